"""`python bench.py --gpus N` with WORLD_SIZE unset must itself start N rank processes (one per GPU) and print ONE
line with n_gpus == N.  Checked on CPU with the rendezvous-only mode (gloo): same launcher, same barrier /
MAX-over-ranks protocol, no GPU work."""
import json
import os
import subprocess
import sys

from ldtest import ROOT


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=300)


def test_gpus_flag_starts_that_many_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--rendezvous-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["verified"] is True


def test_gpus_flag_must_agree_with_world_size():
    r = _run(["--gpus", "4", "--rendezvous-only"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE is 2" in r.stderr


def test_failing_rank_fails_the_launcher():
    from libdeflate_amd import shard
    rc = shard.launch_ranks([sys.executable, "-c", "import os,sys; sys.exit(7 if os.environ['RANK']=='1' else 0)"], 2)
    assert rc == 7


def test_launcher_stops_the_siblings_of_a_rank_that_died():
    # VERDICT r2 item 7 / ADVICE r2: rank 1 exits 7 at once while rank 0 sits in a gloo barrier (which, alone, would
    # wait for the distributed timeout); the launcher must return 7 within seconds, having stopped rank 0
    import time
    from libdeflate_amd import shard
    code = ("import os, sys\n"
            "if os.environ['RANK'] == '1': sys.exit(7)\n"
            "import torch.distributed as dist\n"
            "dist.init_process_group('gloo', rank=0, world_size=2)\n"
            "dist.barrier()\n")
    t0 = time.time()
    rc = shard.launch_ranks([sys.executable, "-c", code], 2)
    assert rc == 7 and time.time() - t0 < 60


def test_launcher_timeout_stops_everything():
    import time
    from libdeflate_amd import shard
    t0 = time.time()
    rc = shard.launch_ranks([sys.executable, "-c", "import time; time.sleep(600)"], 2, timeout=1.0, grace=2.0)
    assert rc == 124 and time.time() - t0 < 30


def test_launcher_process_loads_no_gpu_stack():
    # the launcher branch of bench.py runs before anything HIP-related is loaded (shard.py by path, not
    # through the package, whose __init__ loads libdeflate_hip.so)
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[: src.index("shard.launch_ranks(")]
    assert "from libdeflate_amd import" not in head[head.index("def main()"):]
    assert "import torch" not in head[head.index("def main()"):].replace("import torch.distributed", "")
