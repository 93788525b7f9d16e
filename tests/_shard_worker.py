"""Rank body of tests/test_shard_gloo.py: started once per rank with RANK/WORLD_SIZE/MASTER_* set.
The product's compute needs a GPU; here the oracle stands in for it (tests may use the oracle), because
what is under test is the sharding and the rendezvous code of libdeflate_amd/shard.py that bench.py runs."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import ldtest as T  # noqa: E402
from libdeflate_amd import shard, synth  # noqa: E402

out_path, total_streams, length = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
rank, _, world = shard.env_world()
dist = shard.init("gloo")
lo, hi = shard.shard_of(total_streams, rank, world)
rows = [r.tobytes() for r in synth.text_streams(hi - lo, length, first_stream=lo)]
sums = []


def step():
    sums.clear()
    for s in rows:
        c, _ = T.o_compress(s, level=5)
        d, left = T.o_decompress(c, cap=len(s))
        assert d == s and left == 0
        sums.append(T.o_adler32(c))
    time.sleep(0.05 * (rank + 1))  # rank 1 is the slow one: the reported time must be its time


dt = shard.timed_steps(step, 2, 1, lambda: None, dist, torch, "cpu")
ok = shard.all_true(True, dist, torch, "cpu")
nbytes = shard.total(float(sum(len(s) for s in rows)), dist, torch, "cpu")
# a rank that fails verification must turn the job's flag off for everybody
ok_if_rank1_fails = shard.all_true(rank != 1, dist, torch, "cpu")
with open("%s.%d" % (out_path, rank), "w") as f:
    json.dump({"rank": rank, "lo": lo, "hi": hi, "dt": dt, "ok": ok, "nbytes": nbytes,
               "ok_if_rank1_fails": ok_if_rank1_fails, "sums": sums}, f)
if dist is not None:
    dist.barrier()
    dist.destroy_process_group()
