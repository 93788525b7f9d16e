"""The N>1 path on CPU: two processes, gloo, 127.0.0.1.  Shards partition the job, nothing but the
rendezvous is communicated, the elapsed time is the MAX over ranks and verification is agreed on."""
import json
import os
import socket
import subprocess
import sys

import ldtest as T
from libdeflate_amd import shard, synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_shard_of_partitions():
    for total in (0, 1, 7, 8, 8192, 8193):
        for world in (1, 2, 3, 8):
            spans = [shard.shard_of(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_world_size_2_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    total, length, world = 6, 4096, 2
    out = str(tmp_path / "r")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_shard_worker.py"), out, str(total),
                                       str(length)], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    res = [json.load(open("%s.%d" % (out, r))) for r in range(world)]
    # shards: contiguous, disjoint, complete
    assert [(r["lo"], r["hi"]) for r in res] == [(0, 3), (3, 6)]
    # every rank reports the same elapsed time, and it is the slow rank's (2 steps x 0.1 s sleep)
    assert res[0]["dt"] == res[1]["dt"] and res[0]["dt"] >= 0.2
    assert all(r["ok"] for r in res) and not any(r["ok_if_rank1_fails"] for r in res)
    assert all(r["nbytes"] == total * length for r in res)
    # checksum of checksums: the two shards together are exactly the single-process job
    want = []
    for s in synth.text_streams(total, length, first_stream=0):
        c, _ = T.o_compress(s.tobytes(), level=5)
        want.append(T.o_adler32(c))
    assert res[0]["sums"] + res[1]["sums"] == want
