#!/usr/bin/env python3
"""tests/test_gpu_round3.py::test_inflate_of_short_distance_matches at full occupancy (4096 / 8192 streams: eight
wavefronts a SIMD): matches that read the bytes of the round before, which the one-wavefront inflate stores a round late.
usage (GPU box): python tools/stress_inflate_short.py [counts...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_round3 as R
from libdeflate_amd import batch as B
for count in [int(a) for a in sys.argv[1:]] or [4096, 8192, 8192]:
    R.test_inflate_of_short_distance_matches(B, count)
    print("ok: %d streams of short periods, levels 5 and 1, round trip exact" % count, flush=True)
