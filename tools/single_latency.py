#!/usr/bin/env python3
"""Latency of single-string calls through the host API (the reference's own use: one string at a time)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libdeflate_amd import LibDeflate as LD, synth
for n in (4096, 131072):
    s = synth.text_streams(1, n, first_stream=5)[0].tobytes()
    for level in (1, 5, 8):
        LD.CompressDeflate(s, {"level": level})
        t0 = time.perf_counter()
        for _ in range(20):
            c, _ = LD.CompressDeflate(s, {"level": level})
        tc = (time.perf_counter() - t0) / 20
        LD.DecompressDeflate(c)
        t0 = time.perf_counter()
        for _ in range(20):
            d, _ = LD.DecompressDeflate(c)
        td = (time.perf_counter() - t0) / 20
        assert d == s
        print("n=%6d level %d: compress %.2f ms, decompress %.2f ms" % (n, level, tc * 1e3, td * 1e3), flush=True)
