#!/usr/bin/env python3
"""BASELINE configs[1] alone (65536 x 4 KiB addon-style strings, levels 1 / 5 / 8): kernel times per level.
usage (GPU box): python tools/small_rate.py [count] [length]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ldtest as T
import bench_rows as R
from libdeflate_amd import _native as N
N.init(0)
count = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
length = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
r = R.row_small_strings(torch, torch.device("cuda", 0), 0x4C69624465666C61, count=count, length=length, T=T)
for lv, e in r["levels"].items():
    print("level %s: %.2f GB/s round trip (compress %.2f, decompress %.2f) ratio %.2f ok=%s %s" % (
        lv, e["roundtrip_GBps"], e["compress_GBps"], e["decompress_GBps"], e["compress_ratio"], e["verified"], e["kernel_ms"]))
