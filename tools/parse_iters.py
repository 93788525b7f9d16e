#!/usr/bin/env python3
"""How many relaxation repeats does k_parse_lazy need?  Compares capped runs (LDH_TUNE2=cap) with the uncapped one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libdeflate_amd import batch as B, synth
text = [r.tobytes() for r in synth.text_streams(512, 131072, first_stream=0)]
os.environ.pop("LDH_TUNE2", None)
ref = [c for c, _ in B.compress_batch(text, level=5)]
for cap in (2, 3, 4, 5, 6, 8, 12, 20, 40, 80, 160, 250):
    os.environ["LDH_TUNE2"] = str(cap)
    got = [c for c, _ in B.compress_batch(text, level=5)]
    bad = sum(1 for a, b in zip(ref, got) if a != b)
    print("cap", cap, "streams differing:", bad, "of", len(text), flush=True)
