import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import ldtest as T
import gpu_debug as G
from libdeflate_amd import synth, _native as N
N.init()
cases = [T.golden("totalrp3.txt"), T.golden("encode_6bit_weakaura.txt"), synth.text_streams(1, 131072)[0].tobytes(),
         T.golden("64kplus1.txt"), b"abcd" * 50000, bytes(np.random.default_rng(3).integers(97, 100, 100000, dtype=np.uint8))]
for lv in (1, 2):
    for s in cases:
        print("level", lv, "len", len(s), "ok" if G.check(s, lv) else "")
