import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ldtest as T
from libdeflate_amd import batch as B, synth
cases = [("item", T.golden("item_strings.txt")), ("text20k", synth.text_streams(1, 20000)[0].tobytes()),
         ("a1000", b"a" * 1000), ("lits", bytes(np.random.default_rng(1).integers(0, 256, 3000, dtype=np.uint8)))]
for name, s in cases:
    for level in (1, 5):
        c = T.o_compress(s, level=level)[0]
        for w in ("1", "4", "8"):
            os.environ["LDH_INFLATE_WAVES"] = w
            d, left = B.decompress_batch([c], max_out=len(s) + 16)[0]
            if d == s:
                print(name, level, "W", w, "ok")
            else:
                if d is None:
                    print(name, level, "W", w, "status", left)
                    continue
                n = min(len(d), len(s))
                k = next((i for i in range(n) if d[i] != s[i]), n)
                print(name, level, "W", w, "MISMATCH at", k, "of", len(s), "got len", len(d), d[max(0,k-8):k+24], s[max(0,k-8):k+24])
