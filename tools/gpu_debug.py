#!/usr/bin/env python3
"""Stage-by-stage parity probe (GPU box): links -> tokens -> bytes, against the oracle."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ldtest as T
from libdeflate_amd import LibDeflate, batch
from libdeflate_amd import _native as N

def ref_prevd(V):
    last = {}
    out = np.zeros(len(V), dtype=np.uint16)
    Vp = V + b"\0\0\0"
    for v in range(len(V)):
        k = Vp[v:v+3]
        u = last.get(k)
        if u is not None and v - u <= 32768:
            out[v] = v - u
        last[k] = v
    return out

def gpu_tokens(s, level, nblocks_unused=None):
    n = len(s)
    raw = batch.debug_fetch(2, 0, 4 * (n + 64)).view(np.uint32)
    blk = batch.debug_fetch(3, 0, 1 << 20)
    return raw, blk

def check(s, level, dct=None, fmt=0, strategy=None, verbose=True):
    d = None
    if dct is not None:
        d = LibDeflate.CreateDictionary(dct, len(dct), T.o_adler32(dct))
    res = batch.compress_batch([s], fmt=fmt, level=level, strategy=strategy, dictionary=d)[0]
    want = T.o_compress(s, zlib=bool(fmt), level=level, strategy=strategy, dict=dct)
    ok = (res == want)
    if ok or not verbose:
        return ok
    print("MISMATCH level", level, "len", len(s), "dict", None if dct is None else len(dct), "fmt", fmt, "strategy", strategy,
          "got", len(res[0]), res[1], "want", len(want[0]), want[1])
    a, b = res[0], want[0]
    m = min(len(a), len(b))
    diff = [i for i in range(m) if a[i] != b[i]]
    print("  first diff byte", diff[0] if diff else m, "ndiff", len(diff))
    if level and strategy != "huffman_only":
        D = len(dct) if dct else 0
        V = (dct or b"") + s
        pd = batch.debug_fetch(0, 0, 2 * (len(V) + 64)).view(np.uint16)[: len(V)]
        rp = ref_prevd(V)
        bad = np.nonzero(pd != rp)[0]
        print("  prevd mismatches:", len(bad), bad[:10], pd[bad[:10]], rp[bad[:10]])
        tl, td = T.o_tokens(s, level, dct)
        raw, _ = gpu_tokens(s, level)
        # rebuild GPU token list block by block
        nblk = 1 if len(s) <= 65535 else 1 + (len(s) - 65535 + 32767) // 32768
        gl, gd = [], []
        import ctypes
        bi_size = None
        for b in range(nblk):
            bs = 1 if b == 0 else 65536 + (b - 1) * 32768
            be = min(len(s), 65535 if b == 0 else 65535 + b * 32768)
            # ntok unknown on host without struct size; walk until covered length == block length
            cov = 0; i = bs - 1
            while cov < be - bs + 1:
                t = int(raw[i]); i += 1
                if t & 0x80000000:
                    L = ((t >> 16) & 0xFF) + 3; Dd = (t & 0xFFFF) + 1
                    gl.append(L); gd.append(Dd); cov += L
                else:
                    gl.append(0); gd.append(t); cov += 1
        gl = np.array(gl, dtype=np.uint16); gd = np.array(gd, dtype=np.uint16)
        m2 = min(len(gl), len(tl))
        bt = np.nonzero((gl[:m2] != tl[:m2]) | (gd[:m2] != td[:m2]))[0]
        print("  tokens: gpu", len(gl), "oracle", len(tl), "first mismatch idx", bt[:1], 
              "gpu", (gl[bt[:3]], gd[bt[:3]]), "oracle", (tl[bt[:3]], td[bt[:3]]))
        if len(bt):
            pos = int(np.sum(np.where(tl[:bt[0]] == 0, 1, tl[:bt[0]])))
            print("  stream position of first token mismatch:", pos + 1)
    return ok

def main():
    N.init()
    rng = np.random.default_rng(1)
    G = T.golden
    cases = [b"", b"a", b"abc", b"a" * 1000, bytes(range(256)) + bytes(range(255, -1, -1)), G("item_strings.txt"),
             G("smalltest_no_newline.txt"), bytes(rng.integers(97, 101, 5000, dtype=np.uint8)),
             G("64kminus1.txt"), G("64k.txt"), G("64kplus1.txt"), G("encode_6bit_weakaura.txt"), G("totalrp3.txt")]
    fails = 0
    t0 = time.time()
    for lvl in (5, 7, 8, 4, 6, 0, 1, 2, 3, 9):
        for s in cases:
            if lvl == 9 and len(s) > 70000: continue
            ok = check(s, lvl)
            fails += (not ok)
            if fails > 6: print("too many failures"); return 1
        print("level", lvl, "done, fails so far", fails, "t=%.1f" % (time.time() - t0), flush=True)
    for st in ("fixed", "dynamic", "huffman_only"):
        for s in cases[:8]:
            for lvl in (0, 1, 5):
                fails += (not check(s, lvl, strategy=st))
    print("strategies done, fails", fails, flush=True)
    for s in cases[:8]:
        for lvl in (1, 5):
            fails += (not check(s, lvl, fmt=1))
    print("zlib done, fails", fails, flush=True)
    dct = G("dictionary32768.txt")
    for dl in (1, 2, 3, 100, 258, 259, 32768):
        for s in (b"", G("item_strings.txt"), dct[-dl:], cases[7][:700]):
            for lvl in (1, 3, 5, 7):
                for fmt in (0, 1):
                    fails += (not check(s, lvl, dct=dct[-dl:], fmt=fmt))
        print("dict", dl, "fails", fails, flush=True)
    print("TOTAL FAILS", fails)
    return 1 if fails else 0

if __name__ == "__main__":
    sys.exit(main())
