"""diagnostic (variants/wide_stamps.so only): cycles per phase of decode_block_wide's rounds, one 128 KiB stream"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ldtest as T
from libdeflate_amd import batch as B, synth, _native as N
names = ["reload", "decode", "jump", "ex+B1", "chain", "scan+stops+tk", "B2", "after-B2 reads", "output", "loop"]
s = synth.text_streams(1, 131072)[0].tobytes()
for strat in (None, "huffman_only"):
    c = T.o_compress(s, level=5, strategy=strat)[0]
    for w in ("4", "8"):
        os.environ["LDH_INFLATE_WAVES"] = w
        B.decompress_batch([c], max_out=131072)
        buf = (C.c_ulonglong * 16)()
        N.lib.ldh_debug_wide_stamps(buf, 1)
        B.decompress_batch([c], max_out=131072)
        N.lib.ldh_debug_wide_stamps(buf, 1)
        rounds = max(buf[15], 1)
        print("strategy %s W=%s rounds %d: " % (strat, w, rounds) + ", ".join("%s %.0f" % (n, buf[i] / rounds) for i, n in enumerate(names)),
              " total/round %.0f" % (sum(buf[:10]) / rounds))
