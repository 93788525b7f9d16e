"""diagnostic (variants/parse_stamps.so only): cycles per phase of k_parse_lazy's tiles, 8192 x 128 KiB at level 5"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as BN
from libdeflate_amd import batch as B, _native as N
N.init(0)
S, L = 8192, 131072
dev = torch.device("cuda", 0)
data = BN.gen_text_device(torch, S, L, dev, BN.SEED)
off = torch.arange(S + 1, dtype=torch.int64, device=dev) * L
bt = B.DeviceBatch(data, off, L)
stride = N.lib.ldh_compress_bound(L)
comp = torch.empty(S * stride, dtype=torch.uint8, device=dev)
comp_len = torch.zeros(S, dtype=torch.int32, device=dev)
pad = torch.zeros(S, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
names = ["staging", "steps", "relaxation", "scan", "emit+barrier", "loop top"]
buf = (C.c_ulonglong * 16)()
for rep in range(2):
    B.device_compress(bt, comp, stride, comp_len, pad, fmt=0, level=5)
    B.sync()
    N.lib.ldh_debug_parse_stamps(buf, 1)
tiles = max(buf[15], 1)
print("tiles %d, repeats per tile %.2f: " % (tiles, buf[14] / tiles) + ", ".join("%s %.0f" % (n, buf[i] / tiles) for i, n in enumerate(names)),
      " total/tile %.0f cycles" % (sum(buf[:8]) / tiles))
