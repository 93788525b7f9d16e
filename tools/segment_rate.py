#!/usr/bin/env python3
"""Device-resident rates of the two shapes whose virtual buffer is sorted in segments:
BASELINE configs[3] (128 KiB streams behind the 32 KiB preset dictionary, zlib) and one long stream
(SURVEY 8.f N4).  Prints kernel times from the library's HIP events; outputs are verified by round trip."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libdeflate_amd import LibDeflate as LD, batch as B, _native as N

dev = torch.device("cuda", 0)
N.init(0)


def run(tag, data, off, L, fmt, level, dictionary, check_inflate=True):
    S = off.numel() - 1
    bt = B.DeviceBatch(data, off, L)
    stride = N.lib.ldh_compress_bound(L)
    comp = torch.empty(S * stride, dtype=torch.uint8, device=dev)
    comp_len = torch.zeros(S, dtype=torch.int32, device=dev)
    pad = torch.zeros(S, dtype=torch.uint8, device=dev)
    back = torch.empty(S * L, dtype=torch.uint8, device=dev)
    back_len = torch.zeros(S, dtype=torch.int32, device=dev)
    left = torch.zeros(S, dtype=torch.int32, device=dev)
    status = torch.zeros(S, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()  # the library runs on a stream of its own: torch's fills and copies must have landed
    for rep in range(2):
        B.profile(True)
        B.profile_reset()
        B.device_compress(bt, comp, stride, comp_len, pad, fmt=fmt, level=level, dictionary=dictionary)
        if check_inflate:
            B.device_decompress_slots(comp, stride, comp_len, S, back, L, back_len, left, status, fmt=fmt, dictionary=dictionary)
        B.sync()
        pr = B.profile_report()
        B.profile(False)
    tc = sum(v[0] for k, v in pr.items() if k not in ("k_inflate", "k_zlib_check")) - (pr.get("k_adler32", (0, 0))[0] / 2 if fmt == 1 and check_inflate else 0)
    td = pr.get("k_inflate", (0, 0))[0]
    ok = (not check_inflate) or (bool(torch.equal(back, data)) and int(status.abs().sum()) == 0)
    nbytes = S * L
    print("%s: %d x %d B level %d: compress %.2f ms = %.2f GB/s, inflate %.2f ms, ratio %.3f, round trip ok=%s" %
          (tag, S, L, level, tc, nbytes / 1e6 / tc, td, nbytes / max(int(comp_len.sum().item()), 1), ok), flush=True)
    print("   ", {k: round(v[0], 2) for k, v in sorted(pr.items())}, flush=True)


S, L = 4096, 131072
data = bench.gen_text_device(torch, S, L, dev, bench.SEED)
off = torch.arange(S + 1, dtype=torch.int64, device=dev) * L
dct = bench.gen_text_device(torch, 1, 32768, dev, bench.SEED ^ 0x5555).cpu().numpy().tobytes()
d = LD.CreateDictionary(dct, len(dct), LD.Adler32(dct))
run("no dictionary, zlib", data, off, L, 1, 5, None)
run("32 KiB dictionary, zlib (configs[3])", data, off, L, 1, 5, d)
del data
for mib in (16, 256):
    L1 = mib << 20
    one = bench.gen_text_device(torch, 1, L1, dev, bench.SEED ^ 0x77)
    off1 = torch.tensor([0, L1], dtype=torch.int64, device=dev)
    run("one long stream", one, off1, L1, 0, 5, None, check_inflate=(mib <= 16))
    del one
