#!/usr/bin/env python3
"""k_inflate alone, device-resident: kernel time by batch size and by wavefronts per stream (LDH_INFLATE_WAVES = 1 / 4 / 8,
default: the library's own choice).  Level-5 streams of G-text; every result compared with the input on the device.
usage (GPU box): python tools/inflate_rate.py [counts...]     e.g. 1 16 256 2048 8192;  `long` adds one 16 MiB stream"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench as BN
from libdeflate_amd import _native as N, batch as B
N.init(0)
dev = torch.device("cuda", 0)
args = sys.argv[1:] or ["1", "16", "256", "2048", "8192", "long"]


STRATEGY = os.environ.get("RATE_STRATEGY")  # e.g. huffman_only: literals alone (no match is ever copied)


def run(S, L, label):
    data = BN.gen_text_device(torch, S, L, dev, BN.SEED)
    off = torch.arange(S + 1, dtype=torch.int64, device=dev) * L
    stride = N.lib.ldh_compress_bound(L)
    comp = torch.empty(S * stride, dtype=torch.uint8, device=dev)
    comp_len = torch.zeros(S, dtype=torch.int32, device=dev)
    pad = torch.zeros(S, dtype=torch.uint8, device=dev)
    back = torch.empty(S * L, dtype=torch.uint8, device=dev)
    back_len = torch.zeros(S, dtype=torch.int32, device=dev)
    left = torch.zeros(S, dtype=torch.int32, device=dev)
    status = torch.zeros(S, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    B.device_compress(B.DeviceBatch(data, off, L), comp, stride, comp_len, pad, level=5, strategy=STRATEGY)
    B.sync()
    out = []
    for waves in ("", "1", "2", "4", "8"):
        if waves:
            os.environ["LDH_INFLATE_WAVES"] = waves
        else:
            os.environ.pop("LDH_INFLATE_WAVES", None)
        back.zero_()
        torch.cuda.synchronize()
        reps = 3 if S * L > (64 << 20) else 10
        B.device_decompress_slots(comp, stride, comp_len, S, back, L, back_len, left, status)
        B.sync()
        B.profile(True)
        B.profile_reset()
        for _ in range(reps):
            B.device_decompress_slots(comp, stride, comp_len, S, back, L, back_len, left, status)
        B.sync()
        ms = B.profile_report()["k_inflate"][0] / reps
        B.profile(False)
        ok = bool(torch.equal(back, data)) and int(status.abs().sum()) == 0
        out.append("%s %.3f ms %.2f GB/s%s" % (waves or "auto", ms, S * L / 1e6 / ms, "" if ok else " WRONG"))
    os.environ.pop("LDH_INFLATE_WAVES", None)
    print("%-22s %s" % (label, " | ".join(out)), flush=True)


for a in args:
    if a == "long":
        run(1, 16 << 20, "1 x 16 MiB")
    else:
        run(int(a), 131072, "%d x 128 KiB" % int(a))
