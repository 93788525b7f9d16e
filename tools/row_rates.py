#!/usr/bin/env python3
"""Device-resident kernel rates of the hot-path rows that are not on the bench line (SURVEY 8.a a12, a18, N2 and the
4 KiB shape of BASELINE configs[1]): Adler-32, the 6-bit print codec, the WoW addon codec, 65536 x 4096 B streams."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from libdeflate_amd import LibDeflate as LD, batch as B, _native as N

dev = torch.device("cuda", 0)
N.init(0)


def prof(fn, reps=3):
    torch.cuda.synchronize()  # the library runs on a stream of its own: torch's fills and copies must have landed
    for _ in range(reps):
        B.profile(True)
        B.profile_reset()
        fn()
        B.sync()
        pr = B.profile_report()
        B.profile(False)
    return {k: round(v[0], 3) for k, v in sorted(pr.items())}


S, L = 4096, 131072
data = bench.gen_text_device(torch, S, L, dev, bench.SEED)
off = torch.arange(S + 1, dtype=torch.int64, device=dev) * L
nbytes = S * L
ad = torch.zeros(S, dtype=torch.int32, device=dev)
pr = prof(lambda: N.check(N.lib.ldh_adler32_batch_dev(data.data_ptr(), off.data_ptr(), S, ad.data_ptr())))
print("adler32: %d x %d B:" % (S, L), pr, "-> %.0f GB/s" % (nbytes / 1e6 / pr["k_adler32"]), flush=True)

stride = ((L * 4 + 2) // 3 + 16 + 15) // 16 * 16
enc = torch.empty(S * stride, dtype=torch.uint8, device=dev)
enc_len = torch.zeros(S, dtype=torch.int32, device=dev)
pr = prof(lambda: N.check(N.lib.ldh_print6_encode_batch_dev(data.data_ptr(), off.data_ptr(), S, enc.data_ptr(), stride, enc_len.data_ptr())))
t = sum(pr.values())
print("print6 encode:", pr, "-> %.0f GB/s of input, %.0f GB/s read+written" % (nbytes / 1e6 / t, nbytes * 7 / 3 / 1e6 / t), flush=True)
elen = int(enc_len[0].item())
packed = enc.view(S, stride)[:, :elen].contiguous().view(-1)
off2 = torch.arange(S + 1, dtype=torch.int64, device=dev) * elen
dec = torch.empty(S * L, dtype=torch.uint8, device=dev)
dec_len = torch.zeros(S, dtype=torch.int32, device=dev)
ok = torch.zeros(S, dtype=torch.uint8, device=dev)
pr = prof(lambda: N.check(N.lib.ldh_print6_decode_batch_dev(packed.data_ptr(), off2.data_ptr(), S, dec.data_ptr(), L, dec_len.data_ptr(), ok.data_ptr())))
t = sum(pr.values())
print("print6 decode:", pr, "-> %.0f GB/s of output, %.0f GB/s read+written; exact=%s" %
      (nbytes / 1e6 / t, nbytes * 7 / 3 / 1e6 / t, bool(torch.equal(dec, data)) and int(ok.sum()) == S), flush=True)
del enc, packed, dec

# the WoW addon channel codec on the compressed streams of the same batch (what a WoW caller sends): codec:Encode / codec:Decode
bt0 = B.DeviceBatch(data, off, L)
st0 = N.lib.ldh_compress_bound(L)
comp0 = torch.empty(S * st0, dtype=torch.uint8, device=dev)
clen0 = torch.zeros(S, dtype=torch.int32, device=dev)
pad0 = torch.zeros(S, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
B.device_compress(bt0, comp0, st0, clen0, pad0, fmt=0, level=5)
B.sync()
lens = clen0.to(torch.int64)
coff = torch.zeros(S + 1, dtype=torch.int64, device=dev)
coff[1:] = torch.cumsum(lens, 0)
idx = torch.arange(st0, device=dev)[None, :] < lens[:, None]
cpacked = comp0.view(S, st0)[idx].contiguous()
cbytes = int(coff[-1].item())
codec = LD._addon_codec()
es = (2 * int(lens.max().item()) + 15) // 16 * 16
eout = torch.empty(S * es, dtype=torch.uint8, device=dev)
elen = torch.zeros(S, dtype=torch.int32, device=dev)
pr = prof(lambda: N.check(N.lib.ldh_codec_encode_batch_dev(codec.handle(), cpacked.data_ptr(), coff.data_ptr(), S, eout.data_ptr(), es, elen.data_ptr())))
t = sum(pr.values())
ebytes = int(elen.sum().item())
print("addon codec encode: %d B -> %d B:" % (cbytes, ebytes), pr, "-> %.0f GB/s read+written" % ((cbytes + ebytes) / 1e6 / t), flush=True)
l2 = elen.to(torch.int64)
eoff = torch.zeros(S + 1, dtype=torch.int64, device=dev)
eoff[1:] = torch.cumsum(l2, 0)
epacked = eout.view(S, es)[torch.arange(es, device=dev)[None, :] < l2[:, None]].contiguous()
ds = (int(lens.max().item()) + 16 + 15) // 16 * 16
dout = torch.empty(S * ds, dtype=torch.uint8, device=dev)
dlen = torch.zeros(S, dtype=torch.int32, device=dev)
dok = torch.zeros(S, dtype=torch.uint8, device=dev)
pr = prof(lambda: N.check(N.lib.ldh_codec_decode_batch_dev(codec.handle(), epacked.data_ptr(), eoff.data_ptr(), S, dout.data_ptr(), ds, dlen.data_ptr(), dok.data_ptr())))
t = sum(pr.values())
same = bool(torch.equal(dlen, clen0)) and int(dok.sum()) == S and bool(torch.equal(dout.view(S, ds)[torch.arange(ds, device=dev)[None, :] < lens[:, None]], cpacked))
print("addon codec decode:", pr, "-> %.0f GB/s read+written; exact=%s" % ((cbytes + ebytes) / 1e6 / t, same), flush=True)
del comp0, eout, dout, cpacked, epacked, idx

# BASELINE configs[1] shape: 65536 streams of 4096 bytes
S2, L2 = 65536, 4096
bt = B.DeviceBatch(data[: S2 * L2], torch.arange(S2 + 1, dtype=torch.int64, device=dev) * L2, L2)
st2 = N.lib.ldh_compress_bound(L2)
comp = torch.empty(S2 * st2, dtype=torch.uint8, device=dev)
clen = torch.zeros(S2, dtype=torch.int32, device=dev)
pad = torch.zeros(S2, dtype=torch.uint8, device=dev)
back = torch.empty(S2 * L2, dtype=torch.uint8, device=dev)
blen = torch.zeros(S2, dtype=torch.int32, device=dev)
left = torch.zeros(S2, dtype=torch.int32, device=dev)
status = torch.zeros(S2, dtype=torch.int32, device=dev)
for level in (1, 5, 8):
    def step():
        B.device_compress(bt, comp, st2, clen, pad, fmt=0, level=level)
        B.device_decompress_slots(comp, st2, clen, S2, back, L2, blen, left, status, fmt=0)
    pr = prof(step)
    t = sum(pr.values())
    good = bool(torch.equal(back, data[: S2 * L2])) and int(status.abs().sum()) == 0
    print("65536 x 4096 B level %d: %.2f GB/s round trip, ratio %.2f, ok=%s" % (level, S2 * L2 / 1e6 / t, S2 * L2 / int(clen.sum().item()), good), pr, flush=True)
