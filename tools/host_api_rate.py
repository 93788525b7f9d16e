#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry points (ldh_compress_batch / ldh_decompress_batch), one synchronous
call per batch: inputs and outputs in ordinary (pageable) host memory, then in pinned memory from ldh_host_alloc
(copied from and to directly, no staging).  LDH_HOST_TRACE=1 adds the library's own per-stage host times on stderr.
usage (GPU box): python tools/host_api_rate.py [streams]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from libdeflate_amd import _native as N, synth

S, L = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 131072
N.init()
data = np.ascontiguousarray(np.concatenate([r for r in synth.text_streams(64, L, first_stream=0)] * (S // 64)))
off = (np.arange(S + 1, dtype=np.uint64) * L)
stride = N.lib.ldh_compress_bound(L)
comp = np.empty(S * stride, dtype=np.uint8)
clen = np.zeros(S, dtype=np.uint32)
pad = np.zeros(S, dtype=np.uint8)


def best(fn, reps=3):
    fn()
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t = min(t, time.perf_counter() - t0)
    return t


def compress():
    N.check(N.lib.ldh_compress_batch(data.ctypes.data, off.ctypes.data, S, 0, 5, 0, None, comp.ctypes.data, stride,
                                     clen.ctypes.data, pad.ctypes.data))


def run(label):
    global packed, coff, back
    tc = best(compress)
    # pack the compressed streams back to back, as a caller would hold them
    coff = np.zeros(S + 1, dtype=np.uint64); coff[1:] = np.cumsum(clen)
    pk = np.concatenate([comp[i * stride: i * stride + int(clen[i])] for i in range(S)])
    packed[: pk.size] = pk
    blen = np.zeros(S, dtype=np.uint32); left = np.zeros(S, dtype=np.uint32); st = np.zeros(S, dtype=np.int32)

    def decompress():
        N.check(N.lib.ldh_decompress_batch(packed.ctypes.data, coff.ctypes.data, S, 0, None, back.ctypes.data, L,
                                           blen.ctypes.data, left.ctypes.data, st.ctypes.data))
    td = best(decompress)
    assert (st == 0).all() and np.array_equal(back, data)
    print("host API, %d x %d B, level 5, %s: compress %.2f GB/s, decompress %.2f GB/s, round trip %.2f GB/s "
          "(input bytes; H2D + kernels + D2H)" % (S, L, label, S * L / tc / 1e9, S * L / td / 1e9, S * L / (tc + td) / 1e9), flush=True)


packed = np.empty(S * L // 2, dtype=np.uint8)
back = np.empty(S * L, dtype=np.uint8)
run("pageable memory (staged)")


def pinned(nbytes):  # ldh_host_alloc'ed memory as a numpy array
    p = N.lib.ldh_host_alloc(nbytes)
    assert p
    return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(nbytes,))


d2 = pinned(data.nbytes)
d2[:] = data
data = d2
comp, packed, back = pinned(comp.nbytes), pinned(packed.nbytes), pinned(back.nbytes)
run("memory from ldh_host_alloc (copied from and to directly)")
