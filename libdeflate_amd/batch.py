"""Batched entry points (the north star's "batched entry point"; no counterpart in the
reference, which handles one string per call).

Host form: lists of bytes in, lists of results out (one H2D / D2H round trip).
Device form (`Device*`): torch uint8 tensors resident in HBM; nothing leaves the GPU.
torch is plumbing here (device memory); all computing is done by libdeflate_hip.so.
"""
import ctypes as C

import numpy as np

from . import _native as N
from .LibDeflate import _STRATEGY, MAX_OUT


def _concat(strings):
    lens = np.fromiter((len(s) for s in strings), dtype=np.uint64, count=len(strings))
    off = np.zeros(len(strings) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    data = np.frombuffer(b"".join(strings) + b"\0", dtype=np.uint8)
    return data, off


def compress_batch(strings, fmt=0, level=None, strategy=None, dictionary=None):
    """Returns [(compressed_bytes, pad_bits)] for LibDeflate:Compress{Deflate,Zlib}[WithDict]."""
    N.init()
    n = len(strings)
    if n == 0:
        return []
    data, off = _concat(strings)
    stride = N.lib.ldh_compress_bound(max(len(s) for s in strings))
    out = np.empty(n * stride, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint32)
    pad = np.zeros(n, dtype=np.uint8)
    N.check(N.lib.ldh_compress_batch(data.ctypes.data, off.ctypes.data, n, fmt, -1 if level is None else level,
                                     _STRATEGY[strategy], dictionary.handle() if dictionary is not None else None,
                                     out.ctypes.data, stride, out_len.ctypes.data, pad.ctypes.data))
    return [(out[i * stride: i * stride + int(out_len[i])].tobytes(), int(pad[i])) for i in range(n)]


def decompress_batch(strings, fmt=0, dictionary=None, max_out=None):
    """Returns [(bytes, bytes_left)] or [(None, status)] per stream.  `max_out` is a hard limit on one result's
    size (a stream that needs more gives (None, 1) = LDH_STATUS_OUTPUT_FULL); without it the slots grow (x4) until
    every stream fits, up to the library's 4 GiB limit."""
    N.init()
    n = len(strings)
    if n == 0:
        return []
    data, off = _concat(strings)
    cap = max_out if max_out is not None else max(64, 8 * max(len(s) for s in strings) + 64)
    pending = list(range(n))
    results = [None] * n
    h = dictionary.handle() if dictionary is not None else None
    while pending:
        stride = (cap + 15) // 16 * 16
        sub = [strings[i] for i in pending]
        d, o = (data, off) if len(pending) == n else _concat(sub)
        m = len(pending)
        out = np.empty(m * stride, dtype=np.uint8)
        out_len = np.zeros(m, dtype=np.uint32)
        left = np.zeros(m, dtype=np.uint32)
        status = np.zeros(m, dtype=np.int32)
        N.check(N.lib.ldh_decompress_batch(d.ctypes.data, o.ctypes.data, m, fmt, h, out.ctypes.data, stride,
                                           out_len.ctypes.data, left.ctypes.data, status.ctypes.data))
        again = []
        for k, i in enumerate(pending):
            st = int(status[k])
            if st == 1 and max_out is None and cap < MAX_OUT:
                again.append(i)
            elif st != 0:
                results[i] = (None, st)
            else:
                results[i] = (out[k * stride: k * stride + int(out_len[k])].tobytes(), int(left[k]))
        pending = again
        cap = min(cap * 4, MAX_OUT)
    return results


def adler32_batch(strings):
    N.init()
    n = len(strings)
    if n == 0:
        return []
    data, off = _concat(strings)
    out = np.zeros(n, dtype=np.uint32)
    N.check(N.lib.ldh_adler32_batch(data.ctypes.data, off.ctypes.data, n, out.ctypes.data))
    return [int(x) for x in out]


def print6_encode_batch(strings):
    N.init()
    n = len(strings)
    if n == 0:
        return []
    data, off = _concat(strings)
    stride = ((max(len(s) for s in strings) * 8 + 5) // 6 + 31) // 16 * 16
    out = np.empty(n * stride, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint32)
    N.check(N.lib.ldh_print6_encode_batch(data.ctypes.data, off.ctypes.data, n, out.ctypes.data, stride,
                                          out_len.ctypes.data))
    return [out[i * stride: i * stride + int(out_len[i])].tobytes() for i in range(n)]


def print6_decode_batch(strings):
    N.init()
    n = len(strings)
    if n == 0:
        return []
    data, off = _concat(strings)
    stride = (max(len(s) for s in strings) + 31) // 16 * 16
    out = np.empty(n * stride, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint32)
    ok = np.zeros(n, dtype=np.uint8)
    N.check(N.lib.ldh_print6_decode_batch(data.ctypes.data, off.ctypes.data, n, out.ctypes.data, stride,
                                          out_len.ctypes.data, ok.ctypes.data))
    return [out[i * stride: i * stride + int(out_len[i])].tobytes() if ok[i] else None for i in range(n)]


def codec_encode_batch(codec, strings):
    """codec:Encode for every string (codec: LibDeflate.CreateCodec result)."""
    N.init()
    n = len(strings)
    if n == 0:
        return []
    data, off = _concat(strings)
    stride = (2 * max(len(s) for s in strings) + 31) // 16 * 16
    out = np.empty(n * stride, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint32)
    N.check(N.lib.ldh_codec_encode_batch(codec.handle(), data.ctypes.data, off.ctypes.data, n, out.ctypes.data,
                                         stride, out_len.ctypes.data))
    return [out[i * stride: i * stride + int(out_len[i])].tobytes() for i in range(n)]


def codec_decode_batch(codec, strings):
    """codec:Decode for every string; None stands for nil."""
    N.init()
    n = len(strings)
    if n == 0:
        return []
    data, off = _concat(strings)
    stride = (max(len(s) for s in strings) + 31) // 16 * 16
    out = np.empty(n * stride, dtype=np.uint8)
    out_len = np.zeros(n, dtype=np.uint32)
    ok = np.zeros(n, dtype=np.uint8)
    N.check(N.lib.ldh_codec_decode_batch(codec.handle(), data.ctypes.data, off.ctypes.data, n, out.ctypes.data,
                                         stride, out_len.ctypes.data, ok.ctypes.data))
    return [out[i * stride: i * stride + int(out_len[i])].tobytes() if ok[i] else None for i in range(n)]


class DeviceBatch(object):
    """A batch of streams resident in HBM: `data` (uint8 torch tensor on cuda) holds the
    streams back to back, `off` (int64 tensor on cuda, count+1) their offsets."""

    def __init__(self, data, off, max_len):
        self.data, self.off, self.max_len = data, off, int(max_len)
        self.count = off.numel() - 1


def device_compress(batch, out, out_stride, out_len, pad_bits, fmt=0, level=5, strategy=None, dictionary=None):
    """Enqueue compression of a DeviceBatch into `out` (uint8 cuda tensor, count*out_stride);
    out_len (int32/uint32 cuda tensor), pad_bits (uint8 cuda tensor).  Call sync() before reading."""
    N.init()
    N.check(N.lib.ldh_compress_batch_dev(batch.data.data_ptr(), batch.off.data_ptr(), batch.count, batch.max_len,
                                         fmt, level, _STRATEGY[strategy],
                                         dictionary.handle() if dictionary is not None else None,
                                         out.data_ptr(), out_stride, out_len.data_ptr(), pad_bits.data_ptr()))


def device_decompress(batch, out, out_stride, out_len, bytes_left, status, fmt=0, dictionary=None):
    N.init()
    N.check(N.lib.ldh_decompress_batch_dev(batch.data.data_ptr(), batch.off.data_ptr(), batch.count, fmt,
                                           dictionary.handle() if dictionary is not None else None,
                                           out.data_ptr(), out_stride, out_len.data_ptr(), bytes_left.data_ptr(),
                                           status.data_ptr()))


def device_decompress_slots(data, in_stride, in_len, count, out, out_stride, out_len, bytes_left, status, fmt=0,
                            dictionary=None):
    """Decompress streams left in strided slots by device_compress (no repacking)."""
    N.init()
    N.check(N.lib.ldh_decompress_slots_dev(data.data_ptr(), in_stride, in_len.data_ptr(), count, fmt,
                                           dictionary.handle() if dictionary is not None else None,
                                           out.data_ptr(), out_stride, out_len.data_ptr(), bytes_left.data_ptr(),
                                           status.data_ptr()))


def sync():
    N.check(N.lib.ldh_sync())


def profile(on=True):
    N.check(N.lib.ldh_profile_enable(1 if on else 0))


def profile_reset():
    N.check(N.lib.ldh_profile_reset())


def profile_report():
    """{kernel name: (total ms, launches)} since the last reset."""
    out = {}
    for i in range(N.lib.ldh_profile_count()):
        name = C.create_string_buffer(64)
        ms = C.c_double()
        nl = C.c_uint64()
        N.check(N.lib.ldh_profile_get(i, name, 64, C.byref(ms), C.byref(nl)))
        out[name.value.decode()] = (ms.value, nl.value)
    return out


def debug_fetch(what, stream, nbytes):
    buf = np.zeros(nbytes, dtype=np.uint8)
    got = C.c_size_t()
    N.check(N.lib.ldh_debug_fetch(what, stream, buf.ctypes.data, nbytes, C.byref(got)))
    return buf[: got.value]
