"""Test helpers: ctypes binding of the CPU oracle (oracle/libld_oracle.so), fixtures, generators.

The oracle is the checker only -- product code never imports this module.
"""
import ctypes
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

STRAT = {None: 0, "default": 0, "fixed": 1, "huffman_only": 2, "dynamic": 3}

_c = ctypes
_L = None


def oracle_lib():
    global _L
    if _L is None:
        L = _c.CDLL(os.path.join(ROOT, "oracle", "libld_oracle.so"))
        L.ldo_adler32.restype = _c.c_uint32
        L.ldo_adler32.argtypes = [_c.c_char_p, _c.c_size_t]
        L.ldo_compress_bound.restype = _c.c_size_t
        L.ldo_compress_bound.argtypes = [_c.c_size_t]
        L.ldo_compress.argtypes = [_c.c_char_p, _c.c_size_t, _c.c_int, _c.c_int, _c.c_int, _c.c_char_p,
                                   _c.c_size_t, _c.c_char_p, _c.c_size_t, _c.POINTER(_c.c_size_t),
                                   _c.POINTER(_c.c_int)]
        L.ldo_decompress.argtypes = [_c.c_char_p, _c.c_size_t, _c.c_int, _c.c_char_p, _c.c_size_t,
                                     _c.c_char_p, _c.c_size_t, _c.POINTER(_c.c_size_t),
                                     _c.POINTER(_c.c_size_t)]
        L.ldo_print6_encode.restype = _c.c_size_t
        L.ldo_print6_encode.argtypes = [_c.c_char_p, _c.c_size_t, _c.c_char_p]
        L.ldo_print6_decode.argtypes = [_c.c_char_p, _c.c_size_t, _c.c_char_p, _c.POINTER(_c.c_size_t)]
        L.ldo_tokens.restype = _c.c_long
        L.ldo_tokens.argtypes = [_c.c_char_p, _c.c_size_t, _c.c_int, _c.c_char_p, _c.c_size_t,
                                 _c.c_void_p, _c.c_void_p, _c.c_long]
        L.ldo_codec_build.argtypes = [_c.c_char_p, _c.c_size_t, _c.c_char_p, _c.c_size_t, _c.c_char_p, _c.c_size_t,
                                      _c.POINTER(OCodec), _c.POINTER(_c.c_char_p)]
        L.ldo_codec_encode.restype = _c.c_size_t
        L.ldo_codec_encode.argtypes = [_c.POINTER(OCodec), _c.c_char_p, _c.c_size_t, _c.c_char_p]
        L.ldo_codec_decode.argtypes = [_c.POINTER(OCodec), _c.c_char_p, _c.c_size_t, _c.c_char_p,
                                       _c.POINTER(_c.c_size_t)]
        _L = L
    return _L


class OCodec(_c.Structure):  # ldo_codec of oracle/ld_oracle.h
    _fields_ = [("enc_len", _c.c_uint8 * 256), ("enc0", _c.c_uint8 * 256), ("enc1", _c.c_uint8 * 256),
                ("reserved", _c.c_uint8 * 256), ("has_map", _c.c_int), ("map_back", _c.c_uint8 * 256),
                ("npass", _c.c_int), ("pass_escape", _c.c_uint8 * 256), ("pass_dec", (_c.c_uint16 * 256) * 256)]


def o_codec_build(reserved, escape, mapped):
    """CreateCodec of the oracle: (codec, None) or (None, message)."""
    c = OCodec()
    err = _c.c_char_p()
    rc = oracle_lib().ldo_codec_build(reserved, len(reserved), escape, len(escape), mapped, len(mapped),
                                      _c.byref(c), _c.byref(err))
    return (c, None) if rc == 0 else (None, err.value.decode())


def o_codec_encode(c, s):
    out = _c.create_string_buffer(2 * len(s) + 8)
    n = oracle_lib().ldo_codec_encode(_c.byref(c), s, len(s), out)
    return out.raw[:n]


def o_codec_decode(c, s):
    out = _c.create_string_buffer(len(s) + 8)
    ol = _c.c_size_t()
    rc = oracle_lib().ldo_codec_decode(_c.byref(c), s, len(s), out, _c.byref(ol))
    return None if rc != 0 else out.raw[: ol.value]


ADDON_CODEC = (b"\000", b"\001", b"")                                                   # LD:3060-3062
CHAT_CODEC = (b"sS\000\012\015\174%" + bytes(range(128, 256)), b"\035\037", b"\017\024")  # LD:3118-3124


def o_adler32(s):
    return oracle_lib().ldo_adler32(s, len(s))


def o_compress(s, zlib=False, level=None, strategy=None, dict=None):
    L = oracle_lib()
    cap = L.ldo_compress_bound(len(s))
    out = _c.create_string_buffer(cap)
    ol, pb = _c.c_size_t(), _c.c_int()
    rc = L.ldo_compress(s, len(s), int(zlib), -1 if level is None else level, STRAT[strategy], dict,
                        len(dict) if dict else 0, out, cap, _c.byref(ol), _c.byref(pb))
    assert rc == 0, rc
    return out.raw[: ol.value], pb.value


def o_decompress(s, zlib=False, dict=None, cap=None):
    """Returns (bytes|None, second) like LibDeflate: (data, bytes_left) or (None, status)."""
    L = oracle_lib()
    if cap is None:
        ol, bl = _c.c_size_t(), _c.c_size_t()
        st = L.ldo_decompress(s, len(s), int(zlib), dict, len(dict) if dict else 0, None, 0,
                              _c.byref(ol), _c.byref(bl))
        if st != 0:
            return None, st
        cap = ol.value
    out = _c.create_string_buffer(max(cap, 1))
    ol, bl = _c.c_size_t(), _c.c_size_t()
    st = L.ldo_decompress(s, len(s), int(zlib), dict, len(dict) if dict else 0, out, cap, _c.byref(ol),
                          _c.byref(bl))
    if st != 0:
        return None, st
    return out.raw[: ol.value], bl.value


def o_print6_encode(s):
    out = _c.create_string_buffer(len(s) * 4 // 3 + 8)
    n = oracle_lib().ldo_print6_encode(s, len(s), out)
    return out.raw[:n]


def o_print6_decode(s):
    out = _c.create_string_buffer(len(s) + 8)
    ol = _c.c_size_t()
    rc = oracle_lib().ldo_print6_decode(s, len(s), out, _c.byref(ol))
    return None if rc != 0 else out.raw[: ol.value]


def o_tokens(s, level, dict=None):
    cap = len(s) + 16
    tl = np.zeros(cap, dtype=np.uint16)
    td = np.zeros(cap, dtype=np.uint16)
    n = oracle_lib().ldo_tokens(s, len(s), level, dict, len(dict) if dict else 0, tl.ctypes.data,
                                td.ctypes.data, cap)
    assert n >= 0
    return tl[:n].copy(), td[:n].copy()


# ---------------------------------------------------------------- fixtures

def golden(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()


_RLE = re.compile(r"<([0-9a-f]+)\*(\d+)>")


def kat_hex(s):
    """Expand the <hex*count> run-length notation of kat.json and return bytes."""
    if s is None:
        return None
    return bytes.fromhex(_RLE.sub(lambda m: m.group(1) * int(m.group(2)), s))


def kat():
    with open(os.path.join(GOLDEN, "kat.json")) as f:
        return json.load(f)


# ---------------------------------------------------------------- synthetic inputs

def splitmix64(seed, n):
    """n uint64 values of splitmix64 from `seed` (vectorised)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


# ---------------------------------------------------------------- crafted streams

class BitWriter(object):
    """LSB-first bit packer (RFC 1951 order) for hand-built test streams."""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def bits(self, value, nbits):
        self.acc |= (value & ((1 << nbits) - 1)) << self.n
        self.n += nbits
        while self.n >= 8:
            self.out.append(self.acc & 255)
            self.acc >>= 8
            self.n -= 8

    def done(self):
        if self.n:
            self.out.append(self.acc & 255)
            self.acc, self.n = 0, 0
        return bytes(self.out)


def truncated_dynamic_header_stream():
    """A dynamic block whose header is cut off inside its code lengths (k_inflate): code-length code =
    {0: 1 bit '0', 8: 1 bit '1'}; literal/length lengths 8 x255, 0, 8 (symbols 0..254 and 256: a complete code); then
    the stream ends.  The 29 remaining length codes and the 30 distance lengths decode from the zero bits past the end
    (neither the reference nor puff's order of checks looks at the bits left inside the header, LD:2506-2615), so the
    block body starts some 60 bits beyond the input: the reference decodes literal 0 from eight zero bits and then
    reports 2 (LD:2428)."""
    w = BitWriter()
    w.bits(1, 1)    # BFINAL
    w.bits(2, 2)    # dynamic
    w.bits(29, 5)   # HLIT: 286 literal/length codes
    w.bits(29, 5)   # HDIST: 30 distance codes
    order = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]
    w.bits(1, 4)    # HCLEN: 5 code-length code lengths (16, 17, 18, 0, 8)
    for sym in order[:5]:
        w.bits(1 if sym in (0, 8) else 0, 3)
    for _ in range(255):
        w.bits(1, 1)   # length 8
    w.bits(0, 1)       # symbol 255: length 0
    w.bits(1, 1)       # symbol 256: length 8
    return w.done()


def unique_chars_string(rng, n):
    """GetRandomStringUniqueChars of tests/Test.lua:225-247: n <= 256 distinct bytes in random order; beyond 256, random
    bytes inserted at random places of a permutation of all 256."""
    perm = [int(x) for x in rng.permutation(256)]
    if n <= 256:
        return bytes(perm[:n])
    for _ in range(n - 256):
        perm.insert(int(rng.integers(0, len(perm) + 1)), int(rng.integers(0, 256)))
    return bytes(perm)


def dict_bound_cases(trials=10):
    """(dictionary, string, bound on the raw-deflate size at level 7) of tests/Test.lua:1893-1957."""
    rng = np.random.default_rng(1893)
    out = []
    for _ in range(trials):
        s = unique_chars_string(rng, 3)
        out.append((s, s, 4))                          # TestLength3String1
        s = unique_chars_string(rng, 3)
        out.append((s[:2], s[2:3] + s, 5))             # TestLength3String2: the last two dictionary bytes are hashed
        s = unique_chars_string(rng, 3)
        out.append((s[:1], s[1:3] + s, 6))             # TestLength3String3
        for n, bound in ((257, 5), (258, 4), (259, 5)):
            s = unique_chars_string(rng, n)
            out.append((s, s, bound))
    return out
