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
