"""Host-side mirror of the reference's method table (LibDeflate.lua:2039-2112, 2742-2823,
401, 493, 3311, 3365) over libdeflate_hip.so.

The reference's host language (Lua / LuaJIT) is not present in the build image, so the
operator interface is restated in Python; `lua/LibDeflate.lua` is the LuaJIT-FFI shim a
maintainer would load instead.  Method names, argument order, return
tuples and usage-error strings follow the reference; Lua's `nil` is Python's `None`,
strings are `bytes`.
"""
import ctypes as C

from . import _native as N

_STRATEGY = {None: 0, "fixed": 1, "huffman_only": 2, "dynamic": 3}
MAX_OUT = 0xFFF00000  # k_inflate's clamp on one stream's output slot


def _lua_type(v):
    if v is None:
        return "nil"
    if isinstance(v, (bytes, bytearray, str)):
        return "string"
    if isinstance(v, bool):
        return "boolean"
    if isinstance(v, (int, float)):
        return "number"
    if isinstance(v, (dict, Dictionary)):
        return "table"
    if callable(v):
        return "function"
    return "userdata"


_UNBUILT = object()  # Dictionary.hash_tables before anybody has looked at it


class Dictionary(object):
    """Preset dictionary object (LibDeflate.lua:531-595): same visible fields
    (adler32, strlen, string_table, hash_tables).  The compressor does not read hash_tables -- on the
    device the dictionary's chains are the trigram links that k_sort_links / k_seg_links build over
    dict ++ input -- so the table the reference builds eagerly (LibDeflate.lua:531-577: key = the trigram's
    24-bit value, value = the positions i - strlen, in ascending order) is made here when somebody looks."""

    def __init__(self, data, adler32):
        self.adler32 = adler32
        self.strlen = len(data)
        self.string_table = bytes(data)
        self._hash_tables = _UNBUILT
        self._handle = None

    @property
    def hash_tables(self):
        if self._hash_tables is _UNBUILT:
            t, b, n = {}, self.string_table, len(self.string_table)
            for i in range(n - 2):  # LibDeflate.lua:560-577 (1-based i + 1; the value is (i + 1) - strlen)
                t.setdefault((b[i] << 16) | (b[i + 1] << 8) | b[i + 2], []).append(i + 1 - n)
            self._hash_tables = t
        return self._hash_tables

    @hash_tables.setter
    def hash_tables(self, value):
        self._hash_tables = value

    def handle(self):
        if self._handle is None:
            N.init()
            buf = (C.c_ubyte * len(self.string_table)).from_buffer_copy(self.string_table)
            h = N.lib.ldh_dict_create(buf, len(self.string_table))
            if not h:
                raise N.LdhError((N.lib.ldh_last_error() or b"").decode())
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle:
                N.lib.ldh_dict_free(self._handle)
        except Exception:
            pass


def _is_valid_dictionary(d):  # LibDeflate.lua:602-615
    if not isinstance(d, Dictionary):
        return False, "'dictionary' - table expected got '%s'." % _lua_type(d)
    if (not isinstance(d.adler32, (int, float)) or not isinstance(d.string_table, (bytes, bytearray))
            or not isinstance(d.strlen, int) or d.strlen <= 0 or d.strlen > 32768
            or d.strlen != len(d.string_table)
            or not (d._hash_tables is _UNBUILT or isinstance(d._hash_tables, dict))):
        return False, "'dictionary' - corrupted dictionary."
    return True, ""


def _is_valid_arguments(s, check_dictionary=False, dictionary=None, check_configs=False, configs=None):
    # LibDeflate.lua:679-714
    if not isinstance(s, (bytes, bytearray)):
        return False, "'str' - string expected got '%s'." % _lua_type(s)
    if check_dictionary:
        ok, err = _is_valid_dictionary(dictionary)
        if not ok:
            return False, err
    if check_configs:
        if configs is not None and not isinstance(configs, dict):
            return False, "'configs' - nil or table expected got '%s'." % _lua_type(configs)
        if isinstance(configs, dict):
            for k, v in configs.items():
                if k != "level" and k != "strategy":
                    return False, "'configs' - unsupported table key in the configs: '%s'." % k
                if k == "level" and not (isinstance(v, int) and not isinstance(v, bool) and 0 <= v <= 9):
                    return False, "'configs' - unsupported 'level': %s." % v
                if k == "strategy" and v not in ("fixed", "huffman_only", "dynamic"):
                    return False, "'configs' - unsupported 'strategy': '%s'." % v
    return True, ""


class UsageError(Exception):
    """Lua's error(msg, 2) for argument errors."""


def _buf(b):
    return (C.c_ubyte * max(len(b), 1)).from_buffer_copy(bytes(b) if len(b) else b"\0")


class _LibDeflate(object):
    _VERSION = "1.0.2-release"
    _MAJOR = "LibDeflate"
    _MINOR = 3
    _addon = None  # the two lazily built codec caches (LibDeflate.lua:3058, 3126)
    _chat = None

    # -- LibDeflate.lua:401-442
    def Adler32(self, s):
        if not isinstance(s, (bytes, bytearray)):
            raise UsageError("Usage: LibDeflate:Adler32(str): 'str' - string expected got '%s'." % _lua_type(s))
        N.init()
        out = C.c_uint32()
        N.check(N.lib.ldh_adler32(_buf(s), len(s), C.byref(out)))
        return out.value

    # -- LibDeflate.lua:493-596
    def CreateDictionary(self, s, strlen, adler32):
        pre = "Usage: LibDeflate:CreateDictionary(str, strlen, adler32):"
        if not isinstance(s, (bytes, bytearray)):
            raise UsageError(pre + " 'str' - string expected got '%s'." % _lua_type(s))
        if _lua_type(strlen) != "number":
            raise UsageError(pre + " 'strlen' - number expected got '%s'." % _lua_type(strlen))
        if _lua_type(adler32) != "number":
            raise UsageError(pre + " 'adler32' - number expected got '%s'." % _lua_type(adler32))
        if strlen != len(s):
            raise UsageError(pre + " 'strlen' does not match the actual length of 'str'. 'strlen': %u, '#str': %u ."
                             " Please check if 'str' is modified unintentionally." % (strlen, len(s)))
        if strlen == 0:
            raise UsageError(pre + " 'str' - Empty string is not allowed.")
        if strlen > 32768:
            raise UsageError(pre + " 'str' - string longer than 32768 bytes is not allowed. Got %d bytes." % strlen)
        actual = self.Adler32(s)
        if (adler32 % 4294967296) != (actual % 4294967296):
            raise UsageError(pre + " 'adler32' does not match the actual adler32 of 'str'. 'adler32': %u,"
                             " 'Adler32(str)': %u . Please check if 'str' is modified unintentionally."
                             % (adler32 % 4294967296, actual))
        return Dictionary(s, adler32)

    # -- LibDeflate.lua:1960-2024
    def _compress(self, s, fmt, dictionary, configs):
        N.init()
        level = -1
        strategy = 0
        if configs:
            if configs.get("level") is not None:
                level = configs["level"]
            strategy = _STRATEGY[configs.get("strategy")]
        cap = N.lib.ldh_compress_bound(len(s))
        out = (C.c_ubyte * cap)()
        ol = C.c_size_t()
        pb = C.c_int()
        N.check(N.lib.ldh_compress(_buf(s), len(s), fmt, level, strategy,
                                   dictionary.handle() if dictionary is not None else None, out, cap,
                                   C.byref(ol), C.byref(pb)))
        return bytes(out[: ol.value]), pb.value

    def CompressDeflate(self, s, configs=None):
        ok, err = _is_valid_arguments(s, False, None, True, configs)
        if not ok:
            raise UsageError("Usage: LibDeflate:CompressDeflate(str, configs): " + err)
        return self._compress(s, 0, None, configs)

    def CompressDeflateWithDict(self, s, dictionary, configs=None):
        ok, err = _is_valid_arguments(s, True, dictionary, True, configs)
        if not ok:
            raise UsageError("Usage: LibDeflate:CompressDeflateWithDict(str, dictionary, configs): " + err)
        return self._compress(s, 0, dictionary, configs)

    def CompressZlib(self, s, configs=None):
        ok, err = _is_valid_arguments(s, False, None, True, configs)
        if not ok:
            raise UsageError("Usage: LibDeflate:CompressZlib(str, configs): " + err)
        return self._compress(s, 1, None, configs)

    def CompressZlibWithDict(self, s, dictionary, configs=None):
        ok, err = _is_valid_arguments(s, True, dictionary, True, configs)
        if not ok:
            raise UsageError("Usage: LibDeflate:CompressZlibWithDict(str, dictionary, configs): " + err)
        return self._compress(s, 1, dictionary, configs)

    # -- LibDeflate.lua:2648-2727: (data, bytes_left) or (None, status)
    def _decompress(self, s, fmt, dictionary):
        N.init()
        cap = max(64, 4 * len(s) + 64)
        h = dictionary.handle() if dictionary is not None else None
        while True:
            out = (C.c_ubyte * cap)()
            ol, bl, st = C.c_size_t(), C.c_size_t(), C.c_int()
            N.check(N.lib.ldh_decompress(_buf(s), len(s), fmt, h, out, cap, C.byref(ol), C.byref(bl), C.byref(st)))
            if st.value == 1:  # LDH_STATUS_OUTPUT_FULL: Lua strings simply grow -- up to the library's limit
                if cap >= MAX_OUT:
                    raise N.LdhError("output of 4 GiB or more is not supported")
                cap = min(cap * 4, MAX_OUT)
                continue
            if st.value != 0:
                return None, st.value
            return bytes(out[: ol.value]), bl.value

    def DecompressDeflate(self, s):
        ok, err = _is_valid_arguments(s)
        if not ok:
            raise UsageError("Usage: LibDeflate:DecompressDeflate(str): " + err)
        return self._decompress(s, 0, None)

    def DecompressDeflateWithDict(self, s, dictionary):
        ok, err = _is_valid_arguments(s, True, dictionary)
        if not ok:
            raise UsageError("Usage: LibDeflate:DecompressDeflateWithDict(str, dictionary): " + err)
        return self._decompress(s, 0, dictionary)

    def DecompressZlib(self, s):
        ok, err = _is_valid_arguments(s)
        if not ok:
            raise UsageError("Usage: LibDeflate:DecompressZlib(str): " + err)
        return self._decompress(s, 1, None)

    def DecompressZlibWithDict(self, s, dictionary):
        ok, err = _is_valid_arguments(s, True, dictionary)
        if not ok:
            raise UsageError("Usage: LibDeflate:DecompressZlibWithDict(str, dictionary): " + err)
        return self._decompress(s, 1, dictionary)

    # -- LibDeflate.lua:3311-3417
    def EncodeForPrint(self, s):
        if not isinstance(s, (bytes, bytearray)):
            raise UsageError("Usage: LibDeflate:EncodeForPrint(str): 'str' - string expected got '%s'." % _lua_type(s))
        N.init()
        cap = (len(s) * 8 + 5) // 6 + 16
        out = (C.c_ubyte * cap)()
        ol = C.c_size_t()
        N.check(N.lib.ldh_print6_encode(_buf(s), len(s), out, cap, C.byref(ol)))
        return bytes(out[: ol.value])

    def DecodeForPrint(self, s):
        if not isinstance(s, (bytes, bytearray)):
            raise UsageError("Usage: LibDeflate:DecodeForPrint(str): 'str' - string expected got '%s'." % _lua_type(s))
        N.init()
        cap = len(s) + 16
        out = (C.c_ubyte * cap)()
        ol = C.c_size_t()
        ok = C.c_int()
        N.check(N.lib.ldh_print6_decode(_buf(s), len(s), out, cap, C.byref(ol), C.byref(ok)))
        if not ok.value:
            return None
        return bytes(out[: ol.value])

    # ---- CreateCodec and the WoW channel codecs (LibDeflate.lua:2921-3158) ----------------------

    def CreateCodec(self, reserved_chars, escape_chars, map_chars):
        """(codec, None), or (None, message) exactly as the reference's two return values."""
        for v in (reserved_chars, escape_chars, map_chars):
            if not isinstance(v, (bytes, bytearray)):
                raise UsageError("Usage: LibDeflate:CreateCodec(reserved_chars, escape_chars, map_chars):"
                                 " All arguments must be string.")
        N.init()
        err = C.c_char_p()
        h = N.lib.ldh_codec_create(_buf(reserved_chars), len(reserved_chars), _buf(escape_chars), len(escape_chars),
                                   _buf(map_chars), len(map_chars), C.byref(err))
        if not h:
            return None, err.value.decode()
        return Codec(h), None

    def _addon_codec(self):
        if self._addon is None:  # LibDeflate.lua:3060-3062
            self._addon, _ = self.CreateCodec(b"\000", b"\001", b"")
        return self._addon

    def _chat_codec(self):
        if self._chat is None:  # LibDeflate.lua:3118-3124
            self._chat, _ = self.CreateCodec(b"sS\000\012\015\174%" + bytes(range(128, 256)), b"\035\037", b"\017\024")
        return self._chat

    def _channel(self, name, codec, s, decode):
        if not isinstance(s, (bytes, bytearray)):
            raise UsageError("Usage: LibDeflate:%s(str): 'str' - string expected got '%s'." % (name, _lua_type(s)))
        return codec.Decode(s) if decode else codec.Encode(s)

    def EncodeForWoWAddonChannel(self, s=None):
        return self._channel("EncodeForWoWAddonChannel", self._addon_codec(), s, False)

    def DecodeForWoWAddonChannel(self, s=None):
        return self._channel("DecodeForWoWAddonChannel", self._addon_codec(), s, True)

    def EncodeForWoWChatChannel(self, s=None):
        return self._channel("EncodeForWoWChatChannel", self._chat_codec(), s, False)

    def DecodeForWoWChatChannel(self, s=None):
        return self._channel("DecodeForWoWChatChannel", self._chat_codec(), s, True)


class Codec(object):
    """The table CreateCodec returns: Encode(str) and Decode(str) (nil -> None)."""

    def __init__(self, handle):
        self._h = handle

    def handle(self):
        return self._h

    def __del__(self):
        try:
            if self._h:
                N.lib.ldh_codec_free(self._h)
                self._h = None
        except Exception:
            pass

    def Encode(self, s=None):
        if not isinstance(s, (bytes, bytearray)):
            raise UsageError("Usage: codec:Encode(str): 'str' - string expected got '%s'." % _lua_type(s))
        from . import batch
        return batch.codec_encode_batch(self, [bytes(s)])[0]

    def Decode(self, s=None):
        if not isinstance(s, (bytes, bytearray)):
            raise UsageError("Usage: codec:Decode(str): 'str' - string expected got '%s'." % _lua_type(s))
        from . import batch
        return batch.codec_decode_batch(self, [bytes(s)])[0]


# -- LibDeflate.lua:1184-1197: str[index] goes to t[index - offset] for index in [start, stop] (1-based, as in Lua)
def _load_string_to_table(s, t, start, stop, offset):
    for index in range(start, stop + 1):
        t[index - offset] = s[index - 1]
    return t


_PRINT_ALPHABET = b"abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ0123456789()"  # LibDeflate.lua:3165-3297


class _Internals(object):
    """LibDeflate.internals (LibDeflate.lua:3426-3433): what the reference exposes to its own tests."""
    LoadStringToTable = staticmethod(_load_string_to_table)
    IsValidDictionary = staticmethod(_is_valid_dictionary)
    IsEqualAdler32 = staticmethod(lambda actual, expected: (actual % 4294967296) == (expected % 4294967296))
    _byte_to_6bit_char = {i: _PRINT_ALPHABET[i:i + 1] for i in range(64)}
    _6bit_to_byte = {c: i for i, c in enumerate(_PRINT_ALPHABET)}

    @staticmethod
    def InternalClearCache():  # LibDeflate.lua:3419-3422
        _LibDeflate._addon = None
        _LibDeflate._chat = None
        LibDeflate._addon = None
        LibDeflate._chat = None


LibDeflate = _LibDeflate()
LibDeflate.internals = _Internals

