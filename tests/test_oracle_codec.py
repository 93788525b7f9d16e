"""CreateCodec / WoW channel codecs: the oracle's restatement of LD:2921-3158.

The reference holds no byte fixtures for these codecs: its own tests compare LibDeflate's codec with a second
implementation embedded in tests/Test.lua (the LibCompress encoder, T:826-1030) and check round trips, the creation
messages (T:2138-2175) and that a reserved byte makes Decode fail (T:2108-2136).  The same is done here:
`libcompress_encode_table` restates that second implementation independently (Python, from T:855-968), hand-derived
known answers pin the two WoW codecs, and the messages are the reference's strings.  Parity of these codecs is
therefore pinned by agreement of two restatements and by properties, not by reference-made bytes."""
import numpy as np
import pytest

from ldtest import ADDON_CODEC, CHAT_CODEC, o_codec_build, o_codec_decode, o_codec_encode


def libcompress_encode_table(reserved, escape, mapped):
    """encode_translate of LibCompressEncoder:GetEncodeTable (T:855-968): byte -> bytes, or None + message."""
    if not escape:
        return None, "No escape characters supplied"
    if len(reserved) < len(mapped):
        return None, "Number of reserved characters must be at least as many as the number of mapped chars"
    if not reserved:
        return None, "No characters to encode"
    encode_bytes = reserved + escape + mapped
    taken = set(encode_bytes)
    tr = {}
    for f, t in zip(reserved, mapped):
        tr[f] = bytes([t])
    idx, r = 0, 0
    esc = escape[idx:idx + 1]
    for ch in encode_bytes:
        if ch in tr:
            continue
        while r >= 256 or r in taken:
            r += 1
            if r > 255:
                if not esc:
                    return None, "Out of escape characters"
                idx += 1
                esc = escape[idx:idx + 1]
                r = 0
        if not esc:
            return None, "Out of escape characters"  # (string_sub past the end: "" .. char would not decode)
        tr[ch] = esc + bytes([r])
        r += 1
    return tr, None


def lc_encode(tr, s):
    return b"".join(tr.get(b, bytes([b])) for b in s)


def test_addon_channel_known_answers():
    c, err = o_codec_build(*ADDON_CODEC)
    assert err is None
    # hand-derived from LD:2981-3015: taken = {0, 1}; "\0" -> "\1" + first free suffix 2, "\1" -> "\1" + 3
    assert o_codec_encode(c, b"\x00\x01\x02") == b"\x01\x02\x01\x03\x02"
    assert o_codec_encode(c, b"") == b""
    assert o_codec_decode(c, b"\x01\x02\x01\x03\x02") == b"\x00\x01\x02"
    assert o_codec_decode(c, b"a\x00b") is None          # reserved byte present (LD:3048)
    assert o_codec_decode(c, b"\x01\x07") == b"\x01\x07"  # not an escape pair: gsub leaves it


def test_chat_channel_known_answers():
    c, err = o_codec_build(*CHAT_CODEC)
    assert err is None
    # mapped: "s" -> \015, "S" -> \020; escaped with \029: \0 -> suffix 1 (0 is taken), \n -> 2, \r -> 3 ...
    assert o_codec_encode(c, b"sS") == b"\x0f\x14"
    assert o_codec_encode(c, b"\x00\n\r") == b"\x1d\x01\x1d\x02\x1d\x03"
    enc = o_codec_encode(c, bytes(range(256)))
    bad = set(CHAT_CODEC[0])
    assert not (set(enc) & bad)
    assert o_codec_decode(c, enc) == bytes(range(256))
    # 7 + 128 reserved, 2 escape, 2 mapped = 139 bytes to translate, 2 of them by mapping: 137 escapes; suffixes are the
    # 256 - 139 = 117 free byte values, so the second escape character \031 is needed for the last 20
    assert c.npass == 2 and bytes(c.pass_escape[:2]) == b"\x1d\x1f"
    assert sum(1 for b in range(256) if c.enc_len[b] == 2 and c.enc0[b] == 0x1f) == 20


def test_create_codec_messages():  # T:2138-2175
    assert o_codec_build(b"1", b"", b"2") == (None, "No escape characters supplied.")
    assert o_codec_build(b"1", b"a", b"23")[1] == ("The number of reserved characters must be at least as many as "
                                                   "the number of mapped chars.")
    assert o_codec_build(b"", b"1", b"")[1] == "No characters to encode."
    dup = "There must be no duplicate characters in the concatenation of reserved_chars, escape_chars and map_chars."
    for args in ((b"1", b"2", b"1"), (b"2", b"1", b"1"), (b"1", b"1", b"2")):
        assert o_codec_build(*args)[1] == dup
    assert o_codec_build(CHAT_CODEC[0], b"\x1d", b"\x0f\x14")[1] == "Out of escape characters."
    assert o_codec_build(*CHAT_CODEC)[1] is None


@pytest.mark.parametrize("seed", range(40))
def test_matches_libcompress_restatement(seed):  # CheckEncodeAndDecode, T:1033-1062
    rng = np.random.default_rng(seed)
    perm = bytes(rng.permutation(256).astype(np.uint8))
    nr = int(rng.integers(1, 200 if seed % 4 == 0 else 12))
    ne = int(rng.integers(1, 4))
    nm = int(rng.integers(0, min(nr, 5) + 1))
    spec = (perm[:nr], perm[nr:nr + ne], perm[nr + ne:nr + ne + nm])
    tr, lerr = libcompress_encode_table(*spec)
    c, err = o_codec_build(*spec)
    assert (c is None) == (tr is None), (err, lerr)
    if c is None:
        return
    for n in (0, 1, 50, 3000):
        s = bytes(rng.integers(0, 256, n, dtype=np.uint8))
        enc = o_codec_encode(c, s)
        assert enc == lc_encode(tr, s)
        assert not (set(enc) & set(spec[0]))
        assert o_codec_decode(c, enc) == s
        if enc:  # T:2108-2136: a reserved byte anywhere makes Decode return nil
            k = int(rng.integers(0, len(enc)))
            assert o_codec_decode(c, enc[:k] + spec[0][:1] + enc[k:]) is None


def test_wow_codecs_match_libcompress_restatement():
    rng = np.random.default_rng(3)
    for spec in (ADDON_CODEC, CHAT_CODEC):
        tr, _ = libcompress_encode_table(*spec)
        c, _ = o_codec_build(*spec)
        for n in (1, 255, 256, 70000):
            s = bytes(rng.integers(0, 256, n, dtype=np.uint8))
            assert o_codec_encode(c, s) == lc_encode(tr, s)
            assert o_codec_decode(c, o_codec_encode(c, s)) == s
