"""Pins the CPU oracle (oracle/ld_oracle.c) against every golden vector the reference's own
tests hold for the hot path.  CPU only."""
import zlib

import pytest

from ldtest import (golden, kat, kat_hex, o_adler32, o_compress, o_decompress, o_print6_decode,
                    o_print6_encode)

KAT = kat()
DICT = golden("dictionary32768.txt")
ITEM = golden("item_strings.txt")


def test_golden_streams_byte_exact():
    # tests/data/reference/item_strings_*: LibDeflate's own output, default level (738 B -> 7)
    assert o_compress(ITEM)[0] == golden("item_strings_deflate.bin")
    assert o_compress(ITEM, dict=DICT)[0] == golden("item_strings_deflate_with_dict.bin")
    assert o_compress(ITEM, zlib=True) == (golden("item_strings_zlib.bin"), 0)
    assert o_compress(ITEM, zlib=True, dict=DICT) == (golden("item_strings_zlib_with_dict.bin"), 0)
    assert o_compress(ITEM, level=7)[0] == golden("item_strings_deflate.bin")


def test_golden_streams_inflate():
    assert o_decompress(golden("item_strings_deflate.bin")) == (ITEM, 0)
    assert o_decompress(golden("item_strings_deflate_with_dict.bin"), dict=DICT) == (ITEM, 0)
    assert o_decompress(golden("item_strings_zlib.bin"), zlib=True) == (ITEM, 0)
    assert o_decompress(golden("item_strings_zlib_with_dict.bin"), zlib=True, dict=DICT) == (ITEM, 0)


def test_pinned_sizes():
    s = bytes(range(256)) + bytes(range(255, -1, -1))
    e = KAT["sizes_updown"]
    assert len(o_compress(s)[0]) == e["deflate_default"]
    assert len(o_compress(s, strategy="fixed")[0]) == e["deflate_fixed"]
    assert len(o_compress(s, zlib=True, strategy="fixed")[0]) == e["zlib_fixed"]
    assert len(o_compress(s, strategy="dynamic")[0]) == e["deflate_dynamic"]
    assert len(o_compress(s, zlib=True, strategy="dynamic")[0]) == e["zlib_dynamic"]
    a = b"a" * 1000
    e = KAT["sizes_a1000"]
    assert len(o_compress(a)[0]) == e["deflate_default"]
    assert len(o_compress(a, strategy="huffman_only")[0]) == e["deflate_huffman_only"]
    assert len(o_compress(a, zlib=True)[0]) == e["zlib_default"]
    assert len(o_compress(a, zlib=True, strategy="huffman_only")[0]) == e["zlib_huffman_only"]


def test_smalltest_bounds_are_met_with_equality():
    s = golden("smalltest_no_newline.txt")
    assert len(s) == 28453
    sizes = [len(o_compress(s, level=l)[0]) for l in range(10)]
    assert sizes == KAT["smalltest_bounds"]  # the reference asserts <=; the restatement hits each bound


@pytest.mark.parametrize("name", ["encode_6bit_weakaura.txt", "totalrp3.txt"])
def test_benchmark_ratios(name):
    s = golden(name)
    for lvl, want in zip((1, 5, 8), KAT["ratios"][name]):
        c, _ = o_compress(s, level=lvl)
        assert zlib.decompress(c, -15) == s
        assert round(len(s) / len(c), 2) == want


def test_empty_string():
    assert o_compress(b"") == (bytes.fromhex("03fc"), 6)
    assert o_compress(b"", zlib=True) == (bytes.fromhex("789c03fc00000001"), 0)


def test_adler32_kats():
    for v in KAT["adler32"]:
        assert o_adler32(v["str"].encode()) == v["adler"], v
    for f, want in KAT["adler32_files"].items():
        assert o_adler32(golden(f)) == want


def test_inflate_vectors():
    for v in KAT["inflate"]:
        data, second = o_decompress(kat_hex(v["in"]), zlib=bool(v["zlib"]))
        assert data == kat_hex(v["out"]), v["in"][:60]
        assert isinstance(second, int)


def test_inflate_vectors_match_zlib():
    # cross-implementation check, as the reference does with zdeflate (Test.lua:654-718)
    for v in KAT["inflate"]:
        raw = kat_hex(v["in"])
        d = zlib.decompressobj(15 if v["zlib"] else -15)
        try:
            z = d.decompress(raw)
            ok = d.eof
        except zlib.error:
            ok = False
        want = kat_hex(v["out"])
        assert (want is not None) == ok, v["in"][:60]
        if ok:
            assert z == want


def test_inflate_dict_vectors():
    dct = b"abcdefgh"
    assert o_adler32(dct) == 0x0E000325
    for v in KAT["inflate_dict_abcdefgh"]:
        data, _ = o_decompress(kat_hex(v["in"]), zlib=True, dict=dct if v["use_dict"] else None)
        assert data == kat_hex(v["out"]), v


def test_second_return_counts_trailing_bytes():
    for extra in range(0, 11):
        c, _ = o_compress(b"hello hello hello hello" * 7)
        assert o_decompress(c + b"\x55" * extra) == (b"hello hello hello hello" * 7, extra)
        c, _ = o_compress(b"hello hello hello hello" * 7, zlib=True)
        assert o_decompress(c + b"\xaa" * extra, zlib=True) == (b"hello hello hello hello" * 7, extra)


def test_print6_golden_pair():
    dec = golden("decode_6bit_weakaura.bin")
    enc = golden("encode_6bit_weakaura.txt")
    assert o_print6_encode(dec) == enc
    assert o_print6_decode(enc) == dec
    assert o_print6_decode(b" \n\t" + enc + b"\r\n ") == dec


def test_print6_edges():
    assert o_print6_encode(b"") == b""
    assert o_print6_decode(b"") == b""
    assert o_print6_decode(b"a") is None
    assert o_print6_decode(b"ab!d") is None
    assert o_print6_decode(b"ab d") is None
    for n in range(0, 40):
        s = bytes((i * 37 + 11) & 255 for i in range(n))
        e = o_print6_encode(s)
        assert len(e) == (n * 8 + 5) // 6
        assert o_print6_decode(e) == s
