"""GPU parity tests: libdeflate_hip.so (through the C ABI, via the Python host mirror) against the
CPU oracle and the committed golden fixtures.  Bit-exact everywhere (integer / byte work)."""
import os
import zlib

import numpy as np
import pytest

from ldtest import (ADDON_CODEC, CHAT_CODEC, golden, kat, kat_hex, o_adler32, o_codec_build, o_codec_decode,
                    o_codec_encode, o_compress, o_decompress, o_print6_decode, o_print6_encode, splitmix64)

pytestmark = pytest.mark.gpu

KAT = kat()


@pytest.fixture(scope="module")
def LD():
    from libdeflate_amd import LibDeflate
    return LibDeflate


@pytest.fixture(scope="module")
def B():
    from libdeflate_amd import batch
    return batch


def _files():
    return [golden(f) for f in ("item_strings.txt", "smalltest_no_newline.txt", "64kminus1.txt", "64k.txt",
                                "64kplus1.txt", "encode_6bit_weakaura.txt", "totalrp3.txt")]


def _small():
    rng = np.random.default_rng(7)
    return [b"", b"a", b"ab", b"abc", b"aaa", b"a" * 1000, bytes(range(256)) + bytes(range(255, -1, -1)),
            bytes(rng.integers(0, 256, 3000, dtype=np.uint8)), bytes(rng.integers(97, 101, 5000, dtype=np.uint8)),
            b"abcabcabcabcabcabcabcabc" * 40]


# ------------------------------------------------------------------ compress

def test_golden_streams_byte_exact(LD):
    item, dct = golden("item_strings.txt"), golden("dictionary32768.txt")
    d = LD.CreateDictionary(dct, 32768, 4072834167)  # tests/Test.lua:382-383
    assert LD.CompressDeflate(item)[0] == golden("item_strings_deflate.bin")
    assert LD.CompressDeflateWithDict(item, d)[0] == golden("item_strings_deflate_with_dict.bin")
    assert LD.CompressZlib(item) == (golden("item_strings_zlib.bin"), 0)
    assert LD.CompressZlibWithDict(item, d) == (golden("item_strings_zlib_with_dict.bin"), 0)


def test_pinned_sizes(LD):
    s = bytes(range(256)) + bytes(range(255, -1, -1))
    e = KAT["sizes_updown"]
    assert len(LD.CompressDeflate(s)[0]) == e["deflate_default"]
    assert len(LD.CompressDeflate(s, {"strategy": "fixed"})[0]) == e["deflate_fixed"]
    assert len(LD.CompressZlib(s, {"strategy": "fixed"})[0]) == e["zlib_fixed"]
    assert len(LD.CompressDeflate(s, {"strategy": "dynamic"})[0]) == e["deflate_dynamic"]
    assert len(LD.CompressZlib(s, {"strategy": "dynamic"})[0]) == e["zlib_dynamic"]
    a = b"a" * 1000
    e = KAT["sizes_a1000"]
    assert len(LD.CompressDeflate(a)[0]) == e["deflate_default"]
    assert len(LD.CompressDeflate(a, {"strategy": "huffman_only"})[0]) == e["deflate_huffman_only"]
    assert len(LD.CompressZlib(a)[0]) == e["zlib_default"]
    assert len(LD.CompressZlib(a, {"strategy": "huffman_only"})[0]) == e["zlib_huffman_only"]
    sm = golden("smalltest_no_newline.txt")
    assert [len(LD.CompressDeflate(sm, {"level": l})[0]) for l in range(10)] == KAT["smalltest_bounds"]
    assert LD.CompressDeflate(b"") == (bytes.fromhex("03fc"), 6)
    assert LD.CompressZlib(b"") == (bytes.fromhex("789c03fc00000001"), 0)


@pytest.mark.parametrize("level", [None, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_compress_matches_oracle_all_levels(B, level):
    cases = _small() + _files()
    if level == 9:
        cases = [c for c in cases if len(c) < 70000]
    for fmt in (0, 1):
        got = B.compress_batch(cases, fmt=fmt, level=level)
        for s, g in zip(cases, got):
            assert g == o_compress(s, zlib=bool(fmt), level=level), (level, fmt, len(s))


@pytest.mark.parametrize("strategy", ["fixed", "huffman_only", "dynamic"])
def test_compress_strategies(B, strategy):
    cases = _small() + _files()[:2]
    for level in (0, 1, 5, None):
        got = B.compress_batch(cases, level=level, strategy=strategy)
        for s, g in zip(cases, got):
            assert g == o_compress(s, level=level, strategy=strategy), (level, strategy, len(s))


@pytest.mark.parametrize("dlen", [1, 2, 3, 4, 100, 257, 258, 259, 260, 1000, 32768])
def test_compress_dictionary(LD, B, dlen):
    full = golden("dictionary32768.txt")
    dct = full[-dlen:]
    d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
    rng = np.random.default_rng(dlen)
    cases = [b"", dct, dct[:1], dct[-3:] * 3, golden("item_strings.txt"), dct[len(dct) // 2:] + b"xyz" + dct[: len(dct) // 2],
             bytes(rng.integers(0, 256, 500, dtype=np.uint8)), golden("smalltest_no_newline.txt"), golden("64kplus1.txt")]
    for level in (1, 2, 3, 5, 7):
        for fmt in (0, 1):
            got = B.compress_batch(cases, fmt=fmt, level=level, dictionary=d)
            for s, g in zip(cases, got):
                assert g == o_compress(s, zlib=bool(fmt), level=level, dict=dct), (dlen, level, fmt, len(s))


def test_ragged_batch_default_levels(B):
    # one call, streams of very different sizes: default level is resolved per stream (LD:1782-1790)
    v = splitmix64(99, 300)
    rng = np.random.default_rng(5)
    base = golden("totalrp3.txt") + golden("encode_6bit_weakaura.txt")
    cases = []
    for i in range(300):
        n = int(v[i] % 6000) if i % 7 else int(v[i] % 90000)
        o = int((v[i] >> 20) % (len(base) - n - 1))
        cases.append(base[o: o + n])
    cases += [b"", b"x", bytes(rng.integers(0, 256, 70000, dtype=np.uint8))]
    got = B.compress_batch(cases, level=None)
    for s, g in zip(cases, got):
        assert g == o_compress(s), len(s)
    got = B.compress_batch(cases, fmt=1, level=5)
    for s, g in zip(cases, got):
        assert g == o_compress(s, zlib=True, level=5), len(s)


def test_host_batches_in_many_pieces(B, monkeypatch):
    # the host-pointer entry points cut a batch into pieces (64 MiB) whose copies overlap the kernels of the next
    # piece; LDH_PIECE_KB shrinks the pieces so that this ragged batch goes through dozens of them, in groups whose
    # streams are not contiguous (default level: three size classes), with empty, failing and truncated streams
    v = splitmix64(7, 400)
    base = golden("totalrp3.txt") + golden("encode_6bit_weakaura.txt")
    cases = []
    for i in range(400):
        n = int(v[i] % 5000) if i % 5 else int(v[i] % 80000)
        o = int((v[i] >> 20) % (len(base) - n - 1))
        cases.append(base[o: o + n])
    cases[17] = b""
    cases[250] = b""
    whole = B.compress_batch(cases, level=None)
    whole5 = B.compress_batch(cases, fmt=1, level=5)
    monkeypatch.setenv("LDH_PIECE_KB", "16")
    assert B.compress_batch(cases, level=None) == whole
    assert B.compress_batch(cases, fmt=1, level=5) == whole5
    for i in (0, 17, 100, 399):
        assert whole[i] == o_compress(cases[i]), i
    comp = [c for c, _ in whole]
    comp[3] = comp[3][: len(comp[3]) // 2]      # truncated
    comp[200] = b"\xff" + comp[200]            # invalid block type
    pieces = B.decompress_batch(comp)
    monkeypatch.delenv("LDH_PIECE_KB")
    assert pieces == B.decompress_batch(comp)
    for i, (raw, g) in enumerate(zip(comp, pieces)):
        if i in (3, 200):
            assert g == o_decompress(raw), i
        else:
            assert g == (cases[i], 0), i
    # zlib streams in many pieces: neighbouring pieces' kernels run side by side on two streams, each piece with
    # trailer / Adler-32 arrays of its own (round 3); one stream with a wrong checksum, one cut inside its trailer
    comp5 = [c for c, _ in whole5]
    comp5[9] = comp5[9][:-1] + bytes([comp5[9][-1] ^ 1])
    comp5[301] = comp5[301][:-2]
    monkeypatch.setenv("LDH_PIECE_KB", "16")
    pieces5 = B.decompress_batch(comp5, fmt=1)
    monkeypatch.delenv("LDH_PIECE_KB")
    assert pieces5 == B.decompress_batch(comp5, fmt=1)
    for i, (raw, g) in enumerate(zip(comp5, pieces5)):
        if i in (9, 301):
            assert g == o_decompress(raw, zlib=True), i
        else:
            assert g == (cases[i], 0), i


def test_synthetic_configs(B):
    from libdeflate_amd import synth
    text = [r.tobytes() for r in synth.text_streams(6, 131072)]      # BASELINE configs[2] shape
    addon = [r.tobytes() for r in synth.addon_streams(256, 4096)]    # BASELINE configs[1] shape
    for level in (1, 5, 8):
        for s, g in zip(text, B.compress_batch(text, level=level)):
            assert g == o_compress(s, level=level), level
        for s, g in zip(addon, B.compress_batch(addon, level=level)):
            assert g == o_compress(s, level=level), level


def test_large_streams(LD, B):
    # streams whose virtual buffer does not fit LDS (the sort's global-memory variant), many deflate blocks,
    # matches at the far edge of the window, long runs; with and without a dictionary; round trip on the GPU too
    rng = np.random.default_rng(11)
    base = golden("totalrp3.txt") + golden("encode_6bit_weakaura.txt") + golden("item_strings.txt")
    blob = bytes(rng.integers(0, 256, 40000, dtype=np.uint8))
    cases = [
        (base * 6)[:400001],
        blob[:32768] * 9,                        # period exactly the window size
        blob[:32769] * 5,                        # one byte beyond it: no match may be found across periods
        b"a" * 300000,
        (b"abcdefgh" * 5000 + blob[:7000]) * 5,
        base[:140000] + blob + base[:140000],
    ]
    dct = golden("dictionary32768.txt")
    d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
    for level in (1, 3, 5, 6):
        got = B.compress_batch(cases, level=level)
        for s, g in zip(cases, got):
            assert g == o_compress(s, level=level), (level, len(s))
        back = B.decompress_batch([c for c, _ in got], max_out=max(len(c) for c in cases))
        assert [x for x, _ in back] == cases
    for level in (2, 5):
        got = B.compress_batch(cases[:2] + cases[4:], fmt=1, level=level, dictionary=d)
        for s, g in zip(cases[:2] + cases[4:], got):
            assert g == o_compress(s, zlib=True, level=level, dict=dct), (level, len(s))


def test_segmented_sort(LD, B):
    # lazy levels build the chains of anything longer than 16 KiB in segments of 32768 positions sorted inside LDS
    # (k_seg_links); a first occurrence's link comes from the previous segment's last-occurrence list, runs of four
    # segments share a workgroup and later runs sort the segment before them once more.  BASELINE configs[3] shape
    # (128 KiB streams behind the 32 KiB dictionary: five segments, two runs), one long stream next to
    # short ones (runs past a stream's end), and lengths around the segment and run edges
    from libdeflate_amd import synth
    dct = golden("dictionary32768.txt")
    d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
    text = [r.tobytes() for r in synth.text_streams(3, 131072)]
    text += [dct[5000:9000] + text[0][:70000], text[1][:98304 - 32768 + 1], text[2][:32768], text[2][:32769], b""]
    for level in (5, 8):
        for s, g in zip(text, B.compress_batch(text, fmt=1, level=level, dictionary=d)):
            assert g == o_compress(s, zlib=True, level=level, dict=dct), (level, len(s))
    long_one = b"".join(r.tobytes() for r in synth.text_streams(20, 131072))[:2500001]
    edge = [long_one, long_one[:98304], long_one[:98305], long_one[:131073], long_one[:196608], long_one[:196609], b"q"]
    edge += [long_one[7:7 + n] for n in (16385, 32767, 32768, 32769, 65535, 65536, 65537, 131071, 131072, 163841, 262144, 262145)]
    got = B.compress_batch(edge, level=5)
    for s, g in zip(edge, got):
        assert g == o_compress(s, level=5), len(s)
    back = B.decompress_batch([c for c, _ in got], max_out=len(long_one))
    assert [x for x, _ in back] == edge
    # a short dictionary in front: segment edges no longer fall on multiples of 32768 of the input, and the input is
    # not dword-aligned relative to V (the byte-wise staging path)
    d5 = LD.CreateDictionary(dct[-5:], 5, o_adler32(dct[-5:]))
    some = [long_one[:n] for n in (32763, 32764, 65531, 131067, 140000)]
    for s, g in zip(some, B.compress_batch(some, level=6, dictionary=d5)):
        assert g == o_compress(s, level=6, dict=dct[-5:]), len(s)
    # data whose trigrams are all different inside a segment and recur one segment later (every position is a
    # first occurrence: the look-up lists are full, several pieces), and data with one trigram only
    rng = np.random.default_rng(7)
    blk = rng.integers(0, 256, 32768, dtype=np.uint8).tobytes()
    odd = [blk * 3 + blk[:100], blk[:20000] + blk[:20000] + blk, b"\0" * 100000, (blk[:3] * 40000)[:100003]]
    for s, g in zip(odd, B.compress_batch(odd, level=4)):
        assert g == o_compress(s, level=4), len(s)


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 15, 16, 17, 47, 48, 63, 64, 65, 80, 127, 128, 129])
def test_tiny_batches(B, n):
    # a batch whose longest stream is tiny sizes every per-stream buffer (and every loop bound derived from it) small
    for s in (b"x" * n, bytes(range(n))):
        for level in (1, 5):
            got = B.compress_batch([s], level=level)
            assert got[0] == o_compress(s, level=level), (n, level)
            assert B.decompress_batch([got[0][0]], max_out=max(n, 1))[0][0] == s


def test_random_batch_shapes(LD, B):
    # many small batches of random composition: sizes, content classes, level, format, dictionary -- every buffer size
    # and loop bound that is derived from the batch (not from the stream) gets exercised at odd values
    rng = np.random.default_rng(2024)
    dct = golden("dictionary32768.txt")
    dicts = {n: LD.CreateDictionary(dct[-n:], n, o_adler32(dct[-n:])) for n in (5, 300, 32768)}
    text = golden("totalrp3.txt")

    def stream():
        kind = int(rng.integers(0, 5))
        n = int(rng.choice([0, 1, 2, 3, 7, 31, 32, 33, 95, 96, 97, 200, 1000, 4096, 5000, 66000], p=[.05] * 8 + [.08] * 4 + [.1, .1, .04, .04]))
        if kind == 0:
            return bytes(rng.integers(0, 256, n, dtype=np.uint8))
        if kind == 1:
            return bytes(rng.integers(97, 100, n, dtype=np.uint8))
        if kind == 2:
            o = int(rng.integers(0, len(text) - n - 1))
            return text[o:o + n]
        if kind == 3:
            return (b"ab" * n)[:n]
        return dct[-n:] if n else b""

    for it in range(120):
        cases = [stream() for _ in range(int(rng.integers(1, 5)))]
        level = [None, 0, 1, 2, 3, 4, 5, 6, 7][int(rng.integers(0, 9))]
        fmt = int(rng.integers(0, 2))
        dn = [None, 5, 300, 32768][int(rng.integers(0, 4))]
        d = dicts[dn] if dn else None
        got = B.compress_batch(cases, fmt=fmt, level=level, dictionary=d)
        for s, g in zip(cases, got):
            assert g == o_compress(s, zlib=bool(fmt), level=level, dict=dct[-dn:] if dn else None), (it, level, fmt, dn, len(s))
        back = B.decompress_batch([c for c, _ in got], fmt=fmt, dictionary=d, max_out=max(1, max(len(c) for c in cases)))
        assert [x for x, _ in back] == cases, (it, level, fmt, dn)


def test_corpus_shapes(B):
    # Synthetic stand-ins for the shapes of the reference's third-party corpus (tests/data/3rdparty, exercised by
    # T:1067-1307): incompressible data (stored blocks win), binary records, long zero runs, tiny strings, a short
    # phrase repeated, a string followed by its reverse.  (Those files themselves are not copied into this repo.)
    rng = np.random.default_rng(17)
    rec = np.zeros((6000, 20), dtype=np.uint8)
    rec[:, 0] = 0x0A
    rec[:, 1:5] = rng.integers(0, 4, (6000, 4))
    rec[:, 5:9] = np.arange(6000, dtype=np.uint32).view(np.uint8).reshape(-1, 4)
    rec[:, 9:] = rng.integers(0, 256, (6000, 11)) * (rng.integers(0, 8, (6000, 11)) == 0)
    fox = b"The quick brown fox jumps over the lazy dog"
    walk = bytes(rng.integers(0, 256, 33000, dtype=np.uint8))
    cases = [b"", b"x", b"xyzzy", b"xxxxxxxxxxyyyyyyyyyy", b"x" * 64, fox, fox * 4096, bytes(rng.integers(0, 256, 123093, dtype=np.uint8)),
             rec.tobytes(), b"\x00" * 262144, walk + walk[::-1], bytes(rng.integers(0, 256, 10000, dtype=np.uint8)),
             bytes((rng.integers(0, 256, 50000) * (rng.integers(0, 3, 50000) == 0)).astype(np.uint8))]
    for level, strategy in ((0, None), (1, None), (3, None), (5, None), (8, None), (5, "fixed"), (5, "dynamic"),
                            (1, "huffman_only"), (None, None)):
        got = B.compress_batch(cases, level=level, strategy=strategy)
        for s, g in zip(cases, got):
            assert g == o_compress(s, level=level, strategy=strategy), (level, strategy, len(s))
        back = B.decompress_batch([c for c, _ in got], max_out=262144)
        assert [x for x, _ in back] == cases


def test_no_dependence_on_stale_buffers(B):
    cases = _small() + _files()
    os.environ["LDH_POISON"] = "1"
    try:
        for level, fmt in ((5, 0), (1, 1), (0, 1), (8, 0)):
            for s, g in zip(cases, B.compress_batch(cases, fmt=fmt, level=level)):
                assert g == o_compress(s, zlib=bool(fmt), level=level), (level, fmt, len(s))
    finally:
        del os.environ["LDH_POISON"]


def test_full_size_roundtrip_property(B):
    # size-independent property at the bench's stream size: inflate(deflate(x)) == x, on the GPU
    from libdeflate_amd import synth
    text = [r.tobytes() for r in synth.text_streams(64, 131072, first_stream=1000)]
    for level in (1, 5, 8):
        comp = [c for c, _ in B.compress_batch(text, level=level)]
        back = B.decompress_batch(comp, max_out=131072)
        assert [d for d, _ in back] == text
        assert all(left == 0 for _, left in back)


# ------------------------------------------------------------------ decompress

def test_inflate_kat_vectors(LD):
    for v in KAT["inflate"]:
        raw = kat_hex(v["in"])
        f = LD.DecompressZlib if v["zlib"] else LD.DecompressDeflate
        data, second = f(raw)
        assert data == kat_hex(v["out"]), v["in"][:60]
        assert (data, second) == o_decompress(raw, zlib=bool(v["zlib"])), v["in"][:60]  # same status code too


def test_inflate_dict_vectors(LD):
    dct = b"abcdefgh"
    d = LD.CreateDictionary(dct, 8, 0x0E000325)
    for v in KAT["inflate_dict_abcdefgh"]:
        raw = kat_hex(v["in"])
        got = LD.DecompressZlibWithDict(raw, d) if v["use_dict"] else LD.DecompressZlib(raw)
        assert got[0] == kat_hex(v["out"]), v
        assert got == o_decompress(raw, zlib=True, dict=dct if v["use_dict"] else None)


def test_inflate_golden_streams(LD):
    item, dct = golden("item_strings.txt"), golden("dictionary32768.txt")
    d = LD.CreateDictionary(dct, 32768, 4072834167)
    assert LD.DecompressDeflate(golden("item_strings_deflate.bin")) == (item, 0)
    assert LD.DecompressDeflateWithDict(golden("item_strings_deflate_with_dict.bin"), d) == (item, 0)
    assert LD.DecompressZlib(golden("item_strings_zlib.bin")) == (item, 0)
    assert LD.DecompressZlibWithDict(golden("item_strings_zlib_with_dict.bin"), d) == (item, 0)


def test_inflate_own_and_foreign_streams(B):
    cases = _small() + _files()
    comp, want = [], []
    for s in cases:
        for lvl in (0, 1, 5, 8):
            comp.append(o_compress(s, level=lvl)[0])
            want.append(s)
        for st in ("fixed", "huffman_only", "dynamic"):
            comp.append(o_compress(s, level=5, strategy=st)[0])
            want.append(s)
        for zl in (1, 6, 9):  # zlib-made streams (tests/Test.lua:574-606)
            co = zlib.compressobj(zl, zlib.DEFLATED, -15)
            comp.append(co.compress(s) + co.flush())
            want.append(s)
    got = B.decompress_batch(comp)
    for w, g in zip(want, got):
        assert g == (w, 0)
    zcomp = [zlib.compress(s, 6) for s in cases] + [o_compress(s, zlib=True, level=5)[0] for s in cases]
    got = B.decompress_batch(zcomp, fmt=1)
    for w, g in zip(cases + cases, got):
        assert g == (w, 0)


def test_inflate_second_return_and_truncation(B):
    s = golden("smalltest_no_newline.txt")
    c = o_compress(s, level=5)[0]
    z = o_compress(s, zlib=True, level=5)[0]
    ins = [c + b"\x55" * k for k in range(0, 11)] + [c[:k] for k in range(0, len(c), 97)]
    for raw, g in zip(ins, B.decompress_batch(ins)):
        assert g == o_decompress(raw), len(raw)
    ins = [z + b"\xaa" * k for k in range(0, 11)] + [z[:k] for k in range(0, len(z), 97)] + [z[:-1] + b"\x00"]
    for raw, g in zip(ins, B.decompress_batch(ins, fmt=1)):
        assert g == o_decompress(raw, zlib=True), len(raw)


def test_inflate_with_dictionary(LD, B):
    full = golden("dictionary32768.txt")
    for dlen in (1, 3, 258, 259, 32768):
        dct = full[-dlen:]
        d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
        cases = [b"", dct, golden("item_strings.txt"), dct[-3:] * 5, golden("64kplus1.txt")]
        for fmt in (0, 1):
            comp = [o_compress(s, zlib=bool(fmt), level=7, dict=dct)[0] for s in cases]
            # foreign zlib stream using the dictionary in a later block is Q12 territory; zlib's own
            o = zlib.compressobj(9, zlib.DEFLATED, 15 if fmt else -15, zdict=dct)
            comp.append(o.compress(cases[2]) + o.flush())
            want = cases + [cases[2]]
            for raw, w, g in zip(comp, want, B.decompress_batch(comp, fmt=fmt, dictionary=d)):
                assert g == o_decompress(raw, zlib=bool(fmt), dict=dct)
                assert g == (w, 0)


def test_inflate_garbage_matches_oracle(B):
    # random bytes never crash and give the oracle's (nil, status) or data (tests/Test.lua:2780-2801)
    raw = splitmix64(4242, 8000).tobytes()
    ins = [raw[i * 13: i * 13 + 1 + (i % 61)] for i in range(3000)]
    ins += [bytes([0x78, 0x9C]) + x for x in ins[:500]]
    for x, g in zip(ins, B.decompress_batch(ins)):
        assert g == o_decompress(x), x.hex()
    for x, g in zip(ins, B.decompress_batch(ins, fmt=1)):
        assert g == o_decompress(x, zlib=True), x.hex()


def test_inflate_output_capacity_status(B):
    from libdeflate_amd import _native as N
    import ctypes as C
    s = b"hello " * 100
    c = o_compress(s, level=5)[0]
    out = (C.c_ubyte * 64)()
    ol, bl, st = C.c_size_t(), C.c_size_t(), C.c_int()
    buf = (C.c_ubyte * len(c)).from_buffer_copy(c)
    N.check(N.lib.ldh_decompress(buf, len(c), 0, None, out, 64, C.byref(ol), C.byref(bl), C.byref(st)))
    assert st.value == 1  # LDH_STATUS_OUTPUT_FULL


# ------------------------------------------------------------------ adler / print codec

def test_adler32(LD, B):
    for v in KAT["adler32"]:
        assert LD.Adler32(v["str"].encode()) == v["adler"]
    for f, want in KAT["adler32_files"].items():
        assert LD.Adler32(golden(f)) == want
    cases = _small() + _files() + [b"\xff" * 300000]
    assert B.adler32_batch(cases) == [o_adler32(s) for s in cases]
    assert B.adler32_batch(cases) == [zlib.adler32(s) for s in cases]


def test_print6(LD, B):
    dec, enc = golden("decode_6bit_weakaura.bin"), golden("encode_6bit_weakaura.txt")
    assert LD.EncodeForPrint(dec) == enc
    assert LD.DecodeForPrint(enc) == dec
    assert LD.DecodeForPrint(b" \n\t" + enc + b"\r\n ") == dec
    assert LD.DecodeForPrint(b"a") is None and LD.DecodeForPrint(b"ab!d") is None
    assert LD.DecodeForPrint(b"") == b"" and LD.EncodeForPrint(b"") == b""
    raw = splitmix64(77, 4000).tobytes()
    ins = [raw[i * 5: i * 5 + (i % 50)] for i in range(600)]
    encs = B.print6_encode_batch(ins)
    assert encs == [o_print6_encode(s) for s in ins]
    assert B.print6_decode_batch(encs) == ins
    noisy = [b"  " + e + b"\n" for e in encs] + [e[:3] + b"#" + e[3:] for e in encs if len(e) > 3] + [b" ", b"\x7f(a"]
    assert B.print6_decode_batch(noisy) == [o_print6_decode(s) for s in noisy]


# ------------------------------------------------------------------ device-resident API

def test_device_resident_api_matches_host_api(B):
    import torch
    from libdeflate_amd import synth
    rows = synth.text_streams(32, 20000)
    strings = [r.tobytes() for r in rows]
    data = torch.from_numpy(rows.reshape(-1).copy()).cuda()
    off = (torch.arange(33, dtype=torch.int64) * 20000).cuda()
    bt = B.DeviceBatch(data, off, 20000)
    from libdeflate_amd import _native as N
    stride = N.lib.ldh_compress_bound(20000)
    out = torch.empty(32 * stride, dtype=torch.uint8, device="cuda")
    out_len = torch.zeros(32, dtype=torch.int32, device="cuda")
    pad = torch.zeros(32, dtype=torch.uint8, device="cuda")
    B.device_compress(bt, out, stride, out_len, pad, level=5)
    B.sync()
    lens = out_len.cpu().numpy()
    o = out.cpu().numpy()
    for i, s in enumerate(strings):
        assert (o[i * stride: i * stride + lens[i]].tobytes(), int(pad[i])) == o_compress(s, level=5)
    # decompress on the device: compact the compressed streams first
    coff = torch.zeros(33, dtype=torch.int64)
    coff[1:] = torch.cumsum(torch.from_numpy(lens.astype(np.int64)), 0)
    comp = torch.cat([out[i * stride: i * stride + int(lens[i])] for i in range(32)])
    cb = B.DeviceBatch(comp, coff.cuda(), int(lens.max()))
    dout = torch.empty(32 * 20000, dtype=torch.uint8, device="cuda")
    dlen = torch.zeros(32, dtype=torch.int32, device="cuda")
    left = torch.zeros(32, dtype=torch.int32, device="cuda")
    status = torch.zeros(32, dtype=torch.int32, device="cuda")
    B.device_decompress(cb, dout, 20000, dlen, left, status)
    B.sync()
    assert int(status.abs().sum()) == 0 and int(left.sum()) == 0
    assert torch.equal(dout, data)


# ------------------------------------------------------------------ CreateCodec codecs

def test_codecs_match_oracle(LD, B):
    rng = np.random.default_rng(21)
    specs = [ADDON_CODEC, CHAT_CODEC]
    for seed in range(12):
        perm = bytes(np.random.default_rng(seed).permutation(256).astype(np.uint8))
        nr, ne, nm = (150, 3, 4) if seed % 3 == 0 else (1 + seed, 1 + seed % 3, seed % 2)
        specs.append((perm[:nr], perm[nr:nr + ne], perm[nr + ne:nr + ne + nm]))
    comp = o_compress(golden("totalrp3.txt"), level=5)[0]
    cases = [b"", b"\x00", b"\x01", bytes(range(256)) * 3, comp, bytes(rng.integers(0, 256, 70001, dtype=np.uint8)),
             bytes(rng.integers(0, 4, 5000, dtype=np.uint8)), b"sS%|\r\n" * 700, b"\x1d" * 4099, b"\x1d\x1f" * 2050]
    for spec in specs:
        oc, oerr = o_codec_build(*spec)
        gc, gerr = LD.CreateCodec(*spec)
        assert (oc is None) == (gc is None) and oerr == gerr
        if oc is None:
            continue
        enc = B.codec_encode_batch(gc, cases)
        assert enc == [o_codec_encode(oc, s) for s in cases]
        assert B.codec_decode_batch(gc, enc) == cases
        # Decode of arbitrary strings (pairs that are no escape pairs, reserved bytes -> nil, escape at the very end)
        junk = cases + [e[:-1] for e in enc if e] + [e[:len(e) // 2] + spec[0][:1] + e[len(e) // 2:] for e in enc]
        assert B.codec_decode_batch(gc, junk) == [o_codec_decode(oc, s) for s in junk]


def test_codec_api_surface(LD):
    assert LD.CreateCodec(b"1", b"", b"2") == (None, "No escape characters supplied.")           # T:2140-2142
    assert LD.CreateCodec(b"", b"1", b"")[1] == "No characters to encode."                       # T:2147-2149
    assert LD.CreateCodec(CHAT_CODEC[0], b"\x1d", b"\x0f\x14")[1] == "Out of escape characters."  # T:2168-2170
    with pytest.raises(Exception, match="All arguments must be string"):                         # T:2457-2469
        LD.CreateCodec(None, b"", b"")
    s = golden("item_strings.txt") + bytes(range(256))
    for enc, dec, bad in ((LD.EncodeForWoWAddonChannel, LD.DecodeForWoWAddonChannel, b"\x00"),
                          (LD.EncodeForWoWChatChannel, LD.DecodeForWoWChatChannel, b"sS\x00\n\r|%")):
        e = enc(s)
        assert not (set(e) & set(bad)) and dec(e) == s
        assert dec(e[:10] + bad[:1] + e[10:]) is None                                            # T:2108-2136
    assert LD.EncodeForWoWAddonChannel(b"\x00\x01\x02") == b"\x01\x02\x01\x03\x02"
    with pytest.raises(Exception, match=r"Usage: LibDeflate:EncodeForWoWChatChannel\(str\): 'str' - string expected got 'nil'"):
        LD.EncodeForWoWChatChannel()                                                             # T:2490-2492
