"""Property checks of the CPU oracle: every stream it makes inflates to the input under an
independent inflater (zlib; and the reference's puff when oracle/_ref/puff was built), as the
reference's CheckCompressAndDecompress does (tests/Test.lua:386-642).  CPU only."""
import os
import subprocess
import zlib

import numpy as np
import pytest

from ldtest import ROOT, golden, o_compress, o_decompress, splitmix64

PUFF = os.path.join(ROOT, "oracle", "_ref", "puff")


def _inputs():
    rng = np.random.default_rng(0)
    yield b""
    yield b"a"
    yield b"ab"
    yield b"abc"
    yield b"aaa"
    yield b"abcabcabcabcabcabc"
    yield bytes(rng.integers(0, 256, 3000, dtype=np.uint8))
    yield bytes(rng.integers(97, 101, 5000, dtype=np.uint8))
    yield golden("item_strings.txt")
    yield golden("smalltest_no_newline.txt")
    for f in ("64kminus1.txt", "64k.txt", "64kplus1.txt"):
        yield golden(f)
    yield golden("encode_6bit_weakaura.txt")


@pytest.mark.parametrize("level", [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, None])
def test_roundtrip_levels(level):
    for s in _inputs():
        if level == 9 and len(s) > 70000:
            continue
        c, pad = o_compress(s, level=level)
        assert zlib.decompress(c, -15) == s
        assert o_decompress(c) == (s, 0)
        z, pad0 = o_compress(s, zlib=True, level=level)
        assert pad0 == 0 and zlib.decompress(z) == s
        assert o_decompress(z, zlib=True) == (s, 0)


@pytest.mark.parametrize("strategy", ["fixed", "huffman_only", "dynamic"])
def test_roundtrip_strategies(strategy):
    for s in _inputs():
        for level in (0, 1, 5):
            c, _ = o_compress(s, level=level, strategy=strategy)
            assert zlib.decompress(c, -15) == s
            if s and level != 0:
                btype = (c[0] >> 1) & 3
                assert btype == {"fixed": 1, "dynamic": 2}.get(strategy, btype)


@pytest.mark.parametrize("dlen", [1, 2, 3, 4, 100, 257, 258, 259, 260, 1000, 32768])
def test_roundtrip_dictionary(dlen):
    full = golden("dictionary32768.txt")
    d = full[-dlen:]
    rng = np.random.default_rng(dlen)
    cases = [b"", d, d[:1], d[-3:] * 3, golden("item_strings.txt"), d[len(d) // 2:] + b"xyz" + d[: len(d) // 2],
             bytes(rng.integers(0, 256, 500, dtype=np.uint8))]
    for s in cases:
        for level in (1, 3, 5, 7, 9):
            c, _ = o_compress(s, level=level, dict=d)
            o = zlib.decompressobj(-15, zdict=d)
            assert o.decompress(c) == s and o.eof
            assert o_decompress(c, dict=d) == (s, 0)
            z, _ = o_compress(s, zlib=True, level=level, dict=d)
            o = zlib.decompressobj(zdict=d)
            assert o.decompress(z) == s and o.eof
            assert o_decompress(z, zlib=True, dict=d) == (s, 0)


def test_dictionary_size_pins():
    # tests/Test.lua:1893-1957 (TestLength3String1/2/3, TestLength257/258/259String), all ten trials each, with
    # CheckDictEffectiveness (T:762-807): level 7, round trips, the dictionary must pay off
    from ldtest import dict_bound_cases
    for dct, s, bound in dict_bound_cases():
        c, _ = o_compress(s, level=7, dict=dct)
        assert len(c) <= bound, (len(dct), len(c), bound)
        assert o_decompress(c, dict=dct) == (s, 0)
        plain, _ = o_compress(s, level=7)
        assert len(plain) - len(c) > 0                                  # T:775-777
        z, _ = o_compress(s, zlib=True, level=7, dict=dct)
        assert o_decompress(z, zlib=True, dict=dct) == (s, 0)
        assert len(o_compress(s, zlib=True, level=7)[0]) - len(z) > -4  # T:794-799


def test_foreign_zlib_streams_inflate():
    # zlib-made streams (all levels) must inflate bit-exactly (Test.lua:574-606)
    for s in _inputs():
        for lvl in (1, 6, 9):
            co = zlib.compressobj(lvl, zlib.DEFLATED, -15)
            c = co.compress(s) + co.flush()
            assert o_decompress(c) == (s, 0)
            assert o_decompress(zlib.compress(s, lvl), zlib=True) == (s, 0)


@pytest.mark.skipif(not os.path.exists(PUFF), reason="oracle/_ref/puff not built")
def test_puff_agrees(tmp_path):
    for i, s in enumerate(_inputs()):
        c, _ = o_compress(s, level=5)
        p = tmp_path / ("c%d" % i)
        p.write_bytes(c)
        r = subprocess.run([PUFF, "-w", str(p)], capture_output=True)
        assert r.returncode == 0 and r.stdout == s


def test_random_garbage_never_crashes():
    v = splitmix64(12345, 4000)
    raw = v.tobytes()
    for i in range(0, 2000):
        n = 1 + (i % 37)
        chunk = raw[i * 8: i * 8 + n]
        d, st = o_decompress(chunk)
        assert d is None or isinstance(d, bytes)
        d, st = o_decompress(chunk, zlib=True)
