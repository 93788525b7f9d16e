"""GPU parity tests added in round 2 (through the C ABI): the big-batch path (workspace sub-batches), a header that
runs past the end of its input, streams longer than a *_dev call's bound, and the multi-context pool."""
import numpy as np
import pytest

from ldtest import golden, o_adler32, o_compress, o_decompress, truncated_dynamic_header_stream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def LD():
    from libdeflate_amd import LibDeflate
    return LibDeflate


@pytest.fixture(scope="module")
def B():
    from libdeflate_amd import batch
    return batch


def test_header_running_past_the_input_reports_exhaustion(B):
    # the dynamic header's last code lengths decode from the zero bits past the end, so the block
    # body starts beyond the input; the first token must report 2 (LD:2428), not decode zeros until the output is full
    bad = truncated_dynamic_header_stream()
    assert o_decompress(bad) == (None, 2)
    good = o_compress(b"hello hello hello", level=5)[0]
    # different start alignments of the stream inside the batch buffer (in_off mod 4 = 0..3)
    ins = []
    for k in range(4):
        ins += [good[: 1 + k], bad]
    got = B.decompress_batch(ins + [good], max_out=4096)
    for raw, g in zip(ins + [good], got):
        assert g == o_decompress(raw), raw.hex()
    assert got[1] == (None, 2) and got[-1] == (b"hello hello hello", 0)
    # zlib framing in front of the same block
    z = bytes([0x78, 0x9C]) + bad
    assert B.decompress_batch([z, b"x" + z][:1], fmt=1, max_out=4096)[0] == o_decompress(z, zlib=True) == (None, 2)


@pytest.mark.parametrize("level", [1, 5, 8])
def test_workspace_sub_batches(LD, B, level, monkeypatch):
    # a batch larger than LDH_WORKSPACE_MB is processed in sub-batches (compress_dev_locked's loop);
    # 64 MB of workspace holds about 40 streams of 128 KiB (about 20 behind the dictionary), so 150 streams make
    # at least four trips.  Byte-equal to the oracle, with and without the dictionary, ragged lengths included.
    from libdeflate_amd import synth
    monkeypatch.setenv("LDH_WORKSPACE_MB", "64")
    rows = synth.text_streams(150, 131072, first_stream=5000)
    streams = [r.tobytes() for r in rows]
    streams[7] = streams[7][:70000]
    streams[60] = b""
    streams[149] = streams[149][:1]
    got = B.compress_batch(streams, level=level)
    for i in (0, 7, 39, 40, 41, 60, 79, 80, 81, 120, 148, 149):  # both sides of every sub-batch edge
        assert got[i] == o_compress(streams[i], level=level), (level, i)
    back = B.decompress_batch([c for c, _ in got], max_out=131072)
    assert [d for d, _ in back] == streams and all(left == 0 for _, left in back)
    dct = golden("dictionary32768.txt")
    d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
    sub = streams[:70]
    gotd = B.compress_batch(sub, fmt=1, level=level, dictionary=d)
    for i in (0, 7, 19, 20, 21, 45, 60, 69):
        assert gotd[i] == o_compress(sub[i], zlib=True, level=level, dict=dct), (level, i)
    backd = B.decompress_batch([c for c, _ in gotd], fmt=1, dictionary=d, max_out=131072)
    assert [x for x, _ in backd] == sub


def test_dev_call_rejects_streams_longer_than_its_bound(B):
    # a stream longer than max_in_len was truncated silently (wrong bytes, status 0).  Now: out_len 0 for
    # it, LDH_E_TOO_LONG from ldh_sync(), the other streams of the batch are compressed as always.
    import torch
    from libdeflate_amd import _native as N
    from libdeflate_amd import synth
    rows = [r.tobytes() for r in synth.text_streams(4, 20000)]
    lens = [20000, 12000, 20000, 5]
    streams = [r[:n] for r, n in zip(rows, lens)]
    data = torch.frombuffer(bytearray(b"".join(streams)), dtype=torch.uint8).cuda()
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64).cuda()
    bound = 12000                                    # streams 0 and 2 exceed it
    stride = N.lib.ldh_compress_bound(bound)
    out = torch.zeros(4 * stride, dtype=torch.uint8, device="cuda")
    out_len = torch.full((4,), 77, dtype=torch.int32, device="cuda")
    pad = torch.zeros(4, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    B.device_compress(B.DeviceBatch(data, off, bound), out, stride, out_len, pad, level=5)
    with pytest.raises(N.LdhError, match="longer than max_in_len"):
        B.sync()
    B.sync()                                         # the flag is cleared by the report
    ol = out_len.cpu().numpy()
    assert ol[0] == 0 and ol[2] == 0
    o = out.cpu().numpy()
    for i in (1, 3):
        assert (o[i * stride: i * stride + ol[i]].tobytes(), int(pad[i])) == o_compress(streams[i], level=5)


def test_partition_by_bytes_matches_what_the_pool_does(B):
    from libdeflate_amd import _native as N
    off = np.array([0, 10, 10, 500, 1000, 1001, 4000, 4000, 8000], dtype=np.uint64)
    first = N.partition_by_bytes(off, 3)
    assert first[0] == 0 and first[-1] == 8 and first == sorted(first)


# ---- fixtures of the reference's own tests not yet replayed on the GPU path ("What's missing" 3)

def test_dictionary_size_bounds_T1893(LD, B):
    from ldtest import dict_bound_cases
    for dct, s, bound in dict_bound_cases():
        d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
        c, pad = LD.CompressDeflateWithDict(s, d, {"level": 7})
        assert (c, pad) == o_compress(s, level=7, dict=dct)
        assert len(c) <= bound, (len(dct), len(c), bound)
        assert LD.DecompressDeflateWithDict(c, d) == (s, 0)
        assert len(LD.CompressDeflate(s, {"level": 7})[0]) - len(c) > 0
        z, _ = LD.CompressZlibWithDict(s, d, {"level": 7})
        assert LD.DecompressZlibWithDict(z, d) == (s, 0)
        assert len(LD.CompressZlib(s, {"level": 7})[0]) - len(z) > -4


def test_second_return_T1553(LD, B):
    # Test2ndReturn: 100..300 random bytes from a limited alphabet, default level, 1..10 trailing bytes; raw / zlib,
    # with / without a dictionary of the same kind: the data comes back and the second result is the trailing count
    rng = np.random.default_rng(1553)

    def limited(n):
        return bytes(rng.integers(97, 97 + 6, n, dtype=np.uint8))
    for with_dict in (False, True):
        for fmt in (0, 1):
            ins, want, dcts = [], [], []
            for _ in range(10):
                dct = limited(int(rng.integers(100, 301)))
                s = limited(int(rng.integers(100, 301)))
                extra = limited(int(rng.integers(1, 11)))
                c = o_compress(s, zlib=bool(fmt), dict=dct if with_dict else None)[0]
                ins.append(c + extra)
                want.append((s, len(extra)))
                dcts.append(dct)
            for raw, w, dct in zip(ins, want, dcts):
                if with_dict:
                    d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
                    got = (LD.DecompressZlibWithDict if fmt else LD.DecompressDeflateWithDict)(raw, d)
                    comp = (LD.CompressZlibWithDict if fmt else LD.CompressDeflateWithDict)(w[0], d)[0]
                else:
                    got = (LD.DecompressZlib if fmt else LD.DecompressDeflate)(raw)
                    comp = (LD.CompressZlib if fmt else LD.CompressDeflate)(w[0])[0]
                assert got == w
                assert comp == raw[: len(comp)]        # and the GPU's own stream is the one the oracle made


def _first_block_type(c, is_zlib):                      # GetFirstBlockType, tests/Test.lua:350-360
    first = c[2] if is_zlib else c[0]
    return (first >> 1) & 3


def _randomize_padding(c, pad_bits, rng):               # PutRandomBitsInPaddingBits, tests/Test.lua:362-380
    if not pad_bits:
        return c
    keep = c[-1] & ((1 << (8 - pad_bits)) - 1)
    return c[:-1] + bytes([keep | (int(rng.integers(0, 256)) << (8 - pad_bits)) & 255])


def test_forced_strategies_first_block_type_T485(LD):
    # CheckCompressAndDecompress (T:386-642): strategy "fixed" / "dynamic" fix the first block's type except at level 0
    # (stored); the padding bits may hold anything
    rng = np.random.default_rng(485)
    cases = [golden("smalltest_no_newline.txt"), b"", b"a", golden("item_strings.txt"), bytes(rng.integers(0, 256, 3000, dtype=np.uint8))]
    for s in cases:
        for level in (0, 1, 3, 4, None):
            for strategy, want in (("fixed", 1), ("dynamic", 2)):
                cfg = {"strategy": strategy}
                if level is not None:
                    cfg["level"] = level
                c, pad = LD.CompressDeflate(s, cfg)
                assert _first_block_type(c, False) == (0 if level == 0 else want), (len(s), level, strategy)
                assert LD.DecompressDeflate(_randomize_padding(c, pad, rng)) == (s, 0)
                z, zpad = LD.CompressZlib(s, cfg)
                assert zpad == 0 and _first_block_type(z, True) == (0 if level == 0 else want)
                assert LD.DecompressZlib(z) == (s, 0)


def test_usage_errors_that_need_the_device_T2351(LD):
    from libdeflate_amd.LibDeflate import UsageError
    LD.CreateDictionary(b"1", 1, 0x00320032)                                                  # T:2352
    zeros = b"\0" * 32768
    ad = LD.Adler32(zeros)
    LD.CreateDictionary(zeros, 32768, ad)                                                      # T:2382
    with pytest.raises(UsageError) as e:                                                       # T:2393-2402
        LD.CreateDictionary(zeros, 32768, ad + 1)
    assert str(e.value) == ("Usage: LibDeflate:CreateDictionary(str, strlen, adler32): 'adler32' does not match the actual "
                            "adler32 of 'str'. 'adler32': %u, 'Adler32(str)': %u . Please check if 'str' is modified "
                            "unintentionally." % (ad + 1, ad))
    codec, err = LD.CreateCodec(b"1", b"2", b"")                                               # T:2470-2471
    assert codec is not None and err is None
    codec, _ = LD.CreateCodec(b"\000", b"\001", b"")                                           # T:2474-2480
    for fn, name in ((codec.Encode, "codec:Encode"), (codec.Decode, "codec:Decode")):
        with pytest.raises(UsageError) as e:
            fn()
        assert str(e.value) == "Usage: %s(str): 'str' - string expected got 'nil'." % name
    for name in ("EncodeForWoWAddonChannel", "DecodeForWoWAddonChannel", "EncodeForWoWChatChannel", "DecodeForWoWChatChannel"):
        with pytest.raises(UsageError) as e:                                                   # T:2481-2496
            getattr(LD, name)()
        assert str(e.value) == "Usage: LibDeflate:%s(str): 'str' - string expected got 'nil'." % name


def test_block_plan_of_long_streams(LD, B):
    # streams beyond 1 MiB are planned by a wavefront (k_plan_wave): runs of blocks evaluated for the eight bit
    # phases a run can be entered at, then chained.  The phase only matters through stored blocks (their padding,
    # which can tip the choice of type), so: incompressible stretches between text (stored blocks at odd bit
    # positions, at level 5), level 0 (stored only), forced strategies, zlib framing with its 16 / 48 header bits
    from libdeflate_amd import synth
    rng = np.random.default_rng(11)
    text = b"".join(r.tobytes() for r in synth.text_streams(12, 131072))
    noise = rng.integers(0, 256, 400000, dtype=np.uint8).tobytes()
    mixed = text[:300001] + noise[:131071] + text[300001:700000] + noise[131071:] + text[700000:1100003]
    assert len(mixed) > 17 * 65536
    streams = [mixed, mixed[:1200000], text[:1500000], b"tail"]
    for level in (5, 0):
        for s, g in zip(streams, B.compress_batch(streams, level=level)):
            assert g == o_compress(s, level=level), (level, len(s))
    for strategy in ("fixed", "dynamic", "huffman_only"):
        got = B.compress_batch(streams[:2], level=4, strategy=strategy)
        for s, g in zip(streams[:2], got):
            assert g == o_compress(s, level=4, strategy=strategy), (strategy, len(s))
    dct = golden("dictionary32768.txt")
    d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
    for s, g in zip(streams[:2], B.compress_batch(streams[:2], fmt=1, level=5, dictionary=d)):
        assert g == o_compress(s, zlib=True, level=5, dict=dct), len(s)
    got = B.compress_batch(streams, fmt=1, level=6)
    back = B.decompress_batch([c for c, _ in got], fmt=1, max_out=len(mixed))
    assert [x for x, _ in back] == streams


def test_whole_stream_runs_in_a_big_batch(B):
    # with 1024 streams or more a workgroup of k_seg_links takes a whole stream (here 10 segments, no segment sorted
    # twice) and one of k_match_lazy all its tiles (15, the window kept in LDS between them); smaller batches cut
    # streams into runs of four (covered by the other tests).  1100 streams of unequal length: three compared with the
    # oracle byte for byte, all round-tripped
    from libdeflate_amd import synth
    base = b"".join(r.tobytes() for r in synth.text_streams(6, 131072))
    rng = np.random.default_rng(5)
    streams = []
    for i in range(1100):
        n = 300000 - int(rng.integers(0, 70000))
        o = int(rng.integers(0, len(base) - n))
        streams.append(base[o:o + n])
    streams[7] = streams[7][:32768 * 9]       # ends on a segment edge
    streams[8] = streams[8][:16384 * 11 + 1]  # one position into a tile (tiles are 16384 positions since round 2)
    streams[9] = streams[9][:20480 * 11 + 1]  # (the round-1 tile size, kept as a case)
    got = B.compress_batch(streams, level=5)
    # (the host-pointer entry point cuts this batch into 64 MiB pieces of some 250 streams, so this test covers runs
    # of four segments / tiles behind the piece pipeline; the whole-stream mode itself is tests/test_gpu_round3.py)
    for i in (0, 7, 8, 9, 1099):
        assert got[i] == o_compress(streams[i], level=5), (i, len(streams[i]))
    back = B.decompress_batch([c for c, _ in got], max_out=300000)
    assert [x for x, _ in back] == streams


def test_pool_of_two_contexts_on_device_0(LD, B):
    # one library, many GPUs.  The box has one GPU, so the pool is two contexts on device 0:
    # the host-pointer batch calls split their streams by bytes over both (two host threads, two sets of streams and
    # workspace) and the results land in the caller's one buffer, identical to the single-context results.
    # (last test of this file on purpose: the pool stays for the rest of the process.)
    from libdeflate_amd import _native as N
    from libdeflate_amd import synth
    streams = [r.tobytes() for r in synth.text_streams(24, 40000, first_stream=300)]
    streams += [b"", b"abc" * 5000, golden("item_strings.txt")] + [r.tobytes() for r in synth.addon_streams(40)]
    dct = golden("dictionary32768.txt")
    d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
    single = B.compress_batch(streams, level=5)
    single_d = B.compress_batch(streams, fmt=1, level=None, dictionary=d)
    N.init_devices([0, 0])
    assert N.lib.ldh_pool_size() == 2
    assert B.compress_batch(streams, level=5) == single
    assert B.compress_batch(streams, fmt=1, level=None, dictionary=d) == single_d
    for i in (0, 23, 24, 26, 40):
        assert single[i] == o_compress(streams[i], level=5)
    comp = [c for c, _ in single]
    comp[5] = comp[5][:100]
    back = B.decompress_batch(comp, max_out=40000)
    for i, (raw, g) in enumerate(zip(comp, back)):
        assert g == (o_decompress(raw) if i == 5 else (streams[i], 0)), i
    backd = B.decompress_batch([c for c, _ in single_d], fmt=1, dictionary=d, max_out=40000)
    assert [x for x, _ in backd] == streams
    # a thread may pin itself to one member of the pool; single-string calls keep working on the default context
    N.check(N.lib.ldh_use_context(1))
    assert B.compress_batch(streams[:5], level=5) == single[:5]
    N.check(N.lib.ldh_use_context(-1))
    assert LD.CompressDeflate(streams[26])[0] == golden("item_strings_deflate.bin")
