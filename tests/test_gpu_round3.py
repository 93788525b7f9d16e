"""GPU parity tests added in round 3 (through the C ABI, device-resident entry points).

VERDICT r2, "Parity: green, with one real hole": with 1024 streams or more in one *_dev call a workgroup of
k_seg_links takes a whole stream and one of k_match_lazy all its tiles -- the mode the bench and BASELINE configs[2]
run in.  The host-pointer entry points cut a batch into 64 MiB pieces, so only a *_dev call with >= 1024 streams is
sure to reach it; these tests make such calls at level 5, at level 8 (the deep-chain template instance, REFILL_MIN 12)
and with the 32 KiB dictionary (HAS_DICT, the seed mask carried across tiles), at the greedy levels too, compare a
sample of the streams with the oracle byte for byte, and round-trip all of them on the device.
"""
import numpy as np
import pytest

from ldtest import golden, o_adler32, o_compress

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def LD():
    from libdeflate_amd import LibDeflate
    return LibDeflate


@pytest.fixture(scope="module")
def B():
    from libdeflate_amd import batch
    return batch


# stream lengths around every edge the whole-stream kernels know: k_match_lazy's tiles (16384 positions), k_seg_links'
# segments (32768), the first deflate block (65535) and the ones after it (+32768)
EDGES = [16383, 16384, 16385, 32767, 32768, 32769, 49151, 49152, 49153, 65534, 65535, 65536, 65537, 81919, 81920, 81921,
         98303, 98304, 98305, 114688, 131071, 131072]


def big_batch(count, seed, max_len=131072):
    """`count` text streams of unequal length (the first len(EDGES) on the edges above), cut from G-text."""
    from libdeflate_amd import synth
    base = b"".join(r.tobytes() for r in synth.text_streams(10, 131072, first_stream=seed))
    rng = np.random.default_rng(seed)
    streams = []
    for i in range(count):
        n = EDGES[i] if i < len(EDGES) else int(rng.integers(20000, max_len + 1))
        n = min(n, max_len)
        o = int(rng.integers(0, len(base) - n))
        streams.append(base[o:o + n])
    return streams


def device_round_trip(B, streams, level, fmt=0, dictionary=None, sample=()):
    """Compress `streams` with ONE ldh_compress_batch_dev call, inflate the slots with ONE ldh_decompress_slots_dev call,
    check the round trip of every stream on the device; returns {i: (bytes, pad_bits)} for the sampled streams."""
    import torch
    from libdeflate_amd import _native as N
    lens = np.array([len(s) for s in streams], dtype=np.int64)
    max_len = int(lens.max())
    data = torch.frombuffer(bytearray(b"".join(streams) + b"\0" * 16), dtype=torch.uint8).cuda()
    off = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64).cuda()
    S = len(streams)
    stride = N.lib.ldh_compress_bound(max_len)
    comp = torch.empty(S * stride, dtype=torch.uint8, device="cuda")
    comp_len = torch.zeros(S, dtype=torch.int32, device="cuda")
    pad = torch.zeros(S, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    B.device_compress(B.DeviceBatch(data, off, max_len), comp, stride, comp_len, pad, fmt=fmt, level=level,
                      dictionary=dictionary)
    ostride = (max_len + 15) // 16 * 16
    back = torch.empty(S * ostride, dtype=torch.uint8, device="cuda")
    back_len = torch.zeros(S, dtype=torch.int32, device="cuda")
    left = torch.zeros(S, dtype=torch.int32, device="cuda")
    status = torch.zeros(S, dtype=torch.int32, device="cuda")
    B.device_decompress_slots(comp, stride, comp_len, S, back, ostride, back_len, left, status, fmt=fmt,
                              dictionary=dictionary)
    B.sync()
    assert int(status.abs().sum()) == 0 and int(left.sum()) == 0
    assert np.array_equal(back_len.cpu().numpy().astype(np.int64), lens)
    rows = back.view(S, ostride)
    src = data[: int(lens.sum())]
    # every stream's bytes, compared on the device: gather the valid part of each slot
    idx = torch.arange(ostride, device="cuda")[None, :] < torch.tensor(lens, device="cuda")[:, None]
    assert torch.equal(rows[idx], src)
    cl = comp_len.cpu().numpy()
    pb = pad.cpu().numpy()
    out = {}
    for i in sample:
        out[i] = (comp[i * stride: i * stride + int(cl[i])].cpu().numpy().tobytes(), int(pb[i]))
    return out


SAMPLE = list(range(len(EDGES))) + [len(EDGES) + 1, 500, 1023, 1024, 1099]


@pytest.mark.parametrize("level", [5, 8])
def test_whole_stream_mode_matches_oracle(B, level):
    streams = big_batch(1100, seed=31 + level)
    sample = SAMPLE if level == 5 else SAMPLE[::2] + [1099]
    got = device_round_trip(B, streams, level, sample=sample)
    for i in sample:
        assert got[i] == o_compress(streams[i], level=level), (level, i, len(streams[i]))


@pytest.mark.parametrize("level", [5, 7])
def test_whole_stream_mode_with_dictionary(LD, B, level):
    # zlib framing + the 32 KiB dictionary (BASELINE configs[3]'s shape): V = dictionary ++ input moves every edge by
    # 32768 positions, and the walks carry the dictionary's `seed` rule (LD:1394-1397) across tile changes
    dct = golden("dictionary32768.txt")
    d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
    streams = big_batch(1030, seed=77 + level, max_len=98304)
    # streams that begin with dictionary text, so that matches into the dictionary (and its last bytes) do occur
    for i in range(30, 60):
        k = 1000 * (i - 30)
        streams[i] = dct[k: k + 3000] + streams[i][3000:]
    streams[60] = dct[-300:] + streams[60][300:]
    sample = list(range(0, len(EDGES), 2)) + [30, 45, 59, 60, 1029]
    got = device_round_trip(B, streams, level, fmt=1, dictionary=d, sample=sample)
    for i in sample:
        assert got[i] == o_compress(streams[i], zlib=True, level=level, dict=dct), (level, i, len(streams[i]))


@pytest.mark.parametrize("level", [1, 3])
def test_big_batch_greedy_levels(B, level):
    streams = big_batch(1040, seed=5 + level, max_len=70000)
    sample = list(range(0, 14, 2)) + [1039]
    got = device_round_trip(B, streams, level, sample=sample)
    for i in sample:
        assert got[i] == o_compress(streams[i], level=level), (level, i, len(streams[i]))


def test_host_batches_on_caller_pinned_memory(LD, B):
    # buffers pinned by the caller (ldh_host_register / ldh_host_alloc) are copied from and to
    # directly, no staging.  Same results as the staged path: a ragged batch in many pieces, raw and zlib + dictionary,
    # with a failing stream (its slot may hold anything, its status must not), then the ranges are released again
    import ctypes as C
    from libdeflate_amd import _native as N
    from libdeflate_amd import synth
    N.init(0)
    streams = [r.tobytes() for r in synth.text_streams(40, 50000, first_stream=900)]
    streams[3] = streams[3][:17]
    streams[11] = b""
    streams[39] = streams[39][:40001]
    dct = golden("dictionary32768.txt")
    d = LD.CreateDictionary(dct, len(dct), o_adler32(dct))
    lens = np.array([len(s) for s in streams], dtype=np.uint64)
    off = np.zeros(len(streams) + 1, dtype=np.uint64)
    np.cumsum(lens, out=off[1:])
    n, total = len(streams), int(off[-1])
    stride = N.lib.ldh_compress_bound(50000)
    p_in = N.lib.ldh_host_alloc(total + 16)
    assert p_in
    inp = np.ctypeslib.as_array(C.cast(p_in, C.POINTER(C.c_uint8)), shape=(total + 16,))
    inp[:total] = np.frombuffer(b"".join(streams), dtype=np.uint8)
    # (ldh_host_register pins memory the caller already owns; on hosts where the driver refuses that -- this pool's boxes:
    # hipHostRegister returns "invalid argument" for any range -- it reports LDH_E_HIP and ldh_host_alloc is the way)
    probe = np.zeros(1 << 16, dtype=np.uint8)
    rc = N.lib.ldh_host_register(probe.ctypes.data, probe.nbytes)
    assert rc in (0, -102)
    registered = rc == 0
    if registered:
        N.check(N.lib.ldh_host_unregister(probe.ctypes.data))
    p_out = N.lib.ldh_host_alloc(n * stride)
    out = np.ctypeslib.as_array(C.cast(p_out, C.POINTER(C.c_uint8)), shape=(n * stride,))
    try:
        for fmt, dic, level in ((0, None, 5), (1, d, 7), (0, None, None)):
            want = B.compress_batch(streams, fmt=fmt, level=level, dictionary=dic)   # staged path (ordinary memory)
            out_len = np.zeros(n, dtype=np.uint32)
            pad = np.zeros(n, dtype=np.uint8)
            N.check(N.lib.ldh_compress_batch(p_in, off.ctypes.data, n, fmt, -1 if level is None else level, 0,
                                             dic.handle() if dic else None, p_out, stride, out_len.ctypes.data,
                                             pad.ctypes.data))
            got = [(out[i * stride: i * stride + int(out_len[i])].tobytes(), int(pad[i])) for i in range(n)]
            assert got == want, (fmt, level)
            # inflate from pinned memory into pinned slots; stream 5 is cut short
            comp = [c for c, _ in got]
            comp[5] = comp[5][: len(comp[5]) // 2]
            clens = np.array([len(c) for c in comp], dtype=np.uint64)
            coff = np.zeros(n + 1, dtype=np.uint64)
            np.cumsum(clens, out=coff[1:])
            p_c = N.lib.ldh_host_alloc(int(coff[-1]) + 16)
            cin = np.ctypeslib.as_array(C.cast(p_c, C.POINTER(C.c_uint8)), shape=(int(coff[-1]) + 16,))
            cin[: int(coff[-1])] = np.frombuffer(b"".join(comp), dtype=np.uint8)
            ostride = 50000 + 16
            p_o = N.lib.ldh_host_alloc(n * ostride)
            back = np.ctypeslib.as_array(C.cast(p_o, C.POINTER(C.c_uint8)), shape=(n * ostride,))
            blen = np.zeros(n, dtype=np.uint32)
            left = np.zeros(n, dtype=np.uint32)
            st = np.zeros(n, dtype=np.int32)
            N.check(N.lib.ldh_decompress_batch(p_c, coff.ctypes.data, n, fmt, dic.handle() if dic else None, p_o, ostride,
                                               blen.ctypes.data, left.ctypes.data, st.ctypes.data))
            ref = B.decompress_batch(comp, fmt=fmt, dictionary=dic, max_out=50000)
            for i in range(n):
                if ref[i][0] is None:
                    assert int(st[i]) == ref[i][1], i
                else:
                    assert int(st[i]) == 0 and back[i * ostride: i * ostride + int(blen[i])].tobytes() == ref[i][0], i
                    assert int(left[i]) == ref[i][1]
            N.lib.ldh_host_free(p_c)
            N.lib.ldh_host_free(p_o)
    finally:
        N.lib.ldh_host_free(p_out)
        N.lib.ldh_host_free(p_in)
    with pytest.raises(N.LdhError):
        N.check(N.lib.ldh_host_unregister(probe.ctypes.data))    # not (or no longer) registered


@pytest.mark.parametrize("count", [40, 1100])
def test_inflate_of_short_distance_matches(B, count):
    # The one-wavefront inflate stores a plain round's bytes a round later, just before the next round's loads, and
    # relies on a CU serving its waves' memory operations in order for those loads to see them.
    # Matches that read the bytes of the round before are what exercises that: streams of short periods (distances
    # 1 ... 300, around the 64 bytes a round produces at most), pure, with literal noise between the repeats, and mixed
    # with text.  40 streams take the wide kernels, 1100 one wavefront per stream; round trip of every stream on the
    # device, a few compared with the oracle.
    from libdeflate_amd import synth
    rng = np.random.default_rng(1234 + count)
    text = synth.text_streams(1, 131072)[0].tobytes()
    streams = []
    for i in range(count):
        p = 1 + (i * 7) % 300
        unit = bytes(rng.integers(0, 256, p, dtype=np.uint8))
        n = int(rng.integers(3000, 40000))
        kind = i % 4
        if kind == 0:
            s = (unit * (n // p + 1))[:n]
        elif kind == 1:  # a literal or two between the repeats
            parts = []
            while sum(map(len, parts)) < n:
                parts.append(unit * int(rng.integers(1, 6)))
                parts.append(bytes(rng.integers(0, 256, int(rng.integers(0, 3)), dtype=np.uint8)))
            s = b"".join(parts)[:n]
        elif kind == 2:  # repeats of changing period
            parts = []
            while sum(map(len, parts)) < n:
                q = int(rng.integers(1, 130))
                parts.append((unit * (q // p + 2))[:q] * int(rng.integers(2, 8)))
            s = b"".join(parts)[:n]
        else:
            o = int(rng.integers(0, len(text) - 2000))
            s = ((unit * (600 // p + 1))[:600] + text[o: o + 1500]) * (n // 2100 + 1)
            s = s[:n]
        streams.append(s)
    sample = [0, 1, 2, 3, count - 1]
    for level in (5, 1):
        got = device_round_trip(B, streams, level, sample=sample)
        for i in sample:
            assert got[i] == o_compress(streams[i], level=level), (level, i, len(streams[i]))
