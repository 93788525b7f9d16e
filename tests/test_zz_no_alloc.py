"""The entry point that allocates nothing (ldh_decompress_batch_dev_noalloc) against ldh_decompress_batch_dev:
two calls in a row, the second with a smaller input, through the C ABI."""
import numpy as np
import pytest

from ldtest import golden, o_adler32, o_compress

pytestmark = pytest.mark.gpu


def _device_batch(torch, comp):
    lens = np.array([len(c) for c in comp], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)])
    data = torch.frombuffer(bytearray(b"".join(comp) + b"\0" * 8), dtype=torch.uint8).cuda()
    return data, torch.tensor(off, dtype=torch.int64).cuda()


def _run(N, torch, fn, comp, fmt, dict_handle, cap, no_alloc=False):
    data, off = _device_batch(torch, comp)
    n = len(comp)
    out = torch.full((n * cap,), 0xEE, dtype=torch.uint8, device="cuda")
    out_len = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    left = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    status = torch.full((n,), 99, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    allocs, free = N.lib.ldh_debug_alloc_count(), torch.cuda.mem_get_info()[0]
    N.check(fn(data.data_ptr(), off.data_ptr(), n, fmt, dict_handle, out.data_ptr(), cap, out_len.data_ptr(),
               left.data_ptr(), status.data_ptr()))
    N.check(N.lib.ldh_sync())
    if no_alloc:
        # show that nothing was allocated, not only that the results agree: the library's own count of
        # hipMalloc / hipHostMalloc / hipEventCreate calls, and the device's free memory as the driver reports it
        assert N.lib.ldh_debug_alloc_count() == allocs
        assert torch.cuda.mem_get_info()[0] == free
    ol, st = out_len.cpu().numpy(), status.cpu().numpy()
    rows = out.cpu().numpy().reshape(n, cap)
    return [(rows[i, : ol[i]].tobytes() if st[i] == 0 else None, int(st[i]), int(left[i])) for i in range(n)]


def test_no_alloc_entry_point_matches_the_allocating_one():
    import torch
    from libdeflate_amd import LibDeflate, synth
    from libdeflate_amd import _native as N
    N.init(0)
    dct = golden("dictionary32768.txt")
    d = LibDeflate.CreateDictionary(dct, len(dct), o_adler32(dct))
    big = [r.tobytes() for r in synth.text_streams(40, 30000, first_stream=77)] + [b"", b"q" * 999, golden("item_strings.txt")]
    small = big[5:12]
    cap = 30000 + 16
    for fmt, use_dict in ((1, False), (0, False), (1, True)):
        for streams in (big, small):      # the second call is the smaller one: stale scratch must not matter
            comp = [o_compress(s, zlib=bool(fmt), level=5, dict=dct if use_dict else None)[0] for s in streams]
            comp[2] = comp[2][:-1] + bytes([comp[2][-1] ^ 0x55])   # a trailer / data error
            comp[4] = comp[4][: len(comp[4]) // 2]
            h = d.handle() if use_dict else None
            want = _run(N, torch, N.lib.ldh_decompress_batch_dev, comp, fmt, h, cap)
            got = _run(N, torch, N.lib.ldh_decompress_batch_dev_noalloc, comp, fmt, h, cap, no_alloc=True)
            assert got == want, (fmt, use_dict, len(streams))
            assert [g[0] for i, g in enumerate(got) if i not in (2, 4)] == [s for i, s in enumerate(streams) if i not in (2, 4)]
    # beyond the static scratch: a clear error, no partial work
    rc = N.lib.ldh_decompress_batch_dev_noalloc(None, None, 65537, 0, None, None, 16, None, None, None)
    assert rc == -104 and b"at most 65536" in N.lib.ldh_last_error()
