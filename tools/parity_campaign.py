#!/usr/bin/env python3
"""One-off randomized parity campaign on the GPU (a wider net than tests/: not collected by pytest): random
batch shapes around the sort's narrow / wide / segmented thresholds, all lazy levels, with and without a
dictionary, raw and zlib, compared byte for byte with the oracle and round-tripped on the device.
usage: python tools/parity_campaign.py [seconds] [big]
`big`: every batch is padded to 1024..1100 streams and goes through ONE ldh_compress_batch_dev call, so that the
whole-stream runs of k_seg_links / k_match_lazy (what the bench and BASELINE configs[2] run) are what is compared.
tests/test_gpu_campaign.py runs a bounded sample of both modes (campaign(...) below) under `pytest -m gpu`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ldtest as T
from libdeflate_amd import LibDeflate as LD, batch as B, synth, _native as N

rng = None
pool = b"".join(r.tobytes() for r in synth.text_streams(8, 131072)) + T.golden("totalrp3.txt") + T.golden("encode_6bit_weakaura.txt")
noise = b""
dct = T.golden("dictionary32768.txt")
dicts = {n: LD.CreateDictionary(dct[:n], n, T.o_adler32(dct[:n])) for n in (1, 300, 32768)}
# the narrow / wide boundary (16 KiB), k_seg_links' segments (32768) and runs, k_match_lazy's tiles (16384 + a margin
# of 4096; 20480 in round 1) and runs of four, k_plan_wave's threshold (17 blocks of 65536)
edges = [16384 - 48, 16384 - 47, 16336, 16400, 20480, 32768, 40960, 61440, 65536, 81920, 98304, 98305, 131072 - 48, 131072,
         131105, 163840, 196608, 229376, 262144, 294912, 400000, 1048576, 1114112, 1200000]


def piece(n):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        o = int(rng.integers(0, len(pool) - n)) if n < len(pool) else 0
        return (pool * (n // len(pool) + 1))[o:o + n]
    if kind == 1:
        return (noise * (n // len(noise) + 1))[:n]
    if kind == 2:
        # (half of them short: matches at distances around the 64 bytes an inflate round produces)
        per = int(rng.integers(1, 300)) if rng.random() < 0.5 else int(rng.integers(1, 40000))
        return (noise[:per] * (n // per + 1))[:n]
    if kind == 3:
        return bytes([int(rng.integers(0, 256))]) * n
    a = piece(n // 2)
    return a + piece(n - len(a))


def campaign(budget, BIG=False, max_batches=None, seed=20261110, verbose=True):
    """Random batches for `budget` seconds (or `max_batches`, whichever ends first); returns (batches, streams checked).
    Raises AssertionError on the first stream that differs from the oracle or does not round-trip."""
    global rng, noise
    rng = np.random.default_rng(seed)
    noise = bytes(rng.integers(0, 256, 70000, dtype=np.uint8))
    t0 = time.time()
    rounds = checked = 0
    while time.time() - t0 < budget and (max_batches is None or rounds < max_batches):
        level = int(rng.choice([4, 5, 5, 6, 7, 8, 9, 1, 1, 2, 3]))
        fmt = int(rng.integers(0, 2))
        dn = int(rng.choice([0, 0, 1, 300, 32768]))
        d = dicts.get(dn)
        cnt = int(rng.integers(1, 7))
        lens = [int(rng.choice(edges)) + int(rng.integers(-3, 4)) if rng.random() < 0.5 else int(rng.integers(0, 300000)) for _ in range(cnt)]
        if level >= 7:
            lens = [min(x, 300000) for x in lens]  # (the oracle's deep levels are slow on long streams)
        if level == 9:
            lens = [min(x, 60000) for x in lens]  # (the oracle at level 9 is slow)
        cases = [piece(max(0, x)) for x in lens]
        if BIG:
            # the random streams first (they are the ones compared), padding streams of modest length behind them; one
            # *_dev call for all, round trip of every stream checked on the device (tests/test_gpu_round3.py's helper)
            import test_gpu_round3 as R3
            cases = [c[:200000] for c in cases]
            pad = R3.big_batch(int(rng.integers(1024, 1101)) - len(cases), seed=int(rng.integers(0, 1000)), max_len=40000)
            allc = cases + pad
            got_map = R3.device_round_trip(B, allc, level, fmt=fmt, dictionary=d, sample=range(len(cases)))
            for i, s in enumerate(cases):
                want = T.o_compress(s, zlib=bool(fmt), level=level, dict=dct[:dn] if dn else None)
                assert got_map[i] == want, ("MISMATCH", level, fmt, dn, len(s))
                checked += 1
            rounds += 1
            if rounds % 5 == 0:
                if verbose: print("%d big batches, %d streams checked, %.0f s" % (rounds, checked, time.time() - t0), flush=True)
            continue
        got = B.compress_batch(cases, fmt=fmt, level=level, dictionary=d)
        for s, g in zip(cases, got):
            want = T.o_compress(s, zlib=bool(fmt), level=level, dict=dct[:dn] if dn else None)
            assert g == want, ("MISMATCH", level, fmt, dn, len(s))
            checked += 1
        back = B.decompress_batch([c for c, _ in got], fmt=fmt, dictionary=d, max_out=max(max(len(c) for c in cases), 1))
        assert [x for x, _ in back] == cases, ("ROUNDTRIP", level, fmt, dn, lens)
        rounds += 1
        if rounds % 10 == 0:
            if verbose: print("%d batches, %d streams checked, %.0f s" % (rounds, checked, time.time() - t0), flush=True)
    return rounds, checked


if __name__ == "__main__":
    N.init(0)
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    big = len(sys.argv) > 2 and sys.argv[2] == "big"
    rounds, checked = campaign(budget, big)
    print("campaign ok: %d batches, %d streams byte-identical to the oracle and round-tripped" % (rounds, checked))
