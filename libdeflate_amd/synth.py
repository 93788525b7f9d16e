"""Deterministic synthetic inputs of SURVEY.md 8.d (numpy, CPU): G-text (Zipf words from a
generated vocabulary) and G-addon (WoW addon-style records).  Used by tests/ and bench.py.
Seeds: splitmix64 of (SEED ^ stream index); SEED is recorded in the bench JSON.
"""
import numpy as np

SEED = 0x4C69624465666C61


def _rng(stream):
    return np.random.Generator(np.random.PCG64((SEED ^ stream) & ((1 << 64) - 1)))


def _vocab(nwords=4096):
    rng = np.random.Generator(np.random.PCG64(SEED))
    lens = rng.integers(2, 11, nwords)
    freq = np.array([12, 9, 8, 7.5, 7, 6.7, 6.3, 6, 6, 4.3, 4, 2.8, 2.8, 2.4, 2.4, 2.2, 2, 2, 1.9, 1.5, 1, 0.8, 0.15,
                     0.15, 0.1, 0.07])
    letters = rng.choice(np.frombuffer(b"etaoinshrdlcumwfgypbvkjxqz", dtype=np.uint8), size=(nwords, 10),
                         p=freq / freq.sum())
    return lens, letters


_VOCAB = None


def text_streams(count, length, first_stream=0):
    """`count` G-text streams of exactly `length` bytes: Zipf(1.1) words of a 4096-word
    vocabulary, punctuation/newline every 60-80 characters.  Returns uint8 array
    (count, length)."""
    global _VOCAB
    if _VOCAB is None:
        _VOCAB = _vocab()
    wl, wt = _VOCAB
    nw = len(wl)
    ranks = np.arange(1, nw + 1, dtype=np.float64)
    p = ranks ** -1.1
    p /= p.sum()
    cdf = np.cumsum(p)
    out = np.empty((count, length), dtype=np.uint8)
    table = np.full((nw, 11), 32, dtype=np.uint8)  # word + ' '
    mask = np.zeros((nw, 11), dtype=bool)
    for i in range(nw):
        table[i, : wl[i]] = wt[i, : wl[i]]
        mask[i, : wl[i] + 1] = True
    for s in range(count):
        rng = _rng(first_stream + s)
        need = length // 5 + 64
        idx = np.searchsorted(cdf, rng.random(need))
        idx = np.minimum(idx, nw - 1)
        flat = table[idx][mask[idx]]
        while flat.size < length:
            flat = np.concatenate([flat, flat])
        flat = flat[:length].copy()
        # sentence / line breaks
        pos = np.cumsum(rng.integers(60, 81, length // 60 + 2))
        pos = pos[pos < length]
        flat[pos] = rng.choice(np.frombuffer(b".,;\n\n", dtype=np.uint8), size=pos.size)
        out[s] = flat
    return out


_KEYS = [b"name", b"icon", b"desc", b"ilvl", b"slot", b"qual", b"bind", b"stat", b"gem", b"ench", b"set", b"src",
         b"zone", b"boss", b"diff", b"tier"]


def addon_streams(count, length=4096, first_stream=0):
    """G-addon: AceSerializer-like tokens and item strings (modelled on
    tests/data/itemStrings.txt of the reference), exactly `length` bytes each."""
    out = np.empty((count, length), dtype=np.uint8)
    pool_rng = np.random.Generator(np.random.PCG64(SEED ^ 0xADD0))
    pool = pool_rng.integers(100000, 200000, 2000)
    zipf = (np.arange(1, 2001, dtype=np.float64)) ** -1.1
    zipf /= zipf.sum()
    for s in range(count):
        rng = _rng(first_stream + s)
        parts = [b"^1^T"]
        size = 4
        while size < length:
            if rng.random() < 0.6:
                iid = pool[rng.choice(2000, p=zipf)]
                b1, b2, b3 = rng.integers(1400, 3600, 3)
                rec = b"^Sitem:%d:%d::::::::110:::%d:%d:%d" % (iid, rng.integers(0, 6000) if rng.random() < 0.3 else 0,
                                                            b1, b2, b3)
            else:
                k = _KEYS[rng.integers(0, len(_KEYS))] + (b"%d" % rng.integers(0, 4))
                rec = b"^S" + k + b"^N%d" % rng.integers(0, 100000)
            parts.append(rec)
            size += len(rec)
        buf = b"".join(parts)[: length - 4] + b"^t^^"
        out[s] = np.frombuffer(buf, dtype=np.uint8)
    return out


def addon_record_pool(nrec=16384):
    """A pool of G-addon records (the same two record kinds as addon_streams) for the device-side generator of
    bench.py, which draws records from it with a Zipf law: item strings repeat the way they do in addon data."""
    rng = np.random.Generator(np.random.PCG64(SEED ^ 0xADD1))
    pool = np.random.Generator(np.random.PCG64(SEED ^ 0xADD0)).integers(100000, 200000, 2000)
    zipf = (np.arange(1, 2001, dtype=np.float64)) ** -1.1
    cdf = np.cumsum(zipf / zipf.sum())
    u = rng.random((nrec, 3))
    ids = pool[np.minimum(np.searchsorted(cdf, u[:, 1]), 1999)]
    ench = np.where(u[:, 2] < 0.3, rng.integers(0, 6000, nrec), 0)
    bonus = rng.integers(1400, 3600, (nrec, 3))
    keyi = rng.integers(0, len(_KEYS), nrec)
    keyn = rng.integers(0, 4, nrec)
    val = rng.integers(0, 100000, nrec)
    recs = []
    for i in range(nrec):
        if u[i, 0] < 0.6:
            recs.append(b"^Sitem:%d:%d::::::::110:::%d:%d:%d" % (ids[i], ench[i], bonus[i, 0], bonus[i, 1], bonus[i, 2]))
        else:
            recs.append(b"^S" + _KEYS[keyi[i]] + b"%d^N%d" % (keyn[i], val[i]))
    return recs
