"""Synthetic corpora for tests and bench.py.  All bytes come from
numpy.random.Generator(PCG64(seed)); generation is vectorised so the 64K x 4 KiB corpus of
BASELINE config 3 builds in seconds."""
import zlib

import numpy as np

DEFAULT_SEED = 0xE71A


def uniform(n, size, seed=DEFAULT_SEED):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.integers(0, 256, size=(n, size), dtype=np.uint8)


def _ascii_lines(rng, n, size):
    a = rng.integers(32, 127, size=(n, size), dtype=np.uint8)
    # digits: ~12% of positions in runs (mark run starts, spread by a small maximum filter)
    starts = rng.random((n, size)) < 0.025
    run = starts.copy()
    for k in range(1, 6):
        run[:, k:] |= starts[:, :-k] & (rng.random((n, size - k)) < 0.8 ** k)
    digits = rng.integers(48, 58, size=(n, size), dtype=np.uint8)
    a = np.where(run, digits, a)
    # newline every 16..80 bytes
    gaps = rng.integers(16, 81, size=(n, size // 16 + 1))
    pos = np.cumsum(gaps, axis=1)
    rows = np.repeat(np.arange(n)[:, None], pos.shape[1], axis=1)
    ok = pos < size
    a[rows[ok], pos[ok]] = 10
    return a


def _bracketed(rng, n, size):
    alphabet = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz      \n()[]<>{}\"'0123456789=/.", dtype=np.uint8)
    w = np.ones(len(alphabet))
    w[:26] = 3.0
    w[26:32] = 2.0
    w /= w.sum()
    idx = rng.choice(len(alphabet), size=(n, size), p=w)
    return alphabet[idx]


def _framed(rng, n, size):
    a = rng.integers(0, 256, size=(n, size), dtype=np.uint8)
    for i in range(n):
        hdr = int(rng.integers(0, 16))
        width = int(rng.choice([2, 4]))
        big = bool(rng.integers(0, 2))
        trailer = int(rng.choice([0, 1, 4]))
        body_len = size - hdr - width - trailer
        a[i, hdr:hdr + width] = np.frombuffer(body_len.to_bytes(width, "big" if big else "little"), dtype=np.uint8)
        if trailer == 1:
            a[i, size - 1] = np.bitwise_xor.reduce(a[i, :size - 1])
        elif trailer == 4:
            a[i, size - 4:] = np.frombuffer(zlib.crc32(a[i, :size - 4].tobytes()).to_bytes(4, "big"), dtype=np.uint8)
    return a


def mixed(n, size, seed=DEFAULT_SEED):
    """BASELINE config 3 'mixed-binary': 50% uniform bytes, 25% ASCII lines with embedded decimal
    numbers, 15% bracket/quote-structured text, 10% binary with a BE/LE length field and an
    optional xor8/crc32 trailer.  Returns uint8[n, size]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kind = rng.random(n)
    out = np.empty((n, size), dtype=np.uint8)
    m0, m1, m2, m3 = kind < 0.5, (kind >= 0.5) & (kind < 0.75), (kind >= 0.75) & (kind < 0.9), kind >= 0.9
    out[m0] = rng.integers(0, 256, size=(int(m0.sum()), size), dtype=np.uint8)
    out[m1] = _ascii_lines(rng, int(m1.sum()), size)
    out[m2] = _bracketed(rng, int(m2.sum()), size)
    out[m3] = _framed(rng, int(m3.sum()), size)
    return out


def as_arena(mat):
    """uint8[n, size] -> (flat data, uint64 off[n+1])"""
    n, size = mat.shape
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(size))
    return np.ascontiguousarray(mat).reshape(-1), off
