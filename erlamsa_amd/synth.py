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


def _json_value(rng, depth):
    r = rng.random()
    if depth > 3 or r < 0.45:
        k = int(rng.integers(0, 6))
        if k == 0:
            return str(int(rng.integers(-10 ** 6, 10 ** 9)))
        if k == 1:
            return "%.3f" % float(rng.random() * 1000)
        if k == 2:
            return ["true", "false", "null"][int(rng.integers(0, 3))]
        words = ["alpha", "beta gamma", "http://example.com/a/b?c=d", "dGhpcyBpcyBiYXNlNjQgdGV4dA==", "12 + 34", "x", ""]
        return '"%s"' % words[int(rng.integers(0, len(words)))]
    n = int(rng.integers(0, 5))
    if r < 0.75:
        return "{" + ",".join('"k%d":%s' % (int(rng.integers(0, 100)), _json_value(rng, depth + 1)) for _ in range(n)) + "}"
    return "[" + ",".join(_json_value(rng, depth + 1) for _ in range(n)) + "]"


def json_docs(n, seed=DEFAULT_SEED):
    """Compact JSON documents (what erlamsa_json:fold_ast/1 prints back unchanged), list[bytes]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return [_json_value(rng, 0 if rng.random() < 0.9 else 9).encode() for _ in range(n)]


def _sgml_elem(rng, depth):
    names = ["a", "div", "P", "xmlns:x", "item", "Br"]
    r = rng.random()
    if depth > 3 or r < 0.3:
        words = [b"text ", b"12 apples", b"http://h/p/q", b"line\n", b"c29tZSBiYXNlNjQgZGF0YQ==", b"x"]
        return words[int(rng.integers(0, len(words)))]
    nm = names[int(rng.integers(0, len(names)))].encode()
    attrs = b""
    for _ in range(int(rng.integers(0, 3))):
        k = int(rng.integers(0, 4))
        an = [b"id", b"xmlns", b"class", b"href"][k]
        q = [b"'", b'"', b""][int(rng.integers(0, 3))]
        val = [b"v1", b"http://e.org/ns", b"7"][int(rng.integers(0, 3))]
        attrs += b" " + an + b"=" + q + val + q
    if r < 0.4:
        return b"<" + nm + attrs + b" />"
    if r < 0.45:
        return b"<!-- note -->"
    kids = b"".join(_sgml_elem(rng, depth + 1) for _ in range(int(rng.integers(0, 4))))
    return b"<" + nm + attrs + b">" + kids + b"</" + nm + b">"


def sgml_docs(n, seed=DEFAULT_SEED):
    """Canonical SGML/XML-ish documents (erlamsa_sgml:fold_ast(parse(X)) == X for them), list[bytes]."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for _ in range(n):
        head = b"<?xml version='1.0'?>" if rng.random() < 0.5 else b""
        body = b""
        while not body.startswith(b"<") and not head:
            body = _sgml_elem(rng, 0)
        if head:
            body = _sgml_elem(rng, 0)
        out.append(head + body + b"".join(_sgml_elem(rng, 1) for _ in range(int(rng.integers(0, 3)))))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Counter-hash corpus (BASELINE configs[4]: 1 M x 64 KiB is 64 GiB - generated where it is used, never staged through the
# host).  Byte i of row r is a closed form of (seed, r, i): word j = i >> 3 of the row is splitmix64's finaliser of
# seed + r * G1 + (j + 1) * G2, the row's kind (0 uniform bytes, 1 text of letters / digit runs / newlines for sed_num,
# 2 a big-endian 32-bit length field in front of uniform bytes for the size-field detector, 3 an 8-letter alphabet with long
# repeats for fuse) comes from the same hash.  counter() is the numpy form (tests, oracle side), counter_torch() the same
# arithmetic on a torch device (bench.py fills the HBM arena with it, every rank for itself: no broadcast).
# ---------------------------------------------------------------------------------------------------------------------
_G1, _G2, _M1, _M2 = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def _mix_np(x):
    x = x.astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(_M1)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(_M2)
    return x ^ (x >> np.uint64(31))


def counter(rows, size, seed=DEFAULT_SEED):
    """rows: iterable of row numbers -> uint8[len(rows), size] of the counter-hash corpus"""
    assert size % 8 == 0 and size >= 8
    r = np.asarray(list(rows), dtype=np.uint64)[:, None]
    j = np.arange(size // 8, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        w = _mix_np(np.uint64(seed) + r * np.uint64(_G1) + (j + np.uint64(1)) * np.uint64(_G2))
        kind = (_mix_np(np.uint64(seed ^ 0x5851F42D4C957F2D) + r[:, 0] * np.uint64(_G1)) & np.uint64(3)).astype(np.int64)
    b = w.view(np.uint8).reshape(len(r), size) if w.dtype.byteorder in ("=", "<", "|") else None
    out = b.copy()
    c = b & 63
    text = np.where(c < 10, 48 + c, np.where(c < 36, 97 + c - 10, np.where(c < 62, 65 + c - 36, np.where(c == 62, 32, 10)))).astype(np.uint8)
    out[kind == 1] = text[kind == 1]
    out[kind == 3] = (97 + (b[kind == 3] & 7)).astype(np.uint8)
    fr = kind == 2
    out[fr, 0:4] = np.frombuffer(int(size - 4).to_bytes(4, "big"), dtype=np.uint8)
    return out


def counter_torch(arena, first_row, nrows, size, seed=DEFAULT_SEED, chunk_rows=512):
    """fills arena[first_row * size .. (first_row + nrows) * size) (a flat uint8 torch tensor on any device) with rows
    first_row .. of the counter-hash corpus; int64 arithmetic wraps like uint64, the logical shifts are masked"""
    import torch
    dev = arena.device

    def s64(v):
        v &= (1 << 64) - 1
        return v - (1 << 64) if v >= (1 << 63) else v

    def lsr(x, k):
        return (x >> k) & ((1 << (64 - k)) - 1)

    def mix(x):
        x = (x ^ lsr(x, 30)) * s64(_M1)
        x = (x ^ lsr(x, 27)) * s64(_M2)
        return x ^ lsr(x, 31)
    j = torch.arange(1, size // 8 + 1, dtype=torch.int64, device=dev)[None, :] * s64(_G2)
    sh = (torch.arange(8, dtype=torch.int64, device=dev) * 8)[None, None, :]
    hdr = torch.tensor(list(int(size - 4).to_bytes(4, "big")), dtype=torch.uint8, device=dev)
    for r0 in range(first_row, first_row + nrows, chunk_rows):
        m = min(chunk_rows, first_row + nrows - r0)
        r = torch.arange(r0, r0 + m, dtype=torch.int64, device=dev)
        w = mix(s64(seed) + r[:, None] * s64(_G1) + j)
        kind = mix(s64(seed ^ 0x5851F42D4C957F2D) + r * s64(_G1)) & 3
        b = ((w[:, :, None] >> sh) & 255).reshape(m, size)
        c = b & 63
        text = torch.where(c < 10, 48 + c, torch.where(c < 36, 87 + c, torch.where(c < 62, 29 + c, torch.where(c == 62, torch.full_like(c, 32), torch.full_like(c, 10)))))
        out = torch.where((kind == 1)[:, None], text, torch.where((kind == 3)[:, None], 97 + (b & 7), b)).to(torch.uint8)
        out[kind == 2, 0:4] = hdr
        arena[r0 * size:(r0 + m) * size] = out.reshape(-1)
    return arena


def as_arena(mat):
    """uint8[n, size] -> (flat data, uint64 off[n+1])"""
    n, size = mat.shape
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(size))
    return np.ascontiguousarray(mat).reshape(-1), off
