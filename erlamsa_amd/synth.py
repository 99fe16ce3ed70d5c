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


def as_arena(mat):
    """uint8[n, size] -> (flat data, uint64 off[n+1])"""
    n, size = mat.shape
    off = (np.arange(n + 1, dtype=np.uint64) * np.uint64(size))
    return np.ascontiguousarray(mat).reshape(-1), off
