"""Shared helpers for the parity tests: synthetic corpora (BASELINE/SURVEY §8d) and comparison."""
import numpy as np

SEED_C = 0xE71A


def corpus_uniform(n, size, seed=SEED_C):
    rng = np.random.Generator(np.random.PCG64(seed))
    data = rng.integers(0, 256, size=n * size, dtype=np.uint8)
    return [data[i * size:(i + 1) * size].tobytes() for i in range(n)]


def _ascii_lines(rng, size):
    out = bytearray()
    while len(out) < size:
        ll = int(rng.integers(16, 81))
        line = bytearray(rng.integers(32, 127, size=ll, dtype=np.uint8).tobytes())
        if rng.random() < 0.5:
            num = str(int(rng.integers(0, 10 ** int(rng.integers(1, 12))))).encode()
            pos = int(rng.integers(0, max(1, len(line) - len(num))))
            line[pos:pos + len(num)] = num
        out += line + b"\n"
    return bytes(out[:size])


def _bracketed(rng, size):
    out = bytearray()
    opens, closes = b"([<{\"'", b")]>}\"'"
    stack = []
    while len(out) < size:
        r = rng.random()
        if r < 0.12 and len(stack) < 8:
            k = int(rng.integers(0, 6)); out.append(opens[k]); stack.append(closes[k])
        elif r < 0.24 and stack:
            out.append(stack.pop())
        elif r < 0.30:
            out += b"\n"
        else:
            out += rng.integers(97, 123, size=int(rng.integers(1, 9)), dtype=np.uint8).tobytes() + b" "
    return bytes(out[:size])


def _framed(rng, size):
    import zlib
    hdr = int(rng.integers(0, 16))
    width = int(rng.choice([1, 2, 4]))
    big = bool(rng.integers(0, 2))
    trailer = int(rng.choice([0, 1, 4]))
    body_len = size - hdr - width - trailer
    if width == 1:
        body_len = min(body_len, 255)
    head = rng.integers(0, 256, size=hdr, dtype=np.uint8).tobytes()
    body = rng.integers(0, 256, size=body_len, dtype=np.uint8).tobytes()
    lenf = body_len.to_bytes(width, "big" if big else "little")
    blob = head + lenf + body
    pad = size - len(blob) - trailer
    blob += rng.integers(0, 256, size=max(pad, 0), dtype=np.uint8).tobytes()
    if trailer == 1:
        x = 0
        for b in blob:
            x ^= b
        blob += bytes([x])
    elif trailer == 4:
        blob += zlib.crc32(blob).to_bytes(4, "big")
    return blob[:size]


def corpus_mixed(n, size, seed=SEED_C):
    """C3 'mixed-binary': 50% uniform bytes, 25% ASCII lines with numbers, 15% bracket/quote
    structured text, 10% binary with a length field and xor8/crc32 trailer."""
    rng = np.random.Generator(np.random.PCG64(seed))
    kinds = rng.random(n)
    out = []
    for i in range(n):
        k = kinds[i]
        if k < 0.5:
            out.append(rng.integers(0, 256, size=size, dtype=np.uint8).tobytes())
        elif k < 0.75:
            out.append(_ascii_lines(rng, size))
        elif k < 0.90:
            out.append(_bracketed(rng, size))
        else:
            out.append(_framed(rng, size))
    return out


def first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i
    return n if len(a) != len(b) else -1


# ---------------------------------------------------------------------------------------------
# Oracle result cache.  The GPU box has few host minutes, and most of the time of a `-m gpu` run is
# the single-threaded CPU oracle.  Results are therefore cached as per-case SHA-1 digests under
# tests/.oracle_cache/ (git-ignored, travels with the tree like the built .so files), keyed by the
# hash of the oracle's sources plus every input: a stale or missing entry is recomputed live.
# `EH_PRIME_ORACLE=1 pytest tests -m gpu` fills the cache on a machine without a GPU.
# ---------------------------------------------------------------------------------------------
import hashlib
import os

_CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".oracle_cache")
_so_hash = None


def _oracle_hash():
    """hash of the oracle's sources (not of the binary: the GPU box may rebuild it)"""
    global _so_hash
    if _so_hash is None:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
        h = hashlib.sha1()
        for f in ("oracle.cpp", "oracle.h", "otp_compat.h", "Makefile"):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
        _so_hash = h.hexdigest()
    return _so_hash


class OracleResult:
    """outputs (list[bytes] or None when only digests are known), lens, digests, status, draws, trace"""

    def __init__(self, outs, lens, digests, status, draws, trace):
        self.outs, self.lens, self.digests, self.status, self.draws, self.trace = outs, lens, digests, status, draws, trace

    def same(self, i, got):
        if self.outs is not None:
            return got == self.outs[i]
        return len(got) == int(self.lens[i]) and hashlib.sha1(got).digest() == self.digests[i].tobytes()


def oracle_batch(data, off, live=False, **kw):
    import pyoracle as po
    h = hashlib.sha1()
    h.update(_oracle_hash().encode())
    h.update(np.ascontiguousarray(data).tobytes()); h.update(np.ascontiguousarray(off).tobytes())
    h.update(repr(sorted((k, hashlib.sha1(np.ascontiguousarray(v).tobytes()).hexdigest() if isinstance(v, np.ndarray) else str(v))
                         for k, v in kw.items())).encode())
    path = os.path.join(_CACHE_DIR, h.hexdigest() + ".npz")
    if not live and os.path.exists(path):
        z = np.load(path, allow_pickle=False)
        return OracleResult(None, z["lens"], z["digests"], z["status"], z["draws"], str(z["trace"]).split("\n"))
    outs, st, dr, trace = po.fuzz_batch(data, off, trace=True, **kw)
    lens = np.array([len(o) for o in outs], dtype=np.int64)
    dig = np.frombuffer(b"".join(hashlib.sha1(o).digest() for o in outs), dtype=np.uint8).reshape(len(outs), 20) if outs else np.zeros((0, 20), np.uint8)
    try:
        os.makedirs(_CACHE_DIR, exist_ok=True)
        np.savez_compressed(path, lens=lens, digests=dig, status=st, draws=dr, trace=np.array(trace or ""))
    except OSError:
        pass
    return OracleResult(outs, lens, dig, st, dr, (trace or "").split("\n"))


def priming():
    return os.environ.get("EH_PRIME_ORACLE") == "1"


def oracle_live(data, off, threads=0, chunk=8, **kw):
    """The oracle run LIVE (no digest cache), cases spread over host threads in chunks of `chunk` (the ctypes call releases the
    GIL; a case is a pure function of the run's seed, its number and its input, so a chunk is a sub-range with first_case moved).
    On the GPU box 64 idle host cores make this cheaper than shipping digests computed elsewhere.  -> OracleResult with outputs."""
    import threading
    import pyoracle as po
    po.lib()
    n = len(off) - 1
    data = np.ascontiguousarray(data, dtype=np.uint8)
    off = np.ascontiguousarray(off, dtype=np.uint64)
    threads = threads or max(1, min(os.cpu_count() or 1, 64))
    first = kw.pop("first_case", 1)
    seeds = kw.pop("seeds", None)
    if kw.get("generators") and ("file" in kw["generators"] or "jump" in kw["generators"]) and kw.get("paths") is None:
        kw["paths"] = (data, off)                                  # the Paths of file / jump: the whole corpus, whatever chunk a call runs
    outs, st, dr, tr = [None] * n, np.zeros(n, np.int32), np.zeros(n, np.uint64), [""] * n
    lock, nxt, errs = threading.Lock(), [0], []

    def work():
        while True:
            with lock:
                a = nxt[0]
                if a >= n:
                    return
                nxt[0] = a + chunk
            b = min(a + chunk, n)
            d = data[int(off[a]):int(off[b])] if int(off[b]) > int(off[a]) else np.zeros(1, np.uint8)
            o = off[a:b + 1] - off[a]
            try:
                r = po.fuzz_batch(d, o, first_case=first + a, seeds=None if seeds is None else np.asarray(seeds).reshape(-1, 3)[a:b], trace=True, **kw)
            except Exception as ex:                                # noqa: BLE001 - reported by the caller's thread
                errs.append(ex)
                return
            outs[a:b] = r[0]; st[a:b] = r[1]; dr[a:b] = r[2]
            tl = (r[3] or "").split("\n")
            for j in range(b - a):
                tr[a + j] = tl[j] if j < len(tl) else ""
    ts = [threading.Thread(target=work) for _ in range(min(threads, (n + chunk - 1) // chunk or 1))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    if errs:
        raise errs[0]
    lens = np.array([len(x) for x in outs], dtype=np.int64)
    dig = np.frombuffer(b"".join(hashlib.sha1(x).digest() for x in outs), dtype=np.uint8).reshape(n, 20) if n else np.zeros((0, 20), np.uint8)
    return OracleResult(outs, lens, dig, st, dr, tr)


def meta_matches(eng, i, oracle_case):
    """case i's meta trace, rendered as erlamsa's meta logger prints it (erlamsa_amd/meta.py), against the oracle's full trace of the
    same case (pyoracle.fuzz_batch(trace="full"), cases split at "\\x1e\\n"); a trace the engine had to cut short counts as equal"""
    from erlamsa_amd import meta as _meta
    terms, cut = eng.meta_terms(i)
    return cut or _meta.lines(terms) == oracle_case
