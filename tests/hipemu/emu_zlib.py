#!/usr/bin/env python3
"""The device deflate / inflate (erlamsa_amd/csrc/eh_zlib.h) against zlib itself (Python's zlib module = the image's libz 1.2.11,
the library OTP's zlib module binds): compressed streams BYTE FOR BYTE (level 6, windowBits 15, memLevel 8, one shot), and the
decoders' behaviour on complete, truncated and corrupted inputs as mutate_once_compressed/6 sees it (erlamsa_patterns.erl:216-246).

  ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_zlib.py [quick]
"""
import os
import sys
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import util
import erlamsa_amd as ea


def want_compress(op, data):
    if op == 0:
        c = zlib.compressobj(6, zlib.DEFLATED, -15, 8, zlib.Z_DEFAULT_STRATEGY)
    elif op == 1:
        c = zlib.compressobj(6, zlib.DEFLATED, 31, 8, zlib.Z_DEFAULT_STRATEGY)
    else:
        c = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_DEFAULT_STRATEGY)
    return c.compress(data) + c.flush()


def want_gunzip(data):
    """zlib:gunzip/1 of OTP 20.1 - 23: inflateInit(Z, 16 + 15, reset), inflate, inflateEnd - every concatenated member is decoded
    (inflateReset at each end of stream with input left); anything that is not a sequence of complete members raises"""
    out, rest = b"", data
    while True:
        d = zlib.decompressobj(31)
        try:
            out += d.decompress(rest)
        except zlib.error:
            return None
        if not d.eof:
            return None
        rest = d.unused_data
        if not rest:
            return out


def want_inflate(data):
    """zlib:inflateInit(Z), zlib:inflate(Z, Bin), no inflateEnd: a stream that just stops yields what was decoded"""
    d = zlib.decompressobj(15)
    try:
        return d.decompress(data)
    except zlib.error:
        return None


def samples(rng, quick, small=False):
    out = [b"", b"a", b"ab", b"abc", b"aaaa", b"abcabcabcabc", bytes(1000), bytes(range(256)) * 3]
    sizes = [5, 17, 100, 257, 1000, 4096] + ([] if small else [20000]) + ([] if quick else [70000, 140000, 300000])
    for n in sizes:
        out.append(rng.integers(0, 256, size=n, dtype=np.uint8).tobytes())                       # incompressible: stored blocks
        out.append(rng.integers(0, 4, size=n, dtype=np.uint8).tobytes())                         # short alphabet
        out.append(util.corpus_mixed(1, n, seed=n)[0])
        out.append((b"The quick brown fox jumps over the lazy dog. " * (n // 45 + 1))[:n])        # long matches
        per = rng.integers(0, 256, size=int(rng.integers(3, 300)), dtype=np.uint8).tobytes()
        b = bytearray((per * (n // len(per) + 1))[:n])
        for _ in range(n // 500):
            b[int(rng.integers(0, n))] ^= 1 << int(rng.integers(0, 8))
        out.append(bytes(b))
        words = [rng.integers(97, 123, size=int(rng.integers(2, 9)), dtype=np.uint8).tobytes() for _ in range(200)]
        t = bytearray()
        while len(t) < n:
            t += words[int(rng.integers(0, 200)) if rng.random() < 0.9 else 0] + b" "
        out.append(bytes(t[:n]))
    if not quick:
        # more than 16 383 symbols per block, distances beyond the first window slide, a far length-3 match
        out.append(rng.integers(0, 16, size=200000, dtype=np.uint8).tobytes())
        out.append(bytes(rng.integers(0, 256, size=40000, dtype=np.uint8)) * 4)
        # the edges of zlib's window: sizes around wsize, wsize + MAX_DIST and 2 * wsize, matches at the largest distances
        for n in (32506, 32768, 65274, 65275, 65536, 65537, 98042, 131077):
            out.append(rng.integers(0, 3, size=n, dtype=np.uint8).tobytes())
        for per in (32505, 32506, 32507, 32768, 258, 3, 1):
            base = rng.integers(0, 256, size=per, dtype=np.uint8).tobytes()
            b = bytearray((base * (140000 // per + 2))[:140000])
            for _ in range(5):
                b[int(rng.integers(0, len(b)))] ^= 0x55
            out.append(bytes(b))
    return out


def run(quick=False, small=False):
    """small: the GPU test's set (one lane walks level 6's hash chains: seconds per 20 KB of short-alphabet data)"""
    rng = np.random.Generator(np.random.PCG64(77))
    eng = ea.Engine(0)
    n_c = n_d = 0
    for data in samples(rng, quick, small):
        for op in (0, 1, 2):
            got = eng.selftest_zlib(op, data)
            want = want_compress(op, data)
            assert got == want, "deflate op %d differs on %d bytes: len %s vs %d, first diff %d" % (op, len(data), None if got is None else len(got), len(want), util.first_diff(got or b"", want))
            n_c += 1
        if len(data) > 150000:
            continue
        gz, zl = want_compress(1, data), want_compress(2, data)
        variants = [(4, gz), (5, zl), (4, gz + b"trailing garbage"), (5, zl + b"xyz"), (4, zl), (5, gz), (4, data[:64]), (5, data[:64])]
        # OTP >= 20.1: concatenated members (all decoded), a member + an unfinished member / a stray byte / a member with a wrong CRC
        gz2 = want_compress(1, data[::-1][:777])
        variants += [(4, gz + gz), (4, gz + gz2 + gz), (4, gz + gz2[:len(gz2) // 2]), (4, gz + b"\x1f"), (4, gz + b"\x1f\x8b"), (4, gz + gz2[:-5] + b"\x00" + gz2[-4:]),
                     (4, gz + b"\x00"), (5, zl + zl)]
        # other encoders' streams: stored blocks, fixed trees, level 1 / 9, with a header name
        for lvl, strat in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY)):
            c = zlib.compressobj(lvl, zlib.DEFLATED, 31, 8, strat); variants.append((4, c.compress(data) + c.flush()))
            c = zlib.compressobj(lvl, zlib.DEFLATED, 15, 8, strat); variants.append((5, c.compress(data) + c.flush()))
        named = b"\x1f\x8b\x08\x08" + bytes(6) + b"name.txt\x00" + want_compress(0, data) + zlib.crc32(data).to_bytes(4, "little") + (len(data) & 0xffffffff).to_bytes(4, "little")
        variants.append((4, named))
        cd = zlib.compressobj(6, zlib.DEFLATED, 15, 8, zlib.Z_DEFAULT_STRATEGY, zdict=b"a preset dictionary")     # FDICT: {need_dictionary, _} once the id is there
        withdict = cd.compress(data) + cd.flush()
        variants += [(5, withdict), (5, withdict[:5]), (5, withdict[:2]), (5, withdict[:1])]
        for k in range(6 if small else 10 if quick else 24):                                             # truncations and corruptions of both formats
            for op, full in ((4, gz), (5, zl)):
                cut = int(rng.integers(0, len(full) + 1))
                variants.append((op, full[:cut]))
                if len(full) > 2:
                    b = bytearray(full); b[int(rng.integers(2, len(full)))] ^= 1 << int(rng.integers(0, 8))
                    variants.append((op, bytes(b)))
        for op, blob in variants:
            got = eng.selftest_zlib(op, blob, cap=max(4 * len(data) + 70000, 1 << 16))
            want = want_gunzip(blob) if op == 4 else want_inflate(blob)
            assert got == want, "inflate op %d differs on a %d-byte input (%d bytes plain): %s vs %s" % (
                op, len(blob), len(data), None if got is None else len(got), None if want is None else len(want))
            n_d += 1
    eng.close()
    return n_c, n_d


if __name__ == "__main__":
    print("zlib ok: %d streams compressed byte for byte, %d decoder cases" % run(quick=len(sys.argv) > 1 and sys.argv[1] == "quick"))
