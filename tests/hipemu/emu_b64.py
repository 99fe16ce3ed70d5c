#!/usr/bin/env python3
"""base64_mutator/2 (erlamsa_mutations.erl:658-690) on text full of base64: the wave-parallel decode (csrc/eh_lex.h b64_decode_wave:
groups of four per lane, chunks with white space inside packed first) against the oracle.  Chunks: every tail ("", "xx==", "xxx="),
white space inside groups, between the padding characters and behind them, long blobs, lines of them repeated many times, chunks
base64:decode/1 refuses (bad length, padding in the wrong place, text behind the padding).  Bytes, statuses, draw counts must agree.

  python tests/hipemu/build_emu.py && ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_b64.py [n] [seed] [scale] [small]
(with the real library the same comparison runs on the GPU; scale multiplies blob sizes and repeat counts)"""
import base64
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import erlamsa_amd as ea


def sprinkle(rng, b, p):
    """white space inside the chunk with probability p per position"""
    out = bytearray()
    for ch in b:
        while rng.random() < p:
            out += rng.choice([b" ", b"\t", b"\n", b"\r", b"\r\n", b"  "])
        out.append(ch)
    return bytes(out)


def corpus(n, seed, scale=1, small=False):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for k in range(n):
        def blob(lo, hi):
            return bytes(rng.integers(0, 256, size=int(rng.integers(lo, hi)), dtype=np.uint8))
        pieces = []
        for j in range(int(rng.integers(1, 7))):
            e = base64.b64encode(blob(3, 60))
            kind = int(rng.integers(0, 10))
            if kind == 0: e = sprinkle(rng, e, 0.2)
            elif kind == 1 and e.endswith(b"=="): e = e[:-1] + b" \n=" + b" \t"
            elif kind == 2: e = e + b"  \n"
            elif kind == 3: e = e[:-1]                                 # bad length
            elif kind == 4 and b"=" in e: e = e + b"QQ"                # text behind the padding
            elif kind == 5: e = e[:5] + b"=" + e[5:]                   # padding in the middle
            elif kind == 6: e = sprinkle(rng, base64.b64encode(blob(200, 3000 * scale)), 0.02)
            pieces.append(e)
            pieces.append(rng.choice([b"\x00", b"; ", b"\x01\x02", b"|", b"{", b"\xff"]))
        out.append(b"".join(pieces))
        line = base64.b64encode(blob(30, 90)) + b"\n"
        out.append(b"-----BEGIN-----\x00" + line * (int(rng.integers(10, 40) if small else rng.integers(20, 200)) * scale) + b"\x00-----END-----")       # one chunk of many lines
        out.append(b"\x00".join(base64.b64encode(blob(6, 40)) for _ in range(int(rng.integers(70, 110) if small else rng.integers(70, 300)))))             # more than 64 candidates
        out.append(sprinkle(rng, base64.b64encode(blob(400, 3000) if small else blob(1000 * scale, 20000 * scale)), float(rng.choice([0.0, 0.0, 0.01, 0.3]))))
    return out


def run(n=4, seed=1, scale=1, pats="od,nd,bu", verbose=True, small=False):
    inputs = corpus(n, seed, scale, small)
    data, off = po.pack(inputs)
    t = time.time()
    e = ea.Engine(0)
    e.configure(mutations="b64", patterns=pats, max_case_bytes=64 << 20)
    e.upload_corpus(data, off)
    e.fuzz_batch(seed=(seed, 3, 9))
    got, st = e.download()
    dr, _ = e.diag()
    te = time.time() - t
    e.close()
    import util
    t = time.time()
    o = util.oracle_live(data, off, seed=(seed, 3, 9), mutations="b64", patterns=pats, max_case_bytes=256 << 20, chunk=1)
    to = time.time() - t
    bad = changed = 0
    for i in range(len(inputs)):
        if st[i] in (2, 3) or o.status[i] in (2, 3):
            continue
        changed += got[i] != inputs[i]
        if not (got[i] == o.outs[i] and st[i] == o.status[i] and (st[i] != 0 or dr[i] == o.draws[i])):
            bad += 1
            if verbose and bad <= 8:
                print("case %d (kind %d, len %d): status %d/%d draws %d/%d len %d/%d" % (i, i % 4, len(inputs[i]), st[i], o.status[i], dr[i], o.draws[i], len(got[i]), len(o.outs[i])))
    if verbose:
        print("cases %d bad %d (outputs that differ from their input: %d); engine %.1f s, oracle %.1f s; input bytes %d" % (len(inputs), bad, changed, te, to, sum(map(len, inputs))))
    return len(inputs), bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    total, bad = run(n, seed, scale, small=len(sys.argv) > 4 and sys.argv[4] == "small")   # small: the CPU suite's run on the emulator
    sys.exit(1 if bad else 0)
