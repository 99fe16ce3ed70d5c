#!/usr/bin/env python3
"""erlamsa_fuse:fuse/2 on LARGE pumped blocks, three ways: the engine with the periodic stretches cut short (csrc/eh_fuse_red.h,
the default), the engine searching the lists as they are (EH_FLAG_FUSE_NO_REDUCE) and the oracle.  The corpus is what sr / lr /
tr leave behind: a head, a piece repeated many times, a tail - plain, nested, two such stretches in one block, stretches barely
long enough to be cut, long periods, a period of one byte - 9 KB to ~150 KB.  Bytes, statuses and PRNG draw counts must agree.

  python tests/hipemu/build_emu.py && ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_fuse_red.py [n_per_kind] [seed]
(with ERLAMSA_HIP_LIB unset or pointing at the real library the same comparison runs on the GPU)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import erlamsa_amd as ea
from erlamsa_amd.engine import EH_FLAG_FUSE_NO_REDUCE


def corpus(n, seed, big=1):
    rng = np.random.Generator(np.random.PCG64(seed))

    def rb(lo, hi, alpha=256):
        return rng.integers(0, alpha, size=int(rng.integers(lo, hi)), dtype=np.uint8).tobytes()
    out = []
    for k in range(n):
        out.append(rb(0, 3000) + rb(10, 900) * int(rng.integers(12, 120 * big)) + rb(0, 3000))                       # head, piece x N, tail
        out.append(rb(0, 500, 4) + rb(3, 40, 4) * int(rng.integers(300, 1500 * big)) + rb(0, 500, 4))                 # short period, small alphabet
        inner = rb(5, 60) * int(rng.integers(5, 40))
        out.append(rb(0, 800) + (inner + rb(10, 300)) * int(rng.integers(6, 30 * big)) + rb(0, 800))                  # nested
        out.append(rb(0, 400) + rb(20, 700) * int(rng.integers(10, 60 * big)) + rb(50, 2000) + rb(20, 700) * int(rng.integers(10, 60 * big)) + rb(0, 400))   # two stretches
        p = rb(700, 2500)
        out.append(rb(0, 2000) + p * 5 + p[:int(rng.integers(0, len(p)))] + rb(0, 2000))                               # barely worth cutting (or not)
        out.append(rb(0, 100) + bytes([int(rng.integers(0, 256))]) * int(rng.integers(9000, 40000 * big)) + rb(0, 100))   # one byte
        line = b"".join(bytes([int(x)]) for x in rng.integers(97, 123, size=int(rng.integers(5, 70)))) + b"\n"
        out.append(b"head\n" * int(rng.integers(0, 50)) + line * int(rng.integers(200, 2500 * big)) + b"tail 12345\n")   # repeated line
        out.append(rb(9000, 30000))                                                                                     # nothing periodic
    return out


def run(n=1, seed=1, muts="ft,fn,fo", pats="od,nd,bu", big=1, verbose=True, with_oracle=True):
    inputs = corpus(n, seed, big)
    data, off = po.pack(inputs)
    res = {}
    for name, flags in (("cut", 0), ("plain", EH_FLAG_FUSE_NO_REDUCE)):
        t = time.time()
        e = ea.Engine(0)
        e.configure(mutations=muts, patterns=pats, max_case_bytes=64 << 20, flags=flags)
        e.upload_corpus(data, off)
        e.fuzz_batch(seed=(seed, 3, 5))
        got, st = e.download()
        dr, _ = e.diag()
        res[name] = (got, st, dr, time.time() - t)
        e.close()
    want = wst = wdr = None
    t = time.time()
    if with_oracle:
        import util
        o = util.oracle_live(data, off, seed=(seed, 3, 5), mutations=muts, patterns=pats, max_case_bytes=64 << 20, chunk=1)
        want, wst, wdr = o.outs, o.status, o.draws
    to = time.time() - t
    bad = 0
    a, b = res["cut"], res["plain"]
    for i in range(len(inputs)):
        if a[1][i] in (2, 3) or (with_oracle and wst[i] in (2, 3)):
            continue
        ok_o = (not with_oracle) or (a[0][i] == want[i] and a[1][i] == wst[i] and (a[1][i] != 0 or a[2][i] == wdr[i]))
        ok_n = a[0][i] == b[0][i] and a[1][i] == b[1][i] and a[2][i] == b[2][i]
        if not (ok_o and ok_n):
            bad += 1
            if verbose and bad <= 8:
                print("case %d (kind %d, len %d): cut vs oracle %s, cut vs plain %s; status %d/%d draws %d/%d len %d/%d" % (
                    i, i % 8, len(inputs[i]), ok_o, ok_n, a[1][i], b[1][i], a[2][i], b[2][i], len(a[0][i]), len(b[0][i])))
    if verbose:
        print("cases %d bad %d; cut %.1f s, plain %.1f s, oracle %.1f s; input bytes %d" % (len(inputs), bad, a[3], b[3], to, sum(map(len, inputs))))
    return len(inputs), bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    total, bad = run(n, seed, big=big)
    sys.exit(1 if bad else 0)
