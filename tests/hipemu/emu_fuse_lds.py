#!/usr/bin/env python3
"""erlamsa_fuse:fuse/2 three ways on the CPU wavefront emulator (or, with the real library, on the GPU): the LDS-resident refinement
(csrc/eh_fuse_lds.h, the default for small lists), the node-list refinement (csrc/eh_fuse.h, EH_FLAG_FUSE_NO_LDS) and the oracle.
Corpora that aim at the corners: tiny blocks (every round retires the member whose rest is []), blocks of one repeated byte or a
short period (nodes never split: the big-node path every round), random and text blocks of a few KB (one big node, then nodes of
a few members, then single members), pairs of blocks with little in common (NoDesp =:= [] ends the search), blocks around the
limits of the LDS path.  Compares bytes, statuses and PRNG draw counts.

  python tests/hipemu/build_emu.py && ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_fuse_lds.py [n_per_kind] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import erlamsa_amd as ea
from erlamsa_amd.engine import EH_FLAG_FUSE_NO_LDS


def corpus(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    words = [b"alpha", b"beta", b"gamma", b"delta", b"12345", b"<a>", b"</a>", b"\n", b" ", b"foo=bar;", b"AAAA"]
    for k in range(n):
        out.append(rng.integers(0, 256, size=int(rng.integers(1, 24)), dtype=np.uint8).tobytes())                  # tiny
        out.append(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 300)))                                   # one byte repeated
        per = rng.integers(0, 256, size=int(rng.integers(2, 9)), dtype=np.uint8).tobytes()
        out.append((per * 600)[:int(rng.integers(50, 3000))])                                                         # short period
        out.append(rng.integers(0, 256, size=int(rng.integers(100, 4200)), dtype=np.uint8).tobytes())              # random
        out.append(rng.integers(0, 4, size=int(rng.integers(100, 4200)), dtype=np.uint8).tobytes())                # 4-letter alphabet
        out.append(b"".join(words[int(j)] for j in rng.integers(0, len(words), size=int(rng.integers(5, 900))))[:4300])   # text
        base = rng.integers(0, 256, size=int(rng.integers(20, 200)), dtype=np.uint8).tobytes()
        out.append(base + base[::-1] + base * int(rng.integers(1, 20)))                                              # pumped
        out.append(rng.integers(65, 70, size=int(rng.integers(4000, 8300)), dtype=np.uint8).tobytes())             # near the limits of the LDS path
    return out


def run(n=6, seed=1, muts="ft,fn,fo", pats="od,nd,bu", verbose=True):
    inputs = corpus(n, seed)
    data, off = po.pack(inputs)
    res = {}
    for name, flags in (("lds", 0), ("nodes", EH_FLAG_FUSE_NO_LDS)):
        t = time.time()
        e = ea.Engine(0)
        e.configure(mutations=muts, patterns=pats, max_case_bytes=4 << 20, flags=flags)
        e.upload_corpus(data, off)
        e.fuzz_batch(seed=(seed, 7, 9))
        got, st = e.download()
        dr, _ = e.diag()
        res[name] = (got, st, dr, time.time() - t)
        e.close()
    want, wst, wdr, _ = po.fuzz_batch(data, off, seed=(seed, 7, 9), mutations=muts, patterns=pats, max_case_bytes=4 << 20)
    bad = 0
    for i in range(len(inputs)):
        a, b = res["lds"], res["nodes"]
        if a[1][i] in (2, 3) or wst[i] in (2, 3):
            continue
        ok_o = a[0][i] == want[i] and a[1][i] == wst[i] and (a[1][i] != 0 or a[2][i] == wdr[i])
        ok_n = a[0][i] == b[0][i] and a[1][i] == b[1][i] and a[2][i] == b[2][i]
        if not (ok_o and ok_n):
            bad += 1
            if verbose and bad <= 8:
                print("case %d (kind %d, len %d): lds vs oracle %s, lds vs nodes %s; status %d/%d/%d draws %d/%d/%d len %d/%d/%d" % (
                    i, i % 8, len(inputs[i]), ok_o, ok_n, a[1][i], b[1][i], wst[i], a[2][i], b[2][i], wdr[i], len(a[0][i]), len(b[0][i]), len(want[i])))
    if verbose:
        print("cases %d bad %d; lds %.1f s, nodes %.1f s" % (len(inputs), bad, res["lds"][3], res["nodes"][3]))
    return len(inputs), bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    total, bad = run(n, seed)
    sys.exit(1 if bad else 0)
