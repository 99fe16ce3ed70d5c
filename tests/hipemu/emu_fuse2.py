#!/usr/bin/env python3
"""eh_fuse2.h (position-indexed refinement, two-level next-byte tables, compact lookups) on the emulator against the oracle:
every fuse call goes the streaming way (fuse_stream_min = 64) on inputs made to reach each of its table modes — periodic
data with a few defects (hundreds of nodes with one continuation each: N1 entries, a few escalate to bitmap rows; compact
lookups), text (several continuations per node: bitmap rows, lookups through full rows), random bytes (dense rows, fuel
exhausted after two or three rounds), and pairs of different blocks (fn / fo: the special node, members that leave).
Run with ERLAMSA_HIP_LIB=<emu lib>.  usage: emu_fuse2.py [N] [SIZE]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import util
import erlamsa_amd as ea

n = int(sys.argv[1]) if len(sys.argv) > 1 else 36
size = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
rng = np.random.Generator(np.random.PCG64(77))
inputs = []
for i in range(n):
    kind = i % 4
    if kind == 0:                                   # periodic with defects
        per = rng.integers(0, 256, size=int(rng.integers(40, 700)), dtype=np.uint8)
        b = np.tile(per, size // len(per) + 1)[:size].copy()
        for _ in range(int(rng.integers(0, 6))):
            b[int(rng.integers(0, size))] = rng.integers(0, 256)
        inputs.append(b.tobytes())
    elif kind == 1:                                 # text
        words = [b"alpha", b"beta", b"gamma", b"delta", b"12345", b"<tag>", b"</tag>", b"\n", b" ", b"=", b"\"q\""]
        out = b""
        while len(out) < size:
            out += words[int(rng.integers(0, len(words)))]
        inputs.append(out[:size])
    elif kind == 2:                                 # random bytes
        inputs.append(rng.integers(0, 256, size=size, dtype=np.uint8).tobytes())
    else:                                           # short alphabet
        inputs.append(rng.integers(97, 101, size=size, dtype=np.uint8).tobytes())
data, off = po.pack(inputs)
muts = "ft=3,fn=2,fo=2,sr=1,bd=1"
bad_total = 0
for pats, seed in (("od", (3, 1, 4)), ("nd,bu", (1, 5, 9))):
    want, wst, wdr, _ = po.fuzz_batch(data, off, seed=seed, mutations=muts, patterns=pats, max_case_bytes=256 << 20)
    eng = ea.Engine(0)
    eng.configure(mutations=muts, patterns=pats, max_case_bytes=1 << 20, big_case_bytes=256 << 20, fuse_stream_min=64)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=seed)
    got, gst = eng.download()
    gdr, _ = eng.diag()
    bad = [i for i in range(n) if gst[i] not in (2, 3) and wst[i] not in (2, 3) and (got[i] != want[i] or gst[i] != wst[i] or gdr[i] != wdr[i])]
    print("patterns %s: cases %d, mismatches %s, statuses %s" % (pats, n, bad, np.bincount(gst, minlength=6).tolist()))
    bad_total += len(bad)
    eng.close()
assert bad_total == 0
print("fuse2 ok")
