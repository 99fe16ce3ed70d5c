#!/usr/bin/env python3
"""EH_PROF cost of fuse (ft / fn) on synthetic blocks of one size and three kinds (random, text, periodic).
usage: ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so [FUSE_STREAM_MIN=..] [FUSE_NO_LDS=1] tools/profile_fuse.py SIZE [N] [MUT]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
size = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 64; mut = sys.argv[3] if len(sys.argv) > 3 else "ft"
rng = np.random.Generator(np.random.PCG64(5))
kinds = {"random": rng.integers(0, 256, size=(n, size), dtype=np.uint8),
         "text": rng.integers(32, 127, size=(n, size), dtype=np.uint8)}
per = np.empty((n, size), dtype=np.uint8)
for i in range(n):
    P = int(rng.integers(200, 3000)); pat = rng.integers(0, 256, size=P, dtype=np.uint8)
    per[i] = np.tile(pat, size // P + 1)[:size]
kinds["periodic"] = per
names = [m[0] for m in ea.mutator_table()]
LAB_LDS = {100: "round1", 101: "rounds", 102: "rounds(single)", 103: "select"}
LAB = {97: "cuts", 98: "no-cut", 100: "alloc", 101: "bits0", 102: "etest", 103: "scan", 104: "clear", 105: "pass(lds)", 106: "pass(glob)", 107: "pass(last)", 108: "tail", 109: "select", 110: "pass(test)"}
for kind, mat in kinds.items():
    data = np.ascontiguousarray(mat).reshape(-1); off = np.arange(n + 1, dtype=np.uint64) * np.uint64(size)
    eng = ea.Engine(0)
    eng.configure(fuse_stream_min=int(os.environ.get("FUSE_STREAM_MIN", "0")), mutations=mut + ",nil=0", patterns="od", out_capacity=4 << 30, max_case_bytes=256 << 20, big_case_bytes=256 << 20, max_slots=n, flags=(4 if os.environ.get("FUSE_NO_LDS") == "1" else 0) | (8 if os.environ.get("FUSE_NO_REDUCE") == "1" else 0))
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=(1, 2, 3))
    eng.sync()
    pr = eng.prof().astype(np.float64)
    m = names.index(mut)
    print("%-5s %-9s size %8d calls %5d mean %10.1f kcyc kernel %8.2f ms rounds/call %.2f" % (mut, kind, size, pr[2 * m + 1], pr[2 * m] / max(pr[2 * m + 1], 1) / 1e3, eng.kernel_ms(), pr[2 * 126] / max(pr[2 * 126 + 1], 1)))
    lab = LAB_LDS if (size <= 8192 and os.environ.get("FUSE_NO_LDS") != "1") else LAB
    print("      " + "  ".join("%s %.0fk x%d" % (lab.get(k, str(k)), pr[2 * k] / max(pr[2 * k + 1], 1) / 1e3, pr[2 * k + 1]) for k in list(range(96, 99)) + list(range(100, 112)) if pr[2 * k + 1] > 0))
    eng.close()
