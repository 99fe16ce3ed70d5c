#!/usr/bin/env python3
"""EH_PROF cost of single mutators on the fresh 4 KiB corpus rows (pattern od: one scheduler call per case), by
corpus kind.  usage: ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so python tools/profile_muta.py ft sgm b64 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth

n, size = 2048, 4096
rng = np.random.Generator(np.random.PCG64(7))
kinds = {"uniform": rng.integers(0, 256, size=(n, size), dtype=np.uint8), "ascii-lines": synth._ascii_lines(rng, n, size),
         "bracketed": synth._bracketed(rng, n, size), "framed": synth._framed(rng, 256, size)}
names = [m[0] for m in ea.mutator_table()]
for mut in sys.argv[1:]:
    for kind, mat in kinds.items():
        data, off = synth.as_arena(mat)
        eng = ea.Engine(0)
        eng.configure(mutations=mut + ",nil=0", patterns="od", out_capacity=2 << 30)
        eng.upload_corpus(data, off)
        eng.fuzz_batch(seed=(1, 2, 3))
        eng.sync()
        pr = eng.prof().astype(np.float64)
        m = names.index(mut)
        calls = pr[2 * m + 1]
        extra = "  ".join("p%d %.0fk/%d" % (k, pr[2 * k] / max(pr[2 * k + 1], 1) / 1e3, pr[2 * k + 1]) for k in range(90, 110) if pr[2 * k + 1] > 0)
        print("%-5s %-12s calls %6d  mean %9.1f kcyc  kernel %.2f ms  %s" % (mut, kind, calls, pr[2 * m] / max(calls, 1) / 1e3, eng.kernel_ms(), extra))
        eng.close()
