#!/usr/bin/env python3
"""EH_PROF breakdown of single cases (the slowest ones of a batch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth
n = 16384
mat = synth.mixed(n, 4096)
data, off = synth.as_arena(mat)
eng = ea.Engine(0)
muts = ",".join(ea.gpu_mutators())
eng.configure(mutations=muts, patterns="od,nd,bu", out_capacity=8 << 30)
eng.upload_corpus(data, off)
eng.fuzz_batch(seed=(1, 2, 3))
cyc = eng.cycles().astype(np.float64)
names = [m[0] for m in ea.mutator_table()]
for i in np.argsort(-cyc)[:4]:
    eng.fuzz_batch(seed=(1, 2, 3), first_case=int(i) + 1, corpus_first=int(i), n=1)
    outs, st = eng.download()
    pr = eng.prof().astype(np.float64)
    print("case %d: %.1f Mcyc (batch run), out %d B, status %d, input kind byte0=%d" % (i, cyc[i] / 1e6, len(outs[0]), st[0], mat[i][0]))
    for m in range(len(names)):
        if pr[2 * m + 1] > 0:
            print("    %-6s calls %5d  total %8.2f Mcyc  mean %8.1f kcyc" % (names[m], pr[2 * m + 1], pr[2 * m] / 1e6, pr[2 * m] / pr[2 * m + 1] / 1e3))
    for k, nm in enumerate(["setup", "generator", "pattern+mux", "output"]):
        j = 64 + k
        print("    phase %-12s %8.2f Mcyc" % (nm, pr[2 * j] / 1e6))
