#!/usr/bin/env python3
"""EH_PROF breakdown of given single cases of the bench corpus.  usage: profile_case.py MUTS|default CASE [CASE ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth
muts = None if sys.argv[1] == "default" else sys.argv[1]
mat = synth.mixed(65536, 4096)
data, off = synth.as_arena(mat)
eng = ea.Engine(0)
eng.configure(mutations=muts, patterns="od,nd,bu", out_capacity=4 << 30, max_slots=64)
eng.upload_corpus(data, off)
names = [m[0] for m in ea.mutator_table()]
for i in [int(x) for x in sys.argv[2:]]:
    eng.fuzz_batch(seed=(1, 2, 3), first_case=i + 1, corpus_first=i, n=1)
    outs, st = eng.download()
    pr = eng.prof().astype(np.float64)
    dr, lm = eng.diag()
    print("case %d: %.1f Mcyc, out %d B, status %d, draws %d, input byte0=%d" % (i, eng.cycles()[0] / 1e6, len(outs[0]), st[0], dr[0], mat[i][0]))
    for m in range(len(names)):
        if pr[2 * m + 1] > 0:
            print("    %-6s calls %5d  total %9.2f Mcyc  mean %9.1f kcyc" % (names[m], pr[2 * m + 1], pr[2 * m] / 1e6, pr[2 * m] / pr[2 * m + 1] / 1e3))
    for k in range(64, 128):
        if pr[2 * k + 1] > 0 and k not in range(70, 90):
            print("    slot %3d calls %7d total %9.2f Mcyc mean %9.1f kcyc" % (k, pr[2 * k + 1], pr[2 * k] / 1e6, pr[2 * k] / pr[2 * k + 1] / 1e3))
