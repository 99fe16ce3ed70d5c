#!/usr/bin/env python3
"""Run one pass of the bench workload (case ids BASE+1 ..), list the TOP slowest cases and re-run each alone with the
EH_PROF build's per-mutator breakdown.  usage: ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so find_monsters.py BASE TOP [WORK_MIB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth
base, top = int(sys.argv[1]), int(sys.argv[2])
work = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = 65536
mat = synth.mixed(n, 4096)
data, off = synth.as_arena(mat)
names = [m[0] for m in ea.mutator_table()]
eng = ea.Engine(0)
eng.configure(patterns="od,nd,bu", out_capacity=32 << 30, max_case_bytes=16 << 20, big_case_bytes=1024 << 20, max_case_work=work << 20)
eng.upload_corpus(data, off)
eng.fuzz_batch(seed=(1, 2, 3), first_case=base + 1, corpus_first=0, n=n)
eng.sync()
cyc = eng.cycles().astype(np.float64)
st = eng.status()
dr, lm = eng.diag()
order = np.argsort(-cyc)[:top]
print("kernel %.1f ms; top cases:" % eng.kernel_ms())
for i in order:
    print("  case %d: %.0f Mcyc status %d draws %d last %s" % (i, cyc[i] / 1e6, st[i], dr[i], names[lm[i]] if 0 <= lm[i] < len(names) else "?"), flush=True)
eng.close()
eng = ea.Engine(0)
eng.configure(patterns="od,nd,bu", out_capacity=4 << 30, max_case_bytes=16 << 20, big_case_bytes=1024 << 20, max_case_work=work << 20, max_slots=8)
eng.upload_corpus(data, off)
for i in order:
    i = int(i)
    eng.fuzz_batch(seed=(1, 2, 3), first_case=base + 1 + i, corpus_first=i, n=1)
    eng.sync()
    pr = eng.prof().astype(np.float64)
    _, ob, _ = eng.totals()
    print("case %d alone: %.0f Mcyc, out %d B, status %d" % (i, eng.cycles()[0] / 1e6, ob, eng.status()[0]))
    for m in range(len(names)):
        if pr[2 * m + 1] > 0 and pr[2 * m] > 20e6:
            print("    %-6s calls %6d  total %9.1f Mcyc  mean %9.1f kcyc" % (names[m], pr[2 * m + 1], pr[2 * m] / 1e6, pr[2 * m] / pr[2 * m + 1] / 1e3))
    for k in range(64, 128):
        if pr[2 * k + 1] > 0 and pr[2 * k] > 20e6:
            print("    slot %3d calls %7d total %9.1f Mcyc mean %9.1f kcyc" % (k, pr[2 * k + 1], pr[2 * k] / 1e6, pr[2 * k] / pr[2 * k + 1] / 1e3))
    sys.stdout.flush()
