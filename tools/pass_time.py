#!/usr/bin/env python3
"""One untimed + one timed pass of the bench workload over the first N corpus rows: wall time, kernel time, statuses,
output bytes, for a given (case MiB, big MiB, work budget MiB).  usage: pass_time.py N CASE_MIB BIG_MIB [WORK_MIB] [MUTS]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth
n, case_mib, big_mib = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
work = int(sys.argv[4]) if len(sys.argv) > 4 else 0
muts = sys.argv[5] if len(sys.argv) > 5 and sys.argv[5] != "default" else None
mat = synth.mixed(65536, 4096)[:n]
data, off = synth.as_arena(mat)
eng = ea.Engine(0)
eng.configure(mutations=muts, patterns="od,nd,bu", out_capacity=32 << 30, max_case_bytes=case_mib << 20, big_case_bytes=big_mib << 20, max_case_work=work << 20, max_slots=int(os.environ.get("MAX_SLOTS", "0")))
eng.upload_corpus(data, off)
for rep in range(2):
    t = time.time()
    eng.fuzz_batch(seed=(1, 2, 3), first_case=1 + rep * n, corpus_first=0, n=n)
    eng.sync()
    dt = time.time() - t
    st = eng.status()
    _, ob, _ = eng.totals()
    cyc = eng.cycles().astype(np.float64)
    print("pass %d: n %d case %d MiB big %d MiB work %d MiB: wall %.2f s kernel %.1f ms status %s out %.2f GB total Gcyc %.0f max Mcyc %.0f p50 %.1f p99 %.0f Mcyc" % (
        rep, n, case_mib, big_mib, work, dt, eng.kernel_ms(), np.bincount(st, minlength=6).tolist(), ob / 1e9, cyc.sum() / 1e9, cyc.max() / 1e6, np.median(cyc) / 1e6, np.percentile(cyc, 99) / 1e6), flush=True)
    dr, lm = eng.diag()
    ov = lm[st == 2]
    if len(ov): print("    overflow sites:", dict(zip(*[x.tolist() for x in np.unique(-ov, return_counts=True)])), flush=True)
