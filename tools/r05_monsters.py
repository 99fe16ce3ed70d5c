#!/usr/bin/env python3
"""Which cases decide how long a pass lasts?  Runs the passes bench.py runs (pass k = case numbers k*65536+1 ..), one at a time on an
idle GPU, and lists the heaviest cases of each (wave cycles, draws, last mutator, output bytes).  usage: r05_monsters.py FIRST_PASS N_PASSES TOP OUT.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth
first, npass, top, outp = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
n = 65536
data, off = synth.as_arena(synth.mixed(n, 4096))
names = [m[0] for m in ea.mutator_table()]
eng = ea.Engine(0)
eng.configure(patterns="od,nd,bu", out_capacity=30 << 30, max_case_bytes=4 << 20, big_case_bytes=1024 << 20, pool_bytes=60 << 30)
eng.upload_corpus(data, off)
res = []
for k in range(first, first + npass):
    eng.fuzz_batch(seed=(1, 2, 3), first_case=k * n + 1, corpus_first=0, n=n)
    eng.sync()
    cyc = eng.cycles().astype(np.float64); st = eng.status(); dr, lm = eng.diag(); ln = eng.lens()
    order = np.argsort(-cyc)[:top]
    print("pass %d: kernel %.0f ms, total %.0f Gcyc, cases above 1 / 2 / 3 Gcyc: %d / %d / %d" % (k, eng.kernel_ms(), cyc.sum() / 1e9, (cyc > 1e9).sum(), (cyc > 2e9).sum(), (cyc > 3e9).sum()), flush=True)
    for i in order:
        print("    case %d (+%d): %.0f Mcyc status %d draws %d last %s out %d" % (k * n + 1 + i, i, cyc[i] / 1e6, st[i], dr[i], names[lm[i]] if 0 <= lm[i] < len(names) else str(lm[i]), ln[i]), flush=True)
        res.append({"pass": k, "row": int(i), "mcyc": float(cyc[i] / 1e6), "draws": int(dr[i]), "last": int(lm[i]), "out": int(ln[i]), "status": int(st[i])})
    json.dump(res, open(outp, "w"))
