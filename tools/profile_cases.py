#!/usr/bin/env python3
"""Per-case cost breakdown on the GPU: which mutators dominate the wavefront time."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
muts = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "default" else None   # None = the full default table
pats = sys.argv[3] if len(sys.argv) > 3 else "od,nd,bu"
mat = synth.mixed(n, 4096)
data, off = synth.as_arena(mat)
eng = ea.Engine(0)
import os
eng.configure(mutations=muts, patterns=pats, out_capacity=16 << 30, max_case_work=int(os.environ.get("WORK_MIB", "4")) << 20, max_case_bytes=int(os.environ.get("CASE_MIB", "8")) << 20, big_case_bytes=int(os.environ.get("BIG_MIB", "0")) << 20)
eng.upload_corpus(data, off)
eng.fuzz_batch(seed=(1, 2, 3))
outs, st = eng.download()
cyc = eng.cycles().astype(np.float64)
dr, lm = eng.diag()
names = [m[0] for m in ea.mutator_table()]
print("kernel ms", eng.kernel_ms(), "cases", n, "total Mcycles", cyc.sum() / 1e6, "max Mcycles", cyc.max() / 1e6, "median kcycles", np.median(cyc) / 1e3)
print("status counts", np.bincount(st, minlength=6), "output MB", sum(map(len, outs)) / 1e6)
order = np.argsort(-cyc)
print("top cases:")
for i in order[:12]:
    print("  case %d: %.2f Mcyc, out %d B, draws %d, last %s, status %d" % (i, cyc[i] / 1e6, len(outs[i]), dr[i], names[lm[i]] if lm[i] >= 0 else "-", st[i]))
print("by last mutator (count, mean kcyc, share of total cycles):")
for m in range(len(names)):
    sel = lm == m
    if sel.any():
        print("  %-6s %6d %10.1f %6.1f%%" % (names[m], sel.sum(), cyc[sel].mean() / 1e3, 100 * cyc[sel].sum() / cyc.sum()))
pct = np.percentile(cyc, [50, 90, 99, 99.9])
print("percentiles kcyc 50/90/99/99.9:", (pct / 1e3).round(1))
pr = eng.prof().astype(np.float64)
if pr.sum() > 0:
    print("EH_PROF per mutator attempt (calls, mean kcyc, total Mcyc):")
    for m in range(len(names)):
        if pr[2 * m + 1] > 0:
            print("  %-6s %8d %10.1f %10.1f" % (names[m], pr[2 * m + 1], pr[2 * m] / pr[2 * m + 1] / 1e3, pr[2 * m] / 1e6))
    for k, nm in enumerate(["tree:binarish", "tree:parse", "ts:sample", "ts1:match", "ts1:emit", "tree:sum L", "tree:sum N", "ts1:sum nm", "-", "-", "parse:counts", "parse:collect", "parse:match loop", "parse:level end", "parse:compact", "parse:sum events", "parse:sum slots"]):
        i = 70 + k
        if pr[2 * i + 1] > 0:
            print("  %-14s %8d %10.1f %10.1f" % (nm, pr[2 * i + 1], pr[2 * i] / pr[2 * i + 1] / 1e3, pr[2 * i] / 1e6))
    for k, nm in enumerate(["setup", "generator", "pattern+mux", "output"]):
        i = 64 + k
        if pr[2 * i + 1] > 0:
            print("  phase %-12s %8d %10.1f %10.1f" % (nm, pr[2 * i + 1], pr[2 * i] / pr[2 * i + 1] / 1e3, pr[2 * i] / 1e6))
