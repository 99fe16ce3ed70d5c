#!/usr/bin/env python3
"""Re-run single cases of the bench workload alone (idle GPU) with the EH_PROF build: per-mutator and per-phase cycles.
usage: ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so tools/profile_alone.py BASE CASE [CASE ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth
# "@file": lines of "BASE ROW" (tools/r05_monsters.sh); otherwise BASE CASE [CASE ...]
if sys.argv[1].startswith("@"):
    jobs = [tuple(int(x) for x in ln.split()) for ln in open(sys.argv[1][1:]) if ln.strip()]
else:
    jobs = [(int(sys.argv[1]), int(x)) for x in sys.argv[2:]]
mat = synth.mixed(65536, 4096)
data, off = synth.as_arena(mat)
names = [m[0] for m in ea.mutator_table()]
eng = ea.Engine(0)
eng.configure(fuse_stream_min=int(os.environ.get("FUSE_STREAM_MIN", "0")), patterns=(os.environ.get("PATTERNS", "od,nd,bu") if os.environ.get("PATTERNS") != "default" else None), out_capacity=4 << 30, max_case_bytes=16 << 20, big_case_bytes=1024 << 20, max_slots=8)
eng.upload_corpus(data, off)
for base, i in jobs:
    eng.fuzz_batch(seed=(1, 2, 3), first_case=base + 1 + i, corpus_first=i, n=1)
    eng.sync()
    pr = eng.prof().astype(np.float64)
    _, ob, _ = eng.totals()
    dr, lm = eng.diag()
    print("case %d (base %d) alone: %.0f Mcyc, out %d B, status %d, draws %d, kernel %.1f ms" % (i, base, eng.cycles()[0] / 1e6, ob, eng.status()[0], dr[0], eng.kernel_ms()))
    for m in range(len(names)):
        if pr[2 * m + 1] > 0 and pr[2 * m] > 10e6:
            print("    %-6s calls %6d  total %9.1f Mcyc  mean %9.1f kcyc" % (names[m], pr[2 * m + 1], pr[2 * m] / 1e6, pr[2 * m] / pr[2 * m + 1] / 1e3))
    for k in range(len(names), 128):
        if pr[2 * k + 1] > 0 and pr[2 * k] > 10e6:
            print("    slot %3d calls %7d total %9.1f Mcyc mean %9.1f kcyc" % (k, pr[2 * k + 1], pr[2 * k] / 1e6, pr[2 * k] / pr[2 * k + 1] / 1e3))
    print("    sgm tokenizer phase 2: between attempts %.1f Mcyc x%d, failed attempts %.1f Mcyc x%d (mean %.1f kcyc), accepted tags %.1f Mcyc x%d (mean %.1f kcyc)" % (
        pr[2 * 86] / 1e6, pr[2 * 86 + 1], pr[2 * 87] / 1e6, pr[2 * 87 + 1], pr[2 * 87] / max(pr[2 * 87 + 1], 1) / 1e3, pr[2 * 88] / 1e6, pr[2 * 88 + 1], pr[2 * 88] / max(pr[2 * 88 + 1], 1) / 1e3))
    print("    sgm replay: large documents %d, periodic %d, replays %d (tokens %d), checked but not replayed %d (look-ahead too long %d), gave up aligning %d" % (
        pr[2 * 95 + 1], pr[2 * 95], pr[2 * 94 + 1], pr[2 * 94], pr[2 * 99 + 1], pr[2 * 99], pr[2 * 89 + 1]))
    sys.stdout.flush()
