#!/usr/bin/env python3
"""One full pass of the bench workload (BASELINE configs[2], case ids BASE+1 .. BASE+65536) with the EH_PROF build:
per-case cycles / status / draws / last mutator / output length saved to gpurun_out/<tag>_cases.npz, the per-mutator
and per-size fuse totals printed, the heaviest TOP cases listed.
usage: ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so tools/survey_pass.py TAG [BASE] [TOP] [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth

tag = sys.argv[1]
base = int(sys.argv[2]) if len(sys.argv) > 2 else 0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 200
n = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
mat = synth.mixed(65536, 4096)[:n]
data, off = synth.as_arena(mat)
names = [m[0] for m in ea.mutator_table()]
eng = ea.Engine(0)
eng.configure(fuse_stream_min=int(os.environ.get("FUSE_STREAM_MIN", "0")), patterns=(os.environ.get("PATTERNS", "od,nd,bu") if os.environ.get("PATTERNS") != "default" else None), out_capacity=40 << 30, max_case_bytes=int(os.environ.get("CASE_MIB", "4")) << 20, big_case_bytes=1024 << 20,
              max_slots=int(os.environ.get("MAX_SLOTS", "0")), flags=int(os.environ.get("ENGINE_FLAGS", "0")))
eng.upload_corpus(data, off)
eng.fuzz_batch(seed=(1, 2, 3), first_case=base + 1, corpus_first=0, n=n)
eng.sync()
cyc = eng.cycles().astype(np.float64)
st = eng.status()
dr, lm = eng.diag()
_, ob, _ = eng.totals()
import ctypes as C
lens = np.zeros(n, dtype=np.uint64)
print("kernel %.1f ms, total %.1f Gcyc, max %.1f Mcyc, out %.2f GB, status %s" % (
    eng.kernel_ms(), cyc.sum() / 1e9, cyc.max() / 1e6, ob / 1e9, np.bincount(st, minlength=6).tolist()))
pk = eng.peak().astype(np.float64)
print("work memory high-water MiB 50/90/99/99.9/max:", (np.percentile(pk, [50, 90, 99, 99.9, 100]) / 2**20).round(2).tolist(),
      " cases above 1/2/4/8/16/32/64/256 MiB:", [int((pk > (m << 20)).sum()) for m in (1, 2, 4, 8, 16, 32, 64, 256)])
print("pool:", eng.pool_stats())
print("cooperative execution:", eng.coop_stats())
pct = np.percentile(cyc, [50, 90, 99, 99.9, 99.99])
print("percentiles Mcyc 50/90/99/99.9/99.99:", (pct / 1e6).round(2).tolist())
srt = np.sort(cyc)[::-1]
for k in (1, 10, 100, 1000, 4096):
    print("  sum of the %d heaviest: %.1f Gcyc (%.1f%%)" % (k, srt[:k].sum() / 1e9, 100 * srt[:k].sum() / cyc.sum()))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", tag + "_cases.npz"), cycles=cyc, status=st, draws=dr, lastm=lm, peak=pk)
order = np.argsort(-cyc)[:top]
print("top cases:")
for i in order[:40]:
    print("  case %d: %.0f Mcyc status %d draws %d last %s" % (i, cyc[i] / 1e6, st[i], dr[i], names[lm[i]] if 0 <= lm[i] < len(names) else str(lm[i])))
pr = eng.prof().astype(np.float64)
if pr.sum() > 0:
    print("EH_PROF per mutator attempt (calls, mean kcyc, total Gcyc, share):")
    tot = sum(pr[2 * m] for m in range(len(names)))
    for m in range(len(names)):
        if pr[2 * m + 1] > 0:
            print("  %-6s %9d %10.1f %10.2f %5.1f%%" % (names[m], pr[2 * m + 1], pr[2 * m] / pr[2 * m + 1] / 1e3, pr[2 * m] / 1e9, 100 * pr[2 * m] / tot))
    print("fuse_lists calls by la+lb (<= bytes: calls, mean kcyc, total Gcyc):")
    for b in range(14):
        k = 112 + b
        if pr[2 * k + 1] > 0:
            print("  <=%8d %9d %10.1f %10.2f" % (256 << b, pr[2 * k + 1], pr[2 * k] / pr[2 * k + 1] / 1e3, pr[2 * k] / 1e9))
    if pr[2 * 126 + 1] > 0:
        print("  rounds per fuse_lists call: %.2f" % (pr[2 * 126] / pr[2 * 126 + 1]))
    if pr[2 * 127 + 1] > 0:
        print("  workgroup starts: first .. last = %.3f ms (100 MHz clock; all workgroups resident at once when this is far below the pass)" % ((pr[2 * 127 + 1] - pr[2 * 127]) / 1e5))
    print("  pattern phase outside the mutators (Gcyc, calls): " + ", ".join("%s %.2f x%d" % (nm, pr[2 * k] / 1e9, pr[2 * k + 1]) for k, nm in (
        (59, "csum finder"), (60, "length-field finder (sz)"), (61, "cs: gather"), (62, "cs: checksum"), (63, "cp: decode"), (69, "cp: encode + compare"), (78, "ar: open + first file"), (79, "ar: file done / create"))))
    for k in range(64, 112):
        if pr[2 * k + 1] > 0:
            print("  slot %3d calls %9d mean %10.1f kcyc total %8.2f Gcyc" % (k, pr[2 * k + 1], pr[2 * k] / pr[2 * k + 1] / 1e3, pr[2 * k] / 1e9))

    print("  work memory taken by mutator attempts (what they wrote, nearly): failed attempts %.2f GB in %d; used candidates %.2f GB in %d; the used attempts' tables and temporaries %.2f GB; output %.2f GB" % (
        pr[2 * 56] / 1e9, pr[2 * 56 + 1], pr[2 * 57] / 1e9, pr[2 * 57 + 1], pr[2 * 58] / 1e9, ob / 1e9))
    print("  fuse on shortened lists: the search %d calls mean %.0f kcyc total %.1f Gcyc; the node's members found in the original lists (fr_occ) mean %.0f kcyc total %.1f Gcyc" % (
        pr[2 * 54 + 1], pr[2 * 54] / max(pr[2 * 54 + 1], 1) / 1e3, pr[2 * 54] / 1e9, pr[2 * 55] / max(pr[2 * 55 + 1], 1) / 1e3, pr[2 * 55] / 1e9))
    print("  large fuse calls that ran on shortened lists (eh_fuse_red.h): %d (finding + making the cuts: mean %.0f kcyc), that found no cut worth it: %d (mean %.0f kcyc)   # fuse_red" % (
        pr[2 * 97 + 1], pr[2 * 97] / max(pr[2 * 97 + 1], 1) / 1e3, pr[2 * 98 + 1], pr[2 * 98] / max(pr[2 * 98 + 1], 1) / 1e3))
    print("  sgm tokenizer replays of periodic documents: %d, tokens written by them: %d" % (pr[2 * 94 + 1], pr[2 * 94]))
    print("  sgm replay: large documents %d, periodic %d; checked but not replayed %d (of them look-ahead too long %d); gave up aligning %d" % (pr[2 * 95 + 1], pr[2 * 95], pr[2 * 99 + 1], pr[2 * 99], pr[2 * 89 + 1]))
    print("  sgm phases: tokenizer %d calls mean %.0f kcyc total %.1f Gcyc; pairing+flags total %.1f Gcyc; edit script total %.1f Gcyc; gather total %.1f Gcyc" % (
        pr[2 * 90 + 1], pr[2 * 90] / max(pr[2 * 90 + 1], 1) / 1e3, pr[2 * 90] / 1e9, pr[2 * 91] / 1e9, pr[2 * 92] / 1e9, pr[2 * 93] / 1e9))
