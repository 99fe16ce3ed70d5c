#!/usr/bin/env python3
"""Chunked work areas on the emulator: the reference's full default tables (41 mutators with sgm / js / b64 nesting, 10 patterns)
with a tier-0 area so small (16 KiB) that most cases go on in borrowed areas, often several times and from inside nested
scheduler calls — every mutator attempt that ran out of memory is repeated after the case has grown, nothing else is.  A case
that completes must give the oracle's bytes, statuses and draw counts.  Run with ERLAMSA_HIP_LIB=<emu lib>.
usage: emu_chunks.py [N] [CAP]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import util
import erlamsa_amd as ea
from erlamsa_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 16 << 10
inputs = util.corpus_mixed(n, 700, seed=9) + synth.sgml_docs(n // 3, seed=5) + synth.json_docs(n // 3, seed=6)
data, off = po.pack(inputs)
want, wst, wdr, _ = po.fuzz_batch(data, off, seed=(4, 5, 6), max_case_bytes=32 << 20)
eng = ea.Engine(0)
eng.configure(max_case_bytes=cap, big_case_bytes=32 << 20)
eng.upload_corpus(data, off)
eng.fuzz_batch(seed=(4, 5, 6))
got, gst = eng.download()
gdr, glm = eng.diag()
pk = eng.peak()
ps = eng.pool_stats()
bad = 0
for i in range(len(inputs)):
    if gst[i] in (2, 3) or wst[i] in (2, 3):
        continue
    if got[i] != want[i] or gst[i] != wst[i] or gdr[i] != wdr[i]:
        bad += 1
        print("MISMATCH case", i, "status", gst[i], wst[i], "len", len(got[i]), len(want[i]), "draws", gdr[i], wdr[i], "peak", pk[i])
grown = int((pk > cap).sum())
print("cases %d, needed more than the slot's %d bytes: %d, areas taken per tier %s, engine-only statuses %d, mismatches %d" % (
    len(inputs), cap, grown, ps["taken"], int(((gst == 2) | (gst == 3)).sum()), bad))
assert bad == 0
assert grown > len(inputs) // 4 and sum(ps["taken"]) >= grown
eng.close()

# A slot that is ALMOST full when the pattern starts: the random generator's stream (up to ~30 KB of blocks in the work area) in a
# 32 KiB slot, patterns whose own scans borrow an area and give it back (cs, sz) or that mutate the tail of a block (sk), and
# mutators whose result goes through flush_bvecs (num).  The chunk borrowed by the pattern starts exactly where the mutator
# attempt begins; the candidate, made a few bytes above that point in the chunk BELOW, is moved down onto an overlapping range
# (a plain wave_copy there corrupted 11 - 48 bytes of such cases until round 3).
bad2 = total2 = 0
data2, off2 = po.pack(util.corpus_uniform(24, 64, seed=3))
for spec, pats in (("num", "cs"), ("num", "sz"), ("num", "sk"), ("num,bd,sr,lr", "cs,sz,sk,od,nd")):
    for s in (1, 11) if n >= 24 else (11,):
        seed = (s + 1, 77, 5)
        want2, wst2, wdr2, _ = po.fuzz_batch(data2, off2, seed=seed, mutations=spec, patterns=pats, generators="random=1", max_case_bytes=32 << 20)
        eng = ea.Engine(0)
        eng.configure(mutations=spec, patterns=pats, generators="random=1", max_case_bytes=32768, big_case_bytes=32 << 20)
        eng.upload_corpus(data2, off2); eng.fuzz_batch(seed=seed); got2, gst2 = eng.download(); gdr2, _ = eng.diag()
        eng.close()
        for i in range(24):
            if gst2[i] in (2, 3) or wst2[i] in (2, 3):
                continue
            total2 += 1
            if got2[i] != want2[i] or gst2[i] != wst2[i] or gdr2[i] != wdr2[i]:
                bad2 += 1
                print("MISMATCH (nearly full slot)", spec, pats, seed, "case", i, "len", len(got2[i]), len(want2[i]), "first diff", util.first_diff(got2[i], want2[i]))
print("nearly full slots: %d cases, mismatches %d" % (total2, bad2))
assert bad2 == 0 and total2 > (80 if n >= 24 else 20)
print("chunks ok")
