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
print("chunks ok")
