#!/usr/bin/env python3
"""Work areas on the emulator: the same cases with a small slot area and (a) no larger tier in the pool, (b) tiers up to
32 MiB.  Whatever the tiers, a case that completes gives the oracle's bytes, statuses and draw counts; with (b) nothing
overflows: a case that outgrows what it holds borrows an area of a higher tier and goes on.  Run with ERLAMSA_HIP_LIB=<emu lib>."""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
inputs = util.corpus_mixed(n, 600, seed=4)
data, off = po.pack(inputs)
muts = "sr,lr,bd,num,ft,sgm"
want, wst, wdr, _ = po.fuzz_batch(data, off, seed=(6, 6, 6), mutations=muts, patterns="od,nd,bu")
assert (wst < 2).all()
seen_overflow = False
for cap, big in [(64 << 10, 64 << 10), (64 << 10, 32 << 20), (16 << 10, 32 << 20)]:
    eng = ea.Engine(0)
    eng.configure(mutations=muts, patterns="od,nd,bu", max_case_bytes=cap, big_case_bytes=big)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=(6, 6, 6))
    got, gst = eng.download()
    gdr, glm = eng.diag()
    for i in range(n):
        if gst[i] == 2:
            assert glm[i] < 0, "an overflow reports its site"
            print("overflow: cap", cap, "big", big, "case", i, "site", -glm[i], "oracle out", len(want[i]))
            seen_overflow = True
            continue
        assert got[i] == want[i] and gst[i] == wst[i] and gdr[i] == wdr[i], (cap, big, i)
    if big > cap:
        assert (gst != 2).all(), "larger tiers must absorb every case of this set"
    eng.close()
assert seen_overflow, "the single-tier run is meant to overflow somewhere"
print("tiers ok")
