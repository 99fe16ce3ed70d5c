#!/usr/bin/env python3
"""Differential campaign (development tool, not part of the test suite): the engine on the CPU wavefront
emulator against the oracle over random mutator / pattern / generator subsets, priorities, input shapes
and seeds.  Prints MISMATCH lines with everything needed to reproduce a case.

  python tests/hipemu/build_emu.py
  ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_fuzz.py <rng seed> <seconds>
"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, pyoracle as po, util, erlamsa_amd as ea
ALL = [m for m in ea.gpu_mutators() if m != "b64"]
PATS = ["od","nd","bu","sk","sz","cs","ar","cp","co","nu"]
rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + float(sys.argv[2]) if len(sys.argv) > 2 else time.time() + 600
trial = 0; total = 0; skipped = 0
while time.time() < t_end:
    trial += 1
    k = rnd.randint(1, len(ALL)); muts = rnd.sample(ALL, k)
    if rnd.random() < 0.5: muts = [m for m in muts if m not in ("ft","fn","fo")] or ["bd"]
    spec = ",".join(m if rnd.random() < 0.7 else "%s=%d" % (m, rnd.randint(1, 9)) for m in muts)
    pats = ",".join(rnd.sample(PATS, rnd.randint(1, len(PATS))))
    n = rnd.randint(1, 24)
    seed = (rnd.randint(0, 99999), rnd.randint(0, 99999), rnd.randint(0, 99999))
    kind = rnd.choice(["uniform", "mixed", "ragged", "tiny"])
    if kind == "uniform": inputs = util.corpus_uniform(n, rnd.choice([1, 7, 64, 256, 1000, 4096]), seed=seed[0])
    elif kind == "mixed": inputs = util.corpus_mixed(n, rnd.choice([64, 300, 1024, 2048]), seed=seed[1])
    elif kind == "tiny": inputs = [bytes(rnd.getrandbits(8) for _ in range(rnd.randint(0, 5))) for _ in range(n)]
    else:
        base = util.corpus_mixed(n, 1500, seed=seed[2]); inputs = [b[:rnd.randint(0, 1500)] for b in base]
    gens = rnd.choice([None, None, "random=1", "direct=3,random=1"])
    data, off = po.pack(inputs)
    try:
        want, wst, wdr, tr = po.fuzz_batch(data, off, seed=seed, mutations=spec, patterns=pats, generators=gens, max_case_bytes=2 << 20, max_case_work=4 << 20, trace=True)
    except RuntimeError as e:
        print("oracle error", e, spec, pats); continue
    eng = ea.Engine(0)
    eng.configure(mutations=spec, patterns=pats, generators=gens, max_case_bytes=8 << 20, max_case_work=4 << 20)
    eng.upload_corpus(data, off); eng.fuzz_batch(seed=seed); got, gst = eng.download(); gdr, _ = eng.diag(); eng.close()
    for i in range(n):
        total += 1
        if gst[i] in (2, 3) or wst[i] in (2, 3): skipped += 1; continue
        if got[i] != want[i] or gst[i] != wst[i] or (gst[i] == 0 and gdr[i] != wdr[i]):
            print("MISMATCH trial", trial, "case", i, "spec", spec, "pats", pats, "gens", gens, "seed", seed, "kind", kind, "n", n,
                  "len", len(got[i]), len(want[i]), "status", gst[i], wst[i], "draws", gdr[i], wdr[i], "firstdiff", util.first_diff(got[i], want[i]), flush=True)
            print("   trace:", tr.split("\n")[i][:300], flush=True)
            break
    if trial % 20 == 0: print("trials", trial, "cases", total, "skipped", skipped, flush=True)
print("done trials", trial, "cases", total, "skipped", skipped)
