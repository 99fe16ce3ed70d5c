#!/usr/bin/env python3
"""Differential campaign no. 3 (development tool, not part of the test suite): this round's additions under random configurations -
gzip / zlib / zip inputs with the cp and ar patterns and the zip mutator among random mutator / pattern subsets, the file and jump
generators, small slots (the codecs' scratch blocks come from borrowed areas), the meta trace.  Prints MISMATCH lines with
everything needed to reproduce a case.

  ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_fuzz3.py <rng seed> <seconds>
"""
import os, sys, time, random, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'hipemu'))
import numpy as np, pyoracle as po, util, erlamsa_amd as ea
import emu_containers as ec
warnings.simplefilter("ignore")
ALL = list(ea.gpu_mutators())
PATS = ["od", "nd", "bu", "sk", "sz", "cs", "ar", "cp", "co", "nu"]
def _races():
    """with the race build (build_emu.py --race): cross-lane accesses without a rendezvous + out-of-bounds accesses so far"""
    try:
        import ctypes
        l = ctypes.CDLL(os.environ["ERLAMSA_HIP_LIB"]); l.hipemu_race_count.restype = ctypes.c_ulong; l.hipemu_oob_count.restype = ctypes.c_ulong
        return "races %d oob %d" % (l.hipemu_race_count(), l.hipemu_oob_count())
    except (AttributeError, OSError, KeyError):
        return ""


rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
t_end = time.time() + (float(sys.argv[2]) if len(sys.argv) > 2 else 600)
trial = total = skipped = traced = 0
stat = np.zeros(8, dtype=np.int64)
while time.time() < t_end:
    trial += 1
    k = rnd.randint(1, 12); muts = rnd.sample(ALL, k)
    if rnd.random() < 0.5: muts = [m for m in muts if m not in ("sgm", "js", "b64", "ft", "fn", "fo")] or ["bd"]
    if rnd.random() < 0.4: muts = list(set(muts + ["zip"]))
    spec = ",".join(m if rnd.random() < 0.7 else "%s=%d" % (m, rnd.randint(1, 9)) for m in muts)
    pl = rnd.sample(PATS, rnd.randint(1, len(PATS)))
    if rnd.random() < 0.6: pl = list(set(pl + [rnd.choice(["ar", "cp"])]))
    pats = ",".join(p if rnd.random() < 0.6 else "%s=%d" % (p, rnd.randint(1, 5)) for p in pl)
    n = rnd.randint(2, 12)
    seed = (rnd.randint(0, 99999), rnd.randint(0, 99999), rnd.randint(0, 99999))
    kind = rnd.choice(["gz", "zip", "mixed", "both"])
    if kind == "gz": inputs = ec.compressed_corpus(n, seed[0])
    elif kind == "zip": inputs = ec.zip_corpus(n, seed[0])
    elif kind == "mixed": inputs = util.corpus_mixed(n, rnd.choice([300, 1024, 5000]), seed=seed[1])
    else: inputs = (ec.compressed_corpus(n, seed[0]) + ec.zip_corpus(n, seed[1]))[::2]
    n = len(inputs)
    gens = rnd.choice([None, None, "file", "jump", "file=3,jump=2,direct=2,random=1"])
    slot = rnd.choice([32, 64, 256, 1024]) << 10
    bs = rnd.choice([1.0, 1.0, 0.25, 2.0, 0.0625])                 # blockscale: block sizes of the file / jump / random streams
    first = rnd.choice([1, 1, 7, 1000])
    data, off = po.pack(inputs)
    try:
        want, wst, wdr, tr = po.fuzz_batch(data, off, seed=seed, mutations=spec, patterns=pats, generators=gens, blockscale=bs, first_case=first, max_case_bytes=32 << 20, trace="full", max_case_seconds=20.0)
    except RuntimeError as e:
        print("oracle error", e, spec, pats); continue
    eng = ea.Engine(0)
    eng.configure(mutations=spec, patterns=pats, generators=gens, blockscale=bs, max_case_bytes=slot, big_case_bytes=32 << 20, flags=ea.engine.EH_FLAG_META_TRACE)
    eng.upload_corpus(data, off); eng.fuzz_batch(seed=seed, first_case=first); got, gst = eng.download(); gdr, _ = eng.diag()
    lines = tr.split("\x1e\n")
    for i in range(n):
        total += 1
        stat[min(int(gst[i]), 7)] += 1
        if gst[i] == 2 or wst[i] in (2, 6): skipped += 1; continue
        bad = got[i] != want[i] or gst[i] != wst[i] or (gst[i] == 0 and gdr[i] != wdr[i])
        if not bad and gst[i] == 0:
            traced += 1
            if not util.meta_matches(eng, i, lines[i]):
                bad = True
                from erlamsa_amd import meta as _meta
                print("   TRACE differs: engine", _meta.lines(eng.meta_terms(i)[0]).replace("\n", " ")[:300], "| oracle", lines[i].replace("\n", " ")[:300], flush=True)
        if bad:
            print("MISMATCH trial", trial, "case", i, "spec", spec, "pats", pats, "gens", gens, "seed", seed, "kind", kind, "n", n, "slot", slot, "blockscale", bs, "first_case", first,
                  "len", len(got[i]), len(want[i]), "status", gst[i], wst[i], "draws", gdr[i], wdr[i], "firstdiff", util.first_diff(got[i], want[i]), flush=True)
            print("   trace:", lines[i][:300], flush=True)
            break
    eng.close()
    if trial % 10 == 0: print("trials", trial, "cases", total, "skipped", skipped, "traces compared", traced, "statuses", stat.tolist(), _races(), flush=True)
print("done trials", trial, "cases", total, "skipped", skipped, "traces compared", traced, "statuses", stat.tolist())
