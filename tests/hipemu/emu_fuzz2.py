#!/usr/bin/env python3
"""Differential campaign no. 2 (development tool, not part of the test suite): like emu_fuzz.py, aimed at this round's code —
small slots (the case goes on in borrowed areas, attempts are repeated after running out of memory, also inside nested
scheduler calls), the streaming fuse on small lists (fuse_stream_min down to 64), the meta trace — over random mutator /
pattern / generator subsets incl. sgm, js and b64.  Prints MISMATCH lines with everything needed to reproduce a case.

  ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_fuzz2.py <rng seed> <seconds>
"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, pyoracle as po, util, erlamsa_amd as ea
from erlamsa_amd import synth
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
trial = total = skipped = grown = traced = 0
while time.time() < t_end:
    trial += 1
    k = rnd.randint(1, len(ALL)); muts = rnd.sample(ALL, k)
    if rnd.random() < 0.4: muts = [m for m in muts if m not in ("sgm", "js", "b64")] or ["bd"]
    if rnd.random() < 0.3: muts = list(set(muts + ["ft", "fn", "fo"]))
    spec = ",".join(m if rnd.random() < 0.7 else "%s=%d" % (m, rnd.randint(1, 9)) for m in muts)
    pats = ",".join(rnd.sample(PATS, rnd.randint(1, len(PATS))))
    n = rnd.randint(1, 12)
    seed = (rnd.randint(0, 99999), rnd.randint(0, 99999), rnd.randint(0, 99999))
    kind = rnd.choice(["uniform", "mixed", "ragged", "sgml", "json", "periodic"])
    if kind == "uniform": inputs = util.corpus_uniform(n, rnd.choice([7, 64, 256, 1000, 3000]), seed=seed[0])
    elif kind == "mixed": inputs = util.corpus_mixed(n, rnd.choice([64, 300, 1024, 2048]), seed=seed[1])
    elif kind == "sgml": inputs = synth.sgml_docs(n, seed=seed[0])
    elif kind == "json": inputs = synth.json_docs(n, seed=seed[1])
    elif kind == "periodic":
        inputs = []
        for _ in range(n):
            per = bytes(rnd.getrandbits(8) for _ in range(rnd.randint(3, 200)))
            b = bytearray((per * (3000 // len(per) + 1))[:rnd.randint(200, 3000)])
            for _ in range(rnd.randint(0, 4)): b[rnd.randrange(len(b))] = rnd.getrandbits(8)
            inputs.append(bytes(b))
    else:
        base = util.corpus_mixed(n, 1500, seed=seed[2]); inputs = [b[:rnd.randint(0, 1500)] for b in base]
    gens = rnd.choice([None, None, None, "random=1", "direct=3,random=1"])
    slot = rnd.choice([16, 32, 64, 256, 1024]) << 10
    fsm = rnd.choice([64, 256, 2048, 16384])
    want_trace = rnd.random() < 0.5
    data, off = po.pack(inputs)
    try:
        want, wst, wdr, tr = po.fuzz_batch(data, off, seed=seed, mutations=spec, patterns=pats, generators=gens, max_case_bytes=32 << 20, trace="full", max_case_seconds=20.0)
    except RuntimeError as e:
        print("oracle error", e, spec, pats); continue
    eng = ea.Engine(0)
    eng.configure(mutations=spec, patterns=pats, generators=gens, max_case_bytes=slot, big_case_bytes=32 << 20, fuse_stream_min=fsm,
                  flags=ea.engine.EH_FLAG_META_TRACE if want_trace else 0)
    eng.upload_corpus(data, off); eng.fuzz_batch(seed=seed); got, gst = eng.download(); gdr, _ = eng.diag(); pk = eng.peak()
    lines = tr.split("\x1e\n")
    for i in range(n):
        total += 1
        if gst[i] in (2, 3) or wst[i] in (2, 3, 6): skipped += 1; continue
        grown += int(pk[i] > slot)
        bad = got[i] != want[i] or gst[i] != wst[i] or (gst[i] == 0 and gdr[i] != wdr[i])
        if not bad and want_trace and gst[i] == 0:
            traced += 1
            if not util.meta_matches(eng, i, lines[i]):
                bad = True
                from erlamsa_amd import meta as _meta
                print("   TRACE differs: engine", _meta.lines(eng.meta_terms(i)[0]).replace("\n", " ")[:300], "| oracle", lines[i].replace("\n", " ")[:300], flush=True)
        if bad:
            print("MISMATCH trial", trial, "case", i, "spec", spec, "pats", pats, "gens", gens, "seed", seed, "kind", kind, "n", n, "slot", slot, "fsm", fsm,
                  "len", len(got[i]), len(want[i]), "status", gst[i], wst[i], "draws", gdr[i], wdr[i], "firstdiff", util.first_diff(got[i], want[i]), flush=True)
            print("   trace:", lines[i][:300], flush=True)
            break
    eng.close()
    if trial % 10 == 0: print("trials", trial, "cases", total, "skipped", skipped, "grew beyond the slot", grown, "traces compared", traced, _races(), flush=True)
print("done trials", trial, "cases", total, "skipped", skipped, "grew beyond the slot", grown, "traces compared", traced)
