#!/usr/bin/env python3
"""Runs golden vector sets through the engine and the oracle (with its meta trace) and prints the differing cases."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po
import erlamsa_amd as ea
vecs = json.load(open(os.path.join(ROOT, "tests/golden/vectors.json")))["vectors"]
for vec in vecs:
    if len(sys.argv) > 1 and vec["name"] not in sys.argv[1:]:
        continue
    inputs = [bytes.fromhex(h) for h in vec["inputs_hex"]]
    data, off = po.pack(inputs)
    want, wst, wdr, tr = po.fuzz_batch(data, off, seed=tuple(vec["seed"]), mutations=vec["mutations"], patterns=vec["patterns"], first_case=vec["first_case"], max_case_bytes=256 << 10, trace=True)
    for rep in range(1):
        eng = ea.Engine(0)
        eng.configure(mutations=vec["mutations"], patterns=vec["patterns"])
        eng.upload_corpus(data, off)
        eng.fuzz_batch(seed=tuple(vec["seed"]), first_case=vec["first_case"])
        outs, st = eng.download()
        dr, lm = eng.diag()
        eng.close()
        bad = [i for i in range(len(inputs)) if not (wst[i] in (2, 3) or st[i] in (2, 3)) and (outs[i] != want[i] or st[i] != wst[i] or dr[i] != wdr[i])]
        print(vec["name"], "rep", rep, "bad", bad)
        for i in bad[:3]:
            print("  case", i, "status", st[i], wst[i], "draws", dr[i], wdr[i], "len", len(outs[i]), len(want[i]), "trace:", tr.split("\n")[i][:300])
            fd = next((k for k in range(min(len(outs[i]), len(want[i]))) if outs[i][k] != want[i][k]), -1)
            nd = sum(1 for a, b in zip(outs[i], want[i]) if a != b)
            print("   first diff at", fd, "differing bytes", nd, "full trace:", tr.split("\n")[i])
            print("   got ", outs[i][max(0, fd - 40):fd + 80]); print("   want", want[i][max(0, fd - 40):fd + 80])
