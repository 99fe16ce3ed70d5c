#!/usr/bin/env python3
"""eh_result_meta against the oracle's meta trace: {pattern, P} / {used, Name} / {failed, Name} entries in the order the
reference makes them (erlamsa_patterns.erl, erlamsa_mutations.erl:1269-1279), nested scheduler calls (b64, sgm, js) included,
default tables, all patterns.  Works on the emulator and on the GPU (the caller picks the library).  usage: emu_meta.py [N]"""
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


def run(n=24, size=500, seed=(2, 7, 1)):
    inputs = util.corpus_mixed(n, size, seed=13) + synth.sgml_docs(n // 3, seed=8) + synth.json_docs(n // 3, seed=9)
    data, off = po.pack(inputs)
    want, wst, _, trace = po.fuzz_batch(data, off, seed=seed, max_case_bytes=64 << 20, trace=True)
    lines = trace.split("\n")
    eng = ea.Engine(0)
    eng.configure(flags=ea.engine.EH_FLAG_META_TRACE, max_case_bytes=1 << 20, big_case_bytes=64 << 20)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=seed)
    got, gst = eng.download()
    checked = 0
    for i in range(len(inputs)):
        if gst[i] != 0 or wst[i] != 0:
            continue
        assert got[i] == want[i], i
        ev = eng.meta(i)
        mine = " ".join("%s:%s" % (k, v) for k, v in ev)
        theirs = " ".join(lines[i].split())
        assert mine == theirs, "case %d:\n engine %s\n oracle %s" % (i, mine[:400], theirs[:400])
        checked += 1
    eng.close()
    assert checked >= 0.8 * len(inputs)
    return checked


if __name__ == "__main__":
    print("meta ok: %d traces equal the oracle's" % run(int(sys.argv[1]) if len(sys.argv) > 1 else 24))
