#!/usr/bin/env python3
"""Runs the engine (built against tests/hipemu, i.e. on the CPU wavefront emulator) on a small corpus and
compares it with the oracle.  Used by tests/test_emulated_kernel.py in a subprocess, because the library
path (ERLAMSA_HIP_LIB) is read once per process; also handy on its own while developing a kernel change:

  python tests/hipemu/build_emu.py && ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so \\
      python tests/hipemu/emu_parity.py "bd,bf,sr,tr2" "od,nd,bu" 64 1024 mixed
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import util
import erlamsa_amd as ea


def main():
    muts, pats, n, size = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    kind = sys.argv[5] if len(sys.argv) > 5 else "uniform"
    seed = tuple(int(x) for x in sys.argv[6].split(",")) if len(sys.argv) > 6 else (1, 2, 3)
    assert "emu" in os.environ.get("ERLAMSA_HIP_LIB", ""), "point ERLAMSA_HIP_LIB at the emulator build"
    inputs = util.corpus_uniform(n, size) if kind == "uniform" else util.corpus_mixed(n, size)
    data, off = po.pack(inputs)
    per_call = os.environ.get("EMU_PER_CALL") == "1"                 # eh_fuzz_calls: one seed per case
    seeds = None
    if per_call:
        seeds = np.random.Generator(np.random.PCG64(seed[0])).integers(0, 99999, size=(n, 3)).astype(np.int64) + 1
    want, wst, wdr, tr = po.fuzz_batch(data, off, seed=seed, seeds=seeds, mutations=muts, patterns=pats, max_case_bytes=4 << 20,
                                       max_case_work=8 << 20, trace=True)
    t = time.time()
    eng = ea.Engine(0)
    eng.configure(mutations=muts, patterns=pats, max_case_bytes=4 << 20, max_case_work=8 << 20)
    eng.upload_corpus(data, off)
    if per_call:
        eng.fuzz_calls(seeds)
    else:
        eng.fuzz_batch(seed=seed)
    got, gst = eng.download()
    gdr, _ = eng.diag()
    dt = time.time() - t
    skip = [i for i in range(n) if gst[i] in (2, 3) or wst[i] in (2, 3)]
    bad = [i for i in range(n) if i not in skip and (got[i] != want[i] or gst[i] != wst[i] or (gst[i] == 0 and gdr[i] != wdr[i]))]
    print("cases %d bad %d skipped %d emulated in %.1f s" % (n, len(bad), len(skip), dt))
    for i in bad[:3]:
        print("case %d: first diff %d, len %d vs %d, status %d vs %d, %s" % (i, util.first_diff(got[i], want[i]), len(got[i]), len(want[i]), gst[i], wst[i], tr.split("\n")[i][:200]))
    sys.exit(1 if bad or len(skip) > n // 4 else 0)


if __name__ == "__main__":
    main()
