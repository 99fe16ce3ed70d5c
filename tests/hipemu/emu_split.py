#!/usr/bin/env python3
"""split/1 + split_into_maxblocks/2 (erlamsa_patterns.erl:44-59): inputs of more than ABSMAX_BINARY_BLOCK = 1 000 000 bytes are
cut at 500 000 + rand(500 000) - 1 before mutate_once_loop/6 walks the pieces.  Engine against the oracle on inputs just above the
limit, one and a half, two and three times it.  Run with ERLAMSA_HIP_LIB=<emu lib> (tests/test_emulated_kernel.py: two sizes, the
byte mutators - the emulator moves a megabyte per lane fiber), and by tests/test_gpu_round6.py on the GPU with the full sizes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np


def big_inputs(sizes, per_size=2, seed=4459):
    rng = np.random.Generator(np.random.PCG64(seed))
    out = []
    for s in sizes:
        for k in range(per_size):
            if k % 3 == 0:
                out.append(rng.integers(0, 256, size=s, dtype=np.uint8).tobytes())
            elif k % 3 == 1:                                       # lines of text with numbers: the line and num mutators find work
                words = [b"alpha ", b"beta ", b"12345 ", b"-77 ", b"gamma\n", b"<d>", b"</d>\n"]
                idx = rng.integers(0, len(words), size=s // 4 + 8)
                out.append(b"".join(words[i] for i in idx)[:s])
            else:                                                  # periodic
                per = rng.integers(32, 127, size=int(rng.integers(5, 90)), dtype=np.uint8)
                out.append(np.tile(per, s // len(per) + 1)[:s].tobytes())
    return out


def run(ea, inputs, pats, seed, muts="bd,bf,bi,ber,sd,sr=2,ld,lr,num", engine_cap=4 << 20, big=64 << 20, oracle_threads=False):
    import pyoracle as po
    import util
    data, off = po.pack(inputs)
    if oracle_threads:
        ora = util.oracle_live(data, off, seed=seed, mutations=muts, patterns=pats, max_case_bytes=big, max_case_seconds=60.0)
        want, wst, wdr = ora.outs, ora.status, ora.draws
    else:
        want, wst, wdr, _ = po.fuzz_batch(data, off, seed=seed, mutations=muts, patterns=pats, max_case_bytes=big)
    eng = ea.Engine(0)
    eng.configure(mutations=muts, patterns=pats, max_case_bytes=engine_cap, big_case_bytes=big, out_capacity=max(4 << 30, 64 * len(data)) if oracle_threads else 1 << 30)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=seed)
    got, gst = eng.download()
    gdr, _ = eng.diag()
    eng.close()
    checked = 0
    for i in range(len(inputs)):
        if gst[i] in (2, 3) or wst[i] in (2, 3, 6):
            continue
        assert int(gst[i]) == int(wst[i]) and got[i] == want[i] and (gst[i] != 0 or int(gdr[i]) == int(wdr[i])), \
            "patterns %s case %d (%d bytes): status %d / %d, length %d / %d, draws %d / %d" % (pats, i, len(inputs[i]), gst[i], wst[i], len(got[i]), len(want[i]), gdr[i], wdr[i])
        checked += 1
    return checked


if __name__ == "__main__":
    import erlamsa_amd as ea
    ins = big_inputs(sizes=(1000001, 1600000), per_size=3)
    n = run(ea, ins, "od", (1, 2, 3), muts="bd,bf,bi,ber,sd") + run(ea, ins, "nd,sk", (5, 5, 6), muts="bd,bf,bi,ber,sd")
    assert n >= 2 * len(ins) - 1, n
    print("split ok: %d cases" % n)
