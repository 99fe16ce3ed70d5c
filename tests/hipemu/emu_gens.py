#!/usr/bin/env python3
"""The `file` and `jump` generators (erlamsa_gen.erl:59-150) on the emulator against the oracle: multi-block inputs cut by
rand_block_size, the lazy stream forced by the pattern's first uncons (so its draws come AFTER the pattern's), cross-entry
splices of jump_somewhere, sub-ranges of the corpus as the batch (Paths stay the whole corpus), all patterns, per-call seeds.

  ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_gens.py [cases per configuration]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import util
import erlamsa_amd as ea

CONFIGS = [
    # generators, mutators, patterns, corpus kind, entry size
    ("file", "bd,bf,bi,sr,sd,num,ld,lr,ab,uw,len,ft,fn,fo", "od,nd,bu", "mixed", 9000),
    ("jump", "bd,bf,bi,sr,sd,num,ld,lr,ab,uw,len,ft,fn,fo", "od,nd,bu", "mixed", 9000),
    ("file,jump", None, "od,nd,bu,sk,sz,cs,ar,cp,co,nu", "mixed", 3000),
    ("file=3,jump=2,direct=2,random=1", "bd,sr,lr,tr2,num,fo", "od,nd,bu,sk,sz,cs,co,nu", "ragged", 20000),
    ("jump", "bd,br,sp", "nu,co,od", "ragged", 700),
    # the shape of BASELINE configs[4]: counter-hash corpus, cross-seed splices, fuse family + sed_num + length fields, pattern sz
    ("jump", "ft,fn,fo,num,len", "sz", "counter", 16384),
]


def run(n=16):
    """the comparison itself (also called by the GPU test with the real library); returns the number of cases compared"""
    total = 0
    for ci, (gens, muts, pats, kind, size) in enumerate(CONFIGS):
        rng = np.random.Generator(np.random.PCG64(100 + ci))
        inputs = util.corpus_mixed(n, size, seed=20 + ci)
        if kind == "counter":
            from erlamsa_amd import synth
            inputs = [bytes(r) for r in synth.counter(range(n), size)]
        if kind == "ragged":
            inputs = [b[:int(rng.integers(0, size + 1))] for b in inputs]
            inputs[0] = b""                                   # an empty file: finish(0) alone
            inputs[1] = inputs[1][:256] if len(inputs[1]) >= 256 else inputs[1]
        data, off = po.pack(inputs)
        for per_call in (False, True):
            seed = (7 + ci, 11, 13)
            seeds = rng.integers(1, 99999, size=(n, 3)).astype(np.int64) if per_call else None
            # the batch is a sub-range of the corpus; the generators' Paths are the whole corpus
            first, cnt = (0, n) if ci % 2 == 0 else (n // 4, n // 2)
            sub_off = off[first:first + cnt + 1] - off[first]
            sub_data = data[int(off[first]):int(off[first + cnt])] if int(off[first + cnt]) > int(off[first]) else np.zeros(1, np.uint8)
            want, wst, wdr, tr = po.fuzz_batch(sub_data, sub_off, seed=seed, seeds=None if seeds is None else seeds[:cnt], mutations=muts, patterns=pats,
                                               generators=gens, max_case_bytes=32 << 20, trace=True, paths=(data, off))
            eng = ea.Engine(0)
            eng.configure(mutations=muts, patterns=pats, generators=gens, max_case_bytes=256 << 10, big_case_bytes=32 << 20)
            eng.upload_corpus(data, off)
            if per_call:
                eng.fuzz_calls(seeds[:cnt], corpus_first=first)
            else:
                eng.fuzz_batch(seed=seed, corpus_first=first, n=cnt)
            got, gst = eng.download()
            gdr, _ = eng.diag()
            eng.close()
            skip = [i for i in range(cnt) if gst[i] in (2, 3) or wst[i] in (2, 3)]
            bad = [i for i in range(cnt) if i not in skip and (got[i] != want[i] or gst[i] != wst[i] or (gst[i] == 0 and gdr[i] != wdr[i]))]
            total += cnt
            print("config %d %s per_call=%s: cases %d bad %d skipped %d, statuses %s" % (ci, gens, per_call, cnt, len(bad), len(skip), np.bincount(gst, minlength=2).tolist()), flush=True)
            for i in bad[:3]:
                print("  case %d: first diff %d, len %d vs %d, status %d vs %d, draws %d vs %d, %s" % (i, util.first_diff(got[i], want[i]), len(got[i]), len(want[i]), gst[i], wst[i], gdr[i], wdr[i], tr.split("\n")[i][:200]))
            assert not bad and len(skip) <= cnt // 3, "config %d (%s): %d of %d cases differ, %d skipped" % (ci, gens, len(bad), cnt, len(skip))
    # jump needs two paths, file one (make_generator_fun drops them otherwise: reported here)
    eng = ea.Engine(0)
    eng.configure(generators="jump")
    one = po.pack([b"only one entry"])
    eng.upload_corpus(*one)
    try:
        eng.fuzz_batch(seed=(1, 2, 3))
        raise AssertionError("jump with a single corpus entry must be refused")
    except ea.EngineError as e:
        assert "jump" in str(e), e
    eng.close()
    return total


def main():
    assert "emu" in os.environ.get("ERLAMSA_HIP_LIB", ""), "point ERLAMSA_HIP_LIB at the emulator build"
    print("gens ok: %d cases" % run(int(sys.argv[1]) if len(sys.argv) > 1 else 16))


if __name__ == "__main__":
    main()
