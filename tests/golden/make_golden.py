#!/usr/bin/env python3
"""Generates tests/golden/vectors.json.

IMPORTANT: the reference (Erlang) cannot run on this image, so these vectors are produced by the
CPU oracle (oracle/), not by erlamsa itself.  They pin the oracle against regressions and give the
GPU tests a fixture that does not need the oracle at run time.  Format per vector:
  {seed, first_case, mutations, patterns, inputs_hex[], outputs_hex[], status[]}
When an Erlang host is available, tests/golden/capture.escript runs the real erlamsa_main:fuzzer/1 over the same inputs
(vectors.eterm, written here too) and tests/golden/apply_beam_capture.py rewrites vectors.json from its output: that pins
oracle-vs-BEAM parity ("Oracle").  `generator` in the file says which of the two produced it.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
from erlamsa_amd import synth  # noqa: E402

SETS = [
    ("c2_bytes", (1, 2, 3), "bd,bf,bi", "od", 24, 256, "uniform"),
    ("bytes_seq", (4, 5, 6), "bd,bei,bed,bf,bi,ber,br,sp,sr,sd,snand,srnd,uw,ui,nil", "od,nd,bu", 24, 200, "uniform"),
    ("text", (7, 8, 9), "num,ld,lds,lr2,lri,lr,ls,lp,lis,lrs,ab,ad,uri,b64,zip", "od,nd,bu", 24, 300, "mixed"),
    ("trees_len", (10, 11, 12), "tr2,td,ts1,ts2,tr,len", "od,nd,bu,sk,sz,cs", 24, 300, "mixed"),
    ("hello", (1, 2, 3), None, None, 1, 0, "hello"),
    # the full default tables (41 mutators incl. sgm/js/b64/fuse, all 10 patterns)
    ("default_mixed", (13, 14, 15), None, None, 48, 400, "mixed"),
    ("default_json", (16, 17, 18), None, None, 32, 0, "json"),
    ("default_sgml", (19, 20, 21), None, None, 32, 0, "sgml"),
    ("sgm_js_only", (22, 23, 24), "sgm,js", "od,nd,bu", 48, 0, "docs"),
    # containers (round 3): real gzip / zlib inputs through pattern cp, real zip archives through pattern ar and mutator zip.
    # A BEAM capture of these two sets also pins the restated corners of OTP's zlib / prim_zip / zip ("Oracle").
    ("containers_cp", (25, 26, 27), "bd,bf,bi,sr,num,lr,uw", "cp,od", 30, 0, "gz"),
    ("containers_zip", (28, 29, 30), "zip=3,bd,bf,sr,num", "ar=3,od", 20, 0, "zip"),
    # the file / jump generators (erlamsa_gen.erl:59-150): the inputs are the Paths.  Not in vectors.eterm: with paths other than
    # [direct] erlamsa_main:fuzzer/1 records nothing (erlamsa_main.erl:139-146), a capture would have to go through -o files
    ("gen_file", (31, 32, 33), "bd,bf,sr,num,lr,ft", "od,nd,bu,sz", 16, 6000, "mixed", "file"),
    ("gen_jump", (34, 35, 36), "bd,bf,sr,num,lr,fo", "od,nd,bu,sz", 16, 6000, "mixed", "jump"),
]


def main():
    vecs = []
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import warnings
    import emu_containers as ec                      # the container corpora of the emulator / GPU tests
    for entry in SETS:
        name, seed, muts, pats, n, size, kind = entry[:7]
        gens = entry[7] if len(entry) > 7 else None
        if kind == "gz":
            inputs = ec.compressed_corpus(n, seed[0])
        elif kind == "zip":
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                inputs = ec.zip_corpus(n, seed[0])
        elif kind == "uniform":
            inputs = [bytes(r) for r in synth.uniform(n, size, seed=seed[0])]
        elif kind == "mixed":
            inputs = [bytes(r) for r in synth.mixed(n, size, seed=seed[0])]
        elif kind == "json":
            inputs = synth.json_docs(n, seed=seed[0])
        elif kind == "sgml":
            inputs = synth.sgml_docs(n, seed=seed[0])
        elif kind == "docs":
            inputs = synth.json_docs(n // 2, seed=seed[0]) + synth.sgml_docs(n - n // 2, seed=seed[1])
        else:
            inputs = [b"Hello erlamsa!\n"]          # BASELINE configs[0] input; default mutators and patterns
        data, off = po.pack(inputs)
        outs, st, _, _ = po.fuzz_batch(data, off, seed=seed, mutations=muts, patterns=pats, generators=gens, max_case_bytes=256 << 10)
        vecs.append({"name": name, "seed": list(seed), "first_case": 1, "mutations": muts, "patterns": pats,
                     "inputs_hex": [b.hex() for b in inputs], "outputs_hex": [o.hex() for o in outs], "status": [int(x) for x in st]})
        if gens:
            vecs[-1]["generators"] = gens
    with open(os.path.join(HERE, "vectors.json"), "w") as f:
        json.dump({"generator": "oracle (C++ restatement) — NOT a BEAM run", "vectors": vecs}, f, indent=0)
    # the same inputs as Erlang terms for tests/golden/capture.escript (file:consult/1)
    def term(v):
        q = lambda x: "default" if x is None else '"%s"' % x
        return '{"%s", {%d,%d,%d}, %d, %s, %s, [%s]}' % (v["name"], *v["seed"], v["first_case"], q(v["mutations"]), q(v["patterns"]),
                                                         ", ".join('"%s"' % h for h in v["inputs_hex"]))
    with open(os.path.join(HERE, "vectors.eterm"), "w") as f:
        f.write("[\n" + ",\n".join(term(v) for v in vecs if "generators" not in v) + "\n].\n")
    print("wrote", len(vecs), "vector sets")


if __name__ == "__main__":
    main()
