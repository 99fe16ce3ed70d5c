#!/usr/bin/env python3
"""eh_result_meta against the oracle's meta trace IN FULL (ABI 7): the text erlamsa's meta logger prints for a case
(erlamsa_main.erl:58-70) - {pattern, _}, the patterns' own entries, every mutator's own entry, {used, Name} / {failed, Name},
nested scheduler calls (b64, sgm, js) included, and the short form derived from it - default tables, all patterns; documents,
compressed inputs and zip archives through their patterns.  Works on the emulator and on the GPU (the caller picks the library).  usage: emu_meta.py [N]"""
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
    want, wst, _, trace = po.fuzz_batch(data, off, seed=seed, max_case_bytes=64 << 20, trace="full")
    lines = trace.split("\x1e\n")
    _, _, _, short = po.fuzz_batch(data, off, seed=seed, max_case_bytes=64 << 20, trace=True)
    short = short.split("\n")
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
        from erlamsa_amd import meta as M
        terms, cut = eng.meta_terms(i)
        assert not cut
        mine = M.lines(terms)
        assert mine == lines[i], "case %d:\n engine %s\n oracle %s" % (i, mine.replace("\n", " ")[:600], lines[i].replace("\n", " ")[:600])
        # the short form ({failed | used, Mutator}, {pattern, Code}, {skipped_big, _}) agrees with the oracle's short trace wherever the
        # reference's own quirk has not dropped the front of the list (sgml_mutate / json_mutate returning NewMeta alone) and no `co` ran
        ev = " ".join("%s:%s" % (k, v) for k, v in eng.meta(i))
        theirs = " ".join(short[i].split())
        if "pattern:co" not in theirs and len(ev.split()) == len(theirs.split()):
            assert ev == theirs, (i, ev[:300], theirs[:300])
        checked += 1
    eng.close()
    assert checked >= 0.8 * len(inputs)
    return checked


def size_query_then_fetch(eng, i):
    """what erlang/c_src/erlamsa_hip_nif.c nif_meta does: ask for the length with an empty buffer, then fetch into a binary of that
    size (ABI 8: eh_result_meta copies min(len, cap) bytes and returns EH_OK either way)"""
    import ctypes as C
    raw = eng.meta_raw(i)
    n = C.c_uint64(12345)
    assert eng.lib.eh_result_meta(eng.h, i, None, 0, C.byref(n)) == 0 and n.value == len(raw), (n.value, len(raw))
    buf = (C.c_uint8 * max(n.value, 1))()
    assert eng.lib.eh_result_meta(eng.h, i, buf, n.value, C.byref(n)) == 0 and bytes(buf[:n.value]) == raw
    if len(raw) > 3:
        short = (C.c_uint8 * 8)(*([0xAA] * 8))
        assert eng.lib.eh_result_meta(eng.h, i, short, 3, C.byref(n)) == 0 and n.value == len(raw)
        assert bytes(short[:3]) == raw[:3] and bytes(short[3:]) == b"\xaa" * 5


def run_sets(n=24):
    """the full text on the inputs that reach every kind of entry: documents through js / sgm and their inner runs, the complex
    patterns (skipper, sizer, csum, compressed, archiver, co, nu), gzip / zlib inputs through cp, zip archives through ar and zip"""
    import warnings
    sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
    import emu_containers as ec
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        zips = ec.zip_corpus(max(10, n * 2 // 3), 501)
    sets = [
        (synth.json_docs(n, seed=4) + synth.sgml_docs(n, seed=5), (4, 5, 6), "js,sgm,b64,uri,num,bd,ab,ad", "od,nd,bu"),
        ([bytes(r) for r in synth.mixed(2 * n, 900, seed=9)], (7, 8, 9), None, "sk,sz,cs,cp,ar,co,nu,od"),
        (ec.compressed_corpus(n + n // 4, 301), (2, 4, 9), "bd,bf,bi,sr,num,lr,uw", "cp,od"),
        (zips, (3, 4, 9), "zip=3,bd,bf,sr,num", "ar=3,od"),
    ]
    checked = 0
    for inputs, seed, muts, pats in sets:
        data, off = po.pack(inputs)
        want, wst, _, tr = po.fuzz_batch(data, off, seed=seed, mutations=muts, patterns=pats, max_case_bytes=64 << 20, trace="full")
        lines = tr.split("\x1e\n")
        eng = ea.Engine(0)
        eng.configure(mutations=muts, patterns=pats, flags=ea.engine.EH_FLAG_META_TRACE, max_case_bytes=1 << 20, big_case_bytes=64 << 20)
        eng.upload_corpus(data, off)
        eng.fuzz_batch(seed=seed)
        got, gst = eng.download()
        for i in range(len(inputs)):
            if gst[i] != 0 or wst[i] != 0:
                continue
            assert got[i] == want[i], i
            assert util.meta_matches(eng, i, lines[i]), "set %s case %d: engine %r oracle %r" % (pats, i, eng.meta_terms(i)[0][:12], lines[i][:300])
            checked += 1
            if checked % 16 == 1:
                size_query_then_fetch(eng, i)
        eng.close()
    return checked


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    print("meta ok: %d + %d traces equal the oracle's" % (run(n), run_sets(n)))
