#!/usr/bin/env python3
"""The delimiter matcher of the tree mutators (csrc/eh_tree.h tree_parse: partial_parse/1 + grow/3, erlamsa_mutations.erl:800-905)
against the oracle on inputs made for it: nesting deeper than the 64 entries the stack keeps in lane registers (spills and refills
of 32 entries), quotes that open or close depending on what is on top, closers that match nothing, openers that never close,
batches of exactly 64 events.  Mutators tr2, td, ts1, ts2, tr under od / nd.

  ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_tree.py [rng seed] [trials]
"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import pyoracle as po, util, erlamsa_amd as ea

OPEN, CLOSE = b'([<{', b')]>}'


def doc(rnd, kind):
    out = bytearray()
    if kind == "deep":                                   # one chain of openers deeper than 64 (+ some that never close), closed again
        d = rnd.randint(60, 210)
        ks = [rnd.randrange(4) for _ in range(d)]
        for k in ks: out += bytes([OPEN[k]]) + (b'a' if rnd.random() < 0.3 else b'')
        keep = rnd.randint(0, d)
        for k in reversed(ks[d - keep:]): out += bytes([CLOSE[k]]) + (b'x' if rnd.random() < 0.2 else b'')
    elif kind == "mix":                                  # random delimiters, quotes included, with plain bytes between
        for _ in range(rnd.randint(10, 600)):
            r = rnd.random()
            if r < 0.35: out.append(rnd.choice(OPEN))
            elif r < 0.7: out.append(rnd.choice(CLOSE))
            elif r < 0.8: out.append(rnd.choice(b'"\''))
            else: out += bytes(rnd.choice(b'abc 123\n') for _ in range(rnd.randint(1, 5)))
    elif kind == "lines":                                # many equal small nodes, an event count that is a multiple of 64 now and then
        unit = rnd.choice([b'(ab)', b'[x]{y}', b'"q"', b'<a>(b[c])'])
        out += unit * rnd.randint(8, 96)
    else:                                                # saw: down 40..120, up part of the way, several times
        depth = []
        for _ in range(rnd.randint(2, 6)):
            for _ in range(rnd.randint(40, 120)): k = rnd.randrange(4); depth.append(k); out.append(OPEN[k])
            for _ in range(rnd.randint(0, len(depth))): out.append(CLOSE[depth.pop()])
            if rnd.random() < 0.3: out.append(rnd.choice(CLOSE))             # a closer that matches nothing
    return bytes(out) or b'()'


def main():
    rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    trials = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    cases = 0
    for trial in range(trials):
        n = 8
        inputs = [doc(rnd, rnd.choice(["deep", "mix", "lines", "saw"])) for _ in range(n)]
        seed = (rnd.randint(0, 99999), rnd.randint(0, 99999), rnd.randint(0, 99999))
        spec = rnd.choice(["tr2,td,ts1,ts2,tr", "ts1,ts2", "tr,tr2=3", "td,ts1=4,tr"])
        pats = rnd.choice(["od", "nd", "od,nd,bu"])
        data, off = po.pack(inputs)
        want, wst, wdr, _ = po.fuzz_batch(data, off, seed=seed, mutations=spec, patterns=pats, max_case_bytes=32 << 20, max_case_seconds=20.0)
        eng = ea.Engine(0)
        eng.configure(mutations=spec, patterns=pats, max_case_bytes=1 << 20, big_case_bytes=32 << 20)
        eng.upload_corpus(data, off); eng.fuzz_batch(seed=seed); got, gst = eng.download(); gdr, _ = eng.diag()
        for i in range(n):
            if gst[i] in (2, 3) or wst[i] in (2, 3, 6): continue
            cases += 1
            assert got[i] == want[i] and gst[i] == wst[i] and (gst[i] != 0 or gdr[i] == wdr[i]), (
                "tree matcher: trial %d case %d spec %s pats %s seed %s: len %d vs %d, status %d vs %d, draws %d vs %d, first difference at %s; input %r" % (
                    trial, i, spec, pats, seed, len(got[i]), len(want[i]), gst[i], wst[i], gdr[i], wdr[i], util.first_diff(got[i], want[i]), inputs[i][:80]))
        eng.close()
    print("emu_tree ok: %d trials, %d cases compared" % (trials, cases))


if __name__ == "__main__":
    main()
