#!/usr/bin/env python3
"""erlamsa_sgml:sgml_mutate/2 on periodic documents three ways: the tokenizer replaying one period's tokens (csrc/eh_sgml.h, the
default for blocks of 16 KiB and more), walking them tag by tag (EH_FLAG_SGML_NO_REPLAY) and the oracle.  Documents: a run of
elements repeated many times - plain, with attributes in every quoting style, comments, processing instructions, failed tags whose
white space is eaten, text between the tags -, a period that starts in the middle of a tag, a stretch followed by an unterminated
quote or comment (look-ahead to the end of the block: nothing may be replayed), a stretch too short to replay.
Bytes, statuses and PRNG draw counts must agree.

  python tests/hipemu/build_emu.py && ERLAMSA_HIP_LIB=build/liberlamsa_hip_emu.so python tests/hipemu/emu_sgml_replay.py [n] [seed] [scale] [small]
(with the real library the same comparison runs on the GPU; scale multiplies the repeat counts)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyoracle as po
import erlamsa_amd as ea
from erlamsa_amd.engine import EH_FLAG_SGML_NO_REPLAY, EH_FLAG_SGML_NO_LANES

FAILRUN_UNITS = [b"<a =1 <b =2 <c =3 <d>x</d>", b"< p =1< q =2 < r =3 <s t='u'>v</s> ", b"<a =1 <b =2 <c =3 <e =4 <f =5 <g h=i j>k</g>\n", b"<x y=\"1 <x y=\"2 <x y=\"3 <z>\"w</z>",
                 b"<m<n =1 <o =2 <p =3 </m>text"]
UNITS = [
    b"<a>x</a>", b"<b k='v' j=\"w\" u=z>text <i/> more</b>\n", b"<c  x = 'q q' ><d/></c> ", b"<!-- note --><e>1</e>", b"<?pi data?><f g=h>t</f>",
    b"< g>lost white space</g>", b"<h a b c>t</h>", b"plain text without tags ", b"<i j='k'>l<m n=\"o\"/>p</i>\r\n\t", b"<q =bad>r</q><s>t</s>",
    b"<u v='w>x</u>' y>z</u>", b"</stray><t>u</t>", b"<x/><y /><z  />",
]


def corpus(n, seed, scale=1, small=False):
    """small: a tenth of the repeats (no document reaches the 16 KiB the replay wants; the lane batches, the memos and the failing runs are
    all there) - what the CPU suite can afford on the emulator"""
    rng0 = np.random.Generator(np.random.PCG64(seed))
    class _R:                                                                              # the generator, with the repeat counts scaled down
        def __getattr__(self, k): return getattr(rng0, k)
        def integers(self, lo, hi=None, **kw):
            v = rng0.integers(lo, hi, **kw)
            return max(1, int(v) // 10) if small and hi is not None and hi >= 200 and not kw else v
    rng = _R()
    out = []
    for k in range(n):
        def unit():
            return b"".join(UNITS[int(j)] for j in rng.integers(0, len(UNITS), size=int(rng.integers(1, 5))))
        u = unit()
        reps = int(rng.integers(200, 900)) * scale
        head = b"<doc>" + unit() * int(rng.integers(0, 4))
        tail = unit() * int(rng.integers(0, 4)) + b"</doc>"
        out.append(head + u * reps + tail)                                                 # plain pump
        cut = int(rng.integers(1, max(2, len(u) - 1)))
        out.append(head + u[cut:] + u * reps + u[:cut] + tail)                             # the period starts inside a tag / text
        out.append(head + u * reps + b"<v w='unterminated " + unit() * 3)                  # a quote that never ends behind the stretch
        out.append(head + u * reps + b"<!-- never closed " + unit() * 3)
        out.append(head + u * (16384 // len(u) + 3) + tail)                                # barely long enough / too short to replay
        out.append((head + u * reps + tail) * 2)                                           # two stretches of the same period
        out.append(head + b"".join(UNITS[int(j)] for j in rng.integers(0, len(UNITS), size=int(rng.integers(300, 1500)) * scale)) + tail)   # no period: tag attempts 64 at a time
        junk = [b"<", b"< ", b"<a ", b"<a b", b"<!", b"<!--", b"<?", b"</", b"</x ", b"='", b"\"", b" > ", b"/>", b"-->", b"?>", b"x", b" ", b"\n", b"=", b"y z"]
        out.append(head + b"".join(junk[int(j)] for j in rng.integers(0, len(junk), size=int(rng.integers(2000, 6000)) * scale)) + tail)     # tag soup: failed attempts, quotes and comments that run far
        w = lambda a, b: bytes(rng.choice(list(b"abcdefghijklmnopqrstuvwxyz0123456789{}[]()."), size=int(rng.integers(a, b))).astype(np.uint8))
        run = b"<" + w(3, 9) + b" " + w(4, 12)                                              # "<name attr<name attr ..." without a '>': every attempt walks the attributes ahead and fails
        out.append(head + (b"<t>x</t>" + w(20, 100)) * 20 + run * (int(rng.integers(1500, 5000)) * scale) + w(3, 9) + b"=" + tail)
        words = lambda n: b" ".join(w(1, 9) + (b"=" + w(1, 5) if rng.random() < 0.1 else b"") for _ in range(n))
        out.append(head + b"".join(b"<" + words(int(rng.integers(20, 3000)) * scale) + rng.choice([b">", b"/>", b" >", b"=>", b""]) + w(0, 30) for _ in range(int(rng.integers(2, 12)))) + tail)   # tags of thousands of attributes, few '<'
        soup = b"".join(rng.choice([b"<", b" ", b"  ", b"='", b"=\"", b"=", b">", b"a", b"bc", b"\n"], p=[.2, .2, .05, .03, .03, .05, .04, .2, .15, .05]) for _ in range(int(rng.integers(3000, 9000)) * scale))
        out.append(head + soup + tail)                                                     # failing runs without a period, quotes that never close
        unit2 = b"<" + w(2, 6) + b"/" + w(5, 30)                                            # names that run over thousands of '<' and '/' (neither ends a name)
        out.append(head + unit2 * (int(rng.integers(1500, 4000)) * scale) + rng.choice([b" x=1>", b">", b"/>", b"", b" "]) + words(30) + tail)
        out.append(head + b"".join(b"<" + w(1, 12) + rng.choice([b"", b"/", b"<", b"'", b"\""]) for _ in range(int(rng.integers(2000, 5000)) * scale)) + rng.choice([b" y>", b"", b"=", b"/>"]) + tail)
        # runs of three and more failing attempts INSIDE a periodic stretch: lane batches while the replay records its template / check
        # period (one accepted tag per batch there).  Its own generator: the documents above stay what they were before this kind came.
        rng2 = np.random.Generator(np.random.PCG64([seed, k, 77]))
        fu = FAILRUN_UNITS[int(rng2.integers(0, len(FAILRUN_UNITS)))]
        reps2 = max(4, int(rng2.integers(300, 900)) * scale // (10 if small else 1))
        cut2 = int(rng2.integers(0, len(fu)))
        out.append(b"<doc>" + fu[cut2:] + fu * reps2 + fu[:cut2] + (b"</doc>", b"<!-- ", b"<v w='")[int(rng2.integers(0, 3))])
        # round 5: tags of thousands of attributes in every syntax the attribute loop knows (erlamsa_sgml.erl:134-160) - the wave-wide machine
        # takes them one attribute per lane (sg_lane_attr): values quoted with blanks, '<', '>', "/>" and '=' inside, unquoted values, blanks around
        # '=', tabs / CR / LF runs, attributes glued to a closing quote, quotes inside names, an '=' where a name should start (the tag fails),
        # tags that end in '>', '/>', ' />' or never, a quote that never closes far into the tag
        rng3 = np.random.Generator(np.random.PCG64([seed, k, 78]))
        w3 = lambda a, b: bytes(rng3.choice(list(b"abcdefghijklmnopqrstuvwxyz0123456789{}[]().-!?"), size=int(rng3.integers(a, b))).astype(np.uint8))
        def attr():
            r = rng3.random()
            ws = [b" ", b"  ", b"\t", b"\n", b"\r\n", b" \t "][int(rng3.integers(0, 6))]
            if r < 0.55: return w3(1, 9) + ws
            if r < 0.65: return w3(1, 6) + b"=" + w3(0, 6) + ws
            if r < 0.75: return w3(1, 6) + b"='" + [w3(0, 9), b"a b  c", b"<x y>", b"/>", b"=", b"\"", b""][int(rng3.integers(0, 7))] + b"'" + ws
            if r < 0.85: return w3(1, 6) + b" = \"" + [w3(0, 9), b"p q", b"<<<", b" ", b"'"][int(rng3.integers(0, 5))] + b"\"" + (ws if rng3.random() < 0.7 else b"")
            if r < 0.90: return w3(1, 4) + b"\"" + w3(0, 4) + ws
            if r < 0.94: return w3(1, 5) + b"/" + w3(0, 3) + ws
            if r < 0.97: return w3(1, 5) + b"<" + w3(0, 5) + ws
            return w3(1, 4) + b"= " + w3(1, 4) + ws
        def bigtag(n):
            body = b"".join(attr() for _ in range(n))
            end = [b">", b"/>", b" />", b"", b"=>", b"='never closed ", b" = ", b">"][int(rng3.integers(0, 8))]
            return b"<" + w3(1, 6) + b" " + body + end
        natt = max(40, int(rng3.integers(300, 6000)) * scale // (40 if small else 1))
        out.append(b"<doc>" + b"".join(bigtag(int(rng3.integers(17, natt))) + w3(0, 40) + (b"</x>" if rng3.random() < 0.3 else b"") for _ in range(int(rng3.integers(1, 7)))) + b"</doc>")
    if small:
        out = [d for k, d in enumerate(out) if k % 15 >= 6]                                  # the kinds without a period: what the lane batches are for (the periodic ones need their full length anyway)
    return out


def run(n=1, seed=1, scale=1, pats="od,nd,bu", verbose=True, small=False):
    inputs = corpus(n, seed, scale, small)
    data, off = po.pack(inputs)
    res = {}
    modes = (("replay", 0), ("walk", EH_FLAG_SGML_NO_REPLAY | EH_FLAG_SGML_NO_LANES), ("lanes", EH_FLAG_SGML_NO_REPLAY), ("nolanes", EH_FLAG_SGML_NO_LANES))
    for name, flags in modes[:2] if small else modes:
        t = time.time()
        e = ea.Engine(0)
        e.configure(mutations="sgm", patterns=pats, max_case_bytes=64 << 20, flags=flags)
        e.upload_corpus(data, off)
        e.fuzz_batch(seed=(seed, 6, 2))
        got, st = e.download()
        dr, _ = e.diag()
        res[name] = (got, st, dr, time.time() - t)
        e.close()
    import util
    t = time.time()
    o = util.oracle_live(data, off, seed=(seed, 6, 2), mutations="sgm", patterns=pats, max_case_bytes=256 << 20, chunk=1)
    to = time.time() - t
    bad = 0
    a, b = res["replay"], res["walk"]
    for i in range(len(inputs)):
        if a[1][i] in (2, 3) or o.status[i] in (2, 3):
            continue
        ok_o = a[0][i] == o.outs[i] and a[1][i] == o.status[i] and (a[1][i] != 0 or a[2][i] == o.draws[i])
        ok_n = all(a[0][i] == x[0][i] and a[1][i] == x[1][i] and a[2][i] == x[2][i] for x in res.values())
        if not (ok_o and ok_n):
            bad += 1
            if verbose and bad <= 8:
                print("case %d (kind %d, len %d): replay vs oracle %s, replay vs walk %s; status %d/%d/%d draws %d/%d/%d len %d/%d/%d" % (
                    i, i % 15, len(inputs[i]), ok_o, ok_n, a[1][i], b[1][i], o.status[i], a[2][i], b[2][i], o.draws[i], len(a[0][i]), len(b[0][i]), len(o.outs[i])))
    if verbose:
        print("cases %d bad %d; %s, oracle %.1f s; input bytes %d" % (len(inputs), bad, ", ".join("%s %.1f s" % (
            {"replay": "replay+lanes", "walk": "tag by tag", "lanes": "lanes only", "nolanes": "replay only"}[k], v[3]) for k, v in res.items()), to, sum(map(len, inputs))))
    return len(inputs), bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    total, bad = run(n, seed, scale, small=len(sys.argv) > 4 and sys.argv[4] == "small")
    sys.exit(1 if bad else 0)
