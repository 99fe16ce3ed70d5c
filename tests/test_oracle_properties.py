"""Oracle pinning, part 3: the reference's own eunit properties
(/root/reference/src/erlamsa_mutations_test.erl) re-expressed against the oracle.  The reference
seeds from now(); here every run loops over explicit seeds."""
import re

import numpy as np

import pyoracle as po


def _tries(name, data, pred, n, base=0):
    for k in range(n):
        d, out, _ = po.run_mutator(name, (base + k, 2 * k + 1, 3 * k + 7), data)
        if d is not None and pred(out):
            return True
    return False


def test_sed_num():                       # erlamsa_mutations_test.erl:74-77
    assert _tries("num", b" 100 + 100 + 100 ", lambda o: b"101" in o, 1500)


def test_string_lexer_roundtrip():        # :84-93 (unlex(lex(X)) =:= X)
    rng = np.random.Generator(np.random.PCG64(4))
    assert po.lex_roundtrip(bytes([233, 39, 39, 97, 97, 97, 0]))[1] == bytes([233, 39, 39, 97, 97, 97, 0])
    for _ in range(10000):
        n = int(rng.integers(0, 42))
        t = rng.integers(0, 8, size=n)
        s = bytes([92 if x == 0 else 34 if x == 1 else 39 if x == 2 else 0 if x == 3 else int(rng.integers(0, 256)) if x == 4 else 97 for x in t])
        cnt, out = po.lex_roundtrip(s)
        assert cnt >= 0 and out == s


DASHES = b"-" * 40 + b'""' + b"-" * 50


def test_ascii_bad():                     # :96-100
    rx = re.compile(rb'^-*".*[%|a].*"-*$', re.S)
    assert _tries("ab", DASHES, lambda o: rx.match(o) is not None, 50)


def test_ascii_delimeter():               # :102-109
    rx = re.compile(rb'^-*"-*$', re.S)
    assert _tries("ad", DASHES, lambda o: rx.match(o) is not None, 50)


def test_sed_fuse_this():                 # :115-119
    src = b"kittenslartibartfasterthaneelslartibartfastenyourseatbelts"
    assert _tries("ft", src, lambda o: o == b"kittenslartibartfastenyourseatbelts", 500)


def test_sed_tree_stutter():              # :126-130
    assert _tries("tr", b"(x (Y x))", lambda o: o == b"(x (x (x (x (Y x)))))", 500)


def _distinct(name, data, n):
    outs = set()
    for k in range(n):
        d, out, _ = po.run_mutator(name, (k + 1, k * k + 3, 17 * k + 5), data)
        outs.add(out)
    return outs


def test_sed_tree_dup_swap_counts():      # :145-152: number of DISTINCT outputs
    assert len(_distinct("tr2", b"(a)", 20)) == 1          # "(a)(a)" only
    assert len(_distinct("tr2", b"(a) (b)", 200)) == 2
    assert len(_distinct("ts1", b"(a) (b) (c)", 400)) == 6
    assert len(_distinct("ts2", b"(a) (b) (c)", 400)) == 3


def test_line_mutators():                 # :167-217
    d, out, _ = po.run_mutator("lr2", (1, 2, 3), b"1\n")
    assert out == b"1\n1\n"
    d, out, _ = po.run_mutator("ls", (1, 2, 3), b"A\n B\n")
    assert out == b" B\nA\n"
    outs = _distinct("lri", b"A\nB\n", 200)
    assert outs == {b"A\nA\n", b"A\nB\n", b"B\nB\n"}
    src = b"1\n 2\n  3\n   4\n"
    for k in range(100):
        d, out, _ = po.run_mutator("ld", (k, 1, 1), src)
        assert out.count(b"\n") == 3
        d, out, _ = po.run_mutator("lds", (k, 1, 1), src)
        assert out.count(b"\n") < 4
        d, out, _ = po.run_mutator("lp", (k, 1, 1), src)
        assert sorted(out.split(b"\n")) == sorted(src.split(b"\n"))
        d, out, _ = po.run_mutator("lr", (k, 1, 1), src)
        assert out.count(b"\n") > 4


def test_st_line_ins_single_line():       # :223-228, the only fixed seed {1,2,3} of the reference
    d, out, _ = po.run_mutator("lis", (1, 2, 3), b"Hello\n")
    assert out == b"Hello\nHello\n"
    h = len(out) // 2
    assert out[:h] == out[h:]


def test_byte_mutators_size_and_sum():    # :247-310
    rng = np.random.Generator(np.random.PCG64(6))
    for k in range(300):
        blk = rng.integers(0, 256, size=int(rng.integers(1, 4097)), dtype=np.uint8).tobytes()
        seed = (k, k + 1, k + 2)
        assert len(po.run_mutator("bd", seed, blk)[1]) == len(blk) - 1
        assert len(po.run_mutator("bi", seed, blk)[1]) == len(blk) + 1
        assert len(po.run_mutator("br", seed, blk)[1]) == len(blk) + 1
        for nm, delta in (("bei", 1), ("bed", -1)):
            out = po.run_mutator(nm, seed, blk)[1]
            assert len(out) == len(blk)
            assert (sum(out) - sum(blk)) % 256 == delta % 256
        out = po.run_mutator("bf", seed, blk)[1]
        diff = [a ^ b for a, b in zip(out, blk) if a != b]
        assert len(diff) == 1 and bin(diff[0]).count("1") == 1
        assert len(po.run_mutator("sd", seed, blk)[1]) < len(blk)
        assert len(po.run_mutator("sr", seed, blk)[1]) > len(blk)
        out = po.run_mutator("sp", seed, blk)[1]
        assert sorted(out) == sorted(blk)


def test_utf8_mutators():                 # untested in the reference (:7-11); structural properties
    assert po.run_mutator("uw", (1, 1, 1), b"\x20")[1] == b"\xc0\xa0"
    assert po.run_mutator("uw", (1, 1, 1), b"\x7f")[1] == b"\x7f"
    for k in range(50):
        out = po.run_mutator("ui", (k, 2, 3), b"abc")[1]
        assert len(out) > 3 and out[0:1] == b"a"


def test_json_mutator_properties():
    """erlamsa_json:json_mutate/2 restated in the oracle (branch work: not on the GPU yet).  No reference
    test pins bytes for it; these are the structural properties of the reference code itself: non-JSON
    input fails with delta -1 and is left alone, structural mutations of a valid document re-tokenize,
    and the mutator is a pure function of the seed."""
    import json
    import pyoracle as po
    doc = b'{"a": 1, "b": [true, false, null, "x"], "c": {"d": "hello", "e": -12}}'
    changed = 0
    for s in range(1, 120):
        seed = (s, s * 7 + 1, s * 13 + 5)
        d1, out1, _ = po.run_mutator("js", seed, doc)
        d2, out2, _ = po.run_mutator("js", seed, doc)
        assert (d1, out1) == (d2, out2)
        if d1 is None:
            continue
        assert out1 != doc                      # whitespace is dropped by the tokenizer, so a valid document always changes
        changed += 1
        if d1 == 1 and b"://" not in out1:
            try:
                json.loads(out1.decode("latin1"))
            except ValueError:
                pass                             # inner text mutations may break the syntax; structural ones must not crash the oracle
    assert changed > 100
    for bad in (b"not json at all", b'{"a" "b"}', b"]", b'"str"', b"42"):
        d, out, _ = po.run_mutator("js", (1, 2, 3), bad)
        assert d == -1 and out == bad


def test_sgml_verify_roundtrip():
    """erlamsa_sgml:verify/1 (:768-773): fold_ast(parse(Str)) =:= Str for canonically written documents;
    same for erlamsa_json fold_ast(tokenize(Bin)) on compact JSON."""
    from erlamsa_amd import synth
    for d in synth.sgml_docs(400, seed=11):
        rc, out, (n, nt, _) = po.parse_fold("sgml", d)
        assert rc == 0 and out == d and n >= 1 and nt <= n
    for d in synth.json_docs(400, seed=11):
        rc, out, (n, nt, nv) = po.parse_fold("json", d)
        assert rc == 0 and out == d and nv <= n and nt <= n


def test_sgml_tokenizer_quirks():
    """Behaviour that follows from the reference's clauses and is easy to get wrong."""
    # leading text before the first '<' is dropped (tz(nil, ..) :100-101); ws after '<' and inside tags is dropped
    assert po.parse_fold("sgml", b"junk < a  b = 'c'  d >t</a >")[1] == b"<a b='c' d>t</a>"
    # a '<' whose tag does not parse becomes text, minus the white space that followed it (:79-96)
    assert po.parse_fold("sgml", b"<a>1 <  2 = 3</a>")[1] == b"<a>1 <2 = 3</a>"
    # ?ok(X) is always true: quotes, '<' and '/' are name characters
    assert po.parse_fold("sgml", b"<a/b c\"d>x")[1] == b"<a/b c\"d>x"
    # empty attribute values lose their quotes (fold_params :297-298)
    assert po.parse_fold("sgml", b"<a b=\"\" c=''>")[1] == b"<a b c>"
    # close of an outer tag closes it early; the inner open tag is kept as a bare {open,..} (:226-235)
    rc, out, (n, nt, _) = po.parse_fold("sgml", b"<a><b>x</a>y</b>")
    assert (rc, out, n, nt) == (0, b"<a><b>x</a>y</b>", 5, 1)
    # unterminated comment as the FIRST tag: function_clause outside any try -> the worker dies
    assert po.parse_fold("sgml", b"<!-- never closed")[0] == -2
    assert po.run_mutator("sgm", (1, 2, 3), b"<!-- never closed")[0] is None
    # ... but inside the text state it is caught and the '<' becomes text
    assert po.parse_fold("sgml", b"<a><!-- never closed")[1] == b"<a><!-- never closed"
    # binarish blocks and blocks without '<' fail with delta -1 and no draw
    assert po.run_mutator("sgm", (1, 2, 3), b"\x00<a>")[0] == -1
    assert po.run_mutator("sgm", (1, 2, 3), b"plain")[:2] == (-1, b"plain")
    # case-insensitive pairing, the close tag keeps its spelling
    assert po.parse_fold("sgml", b"<DIV>x</div>") == (0, b"<DIV>x</div>", (2, 1, 0))


def test_sgml_mutator_properties():
    """sgml_mutate/2 :739-757 restated: pure function of the seed, every output parses again, and the
    twelve mutation kinds are all reachable on a small document."""
    doc = b"<?xml v='1'?><r xmlns='u' a=b><i>one</i><i k=\"v\">two 22</i><e /><!-- c --></r>tail"
    outs = set()
    for s in range(1, 400):
        seed = (s, 3 * s + 1, 5 * s + 2)
        d1, o1, _ = po.run_mutator("sgm", seed, doc)
        assert (d1, o1) == po.run_mutator("sgm", seed, doc)[:2]
        assert d1 is not None
        outs.add(o1)
        if o1 != doc and b"<" in o1 and not o1.startswith(b"\x00"):
            assert po.parse_fold("sgml", o1)[0] in (0, -1)
    assert len(outs) > 100
    assert any(o.count(b"<r ") > 1 for o in outs)            # pump / dup / repeat
    assert any(b"xmlns:xsi" in o or b"u http://" in o or b"'http://localhost:51234/'" in o for o in outs)   # xmlns features


def test_counter_corpus_numpy_and_torch_forms_agree():
    """synth.counter (numpy: tests, oracle side) and synth.counter_torch (what bench.py --corpus counter writes into the HBM arena)
    are the same closed form of (seed, row, byte); rows are independent of how they are batched; all four row kinds occur."""
    import torch
    from erlamsa_amd import synth
    a = synth.counter(range(40, 340), 1024, seed=11)
    t = torch.zeros(400 * 1024, dtype=torch.uint8)
    synth.counter_torch(t, 40, 300, 1024, seed=11, chunk_rows=37)
    assert (t.numpy().reshape(400, 1024)[40:340] == a).all() and not t.numpy()[:40 * 1024].any()
    assert (synth.counter([77], 1024, seed=11)[0] == a[37]).all()
    framed = (a[:, :4] == np.frombuffer((1020).to_bytes(4, "big"), dtype=np.uint8)).all(axis=1)
    texty = ((a >= 32) & (a < 127) | (a == 10)).all(axis=1)
    assert framed.sum() > 30 and texty.sum() > 60 and (~framed & ~texty).sum() > 30
