"""GPU parity: HIP engine (through the C ABI) vs the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

BYTE_MUTAS = "bd,bf,bi"
BYTE_ALL = "bd,bei,bed,bf,bi,ber,br"
SEQ = "sp,sr,sd,snand,srnd"


def _compare(inputs, mutations, patterns, seed=(1, 2, 3), generators=None, first_case=1, max_report=5, max_skipped=1,
             oracle_cap=8 << 20, engine_cap=0, work=0, live=False):
    """live=True: the oracle runs here and now on the host's threads (util.oracle_live) instead of through the digest cache that
    travels with the tree - the tests of the default tables and of the bench workload do, so that their green means "engine ==
    the oracle as built on this box", not "engine == bytes hashed elsewhere"."""
    import pyoracle as po
    data, off = po.pack(inputs)
    okw = dict(seed=seed, mutations=mutations, patterns=patterns, generators=generators, first_case=first_case,
               max_case_bytes=oracle_cap, max_case_work=work)
    if live and util.priming():
        pytest.skip("live oracle: nothing to prime")
    ora = util.oracle_live(data, off, **okw) if live else util.oracle_batch(data, off, **okw)
    assert not (ora.status == 3).any(), "the oracle says UNSUPPORTED for an input that is not a zip archive"
    if util.priming():
        pytest.skip("oracle cache primed")
    import erlamsa_amd as ea
    eng = ea.Engine(0)
    eng.configure(mutations=mutations, patterns=patterns, generators=generators, max_case_bytes=engine_cap, max_case_work=work)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=seed, first_case=first_case)
    got, gst = eng.download()
    gdr, glm = eng.diag()
    eng.close()
    wst, wdr = ora.status, ora.draws

    def diff():
        bad, skipped = [], 0
        for i in range(len(inputs)):
            # engine-only statuses (work-area cap, paths the GPU build reports as UNSUPPORTED) have no
            # counterpart in the reference semantics; they are tolerated in small numbers and counted
            if gst[i] in (2, 3) or wst[i] in (2, 3):
                skipped += 1
                continue
            if gst[i] != wst[i] or not ora.same(i, got[i]):
                bad.append(i)
        return bad, skipped

    bad, skipped = diff()
    if bad and ora.outs is None:
        ora = util.oracle_batch(data, off, live=True, **okw)      # digests only: recompute for the report
    msg = "\n".join("case %d: first diff at %d, len gpu %d vs oracle %d, status %d vs %d, draws %d vs %d, trace: %s"
                    % (i, util.first_diff(got[i], ora.outs[i]), len(got[i]), len(ora.outs[i]), int(gst[i]), int(wst[i]), int(gdr[i]), int(wdr[i]),
                       ora.trace[i] if i < len(ora.trace) else "") for i in bad[:max_report])
    assert not bad, "%d/%d cases differ\n%s" % (len(bad), len(inputs), msg)
    # EH_CASE_UNSUPPORTED is what is left for zip archives with features outside the restated prim_zip: none of these corpora holds
    # an archive, and gzip / zlib look-alikes (a random two-byte header passes zlib's check once in ~2 000 inputs) are decoded now
    assert not (gst == 3).any() and not (wst == 3).any(), "EH_CASE_UNSUPPORTED on an input that is not a zip archive"
    if os.environ.get("EH_REPORT_SKIPS"):                              # observed counts per test, for setting max_skipped
        with open(os.environ["EH_REPORT_SKIPS"], "a") as fh:
            fh.write("%s skipped %d of %d (engine-only status: %d, oracle cap: %d) allowed %d\n" % (
                os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], skipped, len(inputs), int(((gst == 2) | (gst == 3)).sum()), int(((wst == 2) | (wst == 3)).sum()), max_skipped))
    # max_skipped is a COUNT: what the GPU runs of round 6 showed for the test (gpurun_out/r06j/skips.txt = profiles/r06_observed_skips.txt)
    # plus one - two where the engine's own work-area cap is among the reasons; a test without an entry there skipped nothing and
    # may skip one.  Nearly every skipped case is one the ORACLE's test cap cut (oracle_cap).
    assert skipped <= max_skipped, "%d cases skipped as overflow/unsupported (allowed: %d)" % (skipped, max_skipped)
    eng_only = int(((gst == 2) | (gst == 3)).sum())
    assert eng_only <= max_skipped, "%d cases ended with an engine-only status" % eng_only
    ok = (wst == 0) & (gst == 0)
    assert (gdr[ok] == wdr[ok]).all(), "draw counts differ"


def test_c2_byte_mutators_od():
    """BASELINE config 2: 1024 x 256 B, bd/bf/bi, pattern od."""
    _compare(util.corpus_uniform(1024, 256), BYTE_MUTAS, "od")


@pytest.mark.parametrize("seed", [(1, 2, 3), (42, 4242, 424242), (0, 0, 0), (30268, 30306, 30322)])
def test_byte_seq_all_patterns(seed):
    _compare(util.corpus_uniform(512, 256, seed=seed[0] + 7), BYTE_ALL + "," + SEQ + ",uw,ui,nil", "od,nd,bu,co,nu", seed=seed)


def test_ragged_and_empty_inputs():
    rng = np.random.Generator(np.random.PCG64(5))
    inputs = [b"", b"a", b"ab", b"\x00", b"Hello erlamsa!\n"] + [rng.integers(0, 256, size=int(s), dtype=np.uint8).tobytes()
                                                                 for s in rng.integers(0, 700, size=400)]
    _compare(inputs, BYTE_ALL + "," + SEQ + ",uw,ui,nil", "od,nd,bu,co,nu")


def test_random_generator_only():
    _compare(util.corpus_uniform(256, 64), BYTE_ALL + ",sd,sr", "od,nd,bu", generators="random=1", max_skipped=2)


def test_first_case_offset_is_a_pure_function_of_index():
    """Cutting a run into calls must not change results."""
    import pyoracle as po
    import erlamsa_amd as ea
    inputs = util.corpus_uniform(300, 128)
    data, off = po.pack(inputs)
    eng = ea.Engine(0)
    eng.configure(mutations=BYTE_ALL + "," + SEQ, patterns="od,nd,bu")
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=(7, 8, 9))
    whole, _ = eng.download()
    eng.fuzz_batch(seed=(7, 8, 9), first_case=101, corpus_first=100, n=200)
    part, _ = eng.download()
    assert part == whole[100:]
    eng.close()


def test_4k_blocks_sequences():
    _compare(util.corpus_uniform(256, 4096), BYTE_ALL + "," + SEQ + ",uw,ui", "od,nd,bu")


LINES = "ld,lds,lr2,lri,lr,ls,lp,lis,lrs"


def _texty(n, size, seed):
    from erlamsa_amd import synth
    m = synth.mixed(n * 2, size, seed=seed)
    return [bytes(r) for r in m]


@pytest.mark.parametrize("seed", [(1, 2, 3), (11, 22, 33)])
def test_lines_and_num_mixed_corpus(seed):
    _compare(_texty(300, 1024, seed[0]), LINES + ",num," + BYTE_ALL + ",sd", "od,nd,bu", seed=seed)


def test_num_edge_cases():
    inputs = [b" 100 + 100 + 100 ", b"-1", b"--5 x -0 007", b"9" * 400 + b"\n", b"x" * 2048, b"1\n" * 1100,
              b"18446744073709551615 340282366920938463463374607431768211455\n", b"abc", b"0", b"1-2-3",
              b"12345678901234567890123456789012345678901234567890" * 7, b"A\n B\n", b"1\n"] * 40
    _compare(inputs, "num", "od,nd,bu")
    _compare(inputs, "num," + LINES, "od,nd,bu", seed=(5, 6, 7), max_skipped=2)


def test_lines_small_texts():
    rng = np.random.Generator(np.random.PCG64(9))
    inputs = []
    for i in range(600):
        nl = int(rng.integers(0, 12))
        parts = [bytes(rng.integers(97, 123, size=int(rng.integers(0, 9)), dtype=np.uint8)) for _ in range(nl + 1)]
        s = b"\n".join(parts)
        if rng.random() < 0.5:
            s += b"\n"
        inputs.append(s)
    _compare(inputs, LINES, "od,nd,bu")


LEXERS = "ab,ad,uri,b64,zip"


def _lexy_inputs(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    frags = [b"hello world ", b"\"quoted text\" ", b"'single' ", b"key=value; ", b"http://example.com/a/b/c?x=1 ", b"file:///etc/passwd ",
             b"ftp://host ", b"back\\\"slash ", b"\x00\x01\x02", b"\xff\xfe", b"path/to/file.txt\n", b"AAAA%d%n", b"tab\tsep\r\n",
             b"://", b"a://b", b"unterminated \"quote ", b"it's ", b"1234567 ", b"x-" * 20, b"\"\"", b"PK\x03\x04data"]
    out = []
    for _ in range(n):
        k = int(rng.integers(1, 30))
        s = b"".join(frags[int(i)] for i in rng.integers(0, len(frags), size=k))
        if rng.random() < 0.3:
            s += rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8).tobytes()
        out.append(s)
    return out


@pytest.mark.parametrize("seed", [(1, 2, 3), (9, 8, 7)])
def test_lexer_mutators(seed):
    _compare(_lexy_inputs(500, seed[0]), LEXERS, "od,nd,bu", seed=seed)


def test_lexer_mutators_on_mixed_corpus():
    _compare(_texty(200, 2048, 77), LEXERS + ",bd,bf", "od,nd,bu")


def test_lexer_with_everything_else():
    _compare(_lexy_inputs(300, 5) + _texty(100, 1024, 5), LEXERS + "," + LINES + ",num," + BYTE_ALL + "," + SEQ + ",uw,ui,nil", "od,nd,bu")


def _docs(n, seed):
    from erlamsa_amd import synth
    return synth.sgml_docs(n // 2, seed=seed) + synth.json_docs(n - n // 2, seed=seed + 1)


ADVERSARIAL_DOCS = [
    b'{"a" "b"', b'{"a" "b":1}', b'[1, 2', b'[1,2]x', b'{"a":1}', b'{1:2,[3]:{"x":null}}', b'"junk', b'true', b'truex', b'tru', b'nul',
    b'[true,false,null]', b'  [ ]  ', b'{}', b'{ }', b'[[[[]]]]', b'{"k":[1,{"z":"dGhpcyBpcyBiYXNlNjQ="},"<a>x</a>"],"n":-0012,"m":+5,"e":1e5}',
    b'[', b'{', b'{"a":', b'{"a"', b'{"a":1,', b'[1,,2]', b'[1 2]', b'"a":1', b'x', b'12 ', b' 12', b'"s"', b'[null]',
    b'{"a":{"b":{"c":[1,2,3,{"d":"e"}]}}}', b'{"a":1}{"b":2}', b'[1]]', b':', b',', b']', b'{"a":1 "b":2}', b'{"a"::1}',
    b'[{"a":1},{"a":1},{"a":1}]', b'{"long":"' + b'x' * 300 + b'"}', b'[' + b','.join(b'%d' % i for i in range(200)) + b']',
    b'<a', b'<a>', b'< a >', b'<>', b'</>', b'<a/>', b'<a b/>', b'<a b=/>', b'<a b= c>', b'<a =b>', b'<a b="c>', b"<a b='c'd>", b'<!-->',
    b'<!---->', b'<!-- x --', b'<a><!-- x --', b'<?x?>', b'<?x', b'<!x', b'<a></b></a>', b'<a><b><a></a></b></a>', b'<A></a><a></A>',
    b'x<a>y<  z<b>w</b>', b'<a\n b\t=\r"v"\n>t</a\n>', b'<a b=c/>', b'<a b=c />', b'<a/b>', b'<a b="">', b'<a>1<b>2<c>3</a>4</c>5</b>6',
    b'<p>' * 70 + b'x', b'<p>' * 70 + b'</p>' * 70, b'<a x=1 y=2 z=3 xmlns=u xmlns:q="v">t</a>', b'text only < not a tag',
    b'<a>' + b'w' * 5000 + b'</a>', b'<a ' + b'k=v ' * 100 + b'>x</a>', b''.join(b'<i n="%d">v%d</i>' % (i, i) for i in range(150)),
    b'<\xc9L>x</\xe9l>', b'<a>\x00</a>', b'<r><![CDATA[ <x> ]]></r>',
    # retried tags that scan far: names running over many events, terminators that never come
    b'<a>' + b'<' * 700 + b' =x>' + b'<' * 300 + b' y=1>t', b'<a>' + b'</' * 400 + b'x >' + b'</' * 200 + b'/>/>/> >', b'<a>' + b'<!--' * 200 + b'x',
    b'<a>' + b'<!--' * 200 + b'-->y', b'<a>' + b'<!x' * 300, b'<a>' + b'<?x' * 300 + b'?', b'<a>' + b"<a b='" * 150, b'<a>' + b'<a b="' * 151 + b'>',
    b'<a>' + b'<b c=d/' * 200 + b'>' + b'<' * 200 + b'/>',
    # base64 of base64 of base64 of "<a>hello world</a>": nested scheduler calls three deep
    b'UEVFK2FHVnNiRzhnZDI5eWJHUThMMkUr', b'say "PGE+aGVsbG8gd29ybGQ8L2E+" and eyJrIjoiYUdWc2JHOD0ifQ== twice',
]


@pytest.mark.parametrize("seed", [(1, 2, 3), (4, 5, 6)])
def test_sgml_json_documents(seed):
    """sgm (erlamsa_sgml) and js (erlamsa_json) on well-formed documents: every mutation kind incl. the inner text
    mutations that re-enter the scheduler with inner_mutations(sgml | json)."""
    _compare(_docs(400, seed[0]), "sgm,js", "od,nd,bu", seed=seed, max_skipped=3, oracle_cap=4 << 20, engine_cap=4 << 20)


def test_sgml_json_adversarial():
    """Corner cases of the two tokenizers (tz/2 clause order, `catch _:_` in tokenize/1, contexts that never produce a
    token, crashes outside any try), 12 runs each."""
    _compare([d for d in ADVERSARIAL_DOCS for _ in range(12)], "sgm,js,b64", "od", seed=(2, 7, 1), oracle_cap=4 << 20, engine_cap=4 << 20)


def test_b64_nested_default_table():
    """base64_mutator success path: decoded chunks are mutated by a fresh mutators_mutator over the whole default table."""
    _compare(_docs(300, 9) + [d for d in ADVERSARIAL_DOCS[-2:] for _ in range(50)], "b64", "od,nd,bu", max_skipped=2, oracle_cap=4 << 20, engine_cap=4 << 20)


@pytest.mark.parametrize("kind", ["docs", "mixed"])
def test_default_tables(kind):
    """eh_options.mutations = patterns = NULL: the reference's full default tables (41 mutators, 10 patterns)."""
    inputs = _docs(240, 3) if kind == "docs" else _texty(150, 1200, 6)
    _compare(inputs, None, None, seed=(3, 4, 5), max_skipped=7, oracle_cap=4 << 20, engine_cap=4 << 20, live=True)


TREES = "tr2,td,ts1,ts2,tr"


def _bracket_inputs(n, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    frags = [b"(", b")", b"[", b"]", b"<", b">", b"{", b"}", b"\"", b"'", b"x", b"Y", b" ", b"ab", b"()", b"(x)", b"(x (Y x))", b"\n", b"foo", b"[1,2]"]
    out = [b"(x (Y x))", b"()", b"(", b")", b"(()", b"(())", b"((x)(x))", b"\"a\"\"a\"", b"(a)(a)(a)", b"[(a)](a)", b"x"]
    for _ in range(n):
        k = int(rng.integers(1, 60))
        out.append(b"".join(frags[int(i)] for i in rng.integers(0, len(frags), size=k)))
    return out


@pytest.mark.parametrize("seed", [(1, 2, 3), (3, 1, 4), (2, 7, 1)])
def test_tree_mutators(seed):
    _compare(_bracket_inputs(300, seed[1]), TREES, "od,nd,bu", seed=seed, max_skipped=15, oracle_cap=4 << 20, engine_cap=4 << 20)   # (tr stutters beyond the 4 MiB test cap: 13 of 311 for seed 0)


def test_tree_mutators_mixed_corpus():
    _compare(_texty(150, 2048, 31), TREES + ",bd", "od,nd,bu", max_skipped=10, oracle_cap=4 << 20, engine_cap=4 << 20)


def _framed_inputs(n, size, seed):
    from erlamsa_amd import synth
    rng = np.random.Generator(np.random.PCG64(seed))
    m = synth._framed(rng, n, size)
    return [bytes(r) for r in m]


def test_len_mutator():
    ins = _framed_inputs(200, 300, 1) + _framed_inputs(100, 64, 2) + [b"\x00\x05hello", b"\x03abc", b"abc", b"\x00\x00\x00\x04abcd\x00"] * 10
    _compare(ins, "len", "od,nd,bu", max_skipped=1, oracle_cap=1 << 20, engine_cap=4 << 20)
    _compare(ins + util.corpus_uniform(100, 700), "len,bd,bf,sd", "od,nd,bu", seed=(4, 4, 4), max_skipped=1, oracle_cap=1 << 20, engine_cap=4 << 20)


@pytest.mark.parametrize("pats", ["sk", "sz", "cs", "ar", "cp", "od,nd,bu,sk,sz,cs,ar,cp,co,nu"])
def test_complex_patterns(pats):
    ins = _framed_inputs(150, 400, 3) + util.corpus_uniform(100, 300, seed=8) + _texty(60, 512, 9)
    _compare(ins, BYTE_ALL + ",sd,sr,num,ld", pats, max_skipped=3, oracle_cap=1 << 20, engine_cap=4 << 20)


FUSE = "ft,fn,fo"


@pytest.mark.parametrize("seed", [(1, 2, 3), (6, 6, 6)])
def test_fuse_mutators(seed):
    ins = [b"kittenslartibartfasterthaneelslartibartfastenyourseatbelts", b"a", b"ab", b"aaaaaaaa", b"abcabcabc", b""] * 10
    ins += util.corpus_uniform(80, 200, seed=seed[0]) + _texty(60, 300, seed[1]) + [b"xy" * 300, b"\x00" * 500]
    _compare(ins, FUSE, "od,nd,bu", seed=seed, max_skipped=1, oracle_cap=1 << 20, engine_cap=8 << 20)


def test_fuse_with_other_mutators_and_blocks():
    ins = util.corpus_uniform(60, 3000, seed=3) + _texty(60, 2500, 4)
    _compare(ins, FUSE + ",num,bd,sd,ld", "od,nd,bu", max_skipped=1, oracle_cap=1 << 20, engine_cap=8 << 20)


PROD = "uw,ui,ab,ad,tr2,td,num,ts1,tr,ts2,bd,bei,bed,bf,bi,ber,br,sp,sr,sd,snand,srnd,ld,lds,lr2,lri,lr,ls,lp,lis,lrs,len,uri,zip,nil"


def test_bench_workload_sample_with_bench_limits():
    """The first 8192 rows of the bench corpus generator (BASELINE configs[2], mixed-binary 4 KiB seeds) with
    the bench's mutator set, patterns and limits: this is where blocks grow to megabytes (sr/lr/tr chains)
    and the tree/lexer paths see large inputs."""
    from erlamsa_amd import synth
    mat = synth.mixed(8192, 4096)
    _compare([mat[i].tobytes() for i in range(mat.shape[0])], PROD, "od,nd,bu", max_skipped=25,
             oracle_cap=8 << 20, engine_cap=8 << 20, work=8 << 20)


# EH_SET_OVERFLOW sites (csrc/) that are limits of the ENGINE on a single result, not bugs: 102 the largest work area
# (big_case_bytes) is exhausted, 802 a tree stutter (tr) whose result k^reps x |node| exceeds it (the reference builds the
# same binary until its 256 MB process guard truncates the stutter, erlamsa_mutations.erl:978-984), 803 / 103 a single
# result of 4 GiB or more.
ENGINE_LIMIT_SITES = (102, 103, 802, 803)


def test_bench_workload_full_table_vs_oracle():
    """THE WORKLOAD bench.py TIMES, against the oracle: BASELINE configs[2] — synth.mixed(65536, 4096), the reference's full
    default mutator table (41 entries), patterns od,nd,bu, no work budget, max_case_bytes 4 MiB / big_case_bytes 1 GiB —
    run as one pass of 65 536 cases; compared with tests/golden/bench_r03.npz (made by tests/golden/make_bench_golden.py
    from the oracle with the same 1 GiB cap) on rows 0..4095 and on the 214 heaviest cases of the pass (round 3's 200 and what round 4's last survey added)
    (tests/golden/bench_heavy_cases.json: multi-megabyte blocks under fuse / sgm / b64 / tree mutators, outputs up to
    1 GB): status, PRNG draw count, length and SHA-1 of every output.  A case may only end with an engine-only status when
    the capacity check that gave up is one of ENGINE_LIMIT_SITES — and then the oracle, under the same cap, must agree."""
    import hashlib
    import os
    if util.priming():
        pytest.skip("golden file, no live oracle")
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bench_r03.npz"))
    import erlamsa_amd as ea
    from erlamsa_amd import synth
    n = 65536
    mat = synth.mixed(n, 4096)
    data, off = synth.as_arena(mat)
    eng = ea.Engine(0)
    eng.configure(patterns="od,nd,bu", max_case_bytes=4 << 20, big_case_bytes=1 << 30, out_capacity=40 << 30)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=(1, 2, 3))
    st = eng.status(); draws, lm = eng.diag(); lens = eng.lens()
    assert (st != 4).all(), "output arena too small"
    limited = np.nonzero(st == 2)[0]
    assert all(-int(lm[i]) in ENGINE_LIMIT_SITES for i in limited), "a case overflowed at a site that is not a single-result limit: %s" % \
        sorted(set(-int(lm[i]) for i in limited))
    assert len(limited) <= n // 400 and (st == 3).sum() == 0 and (st == 5).sum() == 0
    bad = []
    for k, i in enumerate(z["idx"]):
        i = int(i)
        if int(st[i]) != int(z["status"][k]):
            bad.append((i, "status %d vs oracle %d (site %d)" % (st[i], z["status"][k], -lm[i] if st[i] == 2 else 0)))
            continue
        if st[i] != 0:
            continue
        if int(lens[i]) != int(z["lens"][k]) or int(draws[i]) != int(z["draws"][k]):
            bad.append((i, "len %d vs %d, draws %d vs %d" % (lens[i], z["lens"][k], draws[i], z["draws"][k])))
        elif hashlib.sha1(eng.fetch(i, int(lens[i]))).digest() != z["sha1"][k].tobytes():
            bad.append((i, "bytes differ (len %d)" % lens[i]))
    # ... and LIVE: rows 0..2047 through the oracle as built on this box, on the host's threads (no golden file, no digest cache)
    m = 2048
    ora = util.oracle_live(data[:m * 4096], off[:m + 1], seed=(1, 2, 3), patterns="od,nd,bu", max_case_bytes=1 << 30, max_case_seconds=20.0)
    nlive = 0
    for i in range(m):
        if st[i] in (2, 3) or ora.status[i] in (2, 3, 6):
            continue
        nlive += 1
        if int(st[i]) != int(ora.status[i]) or int(lens[i]) != len(ora.outs[i]) or (st[i] == 0 and int(draws[i]) != int(ora.draws[i])) \
                or eng.fetch(i, int(lens[i])) != ora.outs[i]:
            bad.append((i, "differs from the live oracle run"))
    eng.close()
    assert nlive >= m - 16, "only %d of %d cases could be compared with the live oracle run" % (nlive, m)
    assert not bad, "%d of %d cases differ from the oracle: %s" % (len(bad), len(z["idx"]) + m, bad[:8])


def test_results_do_not_depend_on_slot_count_or_batch_cut():
    """Full-size property (no oracle): the same 16384 cases through 4096 slots in one call, and through 256
    slots in three calls, give identical bytes and statuses — results are a pure function of
    (seed, case number, input, options), never of scheduling."""
    if util.priming():
        pytest.skip("no oracle involved")
    import erlamsa_amd as ea
    from erlamsa_amd import synth
    import hashlib
    n = 16384
    mat = synth.mixed(n, 4096)
    data, off = synth.as_arena(mat)

    def run(max_slots, cuts):
        eng = ea.Engine(0)
        eng.configure(mutations=PROD, patterns="od,nd,bu", max_slots=max_slots)
        eng.upload_corpus(data, off)
        dig, sts = [], []
        for a, b in zip(cuts[:-1], cuts[1:]):
            eng.fuzz_batch(seed=(7, 8, 9), first_case=a + 1, corpus_first=a, n=b - a)
            outs, st = eng.download()
            dig += [hashlib.sha1(o).digest() for o in outs]
            sts += [int(x) for x in st]
        eng.close()
        return dig, sts

    d1, s1 = run(0, [0, n])
    d2, s2 = run(256, [0, 5000, 5001, n])
    assert s1 == s2
    assert d1 == d2


def test_per_call_seeds_mode():
    """eh_fuzz_calls: every case is its own fuzzer/1 run with its own seed — what erlamsa_esi:call_fuzzer/3
    does per HTTP request (erlamsa_app:fuzz(Bin, #{seed => S}))."""
    import pyoracle as po
    n = 600
    inputs = util.corpus_mixed(n, 512, seed=21)
    data, off = po.pack(inputs)
    rng = np.random.Generator(np.random.PCG64(9))
    seeds = rng.integers(0, 99999, size=(n, 3)).astype(np.int64) + 1
    seeds[0] = (0, 0, 0); seeds[1] = (30268, 30306, 30322); seeds[2] = (-5, 7, -9)
    muts = "bd,bf,bi,sr,sd,num,ld,lr,tr2,ab,uw,len"
    ora = util.oracle_batch(data, off, seeds=seeds, mutations=muts, patterns="od,nd,bu", max_case_bytes=8 << 20, max_case_work=8 << 20)
    if util.priming():
        pytest.skip("oracle cache primed")
    import erlamsa_amd as ea
    eng = ea.Engine(0)
    eng.configure(mutations=muts, patterns="od,nd,bu", max_case_bytes=8 << 20, max_case_work=8 << 20)
    eng.upload_corpus(data, off)
    eng.fuzz_calls(seeds)
    got, gst = eng.download()
    gdr, _ = eng.diag()
    eng.close()
    cmp = [i for i in range(n) if gst[i] not in (2, 3) and ora.status[i] not in (2, 3)]
    assert len(cmp) >= 0.97 * n
    bad = [i for i in cmp if gst[i] != ora.status[i] or not ora.same(i, got[i]) or (gst[i] == 0 and gdr[i] != ora.draws[i])]
    assert not bad, "cases differ: %s" % bad[:10]


def test_ordered_output_flag():
    """EH_FLAG_ORDERED_OUTPUT: the completion-ordered arena is compacted into case order on the device
    (prefix sum of out_len + gather); download becomes one contiguous copy and eh_result_device hands
    out the compact buffer.  Two batches in a row, because the two arenas swap roles every batch."""
    import pyoracle as po
    inputs = util.corpus_mixed(3000, 700, seed=3) + [b"", b"q"]
    data, off = po.pack(inputs)
    muts = "bd=3,bf,bi=7,sr,ld,num,tr2"
    ora = util.oracle_batch(data, off, seed=(9, 9, 9), mutations=muts, patterns="od,nd,bu", max_case_bytes=8 << 20, max_case_work=8 << 20)
    if util.priming():
        pytest.skip("oracle cache primed")
    import erlamsa_amd as ea
    eng = ea.Engine(0)
    eng.configure(mutations=muts, patterns="od,nd,bu", max_case_bytes=8 << 20, max_case_work=8 << 20, flags=ea.engine.EH_FLAG_ORDERED_OUTPUT)
    eng.upload_corpus(data, off)
    n = len(inputs)
    for _ in range(2):
        eng.fuzz_batch(seed=(9, 9, 9))
        got, gst = eng.download()
        assert all(gst[i] == ora.status[i] and ora.same(i, got[i]) for i in range(n) if gst[i] not in (2, 3) and ora.status[i] not in (2, 3))
        dptr, optr, lptr, sptr, tot = eng.result_device()
        lens = np.array([len(g) for g in got], dtype=np.uint64)
        assert tot == int(lens.sum())
        # the device-side offsets are the prefix sums of the lengths (compact, case order)
        # plain hipMemcpy D2H through the runtime the engine already initialised (no torch: a second
        # HIP runtime user in this process must not be needed to read an engine result)
        import ctypes
        assert ea.load_library() is not None
        hip = ctypes.CDLL("libamdhip64.so")
        offs = np.empty(n, dtype=np.uint64)
        assert hip.hipMemcpy(ctypes.c_void_p(offs.ctypes.data), ctypes.c_void_p(optr), ctypes.c_size_t(n * 8), 2) == 0     # device to host
        assert offs.tolist() == np.concatenate(([0], np.cumsum(lens)[:-1])).astype(np.uint64).tolist()
    eng.close()


def test_download_in_many_chunks_and_into_caller_memory():
    """eh_result_download gathers case-ordered chunks on the device into two alternating bounce buffers and copies chunk k
    out while chunk k+1 is gathered: forced to ~200 small chunks here, plus download_into a caller buffer."""
    import os
    import pyoracle as po
    inputs = util.corpus_mixed(2500, 600, seed=12) + [b"", b"z"]
    data, off = po.pack(inputs)
    muts = "bd=3,bf,bi=7,sr,ld,num,lr"
    ora = util.oracle_batch(data, off, seed=(5, 5, 5), mutations=muts, patterns="od,nd,bu", max_case_bytes=8 << 20)
    if util.priming():
        pytest.skip("oracle cache primed")
    import erlamsa_amd as ea
    n = len(inputs)
    eng = ea.Engine(0)
    eng.configure(mutations=muts, patterns="od,nd,bu", max_case_bytes=8 << 20)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=(5, 5, 5))
    whole, st = eng.download()
    assert all(st[i] == ora.status[i] and ora.same(i, whole[i]) for i in range(n) if st[i] not in (2, 3) and ora.status[i] not in (2, 3))
    eng.configure(mutations=muts, patterns="od,nd,bu", max_case_bytes=8 << 20, download_chunk_bytes=16384)   # results stay valid
    chunked, st2 = eng.download()
    _, total, _ = eng.totals()
    buf = np.full(total + 64, 0xAB, dtype=np.uint8)
    offs, st3 = eng.download_into(buf.ctypes.data, total)
    assert eng.fetch(7) == whole[7] and eng.fetch(n - 1) == whole[n - 1] and list(eng.lens()) == [len(x) for x in whole]
    assert chunked == whole and list(st2) == list(st) and list(st3) == list(st)
    assert int(offs[-1]) == total and bytes(buf[:total]) == b"".join(whole) and (buf[total:] == 0xAB).all()
    eng.close()


def test_request_coalescing_submit_flush_poll():
    """eh_submit / eh_flush / eh_poll: single requests collected into eh_fuzz_calls batches; every ticket gets the bytes
    the request gets alone (erlamsa_app:fuzz(Bin, #{seed => S}) per request, erlamsa_fsupervisor.erl:60-86)."""
    import pyoracle as po
    n = 300
    inputs = util.corpus_mixed(n, 400, seed=33)
    data, off = po.pack(inputs)
    rng = np.random.Generator(np.random.PCG64(5))
    seeds = rng.integers(1, 99999, size=(n, 3)).astype(np.int64)
    muts = "bd,bf,bi,sr,sd,num,ld,lr,ab,uw"
    ora = util.oracle_batch(data, off, seeds=seeds, mutations=muts, patterns="od,nd,bu", max_case_bytes=8 << 20)
    if util.priming():
        pytest.skip("oracle cache primed")
    import erlamsa_amd as ea
    eng = ea.Engine(0)
    eng.configure(mutations=muts, patterns="od,nd,bu", max_case_bytes=8 << 20)
    eng.coalesce_limits(64, 1 << 20)
    tickets = [eng.submit(inputs[i], tuple(int(x) for x in seeds[i])) for i in range(n)]      # 4 full batches + 44 pending
    assert eng.poll(tickets[-1]) is None
    eng.flush()
    order = list(rng.permutation(n))
    for i in order:
        st, out = eng.poll(tickets[i], cap=1 << 12)
        if st in (2, 3) or ora.status[i] in (2, 3):
            continue
        assert st == ora.status[i] and ora.same(i, out), i
    with pytest.raises(ea.EngineError):
        eng.poll(tickets[0])
    eng.close()


def test_full_size_bench_workload_is_independent_of_tiers_and_slots():
    """BASELINE configs[2] at its full size (65 536 x 4 KiB, the full default mutator table, no work budget — the bench
    workload): two very different memory configurations (2 048 slots of 4 MiB — the bench's — vs 1 024 slots of 64 MiB, hence
    different chains of borrowed areas and different attempts repeated after running out of memory) must agree on every case's status, output length and PRNG draw count, and byte for byte on a
    4 096-case sample.  No oracle at this size (the CPU needs ~3 h for it); the oracle pins the same table on smaller sets."""
    if util.priming():
        pytest.skip("no oracle involved")
    import hashlib
    import erlamsa_amd as ea
    from erlamsa_amd import synth
    n = 65536
    mat = synth.mixed(n, 4096)
    data, off = synth.as_arena(mat)

    def run(max_slots, case_mib):
        eng = ea.Engine(0)
        eng.configure(patterns="od,nd,bu", max_slots=max_slots, max_case_bytes=case_mib << 20, big_case_bytes=1 << 30, out_capacity=30 << 30)
        eng.upload_corpus(data, off)
        eng.fuzz_batch(seed=(1, 2, 3))
        lens = np.zeros(n + 1, dtype=np.uint64)
        st = np.zeros(n, dtype=np.int32)
        eng._chk(eng.lib.eh_result_download(eng.h, None, 0, lens.ctypes.data, st.ctypes.data))
        draws, _ = eng.diag()
        eng.fuzz_batch(seed=(1, 2, 3), first_case=20001, corpus_first=20000, n=4096)
        outs, st2 = eng.download()
        eng.close()
        return np.diff(lens), st, draws.copy(), [hashlib.sha1(o).digest() for o in outs], [int(x) for x in st2]

    a = run(0, 4)
    b = run(1024, 64)
    assert (a[1] == b[1]).all(), "statuses differ at %s" % np.nonzero(a[1] != b[1])[0][:10]
    ok = a[1] == 0
    assert (a[0][ok] == b[0][ok]).all() and (a[2][ok] == b[2][ok]).all()
    assert a[4] == b[4] and a[3] == b[3]
    assert ok.mean() > 0.995 and int(a[0].sum()) > 15 << 30          # the workload really is the heavy one


def test_meta_trace_equals_the_oracles():
    """the per-case meta trace in the reference's term format - the TEXT erlamsa's meta logger prints
    (erlamsa_main.erl:58-70, ~p per element): {pattern, _}, the patterns' own entries ({sizer, Elem}, {csum, Elem}, {skipped, F},
    {archiver, _}, {compressed | decompressed, _}), every mutator's own entry ({byte_drop, D}, {seq_repeat, BSize}, {muta_num, 0 | 1},
    {sgml_swap, 1}, {json_innertext, _} ...), {used | failed, Name}, nested scheduler calls included - through eh_result_meta and
    erlamsa_amd/meta.py, line by line against the oracle's full trace of the same run: default mutator and pattern tables on mixed /
    SGML / JSON inputs, documents through js / sgm, the complex patterns, gzip / zlib inputs through cp, zip archives through ar."""
    if util.priming():
        pytest.skip("the trace is not part of the digest cache")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    import emu_meta
    assert emu_meta.run(n=96, size=900, seed=(5, 3, 8)) >= 100
    assert emu_meta.run_sets(n=60) >= 250


def test_two_rank_nccl_bench_smoke():
    """bench.py over RCCL with 2 ranks on one node (arena broadcast, case-range sharding, MAX-over-ranks timing), weak and
    strong: skipped on boxes with fewer than 2 GPUs (the builder's and the driver's test boxes have one; the 8-GPU runs are
    the driver's)."""
    if util.priming():
        pytest.skip("no oracle involved")
    import json
    import os
    import subprocess
    import sys
    # (asked in a child: this process may have loaded the engine's HIP runtime already, and torch must initialise first)
    try:
        q = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=240)
        ngpu = int((q.stdout.strip().splitlines() or ["0"])[-1]) if q.returncode == 0 else 0
    except Exception:                                     # noqa: BLE001 - a slow or broken torch import only means "cannot tell": skip
        ngpu = 0
    if ngpu < 2:
        pytest.skip("needs 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for scaling in ("weak", "strong"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
               os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cases", "4096", "--inflight", "2", "--scaling", scaling,
               "--out-gib", "4", "--pool-gib", "8", "--cpu-sample", "0", "--budget-mib", "0", "--pcie", "0"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert r.returncode == 0 and len(lines) == 1, r.stdout[-1500:] + r.stderr[-1500:]
        d = json.loads(lines[0])
        assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["value"] > 0
        assert d["config"]["world_size_seen_by_torch_distributed"] == 2 and d["config"]["arena_equal_on_all_ranks"] is True
