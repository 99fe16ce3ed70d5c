"""GPU parity: HIP engine (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

BYTE_MUTAS = "bd,bf,bi"
BYTE_ALL = "bd,bei,bed,bf,bi,ber,br"
SEQ = "sp,sr,sd,snand,srnd"


def _compare(inputs, mutations, patterns, seed=(1, 2, 3), generators=None, first_case=1, max_report=5):
    import pyoracle as po
    import erlamsa_amd as ea
    data, off = po.pack(inputs)
    want, wst, wdr, trace = po.fuzz_batch(data, off, seed=seed, mutations=mutations, patterns=patterns, generators=generators,
                                          first_case=first_case, max_case_bytes=8 << 20, trace=True)
    eng = ea.Engine(0)
    eng.configure(mutations=mutations, patterns=patterns, generators=generators)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=seed, first_case=first_case)
    got, gst = eng.download()
    gdr, glm = eng.diag()
    tr = trace.split("\n")
    bad = []
    for i in range(len(inputs)):
        if got[i] != want[i] or gst[i] != wst[i]:
            bad.append((i, util.first_diff(got[i], want[i]), len(got[i]), len(want[i]), int(gst[i]), int(wst[i]), int(gdr[i]), int(wdr[i]), tr[i]))
    msg = "\n".join("case %d: first diff at %d, len gpu %d vs oracle %d, status %d vs %d, draws %d vs %d, trace: %s" % b for b in bad[:max_report])
    assert not bad, "%d/%d cases differ\n%s" % (len(bad), len(inputs), msg)
    ok = wst == 0
    assert (gdr[ok] == wdr[ok]).all(), "draw counts differ"
    eng.close()


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
    _compare(util.corpus_uniform(256, 64), BYTE_ALL + ",sd,sr", "od,nd,bu", generators="random=1")


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
