"""GPU parity, round 3's additions - run AFTER the established tests (the file sorts behind test_gpu_parity.py and
test_gpu_primitives.py): the file / jump generators, the device codecs against libz, the container patterns and zip archives,
the golden sets added this round, and the regression test of the overlapping move.  All of them were written after the round's GPU
minutes had been spent and are green so far only on the CPU wavefront emulator (tests/test_emulated_kernel.py runs the same
bodies there)."""
import pytest

import util
from test_gpu_parity import _compare

pytestmark = pytest.mark.gpu


def test_file_and_jump_generators_vs_oracle():
    """SURVEY §8(f)-2, erlamsa_gen.erl:59-150: the `file` generator (multi-block streams cut by rand_block_size, finish/1) and
    the `jump` generator (jump_somewhere/2 splices across corpus entries) on the device; their streams are forced by the
    pattern's first uncons, so the draw order differs from `direct`.  Batches that are sub-ranges of the corpus, batch and
    per-call seeding, all patterns: bytes, statuses and draw counts against the oracle."""
    if util.priming():
        pytest.skip("live oracle (small)")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    import emu_gens
    assert emu_gens.run(n=96) >= 700


@pytest.mark.parametrize("seed", [(1, 77, 5), (2, 77, 5), (12, 77, 5)])
def test_nearly_full_slot_pattern_scans_then_flushed_results(seed):
    """A 32 KiB slot that the random generator's stream nearly fills, patterns whose own scans borrow an area and give it back (cs,
    sz) or that mutate the tail of a block (sk), a mutator whose result goes through flush_bvecs (num): the candidate is moved down
    onto a range it overlaps (found by tests/hipemu/emu_fuzz2.py in round 3: 11 - 48 bytes of such cases were corrupted)."""
    _compare(util.corpus_uniform(192, 64, seed=3), "num,bd,lr", "cs,sz,sk,od,nd", seed=seed, generators="random=1", engine_cap=32768, max_skipped=1)


def test_device_zlib_against_libz():
    """csrc/eh_zlib.h on the GPU against zlib itself (Python's zlib module, the image's libz 1.2.11 - the library OTP's zlib module
    binds): the level-6 deflate stream byte for byte with the raw / gzip / zlib wrappers, and the decoders on complete,
    truncated and corrupted inputs with the semantics of zlib:gunzip/1 and zlib:inflate/2 (erlamsa_patterns.erl:216-246)."""
    if util.priming():
        pytest.skip("no oracle involved")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    import emu_zlib
    nc, nd = emu_zlib.run(quick=True, small=True)
    assert nc >= 100 and nd >= 1500


def test_container_patterns_vs_oracle():
    """Pattern cp (erlamsa_patterns.erl:216-260) on real gzip / zlib inputs: zlib:gunzip / zlib:inflate on the device, the rest of the
    pattern chain on the payload with the Mutator put back afterwards, zlib:gzip / zlib:deflate(default) byte for byte - bytes,
    statuses, draw counts and the meta trace against the oracle, whose zlib calls are libz's."""
    if util.priming():
        pytest.skip("live oracle (small)")
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    import emu_containers
    assert emu_containers.run_cp(n=60) >= 180


def test_zip_archives_vs_oracle():
    """Pattern ar (erlamsa_patterns.erl:165-214) and mutator zip (erlamsa_mutations.erl:1149-1163) on real zip archives (several
    files, stored and deflated, small and empty files, the extensions zip:create stores, an archive comment, cut and corrupted
    archives, the empty archive): zip:foldl's walk and the files' inflate on the device, every file's evaluation from the Mutator
    the pattern was given, zip:create's layout with raw deflate byte for byte - bytes, statuses, draw counts and the meta trace
    against the oracle (prim_zip / zip restated there, its zlib calls are libz's)."""
    if util.priming():
        pytest.skip("live oracle (small)")
    import os
    import sys
    import warnings
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hipemu"))
    import emu_containers
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert emu_containers.run_zip(n=50) >= 200


def test_engine_reproduces_round3_golden_sets():
    """tests/golden/vectors.json, the sets added in round 3 (gzip / zlib inputs through cp, zip archives through ar and the zip
    mutator, the file and jump generators): the engine reproduces the committed outputs without the oracle being present."""
    if util.priming():
        pytest.skip("golden file, no live oracle")
    import test_golden
    sets = [v for v in test_golden._vectors() if v["name"] in test_golden.LATE_SETS]
    assert len(sets) == len(test_golden.LATE_SETS)
    for v in sets:
        test_golden.engine_reproduces(v)
