"""The engine's kernel code, UNMODIFIED, on a CPU wavefront emulator (tests/hipemu) against the oracle.

This does not replace `-m gpu` (the parity tests proper run the gfx950 binary on an MI355X); it checks the
kernel's logic — wave-uniform control flow, cross-lane exchanges, LDS uniformity, work-area bookkeeping —
where there is no GPU, and it aborts when a cross-lane operation is reached by only part of a wavefront
or when lanes store different values to LDS."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))

PROD = "uw,ui,ab,ad,tr2,td,num,ts1,tr,ts2,bd,bei,bed,bf,bi,ber,br,sp,sr,sd,snand,srnd,ld,lds,lr2,lri,lr,ls,lp,lis,lrs,len,uri,zip,nil"


@pytest.fixture(scope="module")
def emu_lib():
    import build_emu
    return build_emu.build()


def _run(lib, *args, per_call=False):
    env = dict(os.environ, ERLAMSA_HIP_LIB=lib, EMU_PER_CALL="1" if per_call else "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_parity.py")] + [str(a) for a in args],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


def test_emulated_byte_mutators(emu_lib):
    _run(emu_lib, "bd,bf,bi,bei,bed,ber,br,sp,sr,sd,snand,srnd", "od,nd,bu", 64, 256, "uniform")


def test_emulated_production_set_on_mixed_corpus(emu_lib):
    _run(emu_lib, PROD, "od,nd,bu", 48, 1024, "mixed")


def test_emulated_all_patterns_and_fuse(emu_lib):
    _run(emu_lib, PROD + ",ft,fn,fo", "od,nd,bu,sk,sz,cs,ar,cp,co,nu", 32, 512, "mixed", "5,6,7")


def test_emulated_per_call_seeds(emu_lib):
    _run(emu_lib, "bd,bf,bi,sr,sd,num,ld,lr,tr2,ab,uw,len", "od,nd,bu", 48, 512, "mixed", per_call=True)


def test_emulated_abi_behaviour(emu_lib):
    """call order, error codes, option parsing, buffer growth, sub-range reproducibility, empty inputs"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_abi.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "abi behaviour ok" in r.stdout, r.stdout + r.stderr


def test_emulated_tiered_work_areas(emu_lib):
    """a case that outgrows its slot goes on in areas borrowed from the pool's tiers; without larger tiers it overflows"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_tiers.py"), "16"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "tiers ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_emulated_chunked_work_areas_default_tables(emu_lib):
    """16 KiB slots, the full default tables: attempts that run out of memory (also inside nested scheduler calls) are
    repeated after the case has borrowed a larger area; bytes, statuses and draw counts are the oracle's"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_chunks.py"), "16"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "chunks ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_emulated_fuse2_table_modes(emu_lib):
    """eh_fuse2.h on periodic, textual, random and short-alphabet blocks of 6 KB: two-level next-byte tables, compact lookups,
    bitmap rows, the special node — bytes, statuses and draw counts are the oracle's"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_fuse2.py"), "16", "6000"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "fuse2 ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_emulated_meta_trace(emu_lib):
    """eh_result_meta = the oracle's meta trace ({pattern,P} / {used,Name} / {failed,Name}, nested calls included), default tables"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_meta.py"), "18"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "meta ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_emulated_coalescing_from_threads(emu_lib):
    """submitters, pollers and a flusher on one context at once: every ticket gets its request's own result"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_coalesce_threads.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "threads ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_emulated_file_and_jump_generators(emu_lib):
    """erlamsa_gen.erl:59-150 on the device: multi-block `file` streams, `jump` splices across corpus entries, the stream forced by
    the pattern's first uncons (its draws come after the pattern's), sub-range batches with the whole corpus as Paths"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_gens.py"), "8"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "gens ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_emulated_device_zlib_against_libz(emu_lib):
    """csrc/eh_zlib.h against zlib itself (Python's zlib module = libz 1.2.11): level-6 streams byte for byte (raw, gzip, zlib wrappers),
    the decoders on complete / truncated / corrupted inputs with the semantics of zlib:gunzip/1 and zlib:inflate/2"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_zlib.py"), "quick"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "zlib ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_emulated_container_patterns(emu_lib):
    """pattern cp on real gzip / zlib inputs (complete, with header fields, truncated, corrupted, nested, behind a length field):
    decoded, mutated through the rest of the pattern chain and compressed again; pattern ar and mutator zip on real zip archives
    (stored / deflated / small / empty files, stored extensions, archive comment, cut and corrupted archives) - bytes, statuses,
    draw counts and the meta trace are the oracle's, whose zlib calls are libz's"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_containers.py"), "30"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "containers ok" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_emulated_fuse_routes_agree(emu_lib):
    """Round 4's fuse routes on the emulator: LDS-resident vs node lists vs oracle on the corner corpora (csrc/eh_fuse_lds.h), lists
    with their periodic stretches cut short vs the lists as they are vs oracle (csrc/eh_fuse_red.h).  The GPU runs the same scripts
    with more cases (tests/test_gpu_round4.py)."""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_fuse_lds.py"), "1", "3"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_fuse_red.py"), "1", "3"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout + r.stderr


def test_emulated_sgml_lane_batches_and_base64_wave_decode(emu_lib):
    """Round 4's last two changes on the emulator.  csrc/eh_sgml.h: tag attempts one per lane from any text state (runs of failed
    attempts, the machine's memos read and fed by the lanes, names over thousands of events through the next-stop table, a batch per
    accepted tag inside periodic stretches) against the wave-wide machine alone (EH_FLAG_SGML_NO_REPLAY | EH_FLAG_SGML_NO_LANES) and the
    oracle, on the 8 kinds of documents without a period (round 5: tags of thousands of attributes in every attribute syntax) and the one with failing runs inside its periods of tests/hipemu/emu_sgml_replay.py (tag soup, runs of failing attempts, tags of
    thousands of attributes, names over thousands of '<' ...) at a tenth of their length.  csrc/eh_lex.h
    b64_decode_wave: every padding, white space inside groups / the padding, long blobs, refused chunks (tests/hipemu/emu_b64.py).
    The GPU runs both scripts at full length (tests/test_gpu_round4.py)."""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_sgml_replay.py"), "1", "5", "1", "small"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "cases 9 bad 0" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_b64.py"), "3", "2", "1", "small"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout + r.stderr


def test_emulated_blocks_above_a_million_bytes_are_split(emu_lib):
    """split/1 + split_into_maxblocks/2 (erlamsa_patterns.erl:44-59) on inputs of 1.0 and 1.6 MB, patterns od / nd / sk"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_split.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "split ok" in r.stdout, r.stdout + r.stderr


def test_emulated_tree_matcher_deep_nesting_quotes_and_stray_closers(emu_lib):
    """partial_parse/1 + grow/3 (erlamsa_mutations.erl:800-905) on inputs made for the matcher: nesting beyond the 64 stack entries kept
    in lane registers (spill / refill), quotes, closers that match nothing, openers that never close; tr2, td, ts1, ts2, tr"""
    env = dict(os.environ, ERLAMSA_HIP_LIB=emu_lib)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_tree.py"), "1", "10"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "emu_tree ok" in r.stdout, r.stdout + r.stderr


def test_emulated_race_detector_finds_no_cross_lane_access_without_a_rendezvous():
    """The engine built with every load / store of the kernel code instrumented (build_emu.py --race, tests/hipemu/race_hooks.cpp):
    between two rendezvous points no lane reads what another lane wrote or overwrites what another lane read - the class of bug
    (a missing wave_sync() after lane 0 filled something in) that the plain emulator cannot see and a real wavefront does not
    forgive.  Generators, containers (lane-0 codecs), nearly full and tiny slots with the default tables, streaming fuse + trace.
    (Checked by hand that it fires: without the wave_sync() at the end of z_compress the same run reports 4 536 accesses while
    every parity test still passes.)"""
    import build_emu
    lib = build_emu.build(race=True)
    env = dict(os.environ, ERLAMSA_HIP_LIB=lib, HIPEMU_RACE_QUIET="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hipemu", "emu_race.py"), "3"], env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0 and "race ok" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
