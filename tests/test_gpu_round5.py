"""Round 5 GPU parity tests (through the C ABI, against oracle/ run LIVE on the box's host threads).

* BASELINE configs[3]: the 4 KiB mixed corpus x the full default mutator table x ALL TEN default patterns
  (erlamsa_patterns.erl:395-405 at their default priorities), rows 0..2047 of the corpus bench.py times.
* BASELINE configs[4] at its real seed size: 64 KiB counter-hash seeds, generator jump (erlamsa_gen.erl:124-150),
  mutators ft,fn,fo,num,len, pattern sz (erlamsa_patterns.erl:81-111).
* zlib:gunzip/1 with the semantics of OTP 20.1 - 23 (concatenated members, trailing bytes -> data_error) through pattern cp.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
import util  # noqa: E402

pytestmark = pytest.mark.gpu


def _compare_live(eng, ora, n, min_compared):
    st = eng.status(); draws, lm = eng.diag(); lens = eng.lens()
    bad, ncmp = [], 0
    for i in range(n):
        if st[i] in (2, 3) or ora.status[i] in (2, 3, 6):          # engine-only limits / the oracle's watchdog
            continue
        ncmp += 1
        if int(st[i]) != int(ora.status[i]) or int(lens[i]) != len(ora.outs[i]) or (st[i] == 0 and int(draws[i]) != int(ora.draws[i])) \
                or eng.fetch(i, int(lens[i])) != ora.outs[i]:
            bad.append((i, int(st[i]), int(ora.status[i]), int(lens[i]), len(ora.outs[i]), int(draws[i]), int(ora.draws[i])))
    assert ncmp >= min_compared, "only %d of %d cases could be compared with the live oracle run (statuses %s)" % (ncmp, n, np.bincount(st, minlength=6).tolist())
    assert not bad, "%d of %d cases differ from the oracle (case, status, oracle status, len, oracle len, draws, oracle draws): %s" % (len(bad), ncmp, bad[:8])
    return ncmp


def test_config4_all_ten_patterns_vs_live_oracle():
    """BASELINE configs[3] on one GPU: rows 0..2047 of the bench corpus, default mutators, the reference's own pattern table."""
    import erlamsa_amd as ea
    from erlamsa_amd import synth
    m = 2048
    mat = synth.mixed(65536, 4096)[:m]
    data, off = synth.as_arena(mat)
    eng = ea.Engine(0)
    eng.configure(patterns=None, max_case_bytes=4 << 20, big_case_bytes=1 << 30, out_capacity=8 << 30)
    eng.upload_corpus(data, off)
    eng.fuzz_batch(seed=(1, 2, 3))
    ora = util.oracle_live(data, off, seed=(1, 2, 3), patterns=None, max_case_bytes=1 << 30, max_case_seconds=20.0)
    # every pattern of the table must have been exercised: the meta trace of the oracle names them
    seen = set()
    for ln in ora.trace:
        for tok in ln.split():
            if tok.startswith("pattern:"):
                seen.add(tok.split(":", 1)[1])
    assert {"od", "nd", "bu", "sk", "sz", "cs", "ar", "cp"} <= seen, "patterns drawn: %s" % sorted(seen)
    ncmp = _compare_live(eng, ora, m, m - 24)
    eng.close()
    print("configs[3]: %d cases bit-exact vs the live oracle, patterns seen %s" % (ncmp, sorted(seen)))


def test_config5_jump_fuse_num_len_sz_on_64k_seeds_vs_live_oracle():
    """BASELINE configs[4] at its real seed size: 2048 counter-hash seeds of 64 KiB (128 MiB arena), generator jump with the
    whole arena as Paths, mutators ft,fn,fo,num,len, pattern sz; cases 1..2048 and a second range of case numbers."""
    import erlamsa_amd as ea
    from erlamsa_amd import synth
    n, size = 2048, 65536
    mat = np.concatenate([synth.counter(range(r0, r0 + 512), size) for r0 in range(0, n, 512)])
    data, off = synth.as_arena(mat)
    eng = ea.Engine(0)
    eng.configure(mutations="ft,fn,fo,num,len", patterns="sz", generators="jump", max_case_bytes=4 << 20, big_case_bytes=1 << 30, out_capacity=4 << 30)
    eng.upload_corpus(data, off)
    total = 0
    for first in (1, 1 + 5 * 131072):
        eng.fuzz_batch(seed=(1, 2, 3), first_case=first, corpus_first=0, n=n)
        ora = util.oracle_live(data, off, seed=(1, 2, 3), first_case=first, mutations="ft,fn,fo,num,len", patterns="sz", generators="jump",
                               max_case_bytes=1 << 30, max_case_seconds=20.0)
        total += _compare_live(eng, ora, n, n - 16)
    eng.close()
    print("configs[4] shape at 64 KiB seeds: %d cases bit-exact vs the live oracle" % total)


def test_gunzip_of_otp_20_1_concatenated_members_and_trailing_bytes():
    """zlib:gunzip/1 as OTP 20.1 - 23 define it (inflateInit(Z, 16 + MAX_WBITS, reset)): gz + gz is decoded whole, gz + anything else
    raises data_error - on the device decoder itself (vs libz through Python) and through pattern cp (vs the oracle, which calls libz)."""
    import zlib
    import erlamsa_amd as ea
    import emu_containers
    import emu_zlib
    eng = ea.Engine(0)
    a, b = b"first member " * 40, bytes(range(256)) * 5
    def gz(x, lvl=6):
        c = zlib.compressobj(lvl, zlib.DEFLATED, 31, 8, zlib.Z_DEFAULT_STRATEGY)
        return c.compress(x) + c.flush()
    cases = [(gz(a) + gz(b), a + b), (gz(a) + gz(b, 1) + gz(a, 9), a + b + a), (gz(a) + b"tail", None), (gz(a) + gz(b)[:-1], None), (gz(a) + b"\x1f\x8b", None),
             (gz(a) + gz(b)[:-8] + bytes(8), None), (gz(b""), b""), (gz(b"") + gz(b""), b""), (b"", None)]
    for blob, want in cases:
        got = eng.selftest_zlib(4, blob, cap=1 << 16)
        assert got == want and emu_zlib.want_gunzip(blob) == want, "gunzip of %d bytes: %r vs %r" % (len(blob), None if got is None else len(got), None if want is None else len(want))
    eng.close()
    assert emu_containers.run_cp(n=45) >= 135


def test_work_budget_counts_the_device_codecs():
    """eh_options.max_case_work (the engine's deterministic stand-in for maxrunningtime) counts what pattern cp / ar and mutator zip
    inflate and deflate on one lane: the engine and the oracle's EngineGuard stop the same cases (EH_CASE_BUDGET), the others are
    byte-identical (csrc/eh_device.h codec_work)."""
    import emu_containers
    assert emu_containers.run_budget(n=40) == 120


def test_rccl_called_from_inside_the_library_on_this_gpu():
    """ABI 7 (include/erlamsa_hip.h "multi-GPU"): librccl.so loaded by the library, a communicator of ONE rank on this GPU, and the
    three ways an arena reaches a context - eh_corpus_broadcast, eh_corpus_allgather, eh_corpus_broadcast_local - each giving the
    results of eh_corpus_upload.  (One rank is all a one-GPU box has; the W-rank logic runs in tests/test_comm_abi.py on CPU ranks,
    two real ranks in test_two_ranks_rccl_strong_scaling_equals_one_rank below.)"""
    import hashlib
    import erlamsa_amd as ea
    from erlamsa_amd import synth
    mat = synth.mixed(512, 1500, seed=21)
    data, off = synth.as_arena(mat)

    def digests(e):
        e.fuzz_batch(seed=(2, 7, 1))
        outs, st = e.download()
        return [hashlib.sha1(x).digest() for x in outs], [int(x) for x in st]

    ref = ea.Engine(0); ref.configure(max_case_bytes=4 << 20); ref.upload_corpus(data, off)
    want = digests(ref)
    uid = ea.Engine.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    e = ea.Engine(0); e.configure(max_case_bytes=4 << 20)
    e.comm_init(uid, 0, 1)
    e.corpus_broadcast(0, data, off)
    assert e.n_corpus == 512 and digests(e) == want, "eh_corpus_broadcast: other results than eh_corpus_upload"
    e.corpus_allgather(data, off)
    assert e.n_corpus == 512 and digests(e) == want, "eh_corpus_allgather: other results than eh_corpus_upload"
    e.comm_destroy()
    ea.Engine.comm_init_local([e])
    e.upload_corpus(data[:int(off[100])], off[:101])
    ea.Engine.corpus_broadcast_local([e], 0)
    assert e.n_corpus == 100
    with pytest.raises(ea.EngineError):
        ea.Engine.comm_init_local([e, ref])                     # two contexts of one device
    with pytest.raises(ea.EngineError):
        ref.corpus_broadcast(0, data, off)                      # no communicator
    with pytest.raises(ea.EngineError) as ex:
        ref.configure(sequence_muta=True)
    assert ex.value.code == -6
    e.close(); ref.close()


TWO_RANK = r'''
import hashlib, json, os, sys, time
sys.path.insert(0, %(root)r)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import shard, synth
rank, world, idfile, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
n = 4096
e = ea.Engine(rank)
e.configure(max_case_bytes=4 << 20, out_capacity=4 << 30)
if rank == 0:
    open(idfile + ".tmp", "wb").write(ea.Engine.comm_unique_id()); os.rename(idfile + ".tmp", idfile)
while not os.path.exists(idfile):
    time.sleep(0.01)
e.comm_init(open(idfile, "rb").read(), rank, world)
if rank == 0:
    e.corpus_broadcast(0, *synth.as_arena(synth.mixed(n, 4096, seed=3)))
else:
    e.corpus_broadcast(0)
first, cnt = shard.case_range(n, rank, world)
e.fuzz_batch(seed=(1, 2, 3), first_case=first + 1, corpus_first=first, n=cnt)
outs, st = e.download()
json.dump({"first": first, "sha1": [hashlib.sha1(x).hexdigest() for x in outs], "status": [int(x) for x in st]}, open(out, "w"))
'''


def test_two_ranks_rccl_strong_scaling_equals_one_rank(tmp_path):
    """Two OS processes, one GPU each, the arena RCCL-broadcast over xGMI from inside the library, ONE run of 4096 cases split by
    shard.case_range: byte for byte the results of a 1-rank run.  Skipped on a box with one GPU."""
    import hashlib
    import subprocess
    import erlamsa_amd as ea
    import ctypes
    n_dev = ctypes.c_int(0)
    hip = ctypes.CDLL("libamdhip64.so")
    if hip.hipGetDeviceCount(ctypes.byref(n_dev)) != 0 or n_dev.value < 2:
        pytest.skip("needs two GPUs (this box has %d)" % n_dev.value)
    from erlamsa_amd import synth
    script = tmp_path / "rank.py"
    script.write_text(TWO_RANK % {"root": ROOT})
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), "2", str(tmp_path / "uid"), str(tmp_path / ("r%d.json" % r))], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    import json
    res = [json.load(open(tmp_path / ("r%d.json" % r))) for r in range(2)]
    e = ea.Engine(0); e.configure(max_case_bytes=4 << 20, out_capacity=4 << 30)
    e.upload_corpus(*synth.as_arena(synth.mixed(4096, 4096, seed=3)))
    e.fuzz_batch(seed=(1, 2, 3))
    outs, st = e.download()
    want = [hashlib.sha1(x).hexdigest() for x in outs]
    for r in res:
        assert r["sha1"] == want[r["first"]:r["first"] + len(r["sha1"])] and r["status"] == [int(x) for x in st[r["first"]:r["first"] + len(r["sha1"])]]
    assert sum(len(r["sha1"]) for r in res) == 4096
