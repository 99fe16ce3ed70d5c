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
    gz = lambda x, lvl=6: emu_zlib.want_compress(1, x) if lvl == 6 else zlib.compress(x, lvl, 31)
    cases = [(gz(a) + gz(b), a + b), (gz(a) + gz(b, 1) + gz(a, 9), a + b + a), (gz(a) + b"tail", None), (gz(a) + gz(b)[:-1], None), (gz(a) + b"\x1f\x8b", None),
             (gz(a) + gz(b)[:-8] + bytes(8), None), (gz(b""), b""), (gz(b"") + gz(b""), b""), (b"", None)]
    for blob, want in cases:
        got = eng.selftest_zlib(4, blob, cap=1 << 16)
        assert got == want and emu_zlib.want_gunzip(blob) == want, "gunzip of %d bytes: %r vs %r" % (len(blob), None if got is None else len(got), None if want is None else len(want))
    eng.close()
    assert emu_containers.run_cp(n=45) >= 135
