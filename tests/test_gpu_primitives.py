"""Kernel-level tests of the wave-parallel byte movers against numpy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_wave_copy_fill_equal_random_offsets():
    import erlamsa_amd as ea
    rng = np.random.Generator(np.random.PCG64(3))
    region = 1 << 16
    njobs = 600
    buf = rng.integers(0, 256, size=region * njobs * 2, dtype=np.uint8)
    want = buf.copy()
    jobs = []
    sizes = list(range(0, 70)) + [255, 256, 257, 1000, 1023, 1024, 1025, 1971, 3798, 4095, 4096, 4097, 8191, 20000]
    eq_expect = {}
    for j in range(njobs):
        base = 2 * j * region
        kind = j % 3
        n = int(sizes[int(rng.integers(0, len(sizes)))]) if rng.random() < 0.7 else int(rng.integers(0, 30000))
        so, do = int(rng.integers(0, 64)), int(rng.integers(0, 64))
        src, dst = base + so, base + region + do
        if kind == 0:
            want[dst:dst + n] = want[src:src + n]
            jobs.append([0, dst, src, n, 0])
        elif kind == 1:
            pl = int(rng.choice([1, 2, 3, 7, 16, 17, 100, 255, 256, 300, 4096]))
            total = min(n * int(rng.integers(1, 6)), 65000)
            pat = want[src:src + pl].copy()
            reps = np.resize(pat, total) if total else pat[:0]
            want[dst:dst + total] = reps
            jobs.append([1, dst, src, total, pl])
        else:
            same = rng.random() < 0.5
            if same:
                buf[dst:dst + n] = buf[src:src + n]
                want[dst:dst + n] = want[src:src + n]
                if n > 0 and rng.random() < 0.5:
                    k = int(rng.integers(0, n))
                    buf[dst + k] ^= 1
                    want[dst + k] ^= 1
                    same = False
            else:
                same = bool((buf[dst:dst + n] == buf[src:src + n]).all())
            eq_expect[j] = same
            jobs.append([2, dst, src, n, 0])
    eng = ea.Engine(0)
    got, eq = eng.selftest_movers(buf, np.array(jobs, dtype=np.uint32))
    eng.close()
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, "first mismatch at %d (job %d: %s)" % (bad[0], bad[0] // (2 * region), jobs[bad[0] // (2 * region)])
    for j, e in eq_expect.items():
        assert bool(eq[j]) == e, "wave_equal wrong for job %s" % jobs[j]


def test_mask_window_matches_reference():
    import erlamsa_amd as ea
    rng = np.random.Generator(np.random.PCG64(11))
    buf = rng.integers(0, 256, size=1 << 16, dtype=np.uint8)
    text = np.frombuffer(b"hello \"quoted\" it's (a [b] <c> {d}) back\\slash\n" * 400, dtype=np.uint8)
    buf[20000:20000 + len(text)] = text
    jobs = []
    for (s0, n) in [(0, 4096), (0, 100), (3, 5000), (20000, 9000), (20001, 4095), (7, 64), (9, 63), (11, 1), (13, 0), (20000, 4097), (5, 1024 + 17)]:
        for base in (0, 64, 4032):
            if base <= n:
                jobs.append([3, base, s0, n, 0])
    eng = ea.Engine(0)
    _, bad = eng.selftest_movers(buf, np.array(jobs, dtype=np.uint32))
    eng.close()
    assert bad.tolist() == [0] * len(jobs), list(zip(jobs, bad.tolist()))
