"""An independent Python model of a subset of erlamsa_main:fuzzer/1, written from the reference's .erl sources (cited per
function) WITHOUT consulting oracle/oracle.cpp: paths = [direct], generators direct + random, patterns od / nd / bu / sk / sz / cs / co / nu (a complex pattern whose continuation is an archiver or compressed pattern is
reported as unmodelled: fuzzer/5 returns None for that case), and
the mutators uw ui num bd bei bed bf bi ber br sp sr sd snand srnd ld lds lr2 lri lr ls lp lis lrs ft fn fo tr2 td ts1 ts2 tr ab ad uri len nil.  tests/test_pymodel.py diffs it against the C++ oracle.

Everything is a literal, clause-by-clause transcription — Erlang lists are Python lists, binaries are bytes, lazy
stream tails are forced in the order erlamsa_out:blocks_port forces them.  OTP pieces (random, lists:sort/2) are
restated from the OTP sources as the author remembers them (there is no OTP in this image): they are the part of
"parity" that only a real BEAM run can pin (tests/golden/capture.escript).
"""
import math
import sys

sys.setrecursionlimit(max(sys.getrecursionlimit(), 6000))
if hasattr(sys, "set_int_max_str_digits"):
    sys.set_int_max_str_digits(0)            # numbers grow to 10^5 digits under sr

class Unmodelled(Exception):
    """the case took a path this model does not cover (a sizer / csum / archiver / compressed continuation, a tree result
    beyond the model's size limit)"""


class ErlCrash(Exception):
    """the worker process dies: fuzzer/1 times out on it and records <<>>"""


# ------------------------------------------------------------------------------------------------ OTP random (AS183)
P1, P2, P3 = 30269, 30307, 30323


class Random:
    """OTP stdlib random.erl: seed/3, uniform/0, uniform/1 (process dictionary state)."""

    def __init__(self):
        self.s = (3172, 9814, 20125)           # random:seed0()
        self.draws = 0

    def seed(self, t):                          # random:seed({A1,A2,A3}) -> seed/3
        a1, a2, a3 = t
        self.s = (abs(a1) % (P1 - 1) + 1, abs(a2) % (P2 - 1) + 1, abs(a3) % (P3 - 1) + 1)

    def uniform(self):
        a1, a2, a3 = self.s
        b1, b2, b3 = (a1 * 171) % P1, (a2 * 172) % P2, (a3 * 170) % P3
        self.s = (b1, b2, b3)
        self.draws += 1
        r = b1 / P1 + b2 / P2 + b3 / P3
        return r - math.trunc(r)

    def uniform_n(self, n):                     # uniform(N) -> trunc(uniform() * N) + 1
        u = self.uniform()
        try:
            return math.trunc(u * n) + 1
        except OverflowError:                   # a bignum beyond the float range: badarith, the worker dies
            raise ErlCrash("badarith: float * bignum")


# ------------------------------------------------------------------------------------------------ OTP lists:sort/2
def lists_sort(fun, lst):
    """stdlib lists.erl sort/2 with its fsplit_* / fmergel / rfmergel / fmerge2_* / rfmerge2_* helpers, one Python
    function per Erlang function, one branch per clause.  Lists are Python lists with the head at index 0."""
    if len(lst) < 2:
        return list(lst)
    x, y, t = lst[0], lst[1], lst[2:]
    if fun(x, y):
        return _fsplit_1(y, x, fun, t, [], [])
    return _fsplit_2(y, x, fun, t, [], [])


def _fsplit_1(y, x, fun, l, r, rs):
    s, in_x1 = None, False                      # in_x1: we are in fsplit_1_1 with S
    while True:
        if not l:
            if in_x1:
                return _rfmergel([[s], [y, x] + r] + rs, [], fun, "asc")
            return _rfmergel([[y, x] + r] + rs, [], fun, "asc")
        z, l = l[0], l[1:]
        if fun(y, z):
            y, x, r = z, y, [x] + r
        elif fun(x, z):
            x, r = z, [x] + r                   # fsplit_1(Y, Z, Fun, L, [X | R], Rs)
        elif not in_x1 and r == []:
            r = [z]                             # fsplit_1(Y, X, Fun, L, [Z], Rs)
        elif not in_x1:
            s, in_x1 = z, True                  # fsplit_1_1(Y, X, Fun, L, R, Rs, Z)
        else:
            rs = [[y, x] + r] + rs
            if fun(s, z):
                y, x = z, s                     # fsplit_1(Z, S, Fun, L, [], [[Y, X | R] | Rs])
            else:
                y, x = s, z                     # fsplit_1(S, Z, ...)
            r, s, in_x1 = [], None, False


def _fsplit_2(y, x, fun, l, r, rs):
    s, in_x1 = None, False
    while True:
        if not l:
            if in_x1:
                return _fmergel([[s], [y, x] + r] + rs, [], fun, "desc")
            return _fmergel([[y, x] + r] + rs, [], fun, "desc")
        z, l = l[0], l[1:]
        if not fun(y, z):
            y, x, r = z, y, [x] + r
        elif not fun(x, z):
            x, r = z, [x] + r
        elif not in_x1 and r == []:
            r = [z]
        elif not in_x1:
            s, in_x1 = z, True
        else:
            rs = [[y, x] + r] + rs
            if not fun(s, z):
                y, x = z, s
            else:
                y, x = s, z
            r, s, in_x1 = [], None, False


def _fmergel(ls, acc, fun, o):
    while True:
        if len(ls) >= 2 and o == "asc":          # fmergel([T1, [H2 | T2] | L], Acc, Fun, asc)
            t1, l2, ls = ls[0], ls[1], ls[2:]
            acc = [_fmerge2_1(t1, l2[0], fun, l2[1:], [])] + acc
        elif len(ls) >= 2:                       # fmergel([[H2 | T2], T1 | L], Acc, Fun, desc)
            l2, t1, ls = ls[0], ls[1], ls[2:]
            acc = [_fmerge2_1(t1, l2[0], fun, l2[1:], [])] + acc
        elif len(ls) == 1 and acc == []:
            return ls[0]
        elif len(ls) == 1:
            return _rfmergel([ls[0][::-1]] + acc, [], fun, o)
        else:
            return _rfmergel(acc, [], fun, o)


def _rfmergel(ls, acc, fun, o):
    while True:
        if len(ls) >= 2 and o == "asc":          # rfmergel([[H2 | T2], T1 | L], Acc, Fun, asc)
            l2, t1, ls = ls[0], ls[1], ls[2:]
            acc = [_rfmerge2_1(t1, l2[0], fun, l2[1:], [])] + acc
        elif len(ls) >= 2:                       # rfmergel([T1, [H2 | T2] | L], Acc, Fun, desc)
            t1, l2, ls = ls[0], ls[1], ls[2:]
            acc = [_rfmerge2_1(t1, l2[0], fun, l2[1:], [])] + acc
        elif len(ls) == 1:
            return _fmergel([ls[0][::-1]] + acc, [], fun, o)
        else:
            return _fmergel(acc, [], fun, o)


def _fmerge2_1(t1, h2, fun, t2, m):
    # fmerge2_1 / fmerge2_2 as one loop; m is the Erlang accumulator M (head at index 0)
    state, h1 = 1, None
    while True:
        if state == 1:
            if not t1:
                return t2[::-1] + [h2] + m       # lists:reverse(T2, [H2 | M])
            h1, t1 = t1[0], t1[1:]
            if fun(h1, h2):
                m = [h1] + m
            else:
                m, state = [h2] + m, 2
        else:
            if not t2:
                return t1[::-1] + [h1] + m       # lists:reverse(T1, [H1 | M])
            h2, t2 = t2[0], t2[1:]
            if fun(h1, h2):
                m, state = [h1] + m, 1
            else:
                m = [h2] + m


def _rfmerge2_1(t1, h2, fun, t2, m):
    state, h1 = 1, None
    while True:
        if state == 1:
            if not t1:
                return t2[::-1] + [h2] + m
            h1, t1 = t1[0], t1[1:]
            if fun(h1, h2):
                m, state = [h2] + m, 2           # rfmerge2_2(H1, T1, Fun, T2, [H2 | M])
            else:
                m = [h1] + m
        else:
            if not t2:
                return t1[::-1] + [h1] + m
            h2, t2 = t2[0], t2[1:]
            if fun(h1, h2):
                m = [h2] + m
            else:
                m, state = [h1] + m, 1           # rfmerge2_1(T1, H2, Fun, T2, [H1 | M])


# ------------------------------------------------------------------------------------------------ erlamsa_rnd.erl
class Rnd:
    def __init__(self):
        self.r = Random()

    def seed(self, t):
        self.r.seed(t)                                                     # :65

    def rand(self, n):
        return 0 if n == 0 else self.r.uniform_n(n) - 1                    # :69-70

    def erand(self, n):
        return 0 if n == 0 else self.r.uniform_n(n)                        # :73-74

    def rand_range(self, l, r):                                            # :78-83
        if r > l:
            return self.rand(r - l) + l
        return l if r == l else 0

    def rand_bit(self):
        return int(math.floor(self.r.uniform() + 0.5))                     # round(random:uniform()) :95

    def rand_occurs_fixed(self, nom, denom):                               # :111-118
        n = self.rand(denom)
        return n != 0 if nom == 1 else n < nom

    def rand_nbit(self, n):                                                # :122-125
        if n == 0:
            return 0
        hi = 1 << (n - 1)
        return hi | self.rand(hi)

    def rand_log(self, n):
        return 0 if n == 0 else self.rand_nbit(self.rand(n))               # :128-130

    def rand_elem(self, l):
        return [] if not l else l[self.r.uniform_n(len(l)) - 1]            # :133-136

    def random_numbers(self, bound, cnt):                                  # :163-169: built by prepending
        acc = [self.rand(bound) for _ in range(cnt)]
        return acc[::-1]

    def random_block(self, n):
        return bytes(self.random_numbers(256, n))                          # :154-161 (same shape)

    def random_permutation(self, l):                                       # :172-178
        if len(l) == 2:
            return [l[1], l[0]] if self.rand(2) == 1 else l
        return [y for _, y in sorted(((self.r.uniform(), x) for x in l))]   # lists:sort/1 of {float, X}

    def rand_delta(self):
        return 1 if self.rand_bit() == 0 else -1                           # :199-206

    def gen_predictable_seed(self):
        return (self.erand(99999), self.erand(99999), self.erand(99999))   # :57


AVG_BLOCK_SIZE, MIN_BLOCK_SIZE = 2048, 256
MAX_BLOCK_SIZE = 2 * AVG_BLOCK_SIZE
ABSMAX_BINARY_BLOCK = 1000000
MIN_SCORE, MAX_SCORE = 2.0, 10.0


# ------------------------------------------------------------------------------------------------ erlamsa_utils.erl
def sort_by_priority(l):                                                   # :113-117
    sl = lists_sort(lambda a, b: a[0] > b[0], l)
    return sl, sum(a for a, _ in sl)


def choose_pri(l, n):                                                      # :154-160
    for this, el in l:
        if n == 0 or n < this:
            return el
        n -= this
    raise ErlCrash("choose_pri: function_clause")


def binarish(b):                                                           # :237-247
    for p in range(len(b) + 1):
        rest = b[p:]
        if rest[:3] == b"\xef\xbb\xbf" or rest[:2] == b"\xfe\x0f":
            return False
        if p == 8 or not rest:
            return False
        if rest[0] == 0 or rest[0] & 128:
            return True
    return False


def flush_bvecs(b, tail):                                                  # :168-175
    out = []
    while len(b) >= AVG_BLOCK_SIZE:
        out.append(b[:AVG_BLOCK_SIZE])
        b = b[AVG_BLOCK_SIZE:]
    return out + [b] + tail


# ------------------------------------------------------------------------------------------------ erlamsa_mutations.erl
def interesting_numbers():                                                 # :65-73 (foldl prepends)
    acc = []
    for i in [1, 7, 8, 15, 16, 31, 32, 63, 64, 127, 128]:
        x = 1 << i
        acc = [x - 1, x, x + 1] + acc
    return acc


def mutate_num(rnd, num):                                                  # :91-112
    n = rnd.rand(12)
    if n == 0:
        return num + 1
    if n == 1:
        return num - 1
    if n == 2:
        return 0
    if n == 3:
        return 1
    if 3 < n < 6:
        return rnd.rand_elem(interesting_numbers())
    if n == 7:
        return num + rnd.rand_elem(interesting_numbers())
    if n == 8:
        return num - rnd.rand_elem(interesting_numbers())
    if n == 9:
        return num - rnd.rand(abs(num) * 2) * (1 if num >= 0 else -1)
    if n == 10:
        return -num
    k = rnd.rand_range(1, 129)
    l = rnd.rand_log(k)
    s = rnd.rand(3)
    return num - l if s == 0 else num + l


def get_num(b, pos):                                                       # :114-125 -> (value | None, position after)
    sign = 1
    while pos < len(b) and b[pos] == 45:                                   # '-' while no digit has been read
        sign, pos = -1, pos + 1
    e = pos
    while e < len(b) and 48 <= b[e] <= 57:
        e += 1
    if e == pos:
        return None, pos
    return int(b[pos:e]) * sign, e                                         # D - 48 + N * 10 per digit


def sed_num(rnd, ll):                                                      # :131-170
    h, t = ll[0], ll[1:]
    found, pos = [], 0
    while pos < len(h):                                                    # mutate_a_num/2 on the way down
        val, after = get_num(h, pos)
        if val is not None:
            found.append((pos, after, val))
            pos = after
        else:
            pos += 1
    which = rnd.rand(len(found))                                           # mutate_a_num(<<>>, NFound)
    if found:
        s, e, val = found[len(found) - 1 - which]                          # Which counts back from the last number
        new = mutate_num(rnd, val)
        lst = h[:s] + str(new).encode() + h[e:]
        n = -1
    else:
        lst, n = h, 0
    isbin = binarish(lst)
    flushed = flush_bvecs(lst, t)
    if n == 0:
        return flushed, (-1 if rnd.rand(10) == 0 else 0)
    return flushed, (-1 if isbin else 2)


def sed_byte(rnd, ll, f):                                                  # construct_sed_byte_muta :179-185
    h, t = ll[0], ll[1:]
    p = rnd.rand(len(h))
    d = rnd.rand_delta()
    if h == b"":                                                           # edit_byte_vector(<<>>, ..) :57
        return [h] + t, d
    return [h[:p] + f(h[p]) + h[p + 1:]] + t, d


def sed_bytes(rnd, ll, f):                                                 # construct_sed_bytes_muta :239-258
    bvec, btail = ll[0], ll[1:]
    if bvec == b"":
        return ll, -1
    bsize = len(bvec)
    s = rnd.rand(bsize)
    l = rnd.rand_range(1, bsize - s + 1)
    c = f(bvec[:s], bvec[s:s + l], bvec[s + l:], btail)
    d = rnd.rand_delta()
    return c, d


def lines(b):                                                              # :326-331
    out, buff = [], []
    for ch in b:
        buff.append(ch)
        if ch == 10:
            out.append(buff)
            buff = []
    if buff:
        out.append(buff)
    return out


def line_muta(rnd, ll, op):                                                # construct_line_muta :351-362
    h, t = ll[0], ll[1:]
    ls = lines(h)
    if ls == [] or binarish(h):                                            # try_lines :341-348
        return ll, -1
    mls = op(ls, len(ls))
    return [b"".join(bytes(x) for x in mls)] + t, 1


def list_del(rnd, l, length):                                              # erlamsa_generic.erl:54-57
    p = rnd.erand(length)
    return l[:p - 1] + l[p:]


def applynth(i, l, fun):                                                   # erlamsa_utils.erl:188-190 (1-based)
    if i < 1 or i > len(l):
        raise ErlCrash("applynth: function_clause")
    return l[:i - 1] + fun(l[i - 1], l[i:])


# erlamsa_generic.erl:54-116
def list_del_seq(rnd, l, length):
    start = rnd.erand(length)
    n = rnd.erand(length - start + 1)
    return applynth(start, l, lambda _e, r: r[n - 1:n - 1 + length])          # lists:sublist(R, N, Len)


def list_dup(rnd, l, length):
    return applynth(rnd.erand(length), l, lambda e, r: [e, e] + r)


def list_repeat(rnd, l, length):
    p = rnd.erand(length)
    n = max(2, rnd.rand_log(10))
    return applynth(p, l, lambda e, r: [e] * n + r)


def list_clone(rnd, l, length):
    frm = rnd.erand(length)
    to = rnd.erand(length)
    elem = l[frm - 1]
    return applynth(to, l, lambda _e, r: [elem] + r)


def list_swap(rnd, l, length):
    if length < 2:
        return l
    return applynth(rnd.erand(length - 1), l, lambda e, r: [r[0], e] + r[1:])


def list_perm(rnd, l, length):
    if length < 3:
        return l
    frm = rnd.erand(length - 1)
    a = rnd.rand_range(2, length - frm)
    b = rnd.rand_log(10)
    n = max(2, min(a, b))

    def f(e, r):
        if n - 1 > len(r):
            raise ErlCrash("lists:split badarg")
        return rnd.random_permutation([e] + r[:n - 1]) + r[n - 1:]
    return applynth(frm, l, f)


def flat(x):                                                               # iolist of a stored line -> bytes
    out = bytearray()
    stack = [x]
    while stack:
        y = stack.pop()
        if isinstance(y, list):
            stack.extend(reversed(y))
        else:
            out.append(y)
    return bytes(out)


def step_state(rnd, st, l, length):                                        # erlamsa_generic.erl:121-139
    while st[0] < 10:
        p = rnd.erand(length)
        st = [st[0] + 1, l[p - 1]] + st[1:]
    up = rnd.erand(20)
    if up < 10:
        new = l[rnd.erand(length) - 1]

        def f(e, r):
            if not isinstance(e, list) or not e:
                raise ErlCrash("step_state: function_clause")
            return [[new] + e[1:]] + r                                     # [[New | T] | R]: an iolist
        return applynth(up + 1, st, f)
    return st


def st_list_mod(rnd, st, l, fun):                                          # :146-153
    n = len(l)
    stp = step_state(rnd, st, l, n)
    x = stp[1:][rnd.erand(stp[0]) - 1]                                     # pick_state/1
    p = rnd.erand(n)
    return stp, applynth(p, l, fun(x))


def st_line_muta(rnd, ll, st, fun):                                        # construct_st_line_muta :366-378
    h, t = ll[0], ll[1:]
    ls = lines(h)
    if ls == [] or binarish(h):
        return ll, -1, st
    stp, newls = st_list_mod(rnd, st, ls, fun)
    return [b"".join(flat(x) for x in newls)] + t, 1, stp


def funny_unicode():                                                       # :1051-1078
    manual = [[239, 191, 191], [240, 144, 128, 128], [0xef, 0xbb, 0xbf], [0xfe, 0xff], [0xff, 0xfe], [0, 0, 0xff, 0xff],
              [0xff, 0xff, 0, 0], [43, 47, 118, 56], [43, 47, 118, 57], [43, 47, 118, 43], [43, 47, 118, 47], [247, 100, 76],
              [221, 115, 102, 115], [14, 254, 255], [251, 238, 40], [251, 238, 40, 255], [132, 49, 149, 51]]
    codes = [[0x0009, 0x000d], 0x008D, 0x00a0, 0x1680, 0x180e, [0x2000, 0x200a], 0x2028, 0x2029, 0x202f, 0x205f, 0x3000,
             [0x200e, 0x200f], [0x202a, 0x202e], [0x200c, 0x200d], 0x0345, 0x00b7, [0x02d0, 0x02d1], 0xff70, [0x02b0, 0x02b8],
             0xfdd0, 0x034f, [0x115f, 0x1160], [0x2065, 0x2069], 0x3164, 0xffa0, 0xe0001, [0xe0020, 0xe007f], [0x0e40, 0x0e44],
             0x1f4a9]
    numbers = []
    for c in codes:                                                        # foldl: seq(X,Y) ++ Acc | [X | Acc]
        numbers = (list(range(c[0], c[1] + 1)) if isinstance(c, list) else [c]) + numbers

    def ext(n):
        return (n & 0x3f) | 0x80

    def enc(pt):                                                           # encode_point/1 :1034-1048
        if pt < 0x80:
            return [pt]
        if pt < 0x800:
            return [0xc0 | (0x1f & (pt >> 6)), ext(pt)]
        if pt < 0x10000:
            return [0xe0 | (0x0f & (pt >> 12)), ext(pt >> 6), ext(pt)]
        return [0xf0 | (7 & (pt >> 18)), ext(pt >> 12), ext(pt >> 6), ext(pt)]
    return manual + [enc(x) for x in numbers]


FUNNY = funny_unicode()


def randmask(rnd, maskfun, bs):                                            # :281-293
    prob = rnd.erand(100)
    flag = rnd.rand_occurs_fixed(prob, 100)
    out = []
    for b in bs:
        nxt = rnd.rand_occurs_fixed(prob, 100)                             # argument order: the next flag is drawn first
        out.append(maskfun(b) if flag else b)
        flag = nxt
    return out


# ------------------------------------------------------------------------------------------------ erlamsa_field_predict.erl
SIZER_MAX_FIRST_BYTES, PREAMBLE_MAX_BYTES = 512, 32


def basic_u8len(a, b, x):                                                  # :51-59
    if a < b and b > 0 and a < len(x) and len(x) >= a + 1:
        ln = x[a]
        if ln == b - a - 1 and ln > 2:
            return [("ok", 8, "big", ln, a, b)]
    return []


def simple_u8len(a, x):                                                    # :62-65
    return [e for k in range(0, 9) for e in basic_u8len(a, len(x) - k, x)]


def basic_len(a, b, x):                                                    # :68-80: first matching clause
    if a < b and b > 0 and a < len(x):
        for size, endian in ((16, "big"), (32, "big"), (64, "big"), (16, "little"), (32, "little"), (64, "little")):
            nb = size // 8
            if len(x) >= a + nb:
                ln = int.from_bytes(x[a:a + nb], endian)
                if ln == b - a - nb and ln > 2:
                    return [("ok", size, endian, ln, a, b)]
    return []


def simple_len(a, b, x):                                                   # :83-90
    return [e for bb in (b, b - 1, b - 2, b - 4, b - 8) for e in basic_len(a, bb, x)]


PY_FIELD_LIMIT = 700               # blocks beyond it are reported unmodelled: 1.3 M candidate checks / 961 CRCs per megabyte in Python


def get_possible_simple_lens(rnd, x):                                      # :93-109
    if len(x) > PY_FIELD_LIMIT:
        raise Unmodelled("block too large for the Python field_predict model")
    if len(x) > 10:
        ln = len(x)
        sub = min(math.trunc(ln / 5), SIZER_MAX_FIRST_BYTES)
        first = list(range(0, sub + 1))
        varb = [rnd.rand_range(sub, ln) for _ in first]
        ranges = [(a, b) for a in first for b in varb]
        allr = [(a, ln) for a in first] + ranges
        big = []
        for a, b in allr:                                                  # foldl prepends: the last range comes first
            big = [simple_len(a, b, x)] + big
        small = [simple_u8len(a, x) for a in first]
        return [e for grp in small for e in grp] + [e for grp in big for e in grp]
    return [e for a in range(0, 4) for e in simple_len(a, len(x), x) + simple_u8len(a, x)]


def field(v, size, endian):
    return (v % (1 << size)).to_bytes(size // 8, endian)


def length_predict(rnd, ll):                                               # :1140-1143 + mutate_length :1113-1137
    h, t = ll[0], ll[1:]
    elem = rnd.rand_elem(get_possible_simple_lens(rnd, h))
    if elem == []:
        return [h] + t, -2
    _ok, size, endian, ln, a, _b = elem
    nb = size // 8
    head, blob, rest = h[:a], h[a + nb:a + nb + ln], h[a + nb + ln:]
    tmp = int.from_bytes(rnd.random_block(nb), "big")
    newlen = min(ABSMAX_BINARY_BLOCK, tmp * 2)
    k = rnd.rand(7)
    if k == 0:
        res = head + bytes(nb) + blob + rest
    elif k == 1:
        res = head + b"\xff" * nb + blob + rest
    elif k == 2:                                                           # fast_pseudorandom_block/1 erlamsa_rnd.erl:155-160
        if newlen < 500000:
            rb = rnd.random_block(newlen)
        else:
            z = newlen - 500000                                            # <<42:Z8L>> is Z8L BITS wide (sic)
            blk = rnd.random_block(500000)
            if z % 8:
                raise ErlCrash("badarg: a bitstring where a binary is needed")
            rb = (42).to_bytes(z // 8, "big") + blk if z else blk
        res = head + field(ln, size, endian) + blob + rb + rest
    elif k == 3:
        res = head + field(newlen, size, endian) + rest
    else:
        res = head + field(newlen, size, endian) + blob + rest
    return [res] + t, 1


def crc32(b):
    import zlib
    return zlib.crc32(b) & 0xffffffff


def get_possible_csum_locations(x):                                        # :155-161
    if not x:
        return []
    if len(x) > PY_FIELD_LIMIT:
        raise Unmodelled("block too large for the Python field_predict model")
    ln = len(x)
    seq = range(0, min(math.trunc(2 * ln / 3), 30 * PREAMBLE_MAX_BYTES) + 1)
    out = []
    for a in seq:                                                          # has_xor8_checksum/3 :131-138
        body = x[a:ln - 1]
        v = 0
        for c in body:
            v ^= c
        if v == x[ln - 1]:
            out.append(("xor8", 8, a, ln - a - 1))
    for a in seq:                                                          # has_crc32_checksum/3 :141-150
        if ln - a >= 4 and crc32(x[a:ln - 4]) == int.from_bytes(x[ln - 4:], "big"):
            out.append(("crc32", 32, a, ln - a - 4))
    return out


# ------------------------------------------------------------------------------------------------ erlamsa_strlex.erl
def texty(b):                                                              # :38-45
    return (31 < b <= 126) or b in (9, 10, 13)


def texty_enough(s, i):                                                    # :46-55 (?MIN_TEXTY = 6)
    n = 6
    while i < len(s) and n > 0:
        if not texty(s[i]):
            return False
        i, n = i + 1, n - 1
    return True


def lex(s):
    """lex/1 :59-111 -> [('byte', l) | ('text', l) | ('delimited', L, l, R)] (chunks in order)"""
    chunks, raw, i, n = [], [], 0, len(s)
    while True:                                                            # string_lex_step/3
        if i >= n:
            if raw:
                chunks.append(("byte", raw))
            return chunks
        if not texty_enough(s, i):
            raw.append(s[i])
            i += 1
            continue
        if raw:
            chunks.append(("byte", raw))
            raw = []
        seen = []                                                          # step_text/3 (seen kept in order)
        while True:
            if i >= n:
                chunks.append(("text", seen))
                return chunks
            h = s[i]
            if h in (34, 39):                                              # step_delimited/6: PrevR = [H | Seenr]
                start, after, i = h, [], i + 1
                while True:
                    if i >= n:
                        chunks.append(("text", seen + [start] + after))    # reverse(AfterR ++ PrevR)
                        return chunks
                    c = s[i]
                    if c == start:
                        if seen:
                            chunks.append(("text", seen))
                        chunks.append(("delimited", start, after, start))
                        i += 1
                        break
                    if c == 92 and i + 1 >= n:
                        after.append(92)
                        i += 1
                        continue
                    if c == 92:
                        if texty(s[i + 1]):
                            after += [92, s[i + 1]]
                            i += 2
                        else:
                            after.append(92)
                            i += 1
                        continue
                    if texty(c):
                        after.append(c)
                        i += 1
                        continue
                    chunks.append(("text", seen + [start] + after))
                    break
                break                                                      # back to string_lex_step with Rawr = []
            if texty(h):
                seen.append(h)
                i += 1
                continue
            chunks.append(("text", seen))
            break


def unlex(chunks):                                                         # :113-124
    out = []
    for c in chunks:
        out += ([c[1]] + c[2] + [c[3]]) if c[0] == "delimited" else c[1]
    return out


# ------------------------------------------------------------------------------------------------ ASCII mutators :430-651
SILLY = [list(b"%n"), list(b"%n"), list(b"%s"), list(b"%d"), list(b"%p"), list(b"%#x"), [0], list(b"aaaa%d%n"), [10], [13], [9], [8]]
DELIMS = [list(x) for x in (b"'", b'"', b"'", b'"', b"'", b'"', b"&", b":", b"|", b";", b"\\", b"\n", b"\r", b"\t", b" ", b"`", b"\0", b"]", b"[", b">", b"<")]
SHELLINJ = ["';%s;'", '";%s;"', ";%s;", "|%s#", "^ %s ^", "& %s &", "&& %s &&", "|| %s ||", "%%0D%s%%0D", "`%s`"]
REVCONN = ["calc.exe & notepad.exe %s %d ", "nc %s %d", "wget http://%s:%d", "curl %s %d", "exec 3<>/dev/tcp/%s/%d",
           "sleep 100000 # %s %d ", "echo>/tmp/erlamsa.%s.%d"]
SSRF = ("localhost", 51234)                                                # get_ssrf_ep/0 without the ETS table :698-703


def stringy(cs):                                                           # :440-442
    return any(c[0] != "byte" for c in cs)


def random_badness(rnd):                                                   # :469-477
    out = []
    for _ in range(rnd.rand(20) + 1):
        out = rnd.rand_elem(SILLY) + out
    return out


def rand_as_count(rnd):                                                    # :487-501
    t = rnd.rand(11)
    return [127, 128, 255, 256, 16383, 16384, 32767, 32768, 65535, 65536][t] if t < 10 else rnd.rand(1024)


def insert_traversal(rnd, symb):                                           # :509-511
    return symb + [x for _ in range(rnd.erand(10)) for x in [46, 46] + symb]


def mutate_text(rnd, kind, lst):                                           # :524-563
    if kind == "insert_badness":
        if not lst:
            return random_badness(rnd)
        p = rnd.erand(len(lst))
        bad = random_badness(rnd)
        return lst[:p - 1] + bad + lst[p - 1:]
    if kind == "replace_badness":
        if not lst:
            return random_badness(rnd)
        p = rnd.erand(len(lst))
        bad = random_badness(rnd)
        tail = lst[p:]
        return lst[:p - 1] + tail + bad[len(tail):]                        # overwrite(nthtail(P, Lst), Bad) :479-484
    if kind == "insert_aaas":
        if not lst:
            return [97] * rand_as_count(rnd)
        n = rand_as_count(rnd)
        p = rnd.erand(len(lst))
        return lst[:p - 1] + [97] * n + lst[p:]
    if kind == "insert_traversal":
        if not lst:
            return insert_traversal(rnd, [47])
        p = rnd.erand(len(lst))
        return lst[:p - 1] + insert_traversal(rnd, rnd.rand_elem([[92], [47]])) + lst[p:]
    if kind == "insert_null":
        return lst + [0]
    if kind == "insert_delimeter" or (kind == "insert_shellinj" and not lst):
        if not lst:
            return rnd.rand_elem(DELIMS)
        p = rnd.erand(len(lst))
        bad = rnd.rand_elem(DELIMS)
        return lst[:p - 1] + bad + lst[p - 1:]
    if kind == "insert_shellinj":
        p = rnd.erand(len(lst))
        inj = rnd.rand_elem(SHELLINJ)                                      # buildrevconnect/0 :516-520
        rev = rnd.rand_elem(REVCONN)
        sh = list((inj % (rev % SSRF)).encode())
        return lst[:p - 1] + sh + lst[p - 1:]
    raise AssertionError(kind)


def mutate_text_data(rnd, lst, kinds):
    return mutate_text(rnd, rnd.rand_elem(kinds), lst)                     # :513-514


def ascii_mutator(rnd, ll, fun):                                           # construct_ascii_mutator :586-602
    h, t = ll[0], ll[1:]
    cs = lex(list(h))
    if not stringy(cs):
        return ll, -1
    ms = fun(cs)
    d = rnd.rand_delta()
    return [bytes(unlex(ms))] + t, d


def string_generic_mutate(rnd, cs, kinds):                                 # :571-583
    l, r = len(cs), 0
    while True:
        if r > l / 4:
            return cs
        p = rnd.erand(l)
        el = cs[p - 1]
        if el[0] == "text":
            return cs[:p - 1] + [("text", mutate_text_data(rnd, el[1], kinds))] + cs[p:]
        if el[0] == "byte":
            r += 1
            continue
        return cs[:p - 1] + [("delimited", el[1], mutate_text_data(rnd, el[2], kinds), el[3])] + cs[p:]


def string_delimeter_mutate(rnd, cs):                                      # :626-644
    l, r = len(cs), 0
    while True:
        if r > l / 4:
            return cs
        p = rnd.erand(l)
        el = cs[p - 1]
        if el[0] == "text":
            kind = rnd.rand_elem(["insert_delimeter", "insert_delimeter", "insert_delimeter", "insert_shellinj"])
            return cs[:p - 1] + [("text", mutate_text_data(rnd, el[1], [kind]))] + cs[p:]
        if el[0] == "byte":
            r += 1
            continue
        k = rnd.rand(4)                                                    # drop_delimeter/2 :615-622
        drop = [("text", [el[1]] + el[2]), ("text", el[2] + [el[3]]), ("text", el[2]), el][k]
        return cs[:p - 1] + [drop] + cs[p:]


AB_KINDS = ["insert_badness", "replace_badness", "insert_traversal", "insert_aaas", "insert_null"]


# ------------------------------------------------------------------------------------------------ uri :727-784
def change_scheme(acc_rev):                                                # :727-729 (Acc is reversed)
    if acc_rev[:4] == list(b"elif"):
        return (list(b"ptth") + acc_rev[4:])[::-1]
    return acc_rev[::-1]


def tokens(t):                                                             # string:tokens(T, "/")
    return [list(x) for x in bytes(t).split(b"/") if x]


def try_uri_mutate(rnd, a):                                                # :764-768 + rand_uri_mutate :739-758
    for i in range(len(a) - 2):
        if a[i:i + 3] == [58, 47, 47]:
            acc_rev, t = a[:i][::-1], a[i + 3:]
            mode = rnd.erand(3)
            host, port = SSRF
            if mode == 1:
                return change_scheme(acc_rev) + list(("://%s:%d/" % (host, port)).encode()) + t, 1
            if mode == 2:
                at = list((rnd.rand_elem([" @%s:%d", "@%s:%d"]) % (host, port)).encode())
                tk = tokens(t)
                if not tk:
                    raise ErlCrash("badmatch: [Domain | Query] = []")
                q = [x for k, part in enumerate(tk[1:]) for x in ([47] if k else []) + part]
                return change_scheme(acc_rev) + list(b"://") + tk[0] + at + [47] + q, 1
            tk = tokens(t)
            if not tk:
                raise ErlCrash("badmatch: [Domain | Query] = []")
            trav = [47] + list(b"../") * rnd.erand(10)
            k = rnd.erand(4)
            q = [x for j, part in enumerate(tk[1:]) for x in ([47] if j else []) + part]
            tailq = [q, list(b"Windows/win.ini"), list(b"etc/shadow"), list(b"etc/passwd")][k - 1]
            return acc_rev[::-1] + list(b"://") + tk[0] + trav + tailq, 1
    return a, 0


def uri_mutator(rnd, ll, st):                                              # :771-784
    if st == "b64":
        raise Unmodelled("uri_mutator hands back fun base64_mutator/2 (:784): the slot is b64 from its second call on")
    h, t = ll[0], ll[1:]
    cs, d, ms = lex(list(h)), -1, []
    for c in cs:
        if c[0] == "text" and len(c[1]) > 5:
            na, nd = try_uri_mutate(rnd, c[1])
            ms.append(("text", na))
            d += nd
        else:
            ms.append(c)
    return [bytes(unlex(ms))] + t, d, "b64"


# ------------------------------------------------------------------------------------------------ guessed parse trees
# erlamsa_mutations.erl:787-1023, transcribed with Python lists as Erlang lists (bytes are ints, nodes are lists, all
# comparisons are by VALUE like =:=).  FALSE stands for the atom false (a Python False would equal the byte 0).
FALSE = object()
TREE_LIMIT = 128 << 10                                                      # bytes; beyond it the case is reported unmodelled


def usual_delims(c):                                                       # :793-799
    return {40: 41, 91: 93, 60: 62, 123: 125, 34: 34, 39: 39}.get(c, FALSE)


def grow(s, i, close, depth=0):                                            # :805-823 -> (node, next index | None)
    if depth > 1200:
        raise Unmodelled("nesting deeper than this model's Python recursion allows")
    rout = []
    while True:
        if i >= len(s):
            return rout, None                                              # out of data: partial parse
        h = s[i]
        if h == close:
            return rout + [close], i + 1
        nc = usual_delims(h)
        if nc is FALSE:
            rout.append(h)
            i += 1
            continue
        this, nxt = grow(s, i + 1, nc, depth + 1)
        if nxt is None:
            return rout + [h] + this, None                                 # lists:reverse(Rout) ++ [H | This]
        rout.append([h] + this)
        i = nxt


def partial_parse(s):                                                      # :892-905
    rout, i = [], 0
    while i < len(s):
        h = s[i]
        cp = usual_delims(h)
        if cp is FALSE:
            rout.append(h)
            i += 1
            continue
        this, nxt = grow(s, i + 1, cp)
        if nxt is None:
            return rout + [h] + this
        rout.append([h] + this)
        i = nxt
    return rout


def sublists(lst, found):                                                  # :838-845 (latest found first)
    for h in lst:
        if isinstance(h, list):
            found = sublists(h, [h] + found)
    return found


def edit_sublist(lst, sub, op):                                            # :858-869
    if not isinstance(lst, list):
        return [lst]                                                       # a byte: wrapped, an iolist all the same
    out = []
    for i, h in enumerate(lst):
        if h == sub:
            out.append(op(lst[i:]))                                        # Op gets the node AND its right siblings, and the walk ends
            return out
        out.append(edit_sublist(h, sub, op))
    out.append([])                                                         # the list's own [] tail goes through the last clause too
    return out


def edit_sublists(lst, mapping):                                           # :873-884; mapping = [(key node, replacement)]
    if not isinstance(lst, list):
        return lst
    out = []
    for h in lst:
        if isinstance(h, list):
            rep = next((r for k, r in mapping if k == h), FALSE)           # gb_trees lookup by value
            out.append(edit_sublists(h, mapping) if rep is FALSE else rep)
        else:
            out.append(h)
    return out


def io_len(x, memo):
    """flattened length of an iolist whose sublists may be shared (a DAG): iterative post-order, memo by identity"""
    if not isinstance(x, list):
        return 1
    stack = [(x, False)]
    while stack:
        node, done = stack.pop()
        k = id(node)
        if k in memo:
            continue
        if done:
            memo[k] = sum(memo[id(y)] if isinstance(y, list) else 1 for y in node)
        else:
            stack.append((node, True))
            stack.extend((y, False) for y in node if isinstance(y, list) and id(y) not in memo)
    return memo[id(x)]


def iolist_to_binary(x):
    if io_len(x, {}) > TREE_LIMIT:
        raise Unmodelled("tree result beyond the model's size limit")
    return flat(x)


def reservoir_sample(rnd, ll, k):                                          # erlamsa_rnd.erl:201-214
    if k >= len(ll):
        return ll
    r = ll[:k]
    for i in range(k + 1, len(ll) + 1):
        j = rnd.erand(i)
        if j <= k:
            r = r[:j - 1] + [ll[i - 1]] + r[j:]
    return r


def sed_tree_op(rnd, ll, op):                                              # :917-928
    h, t = ll[0], ll[1:]
    if binarish(h):
        return ll, -1
    if len(h) > TREE_LIMIT:
        raise Unmodelled("block too large for the Python tree model")
    lst = partial_parse(list(h))
    subs = sublists(lst, [])
    sub = rnd.rand_elem(subs) if subs else FALSE                           # pick_sublist/1 :847-853
    return [iolist_to_binary(edit_sublist(lst, sub, op))] + t, 1


def sed_tree_swap(rnd, ll, two):                                           # :940-971
    h, t = ll[0], ll[1:]
    if binarish(h):
        return ll, -1
    lst = partial_parse(list(h))
    subs = sublists(lst, [])
    if len(subs) < 2:
        return ll, -1
    if len(h) > TREE_LIMIT:
        raise Unmodelled("block too large for the Python tree model")
    toswap = reservoir_sample(rnd, subs, 2)
    if two:
        a, b = toswap[0], toswap[1]
        mapping = [(b, a)] if a == b else [(a, b), (b, a)]                 # enter(A -> B) then enter(B -> A): equal keys overwrite
        new = edit_sublists(lst, mapping)
    else:
        perm = rnd.random_permutation(toswap)
        a, b = perm[0], perm[1]
        new = edit_sublist(lst, a, lambda l: [b] + l[1:])
    return [iolist_to_binary(new)] + t, 1


def sed_tree_stutter(rnd, ll):                                             # :1005-1023
    h, t = ll[0], ll[1:]
    if binarish(h):
        return ll, -1
    if len(h) > TREE_LIMIT:
        raise Unmodelled("block too large for the Python tree model")
    lst = partial_parse(list(h))
    subs = sublists(lst, [])
    randsubs = rnd.random_permutation(subs)
    parent, child = FALSE, FALSE
    for cand in randsubs:                                                  # choose_stutr_nodes/1 :996-1002
        csubs = sublists(cand, [])
        if csubs:
            parent, child = cand, rnd.rand_elem(csubs)                     # choose_child/1 :988-993
            break
    n_reps = rnd.rand_log(10)
    if parent is FALSE:
        return ll, -1
    r = parent                                                             # repeat_path/3 :975-985, bottom-up (the 256 MB guard is the
    memo = {}                                                              # engine's work-area cap: not modelled, see TREE_LIMIT)
    for _ in range(2, n_reps + 1):
        prev = r
        r = edit_sublist(parent, child, lambda l, prev=prev: [prev] + l[1:])
        if io_len(r, memo) > TREE_LIMIT:
            raise Unmodelled("tree stutter beyond the model's size limit")
    new = edit_sublist(lst, child, lambda l: [r] + l[1:])
    return [iolist_to_binary(new)] + t, 1


# ------------------------------------------------------------------------------------------------ erlamsa_fuse.erl
# A suffix of a list is represented by its start position (len = the empty suffix []): suffixes of one list have distinct
# lengths, so the value comparisons of the reference (jump/3's first clause, fix_empty_list/1) are position comparisons.
def char_suffixes(x, sufs):                                                # :62-71 -> {char: [suffix, ...]} (latest first)
    subs = {}
    for p in sufs:
        if p == len(x):
            continue                                                       # ([], Subs) -> Subs
        el = [p + 1] + subs.get(x[p], [])
        if el == [len(x)]:
            el = []                                                        # fix_empty_list([[]]) -> []
        subs[x[p]] = el
    return subs


def fuse_split(a, b, node, acc):                                           # split/2 :85-100
    froms, tos = node
    sas, sbs = char_suffixes(a, froms), char_suffixes(b, tos)
    for ch in sorted(sas):                                                 # gb_trees:to_list/1: ascending keys
        sufs = sas[ch]
        if sufs == []:
            acc = [([len(a)], [len(b)])] + acc                             # [[[]], []]
        elif ch in sbs:
            acc = [(sufs, sbs[ch])] + acc
    return acc


def fuse(rnd, al, bl):                                                     # fuse/2 :131-134
    if not al:
        return bl
    if not bl:
        return al
    nodes = [(list(range(len(al))), list(range(len(bl))))]                 # find_jump_points/2 :103-107 (non-empty suffixes)
    fuel = 100000
    while True:                                                            # find_jump_points_loop/2 :115-128
        if fuel < 0:
            break
        if rnd.rand(8) == 0:
            break
        nodesp = []
        for nd in nodes:
            nodesp = fuse_split(al, bl, nd, nodesp)
        if not nodesp:
            break
        nodes, fuel = nodesp, fuel - len(nodesp)
    froms, tos = rnd.rand_elem(nodes)                                      # any_position_pair/1 :73-77
    frm = rnd.rand_elem(froms)
    to = rnd.rand_elem(tos)
    frm = len(al) if frm == [] else frm                                    # rand_elem([]) -> [] = the empty suffix
    to = len(bl) if to == [] else to
    return al[:frm] + bl[to:]                                              # jump/3 :47-50


def halve(l):                                                              # erlamsa_utils.erl:136-145
    k = len(l) // 2
    return l[:k], l[k:]


def sed_fuse_this(rnd, ll, st):                                            # :386-390
    b = fuse(rnd, ll[0], ll[0])
    return [b] + ll[1:], rnd.rand_delta(), st


def sed_fuse_next(rnd, ll, st):                                            # :393-402
    h, t = ll[0], ll[1:]
    al1, al2 = halve(h)
    b, rest = (t[0], t[1:]) if t else (h, [])                              # uncons(T, H)
    abl = fuse(rnd, al1, b)
    abal = fuse(rnd, abl, al2)
    d = rnd.rand_delta()
    return flush_bvecs(abal, rest), d, st


def sed_fuse_old(rnd, ll, st):                                             # :405-427, state = the remembered block
    h, t = ll[0], ll[1:]
    block = h if st is None else st
    al1, al2 = halve(h)
    ol1, ol2 = halve(block)
    a = fuse(rnd, al1, ol1)
    b = fuse(rnd, ol2, al2)
    swap = rnd.rand(3)
    d = rnd.rand_delta()
    return flush_bvecs(a, flush_bvecs(b, t)), d, (h if swap == 0 else block)


def make_table(rnd, snand_mask):
    """mutation functions by name; each: (ll, state) -> (ll', delta, state')"""
    def sr(h, bs, t, btail):                                               # construct_sed_bytes_repeat :273-281
        n = max(2, rnd.rand_log(10))
        return [h + bs * n + t] + btail

    masks = {"mask_nand": lambda b: b & ~(1 << rnd.rand(8)) & 255, "mask_or": lambda b: b | (1 << rnd.rand(8)),
             "mask_xor": lambda b: b ^ (1 << rnd.rand(8)), "mask_replace": lambda b: rnd.rand(256)}       # :295-307

    def uw(ll):                                                            # sed_utf8_widen :1081-1089
        return sed_byte(rnd, ll, lambda b: bytes([0xC0, b | 0x80]) if b == b & 0x3f else bytes([b]))

    def ui(ll):                                                            # sed_utf8_insert :1092-1099
        h, t = ll[0], ll[1:]
        p = rnd.rand(len(h))
        d = rnd.rand_delta()
        ins = bytes(rnd.rand_elem(FUNNY))
        if h == b"":
            return [h] + t, d
        return [h[:p + 1] + ins + h[p + 1:]] + t, d

    def stateless(f):
        return lambda ll, st: f(ll) + (st,)

    def line(op):
        return stateless(lambda ll: line_muta(rnd, ll, lambda ls, n: op(rnd, ls, n)))

    tab = {
        "num": lambda ll: sed_num(rnd, ll),
        "bd": lambda ll: sed_byte(rnd, ll, lambda b: b""),
        "bei": lambda ll: sed_byte(rnd, ll, lambda b: bytes([(b + 1) & 255])),
        "bed": lambda ll: sed_byte(rnd, ll, lambda b: bytes([(b - 1) & 255])),
        "bf": lambda ll: sed_byte(rnd, ll, lambda b: bytes([b ^ (1 << rnd.rand(8))])),
        "bi": lambda ll: sed_byte(rnd, ll, lambda b: bytes([rnd.rand(256), b])),
        "ber": lambda ll: sed_byte(rnd, ll, lambda b: bytes([rnd.rand(256)])),
        "br": lambda ll: sed_byte(rnd, ll, lambda b: bytes([b, b])),
        "sp": lambda ll: sed_bytes(rnd, ll, lambda h, bs, t, bt: [h + bytes(rnd.random_permutation(list(bs))) + t] + bt),
        "sr": lambda ll: sed_bytes(rnd, ll, sr),
        "sd": lambda ll: sed_bytes(rnd, ll, lambda h, bs, t, bt: [h + t] + bt),
        "snand": lambda ll: sed_bytes(rnd, ll, lambda h, bs, t, bt: [h + bytes(randmask(rnd, masks[snand_mask], list(bs))) + t] + bt),
        "srnd": lambda ll: sed_bytes(rnd, ll, lambda h, bs, t, bt: [h + bytes(randmask(rnd, masks["mask_replace"], list(bs))) + t] + bt),
        "uw": uw, "ui": ui,
        "len": lambda ll: length_predict(rnd, ll),
        "ab": lambda ll: ascii_mutator(rnd, ll, lambda cs: string_generic_mutate(rnd, cs, AB_KINDS)),       # :605-611
        "ad": lambda ll: ascii_mutator(rnd, ll, lambda cs: string_delimeter_mutate(rnd, cs)),               # :647-651
        "tr2": lambda ll: sed_tree_op(rnd, ll, lambda node: [node[0]] + node),                    # sed_tree_dup :931-932
        "td": lambda ll: sed_tree_op(rnd, ll, lambda node: node[1:]),                             # sed_tree_del :935-936
        "ts1": lambda ll: sed_tree_swap(rnd, ll, False), "ts2": lambda ll: sed_tree_swap(rnd, ll, True),
        "tr": lambda ll: sed_tree_stutter(rnd, ll),
        "nil": lambda ll: (ll, -1),                                        # nomutation/2 :1104-1105
    }
    tab = {k: stateless(v) for k, v in tab.items()}
    tab.update({"ld": line(list_del), "lds": line(list_del_seq), "lr2": line(list_dup), "lri": line(list_clone),
                "lr": line(list_repeat), "ls": line(list_swap), "lp": line(list_perm),
                "uri": lambda ll, st: uri_mutator(rnd, ll, st),
                "ft": lambda ll, st: sed_fuse_this(rnd, ll, st), "fn": lambda ll, st: sed_fuse_next(rnd, ll, st),
                "fo": lambda ll, st: sed_fuse_old(rnd, ll, st),
                "lis": lambda ll, st: st_line_muta(rnd, ll, st, lambda x: lambda t, r: [x, t] + r),          # st_list_ins :156-158
                "lrs": lambda ll, st: st_line_muta(rnd, ll, st, lambda x: lambda _t, r: [x] + r)})           # st_list_replace :161-163
    return tab


# table order of mutations/1 (:1290-1331), restricted to what this model implements
TABLE_ORDER = ["uw", "ui", "ab", "ad", "tr2", "td", "num", "ts1", "tr", "ts2", "bd", "bei", "bed", "bf", "bi", "ber", "br", "sp", "sr", "sd", "snand", "srnd",
               "ld", "lds", "lr2", "lri", "lr", "ls", "lp", "lis", "lrs", "ft", "fn", "fo", "len", "uri", "nil"]


def adjust_priority(pri, delta):                                           # :1240-1242
    return pri if delta == 0 else max(MIN_SCORE, min(MAX_SCORE, pri + delta))


def weighted_permutations(rnd, pris):                                      # :1246-1250
    ppris = [(rnd.rand(math.trunc(x[0] * x[1])), x) for x in pris]
    return [x for _, x in lists_sort(lambda a, b: a[0] >= b[0], ppris)]


def mux_fuzzers(rnd, table, fs, ll):
    """the fun mux_fuzzers/1 returns (:1258-1265) + mux_fuzzers_loop/4 (:1268-1281) -> (fs', ll')"""
    if ll == [b""]:
        return fs, ll
    if ll == []:
        return fs, b""                                                     # a binary, not a list: the caller's ++ crashes
    nodes, out = weighted_permutations(rnd, fs), []
    while nodes:
        node, tail = nodes[0], nodes[1:]
        if len(ll[0]) > ABSMAX_BINARY_BLOCK:
            return out + tail, ll
        score, pri, name, st = node
        mll, delta, st = table[name](ll, st)
        out = [(adjust_priority(score, delta), pri, name, st)] + out
        if mll[0] == ll[0]:
            nodes = tail
            continue
        return out + tail, mll
    return out, ll


# ------------------------------------------------------------------------------------------------ erlamsa_gen.erl
def rand_block_size(rnd, scale):                                           # :54-56
    return max(rnd.rand(round(MAX_BLOCK_SIZE * scale)), round(MIN_BLOCK_SIZE * scale))


def finish(rnd, length):                                                   # :42-51
    n = rnd.rand(length + 1)
    if n == length:
        bits = rnd.rand_range(1, 16)
        nlen = rnd.rand(1 << bits)
        blk = bytes(rnd.random_numbers(256, nlen))
        return [] if blk == b"" else [blk]
    return []


def direct_generator(rnd, data, scale):                                    # :152-164
    rand_block_size(rnd, scale)                 # the argument is evaluated, then split_binary's first guard
    return [data] + finish(rnd, len(data))      # (byte_size(Wanted) of an integer) fails: never split


def random_stream(rnd, scale):                                             # :167-177
    out = []
    while True:
        n = rnd.rand_range(32, round(MAX_BLOCK_SIZE * scale))
        out.append(rnd.random_block(n))
        ip = rnd.rand_range(1, 100)
        if rnd.rand(ip) == 0:
            return out


# ------------------------------------------------------------------------------------------------ erlamsa_patterns.erl
REMUTATE = (4, 5)
ALL_PATTERNS = ["od", "nd", "bu", "sk", "sz", "cs", "ar", "cp", "co", "nu"]       # patterns/0 :395-404, in table order
MODELLED_PATTERNS = ("od", "nd", "bu", "sk", "sz", "cs", "co", "nu")


class Patterns:
    """pat_once_dec / pat_many_dec / pat_burst / skipper / pat_50_muta / pat_nomuta (:146-163, :308-394) over mutate_once/4
    (:267-279) and mutate_once_loop/6 (:283-297).  Blocks go to `written` in the order erlamsa_out:blocks_port/5
    (:642-655) writes them; continuations are Python closures (l, fs) -> None."""

    def __init__(self, rnd, table, written):
        self.rnd, self.table, self.written = rnd, table, written

    def emit(self, l):                                                     # L ++ [{M, Mt}], then written by blocks_port
        if not isinstance(l, list):
            raise ErlCrash("badarg: <<>> ++ [..]")
        self.written.extend(l)

    def split(self, this, rest):                                           # split/1 + split_into_maxblocks/2 :44-59
        if len(this) > ABSMAX_BINARY_BLOCK:
            pieces = []
            while len(this) > ABSMAX_BINARY_BLOCK:
                cut = 500000 + self.rnd.rand(500000) - 1
                pieces.append(this[:cut])
                this = this[cut:]
            pieces.append(this)                                            # cons_revlst(Lst, LlN): pieces in order, then LlN
            this, rest = pieces[0], pieces[1:] + rest
        return this, rest

    def mutate_once(self, ll, fs, cont):
        if ll == [b""]:
            return                                                         # {Mutator, Meta}: nothing more is written
        if not isinstance(ll, list):
            raise ErlCrash("a binary where the block list should be")      # unreachable with these mutators
        ip = self.rnd.rand(24)                                             # ?INITIAL_IP
        if not ll:
            return cont([], fs)                                            # uncons([], false): Cont([], Mutator, Meta)
        this, rest = self.split(ll[0], ll[1:])
        return self.mutate_once_loop(ip, this, rest, fs, cont)

    def mutate_once_loop(self, ip, this, rest, fs, cont):
        while True:
            n = self.rnd.rand(ip)
            if n == 0 or rest == []:
                fs, l = mux_fuzzers(self.rnd, self.table, fs, [this] + rest)
                return cont(l, fs)
            self.written.append(this)
            this, rest = rest[0], rest[1:]

    def run(self, name, ll, fs):
        rnd = self.rnd
        if name == "od":                                                   # pat_once_dec :308-309
            return self.mutate_once(ll, fs, lambda l, fs2: self.emit(l))
        if name == "nd":                                                   # pat_many_dec :325-326 + _cont :315-321
            def cont(l, fs2):
                if rnd.rand_occurs_fixed(*REMUTATE):
                    return self.run("nd", l, fs2)
                return self.emit(l)
            return self.mutate_once(ll, fs, cont)
        if name == "bu":                                                   # pat_burst :348-349 + _cont :332-345
            def cont(l, fs2):
                n = 1
                while True:
                    p = rnd.rand_occurs_fixed(*REMUTATE)
                    if p or n < 2:
                        if not isinstance(l, list):
                            raise ErlCrash("mux_fuzzers: no clause for a binary")
                        fs2, l = mux_fuzzers(rnd, self.table, fs2, l)
                        n += 1
                    else:
                        return self.emit(l)
            return self.mutate_once(ll, fs, cont)
        if name == "sk":                                                   # make_complex_pat :351-355 + mutate_once_skipper :147-163
            nxt = rnd.rand_elem(ALL_PATTERNS)
            ip = rnd.rand(24)
            if not ll:
                raise ErlCrash("size(false)")
            b, rest = ll[0], ll[1:]
            length = rnd.rand(math.trunc(len(b) / 2))
            head, tail = b[:length], b[length:]
            this, rest = self.split(tail, rest)
            self.written.append(head)                                      # [<<HeadBin:Len>> | Res]
            if nxt not in MODELLED_PATTERNS:
                raise Unmodelled(nxt)
            return self.mutate_once_loop(ip, this, rest, fs, lambda l, fs2: self.run(nxt, l, fs2))
        if name in ("sz", "cs"):                                           # make_complex_pat + mutate_once_sizer / _csum :83-145
            nxt = rnd.rand_elem(ALL_PATTERNS)
            ip = rnd.rand(24)
            if not ll:
                raise ErlCrash("size(false)")
            b, rest = ll[0], ll[1:]
            locs = get_possible_simple_lens(rnd, b) if name == "sz" else get_possible_csum_locations(b)
            elem = rnd.rand_elem(locs)
            if nxt not in MODELLED_PATTERNS:
                raise Unmodelled(nxt)
            cont = lambda l, fs2: self.run(nxt, l, fs2)
            if elem == []:                                                 # "failed": the whole block, next pattern as continuation
                this, rest = self.split(b, rest)
                return self.mutate_once_loop(ip, this, rest, fs, cont)
            if name == "sz":
                _ok, size, endian, ln, a, _b = elem
                nb = size // 8
                head, blob, tail = b[:a], b[a + nb:a + nb + ln], b[a + nb + ln:]
            else:
                typ, size, plen, blen = elem
                head, blob, tail = b[:plen], b[plen:plen + blen], None
            this, rest = self.split(blob, rest)
            outer, self.written = self.written, []                         # prepare4sizer/1 :62-78 forces and joins everything
            try:                                                           # the inner evaluation writes
                self.mutate_once_loop(ip, this, rest, fs, cont)
                newblob = b"".join(self.written)
            finally:
                self.written = outer
            if name == "sz":
                self.written.extend([head + field(len(newblob), size, endian) + newblob, tail])
            else:
                v = crc32(newblob) if typ == "crc32" else 0
                if typ == "xor8":
                    for c in newblob:
                        v ^= c
                self.written.append(head + newblob + field(v, size, "big"))
            return
        if name == "co":                                                   # pat_50_muta :378-382
            return self.run("nu" if rnd.erand(2) == 1 else "od", ll, fs)
        if name == "nu":                                                   # pat_nomuta :386-388
            if ll:
                this, rest = self.split(ll[0], ll[1:])
                self.written.extend([this] + rest)
            return
        raise Unmodelled(name)


# ------------------------------------------------------------------------------------------------ erlamsa_main.erl
def fuzzer(inputs, seed, mutations, patterns, blockscale=1.0, first_case=1):
    """erlamsa_main:fuzzer/1 (:124-232) for paths = [direct], generators = default, workers = 1; iteration I mutates
    inputs[I - first_case].  mutations / patterns: [(name, pri)].  -> [(status, bytes)], status 0 ok / 1 crashed."""
    parent = Rnd()
    parent.seed(seed)                                                      # :135
    # make_mutator/2 :1371-1384 (foldl prepends) + mutators_mutator/2 :1391-1395
    # mutations/1 (:1290-1331) is evaluated as make_mutator's fold argument: building the table runs
    # construct_sed_bytes_randmask/1 (:311-312) for snand and srnd, and each draws its MaskFun with rand_elem/1 — two draws
    # of the parent stream whether or not those mutators are selected.  (The first version of this model missed them;
    # diffing against the oracle found it — the oracle had it right.)
    snand_mask = parent.rand_elem(["mask_nand", "mask_or", "mask_xor"])
    parent.rand_elem(["mask_replace"])
    sel = dict(mutations)
    mutas = []
    for name in TABLE_ORDER:
        if name in sel:
            mutas.insert(0, (sel[name], name))
    fs = []
    for pri, name in mutas:
        n = parent.rand(math.trunc(MAX_SCORE))
        fs.insert(0, (max(2, n), pri, name, None if name == "fo" else [0]))   # [0]: InitialState of lis / lrs; fo: no block remembered yet
    # make_generator/5 :244-247 with Args = [direct]: random (1) and direct (500) survive; mux_generators/2 :193-199
    gens, total = sort_by_priority([(1, "random"), (500, "direct")])
    gen = choose_pri(gens, parent.rand(total))
    # make_pattern/1 :417-429 (foldl prepends) + mux_patterns/1 :438-443
    psel = dict(patterns)
    pats = []
    for name in ALL_PATTERNS:
        if name in psel:
            pats.insert(0, (psel[name], name))
    spats, ptotal = sort_by_priority(pats)
    res = []
    # the loop runs from case 1: skip the ThreadSeed draws of the cases before first_case
    for _ in range(3 * (first_case - 1)):
        parent.erand(99999)
    for data in inputs:
        tseed = parent.gen_predictable_seed()                              # :179
        rnd = Rnd()
        rnd.seed(tseed)                                                    # :183
        table = make_table(rnd, snand_mask)
        try:
            ll = direct_generator(rnd, data, blockscale) if gen == "direct" else random_stream(rnd, blockscale)   # :185
            pat = choose_pri(spats, rnd.rand(ptotal))                      # choose_pattern_fun :431-434
            written = []
            Patterns(rnd, table, written).run(pat, ll, list(fs))
            res.append((0, b"".join(written)))
        except ErlCrash:
            res.append((1, b""))
        except Unmodelled:
            res.append(None)
    return res
