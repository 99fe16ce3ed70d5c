// ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// otp_compat.h: CPU restatement of the pieces of Erlang/OTP stdlib that
// erlamsa's hot path depends on but which live OUTSIDE /root/reference
// (un-vendored, OTP version unpinned: reference README.md:32 "OTP 18.0+",
// .travis.yml:5-11 matrix 18.0..23.0).
//
//   * `random` (AS183, Wichmann-Hill 1982)  -- used at erlamsa_rnd.erl:73,78,83,101,105,151,196
//   * `lists:sort/2` merge-sort tie behaviour -- erlamsa_mutations.erl:1249, erlamsa_utils.erl:115
//   * erts big_to_double (bignum -> float)   -- erlamsa_rnd.erl:78 with bignum N
//   * `base64`, `erlang:crc32`               -- erlamsa_mutations.erl:667,674; erlamsa_field_predict.erl:148,165
//
// PARITY UNPINNED: these are restated from the published algorithms / recalled
// OTP sources; no Erlang runtime exists on this image to check them against and
// the reference's own tests (erlamsa_mutations_test.erl) pin no byte-exact
// vectors (they seed from now()).  See DESIGN.md "Oracle".
#pragma once
#include <memory>
#include <functional>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

namespace otp {

// A crash of the Erlang worker process (badmatch / function_clause / badarith ...).
// erlamsa_main.erl:211-220: the case's result is then <<>>.
struct ErlCrash : std::runtime_error {
  explicit ErlCrash(const char* why) : std::runtime_error(why) {}
};

// ---------------------------------------------------------------------------
// random.erl (AS183).  seed/3: Ai := (|x| rem (Pi-1)) + 1 ; uniform/0:
//   B1=(A1*171) rem 30269, B2=(A2*172) rem 30307, B3=(A3*170) rem 30323,
//   R = B1/30269 + B2/30307 + B3/30323,  U = R - trunc(R)
// ---------------------------------------------------------------------------
struct Random {
  uint32_t a1 = 3172, a2 = 9814, a3 = 20125;  // seed0()
  uint64_t draws = 0;                          // diagnostic only
  void seed(int64_t s1, int64_t s2, int64_t s3) {
    a1 = (uint32_t)((s1 < 0 ? -s1 : s1) % 30268) + 1;
    a2 = (uint32_t)((s2 < 0 ? -s2 : s2) % 30306) + 1;
    a3 = (uint32_t)((s3 < 0 ? -s3 : s3) % 30322) + 1;
  }
  double uniform() {
    a1 = (a1 * 171u) % 30269u;
    a2 = (a2 * 172u) % 30307u;
    a3 = (a3 * 170u) % 30323u;
    ++draws;
    volatile double q1 = (double)a1 / 30269.0;
    volatile double q2 = (double)a2 / 30307.0;
    volatile double q3 = (double)a3 / 30323.0;
    volatile double r = q1 + q2;
    r = r + q3;
    return r - std::trunc(r);
  }
  // random:uniform(N) = trunc(uniform() * N) + 1 for machine-size N.
  uint64_t uniform_n(uint64_t n) {
    volatile double x = uniform() * (double)n;
    return (uint64_t)std::trunc(x) + 1;
  }
};

// ---------------------------------------------------------------------------
// lists:sort/2 — stdlib lists.erl, the oracle's OWN restatement (the engine's host set-up has another one, written
// independently over index cursors: erlamsa_amd/csrc/eh_otp_sort.h; tests/test_oracle_otp.py diffs the two and the third
// one in tests/pymodel.py on every priority list of up to 7 entries and on random ones).  erlamsa_utils:sort_by_priority/1
// (erlamsa_utils.erl:113-117) passes a strict '>' — not a total "=<" — so the order among equal priorities is whatever
// this exact merge sort does with it.  Clause for clause over immutable cons lists, as the Erlang reads:
//   sort/2, fsplit_1, fsplit_1_1, fsplit_2, fsplit_2_1, fmergel, rfmergel, fmerge2_1/_2, rfmerge2_1/_2.
// ---------------------------------------------------------------------------
template <class T>
class ListsSort {
  struct Cell; using P = std::shared_ptr<const Cell>;
  struct Cell { T hd; P tl; };
  struct LCell; using PL = std::shared_ptr<const LCell>;       // a list of lists
  struct LCell { P hd; PL tl; };
  using F = std::function<bool(const T&, const T&)>;
  F fun;
  static P cons(const T& h, P t) { return std::make_shared<const Cell>(Cell{h, std::move(t)}); }
  static PL lcons(P h, PL t) { return std::make_shared<const LCell>(LCell{std::move(h), std::move(t)}); }
  static P reverse(P l, P tail) { while (l) { tail = cons(l->hd, tail); l = l->tl; } return tail; }   // lists:reverse/2
  enum Ord { asc, desc };

  // fmerge2_1([H1|T1], H2, Fun, T2, M) / fmerge2_2(H1, T1, Fun, [H2|T2], M): elements of the first list are prioritised
  P fmerge2_1(P l1, const T& h2, P t2, P m) {
    for (;;) {
      if (!l1) return reverse(t2, cons(h2, m));                                   // fmerge2_1([], H2, _, T2, M)
      if (fun(l1->hd, h2)) { m = cons(l1->hd, m); l1 = l1->tl; continue; }        // true -> fmerge2_1(T1, H2, Fun, T2, [H1|M])
      return fmerge2_2(l1->hd, l1->tl, t2, cons(h2, m));                          // false -> fmerge2_2(H1, T1, Fun, T2, [H2|M])
    }
  }
  P fmerge2_2(const T& h1, P t1, P l2, P m) {
    for (;;) {
      if (!l2) return reverse(t1, cons(h1, m));                                   // fmerge2_2(H1, T1, _, [], M)
      if (fun(h1, l2->hd)) return fmerge2_1(t1, l2->hd, l2->tl, cons(h1, m));     // true -> fmerge2_1(T1, H2, Fun, T2, [H1|M])
      m = cons(l2->hd, m); l2 = l2->tl;                                           // false -> fmerge2_2(H1, T1, Fun, T2, [H2|M])
    }
  }
  P rfmerge2_1(P l1, const T& h2, P t2, P m) {
    for (;;) {
      if (!l1) return reverse(t2, cons(h2, m));
      if (fun(l1->hd, h2)) return rfmerge2_2(l1->hd, l1->tl, t2, cons(h2, m));    // true -> rfmerge2_2(H1, T1, Fun, T2, [H2|M])
      m = cons(l1->hd, m); l1 = l1->tl;                                           // false -> rfmerge2_1(T1, H2, Fun, T2, [H1|M])
    }
  }
  P rfmerge2_2(const T& h1, P t1, P l2, P m) {
    for (;;) {
      if (!l2) return reverse(t1, cons(h1, m));
      if (fun(h1, l2->hd)) { m = cons(l2->hd, m); l2 = l2->tl; continue; }        // true -> rfmerge2_2(H1, T1, Fun, T2, [H2|M])
      return rfmerge2_1(t1, l2->hd, l2->tl, cons(h1, m));                         // false -> rfmerge2_1(T1, H2, Fun, T2, [H1|M])
    }
  }
  P fmergel(PL l, PL acc, Ord o) {
    for (;;) {
      if (l && l->tl) {
        P a = l->hd, b = l->tl->hd; PL rest = l->tl->tl;
        // asc: fmergel([T1, [H2|T2] | L]) ; desc: fmergel([[H2|T2], T1 | L])  -> [fmerge2_1(T1, H2, Fun, T2, []) | Acc]
        P t1 = o == asc ? a : b, l2 = o == asc ? b : a;
        acc = lcons(fmerge2_1(t1, l2->hd, l2->tl, nullptr), acc); l = rest; continue;
      }
      if (l && !acc) return l->hd;                                                 // fmergel([L], [], _, _) -> L
      if (l) return rfmergel(lcons(reverse(l->hd, nullptr), acc), nullptr, o);     // fmergel([L], Acc, Fun, O)
      return rfmergel(acc, nullptr, o);                                            // fmergel([], Acc, Fun, O)
    }
  }
  P rfmergel(PL l, PL acc, Ord o) {
    for (;;) {
      if (l && l->tl) {
        P a = l->hd, b = l->tl->hd; PL rest = l->tl->tl;
        // asc: rfmergel([[H2|T2], T1 | L]) ; desc: rfmergel([T1, [H2|T2] | L]) -> [rfmerge2_1(T1, H2, Fun, T2, []) | Acc]
        P t1 = o == asc ? b : a, l2 = o == asc ? a : b;
        acc = lcons(rfmerge2_1(t1, l2->hd, l2->tl, nullptr), acc); l = rest; continue;
      }
      if (l) return fmergel(lcons(reverse(l->hd, nullptr), acc), nullptr, o);      // rfmergel([L], Acc, Fun, O)
      return fmergel(acc, nullptr, o);                                             // rfmergel([], Acc, Fun, O)
    }
  }
  // fsplit_1 / fsplit_1_1 (ascending runs; inv = false) and fsplit_2 / fsplit_2_1 (descending; every test negated)
  P fsplit(bool inv, T y, T x, P l, P r, PL rs) {
    auto t = [&](const T& a, const T& b) { return fun(a, b) != inv; };
    for (;;) {
      if (!l) { PL all = lcons(cons(y, cons(x, r)), rs); return inv ? fmergel(all, nullptr, desc) : rfmergel(all, nullptr, asc); }
      const T z = l->hd; l = l->tl;
      if (t(y, z)) { r = cons(x, r); x = y; y = z; continue; }                     // fsplit_1(Z, Y, Fun, L, [X|R], Rs)
      if (t(x, z)) { r = cons(x, r); x = z; continue; }                            // fsplit_1(Y, Z, Fun, L, [X|R], Rs)
      if (!r) { r = cons(z, nullptr); continue; }                                  // when R == [] -> fsplit_1(Y, X, Fun, L, [Z], Rs)
      return fsplit_x_1(inv, y, x, l, r, rs, z);                                   // fsplit_1_1(Y, X, Fun, L, R, Rs, Z)
    }
  }
  P fsplit_x_1(bool inv, T y, T x, P l, P r, PL rs, T s) {
    auto t = [&](const T& a, const T& b) { return fun(a, b) != inv; };
    for (;;) {
      if (!l) { PL all = lcons(cons(s, nullptr), lcons(cons(y, cons(x, r)), rs)); return inv ? fmergel(all, nullptr, desc) : rfmergel(all, nullptr, asc); }
      const T z = l->hd; l = l->tl;
      if (t(y, z)) { r = cons(x, r); x = y; y = z; continue; }
      if (t(x, z)) { r = cons(x, r); x = z; continue; }
      PL rs2 = lcons(cons(y, cons(x, r)), rs);
      if (t(s, z)) return fsplit(inv, z, s, l, nullptr, rs2);                      // fsplit_1(Z, S, Fun, L, [], [[Y, X | R] | Rs])
      return fsplit(inv, s, z, l, nullptr, rs2);                                   // fsplit_1(S, Z, Fun, L, [], [[Y, X | R] | Rs])
    }
  }

 public:
  explicit ListsSort(F f) : fun(std::move(f)) {}
  std::vector<T> sort(const std::vector<T>& in) {                                  // sort/2
    if (in.size() < 2) return in;
    P l = nullptr;
    for (size_t i = in.size(); i-- > 2;) l = cons(in[i], l);
    const T &x = in[0], &y = in[1];
    P res = fun(x, y) ? fsplit(false, y, x, l, nullptr, nullptr) : fsplit(true, y, x, l, nullptr, nullptr);
    std::vector<T> out;
    for (; res; res = res->tl) out.push_back(res->hd);
    return out;
  }
};

// ---------------------------------------------------------------------------
// erlang:crc32/1 — zlib CRC-32 (reflected 0xEDB88320, init/xorout 0xFFFFFFFF)
// ---------------------------------------------------------------------------
inline uint32_t crc32(const uint8_t* p, size_t n) {
  static uint32_t tab[256]; static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; tab[i] = c; }
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  for (size_t i = 0; i < n; i++) c = tab[(c ^ p[i]) & 0xFF] ^ (c >> 8);
  return c ^ 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------
// Minimal sign-magnitude bignum (base 2^32) with the Erlang semantics needed by
// sed_num (erlamsa_mutations.erl:93-169): decimal parse/print, +, -, *2,
// float conversion as erts big_to_double (d = d*2^64 + digit, 64-bit digits),
// and trunc(float) -> integer.
// ---------------------------------------------------------------------------
struct Big {
  bool neg = false;
  std::vector<uint32_t> mag;  // little endian, no leading zero limbs; empty = 0
  Big() {}
  Big(int64_t v) { neg = v < 0; uint64_t u = neg ? (uint64_t)(-(v + 1)) + 1 : (uint64_t)v; while (u) { mag.push_back((uint32_t)u); u >>= 32; } }
  static Big from_u64(uint64_t u) { Big b; while (u) { b.mag.push_back((uint32_t)u); u >>= 32; } return b; }
  bool is_zero() const { return mag.empty(); }
  void trim() { while (!mag.empty() && mag.back() == 0) mag.pop_back(); if (mag.empty()) neg = false; }
  static int cmp_mag(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
    for (size_t i = a.size(); i-- > 0;) if (a[i] != b[i]) return a[i] < b[i] ? -1 : 1;
    return 0;
  }
  static std::vector<uint32_t> add_mag(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {
    std::vector<uint32_t> r; uint64_t c = 0; size_t n = std::max(a.size(), b.size());
    for (size_t i = 0; i < n; i++) { uint64_t s = c + (i < a.size() ? a[i] : 0) + (i < b.size() ? b[i] : 0); r.push_back((uint32_t)s); c = s >> 32; }
    if (c) r.push_back((uint32_t)c);
    return r;
  }
  static std::vector<uint32_t> sub_mag(const std::vector<uint32_t>& a, const std::vector<uint32_t>& b) {  // a >= b
    std::vector<uint32_t> r; int64_t br = 0;
    for (size_t i = 0; i < a.size(); i++) { int64_t d = (int64_t)a[i] - (i < b.size() ? b[i] : 0) - br; br = d < 0; if (d < 0) d += ((int64_t)1 << 32); r.push_back((uint32_t)d); }
    while (!r.empty() && r.back() == 0) r.pop_back();
    return r;
  }
  Big operator-() const { Big r = *this; if (!r.is_zero()) r.neg = !neg; return r; }
  Big operator+(const Big& o) const {
    Big r;
    if (neg == o.neg) { r.mag = add_mag(mag, o.mag); r.neg = neg; }
    else { int c = cmp_mag(mag, o.mag); if (c == 0) return Big(); if (c > 0) { r.mag = sub_mag(mag, o.mag); r.neg = neg; } else { r.mag = sub_mag(o.mag, mag); r.neg = o.neg; } }
    r.trim(); return r;
  }
  Big operator-(const Big& o) const { return *this + (-o); }
  Big abs() const { Big r = *this; r.neg = false; return r; }
  Big mul_small(uint32_t m) const { Big r; r.neg = neg; uint64_t c = 0; for (uint32_t d : mag) { uint64_t p = (uint64_t)d * m + c; r.mag.push_back((uint32_t)p); c = p >> 32; } if (c) r.mag.push_back((uint32_t)c); r.trim(); return r; }
  Big mul(const Big& o) const {
    Big r; if (is_zero() || o.is_zero()) return r;
    r.mag.assign(mag.size() + o.mag.size(), 0);
    for (size_t i = 0; i < mag.size(); i++) { uint64_t c = 0; for (size_t j = 0; j < o.mag.size(); j++) { uint64_t p = (uint64_t)mag[i] * o.mag[j] + r.mag[i + j] + c; r.mag[i + j] = (uint32_t)p; c = p >> 32; } r.mag[i + o.mag.size()] += (uint32_t)c; }
    r.neg = neg != o.neg; r.trim(); return r;
  }
  static Big pow2(unsigned k) { Big r; r.mag.assign(k / 32 + 1, 0); r.mag[k / 32] = 1u << (k % 32); return r; }
  Big bor(const Big& o) const {  // both non-negative
    Big r; size_t n = std::max(mag.size(), o.mag.size()); r.mag.assign(n, 0);
    for (size_t i = 0; i < n; i++) r.mag[i] = (i < mag.size() ? mag[i] : 0) | (i < o.mag.size() ? o.mag[i] : 0);
    r.trim(); return r;
  }
  // N*10 + d accumulation (get_num, erlamsa_mutations.erl:119-120)
  void mul10_add(uint32_t d) { uint64_t c = d; for (auto& l : mag) { uint64_t p = (uint64_t)l * 10 + c; l = (uint32_t)p; c = p >> 32; } if (c) mag.push_back((uint32_t)c); }
  std::string to_dec() const {  // integer_to_list/1
    if (is_zero()) return "0";
    std::vector<uint32_t> m = mag; std::string out;
    while (!m.empty()) {
      uint64_t rem = 0;
      for (size_t i = m.size(); i-- > 0;) { uint64_t cur = (rem << 32) | m[i]; m[i] = (uint32_t)(cur / 1000000000u); rem = cur % 1000000000u; }
      while (!m.empty() && m.back() == 0) m.pop_back();
      for (int k = 0; k < 9; k++) { out.push_back((char)('0' + rem % 10)); rem /= 10; if (m.empty() && rem == 0) break; }
    }
    while (out.size() > 1 && out.back() == '0') out.pop_back();
    if (neg) out.push_back('-');
    return std::string(out.rbegin(), out.rend());
  }
  // erts big_to_double (64-bit ErtsDigit): d = d*2^64 + digit, most significant first.
  // Returns false on overflow (-> badarith in the caller).
  bool to_double(double* out) const {
    volatile double d = 0.0;
    size_t nd = (mag.size() + 1) / 2;
    for (size_t k = nd; k-- > 0;) {
      uint64_t lo = mag[2 * k], hi = (2 * k + 1 < mag.size()) ? mag[2 * k + 1] : 0;
      uint64_t dig = (hi << 32) | lo;
      d = d * 18446744073709551616.0 + (double)dig;
      if (!std::isfinite((double)d)) return false;
    }
    *out = neg ? -(double)d : (double)d;
    return true;
  }
  // trunc(F) for a finite non-negative double -> exact integer
  static Big from_double_trunc(double f) {
    Big r; bool ng = f < 0; if (ng) f = -f; f = std::trunc(f);
    if (f == 0) return r;
    int e; double m = std::frexp(f, &e);           // f = m * 2^e, m in [0.5,1)
    uint64_t mant = (uint64_t)std::ldexp(m, 53);   // 53-bit integer mantissa
    int sh = e - 53;
    if (sh <= 0) { r = from_u64(mant >> (-sh)); }
    else { r = from_u64(mant); r = r.mul(pow2((unsigned)sh)); }
    r.neg = ng && !r.is_zero(); return r;
  }
  bool fits_i64(int64_t* v) const {
    if (mag.size() > 2) return false;
    uint64_t u = 0; for (size_t i = mag.size(); i-- > 0;) u = (u << 32) | mag[i];
    if (!neg && u <= (uint64_t)INT64_MAX) { *v = (int64_t)u; return true; }
    if (neg && u <= (uint64_t)INT64_MAX + 1) { *v = (int64_t)(0 - u); return true; }
    return false;
  }
};

// ---------------------------------------------------------------------------
// base64 (stdlib base64.erl): decode/1 skips whitespace (\t \n \r space), '='
// padding terminates; any other byte -> error (function_clause/badarg).
// ---------------------------------------------------------------------------
inline bool base64_decode(const std::vector<uint8_t>& in, std::vector<uint8_t>* out) {
  auto val = [](int c) -> int {
    if (c >= 'A' && c <= 'Z') return c - 'A';
    if (c >= 'a' && c <= 'z') return c - 'a' + 26;
    if (c >= '0' && c <= '9') return c - '0' + 52;
    if (c == '+') return 62;
    if (c == '/') return 63;
    return -1;
  };
  auto ws = [](int c) { return c == '\t' || c == '\n' || c == '\r' || c == ' '; };
  // Restatement of base64:decode_list/2 .. (stdlib >= R13): groups of 4 sextets,
  // "xx==" and "xxx=" tails; whitespace is skipped between any characters;
  // after the '=' tail only whitespace may follow.
  out->clear();
  std::vector<int> q; size_t i = 0, n = in.size();
  while (true) {
    // collect up to 4 symbols
    q.clear();
    while (i < n && q.size() < 4) {
      int c = in[i];
      if (ws(c)) { i++; continue; }
      if (c == '=') break;
      int v = val(c); if (v < 0) return false;
      q.push_back(v); i++;
    }
    if (q.size() == 4) {
      out->push_back((uint8_t)((q[0] << 2) | (q[1] >> 4)));
      out->push_back((uint8_t)((q[1] << 4) | (q[2] >> 2)));
      out->push_back((uint8_t)((q[2] << 6) | q[3]));
      continue;
    }
    if (i >= n) { return q.empty(); }  // input exhausted mid-quantum -> error unless clean
    // in[i] == '='
    if (q.size() == 2) {
      i++;  // first '='
      while (i < n && ws(in[i])) i++;
      if (i >= n || in[i] != '=') return false;
      i++;
      out->push_back((uint8_t)((q[0] << 2) | (q[1] >> 4)));
    } else if (q.size() == 3) {
      i++;
      out->push_back((uint8_t)((q[0] << 2) | (q[1] >> 4)));
      out->push_back((uint8_t)((q[1] << 4) | (q[2] >> 2)));
    } else return false;
    while (i < n) { if (!ws(in[i])) return false; i++; }
    return true;
  }
}
inline std::vector<uint8_t> base64_encode(const std::vector<uint8_t>& in) {
  static const char* T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
  std::vector<uint8_t> o; size_t i = 0;
  for (; i + 3 <= in.size(); i += 3) {
    uint32_t v = (in[i] << 16) | (in[i + 1] << 8) | in[i + 2];
    o.push_back(T[v >> 18]); o.push_back(T[(v >> 12) & 63]); o.push_back(T[(v >> 6) & 63]); o.push_back(T[v & 63]);
  }
  if (in.size() - i == 1) { uint32_t v = in[i] << 16; o.push_back(T[v >> 18]); o.push_back(T[(v >> 12) & 63]); o.push_back('='); o.push_back('='); }
  else if (in.size() - i == 2) { uint32_t v = (in[i] << 16) | (in[i + 1] << 8); o.push_back(T[v >> 18]); o.push_back(T[(v >> 12) & 63]); o.push_back(T[(v >> 6) & 63]); o.push_back('='); }
  return o;
}

}  // namespace otp
