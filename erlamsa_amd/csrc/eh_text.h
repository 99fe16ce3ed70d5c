// eh_text.h — device code for the line mutators (erlamsa_mutations.erl:320-378 +
// erlamsa_generic.erl) and the textual-number mutator sed_num (erlamsa_mutations.erl:63-169).
//
// Scans (newlines, digit runs) are wave-parallel: each lane classifies a 16-byte chunk into a
// bit mask, lanes are combined with a shuffle prefix sum ("ballot/popcount segment scan"), and a
// k-th-match query walks 1 KiB tiles.  The decimal bignum arithmetic of sed_num runs on lane 0.
#pragma once
#include "eh_device.h"

namespace eh {

// ---------------------------------------------------------------------------------------------
// generic tile scan: pred(byte, prev_byte) with prev_byte = 256 at offset 0
// ---------------------------------------------------------------------------------------------
template <class Pred>
EH_DEV uint32_t tile_mask(cbptr p, uint32_t n, uint32_t tile_base, Pred pred) {
  uint32_t i0 = tile_base + 16u * (uint32_t)EH_LANE;
  if (i0 >= n) return 0;
  uint32_t cnt = n - i0 < 16 ? n - i0 : 16;
  uint8_t b[16];
  if (cnt == 16) { uint4 v = ldg16(p + i0); __builtin_memcpy(b, &v, 16); }
  else { for (uint32_t k = 0; k < 16; k++) b[k] = k < cnt ? p[i0 + k] : 0; }
  uint32_t prev = i0 > 0 ? p[i0 - 1] : 256u;
  uint32_t m = 0;
#pragma unroll
  for (uint32_t k = 0; k < 16; k++) {
    if (k < cnt && pred((uint32_t)b[k], prev)) m |= 1u << k;
    prev = b[k];
  }
  return m;
}
EH_DEV uint32_t wave_incl_scan(uint32_t v) {
  const int l = EH_LANE;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { uint32_t t = __shfl_up(v, d); if (l >= d) v += t; }
  return v;
}
EH_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
  return uni(v);
}
template <class Pred>
EH_DEV uint32_t wave_count(cbptr p, uint32_t n, Pred pred) {
  uint32_t c = 0;
  for (uint32_t tb = 0; tb < n; tb += 1024) c += __popc(tile_mask(p, n, tb, pred));
  return wave_sum(c);
}
// position of the k-th (0-based) match, or n if there are fewer
template <class Pred>
EH_DEV uint32_t wave_find_kth(cbptr p, uint32_t n, uint32_t k, Pred pred) {
  uint32_t before = 0;
  for (uint32_t tb = 0; tb < n; tb += 1024) {
    uint32_t m = tile_mask(p, n, tb, pred);
    uint32_t inc = wave_incl_scan(__popc(m));
    uint32_t tile_total = uni((uint32_t)__shfl(inc, 63));
    if (k < before + tile_total) {
      uint32_t kk = k - before;
      bool mine = inc > kk && inc - __popc(m) <= kk;
      unsigned long long who = __ballot(mine);
      int src = (int)__builtin_ctzll(who);
      uint32_t pos = 0;
      if (mine) {
        uint32_t r = kk - (inc - __popc(m));  // r-th set bit of m
        uint32_t mm = m;
        for (uint32_t t = 0; t < r; t++) mm &= mm - 1;
        pos = tb + 16u * (uint32_t)EH_LANE + (uint32_t)__builtin_ctz(mm);
      }
      return uni((uint32_t)__shfl(pos, src));
    }
    before += tile_total;
  }
  return n;
}
// writes the positions of matches with rank in [r0, r1) to out[rank - r0]
template <class Pred>
EH_DEV void wave_collect(cbptr p, uint32_t n, uint32_t r0, uint32_t r1, wptr out, Pred pred) {
  uint32_t before = 0;
  for (uint32_t tb = 0; tb < n && before < r1; tb += 1024) {
    uint32_t m = tile_mask(p, n, tb, pred);
    uint32_t inc = wave_incl_scan(__popc(m));
    uint32_t rank = before + inc - __popc(m);
    uint32_t mm = m;
    while (mm) {
      uint32_t bit = (uint32_t)__builtin_ctz(mm); mm &= mm - 1;
      if (rank >= r0 && rank < r1) out[rank - r0] = tb + 16u * (uint32_t)EH_LANE + bit;
      rank++;
    }
    before += uni((uint32_t)__shfl(inc, 63));
  }
  wave_sync();
}

// erlamsa_utils:binarish/1 (erlamsa_utils.erl:238-247): one load of the first 11 bytes, the
// clause order is then replayed on registers.
EH_DEV bool binarish(cbptr p, uint32_t n) {
  const int l = EH_LANE;
  uint32_t mine = (uint32_t)l < n && l < 11 ? p[l] : 0;
  uint32_t b[11];
#pragma unroll
  for (int k = 0; k < 11; k++) b[k] = (uint32_t)__builtin_amdgcn_readlane((int)mine, k);
#pragma unroll
  for (uint32_t pos = 0; pos <= 8; pos++) {
    uint32_t rem = n > pos ? n - pos : 0;
    if (rem >= 3 && b[pos] == 0xEF && b[pos + 1] == 0xBB && b[pos + 2] == 0xBF) return false;
    if (rem >= 2 && b[pos] == 0xFE && b[pos + 1] == 0x0F) return false;
    if (pos == 8) return false;
    if (rem == 0) return false;
    if (b[pos] == 0) return true;
    if (b[pos] & 128) return true;
  }
  return false;
}

// ---------------------------------------------------------------------------------------------
// small piece lists -> new block in the work area
// ---------------------------------------------------------------------------------------------
struct Pieces {
  uint64_t p[6]; uint32_t n[6]; uint32_t rep[6]; int k;
};
EH_DEV void pc_init(Pieces& q) { q.k = 0; }
EH_DEV void pc_add(Pieces& q, cbptr p, uint32_t n, uint32_t rep = 1) { q.p[q.k] = (uint64_t)p; q.n[q.k] = n; q.rep[q.k] = rep; q.k++; }
EH_DEV bool pc_emit(Ctx& c, const Pieces& q) {
  uint64_t total = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) if (i < q.k) total += (uint64_t)q.n[i] * q.rep[i];
  if (total > 0xFFFFFFF0ull) { EH_SET_OVERFLOW(c, 701); return false; }
  bptr dst = ws_alloc(c, total);
  if (!dst) return false;
  uint64_t pos = 0;
#pragma unroll
  for (int i = 0; i < 6; i++) if (i < q.k) {
    if (q.rep[i] == 1) wave_copy(dst + pos, (cbptr)q.p[i], q.n[i]);
    else wave_fill_periodic(dst + pos, (cbptr)q.p[i], q.n[i], (uint64_t)q.n[i] * q.rep[i]);
    pos += (uint64_t)q.n[i] * q.rep[i];
  }
  wave_sync();
  c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)total;
  return true;
}

// ---------------------------------------------------------------------------------------------
// Lines
// ---------------------------------------------------------------------------------------------
struct IsNl { EH_DEV bool operator()(uint32_t b, uint32_t) const { return b == 10; } };

struct LineIdx {  // lines(Bvec): cut after each \n  (erlamsa_mutations.erl:326-331)
  cbptr p; uint32_t L, nnl, N;
};
EH_DEV void li_init(LineIdx& li, cbptr p, uint32_t L) {
  li.p = p; li.L = L;
  li.nnl = wave_count(p, L, IsNl());
  li.N = li.nnl + ((L > 0 && uni(p[L - 1]) != 10) ? 1u : 0u);
}
// start offset of line j (1-based); j == N+1 gives L
EH_DEV uint32_t li_start(const LineIdx& li, uint32_t j) {
  if (j <= 1) return 0;
  if (j > li.N) return li.L;
  return wave_find_kth(li.p, li.L, j - 2, IsNl()) + 1;
}

// per-mutator state of lis/lrs: [Count | Lines], each stored line = optional nested line + plain tail
struct StLineRef { uint64_t nptr; uint64_t pptr; uint32_t nlen; uint32_t plen; uint32_t has_nested; uint32_t pad; };
struct StState { int32_t count; int32_t pad[3]; StLineRef ln[10]; };
static_assert(sizeof(StState) == 4 * ST_STATE_WORDS, "ST_STATE_WORDS");

__device__ __noinline__ int muta_line(Ctx&, int fn) {
  EH_CTX;                       // construct_line_muta :351-362
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  c.r_kind = R_SAME;
  if (L == 0 || binarish(H, L)) return -1;                    // try_lines :341-348
  LineIdx li; li_init(li, H, L);
  uint32_t N = li.N;
  Pieces q; pc_init(q);
  switch (fn) {
    case M_LD: {                                              // list_del erlamsa_generic.erl:54-57
      uint32_t P = rng_erand(c.rng, N);
      uint32_t a = li_start(li, P), b = li_start(li, P + 1);
      pc_add(q, H, a); pc_add(q, H + b, L - b); break;
    }
    case M_LDS: {                                             // list_del_seq :61-66
      uint32_t S = rng_erand(c.rng, N);
      uint32_t n = rng_erand(c.rng, N - S + 1);
      uint32_t a = li_start(li, S), b = li_start(li, S + n);
      pc_add(q, H, a); pc_add(q, H + b, L - b); break;
    }
    case M_LR2: {                                             // list_dup :70-73
      uint32_t P = rng_erand(c.rng, N);
      uint32_t a = li_start(li, P), b = li_start(li, P + 1);
      pc_add(q, H, b); pc_add(q, H + a, b - a); pc_add(q, H + b, L - b); break;
    }
    case M_LR: {                                              // list_repeat :77-82
      uint32_t P = rng_erand(c.rng, N);
      uint32_t n = rng_log(c.rng, 10); if (n < 2) n = 2;
      uint32_t a = li_start(li, P), b = li_start(li, P + 1);
      pc_add(q, H, a); pc_add(q, H + a, b - a, n); pc_add(q, H + b, L - b); break;
    }
    case M_LRI: {                                             // list_clone :86-91
      uint32_t F = rng_erand(c.rng, N), T = rng_erand(c.rng, N);
      uint32_t fa = li_start(li, F), fb = li_start(li, F + 1), ta = li_start(li, T), tb = li_start(li, T + 1);
      pc_add(q, H, ta); pc_add(q, H + fa, fb - fa); pc_add(q, H + tb, L - tb); break;
    }
    case M_LS: {                                              // list_swap :95-100
      if (N < 2) return 1;
      uint32_t P = rng_erand(c.rng, N - 1);
      uint32_t a = li_start(li, P), b = li_start(li, P + 1), e = li_start(li, P + 2);
      pc_add(q, H, a); pc_add(q, H + b, e - b); pc_add(q, H + a, b - a); pc_add(q, H + e, L - e); break;
    }
    case M_LP: {                                              // list_perm :105-116
      if (N < 3) return 1;
      uint32_t F = rng_erand(c.rng, N - 1);
      uint32_t A = rng_range(c.rng, 2, (int64_t)N - F);
      uint32_t B = rng_log(c.rng, 10);
      uint32_t n = A < B ? A : B; if (n < 2) n = 2;
      // line boundaries F .. F+n  (n+1 offsets)
      uint64_t mark = c.ws_used;
      wptr bnd = (wptr)ws_alloc(c, (uint64_t)(n + 1) * 4);
      EH_G Key2* keys = (EH_G Key2*)ws_alloc(c, 512 * sizeof(Key2));
      wptr order = (wptr)ws_alloc(c, 512 * 4);
      if (!bnd || !keys || !order) return 1;
      // newline ranks F-2 .. F+n-2 give the starts of lines F .. F+n (start = pos+1); line 1 starts at 0
      const int l = EH_LANE;
      if (F == 1) { if (l == 0) bnd[0] = 0; wave_collect(H, L, 0, n, bnd + 1, IsNl()); }
      else wave_collect(H, L, F - 2, F - 2 + n + 1, bnd, IsNl());
      // convert newline positions to line starts, fix the end when the last line has no '\n'
      for (uint32_t i = l; i <= n; i += 64) {
        bool is_start0 = (F == 1 && i == 0);
        uint32_t rank_nl = F - 2 + i;  // newline rank whose pos+1 is this boundary
        if (!is_start0) { if (rank_nl < li.nnl) bnd[i] = bnd[i] + 1; else bnd[i] = L; }
      }
      wave_sync();
      if (n == 2) {                                           // random_permutation([A,B]) erlamsa_rnd.erl:190-194
        uint32_t sw = rng_rand(c.rng, 2);
        if (l == 0) { order[0] = sw == 1 ? 1 : 0; order[1] = sw == 1 ? 0 : 1; }
        wave_sync();
      } else {
        for (uint32_t base = 0; base < 512; base += 64) {
          uint32_t idx = base + l;
          if (idx < n) { double u = rng_peek(c.rng, (uint32_t)l + 1); keys[idx].hi = (uint64_t)__double_as_longlong(u); keys[idx].lo = idx; }
          else { keys[idx].hi = ~(uint64_t)0; keys[idx].lo = idx; }
          if (base < n) rng_skip(c.rng, n - base < 64 ? n - base : 64);
        }
        uint32_t np2 = 4; while (np2 < n) np2 <<= 1;
        wave_sort_key2(keys, np2);
        // ties on the float key fall back to comparing the lines (term order) — fix up sequentially
        if (l == 0) {
          for (uint32_t i = 1; i < n; i++) {
            uint32_t j = i;
            while (j > 0 && keys[j - 1].hi == keys[j].hi) {
              uint32_t x = (uint32_t)keys[j - 1].lo, y = (uint32_t)keys[j].lo;
              uint32_t xl = bnd[x + 1] - bnd[x], yl = bnd[y + 1] - bnd[y];
              int cmp = 0;
              for (uint32_t t = 0; t < xl && t < yl && cmp == 0; t++) { int d = (int)H[bnd[x] + t] - (int)H[bnd[y] + t]; cmp = d; }
              if (cmp == 0) cmp = xl < yl ? -1 : (xl > yl ? 1 : 0);
              if (cmp <= 0) break;
              Key2 tmp = keys[j - 1]; keys[j - 1] = keys[j]; keys[j] = tmp; j--;
            }
          }
        }
        wave_sync();
        for (uint32_t i = l; i < n; i += 64) order[i] = (uint32_t)keys[i].lo;
        wave_sync();
      }
      uint32_t a = uni(bnd[0]), e = uni(bnd[n]);
      bptr dst = ws_alloc(c, L);
      if (!dst) return 1;
      wave_copy(dst, H, a);
      uint32_t pos = a;
      for (uint32_t i = 0; i < n; i++) {
        uint32_t x = uni(order[i]);
        uint32_t s = uni(bnd[x]), t = uni(bnd[x + 1]);
        wave_copy(dst + pos, H + s, t - s); pos += t - s;
      }
      wave_copy(dst + pos, H + e, L - e);
      wave_sync();
      // compact: move dst down over the temporaries (keeps the linear allocator tidy)
      (void)mark;
      c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = L;
      return 1;
    }
  }
  pc_emit(c, q);
  return 1;
}

__device__ __noinline__ int muta_st_line(Ctx&, int fn, EH_G StState* st) {
  EH_CTX;        // construct_st_line_muta :366-378
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  c.r_kind = R_SAME;
  if (L == 0 || binarish(H, L)) return -1;
  LineIdx li; li_init(li, H, L);
  uint32_t N = li.N;
  const int l = EH_LANE;
  int count = (int)uni((uint32_t)st->count);
  // step_state/3 erlamsa_generic.erl:123-139
  if (count < 10) {
    while (count < 10) {
      uint32_t P = rng_erand(c.rng, N);
      uint32_t a = li_start(li, P), b = li_start(li, P + 1);
      wave_sync();
      if (l == 0) {
        for (int i = count; i > 0; i--) st->ln[i] = st->ln[i - 1];   // prepend
        st->ln[0].nptr = 0; st->ln[0].nlen = 0; st->ln[0].has_nested = 0; st->ln[0].pptr = (uint64_t)(H + a); st->ln[0].plen = b - a;
      }
      count++;
      wave_sync();
    }
    if (l == 0) st->count = count;
  }
  {                                                                   // clause 1 recurses into clause 2: also on the filling call
    uint32_t up = rng_erand(c.rng, 20);
    if (up < 10) {
      uint32_t ep = rng_erand(c.rng, N);
      uint32_t a = li_start(li, ep), b = li_start(li, ep + 1);
      EH_G StLineRef* e = &st->ln[up - 1];
      uint32_t hn = uni(e->has_nested), pl = uni(e->plen);
      if (!hn && pl == 0) { c.status = CASE_CRASHED; return 0; }       // fun([_|T], R) on []
      if (l == 0) {
        if (!hn) { e->pptr += 1; e->plen -= 1; }                      // drop the first item
        e->nptr = (uint64_t)(H + a); e->nlen = b - a; e->has_nested = 1;
      }
    }
  }
  wave_sync();
  uint32_t pk = rng_erand(c.rng, (uint32_t)count);                    // pick_state :141-143
  StLineRef x = st->ln[pk - 1];
  uint64_t xn = uni64(x.nptr), xp = uni64(x.pptr); uint32_t xnl = uni(x.has_nested) ? uni(x.nlen) : 0, xpl = uni(x.plen);
  uint32_t P = rng_erand(c.rng, N);                                   // st_list_mod :146-152
  uint32_t a = li_start(li, P), b = li_start(li, P + 1);
  Pieces q; pc_init(q);
  pc_add(q, H, a);
  pc_add(q, (cbptr)xn, xnl); pc_add(q, (cbptr)xp, xpl);
  if (fn == M_LIS) pc_add(q, H + a, L - a); else pc_add(q, H + b, L - b);   // [X, T | R]  /  [X | R]
  pc_emit(c, q);
  return 1;
}

// ---------------------------------------------------------------------------------------------
// sed_num
// ---------------------------------------------------------------------------------------------
struct IsDigitStart { EH_DEV bool operator()(uint32_t b, uint32_t prev) const { return b >= 48 && b <= 57 && !(prev >= 48 && prev <= 57); } };

// Decimal bignum, base 1e9, little endian, lane-0 only.
struct BD { uint32_t* d; int n; bool neg; };   // (digits in work memory or in a caller's array: generic)
__device__ inline void bd_trim(BD& a) { while (a.n > 0 && a.d[a.n - 1] == 0) a.n--; if (a.n == 0) a.neg = false; }
__device__ inline void bd_from_text(BD& r, cbptr t, uint32_t nd) {
  r.n = (int)((nd + 8) / 9); r.neg = false;
  for (int i = 0; i < r.n; i++) {
    uint32_t hi = nd - 9u * (uint32_t)i, lo = hi >= 9 ? hi - 9 : 0, v = 0;
    for (uint32_t k = lo; k < hi; k++) v = v * 10 + (t[k] - 48);
    r.d[i] = v;
  }
  bd_trim(r);
}
__device__ inline void bd_from_u128(BD& r, unsigned __int128 v) {
  r.n = 0; r.neg = false;
  while (v) { r.d[r.n++] = (uint32_t)(v % 1000000000u); v /= 1000000000u; }
}
__device__ inline int bd_cmp_abs(const BD& a, const BD& b) {
  if (a.n != b.n) return a.n < b.n ? -1 : 1;
  for (int i = a.n - 1; i >= 0; i--) if (a.d[i] != b.d[i]) return a.d[i] < b.d[i] ? -1 : 1;
  return 0;
}
__device__ inline void bd_add_abs(BD& r, const BD& a, const BD& b) {  // r.d has room for max(n)+1
  uint32_t carry = 0; int n = a.n > b.n ? a.n : b.n;
  for (int i = 0; i < n; i++) {
    uint32_t s = (i < a.n ? a.d[i] : 0) + (i < b.n ? b.d[i] : 0) + carry;
    carry = s >= 1000000000u; r.d[i] = carry ? s - 1000000000u : s;
  }
  r.n = n; if (carry) r.d[r.n++] = 1;
}
__device__ inline void bd_sub_abs(BD& r, const BD& a, const BD& b) {  // |a| >= |b|
  int borrow = 0;
  for (int i = 0; i < a.n; i++) {
    int64_t s = (int64_t)a.d[i] - (i < b.n ? b.d[i] : 0) - borrow;
    borrow = s < 0; r.d[i] = (uint32_t)(borrow ? s + 1000000000 : s);
  }
  r.n = a.n; bd_trim(r);
}
// r = a + (bneg ? -|b| : |b|) with a signed
__device__ inline void bd_add_signed(BD& r, const BD& a, const BD& b, bool bneg) {
  if (a.neg == bneg) { bd_add_abs(r, a, b); r.neg = a.neg; }
  else {
    int cm = bd_cmp_abs(a, b);
    if (cm == 0) { r.n = 0; r.neg = false; }
    else if (cm > 0) { bd_sub_abs(r, a, b); r.neg = a.neg; }
    else { bd_sub_abs(r, b, a); r.neg = bneg; }
  }
  bd_trim(r);
}
__device__ inline uint32_t bd_to_text(const BD& a, bptr out) {      // integer_to_list/1
  uint32_t pos = 0;
  if (a.n == 0) { out[0] = '0'; return 1; }
  if (a.neg) out[pos++] = '-';
  uint32_t top = a.d[a.n - 1];
  uint32_t div = 1; while (top / div >= 10) div *= 10;
  while (div) { out[pos++] = (uint8_t)('0' + (top / div) % 10); div /= 10; }
  for (int i = a.n - 2; i >= 0; i--) {
    uint32_t v = a.d[i]; uint32_t dv = 100000000u;
    for (int k = 0; k < 9; k++) { out[pos++] = (uint8_t)('0' + v / dv); v %= dv; dv /= 10; }
  }
  return pos;
}
// |a|*2 as base-2^64 digits (little endian); returns number of digits (<= cap) or -1 if it does not fit
__device__ inline int bd_times2_to_bin(const BD& a, qptr w, int cap) {
  int n = 0;
  for (int i = a.n - 1; i >= 0; i--) {
    unsigned __int128 carry = a.d[i];
    for (int k = 0; k < n; k++) { unsigned __int128 t = (unsigned __int128)w[k] * 1000000000u + carry; w[k] = (uint64_t)t; carry = t >> 64; }
    if (carry) { if (n >= cap) return -1; w[n++] = (uint64_t)carry; }
  }
  uint64_t c = 0;
  for (int k = 0; k < n; k++) { uint64_t nv = (w[k] << 1) | c; c = w[k] >> 63; w[k] = nv; }
  if (c) { if (n >= cap) return -1; w[n++] = c; }
  return n;
}
// base-2^64 digits -> BD (destroys w)
__device__ inline void bd_from_bin(BD& r, qptr w, int n) {
  r.n = 0; r.neg = false;
  while (n > 0) {
    unsigned __int128 rem = 0;
    for (int k = n - 1; k >= 0; k--) { unsigned __int128 cur = (rem << 64) | w[k]; w[k] = (uint64_t)(cur / 1000000000u); rem = cur % 1000000000u; }
    r.d[r.n++] = (uint32_t)rem;
    while (n > 0 && w[n - 1] == 0) n--;
  }
  bd_trim(r);
}

// interesting_numbers/0 (erlamsa_mutations.erl:68-75): list index -> (exponent, -1/0/+1)
__device__ inline void bd_interesting(BD& r, uint32_t idx, wptr t1) {
  const int is[11] = {128, 127, 64, 63, 32, 31, 16, 15, 8, 7, 1};
  int e = 1; for (int k = 0; k < 11; k++) if ((int)(idx / 3) == k) e = is[k];
  int which = (int)(idx % 3);                                  // X-1, X, X+1
  if (e < 128) { unsigned __int128 x = (unsigned __int128)1 << e; bd_from_u128(r, which == 0 ? x - 1 : (which == 1 ? x : x + 1)); return; }
  BD m{t1, 0, false}; bd_from_u128(m, ~(unsigned __int128)0);   // 2^128 - 1
  if (which == 0) { for (int i = 0; i < m.n; i++) r.d[i] = m.d[i]; r.n = m.n; r.neg = false; return; }
  uint32_t small[1]; BD one{small, 0, false}; bd_from_u128(one, (unsigned __int128)which);   // +1 -> 2^128, +2 -> 2^128+1
  bd_add_abs(r, m, one); r.neg = false;
}

// mutate_num/1,2 (erlamsa_mutations.erl:93-112) on the decimal number whose digits are H[s, e) (sign given
// separately): draws first, then the arithmetic; the decimal text of the result (integer_to_list/1) is left
// in *txt / *tlen (work-area memory).  Returns false after setting c.status (crash: float overflow in rand/1).
__device__ __noinline__ bool num_core(Ctx&, cbptr H, uint32_t L, uint32_t s, uint32_t e, bool negsign, bptr* txt_out, uint32_t* tlen_out) {
  EH_CTX;
  const int l = EH_LANE;
  uint32_t nd = e - s;
  bool fast_parse = false; uint64_t mag = 0;
  if (nd <= 18) {
    fast_parse = true;
    uint32_t idx = s + (uint32_t)l;
    uint32_t ch = (uint32_t)l < nd && idx < L ? H[idx] : 48;
    uint64_t part = 0;
    if ((uint32_t)l < nd) { uint64_t pw = 1; for (uint32_t k = (uint32_t)l + 1; k < nd; k++) pw *= 10; part = (uint64_t)(ch - 48) * pw; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) part += ((uint64_t)(uint32_t)__shfl_xor((int)(part >> 32), d) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)part, d);
    mag = uni64(part);
  }
  // ---- mutate_num/2 :93-112 : draws first (uniform), arithmetic afterwards
  uint32_t op = rng_rand(c.rng, 12);
  uint32_t ielem = 0, rl_n = 0, rl_s = 0; double u9 = 0.0; double u_hi = 0.0; uint32_t rl_k = 0;
  bool is_zero_num = false;
  if (fast_parse) is_zero_num = mag == 0;
  else {  // is the parsed value zero? (needed to know whether case 9 draws)
    uint32_t nz = 0;
    if (l == 0) { for (uint32_t k = s; k < e; k++) if (H[k] != 48) { nz = 1; break; } }
    is_zero_num = uni(nz) == 0;
  }
  if (op == 4 || op == 5 || op == 7 || op == 8) ielem = rng_rand(c.rng, 33);
  else if (op == 9) { if (!is_zero_num) u9 = rng_uniform(c.rng); }
  else if (op == 6 || op == 11) {
    rl_n = rng_range(c.rng, 1, 129);
    rl_k = rng_rand(c.rng, rl_n);                              // rand_log: rand_nbit(rand(N))
    if (rl_k > 0) u_hi = rng_uniform(c.rng);                   // rand(Hi), Hi = 2^(k-1)
    rl_s = rng_rand(c.rng, 3);
  }
  // ---- register fast path: |Num| < 10^18 and an operand below 2^120: signed 128-bit arithmetic,
  // digits produced one per lane.
  bool fast_op = fast_parse && !((op == 4 || op == 5 || op == 7 || op == 8) && ielem < 6) && !((op == 6 || op == 11) && rl_k > 120);
  if (fast_op) {
    __int128 v = negsign ? -(__int128)mag : (__int128)mag;
    __int128 r = 0;
    auto inum = [](uint32_t idx) -> __int128 {                 // interesting_numbers/0, idx >= 6
      const int is[11] = {128, 127, 64, 63, 32, 31, 16, 15, 8, 7, 1};
      int ex = 1;
#pragma unroll
      for (int k = 2; k < 11; k++) if ((int)(idx / 3) == k) ex = is[k];
      __int128 x = (__int128)1 << ex; int w = (int)(idx % 3);
      return w == 0 ? x - 1 : (w == 1 ? x : x + 1);
    };
    switch (op) {
      case 0: r = v + 1; break;
      case 1: r = v - 1; break;
      case 2: r = 0; break;
      case 3: r = 1; break;
      case 4: case 5: r = inum(ielem); break;
      case 7: r = v + inum(ielem); break;
      case 8: r = v - inum(ielem); break;
      case 9: {
        if (mag == 0) { r = 0; break; }
        double f = (double)(mag * 2);
        double x = u9 * f;
        uint64_t rr = (uint64_t)x;
        r = negsign ? v + (__int128)rr : v - (__int128)rr;
        break;
      }
      case 10: r = -v; break;
      default: {
        unsigned __int128 lv = 0;
        if (rl_k > 0) {
          unsigned __int128 hi = (unsigned __int128)1 << (rl_k - 1), rv = 0;
          if (u_hi > 0.0) {
            int ex; double m = frexp(u_hi, &ex); uint64_t mant = (uint64_t)ldexp(m, 53);
            int sh = ex - 53 + (int)(rl_k - 1);
            rv = sh >= 0 ? ((unsigned __int128)mant << sh) : (sh > -64 ? (unsigned __int128)(mant >> (-sh)) : 0);
          }
          lv = hi | rv;
        }
        r = rl_s == 0 ? v - (__int128)lv : v + (__int128)lv;
        break;
      }
    }
    bool rneg = r < 0;
    unsigned __int128 m = rneg ? (unsigned __int128)(-r) : (unsigned __int128)r;
    const unsigned __int128 P19 = (unsigned __int128)10000000000000000000ull;
    uint64_t hi64 = (uint64_t)(m / P19), lo64 = (uint64_t)(m % P19);
    hi64 = uni64(hi64); lo64 = uni64(lo64);
    uint32_t dig = 0;
    if (l < 39) {
      uint64_t src = l < 19 ? lo64 : hi64; uint32_t j = l < 19 ? (uint32_t)l : (uint32_t)l - 19;
      uint64_t pw = 1; for (uint32_t k = 0; k < j; k++) pw *= 10;
      dig = (uint32_t)((src / pw) % 10);
    }
    unsigned long long nzm = __ballot(dig != 0);
    uint32_t ndig = nzm == 0 ? 1u : 64u - (uint32_t)__builtin_clzll(nzm);
    uint32_t sg = rneg ? 1u : 0u;
    uint32_t tlen = sg + ndig;
    bptr t = ws_alloc(c, 48);
    if (!t) return false;
    if ((uint32_t)l < ndig) t[sg + (ndig - 1 - (uint32_t)l)] = (uint8_t)(48 + dig);
    if (rneg && l == 63) t[0] = 45;
    wave_sync();
    *txt_out = t; *tlen_out = tlen;
    return true;
  }
  // work arrays (lane 0): limbs for value, operand, result
  uint32_t nl = (nd + 8) / 9 + 8;
  wptr va = (wptr)ws_alloc(c, (uint64_t)nl * 4 + 64);
  wptr vb = (wptr)ws_alloc(c, (uint64_t)nl * 4 + 256);
  wptr vr = (wptr)ws_alloc(c, (uint64_t)nl * 4 + 256);
  qptr wb = (qptr)ws_alloc(c, 20 * 8);
  bptr txt = ws_alloc(c, (uint64_t)nl * 9 + 64);
  if (!va || !vb || !vr || !wb || !txt) return false;
  uint32_t tlen = 0, crashed = 0;
  if (l == 0) {
    BD num{va, 0, false}, opd{vb, 0, false}, res{vr, 0, false};
    bd_from_text(num, H + s, nd);
    if (negsign && num.n > 0) num.neg = true;
    switch (op) {
      case 0: bd_from_u128(opd, 1); bd_add_signed(res, num, opd, false); break;
      case 1: bd_from_u128(opd, 1); bd_add_signed(res, num, opd, true); break;
      case 2: res.n = 0; break;
      case 3: bd_from_u128(res, 1); break;
      case 4: case 5: case 7: case 8: {
        bd_interesting(opd, ielem, (wptr)wb);
        if (op == 4 || op == 5) { res.d = opd.d; res.n = opd.n; res.neg = false; }
        else bd_add_signed(res, num, opd, op == 8);
        break;
      }
      case 9: {                                                // Num - rand(abs(Num)*2) * sign(Num)
        if (num.n == 0) { res.n = 0; break; }
        if (num.n > 39) { crashed = 1; break; }                // >= 10^342: float overflow -> badarith
        int nw = bd_times2_to_bin(num, wb, 18);
        if (nw < 0) { crashed = 1; break; }
        double d = 0.0;                                        // erts big_to_double: d = d*2^64 + digit
        for (int k = nw - 1; k >= 0; k--) { d = d * 18446744073709551616.0 + (double)wb[k]; if (isinf(d)) { crashed = 1; break; } }
        if (crashed) break;
        double x = u9 * d;
        x = trunc(x);
        // exact integer of x -> base 2^64 digits
        for (int k = 0; k < 18; k++) wb[k] = 0;
        int nwr = 0;
        if (x >= 1.0) {
          int ex; double m = frexp(x, &ex);                    // x = m * 2^ex
          uint64_t mant = (uint64_t)ldexp(m, 53);
          int sh = ex - 53;
          if (sh <= 0) { wb[0] = mant >> (-sh); nwr = 1; }
          else { int wi = sh / 64, bi = sh % 64; wb[wi] = mant << bi; if (bi) wb[wi + 1] = mant >> (64 - bi); nwr = wi + 2; }
          while (nwr > 0 && wb[nwr - 1] == 0) nwr--;
        }
        bd_from_bin(opd, wb, nwr);
        bd_add_signed(res, num, opd, !num.neg);                // minus R*sign: sign(X>=0)=1
        break;
      }
      case 10: res.d = num.d; res.n = num.n; res.neg = num.n > 0 ? !num.neg : false; break;
      default: {                                               // 6, 11: Num -/+ rand_log(rand_range(1,129))
        unsigned __int128 lv = 0;
        if (rl_k > 0) {
          unsigned __int128 hi = (unsigned __int128)1 << (rl_k - 1);
          // rand(Hi) = trunc(U * 2^(k-1)) : exact scaling of the double U
          unsigned __int128 rv = 0;
          if (u_hi > 0.0) {
            int ex; double m = frexp(u_hi, &ex); uint64_t mant = (uint64_t)ldexp(m, 53);
            int sh = ex - 53 + (int)(rl_k - 1);
            rv = sh >= 0 ? ((unsigned __int128)mant << sh) : (sh > -64 ? (unsigned __int128)(mant >> (-sh)) : 0);
          }
          lv = hi | rv;
        }
        bd_from_u128(opd, lv);
        bd_add_signed(res, num, opd, rl_s == 0);
        break;
      }
    }
    if (!crashed) tlen = bd_to_text(res, txt);
  }
  wave_sync();
  if (uni(crashed)) { c.status = CASE_CRASHED; return false; }
  *txt_out = txt; *tlen_out = uni(tlen);
  return true;
}

__device__ __noinline__ int muta_num(Ctx&) {
  EH_CTX;                                  // sed_num :154-169
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  const int l = EH_LANE;
  c.r_kind = R_SAME;
  // mutate_a_num/2: numbers = maximal digit runs, each extended left over the dashes before it
  uint32_t nfound = wave_count(H, L, IsDigitStart());
  c.m_aux = nfound == 0 ? 0 : 1;                              // [{muta_num, 0 | 1} | Meta] :162-168
  uint32_t which = rng_rand(c.rng, nfound);
  if (nfound == 0) {
    // nothing to change; the data still goes through flush_bvecs (re-chunking counts as a change
    // for blocks >= 2048 bytes, erlamsa_mutations.erl:157 + mux_fuzzers_loop :1278)
    c.r_kind = R_NEW; c.r_ptr = (bptr)H; c.r_len = L; c.r_flush = 1;
    uint32_t r = rng_rand(c.rng, 10);
    return r == 0 ? -1 : 0;
  }
  uint32_t s = wave_find_kth(H, L, nfound - 1 - which, IsDigitStart());
  // end of the digit run / start of the dash run: one 64-byte window each way resolved with ballots;
  // runs longer than the window fall back to a lane-0 walk.
  uint32_t e, a;
  {
    uint32_t idx = s + (uint32_t)l;
    uint32_t ch = idx < L ? H[idx] : 0;
    unsigned long long dm = __ballot(ch >= 48 && ch <= 57);
    uint32_t run = dm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~dm);
    uint32_t ch2 = (uint32_t)l < s ? H[s - 1 - (uint32_t)l] : 0;
    unsigned long long mm = __ballot((uint32_t)l < s && ch2 == 45);
    uint32_t dr = mm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~mm);
    if (run < 64 && dr < 64) { e = s + run; a = s - dr; }
    else {
      uint32_t ee = s, aa = s;
      if (l == 0) {
        while (ee < L && H[ee] >= 48 && H[ee] <= 57) ee++;
        while (aa > 0 && H[aa - 1] == 45) aa--;
      }
      e = uni(ee); a = uni(aa);
    }
  }
  bptr txt; uint32_t tlen;
  if (!num_core(c, H, L, s, e, a < s, &txt, &tlen)) return 0;
  uint32_t nlen = a + tlen + (L - e);
  bptr dst = ws_alloc(c, nlen);
  if (!dst) return 0;
  wave_copy(dst, H, a);
  wave_copy(dst + a, txt, tlen);
  wave_copy(dst + a + tlen, H + e, L - e);
  wave_sync();
  c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = nlen; c.r_flush = 1;
  return binarish(dst, nlen) ? -1 : 2;
}

}  // namespace eh
