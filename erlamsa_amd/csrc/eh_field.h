// eh_field.h — device code for erlamsa_field_predict.erl (length-field and checksum-trailer
// finders) and the `len` mutator (erlamsa_mutations.erl:1107-1143).
//
// get_possible_simple_lens/1 as written enumerates (SubLen+1)^2 + SubLen+1 ranges x 5 offsets x 6
// binary patterns (7.9 M pattern matches for a 4 KiB block).  Here the candidate LIST is never
// materialised: a range {A,B} matches pattern c iff B == A + w_c + field_c(A), so lanes own the
// offsets A (<= 513 of them), the SubLen+1 random right ends are broadcast one by one, and only
// per-offset match COUNTS are kept.  rand_elem's index is then resolved by walking the counts in
// the reference's list order (SmallLens ascending, then BigLens = reversed range order).
#pragma once
#include "eh_tree.h"

namespace eh {

struct SizerElem { uint32_t size_bits; uint32_t big; uint32_t len; uint32_t a; uint32_t b; };

// field value at offset A for clause c (0..5: 16/32/64 big, 16/32/64 little); returns false if the
// binary is too short for the pattern
EH_DEV bool field_at(cbptr H, uint32_t L, uint32_t A, int c, uint64_t* v, uint32_t* w) {
  const uint32_t ws[3] = {2, 4, 8};
  uint32_t wd = ws[c % 3]; *w = wd;
  if (A + wd > L) return false;
  uint64_t x = 0;
  if (c < 3) { for (uint32_t k = 0; k < wd; k++) x = (x << 8) | H[A + k]; }
  else { for (uint32_t k = 0; k < wd; k++) x |= (uint64_t)H[A + k] << (8 * k); }
  *v = x; return true;
}
// basic_len/2 (:66-79): first matching clause for the range {A,B}; returns clause index or -1
EH_DEV int basic_len_clause(const uint64_t fv[6], uint32_t fmask, uint32_t L, uint32_t A, int64_t B) {
  if (!((int64_t)A < B && B > 0 && A < L)) return -1;
  const int64_t ws[6] = {2, 4, 8, 2, 4, 8};
#pragma unroll
  for (int c = 0; c < 6; c++) {
    if (!((fmask >> c) & 1)) continue;
    int64_t want = B - (int64_t)A - ws[c];
    if (want > 2 && fv[c] == (uint64_t)want) return c;
  }
  return -1;
}

// The offset that holds the k-th element when the per-offset counts cnt[0..top] are walked from `top` down (BigLens is built in
// reversed range order); k becomes the rank within that offset.  64 offsets a step.
EH_DEV int64_t sizer_pick_desc(const uint32_t* cnt, uint32_t top, uint32_t& k) {
  const int l = EH_LANE;
  for (int64_t xtop = (int64_t)top; xtop >= 0; xtop -= 64) {
    const int64_t X = xtop - (int64_t)l;
    const uint32_t m = X >= 0 ? cnt[X] : 0u;
    const uint32_t inc = wave_incl_scan(m);
    const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    if (k < all) {
      const int j = (int)__builtin_ctzll(__ballot(inc > k));
      k -= (uint32_t)__builtin_amdgcn_readlane((int)(inc - m), j);
      return xtop - j;
    }
    k -= all;
  }
  return -1;
}

// Picks rand_elem(get_possible_simple_lens(Bin)).  Returns 1 and fills *e, 0 when the list is
// empty (no draw for rand_elem), -1 on allocation failure.  Consumes SubLen+1 draws when L > 10.
__device__ __noinline__ int pick_simple_len(Ctx&, cbptr H, uint32_t L, SizerElem* e) {
  EH_CTX;
  const int l = EH_LANE;
  const int64_t adjs[5] = {0, 1, 2, 4, 8};
  if (L <= 10) {                                                 // :102-105: offsets 0..3, [simple_len, simple_u8len] per offset
    // tiny: evaluate on every lane redundantly, in list order
    uint32_t total = 0; SizerElem found[1];
    // first pass count, second pass pick
    uint32_t pickidx = 0xFFFFFFFFu; bool have = false; SizerElem res{0, 0, 0, 0, 0};
    for (int pass = 0; pass < 2; pass++) {
      uint32_t k = 0;
      for (uint32_t X = 0; X <= 3; X++) {
        uint64_t fv[6]; uint32_t fm = 0;
        for (int cc = 0; cc < 6; cc++) { uint32_t w; uint64_t v = 0; if (field_at(H, L, X, cc, &v, &w)) { fm |= 1u << cc; } fv[cc] = v; }
        for (int a = 0; a < 5; a++) {
          int64_t B = (int64_t)L - adjs[a];
          int cl = basic_len_clause(fv, fm, L, X, B);
          if (cl >= 0) { if (pass == 1 && k == pickidx) { const uint32_t ws[6] = {2, 4, 8, 2, 4, 8}; res = SizerElem{ws[cl] * 8, cl < 3 ? 1u : 0u, (uint32_t)(B - X - ws[cl]), X, (uint32_t)B}; have = true; } k++; }
        }
        for (int x8 = 0; x8 <= 8; x8++) {                        // simple_u8len :60-64
          int64_t B = (int64_t)L - x8;
          if ((int64_t)X < B && B > 0 && X < L) { uint32_t v = H[X]; if ((int64_t)v == B - X - 1 && v > 2) { if (pass == 1 && k == pickidx) { res = SizerElem{8, 1, v, X, (uint32_t)B}; have = true; } k++; } }
        }
      }
      if (pass == 0) { total = uni(k); if (total == 0) return 0; pickidx = rng_rand(c.rng, total); }
    }
    (void)found; (void)have;
    e->size_bits = uni(res.size_bits); e->big = uni(res.big); e->len = uni(res.len); e->a = uni(res.a); e->b = uni(res.b);
    return 1;
  }
  uint32_t sub = L / 5 < SIZER_MAX_FIRST_BYTES ? L / 5 : SIZER_MAX_FIRST_BYTES;
  uint32_t ny = sub + 1;
  // the <= 513 right ends in LDS (g_fuse_lds is free outside the fuse / sgm mutators): in draw order, sorted, and the per-offset
  // counts of parts II and III
  static_assert(4 * (SIZER_MAX_FIRST_BYTES + 8) <= EH_FUSE_LDS_WORDS, "the sizer's tables fit the fuse band");
  uint32_t* varb = g_fuse_lds; uint32_t* vsort = g_fuse_lds + (SIZER_MAX_FIRST_BYTES + 8); uint32_t* cnt2 = g_fuse_lds + 2 * (SIZER_MAX_FIRST_BYTES + 8);
  uint32_t* cnt3 = g_fuse_lds + 3 * (SIZER_MAX_FIRST_BYTES + 8);
  wave_sync();
  // VarBSeq = [rand_range(SubLen, Len) || _ <- FirstSeq]   (:94)
  for (uint32_t base = 0; base < ny; base += 64) {
    uint32_t j = base + (uint32_t)l;
    if (j < ny) varb[j] = sub + (uint32_t)(rng_peek(c.rng, (uint32_t)l + 1) * (double)(L - sub));
    rng_skip(c.rng, ny - base < 64 ? ny - base : 64);
  }
  wave_sync();
  // sorted copy: every lane ranks its own values against all of them (ties by position)
  for (uint32_t base = 0; base < ny; base += 64) {
    uint32_t j = base + (uint32_t)l;
    if (j < ny) {
      const uint32_t v = varb[j]; uint32_t rank = 0;
      for (uint32_t i = 0; i < ny; i++) { const uint32_t u = varb[i]; rank += (u < v || (u == v && i < j)) ? 1u : 0u; }
      vsort[rank] = v;
    }
  }
  wave_sync();
  // lane handles offsets A = l, l+64, ... (<= 9 of them)
  uint32_t tot1 = 0, tot2 = 0, tot3 = 0;
  for (uint32_t base = 0; base < ny; base += 64) {
    uint32_t A = base + (uint32_t)l;
    uint32_t c1 = 0, c2 = 0, c3 = 0;
    if (A < ny) {
      uint64_t fv[6]; uint32_t fm = 0;
      for (int cc = 0; cc < 6; cc++) { uint32_t w; uint64_t v = 0; if (field_at(H, L, A, cc, &v, &w)) fm |= 1u << cc; fv[cc] = v; }
      // I: simple_u8len(A): B = L - X, X in 0..8
      { uint32_t v = H[A]; int64_t B = (int64_t)A + 1 + v; int64_t X = (int64_t)L - B; if (v > 2 && X >= 0 && X <= 8 && A < L) c1 = 1; }
      // III: simple_len({A, L})
      for (int a = 0; a < 5; a++) if (basic_len_clause(fv, fm, L, A, (int64_t)L - adjs[a]) >= 0) c3++;
      // II: simple_len({A, VarB[y] - adj}) for every y and adj: a pair matches when VarB[y] - adj is one of the (at most six, here
      // made distinct) right ends A + w_c + field_c(A) the clauses ask for - counted in the sorted copy
      const uint32_t wsc[6] = {2, 4, 8, 2, 4, 8};
      uint32_t te[6]; uint32_t use = 0;                            // (indexed by unrolled loops only: registers)
#pragma unroll
      for (int cc = 0; cc < 6; cc++) {
        te[cc] = 0;
        if (!((fm >> cc) & 1) || fv[cc] <= 2 || fv[cc] > (uint64_t)L) continue;      // want > 2; B <= VarB < L
        const uint64_t t = (uint64_t)A + wsc[cc] + fv[cc];
        if (t > (uint64_t)L) continue;
        te[cc] = (uint32_t)t;
        bool dup = false;
#pragma unroll
        for (int q = 0; q < 6; q++) if (q < cc && ((use >> q) & 1) && te[q] == (uint32_t)t) dup = true;
        if (!dup) use |= 1u << cc;
      }
#pragma unroll 1
      for (int cc = 0; cc < 6; cc++) {
        if (!((use >> cc) & 1)) continue;
        const uint32_t t = cc == 0 ? te[0] : cc == 1 ? te[1] : cc == 2 ? te[2] : cc == 3 ? te[3] : cc == 4 ? te[4] : te[5];
        uint32_t lo = 0, hi = ny;                                   // first sorted value >= t
        while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (vsort[mid] < t) lo = mid + 1; else hi = mid; }
        for (uint32_t k = lo; k < ny; k++) {
          const uint32_t d = vsort[k] - t;
          if (d > 8) break;
          if (d == 0 || d == 1 || d == 2 || d == 4 || d == 8) c2++;
        }
      }
      cnt2[A] = c2; cnt3[A] = c3;
    }
    tot1 += c1; tot2 += c2; tot3 += c3;
  }
  tot1 = wave_sum(tot1); tot2 = wave_sum(tot2); tot3 = wave_sum(tot3);
  wave_sync();
  uint32_t total = tot1 + tot2 + tot3;
  if (total == 0) return 0;
  uint32_t idx = rng_rand(c.rng, total);                         // rand_elem/1
  SizerElem res{0, 0, 0, 0, 0};
  const uint32_t wsz[6] = {2, 4, 8, 2, 4, 8};
  if (idx < tot1) {
    // idx-th flagged offset, ascending
    uint32_t before = 0;
    for (uint32_t base = 0; base < ny; base += 64) {
      uint32_t A = base + (uint32_t)l; bool f = false; uint32_t v = 0;
      if (A < ny) { v = H[A]; int64_t B = (int64_t)A + 1 + v; int64_t X = (int64_t)L - B; f = v > 2 && X >= 0 && X <= 8; }
      unsigned long long m = __ballot(f);
      uint32_t cntm = (uint32_t)__popcll(m);
      if (idx < before + cntm) {
        uint32_t r = idx - before; unsigned long long mm = m; for (uint32_t t = 0; t < r; t++) mm &= mm - 1;
        int src = (int)__builtin_ctzll(mm);
        uint32_t vv = (uint32_t)__builtin_amdgcn_readlane((int)v, src);
        res = SizerElem{8, 1, vv, base + (uint32_t)src, base + (uint32_t)src + 1 + vv};
        break;
      }
      before += cntm;
    }
  } else if (idx < tot1 + tot2) {
    // BigLens over {X, VarB[y]}: X from SubLen down to 0, y from last to first, then the 5 offsets
    uint32_t k = idx - tot1;
    const int64_t Xs = sizer_pick_desc(cnt2, sub, k);
    uint32_t A = (uint32_t)Xs;
    uint64_t fv[6]; uint32_t fm = 0;
    for (int cc = 0; cc < 6; cc++) { uint32_t w; uint64_t v = 0; if (field_at(H, L, A, cc, &v, &w)) fm |= 1u << cc; fv[cc] = v; }
    // which y: 64 right ends at a time from the last one down, every lane counts the matches of its own (over the five adjustments)
    int64_t ys = -1;
    for (int64_t ytop = (int64_t)ny - 1; ytop >= 0 && ys < 0; ytop -= 64) {
      const int64_t y = ytop - (int64_t)l;
      uint32_t m = 0;
      if (y >= 0) { const int64_t vb = (int64_t)varb[y]; for (int a = 0; a < 5; a++) if (basic_len_clause(fv, fm, L, A, vb - adjs[a]) >= 0) m++; }
      const uint32_t inc = wave_incl_scan(m);
      const uint32_t all = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      if (k < all) {
        const int j = (int)__builtin_ctzll(__ballot(inc > k));
        k -= (uint32_t)__builtin_amdgcn_readlane((int)(inc - m), j);
        ys = ytop - j;
      } else k -= all;
    }
    bool done = false;
    if (ys >= 0) {
      const int64_t vb = (int64_t)uni(varb[ys]);
      for (int a = 0; a < 5 && !done; a++) {
        int cl = basic_len_clause(fv, fm, L, A, vb - adjs[a]);
        if (cl >= 0) { if (k == 0) { uint32_t B = (uint32_t)(vb - adjs[a]); res = SizerElem{wsz[cl] * 8, cl < 3 ? 1u : 0u, B - A - wsz[cl], A, B}; done = true; } else k--; }
      }
    }
  } else {
    uint32_t k = idx - tot1 - tot2;
    const int64_t Xs = sizer_pick_desc(cnt3, sub, k);
    uint32_t A = (uint32_t)Xs;
    uint64_t fv[6]; uint32_t fm = 0;
    for (int cc = 0; cc < 6; cc++) { uint32_t w; uint64_t v = 0; if (field_at(H, L, A, cc, &v, &w)) fm |= 1u << cc; fv[cc] = v; }
    bool done = false;
    for (int a = 0; a < 5 && !done; a++) {
      int cl = basic_len_clause(fv, fm, L, A, (int64_t)L - adjs[a]);
      if (cl >= 0) { if (k == 0) { uint32_t B = (uint32_t)((int64_t)L - adjs[a]); res = SizerElem{wsz[cl] * 8, cl < 3 ? 1u : 0u, B - A - wsz[cl], A, B}; done = true; } else k--; }
    }
  }
  wave_sync();
  e->size_bits = uni(res.size_bits); e->big = uni(res.big); e->len = uni(res.len); e->a = uni(res.a); e->b = uni(res.b);
  return 1;
}

// writes V as a Size-bit big/little endian field (two's-complement truncation) — lane 0
__device__ inline void put_field(bptr o, uint64_t v, uint32_t bits, bool big) {
  uint32_t n = bits / 8;
  for (uint32_t i = 0; i < n; i++) { uint32_t sh = big ? 8 * (n - 1 - i) : 8 * i; o[i] = (uint8_t)(v >> sh); }
}

// length_predict/2 + mutate_length/2 (erlamsa_mutations.erl:1107-1143)
__device__ __noinline__ int muta_len(Ctx&) {
  EH_CTX;
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  const int l = EH_LANE;
  c.r_kind = R_SAME;
  SizerElem e;
  int r = pick_simple_len(c, H, L, &e);
  if (r < 0) return 0;
  if (r == 0) return -2;                                         // mutate_length(Binary, []) :1112
  uint32_t nb = e.size_bits / 8;
  if ((uint64_t)e.a + nb + e.len > L) { c.status = CASE_CRASHED; return 0; }   // extract_blob badmatch
  uint32_t blob0 = e.a + nb, rest0 = blob0 + e.len;
  // <<TmpNewLen:Size>> = random_block(Size/8) ; NewLen = min(1000000, TmpNewLen*2)
  uint64_t tmp = 0; bool huge = false;
  {
    // random_block: first draw is the LAST byte (:173-174)
    uint32_t bytes[8];
#pragma unroll
    for (int k = 0; k < 8; k++) bytes[k] = (uint32_t)k < nb ? rng_rand(c.rng, 256) : 0;
    for (uint32_t k = 0; k < nb; k++) {                           // big-endian value of the block: block[i] = draw[nb-1-i]
      uint32_t bv = 0;
#pragma unroll
      for (int t = 0; t < 8; t++) if ((uint32_t)t == nb - 1 - k) bv = bytes[t];
      if (tmp >> 56) huge = true;
      tmp = (tmp << 8) | bv;
    }
  }
  uint32_t newlen = (huge || tmp >= ABSMAX_BINARY_BLOCK) ? ABSMAX_BINARY_BLOCK : (uint32_t)(tmp * 2 > ABSMAX_BINARY_BLOCK ? ABSMAX_BINARY_BLOCK : tmp * 2);
  uint32_t k = rng_rand(c.rng, 7);
  bptr fld = ws_alloc(c, 16);
  if (!fld) return 0;
  Pieces q; pc_init(q);
  pc_add(q, H, e.a);
  switch (k) {
    case 0: if (l == 0) put_field(fld, 0, e.size_bits, true); wave_sync(); pc_add(q, fld, nb); pc_add(q, H + blob0, L - blob0); break;
    case 1: if (l == 0) put_field(fld, ~(uint64_t)0, e.size_bits, true); wave_sync(); pc_add(q, fld, nb); pc_add(q, H + blob0, L - blob0); break;
    case 2: {
      // fast_pseudorandom_block(NewLen) (erlamsa_rnd.erl:155-160)
      bptr rnd; uint32_t rlen;
      if (newlen < ABSMAXHALF_BINARY_BLOCK) { rlen = newlen; rnd = ws_alloc(c, rlen); if (!rnd) return 0; random_block_rev(c, rnd, rlen); }
      else {
        uint32_t z = newlen - ABSMAXHALF_BINARY_BLOCK;          // <<42:Z8L, RndBlk/binary>>: Z8L is a BIT count
        uint32_t padb = z / 8;
        rlen = padb + ABSMAXHALF_BINARY_BLOCK; rnd = ws_alloc(c, rlen); if (!rnd) return 0;
        random_block_rev(c, rnd + padb, ABSMAXHALF_BINARY_BLOCK);
        if (z % 8 != 0) { c.status = CASE_CRASHED; return 0; }   // non byte-aligned bitstring used as /binary -> badarg
        for (uint32_t i = l; i < padb; i += 64) rnd[i] = (i == padb - 1) ? 42 : 0;
      }
      wave_sync();
      pc_add(q, H + e.a, nb + e.len);                             // original length field + blob
      pc_add(q, rnd, rlen); pc_add(q, H + rest0, L - rest0); break;
    }
    case 3: if (l == 0) put_field(fld, newlen, e.size_bits, e.big != 0); wave_sync(); pc_add(q, fld, nb); pc_add(q, H + rest0, L - rest0); break;
    default: if (l == 0) put_field(fld, newlen, e.size_bits, e.big != 0); wave_sync(); pc_add(q, fld, nb); pc_add(q, H + blob0, L - blob0); break;
  }
  pc_emit(c, q);
  return 1;
}

}  // namespace eh
