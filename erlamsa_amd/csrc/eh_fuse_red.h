// eh_fuse_red.h — erlamsa_fuse:fuse/2 on LARGE lists: run the search on a shorter list that has the same nodes.
//
// The large blocks fuse meets are pumped ones (sr, lr, tr, sgm repeat a piece of a 4 KiB seed hundreds of times).  The search
// (find_jump_points/2, erlamsa_fuse.erl:102-128) never looks further than R bytes into a suffix, R - 1 = the rounds its own
// rand(8) draws allow (they do not depend on the data and can be read ahead), and what it computes per generation g <= R is
//   * which g-grams occur on both sides (the nodes, in key order) and how many there are (the fuel, NoDesp =:= []),
//   * what happens to the member whose rest is [] (fix_empty_list/1: is it alone in its group?) - a question about the last
//     bytes of the list and about whether their g-gram occurs anywhere else.
// None of this depends on HOW OFTEN a g-gram occurs.  So a stretch S[a, b) with S[k] = S[k + P] can lose whole periods from
// its middle: cutting [u, u + D) with u >= a + P, D a multiple of P and u + D + R <= b leaves every g-gram (g <= R) of the
// list in place (a removed position has the g-gram of its counterpart in the period before u), creates none (the bytes after
// the cut continue the bytes before it) and leaves the last R bytes alone.  The generations of the shortened lists therefore
// have the same node counts, in the same order, and the search draws the same node - and the members of that node in the
// ORIGINAL lists are simply the occurrences of its g-gram (ascending positions; plus, possibly, the member whose rest is [],
// which the shortened run reports), found by one streaming compare pass instead of g refinement passes over every position.
// Several cuts compose (each is valid on the list the previous one left).  A list without a long periodic stretch, a run
// with a work budget (its accounting counts list members) and runs whose draws allow 64 rounds or more go the ordinary way.
#pragma once

namespace eh {

constexpr uint32_t FR_NONE = 0xFFFFFFFFu;
constexpr uint32_t FR_MAX_PERIOD = 1u << 16;        // how far the next occurrence of an anchor's 8 bytes is looked for
constexpr int FR_LEVELS = 4;

EH_DEV uint64_t fr_ld8(cbptr S, uint32_t q, uint32_t len) {      // 8 bytes from q, zero filled past the end
  uint64_t by = 0;
  if (q + 8 <= len) by = ldg8(S + q);
  else { for (uint32_t k = 0; k < 8; k++) if (q + k < len) by |= (uint64_t)S[q + k] << (8 * k); }
  return by;
}
// how many rand(8) draws from now come out non-zero before the first zero (= rounds the search may run); 64: none of the next 64 is zero
EH_DEV uint32_t fr_peek_rounds(const Rng& r) {
  const uint32_t l = (uint32_t)EH_LANE;
  double x = rng_peek(r, l + 1u) * 8.0;
  unsigned long long z = __ballot((uint32_t)x == 0u);
  return z ? (uint32_t)__builtin_ctzll(z) : 64u;
}
// first i >= i0 with i + p >= n or S[i] != S[i + p]
__device__ __noinline__ uint32_t fr_run_fwd(cbptr S, uint32_t i0, uint32_t p, uint32_t n) {
  const uint32_t l = (uint32_t)EH_LANE;
  const uint32_t lim = n - p;
  for (uint32_t base = i0;; base += 1024) {
    const uint32_t q = base + 16u * l;
    uint32_t mp = FR_NONE;
    if (q + 16 <= lim) {
      uint4 a, b;
      a = ldg16(S + q); b = ldg16(S + q + p);
      const uint32_t x[4] = {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w};
#pragma unroll
      for (int k = 3; k >= 0; k--) if (x[k]) mp = q + 4u * (uint32_t)k + ((uint32_t)__builtin_ctz(x[k]) >> 3);
    } else {
      for (uint32_t k = 0; k < 16; k++) { if (q + k >= lim || S[q + k] != S[q + k + p]) { mp = q + k; break; } }
    }
    unsigned long long hit = __ballot(mp != FR_NONE);
    if (hit) return (uint32_t)__builtin_amdgcn_readlane((int)mp, (int)__builtin_ctzll(hit));
  }
}
// smallest lo <= i0 with S[k] == S[k + p] for every k in [lo, i0)   (i0 + p <= n)
__device__ __noinline__ uint32_t fr_run_bwd(cbptr S, uint32_t i0, uint32_t p) {
  const uint32_t l = (uint32_t)EH_LANE;
  for (uint32_t hi = i0;; hi -= 1024) {                              // lane l looks at [hi - 16 (l + 1), hi - 16 l), lane 0 the highest
    uint32_t mp = FR_NONE;                                           // highest mismatching position of this lane
    const uint32_t top = 16u * l < hi ? hi - 16u * l : 0u, bot = 16u * (l + 1u) < hi ? hi - 16u * (l + 1u) : 0u;
    if (top - bot == 16u) {
      uint4 a, b;
      a = ldg16(S + bot); b = ldg16(S + bot + p);
      const uint32_t x[4] = {a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w};
#pragma unroll
      for (int k = 0; k < 4; k++) if (x[k]) mp = bot + 4u * (uint32_t)k + ((31u - (uint32_t)__builtin_clz(x[k])) >> 3);
    } else {
      for (uint32_t k = bot; k < top; k++) if (S[k] != S[k + p]) mp = k;
    }
    unsigned long long hit = __ballot(mp != FR_NONE);
    if (hit) return (uint32_t)__builtin_amdgcn_readlane((int)mp, (int)__builtin_ctzll(hit)) + 1u;
    if (hi <= 1024) return 0;
  }
}
// first y in [from, lim) with y + 8 <= n and the 8 bytes at y equal to key, FR_NONE: none.  1024 positions per step: a lane
// loads the 24 bytes at its 16 positions and slides an 8-byte window over them.
__device__ __noinline__ uint32_t fr_find8(cbptr S, uint32_t from, uint32_t lim, uint32_t n, uint64_t key) {
  const uint32_t l = (uint32_t)EH_LANE;
  if (n < 8) return FR_NONE;
  if (lim > n - 7) lim = n - 7;
  for (uint32_t base = from; base < lim; base += 1024) {
    const uint32_t q = base + 16u * l;
    uint64_t w0 = 0, w1 = 0, w2 = 0;
    if (q + 24 <= n) { w0 = ldg8(S + q); w1 = ldg8(S + q + 8); w2 = ldg8(S + q + 16); }
    else if (q < lim) { w0 = fr_ld8(S, q, n); w1 = fr_ld8(S, q + 8, n); w2 = fr_ld8(S, q + 16, n); }
    uint32_t first = 16;
#pragma unroll
    for (int k = 15; k >= 0; k--) {
      const uint64_t lo = k < 8 ? w0 : w1, hi = k < 8 ? w1 : w2;
      const int sh = 8 * (k & 7);
      const uint64_t win = sh ? (lo >> sh) | (hi << (64 - sh)) : lo;
      if (win == key && q + (uint32_t)k < lim) first = (uint32_t)k;
    }
    unsigned long long hit = __ballot(first < 16);
    if (hit) { int src = (int)__builtin_ctzll(hit); return base + 16u * (uint32_t)src + (uint32_t)__builtin_amdgcn_readlane((int)first, src); }
  }
  return FR_NONE;
}
// The best cut of S[0, n) for searches of at most R bytes' depth: *u, *D (bytes [u, u + D) can go; *cP: the period: S[k] == S[k + P]
// for k in [u - P, u + D + R - P)); false: nothing worth it.
// Anchors at the eighths of the list: the 8 bytes there, their next occurrences as period candidates, the stretch each holds for.
__device__ __noinline__ bool fr_find_cut(cbptr S, uint32_t n, uint32_t R, uint32_t* cu, uint32_t* cD, uint32_t* cP = nullptr) {
  uint32_t bestD = 0, bestU = 0, blo = 0, bhi = 0, bestP = 0;
  if (n < 4096) return false;
  bool any = false;                                                  // some anchor's bytes came again at all
  for (uint32_t ai = 0; ai < 7; ai++) {
    // the middle and the quarters first: a list in which none of them sees its 8 bytes again within FR_MAX_PERIOD has no stretch
    // worth cutting (it would have to miss all three), and the search must stay cheap for lists that are not periodic at all
    const uint32_t a = ai == 0 ? 4u : ai == 1 ? 2u : ai == 2 ? 6u : ai == 3 ? 1u : ai == 4 ? 3u : ai == 5 ? 5u : 7u;
    if (ai == 3 && !any) break;
    const uint32_t x = (uint32_t)(((uint64_t)n * a) >> 3);
    if (x + 8 > n) continue;
    if (bestD && x >= blo && x < bhi) continue;                      // inside the stretch already found
    const uint64_t key = fr_ld8(S, x, n);
    uint32_t y = x + 1;
    for (int tries = 0; tries < 4; tries++) {
      const uint32_t far = x + 1u + FR_MAX_PERIOD < n ? x + 1u + FR_MAX_PERIOD : n;
      y = fr_find8(S, y, far, n, key);
      if (y == FR_NONE) break;
      any = true;
      const uint32_t p = y - x;
      const uint32_t hi = fr_run_fwd(S, x, p, n), lo = fr_run_bwd(S, x, p);
      const uint32_t u = lo + p, b = hi + p;                          // S[k] == S[k + p] on [lo, hi)
      if (b > u + R && b - u - R >= p) {
        const uint32_t D = (b - R - u) / p * p;
        if (D > bestD) { bestD = D; bestU = u; blo = lo; bhi = b; bestP = p; }
      }
      if (bestD >= n / 2) break;
      y++;
    }
    if (bestD >= n / 2) break;
  }
  if (bestD < 2048 || bestD < n / 8) return false;
  *cu = bestU; *cD = bestD;
  if (cP) *cP = bestP;
  return true;
}
// S[0, *n) -> a shortened copy in the work area (or S itself, untouched, when no cut is worth it); *n its length.  nullptr: work area exhausted.
__device__ __noinline__ cbptr fr_reduce(Ctx&, cbptr S, uint32_t* n, uint32_t R) {
  EH_CTX;
  uint32_t u = 0, D = 0, len = *n;
  if (!fr_find_cut(S, len, R, &u, &D)) return S;
  bptr C = ws_alloc(c, (uint64_t)len - D + 16);
  if (!C) return nullptr;
  wave_copy(C, S, u);
  wave_copy(C + u, S + u + D, len - u - D);
  len -= D;
  wave_sync();
  for (int lv = 1; lv < FR_LEVELS; lv++) {
    if (!fr_find_cut(C, len, R, &u, &D)) break;
    wave_sync();
    wave_move_down(C + u, C + u + D, len - u - D);
    len -= D;
    wave_sync();
  }
  *n = len;
  return C;
}
// Occurrences of key[0, g) in S at positions s < lim (s + g <= len): their number; with want != FR_NONE the position of the
// want-th one (0-based, ascending) goes to *pos and the scan stops there.
// A lane takes 16 consecutive positions per step: their 8-byte windows come out of 24 bytes it loads once (until round 5 every
// position was an 8-byte load of its own: eight times the bytes, 256 positions per memory round trip instead of 4 096 - and the four
// passes a fuse call on shortened lists makes over its ORIGINAL megabyte lists were most of what that call cost).
__device__ __noinline__ uint32_t fr_occ(cbptr S, uint32_t len, uint32_t lim, cbptr key, uint32_t g, uint32_t want, uint32_t* pos) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint64_t k8 = 0;
  for (uint32_t k = 0; k < 8 && k < g; k++) k8 |= (uint64_t)key[k] << (8 * k);
  const uint64_t mask = g >= 8 ? ~0ull : ((1ull << (8 * g)) - 1ull);
  uint32_t seen = 0;
  for (uint32_t base = 0; base < lim; base += 4096) {
    uint64_t lo[4], hi[4], nx[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t s0 = base + 1024u * (uint32_t)u + 16u * l;
      lo[u] = hi[u] = nx[u] = 0;
      if (s0 < lim) {
        if (s0 + 24 <= len) { const uint4 v = ldg16(S + s0); lo[u] = (uint64_t)v.x | ((uint64_t)v.y << 32); hi[u] = (uint64_t)v.z | ((uint64_t)v.w << 32); nx[u] = ldg8(S + s0 + 16); }
        else { lo[u] = fr_ld8(S, s0, len); hi[u] = fr_ld8(S, s0 + 8, len); nx[u] = fr_ld8(S, s0 + 16, len); }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t s0 = base + 1024u * (uint32_t)u + 16u * l;
      uint32_t bits = 0;
#pragma unroll
      for (uint32_t k = 0; k < 16; k++) {
        const uint64_t a = k < 8 ? lo[u] : hi[u], b = k < 8 ? hi[u] : nx[u];
        const uint32_t sh = 8u * (k & 7u);
        const uint64_t w = sh ? (a >> sh) | (b << (64u - sh)) : a;
        if ((w & mask) == k8 && s0 + k < lim) bits |= 1u << k;
      }
      if (g > 8 && __ballot(bits != 0)) {                           // the first 8 bytes agree: the rest, byte by byte
        uint32_t left = bits;
        while (left) {
          const uint32_t k = (uint32_t)__builtin_ctz(left); left &= left - 1u;
          const uint32_t s = s0 + k;
          for (uint32_t q = 8; q < g; q++) if (S[s + q] != key[q]) { bits &= ~(1u << k); break; }
        }
      }
      const uint32_t cnt = (uint32_t)__popc(bits);
      const uint32_t inc = wave_incl_scan(cnt);
      const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      if (want != FR_NONE && want < seen + tot) {
        const uint32_t r = want - seen, ex = inc - cnt;
        const bool mine = r >= ex && r < inc;
        uint32_t p = 0;
        if (mine) { uint32_t b = bits; for (uint32_t t = ex; t < r; t++) b &= b - 1u; p = s0 + (uint32_t)__builtin_ctz(b); }
        const unsigned long long who = __ballot(mine);
        *pos = (uint32_t)__builtin_amdgcn_readlane((int)p, (int)__builtin_ctzll(who));
        return seen + tot;
      }
      seen += tot;
    }
  }
  return seen;
}

}  // namespace eh
