// eh_fuse2.h — find_jump_points/2 (erlamsa_fuse.erl:102-128) for LARGE lists as a position-indexed class refinement.
//
// eh_fuse.h keeps, per generation, the suffix lists of every node (sorted position arrays); a round moves every member
// and costs a dependent gather/scatter per member, which is what 81 % of all wave cycles of the bench workload went
// into (fuse calls on blocks >= 32 KiB, profiles/r03_summary.json).  Nothing of that structure is needed for the
// RESULT: any_position_pair/1 (:73-77) picks ONE node and one member of each of its two lists.  What the loop needs is
//   * n_g, the number of nodes of every generation (fuel, NoDesp =:= []), and
//   * at the end, the members of one node in list order.
// Both follow from the node INDEX of every suffix.  A suffix is identified by its start p0 (a member's current
// position is p0 + g); id_g[p0] = index of its node in array order (eh_fuse.h: reference order on even generations,
// reversed on odd ones; in array order the next bytes ascend on even and descend on odd rounds, and the members of a
// node are always in ascending position, because every round is a stable partition), or DEAD.  One round:
//   1. every live non-empty member sets bit (id, byte') in a bitmap of 256 bits per node, per side;
//   2. children = bits set on both sides, in bitmap order = array order; an exclusive prefix count over the words
//      gives every child its index (RK = {mask, prefix} per 32-bit word), the total is n_{g+1};
//   3. id_{g+1}[p0] = RK[id*8 + byte'/32].prefix + popc(mask below byte'), DEAD if the bit is not in the mask.
// Step 3 of round g and step 1 of round g+1 are ONE streaming pass over the positions (the next round is known to
// happen: fuel and the next rand(8) are known before the pass).  Every array is indexed by position or by node, all
// accesses are coalesced streams except the 8-byte RK gather and the bit set (a returnless atomic).  The bitmaps are
// bounded by the fuel: a round only runs while the node counts sum to <= 100 000.
//
// The entry whose rest is [] (fix_empty_list/1 :58-60): the member at position len-1 is dropped when it is inserted
// FIRST into its group, i.e. (it is the largest position of its group and lists flip every round) when the group has
// no other member or the generation is odd.  Alone on the source side, its group is [] and the child is the special
// node {[[]], [[]]} whatever the target side holds (:91-93); alone on the target side, the child has no targets.
#pragma once
#include "eh_fuse.h"

namespace eh {

constexpr uint32_t FB_DEAD = 0xFFFFFFFFu;
constexpr uint32_t FB_LDS_WORDS = EH_FUSE_LDS_WORDS;          // g_fuse_lds: bitmaps of <= 256 nodes (every first round) are built in LDS
constexpr uint32_t FB_TEST_WORDS = 65536;        // bitmaps up to this size are hot: many members per bit, test (a load at L2) before the atomic
struct RkWord { uint32_t mask, prefix; };

EH_DEV uint32_t fb_tr(uint32_t b, uint32_t g) { return (g & 1u) ? 255u - b : b; }
// bitmap words are written by atomics (performed at L2): read them there, not from a stale L1 line
EH_DEV uint32_t fb_ld(const uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
EH_DEV uint32_t wave_min(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { uint32_t t = (uint32_t)__shfl_xor((int)v, d); v = t < v ? t : v; }
  return uni(v);
}
EH_DEV void fb_clear(uint32_t* p, uint32_t nwords) {               // nwords is a multiple of 4, p 16-byte aligned
  uint4 z; z.x = z.y = z.z = z.w = 0;
  for (uint32_t i = 4u * (uint32_t)EH_LANE; i < nwords; i += 256) *reinterpret_cast<uint4*>(p + i) = z;
}

// Bitmap of generation 0 (one node): the bytes of S[0, len-1) (the member at len-1 is tested separately).  256 LDS flags,
// packed into 8 words by ballots.
__device__ __noinline__ void fb_bits0(const uint8_t* S, uint32_t len, uint32_t* M) {
  const uint32_t l = (uint32_t)EH_LANE;
  for (uint32_t i = l; i < 256; i += 64) g_fuse_lds[i] = 0;
  lanes_sync();
  const uint32_t n = len - 1;
  for (uint32_t base = 0; base < n; base += 4096) {                // 4 vector loads in flight
    uint4 v[4]; uint32_t i0[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { i0[u] = base + 1024u * u + 16u * l; uint4 z; z.x = z.y = z.z = z.w = 0; v[u] = z; if (i0[u] + 16 <= n) __builtin_memcpy(&v[u], S + i0[u], 16); }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (i0[u] + 16 <= n) {
        uint32_t w[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
        for (int k = 0; k < 16; k++) g_fuse_lds[(w[k >> 2] >> (8 * (k & 3))) & 255u] = 1;
      } else {
        for (uint32_t k = 0; k < 16; k++) if (i0[u] + k < n) g_fuse_lds[S[i0[u] + k]] = 1;
      }
    }
  }
  lanes_sync();
#pragma unroll
  for (int i = 0; i < 4; i++) {
    unsigned long long m = __ballot(g_fuse_lds[64 * i + l] != 0);
    if (l == 0) { M[2 * i] = (uint32_t)m; M[2 * i + 1] = (uint32_t)(m >> 32); }
  }
  wave_sync();
}

// Children of one round: RK[w] = {MA[w] & MB[w], children before word w}; returns the number of children.  A source
// group that is [] (its only member was the dropped one) is a child whatever the target side holds: force_bit in
// word force_w; *special = its index.  Lane l owns 16 consecutive words of every 1024-word step (one shuffle scan per
// step, 4 + 4 vector loads in flight).
__device__ __noinline__ uint32_t fb_scan(const uint32_t* MA, const uint32_t* MB, uint32_t nwords, uint32_t force_w, uint32_t force_bit, RkWord* RK, uint32_t* special) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint32_t running = 0, sp = FB_DEAD;
  for (uint32_t base = 0; base < nwords; base += 1024) {
    const uint32_t w0 = base + 16u * l;
    uint32_t m[16];
    uint4 va[4], vb[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      uint4 z; z.x = z.y = z.z = z.w = 0; va[u] = z; vb[u] = z;
      if (w0 + 4u * u < nwords) {                                  // (word by word: bitmap words are read at L2, see fb_ld)
        const uint32_t* a = MA + w0 + 4 * u; va[u].x = fb_ld(a); va[u].y = fb_ld(a + 1); va[u].z = fb_ld(a + 2); va[u].w = fb_ld(a + 3);
        if (MB) { const uint32_t* b = MB + w0 + 4 * u; vb[u].x = fb_ld(b); vb[u].y = fb_ld(b + 1); vb[u].z = fb_ld(b + 2); vb[u].w = fb_ld(b + 3); }
      }
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      uint32_t a4[4] = {va[u].x, va[u].y, va[u].z, va[u].w}, b4[4] = {vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        uint32_t w = w0 + 4u * u + k;
        uint32_t v = MB ? (a4[k] & b4[k]) : a4[k];
        if (w == force_w) v |= force_bit;
        m[4 * u + k] = v; cnt += (uint32_t)__popc(v);
      }
    }
    uint32_t inc = wave_incl_scan(cnt);
    uint32_t ex = running + inc - cnt;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      if (w0 + 4u * u < nwords) {
        uint32_t pre[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
          pre[k] = ex;
          if (w0 + 4u * u + k == force_w && force_bit) sp = ex + (uint32_t)__popc(m[4 * u + k] & (force_bit - 1u));
          ex += (uint32_t)__popc(m[4 * u + k]);
        }
        uint4 s0, s1;
        s0.x = m[4 * u]; s0.y = pre[0]; s0.z = m[4 * u + 1]; s0.w = pre[1];
        s1.x = m[4 * u + 2]; s1.y = pre[2]; s1.z = m[4 * u + 3]; s1.w = pre[3];
        *reinterpret_cast<uint4*>(RK + w0 + 4 * u) = s0;
        *reinterpret_cast<uint4*>(RK + w0 + 4 * u + 2) = s1;
      }
    }
    running += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  }
  *special = wave_min(sp);
  wave_sync();
  return running;
}

// 8 bytes of S from q (fewer at the end of the block: zero filled)
EH_DEV uint64_t fb_ld8(const uint8_t* S, uint32_t q, uint32_t len) {
  uint64_t by = 0;
  if (q + 8 <= len) __builtin_memcpy(&by, S + q, 8);
  else { for (uint32_t k = 0; k < 5; k++) if (q + k < len) by |= (uint64_t)S[q + k] << (8 * k); }
  return by;
}
// One streaming pass over the suffixes of one side: id_{g+1} from id_g (g == 0: every suffix is in node 0) through RK,
// and, when Mn != nullptr, the bitmap of the NEXT round (every live member of generation g+1 that is not empty and not
// the member at len-1).  kill: target members that fall into the special child; e_pos/e_kill: the member at len-1 of
// THIS round and whether it leaves its group.  Returns the number of live members of generation g+1.
// A lone wavefront is bound by memory round trips, not by bytes: 1024 positions per step (16 per lane), the ids and
// bytes of the NEXT step are requested before this step's 16 RK gathers, so a step costs about one round trip.
template <int MODE>   // how the next bitmap is written: 0 global atomics, 1 LDS atomics, 2 global, test (at L2) before the atomic
__device__ __noinline__ uint32_t fb_pass(const uint8_t* S, uint32_t len, uint32_t* ids, uint32_t g, const RkWord* RK, uint32_t kill, uint32_t e_pos, bool e_kill, uint32_t* Mn) {
  const uint32_t l = (uint32_t)EH_LANE;
  constexpr int U = 4;
  uint32_t alive = 0;
  uint4 idv[U]; uint64_t byv[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const uint32_t p = 256u * u + 4u * l;
    uint4 z; z.x = z.y = z.z = z.w = 0; idv[u] = z; byv[u] = 0;
    if (p < len) { if (g > 0) idv[u] = *reinterpret_cast<const uint4*>(ids + p); byv[u] = fb_ld8(S, p + g, len); }
  }
  for (uint32_t base = 0; base < len; base += 256u * U) {
    uint4 nidv[U]; uint64_t nbyv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {                                  // prefetch the next step
      const uint32_t p = base + 256u * U + 256u * u + 4u * l;
      uint4 z; z.x = z.y = z.z = z.w = 0; nidv[u] = z; nbyv[u] = 0;
      if (p < len) { if (g > 0) nidv[u] = *reinterpret_cast<const uint4*>(ids + p); nbyv[u] = fb_ld8(S, p + g, len); }
    }
    RkWord rk[U][4];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t p = base + 256u * u + 4u * l;
      const uint32_t o[4] = {idv[u].x, idv[u].y, idv[u].z, idv[u].w};
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const bool ok = p + k + g < len && o[k] != FB_DEAD;        // (positions past the end have id 0: p + k + g < len excludes them)
        const uint32_t b = fb_tr((uint32_t)(byv[u] >> (8 * k)) & 255u, g);
        RkWord r; r.mask = 0; r.prefix = 0;
        if (ok) r = RK[o[k] * 8u + (b >> 5)];
        rk[u][k] = r;                                              // mask 0: not a member of any child
      }
    }
    // new ids first, then (MODE 2) the 16 bitmap tests of the step as ONE batch of loads, then the atomics that are still
    // needed: tested one after the other behind their branches, every test was a memory round trip of its own and a step
    // cost sixteen of them (4.4 M cycles per pass in the bench workload's profile, 36 % of all wave cycles).
    uint32_t w2v[U][4], bit2v[U][4];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t p = base + 256u * u + 4u * l;
      uint32_t nw[4];
#pragma unroll
      for (uint32_t k = 0; k < 4; k++) {
        const uint32_t pos = p + k;
        uint32_t nid = FB_DEAD;
        const uint32_t b = fb_tr((uint32_t)(byv[u] >> (8 * k)) & 255u, g);
        const uint32_t bit = 1u << (b & 31u);
        if (rk[u][k].mask & bit) nid = rk[u][k].prefix + (uint32_t)__popc(rk[u][k].mask & (bit - 1u));
        if (nid == kill || (pos == e_pos && e_kill)) nid = FB_DEAD;
        nw[k] = nid;
        w2v[u][k] = 0; bit2v[u][k] = 0;                            // bit2 == 0: nothing to set
        if (nid != FB_DEAD) {
          alive++;
          if (Mn && pos + g + 2 < len) {
            uint32_t b2 = fb_tr((uint32_t)(byv[u] >> (8 * k + 8)) & 255u, g + 1);
            w2v[u][k] = nid * 8u + (b2 >> 5); bit2v[u][k] = 1u << (b2 & 31u);
          }
        }
      }
      if (p < len) { uint4 v; v.x = nw[0]; v.y = nw[1]; v.z = nw[2]; v.w = nw[3]; *reinterpret_cast<uint4*>(ids + p) = v; }
    }
    if (Mn) {
      if (MODE == 2) {
        uint32_t have[U][4];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) have[u][k] = fb_ld(&Mn[w2v[u][k]]);      // (word 0 for the lanes with nothing to set)
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) if (bit2v[u][k] & ~have[u][k]) atomicOr(&Mn[w2v[u][k]], bit2v[u][k]);
      } else {
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) if (bit2v[u][k]) { if (MODE == 1) atomicOr(&g_fuse_lds[w2v[u][k]], bit2v[u][k]); else atomicOr(&Mn[w2v[u][k]], bit2v[u][k]); }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) { idv[u] = nidv[u]; byv[u] = nbyv[u]; }
  }
  wave_sync();
  return wave_sum(alive);
}

// members of `node`: how many, and the position of the k-th (0-based, ascending)
__device__ __noinline__ uint32_t fb_count(const uint32_t* ids, uint32_t len, uint32_t node) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint32_t c = 0;
  for (uint32_t base = 0; base < len; base += 1024) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { uint32_t p = base + 256u * u + 4u * l; uint4 z; z.x = z.y = z.z = z.w = FB_DEAD; v[u] = z; if (p < len) v[u] = *reinterpret_cast<const uint4*>(ids + p); }
#pragma unroll
    for (int u = 0; u < 4; u++) c += (v[u].x == node) + (v[u].y == node) + (v[u].z == node) + (v[u].w == node);
  }
  return wave_sum(c);
}
__device__ __noinline__ uint32_t fb_find(const uint32_t* ids, uint32_t len, uint32_t node, uint32_t k) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint32_t before = 0;
  for (uint32_t base = 0; base < len; base += 1024) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { uint32_t p = base + 256u * u + 4u * l; uint4 z; z.x = z.y = z.z = z.w = FB_DEAD; v[u] = z; if (p < len) v[u] = *reinterpret_cast<const uint4*>(ids + p); }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t p = base + 256u * u + 4u * l;
      const uint32_t e[4] = {v[u].x == node, v[u].y == node, v[u].z == node, v[u].w == node};
      uint32_t c = e[0] + e[1] + e[2] + e[3];
      uint32_t inc = wave_incl_scan(c);
      uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
      if (k < before + tot) {
        uint32_t kk = k - before, ex = inc - c;
        bool mine = kk >= ex && kk < inc;
        uint32_t pos = 0;
        if (mine) { uint32_t r = kk - ex; for (uint32_t j = 0; j < 4; j++) { if (e[j]) { if (r == 0) { pos = p + j; break; } r--; } } }
        unsigned long long who = __ballot(mine);
        return (uint32_t)__builtin_amdgcn_readlane((int)pos, (int)__builtin_ctzll(who));
      }
      before += tot;
    }
  }
  return len;
}

// find_jump_points/2 + any_position_pair/1: *from / *tpos = the positions jump/3 (:47-50) cuts at.  Same draws, same fuel,
// same work accounting as fuse_lists' node-list version (eh_fuse.h).  false: work area exhausted / budget.
__device__ __noinline__ bool fuse_jump_stream(Ctx&, const uint8_t* A, uint32_t la, const uint8_t* B, uint32_t lb, bool sym, uint32_t* from, uint32_t* tpos, uint32_t* rounds) {
  EH_CTX;
  const uint32_t l = (uint32_t)EH_LANE;
  uint32_t* ids[2] = {nullptr, nullptr};
  uint32_t* M[2] = {nullptr, nullptr};
  RkWord* RK = nullptr;
  uint32_t rk_rows = 0, m_rows = 0;                                // nodes RK / M[] have room for
  const uint8_t* S[2] = {A, B};
  const uint32_t len[2] = {la, lb};
  const int nside = sym ? 1 : 2;
  uint32_t nn = 1, g = 0;                                          // nodes of generation g
  int64_t fuel = 100000;                                           // ?SEARCH_FUEL
  uint64_t gen_entries = (uint64_t)la + lb;
  bool have_bits = false;                                          // M holds the bitmap of generation g
  EH_PT0;
  uint32_t special_node = FB_DEAD;                                 // generation g's {[[]], [[]]} node (two distinct lists only)
  while (true) {                                                   // find_jump_points_loop (:115-128)
    if (fuel < 0) break;
    if (rng_rand(c.rng, 8) == 0) break;                            // ?SEARCH_STOP_IP
    if (c.work_budget) {
      c.work += 16ull * gen_entries;
      if (c.work > c.work_budget) { c.status = CASE_BUDGET; return false; }
    }
    // Tables grow with the generations (n_g <= 256 n_{g-1}, and the fuel keeps n_g <= 100 000): a table that has become
    // too small is left behind and a larger one allocated — what is left behind is a fraction of what replaces it.  Sizing
    // everything for 100 000 nodes up front made every fuse of a block of 100 KB and more ask for 13 to 21 MB.
    if (!ids[0]) {
      for (int s = 0; s < nside; s++) { ids[s] = (uint32_t*)ws_alloc(c, ((uint64_t)len[s] + 8) * 4); if (!ids[s]) return false; }
    }
    if (nn > rk_rows) {
      rk_rows = nn < 1024u ? 1024u : nn;
      RK = (RkWord*)ws_alloc(c, ((uint64_t)rk_rows + 4) * 64);
      if (!RK) return false;
    }
    if (nn > m_rows) {                                             // (only generation 0 comes here: later bitmaps are sized below)
      m_rows = nn < 1024u ? 1024u : nn;
      for (int s = 0; s < nside; s++) { M[s] = (uint32_t*)ws_alloc(c, ((uint64_t)m_rows + 4) * 32); if (!M[s]) return false; }
    }
    const uint32_t nwords = nn * 8u;
    EH_PT(c, 100);
    if (!have_bits) for (int s = 0; s < nside; s++) fb_bits0(S[s], len[s], M[s]);     // only generation 0 comes here
    EH_PT(c, 101);
    // ---- the member at len-1 of this round: alone in its group?
    uint32_t e_pos[2] = {FB_DEAD, FB_DEAD}, e_alone[2] = {0, 0}, e_w[2] = {0, 0}, e_bit[2] = {0, 0};
    for (int s = 0; s < nside; s++) {
      if (len[s] < g + 1) continue;
      uint32_t ep = len[s] - 1 - g;
      uint32_t id = g == 0 ? 0u : uni(ids[s][ep]);
      if (id == FB_DEAD) continue;
      uint32_t b = fb_tr(uni(S[s][len[s] - 1]), g);
      e_pos[s] = ep; e_w[s] = id * 8u + (b >> 5); e_bit[s] = 1u << (b & 31u);
      uint32_t old = 0;
      if (l == 0) old = atomicOr(&M[s][e_w[s]], e_bit[s]);
      e_alone[s] = (uni(old) & e_bit[s]) ? 0u : 1u;
    }
    wave_sync();
    EH_PT(c, 102);
    // ---- children
    const bool forced = !sym && e_pos[0] != FB_DEAD && e_alone[0];
    uint32_t sp = FB_DEAD;
    uint32_t nchild = fb_scan(M[0], sym ? nullptr : M[1], nwords, forced ? e_w[0] : FB_DEAD, forced ? e_bit[0] : 0u, RK, &sp);
    EH_PT(c, 103);
    if (nchild == 0) break;                                        // NoDesp =:= [] -> any_position_pair(Nodes)
    // ---- commit: ids of generation g+1 (+ the bitmap of the next round when it is going to run)
    fuel -= (int64_t)nchild;
    const bool next = fuel >= 0 && (uint32_t)(rng_peek(c.rng, 1) * 8.0) != 0;
    const bool lds = next && nchild * 8u <= FB_LDS_WORDS;
    if (next && nchild > m_rows) {                                 // the next generation's bitmaps (this one's are in RK now)
      m_rows = nchild;
      for (int s = 0; s < nside; s++) { M[s] = (uint32_t*)ws_alloc(c, ((uint64_t)m_rows + 4) * 32); if (!M[s]) return false; }
    }
    uint64_t entries = 0;
    for (int s = 0; s < nside; s++) {
      if (next) {
        if (lds) { for (uint32_t i = l; i < nchild * 8u; i += 64) g_fuse_lds[i] = 0; lanes_sync(); }
        else { fb_clear(M[s], nchild * 8u); wave_sync(); }
      }
      EH_PT(c, 104);
      // the member at len-1: leaves when it was inserted first (alone, or an odd generation).  One list on both sides:
      // alone = the special node = the member itself, now empty, so it stays.
      bool ek = sym ? (!e_alone[0] && (g & 1u)) : (e_alone[s] || (g & 1u));
      uint32_t kill = (s == 1 && forced) ? sp : FB_DEAD;
      uint32_t alive = lds ? fb_pass<1>(S[s], len[s], ids[s], g, RK, kill, e_pos[s], ek, M[s])
                           : (next && nchild * 8u <= FB_TEST_WORDS) ? fb_pass<2>(S[s], len[s], ids[s], g, RK, kill, e_pos[s], ek, M[s])
                           : fb_pass<0>(S[s], len[s], ids[s], g, RK, kill, e_pos[s], ek, next ? M[s] : nullptr);
      entries += alive;
      EH_PT(c, lds ? 105 : (next ? (nchild * 8u <= FB_TEST_WORDS ? 110 : 106) : 107));
      if (lds) { lanes_sync(); for (uint32_t i = l; i < nchild * 8u; i += 64) M[s][i] = g_fuse_lds[i]; wave_sync(); }
    }
    if (sym) entries *= 2; else if (forced) entries += 2;
    gen_entries = entries;
    special_node = forced ? sp : FB_DEAD;
    nn = nchild; g++; have_bits = next;
    EH_PT(c, 108);
    (*rounds)++;
  }
  // any_position_pair/1 (:73-77); odd generations are stored reversed
  const uint32_t par = g & 1u;
  uint32_t ni = rng_rand(c.rng, nn);
  uint32_t node = par ? nn - 1 - ni : ni;
  if (g == 0) {                                                    // the one node of all suffixes
    *from = rng_rand(c.rng, la);
    *tpos = rng_rand(c.rng, lb);
    return true;
  }
  if (node == special_node) { (void)rng_rand(c.rng, 1); (void)rng_rand(c.rng, 1); *from = la; *tpos = lb; return true; }
  uint32_t fc = fb_count(ids[0], la, node);
  uint32_t tc = sym ? fc : fb_count(ids[1], lb, node);
  uint32_t f = la, t = lb;
  if (fc > 0) { uint32_t j = rng_rand(c.rng, fc); f = fb_find(ids[0], la, node, par ? fc - 1 - j : j) + g; }
  if (tc > 0) { uint32_t j = rng_rand(c.rng, tc); t = fb_find(sym ? ids[0] : ids[1], lb, node, par ? tc - 1 - j : j) + g; }
  *from = f; *tpos = t;
  EH_PT(c, 109);
  return true;
}

}  // namespace eh
