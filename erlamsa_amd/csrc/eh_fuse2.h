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
// g_fuse_lds (16 KiB): words [0, 2048) hold the compact lookup entries of the current generation when it has <= 1024 nodes;
// words [2048, 4096) hold the next generation's table while a pass fills it: bitmap rows of <= 256 nodes (every first
// round), or N1 entries of <= 2048 nodes.  A pass whose tables are both in LDS moves ids and bytes only.
constexpr uint32_t FB_LDS_SC_NODES = 1024, FB_LDS_NEXT = 2048, FB_LDS_WORDS = 2048, FB_LDS_N1_NODES = 2048;
constexpr uint32_t FB_TEST_WORDS = 65536;        // bitmaps up to this size are hot: many members per bit, test (a load at L2) before the atomic
struct RkWord { uint32_t mask, prefix; };

EH_DEV uint32_t fb_tr(uint32_t b, uint32_t g) { return (g & 1u) ? 255u - b : b; }
// bitmap words are written by atomics (performed at L2): read them there, not from a stale L1 line
EH_DEV uint32_t fb_ld(cwptr p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
EH_DEV uint32_t wave_min(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { uint32_t t = (uint32_t)__shfl_xor((int)v, d); v = t < v ? t : v; }
  return uni(v);
}
EH_DEV void fb_clear(wptr p, uint32_t nwords) {               // nwords is a multiple of 4, p 16-byte aligned
  uint4 z; z.x = z.y = z.z = z.w = 0;
  for (uint32_t i = 4u * (uint32_t)EH_LANE; i < nwords; i += 256) stg16a(p + i, z);
}

// Bitmap of generation 0 (one node): the bytes of S[0, len-1) (the member at len-1 is tested separately).  256 LDS flags,
// packed into 8 words by ballots.
__device__ __noinline__ void fb_bits0(cbptr S, uint32_t len, wptr M) {
  const uint32_t l = (uint32_t)EH_LANE;
  for (uint32_t i = l; i < 256; i += 64) g_fuse_lds[i] = 0;
  lanes_sync();
  const uint32_t n = len - 1;
  for (uint32_t base = 0; base < n; base += 4096) {                // 4 vector loads in flight
    uint4 v[4]; uint32_t i0[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { i0[u] = base + 1024u * u + 16u * l; uint4 z; z.x = z.y = z.z = z.w = 0; v[u] = z; if (i0[u] + 16 <= n) v[u] = ldg16(S + i0[u]); }
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

// A node's next bytes are recorded in two levels (fb_pass, INS 3): N1[node] = 0 (none yet), byte' + 1 (exactly one
// byte so far), FB_MULTI (several: they are in the node's bitmap row).  On repetitive data — where the large blocks come
// from: sr, lr, tr and sgm pumps — a k-gram nearly always has ONE continuation, so a position touches a 4-byte entry
// (sixteen nodes to a cache line) instead of a 32-byte row, and the rows of single-byte nodes are never written.
constexpr uint32_t FB_MULTI = 0xFFFFFFFFu;
// The lookup side of the same idea: SC[node] = {mask, prefix | word << 24 | kind << 28}, kind 0: no child, 1: all
// children in ONE mask word (the entry is the whole answer), 2: several words — see RK.
struct ScEnt { uint32_t mask, meta; };

// the eight mask words of `node` (bitmap row, or decoded from N1), and the tables left ZERO for the next generation
EH_DEV void fb_row(wptr M, wptr N1, uint32_t node, bool valid, uint32_t w[8]) {
#pragma unroll
  for (int k = 0; k < 8; k++) w[k] = 0;
  if (!valid) return;
  wptr row = M + 8u * node;
  bool full = true;
  if (N1) {
    uint32_t n1 = fb_ld(&N1[node]);
    full = n1 == FB_MULTI;
    if (n1 != 0) N1[node] = 0;
    if (!full && n1 != 0) {
#pragma unroll
      for (int k = 0; k < 8; k++) if ((uint32_t)k == ((n1 - 1u) >> 5)) w[k] = 1u << ((n1 - 1u) & 31u);
    }
  }
  if (full) {
#pragma unroll
    for (int k = 0; k < 8; k++) w[k] = fb_ld(row + k);
    uint4 z; z.x = z.y = z.z = z.w = 0;
    stg16a(row, z); stg16a(row + 4, z);
  }
}

// Children of one round: RK[w] = {MA[w] & MB[w], children before word w} and SC[node]; returns the number of children,
// *nfull = nodes whose children spread over several mask words.  A source group that is [] (its only member was the
// dropped one) is a child whatever the target side holds: force_bit in word force_w; *special = its index.  Lane l owns
// two nodes (16 consecutive words) of every 128-node step.  The bitmap rows and N1 entries it reads are zeroed: the
// tables of a generation are clean when the next one starts to fill them.
__device__ __noinline__ uint32_t fb_scan(wptr MA, wptr MB, wptr N1A, wptr N1B, uint32_t nwords, uint32_t force_w, uint32_t force_bit,
                                         EH_G RkWord* RK, EH_G ScEnt* SC, uint32_t* special, uint32_t* nfull) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint32_t running = 0, sp = FB_DEAD, full = 0;
  for (uint32_t base = 0; base < nwords; base += 1024) {
    const uint32_t w0 = base + 16u * l;
    uint32_t m[16];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t node = (w0 >> 3) + (uint32_t)h;
      const bool valid = node * 8u < nwords;
      uint32_t a[8], b[8];
      fb_row(MA, N1A, node, valid, a);
      if (MB) fb_row(MB, N1B, node, valid, b);
#pragma unroll
      for (int k = 0; k < 8; k++) {
        uint32_t v = MB ? (a[k] & b[k]) : a[k];
        if (node * 8u + (uint32_t)k == force_w) v |= force_bit;
        m[8 * h + k] = v;
      }
    }
    uint32_t cnt = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) cnt += (uint32_t)__popc(m[k]);
    uint32_t inc = wave_incl_scan(cnt);
    uint32_t ex = running + inc - cnt;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint32_t node = (w0 >> 3) + (uint32_t)h;
      if (node * 8u < nwords) {
        uint32_t pre[8], nz = 0, wsel = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
          pre[k] = ex;
          if (node * 8u + (uint32_t)k == force_w && force_bit) sp = ex + (uint32_t)__popc(m[8 * h + k] & (force_bit - 1u));
          if (m[8 * h + k]) { nz++; wsel = (uint32_t)k; }
          ex += (uint32_t)__popc(m[8 * h + k]);
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          uint4 s0; s0.x = m[8 * h + 2 * q]; s0.y = pre[2 * q]; s0.z = m[8 * h + 2 * q + 1]; s0.w = pre[2 * q + 1];
          stg16a(RK + node * 8u + 2 * q, s0);
        }
        ScEnt e; e.mask = 0; e.meta = 0;
        if (nz == 1) {
#pragma unroll
          for (int k = 0; k < 8; k++) if ((uint32_t)k == wsel) { e.mask = m[8 * h + k]; e.meta = pre[k] | (wsel << 24) | (1u << 28); }
        } else if (nz > 1) { e.meta = 2u << 28; full++; }
        SC[node] = e;
      }
    }
    running += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  }
  *special = wave_min(sp);
  *nfull = wave_sum(full);
  wave_sync();
  return running;
}

// 8 bytes of S from q (fewer at the end of the block: zero filled)
EH_DEV uint64_t fb_ld8(cbptr S, uint32_t q, uint32_t len) {
  uint64_t by = 0;
  if (q + 8 <= len) by = ldg8(S + q);
  else { for (uint32_t k = 0; k < 5; k++) if (q + k < len) by |= (uint64_t)S[q + k] << (8 * k); }
  return by;
}
// One streaming pass over the suffixes of one side: id_{g+1} from id_g (g == 0: every suffix is in node 0) through SC / RK,
// and, when Mn != nullptr, the next-byte table of the NEXT round (every live member of generation g+1 that is not empty and
// not the member at len-1).  kill: target members that fall into the special child; e_pos/e_kill: the member at len-1 of
// THIS round and whether it leaves its group.  Returns the number of live members of generation g+1.
// A lone wavefront is bound by memory round trips, not by bytes: 1024 positions per step (16 per lane), the ids and
// bytes of the NEXT step are requested before this step's 16 lookups, and every kind of access of a step (lookups,
// tests of the next table) is issued as one batch, so a step costs a few round trips.
template <int INS>   // how the next table is written: 0 bitmap, global atomics; 1 bitmap in LDS; 2 bitmap, test (at L2) before the atomic; 3 N1 + bitmap rows; 4 N1 in LDS + bitmap rows
__device__ __noinline__ uint32_t fb_pass(cbptr S, uint32_t len, wptr ids, uint32_t g, const EH_G RkWord* RK, const EH_G ScEnt* SC, bool sc_lds, uint32_t kill, uint32_t e_pos, uint32_t lim, wptr Mn, wptr N1n) {   // e_pos: the member at len-1 when it leaves its group (FB_DEAD: it stays); lim: positions [0, lim) are done - len, or where a chunk of a posted pass ends (a multiple of 1024)
  const uint32_t l = (uint32_t)EH_LANE;
  constexpr int U = 4;
  uint32_t alive = 0;
  uint4 idv[U]; uint64_t byv[U];
#pragma unroll
  for (int u = 0; u < U; u++) {
    const uint32_t p = 256u * u + 4u * l;
    uint4 z; z.x = z.y = z.z = z.w = 0; idv[u] = z; byv[u] = 0;
    if (p < len) { if (g > 0) idv[u] = ldg16a(ids + p); byv[u] = fb_ld8(S, p + g, len); }
  }
  for (uint32_t base = 0; base < lim; base += 256u * U) {
    uint4 nidv[U]; uint64_t nbyv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {                                  // prefetch the next step
      const uint32_t p = base + 256u * U + 256u * u + 4u * l;
      uint4 z; z.x = z.y = z.z = z.w = 0; nidv[u] = z; nbyv[u] = 0;
      if (p < len) { if (g > 0) nidv[u] = ldg16a(ids + p); nbyv[u] = fb_ld8(S, p + g, len); }
    }
    RkWord rk[U][4];
    if (SC) {                                                      // compact entries first, rows only for the nodes that need them
      ScEnt sc[U][4];
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t p = base + 256u * u + 4u * l;
        const uint32_t o[4] = {idv[u].x, idv[u].y, idv[u].z, idv[u].w};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
          const bool ok = p + k + g < len && o[k] != FB_DEAD;
          ScEnt e; e.mask = 0; e.meta = 0;
          if (ok) { if (sc_lds) { e.mask = g_fuse_lds[2u * o[k]]; e.meta = g_fuse_lds[2u * o[k] + 1u]; } else e = SC[o[k]]; }
          sc[u][k] = e;
        }
      }
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t o[4] = {idv[u].x, idv[u].y, idv[u].z, idv[u].w};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
          const uint32_t b = fb_tr((uint32_t)(byv[u] >> (8 * k)) & 255u, g);
          const uint32_t kind = sc[u][k].meta >> 28;
          RkWord r; r.mask = 0; r.prefix = 0;
          if (kind == 1u && (b >> 5) == ((sc[u][k].meta >> 24) & 7u)) { r.mask = sc[u][k].mask; r.prefix = sc[u][k].meta & 0xFFFFFFu; }
          if (kind == 2u) r = RK[o[k] * 8u + (b >> 5)];
          rk[u][k] = r;
        }
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t p = base + 256u * u + 4u * l;
        const uint32_t o[4] = {idv[u].x, idv[u].y, idv[u].z, idv[u].w};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {
          const bool ok = p + k + g < len && o[k] != FB_DEAD;      // (positions past the end have id 0: p + k + g < len excludes them)
          const uint32_t b = fb_tr((uint32_t)(byv[u] >> (8 * k)) & 255u, g);
          RkWord r; r.mask = 0; r.prefix = 0;
          if (ok) r = RK[o[k] * 8u + (b >> 5)];
          rk[u][k] = r;                                            // mask 0: not a member of any child
        }
      }
    }
    // new ids first, then the tests of the next table as ONE batch of loads, then the atomics that are still needed:
    // tested one after the other behind their branches, every test was a memory round trip of its own and a step cost
    // sixteen of them (4.4 M cycles per pass in the bench workload's profile, 36 % of all wave cycles).
    uint32_t w2v[U][4], bit2v[U][4], nidn[U][4];
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
        if (nid == kill || pos == e_pos) nid = FB_DEAD;
        nw[k] = nid;
        w2v[u][k] = 0; bit2v[u][k] = 0; nidn[u][k] = 0;            // bit2 == 0: nothing to set
        if (nid != FB_DEAD) {
          alive++;
          if (Mn && pos + g + 2 < len) {
            uint32_t b2 = fb_tr((uint32_t)(byv[u] >> (8 * k + 8)) & 255u, g + 1);
            w2v[u][k] = nid * 8u + (b2 >> 5); bit2v[u][k] = 1u << (b2 & 31u); nidn[u][k] = nid;
          }
        }
      }
      if (p < len) { uint4 v; v.x = nw[0]; v.y = nw[1]; v.z = nw[2]; v.w = nw[3]; stg16a(ids + p, v); }
    }
    if (Mn) {
      if (INS == 3 || INS == 4) {
        // first level: N1 (global, tests batched; or LDS).  A member whose byte is not the node's single recorded byte
        // has to be in the node's bitmap row: those bits are tested as one batch of loads and set only where missing —
        // under a node with several continuations every member comes this way, and an atomic per member on the same few
        // words serialises at the L2.
        uint32_t have[U][4];
        if (INS == 3) {
#pragma unroll
          for (int u = 0; u < U; u++)
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) have[u][k] = fb_ld(&N1n[nidn[u][k]]);     // (entry 0 for the lanes with nothing to record)
        }
        bool row[U][4];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) {
            row[u][k] = false;
            if (!bit2v[u][k]) continue;
            const uint32_t key = ((w2v[u][k] & 7u) << 5) + (uint32_t)__builtin_ctz(bit2v[u][k]) + 1u;     // byte' + 1
            uint32_t* e = INS == 4 ? &g_fuse_lds[FB_LDS_NEXT + nidn[u][k]] : (uint32_t*)&N1n[nidn[u][k]];   // (INS is a template argument: one address space per instance)
            uint32_t old = INS == 4 ? *e : have[u][k];
            if (old == key) continue;
            if (old == 0) old = atomicCAS(e, 0u, key);
            if (old == 0 || old == key) continue;
            if (old != FB_MULTI) {                                 // a second byte under this node: both go to its bitmap row
              atomicOr(&Mn[nidn[u][k] * 8u + ((old - 1u) >> 5)], 1u << ((old - 1u) & 31u));
              atomicExch(e, FB_MULTI);
            }
            row[u][k] = true;
          }
        uint32_t bits[U][4];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) bits[u][k] = row[u][k] ? fb_ld(&Mn[w2v[u][k]]) : 0xFFFFFFFFu;
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (uint32_t k = 0; k < 4; k++) if (bit2v[u][k] & ~bits[u][k]) atomicOr(&Mn[w2v[u][k]], bit2v[u][k]);
      } else if (INS == 2) {
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
          for (uint32_t k = 0; k < 4; k++) if (bit2v[u][k]) { if (INS == 1) atomicOr(&g_fuse_lds[FB_LDS_NEXT + w2v[u][k]], bit2v[u][k]); else atomicOr(&Mn[w2v[u][k]], bit2v[u][k]); }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) { idv[u] = nidv[u]; byv[u] = nbyv[u]; }
  }
  wave_sync();
  return wave_sum(alive);
}

// chunk t of a posted pass (co_exec): positions [t CO_FB_CHUNK, (t + 1) CO_FB_CHUNK) of one side.  The wavefront that runs it may be
// any wavefront between two cases: the compact lookup entries go into ITS copy of the LDS table first.
// (from which list length a pass is posted, and its chunks: KParams co_fb_min, co_fb_chunk)
__device__ __noinline__ void fb_pass_chunk(const EH_G CoJob* j, uint32_t t) {
  const uint32_t l = (uint32_t)EH_LANE;
  cbptr S = (cbptr)uni64(j->a[0]); const uint32_t len = uni((uint32_t)j->a[1]), g = uni((uint32_t)(j->a[1] >> 32));
  wptr ids = (wptr)uni64(j->a[2]); const EH_G RkWord* RK = (const EH_G RkWord*)uni64(j->a[3]); const EH_G ScEnt* SC = (const EH_G ScEnt*)uni64(j->a[4]);
  const uint32_t kill = uni((uint32_t)j->a[5]), e_pos = uni((uint32_t)(j->a[5] >> 32)), fl = uni((uint32_t)j->a[6]), nn = uni((uint32_t)j->a[9]);
  wptr Mn = (wptr)uni64(j->a[7]), N1n = (wptr)uni64(j->a[8]);
  const bool ek = fl & 1u, sc_lds = (fl & 2u) != 0; const uint32_t ins = (fl >> 8) & 7u;
  if (sc_lds) { cwptr scw = (cwptr)SC; lanes_sync(); for (uint32_t i = l; i < 2u * nn; i += 64) g_fuse_lds[i] = scw[i]; lanes_sync(); }
  // the chunk's positions count from 0: position p of the chunk is p0 + p of the list, and everything fb_pass asks about a position
  // (p + k + g < len, the member at len-1) is a difference to len or e_pos, which move with it
  const uint32_t chunk = uni((uint32_t)j->a[10]);
  const uint32_t p0 = t * chunk, ln = len - p0, plim = ln < chunk ? ln : chunk;
  const uint32_t ep = (ek && e_pos != FB_DEAD && e_pos >= p0) ? e_pos - p0 : FB_DEAD;
  S += p0; ids += p0;
  const uint32_t alive = ins == 3 ? fb_pass<3>(S, ln, ids, g, RK, SC, sc_lds, kill, ep, plim, Mn, N1n)
                       : ins == 2 ? fb_pass<2>(S, ln, ids, g, RK, SC, sc_lds, kill, ep, plim, Mn, nullptr)
                       : fb_pass<0>(S, ln, ids, g, RK, SC, sc_lds, kill, ep, plim, Mn, nullptr);
  if (l == 0 && alive) atomicAdd((EH_G unsigned long long*)&j->acc, (unsigned long long)alive);
}

// members of `node`: how many, and the position of the k-th (0-based, ascending)
__device__ __noinline__ uint32_t fb_count(cwptr ids, uint32_t len, uint32_t node) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint32_t c = 0;
  for (uint32_t base = 0; base < len; base += 1024) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { uint32_t p = base + 256u * u + 4u * l; uint4 z; z.x = z.y = z.z = z.w = FB_DEAD; v[u] = z; if (p < len) v[u] = ldg16a(ids + p); }
#pragma unroll
    for (int u = 0; u < 4; u++) c += (v[u].x == node) + (v[u].y == node) + (v[u].z == node) + (v[u].w == node);
  }
  return wave_sum(c);
}
__device__ __noinline__ uint32_t fb_find(cwptr ids, uint32_t len, uint32_t node, uint32_t k) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint32_t before = 0;
  for (uint32_t base = 0; base < len; base += 1024) {
    uint4 v[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { uint32_t p = base + 256u * u + 4u * l; uint4 z; z.x = z.y = z.z = z.w = FB_DEAD; v[u] = z; if (p < len) v[u] = ldg16a(ids + p); }
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

// The member at len-1 of a round: records its (node, byte') in the next-byte tables of the current generation and tells
// whether another member had put it there already (then it is not alone in its group).
EH_DEV uint32_t fb_end_member(wptr M, wptr N1, uint32_t id, uint32_t b) {
  uint32_t was = 0;
  if (EH_LANE == 0) {
    const uint32_t w = id * 8u + (b >> 5), bit = 1u << (b & 31u), key = b + 1u;
    if (!N1) was = (atomicOr(&M[w], bit) & bit) ? 1u : 0u;
    else {
      const uint32_t n1 = fb_ld(&N1[id]);
      if (n1 == key) was = 1u;
      else if (n1 == 0) (void)atomicExch(&N1[id], key);
      else if (n1 == FB_MULTI) was = (atomicOr(&M[w], bit) & bit) ? 1u : 0u;
      else { (void)atomicOr(&M[id * 8u + ((n1 - 1u) >> 5)], 1u << ((n1 - 1u) & 31u)); (void)atomicOr(&M[w], bit); (void)atomicExch(&N1[id], FB_MULTI); }
    }
  }
  return uni(was);
}

// find_jump_points/2 + any_position_pair/1: *from / *tpos = the positions jump/3 (:47-50) cuts at.  Same draws, same fuel,
// same work accounting as fuse_lists' node-list version (eh_fuse.h).  false: work area exhausted / budget.
__device__ __noinline__ bool fuse_jump_stream(Ctx&, cbptr A, uint32_t la, cbptr B, uint32_t lb, bool sym, uint32_t* from, uint32_t* tpos, uint32_t* rounds) {
  EH_CTX;
  const uint32_t l = (uint32_t)EH_LANE;
  wptr ids[2] = {nullptr, nullptr};
  wptr M[2] = {nullptr, nullptr};
  wptr N1[2] = {nullptr, nullptr};
  EH_G RkWord* RK = nullptr; EH_G ScEnt* SC = nullptr;
  uint32_t rk_rows = 0, m_rows = 0;                                // nodes RK + SC / M[] + N1[] have room for
  cbptr S[2] = {A, B};
  const uint32_t len[2] = {la, lb};
  const int nside = sym ? 1 : 2;
  uint32_t nn = 1, g = 0;                                          // nodes of generation g
  int64_t fuel = 100000;                                           // ?SEARCH_FUEL
  uint64_t gen_entries = (uint64_t)la + lb;
  bool have_bits = false;                                          // M holds the next bytes of generation g
  bool cur_n1 = false;                                             // ... in two levels (N1 + rows of the nodes with several bytes)
  // lists of CO_FB_MIN positions and more: the streaming passes are posted for several wavefronts (co_run), so the next-byte tables stay in HBM
  const uint32_t co_min = c.p->co_fb_min, co_chunk = c.p->co_fb_chunk;
  const bool coop = c.p->board != nullptr && (la >= co_min || lb >= co_min);
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
    // everything for 100 000 nodes up front made every fuse of a block of 100 KB and more ask for 13 to 21 MB.  The
    // next-byte tables (M, N1) are zero when they are allocated and fb_scan leaves zero what it has read.
    if (!ids[0]) {
      for (int s = 0; s < nside; s++) { ids[s] = (wptr)ws_alloc(c, ((uint64_t)len[s] + 8) * 4); if (!ids[s]) return false; }
    }
    if (nn > rk_rows) {
      rk_rows = nn < 1024u ? 1024u : nn;
      RK = (EH_G RkWord*)ws_alloc(c, ((uint64_t)rk_rows + 4) * 64);
      SC = (EH_G ScEnt*)ws_alloc(c, ((uint64_t)rk_rows + 4) * 8);
      if (!RK || !SC) return false;
    }
    if (nn > m_rows) {                                             // (only generation 0 comes here: later tables are sized below)
      m_rows = nn < 1024u ? 1024u : nn;
      for (int s = 0; s < nside; s++) {
        M[s] = (wptr)ws_alloc(c, ((uint64_t)m_rows + 4) * 32); N1[s] = (wptr)ws_alloc(c, ((uint64_t)m_rows + 8) * 4);
        if (!M[s] || !N1[s]) return false;
        fb_clear(M[s], (m_rows + 4) * 8u); fb_clear(N1[s], (m_rows + 8u) & ~3u);
      }
      wave_sync();
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
      e_alone[s] = fb_end_member(M[s], cur_n1 ? N1[s] : nullptr, id, b) ? 0u : 1u;
    }
    wave_sync();
    EH_PT(c, 102);
    // ---- children
    const bool forced = !sym && e_pos[0] != FB_DEAD && e_alone[0];
    uint32_t sp = FB_DEAD, nfull = 0;
    uint32_t nchild = fb_scan(M[0], sym ? nullptr : M[1], cur_n1 ? N1[0] : nullptr, (cur_n1 && !sym) ? N1[1] : nullptr, nwords,
                              forced ? e_w[0] : FB_DEAD, forced ? e_bit[0] : 0u, RK, SC, &sp, &nfull);
    EH_PT(c, 103);
    if (nchild == 0) break;                                        // NoDesp =:= [] -> any_position_pair(Nodes)
    // ---- commit: ids of generation g+1 (+ the next-byte tables of the next round when it is going to run)
    fuel -= (int64_t)nchild;
    const bool next = fuel >= 0 && (uint32_t)(rng_peek(c.rng, 1) * 8.0) != 0;
    const bool lds = next && nchild * 8u <= FB_LDS_WORDS && !coop;
    // two-level tables when the nodes of this round had two children on average at most: low-entropy data, where the node
    // counts stay small over many rounds; lookups through the compact entries when few nodes spread over several words
    const bool n1 = next && !lds && nchild <= 2u * nn;
    const bool n1_lds = n1 && nchild <= FB_LDS_N1_NODES && !coop;
    const EH_G ScEnt* look = nfull * 4u <= nn ? SC : nullptr;
    const bool sc_lds = look && nn <= FB_LDS_SC_NODES;
    if (sc_lds) { cwptr scw = (cwptr)SC; lanes_sync(); for (uint32_t i = l; i < 2u * nn; i += 64) g_fuse_lds[i] = scw[i]; lanes_sync(); }
    if (next && nchild > m_rows) {                                 // the next generation's tables (this one's are in RK now)
      m_rows = nchild;
      for (int s = 0; s < nside; s++) {
        M[s] = (wptr)ws_alloc(c, ((uint64_t)m_rows + 4) * 32); N1[s] = (wptr)ws_alloc(c, ((uint64_t)m_rows + 8) * 4);
        if (!M[s] || !N1[s]) return false;
        fb_clear(M[s], (m_rows + 4) * 8u); fb_clear(N1[s], (m_rows + 8u) & ~3u);
      }
      wave_sync();
    }
    uint64_t entries = 0;
    for (int s = 0; s < nside; s++) {
      if (lds) { for (uint32_t i = l; i < nchild * 8u; i += 64) g_fuse_lds[FB_LDS_NEXT + i] = 0; lanes_sync(); }
      if (n1_lds) { for (uint32_t i = l; i < nchild; i += 64) g_fuse_lds[FB_LDS_NEXT + i] = 0; lanes_sync(); }
      EH_PT(c, 104);
      // the member at len-1: leaves when it was inserted first (alone, or an odd generation).  One list on both sides:
      // alone = the special node = the member itself, now empty, so it stays.
      bool ek = sym ? (!e_alone[0] && (g & 1u)) : (e_alone[s] || (g & 1u));
      uint32_t kill = (s == 1 && forced) ? sp : FB_DEAD;
      const int ins = lds ? 1 : n1_lds ? 4 : n1 ? 3 : (next && nchild * 8u <= FB_TEST_WORDS) ? 2 : 0;
      wptr mn = (ins == 0 && !next) ? nullptr : M[s]; wptr n1n = (ins == 3 || ins == 4) ? N1[s] : nullptr;
      uint32_t alive = 0;
      bool posted = false;
      if (coop && len[s] >= co_min) {                               // the pass as chunks for several wavefronts (tables in HBM: ins is 0, 2 or 3 here)
        CoArgs a; for (uint32_t k = 0; k < CO_ARGS; k++) a.a[k] = 0;
        a.a[0] = (uint64_t)S[s]; a.a[1] = (uint64_t)len[s] | ((uint64_t)g << 32); a.a[2] = (uint64_t)ids[s]; a.a[3] = (uint64_t)RK; a.a[4] = (uint64_t)look;
        a.a[5] = (uint64_t)kill | ((uint64_t)e_pos[s] << 32); a.a[6] = (ek ? 1u : 0u) | ((uint32_t)ins << 8) | (sc_lds ? 2u : 0u); a.a[7] = (uint64_t)mn; a.a[8] = (uint64_t)n1n; a.a[9] = nn; a.a[10] = co_chunk;
        unsigned long long acc = 0;
        posted = co_run(CO_FBPASS, (len[s] + co_chunk - 1) / co_chunk, a, &acc);
        alive = (uint32_t)acc;
      }
      if (!posted)
        alive = ins == 1 ? fb_pass<1>(S[s], len[s], ids[s], g, RK, look, sc_lds, kill, ek ? e_pos[s] : FB_DEAD, len[s], mn, nullptr)
              : ins == 4 ? fb_pass<4>(S[s], len[s], ids[s], g, RK, look, sc_lds, kill, ek ? e_pos[s] : FB_DEAD, len[s], mn, n1n)
              : ins == 3 ? fb_pass<3>(S[s], len[s], ids[s], g, RK, look, sc_lds, kill, ek ? e_pos[s] : FB_DEAD, len[s], mn, n1n)
              : ins == 2 ? fb_pass<2>(S[s], len[s], ids[s], g, RK, look, sc_lds, kill, ek ? e_pos[s] : FB_DEAD, len[s], mn, nullptr)
              : fb_pass<0>(S[s], len[s], ids[s], g, RK, look, sc_lds, kill, ek ? e_pos[s] : FB_DEAD, len[s], mn, nullptr);
      entries += alive;
      EH_PT(c, lds ? 105 : (n1_lds ? 96 : (n1 ? 111 : (next ? (nchild * 8u <= FB_TEST_WORDS ? 110 : 106) : 107))));
      if (lds) { lanes_sync(); for (uint32_t i = l; i < nchild * 8u; i += 64) M[s][i] = g_fuse_lds[FB_LDS_NEXT + i]; wave_sync(); }
      if (n1_lds) { lanes_sync(); for (uint32_t i = l; i < nchild; i += 64) N1[s][i] = g_fuse_lds[FB_LDS_NEXT + i]; wave_sync(); }
    }
    if (sym) entries *= 2; else if (forced) entries += 2;
    gen_entries = entries;
    special_node = forced ? sp : FB_DEAD;
    nn = nchild; g++; have_bits = next; cur_n1 = n1;
    EH_PT(c, 108);
    (*rounds)++;
  }
  // any_position_pair/1 (:73-77); odd generations are stored reversed
  const uint32_t par = g & 1u;
  uint32_t ni = rng_rand(c.rng, nn);
  uint32_t node = par ? nn - 1 - ni : ni;
  if (c.fp_on) {                                                   // eh_fuse_red.h: name the node, the members come from the original lists
    c.fp_g = g; c.fp_special = 0; c.fp_keypos = 0; c.fp_bA = 0; c.fp_bB = 0;
    if (g == 0) return true;
    if (node == special_node) { c.fp_special = 1; return true; }
    const uint32_t fcn = fb_count(ids[0], la, node);
    const uint32_t m0 = fb_find(ids[0], la, node, 0), ml = fb_find(ids[0], la, node, fcn - 1u);
    c.fp_keypos = m0;
    c.fp_special = (fcn == 1u && m0 + g == la) ? 1u : 0u;
    c.fp_bA = (ml + g == la) ? 1u : 0u;
    if (sym) c.fp_bB = c.fp_bA;
    else { const uint32_t tcn = fb_count(ids[1], lb, node); c.fp_bB = (tcn > 0 && fb_find(ids[1], lb, node, tcn - 1u) + g == lb) ? 1u : 0u; }
    EH_PT(c, 109);
    return true;
  }
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
