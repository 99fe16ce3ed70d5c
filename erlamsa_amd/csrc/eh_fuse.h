// eh_fuse.h — device code for erlamsa_fuse:fuse/2 (erlamsa_fuse.erl:47-134) and the mutators
// ft, fn, fo (erlamsa_mutations.erl:380-427).
//
// See fuse_round() below for how find_jump_points/2 is run on one wavefront.
#pragma once
#include "eh_field.h"

namespace eh {

// ---------------------------------------------------------------------------------------------
// find_jump_points/2 as an MSD radix refinement, one byte per round, in ONE ordered pass per round.
//
// A node is {fo, fc, to, tc}: its source suffixes F[fo, fo+fc) and target suffixes T[to, to+tc) (start positions;
// len = the empty suffix).  The reference builds every child list and the new node list by prepending, so its
// lists come out reversed every round; here the arrays are kept in REFERENCE order on even generations and in
// REVERSED reference order on odd ones, which turns every round into the same stable operation:
//   for the nodes in array order, for the next bytes in ascending (even -> odd) or descending (odd -> even)
//   order: child = {source entries with that byte, target entries with that byte}, in array order.
// (fix_empty_list/1 :58-60 drops the entry whose rest is [] when it is inserted FIRST into its group, i.e. when it
// is the group's first member in reference order: first in array order on even generations, last on odd ones.)
// Three paths, chosen per run of nodes, all writing children in node order behind running counters:
//   * a run of {1,1} nodes — the steady state on most data: one node per lane, two byte loads and a compare;
//   * whole nodes packed into <= 64 source and <= 64 target entries: one entry per lane, register bitonic sort by
//     (node, byte), groups from neighbour compares, source/target groups joined by shuffle binary search;
//   * a node with more than 64 entries on a side: 256-bin LDS histograms + ballot-matched stable scatter.
// The PRNG draws (one rand(8) per round, rand_elem x3 at the end) are the reference's.
// ---------------------------------------------------------------------------------------------
// fc == 1: fo IS the suffix position (nothing is stored in F for it); fc > 1: fo = offset of the fc positions in F.
// Likewise to / tc / T.  Most nodes are {1,1} after a round or two and then cost one 16-byte load and store per round.
struct FNode { uint32_t fo, fc, to, tc; };
// big-node path: [0,256) source count -> cursor, [256,512) target, [512,768) child index of the bin, [768,1024) flags
// (g_fuse_lds, EH_FUSE_LDS_WORDS words: eh_device.h; eh_fuse2.h: bitmaps of <= EH_FUSE_LDS_WORDS / 8 nodes)

// ascending bitonic sort of one 32-bit key per lane
EH_DEV uint32_t wave_sort64(uint32_t key) {
  const uint32_t l = (uint32_t)EH_LANE;
#pragma unroll
  for (uint32_t k2 = 2; k2 <= 64; k2 <<= 1) {
#pragma unroll
    for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
      uint32_t other = (uint32_t)__shfl_xor((int)key, (int)j);
      bool up = (l & k2) == 0 || k2 == 64, lower = (l & j) == 0;
      bool take_min = lower == up;
      uint32_t mn = key < other ? key : other, mx = key < other ? other : key;
      key = take_min ? mn : mx;
    }
  }
  return key;
}
// number of leading lanes (of the first cnt) whose ascending value v is <= x  (v held one per lane)
EH_DEV uint32_t lanes_le(uint32_t v, uint32_t cnt, uint32_t x) {
  uint32_t lo = 0;
#pragma unroll
  for (uint32_t s = 32; s > 0; s >>= 1) { uint32_t t = (uint32_t)__shfl((int)v, (int)((lo + s - 1) & 63)); if (lo + s <= cnt && t <= x) lo += s; }
  { uint32_t t = (uint32_t)__shfl((int)v, (int)(lo & 63)); if (lo < cnt && t <= x) lo++; }   // (the shuffle itself must not diverge)
  return lo;
}
EH_DEV uint32_t lanes_lt(uint32_t v, uint32_t cnt, uint32_t x) { uint32_t r = lanes_le(v, cnt, x ? x - 1 : 0u); return x == 0 ? 0u : r; }

EH_DEV uint64_t wave_sum64(uint64_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v += ((uint64_t)(uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)v, d);
  return uni64(v);
}
}  // namespace eh
#include "eh_fuse_lds.h"
namespace eh {
#define FUSE_UNROLL 4
struct FuseGen { EH_G FNode* nd; wptr F; wptr T; };

// One refinement round over all nodes of `g` (generation parity par) into `o`; returns the number of children.
// sym: fuse(H, H) — the target lists are the source lists at every level, only one side is computed.
EH_DEV uint32_t fuse_round(cbptr A, uint32_t la, cbptr B, uint32_t lb, const FuseGen& g, uint32_t nn, const FuseGen& o, uint32_t par, bool sym, uint64_t* entries) {
  const int l = EH_LANE;
  const bool asc = par == 0;
  uint64_t le = 0;                                               // list members of the children made by this lane (work accounting)
  uint32_t cn = 0, cf = 0, ct = 0;                               // children / F entries / T entries written so far
  uint32_t k0 = 0;
  while (k0 < nn) {
    uint32_t k = k0 + (uint32_t)l;
    FNode nd = {0, 0, 0, 0};
    if (k < nn) nd = g.nd[k];
    if (sym) { nd.to = nd.fo; nd.tc = nd.fc; }
    bool valid = k < nn;
    bool big = valid && (nd.fc > 64 || nd.tc > 64);
    bool one = valid && nd.fc == 1 && nd.tc == 1;
    unsigned long long vm = __ballot(valid), bm = __ballot(big), om = __ballot(one);
    uint32_t nvalid = (uint32_t)__popcll(vm);
    uint32_t nones = ~om == 0ull ? 64u : (uint32_t)__builtin_ctzll(~om);
    if (nones > nvalid) nones = nvalid;
    if (nones >= 16 || (nones > 0 && nones == nvalid)) {
      // ---- run of {1,1} nodes: positions are in the descriptor
      bool in = (uint32_t)l < nones;
      uint32_t p = in ? nd.fo : la, q = in ? nd.to : lb;
      bool special = in && p == la - 1;                          // its only member is dropped: {[[]], [[]]}
      bool same;
      if (sym) same = in && p < la;                              // q == p: the bytes are the same byte
      else { uint32_t ba = (in && p < la) ? A[p] : 256u, bb = (in && q < lb) ? B[q] : 257u; same = ba == bb; }
      bool child = special || same;
      bool tkeep = child && !special && q != lb - 1;             // a target whose rest is [] is dropped as well
      unsigned long long cm = __ballot(child);
      uint32_t ci = (uint32_t)__popcll(cm & ((1ull << l) - 1));
      if (child) {
        FNode c2; c2.fo = special ? la : p + 1; c2.fc = 1; c2.to = special ? lb : q + 1; c2.tc = (special || tkeep) ? 1u : 0u;
        o.nd[cn + ci] = c2;
        le += 1u + c2.tc;
      }
      cn += (uint32_t)__popcll(cm);
      k0 += nones;
      continue;
    }
    if (bm & 1ull) {
      // ---- one big node, the whole wave: LDS histograms of the next byte on both sides
      FNode b0; b0.fo = uni(nd.fo); b0.fc = uni(nd.fc); b0.to = uni(nd.to); b0.tc = uni(nd.tc);
      uint32_t* hf = g_fuse_lds; uint32_t* ht = sym ? g_fuse_lds : g_fuse_lds + 256;
      uint32_t* hc = g_fuse_lds + 512; uint32_t* hk = g_fuse_lds + 768;
      for (uint32_t i = l; i < 512; i += 64) g_fuse_lds[i] = 0;
      lanes_sync();
      // the entry whose rest is []: its byte' and how many members of its group precede it (array order)
      uint32_t fdb = 0xFFFFFFFFu, fdbefore = 0, tdb = 0xFFFFFFFFu, tdbefore = 0;
      for (int side = 0; side < (sym ? 1 : 2); side++) {
        cbptr S = side ? B : A; uint32_t slen = side ? lb : la;
        uint32_t cnt = side ? b0.tc : b0.fc;
        cwptr P = side ? g.T + b0.to : g.F + b0.fo;
        uint32_t single = side ? b0.to : b0.fo;                  // cnt == 1: the position itself
        uint32_t* h = side ? ht : hf;
        for (uint32_t base = 0; base < cnt; base += 64 * FUSE_UNROLL) {      // FUSE_UNROLL chunks in flight: the loads are the cost
          uint32_t pp[FUSE_UNROLL], bt[FUSE_UNROLL]; bool v[FUSE_UNROLL];
#pragma unroll
          for (int u = 0; u < FUSE_UNROLL; u++) {
            uint32_t i = base + 64u * (uint32_t)u + (uint32_t)l; bool in = i < cnt;
            pp[u] = in ? (cnt == 1 ? single : P[i]) : slen; v[u] = in && pp[u] < slen;
          }
#pragma unroll
          for (int u = 0; u < FUSE_UNROLL; u++) { bt[u] = v[u] ? (uint32_t)S[pp[u]] : 0u; if (!asc) bt[u] = 255u - bt[u]; }
#pragma unroll
          for (int u = 0; u < FUSE_UNROLL; u++) {
            unsigned long long lastm = __ballot(v[u] && pp[u] == slen - 1);
            if (lastm) {                                         // rare: at most once per side and round
              int j = (int)__builtin_ctzll(lastm);
              uint32_t bj = (uint32_t)__builtin_amdgcn_readlane((int)bt[u], j);
              lanes_sync();
              uint32_t before = h[bj] + (uint32_t)__popcll(__ballot(v[u] && bt[u] == bj) & ((1ull << j) - 1));
              if (side) { tdb = bj; tdbefore = before; } else { fdb = bj; fdbefore = before; }
              lanes_sync();
            }
            if (v[u]) atomicAdd(&h[bt[u]], 1u);
          }
        }
      }
      if (sym) { tdb = fdb; tdbefore = fdbefore; }
      lanes_sync();
      // bins in output order: lane l owns byte' 4l .. 4l+3
      uint32_t rf[4], rt[4], ef[4], et[4]; bool ch[4], sp[4];
      uint32_t lc = 0, lf = 0, lt2 = 0;
      bool anyfd = false, anytd = false;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t bt = 4u * (uint32_t)l + (uint32_t)j;
        rf[j] = hf[bt]; rt[j] = ht[bt];
        bool fdrop = bt == fdb && (asc ? fdbefore == 0 : fdbefore + 1 == rf[j]);
        bool tdrop = bt == tdb && (asc ? tdbefore == 0 : tdbefore + 1 == rt[j]);
        anyfd |= fdrop; anytd |= tdrop;
        ef[j] = rf[j] - (fdrop ? 1u : 0u); et[j] = rt[j] - (tdrop ? 1u : 0u);
        sp[j] = rf[j] > 0 && ef[j] == 0;
        ch[j] = rf[j] > 0 && (sp[j] || rt[j] > 0);
        if (sp[j]) { ef[j] = 1; et[j] = 1; }
        if (!ch[j]) { ef[j] = 0; et[j] = 0; }
        le += ef[j] + et[j];
        lc += ch[j] ? 1u : 0u; lf += ef[j] > 1 ? ef[j] : 0u; lt2 += et[j] > 1 ? et[j] : 0u;   // only lists of > 1 go to F / T
      }
      const bool fdropped = __ballot(anyfd) != 0, tdropped = __ballot(anytd) != 0;   // the entry with rest [] leaves its group
      uint32_t ic = wave_incl_scan(lc), iff = wave_incl_scan(lf), it = wave_incl_scan(lt2);
      uint32_t oc = cn + ic - lc, of = cf + iff - lf, ot = sym ? of : ct + it - lt2;
      lanes_sync();
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t bt = 4u * (uint32_t)l + (uint32_t)j;
        bool scat = ch[j] && !sp[j];
        hf[bt] = scat ? of : 0xFFFFFFFFu; if (!sym) ht[bt] = scat ? ot : 0xFFFFFFFFu;
        hc[bt] = scat ? oc : 0xFFFFFFFFu; hk[bt] = (ef[j] == 1 ? 1u : 0u) | (et[j] == 1 ? 2u : 0u);
        if (ch[j]) {
          FNode c2; c2.fo = sp[j] ? la : of; c2.fc = ef[j]; c2.to = sp[j] ? lb : ot; c2.tc = et[j];
          o.nd[oc] = c2;                                         // a single member's position is filled in by the scatter below
          oc++; if (ef[j] > 1) of += ef[j]; if (sym) ot = of; else if (et[j] > 1) ot += et[j];
        }
      }
      cn += (uint32_t)__builtin_amdgcn_readlane((int)ic, 63); cf += (uint32_t)__builtin_amdgcn_readlane((int)iff, 63); ct += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
      wave_sync();                                               // the child descriptors are patched below
      // stable scatter: rank among the lanes of this chunk with the same byte', behind the bin's cursor
      for (int side = 0; side < (sym ? 1 : 2); side++) {
        cbptr S = side ? B : A; uint32_t slen = side ? lb : la;
        uint32_t cnt = side ? b0.tc : b0.fc;
        cwptr P = side ? g.T + b0.to : g.F + b0.fo;
        uint32_t single = side ? b0.to : b0.fo;
        uint32_t* h = side ? ht : hf; wptr O = side ? o.T : o.F;
        const bool dropped = side ? tdropped : fdropped;
        for (uint32_t base = 0; base < cnt; base += 64 * FUSE_UNROLL) {
          uint32_t ppu[FUSE_UNROLL], btu[FUSE_UNROLL]; bool vu[FUSE_UNROLL];
#pragma unroll
          for (int u = 0; u < FUSE_UNROLL; u++) {
            uint32_t i = base + 64u * (uint32_t)u + (uint32_t)l; bool in = i < cnt;
            ppu[u] = in ? (cnt == 1 ? single : P[i]) : slen; vu[u] = in && ppu[u] < slen;
          }
#pragma unroll
          for (int u = 0; u < FUSE_UNROLL; u++) { btu[u] = vu[u] ? (uint32_t)S[ppu[u]] : 0u; if (!asc) btu[u] = 255u - btu[u]; }
#pragma unroll
          for (int u = 0; u < FUSE_UNROLL; u++) {
            const uint32_t pp = ppu[u], bt = btu[u]; const bool v = vu[u];
            if (base + 64u * (uint32_t)u >= cnt) break;
            bool isdrop = dropped && v && pp == slen - 1;          // positions are distinct: at most one such entry
            bool vv = v && !isdrop;
            unsigned long long eq = __ballot(vv);
#pragma unroll
            for (int bit = 0; bit < 8; bit++) { unsigned long long m = __ballot(vv && ((bt >> bit) & 1u)); eq &= ((bt >> bit) & 1u) ? m : ~m; }
            uint32_t rank = (uint32_t)__popcll(eq & ((1ull << l) - 1)), cntg = (uint32_t)__popcll(eq);
            uint32_t off = vv ? h[bt] : 0xFFFFFFFFu, cidx = vv ? hc[bt] : 0xFFFFFFFFu, kf = vv ? hk[bt] : 0u;
            lanes_sync();
            if (vv && cidx != 0xFFFFFFFFu) {
              bool isone = (kf >> side) & 1u;
              if (isone) { if (side) o.nd[cidx].to = pp + 1; else { o.nd[cidx].fo = pp + 1; if (sym) o.nd[cidx].to = pp + 1; } }
              else { O[off + rank] = pp + 1; if (rank == 0) h[bt] = off + cntg; }
            }
            lanes_sync();
          }
        }
      }
      k0 += 1;
      continue;
    }
    // ---- whole nodes packed into <= 64 + 64 entries, one entry per lane
    uint32_t cumf = wave_incl_scan(valid && !big ? nd.fc : 0u), cumt = wave_incl_scan(valid && !big ? nd.tc : 0u);
    uint32_t firstbig = bm ? (uint32_t)__builtin_ctzll(bm) : 64u;
    unsigned long long fitm = __ballot(valid && (uint32_t)l < firstbig && cumf <= 64 && cumt <= 64);
    uint32_t nfit = ~fitm == 0ull ? 64u : (uint32_t)__builtin_ctzll(~fitm);     // cumulative sums ascend: a prefix
    uint32_t totf = (uint32_t)__builtin_amdgcn_readlane((int)cumf, (int)(nfit - 1)), tott = (uint32_t)__builtin_amdgcn_readlane((int)cumt, (int)(nfit - 1));
    uint32_t key[2], pos[2], tot[2] = {totf, tott};
#pragma unroll
    for (int side = 0; side < 2; side++) {
      if (side == 1 && sym) { key[1] = key[0]; pos[1] = pos[0]; break; }
      cbptr S = side ? B : A; uint32_t slen = side ? lb : la;
      uint32_t cum = side ? cumt : cumf, cntn = side ? nd.tc : nd.fc, offn = side ? nd.to : nd.fo;
      bool in = (uint32_t)l < tot[side];
      uint32_t nj = lanes_le(cum, nfit, (uint32_t)l);                          // node (of this batch) of entry l
      uint32_t ex = (uint32_t)__shfl((int)(cum - cntn), (int)(nj & 63)), ob = (uint32_t)__shfl((int)offn, (int)(nj & 63)), cj = (uint32_t)__shfl((int)cntn, (int)(nj & 63));
      uint32_t pp = slen;
      if (in) pp = cj == 1 ? ob : (side ? g.T : g.F)[ob + ((uint32_t)l - ex)];
      bool v = in && pp < slen;
      uint32_t bt = v ? (uint32_t)S[pp] : 0u; if (!asc) bt = 255u - bt;
      uint32_t kk = in ? (((nj << 9) | (v ? bt : 511u)) << 6) | (uint32_t)l : 0xFFFFFFFFu;
      uint32_t nxt = (uint32_t)__shfl_down((int)kk, 1);
      uint32_t sk = __ballot(l < 63 && kk > nxt) == 0 ? kk : wave_sort64(kk);
      key[side] = sk; pos[side] = (uint32_t)__shfl((int)pp, (int)(sk & 63u));
    }
    // groups = runs of equal (node, byte')
    uint32_t gk[2], lead[2], len[2], drops[2]; bool alive[2], isdrop[2];
#pragma unroll
    for (int side = 0; side < 2; side++) {
      if (side == 1 && sym) { gk[1] = gk[0]; lead[1] = lead[0]; len[1] = len[0]; drops[1] = drops[0]; alive[1] = alive[0]; isdrop[1] = isdrop[0]; break; }
      uint32_t slen = side ? lb : la;
      gk[side] = key[side] >> 6;
      alive[side] = (uint32_t)l < tot[side] && (gk[side] & 511u) != 511u;
      uint32_t prevk = (uint32_t)__shfl_up((int)gk[side], 1), nextk = (uint32_t)__shfl_down((int)gk[side], 1);
      bool leader = alive[side] && (l == 0 || prevk != gk[side]);
      bool last = alive[side] && (l == 63 || (uint32_t)l + 1 >= tot[side] || nextk != gk[side]);
      unsigned long long lm = __ballot(leader);
      uint32_t ll = alive[side] ? 63u - (uint32_t)__builtin_clzll(lm & ((2ull << l) - 1)) : 0u;
      unsigned long long after = lm & ~((2ull << ll) - 1);                     // next leader
      unsigned long long am = __ballot(alive[side]);
      uint32_t endl = after ? (uint32_t)__builtin_ctzll(after) : 64u;
      // dead entries (key byte 511) sit between nodes: the run ends at the next leader or at the first non-alive lane
      unsigned long long dead_after = ~am & ~((2ull << ll) - 1);
      uint32_t endd = dead_after ? (uint32_t)__builtin_ctzll(dead_after) : 64u;
      uint32_t e = endl < endd ? endl : endd;
      lead[side] = ll; len[side] = alive[side] ? e - ll : 0u;
      isdrop[side] = alive[side] && pos[side] == slen - 1 && (asc ? leader : last);
      unsigned long long dm = __ballot(isdrop[side]);
      unsigned long long runm = (e >= 64 ? ~0ull : ((1ull << e) - 1)) & ~((1ull << ll) - 1);
      drops[side] = alive[side] ? (uint32_t)__popcll(dm & runm) : 0u;
    }
    // join: the target run with my (node, byte'), if any
    uint32_t tl = sym ? lead[0] : lanes_lt(gk[1], tott, gk[0]);                 // first target lane with key >= mine
    uint32_t tk_at = (uint32_t)__shfl((int)gk[1], (int)(tl & 63)), tlen_at = (uint32_t)__shfl((int)len[1], (int)(tl & 63)), tdr_at = (uint32_t)__shfl((int)drops[1], (int)(tl & 63));
    bool tldrop = __shfl((int)(isdrop[1] ? 1 : 0), (int)(tl & 63)) != 0;       // the target run's leader is the dropped entry
    bool texists = alive[0] && tl < tott && tk_at == gk[0];
    bool fleader = alive[0] && lead[0] == (uint32_t)l;
    uint32_t fcnt = len[0] - drops[0];
    bool special = fleader && fcnt == 0;
    bool child = fleader && (special || texists);
    uint32_t nf = child ? (special ? 1u : fcnt) : 0u, nt = child ? (special ? 1u : (tlen_at - tdr_at)) : 0u;
    uint32_t af = nf > 1 ? nf : 0u, at = nt > 1 ? nt : 0u;                      // entries that go to the arrays
    uint32_t ic = wave_incl_scan(child ? 1u : 0u), iff = wave_incl_scan(af), it = wave_incl_scan(at);
    uint32_t oc = cn + ic - (child ? 1u : 0u), of = cf + iff - af, ot = sym ? of : ct + it - at;
    // a single surviving member goes into the descriptor: the run's first member that is not dropped
    uint32_t fsurv = (uint32_t)__shfl((int)pos[0], (int)(((uint32_t)l + (isdrop[0] ? 1u : 0u)) & 63));
    uint32_t tsurv = (uint32_t)__shfl((int)pos[1], (int)((tl + (tldrop ? 1u : 0u)) & 63));
    if (child) {
      FNode c2;
      c2.fc = nf; c2.tc = nt;
      c2.fo = special ? la : (nf == 1 ? fsurv + 1 : of);
      c2.to = special ? lb : (nt == 1 ? tsurv + 1 : (nt == 0 ? 0u : ot));
      o.nd[oc] = c2;
      le += nf + nt;
    }
    // source entries of multi-member children: behind their leader's offset
    {
      bool lmulti = __shfl((int)(child && !special && nf > 1 ? 1 : 0), (int)lead[0]) != 0;
      uint32_t lof = (uint32_t)__shfl((int)of, (int)lead[0]);
      bool ldrop = __shfl((int)(isdrop[0] ? 1 : 0), (int)lead[0]) != 0;        // asc: the leader itself was dropped
      if (alive[0] && lmulti && !isdrop[0]) o.F[lof + ((uint32_t)l - lead[0]) - (ldrop ? 1u : 0u)] = pos[0] + 1;
    }
    // target entries: find the source leader with my key
    if (!sym) {
      uint32_t sl = lanes_lt(gk[0], totf, gk[1]);
      uint32_t sk_at = (uint32_t)__shfl((int)gk[0], (int)(sl & 63));
      bool has = alive[1] && sl < totf && sk_at == gk[1];
      bool smulti = __shfl((int)(child && !special && nt > 1 ? 1 : 0), (int)(sl & 63)) != 0;
      uint32_t sot = (uint32_t)__shfl((int)ot, (int)(sl & 63));
      bool ldrop = __shfl((int)(isdrop[1] ? 1 : 0), (int)lead[1]) != 0;
      if (has && smulti && !isdrop[1]) o.T[sot + ((uint32_t)l - lead[1]) - (ldrop ? 1u : 0u)] = pos[1] + 1;
    }
    cn += (uint32_t)__builtin_amdgcn_readlane((int)ic, 63); cf += (uint32_t)__builtin_amdgcn_readlane((int)iff, 63); ct += (uint32_t)__builtin_amdgcn_readlane((int)it, 63);
    k0 += nfit;
  }
  *entries = wave_sum64(le);
  wave_sync();
  return cn;
}

__device__ bool fuse_jump_stream(Ctx& c, cbptr A, uint32_t la, cbptr B, uint32_t lb, bool sym, uint32_t* from, uint32_t* tpos, uint32_t* rounds);   // eh_fuse2.h
}  // namespace eh
#include "eh_fuse_red.h"
namespace eh {

// fuse(Al, Bl) -> new byte list in the work area
#ifdef EH_FUSE_INLINE
EH_DEV bool fuse_lists(Ctx& c, cbptr A, uint32_t la, cbptr B, uint32_t lb, bptr* out, uint32_t* outlen) {
#else
__device__ __noinline__ bool fuse_lists(Ctx&, cbptr A, uint32_t la, cbptr B, uint32_t lb, bptr* out, uint32_t* outlen) {
  EH_CTX;
#endif
  const int l = EH_LANE;
  if (la == 0) { *out = (bptr)B; *outlen = lb; return true; }   // fuse([], Bl) -> Bl
  if (lb == 0) { *out = (bptr)A; *outlen = la; return true; }
  uint64_t mark = c.ws_used;
  const bool sym = A == B && la == lb;                             // sed_fuse_this: fuse(Lst, Lst)
  EH_PT0;
  uint32_t from = la, tpos = lb;
  uint32_t prof_rounds = 0;
  c.fp_on = 0;
  if ((sym ? (uint64_t)la : (uint64_t)la + lb) > FL_NMAX && !c.work_budget && !(c.p->flags & EH_FLAG_FUSE_NO_REDUCE)) {
    // large lists: the same search on lists with the long periodic stretches cut short (eh_fuse_red.h)
    const uint32_t rd = fr_peek_rounds(c.rng);
    if (rd < 64) {
      const uint32_t R = rd + 2u;
      uint32_t la2 = la, lb2 = lb;
      cbptr A2 = fr_reduce(c, A, &la2, R);
      if (!A2) return false;
      cbptr B2 = A2;
      if (sym) lb2 = la2; else { B2 = fr_reduce(c, B, &lb2, R); if (!B2) return false; }
      if ((uint64_t)la2 + lb2 <= ((uint64_t)la + lb) / 4u * 3u) {
        EH_PT(c, 97);                                              // eh_result_prof 97: finding + making the cuts, calls that took them; 98: calls that found none
        c.fp_on = 1;
        bool ok;
        if ((sym ? (uint64_t)la2 : (uint64_t)la2 + lb2) <= FL_NMAX && !(c.p->flags & EH_FLAG_FUSE_NO_LDS)) ok = fuse_jump_lds(c, A2, la2, B2, lb2, sym, &from, &tpos, &prof_rounds);
        else ok = fuse_jump_stream(c, A2, la2, B2, lb2, sym, &from, &tpos, &prof_rounds);
        c.fp_on = 0;
        if (!ok) return false;
        EH_PT(c, 54);                                              // eh_result_prof 54: the search on the shortened lists; 55: the node's members found in the original lists
        // any_position_pair/1 (:73-77) over the ORIGINAL lists: the members of the node are the occurrences of its g-gram
        const uint32_t g = c.fp_g, par = g & 1u;
        from = la; tpos = lb;
        if (g == 0) { from = rng_rand(c.rng, la); tpos = rng_rand(c.rng, lb); }
        else if (c.fp_special) { (void)rng_rand(c.rng, 1); (void)rng_rand(c.rng, 1); }      // {[[]], [[]]}
        else {
          cbptr key = A2 + c.fp_keypos;
          const uint32_t limA = la > g ? la - g : 0u, limB = lb > g ? lb - g : 0u;
          const uint32_t fc0 = fr_occ(A, la, limA, key, g, FR_NONE, nullptr), fc = fc0 + c.fp_bA;
          uint32_t pos = 0;
          if (fc > 0) { uint32_t j = rng_rand(c.rng, fc); j = par ? fc - 1u - j : j; if (j < fc0) { (void)fr_occ(A, la, limA, key, g, j, &pos); from = uni(pos) + g; } }
          const uint32_t tc0 = sym ? fc0 : fr_occ(B, lb, limB, key, g, FR_NONE, nullptr), tc = tc0 + c.fp_bB;
          if (tc > 0) { uint32_t j = rng_rand(c.rng, tc); j = par ? tc - 1u - j : j; if (j < tc0) { (void)fr_occ(B, lb, limB, key, g, j, &pos); tpos = uni(pos) + g; } }
        }
        EH_PT(c, 55);
        goto jump;
      }
    }
    EH_PT(c, 98);
    c.ws_used = mark;                                              // (no cut worth it: the copies go)
  }
  if ((sym ? (uint64_t)la : (uint64_t)la + lb) <= FL_NMAX && !(c.p->flags & EH_FLAG_FUSE_NO_LDS)) {     // small lists: sorted suffix entries in LDS (eh_fuse_lds.h)
    if (!fuse_jump_lds(c, A, la, B, lb, sym, &from, &tpos, &prof_rounds)) { if (c.status == CASE_BUDGET) c.ws_used = mark; return false; }
  } else if ((uint64_t)la + lb >= c.p->fuse_stream_min) {          // large lists: position-indexed refinement (eh_fuse2.h)
    if (!fuse_jump_stream(c, A, la, B, lb, sym, &from, &tpos, &prof_rounds)) { if (c.status == CASE_BUDGET) c.ws_used = mark; return false; }
  } else {
    FuseGen g[2];
    for (int k = 0; k < 2; k++) {
      g[k].nd = (EH_G FNode*)ws_alloc(c, ((uint64_t)la + 4) * sizeof(FNode));     // every node owns >= 1 source entry
      g[k].F = (wptr)ws_alloc(c, ((uint64_t)la + 4) * 4);
      g[k].T = sym ? g[k].F : (wptr)ws_alloc(c, ((uint64_t)lb + 4) * 4);
      if (!g[k].nd || !g[k].F || !g[k].T) return false;
    }
    // find_jump_points (:103-107): one node with all non-empty suffixes of both lists
    if (la > 1) for (uint32_t i = l; i < la; i += 64) g[0].F[i] = i;
    if (lb > 1 && !sym) for (uint32_t i = l; i < lb; i += 64) g[0].T[i] = i;
    if (l == 0) { FNode n0; n0.fo = 0; n0.fc = la; n0.to = 0; n0.tc = lb; g[0].nd[0] = n0; }   // (a single suffix: position 0 = offset 0)
    uint32_t nn = 1, par = 0;
    int64_t fuel = 100000;                                           // ?SEARCH_FUEL
    uint64_t gen_entries = (uint64_t)la + lb;
    wave_sync();
    while (true) {                                                   // find_jump_points_loop (:115-128)
      if (fuel < 0) break;
      if (rng_rand(c.rng, 8) == 0) break;                            // ?SEARCH_STOP_IP
      if (c.work_budget) {                                           // optional engine guard: a round costs its list members
        c.work += 16ull * gen_entries;
        if (c.work > c.work_budget) { c.status = CASE_BUDGET; c.ws_used = mark; return false; }
      }
      uint32_t nchild = fuse_round(A, la, B, lb, g[par], nn, g[par ^ 1], par, sym, &gen_entries);
      if (nchild == 0) break;                                        // NoDesp =:= [] -> any_position_pair(Nodes)
      par ^= 1; nn = nchild;
      fuel -= (int64_t)nchild;
      prof_rounds++;
    }
    // any_position_pair/1 (:73-77); odd generations are stored reversed
    uint32_t ni = rng_rand(c.rng, nn);
    FNode nd = g[par].nd[par ? nn - 1 - ni : ni];
    uint32_t fo = uni(nd.fo), fc = uni(nd.fc), to = sym ? fo : uni(nd.to), tc = sym ? fc : uni(nd.tc);
    if (fc > 0) { uint32_t j = rng_rand(c.rng, fc); from = fc == 1 ? fo : uni(g[par].F[fo + (par ? fc - 1 - j : j)]); }
    if (tc > 0) { uint32_t j = rng_rand(c.rng, tc); tpos = tc == 1 ? to : uni(g[par].T[to + (par ? tc - 1 - j : j)]); }
  }
jump:
#ifdef EH_PROF
  {                                                                // slots 112..125: fuse calls by log2(la + lb), 126: rounds
    uint32_t tot = la + lb, b = 0; while ((256u << b) < tot && b < 13) b++;
    EH_PT(c, 112 + b);
    if (EH_LANE == 0) { atomicAdd(&c.p->prof[2 * 126], (unsigned long long)prof_rounds); atomicAdd(&c.p->prof[2 * 126 + 1], 1ull); }
  }
#endif
  c.ws_used = mark;                                                // release all tables
  // jump/3 (:47-50): Al up to From, then To
  uint32_t nl = from + (lb - tpos);
  bptr dst = ws_alloc(c, nl);
  if (!dst) return false;
  wave_copy(dst, A, from);
  wave_copy(dst + from, B + tpos, lb - tpos);
  wave_sync();
  *out = dst; *outlen = nl;
  return true;
}

struct FoState { uint64_t ptr; uint32_t len; uint32_t has; };

__device__ __noinline__ int muta_fuse(Ctx&, int fn, EH_G FoState* fo) {
  EH_CTX;
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  c.r_kind = R_SAME;
  bptr r; uint32_t rl;
  if (fn == M_FT) {                                               // sed_fuse_this :386-390
    if (!fuse_lists(c, H, L, H, L, &r, &rl)) return 0;
    int d = rng_delta(c.rng);
    c.r_kind = R_NEW; c.r_ptr = r; c.r_len = rl;
    return d;
  }
  uint32_t h1 = L / 2;                                            // erlamsa_utils:halve/1 :137-146
  if (fn == M_FN) {                                               // sed_fuse_next :393-402
    Blk nb = hb; bool have_next = c.cur + 1 < c.nb;
    if (have_next) nb = blk_load(c.bl, c.cur + 1);                // uncons(T, H): next block or H itself
    bptr abl; uint32_t abll;
    if (!fuse_lists(c, H, h1, (cbptr)nb.ptr, nb.len, &abl, &abll)) return 0;
    if (!fuse_lists(c, abl, abll, H + h1, L - h1, &r, &rl)) return 0;
    int d = rng_delta(c.rng);
    c.r_kind = R_NEW; c.r_ptr = r; c.r_len = rl; c.r_flush = 1; c.r_drop_next = have_next ? 1 : 0;
    return d;
  }
  // sed_fuse_old / remember/1 :405-427
  uint32_t has = uni(fo->has);
  uint64_t optr = has ? uni64(fo->ptr) : hb.ptr; uint32_t olen = has ? uni(fo->len) : L;
  uint32_t o1 = olen / 2;
  bptr a; uint32_t al; bptr b; uint32_t bl;
  if (!fuse_lists(c, H, h1, (cbptr)optr, o1, &a, &al)) return 0;              // a -> o
  if (!fuse_lists(c, (cbptr)optr + o1, olen - o1, H + h1, L - h1, &b, &bl)) return 0;   // o -> a
  uint32_t swap = rng_rand(c.rng, 3);
  int d = rng_delta(c.rng);
  if (EH_LANE == 0) { if (!has || swap == 0) { fo->ptr = hb.ptr; fo->len = L; } fo->has = 1; }
  wave_sync();
  // flush_bvecs(A, flush_bvecs(B, T)): two flushed regions — build them contiguously as A-chunks then B-chunks
  c.r_kind = R_NEW; c.r_ptr = a; c.r_len = al; c.r_flush = 1;
  c.r2_ptr = b; c.r2_len = bl; c.r2 = 1;
  return d;
}

}  // namespace eh
