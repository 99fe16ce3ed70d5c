// eh_tree.h — device code for the guessed-parse-tree mutators tr2, td, ts1, ts2, tr
// (erlamsa_mutations.erl:787-1023).
//
// partial_parse/1 + grow/3 build nested lists out of matched delimiter pairs ()[]<>{}""''.  Every
// list node is a contiguous byte range [open, close] of the block, unclosed openers stay plain
// bytes, and completed nodes are properly nested.  So the "tree" is kept as a table of matched
// pairs sorted by opening position (= pre-order); node equality (Erlang =:= on the nested lists)
// is byte equality of the ranges; sublists/1 (reverse pre-order) is that table read backwards; and
// edit_sublist/3 ("first equal node per level, the rest of that level is left alone") is a single
// pre-order sweep with a skip pointer.
#pragma once
#include "eh_lex.h"

namespace eh {

struct TNode { uint32_t open, close, pend, pad; };      // pend = one past the close of the nearest enclosing NODE (level end)

EH_DEV uint32_t usual_delim_close(uint32_t c) {                 // usual_delims/1 :791-798
  switch (c) { case 40: return 41; case 91: return 93; case 60: return 62; case 123: return 125; case 34: return 34; case 39: return 39; }
  return 0;
}
// delimiter code planes: 1 ( 2 ) 3 [ 4 ] 5 < 6 > 7 { 8 } 9 " 10 '  (0 = plain byte)
struct DelimCls {
  EH_DEV uint32_t operator()(uint32_t b) const {
    switch (b) { case 40: return 1; case 41: return 2; case 91: return 3; case 93: return 4; case 60: return 5; case 62: return 6;
                 case 123: return 7; case 125: return 8; case 34: return 9; case 39: return 10; }
    return 0;
  }
};
EH_DEV uint32_t delim_close_code(uint32_t code) { return code == 9 ? 9u : (code == 10 ? 10u : code + 1); }   // opener code -> closer code
EH_DEV bool delim_is_opener(uint32_t code) { return code == 1 || code == 3 || code == 5 || code == 7 || code == 9 || code == 10; }
// byte-class tests as bit tables (word = byte >> 5), a handful of ALU ops per byte instead of a compare chain
EH_DEV bool delim_bit(uint32_t b, uint32_t w1, uint32_t w23) { uint32_t m = b < 32 ? 0u : (b < 64 ? w1 : (b < 128 ? w23 : 0u)); return (m >> (b & 31u)) & 1u; }
struct IsOpener { EH_DEV bool operator()(uint32_t b, uint32_t) const { return delim_bit(b, (1u << 2) | (1u << 7) | (1u << 8) | (1u << 28), 1u << 27); } };

// Parses the block (partial_parse/1 + grow/3, :800-905) and returns the completed nodes sorted by
// open position (pre-order), or -1 on allocation failure.  The matcher only looks at delimiter
// bytes (mask events); its stack lives in lane registers (entry d in lane d - base), spilling 32
// entries at a time to the work area for nesting deeper than 64.
#ifdef EH_PROF
#define TR_PH(k) do { uint64_t now_ = __builtin_readcyclecounter(); if (EH_LANE == 0) { atomicAdd(&c.p->prof[2 * (70 + (k))], (unsigned long long)(now_ - tph)); atomicAdd(&c.p->prof[2 * (70 + (k)) + 1], 1ull); } tph = now_; } while (0)
#define TR_ST(k, v) do { if (EH_LANE == 0) { atomicAdd(&c.p->prof[2 * (70 + (k))], (unsigned long long)(v)); atomicAdd(&c.p->prof[2 * (70 + (k)) + 1], 1ull); } } while (0)
#else
#define TR_PH(k) do { } while (0)
#define TR_ST(k, v) do { } while (0)
#endif
struct IsDelim { EH_DEV bool operator()(uint32_t b, uint32_t) const { return delim_bit(b, (1u << 2) | (1u << 7) | (1u << 8) | (1u << 9) | (1u << 28) | (1u << 30), (1u << 27) | (1u << 29)); } };
__device__ __noinline__ int tree_parse(Ctx&, cbptr H, uint32_t L, EH_G TNode** out) {
  EH_CTX;
  const int l = EH_LANE;
#ifdef EH_PROF
  uint64_t tph = __builtin_readcyclecounter();
#endif
  // 1. positions of all delimiter bytes, in order (parallel scan + compaction)
  uint32_t nev = wave_count(H, L, IsDelim());
  uint32_t nopen = wave_count(H, L, IsOpener());
  EH_G TNode* tab = (EH_G TNode*)ws_alloc(c, (uint64_t)(nopen + 1) * sizeof(TNode));
  wptr spill = (wptr)ws_alloc(c, (uint64_t)(nopen + 64) * 4);
  wptr evp = (wptr)ws_alloc(c, (uint64_t)(nev + 64) * 4);
  if (!tab || !spill || !evp) return -1;
  if (nopen >= (1u << 24)) { EH_SET_OVERFLOW(c, 801); return -1; }
  TR_PH(10);
  wave_collect(H, L, 0, nev, evp, IsDelim());
  TR_PH(11); TR_ST(15, nev);
  // 2. the matcher walks the event list 64 events at a time; positions and delimiter codes sit in
  //    registers (lane i = event i of the batch), the stack in lane registers too.  The inner loop
  //    touches no memory: on this ISA stores count in vmcnt, so a store per event makes every
  //    iteration wait for the previous one's write (~600 cycles).  Pushes and closes of a batch are
  //    buffered in lane registers (k-th push / k-th close of the batch in lane k) and written with one
  //    coalesced store / one scatter per batch.
  for (uint32_t i = (uint32_t)l; i < nopen + 1; i += 64) tab[i].close = 0xFFFFFFFFu;
  wave_sync();
  uint32_t stk = 0;                      // my stack entry: slot << 8 | expected closer code
  uint32_t sp = 0, sbase = 0;            // depth, depth held by lane 0
  uint32_t nslots = 0, top = 0;          // top = copy of the top entry
  for (uint32_t eb = 0; eb < nev; eb += 64) {
    uint32_t mypos = eb + (uint32_t)l < nev ? evp[eb + l] : 0;
    uint32_t mycode = eb + (uint32_t)l < nev ? DelimCls()((uint32_t)H[mypos]) : 0;
    // bit 8: opener; bits 16..23: the closer code an opener waits for
    mycode |= delim_is_opener(mycode) ? (0x100u | (delim_close_code(mycode) << 16)) : 0u;
    uint32_t cnt = nev - eb < 64 ? nev - eb : 64;
    uint32_t first = nslots, npush = 0, nclose = 0;
    uint32_t po = 0, pp = 0, cs = 0, cp = 0;
    for (uint32_t e = 0; e < cnt; e++) {
      uint32_t code = (uint32_t)__builtin_amdgcn_readlane((int)mycode, (int)e);
      uint32_t pos = (uint32_t)__builtin_amdgcn_readlane((int)mypos, (int)e);
      if (sp > 0 && (code & 255u) == (top & 255u)) {              // grow: H =:= Close (:806-807)
        if ((uint32_t)l == nclose) { cs = top >> 8; cp = pos; }
        nclose++;
        sp--;
        if (sp > 0) {
          if (sp == sbase) {                                      // refill the lower 32 entries from the spill area
            uint32_t up = (uint32_t)__shfl_up((int)stk, 32);
            sbase -= 32;
            stk = l < 32 ? spill[sbase + l] : up;
          }
          top = (uint32_t)__builtin_amdgcn_readlane((int)stk, (int)(sp - 1 - sbase));
        }
      } else if (code & 0x100u) {
        if (sp - sbase == 64) {                                   // spill the lower half
          if (l < 32) spill[sbase + l] = stk;
          stk = (uint32_t)__shfl_down((int)stk, 32);
          sbase += 32;
        }
        uint32_t parent = sp > 0 ? (top >> 8) : 0xFFFFFFFFu;
        uint32_t ent = (nslots << 8) | (code >> 16);
        if ((uint32_t)l == sp - sbase) stk = ent;
        if ((uint32_t)l == npush) { po = pos; pp = parent; }
        npush++;
        top = ent; nslots++; sp++;
      }
    }
    if ((uint32_t)l < npush) { tab[first + l].open = po; tab[first + l].pend = pp; }
    if ((uint32_t)l < nclose) tab[cs].close = cp;
  }
  wave_sync();
  TR_PH(12); TR_ST(16, nslots);
  // level end of every slot: one past the close of the nearest ancestor that did close (L at top level).
  // An opener that never closes stays on the matcher's stack for good, and so does everything below
  // it: the ancestors of an unclosed slot are all unclosed.  So the level end is close[parent] + 1 if
  // the parent closed and L otherwise — one gather (before the in-place compaction moves slots).
  for (uint32_t i = (uint32_t)l; i < nslots; i += 64) {
    uint32_t par = tab[i].pend, pe = L;
    if (par != 0xFFFFFFFFu) { uint32_t pc = tab[par].close; if (pc != 0xFFFFFFFFu) pe = pc + 1; }
    tab[i].pad = pe;
  }
  wave_sync();
  TR_PH(13);
  // compact completed nodes (keep pre-order); pend <- level end
  uint32_t n = 0;
  for (uint32_t base = 0; base < nslots; base += 64) {
    uint32_t i = base + (uint32_t)l;
    TNode t{0, 0xFFFFFFFFu, 0, 0};
    if (i < nslots) t = tab[i];
    bool ok = i < nslots && t.close != 0xFFFFFFFFu;
    unsigned long long m = __ballot(ok);
    uint32_t before = (uint32_t)__popcll(m & ((1ull << l) - 1));
    wave_sync();
    if (ok) { t.pend = t.pad; tab[n + before] = t; }   // n + before <= i: never overwrites an unread slot of a later chunk
    n += (uint32_t)__popcll(m);
    wave_sync();
  }
  TR_PH(14);
  *out = tab;
  return (int)n;
}

EH_DEV bool node_eq(cbptr H, TNode a, TNode b) {
  uint32_t la = a.close - a.open + 1, lb = b.close - b.open + 1;
  if (la != lb) return false;
  if (a.open == b.open) return true;
  return wave_equal(H + a.open, H + b.open, la);
}
EH_DEV TNode node_load(const EH_G TNode* t, uint32_t i) { TNode x = t[i]; x.open = uni(x.open); x.close = uni(x.close); x.pend = uni(x.pend); return x; }

// edit_sublist/3 sweep (:858-869) over nodes[lo..hi) in pre-order: the first node of a level that
// equals `sub` is reported and the rest of that level (up to its pend) is skipped.  64 nodes are
// loaded per step; only nodes of the right length are compared.  `level_end` replaces pend for
// nodes whose level is the sweep's own top level (sweeps restricted to a subtree pass its end).
// Lane-parallel equality of up to 64 candidate nodes against `sub` (all slen+1 bytes long): every
// candidate lane walks its own node 8 bytes at a time.  Returns the mask of equal candidates.
EH_DEV unsigned long long nodes_equal_mask(cbptr H, uint32_t my_open, bool cand, TNode sub) {
  uint32_t n = sub.close - sub.open + 1;
  if (n > 256) {
    // long nodes: one wave-wide compare per candidate (1 KiB per step); the node itself is trivially equal
    unsigned long long cm = __ballot(cand), res = 0;
    while (cm) {
      int j = (int)__builtin_ctzll(cm); cm &= cm - 1;
      uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)my_open, j);
      if (o == sub.open || wave_equal(H + o, H + sub.open, n)) res |= 1ull << j;
    }
    return res;
  }
  bool ne = !cand;
  bool self = cand && my_open == sub.open;
  uint32_t k = 0;
  for (; k + 8 <= n; k += 8) {
    uint64_t a = 0, b;
    b = ldg8(H + sub.open + k);
    if (!ne && !self) a = ldg8(H + my_open + k);
    ne = ne || (!self && a != b);
    if (__ballot(!ne && !self) == 0) break;
  }
  if (k + 8 <= n) return __ballot(!ne);                        // left the loop early: only `self` lanes (or none) remain
  for (; k < n; k++) { uint32_t b = H[sub.open + k]; if (!ne && !self && H[my_open + k] != b) ne = true; }
  return __ballot(!ne);
}
template <class F>
EH_DEV void tree_matches(cbptr H, const EH_G TNode* nodes, uint32_t lo, uint32_t hi, uint32_t level_end, TNode sub, wptr anc, F f) {
  (void)anc;
  const int l = EH_LANE;
  (void)l;
  uint32_t skip_until = 0;
  uint32_t slen = sub.close - sub.open;
  for (uint32_t base = lo; base < hi; base += 64) {
    uint32_t i = base + (uint32_t)l;
    TNode q{0, 0, 0, 0};
    if (i < hi) q = nodes[i];
    bool cand = i < hi && q.close - q.open == slen && q.open >= skip_until;
    if (__ballot(cand) == 0) continue;
    unsigned long long cm = nodes_equal_mask(H, q.open, cand, sub);
    while (cm) {
      int j = (int)__builtin_ctzll(cm); cm &= cm - 1;
      TNode x; x.open = (uint32_t)__builtin_amdgcn_readlane((int)q.open, j); x.close = (uint32_t)__builtin_amdgcn_readlane((int)q.close, j);
      x.pend = (uint32_t)__builtin_amdgcn_readlane((int)q.pend, j); x.pad = 0;
      if (x.open < skip_until) continue;
      f(x, base + (uint32_t)j);
      skip_until = x.pend < level_end ? x.pend : level_end;     // rest of the parent's level is left alone
    }
  }
}

// Assembles the edited block from a recorded match list without a serial copy per match (blocks with
// tens of thousands of equal small nodes are common after the line/sequence repeaters).  Match k
// (node index mlist[k] & 0x7fffffff, flag = top bit) contributes the segment
//   gap  = H[start_k, open_k)      start_k = end of the previous match (or its open when keep_node)
//   rep  = flag ? R1[0,r1len) : R0[0,r0len)
// 64 matches are handled per step: lane k owns a segment, a shuffle scan gives every segment its output
// offset, short segments are then written byte-per-lane (each output byte finds its segment by a 6-step
// shuffle search), long ones by a wave-wide copy each.
EH_DEV uint64_t tree_emit(bptr dst, cbptr H, uint32_t L, const EH_G TNode* nodes, cwptr mlist, uint32_t nm,
                          cbptr R0, uint32_t r0len, cbptr R1, uint32_t r1len, bool keep_node) {
  const int l = EH_LANE;
  uint64_t out = 0; uint32_t cur = 0;
  for (uint32_t base = 0; base < nm; base += 64) {
    uint32_t k = base + (uint32_t)l; bool act = k < nm;
    uint32_t qo = 0, qc = 0, flag = 0;
    if (act) { uint32_t e = mlist[k]; flag = e >> 31; TNode q = nodes[e & 0x7fffffffu]; qo = q.open; qc = q.close; }
    uint32_t nxt = keep_node ? qo : qc + 1;
    uint32_t start = (uint32_t)__shfl_up((int)nxt, 1); if (l == 0) start = cur;
    uint32_t g = act ? qo - start : 0;
    uint32_t rl = act ? (flag ? r1len : r0len) : 0;
    uint32_t seg = g + rl;
    bool big = seg > 512;
    uint32_t sseg = big ? 0 : seg;
    uint32_t inc = seg, sinc = sseg;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint32_t t = (uint32_t)__shfl_up((int)inc, d), t2 = (uint32_t)__shfl_up((int)sinc, d);
      if (l >= d) { inc += t; sinc += t2; }
    }
    uint32_t excl = inc - seg, sexcl = sinc - sseg;
    uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63), stotal = (uint32_t)__builtin_amdgcn_readlane((int)sinc, 63);
    for (uint32_t b0 = 0; b0 < stotal; b0 += 64) {
      uint32_t b = b0 + (uint32_t)l;
      uint32_t j = 0;
#pragma unroll
      for (uint32_t step = 32; step; step >>= 1) { uint32_t cand = j + step; uint32_t v = (uint32_t)__shfl((int)sexcl, (int)(cand & 63)); if (v <= b) j = cand; }
      uint32_t within = b - (uint32_t)__shfl((int)sexcl, (int)j);
      uint32_t gj = (uint32_t)__shfl((int)g, (int)j), sj = (uint32_t)__shfl((int)start, (int)j), fj = (uint32_t)__shfl((int)flag, (int)j);
      uint32_t ej = (uint32_t)__shfl((int)excl, (int)j);
      if (b < stotal) {
        uint8_t v = within < gj ? H[sj + within] : (fj ? R1 : R0)[within - gj];
        dst[out + ej + within] = v;
      }
    }
    unsigned long long bm = __ballot(big);
    while (bm) {
      int j = (int)__builtin_ctzll(bm); bm &= bm - 1;
      uint32_t gj = (uint32_t)__builtin_amdgcn_readlane((int)g, j), sj = (uint32_t)__builtin_amdgcn_readlane((int)start, j);
      uint32_t fj = (uint32_t)__builtin_amdgcn_readlane((int)flag, j), ej = (uint32_t)__builtin_amdgcn_readlane((int)excl, j);
      wave_copy(dst + out + ej, H + sj, gj);
      wave_copy(dst + out + ej + gj, fj ? R1 : R0, fj ? r1len : r0len);
    }
    out += total;
    uint32_t lastl = nm - base < 64 ? nm - base - 1 : 63;
    cur = (uint32_t)__builtin_amdgcn_readlane((int)nxt, (int)lastl);
  }
  wave_copy(dst + out, H + cur, L - cur); out += L - cur;
  return out;
}

// sed_tree_op (tr2/td :917-936), construct_sed_tree_swap (ts1/ts2 :940-971), sed_tree_stutter (tr :975-1023)
__device__ __noinline__ int muta_tree(Ctx&, int fn) {
  EH_CTX;
#ifdef EH_PROF
  uint64_t tph = __builtin_readcyclecounter();
#endif
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  const int l = EH_LANE;
  c.r_kind = R_SAME;
  if (binarish(H, L)) { TR_PH(0); return -1; }
  TR_PH(0);
  uint64_t mark = c.ws_used;
  EH_G TNode* nodes;
  int n_ = tree_parse(c, H, L, &nodes);
  TR_PH(1);
  if (n_ < 0) return 0;
  TR_ST(5, L); TR_ST(6, n_);
  uint32_t N = (uint32_t)n_;
  wptr anc = (wptr)ws_alloc(c, (uint64_t)(N + 1) * 4);
  if (!anc) return 0;
  // Subs = sublists(Lst): list position j (0-based) <-> nodes[N-1-j]

  if (fn == M_TR2 || fn == M_TD) {
    if (N == 0) { c.ws_used = mark; return 1; }                  // pick_sublist -> false: nothing is edited
    uint32_t idx = rng_rand(c.rng, N);
    TNode sub = node_load(nodes, N - 1 - idx);
    // one sweep records the matches, the edited block is then assembled in parallel
    uint32_t slen = sub.close - sub.open + 1;
    uint32_t nm = 0;
    tree_matches(H, nodes, 0, N, L, sub, anc, [&](TNode, uint32_t idx) { if (l == 0) anc[nm] = idx; nm++; });
    uint64_t nl = fn == M_TR2 ? (uint64_t)L + (uint64_t)nm * slen : (uint64_t)L - (uint64_t)nm * slen;
    bptr dst = ws_alloc(c, nl);
    if (!dst) return 1;
    wave_sync();
    // tr2: [H | Node] -> the node is written once more in front of itself; td: T -> the node is dropped
    uint64_t out = tree_emit(dst, H, L, nodes, anc, nm, H + sub.open, fn == M_TR2 ? slen : 0, nullptr, 0, fn == M_TR2);
    wave_sync();
    c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)out;
    return 1;
  }

  if (fn == M_TS1 || fn == M_TS2) {
    if (N < 2) { c.ws_used = mark; return -1; }
    // reservoir_sample(Subs, 2) (erlamsa_rnd.erl:201-214) over list positions
    // J_i = erand(i) for i = 3..N are independent draws: lane-parallel by jump-ahead; slot 1 / slot 2 end
    // up holding the LAST i that drew 1 / 2
    uint32_t r0 = 0, r1 = 1;
    {
      uint32_t m0 = 0, m1 = 0;
      for (uint32_t base = 3; base <= N; base += 64) {
        uint32_t i = base + (uint32_t)l;
        if (i <= N) {
          uint32_t j = (uint32_t)(rng_peek(c.rng, (uint32_t)l + 1) * (double)i) + 1;
          if (j == 1) m0 = i; else if (j == 2) m1 = i;
        }
        rng_skip(c.rng, N - base + 1 < 64 ? N - base + 1 : 64);
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) { uint32_t a = (uint32_t)__shfl_xor((int)m0, d), b = (uint32_t)__shfl_xor((int)m1, d); m0 = a > m0 ? a : m0; m1 = b > m1 ? b : m1; }
      m0 = uni(m0); m1 = uni(m1);
      if (m0) r0 = m0 - 1;
      if (m1) r1 = m1 - 1;
    }
    TNode A = node_load(nodes, N - 1 - r0), B = node_load(nodes, N - 1 - r1);
    TR_PH(2);
    if (fn == M_TS1) {                                           // sed_tree_swap_one
      if (rng_rand(c.rng, 2) == 1) { TNode t = A; A = B; B = t; } // random_permutation([A,B])
      uint32_t al = A.close - A.open + 1, bl = B.close - B.open + 1;
      uint32_t nm = 0;
      tree_matches(H, nodes, 0, N, L, A, anc, [&](TNode, uint32_t idx) { if (l == 0) anc[nm] = idx; nm++; });
      TR_PH(3); TR_ST(7, nm);
      uint64_t nl = (uint64_t)L + (uint64_t)nm * bl - (uint64_t)nm * al;
      bptr dst = ws_alloc(c, nl);
      if (!dst) return 1;
      wave_sync();
      uint64_t out = tree_emit(dst, H, L, nodes, anc, nm, H + B.open, bl, nullptr, 0, false);      // [B | Tl]
      wave_sync();
      TR_PH(4);
      c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)out;
      return 1;
    }
    // sed_tree_swap_two: edit_sublists/2 with {A -> B, B -> A}; every occurrence, no descent into a replaced node
    uint32_t al = A.close - A.open + 1, bl = B.close - B.open + 1;
    bool same = node_eq(H, A, B);
    int64_t delta = 0;
    uint32_t nm = 0;
    {
      uint32_t skip_until = 0;
      for (uint32_t base = 0; base < N; base += 64) {
        uint32_t i = base + (uint32_t)l;
        TNode q{0, 0, 0, 0};
        if (i < N) q = nodes[i];
        uint32_t qlen = q.close - q.open + 1;
        bool candB = i < N && qlen == bl && q.open >= skip_until, candA = i < N && qlen == al && q.open >= skip_until;
        if (__ballot(candA || candB) == 0) continue;
        unsigned long long mB = __ballot(candB) ? nodes_equal_mask(H, q.open, candB, B) : 0ull;
        unsigned long long mA = __ballot(candA) ? nodes_equal_mask(H, q.open, candA, A) : 0ull;
        mA &= ~mB;                                              // a node equal to both is looked up as B first
        unsigned long long cm = mA | mB;
        while (cm) {
          int j = (int)__builtin_ctzll(cm); cm &= cm - 1;
          uint32_t qo = (uint32_t)__builtin_amdgcn_readlane((int)q.open, j), qc = (uint32_t)__builtin_amdgcn_readlane((int)q.close, j);
          if (qo < skip_until) continue;
          bool isB = (mB >> j) & 1ull;
          // gb_trees: enter(A,->B) then enter(B,->A); equal keys: A -> A
          uint32_t rl = (isB || same) ? al : bl; uint32_t ql = qc - qo + 1;
          delta += (int64_t)rl - (int64_t)ql;
          if (l == 0) anc[nm] = (base + (uint32_t)j) | ((isB || same) ? 0x80000000u : 0u);
          nm++;
          skip_until = qc + 1;
        }
      }
    }
    bptr dst = ws_alloc(c, (uint64_t)((int64_t)L + delta));
    if (!dst) return 1;
    wave_sync();
    uint64_t out = tree_emit(dst, H, L, nodes, anc, nm, H + B.open, bl, H + A.open, al, false);
    wave_sync();
    c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)out;
    return 1;
  }

  // ---- tree stutter
  // RandSubs = random_permutation(Subs): only its first element WITH descendants matters.
  uint32_t pidx = 0xFFFFFFFFu;                                    // index into nodes[]
  auto has_desc = [&](uint32_t k) -> bool { return k + 1 < N && uni(nodes[k + 1].open) < uni(nodes[k].close); };
  if (N == 2) {
    uint32_t sw = rng_rand(c.rng, 2);                             // [A,B] = Subs[0],Subs[1] = nodes[1],nodes[0]
    uint32_t first = sw == 1 ? 0u : 1u, second = sw == 1 ? 1u : 0u;
    if (has_desc(first)) pidx = first; else if (has_desc(second)) pidx = second;
  } else if (N > 0) {
    // keys in list order: list position j <-> nodes[N-1-j]; the minimum {U, node} among nodes with descendants
    uint64_t best = ~(uint64_t)0; uint32_t bestk = 0xFFFFFFFFu;
    for (uint32_t base = 0; base < N; base += 64) {
      uint32_t j = base + (uint32_t)l;
      if (j < N) {
        uint32_t k = N - 1 - j;
        double u = rng_peek(c.rng, (uint32_t)l + 1);
        bool hd = k + 1 < N && nodes[k + 1].open < nodes[k].close;
        uint64_t key = (uint64_t)__double_as_longlong(u);
        if (hd && key < best) { best = key; bestk = k; }          // (ties on the float key would fall back to term order)
      }
      rng_skip(c.rng, N - base < 64 ? N - base : 64);
    }
    // wave arg-min
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      uint64_t ob = ((uint64_t)(uint32_t)__shfl_xor((int)(best >> 32), d) << 32) | (uint32_t)__shfl_xor((int)(uint32_t)best, d);
      uint32_t ok = (uint32_t)__shfl_xor((int)bestk, d);
      if (ob < best) { best = ob; bestk = ok; }
    }
    pidx = uni(bestk);
  }
  TNode P{0, 0, 0, 0}, C{0, 0, 0, 0}; uint32_t ndesc = 0;
  if (pidx != 0xFFFFFFFFu) {
    P = node_load(nodes, pidx);
    // descendants of P are contiguous in pre-order: nodes[pidx+1 .. pidx+ndesc]
    uint32_t cnt = 0;
    for (uint32_t base = pidx + 1; base < N; base += 64) {
      uint32_t k = base + (uint32_t)l;
      bool in = k < N && nodes[k].open < P.close;
      unsigned long long m = __ballot(in);
      cnt += (uint32_t)__popcll(m);
      if (m != ~0ull) break;
    }
    ndesc = cnt;
    uint32_t ci = rng_rand(c.rng, ndesc);                          // choose_child: rand_elem(sublists(H))
    C = node_load(nodes, pidx + ndesc - ci);                      // reverse pre-order
  }
  uint32_t nreps = rng_log(c.rng, 10);
  if (pidx == 0xFFFFFFFFu) { c.ws_used = mark; return -1; }
  // R_n = repeat_path(Parent, Child, n): matches of Child inside Parent
  uint32_t plen = P.close - P.open + 1, clen = C.close - C.open + 1;
  uint32_t k_in = 0; uint64_t matched_bytes = 0;
  tree_matches(H, nodes, pidx + 1, pidx + 1 + ndesc, P.close + 1, C, anc, [&](TNode q, uint32_t) { k_in++; matched_bytes += q.close - q.open + 1; });
  (void)clen;
  uint64_t fixed = plen - matched_bytes;                          // bytes of P outside the matched children
  // size of R_n
  uint64_t rsz = plen;
  for (uint32_t t = 2; t <= nreps; t++) { rsz = (uint64_t)k_in * rsz + fixed; if (rsz > ws_max_request(c)) { EH_SET_OVERFLOW(c, 802); c.ovf_req = rsz; return 1; } }
  bptr R = nullptr;
  if (nreps < 2) R = (bptr)(H + P.open);
  else if (k_in == 1) {
    // R_n = pre^(n-1) ++ P ++ suf^(n-1)
    TNode m{0, 0, 0, 0};
    tree_matches(H, nodes, pidx + 1, pidx + 1 + ndesc, P.close + 1, C, anc, [&](TNode q, uint32_t) { m = q; });
    uint32_t pre = m.open - P.open, suf = P.close - m.close;
    R = ws_alloc(c, rsz);
    if (!R) return 1;
    wave_fill_periodic(R, H + P.open, pre, (uint64_t)pre * (nreps - 1));
    wave_copy(R + (uint64_t)pre * (nreps - 1), H + P.open, plen);
    wave_fill_periodic(R + (uint64_t)pre * (nreps - 1) + plen, H + m.close + 1, suf, (uint64_t)suf * (nreps - 1));
    wave_sync();
  } else {
    cbptr prev = H + P.open; uint64_t prevsz = plen;
    for (uint32_t t = 2; t <= nreps; t++) {
      uint64_t sz = (uint64_t)k_in * prevsz + fixed;
      bptr cur = ws_alloc(c, sz);
      if (!cur) return 1;
      uint32_t from = P.open; uint64_t out = 0;
      tree_matches(H, nodes, pidx + 1, pidx + 1 + ndesc, P.close + 1, C, anc, [&](TNode q, uint32_t) {
        wave_copy(cur + out, H + from, q.open - from); out += q.open - from;
        uint64_t done = 0; while (done < prevsz) { uint32_t cc = prevsz - done > 0x40000000ull ? 0x40000000u : (uint32_t)(prevsz - done); wave_copy(cur + out + done, prev + done, cc); done += cc; }
        out += prevsz; from = q.close + 1;
      });
      wave_copy(cur + out, H + from, P.close + 1 - from);
      wave_sync();
      prev = cur; prevsz = sz;
    }
    R = (bptr)prev;
  }
  // top level: edit_sublist(Lst, Child, [R_N | Tl])
  uint32_t nm = 0; uint64_t mb = 0;
  tree_matches(H, nodes, 0, N, L, C, anc, [&](TNode q, uint32_t) { nm++; mb += q.close - q.open + 1; });
  uint64_t nl = (uint64_t)L - mb + (uint64_t)nm * rsz;
  if (nl > 0xFFFFFFF0ull) { EH_SET_OVERFLOW(c, 803); c.ovf_req = ~0ull; return 1; }
  bptr dst = ws_alloc(c, nl);
  if (!dst) return 1;
  uint32_t cur = 0; uint64_t out = 0;
  tree_matches(H, nodes, 0, N, L, C, anc, [&](TNode q, uint32_t) {
    wave_copy(dst + out, H + cur, q.open - cur); out += q.open - cur;
    uint64_t done = 0; while (done < rsz) { uint32_t cc = rsz - done > 0x40000000ull ? 0x40000000u : (uint32_t)(rsz - done); wave_copy(dst + out + done, R + done, cc); done += cc; }
    out += rsz; cur = q.close + 1;
  });
  wave_copy(dst + out, H + cur, L - cur); out += L - cur;
  wave_sync();
  c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)out;
  return 1;
}

}  // namespace eh
