// eh_fuse_lds.h — find_jump_points/2 (erlamsa_fuse.erl:102-128) for SMALL lists, entirely in LDS.
//
// What the reference's node lists are, in array terms (eh_fuse.h / eh_fuse2.h derive the same facts): after g rounds the
// nodes are the g-grams that occur on both sides; in the order eh_fuse.h stores them (reference order on even generations,
// reversed on odd ones) they ascend by the key (c1, ~c2, c3, ~c4, ...) and the members of a node ascend by position, because
// every round is ONE stable partition by the next byte' (byte' = the byte on even rounds, 255 - byte on odd ones).  So the
// whole state of a generation is
//     E[0, n)   the live suffixes (source suffix s as s, target suffix s as la + s; fuse(H, H): sources only, the target lists
//               ARE the source lists), sorted by (key, side, position) — 16-bit entries that never change, a member's current
//               position is s + g;
//     M         one bit per entry: "a node starts here".
// A round re-sorts every node by the next byte' in place and drops what dies — a few hundred bytes of LDS traffic per 64
// members instead of eh_fuse.h's dependent global gathers (a 16-byte node descriptor, its member arrays, the data byte: three
// memory round trips per step, which is what a `ft` on a 4 KiB block spent its 3.3 M cycles in, profiles/r03_survey_final.txt).
//   * nodes of at most 64 members: several whole nodes per step, one member per lane; a member's new place follows from lane
//     masks built with ballots (members of its node with a smaller byte', with the same byte', kept lanes) - no sort, no shuffles;
//   * a node of more than 64 members: 257-bin LDS histogram (source count | target count << 16), per-bin verdicts, stable
//     scatter behind per-bin cursors; the node is parked in the work area meanwhile (streamed, 4 bytes a member);
//   * fuse(H, H) once every node is a single member: a round only retires the member whose rest is [].
// Nothing is written before a round has a survivor (kept members are written compacted, behind the read position), so the
// generation a round found is still there when NoDesp =:= [] sends the reference back to it (:124-126).
//
// The entry whose rest is [] (fix_empty_list/1 :58-60; same rules as eh_fuse.h, restated for this layout): the member at
// position len - 1 is the LAST of its group in array order (largest position; the empty suffixes before it are skipped); it is
// dropped when it is inserted first, i.e. when it is alone on its side of its new group (even generations) or always (odd
// ones: the lists are reversed).  A source side that is [] after the drop makes the child {[[]], [[]]} whatever the target side
// holds (:91-93): here the dropped member stays as the node's only entry (its position is then la, and no target entry
// means to = lb with one draw).  A target side that is [] after the drop leaves a node without targets (`ghost`: no draw).
#pragma once

namespace eh {

constexpr uint32_t FL_NMAX = 8192;                      // entries (sources + targets; sources only for fuse(H, H))
constexpr uint32_t FL_E_WORDS = FL_NMAX / 2;
constexpr uint32_t FL_M_OFF = FL_E_WORDS, FL_M_WORDS = 264;
constexpr uint32_t FL_H_OFF = FL_M_OFF + FL_M_WORDS, FL_H_WORDS = 264;
constexpr uint32_t FL_NONE = 0xFFFFFFFFu;
static_assert(FL_H_OFF + FL_H_WORDS <= EH_FUSE_LDS_WORDS, "g_fuse_lds too small for eh_fuse_lds.h");
// histogram word of a bin after the verdicts: source cursor (14 bits) | FL_DEAD | FL_DROPA | target cursor << 16 | FL_DROPB | FL_SP
constexpr uint32_t FL_CUR = 0x3FFFu, FL_DEAD = 1u << 14, FL_DROPA = 1u << 15, FL_DROPB = 1u << 30, FL_SP = 1u << 31;

struct FlState {
  cbptr A; cbptr B; uint32_t la, lb; bool sym;
  uint32_t n, g, nn;        // live entries, generation, nodes
  uint32_t ghost;           // start index of the node without targets, FL_NONE: none
  uint32_t nsp;             // {[[]], [[]]} nodes of this generation (work accounting: they count two members, hold one entry)
  uint32_t multi;           // nodes with more than one entry
  uint32_t dboff;           // byte offset in g_fuse_lds of the staged data (A then B), FL_NONE: read the lists where they are
  wptr T;              // work-area parking space of a big node (n0 words)
  // fuse(H, H) with single-member nodes only: a round retires one known member; it is marked, not removed (see fl_round)
  uint32_t fast, tab_g0, ntomb;
};
struct FlRound { uint32_t w, nn, ghost, nsp, multi; };    // a round's output so far: entries written, nodes, ...

EH_DEV uint16_t* fl_E() { return reinterpret_cast<uint16_t*>(g_fuse_lds); }
EH_DEV uint32_t* fl_M() { return g_fuse_lds + FL_M_OFF; }
EH_DEV uint32_t* fl_H() { return g_fuse_lds + FL_H_OFF; }

// byte' of entry e (combined index) at position e + g of the combined data; the caller knows the member is not empty
EH_DEV uint32_t fl_byte(const FlState& st, uint32_t e, bool side, uint32_t s) {
  uint32_t b;
  if (st.dboff != FL_NONE) b = reinterpret_cast<const uint8_t*>(g_fuse_lds)[st.dboff + e + st.g];
  else b = side ? st.B[s + st.g] : st.A[s + st.g];
  return (st.g & 1u) ? 255u - b : b;
}
// start bits of entries [i, i + 64) and of entry i + 64
EH_DEV uint64_t fl_window(uint32_t i, uint32_t* b64) {
  const uint32_t* M = fl_M();
  uint32_t w0 = M[i >> 5], w1 = M[(i >> 5) + 1], w2 = M[(i >> 5) + 2], sh = i & 31u;
  uint64_t lo = ((uint64_t)w1 << 32) | w0;
  uint64_t r = sh ? (lo >> sh) | ((uint64_t)w2 << (64 - sh)) : lo;
  *b64 = sh ? (w2 >> sh) & 1u : w2 & 1u;
  return r;
}
// smallest index >= j whose start bit is set, n if there is none below n
EH_DEV uint32_t fl_next_start(uint32_t j, uint32_t n) {
  const uint32_t* M = fl_M();
  const uint32_t l = (uint32_t)EH_LANE;
  for (uint32_t wbase = j >> 5; wbase * 32u < n; wbase += 64) {
    uint32_t wd = wbase + l;
    uint32_t v = wd * 32u < n ? M[wd] : 0u;
    if (wd == (j >> 5)) v &= ~0u << (j & 31u);
    unsigned long long any = __ballot(v != 0);
    if (any) {
      int src = (int)__builtin_ctzll(any);
      uint32_t vv = (uint32_t)__builtin_amdgcn_readlane((int)v, src);
      uint32_t idx = (wbase + (uint32_t)src) * 32u + (uint32_t)__builtin_ctz(vv);
      return idx < n ? idx : n;
    }
  }
  return n;
}
// index of the k-th (0-based) start bit; k < number of start bits below n
EH_DEV uint32_t fl_kth_start(uint32_t k, uint32_t n) {
  const uint32_t* M = fl_M();
  const uint32_t l = (uint32_t)EH_LANE;
  uint32_t seen = 0;
  for (uint32_t wbase = 0; wbase * 32u < n; wbase += 64) {
    uint32_t wd = wbase + l;
    uint32_t v = wd * 32u < n ? M[wd] : 0u;
    if (wd * 32u + 32u > n && wd * 32u < n) v &= (1u << (n & 31u)) - 1u;       // (n & 31 != 0 here)
    uint32_t c = (uint32_t)__popc(v);
    uint32_t inc = wave_incl_scan(c);
    uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    if (seen + tot > k) {
      unsigned long long hit = __ballot(seen + inc > k);
      int src = (int)__builtin_ctzll(hit);
      uint32_t vv = (uint32_t)__builtin_amdgcn_readlane((int)v, src);
      uint32_t before = seen + (uint32_t)__builtin_amdgcn_readlane((int)(inc - c), src);
      uint32_t r = k - before;                                                // r-th set bit of vv
      for (uint32_t t = 0; t < r; t++) vv &= vv - 1u;
      return (wbase + (uint32_t)src) * 32u + (uint32_t)__builtin_ctz(vv);
    }
    seen += tot;
  }
  return n;
}
// clears the start bits of entries [a, b)
EH_DEV void fl_clear_bits(uint32_t a, uint32_t b) {
  uint32_t* M = fl_M();
  const uint32_t l = (uint32_t)EH_LANE;
  if (a >= b) return;
  for (uint32_t wbase = a >> 5; wbase <= ((b - 1u) >> 5); wbase += 64) {
    uint32_t wd = wbase + l;
    if (wd <= ((b - 1u) >> 5)) {
      uint32_t lo = wd * 32u, mask = ~0u;
      if (a > lo) mask &= ~0u << (a - lo);
      if (b < lo + 32u) mask &= (1u << (b - lo)) - 1u;
      M[wd] &= ~mask;
    }
  }
  lanes_sync();
}

// The verdict on one new group (all operands are the group's own): how many sources / targets it keeps, whether it is a child.
struct FlVerdict { bool ch, sp, fdrop, tdrop; uint32_t keepA, keepB; };
EH_DEV FlVerdict fl_verdict(bool sym, bool even, uint32_t rf, uint32_t rt, bool hasA, bool hasB) {
  FlVerdict v;
  if (sym) { rt = rf; hasB = hasA; }
  v.fdrop = hasA && (even ? rf == 1u : true);
  v.tdrop = hasB && (even ? rt == 1u : true);
  uint32_t ef = rf - (v.fdrop ? 1u : 0u), et = rt - (v.tdrop ? 1u : 0u);
  v.sp = rf > 0 && ef == 0;
  v.ch = rf > 0 && (v.sp || rt > 0);
  v.keepA = v.ch ? (v.sp ? 1u : ef) : 0u;
  v.keepB = (v.ch && !v.sp && !sym) ? et : 0u;
  return v;
}

// ---- whole nodes of <= 64 members together: entries [i, i + m), one per lane ---------------------------------------------------
// No sort: a member's place in the next generation follows from three lane masks — the members of its node with a smaller byte',
// those with the same byte' (its new group), and which lanes are kept — all built from ballots (8, one per bit of the byte), so the
// step is a straight run of VALU work without a single dependent LDS or cross-lane round trip (a 21-stage bitonic sort through
// ds_bpermute cost ~2 000 cycles of pure latency per step).
EH_DEV void fl_chunk(const FlState& st, uint32_t i, uint32_t m, uint64_t sb, FlRound& o) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint16_t* E = fl_E();
  const bool in = l < m, even = (st.g & 1u) == 0;
  uint32_t e = in ? (uint32_t)E[i + l] : 0u;
  bool side = !st.sym && e >= st.la;
  uint32_t s = side ? e - st.la : e, slen = side ? st.lb : st.la;
  const bool alive = in && s + st.g < slen;                                          // (the empty suffix is skipped, :66-67)
  const bool star = alive && s + st.g == slen - 1u;
  uint32_t bt = alive ? fl_byte(st, e, side, s) : 0u;
  // the lanes of this lane's node: [gs, ge)
  const unsigned long long upto = (2ull << l) - 1ull;
  uint32_t gs = 63u - (uint32_t)__builtin_clzll((sb & upto) | 1ull);
  unsigned long long later = sb & ~upto;
  uint32_t ge = later ? (uint32_t)__builtin_ctzll(later) : m;
  unsigned long long pgm = (ge >= 64 ? ~0ull : ((1ull << ge) - 1ull)) & ~((1ull << gs) - 1ull);
  const unsigned long long am = __ballot(alive);
  unsigned long long eqm = pgm & am, lessm = 0;
#pragma unroll
  for (int bit = 7; bit >= 0; bit--) {
    unsigned long long mb = __ballot(alive && ((bt >> bit) & 1u));
    if ((bt >> bit) & 1u) { lessm |= eqm & ~mb; eqm &= mb; } else eqm &= ~mb;
  }
  if (!alive) { eqm = 0; lessm = 0; }
  unsigned long long amA = __ballot(alive && !side), amB = __ballot(alive && side);
  // Nothing to do? Every member alive, none of them the one whose rest is [], every node's members agree on the next byte
  // (and, two lists: have sources and targets) - then the nodes are their own children, in place.  The steady state of
  // repetitive data, where nodes never split.
  if (o.w == i && __ballot(in && (!alive || star || eqm != pgm || (!st.sym && ((amA & pgm) == 0 || (amB & pgm) == 0)))) == 0) {
    const unsigned long long nxtb = (sb >> 1) | (1ull << (m - 1u));
    const uint32_t groups = (uint32_t)__popcll(sb);
    o.nn += groups; o.multi += groups - (uint32_t)__popcll(sb & nxtb); o.w += m;
    return;
  }
  unsigned long long smA = __ballot(star && !side), smB = __ballot(star && side);
  FlVerdict v = fl_verdict(st.sym, even, (uint32_t)__popcll(amA & eqm), (uint32_t)__popcll(amB & eqm), (smA & eqm) != 0, (smB & eqm) != 0);
  bool keep = alive && v.ch && (v.sp ? (star && !side) : !((star && !side && v.fdrop) || (star && side && v.tdrop)));
  unsigned long long km = __ballot(keep);
  uint32_t kc = (uint32_t)__popcll(km);
  if (kc == 0) return;
  const unsigned long long below = (1ull << l) - 1ull;
  uint32_t wi = o.w + (uint32_t)__popcll(km & ((1ull << gs) - 1ull)) + (uint32_t)__popcll(km & lessm) + (uint32_t)__popcll(km & eqm & below);
  bool first = keep && (km & eqm & below) == 0;
  unsigned long long fm = __ballot(first);
  unsigned long long gm = __ballot(first && !st.sym && !v.sp && v.keepB == 0);
  unsigned long long spm = __ballot(first && v.sp);
  unsigned long long mm = __ballot(first && (uint32_t)__popcll(km & eqm) > 1u);
  fl_clear_bits(o.w, o.w + kc);
  if (keep) E[wi] = (uint16_t)e;
  if (first) atomicOr(&fl_M()[wi >> 5], 1u << (wi & 31u));
  lanes_sync();
  if (gm) o.ghost = (uint32_t)__builtin_amdgcn_readlane((int)wi, (int)__builtin_ctzll(gm));
  o.nn += (uint32_t)__popcll(fm); o.nsp += (uint32_t)__popcll(spm); o.multi += (uint32_t)__popcll(mm);
  o.w += kc;
}
// fuse(H, H), a step of single-member nodes: each stays what it is unless it is the empty suffix now
EH_DEV void fl_singles(const FlState& st, uint32_t i, uint32_t m, FlRound& o) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint16_t* E = fl_E();
  uint32_t e = l < m ? (uint32_t)E[i + l] : 0u;
  bool keep = l < m && e + st.g < st.la;
  unsigned long long km = __ballot(keep);
  uint32_t kc = (uint32_t)__popcll(km);
  if (kc == 0) return;
  if (o.w != i || kc != m) {
    uint32_t wi = o.w + (uint32_t)__popcll(km & ((1ull << l) - 1ull));
    lanes_sync();
    if (keep) E[wi] = (uint16_t)e;
    // start bits of [w, w + kc): all set
    uint32_t* M = fl_M();
    const uint32_t a = o.w, b = o.w + kc;
    uint32_t wd = (a >> 5) + l;
    if (l < 3 && wd <= ((b - 1u) >> 5)) {
      uint32_t lo = wd * 32u, mask = ~0u;
      if (a > lo) mask &= ~0u << (a - lo);
      if (b < lo + 32u) mask &= (1u << (b - lo)) - 1u;
      M[wd] |= mask;
    }
    lanes_sync();
  }
  o.nn += kc; o.w += kc;
}

// ---- one node of more than 64 members: entries [i, i + L) ---------------------------------------------------------------------
#define FL_UNROLL 4
EH_DEV void fl_big(const FlState& st, uint32_t i, uint32_t L, FlRound& o) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint16_t* E = fl_E();
  uint32_t* H = fl_H();
  const bool even = (st.g & 1u) == 0;
  for (uint32_t k = l; k < 260; k += 64) H[k] = 0;
  lanes_sync();
  // pass 1: the next bytes' histogram; the node is parked in T as entry | byte' << 16 (256: the empty suffix) | star << 25
  uint32_t binA = FL_NONE, binB = FL_NONE;                            // bins of the members whose rest is []
  for (uint32_t base = 0; base < L; base += 64 * FL_UNROLL) {
    uint32_t ev[FL_UNROLL], bv[FL_UNROLL]; bool sv[FL_UNROLL], inv[FL_UNROLL], stv[FL_UNROLL];
#pragma unroll
    for (int u = 0; u < FL_UNROLL; u++) {
      uint32_t idx = base + 64u * (uint32_t)u + l;
      inv[u] = idx < L;
      ev[u] = inv[u] ? (uint32_t)E[i + idx] : 0u;
      sv[u] = !st.sym && ev[u] >= st.la;
    }
#pragma unroll
    for (int u = 0; u < FL_UNROLL; u++) {
      uint32_t s = sv[u] ? ev[u] - st.la : ev[u], slen = sv[u] ? st.lb : st.la;
      bool nonempty = inv[u] && s + st.g < slen;
      stv[u] = nonempty && s + st.g == slen - 1u;
      bv[u] = nonempty ? fl_byte(st, ev[u], sv[u], s) : 256u;
    }
#pragma unroll
    for (int u = 0; u < FL_UNROLL; u++) {
      if (base + 64u * (uint32_t)u >= L) break;
      uint32_t idx = base + 64u * (uint32_t)u + l;
      if (inv[u]) { st.T[idx] = ev[u] | (bv[u] << 16) | (stv[u] ? 1u << 25 : 0u); atomicAdd(&H[bv[u]], sv[u] ? 0x10000u : 1u); }
      unsigned long long sa = __ballot(stv[u] && !sv[u]), sb2 = __ballot(stv[u] && sv[u]);
      if (sa) binA = (uint32_t)__builtin_amdgcn_readlane((int)bv[u], (int)__builtin_ctzll(sa));
      if (sb2) binB = (uint32_t)__builtin_amdgcn_readlane((int)bv[u], (int)__builtin_ctzll(sb2));
    }
  }
  wave_sync();                                                         // (T is read back below)
  // verdicts: lane l owns bins 4l .. 4l+3, ascending byte' = array order
  uint32_t keepA[4], keepB[4], flg[4], kb = 0, lc = 0, lsp = 0, lmu = 0; bool gh[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t b = 4u * l + (uint32_t)j, h = H[b];
    FlVerdict v = fl_verdict(st.sym, even, h & 0xFFFFu, h >> 16, b == binA, b == binB);
    keepA[j] = v.keepA; keepB[j] = v.keepB;
    flg[j] = (v.ch ? 0u : FL_DEAD) | (v.fdrop ? FL_DROPA : 0u) | (v.tdrop ? FL_DROPB : 0u) | (v.sp ? FL_SP : 0u);
    gh[j] = v.ch && !st.sym && !v.sp && v.keepB == 0;
    kb += keepA[j] + keepB[j]; lc += v.ch ? 1u : 0u; lsp += v.sp ? 1u : 0u; lmu += (keepA[j] + keepB[j] > 1u) ? 1u : 0u;
  }
  uint32_t inc = wave_incl_scan(kb);
  uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
  uint32_t nch = wave_sum(lc);
  if (total == 0) return;
  lanes_sync();
  fl_clear_bits(o.w, o.w + total);
  uint32_t off = o.w + inc - kb;
  uint32_t ghost_here = FL_NONE;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    uint32_t b = 4u * l + (uint32_t)j;
    H[b] = (off & FL_CUR) | (((off + keepA[j]) & FL_CUR) << 16) | flg[j];
    if (keepA[j] + keepB[j] > 0) atomicOr(&fl_M()[off >> 5], 1u << (off & 31u));
    if (gh[j]) ghost_here = off;
    off += keepA[j] + keepB[j];
  }
  unsigned long long ghm = __ballot(ghost_here != FL_NONE);
  if (ghm) o.ghost = (uint32_t)__builtin_amdgcn_readlane((int)ghost_here, (int)__builtin_ctzll(ghm));
  o.nn += nch; o.nsp += wave_sum(lsp); o.multi += wave_sum(lmu);
  lanes_sync();
  // pass 2: stable scatter behind the bins' cursors
  for (uint32_t base = 0; base < L; base += 64 * FL_UNROLL) {
    uint32_t tv[FL_UNROLL];
#pragma unroll
    for (int u = 0; u < FL_UNROLL; u++) { uint32_t idx = base + 64u * (uint32_t)u + l; tv[u] = idx < L ? st.T[idx] : (256u << 16); }
#pragma unroll
    for (int u = 0; u < FL_UNROLL; u++) {
      if (base + 64u * (uint32_t)u >= L) break;
      uint32_t e = tv[u] & 0xFFFFu, bt = (tv[u] >> 16) & 511u; bool star = (tv[u] >> 25) & 1u;
      bool side = !st.sym && e >= st.la;
      uint32_t h = bt < 256u ? H[bt] : FL_DEAD;
      bool keep = !(h & FL_DEAD);
      if (keep) {
        if (h & FL_SP) keep = star && !side;
        else keep = !((star && !side && (h & FL_DROPA)) || (star && side && (h & FL_DROPB)));
      }
      // rank among the kept lanes of this step with the same (byte', side)
      uint32_t mk = bt | (side ? 256u : 0u);
      unsigned long long eq = __ballot(keep);
#pragma unroll
      for (int bit = 0; bit < 9; bit++) { unsigned long long mb = __ballot(keep && ((mk >> bit) & 1u)); eq &= ((mk >> bit) & 1u) ? mb : ~mb; }
      uint32_t rank = (uint32_t)__popcll(eq & ((1ull << l) - 1ull)), cnt = (uint32_t)__popcll(eq);
      lanes_sync();                                                    // every lane has read its bin's cursor
      if (keep) {
        uint32_t cur = side ? (h >> 16) & FL_CUR : h & FL_CUR;
        E[cur + rank] = (uint16_t)e;
        if (rank == 0) atomicAdd(&H[bt], side ? cnt << 16 : cnt);
      }
      lanes_sync();
    }
  }
  o.w += total;
}

// One round g -> g + 1 over all nodes.  Returns the number of children; 0: nothing was written, the generation is intact.
EH_DEV uint32_t fl_round(FlState& st) {
  const uint32_t l = (uint32_t)EH_LANE;
  FlRound o; o.w = 0; o.nn = 0; o.ghost = FL_NONE; o.nsp = 0; o.multi = 0;
  if (st.sym && st.multi == 0) {
    // every node is one member: it stays one (its next byte is its own), except the member whose rest WAS [] a round ago
    // (position la now: the empty suffix is skipped, :66-67), which leaves; the member at la - 1 becomes {[[]], [[]]} and stays.
    // The leaving member is suffix la - g: a table (H[0, 64): where the suffixes la - tab_g0 - j are in E, j < 64) finds it, and
    // it is only MARKED dead (E = 0xFFFF, its index noted in H[64 ..)) - closing the gap would move half the array every round.
    uint16_t* E = fl_E();
    uint32_t* H = fl_H();
    if (st.ntomb >= 190u) {                                          // (never in practice: a round for every mark) close the gaps
      uint32_t w = 0;
      for (uint32_t base = 0; base < st.n; base += 64) {
        uint32_t e = base + l < st.n ? (uint32_t)E[base + l] : 0xFFFFu;
        unsigned long long km = __ballot(e != 0xFFFFu);
        lanes_sync();
        if (e != 0xFFFFu) E[w + (uint32_t)__popcll(km & ((1ull << l) - 1ull))] = (uint16_t)e;
        w += (uint32_t)__popcll(km);
        lanes_sync();
      }
      st.n = w; st.ntomb = 0; st.fast = 0;
    }
    if (!st.fast || st.g - st.tab_g0 >= 64u) {
      st.fast = 1; st.tab_g0 = st.g;
      if (l < 64) H[l] = FL_NONE;
      lanes_sync();
      for (uint32_t base = 0; base < st.n; base += 64) {
        uint32_t e = base + l < st.n ? (uint32_t)E[base + l] : 0xFFFFu;
        uint32_t d = st.la - st.g - e;                               // (wraps for the others)
        if (e != 0xFFFFu && d < 64u) H[d] = base + l;
      }
      lanes_sync();
    }
    const uint32_t at = st.g <= st.la ? H[st.g - st.tab_g0] : FL_NONE;
    if (at == FL_NONE) { st.g++; return st.nn; }
    if (st.nn == 1) return 0;
    if (l == 0) { E[at] = 0xFFFFu; H[64u + st.ntomb] = at; }
    lanes_sync();
    st.ntomb++; st.nn--; st.g++;
    return st.nn;
  }
  uint32_t i = 0;
  while (i < st.n) {
    uint32_t b64 = 0;
    uint64_t sb = fl_window(i, &b64);
    uint32_t rest = st.n - i;
    uint32_t m;
    if (rest <= 64) m = rest;
    else {
      uint64_t cand = (sb >> 1) | ((uint64_t)b64 << 63);               // bit j-1: a node starts at i + j, j = 1..64
      m = cand ? 64u - (uint32_t)__builtin_clzll(cand) : 0u;
    }
    if (rest < 64) sb &= (1ull << rest) - 1ull;
    if (m == 0) {
      uint32_t end = fl_next_start(i + 65, st.n);
      fl_big(st, i, end - i, o);
      i = end;
    } else {
      if (m < 64) sb &= (1ull << m) - 1ull;
      const uint64_t ones = m < 64 ? (1ull << m) - 1ull : ~0ull;
      if (st.sym && sb == ones) fl_singles(st, i, m, o);            // (the entry after the step starts a node: m ends at a boundary)
      else fl_chunk(st, i, m, sb, o);
      i += m;
    }
  }
  if (o.nn == 0) return 0;
  st.n = o.w; st.nn = o.nn; st.ghost = o.ghost; st.nsp = o.nsp; st.multi = o.multi; st.g++;
  return o.nn;
}

// find_jump_points/2 + any_position_pair/1 for lists whose members fit FL_NMAX entries; draws are the reference's.
__device__ __noinline__ bool fuse_jump_lds(Ctx&, cbptr A, uint32_t la, cbptr B, uint32_t lb, bool sym, uint32_t* from, uint32_t* tpos, uint32_t* rounds) {
  EH_CTX;
  const uint32_t l = (uint32_t)EH_LANE;
  FlState st;
  st.A = A; st.B = B; st.la = la; st.lb = lb; st.sym = sym;
  const uint32_t n0 = sym ? la : la + lb;
  st.n = n0; st.g = 0; st.nn = 1; st.ghost = FL_NONE; st.nsp = 0; st.multi = n0 > 1 ? 1u : 0u; st.dboff = FL_NONE;
  st.fast = 0; st.tab_g0 = 0; st.ntomb = 0;
  st.T = nullptr;
  if (n0 > 64) { st.T = (wptr)ws_alloc(c, 4ull * n0); if (!st.T) return false; }
  uint16_t* E = fl_E();
  for (uint32_t k = l; k < n0; k += 64) E[k] = (uint16_t)k;
  for (uint32_t k = l; k < FL_M_WORDS; k += 64) fl_M()[k] = k == 0 ? 1u : 0u;
  {
    // the lists themselves move to LDS when they fit behind the entries (a `ft` of a 4 KiB block: 8 KiB of entries + 4 KiB of data)
    uint32_t dbytes = sym ? la : la + lb, off = (2u * n0 + 15u) & ~15u;
    if (off + dbytes + 16u <= FL_E_WORDS * 4u) {
      uint8_t* D = reinterpret_cast<uint8_t*>(g_fuse_lds) + off;
      for (uint32_t k = l; k < la; k += 64) D[k] = A[k];
      if (!sym) for (uint32_t k = l; k < lb; k += 64) D[la + k] = B[k];
      st.dboff = off;
    }
  }
  lanes_sync();
  EH_PT0;
  int64_t fuel = 100000;                                             // ?SEARCH_FUEL
  uint64_t gen_entries = (uint64_t)la + lb;
  while (true) {                                                     // find_jump_points_loop (:115-128)
    if (fuel < 0) break;
    if (rng_rand(c.rng, 8) == 0) break;                              // ?SEARCH_STOP_IP
    if (c.work_budget) {
      c.work += 16ull * gen_entries;
      if (c.work > c.work_budget) { c.status = CASE_BUDGET; return false; }
    }
#ifdef EH_PROF
    const int pslot = (st.sym && st.multi == 0) ? 102 : (st.g == 0 ? 100 : 101);   // eh_result_prof: first round / later rounds / rounds of single members
#endif
    uint32_t nchild = fl_round(st);
    EH_PT(c, pslot);
    if (nchild == 0) break;                                          // NoDesp =:= [] -> any_position_pair(Nodes)
    fuel -= (int64_t)nchild;
    gen_entries = sym ? 2ull * (st.n - st.ntomb) : (uint64_t)st.n + st.nsp;
    (*rounds)++;
  }
  // any_position_pair/1 (:73-77); odd generations are stored reversed
  const uint32_t par = st.g & 1u;
  uint32_t ni = rng_rand(c.rng, st.nn);
  uint32_t k = par ? st.nn - 1u - ni : ni;
  uint32_t a, b;
  if (st.sym && st.multi == 0) {                                    // single members: the k-th entry that is not marked dead
    a = k;
    if (st.ntomb) {
      const uint32_t* H = fl_H();
      for (;;) {
        uint32_t cdead = 0;
        for (uint32_t t0 = 0; t0 < st.ntomb; t0 += 64) cdead += (uint32_t)__popcll(__ballot(t0 + l < st.ntomb && H[64u + t0 + l] <= a));
        if (k + cdead == a) break;
        a = k + cdead;
      }
    }
    b = a + 1u;
  } else { a = fl_kth_start(k, st.n); b = fl_next_start(a + 1u, st.n); }
  uint32_t fc = b - a;
  if (!sym) {                                                        // sources come first
    uint32_t cnt = 0;
    for (uint32_t base = a; base < b; base += 64) cnt += (uint32_t)__popcll(__ballot(base + l < b && (uint32_t)E[base + l] < la));
    fc = cnt;
  }
  uint32_t tc = sym ? fc : (b - a) - fc;
  if (c.fp_on) {                                                     // eh_fuse_red.h: name the node, the members come from the original lists
    const uint32_t e0 = uni((uint32_t)E[a]), el = uni((uint32_t)E[a + fc - 1u]);          // (fc >= 1: a node has a source entry)
    c.fp_g = st.g; c.fp_keypos = e0;
    c.fp_special = (fc == 1u && e0 + st.g == la) ? 1u : 0u;
    c.fp_bA = (el + st.g == la) ? 1u : 0u;
    c.fp_bB = sym ? c.fp_bA : ((tc > 0 && uni((uint32_t)E[b - 1u]) - la + st.g == lb) ? 1u : 0u);
    EH_PT(c, 103);
    return true;
  }
  bool special_t = false;
  if (!sym && tc == 0 && a != st.ghost) { tc = 1; special_t = true; }   // {[[]], [[]]}: Tos = [[]]
  *from = la; *tpos = lb;
  if (fc > 0) { uint32_t j = rng_rand(c.rng, fc); *from = (uint32_t)E[a + (par ? fc - 1u - j : j)] + st.g; }
  if (tc > 0) {
    uint32_t j = rng_rand(c.rng, tc);
    if (!special_t) { uint32_t idx = sym ? a + (par ? tc - 1u - j : j) : a + fc + (par ? tc - 1u - j : j); uint32_t e = (uint32_t)E[idx]; *tpos = (sym ? e : e - la) + st.g; }
  }
  *from = uni(*from); *tpos = uni(*tpos);
  EH_PT(c, 103);
  return true;
}

}  // namespace eh
