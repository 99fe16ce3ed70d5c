// eh_fuse.h — device code for erlamsa_fuse:fuse/2 (erlamsa_fuse.erl:47-134) and the mutators
// ft, fn, fo (erlamsa_mutations.erl:380-427).
//
// find_jump_points/2 refines a list of nodes {source suffixes, target suffixes} one character per
// round: suffixes are grouped by their next byte and a source group survives if the target side
// has the same byte.  Suffixes are positions, so a round is: key every live suffix with
// (node, next byte, list index), SORT (the reference builds every group by prepending and the node
// list by prepending, i.e. everything comes out in descending order of (node, byte, index)), cut
// the sorted sequence into groups with ballot/prefix scans, and join source groups with target
// groups by binary search.  The PRNG draws (one rand(8) per round, rand_elem x3 at the end) are
// the reference's.
#pragma once
#include "eh_field.h"

namespace eh {

struct FuseSide {
  const uint8_t* s; uint32_t len;     // the byte list
  uint32_t* pos;                      // suffix start positions (len == empty suffix), grouped by node
  uint32_t* node;                     // node index of every entry
  uint32_t n;                         // live entries
  Key2* keys;                         // entries in group order: hi = group key << 32 | ..., lo = next position
  uint32_t* gid;                      // group index of every sorted entry
  uint32_t* bstart;                   // group starts in sorted order (+1 sentinel)
  uint64_t* bkey;                     // group key = (nn-1-node) << 8 | (255-byte)
  uint32_t* bcnt;                     // group size after fix_empty_list
  uint32_t* child;                    // child node built from this group (or NONE)
  uint32_t* nfirst;                   // per (reversed) node: first group and one past its last group
  uint32_t* nend;
  uint32_t nb;
};
constexpr uint32_t FUSE_NONE = 0xFFFFFFFFu, FUSE_SPECIAL = 0xFFFFFFFEu;

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

// Orders the live suffixes of one side the way the reference's prepending builds its groups and node
// list — descending (node, next byte, list index), the empty suffix dropped (`([], Subs) -> Subs`) —
// and cuts them into groups.  The entries arrive grouped by node (seg[k]..seg[k+1]), so this is a stable
// counting sort by the next byte inside every segment, mirrored: segments of more than 64 entries use
// a 256-bin histogram in the work area (ballot match for the stable ranks), smaller ones are packed up
// to 64 segments per step and ordered by a register bitonic network.  (A full comparison sort of
// 128-bit keys per round cost ~10 ms per fuse call.)
EH_DEV void fuse_group(FuseSide& x, uint32_t nn, const uint32_t* seg, uint32_t* hist) {
  const int l = EH_LANE;
  uint32_t nvalid = 0;
  for (uint32_t base = 0; base < x.n; base += 64) {
    uint32_t i = base + (uint32_t)l;
    nvalid += (uint32_t)__popcll(__ballot(i < x.n && x.pos[i] < x.len));
  }
  auto put = [&](uint32_t asc, uint32_t nd, uint32_t b, uint32_t idx, uint32_t p) {
    uint32_t d = nvalid - 1 - asc;
    x.keys[d].hi = ((uint64_t)(nn - 1 - nd) << 40) | ((uint64_t)(255u - b) << 32) | (uint64_t)(x.n - 1 - idx);
    x.keys[d].lo = p + 1;
  };
  uint32_t vbase = 0, k = 0;
  while (k < nn) {
    uint32_t s0 = uni(seg[k]);
    uint32_t kk = k + (uint32_t)l + 1;
    uint32_t myend = kk <= nn ? seg[kk] : 0xFFFFFFFFu;
    bool fits = kk <= nn && myend - s0 <= 64;
    unsigned long long fm = __ballot(fits);
    uint32_t nfit = fm == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fm);       // offsets ascend: `fits` is a prefix
    if (nfit > 0) {
      uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)myend, (int)(nfit - 1)) - s0;
      uint32_t i = s0 + (uint32_t)l; bool in = (uint32_t)l < cnt;
      uint32_t p = in ? x.pos[i] : 0u; bool v = in && p < x.len;
      uint32_t nd = in ? x.node[i] : 0u;
      uint32_t b = v ? (uint32_t)x.s[p] : 0u;
      uint32_t key = v ? (((nd - k) << 14) | (b << 6) | (uint32_t)l) : 0xFFFFFFFFu;
      // entries arrive ordered by segment; when the bytes happen to ascend inside every segment too (always
      // the case once segments have shrunk to single entries) the 21-step network is skipped
      uint32_t nxt = (uint32_t)__shfl_down((int)key, 1);
      uint32_t sk = __ballot(l < 63 && key > nxt) == 0 ? key : wave_sort64(key);
      bool sv = sk != 0xFFFFFFFFu;
      uint32_t ol = sk & 63u;
      uint32_t sp = (uint32_t)__shfl((int)p, (int)ol), snd = (uint32_t)__shfl((int)nd, (int)ol);
      if (sv) put(vbase + (uint32_t)l, snd, (sk >> 6) & 255u, s0 + ol, sp);
      vbase += (uint32_t)__popcll(__ballot(sv));
      k += nfit;
      continue;
    }
    // one big segment
    uint32_t e0 = uni(seg[k + 1]);
    for (uint32_t b = (uint32_t)l; b < 256; b += 64) hist[b] = 0;
    wave_sync();
    for (uint32_t base = s0; base < e0; base += 64) {
      uint32_t i = base + (uint32_t)l;
      if (i < e0) { uint32_t p = x.pos[i]; if (p < x.len) atomicAdd(&hist[x.s[p]], 1u); }
    }
    wave_sync();
    uint32_t h0 = hist[4 * l], h1 = hist[4 * l + 1], h2 = hist[4 * l + 2], h3 = hist[4 * l + 3];
    uint32_t sum = h0 + h1 + h2 + h3, inc = wave_incl_scan(sum), exc = inc - sum;
    uint32_t segvalid = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    wave_sync();
    hist[4 * l] = exc; hist[4 * l + 1] = exc + h0; hist[4 * l + 2] = exc + h0 + h1; hist[4 * l + 3] = exc + h0 + h1 + h2;
    wave_sync();
    for (uint32_t base = s0; base < e0; base += 64) {
      uint32_t i = base + (uint32_t)l; bool in = i < e0;
      uint32_t p = in ? x.pos[i] : 0u; bool v = in && p < x.len;
      uint32_t b = v ? (uint32_t)x.s[p] : 0u;
      unsigned long long eq = __ballot(v);                        // lanes holding the same byte as me
#pragma unroll
      for (int bit = 0; bit < 8; bit++) { unsigned long long m = __ballot(v && ((b >> bit) & 1u)); eq &= ((b >> bit) & 1u) ? m : ~m; }
      uint32_t rank = (uint32_t)__popcll(eq & ((1ull << l) - 1)), cntg = (uint32_t)__popcll(eq);
      uint32_t off = v ? hist[b] : 0u;
      wave_sync();                                                // every lane has read its bin before the group leaders bump them
      if (v && rank == 0) hist[b] = off + cntg;
      wave_sync();
      if (v) put(vbase + off + rank, k, b, i, p);
    }
    vbase += segvalid;
    k += 1;
  }
  wave_sync();
  // group boundaries: (hi >> 32) changes
  uint32_t nb = 0;
  for (uint32_t base = 0; base < nvalid; base += 64) {
    uint32_t i = base + (uint32_t)l;
    bool valid = i < nvalid;
    uint64_t h = valid ? x.keys[i].hi : 0;
    bool start = valid && (i == 0 || (x.keys[i - 1].hi >> 32) != (h >> 32));
    unsigned long long m = __ballot(start);
    uint32_t upto = (uint32_t)__popcll(m & ((2ull << l) - 1));    // group starts at or before me
    if (start) { x.bstart[nb + upto - 1] = i; x.bkey[nb + upto - 1] = h >> 32; }
    if (valid) x.gid[i] = nb + upto - 1;
    nb += (uint32_t)__popcll(m);
  }
  if (l == 0) x.bstart[nb] = nvalid;
  wave_sync();
  x.nb = nb;
  // group range of every node (keys ascend, so a node's groups are contiguous): lookups then search at most
  // 256 groups, usually one, instead of all of them
  for (uint32_t k2 = (uint32_t)l; k2 < nn; k2 += 64) { x.nfirst[k2] = 0; x.nend[k2] = 0; }
  wave_sync();
  for (uint32_t j = (uint32_t)l; j < nb; j += 64) {
    uint32_t np = (uint32_t)(x.bkey[j] >> 8);
    if (j == 0 || (uint32_t)(x.bkey[j - 1] >> 8) != np) x.nfirst[np] = j;
    if (j + 1 == nb || (uint32_t)(x.bkey[j + 1] >> 8) != np) x.nend[np] = j + 1;
  }
  // fix_empty_list (:58-60): a group whose LAST element (first one inserted) is the empty tail loses it
  for (uint32_t j = l; j < nb; j += 64) {
    uint32_t a = x.bstart[j], b = x.bstart[j + 1];
    uint32_t cnt = b - a;
    if (cnt > 0 && (uint32_t)x.keys[b - 1].lo == x.len) cnt--;
    x.bcnt[j] = cnt;
    x.child[j] = FUSE_NONE;
  }
  wave_sync();
}

// fuse(Al, Bl) -> new byte list in the work area
EH_DEV bool fuse_lists(Ctx& c, const uint8_t* A, uint32_t la, const uint8_t* B, uint32_t lb, uint8_t** out, uint32_t* outlen) {
  const int l = EH_LANE;
  if (la == 0) { *out = (uint8_t*)B; *outlen = lb; return true; }   // fuse([], Bl) -> Bl
  if (lb == 0) { *out = (uint8_t*)A; *outlen = la; return true; }
  uint64_t mark = c.ws_used;
  FuseSide f, t;
  uint32_t capf = la + 2, capt = lb + 2;
  f.s = A; f.len = la; t.s = B; t.len = lb;
  auto u32 = [&](uint64_t n) { return (uint32_t*)ws_alloc(c, n * 4); };
  f.pos = u32(capf); f.node = u32(capf); t.pos = u32(capt); t.node = u32(capt);
  uint32_t* f2 = u32(capf); uint32_t* fn2 = u32(capf); uint32_t* t2 = u32(capt); uint32_t* tn2 = u32(capt);
  f.keys = (Key2*)ws_alloc(c, (uint64_t)capf * sizeof(Key2)); t.keys = (Key2*)ws_alloc(c, (uint64_t)capt * sizeof(Key2));
  f.gid = u32(capf); t.gid = u32(capt);
  f.bstart = u32(capf + 1); t.bstart = u32(capt + 1);
  f.bkey = (uint64_t*)ws_alloc(c, (uint64_t)capf * 8); t.bkey = (uint64_t*)ws_alloc(c, (uint64_t)capt * 8);
  f.bcnt = u32(capf); t.bcnt = u32(capt); f.child = u32(capf); t.child = u32(capt);
  f.nfirst = u32(capf); f.nend = u32(capf); t.nfirst = u32(capf); t.nend = u32(capf);      // indexed by node: nn <= capf
  // node table: per node start/count in pos arrays (current and next)
  uint32_t capn = capf;                                            // every node owns >= 1 source entry
  uint32_t* nfs = u32(capn + 1); uint32_t* nts = u32(capn + 1); uint32_t* nfs2 = u32(capn + 1); uint32_t* nts2 = u32(capn + 1);
  uint32_t* cmatch = u32(capn); uint32_t* hist = u32(256);
  if (!f.pos || !f.node || !t.pos || !t.node || !f2 || !fn2 || !t2 || !tn2 || !f.keys || !t.keys || !f.gid || !t.gid || !f.bstart ||
      !t.bstart || !f.bkey || !t.bkey || !f.bcnt || !t.bcnt || !f.child || !t.child || !f.nfirst || !f.nend || !t.nfirst || !t.nend || !nfs || !nts || !nfs2 || !nts2 || !cmatch || !hist) return false;
  // find_jump_points (:103-107): one node with all non-empty suffixes of both lists
  for (uint32_t i = l; i < la; i += 64) { f.pos[i] = i; f.node[i] = 0; }
  for (uint32_t i = l; i < lb; i += 64) { t.pos[i] = i; t.node[i] = 0; }
  if (l == 0) { nfs[0] = 0; nfs[1] = la; nts[0] = 0; nts[1] = lb; }
  f.n = la; t.n = lb;
  uint32_t nn = 1;
  int64_t fuel = 100000;                                           // ?SEARCH_FUEL
  wave_sync();
  while (true) {                                                   // find_jump_points_loop (:115-128)
    if (fuel < 0) break;
    if (rng_rand(c.rng, 8) == 0) break;                            // ?SEARCH_STOP_IP
    fuse_group(f, nn, nfs, hist);
    fuse_group(t, nn, nts, hist);
    // children, in the order of the source groups: a group with no elements left is the special node
    // {[[]], [[]]}; otherwise it needs a target group with the same (node, byte)
    uint32_t nchild = 0, newf = 0, newt = 0;
    for (uint32_t base = 0; base < f.nb; base += 64) {
      uint32_t j = base + (uint32_t)l;
      bool child = false; uint32_t fc = 0, tc = 0, tj = FUSE_NONE;
      if (j < f.nb) {
        fc = f.bcnt[j];
        if (fc == 0) { child = true; fc = 1; tc = 1; tj = FUSE_SPECIAL; }           // [[[[]], []] | Tl]
        else {
          uint64_t key = f.bkey[j];
          uint32_t np = (uint32_t)(key >> 8);
          uint32_t lo = t.nfirst[np], hi = t.nend[np];             // binary search inside the node's groups (ascending keys)
          uint32_t tend = hi;
          while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (t.bkey[mid] < key) lo = mid + 1; else hi = mid; }
          if (lo < tend && t.bkey[lo] == key) { child = true; tj = lo; tc = t.bcnt[lo]; }
        }
      }
      // exclusive prefix sums of (child, fc, tc) across the wave
      uint32_t ci = child ? 1u : 0u, fi = child ? fc : 0u, ti = child ? tc : 0u;
      uint32_t cs = wave_incl_scan(ci), fs = wave_incl_scan(fi), ts = wave_incl_scan(ti);
      if (child) {
        uint32_t kc = nchild + cs - 1;
        nfs2[kc] = newf + fs - fi; nts2[kc] = newt + ts - ti; cmatch[kc] = tj;
        if (tj != FUSE_SPECIAL) { f.child[j] = kc; t.child[tj] = kc; }
      }
      nchild += uni((uint32_t)__shfl((int)cs, 63)); newf += uni((uint32_t)__shfl((int)fs, 63)); newt += uni((uint32_t)__shfl((int)ts, 63));
    }
    if (l == 0) { nfs2[nchild] = newf; nts2[nchild] = newt; }
    wave_sync();
    if (nchild == 0) break;                                        // NoDesp =:= [] -> any_position_pair(Nodes)
    // the children's suffix lists are the surviving groups in the same order: a compaction
    for (uint32_t i = (uint32_t)l; i < f.bstart[f.nb]; i += 64) {
      uint32_t g = f.gid[i], kc = f.child[g];
      if (kc != FUSE_NONE) { uint32_t o = i - f.bstart[g]; if (o < f.bcnt[g]) { f2[nfs2[kc] + o] = (uint32_t)f.keys[i].lo; fn2[nfs2[kc] + o] = kc; } }
    }
    for (uint32_t i = (uint32_t)l; i < t.bstart[t.nb]; i += 64) {
      uint32_t g = t.gid[i], kc = t.child[g];
      if (kc != FUSE_NONE) { uint32_t o = i - t.bstart[g]; if (o < t.bcnt[g]) { t2[nts2[kc] + o] = (uint32_t)t.keys[i].lo; tn2[nts2[kc] + o] = kc; } }
    }
    for (uint32_t kc = (uint32_t)l; kc < nchild; kc += 64)
      if (cmatch[kc] == FUSE_SPECIAL) { f2[nfs2[kc]] = la; fn2[nfs2[kc]] = kc; t2[nts2[kc]] = lb; tn2[nts2[kc]] = kc; }
    wave_sync();
    // swap generations
    { uint32_t* tmp; tmp = f.pos; f.pos = f2; f2 = tmp; tmp = f.node; f.node = fn2; fn2 = tmp; tmp = t.pos; t.pos = t2; t2 = tmp; tmp = t.node; t.node = tn2; tn2 = tmp;
      tmp = nfs; nfs = nfs2; nfs2 = tmp; tmp = nts; nts = nts2; nts2 = tmp; }
    f.n = newf; t.n = newt; nn = nchild;
    fuel -= (int64_t)nchild;
  }
  // any_position_pair/1 (:73-77)
  uint32_t ni = rng_rand(c.rng, nn);
  uint32_t fo = uni(nfs[ni]), fcnt = uni(nfs[ni + 1]) - fo, to = uni(nts[ni]), tcnt = uni(nts[ni + 1]) - to;
  uint32_t from = la, tpos = lb;
  if (fcnt > 0) from = uni(f.pos[fo + rng_rand(c.rng, fcnt)]);
  if (tcnt > 0) tpos = uni(t.pos[to + rng_rand(c.rng, tcnt)]);
  c.ws_used = mark;                                                // release all tables
  // jump/3 (:47-50): Al up to From, then To
  uint32_t nl = from + (lb - tpos);
  uint8_t* dst = ws_alloc(c, nl);
  if (!dst) return false;
  wave_copy(dst, A, from);
  wave_copy(dst + from, B + tpos, lb - tpos);
  wave_sync();
  *out = dst; *outlen = nl;
  return true;
}

struct FoState { uint64_t ptr; uint32_t len; uint32_t has; };

__device__ __noinline__ int muta_fuse(Ctx&, int fn, FoState* fo) {
  EH_CTX;
  Blk hb = blk_load(c.bl, c.cur);
  const uint8_t* H = (const uint8_t*)hb.ptr; uint32_t L = hb.len;
  c.r_kind = R_SAME;
  uint8_t* r; uint32_t rl;
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
    uint8_t* abl; uint32_t abll;
    if (!fuse_lists(c, H, h1, (const uint8_t*)nb.ptr, nb.len, &abl, &abll)) return 0;
    if (!fuse_lists(c, abl, abll, H + h1, L - h1, &r, &rl)) return 0;
    int d = rng_delta(c.rng);
    c.r_kind = R_NEW; c.r_ptr = r; c.r_len = rl; c.r_flush = 1; c.r_drop_next = have_next ? 1 : 0;
    return d;
  }
  // sed_fuse_old / remember/1 :405-427
  uint32_t has = uni(fo->has);
  uint64_t optr = has ? uni64(fo->ptr) : hb.ptr; uint32_t olen = has ? uni(fo->len) : L;
  uint32_t o1 = olen / 2;
  uint8_t* a; uint32_t al; uint8_t* b; uint32_t bl;
  if (!fuse_lists(c, H, h1, (const uint8_t*)optr, o1, &a, &al)) return 0;              // a -> o
  if (!fuse_lists(c, (const uint8_t*)optr + o1, olen - o1, H + h1, L - h1, &b, &bl)) return 0;   // o -> a
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
