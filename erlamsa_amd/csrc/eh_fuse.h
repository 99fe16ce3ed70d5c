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
  Key2* keys;                         // sort buffer (capacity cap2)
  uint32_t* bstart;                   // group starts in sorted order (+1 sentinel)
  uint64_t* bkey;                     // group key = (nn-1-node) << 8 | (255-byte)
  uint32_t* bcnt;                     // group size after fix_empty_list
  uint32_t nb;
};

// keys + sort + grouping for one side.  Returns false on allocation problems.
EH_DEV void fuse_group(FuseSide& x, uint32_t nn) {
  const int l = EH_LANE;
  uint32_t np2 = 64; while (np2 < x.n) np2 <<= 1;
  for (uint32_t i = l; i < np2; i += 64) {
    Key2 k; k.hi = ~(uint64_t)0; k.lo = 0;
    if (i < x.n) {
      uint32_t p = x.pos[i];
      if (p < x.len) {                                           // ([], Subs) -> Subs : the empty suffix drops out
        k.hi = ((uint64_t)(nn - 1 - x.node[i]) << 40) | ((uint64_t)(255u - x.s[p]) << 32) | (uint64_t)(x.n - 1 - i);
        k.lo = p + 1;
      }
    }
    x.keys[i] = k;
  }
  wave_sort_key2(x.keys, np2);
  // group boundaries: (hi >> 32) changes; invalid keys (hi == ~0) sort last
  uint32_t nb = 0;
  for (uint32_t base = 0; base < np2; base += 64) {
    uint32_t i = base + (uint32_t)l;
    uint64_t h = x.keys[i].hi;
    bool valid = h != ~(uint64_t)0;
    bool start = valid && (i == 0 || (x.keys[i - 1].hi >> 32) != (h >> 32));
    unsigned long long m = __ballot(start);
    uint32_t before = (uint32_t)__popcll(m & ((1ull << l) - 1));
    if (start) { x.bstart[nb + before] = i; x.bkey[nb + before] = h >> 32; }
    nb += (uint32_t)__popcll(m);
    unsigned long long vm = __ballot(valid);
    if (vm != ~0ull) {                                            // end of the valid prefix
      uint32_t nvalid = base + (uint32_t)__popcll(vm);
      if (l == 0) x.bstart[nb] = nvalid;
      break;
    }
    if (base + 64 >= np2 && l == 0) x.bstart[nb] = np2;
  }
  wave_sync();
  x.nb = nb;
  // fix_empty_list (:58-60): a group whose LAST element (first one inserted) is the empty tail loses it
  for (uint32_t j = l; j < nb; j += 64) {
    uint32_t a = x.bstart[j], b = x.bstart[j + 1];
    uint32_t cnt = b - a;
    if (cnt > 0 && (uint32_t)x.keys[b - 1].lo == x.len) cnt--;
    x.bcnt[j] = cnt;
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
  uint32_t np2f = 64; while (np2f < capf) np2f <<= 1;
  uint32_t np2t = 64; while (np2t < capt) np2t <<= 1;
  f.s = A; f.len = la; t.s = B; t.len = lb;
  f.pos = (uint32_t*)ws_alloc(c, (uint64_t)capf * 4); f.node = (uint32_t*)ws_alloc(c, (uint64_t)capf * 4);
  t.pos = (uint32_t*)ws_alloc(c, (uint64_t)capt * 4); t.node = (uint32_t*)ws_alloc(c, (uint64_t)capt * 4);
  uint32_t* f2 = (uint32_t*)ws_alloc(c, (uint64_t)capf * 4); uint32_t* fn2 = (uint32_t*)ws_alloc(c, (uint64_t)capf * 4);
  uint32_t* t2 = (uint32_t*)ws_alloc(c, (uint64_t)capt * 4); uint32_t* tn2 = (uint32_t*)ws_alloc(c, (uint64_t)capt * 4);
  f.keys = (Key2*)ws_alloc(c, (uint64_t)np2f * sizeof(Key2)); t.keys = (Key2*)ws_alloc(c, (uint64_t)np2t * sizeof(Key2));
  f.bstart = (uint32_t*)ws_alloc(c, (uint64_t)(capf + 1) * 4); t.bstart = (uint32_t*)ws_alloc(c, (uint64_t)(capt + 1) * 4);
  f.bkey = (uint64_t*)ws_alloc(c, (uint64_t)capf * 8); t.bkey = (uint64_t*)ws_alloc(c, (uint64_t)capt * 8);
  f.bcnt = (uint32_t*)ws_alloc(c, (uint64_t)capf * 4); t.bcnt = (uint32_t*)ws_alloc(c, (uint64_t)capt * 4);
  // node table: per node start/count in pos arrays (current and next)
  uint32_t capn = capf;                                            // every node owns >= 1 source entry
  uint32_t* nfs = (uint32_t*)ws_alloc(c, (uint64_t)(capn + 1) * 4); uint32_t* nts = (uint32_t*)ws_alloc(c, (uint64_t)(capn + 1) * 4);
  uint32_t* nfs2 = (uint32_t*)ws_alloc(c, (uint64_t)(capn + 1) * 4); uint32_t* nts2 = (uint32_t*)ws_alloc(c, (uint64_t)(capn + 1) * 4);
  uint32_t* cflag = (uint32_t*)ws_alloc(c, (uint64_t)capn * 4); uint32_t* cmatch = (uint32_t*)ws_alloc(c, (uint64_t)capn * 4);
  if (!f.pos || !f.node || !t.pos || !t.node || !f2 || !fn2 || !t2 || !tn2 || !f.keys || !t.keys || !f.bstart || !t.bstart || !f.bkey ||
      !t.bkey || !f.bcnt || !t.bcnt || !nfs || !nts || !nfs2 || !nts2 || !cflag || !cmatch) return false;
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
    fuse_group(f, nn);
    fuse_group(t, nn);
    // children, in the order of the sorted source groups: a group with no elements left is the
    // special node {[[]], [[]]}; otherwise it needs a target group with the same (node, byte)
    uint32_t nchild = 0, newf = 0, newt = 0;
    for (uint32_t base = 0; base < f.nb; base += 64) {
      uint32_t j = base + (uint32_t)l;
      bool child = false; uint32_t fc = 0, tc = 0, tj = 0xFFFFFFFFu;
      if (j < f.nb) {
        fc = f.bcnt[j];
        if (fc == 0) { child = true; fc = 1; tc = 1; tj = 0xFFFFFFFEu; }        // [[[[]], []] | Tl]
        else {
          uint64_t key = f.bkey[j];
          uint32_t lo = 0, hi = t.nb;                              // binary search (ascending keys)
          while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (t.bkey[mid] < key) lo = mid + 1; else hi = mid; }
          if (lo < t.nb && t.bkey[lo] == key) { child = true; tj = lo; tc = t.bcnt[lo]; }
        }
      }
      // exclusive prefix sums of (child, fc, tc) across the wave
      uint32_t ci = child ? 1u : 0u, fi = child ? fc : 0u, ti = child ? tc : 0u;
      uint32_t cs = wave_incl_scan(ci), fs = wave_incl_scan(fi), ts = wave_incl_scan(ti);
      if (child) {
        uint32_t k = nchild + cs - 1, fo = newf + fs - fi, to = newt + ts - ti;
        nfs2[k] = fo; nts2[k] = to; cflag[k] = j; cmatch[k] = tj;
      }
      nchild += uni((uint32_t)__shfl((int)cs, 63)); newf += uni((uint32_t)__shfl((int)fs, 63)); newt += uni((uint32_t)__shfl((int)ts, 63));
    }
    if (l == 0) { nfs2[nchild] = newf; nts2[nchild] = newt; }
    wave_sync();
    if (nchild == 0) break;                                        // NoDesp =:= [] -> any_position_pair(Nodes)
    // materialise the children's suffix lists (one lane per child)
    for (uint32_t base = 0; base < nchild; base += 64) {
      uint32_t k = base + (uint32_t)l;
      if (k < nchild) {
        uint32_t j = cflag[k], tj = cmatch[k];
        uint32_t fo = nfs2[k], fcnt = nfs2[k + 1] - fo, to = nts2[k], tcnt = nts2[k + 1] - to;
        if (tj == 0xFFFFFFFEu) { f2[fo] = la; fn2[fo] = k; t2[to] = lb; tn2[to] = k; }
        else {
          uint32_t fa = f.bstart[j];
          for (uint32_t i = 0; i < fcnt; i++) { f2[fo + i] = (uint32_t)f.keys[fa + i].lo; fn2[fo + i] = k; }
          uint32_t ta = t.bstart[tj];
          for (uint32_t i = 0; i < tcnt; i++) { t2[to + i] = (uint32_t)t.keys[ta + i].lo; tn2[to + i] = k; }
        }
      }
    }
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
