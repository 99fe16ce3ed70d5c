// eh_doc.h — piece lists: the common output form of the document mutators (sgm, js, b64).
//
// A document mutator never edits bytes in place.  Its tokenizer records, for every token, the
// pieces of the token's canonical rendering (erlamsa_sgml:fold_ast/2, erlamsa_json:fold_ast/2) as
// (pointer, length, repeat) triples that reference the input block, a small literal pool or a
// temporary in the work area.  Every AST element of those modules is a contiguous token range, so a
// structural mutation (swap, dup, repeat, insert, pump ...) is an edit script over piece ranges, and
// the new block is produced by ONE gather at the end — the input is read once, the output written
// once.
//
// wave_gather: pieces of <= 48 bytes are copied lane-per-piece (8 byte loads in flight per lane, no
// cross-lane traffic), longer or repeated pieces wave-per-piece with the 16-byte movers.
#pragma once
#include "eh_lex.h"

namespace eh {

struct Piece { uint64_t ptr; uint32_t len; uint32_t rep; };

EH_DEV void piece_put(EH_G Piece* t, uint32_t i, const void* p, uint32_t len, uint32_t rep = 1) {
  if (EH_LANE == 0) { t[i].ptr = (uint64_t)p; t[i].len = len; t[i].rep = rep; }
}
// per-lane source lane (ds_bpermute); readlane64 needs a wave-uniform lane
EH_DEV uint64_t shfl64(uint64_t v, int src) { return ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(v >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)v, src); }
EH_DEV uint32_t wave_max(uint32_t v) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { uint32_t t = (uint32_t)__shfl_xor((int)v, d); v = t > v ? t : v; }
  return uni(v);
}
EH_DEV uint64_t pieces_total(const EH_G Piece* t, uint32_t n) {
  uint64_t s = 0;
  for (uint32_t i = EH_LANE; i < n; i += 64) s += (uint64_t)t[i].len * t[i].rep;
  return wave_sum64(s);
}
// Merges neighbours that are adjacent in memory (ptr + len == next ptr, no repeats) IN PLACE and returns the new
// count.  Tokenizers point their "literal" pieces at the input bytes whenever the input spells the canonical form,
// so an unedited stretch of a document collapses into one long piece and the gather below moves it with 16-byte
// vectors instead of a lane per 1-byte piece.
EH_DEV uint32_t pieces_coalesce(EH_G Piece* t, uint32_t n) {
  const int l = EH_LANE;
  uint32_t nout = 0;
  uint64_t carry_end = 0; bool have_carry = false;                  // end address of the last written piece (if mergeable)
  for (uint32_t base = 0; base < n; base += 64) {
    uint32_t idx = base + (uint32_t)l;
    Piece p = {0, 0, 1};
    if (idx < n) p = t[idx];
    bool valid = idx < n && p.len > 0;                             // empty pieces vanish
    bool plain = p.rep == 1;
    // does my piece continue its valid predecessor?  (predecessor = nearest lower valid lane, or the carry)
    unsigned long long vm = __ballot(valid);
    unsigned long long below = vm & ((1ull << l) - 1);
    int pl = below ? 63 - (int)__builtin_clzll(below) : -1;
    uint64_t pend = shfl64(p.ptr + p.len, pl < 0 ? 0 : pl);
    bool pplain = __shfl((int)(plain ? 1 : 0), pl < 0 ? 0 : pl) != 0;
    if (pl < 0) { pend = carry_end; pplain = have_carry; }
    bool cont = valid && plain && pplain && pend == p.ptr;
    bool head = valid && !cont;
    unsigned long long hm = __ballot(head);
    // run = head .. next head: total length by a segmented sum
    uint32_t hi = (uint32_t)__popcll(hm & ((2ull << l) - 1));       // 1-based index of my run among this batch's heads (0: continues the carry)
    // lengths: sum over the lanes of my run
    uint32_t len = valid ? p.len : 0u;
    uint32_t inc = wave_incl_scan(len);
    // run end lane = (next head lane) - 1 or 63
    unsigned long long after = hm & ~((2ull << l) - 1);
    int endl = after ? (int)__builtin_ctzll(after) - 1 : 63;
    uint32_t run_total = (uint32_t)__shfl((int)inc, endl) - (inc - len);
    wave_sync();                                                    // all lanes have loaded their piece before the table is overwritten
    if (head) { Piece q; q.ptr = p.ptr; q.len = run_total; q.rep = p.rep; t[nout + hi - 1] = q; }
    // pieces before the first head continue the carried piece
    uint32_t first_head = hm ? (uint32_t)__builtin_ctzll(hm) : 64u;
    uint32_t carry_add = first_head > 0 ? (uint32_t)__builtin_amdgcn_readlane((int)inc, (int)(first_head - 1)) : 0u;
    if (carry_add > 0 && l == 0 && nout > 0) t[nout - 1].len += carry_add;
    nout += (uint32_t)__popcll(hm);
    // new carry: the last valid piece of the batch
    if (vm) {
      int last = 63 - (int)__builtin_clzll(vm);
      carry_end = readlane64(p.ptr + p.len, (uint32_t)last);
      have_carry = __shfl((int)(plain ? 1 : 0), last) != 0;
    }
    wave_sync();
  }
  return nout;
}
// dst[0, total) = concatenation of the pieces; the caller allocated `total` = pieces_total() bytes
EH_DEV void wave_gather(bptr dst, const EH_G Piece* t, uint32_t n) {
  const int l = EH_LANE;
  uint64_t pos = 0;
  for (uint32_t base = 0; base < n; base += 64) {
    uint32_t idx = base + (uint32_t)l;
    uint64_t ptr = 0; uint32_t len = 0, rep = 1;
    if (idx < n) { Piece p = t[idx]; ptr = p.ptr; len = p.len; rep = p.rep; }
    uint64_t tl = (uint64_t)len * rep;
    // exclusive offsets inside the batch (64-bit: a repeated piece can be large)
    uint64_t inc = tl;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint64_t up = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), d) << 32) | (uint32_t)__shfl_up((int)(uint32_t)inc, d);
      if (l >= d) inc += up;
    }
    uint64_t off = pos + inc - tl;
    bool big = len > 48 || rep > 1;
    uint32_t maxs = wave_max(big ? 0u : len);
    cbptr sp = (cbptr)ptr;
    bptr dp = dst + off;
    for (uint32_t i0 = 0; i0 < maxs; i0 += 8) {
      uint8_t b[8];
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) if (!big && i0 + k < len) b[k] = sp[i0 + k];
#pragma unroll
      for (uint32_t k = 0; k < 8; k++) if (!big && i0 + k < len) dp[i0 + k] = b[k];
    }
    unsigned long long bm = __ballot(big && tl > 0);
    while (bm) {
      int j = (int)__builtin_ctzll(bm); bm &= bm - 1;
      uint64_t pj = readlane64(ptr, (uint32_t)j), oj = readlane64(off, (uint32_t)j);
      uint32_t lj = (uint32_t)__builtin_amdgcn_readlane((int)len, j), rj = (uint32_t)__builtin_amdgcn_readlane((int)rep, j);
      if (rj == 1) wave_copy(dst + oj, (cbptr)pj, lj);
      else wave_fill_periodic(dst + oj, (cbptr)pj, lj, (uint64_t)lj * rj);
    }
    pos += readlane64(inc, 63);
  }
}
// entries [a, b) of src appended to dst[*n ..)
EH_DEV void pieces_append(EH_G Piece* dst, uint32_t* n, const EH_G Piece* src, uint32_t a, uint32_t b) {
  if (b <= a) return;
  uint32_t cnt = b - a, at = *n;
  for (uint32_t i = EH_LANE; i < cnt; i += 64) dst[at + i] = src[a + i];
  *n = at + cnt;
}
// gathers pieces [a, b) into a fresh temporary of the work area; returns it as one piece (ptr, len)
EH_DEV bool pieces_materialize(Ctx& c, const EH_G Piece* t, uint32_t a, uint32_t b, bptr* out, uint32_t* outlen) {
  wave_sync();
  uint64_t tot = pieces_total(t + a, b - a);
  if (tot > 0xFFFFFFF0ull) { EH_SET_OVERFLOW(c, 201); return false; }
  bptr d = ws_alloc(c, tot ? tot : 16);
  if (!d) return false;
  wave_gather(d, t + a, b - a);
  wave_sync();
  *out = d; *outlen = (uint32_t)tot;
  return true;
}
// number of bytes of [p, p+n) that are not 0, 10, 13 or 32 (mutate_innertext, erlamsa_sgml.erl:675)
struct IsInk { EH_DEV bool operator()(uint32_t b, uint32_t) const { return b != 0 && b != 10 && b != 13 && b != 32; } };

// Muta([Bin], []) of a freshly scored mutator list - the nested scheduler call of base64_mutator (erlamsa_mutations.erl:669-670),
// erlamsa_sgml:mutate_innertext_prob/4 (:669-672) and erlamsa_json:mutate_innertext_prob/4 (:633-639) - is run by the scheduler
// itself: the mutator puts the table (lane i = entry i) into its MuFrame, sets c.call_req / call_bin / call_len / call_nfs and
// returns; it is called again with c.mu_phase == 1 and c.call_nres = the number of blocks of the resulting list, which sit at
// c.bl[c.nb .. c.nb + n), or -1 with c.status set (eh_device.h mux_fuzzers).
EH_DEV EH_G MuFrame* mu_frame(Ctx& c) {                                    // the running attempt's frame (made at its first nested run)
  if (c.mu) return c.mu;
  EH_G MuFrame* m = (EH_G MuFrame*)ws_alloc(c, sizeof(MuFrame));
  if (m && EH_LANE == 0) m->v[23] = 0;
  c.mu = m;
  return m;
}

__constant__ uint8_t c_def_pri[M_COUNT] = {10, 3, 1, 2, 1, 1, 1, 1, 3, 2, 2, 2, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 1, 2, 2, 7, 1, 1, 0};

// inner_mutations(sgml | json) (erlamsa_mutations.erl:1342-1356) + mutators_mutator/1 (:1387-1395):
// mutations([]) is evaluated first (2 draws, :1313-1314), the filtered table is folded into reverse
// table order, mutators_mutator draws rand(10) along that list and prepends => list in table order.
EH_DEV void inner_table(Ctx& c, bool json, EH_G MuFrame* mu, int* nfs) {
  const int l = EH_LANE;
  (void)rng_rand(c.rng, 3); (void)rng_rand(c.rng, 1);
  const uint8_t sg[11] = {M_AB, M_AD, M_NUM, M_BD, M_SD, M_LD, M_LRI, M_LR, M_LP, M_B64, M_URI};
  const uint8_t js[9] = {M_SGM, M_AB, M_AD, M_NUM, M_SP, M_SR, M_SD, M_B64, M_URI};
  int n = json ? 9 : 11;
  uint32_t name = 0;
#pragma unroll
  for (int k = 0; k < 11; k++) if (k == l) name = json ? (k < 9 ? js[k] : 0) : sg[k];
  uint32_t score = 0;
  if (l < n) { uint32_t v = (uint32_t)(rng_peek(c.rng, (uint32_t)(n - 1 - l) + 1) * 10.0); score = v < 2 ? 2 : v; }
  rng_skip(c.rng, (uint64_t)n);
  mu->pri[l] = l < n ? (uint32_t)c_def_pri[name] : 0;
  mu->meta[l] = em_pack(score, name, name, 3u);
  *nfs = n;
}

// base64_mutator/2 (erlamsa_mutations.erl:658-690): every text chunk longer than 6 that base64:decode/1 accepts
// is decoded, mutated once by a FRESH mutators_mutator over the whole default table (mutas_list(mutations([])):
// 2 draws for the table itself, then 41 score draws per chunk, list in REVERSE table order) and encoded again.
// The nested run is the scheduler's (mux_fuzzers, "nested scheduler calls without device recursion"): at a chunk the function leaves
// its loop state in its MuFrame and returns with call_req set; it is called again (mu_phase 1) with the result blocks.
__device__ __noinline__ int muta_b64(Ctx&, EH_G LexCache& lc) {
  EH_CTX;
  const int l = EH_LANE;
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  c.r_kind = R_SAME;
  EH_G LexChunk* tab = nullptr;
  int n = 0, base = 0, dacc = -1, d = 0;
  uint32_t snand_mask = 0, nout = 0, cap = 0, done_to = 0, a = 0, b = 0;
  unsigned long long cand = 0;
  EH_G Piece* out = nullptr;
  bool resume = c.mu_phase == 1;
  EH_G MuFrame* mu = c.mu;
  EH_PT0;
  if (!resume) {
    n = lex_cached(c, lc, H, L, &tab);
    if (n < 0) return 0;
    EH_PT(c, 48);                                                // eh_result_prof 48..53: lexing, candidates refused, decode, nested call, encode, gather
    snand_mask = rng_rand(c.rng, 3); (void)rng_rand(c.rng, 1);   // mutations([]) :661 -> :1313-1314
  } else {
    tab = (EH_G LexChunk*)uni64(mu->v[0]); n = (int)uni((uint32_t)mu->v[1]); base = (int)uni((uint32_t)mu->v[2]); cand = uni64(mu->v[3]);
    snand_mask = uni((uint32_t)mu->v[4]); nout = uni((uint32_t)mu->v[5]); cap = uni((uint32_t)mu->v[6]); done_to = uni((uint32_t)mu->v[7]);
    a = uni((uint32_t)mu->v[8]); b = uni((uint32_t)mu->v[9]); dacc = (int)uni((uint32_t)mu->v[10]); d = (int)uni((uint32_t)mu->v[11]);
    out = (EH_G Piece*)uni64(mu->v[12]);
  }
  // candidate chunks ({text, A} when length(A) > 6, :664) are picked 64 table entries at a time
  for (; base < n; base += 64) {
   int ti = base + l;
   uint32_t cty = 1, ca = 0, cb = 0;
   if (ti < n) { LexChunk e = tab[ti]; cty = e.type; ca = e.a; cb = e.b; }
   if (!resume) cand = __ballot(cty == 0 && cb - ca > 6);
   while (cand || resume) {
    if (!resume) {
      int cj = (int)__builtin_ctzll(cand); cand &= cand - 1;
      a = (uint32_t)__builtin_amdgcn_readlane((int)ca, cj); b = (uint32_t)__builtin_amdgcn_readlane((int)cb, cj);
      uint32_t nalpha, span;
      if (!b64_accepts(H + a, b - a, &nalpha, &span)) continue;    // error:badarg / function_clause :677-684
      uint32_t dl = b64_decoded_len(nalpha);
      if (!out) {                                                  // first hit: the piece list of unlex(Ms)
        cap = 2 * (uint32_t)(n - (base + cj)) + 4;
        out = (EH_G Piece*)ws_alloc(c, (uint64_t)cap * sizeof(Piece));
        if (!out) return 0;
      }
      if (!mu) { mu = mu_frame(c); if (!mu) return 0; }
      bptr dec = ws_alloc(c, (uint64_t)dl + 16);
      bptr pack = nalpha != span ? ws_alloc(c, (uint64_t)nalpha + 16) : nullptr;
      if (!dec || (nalpha != span && !pack)) return 0;
      EH_PT(c, 49);
      b64_decode_wave(H + a, span, nalpha, dec, pack);
      wave_sync();
      EH_PT(c, 50);
      d = rng_delta(c.rng);                                        // :666
      tr_ai(c, AT_base64_mutator, d);                              // [AddedMeta, {base64_mutator, D} | MAcc] :674: D's entry, then what the nested run adds
      // mutators_mutator(MutasList, []) :667: rand(10) per table entry in table order, each prepended
      uint32_t name = l < (int)M_COUNT ? (uint32_t)((int)M_COUNT - 1 - l) : 0;
      uint32_t score = 0;
      if (l < (int)M_COUNT) { uint32_t v = (uint32_t)(rng_peek(c.rng, name + 1) * 10.0); score = v < 2 ? 2 : v; }
      rng_skip(c.rng, (uint64_t)M_COUNT);
      mu->pri[l] = l < (int)M_COUNT ? (uint32_t)c_def_pri[name] : 0;
      mu->meta[l] = em_pack(score, name, name, name == M_SNAND ? snand_mask : 3u);
      if (l == 0) {
        mu->v[0] = (uint64_t)tab; mu->v[1] = (uint32_t)n; mu->v[2] = (uint32_t)base; mu->v[3] = cand; mu->v[4] = snand_mask; mu->v[5] = nout; mu->v[6] = cap;
        mu->v[7] = done_to; mu->v[8] = a; mu->v[9] = b; mu->v[10] = (uint32_t)dacc; mu->v[11] = (uint32_t)d; mu->v[12] = (uint64_t)out;
      }
      wave_sync();
      c.call_req = 1; c.call_bin = (uint64_t)dec; c.call_len = dl; c.call_nfs = (int)M_COUNT;      // Muta([Bin], []) :668
      return 0;
    }
    resume = false;
    const int nres = c.call_nres;
    if (nres < 0) return 0;
    EH_PT(c, 51);
    // NewBin = iolist_to_binary(NewLl) :669
    uint64_t tot = 0;
    for (int k = 0; k < nres; k++) tot += blk_load(c.bl, c.nb + k).len;
    if (tot > 0xBFFFFFF0ull) { EH_SET_OVERFLOW(c, 202); return 0; }
    bptr nb = ws_alloc(c, tot + 16);
    bptr enc = ws_alloc(c, (tot + 2) / 3 * 4 + 16);
    if (!nb || !enc) return 0;
    uint64_t o = 0;
    for (int k = 0; k < nres; k++) { Blk x = blk_load(c.bl, c.nb + k); wave_copy(nb + o, (cbptr)x.ptr, x.len); o += x.len; }
    wave_sync();
    b64_encode(nb, (uint32_t)tot, enc);
    wave_sync();
    piece_put(out, nout, H + done_to, a - done_to); nout++;
    piece_put(out, nout, enc, (uint32_t)((tot + 2) / 3 * 4)); nout++;
    done_to = b;
    dacc += d;
    EH_PT(c, 52);
   }
  }
  EH_PT(c, 49);
  if (!out) return -1;                                           // nothing decoded: unlex(lex(H)) =:= H
  piece_put(out, nout, H + done_to, L - done_to); nout++;
  wave_sync();
  uint64_t total = pieces_total(out, nout);
  if (total > 0xFFFFFFF0ull) { EH_SET_OVERFLOW(c, 203); return 0; }
  bptr dst = ws_alloc(c, total ? total : 16);
  if (!dst) return 0;
  wave_gather(dst, out, nout);
  wave_sync();
  EH_PT(c, 53);
  c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)total;
  return dacc;
}

}  // namespace eh
