// eh_lex.h — device code for erlamsa_strlex:lex/1 (erlamsa_strlex.erl:75-142) and the mutators
// built on it: ab / ad (erlamsa_mutations.erl:430-651), uri (:696-784), b64 (:658-690, decode
// probe), zip (:1149-1163, archive probe).
//
// The lexer is a sequential automaton with 6 bytes of lookahead; it runs wave-uniform over a
// 256-byte register window (one coalesced dword load per lane, bytes fetched with v_readlane),
// so the scan itself never waits on memory.  The chunk table is cached per block: the four
// lexing mutators are usually tried on the same block within one mux_fuzzers call.
#pragma once
#include "eh_text.h"
#include "eh_mask.h"

namespace eh {

// ---------------------------------------------------------------------------------------------
// forward-only byte reader over a register window
// ---------------------------------------------------------------------------------------------
struct ByteReader {
  cbptr p; uint32_t n; uint32_t wbase; uint32_t w; bool valid;
};
EH_DEV void br_init(ByteReader& r, cbptr p, uint32_t n) { r.p = p; r.n = n; r.wbase = 0; r.w = 0; r.valid = false; }
EH_DEV void br_fill(ByteReader& r, uint32_t from) {
  r.wbase = from;
  uint32_t off = from + 4u * (uint32_t)EH_LANE;
  uint32_t v = 0;
  if (off + 4 <= r.n) v = ldg4(r.p + off);
  else { for (uint32_t k = 0; k < 4; k++) if (off + k < r.n) v |= (uint32_t)r.p[off + k] << (8 * k); }
  r.w = v; r.valid = true;
}
// byte at i (i < n); `hint` = lowest position that will still be needed (window start on refill)
EH_DEV uint32_t br_get(ByteReader& r, uint32_t i, uint32_t hint) {
  if (!r.valid || i < r.wbase || i >= r.wbase + 256) br_fill(r, hint);
  uint32_t d = i - r.wbase;
  uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)r.w, (int)uni(d >> 2));
  return (word >> (8 * (d & 3))) & 255u;
}

EH_DEV bool texty(uint32_t b) {                                 // erlamsa_strlex.erl:45-52
  if (b < 9) return false;
  if (b > 126) return false;
  if (b > 31) return true;
  return b == 9 || b == 10 || b == 13;
}

// chunk table entry: type 0 text, 1 byte, 2 delimited; [a,b) byte range in the block
// (for delimited: a = opening quote, b = one past the closing quote)
struct LexChunk { uint32_t type, a, b; };

// Event-driven lexer.  Classes: 0 texty, 1 '"', 2 "'", 3 backslash.  The automaton of
// erlamsa_strlex (string_lex_step / step_text / step_delimited) only changes state at quotes,
// backslashes, non-texty bytes and at positions where texty_enough/1 flips, so it hops between
// those with mask lookups.
struct LexCls { EH_DEV uint32_t operator()(uint32_t b) const { return (texty(b) ? 1u : 0u) | (b == 34 ? 2u : 0u) | (b == 39 ? 4u : 0u) | (b == 92 ? 8u : 0u); } };
struct LexWin { MaskWin<4> w; uint64_t nt, te; };
EH_DEV void lw_load(LexWin& x, uint32_t base) {
#ifdef EH_MW_REF
  mw_load_ref(x.w, base, LexCls());
#else
  mw_load(x.w, base, LexCls());
#endif
  uint64_t T = x.w.m[0];
  x.nt = ~T & x.w.inrange;
  // texty_enough (:54-64): the next 6 bytes are texty, running off the end counts as texty
  uint64_t Tp = T | ~x.w.inrange;
  uint64_t nx = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(Tp >> 32), 1) << 32) | (uint32_t)__shfl_down((int)(uint32_t)Tp, 1);
  if (EH_LANE == 63) nx = ~0ull;
  uint64_t te = Tp;
#pragma unroll
  for (int d = 1; d <= 5; d++) te &= (Tp >> d) | (nx << (64 - d));
  x.te = te & x.w.inrange;
}
// Lexes H[0,L) into tab (capacity cap); returns the number of chunks or -1 on table overflow.
// One explicit state machine with a SINGLE window-load site, so that the event loop stays a few
// hundred bytes of code (an earlier version inlined the window load at every lookup: 60 KB of code,
// instruction-cache misses on every hop).
//   ST_STEP   string_lex_step (:79-97): is texty_enough at pos?
//   ST_RAW    raw bytes: hop to the next position where texty_enough holds
//   ST_TEXT   step_text (:99-112): hop to the next quote or non-texty byte
//   ST_DELIM  step_delimited (:114-142): hop to the next closing quote, backslash or non-texty byte
__device__ __noinline__ int lex_block(cbptr H, uint32_t L, EH_G LexChunk* tab, uint32_t cap) {
  const int l = EH_LANE;
  LexWin x; x.w.p = H; x.w.L = L; x.w.valid = false; x.w.base = 0;
  uint32_t pos = 0, n = 0; uint32_t raw_start = 0xFFFFFFFFu;
  // chunk records are staged in lane registers (record k of the batch in lane k) and written 64 at a
  // time: a store per chunk would make the next window load wait for it
  uint32_t bt = 0, ba = 0, bb = 0, nb = 0;
  auto flush = [&]() {
    uint32_t first = n - nb;
    if ((uint32_t)l < nb && first + (uint32_t)l < cap) { tab[first + l].type = bt; tab[first + l].a = ba; tab[first + l].b = bb; }
    nb = 0;
  };
  auto emit = [&](uint32_t type, uint32_t a, uint32_t b) {
    if ((uint32_t)l == nb) { bt = type; ba = a; bb = b; }
    nb++; n++;
    if (nb == 64) flush();
  };
  enum { ST_STEP, ST_RAW, ST_TEXT, ST_DELIM };
  int state = ST_STEP; uint32_t seen = 0, q = 0, qcls = 1;
  while (pos < L) {
    if (!x.w.valid || pos < x.w.base || pos >= x.w.base + MW_STEP) lw_load(x, pos & ~63u);
    const uint32_t rel = pos - x.w.base;
    if (state == ST_STEP) {
      if (!mw_test(x.te, rel)) { if (raw_start == 0xFFFFFFFFu) raw_start = pos; pos++; state = ST_RAW; }
      else { if (raw_start != 0xFFFFFFFFu) { emit(1, raw_start, pos); raw_start = 0xFFFFFFFFu; } seen = pos; state = ST_TEXT; }
    } else if (state == ST_RAW) {
      uint32_t r = mw_next(x.te, rel);
      if (r >= MW_STEP) pos = x.w.base + MW_STEP; else { pos = x.w.base + r; state = ST_STEP; }
    } else if (state == ST_TEXT) {
      uint32_t r = mw_next(x.w.m[1] | x.w.m[2] | x.nt, rel);
      if (r >= MW_STEP) { pos = x.w.base + MW_STEP; continue; }
      pos = x.w.base + r;
      if (!mw_test(x.w.m[0], r)) { emit(0, seen, pos); state = ST_STEP; }     // non-texty byte ends the text chunk
      else { qcls = mw_test(x.w.m[1], r) ? 1u : 2u; q = pos; pos++; state = ST_DELIM; }
    } else {
      uint64_t quote = qcls == 1 ? x.w.m[1] : x.w.m[2];
      uint32_t r = mw_next(quote | x.w.m[3] | x.nt, rel);
      if (r >= MW_STEP) { pos = x.w.base + MW_STEP; continue; }
      uint32_t p2 = x.w.base + r;
      if (mw_test(quote, r)) {                                   // closing quote
        if (q > seen) emit(0, seen, q);
        emit(2, q, p2 + 1);
        pos = p2 + 1; state = ST_STEP;
      } else if (mw_test(x.w.m[3], r)) {                         // backslash: skips the next byte when that is texty
        if (p2 + 1 >= L) pos = p2 + 1;                           // (r + 1 is inside the lookahead word)
        else pos = mw_test(x.w.m[0], r + 1) ? p2 + 2 : p2 + 1;
      } else { emit(0, seen, p2); pos = p2; state = ST_STEP; }   // non-texty byte: the whole thing was text
    }
  }
  // the block ended inside a chunk
  if (state == ST_TEXT || state == ST_DELIM) emit(0, seen, L);
  if (raw_start != 0xFFFFFFFFu) emit(1, raw_start, L);
  flush();
  wave_sync();
  return n <= cap ? (int)n : -1;
}

// returns chunk count (>= 0) and *tab, or -1 after setting c.status
EH_DEV int lex_cached(Ctx& c, EH_G LexCache& lc, cbptr H, uint32_t L, EH_G LexChunk** tab) {
  if (c.lex_ptr[c.depth] == (uint64_t)H && lc.n >= 0 && lc.ptr == (uint64_t)H && lc.len == L) { *tab = lc.tab; return lc.n; }
  uint32_t cap = L + 2 < (1u << 18) ? L + 2 : (1u << 18);   // every chunk covers >= 1 byte
  // the previous table is reused when it is large enough (a case that keeps lexing changing blocks used to leave one
  // table per miss behind at the top of its work area)
  uint32_t have = uni(lc.tcap);
  EH_G LexChunk* t = have >= cap ? (EH_G LexChunk*)uni64((uint64_t)lc.tab) : (EH_G LexChunk*)ws_alloc_top(c, (uint64_t)cap * sizeof(LexChunk));
  if (!t) return -1;
  wave_sync();
  if (EH_LANE == 0) { lc.n = -1; if (have < cap) { lc.tab = t; lc.tcap = cap; } }
  c.lex_ptr[c.depth] = 0;
  wave_sync();
  int n = lex_block(H, L, t, cap);
  if (n < 0) { EH_SET_OVERFLOW(c, 502); return -1; }
  wave_sync();
  if (EH_LANE == 0) { lc.ptr = (uint64_t)H; lc.len = L; lc.n = n; }
  *tab = t; c.lex_ptr[c.depth] = (uint64_t)H;
  wave_sync();
  return n;
}

// ---------------------------------------------------------------------------------------------
// literal tables (erlamsa_mutations.erl:444-466)
// ---------------------------------------------------------------------------------------------
__constant__ uint8_t c_silly[12][9] = {{2, '%', 'n'}, {2, '%', 'n'}, {2, '%', 's'}, {2, '%', 'd'}, {2, '%', 'p'}, {3, '%', '#', 'x'}, {1, 0},
                                       {8, 'a', 'a', 'a', 'a', '%', 'd', '%', 'n'}, {1, 10}, {1, 13}, {1, 9}, {1, 8}};
__constant__ uint8_t c_delims[21] = {'\'', '"', '\'', '"', '\'', '"', '&', ':', '|', ';', '\\', 10, 13, 9, ' ', '`', 0, ']', '[', '>', '<'};
// shellinjects(): prefix/suffix around ~s ; revconnects(): pieces around ~s and ~p
__constant__ char c_inj_pre[10][8] = {"';", "\";", ";", "|", "^ ", "& ", "&& ", "|| ", "%0D", "`"};
__constant__ char c_inj_suf[10][8] = {";'", ";\"", ";", "#", " ^", " &", " &&", " ||", "%0D", "`"};
__constant__ char c_rev_a[7][24] = {"calc.exe & notepad.exe ", "nc ", "wget http://", "curl ", "exec 3<>/dev/tcp/", "sleep 100000 # ", "echo>/tmp/erlamsa."};
__constant__ char c_rev_b[7][4] = {" ", " ", ":", " ", "/", " ", "."};
__constant__ char c_rev_c[7][4] = {" ", "", "", "", "", " ", ""};

// lane-0 helpers writing C strings
__device__ inline uint32_t put_str(bptr o, uint32_t pos, const char* s) { while (*s) o[pos++] = (uint8_t)*s++; return pos; }

// random_badness/0 (:468-476): N = rand(20)+1 silly strings, each PREPENDED.  Returns length.
EH_DEV uint32_t random_badness(Ctx& c, bptr buf /* >= 168 bytes */) {
  uint32_t n = rng_rand(c.rng, 20) + 1;
  uint32_t idx[20];
#pragma unroll
  for (int i = 0; i < 20; i++) idx[i] = (uint32_t)i < n ? rng_rand(c.rng, 12) : 0;
  uint32_t total = 0;
  if (EH_LANE == 0) {
    // output order: last drawn first
    uint32_t pos = 0;
#pragma unroll
    for (int i = 19; i >= 0; i--) if ((uint32_t)i < n) { uint32_t e = idx[i]; uint32_t ln = c_silly[e][0]; for (uint32_t k = 0; k < ln; k++) buf[pos++] = c_silly[e][1 + k]; }
    total = pos;
  }
  wave_sync();
  return uni(total);
}
EH_DEV uint32_t rand_as_count(Ctx& c) {                        // :485-499
  uint32_t t = rng_rand(c.rng, 11);
  switch (t) {
    case 0: return 127; case 1: return 128; case 2: return 255; case 3: return 256; case 4: return 16383;
    case 5: return 16384; case 6: return 32767; case 7: return 32768; case 8: return 65535; case 9: return 65536;
  }
  return rng_rand(c.rng, 1024);
}
// buildrevconnect/0 (:514-519); returns length written to buf (lane 0)
EH_DEV uint32_t buildrevconnect(Ctx& c, bptr buf) {
  uint32_t inj = rng_rand(c.rng, 10), rev = rng_rand(c.rng, 7);
  uint32_t total = 0;
  if (EH_LANE == 0) {
    uint32_t pos = 0;
    pos = put_str(buf, pos, c_inj_pre[inj]);
    pos = put_str(buf, pos, c_rev_a[rev]); pos = put_str(buf, pos, c.p->cfg.ssrf_host);
    pos = put_str(buf, pos, c_rev_b[rev]); pos = put_str(buf, pos, c.p->cfg.ssrf_port);
    pos = put_str(buf, pos, c_rev_c[rev]);
    pos = put_str(buf, pos, c_inj_suf[inj]);
    total = pos;
  }
  wave_sync();
  return uni(total);
}

enum TextMuta { T_INSERT_BADNESS, T_REPLACE_BADNESS, T_INSERT_TRAVERSAL, T_INSERT_AAAS, T_INSERT_NULL, T_INSERT_DELIMETER, T_INSERT_SHELLINJ };

// mutate_text/2 (:521-563) on the content range [cs,ce) of H; builds the whole new block.
EH_DEV void mutate_text_emit(Ctx& c, int tm, cbptr H, uint32_t L, uint32_t cs, uint32_t ce) {
  uint32_t n = ce - cs;                                         // length(Lst)
  bptr lit = ws_alloc(c, 512);
  if (!lit) return;
  Pieces q; pc_init(q);
  pc_add(q, H, cs);
  switch (tm) {
    case T_INSERT_BADNESS: {
      if (n == 0) { uint32_t bl = random_badness(c, lit); pc_add(q, lit, bl); break; }
      uint32_t P = rng_erand(c.rng, n); uint32_t bl = random_badness(c, lit);
      pc_add(q, H + cs, P - 1); pc_add(q, lit, bl); pc_add(q, H + cs + P - 1, n - (P - 1)); break;
    }
    case T_REPLACE_BADNESS: {
      if (n == 0) { uint32_t bl = random_badness(c, lit); pc_add(q, lit, bl); break; }
      uint32_t P = rng_erand(c.rng, n); uint32_t bl = random_badness(c, lit);
      // sublist(Lst,P-1) ++ overwrite(nthtail(P,Lst), Bad): the old tail wins, Bad only sticks out past it
      uint32_t tail = n - P;
      pc_add(q, H + cs, P - 1); pc_add(q, H + cs + P, tail);
      if (bl > tail) pc_add(q, lit + tail, bl - tail);
      break;
    }
    case T_INSERT_AAAS: {
      if (n == 0) { uint32_t cnt = rand_as_count(c); if (EH_LANE == 0) lit[0] = 97; wave_sync(); pc_add(q, lit, 1, cnt); break; }
      uint32_t cnt = rand_as_count(c); uint32_t P = rng_erand(c.rng, n);
      if (EH_LANE == 0) lit[0] = 97;
      wave_sync();
      pc_add(q, H + cs, P - 1); if (cnt) pc_add(q, lit, 1, cnt); pc_add(q, H + cs + P, n - P); break;
    }
    case T_INSERT_TRAVERSAL: {
      uint32_t P = 0; uint32_t sym = '/';
      if (n > 0) { P = rng_erand(c.rng, n); sym = rng_rand(c.rng, 2) == 0 ? '\\' : '/'; }
      uint32_t k = rng_erand(c.rng, 10);                         // insert_traversal/1 :506-508
      if (EH_LANE == 0) { lit[0] = (uint8_t)sym; lit[1] = '.'; lit[2] = '.'; lit[3] = (uint8_t)sym; }
      wave_sync();
      if (n == 0) { pc_add(q, lit, 1); pc_add(q, lit + 1, 3, k); break; }
      pc_add(q, H + cs, P - 1); pc_add(q, lit, 1); pc_add(q, lit + 1, 3, k); pc_add(q, H + cs + P, n - P); break;
    }
    case T_INSERT_NULL: {
      if (EH_LANE == 0) lit[0] = 0;
      wave_sync();
      pc_add(q, H + cs, n); pc_add(q, lit, 1); break;
    }
    case T_INSERT_DELIMETER: {
      uint32_t P = n > 0 ? rng_erand(c.rng, n) : 0;
      uint32_t d = rng_rand(c.rng, 21);
      if (EH_LANE == 0) lit[0] = c_delims[d];
      wave_sync();
      if (n == 0) { pc_add(q, lit, 1); break; }
      pc_add(q, H + cs, P - 1); pc_add(q, lit, 1); pc_add(q, H + cs + P - 1, n - (P - 1)); break;
    }
    case T_INSERT_SHELLINJ: {
      if (n == 0) { uint32_t d = rng_rand(c.rng, 21); if (EH_LANE == 0) lit[0] = c_delims[d]; wave_sync(); pc_add(q, lit, 1); break; }
      uint32_t P = rng_erand(c.rng, n);
      uint32_t sl = buildrevconnect(c, lit);
      pc_add(q, H + cs, P - 1); pc_add(q, lit, sl); pc_add(q, H + cs + P - 1, n - (P - 1)); break;
    }
  }
  // the rest of the block after the content range
  if (q.k < 6) pc_add(q, H + ce, L - ce);
  else {  // 6 pieces already (insert_traversal): emit in two steps
    Pieces q2; pc_init(q2);
    if (!pc_emit(c, q)) return;
    bptr first = c.r_ptr; uint32_t fl = c.r_len;
    pc_add(q2, first, fl); pc_add(q2, H + ce, L - ce);
    pc_emit(c, q2);
    return;
  }
  pc_emit(c, q);
}

// construct_ascii_mutator (:585-602) with string_generic_mutate (ab, :571-583) or
// string_delimeter_mutate (ad, :626-644)
__device__ __noinline__ int muta_ascii(Ctx&, EH_G LexCache& lc, int fn) {
  EH_CTX;
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  c.r_kind = R_SAME;
  EH_G LexChunk* tab;
  int n = lex_cached(c, lc, H, L, &tab);
  if (n < 0) return 0;
  // stringy/1 :438-442
  uint32_t nontext = 0;
  for (int i = EH_LANE; i < n; i += 64) if (tab[i].type != 1) nontext = 1;
  c.m_aux = 0;
  if (__ballot(nontext != 0) == 0) return -1;                     // {Ascii_mutator, Ll, Meta, -1} :600-601
  c.m_aux = 1;                                                    // [{Name, D} | Meta] :598
  // R > L/4 -> give up  (R starts at 0; compares R > n/4 as floats)
  for (uint32_t r = 0; !((double)r > (double)n / 4.0); r++) {
    uint32_t P = rng_erand(c.rng, (uint32_t)n);
    LexChunk e = tab[P - 1];
    uint32_t ty = uni(e.type), a = uni(e.a), b = uni(e.b);
    if (ty == 1) continue;
    if (fn == M_AB) {
      const int tms[5] = {T_INSERT_BADNESS, T_REPLACE_BADNESS, T_INSERT_TRAVERSAL, T_INSERT_AAAS, T_INSERT_NULL};
      uint32_t k = rng_rand(c.rng, 5);                           // mutate_text_data: rand_elem(TxtMutators)
      int tm = T_INSERT_BADNESS;
#pragma unroll
      for (int t = 0; t < 5; t++) if ((uint32_t)t == k) tm = tms[t];
      if (ty == 0) mutate_text_emit(c, tm, H, L, a, b); else mutate_text_emit(c, tm, H, L, a + 1, b - 1);
    } else {
      if (ty == 0) {
        uint32_t k = rng_rand(c.rng, 4);                         // [insert_delimeter x3, insert_shellinj]
        (void)rng_rand(c.rng, 1);                                // rand_elem over the 1-element list
        mutate_text_emit(c, k == 3 ? T_INSERT_SHELLINJ : T_INSERT_DELIMETER, H, L, a, b);
      } else {
        uint32_t dr = rng_rand(c.rng, 4);                        // drop_delimeter/2 :615-622
        Pieces q; pc_init(q);
        if (dr == 0) { pc_add(q, H, b - 1); pc_add(q, H + b, L - b); pc_emit(c, q); }          // drop right
        else if (dr == 1) { pc_add(q, H, a); pc_add(q, H + a + 1, L - a - 1); pc_emit(c, q); }  // drop left
        else if (dr == 2) { pc_add(q, H, a); pc_add(q, H + a + 1, b - a - 2); pc_add(q, H + b, L - b); pc_emit(c, q); }
        // 3: unchanged
      }
    }
    break;
  }
  if (c.status != CASE_OK) return 0;
  return rng_delta(c.rng);
}

// 4-byte history predicate scan (for "://" and the zip EOCD signature)
template <class Pred>
EH_DEV uint32_t tile_mask_h(cbptr p, uint32_t n, uint32_t tile_base, Pred pred) {
  uint32_t i0 = tile_base + 16u * (uint32_t)EH_LANE;
  if (i0 >= n) return 0;
  uint32_t cnt = n - i0 < 16 ? n - i0 : 16;
  uint32_t hist = 0xFFFFFF00u;                                  // [b-3,b-2,b-1] in bits 31..8 after shifting
  for (uint32_t k = 3; k >= 1; k--) hist = (hist << 8) | (i0 >= k ? (uint32_t)p[i0 - k] : 0xFFu);
  // hist low 24 bits now = b-3,b-2,b-1 (b-1 in the low byte)
  uint32_t m = 0;
  for (uint32_t k = 0; k < cnt; k++) {
    uint32_t b = p[i0 + k];
    if (pred(b, hist & 0xFFFFFFu)) m |= 1u << k;
    hist = (hist << 8) | b;
  }
  return m;
}
struct IsUriSep { EH_DEV bool operator()(uint32_t b, uint32_t h) const { return b == '/' && (h & 0xFFFF) == ((uint32_t)':' << 8 | '/'); } };   // last byte of "://"
struct IsEocd { EH_DEV bool operator()(uint32_t b, uint32_t h) const { return b == 0x06 && h == (0x50u << 16 | 0x4Bu << 8 | 0x05u); } };

// has an end-of-central-directory signature within the last 22+65535 bytes (zip:foldl fails with
// bad_eocd otherwise)
EH_DEV bool has_zip_eocd(cbptr H, uint32_t L) {
  if (L < 22) return false;
  uint32_t lo = L > 22 + 65535 ? L - 22 - 65535 : 0;
  uint32_t found = 0;
  for (uint32_t tb = lo & ~1023u; tb < L; tb += 1024) {
    uint32_t m = tile_mask_h(H, L, tb, IsEocd());
    // signature ends at position e (byte 0x06): start = e-3 must satisfy lo <= start <= L-22
    uint32_t mm = m;
    while (mm) { uint32_t bit = (uint32_t)__builtin_ctz(mm); mm &= mm - 1; uint32_t e = tb + 16u * (uint32_t)EH_LANE + bit; if (e >= 3 && e - 3 >= lo && e - 3 + 22 <= L) found = 1; }
  }
  return __ballot(found != 0) != 0;
}
// base64:decode/1 acceptance (stdlib, restated in oracle/otp_compat.h): groups of four sextets, "xx==" / "xxx="
// tails, white space skipped anywhere, only white space after the padding.  Wave-parallel and without decoding:
// 64 bytes per step are classified (alphabet, white space, '=', other) with ballots; everything before the first
// '=' must be alphabet or white space, the number of alphabet characters there decides which padding is legal.
// Almost every text chunk fails within its first step, which is what keeps b64 cheap as a failing probe.
// On acceptance *nalpha = alphabet characters before the padding and *span = the bytes they sit in (offset of the
// first '=' or n): the decoded length is nalpha / 4 * 3 + {0, -, 1, 2}[nalpha % 4].
EH_DEV bool b64_accepts(cbptr t, uint32_t n, uint32_t* nalpha_out, uint32_t* span_out) {
  const int l = EH_LANE;
  uint32_t nalpha = 0, eqpos = 0xFFFFFFFFu;
  for (uint32_t base = 0; base < n && eqpos == 0xFFFFFFFFu; base += 64) {
    uint32_t i = base + (uint32_t)l; bool in = i < n;
    uint32_t ch = in ? t[i] : 32u;
    bool ws = ch == 9 || ch == 10 || ch == 13 || ch == 32;
    bool al = (ch >= 'A' && ch <= 'Z') || (ch >= 'a' && ch <= 'z') || (ch >= '0' && ch <= '9') || ch == '+' || ch == '/';
    unsigned long long em = __ballot(in && ch == '='), om = __ballot(in && !ws && !al && ch != '='), am = __ballot(in && al);
    unsigned long long upto = em ? ((1ull << __builtin_ctzll(em)) - 1) : ~0ull;   // lanes before the first '='
    if (om & upto) return false;
    nalpha += (uint32_t)__popcll(am & upto);
    if (em) eqpos = base + (uint32_t)__builtin_ctzll(em);
  }
  uint32_t q = nalpha & 3u;
  *nalpha_out = nalpha; *span_out = eqpos == 0xFFFFFFFFu ? n : eqpos;
  if (eqpos == 0xFFFFFFFFu) return q == 0;
  if (q != 2 && q != 3) return false;
  // after the first '=': q == 2 needs one more '=' (white space may sit in between), then only white space
  uint32_t need = q == 2 ? 1u : 0u; bool ok = true;
  for (uint32_t base = eqpos + 1; base < n && ok; base += 64) {
    uint32_t i = base + (uint32_t)l; bool in = i < n;
    uint32_t ch = in ? t[i] : 32u;
    bool ws = ch == 9 || ch == 10 || ch == 13 || ch == 32;
    unsigned long long nm = __ballot(in && !ws);                 // non white space
    while (nm && ok) {
      int j = (int)__builtin_ctzll(nm); nm &= nm - 1;
      uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)ch, j);
      if (need && cj == '=') need = 0; else ok = false;
    }
  }
  return ok && need == 0;
}
EH_DEV uint32_t b64_decoded_len(uint32_t nalpha) { uint32_t q = nalpha & 3u; return nalpha / 4 * 3 + (q == 2 ? 1u : (q == 3 ? 2u : 0u)); }
EH_DEV uint32_t b64_sextet(uint32_t ch) {
  return ch >= 'a' ? ch - 'a' + 26 : (ch >= 'A' ? ch - 'A' : (ch >= '0' ? ch - '0' + 52 : (ch == '+' ? 62u : 63u)));
}
// The bytes of an accepted chunk (b64_accepts: nalpha alphabet characters within t[0, span), the rest of the span white
// space).  A chunk with white space inside is first packed into pack[0, nalpha) (ballot ranks, 64 bytes per step); then
// lane g turns the g-th group of four characters into three bytes, the last lane the "xx" / "xxx" tail.  dst holds
// b64_decoded_len(nalpha) bytes.  (Round 3 decoded on lane 0, byte by byte: the longest case of the bench workload
// spent 9 of its 9.2 Gcyc there, on megabytes of repeated base64 lines - profiles/r04_heaviest_cases.txt.)
EH_DEV void b64_decode_wave(cbptr t, uint32_t span, uint32_t nalpha, bptr dst, bptr pack) {
  const int l = EH_LANE;
  cbptr src = t;
  if (nalpha != span) {
    uint32_t cnt = 0;
    for (uint32_t base = 0; base < span; base += 64) {
      uint32_t i = base + (uint32_t)l; bool in = i < span;
      uint32_t ch = in ? t[i] : 32u;
      bool al = in && !(ch == 9 || ch == 10 || ch == 13 || ch == 32);
      unsigned long long am = __ballot(al);
      if (al) pack[cnt + (uint32_t)__popcll(am & ((1ull << l) - 1))] = (uint8_t)ch;
      cnt += (uint32_t)__popcll(am);
    }
    wave_sync();
    src = pack;
  }
  uint32_t ng = nalpha / 4, q = nalpha & 3u;
  for (uint32_t g = (uint32_t)l; g < ng; g += 64) {
    uint32_t v = (b64_sextet(src[4 * g]) << 18) | (b64_sextet(src[4 * g + 1]) << 12) | (b64_sextet(src[4 * g + 2]) << 6) | b64_sextet(src[4 * g + 3]);
    dst[3 * g] = (uint8_t)(v >> 16); dst[3 * g + 1] = (uint8_t)(v >> 8); dst[3 * g + 2] = (uint8_t)v;
  }
  if (l == 63 && q >= 2) {
    uint32_t v = (b64_sextet(src[4 * ng]) << 6) | b64_sextet(src[4 * ng + 1]);        // 12 bits
    if (q == 2) dst[3 * ng] = (uint8_t)(v >> 4);
    else { v = (v << 6) | b64_sextet(src[4 * ng + 2]); dst[3 * ng] = (uint8_t)(v >> 10); dst[3 * ng + 1] = (uint8_t)(v >> 2); }
  }
}
// base64:encode_to_string/1: lane g encodes the g-th 3-byte group
EH_DEV void b64_encode(cbptr src, uint32_t n, bptr dst) {
  uint32_t ng = (n + 2) / 3;
  for (uint32_t g = EH_LANE; g < ng; g += 64) {
    uint32_t i = 3 * g, rem = n - i;
    uint32_t b0 = src[i], b1 = rem > 1 ? src[i + 1] : 0, b2 = rem > 2 ? src[i + 2] : 0;
    uint32_t v = (b0 << 16) | (b1 << 8) | b2;
    auto ch = [](uint32_t x) -> uint8_t { return (uint8_t)(x < 26 ? 'A' + x : (x < 52 ? 'a' + (x - 26) : (x < 62 ? '0' + (x - 52) : (x == 62 ? '+' : '/')))); };
    dst[4 * g] = ch((v >> 18) & 63); dst[4 * g + 1] = ch((v >> 12) & 63);
    dst[4 * g + 2] = rem > 1 ? ch((v >> 6) & 63) : (uint8_t)'=';
    dst[4 * g + 3] = rem > 2 ? ch(v & 63) : (uint8_t)'=';
  }
}
// uri_mutator :770-784 (+ try_uri_mutate :760-768, rand_uri_mutate :737-758)
__device__ __noinline__ int muta_uri(Ctx&, EH_G LexCache& lc) {
  EH_CTX;
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  c.r_kind = R_SAME;
  EH_G LexChunk* tab;
  int n = lex_cached(c, lc, H, L, &tab);
  if (n < 0) return 0;
  // positions of the last byte of every "://" in the block
  uint32_t nsep = 0;
  for (uint32_t tb = 0; tb < L; tb += 1024) nsep += __popc(tile_mask_h(H, L, tb, IsUriSep()));
  nsep = wave_sum(nsep);
  if (nsep == 0) return -1;
  // rare path: walk chunks on lane 0, build the new block byte by byte
  const DevConfig& cfg = c.p->cfg;
  uint32_t hostlen = 0; while (cfg.ssrf_host[hostlen]) hostlen++;
  uint64_t bound = (uint64_t)L + (uint64_t)nsep * (hostlen + 96) + 64;
  bptr dst = ws_alloc(c, bound);
  if (!dst) return 0;
  // draws are wave-uniform: iterate chunks uniformly, let lane 0 write
  uint32_t out = 0; int dacc = -1; bool crashed = false;
  auto copy_range = [&](uint32_t a, uint32_t b) { wave_copy(dst + out, H + a, b - a); out += b - a; };
  for (int i = 0; i < n && !crashed; i++) {
    LexChunk e = tab[i];
    uint32_t ty = uni(e.type), a = uni(e.a), b = uni(e.b);
    if (ty != 0 || b - a <= 5) { copy_range(a, b); continue; }
    // first "://" inside [a,b)
    uint32_t sp = 0xFFFFFFFFu;
    if (EH_LANE == 0) { for (uint32_t k = a; k + 2 < b; k++) if (H[k] == ':' && H[k + 1] == '/' && H[k + 2] == '/') { sp = k; break; } }
    sp = uni((uint32_t)__shfl((int)sp, 0));
    if (sp == 0xFFFFFFFFu) { copy_range(a, b); continue; }
    uint32_t tstart = sp + 3;                                    // T = rest of the chunk
    uint32_t mode = rng_erand(c.rng, 3);
    bool file_scheme = sp - a >= 4 && uni(H[sp - 4]) == 'f' && uni(H[sp - 3]) == 'i' && uni(H[sp - 2]) == 'l' && uni(H[sp - 1]) == 'e';
    // Domain / Query tokens of T (string:tokens(T, "/"))
    uint32_t dom_a = 0, dom_b = 0; bool have_dom = false;
    if (mode != 1) {
      uint32_t da = 0, db = 0, hv = 0;
      if (EH_LANE == 0) { uint32_t k = tstart; while (k < b && H[k] == '/') k++; if (k < b) { hv = 1; da = k; while (k < b && H[k] != '/') k++; db = k; } }
      dom_a = uni((uint32_t)__shfl((int)da, 0)); dom_b = uni((uint32_t)__shfl((int)db, 0)); have_dom = uni((uint32_t)__shfl((int)hv, 0)) != 0;
    }
    uint32_t at_sp = 0, ntrav = 0, which = 0;
    if (mode == 2) at_sp = rng_rand(c.rng, 2);                   // rand_elem([" @~s:~p", "@~s:~p"]) precedes the token match
    if (mode != 1 && !have_dom) { crashed = true; break; }       // [Domain | Query] = [] -> badmatch
    if (mode == 3) { ntrav = rng_erand(c.rng, 10); which = rng_erand(c.rng, 4); }
    wave_sync();
    uint32_t newout = out;
    if (EH_LANE == 0) {
      uint32_t o = out;
      auto putb = [&](uint8_t v) { dst[o++] = v; };
      auto puts = [&](const char* s) { while (*s) dst[o++] = (uint8_t)*s++; };
      auto putr = [&](uint32_t x, uint32_t y) { for (uint32_t k = x; k < y; k++) dst[o++] = H[k]; };
      auto put_query_joined = [&]() {                            // string:join(Query, "/")
        uint32_t k = dom_b; bool first = true;
        while (k < b) { while (k < b && H[k] == '/') k++; if (k >= b) break; if (!first) putb('/'); first = false; while (k < b && H[k] != '/') putb(H[k++]); }
      };
      if (mode == 1) {
        if (file_scheme) { putr(a, sp - 4); puts("http"); } else putr(a, sp);            // change_scheme :733-735
        puts("://"); puts(cfg.ssrf_host); putb(':'); puts(cfg.ssrf_port); putb('/');   // get_ssrf_uri :727-731
        putr(tstart, b);
      } else if (mode == 2) {
        if (file_scheme) { putr(a, sp - 4); puts("http"); } else putr(a, sp);
        puts("://"); putr(dom_a, dom_b);
        if (at_sp == 0) putb(' ');
        putb('@'); puts(cfg.ssrf_host); putb(':'); puts(cfg.ssrf_port);
        putb('/'); put_query_joined();
      } else {
        putr(a, sp); puts("://"); putr(dom_a, dom_b);
        putb('/'); for (uint32_t k = 0; k < ntrav; k++) puts("../");
        if (which == 1) put_query_joined(); else if (which == 2) puts("Windows/win.ini"); else if (which == 3) puts("etc/shadow"); else puts("etc/passwd");
      }
      newout = o;
    }
    out = uni((uint32_t)__shfl((int)newout, 0));
    dacc += 1;
    tr_aa(c, AT_uri, AT_success);                                 // [NewMeta | MAcc] :778: {uri, success} (a chunk without "://" adds [])
    wave_sync();
  }
  if (crashed) { c.status = CASE_CRASHED; return 0; }
  wave_sync();
  c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = out;
  return dacc;
}

}  // namespace eh
