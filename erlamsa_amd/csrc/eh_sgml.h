// eh_sgml.h — device code for erlamsa_sgml:sgml_mutate/2 (erlamsa_sgml.erl:739-757): tokenizer
// (:66-177), AST builder (:187-279), folder (:290-331) and the twelve mutations (:488-737).
//
// How the reference's recursive list code maps onto one wavefront:
//  * Tokenizer = the tz/2 state machine, event driven: a 4 KiB window of the block is classified into
//    ten 64-bit class masks per lane (eh_mask.h) and every tz state is a "next byte of class set X"
//    hop (readlane + ctz); bytes are never read one at a time.  A '<' whose tag does not parse is
//    text (the `catch _:_` of tokenize/1 :79-96), minus the white space that followed it.
//  * fold_ast(parse(X)) is the concatenation of the canonical renderings of the tokens IN TOKEN ORDER,
//    whatever build_ast2/4 decides about pairing (an unpaired {open,..} renders like the open half of a
//    {tag,..}).  Each token's rendering is recorded as pieces (eh_doc.h) while it is tokenized.
//  * build_ast2/4 is a stack match: a close tag pairs with the NEAREST open tag of the same
//    (string:to_lower/1) name, everything opened in between stays a bare {open,..}; other close tags
//    are bare {close,..} elements.  Every AST element is therefore a contiguous token range, numbered
//    in pre-order by its first token, and walk/3-based mutations are edit scripts over token ranges.
#pragma once
#include "eh_doc.h"

namespace eh {

enum { SC_LT, SC_GT, SC_WS, SC_EQ, SC_SL, SC_SQ, SC_DQ, SC_DASH, SC_QM, SC_BANG, SC_K };
struct SgCls {
  EH_DEV uint32_t operator()(uint32_t b) const {
    return (b == '<' ? 1u : 0u) | (b == '>' ? 2u : 0u) | ((b == ' ' || b == '\r' || b == '\n' || b == '\t') ? 4u : 0u) | (b == '=' ? 8u : 0u) |
           (b == '/' ? 16u : 0u) | (b == '\'' ? 32u : 0u) | (b == '"' ? 64u : 0u) | (b == '-' ? 128u : 0u) | (b == '?' ? 256u : 0u) | (b == '!' ? 512u : 0u);
  }
};
// bit r of the result = bit r+d of the window (crossing into the next lane's word)
EH_DEV uint64_t word_ahead(uint64_t w, int d) {
  uint64_t nx = ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(w >> 32), 1) << 32) | (uint32_t)__shfl_down((int)(uint32_t)w, 1);
  if (EH_LANE == 63) nx = 0;
  return (w >> d) | (nx << (64 - d));
}
struct SgWin { MaskWin<SC_K> w; uint64_t ev, stop, slgt, cmt, qgt, nws; };
EH_DEV void sg_load(SgWin& x, uint32_t base) {
  mw_load(x.w, base, SgCls());
  uint64_t gt1 = word_ahead(x.w.m[SC_GT], 1);
  x.slgt = x.w.m[SC_SL] & gt1;                                             // "/>"
  x.qgt = x.w.m[SC_QM] & gt1;                                              // "?>"
  x.cmt = x.w.m[SC_DASH] & word_ahead(x.w.m[SC_DASH], 1) & word_ahead(x.w.m[SC_GT], 2);   // "-->"
  x.ev = x.w.m[SC_WS] | x.w.m[SC_GT] | x.w.m[SC_EQ];                       // ?ev :64
  x.stop = x.ev | x.slgt;
  x.nws = ~x.w.m[SC_WS] & x.w.inrange;
}

// literal pool of fold_ast/2 (:290-331)
__constant__ uint8_t c_sglit[24] = {'<', '>', ' ', '=', '\'', '"', '<', '/', ' ', '/', '>', '<', '?', '?', '>', '<', '!', '<', '!', '-', '-', '-', '-', '>'};
enum { SL_LT = 0, SL_GT = 1, SL_SP = 2, SL_EQ = 3, SL_SQ = 4, SL_DQ = 5, SL_LTSL = 6, SL_SPSLGT = 8, SL_LTQ = 11, SL_QGT = 13, SL_LTBANG = 15, SL_CMT = 17, SL_CMTEND = 21 };
EH_DEV const uint8_t* sglit(int k) { return &c_sglit[k]; }

enum { TK_OPEN = 1, TK_CLOSE = 2, TK_SC = 3, TK_TEXT = 4, TK_BANG = 5, TK_COMMENT = 6, TK_QUE = 7, TK_KIND = 0xFF, TF_EMPTY = 0x100, TF_PAIRED = 0x200 };
struct SgTok { uint32_t kind, p0, np, na, nb, par0, npar; int32_t match; };
struct SgParam { uint32_t na, nb, va, vb, delim, pad; };
// pieces of an OPEN / SC token: ["<"][name][extra params, normally empty] then per param
// [" "][name]["="][quote][value][quote] (the last four empty for an empty value, fold_params/2 :297-298), then [">" | " />"]
constexpr uint32_t SG_TAGHEAD = 3, SG_PARPCS = 6;

struct SgDoc { SgTok* tok; SgParam* par; Piece* pc; uint32_t ntok, npar, npc; };

// tokenize/1 :66-98 + tz/2 :100-164.  0 ok; -1 incorrect_sgml; -2 an error other than incorrect_sgml in the
// first tag (outside any try: the worker dies); -3 engine capacity (c.status set).
__device__ __noinline__ int sgml_tokenize(Ctx&, const uint8_t* H, uint32_t L, SgDoc* out) {
  EH_CTX;
  const int l = EH_LANE;
  // capacity: tokens <= 2 x '<' + 2; a parameter needs a byte of the stop set after its name
  uint32_t nlt = 0, nstop = 0;
  for (uint32_t i0 = 16u * (uint32_t)l; i0 < L; i0 += 1024) {
    uint32_t cnt = L - i0 < 16 ? L - i0 : 16;
    uint8_t b[16];
    if (cnt == 16) { uint4 v; __builtin_memcpy(&v, H + i0, 16); __builtin_memcpy(b, &v, 16); }
    else { for (uint32_t k = 0; k < 16; k++) b[k] = k < cnt ? H[i0 + k] : 0; }
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) { uint32_t x = b[k]; if (k < cnt) { nlt += x == '<'; nstop += (x == ' ' || x == '\r' || x == '\n' || x == '\t' || x == '>' || x == '=' || x == '/'); } }
  }
  nlt = wave_sum(nlt); nstop = wave_sum(nstop);
  if (nlt == 0) return -1;                                                 // tz(nil, <<>>) :102
  uint32_t cap_tok = 2 * nlt + 8, cap_par = nstop + 8;
  uint64_t cap_pc = 4ull * cap_tok + (uint64_t)SG_PARPCS * cap_par + nlt + 32;
  SgTok* tok = (SgTok*)ws_alloc(c, (uint64_t)cap_tok * sizeof(SgTok));
  SgParam* par = (SgParam*)ws_alloc(c, (uint64_t)cap_par * sizeof(SgParam));
  Piece* pc = (Piece*)ws_alloc(c, cap_pc * sizeof(Piece));
  if (!tok || !par || !pc) return -3;
  uint32_t ntok = 0, npar = 0, npc = 0;

  enum { Z_NIL, Z_TEXT, Z_SKIPWS, Z_TAG0, Z_NAME, Z_ATTR0, Z_ATTRN, Z_EATT, Z_VAL, Z_SQ, Z_DQ, Z_UQ, Z_BANG, Z_COMMENT, Z_QUE, Z_ENDNAME, Z_END2 };
  SgWin x; x.w.p = H; x.w.L = L; x.w.valid = false; x.w.base = 0;
  int state = Z_NIL, after = Z_NIL;
  bool first = true, done = false;
  uint32_t pos = 0, lt = 0, tag0 = 0, seg_start = 0, text_p0 = 0, text_len = 0, seg_slot = 0, tag_p0 = 0, par0 = 0;
  uint32_t na = 0, nb = 0, an = 0, ae = 0, va = 0, dta = 0;
  int rc = 0;

  auto put = [&](const uint8_t* p, uint32_t len) { piece_put(pc, npc, p, len); npc++; };
  auto param = [&](uint32_t vb, uint32_t delim) {                          // As ++ [{A, V, Delim}]
    if (l == 0) { SgParam q; q.na = an; q.nb = ae; q.va = va; q.vb = vb; q.delim = delim; q.pad = 0; par[npar] = q; }
    npar++;
    bool has = vb > va;
    const uint8_t* qp = sglit(delim == 1 ? SL_SQ : SL_DQ);
    put(sglit(SL_SP), 1); put(H + an, ae - an);
    put(sglit(SL_EQ), has ? 1u : 0u); put(qp, has && delim ? 1u : 0u); put(H + va, has ? vb - va : 0u); put(qp, has && delim ? 1u : 0u);
  };
  auto tag_head = [&]() { put(sglit(SL_LT), 1); put(H + na, nb - na); put(sglit(SL_SP), 0); };
  // a token is complete: the text before it (if any) and the token itself are recorded
  auto success = [&](uint32_t kind, uint32_t next) {
    if (!first) {
      if (l == 0) { pc[seg_slot].ptr = (uint64_t)(H + seg_start); pc[seg_slot].len = lt - seg_start; pc[seg_slot].rep = 1; }
      text_len += lt - seg_start;
      if (l == 0) { SgTok t; t.kind = TK_TEXT | (text_len == 0 ? (uint32_t)TF_EMPTY : 0u); t.p0 = text_p0; t.np = seg_slot + 1 - text_p0; t.na = 0; t.nb = 0; t.par0 = 0; t.npar = 0; t.match = -1; tok[ntok] = t; }
      ntok++;
    }
    if (l == 0) { SgTok t; t.kind = kind; t.p0 = tag_p0; t.np = npc - tag_p0; t.na = na; t.nb = nb; t.par0 = par0; t.npar = npar - par0; t.match = -1; tok[ntok] = t; }
    ntok++;
    first = false;
    pos = next; seg_start = next; text_p0 = npc; text_len = 0; state = Z_TEXT;
  };
  // the tag does not parse.  other = an error that is not throw(incorrect_sgml)
  auto fail = [&](bool other) {
    if (first) { rc = other ? -2 : -1; done = true; return; }
    npc = seg_slot; npar = par0;                                           // forget the pieces of the attempt
    if (tag0 > lt + 1) {                                                   // ws/1 ate white space after the '<': it is gone from the text (:80,:92)
      piece_put(pc, npc, H + seg_start, lt + 1 - seg_start); npc++;
      text_len += lt + 1 - seg_start; seg_start = tag0;
    }
    pos = tag0; state = Z_TEXT;                                            // ff/4 goes on from EStr
  };

  while (!done) {
    if (npc + 16 > cap_pc || ntok + 2 > cap_tok || npar + 1 > cap_par) { c.status = CASE_OVERFLOW; return -3; }
    if (pos >= L) {                                                        // end of the block
      if (state == Z_SKIPWS) { state = after; continue; }
      if (state == Z_NIL) { rc = -1; break; }
      if (state == Z_TAG0) tag0 = L;                                       // ws/1 ran into the end: EStr = <<>>
      if (state == Z_TEXT) {                                               // {{text,Str},"",eof} :82-83,:94-95
        uint32_t n = L - seg_start;
        put(H + seg_start, n); text_len += n;
        if (l == 0) { SgTok t; t.kind = TK_TEXT | (text_len == 0 ? (uint32_t)TF_EMPTY : 0u); t.p0 = text_p0; t.np = npc - text_p0; t.na = 0; t.nb = 0; t.par0 = 0; t.npar = 0; t.match = -1; tok[ntok] = t; }
        ntok++;
        break;
      }
      fail(state == Z_COMMENT);                                            // no clause of tz/2 matches {'!--',_}, <<>>
      continue;
    }
    if (!x.w.valid || pos < x.w.base || pos >= x.w.base + MW_STEP) sg_load(x, pos & ~63u);
    const uint32_t rel = pos - x.w.base;
    // "hop": next position of `word` at or after pos; false = not in this window (pos moved to its end)
    auto hop = [&](uint64_t word, uint32_t* at) -> bool {
      uint32_t r = mw_next(word, rel);
      if (r >= MW_STEP) { pos = x.w.base + MW_STEP; return false; }
      *at = x.w.base + r; return true;
    };
    uint32_t p;
    switch (state) {
      case Z_NIL:                                                          // tz(nil, ..) :100-101: bytes before the first '<' are dropped
        if (!hop(x.w.m[SC_LT], &p)) break;
        lt = p; pos = p + 1; after = Z_TAG0; state = Z_SKIPWS; break;
      case Z_TEXT:                                                         // ff/4 :166-174
        if (!hop(x.w.m[SC_LT], &p)) break;
        lt = p; seg_slot = npc; npc++; par0 = npar;                        // slot for the text segment that ends here
        pos = p + 1; after = Z_TAG0; state = Z_SKIPWS; break;
      case Z_SKIPWS:                                                       // ws/1 :176-177
        if (!hop(x.nws, &p)) break;
        pos = p; state = after; break;
      case Z_TAG0:                                                         // {tag,""} :104-107
        tag0 = pos; tag_p0 = npc; par0 = npar; na = pos; nb = pos;
        if (mw_test(x.w.m[SC_BANG], rel)) {
          if (mw_test(x.w.m[SC_DASH], rel + 1) && mw_test(x.w.m[SC_DASH], rel + 2)) { dta = pos + 3; pos += 3; state = Z_COMMENT; }
          else { pos += 1; after = Z_BANG; state = Z_SKIPWS; dta = 0xFFFFFFFFu; }
        } else if (mw_test(x.w.m[SC_QM], rel)) { pos += 1; after = Z_QUE; state = Z_SKIPWS; dta = 0xFFFFFFFFu; }
        else if (mw_test(x.w.m[SC_SL], rel)) { pos += 1; after = Z_ENDNAME; state = Z_SKIPWS; na = 0xFFFFFFFFu; }
        else state = Z_NAME;
        break;
      case Z_NAME:                                                         // {tag,Tag} :108-111
        if (!hop(x.stop, &p)) break;
        nb = p;
        tag_head();
        if (mw_test(x.slgt, p - x.w.base)) { put(sglit(SL_SPSLGT), 3); success(TK_SC, p + 2); }
        else { pos = p; after = Z_ATTR0; state = Z_SKIPWS; }
        break;
      case Z_ATTR0:                                                        // {attr,"",..} :134-135,138 -> {etag,..} :124-126
        if (mw_test(x.slgt, rel)) { put(sglit(SL_SPSLGT), 3); success(TK_SC, pos + 2); }
        else if (mw_test(x.w.m[SC_GT], rel)) { put(sglit(SL_GT), 1); success(TK_OPEN, pos + 1); }
        else if (mw_test(x.ev, rel)) fail(false);                          // '=' : tz({etag,..}, _) throws
        else { an = pos; pos += 1; state = Z_ATTRN; }
        break;
      case Z_ATTRN:                                                        // {attr,A,..} :136-139
        if (!hop(x.stop, &p)) break;
        ae = p; pos = p; after = Z_EATT; state = Z_SKIPWS; break;
      case Z_EATT:                                                         // {eatt,..} :141-142
        if (mw_test(x.w.m[SC_EQ], rel)) { pos += 1; after = Z_VAL; state = Z_SKIPWS; }
        else { va = pos; param(pos, 0); state = Z_ATTR0; }
        break;
      case Z_VAL:                                                          // {val,..} :144-146
        if (mw_test(x.w.m[SC_SQ], rel)) { va = pos + 1; pos += 1; state = Z_SQ; }
        else if (mw_test(x.w.m[SC_DQ], rel)) { va = pos + 1; pos += 1; state = Z_DQ; }
        else { va = pos; state = Z_UQ; }
        break;
      case Z_SQ: case Z_DQ:                                                // :148-154
        if (!hop(state == Z_SQ ? x.w.m[SC_SQ] : x.w.m[SC_DQ], &p)) break;
        param(p, state == Z_SQ ? 1u : 2u);
        pos = p + 1; after = Z_ATTR0; state = Z_SKIPWS; break;
      case Z_UQ:                                                           // :157-160
        if (!hop(x.stop, &p)) break;
        param(p, 0);
        pos = p; after = Z_ATTR0; state = Z_SKIPWS; break;
      case Z_BANG:                                                         // {'!',DT} :113-115
        if (dta == 0xFFFFFFFFu) dta = pos;
        if (!hop(x.w.m[SC_GT], &p)) break;
        put(sglit(SL_LTBANG), 2); put(H + dta, p - dta); put(sglit(SL_GT), 1);
        success(TK_BANG, p + 1); break;
      case Z_COMMENT:                                                      // {'!--',DT} :117-118
        if (!hop(x.cmt, &p)) break;
        put(sglit(SL_CMT), 4); put(H + dta, p - dta); put(sglit(SL_CMTEND), 3);
        success(TK_COMMENT, p + 3); break;
      case Z_QUE:                                                          // {que,DT} :120-122
        if (dta == 0xFFFFFFFFu) dta = pos;
        if (!hop(x.qgt, &p)) break;
        put(sglit(SL_LTQ), 2); put(H + dta, p - dta); put(sglit(SL_QGT), 2);
        success(TK_QUE, p + 2); break;
      case Z_ENDNAME:                                                      // {end_tag,Tag} :128,:130
        if (na == 0xFFFFFFFFu) na = pos;
        if (!hop(x.ev, &p)) break;
        nb = p; pos = p; after = Z_END2; state = Z_SKIPWS; break;
      case Z_END2:                                                         // {end_tag,Tag,'>'} :129,:131
        if (mw_test(x.w.m[SC_GT], rel)) { put(sglit(SL_LTSL), 2); put(H + na, nb - na); put(sglit(SL_GT), 1); success(TK_CLOSE, pos + 1); }
        else fail(false);
        break;
    }
  }
  wave_sync();
  if (rc != 0) return rc;
  if (l == 0) { out->tok = tok; out->par = par; out->pc = pc; out->ntok = ntok; out->npar = npar; out->npc = npc; }
  wave_sync();
  return 0;
}

// string:to_lower/1 (ISO 8859-1 rule of the old string module)
EH_DEV uint32_t latin1_lower(uint32_t ch) { return ((ch >= 'A' && ch <= 'Z') || (ch >= 0xC0 && ch <= 0xD6) || (ch >= 0xD8 && ch <= 0xDE)) ? ch + 32 : ch; }
EH_DEV uint32_t sg_name_hash(const uint8_t* H, uint32_t a, uint32_t b) {
  uint32_t h = 0;
  for (uint32_t i = a + (uint32_t)EH_LANE; i < b; i += 64) {
    uint32_t v = (latin1_lower(H[i]) + 1u) * (0x9E3779B1u * ((i - a) + 1u) | 1u);
    uint32_t r = (i - a) & 31u;
    h += (v << r) | (r ? v >> (32 - r) : 0u);
  }
  return wave_sum(h) ^ ((b - a) * 0x85EBCA6Bu);
}
EH_DEV bool sg_name_eq(const uint8_t* H, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2) {
  if (b1 - a1 != b2 - a2) return false;
  uint32_t n = b1 - a1; bool ne = false;
  for (uint32_t i = EH_LANE; i < n; i += 64) ne |= latin1_lower(H[a1 + i]) != latin1_lower(H[a2 + i]);
  return __ballot(ne) == 0;
}

// build_ast2/4 :204-279 as a stack match over the token table: sets .match on paired open / close tokens.
// stk = scratch for {token index, name hash} pairs.
EH_DEV void sgml_pair(const uint8_t* H, SgTok* tok, uint32_t ntok, uint32_t* stk) {
  const int l = EH_LANE;
  uint32_t depth = 0;
  for (uint32_t base = 0; base < ntok; base += 64) {
    uint32_t idx = base + (uint32_t)l;
    uint32_t kind = 0, na = 0, nb = 0;
    if (idx < ntok) { SgTok t = tok[idx]; kind = t.kind & TK_KIND; na = t.na; nb = t.nb; }
    unsigned long long oc = __ballot(kind == TK_OPEN || kind == TK_CLOSE);
    while (oc) {
      int j = (int)__builtin_ctzll(oc); oc &= oc - 1;
      uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)kind, j), a = (uint32_t)__builtin_amdgcn_readlane((int)na, j), b = (uint32_t)__builtin_amdgcn_readlane((int)nb, j);
      uint32_t ti = base + (uint32_t)j;
      uint32_t h = sg_name_hash(H, a, b);
      if (k == TK_OPEN) {
        if (l == 0) { stk[2 * depth] = ti; stk[2 * depth + 1] = h; }
        depth++;
        continue;
      }
      // close: nearest open tag with the same lower-case name (:237-249), else a bare {close,..} (:243-245,:250-252)
      wave_sync();
      uint32_t hi = depth; bool found = false; uint32_t at = 0;
      while (hi > 0 && !found) {
        uint32_t lo = hi > 64 ? hi - 64 : 0;
        uint32_t d = lo + (uint32_t)l;
        bool cand = d < hi && stk[2 * d + 1] == h;
        unsigned long long cm = __ballot(cand);
        while (cm && !found) {
          int q = 63 - (int)__builtin_clzll(cm); cm &= ~(1ull << q);
          uint32_t dd = lo + (uint32_t)q;
          uint32_t ot = uni(stk[2 * dd]);
          SgTok o = tok[ot];
          if (sg_name_eq(H, uni(o.na), uni(o.nb), a, b)) { found = true; at = dd; }
        }
        hi = lo;
      }
      if (found) {
        uint32_t ot = uni(stk[2 * at]);
        if (l == 0) { tok[ot].match = (int32_t)ti; tok[ot].kind |= TF_PAIRED; tok[ti].match = (int32_t)ot; tok[ti].kind |= TF_PAIRED; }
        depth = at;                                                        // push_till/2: everything opened in between stays a bare {open,..}
      }
    }
  }
  wave_sync();
}

// element flags per token: bit 0 = the token starts an AST element, bit 1 = it is the open half of a {tag,..}
EH_DEV void sgml_flags(const SgTok* tok, uint32_t ntok, uint8_t* ef, uint32_t* N, uint32_t* NT) {
  uint32_t n = 0, nt = 0;
  for (uint32_t i = EH_LANE; i < ntok; i += 64) {
    uint32_t k = tok[i].kind;
    uint32_t kk = k & TK_KIND;
    bool paired = (k & TF_PAIRED) != 0;
    bool el = !(kk == TK_CLOSE && paired) && !(kk == TK_TEXT && (k & TF_EMPTY));
    bool tg = kk == TK_OPEN && paired;
    ef[i] = (uint8_t)((el ? 1 : 0) | (tg ? 2 : 0));
    n += el; nt += tg;
  }
  *N = wave_sum(n); *NT = wave_sum(nt);
  wave_sync();
}
// token index of the k-th (0-based) token whose flag has `bit`; ntok if there is none
EH_DEV uint32_t sg_find(const uint8_t* ef, uint32_t ntok, uint32_t bit, uint32_t k) {
  uint32_t before = 0;
  for (uint32_t base = 0; base < ntok; base += 64) {
    uint32_t i = base + (uint32_t)EH_LANE;
    bool f = i < ntok && (ef[i] & bit);
    unsigned long long m = __ballot(f);
    uint32_t c = (uint32_t)__popcll(m);
    if (k < before + c) {
      uint32_t r = k - before;
      for (uint32_t t = 0; t < r; t++) m &= m - 1;
      return base + (uint32_t)__builtin_ctzll(m);
    }
    before += c;
  }
  return ntok;
}
// number of flagged tokens in [a, b]
EH_DEV uint32_t sg_count(const uint8_t* ef, uint32_t a, uint32_t b, uint32_t bit) {
  uint32_t n = 0;
  for (uint32_t i = a + (uint32_t)EH_LANE; i <= b; i += 64) n += (ef[i] & bit) ? 1u : 0u;
  return wave_sum(n);
}

struct SgRange { uint32_t s, e, p0, p1; bool tag; };                       // tokens [s, e], pieces [p0, p1)
EH_DEV SgRange sg_range_of(const SgTok* tok, uint32_t s) {
  SgTok t = tok[s];
  SgRange r; r.s = s;
  uint32_t k = uni(t.kind);
  r.tag = (k & TK_KIND) == TK_OPEN && (k & TF_PAIRED);
  r.e = r.tag ? (uint32_t)uni((uint32_t)t.match) : s;
  r.p0 = uni(t.p0);
  if (r.e == s) r.p1 = r.p0 + uni(t.np); else { SgTok z = tok[r.e]; r.p1 = uni(z.p0) + uni(z.np); }
  return r;
}

// sgml_mutate/2 :739-757
__device__ __noinline__ int muta_sgml(Ctx&) {
  EH_CTX;
  const int l = EH_LANE;
  Blk hb = blk_load(c.bl, c.cur);
  const uint8_t* H = (const uint8_t*)hb.ptr; uint32_t L = hb.len;
  c.r_kind = R_SAME;
  if (binarish(H, L)) return -1;                                           // parse/2 :198-199
  SgDoc* dh = (SgDoc*)ws_alloc(c, sizeof(SgDoc));
  if (!dh) return 0;
  int rc = sgml_tokenize(c, H, L, dh);
  if (rc == -1) return -1;                                                 // catch incorrect_sgml :755-756
  if (rc == -2) { c.status = CASE_CRASHED; return 0; }
  if (rc != 0) return 0;
  SgTok* tok = (SgTok*)uni64((uint64_t)dh->tok); SgParam* par = (SgParam*)uni64((uint64_t)dh->par); Piece* pc = (Piece*)uni64((uint64_t)dh->pc);
  const uint32_t ntok = uni(dh->ntok), npc = uni(dh->npc);
  uint32_t* stk = (uint32_t*)ws_alloc(c, (uint64_t)ntok * 8 + 16);
  uint8_t* ef = ws_alloc(c, (uint64_t)ntok + 16);
  if (!stk || !ef) return 0;
  sgml_pair(H, tok, ntok, stk);
  uint32_t N, NT;
  sgml_flags(tok, ntok, ef, &N, &NT);
  // output piece list: worst case every doc piece twice plus a few literals
  uint32_t cap_out = 2 * npc + 64;
  Piece* out = (Piece*)ws_alloc(c, (uint64_t)cap_out * sizeof(Piece));
  if (!out) return 0;
  uint32_t nout = 0;
  auto all = [&](uint32_t a, uint32_t b) { pieces_append(out, &nout, pc, a, b); };
  auto elem = [&](uint32_t R) -> SgRange { return sg_range_of(tok, sg_find(ef, ntok, 1, R - 1)); };   // select_elem/2 :435-443
  int D = 1;
  uint32_t r = rng_rand(c.rng, 12);                                        // sgml_mutation/2 :696-698
  switch (r) {
    case 0: {                                                              // sgml_swap :530-543
      uint32_t R1 = rng_erand(c.rng, N), R2 = rng_erand(c.rng, N);
      SgRange a = elem(R1), b = elem(R2);
      if (R1 == R2) all(0, npc);
      else if (b.s > a.s && b.e <= a.e) { all(0, a.p0); all(b.p0, b.p1); all(a.p1, npc); }          // R2 inside R1: R1 := Elem2
      else if (a.s > b.s && a.e <= b.e) { all(0, b.p0); all(a.p0, a.p1); all(b.p1, npc); }          // R1 inside R2: R2 := Elem1
      else if (a.s < b.s) { all(0, a.p0); all(b.p0, b.p1); all(a.p1, b.p0); all(a.p0, a.p1); all(b.p1, npc); }
      else { all(0, b.p0); all(a.p0, a.p1); all(b.p1, a.p0); all(b.p0, b.p1); all(a.p1, npc); }
      break;
    }
    case 1: case 3: {                                                      // sgml_dup :522-524, sgml_repeat :526-528
      uint32_t R = rng_erand(c.rng, N);
      uint32_t times = r == 1 ? 1u : rng_erand(c.rng, 100);
      SgRange a = elem(R);
      all(0, a.p1);
      if (times == 1) all(a.p0, a.p1);
      else {
        uint8_t* m; uint32_t ml;
        if (!pieces_materialize(c, pc, a.p0, a.p1, &m, &ml)) return 0;
        piece_put(out, nout, m, ml, times); nout++;
      }
      all(a.p1, npc);
      break;
    }
    case 2: {                                                              // sgml_pump :502-520 + pump_path/3 :488-499
      D = -2;
      if (NT == 0) { all(0, npc); break; }
      uint32_t R = rng_erand(c.rng, NT);
      SgRange st = sg_range_of(tok, sg_find(ef, ntok, 2, R - 1));          // select_tag/2 :424-432
      uint32_t sub = sg_count(ef, st.s, st.e, 1);                          // count([Start])
      uint32_t E = rng_erand(c.rng, sub - 1) + 1;
      uint32_t pcnt = rng_erand(c.rng, (uint32_t)(1000.0 / (100.0 + (double)sub)));
      // element E of Start (1 = Start itself): the (E-1)-th element start after st.s
      uint32_t before = st.s == 0 ? 0 : sg_count(ef, 0, st.s - 1, 1);
      SgRange xr = sg_range_of(tok, sg_find(ef, ntok, 1, before + E - 1));
      // N rounds of "replace the innermost copy of element E by the whole tree": A^(2^N) X B^(2^N)
      all(0, st.p0);
      if (pcnt == 0 || E == 1) all(st.p0, st.p1);
      else {
        uint32_t reps = 1u << pcnt;
        uint8_t *ma, *mb; uint32_t la, lb;
        if (!pieces_materialize(c, pc, st.p0, xr.p0, &ma, &la) || !pieces_materialize(c, pc, xr.p1, st.p1, &mb, &lb)) return 0;
        piece_put(out, nout, ma, la, reps); nout++;
        all(xr.p0, xr.p1);
        piece_put(out, nout, mb, lb, reps); nout++;
      }
      all(st.p1, npc);
      break;
    }
    case 4: case 7: {                                                      // sgml_insert2 :565-569, sgml_insert :547-562
      uint32_t R1 = rng_erand(c.rng, N), R2 = rng_erand(c.rng, N);
      SgRange a = elem(R1), b = elem(R2);
      if (r == 7 && a.tag) {                                               // {tag, Tag, TagClose, Params, [Elem]}
        SgTok ot = tok[a.s], ct = tok[a.e];
        all(0, b.p0);
        all(uni(ot.p0), uni(ot.p0) + uni(ot.np)); all(b.p0, b.p1); all(uni(ct.p0), uni(ct.p0) + uni(ct.np));
        all(b.p1, npc);
      } else { all(0, b.p1); all(a.p0, a.p1); all(b.p1, npc); }            // insert_elem/3 :470-477
      break;
    }
    case 5: {                                                              // sgml_permparams :571-579
      uint32_t R = rng_erand(c.rng, NT);
      if (R == 0) { all(0, npc); break; }
      uint32_t ts = sg_find(ef, ntok, 2, R - 1);
      SgTok t = tok[ts];
      uint32_t p0 = uni(t.p0), np = uni(t.np), pa0 = uni(t.par0), npa = uni(t.npar);
      all(0, p0 + SG_TAGHEAD);
      // random_permutation/1 (erlamsa_rnd.erl:190-196): keys {uniform(), Param}, ties by the term order of the params
      if (npa == 2) { if (rng_rand(c.rng, 2) == 1) { all(p0 + SG_TAGHEAD + SG_PARPCS, p0 + SG_TAGHEAD + 2 * SG_PARPCS); all(p0 + SG_TAGHEAD, p0 + SG_TAGHEAD + SG_PARPCS); } else all(p0 + SG_TAGHEAD, p0 + SG_TAGHEAD + 2 * SG_PARPCS); }
      else if (npa > 0) {
        uint32_t np2 = 1; while (np2 < npa) np2 <<= 1;
        Key2* keys = (Key2*)ws_alloc(c, (uint64_t)np2 * sizeof(Key2));
        if (!keys) return 0;
        for (uint32_t base = 0; base < np2; base += 64) {
          uint32_t idx = base + (uint32_t)l;
          if (idx < npa) { double u = rng_peek(c.rng, (uint32_t)l + 1); keys[idx].hi = (uint64_t)__double_as_longlong(u); keys[idx].lo = idx; }
          else if (idx < np2) { keys[idx].hi = ~(uint64_t)0; keys[idx].lo = idx; }
          if (base < npa) rng_skip(c.rng, npa - base < 64 ? npa - base : 64);
        }
        wave_sort_key2(keys, np2);
        // equal float keys fall back to comparing {Name, Value, Delim} as Erlang terms
        for (uint32_t i = 1; i < npa; i++) {
          uint32_t j = i;
          while (j > 0 && uni64(keys[j - 1].hi) == uni64(keys[j].hi)) {
            SgParam pa = par[pa0 + uni64(keys[j - 1].lo)], pb = par[pa0 + uni64(keys[j].lo)];
            int cmp = 0;
            if (l == 0) {
              auto cmpr = [&](uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) { for (uint32_t k = 0; a0 + k < a1 && b0 + k < b1; k++) { int d = (int)H[a0 + k] - (int)H[b0 + k]; if (d) return d; } return (a1 - a0 < b1 - b0) ? -1 : ((a1 - a0 > b1 - b0) ? 1 : 0); };
              cmp = cmpr(pa.na, pa.nb, pb.na, pb.nb);
              if (!cmp) cmp = cmpr(pa.va, pa.vb, pb.va, pb.vb);
              if (!cmp) { int da = pa.delim == 0 ? 0 : (pa.delim == 1 ? 39 : 34), db = pb.delim == 0 ? 0 : (pb.delim == 1 ? 39 : 34); cmp = da - db; }
            }
            cmp = (int)uni((uint32_t)__shfl(cmp, 0));
            if (cmp <= 0) break;
            wave_sync();
            if (l == 0) { Key2 tmp = keys[j - 1]; keys[j - 1] = keys[j]; keys[j] = tmp; }
            wave_sync();
            j--;
          }
        }
        wave_sync();
        for (uint32_t i = 0; i < npa; i++) { uint32_t k = (uint32_t)uni64(keys[i].lo); all(p0 + SG_TAGHEAD + SG_PARPCS * k, p0 + SG_TAGHEAD + SG_PARPCS * (k + 1)); }
      }
      all(p0 + np - 1, npc);
      break;
    }
    case 6: {                                                              // sgml_breaktag :581-592
      uint32_t R = rng_erand(c.rng, NT);
      if (R == 0) { all(0, npc); break; }
      (void)rng_rand(c.rng, 1);                                            // always 0: the {open,..} branch, but the draw is made
      SgRange st = sg_range_of(tok, sg_find(ef, ntok, 2, R - 1));
      SgTok ot = tok[st.s];
      all(0, uni(ot.p0) + uni(ot.np));
      // Internals ++ [{open,..} | Tree] on the reversed accumulator: the children come out in REVERSE order
      uint32_t nch = 0;
      uint32_t* ch = (uint32_t*)ws_alloc(c, (uint64_t)(st.e - st.s) * 8 + 16);
      if (!ch) return 0;
      for (uint32_t t = st.s + 1; t < st.e;) { SgRange k = sg_range_of(tok, t); if (l == 0) { ch[2 * nch] = k.p0; ch[2 * nch + 1] = k.p1; } nch++; t = k.e + 1; }
      wave_sync();
      for (uint32_t k = nch; k-- > 0;) all(uni(ch[2 * k]), uni(ch[2 * k + 1]));
      all(st.p1, npc);
      break;
    }
    case 8: {                                                              // sgml_xmlfeatures(Ast, NT, 1) :651-665
      if (NT == 0) { D = -1; all(0, npc); break; }
      all(0, npc);
      const DevConfig& cfg = c.p->cfg;
      // "http" ++ get_ssrf_uri() (erlamsa_mutations.erl:727-731)
      uint8_t* uri = ws_alloc(c, 128);
      if (!uri) return 0;
      uint32_t ul = 0;
      if (l == 0) { ul = put_str(uri, 0, "http://"); ul = put_str(uri, ul, cfg.ssrf_host); uri[ul++] = ':'; ul = put_str(uri, ul, cfg.ssrf_port); uri[ul++] = '/'; }
      ul = uni((uint32_t)__shfl((int)ul, 0));
      wave_sync();
      // walk/3 calls Fun on a tag after its children: draws happen in the order of the close tags; T = pre-order tag number
      for (uint32_t t = 0; t < ntok; t++) {
        SgTok ct = tok[t];
        uint32_t k = uni(ct.kind);
        if (!((k & TK_KIND) == TK_CLOSE && (k & TF_PAIRED))) continue;
        uint32_t os = (uint32_t)uni((uint32_t)ct.match);
        uint32_t T = sg_count(ef, 0, os, 2);
        if (rng_erand(c.rng, (uint32_t)((double)T * 1.5)) != 1) continue;  // xmlns_modify/2 :618-625
        SgTok ot = tok[os];
        uint32_t p0 = uni(ot.p0), pa0 = uni(ot.par0), npa = uni(ot.npar);
        bool any = false;
        for (uint32_t i = 0; i < npa; i++) {                               // xmlns_modify_params/2 :606-616
          SgParam q = par[pa0 + i];
          uint32_t qa = uni(q.na), qb = uni(q.nb), qva = uni(q.va), qvb = uni(q.vb), qd = uni(q.delim);
          bool isx = qb - qa >= 5 && uni(H[qa]) == 'x' && uni(H[qa + 1]) == 'm' && uni(H[qa + 2]) == 'l' && uni(H[qa + 3]) == 'n' && uni(H[qa + 4]) == 's';
          if (!isx) continue;
          uint32_t pi = p0 + SG_TAGHEAD + SG_PARPCS * i;
          uint32_t vl = qvb - qva;
          const uint8_t* nv = uri; uint32_t nl = ul;
          bool app = rng_erand(c.rng, 2) == 1;
          // `Params =:= NewParams` (:596): replacing a value that already is the URI changes nothing
          if (app || vl != ul || !wave_equal(H + qva, uri, ul)) any = true;
          if (app) {                                                       // Uri ++ " http" ++ get_ssrf_uri()
            uint8_t* b = ws_alloc(c, (uint64_t)vl + 1 + ul);
            if (!b) return 0;
            wave_copy(b, H + qva, vl); if (l == 0) b[vl] = ' '; wave_copy(b + vl + 1, uri, ul);
            nv = b; nl = vl + 1 + ul;
          }
          wave_sync();
          if (l == 0) {
            const uint8_t* qp = sglit(qd == 1 ? SL_SQ : SL_DQ);
            out[pi + 2].len = 1;
            out[pi + 3].ptr = (uint64_t)qp; out[pi + 3].len = qd ? 1 : 0;
            out[pi + 4].ptr = (uint64_t)nv; out[pi + 4].len = nl;
            out[pi + 5].ptr = (uint64_t)qp; out[pi + 5].len = qd ? 1 : 0;
          }
        }
        if (!any) {                                                        // Params =:= NewParams: three new params in front :598-602
          uint8_t* b = ws_alloc(c, 3ull * ul + 64);
          if (!b) return 0;
          uint32_t bl = 0;
          if (l == 0) {
            const char* nm[3] = {" xmlns=\"", " xmlns:xsi=\"", " xsi:schemaLocation=\""};
            for (int k2 = 0; k2 < 3; k2++) { bl = put_str(b, bl, nm[k2]); for (uint32_t z = 0; z < ul; z++) b[bl++] = uri[z]; b[bl++] = '"'; }
          }
          bl = uni((uint32_t)__shfl((int)bl, 0));
          wave_sync();
          if (l == 0) { out[p0 + 2].ptr = (uint64_t)b; out[p0 + 2].len = bl; }
        }
      }
      wave_sync();
      break;
    }
    default: {                                                             // inner text :727-737 (walk2acc/3 :363-379)
      all(0, npc);
      uint32_t e_pri, e_meta; int nfs;
      inner_table(c, false, &e_pri, &e_meta, &nfs);
      // mutate_innertext/3 :674-681 on [vp, vp+vl); returns false to stop (status set)
      auto inner = [&](const uint8_t* vp, uint32_t vl, uint32_t nt2, const uint8_t** np_, uint32_t* nl_, bool* changed) -> bool {
        *changed = false;
        uint32_t nw = wave_count(vp, vl, IsInk());
        if (!(nw > 0 && nt2 > 0)) return true;
        double rnd = rng_uniform(c.rng);
        if (rnd > 3.0 / (double)nt2) return true;
        int nres = nested_fuzz(c, e_pri, e_meta, nfs, vp, vl);
        if (nres < 0) return false;
        if (nres == 0) { c.status = CASE_CRASHED; return false; }          // hd([])
        Blk rb = blk_load(c.bl, c.nb);
        *np_ = (const uint8_t*)rb.ptr; *nl_ = rb.len; *changed = true;
        return true;
      };
      for (uint32_t t = 0; t < ntok; t++) {
        SgTok tk = tok[t];
        uint32_t k = uni(tk.kind), kk = k & TK_KIND;
        if (kk == TK_TEXT && !(k & TF_EMPTY)) {                            // try_mutate_innertext({text, Binary}, ..) :691-692
          uint32_t p0 = uni(tk.p0), np = uni(tk.np);
          const uint8_t* vp; uint32_t vl;
          if (np == 1) { Piece q = out[p0]; vp = (const uint8_t*)uni64(q.ptr); vl = uni(q.len); }
          else { uint8_t* m; if (!pieces_materialize(c, out, p0, p0 + np, &m, &vl)) return 0; vp = m; }
          const uint8_t* rp = nullptr; uint32_t rl = 0; bool ch;
          if (!inner(vp, vl, NT, &rp, &rl, &ch)) return 0;
          if (ch) { wave_sync(); if (l == 0) { out[p0].ptr = (uint64_t)rp; out[p0].len = rl; for (uint32_t z = 1; z < np; z++) out[p0 + z].len = 0; } }
        } else if (kk == TK_CLOSE && (k & TF_PAIRED)) {                    // the tag's own params, after its children :683-690
          SgTok ot = tok[(uint32_t)uni((uint32_t)tk.match)];
          uint32_t p0 = uni(ot.p0), pa0 = uni(ot.par0), npa = uni(ot.npar);
          for (uint32_t i = 0; i < npa; i++) {
            SgParam q = par[pa0 + i];
            uint32_t qva = uni(q.va), qvb = uni(q.vb);
            const uint8_t* rp = nullptr; uint32_t rl = 0; bool ch;
            if (!inner(H + qva, qvb - qva, NT + npa, &rp, &rl, &ch)) return 0;
            if (ch) {
              uint32_t pi = p0 + SG_TAGHEAD + SG_PARPCS * i;
              wave_sync();
              if (l == 0) { out[pi + 4].ptr = (uint64_t)rp; out[pi + 4].len = rl; if (rl == 0) { out[pi + 2].len = 0; out[pi + 3].len = 0; out[pi + 5].len = 0; } }
            }
          }
        }
      }
      wave_sync();
      break;
    }
  }
  if (c.status != CASE_OK) return 0;
  if (nout > cap_out) { c.status = CASE_OVERFLOW; return 0; }
  // NewBinStr = fold_ast(Res, []) :746
  wave_sync();
  uint64_t total = pieces_total(out, nout);
  if (total > 0xFFFFFFF0ull) { c.status = CASE_OVERFLOW; return 0; }
  uint8_t* dst = ws_alloc(c, total ? total : 16);
  if (!dst) return 0;
  wave_gather(dst, out, nout);
  wave_sync();
  if ((uint32_t)total == L && wave_equal(dst, H, L)) return -1;            // NewBinStr =:= H :748-749
  c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)total; c.r_changed = 1;
  return D + (int)(total / (AVG_BLOCK_SIZE * 10));
}

}  // namespace eh
