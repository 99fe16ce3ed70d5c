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

// byte classes of the tokenizer; 0 = ordinary byte (no event).  The three two/three byte terminators are resolved
// when the events are extracted: '/' of "/>" is E_SLGT, '?' of "?>" is E_QGT, the first '-' of "-->" is E_CMTEND.
enum { E_LT = 1, E_GT, E_WS, E_EQ, E_SL, E_SQ, E_DQ, E_DASH, E_QM, E_BANG, E_SLGT, E_QGT, E_CMTEND, E_SPACE };   // E_SPACE: the blank itself; E_WS: \t \n \r
constexpr uint32_t ES_WS = (1u << E_WS) | (1u << E_SPACE);                                // ?ws :58
constexpr uint32_t ES_EV = ES_WS | (1u << E_GT) | (1u << E_EQ);                           // ?ev :64
constexpr uint32_t ES_STOP = ES_EV | (1u << E_SLGT);
constexpr uint32_t ES_DASH = (1u << E_DASH) | (1u << E_CMTEND);
// class of a byte below 64, 4 bits each (every special byte is below 64)
EH_DEV uint32_t sg_class(uint32_t b) {
  // 9 \t, 10 \n, 13 \r, 32 ' ' -> WS; 33 '!'; 34 '"'; 39 '\''; 45 '-'; 47 '/'; 60 '<'; 61 '='; 62 '>'; 63 '?'
  const uint64_t t0 = ((uint64_t)E_WS << 36) | ((uint64_t)E_WS << 40) | ((uint64_t)E_WS << 52);                  // bytes 0..15
  const uint64_t t2 = ((uint64_t)E_SPACE << 0) | ((uint64_t)E_BANG << 4) | ((uint64_t)E_DQ << 8) | ((uint64_t)E_SQ << 28) | ((uint64_t)E_DASH << 52) | ((uint64_t)E_SL << 60);   // 32..47
  const uint64_t t3 = ((uint64_t)E_LT << 48) | ((uint64_t)E_EQ << 52) | ((uint64_t)E_GT << 56) | ((uint64_t)E_QM << 60);   // 48..63
  uint64_t t = b < 16 ? t0 : (b < 32 ? 0ull : (b < 48 ? t2 : t3));
  return b < 64 ? (uint32_t)((t >> (4 * (b & 15))) & 15) : 0u;
}

// literal pool of fold_ast/2 (:290-331)
__constant__ uint8_t c_sglit[24] = {'<', '>', ' ', '=', '\'', '"', '<', '/', ' ', '/', '>', '<', '?', '?', '>', '<', '!', '<', '!', '-', '-', '-', '-', '>'};
enum { SL_LT = 0, SL_GT = 1, SL_SP = 2, SL_EQ = 3, SL_SQ = 4, SL_DQ = 5, SL_LTSL = 6, SL_SPSLGT = 8, SL_LTQ = 11, SL_QGT = 13, SL_LTBANG = 15, SL_CMT = 17, SL_CMTEND = 21 };
EH_DEV cbptr sglit(int k) { return (cbptr)&c_sglit[k]; }   // (constant data is global memory)

enum { TK_OPEN = 1, TK_CLOSE = 2, TK_SC = 3, TK_TEXT = 4, TK_BANG = 5, TK_COMMENT = 6, TK_QUE = 7, TK_KIND = 0xFF, TF_EMPTY = 0x100, TF_PAIRED = 0x200 };
struct SgTok { uint32_t kind, p0, np, na, nb, par0, npar; int32_t match; };
struct SgParam { uint32_t na, nb, va, vb, delim, pad; };
// pieces of an OPEN / SC token: ["<"][name][extra params, normally empty] then per param
// [" "][name]["="][quote][value][quote] (the last four empty for an empty value, fold_params/2 :297-298), then [">" | " />"]
constexpr uint32_t SG_TAGHEAD = 3, SG_PARPCS = 6;

struct SgDoc { EH_G SgTok* tok; EH_G SgParam* par; EH_G Piece* pc; uint32_t ntok, npar, npc; };

// ---- one tag attempt PER LANE (round 4) ------------------------------------------------------------------------------------------
// An attempt of tz/2 from a '<' depends on nothing before that '<' (the machine is in its text state there), so the attempts at
// the next 64 '<' of the block can be made at once, one per lane, as plain per-lane loops over the events (LDS window, the event
// list in the work area outside it); which of them the machine really makes is decided afterwards (an accepted tag swallows the
// '<' inside it).  This is sgml_tokenize's attempt (below) statement for statement.  It pays while the events are in LDS: outside
// the window every event is a dependent global load, and a tag of thousands of attributes walked that way by one lane (tried: a
// batch of one for such tags) took twice the time of the wave-wide machine - so a lane gives up (bail) once it has left the
// window for more than a few events or looked at more than its budget, and the wave-wide machine, which also searches 64 events
// per ballot, takes that '<'.
// MODE 0 sizes the tag, MODE 1 writes its pieces and parameters, MODE 2 marks the attribute-loop entries of a failed attempt.
// The memos of the wave-wide machine are read here and fed from here: ff (a search for one class that found nothing from event f
// finds nothing from a later event) and bad (the attribute loop entered at byte p fails) - both facts about the block, whichever
// attempt establishes them.
struct SgLaneTag { uint32_t ok, bail, kind, next, nexte, tag0, lt, npc, npar, na, nb, reach, ffc, fff, unmarked, capped; };
struct SgLaneMemo { uint32_t gt, qgt, cmt, sq, dq; cbptr bad; cwptr nstop; };
constexpr uint32_t SG_LANE_BUDGET = 8192;           // events a lane may look at in all,
constexpr uint32_t SG_LANE_FAR = 64;                // ... of them outside the LDS window (the event list in the work area: a dependent global load each),
constexpr uint32_t SG_LANE_ATTRS = 48, SG_LANE_ATTRS_MIN = 6;   // ... and attributes it walks (the cap adapts between these, lane_batches): a longer tag is the wave-wide machine's, which takes its attributes one per lane (round 5)
constexpr uint32_t SG_LANE_SEARCH = 2048;           // ... and in one search for a single class ('>', a quote, "-->", "?>"): the wave-wide machine looks at 64 per step
// (a function of its own: inlined three times into sgml_tokenize it cost the wave-wide machine there its registers)
template <int MODE>
__device__ __noinline__ SgLaneTag sg_lane_attempt(cbptr H, uint32_t L, cwptr ev, uint32_t nev, uint32_t cbase, uint32_t e, const SgLaneMemo& mm, EH_G Piece* pc, EH_G SgParam* par, bptr mark, uint32_t attr_cap) {
  SgLaneTag R; R.ok = 0; R.bail = 0; R.kind = 0; R.next = 0; R.nexte = 0; R.npc = 0; R.npar = 0; R.ffc = 0; R.fff = 0; R.unmarked = 0; R.capped = 0;
  bool bail = false; uint32_t budget = SG_LANE_BUDGET, reach = e + 1, far = 0;
  auto EV = [&](uint32_t i) -> uint32_t {                                  // (an unconditional LDS read, kept apart from the global one: EH_KEEP)
    const uint32_t o = i - cbase; const bool in = o < 4096u;
    uint32_t v = g_fuse_lds[in ? o : 0u];
    EH_KEEP(v);
    if (!in) { v = ev[i]; if (++far > SG_LANE_FAR) bail = true; }
    return v;
  };
  auto have = [&](uint32_t i) -> bool {                                    // is there an event i?
    if (i >= nev) return false;
    if (budget == 0) { bail = true; return false; }
    budget--;
    if (i >= reach) reach = i + 1;
    return true;
  };
  const uint32_t lt = EV(e) >> 4;
  uint32_t pos = lt + 1, ei = e + 1, np = 0, nq = 0;
  bool ws_sp = false;
  auto cls_here = [&]() -> uint32_t { if (!have(ei)) return 0u; uint32_t v = EV(ei); return (v >> 4) == pos ? (v & 15u) : 0u; };
  auto step1 = [&]() { if (have(ei) && (EV(ei) >> 4) == pos) ei++; pos++; };
  auto skipws = [&]() {                                                    // ws/1 :176-177
    ws_sp = false;
    while (have(ei)) { uint32_t v = EV(ei); if (!((ES_WS >> (v & 15u)) & 1u) || (v >> 4) != pos) break; ws_sp = (v & 15u) == E_SPACE; pos++; ei++; }
  };
  auto find_set = [&](uint32_t from, uint32_t set) -> uint32_t {           // find_stop of the wave-wide machine: set is ES_STOP or ES_EV
    uint32_t i = from;
    for (int k = 0; k < 8; k++) { if (!have(i)) return nev; if ((set >> (EV(i) & 15u)) & 1u) return i; i++; }
    for (;;) {                                                             // a name that runs over many events: the next-stop table
      if (!have(i)) return nev;
      const uint32_t j = mm.nstop[i];
      if (j >= nev) return nev;
      if (j >= reach) reach = j + 1;
      if (set == ES_STOP || ((set >> (EV(j) & 15u)) & 1u)) return j;       // ES_EV: a "/>" does not end the name
      i = j + 1;
    }
  };
  auto find_one = [&](uint32_t from, uint32_t cls, uint32_t ff) -> uint32_t {      // find1 of the wave-wide machine
    if (from >= ff) return nev;
    uint32_t i = from;
    while (have(i)) { if ((EV(i) & 15u) == cls) return i; i++; if (i - from > SG_LANE_SEARCH) { bail = true; return nev; } }
    if (!bail) { R.ffc = cls; R.fff = from; reach = nev; }                // ran to the end of the block
    return nev;
  };
  auto put = [&](cbptr p, uint32_t len) { if (MODE == 1) { Piece q; q.ptr = (uint64_t)p; q.len = len; q.rep = 1; pc[np] = q; } np++; };
  skipws();
  const uint32_t tag0 = pos;
  const bool tight = tag0 == lt + 1;
  uint32_t kind = 0, na = pos, nb = pos, next = 0, nexte = 0;
  bool ok = false;
  do {
    if (pos >= L) break;
    uint32_t c0 = cls_here();
    if (c0 == E_BANG) {                                                    // :104-105
      bool d1 = false, d2 = false;
      if (ei + 2 < nev) {
        if (!have(ei + 2)) break;
        uint32_t e1 = EV(ei + 1), e2 = EV(ei + 2);
        d1 = (e1 >> 4) == pos + 1 && ((ES_DASH >> (e1 & 15u)) & 1u); d2 = (e2 >> 4) == pos + 2 && ((ES_DASH >> (e2 & 15u)) & 1u);
      }
      if (d1 && d2) {                                                      // {'!--',DT} :117-118
        uint32_t dta = pos + 3;
        uint32_t j = find_one(ei + 3, E_CMTEND, mm.cmt);
        if (j >= nev) break;
        uint32_t en = EV(j) >> 4;
        put(tight ? H + lt : sglit(SL_CMT), 4); put(H + dta, en - dta); put(H + en, 3);
        kind = TK_COMMENT; next = en + 3; nexte = j + 3; ok = true; break;
      }
      step1(); skipws();                                                   // {'!',DT} :113-115
      uint32_t dta = pos;
      uint32_t j = find_one(ei, E_GT, mm.gt);
      if (j >= nev) break;
      uint32_t en = EV(j) >> 4;
      put(tight ? H + lt : sglit(SL_LTBANG), 2); put(H + dta, en - dta); put(H + en, 1);
      kind = TK_BANG; next = en + 1; nexte = j + 1; ok = true; break;
    }
    if (c0 == E_QM || c0 == E_QGT) {                                       // {que,DT} :106,:120-122
      step1(); skipws();
      uint32_t dta = pos;
      uint32_t j = find_one(ei, E_QGT, mm.qgt);
      if (j >= nev) break;
      uint32_t en = EV(j) >> 4;
      put(tight ? H + lt : sglit(SL_LTQ), 2); put(H + dta, en - dta); put(H + en, 2);
      kind = TK_QUE; next = en + 2; nexte = j + 2; ok = true; break;
    }
    if (c0 == E_SL || c0 == E_SLGT) {                                      // {end_tag,Tag} :107,:128-132
      step1(); skipws();
      na = pos;
      uint32_t j = find_set(ei, ES_EV);
      if (j >= nev) break;
      nb = EV(j) >> 4; pos = nb; ei = j;
      skipws();
      if (cls_here() != E_GT) break;
      put(tight ? H + lt : sglit(SL_LTSL), 2); put(H + na, nb - na); put(H + pos, 1);
      kind = TK_CLOSE; next = pos + 1; nexte = ei + 1; ok = true; break;
    }
    // {tag,Tag} :108-111
    uint32_t j = find_set(ei, ES_STOP);
    if (j >= nev) break;
    uint32_t ej = EV(j);
    nb = ej >> 4;
    put(H + lt, 1); put(H + na, nb - na); put(sglit(SL_SP), 0);
    if ((ej & 15u) == E_SLGT) { put(sglit(SL_SPSLGT), 3); kind = TK_SC; next = nb + 2; nexte = j + 2; ok = true; break; }
    pos = nb; ei = j;
    skipws();
    for (;;) {                                                             // attribute loop :134-160
      if (pos >= L || bail) break;
      if (nq >= attr_cap) { bail = true; R.capped = 1; break; }
      if (nq >= 16) {                                                      // the memo of failed attribute-loop entries
        if (mm.bad && mm.bad[pos]) break;
        if (MODE == 2) mark[pos] = 1;
        R.unmarked++;
      }
      uint32_t ca = cls_here();
      if (ca == E_SLGT) { put(ws_sp ? H + pos - 1 : sglit(SL_SPSLGT), 3); kind = TK_SC; next = pos + 2; nexte = ei + 2; ok = true; break; }
      if (ca == E_GT) { put(H + pos, 1); kind = TK_OPEN; next = pos + 1; nexte = ei + 1; ok = true; break; }
      const bool sp_before = ws_sp;
      if (ca == E_EQ) break;
      uint32_t an = pos;
      step1();
      uint32_t ja = find_set(ei, ES_STOP);
      if (ja >= nev) break;
      uint32_t ae = EV(ja) >> 4; pos = ae; ei = ja;
      skipws();
      uint32_t va = pos, vb = pos, delim = 0, eqpos = 0xFFFFFFFFu;
      if (pos < L && cls_here() == E_EQ) {
        eqpos = pos;
        step1(); skipws();
        if (pos >= L) break;
        uint32_t cv = cls_here();
        if (cv == E_SQ || cv == E_DQ) {
          uint32_t jq = find_one(ei + 1, cv, cv == E_SQ ? mm.sq : mm.dq);
          if (jq >= nev) break;
          va = pos + 1; vb = EV(jq) >> 4; delim = cv == E_SQ ? 1u : 2u;
          pos = vb + 1; ei = jq + 1;
        } else {
          uint32_t ju = find_set(ei, ES_STOP);
          if (ju >= nev) break;
          va = pos; vb = EV(ju) >> 4; pos = vb; ei = ju;
        }
        skipws();
      }
      if (MODE == 1) { SgParam q; q.na = an; q.nb = ae; q.va = va; q.vb = vb; q.delim = delim; q.pad = 0; par[nq] = q; }
      nq++;
      const bool has = vb > va;
      cbptr qp = sglit(delim == 1 ? SL_SQ : SL_DQ);
      put(sp_before ? H + an - 1 : sglit(SL_SP), 1); put(H + an, ae - an);
      put(has ? H + eqpos : sglit(SL_EQ), has ? 1u : 0u); put(has && delim ? H + va - 1 : qp, has && delim ? 1u : 0u);
      put(H + va, has ? vb - va : 0u); put(has && delim ? H + vb : qp, has && delim ? 1u : 0u);
    }
  } while (false);
  R.ok = (ok && !bail) ? 1u : 0u; R.bail = bail ? 1u : 0u; R.kind = kind; R.next = next; R.nexte = nexte; R.tag0 = tag0; R.lt = lt;
  R.npc = np; R.npar = nq; R.na = na; R.nb = nb; R.reach = reach;
  if (ok || bail) { R.unmarked = 0; R.ffc = 0; }
  return R;
}

// ---- one ATTRIBUTE per lane (round 5) ---------------------------------------------------------------------------------------------
// The heaviest cases of every pass are `sgm` on documents whose tags hold tens of thousands of attributes (a megabyte of words
// between a '<' and the next '>': 100 000 iterations of the attribute loop :134-160 at ~4 000 cycles each in the wave-wide machine,
// 1 600 in a lane of a batch - and nd / bu call sgm a dozen times per case).  One iteration of that loop depends on nothing but
// the place it starts at (pos after ws/1, the first event there, whether the last blank skipped was 0x20), and nearly every
// iteration starts right behind a run of white space or behind a closing quote.  So the wave-wide machine, once a tag has shown
// to be long, takes the next 63 such places from the event window as CANDIDATES, lets every lane run ONE iteration from its
// candidate (lane 0: from where the machine is), and then walks the chain - lane k's iteration ends where some later lane's began -
// to find the iterations the machine would really have made; their parameters and pieces are written by all lanes at once.
// A lane reports `term` for everything that is not a plain completed attribute (the tag's end, '=', a search that fails or runs
// far, the end of the block): the chain stops in front of it and the wave-wide machine makes that iteration itself, statement for
// statement as before - so this function only has to be right about the plain case.
struct SgLaneAttr { uint32_t term, npos, nei, nws, an, ae, va, vb, delim, eqpos, reach; };
constexpr uint32_t SG_ATTR_BUDGET = 512;            // events one lane may look at for its attribute
__device__ __noinline__ SgLaneAttr sg_lane_attr(cbptr H, uint32_t L, cwptr ev, uint32_t nev, uint32_t cbase, uint32_t pos, uint32_t ei, uint32_t ws0, const SgLaneMemo& mm) {
  (void)H;
  SgLaneAttr R; R.term = 1; R.npos = pos; R.nei = ei; R.nws = ws0; R.an = pos; R.ae = pos; R.va = pos; R.vb = pos; R.delim = 0; R.eqpos = 0xFFFFFFFFu; R.reach = ei;
  bool bail = false; uint32_t budget = SG_ATTR_BUDGET, reach = ei, far = 0;
  auto EV = [&](uint32_t i) -> uint32_t {
    const uint32_t o = i - cbase; const bool in = o < 4096u;
    uint32_t v = g_fuse_lds[in ? o : 0u];
    EH_KEEP(v);
    if (!in) { v = ev[i]; if (++far > 8u) bail = true; }
    return v;
  };
  auto have = [&](uint32_t i) -> bool {
    if (i >= nev) return false;
    if (budget == 0) { bail = true; return false; }
    budget--;
    if (i >= reach) reach = i + 1;
    return true;
  };
  bool ws_sp = ws0 != 0;
  auto cls_here = [&]() -> uint32_t { if (!have(ei)) return 0u; uint32_t v = EV(ei); return (v >> 4) == pos ? (v & 15u) : 0u; };
  auto step1 = [&]() { if (have(ei) && (EV(ei) >> 4) == pos) ei++; pos++; };
  auto skipws = [&]() {
    ws_sp = false;
    while (have(ei)) { uint32_t v = EV(ei); if (!((ES_WS >> (v & 15u)) & 1u) || (v >> 4) != pos) break; ws_sp = (v & 15u) == E_SPACE; pos++; ei++; }
  };
  auto find_stop = [&](uint32_t from) -> uint32_t {                        // ES_STOP
    uint32_t i = from;
    for (int k = 0; k < 8; k++) { if (!have(i)) return nev; if ((ES_STOP >> (EV(i) & 15u)) & 1u) return i; i++; }
    if (!have(i)) return nev;
    const uint32_t j = mm.nstop[i];                                        // a name that runs over many events: the next-stop table
    if (j < nev && j >= reach) reach = j + 1;
    return j;
  };
  do {
    if (pos >= L) break;
    const uint32_t ca = cls_here();
    if (ca == E_SLGT || ca == E_GT || ca == E_EQ) break;                   // the machine's: {etag,..}, the tag's end, the throw
    R.an = pos;
    step1();
    const uint32_t ja = find_stop(ei);
    if (ja >= nev || bail) break;
    R.ae = EV(ja) >> 4; pos = R.ae; ei = ja;
    skipws();
    R.va = pos; R.vb = pos;
    if (pos < L && cls_here() == E_EQ) {
      R.eqpos = pos;
      step1(); skipws();
      if (pos >= L) break;
      const uint32_t cv = cls_here();
      if (cv == E_SQ || cv == E_DQ) {
        const uint32_t ff = cv == E_SQ ? mm.sq : mm.dq;
        uint32_t i = ei + 1, jq = nev;
        if (i < ff) { while (have(i)) { if ((EV(i) & 15u) == cv) { jq = i; break; } i++; } }
        if (jq >= nev) break;                                              // not found (or not within reach): the machine's
        R.va = pos + 1; R.vb = EV(jq) >> 4; R.delim = cv == E_SQ ? 1u : 2u;
        pos = R.vb + 1; ei = jq + 1;
      } else {
        const uint32_t ju = find_stop(ei);
        if (ju >= nev || bail) break;
        R.va = pos; R.vb = EV(ju) >> 4; pos = R.vb; ei = ju;
      }
      skipws();
    }
    if (bail) break;
    R.term = 0; R.npos = pos; R.nei = ei; R.nws = ws_sp ? 1u : 0u;
  } while (false);
  R.reach = reach;
  return R;
}

// tokenize/1 :66-98 + tz/2 :100-164.  0 ok; -1 incorrect_sgml; -2 an error other than incorrect_sgml in the
// first tag (outside any try: the worker dies); -3 engine capacity (c.status set).
//
// Phase 1 turns the block into an EVENT list — (position, class) of every special byte, extracted 1 KiB per step by
// the whole wave.  Phase 2 is tz/2 written as plain loops over that list: every tz state is "the next event whose
// class is in set X", answered for 64 events at a time with one ballot; ordinary bytes are never looked at.
// A failed tag is retried from its next '<' by the reference (catch _:_ :79-96), which re-scans the same attribute
// list again and again on text like "<a <b <c ... " without a '>' — quadratic, minutes on BEAM.  The attribute
// loop entered at byte p always ends the same way, whatever tag led there, so the positions at which a FAILED
// attempt entered it are remembered (after its 16th attribute) and a later attempt arriving at one of them fails
// at once: same tokens, linear time.
__device__ __noinline__ int sgml_tokenize(Ctx&, cbptr H, uint32_t L, SgDoc* out) {
  EH_CTX;
  const int l = EH_LANE;
#ifdef EH_PROF
  const uint64_t ph1_t0 = __builtin_readcyclecounter();
#endif
  // ---- phase 1: events
  wptr ev = (wptr)ws_alloc(c, ((uint64_t)L + 80) * 4);
  if (!ev) return -3;
  uint32_t nev = 0, nlt = 0, nstop = 0;
  for (uint32_t tb = 0; tb < L; tb += 1024) {
    uint32_t i0 = tb + 16u * (uint32_t)l;
    uint8_t b[18];
    if (i0 + 18 <= L) { uint4 v = ldg16(H + i0); __builtin_memcpy(b, &v, 16); b[16] = H[i0 + 16]; b[17] = H[i0 + 17]; }
    else { for (uint32_t k = 0; k < 18; k++) b[k] = i0 + k < L ? H[i0 + k] : 0; }
    uint32_t cls[16]; uint32_t cnt = 0, lts = 0, stops = 0;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) {
      uint32_t x = b[k];
      uint32_t cl = i0 + k < L ? sg_class(x) : 0u;
      if (cl == E_SL && b[k + 1] == '>') cl = E_SLGT;
      if (cl == E_QM && b[k + 1] == '>') cl = E_QGT;
      if (cl == E_DASH && b[k + 1] == '-' && b[k + 2] == '>') cl = E_CMTEND;
      cls[k] = cl; cnt += cl ? 1u : 0u; lts += cl == E_LT ? 1u : 0u; stops += (cl == E_WS || cl == E_SPACE || cl == E_GT || cl == E_EQ || cl == E_SL || cl == E_SLGT) ? 1u : 0u;
    }
    uint32_t inc = wave_incl_scan(cnt);
    uint32_t o = nev + inc - cnt;
#pragma unroll
    for (uint32_t k = 0; k < 16; k++) if (cls[k]) ev[o++] = ((i0 + k) << 4) | cls[k];
    nev += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    nlt += lts; nstop += stops;
  }
  nlt = wave_sum(nlt); nstop = wave_sum(nstop);
#ifdef EH_PROF
  uint64_t sub_t0 = __builtin_readcyclecounter();
#define SG_SUB0() do { sub_t0 = __builtin_readcyclecounter(); } while (0)
#define SG_SUB(k) do { uint64_t n_ = __builtin_readcyclecounter(); if (l == 0) { atomicAdd(&c.p->prof[2 * (k)], (unsigned long long)(n_ - sub_t0)); atomicAdd(&c.p->prof[2 * (k) + 1], 1ull); } sub_t0 = n_; } while (0)
#define SG_CNT(k, v) do { if (l == 0) { atomicAdd(&c.p->prof[2 * (k)], (unsigned long long)(v)); atomicAdd(&c.p->prof[2 * (k) + 1], 1ull); } } while (0)
  if (l == 0) { atomicAdd(&c.p->prof[2 * 117], (unsigned long long)(sub_t0 - ph1_t0)); atomicAdd(&c.p->prof[2 * 117 + 1], 1ull); atomicAdd(&c.p->prof[2 * 119], (unsigned long long)nev); atomicAdd(&c.p->prof[2 * 119 + 1], (unsigned long long)L); }
  uint64_t tz_t0 = __builtin_readcyclecounter();
#define SG_TZ(k) do { uint64_t n_ = __builtin_readcyclecounter(); if (l == 0) { atomicAdd(&c.p->prof[2 * (k)], (unsigned long long)(n_ - tz_t0)); atomicAdd(&c.p->prof[2 * (k) + 1], 1ull); } tz_t0 = n_; } while (0)
#else
#define SG_TZ(k) do {} while (0)
#define SG_SUB0() do {} while (0)
#define SG_SUB(k) do {} while (0)
#define SG_CNT(k, v) do {} while (0)
#endif
  if (nlt == 0) return -1;                                                 // tz(nil, <<>>) :102
  // capacity: tokens <= 2 x '<' + 2; a parameter needs a byte of the stop set after its name
  uint32_t cap_tok = 2 * nlt + 8, cap_par = nstop + 8;
  uint64_t cap_pc = 4ull * cap_tok + (uint64_t)SG_PARPCS * cap_par + nlt + 32;
  EH_G SgTok* tok = (EH_G SgTok*)ws_alloc(c, (uint64_t)cap_tok * sizeof(SgTok));
  EH_G SgParam* par = (EH_G SgParam*)ws_alloc(c, (uint64_t)cap_par * sizeof(SgParam));
  EH_G Piece* pc = (EH_G Piece*)ws_alloc(c, cap_pc * sizeof(Piece));
  if (!tok || !par || !pc) return -3;
  wave_sync();
  uint32_t ntok = 0, npar = 0, npc = 0;

  // ---- phase 2
  // The events the state machine is walking through sit in LDS (g_fuse_lds is idle outside fuse): SG_EVC of them around the
  // place of work, refilled in one batch of loads when it is left.  Read from the work area 64 at a time, every refill of
  // the lane window was a global load that had to wait for all the piece / token stores issued before it (stores count in
  // vmcnt on this ISA) - two memory round trips every few tokens, which is what the tokenizer of a megabyte block spent its
  // hundreds of millions of cycles in (the heaviest cases of a pass are sgm on pumped documents).
  constexpr uint32_t SG_EVC = 4096, SG_EVBACK = 512;
  uint32_t cbase = 0xFFFFFFFFu;                                            // events [cbase, cbase + SG_EVC) are in g_fuse_lds
  uint32_t bev = 0, bbase = 0xFFFFFFFFu;                                   // lane k holds event bbase + k
  uint32_t reach_ev = 0;                                                  // events below this index may have been looked at (replay, below)
  auto need = [&](uint32_t i) {
    if (bbase != 0xFFFFFFFFu && i >= bbase && i < bbase + 64) return;
    bbase = i & ~63u;
    if (bbase + 64 > reach_ev) reach_ev = bbase + 64;
    if (cbase == 0xFFFFFFFFu || bbase < cbase || bbase + 64 > cbase + SG_EVC) {
      cbase = bbase > SG_EVBACK ? bbase - SG_EVBACK : 0u;                 // (a failed tag is retried from its next '<': a little history stays)
      lanes_sync();
      for (uint32_t k0 = 0; k0 < SG_EVC; k0 += 256) {
        uint32_t v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { uint32_t j = cbase + k0 + 64u * (uint32_t)u + (uint32_t)l; v[u] = j < nev ? ev[j] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; u++) g_fuse_lds[k0 + 64u * (uint32_t)u + (uint32_t)l] = v[u];
      }
      lanes_sync();
    }
    bev = g_fuse_lds[bbase - cbase + (uint32_t)l];
  };
  // index of the first event >= from whose class is in `set`; nev if there is none
  auto find = [&](uint32_t from, uint32_t set) -> uint32_t {
    while (from < nev) {
      need(from);
      unsigned long long m = __ballot(((set >> (bev & 15u)) & 1u) != 0 && bbase + (uint32_t)l >= from && bbase + (uint32_t)l < nev);
      if (m) return bbase + (uint32_t)__builtin_ctzll(m);
      from = bbase + 64;
    }
    return nev;
  };
  auto evget = [&](uint32_t i) -> uint32_t { need(i); return (uint32_t)__builtin_amdgcn_readlane((int)bev, (int)(i - bbase)); };
  // The same two guards against quadratic rescans by retried tags (see below):
  // a search for ONE class that found nothing from event f finds nothing from any later event either ...
  uint32_t ff_gt = nev, ff_qgt = nev, ff_cmt = nev, ff_sq = nev, ff_dq = nev;
  auto find1 = [&](uint32_t from, uint32_t cls, uint32_t& ff) -> uint32_t {
    if (from >= ff) return nev;
    uint32_t j = find(from, 1u << cls);
    if (j >= nev) ff = from;
    return j;
  };
  // ... and a name that runs over many events ("<<<<<< ... ") is skipped through a next-stop table, built on first need
  wptr nstop_tab = nullptr;
  auto build_nstop = [&]() __attribute__((always_inline)) -> bool {                                       // nstop_tab[i] = the first event >= i of the stop set (nev: none)
    EH_CTX;
    nstop_tab = (wptr)ws_alloc(c, ((uint64_t)nev + 64) * 4);
    if (!nstop_tab) return false;
#ifdef EH_PROF
    const uint64_t bn_t0 = __builtin_readcyclecounter();
#endif
    uint32_t carry = nev;
    for (uint32_t base = (nev - 1) & ~63u;; base -= 64) {
      uint32_t j = base + (uint32_t)l; uint32_t e = j < nev ? ev[j] : 0u;
      unsigned long long ms = __ballot(j < nev && ((ES_STOP >> (e & 15u)) & 1u));
      unsigned long long mm = ms & ~((1ull << l) - 1);
      if (j < nev) nstop_tab[j] = mm ? base + (uint32_t)__builtin_ctzll(mm) : carry;
      if (ms) carry = base + (uint32_t)__builtin_ctzll(ms);
      if (base == 0) break;
    }
    wave_sync();
#ifdef EH_PROF
    if (l == 0) { atomicAdd(&c.p->prof[2 * 116], (unsigned long long)(__builtin_readcyclecounter() - bn_t0)); atomicAdd(&c.p->prof[2 * 116 + 1], 1ull); }
#endif
    return true;
  };
  auto find_stop = [&](uint32_t from, uint32_t set) -> uint32_t {          // set: ES_STOP or ES_EV
    if (from >= nev) return nev;
    need(from);
    unsigned long long m = __ballot(((set >> (bev & 15u)) & 1u) != 0 && bbase + (uint32_t)l >= from && bbase + (uint32_t)l < nev);
    if (m) return bbase + (uint32_t)__builtin_ctzll(m);
    from = bbase + 64;
    if (from >= nev) return nev;
    if (!nstop_tab && !build_nstop()) return 0xFFFFFFFFu;
    for (;;) {
      uint32_t j = uni(nstop_tab[from]);
      if (j >= nev) return nev;
      if (set == ES_STOP || ((set >> (evget(j) & 15u)) & 1u)) return j;    // ES_EV: a "/>" does not end the name
      from = j + 1;
      if (from >= nev) return nev;
    }
  };
  uint32_t pos = 0, ei = 0;                                                // ei = first event at or after byte pos
  auto cls_here = [&]() -> uint32_t { if (ei >= nev) return 0u; uint32_t e = evget(ei); return (e >> 4) == pos ? (e & 15u) : 0u; };
  auto step1 = [&]() { if (ei < nev && (evget(ei) >> 4) == pos) ei++; pos++; };        // consume one byte
  bool ws_sp = false;                                                      // the last skipws() ended on a blank (0x20)
  auto skipws = [&]() {                                                    // ws/1 :176-177
    ws_sp = false;
    while (ei < nev) {
      need(ei);
      uint32_t off = ei - bbase;
      bool ok = (uint32_t)l >= off && bbase + (uint32_t)l < nev && ((ES_WS >> (bev & 15u)) & 1u) && (bev >> 4) == pos + ((uint32_t)l - off);
      unsigned long long notok = ~__ballot(ok) & ~((1ull << off) - 1);
      uint32_t run = notok ? (uint32_t)__builtin_ctzll(notok) - off : 64u - off;
      if (run > 0) ws_sp = ((uint32_t)__builtin_amdgcn_readlane((int)bev, (int)(off + run - 1)) & 15u) == E_SPACE;
      pos += run; ei += run;
      if (off + run < 64) break;
    }
  };
  auto put = [&](cbptr p, uint32_t len) { if (l == 0) { Piece q; q.ptr = (uint64_t)p; q.len = len; q.rep = 1; pc[npc] = q; } npc++; };

  // ---- replay of periodic documents.  The heaviest cases of a pass are `sgm` on documents that sr / lr / sgm pumps have grown to
  // megabytes: thousands of copies of one run of elements, ~3 000 cycles of this sequential machine per tag.  Right after an accepted
  // tag the machine's state is a function of the position alone (text state, nothing pending), and what it does next depends only on
  // the bytes it looks at.  So when the block is periodic around here (H[k] == H[k + P], found like eh_fuse_red.h finds its cuts) and
  // the machine, started at s, is back in that state at exactly s + P, the tokens of [s, s + P) repeat - shifted by P - for every
  // further period whose look-ahead stays inside the periodic stretch.  One period is tokenized as a template, the next one as a
  // check (every token, piece and parameter must be the template's, shifted), the rest are written by the whole wave from the template.
  const bool lanes_on = !(c.p->flags & EH_FLAG_SGML_NO_LANES);
  static_assert(4096 + 64 <= EH_FUSE_LDS_WORDS, "g_fuse_lds: event window + the batch's '<' list");
  uint32_t rp_P = 0, rp_lo = 0, rp_end = 0;                                // the stretch in hand: H[k] == H[k + P] for k in [rp_lo, rp_end - P)
  int rp_state = 0, rp_tries = 0;                                          // 0 idle, 1 in the template period, 2 in the check period
  // A document may hold several stretches (an element nested a thousand times: a run of open tags, a run of close tags; a block pumped
  // twice).  Stretches are found best-first in what lies ahead (fr_find_cut: anchors at the eighths), then, when the best one starts far
  // ahead, in the gap before it; they are replayed in the order the machine reaches them.  At most 8 searches per document.
  uint32_t rq_lo[4], rq_end[4], rq_P[4]; int rq_n = 0, rp_budget = 8; uint32_t rp_scan_from = 0;
  auto rp_detect = [&](uint32_t a, uint32_t b) __attribute__((always_inline)) -> bool {                   // the best stretch of H[a, b) goes onto the stack
    EH_CTX;
    if (rp_budget <= 0 || rq_n >= 4 || b <= a || b - a < 16384u) return false;
    rp_budget--;
    uint32_t u = 0, D = 0, P = 0;
    const bool found = fr_find_cut(H + a, b - a, 0, &u, &D, &P) && P > 0;
#ifdef EH_PROF
    if (l == 0) { atomicAdd(&c.p->prof[2 * 95 + 1], 1ull); atomicAdd(&c.p->prof[2 * 95], found ? 1ull : 0ull); }      // searches, stretches found
#endif
    if (!found) return false;
    rq_lo[rq_n] = a + u - P; rq_end[rq_n] = a + u + D; rq_P[rq_n] = P; rq_n++;
    return true;
  };
  auto rp_load = [&](uint32_t at) __attribute__((always_inline)) {                                        // the next stretch the machine, now at byte `at`, will reach
    rp_P = 0; rp_state = 0; rp_tries = 0;
    for (;;) {
      while (rq_n > 0 && (uint64_t)at + 4ull * rq_P[rq_n - 1] > rq_end[rq_n - 1]) rq_n--;        // behind us (or too little of it left)
      if (rq_n == 0) {
        const uint32_t from = at > rp_scan_from ? at : rp_scan_from;
        if (!rp_detect(from, L)) return;
        rp_scan_from = rq_end[rq_n - 1];
        continue;
      }
      const int top = rq_n - 1;
      if (rq_lo[top] > at && rq_lo[top] - at >= 32768u && rp_detect(at, rq_lo[top])) continue;   // something earlier in the gap
      rp_P = rq_P[top]; rp_lo = rq_lo[top]; rp_end = rq_end[top]; rq_n--;
      return;
    }
  };
  if (L >= 16384 && nlt >= 64 && !(c.p->flags & EH_FLAG_SGML_NO_REPLAY)) rp_load(0);
  uint32_t rp_s0 = 0, rp_tok0 = 0, rp_pc0 = 0, rp_par0 = 0, rp_e0 = 0, rp_dtok = 0, rp_dpc = 0, rp_dpar = 0, rp_de = 0, rp_reach1 = 0;
  // memo of attribute-loop entries of failed attempts (see above)
  bptr bad = nullptr; wptr chain = nullptr; uint32_t nchain = 0;
  int rc = 0;
  bool first = true;
  uint32_t lt = 0, seg_start = 0, text_p0 = 0, text_len = 0, seg_slot = 0;
  // ---- replay bookkeeping: the machine has just accepted a tag (text state, nothing pending)
  // (always_inline, like lane_batches below: a [&] lambda that is called, not inlined, moves every local it captures - pos, ei, npc ... -
  // from registers into stack memory, for the wave-wide machine as well: three times its time on tags of thousands of attributes)
  auto after_accept = [&]() __attribute__((always_inline)) -> int {
    EH_CTX;                                                                // (not the captured reference: a lambda that is not inlined would carry a generic pointer to the LDS context)
      if (rp_P) {                                                          // ---- replay (see above): this is the state "right after an accepted tag"
        if (rp_state == 1 && pos == rp_s0 + rp_P) {                        // the template period is complete
          rp_dtok = ntok - rp_tok0; rp_dpc = npc - rp_pc0; rp_dpar = npar - rp_par0; rp_de = ei - rp_e0; rp_reach1 = reach_ev; rp_state = 2;
          reach_ev = (bbase != 0xFFFFFFFFu && bbase + 64 > ei) ? bbase + 64 : ei;   // (the lane window that is loaded counts as looked at)
        } else if (rp_state == 2 && pos == rp_s0 + 2u * rp_P) {            // the check period is complete
          wave_sync();                                                     // (lane 0's stores of the two periods are read below)
          bool same = ntok - rp_tok0 == 2u * rp_dtok && npc - rp_pc0 == 2u * rp_dpc && npar - rp_par0 == 2u * rp_dpar && ei - rp_e0 == 2u * rp_de && rp_dtok > 0;
          const uint64_t h0 = (uint64_t)(uintptr_t)H, h1 = h0 + L;
          if (same) {
            bool ne = false;
            for (uint32_t t = l; t < rp_dtok; t += 64) {
              SgTok a = tok[rp_tok0 + t], b = tok[rp_tok0 + rp_dtok + t];
              const bool text = (a.kind & TK_KIND) == TK_TEXT;
              const uint32_t sh = text ? 0u : rp_P, shp = text ? 0u : rp_dpar;
              ne |= b.kind != a.kind || b.p0 != a.p0 + rp_dpc || b.np != a.np || b.na != a.na + sh || b.nb != a.nb + sh || b.par0 != a.par0 + shp || b.npar != a.npar;
            }
            for (uint32_t t = l; t < rp_dpc; t += 64) {
              Piece a = pc[rp_pc0 + t], b = pc[rp_pc0 + rp_dpc + t];
              const bool inblk = a.ptr >= h0 && a.ptr < h1;
              ne |= b.len != a.len || b.rep != a.rep || b.ptr != a.ptr + (inblk ? rp_P : 0u);
            }
            for (uint32_t t = l; t < rp_dpar; t += 64) {
              SgParam a = par[rp_par0 + t], b = par[rp_par0 + rp_dpar + t];
              ne |= b.na != a.na + rp_P || b.nb != a.nb + rp_P || b.va != a.va + rp_P || b.vb != a.vb + rp_P || b.delim != a.delim;
            }
            same = __ballot(ne) == 0;
          }
          // how far ahead of its start a period looked (events -> bytes; a search that ran to the end of the block leaves none to replay)
          uint32_t r1 = rp_reach1 > rp_e0 ? rp_reach1 : rp_e0, r2 = reach_ev > rp_e0 + rp_de ? reach_ev : rp_e0 + rp_de;
          uint32_t b1 = r1 >= nev ? L : (evget(r1) >> 4), b2 = r2 >= nev ? L : (evget(r2) >> 4);
          uint32_t rel1 = b1 - rp_s0, rel2 = b2 - (rp_s0 + rp_P);
          uint32_t rel = (rel1 > rel2 ? rel1 : rel2) + 4u;                 // (+ the bytes a class looks ahead: "/>", "?>", "-->")
          uint32_t J = 0;
          if (same && rp_end > rp_s0 + rel) J = (rp_end - rel - rp_s0) / rp_P;   // periods 0 .. J-1 see only bytes of the periodic stretch
          if (J > 3) {
            const uint32_t N = J - 2u;
            if ((uint64_t)npc + (uint64_t)N * rp_dpc + 16 > cap_pc || (uint64_t)ntok + (uint64_t)N * rp_dtok + 2 > cap_tok || (uint64_t)npar + (uint64_t)N * rp_dpar + 1 > cap_par) { EH_SET_OVERFLOW(c, 603); return -3; }
            for (uint64_t idx = l; idx < (uint64_t)N * rp_dtok; idx += 64) {
              const uint32_t j = 2u + (uint32_t)(idx / rp_dtok), t = (uint32_t)(idx % rp_dtok);
              SgTok a = tok[rp_tok0 + t];
              const bool text = (a.kind & TK_KIND) == TK_TEXT;
              a.p0 += j * rp_dpc;
              if (!text) { a.na += j * rp_P; a.nb += j * rp_P; a.par0 += j * rp_dpar; }
              tok[rp_tok0 + j * rp_dtok + t] = a;
            }
            for (uint64_t idx = l; idx < (uint64_t)N * rp_dpc; idx += 64) {
              const uint32_t j = 2u + (uint32_t)(idx / rp_dpc), t = (uint32_t)(idx % rp_dpc);
              Piece a = pc[rp_pc0 + t];
              if (a.ptr >= h0 && a.ptr < h1) a.ptr += (uint64_t)j * rp_P;
              pc[rp_pc0 + j * rp_dpc + t] = a;
            }
            for (uint64_t idx = l; idx < (uint64_t)N * rp_dpar; idx += 64) {
              const uint32_t j = 2u + (uint32_t)(idx / rp_dpar), t = (uint32_t)(idx % rp_dpar);
              SgParam a = par[rp_par0 + t];
              a.na += j * rp_P; a.nb += j * rp_P; a.va += j * rp_P; a.vb += j * rp_P;
              par[rp_par0 + j * rp_dpar + t] = a;
            }
            wave_sync();
#ifdef EH_PROF
            if (l == 0) { atomicAdd(&c.p->prof[2 * 94], (unsigned long long)N * rp_dtok); atomicAdd(&c.p->prof[2 * 94 + 1], 1ull); }   // tokens written by replay, replays
#endif
            ntok += N * rp_dtok; npc += N * rp_dpc; npar += N * rp_dpar;
            pos = rp_s0 + J * rp_P; ei = rp_e0 + J * rp_de;
            seg_start = pos; text_p0 = npc; text_len = 0;
            bbase = 0xFFFFFFFFu;
          }
#ifdef EH_PROF
          if (l == 0 && J <= 3) { atomicAdd(&c.p->prof[2 * 99 + 1], 1ull); atomicAdd(&c.p->prof[2 * 99], same ? 1ull : 0ull); }   // checked but not replayed; of them: look-ahead too long
#endif
          rp_load(pos);                                                    // this stretch is done; is there another one ahead?
        } else if (rp_state != 0 && pos > rp_s0 + (uint32_t)rp_state * rp_P) {
          rp_state = 0;                                                    // the machine was not back in this state a period later: try from here
          if (++rp_tries >= 8) {
            rp_load(rp_end);
#ifdef EH_PROF
            if (l == 0) atomicAdd(&c.p->prof[2 * 89 + 1], 1ull);        // gave up: never back in the state a period later
#endif
          }
        }
        if (rp_P && rp_state == 0 && (uint64_t)pos + 4ull * rp_P > rp_end) rp_load(pos);          // walked past it
        if (rp_P && rp_state == 0 && pos >= rp_lo + rp_P && (uint64_t)pos + 4ull * rp_P <= rp_end) {
          rp_state = 1; rp_s0 = pos; rp_tok0 = ntok; rp_pc0 = npc; rp_par0 = npar; rp_e0 = ei;
          reach_ev = (bbase != 0xFFFFFFFFu && bbase + 64 > ei) ? bbase + 64 : ei;
        }
      }
    return 0;
  };
  // ---- lane batches: the next '<' of the block attempted one per lane (sg_lane_attempt).  Called in the text state - behind an accepted
  // tag or a failed attempt - where what the machine does next depends on the bytes ahead alone.  A batch commits exactly what the
  // machine would have done tag by tag: the attempts it would make (an accepted tag hides the '<' inside it; a failed one hands over to
  // the next '<'; failed ones that lost white space behind their '<' leave a text piece), up to the first lane that gave up.  While
  // a periodic stretch is in hand a batch stops behind its first accepted tag, so that the replay's bookkeeping sees every one.
  // Batches are made where they pay: four '<' or more within reach, runs of failing attempts ("<a <b <c ...").
  // A batch whose first lane gave up has cost its time for nothing: the next ones are skipped, twice as many each time it happens.
  uint32_t fail_run = 0, lb_skip = 0, lb_penalty = 1;
  uint32_t lb_attr_cap = SG_LANE_ATTRS;                                    // attributes a lane of a tag batch walks before it gives up (adapts, below)
  uint32_t ab_skip = 0, ab_penalty = 1;                                    // attribute batches (in the attempt below) back off the same way
  auto lane_batches = [&]() __attribute__((always_inline)) -> int {
    EH_CTX;                                                                // (not the captured reference: a lambda that is not inlined would carry a generic pointer to the LDS context)
    while (lanes_on && ei < nev) {
      const bool single = rp_P != 0 && (rp_state != 0 || pos >= rp_lo);
      if (single && fail_run < 3) break;
      if (lb_skip > 0) { lb_skip--; break; }
      if (cbase != 0xFFFFFFFFu && cbase + SG_EVC < nev && ei + 2048u > cbase + SG_EVC) { cbase = 0xFFFFFFFFu; bbase = 0xFFFFFFFFu; }   // little of the window left: move it
      need(ei);
      const uint32_t wend = cbase + SG_EVC < nev ? cbase + SG_EVC : nev;
      const uint32_t send = ei + 1024u < wend ? ei + 1024u : wend;         // '<' further ahead than this are the text state's to find
      const uint32_t bound = rp_P && !single ? rp_lo : 0xFFFFFFFFu;        // before a stretch: '<' at or beyond it are left to the batches made there
      // (the list of '<' goes into the 64 words behind the window, g_fuse_lds[SG_EVC ..): indexed, not through a pointer - a generic
      // pointer into LDS here sent this compiler into "Illegal instruction detected: V_CMP_NE_U32 0, $src_shared_base")
      uint32_t nl = 0;
      lanes_sync();
      for (uint32_t base = ei; base < send && nl < 64; base += 64) {
        const uint32_t idx = base + (uint32_t)l;
        const uint32_t v = idx < send ? g_fuse_lds[idx - cbase] : 0u;
        const bool is = idx < send && (v & 15u) == E_LT && (v >> 4) < bound;
        const unsigned long long m = __ballot(is);
        const uint32_t rank = nl + (uint32_t)__popcll(m & ((1ull << l) - 1ull));
        if (is && rank < 64) g_fuse_lds[SG_EVC + rank] = idx;
        nl += (uint32_t)__popcll(m);
      }
      if (nl > 64) nl = 64;
      lanes_sync();
      if (nl == 0 || (nl < 4 && fail_run < 3)) break;                       // not worth a batch: tag by tag
      const bool mine = (uint32_t)l < nl;
      const uint32_t e = mine ? g_fuse_lds[SG_EVC + (uint32_t)l] : 0xFFFFFFFFu;
      if (!nstop_tab && !build_nstop()) return -3;
      SgLaneMemo mm; mm.gt = ff_gt; mm.qgt = ff_qgt; mm.cmt = ff_cmt; mm.sq = ff_sq; mm.dq = ff_dq; mm.bad = bad; mm.nstop = nstop_tab;
      SgLaneTag R; R.ok = 0; R.bail = 1; R.kind = 0; R.next = 0; R.nexte = 0; R.tag0 = 0; R.lt = 0; R.npc = 0; R.npar = 0; R.na = 0; R.nb = 0; R.reach = 0; R.ffc = 0; R.fff = 0; R.unmarked = 0; R.capped = 0;
      const uint32_t acap = lb_attr_cap;                                   // (the three passes of one batch walk with the same cap)
      if (mine) R = sg_lane_attempt<0>(H, L, ev, nev, cbase, e, mm, nullptr, nullptr, nullptr, acap);
      // what the attempts found out about the block: searches that ran to its end, attribute loops that fail
      if (__ballot(R.ffc != 0)) {
        const uint32_t m1 = wave_min(R.ffc == E_GT ? R.fff : nev), m2 = wave_min(R.ffc == E_QGT ? R.fff : nev), m3 = wave_min(R.ffc == E_CMTEND ? R.fff : nev);
        const uint32_t m4 = wave_min(R.ffc == E_SQ ? R.fff : nev), m5 = wave_min(R.ffc == E_DQ ? R.fff : nev);
        if (m1 < ff_gt) ff_gt = m1;
        if (m2 < ff_qgt) ff_qgt = m2;
        if (m3 < ff_cmt) ff_cmt = m3;
        if (m4 < ff_sq) ff_sq = m4;
        if (m5 < ff_dq) ff_dq = m5;
      }
      if (__ballot(R.unmarked != 0)) {
        if (!bad) {
          bad = ws_alloc(c, (uint64_t)L + 16);
          if (!bad) return -3;
          for (uint32_t i = 16u * (uint32_t)l; i < L + 16; i += 1024) { uint4 z = {0, 0, 0, 0}; stg16(bad + i, z); }
          wave_sync();
        }
        if (R.unmarked != 0) (void)sg_lane_attempt<2>(H, L, ev, nev, cbase, e, mm, nullptr, nullptr, bad, acap);
        wave_sync();
      }
      // which attempts does the machine make?
      unsigned long long visited = 0; uint32_t k = 0; bool any_acc = false;
      while (k < nl) {
        if ((uint32_t)__builtin_amdgcn_readlane((int)R.bail, (int)k)) break;
        visited |= 1ull << k;
        if (!(uint32_t)__builtin_amdgcn_readlane((int)R.ok, (int)k)) { k++; continue; }
        any_acc = true;
        if (single) break;
        const uint32_t ne = (uint32_t)__builtin_amdgcn_readlane((int)R.nexte, (int)k);
        const unsigned long long m = __ballot(mine && e >= ne);
        if (!m) break;
        k = (uint32_t)__builtin_ctzll(m);
      }
      // Lanes that walk a long tag to the cap before they give up make the whole batch wait for nothing - and where tags hold '<'
      // themselves most lanes' work is thrown away anyway: few tags committed with lanes at the cap -> the cap halves (long tags
      // go to the wave-wide machine sooner, whose attribute batches are cheap); many committed -> it grows back.
      {
        const uint32_t ncom = (uint32_t)__popcll(visited);
        const bool hitcap = __ballot(mine && R.capped != 0) != 0;
        if (hitcap && ncom < 8 && lb_attr_cap > SG_LANE_ATTRS_MIN) lb_attr_cap = lb_attr_cap / 2 < SG_LANE_ATTRS_MIN ? SG_LANE_ATTRS_MIN : lb_attr_cap / 2;
        else if (ncom >= 24 && lb_attr_cap < SG_LANE_ATTRS) lb_attr_cap = lb_attr_cap * 2 > SG_LANE_ATTRS ? SG_LANE_ATTRS : lb_attr_cap * 2;
      }
      if (!visited) { lb_skip = lb_penalty; if (lb_penalty < 1024) lb_penalty *= 2; break; }   // the first lane gave up: the wave-wide machine takes this '<'
      if (__popcll(visited) >= 4) lb_penalty = 1;
      const uint32_t lastc = 63u - (uint32_t)__builtin_clzll(visited);
      const bool cm = (visited >> l) & 1ull;
      const bool acc = cm && R.ok, fw = cm && !R.ok && R.tag0 > R.lt + 1u;   // accepted; failed with white space eaten behind its '<' (:80,:92)
      const uint32_t pcs = acc ? 1u + R.npc : (fw ? 1u : 0u), tks = acc ? 2u : 0u, prs = acc ? R.npar : 0u;
      const uint32_t ipc = wave_incl_scan(pcs), itk = wave_incl_scan(tks), ipr = wave_incl_scan(prs);
      const uint32_t tpc = (uint32_t)__builtin_amdgcn_readlane((int)ipc, 63), ttk = (uint32_t)__builtin_amdgcn_readlane((int)itk, 63), tpr = (uint32_t)__builtin_amdgcn_readlane((int)ipr, 63);
      if ((uint64_t)npc + tpc + 16 > cap_pc || (uint64_t)ntok + ttk + 2 > cap_tok || (uint64_t)npar + tpr + 1 > cap_par) { EH_SET_OVERFLOW(c, 604); return -3; }
      const uint32_t bpc = npc + ipc - pcs, btk = ntok + itk - tks, bpr = npar + ipr - prs;
      // where the text in front of this '<' begins: behind the last accepted tag, or behind the white space the last failed '<' lost
      const unsigned long long defm = __ballot(acc || fw), accm = __ballot(acc), fwm = __ballot(fw);
      const unsigned long long below = (1ull << l) - 1ull;
      const uint32_t defv = acc ? R.next : R.tag0;
      const uint32_t pd = (defm & below) ? 63u - (uint32_t)__builtin_clzll(defm & below) : 0u;
      const uint32_t pdv = (uint32_t)__shfl((int)defv, (int)pd);
      const uint32_t segb = (defm & below) ? pdv : seg_start;
      const uint32_t pa = (accm & below) ? 63u - (uint32_t)__builtin_clzll(accm & below) : 0u;
      const uint32_t pav = (uint32_t)__shfl((int)(bpc + pcs), (int)pa);
      const uint32_t tp0 = (accm & below) ? pav : text_p0;                 // first piece of the text token in front of this tag
      if (fw) { Piece q; q.ptr = (uint64_t)(H + segb); q.len = R.lt + 1u - segb; q.rep = 1; pc[bpc] = q; }
      if (acc) {
        Piece q; q.ptr = (uint64_t)(H + segb); q.len = R.lt - segb; q.rep = 1; pc[bpc] = q;
        const bool empty = bpc == tp0 && R.lt == segb;                     // (every earlier piece of the token holds a '<')
        SgTok t; t.kind = TK_TEXT | (empty ? (uint32_t)TF_EMPTY : 0u); t.p0 = tp0; t.np = bpc + 1u - tp0; t.na = 0; t.nb = 0; t.par0 = 0; t.npar = 0; t.match = -1;
        tok[btk] = t;
        SgLaneTag W = sg_lane_attempt<1>(H, L, ev, nev, cbase, e, mm, pc + bpc + 1u, par + bpr, nullptr, acap);
        SgTok g; g.kind = W.kind; g.p0 = bpc + 1u; g.np = W.npc; g.na = W.na; g.nb = W.nb; g.par0 = bpr; g.npar = W.npar; g.match = -1;
        tok[btk + 1u] = g;
      }
      if (rp_state != 0) { const uint32_t r = wave_max(cm ? R.reach : 0u); if (r > reach_ev) reach_ev = r; }
      // the state behind the last committed attempt
      const bool last_is_acc = (accm >> lastc) & 1ull;
      if (last_is_acc) {
        pos = (uint32_t)__builtin_amdgcn_readlane((int)R.next, (int)lastc); ei = (uint32_t)__builtin_amdgcn_readlane((int)R.nexte, (int)lastc);
        seg_start = pos; text_p0 = npc + tpc; text_len = 0;
        fail_run = 0;
      } else {
        const unsigned long long trail = accm ? fwm & ~((2ull << (63u - (uint32_t)__builtin_clzll(accm))) - 1ull) : fwm;   // lost white space behind the last accepted tag
        const uint32_t add = wave_sum(((trail >> l) & 1ull) ? R.lt + 1u - segb : 0u);
        if (accm) {
          const uint32_t la = 63u - (uint32_t)__builtin_clzll(accm);
          text_p0 = (uint32_t)__shfl((int)(bpc + pcs), (int)la); text_len = 0; seg_start = (uint32_t)__builtin_amdgcn_readlane((int)R.next, (int)la);
        }
        text_len += add;
        if (trail) seg_start = (uint32_t)__builtin_amdgcn_readlane((int)R.tag0, (int)(63u - (uint32_t)__builtin_clzll(trail)));
        pos = (uint32_t)__builtin_amdgcn_readlane((int)R.lt, (int)lastc) + 1u; ei = (uint32_t)__builtin_amdgcn_readlane((int)e, (int)lastc) + 1u;
        fail_run += (uint32_t)__popcll(visited & ~accm & (accm ? ~((2ull << (63u - (uint32_t)__builtin_clzll(accm))) - 1ull) : ~0ull));
      }
      ntok += ttk; npc += tpc; npar += tpr;
      bbase = 0xFFFFFFFFu;
      wave_sync();
#ifdef EH_PROF
      if (l == 0) { atomicAdd(&c.p->prof[2 * 85], (unsigned long long)(ttk / 2u)); atomicAdd(&c.p->prof[2 * 85 + 1], 1ull); }   // tags written by lane batches, batches
      if (l == 0) { atomicAdd(&c.p->prof[2 * 68], (unsigned long long)__popcll(visited & ~accm)); atomicAdd(&c.p->prof[2 * 68 + 1], 1ull); }   // failed attempts committed by batches
#endif
      if (single && any_acc) { const int r = after_accept(); if (r) return r; }
    }
    return 0;
  };
  // tz(nil, ..) :100-101: bytes before the first '<' are dropped
  {
    uint32_t j = find(0, 1u << E_LT);
    lt = evget(j) >> 4; pos = lt + 1; ei = j + 1;
  }
  SG_SUB(112);                                                             // set-up: tables, the first stretch search, the first '<'
  for (;;) {
    if (npc + 16 > cap_pc || ntok + 2 > cap_tok) { EH_SET_OVERFLOW(c, 601); return -3; }
    SG_TZ(86);                                                             // eh_result_prof 86: text state + bookkeeping between attempts
    // ---- one tag attempt: '<' at lt, pos/ei just behind it
    skipws();
    const uint32_t tag0 = pos, ei0 = ei, tag_p0 = npc, par0 = npar;
    const bool tight = tag0 == lt + 1;                                     // no white space behind the '<': "<!", "</" ... are in the input as such
    uint32_t kind = 0, na = pos, nb = pos, next = 0, nexte = 0;
    bool ok = false, other = false;
    nchain = 0;
    do {
      if (pos >= L) break;
      uint32_t c0 = cls_here();
      if (c0 == E_BANG) {                                                  // :104-105
        bool d1 = false, d2 = false;
        if (ei + 2 < nev + 0u) { uint32_t e1 = evget(ei + 1), e2 = evget(ei + 2); d1 = (e1 >> 4) == pos + 1 && ((ES_DASH >> (e1 & 15u)) & 1u); d2 = (e2 >> 4) == pos + 2 && ((ES_DASH >> (e2 & 15u)) & 1u); }
        if (d1 && d2) {                                                    // {'!--',DT} :117-118
          uint32_t dta = pos + 3;
          uint32_t j = find1(ei + 3, E_CMTEND, ff_cmt);
          if (j >= nev) { other = true; break; }                           // no clause of tz/2 matches {'!--',_}, <<>>
          uint32_t e = evget(j) >> 4;
          put(tight ? H + lt : sglit(SL_CMT), 4); put(H + dta, e - dta); put(H + e, 3);
          kind = TK_COMMENT; next = e + 3; nexte = j + 3; ok = true; break;
        }
        step1(); skipws();                                                 // {'!',DT} :113-115
        uint32_t dta = pos;
        uint32_t j = find1(ei, E_GT, ff_gt);
        if (j >= nev) break;
        uint32_t e = evget(j) >> 4;
        put(tight ? H + lt : sglit(SL_LTBANG), 2); put(H + dta, e - dta); put(H + e, 1);
        kind = TK_BANG; next = e + 1; nexte = j + 1; ok = true; break;
      }
      if (c0 == E_QM || c0 == E_QGT) {                                     // {que,DT} :106,:120-122
        step1(); skipws();
        uint32_t dta = pos;
        uint32_t j = find1(ei, E_QGT, ff_qgt);
        if (j >= nev) break;
        uint32_t e = evget(j) >> 4;
        put(tight ? H + lt : sglit(SL_LTQ), 2); put(H + dta, e - dta); put(H + e, 2);
        kind = TK_QUE; next = e + 2; nexte = j + 2; ok = true; break;
      }
      if (c0 == E_SL || c0 == E_SLGT) {                                    // {end_tag,Tag} :107,:128-132
        step1(); skipws();
        na = pos;
        uint32_t j = find_stop(ei, ES_EV);
        if (j == 0xFFFFFFFFu) return -3;
        if (j >= nev) break;
        nb = evget(j) >> 4; pos = nb; ei = j;
        skipws();
        if (cls_here() != E_GT) break;
        put(tight ? H + lt : sglit(SL_LTSL), 2); put(H + na, nb - na); put(H + pos, 1);
        kind = TK_CLOSE; next = pos + 1; nexte = ei + 1; ok = true; break;
      }
      // {tag,Tag} :108-111
      uint32_t j = find_stop(ei, ES_STOP);
      if (j == 0xFFFFFFFFu) return -3;
      if (j >= nev) break;
      uint32_t ej = evget(j);
      nb = ej >> 4;
      put(H + lt, 1); put(H + na, nb - na); put(sglit(SL_SP), 0);           // "<", name, slot for extra params
      if ((ej & 15u) == E_SLGT) { put(sglit(SL_SPSLGT), 3); kind = TK_SC; next = nb + 2; nexte = j + 2; ok = true; break; }
      pos = nb; ei = j;
      skipws();
      // attribute loop: {attr,..} {eatt,..} {val,..} {sqval|dqval|uqval,..} :134-160
      uint32_t nattr = 0;
      for (;;) {
        if (npc + 16 > cap_pc || npar + 1 > cap_par) { EH_SET_OVERFLOW(c, 602); return -3; }
        if (pos >= L) break;
        // ---- a long tag: the next iterations one per lane (sg_lane_attr, above).  The chain walk commits exactly the iterations this
        // loop would make, in its order; what is not a plain attribute stays for the statements below.
        // (below the 16th attribute the memo of failed entries is neither asked nor fed: a batch there takes at most 16 - nattr iterations)
        if (nattr >= 2 && lanes_on && (uint64_t)npc + 6u * 64u + 16u <= cap_pc && (uint64_t)npar + 65u <= cap_par) {
          if (ab_skip > 0) ab_skip--;
          else {
            if (!nstop_tab && !build_nstop()) return -3;
            const bool memo = nattr >= 16;
            const uint32_t maxc = memo ? 64u : 16u - nattr;
            if (memo && !chain) { chain = (wptr)ws_alloc(c, ((uint64_t)nstop + 8) * 4); if (!chain) return -3; }
            if (cbase != 0xFFFFFFFFu && cbase + SG_EVC < nev && ei + 1024u > cbase + SG_EVC) { cbase = 0xFFFFFFFFu; bbase = 0xFFFFFFFFu; }   // little of the window left: move it
            need(ei);
            const uint32_t wend = cbase + SG_EVC < nev ? cbase + SG_EVC : nev;
            const uint32_t lim = wend < nev ? wend - 1u : wend;            // (a candidate's event is looked at together with the one behind it)
            const uint32_t send = ei + 1024u < lim ? ei + 1024u : lim;
            uint32_t nc = 1;                                               // lane 0: where the machine is
            lanes_sync();
            for (uint32_t base = ei; base < send && nc < maxc; base += 64) {
              const uint32_t idx = base + (uint32_t)l;
              const bool in = idx < send;
              const uint32_t v = in ? g_fuse_lds[idx - cbase] : 0u;
              const uint32_t vn = (in && idx + 1u < nev) ? g_fuse_lds[idx + 1u - cbase] : 0u;
              const uint32_t cl = v & 15u;
              const bool adj = in && idx + 1u < nev && (vn >> 4) == (v >> 4) + 1u && ((ES_WS >> (vn & 15u)) & 1u);
              const bool is = in && (((ES_WS >> cl) & 1u) || cl == E_SQ || cl == E_DQ) && !adj;   // the end of a run of white space, or a quote with none behind it
              const unsigned long long m = __ballot(is);
              const uint32_t rank = nc + (uint32_t)__popcll(m & ((1ull << l) - 1ull));
              if (is && rank < 64) g_fuse_lds[SG_EVC + rank] = idx;
              nc += (uint32_t)__popcll(m);
            }
            if (nc > maxc) nc = maxc;
            lanes_sync();
            const bool mine = (uint32_t)l < nc;
            const uint32_t cj = (l > 0 && mine) ? g_fuse_lds[SG_EVC + (uint32_t)l] : cbase;
            const uint32_t cv = g_fuse_lds[cj - cbase];
            const uint32_t cpos = l == 0 ? pos : (cv >> 4) + 1u, cei = l == 0 ? ei : cj + 1u, cws = l == 0 ? (ws_sp ? 1u : 0u) : ((cv & 15u) == E_SPACE ? 1u : 0u);
            SgLaneMemo mm; mm.gt = ff_gt; mm.qgt = ff_qgt; mm.cmt = ff_cmt; mm.sq = ff_sq; mm.dq = ff_dq; mm.bad = bad; mm.nstop = nstop_tab;
            SgLaneAttr A; A.term = 1; A.npos = 0; A.nei = 0; A.nws = 0; A.an = 0; A.ae = 0; A.va = 0; A.vb = 0; A.delim = 0; A.eqpos = 0; A.reach = 0;
            if (mine) A = sg_lane_attr(H, L, ev, nev, cbase, cpos, cei, cws, mm);
            const uint32_t isbad = (memo && mine && bad && cpos < L) ? (uint32_t)bad[cpos] : 0u;
            // the chain: lane k's iteration is followed by the lane whose candidate is where it ended.  Mostly that is the next lane
            // (a word, white space, the next word), so runs of such lanes are taken with bit operations, the rest by a search.
            const uint32_t cpos_nx = (uint32_t)__shfl((int)cpos, (l + 1) & 63);
            const unsigned long long termm = __ballot(!mine || A.term != 0 || isbad != 0);   // isbad: the memo of failed attribute-loop entries (nattr >= 16 here)
            const unsigned long long linkm = __ballot(mine && (uint32_t)l + 1u < nc && A.npos == cpos_nx);
            const unsigned long long chainable = linkm & ~termm & (~termm >> 1);              // lane k and its successor k + 1 both make a plain iteration
            unsigned long long visited = 0; uint32_t k = 0, cnt = 0;
            for (;;) {
              if ((termm >> k) & 1ull) break;
              const uint32_t r = (uint32_t)__builtin_ctzll(~(chainable >> k));                // (bit 63 is never chainable: the shift brings in zeros)
              visited |= ((2ull << r) - 1ull) << k; cnt += r + 1u; k += r;
              if ((linkm >> k) & 1ull) break;                                                  // ends at lane k + 1's candidate, which is not a plain iteration
              const uint32_t np = (uint32_t)__builtin_amdgcn_readlane((int)A.npos, (int)k);
              const unsigned long long m = __ballot(mine && (uint32_t)l > k && cpos == np);
              if (!m) break;
              k = (uint32_t)__builtin_ctzll(m);
            }
            if (cnt == 0) { ab_skip = ab_penalty; if (ab_penalty < 64) ab_penalty *= 2; }
            else {
              const bool cm = (visited >> l) & 1ull;
              const uint32_t rank = (uint32_t)__popcll(visited & ((1ull << l) - 1ull));
              if (cm) {
                SgParam q; q.na = A.an; q.nb = A.ae; q.va = A.va; q.vb = A.vb; q.delim = A.delim; q.pad = 0; par[npar + rank] = q;
                const bool has = A.vb > A.va, hd = has && A.delim != 0;
                cbptr qp = sglit(A.delim == 1 ? SL_SQ : SL_DQ);
                EH_G Piece* o = pc + npc + 6u * rank; Piece x; x.rep = 1;
                x.ptr = (uint64_t)(cws ? H + A.an - 1 : sglit(SL_SP)); x.len = 1; o[0] = x;
                x.ptr = (uint64_t)(H + A.an); x.len = A.ae - A.an; o[1] = x;
                x.ptr = (uint64_t)(has ? H + A.eqpos : sglit(SL_EQ)); x.len = has ? 1u : 0u; o[2] = x;
                x.ptr = (uint64_t)(hd ? H + A.va - 1 : qp); x.len = hd ? 1u : 0u; o[3] = x;
                x.ptr = (uint64_t)(H + A.va); x.len = has ? A.vb - A.va : 0u; o[4] = x;
                x.ptr = (uint64_t)(hd ? H + A.vb : qp); x.len = hd ? 1u : 0u; o[5] = x;
                if (memo) chain[nchain + rank] = cpos;
              }
              const uint32_t lastc = 63u - (uint32_t)__builtin_clzll(visited);
              pos = (uint32_t)__builtin_amdgcn_readlane((int)A.npos, (int)lastc); ei = (uint32_t)__builtin_amdgcn_readlane((int)A.nei, (int)lastc);
              ws_sp = (uint32_t)__builtin_amdgcn_readlane((int)A.nws, (int)lastc) != 0;
              npar += cnt; npc += 6u * cnt; nattr += cnt; if (memo) nchain += cnt;
              if (rp_state != 0) { const uint32_t r = wave_max(cm ? A.reach : 0u); if (r > reach_ev) reach_ev = r; }
              bbase = 0xFFFFFFFFu;
              wave_sync();
              ab_penalty = 1;
#ifdef EH_PROF
              if (l == 0) { atomicAdd(&c.p->prof[2 * 120], (unsigned long long)cnt); atomicAdd(&c.p->prof[2 * 120 + 1], 1ull); }   // attributes written by attribute batches, batches
#endif
              continue;
            }
          }
        }
        if (nattr >= 16) {                                                 // quadratic-rescan guard
          if (bad && uni(bad[pos])) break;
          if (!chain) { chain = (wptr)ws_alloc(c, ((uint64_t)nstop + 8) * 4); if (!chain) return -3; }
          if (l == 0) chain[nchain] = pos;
          nchain++;
        }
        uint32_t ca = cls_here();
        if (ca == E_SLGT) { put(ws_sp ? H + pos - 1 : sglit(SL_SPSLGT), 3); kind = TK_SC; next = pos + 2; nexte = ei + 2; ok = true; break; }   // {etag,..} :124-125
        if (ca == E_GT) { put(H + pos, 1); kind = TK_OPEN; next = pos + 1; nexte = ei + 1; ok = true; break; }
        const bool sp_before = ws_sp;                                      // the blank in front of the name is in the input
        if (ca == E_EQ) break;                                             // tz({etag,..}, _) throws
        uint32_t an = pos;
        step1();
        uint32_t ja = find_stop(ei, ES_STOP);
        if (ja == 0xFFFFFFFFu) return -3;
        if (ja >= nev) break;
        uint32_t ae = evget(ja) >> 4; pos = ae; ei = ja;
        skipws();
        uint32_t va = pos, vb = pos, delim = 0, eqpos = 0xFFFFFFFFu;
        if (pos < L && cls_here() == E_EQ) {                               // {eatt,..} "=" -> {val,..} :141,:144-146
          eqpos = pos;
          step1(); skipws();
          if (pos >= L) break;
          uint32_t cv = cls_here();
          if (cv == E_SQ || cv == E_DQ) {
            uint32_t jq = cv == E_SQ ? find1(ei + 1, E_SQ, ff_sq) : find1(ei + 1, E_DQ, ff_dq);
            if (jq >= nev) break;                                          // unterminated quote
            va = pos + 1; vb = evget(jq) >> 4; delim = cv == E_SQ ? 1u : 2u;
            pos = vb + 1; ei = jq + 1;
          } else {
            uint32_t ju = find_stop(ei, ES_STOP);
            if (ju == 0xFFFFFFFFu) return -3;
            if (ju >= nev) break;
            va = pos; vb = evget(ju) >> 4; pos = vb; ei = ju;
          }
          skipws();
        }
        // As ++ [{A, V, Delim}]
        if (l == 0) { SgParam q; q.na = an; q.nb = ae; q.va = va; q.vb = vb; q.delim = delim; q.pad = 0; par[npar] = q; }
        npar++; nattr++;
        bool has = vb > va;
        cbptr qp = sglit(delim == 1 ? SL_SQ : SL_DQ);
        put(sp_before ? H + an - 1 : sglit(SL_SP), 1); put(H + an, ae - an);
        put(has ? H + eqpos : sglit(SL_EQ), has ? 1u : 0u); put(has && delim ? H + va - 1 : qp, has && delim ? 1u : 0u);
        put(H + va, has ? vb - va : 0u); put(has && delim ? H + vb : qp, has && delim ? 1u : 0u);
      }
    } while (false);
    if (c.status != CASE_OK) return -3;
    if (ok) SG_TZ(88); else SG_TZ(87);                                     // 88: accepted tags, 87: failed attempts
    SG_CNT(118, npar - par0);                                              // attributes the attempt walked
    SG_SUB0();
    if (ok) {
      // the text before the tag (if any) and the tag itself
      if (!first) {
        if (l == 0) { Piece q; q.ptr = (uint64_t)(H + seg_start); q.len = lt - seg_start; q.rep = 1; pc[seg_slot] = q; }
        text_len += lt - seg_start;
        if (l == 0) { SgTok t; t.kind = TK_TEXT | (text_len == 0 ? (uint32_t)TF_EMPTY : 0u); t.p0 = text_p0; t.np = seg_slot + 1 - text_p0; t.na = 0; t.nb = 0; t.par0 = 0; t.npar = 0; t.match = -1; tok[ntok] = t; }
        ntok++;
      }
      if (l == 0) { SgTok t; t.kind = kind; t.p0 = tag_p0; t.np = npc - tag_p0; t.na = na; t.nb = nb; t.par0 = par0; t.npar = npar - par0; t.match = -1; tok[ntok] = t; }
      ntok++;
      first = false;
      pos = next; ei = nexte; seg_start = next; text_p0 = npc; text_len = 0;
      fail_run = 0;
      { const int r = after_accept(); if (r) return r; }
      SG_SUB(113);
    } else {
      if (first) { rc = other ? -2 : -1; break; }
      if (nchain > 0) {                                                    // remember where this attempt entered the attribute loop
        if (!bad) {
          bad = ws_alloc(c, (uint64_t)L + 16);
          if (!bad) return -3;
          for (uint32_t i = 16u * (uint32_t)l; i < L + 16; i += 1024) { uint4 z = {0, 0, 0, 0}; stg16(bad + i, z); }
        }
        wave_sync();
        for (uint32_t i = l; i < nchain; i += 64) bad[chain[i]] = 1;
        wave_sync();
      }
      npc = seg_slot; npar = par0;                                         // forget the pieces of the attempt
      if (tag0 > lt + 1) {                                                 // ws/1 ate white space after the '<': it is gone from the text (:80,:92)
        put(H + seg_start, lt + 1 - seg_start);
        text_len += lt + 1 - seg_start; seg_start = tag0;
      }
      pos = tag0; ei = ei0;                                                // ff/4 goes on from EStr
      fail_run++;
    }
    SG_SUB0();
    { const int r = lane_batches(); if (r) return r; }
    SG_SUB(114);
    // ---- text state: ff/4 :166-174
    uint32_t j = find(ei, 1u << E_LT);
    SG_SUB(115);
    if (j >= nev) {                                                        // {{text,Str},"",eof} :82-83,:94-95
      uint32_t n = L - seg_start;
      put(H + seg_start, n); text_len += n;
      if (l == 0) { SgTok t; t.kind = TK_TEXT | (text_len == 0 ? (uint32_t)TF_EMPTY : 0u); t.p0 = text_p0; t.np = npc - text_p0; t.na = 0; t.nb = 0; t.par0 = 0; t.npar = 0; t.match = -1; tok[ntok] = t; }
      ntok++;
      break;
    }
    lt = evget(j) >> 4; seg_slot = npc; npc++;                             // slot for the text segment that ends here
    pos = lt + 1; ei = j + 1;
  }
  wave_sync();
  if (rc != 0) return rc;
  if (l == 0) { out->tok = tok; out->par = par; out->pc = pc; out->ntok = ntok; out->npar = npar; out->npc = npc; }
  wave_sync();
  return 0;
}

// string:to_lower/1 (ISO 8859-1 rule of the old string module)
EH_DEV uint32_t latin1_lower(uint32_t ch) { return ((ch >= 'A' && ch <= 'Z') || (ch >= 0xC0 && ch <= 0xD6) || (ch >= 0xD8 && ch <= 0xDE)) ? ch + 32 : ch; }
EH_DEV uint32_t sg_name_hash(cbptr H, uint32_t a, uint32_t b) {
  uint32_t h = 0;
  for (uint32_t i = a + (uint32_t)EH_LANE; i < b; i += 64) {
    uint32_t v = (latin1_lower(H[i]) + 1u) * (0x9E3779B1u * ((i - a) + 1u) | 1u);
    uint32_t r = (i - a) & 31u;
    h += (v << r) | (r ? v >> (32 - r) : 0u);
  }
  return wave_sum(h) ^ ((b - a) * 0x85EBCA6Bu);
}
EH_DEV bool sg_name_eq(cbptr H, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2) {
  if (b1 - a1 != b2 - a2) return false;
  uint32_t n = b1 - a1; bool ne = false;
  for (uint32_t i = EH_LANE; i < n; i += 64) ne |= latin1_lower(H[a1 + i]) != latin1_lower(H[a2 + i]);
  return __ballot(ne) == 0;
}

// build_ast2/4 :204-279 as a stack match over the token table: sets .match on paired open / close tokens.
// stk = scratch for {token index, name hash} pairs.
EH_DEV void sgml_pair(cbptr H, EH_G SgTok* tok, uint32_t ntok, wptr stk) {
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
EH_DEV void sgml_flags(const EH_G SgTok* tok, uint32_t ntok, bptr ef, uint32_t* N, uint32_t* NT) {
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
EH_DEV uint32_t sg_find(cbptr ef, uint32_t ntok, uint32_t bit, uint32_t k) {
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
EH_DEV uint32_t sg_count(cbptr ef, uint32_t a, uint32_t b, uint32_t bit) {
  uint32_t n = 0;
  for (uint32_t i = a + (uint32_t)EH_LANE; i <= b; i += 64) n += (ef[i] & bit) ? 1u : 0u;
  return wave_sum(n);
}

struct SgRange { uint32_t s, e, p0, p1; bool tag; };                       // tokens [s, e], pieces [p0, p1)
EH_DEV SgRange sg_range_of(const EH_G SgTok* tok, uint32_t s) {
  SgTok t = tok[s];
  SgRange r; r.s = s;
  uint32_t k = uni(t.kind);
  r.tag = (k & TK_KIND) == TK_OPEN && (k & TF_PAIRED);
  r.e = r.tag ? (uint32_t)uni((uint32_t)t.match) : s;
  r.p0 = uni(t.p0);
  if (r.e == s) r.p1 = r.p0 + uni(t.np); else { SgTok z = tok[r.e]; r.p1 = uni(z.p0) + uni(z.np); }
  return r;
}

// the end of sgml_mutate/2: NewBinStr = fold_ast(Res, []) :746 and its comparison with the block
EH_DEV int sgml_finish(Ctx& c, cbptr H, uint32_t L, EH_G Piece* out, uint32_t nout, uint32_t cap_out, int D, uint32_t meta0) {
  if (c.status != CASE_OK) return 0;
  if (nout > cap_out) { EH_SET_OVERFLOW(c, 603); return 0; }
  wave_sync();
  EH_PT0;
  nout = pieces_coalesce(out, nout);
  uint64_t total = pieces_total(out, nout);
  if (total > 0xFFFFFFF0ull) { EH_SET_OVERFLOW(c, 604); return 0; }
  bptr dst = ws_alloc(c, total ? total : 16);
  if (!dst) return 0;
  wave_gather(dst, out, nout);
  wave_sync();
  EH_PT(c, 93);
  if ((uint32_t)total == L && wave_equal(dst, H, L)) { tr_drop_before(c, meta0); return -1; }   // NewBinStr =:= H: {fun sgml_mutate/2, Ll, NewMeta, -1} :748-749 - NewMeta ALONE
  c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)total; c.r_changed = 1;
  return D + (int)(total / (AVG_BLOCK_SIZE * 10));
}

// inner text :727-737 (walk2acc/3 :363-379).  mutate_innertext/3 :674-681 runs a nested scheduler call on a text or on a
// parameter's value: the walk leaves its place (token, parameter) in the MuFrame, asks the scheduler for the run (eh_device.h
// mux_fuzzers) and goes on from there when it is called again (c.mu_phase 1).  v[0..8]: what sgml_mutate had in hand when it came
// here (tokens, parameters, the piece list, counts, D, where its Meta began), v[9..13]: the place.
__device__ __noinline__ int sgml_inner(Ctx&, cbptr H, uint32_t L) {
  EH_CTX;
  const int l = EH_LANE;
  EH_G MuFrame* mu = c.mu;
  bool resume = c.mu_phase == 1;
  EH_G SgTok* tok = (EH_G SgTok*)uni64(mu->v[0]); EH_G SgParam* par = (EH_G SgParam*)uni64(mu->v[1]); EH_G Piece* out = (EH_G Piece*)uni64(mu->v[2]);
  const uint32_t ntok = uni((uint32_t)mu->v[3]), nout = uni((uint32_t)mu->v[4]), cap_out = uni((uint32_t)mu->v[5]), NT = uni((uint32_t)mu->v[6]), meta0 = uni((uint32_t)mu->v[8]);
  const int D = (int)uni((uint32_t)mu->v[7]), nfs = (int)uni((uint32_t)mu->v[11]);
  uint32_t t = uni((uint32_t)mu->v[9]), i = uni((uint32_t)mu->v[10]);
  // asks for Muta([Bin], []) on [vp, vp + vl) when mutate_innertext/3 draws it; true: the request is made, the caller returns
  auto wants = [&](cbptr vp, uint32_t vl, uint32_t nt2) -> bool {
    uint32_t nw = wave_count(vp, vl, IsInk());
    if (!(nw > 0 && nt2 > 0)) return false;
    double rnd = rng_uniform(c.rng);
    if (rnd > 3.0 / (double)nt2) return false;
    if (l == 0) { mu->v[9] = t; mu->v[10] = i; }
    wave_sync();
    c.call_req = 1; c.call_bin = (uint64_t)vp; c.call_len = vl; c.call_nfs = nfs;
    return true;
  };
  for (; t < ntok; t++, i = 0) {
    SgTok tk = tok[t];
    uint32_t k = uni(tk.kind), kk = k & TK_KIND;
    if (kk == TK_TEXT && !(k & TF_EMPTY)) {                                // try_mutate_innertext({text, Binary}, ..) :691-692
      uint32_t p0 = uni(tk.p0), np = uni(tk.np);
      if (!resume) {
        cbptr vp; uint32_t vl;
        if (np == 1) { Piece q = out[p0]; vp = (cbptr)uni64(q.ptr); vl = uni(q.len); }
        else { bptr m; if (!pieces_materialize(c, out, p0, p0 + np, &m, &vl)) return 0; vp = m; }
        if (wants(vp, vl, NT)) return 0;
        continue;
      }
      resume = false;
      const int nres = c.call_nres;
      if (nres < 0) return 0;
      if (nres == 0) { c.status = CASE_CRASHED; return 0; }                // hd([])
      Blk rb = blk_load(c.bl, c.nb);
      wave_sync();
      if (l == 0) { out[p0].ptr = rb.ptr; out[p0].len = rb.len; for (uint32_t z = 1; z < np; z++) out[p0 + z].len = 0; }
    } else if (kk == TK_CLOSE && (k & TF_PAIRED)) {                        // the tag's own params, after its children :683-690
      SgTok ot = tok[(uint32_t)uni((uint32_t)tk.match)];
      uint32_t p0 = uni(ot.p0), pa0 = uni(ot.par0), npa = uni(ot.npar);
      for (; i < npa; i++) {
        if (!resume) {
          SgParam q = par[pa0 + i];
          uint32_t qva = uni(q.va), qvb = uni(q.vb);
          if (wants(H + qva, qvb - qva, NT + npa)) return 0;
          continue;
        }
        resume = false;
        const int nres = c.call_nres;
        if (nres < 0) return 0;
        if (nres == 0) { c.status = CASE_CRASHED; return 0; }
        Blk rb = blk_load(c.bl, c.nb);
        uint32_t pi = p0 + SG_TAGHEAD + SG_PARPCS * i;
        wave_sync();
        if (l == 0) { out[pi + 4].ptr = rb.ptr; out[pi + 4].len = rb.len; if (rb.len == 0) { out[pi + 2].len = 0; out[pi + 3].len = 0; out[pi + 5].len = 0; } }
      }
    }
  }
  wave_sync();
  return sgml_finish(c, H, L, out, nout, cap_out, D, meta0);
}

// sgml_mutate/2 :739-757
__device__ __noinline__ int muta_sgml(Ctx&) {
  EH_CTX;
  const int l = EH_LANE;
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  c.r_kind = R_SAME;
  if (c.mu_phase == 1) return sgml_inner(c, H, L);                         // back from a nested scheduler call of the inner-text walk
  if (binarish(H, L)) return -1;                                           // parse/2 :198-199
  SgDoc* dh = (SgDoc*)ws_alloc(c, sizeof(SgDoc));
  if (!dh) return 0;
  EH_PT0;
  int rc = sgml_tokenize(c, H, L, dh);
  EH_PT(c, 90);
  if (rc == -1) return -1;                                                 // catch incorrect_sgml :755-756
  if (rc == -2) { c.status = CASE_CRASHED; return 0; }
  if (rc != 0) return 0;
  EH_G SgTok* tok = (EH_G SgTok*)uni64((uint64_t)dh->tok); EH_G SgParam* par = (EH_G SgParam*)uni64((uint64_t)dh->par); EH_G Piece* pc = (EH_G Piece*)uni64((uint64_t)dh->pc);
  const uint32_t ntok = uni(dh->ntok), npc = uni(dh->npc);
  wptr stk = (wptr)ws_alloc(c, (uint64_t)ntok * 8 + 16);
  bptr ef = ws_alloc(c, (uint64_t)ntok + 16);
  if (!stk || !ef) return 0;
  sgml_pair(H, tok, ntok, stk);
  uint32_t N, NT;
  sgml_flags(tok, ntok, ef, &N, &NT);
  EH_PT(c, 91);
  // output piece list: worst case every doc piece twice plus a few literals
  uint32_t cap_out = 2 * npc + 64;
  EH_G Piece* out = (EH_G Piece*)ws_alloc(c, (uint64_t)cap_out * sizeof(Piece));
  if (!out) return 0;
  uint32_t nout = 0;
  auto all = [&](uint32_t a, uint32_t b) { pieces_append(out, &nout, pc, a, b); };
  auto elem = [&](uint32_t R) -> SgRange { return sg_range_of(tok, sg_find(ef, ntok, 1, R - 1)); };   // select_elem/2 :435-443
  int D = 1;
  const uint32_t meta0 = c.ntrace;                                         // NewMeta = what sgml_mutation/2 adds from here on
  uint32_t r = rng_rand(c.rng, 12);                                        // sgml_mutation/2 :696-698
  if (r < 8) {                                                             // {[{sgml_swap, 1}], Res, 1} ... :700-723
    const int sa = r == 0 ? AT_sgml_swap : r == 1 ? AT_sgml_dup : r == 2 ? AT_sgml_pump : r == 3 ? AT_sgml_repeat : r == 4 ? AT_sgml_insert2 : r == 5 ? AT_sgml_permparams : r == 6 ? AT_sgml_breaktag : AT_sgml_insert;
    tr_ai(c, sa, 1);
  } else if (r > 8) tr_ai(c, AT_sgml_innertext, 1);                        // {[Meta, {sgml_innertext, 1}], Res, 1} :737: in front of what the walk adds
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
        bptr m; uint32_t ml;
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
        bptr ma, mb; uint32_t la, lb;
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
        EH_G Key2* keys = (EH_G Key2*)ws_alloc(c, (uint64_t)np2 * sizeof(Key2));
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
      wptr ch = (wptr)ws_alloc(c, (uint64_t)(st.e - st.s) * 8 + 16);
      if (!ch) return 0;
      for (uint32_t t = st.s + 1; t < st.e;) { SgRange k = sg_range_of(tok, t); if (l == 0) { ch[2 * nch] = k.p0; ch[2 * nch + 1] = k.p1; } nch++; t = k.e + 1; }
      wave_sync();
      for (uint32_t k = nch; k-- > 0;) all(uni(ch[2 * k]), uni(ch[2 * k + 1]));
      all(st.p1, npc);
      break;
    }
    case 8: {                                                              // sgml_xmlfeatures(Ast, NT, 1) :651-665
      if (NT == 0) { D = -1; all(0, npc); tr_ai(c, AT_sgml_xmlfeatures, -1); break; }   // sgml_xmlfeatures(Ast, _NT, _) :664-665
      bool xm_changed = false;
      all(0, npc);
      wave_sync();
      const DevConfig& cfg = c.p->cfg;
      // "http" ++ get_ssrf_uri() (erlamsa_mutations.erl:727-731)
      bptr uri = ws_alloc(c, 128);
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
        xm_changed = true;                                                 // (a tag that is picked always comes out different)
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
          cbptr nv = uri; uint32_t nl = ul;
          bool app = rng_erand(c.rng, 2) == 1;
          // `Params =:= NewParams` (:596): replacing a value that already is the URI changes nothing
          if (app || vl != ul || !wave_equal(H + qva, uri, ul)) any = true;
          if (app) {                                                       // Uri ++ " http" ++ get_ssrf_uri()
            bptr b = ws_alloc(c, (uint64_t)vl + 1 + ul);
            if (!b) return 0;
            wave_copy(b, H + qva, vl); if (l == 0) b[vl] = ' '; wave_copy(b + vl + 1, uri, ul);
            nv = b; nl = vl + 1 + ul;
          }
          wave_sync();
          if (l == 0) {
            cbptr qp = sglit(qd == 1 ? SL_SQ : SL_DQ);
            out[pi + 2].len = 1;
            out[pi + 3].ptr = (uint64_t)qp; out[pi + 3].len = qd ? 1 : 0;
            out[pi + 4].ptr = (uint64_t)nv; out[pi + 4].len = nl;
            out[pi + 5].ptr = (uint64_t)qp; out[pi + 5].len = qd ? 1 : 0;
          }
        }
        if (!any) {                                                        // Params =:= NewParams: three new params in front :598-602
          bptr b = ws_alloc(c, 3ull * ul + 64);
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
      tr_aa(c, AT_sgml_xmlfeatures, xm_changed ? AT_xmlns : AT_failed);     // Ast =:= NewAst -> failed :661-662
      wave_sync();
      break;
    }
    default: {                                                             // inner text :727-737 (walk2acc/3 :363-379): sgml_inner below
      all(0, npc);
      wave_sync();                                                         // the list is read back below (other lanes wrote it)
      EH_G MuFrame* mu = mu_frame(c);
      if (!mu) return 0;
      int nfs;
      inner_table(c, false, mu, &nfs);
      if (l == 0) {
        mu->v[0] = (uint64_t)tok; mu->v[1] = (uint64_t)par; mu->v[2] = (uint64_t)out; mu->v[3] = ntok; mu->v[4] = nout; mu->v[5] = cap_out; mu->v[6] = NT;
        mu->v[7] = (uint32_t)D; mu->v[8] = meta0; mu->v[9] = 0; mu->v[10] = 0; mu->v[11] = (uint32_t)nfs;
      }
      wave_sync();
      return sgml_inner(c, H, L);
    }
  }
  if (c.status != CASE_OK) return 0;
  EH_PT(c, 92);
  return sgml_finish(c, H, L, out, nout, cap_out, D, meta0);
}

}  // namespace eh
