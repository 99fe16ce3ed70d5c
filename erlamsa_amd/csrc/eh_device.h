// eh_device.h — device-side building blocks of the batch mutation engine (gfx950, wave64).
//
// Execution model: ONE WAVEFRONT PER CASE, workgroup = 1 wavefront (64 threads).
//  * All decision state (PRNG, block list, mutator scores) is wave-uniform; every lane
//    evaluates the same scalar program, so branches are uniform and the compiler keeps
//    integer state in SGPRs.  The FP64 part of AS183 runs on the VALU.
//  * Draw *runs* whose length is known up front (weighted_permutations keys, random
//    blocks, permutation keys) are evaluated lane-parallel with AS183 jump-ahead:
//    state after k draws = (A1*171^k mod 30269, A2*172^k mod 30307, A3*170^k mod 30323),
//    so lane l computes draw l+1 directly ("counter-based" use of erlamsa_rnd).
//  * Byte work (copies, fills, compares, scans) is spread over the 64 lanes with 16-byte
//    accesses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eh_common.h"

namespace eh {

#define EH_LANE ((int)threadIdx.x)
#define EH_DEV __device__ __forceinline__

EH_DEV uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
EH_DEV uint64_t uni64(uint64_t v) { return ((uint64_t)uni((uint32_t)(v >> 32)) << 32) | uni((uint32_t)v); }
EH_DEV void wave_sync() { __syncthreads(); }  // workgroup == one wavefront

// ---------------------------------------------------------------------------------------------
// Tables in constant memory, computed by the compiler.  (Until round 4 eh_create filled them with hipMemcpyToSymbol: the only
// GPU write between process start and the first launch, into the code object's read-only segment - and the place where, on
// two boxes of round 4 and on the driver's box of round 3, every process died with "Memory access fault by GPU".  Whether
// those boxes were sick or that write was the cause never showed on a healthy box; now there is no such write.)
// AS183 jump tables: T?[k] = mult^k mod prime, k = 0..64
// ---------------------------------------------------------------------------------------------
struct alignas(16) TabU16x65 { uint16_t v[65]; };   // (16-byte aligned like the arrays they replace: at the struct's natural alignment of 2 the 16-bit reads become vector loads instead of scalar ones)
constexpr TabU16x65 as183_powers(uint32_t mult, uint32_t prime) {
  TabU16x65 t{};
  uint32_t a = 1;
  for (int k = 0; k <= 64; k++) { t.v[k] = (uint16_t)a; a = a * mult % prime; }
  return t;
}
__constant__ TabU16x65 c_T1 = as183_powers(171, 30269);
__constant__ TabU16x65 c_T2 = as183_powers(172, 30307);
__constant__ TabU16x65 c_T3 = as183_powers(170, 30323);
// funny_unicode/0 (erlamsa_mutations.erl:1053-1078): 17 hand-written sequences, then the UTF-8 encodings of the code points
// produced by folding (with prepend) over the Codes list.  [len, b0..b3] per entry.
struct alignas(16) FunnyTab { uint8_t v[192][5]; int n; };
constexpr FunnyTab funny_unicode() {
  constexpr uint8_t manual[17][5] = {{3, 239, 191, 191}, {4, 240, 144, 128, 128}, {3, 0xef, 0xbb, 0xbf}, {2, 0xfe, 0xff}, {2, 0xff, 0xfe},
                                     {4, 0, 0, 0xff, 0xff}, {4, 0xff, 0xff, 0, 0}, {4, 43, 47, 118, 56}, {4, 43, 47, 118, 57},
                                     {4, 43, 47, 118, 43}, {4, 43, 47, 118, 47}, {3, 247, 100, 76}, {4, 221, 115, 102, 115},
                                     {3, 14, 254, 255}, {3, 251, 238, 40}, {4, 251, 238, 40, 255}, {4, 132, 49, 149, 51}};
  constexpr uint32_t codes[29][2] = {{0x0009, 0x000d}, {0x008D, 0x008D}, {0x00a0, 0x00a0}, {0x1680, 0x1680}, {0x180e, 0x180e},
                                     {0x2000, 0x200a}, {0x2028, 0x2028}, {0x2029, 0x2029}, {0x202f, 0x202f}, {0x205f, 0x205f},
                                     {0x3000, 0x3000}, {0x200e, 0x200f}, {0x202a, 0x202e}, {0x200c, 0x200d}, {0x0345, 0x0345},
                                     {0x00b7, 0x00b7}, {0x02d0, 0x02d1}, {0xff70, 0xff70}, {0x02b0, 0x02b8}, {0xfdd0, 0xfdd0},
                                     {0x034f, 0x034f}, {0x115f, 0x1160}, {0x2065, 0x2069}, {0x3164, 0x3164}, {0xffa0, 0xffa0},
                                     {0xe0001, 0xe0001}, {0xe0020, 0xe007f}, {0x0e40, 0x0e44}, {0x1f4a9, 0x1f4a9}};
  FunnyTab t{};
  int n = 0;
  for (int i = 0; i < 17; i++, n++) for (int k = 0; k < 5; k++) t.v[n][k] = manual[i][k];
  for (int g = 28; g >= 0; g--)
    for (uint32_t p = codes[g][0]; p <= codes[g][1]; p++, n++) {
      uint8_t* e = t.v[n];
      if (p < 0x80) { e[0] = 1; e[1] = (uint8_t)p; }
      else if (p < 0x800) { e[0] = 2; e[1] = (uint8_t)(0xc0 | (0x1f & (p >> 6))); e[2] = (uint8_t)((p & 0x3f) | 0x80); }
      else if (p < 0x10000) { e[0] = 3; e[1] = (uint8_t)(0xe0 | (0x0f & (p >> 12))); e[2] = (uint8_t)(((p >> 6) & 0x3f) | 0x80); e[3] = (uint8_t)((p & 0x3f) | 0x80); }
      else { e[0] = 4; e[1] = (uint8_t)(0xf0 | (0x7 & (p >> 18))); e[2] = (uint8_t)(((p >> 12) & 0x3f) | 0x80); e[3] = (uint8_t)(((p >> 6) & 0x3f) | 0x80); e[4] = (uint8_t)((p & 0x3f) | 0x80); }
    }
  t.n = n;
  return t;
}
__constant__ FunnyTab c_funny = funny_unicode();
static_assert(funny_unicode().n == 179, "funny_unicode/0: 17 sequences + 162 code points");

constexpr uint32_t P1 = 30269, P2 = 30307, P3 = 30323;

EH_DEV uint32_t modpow(uint32_t base, uint64_t e, uint32_t p) {
  uint32_t r = 1;
  while (e) { if (e & 1) r = (r * base) % p; base = (base * base) % p; e >>= 1; }
  return r;
}

struct Rng {
  uint32_t a1, a2, a3;
  uint64_t draws;
};

// U from a POST-STEP state: R = B1/30269 + B2/30307 + B3/30323 ; U = R - trunc(R)
// The three quotients must be the correctly rounded IEEE quotients (that is what the BEAM computes).
// A full f64 division expands to ~12 dependent instructions; for these divisors and integer numerators
// below 2^15 one Newton-style correction of the reciprocal product is already correctly rounded:
//   q0 = b * RN(1/P);  e = fma(-P, q0, b);  q = fma(e, RN(1/P), q0)
// which tests/test_oracle_rng.py checks exhaustively (all 90 899 numerators) against real division.
EH_DEV double as183_div(uint32_t b, double P, double rP) {
  double a = (double)b;
  double q0 = a * rP;
  double e = fma(-P, q0, a);
  return fma(e, rP, q0);
}
EH_DEV double u_of(uint32_t b1, uint32_t b2, uint32_t b3) {
  double q1 = as183_div(b1, 30269.0, 1.0 / 30269.0);
  double q2 = as183_div(b2, 30307.0, 1.0 / 30307.0);
  double q3 = as183_div(b3, 30323.0, 1.0 / 30323.0);
  double r = (q1 + q2) + q3;
  return r - trunc(r);
}
EH_DEV void rng_seed(Rng& r, int64_t s1, int64_t s2, int64_t s3) {  // random:seed/3
  uint64_t x1 = (uint64_t)(s1 < 0 ? -s1 : s1), x2 = (uint64_t)(s2 < 0 ? -s2 : s2), x3 = (uint64_t)(s3 < 0 ? -s3 : s3);
  r.a1 = (uint32_t)(x1 % (P1 - 1)) + 1;
  r.a2 = (uint32_t)(x2 % (P2 - 1)) + 1;
  r.a3 = (uint32_t)(x3 % (P3 - 1)) + 1;
}
EH_DEV double rng_uniform(Rng& r) {
  r.a1 = (r.a1 * 171u) % P1;
  r.a2 = (r.a2 * 172u) % P2;
  r.a3 = (r.a3 * 170u) % P3;
  r.draws++;
  return u_of(r.a1, r.a2, r.a3);
}
// advance by k draws (k <= 64 via table, else modpow)
EH_DEV void rng_skip(Rng& r, uint64_t k) {
  if (k <= 64) {
    r.a1 = (r.a1 * (uint32_t)c_T1.v[k]) % P1; r.a2 = (r.a2 * (uint32_t)c_T2.v[k]) % P2; r.a3 = (r.a3 * (uint32_t)c_T3.v[k]) % P3;
  } else {
    r.a1 = (r.a1 * modpow(171, k, P1)) % P1; r.a2 = (r.a2 * modpow(172, k, P2)) % P2; r.a3 = (r.a3 * modpow(170, k, P3)) % P3;
  }
  r.draws += k;
}
// the j-th (1-based, j <= 64) NEXT uniform without advancing the state: per-lane j allowed
EH_DEV double rng_peek(const Rng& r, uint32_t j) {
  uint32_t b1 = (r.a1 * (uint32_t)c_T1.v[j]) % P1, b2 = (r.a2 * (uint32_t)c_T2.v[j]) % P2, b3 = (r.a3 * (uint32_t)c_T3.v[j]) % P3;
  return u_of(b1, b2, b3);
}
// erlamsa_rnd:rand/1 (erlamsa_rnd.erl:77): rand(0) = 0 without a draw
EH_DEV uint32_t rng_rand(Rng& r, uint32_t n) {
  if (n == 0) return 0;
  double x = rng_uniform(r) * (double)n;
  return uni((uint32_t)x);
}
EH_DEV uint64_t rng_rand64(Rng& r, uint64_t n) {  // n < 2^53
  if (n == 0) return 0;
  double x = rng_uniform(r) * (double)n;
  return uni64((uint64_t)x);
}
EH_DEV uint32_t rng_erand(Rng& r, uint32_t n) { return n == 0 ? 0 : rng_rand(r, n) + 1; }  // :82
EH_DEV uint32_t rng_range(Rng& r, int64_t lo, int64_t hi) {                                  // :87-92
  if (hi > lo) return rng_rand(r, (uint32_t)(hi - lo)) + (uint32_t)lo;
  if (hi == lo) return (uint32_t)lo;
  return 0;
}
EH_DEV int rng_bit(Rng& r) { return uni(rng_uniform(r) >= 0.5 ? 1u : 0u); }                   // :105 round/1
EH_DEV int rng_delta(Rng& r) { return rng_bit(r) == 0 ? 1 : -1; }                             // :224-231
EH_DEV bool rng_occurs(Rng& r, uint32_t nom, uint32_t den) {                                  // :123-130
  uint32_t n = rng_rand(r, den);
  return nom == 1 ? (n != 0) : (n < nom);
}
// rand_log(n) for n <= 32 (values < 2^31): rand_nbit(rand(n))  :134-143
EH_DEV uint32_t rng_log(Rng& r, uint32_t n) {
  if (n == 0) return 0;
  uint32_t k = rng_rand(r, n);
  if (k == 0) return 0;
  uint32_t hi = 1u << (k - 1);
  return hi | rng_rand(r, hi);
}

// ---------------------------------------------------------------------------------------------
// 4 / 8 / 16 bytes of HBM in one instruction.  ldg* / stg*: any alignment (the device runs with unaligned access on: one
// global_load_dwordx4 whatever the address), *a: the address is a multiple of the size.  (The vector types of the HIP headers are
// classes, which cannot be copied out of an address-space qualified lvalue: the accesses go through the compiler's own vectors.)
// ---------------------------------------------------------------------------------------------
#ifdef HIPEMU
EH_DEV uint4 ldg16(const void* p) { uint4 v; memcpy(&v, p, 16); return v; }
EH_DEV uint4 ldg16a(const void* p) { uint4 v; memcpy(&v, p, 16); return v; }
EH_DEV void stg16(void* p, const uint4& v) { memcpy(p, &v, 16); }
EH_DEV void stg16a(void* p, const uint4& v) { memcpy(p, &v, 16); }
EH_DEV uint64_t ldg8(const void* p) { uint64_t v; memcpy(&v, p, 8); return v; }
EH_DEV uint32_t ldg4(const void* p) { uint32_t v; memcpy(&v, p, 4); return v; }
#else
typedef uint32_t eh_v4 __attribute__((ext_vector_type(4)));
typedef eh_v4 __attribute__((aligned(1))) eh_v4u;
typedef uint64_t __attribute__((aligned(1))) eh_u64u;
typedef uint32_t __attribute__((aligned(1))) eh_u32u;
EH_DEV uint4 ldg16(const EH_G void* p) { eh_v4 v = *reinterpret_cast<const EH_G eh_v4u*>(p); uint4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
EH_DEV uint4 ldg16a(const EH_G void* p) { eh_v4 v = *reinterpret_cast<const EH_G eh_v4*>(p); uint4 r; r.x = v.x; r.y = v.y; r.z = v.z; r.w = v.w; return r; }
EH_DEV void stg16(EH_G void* p, const uint4& r) { eh_v4 v; v.x = r.x; v.y = r.y; v.z = r.z; v.w = r.w; *reinterpret_cast<EH_G eh_v4u*>(p) = v; }
EH_DEV void stg16a(EH_G void* p, const uint4& r) { eh_v4 v; v.x = r.x; v.y = r.y; v.z = r.z; v.w = r.w; *reinterpret_cast<EH_G eh_v4*>(p) = v; }
EH_DEV uint64_t ldg8(const EH_G void* p) { return *reinterpret_cast<const EH_G eh_u64u*>(p); }
EH_DEV uint32_t ldg4(const EH_G void* p) { return *reinterpret_cast<const EH_G eh_u32u*>(p); }
#endif

// ---------------------------------------------------------------------------------------------
// wave-parallel byte movers (HBM to HBM: corpus arena, slot and pool work memory, output arena)
// ---------------------------------------------------------------------------------------------
// Copies of CO_COPY_MIN bytes and more inside a case are posted for several wavefronts (co_copy below); the _raw forms are the
// loops themselves (one wavefront), for the kernels that are not cases and for the chunks of a posted copy.
// CO_COPY_MIN: what the inlined test at every call site compares with; the size from which a copy IS posted and its chunks are run-time
// parameters (KParams co_copy_min >= CO_COPY_MIN, co_copy_chunk; eh_engine.hip co_defaults).
#ifdef HIPEMU
constexpr uint32_t CO_COPY_MIN = 8u << 10;         // (the emulator's tests are small: they take the posted path too, one wavefront running every chunk)
#else
constexpr uint32_t CO_COPY_MIN = 256u << 10;
#endif
__device__ void co_copy(bptr dst, cbptr src, uint32_t n);
__device__ bool co_equal(cbptr a, cbptr b, uint32_t n);
EH_DEV void wave_copy_raw(bptr dst, cbptr src, uint32_t n) {
  const int l = EH_LANE;
  // head: bring dst to 16-byte alignment
  uint32_t head = (uint32_t)((16 - ((uintptr_t)dst & 15)) & 15);
  if (head > n) head = n;
  if ((uint32_t)l < head) dst[l] = src[l];
  dst += head; src += head; n -= head;
  uint32_t nv = n >> 4;
  uint32_t i = l;
  // 4 independent 16-byte loads in flight per lane (4 KiB per wave per round trip)
  for (; i + 192 < nv; i += 256) {
    const uint4 v0 = ldg16(src + 16 * (size_t)i), v1 = ldg16(src + 16 * (size_t)(i + 64));
    const uint4 v2 = ldg16(src + 16 * (size_t)(i + 128)), v3 = ldg16(src + 16 * (size_t)(i + 192));
    stg16a(dst + 16 * (size_t)i, v0);
    stg16a(dst + 16 * (size_t)(i + 64), v1);
    stg16a(dst + 16 * (size_t)(i + 128), v2);
    stg16a(dst + 16 * (size_t)(i + 192), v3);
  }
  for (; i < nv; i += 64) {
    const uint4 v = ldg16(src + 16 * (size_t)i);      // unaligned dwordx4 load
    stg16a(dst + 16 * (size_t)i, v);
  }
  uint32_t done = nv << 4;
  if (done + l < n) dst[done + l] = src[done + l];
}
EH_DEV void wave_copy(bptr dst, cbptr src, uint32_t n) {
  if (__builtin_expect(n >= CO_COPY_MIN, 0)) { co_copy(dst, src, n); return; }
  wave_copy_raw(dst, src, n);
}
// memmove towards lower addresses (dst < src, ranges may overlap): every iteration is executed by
// the whole wave — all loads of a 4 KiB stripe are issued before its stores — so no lane can read
// bytes that a lane running ahead has already overwritten (wave_copy's per-lane loop trip counts
// differ, which is only safe for disjoint ranges).
EH_DEV void wave_move_down(bptr dst, cbptr src, uint32_t n) {
  const int l = EH_LANE;
  uint32_t nv = n >> 4;
  for (uint32_t base = 0; base < nv; base += 256) {
    uint4 v[4]; bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { uint32_t i = base + 64u * u + (uint32_t)l; ok[u] = i < nv; if (ok[u]) v[u] = ldg16(src + 16 * (size_t)i); }
    wave_sync();                                     // every lane has loaded its 4 chunks before any lane stores
#pragma unroll
    for (int u = 0; u < 4; u++) { uint32_t i = base + 64u * u + (uint32_t)l; if (ok[u]) stg16(dst + 16 * (size_t)i, v[u]); }
  }
  uint32_t done = nv << 4;
  uint8_t t = 0; bool tk = done + (uint32_t)l < n;
  if (tk) t = src[done + l];
  wave_sync();
  if (tk) dst[done + l] = t;
}
// dst[i] = pat[i % plen], i < total.  The first copy is written directly, the rest by doubling
// (dst[0,k) -> dst[k,2k)) so that all but the first plen bytes move as 16-byte vectors.
EH_DEV void wave_fill_periodic(bptr dst, cbptr pat, uint32_t plen, uint64_t total) {
  if (total == 0) return;
  uint64_t have = plen < total ? plen : total;
  wave_copy(dst, pat, (uint32_t)have);
  while (have < total) {
    wave_sync();
    uint64_t chunk = have < total - have ? have : total - have;
    // copy in <= 1 GiB pieces (32-bit length of wave_copy)
    uint64_t done = 0;
    while (done < chunk) { uint32_t c = chunk - done > 0x40000000ull ? 0x40000000u : (uint32_t)(chunk - done); wave_copy(dst + have + done, dst + done, c); done += c; }
    have += chunk;
  }
}
// returns true if the two byte ranges are equal
EH_DEV bool wave_equal_raw(cbptr a, cbptr b, uint32_t n) {
  const int l = EH_LANE;
  uint32_t nv = n >> 4;
  // uniform trip count (the early exit must be taken by the whole wave)
  for (uint32_t base = 0; base < nv; base += 128) {
    uint32_t i0 = base + l, i1 = base + 64 + l;
    bool ne = false;
    uint4 x0, y0, x1, y1;
    if (i0 < nv) { x0 = ldg16(a + 16 * (size_t)i0); y0 = ldg16(b + 16 * (size_t)i0); }
    if (i1 < nv) { x1 = ldg16(a + 16 * (size_t)i1); y1 = ldg16(b + 16 * (size_t)i1); }
    if (i0 < nv) ne |= (x0.x != y0.x) | (x0.y != y0.y) | (x0.z != y0.z) | (x0.w != y0.w);
    if (i1 < nv) ne |= (x1.x != y1.x) | (x1.y != y1.y) | (x1.z != y1.z) | (x1.w != y1.w);
    if (__ballot(ne) != 0) return false;
  }
  uint32_t done = nv << 4;
  bool ne = false;
  if (done + l < n) ne = a[done + l] != b[done + l];
  return __ballot(ne) == 0;
}
EH_DEV bool wave_equal(cbptr a, cbptr b, uint32_t n) {
  if (__builtin_expect(n >= CO_COPY_MIN, 0)) return co_equal(a, b, n);
  return wave_equal_raw(a, b, n);
}

// ---------------------------------------------------------------------------------------------
// Per-case context (all wave-uniform)
// ---------------------------------------------------------------------------------------------
// site: 1xx eh_device.h, 2xx eh_doc.h, 3xx eh_engine.hip, 4xx eh_json.h, 5xx eh_lex.h, 6xx eh_sgml.h, 7xx eh_text.h, 8xx eh_tree.h
constexpr int MAX_CHUNKS = POOL_TIERS + 1;
constexpr int MAX_NEST = 6;         // nested scheduler calls (b64 / sgm / js inner mutations)
constexpr int LEX_LEVELS = MAX_NEST + 1;
constexpr int ST_STATE_WORDS = 84;  // sizeof(StState) / 4 (eh_text.h; checked there)
#define EH_SET_OVERFLOW(c, site) ((c).ovf_line = (site), (c).ovf_need = 0, (c).ovf_req = 0, (c).status = CASE_OVERFLOW)
struct MuFrame;
struct Ctx {
  Rng rng;
  const EH_G KParams* p;
  // block list: bl[cur..nb) is the list handed to the mutator; bl[0..cur) already emitted
  EH_G Blk* bl;
  EH_G Blk* bl2;
  int nb, cur;
  EH_G Blk* em;
  int nem;
  bptr aux;        // per-slot mutator state (lis/lrs lines, fo block)
  // linear work allocator
  bptr ws;
  // ws + ws_used is the next free byte; ws_used and ws_cap are VIRTUAL offsets that run on across the chunks of the work
  // area (chunk 0 = the slot's own area, chunks 1.. = larger areas borrowed from the pool when the case outgrows what it has,
  // see ws_grow), and ws is the current chunk's base minus the chunk's virtual start, so the arithmetic is that of one area
  uint64_t ws_used, ws_cap;
  int nchunk;          // chunks above the slot's own
  int view;            // chunk ws / ws_lo / ws_cap describe (the one the last allocation came from)
  uint64_t ws_lo;      // its virtual start
  uint64_t ch_vstart[MAX_CHUNKS], ch_vend[MAX_CHUNKS]; bptr ch_base[MAX_CHUNKS]; uint32_t ch_area[MAX_CHUNKS]; int32_t ch_tier[MAX_CHUNKS];
  uint64_t lex_ptr[LEX_LEVELS];   // block the lex cache of nesting level d holds a table for (mirror of LexCache::ptr; 0: none)
  bptr trace;      // EH_FLAG_META_TRACE: the case's event bytes (slot memory), nullptr = off
  uint32_t ntrace, tr_base;   // bytes written; where the Meta list in hand begins (tr_drop_before)
  uint32_t co_posted;  // the case has posted a loop for other wavefronts (counted in CoBoard::posters until it ends)
  // nested scheduler calls without device recursion (mux_fuzzers): what a mutator that wants Muta([Bin], []) run leaves for the scheduler,
  // and what the scheduler leaves for the mutator when it calls it again
  int32_t call_req, mu_phase, call_nres, call_nfs;
  uint64_t call_bin; uint32_t call_len;
  EH_G MuFrame* mu;
  uint64_t t_case;     // cycle stamp at which the case began (mux_fuzzers raises the wavefront's issue priority for cases that run long)
  int32_t m_aux;       // set by the mutators whose own Meta entry does not follow from their result alone (num: a number found; ab / ad: stringy)
  uint64_t ws_peak, ws_top;   // diagnostics: highest ws_used, bytes taken from the top of chunks (eh_result_peak)
  int status;
  int lastm;
  uint64_t work;       // bytes handed to mutators so far (deterministic stand-in for maxrunningtime)
  // mutator result (candidate new head of the list)
  int r_kind;          // R_SAME: list unchanged ; R_NEW: r_ptr/r_len replace bl[cur]
  bptr r_ptr;
  uint32_t r_len;
  int r_flush;         // result goes through flush_bvecs/2
  int r_drop_next;     // fn consumed the following block
  int r_changed;       // mutator guarantees hd(result) != hd(input): skip the compare
  int r2; bptr r2_ptr; uint32_t r2_len;   // optional second flushed region (fo: flush_bvecs(A, flush_bvecs(B, T)))
  int nfs;             // entries of the mux_fuzzers list (the entries themselves: LaneTab)
  uint64_t work_budget;
  uint64_t ovf_need, ovf_req;   // work area the case had asked for in total / in the request that failed (0: unknown): picks the tier that runs it again
  int ovf_line;        // site id that set CASE_OVERFLOW (diagnostic: reported as -line in the last-mutator array)
  int depth;           // nesting depth of mux_fuzzers (b64 / sgm / js inner mutations re-enter the scheduler)
  int gen_pending;     // G_FILE / G_JUMP: the generator's fun has not been called yet (gen_force, eh_engine.hip); 0 = Ll is a list
  uint32_t gen_e1, gen_e2;   // the corpus entries (paths) it was made for
  int pat_ret; uint32_t pat_ip; int pat_cont;   // results of the container patterns' helpers (cp_end / ar_step, eh_engine.hip)  // eh_fuse_red.h: the search runs on shortened lists and only names its node (fp_on); the members are found in the originals
  uint32_t fp_on, fp_g, fp_special, fp_keypos, fp_bA, fp_bB;
};
// The per-case context lives in LDS.  One workgroup is one wavefront, so there is exactly one Ctx per
// workgroup and no synchronisation is needed.  As a stack object it was reached through generic
// pointers from every non-inlined mutator function: each field access was a flat scratch access with
// a full memory round trip behind it (72 % of all wave cycles were s_waitcnt).  Non-inlined functions
// therefore ignore their Ctx& argument and bind `c` to the LDS object directly (EH_CTX), which lets the
// compiler emit ds_read/ds_write.
__shared__ Ctx g_ctx;
// LDS scratch that lanes write individually (histograms).  The CPU wavefront emulator of tests/hipemu models
// __shared__ as wave-uniform state, so there the array is plain memory (one wavefront runs at a time).
#ifdef HIPEMU
#define EH_LDS_ARRAY(type, name, n) static type name[n]
#define EH_KEEP(v) do {} while (0)
#else
#define EH_LDS_ARRAY(type, name, n) __shared__ type name[n]
// The value is used here, as far as the optimizer can tell: a load that produced it stays where it is.  (An LDS load and a global load in
// the two arms of a condition are otherwise merged into ONE load through a generic pointer: slower, and it has sent this compiler's
// back end into "Illegal instruction detected".)
#define EH_KEEP(v) asm volatile("" : "+v"(v))
#endif
// orders LDS accesses of different lanes of the wavefront.  The hardware runs a wavefront's LDS instructions in
// program order, so nothing is emitted; on the emulator the ballot is a rendezvous of the lane fibers.
EH_DEV void lanes_sync() { (void)__ballot(1); }
#define EH_CTX Ctx& c = g_ctx
// The LDS band of the fuse mutators (eh_fuse.h, eh_fuse2.h, eh_fuse_lds.h) and the sgm tokenizer.  Outside of them it is free, and
// the pattern-level finders stage their tables there: the CRC tables of wave_crc32 (eh_zlib.h), the right ends and counts of
// pick_simple_len (eh_field.h), the prefix CRCs and hit masks of pick_csum (eh_engine.hip).  Nobody keeps anything in it across calls.
#ifndef EH_FUSE_LDS_WORDS
#define EH_FUSE_LDS_WORDS 4800
#endif
EH_LDS_ARRAY(uint32_t, g_fuse_lds, EH_FUSE_LDS_WORDS);
EH_LDS_ARRAY(uint32_t, g_st_save, ST_STATE_WORDS);   // lis / lrs store as it was before the running attempt (mux_fuzzers)
// per-lane mux_fuzzers entry (lane i = list position i); private registers, never in LDS
struct LaneTab {
  uint32_t e_pri;
  uint32_t e_meta;     // score | fn<<8 | name<<16 | mask<<24
};
enum { R_SAME = 0, R_NEW = 1 };

// EH_PROF builds: cycle counters per code region (eh_result_prof slots; 0..63 mutators, 64.. phases)
#ifdef EH_PROF
#define EH_PT0 uint64_t pt_ = __builtin_readcyclecounter()
#define EH_PT(c, k) do { uint64_t n_ = __builtin_readcyclecounter(); if (EH_LANE == 0) { atomicAdd(&(c).p->prof[2 * (k)], (unsigned long long)(n_ - pt_)); atomicAdd(&(c).p->prof[2 * (k) + 1], 1ull); } pt_ = n_; } while (0)
#else
#define EH_PT0 do {} while (0)
#define EH_PT(c, k) do {} while (0)
#endif

// ---- meta trace (erlamsa_main.erl:58-70 prints the Meta list a case has built).  Events (eh_common.h TraceKind) are appended in the
// order the reference PRINTS the elements: lists:reverse(lists:flatten(Meta)) - the order in time for everything that is consed as
// it happens, which is nearly everything; the few literal lists ([A, B | Meta]) are written back to front at their sites.
// tr_base = where the Meta list in hand begins: Muta([Bin], []) (b64 / sgm / js inner runs, nested_fuzz) and
// mutate_once_loop(Mutator, [], ..) (cp, ar) start lists of their own, and sgml_mutate / json_mutate return NewMeta ALONE when the
// block comes back unchanged (erlamsa_sgml.erl:748-749, erlamsa_json.erl:722-723), which drops what the list in hand held before.
// (The emitters are functions of their own, called only when a trace is kept: inlined into the scheduler loop they cost the
// kernel 7 % of its wave cycles with tracing OFF - 18 % on cases that nest thousands of scheduler calls -, measured on the
// build that first had them, profiles/r05_bench_after_tests_same_call.json against r05_bench_driver_command_mid_round.json.)
EH_DEV void tr_b(Ctx& c, uint32_t v) {
  if (c.ntrace < TRACE_CAP - 1) { if (EH_LANE == 0) c.trace[c.ntrace] = (uint8_t)v; c.ntrace++; }
  else { if (EH_LANE == 0) c.trace[TRACE_CAP - 1] = 0xFF; c.ntrace = TRACE_CAP; }
}
EH_DEV void tr_v(Ctx& c, uint64_t v) { while (v >= 128) { tr_b(c, (uint32_t)(v & 127u) | 128u); v >>= 7; } tr_b(c, (uint32_t)v); }
__device__ __noinline__ void tr_aa_emit(int a, int b) { Ctx& c = g_ctx; tr_b(c, TRK_AA); tr_b(c, (uint32_t)a); tr_b(c, (uint32_t)b); }
__device__ __noinline__ void tr_ai_emit(int a, int64_t v) { Ctx& c = g_ctx; tr_b(c, TRK_AI); tr_b(c, (uint32_t)a); tr_v(c, ((uint64_t)v << 1) ^ (uint64_t)(v >> 63)); }
// kind byte + up to five LEB128 operands (sizer, csum, skipped)
__device__ __noinline__ void tr_kv_emit(int kind, int nb, uint32_t b0, uint32_t b1, int nv, uint64_t v0, uint64_t v1, uint64_t v2) {
  Ctx& c = g_ctx;
  tr_b(c, (uint32_t)kind);
  if (nb > 0) tr_b(c, b0);
  if (nb > 1) tr_b(c, b1);
  if (nv > 0) tr_v(c, v0);
  if (nv > 1) tr_v(c, v1);
  if (nv > 2) tr_v(c, v2);
}
// {archiver, Name}: `up` x "../" in front of the name
__device__ __noinline__ void tr_name_emit(cbptr nm, uint32_t nl, uint32_t up) {
  Ctx& c = g_ctx;
  tr_b(c, TRK_ARCHIVER); tr_v(c, nl + 3u * up);
  for (uint32_t k = 0; k < up; k++) { tr_b(c, '.'); tr_b(c, '.'); tr_b(c, '/'); }
  for (uint32_t k = 0; k < nl; k++) tr_b(c, uni(nm[k]));
}
EH_DEV void tr_aa(Ctx& c, int a, int b) { if (__builtin_expect(c.trace != nullptr, 0)) tr_aa_emit(a, b); }
EH_DEV void tr_ai(Ctx& c, int a, int64_t v) { if (__builtin_expect(c.trace != nullptr, 0)) tr_ai_emit(a, v); }
// drops the events [tr_base, start) of the list in hand (see above); what came after start moves down
EH_DEV void tr_drop_before(Ctx& c, uint32_t start) {
  if (!c.trace || start <= c.tr_base || c.ntrace >= TRACE_CAP) return;
  const uint32_t n = c.ntrace - start;
  wave_sync();
  if (n) wave_move_down(c.trace + c.tr_base, c.trace + start, n);
  wave_sync();
  c.ntrace = c.tr_base + n;
}

// ---- work-area pool (see KParams): lane 0 talks to the rings, the wave takes the result.  A popper owns ring entry
// (ticket mod count) and waits until a pusher has filled it; a pusher waits until the entry's previous value has been
// taken.  A wavefront only ever waits for an area of a HIGHER tier than any it holds, so the waits end.
EH_DEV void pool_nap() {
#ifndef HIPEMU
  __builtin_amdgcn_s_sleep(64);
#else
  fprintf(stderr, "hipemu: a wait for a pool area can never end with one wavefront running (block %u)\n", blockIdx.x); abort();
#endif
}
EH_DEV uint32_t pool_pop(const EH_G KParams& p, int t) {
  uint32_t v = 0xFFFFFFFFu;
  if (EH_LANE == 0) {
    unsigned long long h = atomicAdd(&p.pool_ctr[2 * t], 1ull);
    wptr e = p.pool_ring[t] + (h % p.pool_cnt[t]);
    v = atomicExch(e, 0xFFFFFFFFu);
    if (v == 0xFFFFFFFFu) {                                          // every area of the tier is out: wait for a push (eh_pool_stats counts the cycles)
      uint64_t w0 = __builtin_readcyclecounter();
      for (;;) { pool_nap(); v = atomicExch(e, 0xFFFFFFFFu); if (v != 0xFFFFFFFFu) break; }
      atomicAdd(&p.pool_ctr[20 + t], (unsigned long long)(__builtin_readcyclecounter() - w0));
      atomicAdd(&p.pool_ctr[30 + t], 1ull);
    }
    // the most areas of the tier wanted at the same time (eh_pool_stats): what says how to split the pool's memory over the tiers
    // (holders + wavefronts waiting for one, so it can exceed the tier's size)
    unsigned long long back = atomicAdd(&p.pool_ctr[2 * t + 1], 0ull);
    unsigned long long outn = h + 1 + p.pool_cnt[t] > back ? h + 1 + p.pool_cnt[t] - back : 0;
    atomicMax(&p.pool_ctr[40 + t], outn);
  }
#ifndef EH_NO_POOL_FENCE
  __threadfence();                                                   // the previous owner's stores (another XCD's L2) are behind us
#endif
  return uni(v);
}
EH_DEV void pool_push(const EH_G KParams& p, int t, uint32_t v) {
#ifndef EH_NO_POOL_FENCE
  __threadfence();                                                   // our stores to the area are written back before it changes hands
#endif
  if (EH_LANE == 0) {
    unsigned long long h = atomicAdd(&p.pool_ctr[2 * t + 1], 1ull);
    wptr e = p.pool_ring[t] + (h % p.pool_cnt[t]);
    while (atomicCAS(e, 0xFFFFFFFFu, v) != 0xFFFFFFFFu) pool_nap();
  }
}

// ---- cooperative execution (eh_common.h CoBoard) -----------------------------------------------------------------------------
// co_exec runs chunk t of a posted job with the whole wavefront (eh_engine.hip: it knows every kind of loop).
struct CoArgs { uint64_t a[CO_ARGS]; };
__device__ void co_exec(const EH_G CoJob* j, uint32_t t);
EH_DEV void co_nap() {
#ifndef HIPEMU
  __builtin_amdgcn_s_sleep(16);
#else
  fprintf(stderr, "hipemu: a wait for a helper's chunk can never end with one wavefront running (block %u)\n", blockIdx.x); abort();
#endif
}
// Agent-scope release / acquire around what changes hands between wavefronts on different compute units and XCDs (the per-XCD L2s
// are not coherent with each other, a CU's vector L1 is never refreshed by other CUs' stores).  Release: this XCD's dirty L2 lines are
// written back - and the wait is spelled out: the compiler drops the s_waitcnt vmcnt(0) behind buffer_wbl2 when it believes no
// memory operation is in flight, and the flag would overtake the data.  Acquire: this CU's L1 is invalidated.  Microseconds each:
// once per posted loop / per chunk, and chunks are hundreds of kilobytes.
EH_DEV void co_release() {
#ifndef HIPEMU
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}
EH_DEV void co_acquire() {
#ifndef HIPEMU
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
}
EH_DEV uint32_t co_ld32(const EH_G uint32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
EH_DEV unsigned long long co_ld64(const EH_G unsigned long long* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
// Posts a loop of `nchunks` chunks, takes chunks itself, waits for the ones others took.  false: no board, or no free entry - the
// caller runs the loop itself.  *acc: what the chunks added up.  Everything the case has written is visible to the helpers (fence
// before the entry opens), everything they wrote is visible to the case when this returns (fence after the last chunk).
__device__ __noinline__ bool co_run(uint32_t kind, uint32_t nchunks, CoArgs args, unsigned long long* acc) {
  Ctx& c = g_ctx;
  EH_G CoBoard* bd = c.p ? c.p->board : nullptr;
  const int l = EH_LANE;
  if (!bd || nchunks < 2) return false;
  uint32_t slot = CO_JOBS;
  if (l == 0) {
    const uint32_t h = (blockIdx.x * 2654435761u) >> 26;
    for (uint32_t k = 0; k < 4 && slot == CO_JOBS; k++) { const uint32_t s = (h + 17u * k) & (CO_JOBS - 1); if (atomicCAS(&bd->owner[s], 0u, 1u) == 0u) slot = s; }
    if (slot == CO_JOBS) atomicAdd(&bd->stat[4], 1ull);
  }
  slot = uni(slot);
  if (slot == CO_JOBS) return false;
  EH_G CoJob* j = &bd->job[slot];
  unsigned long long gen = 0;
  if (l == 0) {
#pragma unroll
    for (uint32_t k = 0; k < CO_ARGS; k++) j->a[k] = args.a[k];
    j->kind = kind; j->nchunks = nchunks; j->done = 0; j->acc = 0;
  }
  wave_sync();                                                             // every lane's stores of the case so far are issued ...
  co_release();                                                            // ... and written back: the helpers read the loop's inputs from memory
  if (l == 0) {
    if (!c.co_posted) atomicAdd(&bd->posters, 1u);                          // (until the case ends: eh_mutate_kernel)
    gen = ((co_ld64(&bd->word[slot]) >> 32) + 1ull) | 1ull;                 // (closed words carry even generations)
    atomicExch(&bd->word[slot], gen << 32);
    atomicAdd(&bd->open, 1u);
    atomicAdd(&bd->stat[0], 1ull);
  }
  c.co_posted = 1;
  uint32_t mine = 0;
  for (;;) {
    unsigned long long old = 0;
    if (l == 0) old = atomicAdd(&bd->word[slot], 1ull);
    const uint32_t t = uni((uint32_t)old);
    if (t >= nchunks) break;
    co_exec(j, t);
    mine++;
  }
  wave_sync();
  unsigned long long sum = 0;
  if (l == 0) {
    atomicAdd(&j->done, mine);
    const uint64_t w0 = __builtin_readcyclecounter();
    while (co_ld32(&j->done) < nchunks) co_nap();
    atomicAdd(&bd->open, 0xFFFFFFFFu);                                       // (- 1)
    atomicExch(&bd->word[slot], (gen + 1ull) << 32);                        // closed: a chunk number drawn from here on belongs to no job
    sum = co_ld64(&j->acc);
    atomicAdd(&bd->stat[2], (unsigned long long)mine);
    atomicAdd(&bd->stat[3], (unsigned long long)(__builtin_readcyclecounter() - w0));
  }
  sum = uni64(sum);
  co_acquire();                                                            // what the helpers wrote (they released it) is read from memory, not from this CU's L1
  if (l == 0) atomicExch(&bd->owner[slot], 0u);
  if (acc) *acc = sum;
  return true;
}
// A wavefront between two cases: up to `most` chunks of whatever is posted (a bound, so that the wavefront's own pass - the
// kernel the host waits for - is never kept alive by the loops of other passes).
__device__ __noinline__ void co_help(EH_G CoBoard* bd, uint32_t most) {
  const int l = EH_LANE;
  uint32_t ran = 0;
  for (uint32_t guard = 0; guard < 64 && ran < most; guard++) {
    if (uni(co_ld32(&bd->open)) == 0) return;
    const unsigned long long w = co_ld64(&bd->word[l]);
    const bool cand = ((w >> 32) & 1ull) && (uint32_t)w < co_ld32(&bd->job[l].nchunks);
    unsigned long long m = __ballot(cand);
    if (!m) return;
    const uint32_t rot = blockIdx.x & 63u;                                   // (helpers start at different entries)
    const unsigned long long mr = rot ? ((m >> rot) | (m << (64u - rot))) : m;
    const uint32_t s = ((uint32_t)__builtin_ctzll(mr) + rot) & 63u;
    unsigned long long old = 0;
    if (l == 0) old = atomicAdd(&bd->word[s], 1ull);
    old = uni64(old);
    if (!((old >> 32) & 1ull)) continue;                                     // closed in the meantime
    co_acquire();                                                            // the job's arguments and what its case wrote
    const EH_G CoJob* j = &bd->job[s];
    const uint32_t nchunks = uni(co_ld32(&j->nchunks)), t = (uint32_t)old;
    if (uni64(co_ld64(&bd->word[s])) >> 32 != old >> 32 || t >= nchunks) continue;   // (nchunks read under the generation the chunk was drawn from)
    co_exec(j, t);
    ran++;
    wave_sync();
    co_release();
    if (l == 0) { atomicAdd(&bd->job[s].done, 1u); atomicAdd(&bd->stat[1], 1ull); }
  }
}
__device__ __noinline__ void co_copy(bptr dst, cbptr src, uint32_t n) {
  const EH_G KParams* kp = g_ctx.p;
  if (!kp || !kp->board || n < kp->co_copy_min) { wave_copy_raw(dst, src, n); return; }
  const uint32_t ch = kp->co_copy_chunk;
  CoArgs a; for (uint32_t k = 0; k < CO_ARGS; k++) a.a[k] = 0;
  a.a[0] = (uint64_t)dst; a.a[1] = (uint64_t)src; a.a[2] = n; a.a[3] = ch;
  if (!co_run(CO_COPY, (n + ch - 1) / ch, a, nullptr)) wave_copy_raw(dst, src, n);
}
__device__ __noinline__ bool co_equal(cbptr x, cbptr y, uint32_t n) {
  const EH_G KParams* kp = g_ctx.p;
  if (!kp || !kp->board || n < 2 * kp->co_copy_min) return wave_equal_raw(x, y, n);
  const uint32_t ch = kp->co_copy_chunk;
  CoArgs a; for (uint32_t k = 0; k < CO_ARGS; k++) a.a[k] = 0;
  a.a[0] = (uint64_t)x; a.a[1] = (uint64_t)y; a.a[2] = n; a.a[3] = ch;
  unsigned long long ne = 0;
  if (!co_run(CO_EQUAL, (n + ch - 1) / ch, a, &ne)) return wave_equal_raw(x, y, n);
  return ne == 0;
}

// Lex caches (eh_lex.h), one per nesting level of the scheduler: a mutator that walks its block's chunk table may call the
// scheduler on a piece of it (base64 chunks, inner texts), and the mutators down there lex their own blocks.  A table lives at
// the top of a work-area chunk and survives candidate discards; tcap = entries it has room for (0: no table).
struct LexChunk;
struct LexCache { uint64_t ptr; uint32_t len; int32_t n; EH_G LexChunk* tab; uint32_t tcap; uint32_t pad; };
constexpr uint32_t AUX_LEXCACHE = 2048;                              // offset in Ctx::aux of LexCache[LEX_LEVELS]
constexpr uint64_t AUX_BYTES = 4096;
// a slot: block list, scratch list, emit list, aux, meta-trace bytes, then the work area
constexpr uint64_t SLOT_TABLE_BYTES = (uint64_t)(2 * MAX_BLOCKS + MAX_EMITS) * sizeof(Blk) + AUX_BYTES + TRACE_CAP;
EH_DEV EH_G LexCache& lex_slot(Ctx& c) { return ((EH_G LexCache*)(c.aux + AUX_LEXCACHE))[c.depth]; }
// Work memory at virtual offsets >= v0 is about to be reused: a lexed block that lives there (a decoded base64 chunk, an
// inner text — temporaries of a mutator attempt) is gone, and the cache is keyed by address.  Only this level's key can
// be up there: the blocks of the levels above are older than anything the running attempt allocated.
EH_DEV void lex_forget_from(Ctx& c, uint64_t v0) {
  const uint64_t key = c.lex_ptr[c.depth];
  if (key == 0) return;
  bool dead = false;
  for (int j = 0; j <= c.nchunk; j++) {
    uint64_t v = key - (uint64_t)(uintptr_t)c.ch_base[j], lo = v0 > c.ch_vstart[j] ? v0 : c.ch_vstart[j];
    if (v >= lo && v < c.ch_vstart[j] + c.p->pool_cap[c.ch_tier[j]]) dead = true;
  }
  if (dead) { c.lex_ptr[c.depth] = 0; if (EH_LANE == 0) lex_slot(c).n = -1; }
}
// ---- the work area of a case: a stack of chunks ---------------------------------------------------------------------
// Chunk 0 is the slot's own area; chunks 1..nchunk are larger areas the case borrowed when it outgrew what it had.  The
// allocator below never talks to the pool — mutators stay leaf functions without a frame — it only moves between the
// chunks the case HOLDS (`view`): memory is released by plain assignments to ws_used all over the code, so an allocation
// may find ws_used below the chunk it last used, or find that the request fits the next chunk up.  When nothing the case
// holds can serve a request the allocation fails with CASE_OVERFLOW and ovf_need / ovf_req say what was asked for; the
// callers that can afford it (the scheduler around a mutator attempt, the pattern code) then borrow an area of a higher
// tier and repeat ONLY that attempt (ws_regrow), so nothing a case has done is run again.
EH_DEV void ws_set_view(Ctx& c, int j) { c.view = j; c.ws = c.ch_base[j]; c.ws_lo = c.ch_vstart[j]; c.ws_cap = c.ch_vend[j]; }
EH_DEV bool ws_slow(Ctx& c, uint64_t need, int site) {
  int j = c.nchunk;
  while (j > 0 && c.ws_used < c.ch_vstart[j]) j--;
  for (; j <= c.nchunk; j++) {
    uint64_t at = c.ws_used > c.ch_vstart[j] ? c.ws_used : c.ch_vstart[j];
    if (at + need <= c.ch_vend[j]) { c.ws_used = at; ws_set_view(c, j); return true; }
  }
  // (what the top of the chunk holds — lex tables made above the caller's mark — is part of what was asked for)
  const int k = c.nchunk;
  EH_SET_OVERFLOW(c, site); c.ovf_need = c.ws_used + need + (c.ch_vstart[k] + c.p->pool_cap[c.ch_tier[k]] - c.ch_vend[k]); c.ovf_req = need;
  return false;
}
EH_DEV uint64_t ws_max_request(const Ctx& c) { return c.p->pool_cap[c.p->ntiers]; }   // the largest single allocation a case can get
EH_DEV bptr ws_alloc(Ctx& c, uint64_t n) {
  uint64_t need = (n + 15) & ~(uint64_t)15;
  if ((c.ws_used < c.ws_lo || c.ws_used + need > c.ws_cap) && !ws_slow(c, need, 102)) return nullptr;
  bptr p = c.ws + c.ws_used;
  c.ws_used += need;
  if (c.ws_used > c.ws_peak) c.ws_peak = c.ws_used;
  return p;
}
// from the TOP of the chunk (tables that must survive candidate discards: the lex caches)
EH_DEV bptr ws_alloc_top(Ctx& c, uint64_t n) {
  uint64_t need = (n + 15) & ~(uint64_t)15;
  if ((c.ws_used < c.ws_lo || c.ws_used + need > c.ws_cap) && !ws_slow(c, need, 501)) return nullptr;
  c.ws_cap -= need; c.ch_vend[c.view] = c.ws_cap; c.ws_top += need;
  return c.ws + c.ws_cap;
}
// ws_used = mark, and borrowed areas that are empty now go back to the pool.  Not for leaf functions.
__device__ __noinline__ void ws_release_to(Ctx&, uint64_t mark) {
  Ctx& c = g_ctx;
  c.ws_used = mark;
  while (c.nchunk > 0 && mark <= c.ch_vstart[c.nchunk]) {
    const int k = c.nchunk;
    const EH_G KParams& p = *c.p;
    bptr lo = c.ch_base[k] + c.ch_vstart[k]; bptr hi = lo + p.pool_cap[c.ch_tier[k]];
    for (int d = 0; d < LEX_LEVELS; d++) {                           // lex tables and lexed blocks inside the chunk: forget them
      EH_G LexCache* lc = (EH_G LexCache*)(c.aux + AUX_LEXCACHE) + d;
      uint64_t tab = uni64((uint64_t)lc->tab), key = c.lex_ptr[d];
      bool tin = uni(lc->tcap) != 0 && tab >= (uint64_t)lo && tab < (uint64_t)hi, kin = key >= (uint64_t)lo && key < (uint64_t)hi;
      if (tin || kin) { c.lex_ptr[d] = 0; if (EH_LANE == 0) { lc->n = -1; if (tin) lc->tcap = 0; } }
    }
    wave_sync();
    pool_push(p, c.ch_tier[k], c.ch_area[k]);
    c.nchunk = k - 1;
  }
  int j = c.nchunk;
  while (j > 0 && mark < c.ch_vstart[j]) j--;
  ws_set_view(c, j);
}
// An allocation above `mark` failed (CASE_OVERFLOW with ovf_need set) and everything above mark has been given up: borrow
// an area of a higher tier, large enough for what had been asked for above mark when it failed plus a quarter (allocations
// mostly come before the work, so guessing low costs little), and make it the top chunk starting at mark.  Waits for an area if the tier is out.  *last_tier: the tier the
// previous repetition of the same attempt was given (0: none) — the next one is strictly higher, so the repetitions end.
// true: status is CASE_OK again and the caller repeats its attempt; false: CASE_OVERFLOW stands for good (ovf_need = 0,
// so that callers further out do not try again): request above the largest area, or the case is already there.
__device__ __noinline__ bool ws_regrow(Ctx&, uint64_t mark, int* last_tier) {
  Ctx& c = g_ctx;
  const EH_G KParams& p = *c.p;
  const uint64_t asked = c.ovf_need > mark ? c.ovf_need - mark : c.ovf_req, req = c.ovf_req;
  const int site = c.ovf_line;
  ws_release_to(c, mark);
  int tt = c.ch_tier[c.nchunk] + 1;
  if (tt <= *last_tier) tt = *last_tier + 1;
  while (tt < p.ntiers && p.pool_cap[tt] < asked + asked / 4) tt++;
  if (tt > p.ntiers || c.nchunk + 1 >= MAX_CHUNKS || req > p.pool_cap[tt] || asked > p.pool_cap[tt]) { EH_SET_OVERFLOW(c, site); return false; }
  wave_sync();
  const uint32_t area = pool_pop(p, tt);
  const int k = c.nchunk + 1;
  c.ch_vstart[k] = mark; c.ch_vend[k] = mark + p.pool_cap[tt]; c.ch_tier[k] = tt; c.ch_area[k] = area;
  c.ch_base[k] = (bptr)((uintptr_t)(p.pool_base[tt] + (uint64_t)area * p.pool_stride[tt]) - (uintptr_t)mark);
  c.nchunk = k;
  ws_set_view(c, k);
  c.status = CASE_OK;
  *last_tier = tt;
  return true;
}
// ws_alloc for code that is not inside a mutator attempt (patterns, generators): grows the work area on the spot
__device__ __noinline__ bptr ws_alloc_grow(Ctx&, uint64_t n) {
  Ctx& c = g_ctx;
  int last_tier = 0;
  for (;;) {
    bptr q = ws_alloc(c, n);
    if (q || c.status != CASE_OVERFLOW || c.ovf_need == 0) return q;
    if (!ws_regrow(c, c.ws_used, &last_tier)) return nullptr;
  }
}
EH_DEV Blk blk_load(const EH_G Blk* t, int i) {
  Blk b = t[i];
  b.ptr = uni64(b.ptr); b.len = uni(b.len); b.aux = 0;
  return b;
}
EH_DEV void blk_store(EH_G Blk* t, int i, uint64_t ptr, uint32_t len) {
  if (EH_LANE == 0) { t[i].ptr = ptr; t[i].len = len; t[i].aux = 0; }
}

// random_block/1 (erlamsa_rnd.erl:165,173-174) / random_numbers(256, N) (:178-183): N draws of
// rand(256), list built by prepending => draw j (0-based) lands at byte N-1-j.  Lane-parallel.
EH_DEV void random_block_rev(Ctx& c, bptr dst, uint32_t n) {
  const int l = EH_LANE;
  for (uint32_t base = 0; base < n; base += 64) {
    uint32_t chunk = n - base < 64 ? n - base : 64;
    if ((uint32_t)l < chunk) {
      double u = rng_peek(c.rng, (uint32_t)l + 1);
      dst[n - 1 - (base + l)] = (uint8_t)(uint32_t)(u * 256.0);
    }
    rng_skip(c.rng, chunk);
  }
}

// ---------------------------------------------------------------------------------------------
// Single-byte mutators  (erlamsa_mutations.erl:56-61,176-223) and UTF-8 (:1081-1099)
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ int muta_byte(Ctx&, int fn) {
  EH_CTX;
  Blk h = blk_load(c.bl, c.cur);
  cbptr src = (cbptr)h.ptr;
  uint32_t L = h.len;
  uint32_t p = rng_rand(c.rng, L);
  int d = rng_delta(c.rng);
  uint32_t ins_idx = 0;
  if (fn == M_UI) ins_idx = rng_rand(c.rng, (uint32_t)c_funny.n);  // rand_elem(funny_unicode()) is drawn even for <<>>
  c.r_kind = R_SAME;
  if (L == 0) return d;  // edit_byte_vector(<<>>, _, _) -> <<>>
  uint32_t b = uni(src[p]);
  uint32_t nb0 = 0, nb1 = 0;  // replacement bytes
  uint32_t repl = 1;          // number of bytes that replace byte p
  switch (fn) {
    case M_BD: repl = 0; break;
    case M_BEI: nb0 = (b + 1) & 255; break;
    case M_BED: nb0 = (b - 1) & 255; break;
    case M_BR: nb0 = b; nb1 = b; repl = 2; break;
    case M_BF: nb0 = b ^ (1u << rng_rand(c.rng, 8)); break;
    case M_BI: nb0 = rng_rand(c.rng, 256); nb1 = b; repl = 2; break;
    case M_BER: nb0 = rng_rand(c.rng, 256); break;
    case M_UW:
      if (b == (b & 0x3f)) { nb0 = 0xC0; nb1 = b | 0x80; repl = 2; } else return d;  // unchanged
      break;
    case M_UI: break;
  }
  if (fn == M_UI) {
    uint32_t il = c_funny.v[ins_idx][0];
    bptr dst = ws_alloc(c, (uint64_t)L + il);
    if (!dst) return d;
    wave_copy(dst, src, p + 1);
    if ((uint32_t)EH_LANE < il) dst[p + 1 + EH_LANE] = c_funny.v[ins_idx][1 + EH_LANE];
    wave_copy(dst + p + 1 + il, src + p + 1, L - p - 1);
    c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = L + il;
    return d;
  }
  if (repl == 1 && nb0 == b) return d;  // same byte value: hd(Mll) == hd(Ll)
  uint32_t nl = L - 1 + repl;
  bptr dst = ws_alloc(c, nl);
  if (!dst) return d;
  wave_copy(dst, src, p);
  if (EH_LANE == 0) { if (repl >= 1) dst[p] = (uint8_t)nb0; if (repl == 2) dst[p + 1] = (uint8_t)nb1; }
  wave_copy(dst + p + repl, src + p + 1, L - p - 1);
  c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = nl; c.r_changed = 1;   // length or the byte value differs
  return d;
}

// ---------------------------------------------------------------------------------------------
// Wave-wide bitonic sort of n_pow2 (hi,lo) 128-bit keys, ascending by hi then lo.  Padding
// entries carry hi = ~0.
// ---------------------------------------------------------------------------------------------
struct Key2 { uint64_t hi, lo; };
EH_DEV bool key2_gt(const Key2& a, const Key2& b) { return a.hi > b.hi || (a.hi == b.hi && a.lo > b.lo); }
EH_DEV void wave_sort_key2(EH_G Key2* k, uint32_t n_pow2) {
  const int l = EH_LANE;
  const uint32_t half = n_pow2 >> 1;
  for (uint32_t size = 2; size <= n_pow2; size <<= 1) {
    for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
      wave_sync();
      uint32_t t = l;
      for (; t + 192 < half; t += 256) {   // 4 compare-exchanges (8 loads) in flight per lane
        uint32_t i0 = 2 * t - (t & (stride - 1)), i1 = 2 * (t + 64) - ((t + 64) & (stride - 1));
        uint32_t i2 = 2 * (t + 128) - ((t + 128) & (stride - 1)), i3 = 2 * (t + 192) - ((t + 192) & (stride - 1));
        Key2 a0 = k[i0], b0 = k[i0 + stride], a1 = k[i1], b1 = k[i1 + stride];
        Key2 a2 = k[i2], b2 = k[i2 + stride], a3 = k[i3], b3 = k[i3 + stride];
        if (key2_gt(a0, b0) == ((i0 & size) == 0)) { k[i0] = b0; k[i0 + stride] = a0; }
        if (key2_gt(a1, b1) == ((i1 & size) == 0)) { k[i1] = b1; k[i1 + stride] = a1; }
        if (key2_gt(a2, b2) == ((i2 & size) == 0)) { k[i2] = b2; k[i2 + stride] = a2; }
        if (key2_gt(a3, b3) == ((i3 & size) == 0)) { k[i3] = b3; k[i3 + stride] = a3; }
      }
      for (; t < half; t += 64) {
        uint32_t i = 2 * t - (t & (stride - 1));  // lower index of the pair
        uint32_t j = i + stride;
        bool up = ((i & size) == 0);
        Key2 a = k[i], b = k[j];
        if (key2_gt(a, b) == up) { k[i] = b; k[j] = a; }
      }
    }
  }
  wave_sync();
}

// ---------------------------------------------------------------------------------------------
// Multi-byte mutators (erlamsa_mutations.erl:232-318)
// ---------------------------------------------------------------------------------------------
__device__ __noinline__ int muta_seq(Ctx&, int fn, int mask_fun) {
  EH_CTX;
  Blk hb = blk_load(c.bl, c.cur);
  cbptr src = (cbptr)hb.ptr;
  uint32_t B = hb.len;
  c.r_kind = R_SAME;
  if (B == 0) return -1;                                   // Self([<<>>|BTail], Meta)
  uint32_t S = rng_rand(c.rng, B);
  uint32_t Lp = rng_range(c.rng, 1, (int64_t)B - S + 1);
  cbptr P = src + S;
  uint32_t tl = B - S - Lp;
  const int l = EH_LANE;
  switch (fn) {
    case M_SD: {                                           // :273-276
      bptr dst = ws_alloc(c, B - Lp);
      if (dst) { wave_copy(dst, src, S); wave_copy(dst + S, P + Lp, tl); c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = B - Lp; c.r_changed = 1; }
      break;
    }
    case M_SR: {                                           // :263-270
      uint32_t n = rng_log(c.rng, 10); if (n < 2) n = 2;
      uint64_t nl = (uint64_t)S + (uint64_t)Lp * n + tl;
      bptr dst = nl > 0xFFFFFFFFull ? (EH_SET_OVERFLOW(c, 103), c.ovf_req = ~0ull, nullptr) : ws_alloc(c, nl);
      if (dst) {
        wave_copy(dst, src, S);
        wave_fill_periodic(dst + S, P, Lp, (uint64_t)Lp * n);
        wave_copy(dst + S + (uint64_t)Lp * n, P + Lp, tl);
        c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = (uint32_t)nl; c.r_changed = 1;
      }
      break;
    }
    case M_SP: {                                           // :253-260 + random_permutation erlamsa_rnd.erl:190-196
      bptr dst = ws_alloc(c, B);
      if (!dst) break;
      wave_copy(dst, src, S);
      wave_copy(dst + S + Lp, P + Lp, tl);
      if (Lp == 2) {
        uint32_t sw = rng_rand(c.rng, 2);
        if (l == 0) { dst[S] = sw == 1 ? P[1] : P[0]; dst[S + 1] = sw == 1 ? P[0] : P[1]; }
      } else {
        // keys {uniform(), X}: lists:sort/1 ascending by float, ties by byte.  IEEE bits of a
        // non-negative double order like the value.
        uint32_t np2 = 1; while (np2 < Lp) np2 <<= 1;
        uint64_t mark = c.ws_used;
        EH_G Key2* keys = (EH_G Key2*)ws_alloc(c, (uint64_t)np2 * sizeof(Key2));
        if (!keys) break;
        for (uint32_t base = 0; base < np2; base += 64) {
          uint32_t idx = base + l;
          if (idx < Lp) {
            double u = rng_peek(c.rng, (uint32_t)l + 1);
            keys[idx].hi = (uint64_t)__double_as_longlong(u); keys[idx].lo = P[idx];
          } else if (idx < np2) { keys[idx].hi = ~(uint64_t)0; keys[idx].lo = 0; }
          if (base < Lp) rng_skip(c.rng, Lp - base < 64 ? Lp - base : 64);
        }
        wave_sort_key2(keys, np2);
        for (uint32_t i = l; i < Lp; i += 64) dst[S + i] = (uint8_t)keys[i].lo;
        wave_sync();
        c.ws_used = mark;  // release the key array (dst was allocated before it)
      }
      c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = B;
      break;
    }
    case M_SNAND:
    case M_SRND: {
      // randmask/2 + randmask_loop/5 (:281-293).  Draw stream after MaskProb and the first
      // rand_occurs: per byte one "occurs" draw (for the NEXT byte; argument order of the
      // recursive call) and, when the byte's own flag is set, one mask draw.  Over the draw index
      // this is a 4-state automaton (O0,O1 = occurs-draw with the current flag 0/1; M0,M1 = mask
      // draw carrying the next flag), so 64 draws are resolved at once: every lane evaluates its
      // uniform by jump-ahead, the transition maps are composed with a shuffle prefix scan, and
      // ballots assign draws to bytes.
      bptr dst = ws_alloc(c, B);
      if (!dst) break;
      wave_copy(dst, src, B);
      uint32_t prob = rng_erand(c.rng, 100);
      bool occ = rng_occurs(c.rng, prob, 100);
      uint32_t byte0 = 0;                    // first byte whose occurs-draw is still pending
      uint32_t st = occ ? 1u : 0u;           // automaton state before the next draw: 0=O0 1=O1 2=M0 3=M1
      wave_sync();
      while (byte0 < Lp) {
        double u = rng_peek(c.rng, (uint32_t)l + 1);
        uint32_t n100 = (uint32_t)(u * 100.0);
        uint32_t F = prob == 1 ? (n100 != 0 ? 1u : 0u) : (n100 < prob ? 1u : 0u);
        // transition map of this draw, 2 bits per source state: O0->O_F, O1->M_F, M0->O0, M1->O1
        uint32_t tm = (F) | ((2u + F) << 2) | (0u << 4) | (1u << 6);
        // inclusive scan of map composition: comp(a then b)[s] = b[a[s]]
        uint32_t inc = tm;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          uint32_t prev = (uint32_t)__shfl_up((int)inc, d);
          if (l >= d) {
            uint32_t r = 0;
#pragma unroll
            for (int sidx = 0; sidx < 4; sidx++) { uint32_t mid = (prev >> (2 * sidx)) & 3u; r |= ((inc >> (2 * mid)) & 3u) << (2 * sidx); }
            inc = r;
          }
        }
        uint32_t excl = (uint32_t)__shfl_up((int)inc, 1);
        uint32_t my_state = l == 0 ? st : ((excl >> (2 * st)) & 3u);   // state BEFORE my draw
        bool is_o = my_state < 2;
        unsigned long long omask = __ballot(is_o);
        uint32_t o_before = (uint32_t)__popcll(omask & ((1ull << l) - 1));
        // byte index served by my draw: an O-draw belongs to byte byte0+o_before; an M-draw to the
        // byte of the preceding O-draw
        uint32_t bidx = is_o ? byte0 + o_before : byte0 + o_before - 1;
        // the stream ends after the O-draw of byte Lp-1 and, if that byte's flag is set, its M-draw
        bool valid = is_o ? (bidx < Lp) : (bidx < Lp);
        // an M-draw at the very start of a chunk (state M*) belongs to byte0-1 which is < Lp by construction
        unsigned long long vmask = __ballot(valid);
        uint32_t nvalid = vmask == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~vmask);   // valid draws form a prefix
        if (!is_o && (uint32_t)l < nvalid) {
          uint32_t old = P[bidx];
          uint32_t nbv;
          if (mask_fun == 3) nbv = (uint32_t)(u * 256.0);
          else { uint32_t m = 1u << (uint32_t)(u * 8.0); nbv = mask_fun == 0 ? (old & ~m) : (mask_fun == 1 ? (old | m) : (old ^ m)); }
          dst[S + bidx] = (uint8_t)nbv;
        }
        // advance: state after the last valid draw, bytes whose O-draw happened
        uint32_t last = nvalid - 1;
        uint32_t new_st = (uint32_t)__builtin_amdgcn_readlane((int)((inc >> (2 * st)) & 3u), (int)last);
        uint32_t o_done = (uint32_t)__popcll(omask & (nvalid == 64 ? ~0ull : ((1ull << nvalid) - 1)));
        rng_skip(c.rng, nvalid);
        byte0 += o_done;
        st = uni(new_st);
        if (nvalid < 64) break;
      }
      // If the loop ended exactly at a chunk boundary with the last byte's M-draw still pending
      // (state M*), it is consumed here.
      if (byte0 >= Lp && st >= 2) {
        double u = rng_uniform(c.rng);
        if (l == 0) {
          uint32_t old = P[Lp - 1];
          uint32_t nbv;
          if (mask_fun == 3) nbv = (uint32_t)(u * 256.0);
          else { uint32_t m = 1u << (uint32_t)(u * 8.0); nbv = mask_fun == 0 ? (old & ~m) : (mask_fun == 1 ? (old | m) : (old ^ m)); }
          dst[S + Lp - 1] = (uint8_t)nbv;
        }
        st = 0;
      }
      c.r_kind = R_NEW; c.r_ptr = dst; c.r_len = B;
      break;
    }
  }
  wave_sync();
  return rng_delta(c.rng);
}

// ---------------------------------------------------------------------------------------------
// Scheduler: weighted_permutations + mux_fuzzers_loop  (erlamsa_mutations.erl:1244-1280)
// ---------------------------------------------------------------------------------------------
EH_DEV uint32_t em_score(uint32_t m) { return m & 0xFF; }
EH_DEV uint32_t em_fn(uint32_t m) { return (m >> 8) & 0xFF; }
EH_DEV uint32_t em_name(uint32_t m) { return (m >> 16) & 0xFF; }
EH_DEV uint32_t em_mask(uint32_t m) { return (m >> 24) & 0xFF; }
EH_DEV uint32_t em_pack(uint32_t score, uint32_t fn, uint32_t name, uint32_t mask) { return score | (fn << 8) | (name << 16) | (mask << 24); }

// Cost weight of one byte handed to mutator `fn` in the work budget (measured cycles per byte, rounded
// to powers of two): parsers and per-byte-draw mutators are an order of magnitude dearer than byte
// movers, and the budget stands in for a time limit.  oracle/oracle.cpp mirrors this table.
EH_DEV uint32_t work_weight(uint32_t fn) {
  switch (fn) {
    case M_SGM: case M_JS: case M_AB: case M_AD: case M_TR2: case M_TD: case M_TS1: case M_TR: case M_TS2:
    case M_SNAND: case M_SRND: case M_B64: case M_URI: return 8;
    case M_NUM: return 4;
    case M_FT: case M_FN: case M_FO: case M_ZIP: return 64;       // (zip: every file of the archive is inflated and deflated again on one lane)
    default: return 1;
  }
}
// The device codecs of the container patterns (cp: gunzip / inflate and gzip / deflate; ar: zip:foldl and zip:create) run on one
// lane at a few thousand cycles per byte (profiles/r05_zlib_rate.json): their bytes count towards the work budget at the weight of
// the dearest mutators.  false: the budget is spent (status set).  oracle/oracle.cpp EngineGuard::codec mirrors the sites.
constexpr uint32_t CODEC_WEIGHT = 64;
EH_DEV bool codec_work(Ctx& c, uint64_t bytes) {
  if (!c.work_budget) return true;
  c.work += bytes * CODEC_WEIGHT;
  if (c.work > c.work_budget) { c.status = CASE_BUDGET; return false; }
  return true;
}

EH_DEV int run_mutator_ext(Ctx& c, uint32_t fn, uint32_t mask);   // eh_engine.hip: text/tree/... mutators
EH_DEV int run_mutator(Ctx& c, uint32_t fn, uint32_t mask) {
  switch (fn) {
    case M_BD: case M_BEI: case M_BED: case M_BF: case M_BI: case M_BER: case M_BR: case M_UW: case M_UI:
      return muta_byte(c, (int)fn);
    case M_SP: case M_SR: case M_SD: case M_SNAND: case M_SRND:
      return muta_seq(c, (int)fn, (int)mask);
    case M_NIL: c.r_kind = R_SAME; return -1;               // nomutation :1104-1105
    default: return run_mutator_ext(c, fn, mask);
  }
}

// Applies the candidate to the block list: bl[cur] is replaced (possibly by several flush_bvecs
// chunks) and, for fn, bl[cur+1] is dropped.
EH_DEV void commit_result(Ctx& c) {
  uint32_t chunks1 = 1;
  if (c.r_flush) chunks1 = c.r_len / AVG_BLOCK_SIZE + 1;    // flush_bvecs: k full blocks + remainder (may be <<>>)
  uint32_t chunks2 = c.r2 ? c.r2_len / AVG_BLOCK_SIZE + 1 : 0;
  uint32_t chunks = chunks1 + chunks2;
  int drop = 1 + (c.r_drop_next && c.cur + 1 < c.nb ? 1 : 0);
  int tail = c.nb - c.cur - drop;
  int newnb = c.cur + (int)chunks + tail;
  if (newnb > MAX_BLOCKS) { EH_SET_OVERFLOW(c, 104); return; }
  if (chunks != (uint32_t)drop) {                           // move the tail
    const int l = EH_LANE;
    for (int i = l; i < tail; i += 64) c.bl2[i] = c.bl[c.cur + drop + i];
    wave_sync();
    for (int i = l; i < tail; i += 64) c.bl[c.cur + (int)chunks + i] = c.bl2[i];
  }
  for (uint32_t k = EH_LANE; k < chunks; k += 64) {
    bool second = k >= chunks1;
    uint32_t kk = second ? k - chunks1 : k, nchk = second ? chunks2 : chunks1;
    bptr base = second ? c.r2_ptr : c.r_ptr; uint32_t blen = second ? c.r2_len : c.r_len;
    bool fl = second || c.r_flush;
    uint64_t off = (uint64_t)kk * AVG_BLOCK_SIZE;
    uint32_t len = fl ? (kk + 1 < nchk ? AVG_BLOCK_SIZE : blen - (uint32_t)off) : blen;
    c.bl[c.cur + k].ptr = (uint64_t)(base + off); c.bl[c.cur + k].len = len; c.bl[c.cur + k].aux = 0;
  }
  c.nb = newnb;
  wave_sync();
}

// The Meta entry a mutator conses itself (erlamsa_mutations.erl:162-168 muta_num, :180 {Name, D}, :235/:248 {Name, -1 | BSize}, :360/:376
// {Name, 1} when the block is lines, :390/:402/:421 fuse, :598 ascii when stringy, :922/:968/:1021 tree when it ran, :1089/:1099 utf8,
// :1105, :1143, :1160-1162); b64 / uri / sgm / js write theirs where they happen.  oracle.cpp own_meta is the same table.
struct alignas(16) OwnTab { uint8_t v[M_COUNT]; };
constexpr OwnTab own_atoms() {
  OwnTab t{};
  const int a[M_COUNT] = {0, 0, AT_sed_utf8_widen, AT_sed_utf8_insert, AT_ascii_bad, AT_ascii_delimeter, AT_tree_dup, AT_tree_del, AT_muta_num, AT_tree_swap_one,
                          AT_tree_stutter, AT_tree_swap_two, AT_byte_drop, AT_byte_inc, AT_byte_dec, AT_byte_flip, AT_byte_insert, AT_byte_swap_random, AT_byte_repeat,
                          AT_seq_perm, AT_seq_repeat, AT_seq_drop, AT_seq_randmask, AT_seq_randmask, AT_line_del, AT_line_del_seq, AT_line_dup, AT_line_clone,
                          AT_line_repeat, AT_line_swap, AT_line_perm, AT_list_ins, AT_list_replace, AT_fuse_this, AT_fuse_next, AT_fuse_old, AT_muta_len, 0, 0,
                          AT_muta_zippath, AT_nomutation};
  for (int i = 0; i < M_COUNT; i++) t.v[i] = (uint8_t)a[i];
  return t;
}
__constant__ OwnTab c_own_atom = own_atoms();
__device__ __noinline__ void own_meta_emit(uint32_t fn, int delta, uint32_t hlen) {
  Ctx& c = g_ctx;
  const int nm = (int)c_own_atom.v[fn];
  switch (fn) {
    case M_BD: case M_BEI: case M_BED: case M_BF: case M_BI: case M_BER: case M_BR: case M_UW: case M_UI:
    case M_FT: case M_FN: case M_FO: case M_LEN: case M_ZIP: case M_NIL:
      tr_ai(c, nm, delta); break;
    case M_SP: case M_SR: case M_SD: case M_SNAND: case M_SRND:
      tr_ai(c, nm, hlen == 0 ? -1 : (int64_t)hlen); break;
    case M_NUM: tr_ai(c, nm, c.m_aux); break;
    case M_AB: case M_AD: if (c.m_aux == 1) tr_ai(c, nm, delta); break;
    case M_LD: case M_LDS: case M_LR2: case M_LRI: case M_LR: case M_LS: case M_LP: case M_LIS: case M_LRS:
    case M_TR2: case M_TD: case M_TS1: case M_TS2: case M_TR:
      if (delta == 1) tr_ai(c, nm, 1);
      break;
    default: break;
  }
}
EH_DEV void own_meta(Ctx& c, uint32_t fn, int delta, uint32_t hlen) { if (__builtin_expect(c.trace != nullptr, 0)) own_meta_emit(fn, delta, hlen); }

// Nested scheduler calls WITHOUT device recursion.  base64_mutator, sgml_mutate and json_mutate run Muta([Bin], []) of a fresh
// mutator list on pieces of their block (erlamsa_mutations.erl:667-670, erlamsa_sgml.erl:669-681, erlamsa_json.erl:633-639), and
// the mutators down there may do the same - six levels deep at most.  Until round 5 that was a call chain through the scheduler
// (muta_X -> nested_fuzz -> mux_fuzzers -> muta_X ...): device recursion, a stack the compiler cannot bound, hipLimitStackSize.
// Now a mutator that wants a nested run RETURNS: it leaves the block, the nested table (one entry per lane) and its own loop state
// in a MuFrame in work memory and sets call_req; the scheduler below parks the level it is at in an MxFrame, runs the nested level
// in the same loop, and calls the mutator again with mu_phase = 1 and call_nres = the number of result blocks (at c.bl[c.nb ..),
// or -1 with the status set).  The call graph has no cycle and the kernel's stack is what the assembler adds up.
constexpr uint32_t NEST_SAVE = 720;                                          // StState x 2 + FoState (aux + 0 .. 720)
struct MuFrame {                // a nesting mutator's locals across its nested runs + the table it wants run (v[23]: the MxFrame its nested runs park the level in)
  uint64_t v[24];
  uint32_t pri[64], meta[64];
};
struct MxFrame {                // a level of the scheduler that waits for a nested one
  uint32_t e_pri[64], e_meta[64], rank[64];
  uint32_t aux_save[NEST_SAVE / 4];   // lis / lrs / fo of the waiting level (aux + 0 .. 720): the nested table starts from fresh ones
  uint32_t state[64];                 // MxState as it was (eh_device.h g_mx)
};

// One call of the mux_fuzzers closure on the list bl[cur..nb) - and every nested one it leads to.
// The level in hand: what is the same for all lanes lives in LDS (as locals these two dozen values were live across every call of a
// mutator and spilled: 200 scratch loads in the scheduler's loop, a measured 8 % of all wave cycles); lane tables and ranks stay in registers.
struct MxState {
  Rng rng0; uint64_t work0, mark, pt0; Blk h0;
  int32_t nfs, r, tried, j, last_tier, delta; uint32_t meta, fn, ntrace0, dropped, stateful, pad;
  // what a level that waits for a nested one goes back to (filled in when it is parked; the whole struct is parked and brought back
  // by one lane-parallel copy)
  uint64_t up, mu; int32_t lastm0, cur0, nb0, nfs0; uint32_t tr_base0, pad2;
};
static_assert(sizeof(MxState) % 4 == 0 && sizeof(MxState) <= 256, "MxState is copied one word per lane");
__shared__ MxState g_mx;
__device__ __noinline__ void mux_fuzzers(Ctx&, LaneTab& lt) {             // (ONE instance: the pattern code calls it from three places)
  EH_CTX;
  const int l = EH_LANE;
  EH_G MxFrame* up = nullptr;                                                // the level that waits for the one in hand (nullptr: the pattern's call)
  uint32_t e_pri = lt.e_pri, e_meta = lt.e_meta, rank = 0;
  MxState& m = g_mx;
  enum { S_ENTER, S_ATTEMPT, S_RUN, S_AFTER, S_FINISH, S_LEAVE };
  int st = S_ENTER;
  for (;;) {
    if (st == S_ENTER) {
      if (c.nb - c.cur == 1 && blk_load(c.bl, c.cur).len == 0) { st = S_LEAVE; continue; }   // L([<<>>], Meta)
      if (c.nb - c.cur <= 0) { c.status = CASE_CRASHED; st = S_LEAVE; continue; }
      m.nfs = c.nfs;
      // --- weighted_permutations: key_i = rand(trunc(Score*Pri)) in list order, lane-parallel jump-ahead
      uint32_t nkey = (l < m.nfs) ? em_score(e_meta) * e_pri : 0;
      unsigned long long drawing = __ballot(nkey > 0);
      uint32_t my_idx = (uint32_t)__popcll(drawing & ((1ull << l) - 1));
      uint32_t ndraw = (uint32_t)__popcll(drawing);
      uint32_t key = 0;
      if (nkey > 0) key = (uint32_t)(rng_peek(c.rng, my_idx + 1) * (double)nkey);
      rng_skip(c.rng, ndraw);
      // --- stable descending sort => rank per lane
      rank = 0;
      for (int q = 0; q < m.nfs; q++) {
        uint32_t kq = (uint32_t)__builtin_amdgcn_readlane((int)key, q);
        rank += (kq > key || (kq == key && q < l)) ? 1u : 0u;
      }
      // A case that has been running for long decides when its pass ends (one wavefront, seconds): from 20 M cycles on its wavefront
      // issues ahead of the other wavefront of its SIMD (s_setprio; the ticket loop puts it back when the case is done).
#ifndef HIPEMU
      if (__builtin_readcyclecounter() - c.t_case > 20000000ull) __builtin_amdgcn_s_setprio(3);
#endif
      // --- mux_fuzzers_loop
      m.tried = 0; m.dropped = false; m.r = 0;
      m.h0 = blk_load(c.bl, c.cur);
      st = S_ATTEMPT;
    }
    if (st == S_ATTEMPT) {
      if (m.r >= m.nfs) { st = S_FINISH; continue; }
      if (m.h0.len > ABSMAX_BINARY_BLOCK) { m.dropped = true; tr_ai(c, AT_skipped_big, (int64_t)m.h0.len); st = S_FINISH; continue; }   // [{skipped_big, byte_size(H)} | Meta] :1269-1270
      unsigned long long who = __ballot(l < m.nfs && rank == (uint32_t)m.r);
      m.j = (int)__builtin_ctzll(who);
      m.meta = (uint32_t)__builtin_amdgcn_readlane((int)e_meta, m.j);
      m.fn = em_fn(m.meta);
      c.r_kind = R_SAME; c.r_flush = 0; c.r_drop_next = 0; c.r_changed = 0; c.r2 = 0;
      m.mark = c.ws_used;
      // work budget: the reference kills a worker after maxrunningtime and records <<>>
      // (erlamsa_main.erl:211-220); the engine's deterministic analogue counts bytes
      if (c.work_budget) {                                                          // optional (eh_options.max_case_work, 0 = off)
        c.work += (uint64_t)m.h0.len * work_weight(m.fn);
        if (c.work > c.work_budget) { c.status = CASE_BUDGET; st = S_LEAVE; continue; }
      }
#ifdef EH_PROF
      m.pt0 = __builtin_readcyclecounter();
#endif
      // An attempt that runs out of work memory is repeated — that attempt only, from the same PRNG state — after the case
      // has borrowed a larger area (ws_regrow).  lis / lrs update their store before they allocate: it is put back as well.
      m.rng0 = c.rng; m.work0 = c.work; m.ntrace0 = c.ntrace;
      m.stateful = m.fn == M_LIS || m.fn == M_LRS;
      if (m.stateful) { cwptr ax = (cwptr)c.aux + (m.fn == M_LRS ? ST_STATE_WORDS : 0); for (int i = l; i < ST_STATE_WORDS; i += 64) g_st_save[i] = ax[i]; }
      m.last_tier = 0;
      c.mu_phase = 0; c.mu = nullptr;
      st = S_RUN;
    }
    if (st == S_RUN) {
      c.call_req = 0;
      m.delta = run_mutator(c, m.fn, em_mask(m.meta));
      if (c.call_req && c.status == CASE_OK) {
        // ---- the mutator wants Muta([Bin], []) run: park this level, set the nested one up (the inner list [Bin] lives above the
        // outer block list; lis / lrs / fo of the inner table start from their initial state, as the closures of a fresh
        // mutators_mutator/1 do - the outer states are parked in the work area for the duration of the nested level)
        EH_G MxFrame* f = nullptr;
        if (c.depth >= MAX_NEST || c.nb + 2 > MAX_BLOCKS) EH_SET_OVERFLOW(c, 301);
        else {                                                                 // (one frame per attempt, however many nested runs it makes)
          f = (EH_G MxFrame*)uni64(c.mu->v[23]);
          if (!f) { f = (EH_G MxFrame*)ws_alloc(c, sizeof(MxFrame)); if (f && l == 0) c.mu->v[23] = (uint64_t)f; }
        }
        if (f) {
          EH_G uint32_t* save = f->aux_save;
          f->e_pri[l] = e_pri; f->e_meta[l] = e_meta; f->rank[l] = rank;
          EH_G uint32_t* ax = (EH_G uint32_t*)c.aux;
          for (uint32_t i = l; i < NEST_SAVE / 4; i += 64) save[i] = ax[i];
          m.up = (uint64_t)up; m.mu = (uint64_t)c.mu; m.lastm0 = c.lastm; m.tr_base0 = c.tr_base; m.cur0 = c.cur; m.nb0 = c.nb; m.nfs0 = c.nfs;
          lanes_sync();
          if ((uint32_t)l < sizeof(MxState) / 4) f->state[l] = ((const uint32_t*)&m)[l];
          wave_sync();
          // (StState[0].count, StState[1].count, FoState::has: eh_text.h, eh_fuse.h - the same offsets the reclaim test below reads)
          if (l == 0) { ((EH_G int32_t*)c.aux)[0] = 0; ((EH_G int32_t*)(c.aux + 4 * ST_STATE_WORDS))[0] = 0; ((EH_G uint32_t*)(c.aux + 704))[3] = 0; }
          const int nb0 = c.nb;
          blk_store(c.bl, nb0, c.call_bin, c.call_len);
          e_pri = c.mu->pri[l]; e_meta = c.mu->meta[l];                        // the nested table, lane i = list position i
          wave_sync();
          c.cur = nb0; c.nb = nb0 + 1; c.nfs = c.call_nfs; c.depth++;
          c.tr_base = c.ntrace;                                                // Muta([Bin], []): a Meta list of its own
          c.lex_ptr[c.depth] = 0;                                              // this level's last lexed block was a temporary of an earlier call
          if (l == 0) lex_slot(c).n = -1;
          up = f;
          st = S_ENTER;
          continue;
        }
        // (no room for the frame, or nested too deep: the attempt ends like one whose mutator ran out of memory)
      }
      if (c.status == CASE_OVERFLOW && c.ovf_need != 0) {
        wave_sync();
        if (ws_regrow(c, m.mark, &m.last_tier)) {
          lex_forget_from(c, m.mark);
          c.rng = m.rng0; c.work = m.work0; c.ntrace = m.ntrace0;
          c.r_kind = R_SAME; c.r_flush = 0; c.r_drop_next = 0; c.r_changed = 0; c.r2 = 0;
          if (m.stateful) { lanes_sync(); wptr ax = (wptr)c.aux + (m.fn == M_LRS ? ST_STATE_WORDS : 0); for (int i = l; i < ST_STATE_WORDS; i += 64) ax[i] = g_st_save[i]; }
          wave_sync();
          c.mu_phase = 0; c.mu = nullptr;
          continue;                                                            // (st == S_RUN: the same attempt once more)
        }
      }
      st = S_AFTER;
    }
    if (st == S_AFTER) {
#ifdef EH_PROF
      if (l == 0) { atomicAdd(&c.p->prof[2 * m.fn], (unsigned long long)(__builtin_readcyclecounter() - m.pt0)); atomicAdd(&c.p->prof[2 * m.fn + 1], 1ull); }
#endif
      if (c.status != CASE_OK) { st = S_LEAVE; continue; }
      const uint32_t name = em_name(m.meta);
      // adjust_priority :1238-1242
      uint32_t sc = em_score(m.meta);
      if (m.delta != 0) { int ns = (int)sc + m.delta; ns = ns < 2 ? 2 : (ns > 10 ? 10 : ns); sc = (uint32_t)ns; }
      uint32_t nfn = (m.fn == M_URI) ? (uint32_t)M_B64 : m.fn;                          // :784 (sic)
      if (l == m.j) e_meta = em_pack(sc, nfn, name, em_mask(m.meta));
      m.tried++;
      bool changed = false;
      if (c.r_kind == R_NEW) {
        wave_sync();                                                                  // candidate bytes were written by other lanes
        uint32_t hd_len = c.r_flush && c.r_len >= AVG_BLOCK_SIZE ? AVG_BLOCK_SIZE : c.r_len;
        changed = c.r_changed || hd_len != m.h0.len || !wave_equal(c.r_ptr, (cbptr)m.h0.ptr, hd_len);
      }
#ifdef EH_PROF
    // work memory an attempt took (what it wrote, nearly: candidates, tables, temporaries): slot 56 attempts that failed, 57 the candidates
    // that were used, 58 what the used attempts took besides their candidate (eh_result_prof; the write traffic's breakdown)
    if (l == 0) {
      const unsigned long long took = c.ws_used > m.mark ? c.ws_used - m.mark : 0ull, cand = c.r_kind == R_NEW ? c.r_len : 0u;
      if (changed) { atomicAdd(&c.p->prof[2 * 57], cand); atomicAdd(&c.p->prof[2 * 57 + 1], 1ull); atomicAdd(&c.p->prof[2 * 58], took > cand ? took - cand : 0ull); atomicAdd(&c.p->prof[2 * 58 + 1], 1ull); }
      else { atomicAdd(&c.p->prof[2 * 56], took); atomicAdd(&c.p->prof[2 * 56 + 1], 1ull); }
    }
#endif
      own_meta(c, m.fn, m.delta, m.h0.len);                                                // the mutator's own entry is in the Meta it returns, used or failed
      tr_aa(c, changed ? AT_used : AT_failed, (int)name);                            // {used, Name} / {failed, Name} :1278-1279
      if (changed) {
      c.lastm = (int)name;
      // Reclaim work memory before committing: everything between `m.mark` and the candidate is a
      // dead temporary, and the block being replaced is dead too when it is the newest committed
      // allocation and no mutator state (lis/lrs lines, fo block) can point into it.  The
      // candidate slides down (ascending copy, dst < src) so that chains of mutations on one block
      // keep a ~1x footprint instead of growing linearly with the number of rounds.
      // The slide is a second full copy of the block plus two memory round trips, so it is only done
      // once the work area is more than 1/8 full: the typical case (a 4 KiB block, ~10 rounds) never
      // gets there and simply leaves its dead candidates behind.
      // The attempt went on in areas borrowed from the pool (it ran out of memory and was repeated): when the candidate
      // fits where the attempt began, it moves there and the areas go back at once — most borrowers are fuse calls whose
      // tables need tens of megabytes for a result of one or two.
      if (c.nchunk > 0 && m.mark <= c.ch_vstart[c.nchunk] && !c.r2) {
        int jc = c.nchunk;
        while (jc > 0 && m.mark <= c.ch_vstart[jc]) jc--;
        const uint64_t need = ((uint64_t)c.r_len + 15) & ~(uint64_t)15;
        if (m.mark + need <= c.ch_vend[jc]) {
          bptr dst = c.ch_base[jc] + m.mark;
          // (the candidate is not always up in a borrowed area: a chunk the PATTERN borrowed for its scans - pick_csum,
          // pick_simple_len - and gave back by resetting ws_used starts exactly at m.mark, and the attempt then ran in the chunk
          // below it, a few bytes above dst: overlapping ranges, which wave_copy must not be given)
          if (dst != c.r_ptr) {
            wave_sync();
            if (c.r_ptr > dst && c.r_ptr < dst + need) wave_move_down(dst, c.r_ptr, c.r_len); else wave_copy(dst, c.r_ptr, c.r_len);
            wave_sync();
          }
          c.r_ptr = dst;
          ws_release_to(c, m.mark);
          c.ws_used = m.mark + need;
          if (c.ws_used > c.ws_peak) c.ws_peak = c.ws_used;
        }
      }
      bptr lo = c.ws + m.mark;
      // A candidate of more than a few MiB stays where it is: mux_fuzzers never hands out a block above
      // ABSMAX_BINARY_BLOCK again (:1269, split_into_maxblocks), so nothing will copy it as a whole any more, and sliding
      // a 1 GiB tree-stutter result took a lone wavefront 0.7 s.  (m.mark below the chunk the candidate is in: the attempt
      // went on in the next area up; not worth a copy across areas.)
      const uint64_t vs = c.ws_lo;
      if (m.mark >= vs && c.ws_used - vs > (c.ws_cap - vs) / 8 && c.r_len <= (4u << 20) && !c.r2 && c.r_ptr >= lo && c.r_ptr + c.r_len <= c.ws + c.ws_used) {
        bptr dst = lo;
        bptr hp = (bptr)m.h0.ptr;
        bool state_refs = uni(((cwptr)c.aux)[0]) != 0 || uni(((cwptr)(c.aux + 336))[0]) != 0 || uni(((cwptr)(c.aux + 704))[3]) != 0;
        if (!state_refs && hp >= c.ws + vs && hp + ((m.h0.len + 15u) & ~15u) == lo && ((uintptr_t)hp & 15) == 0) dst = hp;
        if (dst != c.r_ptr) { wave_sync(); wave_move_down(dst, c.r_ptr, c.r_len); wave_sync(); c.r_ptr = dst; }   // candidate stores must have landed
        c.ws_used = (uint64_t)(dst - c.ws) + (((uint64_t)c.r_len + 15) & ~(uint64_t)15);
        lex_forget_from(c, (uint64_t)(dst - c.ws));                               // (H's own memory included when the candidate took its place)
      }
        commit_result(c); st = S_FINISH; continue;
      }
      if (c.nchunk > 0) ws_release_to(c, m.mark); else c.ws_used = m.mark;              // discard candidate
      lex_forget_from(c, m.mark);
      m.r++;
      st = S_ATTEMPT;
      continue;
    }
    if (st == S_FINISH) {
      // --- new list: reverse(m.tried) ++ untried (sorted order)   :1268,1270,1279
      // m.dropped (:1270): the entry at sorted position `m.tried` leaves the list.
      int newpos = l;
      if (l < m.nfs) {
        int rk = (int)rank;
        if (rk < m.tried) newpos = m.tried - 1 - rk;
        else if (!m.dropped) newpos = rk;
        else newpos = rk == m.tried ? m.nfs - 1 : rk - 1;
      }
      // ds_permute (forward): lane i sends its value to lane newpos (a bijection)
      e_meta = (uint32_t)__builtin_amdgcn_ds_permute(newpos << 2, (int)e_meta);
      e_pri = (uint32_t)__builtin_amdgcn_ds_permute(newpos << 2, (int)e_pri);
      if (m.dropped) c.nfs = m.nfs - 1;
      st = S_LEAVE;
    }
    // S_LEAVE
    if (!up) { lt.e_pri = e_pri; lt.e_meta = e_meta; return; }
    {
      // ---- a nested level is over: its result blocks are bl[cur..nb); back to the level that waits, whose mutator goes on
      EH_G MxFrame* f = up;
      c.depth--;
      const int nres = c.nb - c.cur;
      wave_sync();
      EH_G uint32_t* ax = (EH_G uint32_t*)c.aux; const EH_G uint32_t* save = f->aux_save;
      for (uint32_t i = l; i < NEST_SAVE / 4; i += 64) ax[i] = save[i];
      lanes_sync();
#ifdef HIPEMU
      for (uint32_t k = 0; k < sizeof(MxState) / 4; k++) ((uint32_t*)&m)[k] = f->state[k];   // (the emulator keeps a copy of the LDS per lane)
#else
      if ((uint32_t)l < sizeof(MxState) / 4) ((uint32_t*)&m)[l] = f->state[l];
#endif
      e_pri = f->e_pri[l]; e_meta = f->e_meta[l]; rank = f->rank[l];
      wave_sync();
      c.tr_base = m.tr_base0;
      c.cur = m.cur0; c.nb = m.nb0; c.nfs = m.nfs0; c.lastm = m.lastm0;
      c.r_kind = R_SAME; c.r_flush = 0; c.r_drop_next = 0; c.r_changed = 0; c.r2 = 0;
      c.call_nres = c.status == CASE_OK ? nres : -1;
      c.mu = (EH_G MuFrame*)m.mu; c.mu_phase = 1;
      up = (EH_G MxFrame*)m.up;
      st = S_RUN;
    }
  }
}

}  // namespace eh
