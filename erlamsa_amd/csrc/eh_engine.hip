// eh_engine.hip — kernels + C ABI of liberlamsa_hip.so (see include/erlamsa_hip.h).
//
// Kernels (gfx950 only):
//   eh_prologue_kernel: counters + argument block of a batch, per-run setup of erlamsa_main:fuzzer/1 (one wavefront)
//   eh_mutate_kernel  : persistent grid, one wavefront per case, cases pulled from a ticket
//                       counter; runs generator -> pattern -> mux_fuzzers -> mutators and
//                       writes each case's output into a bump-allocated arena.
//
// Reference call path restated here (src/ of the reference):
//   erlamsa_main.erl:125-247  fuzzer/1 (setup draw order, per-case ThreadSeed, worker body)
//   erlamsa_gen.erl:43-56,152-199  finish/1, direct_generator/2, random_stream/1, mux_generators/2
//   erlamsa_patterns.erl:45-60,146-161,265-442  split, skipper, mutate_once(_loop), od/nd/bu/co/nu, mux_patterns
//   erlamsa_mutations.erl  (see eh_device.h)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/erlamsa_hip.h"
#include "eh_otp_sort.h"
#include "eh_device.h"
#include "eh_text.h"
#include "eh_lex.h"
#include "eh_tree.h"
#include "eh_field.h"
#include "eh_fuse.h"
#include "eh_fuse2.h"
#include "eh_doc.h"
#include "eh_sgml.h"
#include "eh_json.h"
#include "eh_zlib.h"
#include "eh_zip.h"
#include "eh_comm.h"

namespace eh {


EH_DEV int run_mutator_ext(Ctx& c, uint32_t fn, uint32_t mask) {
  (void)mask;
  EH_G StState* st = (EH_G StState*)c.aux;
  switch (fn) {
    case M_LD: case M_LDS: case M_LR2: case M_LRI: case M_LR: case M_LS: case M_LP: return muta_line(c, (int)fn);
    case M_LIS: return muta_st_line(c, (int)fn, st);
    case M_LRS: return muta_st_line(c, (int)fn, st + 1);
    case M_NUM: return muta_num(c);
    case M_AB: case M_AD: return muta_ascii(c, lex_slot(c), (int)fn);
    case M_URI: return muta_uri(c, lex_slot(c));
    case M_B64: return muta_b64(c, lex_slot(c));
    case M_ZIP: return muta_zip(c);
    case M_LEN: return muta_len(c);
    case M_FT: case M_FN: case M_FO: return muta_fuse(c, (int)fn, (EH_G FoState*)(c.aux + 704));
    case M_TR2: case M_TD: case M_TS1: case M_TS2: case M_TR: return muta_tree(c, (int)fn);
    case M_SGM: return muta_sgml(c);
    case M_JS: return muta_json(c);
    default: c.status = CASE_UNSUPPORTED; c.r_kind = R_SAME; return 0;
  }
}

// =============================================================================================
// device: per-run setup  (erlamsa_main.erl:134-158)
// =============================================================================================
EH_DEV void setup_run(const DevConfig& cfg, int64_t s1, int64_t s2, int64_t s3, Rng& rng, int& gen, int& snand_mask,
                      uint32_t& e_pri, uint32_t& e_meta, int& nfs) {
  const int l = EH_LANE;
  rng.draws = 0;
  rng_seed(rng, s1, s2, s3);                                   // erlamsa_rnd:seed/1 :134
  // mutations/1 evaluates construct_sed_bytes_randmask twice: rand_elem over 3, then over 1 funs
  snand_mask = (int)rng_rand(rng, 3);                          // erlamsa_mutations.erl:311-312,1313
  (void)rng_rand(rng, 1);                                      // :1314
  // make_mutator (:1370-1383): Mutas = selected entries in REVERSE table order;
  // mutators_mutator (:1391-1395) draws rand(10) along that list and prepends => list in table order.
  nfs = cfg.nsel;
  uint32_t score = 0;
  if (l < nfs) {
    double u = rng_peek(rng, (uint32_t)(nfs - 1 - l) + 1);
    uint32_t n = (uint32_t)(u * 10.0);
    score = n < 2 ? 2 : n;
  }
  rng_skip(rng, (uint64_t)nfs);
  uint32_t name = l < nfs ? cfg.sel_name[l] : 0;
  e_pri = l < nfs ? cfg.sel_pri[l] : 0;
  uint32_t mask = name == M_SNAND ? (uint32_t)snand_mask : 3u;
  e_meta = em_pack(score, name, name, mask);
  // mux_generators (erlamsa_gen.erl:194-199): rand(N) over the priority-sorted list
  uint32_t g = rng_rand(rng, (uint32_t)cfg.gen_total);
  gen = cfg.gen_id[cfg.ngen - 1];
  for (int i = 0; i < cfg.ngen; i++) {                         // choose_pri erlamsa_utils.erl:155-161
    if (g == 0 || g < cfg.gen_pri[i]) { gen = cfg.gen_id[i]; break; }
    g -= cfg.gen_pri[i];
  }
}

// One wavefront in front of every eh_mutate_kernel: the batch's counters back to zero, its argument block written to device memory,
// and (mode 0) the run state of the parent seed.  The runtime's own fill and copy kernels (hipMemsetAsync, hipMemcpyAsync from
// pageable memory) are workgroups of 256 work-items: four wavefronts that must find room on ONE compute unit at the same moment.
// On a device whose every slot is held by the one-wavefront workgroups of the passes in flight, and whose freed slots go to the
// next of those one at a time, they starve - 0.17 s on average and up to 0.8 s per launch with six passes in flight, seconds with
// twelve (profiles/r05_kernel_stats_before_prologue.csv) - and the pass behind them with them.  A one-wavefront kernel takes the
// next free slot like any other workgroup of the batch (the old eh_setup_kernel: 29 us on average in the same trace).
__global__ void __launch_bounds__(64) eh_prologue_kernel(KParams p, int64_t s1, int64_t s2, int64_t s3, RunState* out_, KParams* params_out_,
                                                         unsigned long long* counters_) {
  // (kernel arguments are what the host passes: generic pointers)
  EH_G RunState* out = (EH_G RunState*)out_; EH_G KParams* params_out = (EH_G KParams*)params_out_; EH_G unsigned long long* counters = (EH_G unsigned long long*)counters_;
  const int l = EH_LANE;
  for (int i = l; i < 512; i += 64) counters[i] = 0;             // ticket, output cursor, input bytes, prof slots: 4096 bytes
#ifdef EH_PROF
  if (l == 0) counters[8 + 2 * 127] = ~0ull;                     // earliest workgroup start: atomicMin
#endif
  static_assert(sizeof(KParams) % 4 == 0, "KParams is copied word by word");
  cwptr src = (cwptr)&p;
  wptr dst = (wptr)params_out;
  for (int i = l; i < (int)(sizeof(KParams) / 4); i += 64) dst[i] = src[i];
  if (p.mode != 0) return;
  Rng rng; int gen, mask, nfs; uint32_t e_pri, e_meta;
  setup_run(p.cfg, s1, s2, s3, rng, gen, mask, e_pri, e_meta, nfs);
  if (l == 0) { out->a1 = rng.a1; out->a2 = rng.a2; out->a3 = rng.a3; out->gen = gen; out->nfs = nfs; out->snand_mask = mask; }
  if (l < nfs) { out->fs_name[l] = (uint8_t)em_name(e_meta); out->fs_score[l] = (uint8_t)em_score(e_meta); out->fs_pri[l] = e_pri; }
}

// =============================================================================================
// device: emit list, split, patterns
// =============================================================================================
EH_DEV void emit_ref(Ctx& c, uint64_t ptr, uint32_t len) {
  if (len == 0) return;
  if (c.nem >= MAX_EMITS) { EH_SET_OVERFLOW(c, 302); return; }
  blk_store(c.em, c.nem, ptr, len);
  c.nem++;
}
EH_DEV void emit_all(Ctx& c) {
  for (int i = c.cur; i < c.nb; i++) { Blk b = blk_load(c.bl, i); emit_ref(c, b.ptr, b.len); }
  c.cur = c.nb;
}
// split/1 + split_into_maxblocks/2 (erlamsa_patterns.erl:45-60) on the head of bl[cur..nb)
EH_DEV void split_head(Ctx& c) {
  if (c.cur >= c.nb) return;
  Blk h = blk_load(c.bl, c.cur);
  if (h.len <= ABSMAX_BINARY_BLOCK) return;
  // rare path: lane-uniform rebuild through bl2
  int tail = c.nb - c.cur - 1;
  for (int i = EH_LANE; i < tail; i += 64) c.bl2[i] = c.bl[c.cur + 1 + i];
  wave_sync();
  int k = c.cur; uint64_t ptr = h.ptr; uint32_t rem = h.len;
  while (rem > ABSMAX_BINARY_BLOCK) {
    uint32_t as = ABSMAXHALF_BINARY_BLOCK + rng_rand(c.rng, ABSMAXHALF_BINARY_BLOCK) - 1;
    if (k + 1 + tail >= MAX_BLOCKS) { EH_SET_OVERFLOW(c, 303); return; }
    blk_store(c.bl, k++, ptr, as); ptr += as; rem -= as;
  }
  blk_store(c.bl, k++, ptr, rem);
  if (k + tail > MAX_BLOCKS) { EH_SET_OVERFLOW(c, 304); return; }
  wave_sync();
  for (int i = EH_LANE; i < tail; i += 64) c.bl[k + i] = c.bl2[i];
  c.nb = k + tail;
  wave_sync();
}

__device__ __noinline__ void gen_force(Ctx&);   // the file / jump generators' fun, called by the pattern's first uncons (below)
enum Act { A_RUN_PAT, A_MUTATE_ONCE, A_LOOP, A_CONT, A_TERMINAL, A_DONE };
enum ContKind { C_EMIT, C_ND, C_BU, C_PAT };

struct PatFrame {           // a sizer/csum wrapper waiting for its inner evaluation (prepare4sizer)
  int kind;                 // P_SZ or P_CS
  int em_field;             // sz: index of the length-field piece in the emit list; cs: first inner piece
  bptr field;           // sz: the Size/8 bytes to fill in
  uint32_t size_bits, big;  // sz
  uint64_t tail_ptr; uint32_t tail_len;   // sz: TailBin
  uint32_t crc;             // cs: 1 crc32, 0 xor8
};

// get_possible_csum_locations/1 + rand_elem (erlamsa_field_predict.erl:131-161).
// Returns 1 with (*crc,*plen,*blen), 0 for no candidate, -1 on failure.
__device__ __noinline__ int pick_csum(Ctx&, cbptr H, uint32_t L, uint32_t* crc, uint32_t* plen, uint32_t* blen) {
  EH_CTX;
  const int l = EH_LANE;
  if (L == 0) return 0;
  uint32_t maxp = (uint32_t)(2.0 * (double)L / 3.0);
  if (maxp > 30 * PREAMBLE_MAX_BYTES) maxp = 30 * PREAMBLE_MAX_BYTES;
  uint32_t np = maxp + 1;                                          // preambles 0..maxp (<= 961)
  // LDS (g_fuse_lds is free outside the fuse / sgm mutators): [0, 1024) the CRC tables wave_crc32 stages, [1024, 1024 + np) the
  // prefix CRCs, [2048, 2112) the hits as ballot masks - 16 of xor8, then 16 of crc32
  static_assert(30 * PREAMBLE_MAX_BYTES + 1 <= 1024 && 2112 <= EH_FUSE_LDS_WORDS, "pick_csum's tables fit the fuse band");
  uint32_t* T = g_fuse_lds; uint32_t* pre = g_fuse_lds + 1024; uint32_t* msk = g_fuse_lds + 2048;
  const uint32_t nbase = (np + 63) / 64;
  // xor8: xor(bytes[A .. L-1)) == byte[L-1]  <=>  prefix_xor(A) == total_xor ^ last
  uint32_t last = uni(H[L - 1]);
  uint32_t tot = wave_xor8(H, L - 1);
  uint32_t target = tot ^ last;
  wave_sync();
  uint32_t carry = 0, nx = 0;
  for (uint32_t bi = 0; bi < nbase; bi++) {
    uint32_t A = bi * 64 + (uint32_t)l;
    uint32_t v = (A < np && A < L - 1 + 1 && A > 0) ? H[A - 1] : 0;   // prefix_xor(A) = xor of bytes [0,A)
    if (A == 0 || A > L - 1) v = 0;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { uint32_t t = (uint32_t)__shfl_up((int)inc, d); if (l >= d) inc ^= t; }
    uint32_t px = inc ^ carry;
    bool hit = A < np && A <= L - 1 && px == target;               // has_xor8_checksum: needs Len - A - 1 >= 0
    const unsigned long long m = __ballot(hit);
    if (l == 0) { msk[2 * bi] = (uint32_t)m; msk[2 * bi + 1] = (uint32_t)(m >> 32); }
    nx += (uint32_t)__popcll(m);
    carry = uni((uint32_t)__shfl((int)px, 63));
  }
  // crc32: crc(bytes[A .. L-4)) == BE32(last 4), for A with L - A >= 4
  uint32_t nc = 0;
  if (L >= 4) {
    uint32_t stored = (uni(H[L - 4]) << 24) | (uni(H[L - 3]) << 16) | (uni(H[L - 2]) << 8) | uni(H[L - 1]);
    uint32_t E = L - 4;
    uint32_t whole = wave_crc32(H, E);                             // crc(0..E); the tables stay in T
    // prefix CRCs crc(0..A) for A <= maxp < 1024: lane k owns bytes [16k, 16k + 16).  The register state in front of its bytes
    // follows from the lanes before it (state' = state advanced over 16 zero bytes ^ the zero-start remainder of the 16 bytes:
    // the CRC register is linear in its start value), then every lane walks its own bytes.
    {
      const uint32_t o = 16u * (uint32_t)l;
      uint32_t by[16];
#pragma unroll
      for (int j = 0; j < 16; j++) by[j] = (o + (uint32_t)j < L && o < np) ? (uint32_t)H[o + (uint32_t)j] : 0u;
      uint32_t z = 0;                                              // zero-start remainder of my 16 bytes
#pragma unroll
      for (int j = 0; j < 16; j += 4) z = crc32_word(T, z, by[j] | (by[j + 1] << 8) | (by[j + 2] << 16) | (by[j + 3] << 24));
      uint32_t run = 0xFFFFFFFFu, mine = 0xFFFFFFFFu;
      const uint32_t nseg = (np + 15u) / 16u;
      for (uint32_t k = 0; k < nseg; k++) {
        if ((uint32_t)l == k) mine = run;
        const uint32_t zk = (uint32_t)__builtin_amdgcn_readlane((int)z, (int)k);
        run = crc32_word(T, crc32_word(T, crc32_word(T, crc32_word(T, run, 0u), 0u), 0u), 0u) ^ zk;
      }
      if (o < np) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
          if (o + (uint32_t)j < np) pre[o + (uint32_t)j] = mine ^ 0xFFFFFFFFu;      // crc32(bytes[0..A))
          mine = T[(mine ^ by[j]) & 0xFFu] ^ (mine >> 8);
        }
      }
    }
    wave_sync();
    // ... then every lane tests its own preambles, from the last down: crc(A..E) = crc(0..E) ^ crc(0..A) * x^(8(E-A))
    // (crc32_combine solved for the suffix); 64 preambles further down the multiplier is x^512 times the one before
    const uint32_t x512 = gf2_xpow8n(64);
    uint32_t mul = 0; bool have = false;
    for (int bi = (int)nbase - 1; bi >= 0; bi--) {
      uint32_t A = (uint32_t)bi * 64 + (uint32_t)l;
      bool hit = false;
      if (A < np && A <= E) {
        if (!have) { mul = gf2_xpow8n(E - A); have = true; } else mul = gf2_multmodp(x512, mul);
        uint32_t suf = A == 0 ? whole : (whole ^ gf2_multmodp(mul, pre[A]));
        hit = suf == stored;
      }
      const unsigned long long m = __ballot(hit);
      if (l == 0) { msk[32 + 2 * bi] = (uint32_t)m; msk[32 + 2 * bi + 1] = (uint32_t)(m >> 32); }
      nc += (uint32_t)__popcll(m);
    }
  } else { if (l < 32) msk[32 + l] = 0; }
  wave_sync();
  uint32_t total = nx + nc;
  if (total == 0) return 0;
  uint32_t idx = rng_rand(c.rng, total);
  uint32_t want = idx, off = 0, iscrc = 0;
  if (idx >= nx) { want = idx - nx; off = 32; iscrc = 1; }
  uint32_t Asel = 0;
  for (uint32_t bi = 0; bi < nbase; bi++) {
    unsigned long long m = (unsigned long long)uni(msk[off + 2 * bi]) | ((unsigned long long)uni(msk[off + 2 * bi + 1]) << 32);
    const uint32_t cnt = (uint32_t)__popcll(m);
    if (want < cnt) { for (uint32_t t = 0; t < want; t++) m &= m - 1; Asel = bi * 64 + (uint32_t)__builtin_ctzll(m); break; }
    want -= cnt;
  }
  wave_sync();
  *crc = iscrc; *plen = Asel; *blen = iscrc ? L - Asel - 4 : L - Asel - 1;
  return 1;
}

// ---------------------------------------------------------------------------------------------------------------------
// The container patterns (cp: gzip / zlib, ar: zip).  Both evaluate the rest of the pattern chain on a decoded payload
// (prepare4sizer(mutate_once_loop(Mutator, [], NextPat, Ip, Data, []))) and then keep using the Mutator they were GIVEN, not
// the one that evaluation returns: the scheduler's list (LaneTab), the lis / lrs / fo states and, when the result is thrown
// away, the meta trace are put back.  What is parked lives in the work area until the frame is popped.
// ---------------------------------------------------------------------------------------------------------------------
struct MutatorSave { uint32_t lt_pri[64], lt_meta[64]; uint32_t aux_save[180]; int32_t nfs; uint32_t ntrace; };   // aux + 0 .. 720: StState x 2 + FoState
EH_DEV void mutator_save(Ctx& c, EH_G MutatorSave* m, uint32_t e_pri, uint32_t e_meta) {
  const int l = EH_LANE;
  m->lt_pri[l] = e_pri; m->lt_meta[l] = e_meta;
  cwptr ax = (cwptr)c.aux;
  for (int i = l; i < 180; i += 64) m->aux_save[i] = ax[i];
  if (l == 0) { m->nfs = c.nfs; m->ntrace = c.ntrace; }
  wave_sync();
}
EH_DEV uint64_t mutator_restore(Ctx& c, const EH_G MutatorSave* m, bool trace_too) {   // returns the lane's (e_pri, e_meta)
  const int l = EH_LANE;
  wave_sync();
  wptr ax = (wptr)c.aux;
  for (int i = l; i < 180; i += 64) ax[i] = m->aux_save[i];
  c.nfs = (int)uni((uint32_t)m->nfs);
  if (trace_too) c.ntrace = uni(m->ntrace);
  wave_sync();
  return ((uint64_t)m->lt_pri[l] << 32) | m->lt_meta[l];
}
// the pieces em[from..nem) as one binary (iolist_to_binary of prepare4sizer); a single piece is used where it is
EH_DEV bptr gather_emits(Ctx& c, int from, uint64_t* len) {
  uint64_t tot = 0; for (int k = from; k < c.nem; k++) tot += blk_load(c.em, k).len;
  *len = tot;
  if (tot > 0xFFFFFF00ull) { EH_SET_OVERFLOW(c, 316); return nullptr; }
  if (c.nem - from == 1) return (bptr)blk_load(c.em, from).ptr;
  bptr blob = ws_alloc_grow(c, tot + 16);
  if (!blob) return nullptr;
  uint64_t o = 0;
  for (int k = from; k < c.nem; k++) { Blk x = blk_load(c.em, k); wave_copy(blob + o, (cbptr)x.ptr, x.len); o += x.len; }
  wave_sync();
  return blob;
}

struct CpSide { MutatorSave m; Blk orig; int32_t fmt, nrest, ip, contpat; uint32_t tr_base0, pad; };   // + Blk rest[nrest]
// mutate_once_compressed/6 (erlamsa_patterns.erl:216-246) up to the inner evaluation: zlib:gunzip(Bin), on data_error
// zlib:inflate(Bin).  1: bl[cur] is the decoded Data alone and a P_CP frame waits for the evaluation; 0: not compressed
// ({Bin, Meta}); -1: the case stops (status set).
__device__ __noinline__ int cp_begin(Ctx&, uint32_t e_pri, uint32_t e_meta, EH_G PatFrame* frames, int nfr, uint32_t ip, int contpat) {
  EH_CTX;
  const int l = EH_LANE;
  const Blk b = blk_load(c.bl, c.cur);
  cbptr H = (cbptr)b.ptr;
  if (!codec_work(c, b.len)) return -1;                                      // the attempt to decode reads the block
  EH_G ZInf* zi = (EH_G ZInf*)ws_alloc_grow(c, sizeof(ZInf));
  if (!zi) return -1;
  int fmt = 0; bptr data = nullptr; uint64_t dlen = 0;
  for (int f = ZF_GZIP; f <= ZF_ZLIB && fmt == 0; f++) {
    uint64_t outn, off;
    if (!z_uncompress_size(zi, f, H, b.len, &outn, &off)) continue;
    if (outn > 0xFFFFFF00ull) { EH_SET_OVERFLOW(c, 314); return -1; }
    bptr d = ws_alloc_grow(c, outn + 16);
    if (!d) return -1;
    if (z_uncompress_write(zi, f, H, b.len, off, d, outn)) { fmt = f; data = d; dlen = outn; }
  }
  if (fmt == 0) return 0;
  if (!codec_work(c, dlen)) return -1;                                       // ... and the payload was written
  if (nfr >= MAX_FRAMES) { EH_SET_OVERFLOW(c, 315); return -1; }
  const int nrest = c.nb - c.cur - 1;
  EH_G CpSide* sd = (EH_G CpSide*)ws_alloc_grow(c, sizeof(CpSide) + (uint64_t)nrest * sizeof(Blk));
  if (!sd) return -1;
  mutator_save(c, &sd->m, e_pri, e_meta);
  // {NewBin, [{compressed, gzip}, NewMeta, {decompressed, gzip} | Meta]} :223 (zlib :241): printed as decompressed, what the payload's
  // evaluation adds (a Meta list of its own, from []), compressed
  tr_aa(c, AT_decompressed, fmt == ZF_GZIP ? AT_gzip : AT_zlib);
  const uint32_t tr_base0 = c.tr_base; c.tr_base = c.ntrace;
  EH_G Blk* rest = (EH_G Blk*)(sd + 1);
  for (int i = l; i < nrest; i += 64) rest[i] = c.bl[c.cur + 1 + i];
  if (l == 0) {
    sd->tr_base0 = tr_base0;
    sd->orig = b; sd->fmt = fmt; sd->nrest = nrest; sd->ip = (int32_t)ip; sd->contpat = contpat;
    EH_G PatFrame& f = frames[nfr];
    f.kind = P_CP; f.em_field = c.nem; f.field = (bptr)sd; f.size_bits = 0; f.big = 0; f.tail_ptr = 0; f.tail_len = 0; f.crc = 0;
  }
  wave_sync();
  blk_store(c.bl, c.cur, (uint64_t)data, (uint32_t)dlen);                   // mutate_once_loop(Mutator, [], NextPat, Ip, Data, [])
  c.nb = c.cur + 1;
  wave_sync();
  return 1;
}
// ... and after it: NewBin = zlib:gzip(NewData) | deflate(default), split({NewBin, Rest}), NewBin =:= Bin ? (:222-224,236-244,
// 252-260).  c.pat_ret 1: [NewBin | Rest] is the result (bl[cur..nb), to be written as it is); 0: unchanged - the list is
// [Bin | Rest] again, the Mutator and the trace are the ones from before, c.pat_ip / c.pat_cont say how mutate_once_loop goes
// on; -1: the case stops.  Returns the lane's scheduler entry (e_pri << 32 | e_meta) to go on with.
__device__ __noinline__ uint64_t cp_end(Ctx&, uint32_t e_pri, uint32_t e_meta, int em_field, bptr side) {
  EH_CTX;
  const int l = EH_LANE;
  const uint64_t keep = ((uint64_t)e_pri << 32) | e_meta;
  EH_G CpSide* sd = (EH_G CpSide*)side;
  const int fmt = (int)uni((uint32_t)sd->fmt), nrest = (int)uni((uint32_t)sd->nrest);
  Blk orig; orig.ptr = uni64(sd->orig.ptr); orig.len = uni(sd->orig.len); orig.aux = 0;
  c.pat_ret = -1; c.pat_ip = uni((uint32_t)sd->ip); c.pat_cont = (int)uni((uint32_t)sd->contpat);
  c.tr_base = uni(sd->tr_base0);
  tr_aa(c, AT_compressed, fmt == ZF_GZIP ? AT_gzip : AT_zlib);
  uint64_t tot;
  cbptr nd = gather_emits(c, em_field, &tot);
  if (tot > 0 && !nd) return keep;
  c.nem = em_field;
  if (!codec_work(c, tot)) return keep;                                      // zlib:gzip(NewData) | deflate
  EH_G ZDef* zd = (EH_G ZDef*)ws_alloc_grow(c, sizeof(ZDef));
  if (!zd) return keep;
  const uint64_t cap = z_deflate_bound(tot) + z_wrap_bytes(fmt);
  bptr dst = ws_alloc_grow(c, cap);
  if (!dst) return keep;
  const uint64_t nlen = z_compress(zd, fmt, nd, tot, dst, cap);
  if (nlen == 0 || nlen > 0xFFFFFF00ull) { EH_SET_OVERFLOW(c, 317); return keep; }
  const bool changed = nlen != orig.len || !wave_equal(dst, (cbptr)orig.ptr, (uint32_t)nlen);
  if (c.cur + 1 + nrest > MAX_BLOCKS) { EH_SET_OVERFLOW(c, 318); return keep; }
  const EH_G Blk* rest = (const EH_G Blk*)(sd + 1);
  wave_sync();
  for (int i = l; i < nrest; i += 64) c.bl[c.cur + 1 + i] = rest[i];
  if (changed) blk_store(c.bl, c.cur, (uint64_t)dst, (uint32_t)nlen); else blk_store(c.bl, c.cur, orig.ptr, orig.len);
  c.nb = c.cur + 1 + nrest;
  wave_sync();
  c.pat_ret = changed ? 1 : 0;
  if (changed) return keep;
  return mutator_restore(c, &sd->m, true);
}

struct ArSide { MutatorSave m; Blk archive; int32_t n, idx, ip, contpat; EH_G ZipEntry* es; uint32_t tr_base0, pad; };
// mutate_once_archiver/4 (erlamsa_patterns.erl:203-214): UnZip = zip:foldl(fun(N, I, B, Acc) -> [{N, B(), I()} | Acc] end, [], ..) over
// bl[cur] (the whole list as one binary).  c.pat_ret 1: an archive - the side block (returned) holds its entries and the Mutator;
// 0: {error, _}; -1: the case stops.
__device__ __noinline__ bptr ar_begin(Ctx&, uint32_t e_pri, uint32_t e_meta, uint32_t ip, int contpat) {
  EH_CTX;
  const Blk b = blk_load(c.bl, c.cur);
  c.pat_ret = -1;
  if (!codec_work(c, b.len)) return nullptr;                                 // zip:foldl reads the archive
  EH_G ZipRd* rd = (EH_G ZipRd*)ws_alloc_grow(c, sizeof(ZipRd));
  if (!rd) return nullptr;
  int rc = zip_open(rd, (cbptr)b.ptr, b.len);
  if (rc == ZR_UNSUP) { c.status = CASE_UNSUPPORTED; return nullptr; }
  if (rc != ZR_OK) { c.pat_ret = 0; return nullptr; }
  const uint32_t n = uni(rd->entries);
  EH_G ArSide* sd = (EH_G ArSide*)ws_alloc_grow(c, sizeof(ArSide));
  if (!sd) return nullptr;
  EH_G ZipEntry* es = (EH_G ZipEntry*)ws_alloc_grow(c, (uint64_t)(n ? n : 1) * sizeof(ZipEntry));
  if (!es) return nullptr;
  for (uint32_t i = 0; i < n; i++) {
    rc = zip_next<true>(c, rd, &es[i]);
    if (rc == ZR_STOP) return nullptr;
    if (rc == ZR_UNSUP) { c.status = CASE_UNSUPPORTED; return nullptr; }
    if (rc == ZR_CRASH) { c.status = CASE_CRASHED; return nullptr; }
    if (rc != ZR_OK) { c.pat_ret = 0; return nullptr; }
    if (!codec_work(c, uni(es[i].data_len))) return nullptr;                 // ... and inflates every file
  }
  mutator_save(c, &sd->m, e_pri, e_meta);
  if (EH_LANE == 0) { sd->archive = b; sd->n = (int32_t)n; sd->idx = (int32_t)n - 1; sd->ip = (int32_t)ip; sd->contpat = contpat; sd->es = es; }
  wave_sync();
  c.pat_ret = 1;
  return (bptr)sd;
}
// The lists:mapfoldl of mutate_once_archiver/7 (:175-186) from entry sd->idx downwards (FileSpec is in reverse central-directory
// order): R = rand(1000) per file, R > 750 runs the rest of the pattern chain on the file's bytes.  c.pat_ret 2: bl[cur] is such a
// file and a P_AR frame waits for the evaluation; 1: all files done and zip:create gave bl[cur] = NewBin; 0: zip:create failed -
// bl[cur] is the archive again, Mutator and trace are put back (the {error, _} clause, :165-174); -1: the case stops.
// Returns the lane's scheduler entry to go on with.
__device__ __noinline__ uint64_t ar_step(Ctx&, uint32_t e_pri, uint32_t e_meta, bptr side, EH_G PatFrame* frames, int nfr) {
  EH_CTX;
  const uint64_t keep = ((uint64_t)e_pri << 32) | e_meta;
  EH_G ArSide* sd = (EH_G ArSide*)side;
  EH_G ZipEntry* es = (EH_G ZipEntry*)uni64((uint64_t)sd->es);
  const int n = (int)uni((uint32_t)sd->n);
  c.pat_ret = -1; c.pat_ip = uni((uint32_t)sd->ip); c.pat_cont = (int)uni((uint32_t)sd->contpat);
  int idx = (int)uni((uint32_t)sd->idx);
  while (idx >= 0) {
    uint32_t r = rng_rand(c.rng, 1000);
    if (r > 750) {                                                                      // 25 % of the files are mutated
      if (nfr >= MAX_FRAMES) { EH_SET_OVERFLOW(c, 319); return keep; }
      blk_store(c.bl, c.cur, uni64(es[idx].data), uni(es[idx].data_len));
      c.nb = c.cur + 1;
      if (c.trace) tr_name_emit((cbptr)uni64(es[idx].name), uni(es[idx].name_len), uni(es[idx].up));   // [NM, {archiver, N} | Acc] :183: the name, then what the file's evaluation adds (a list of its own)
      if (EH_LANE == 0) sd->tr_base0 = c.tr_base;
      c.tr_base = c.ntrace;
      if (EH_LANE == 0) {
        sd->idx = idx;
        EH_G PatFrame& f = frames[nfr];
        f.kind = P_AR; f.em_field = c.nem; f.field = side; f.size_bits = 0; f.big = 0; f.tail_ptr = 0; f.tail_len = 0; f.crc = 0;
      }
      wave_sync();
      c.pat_ret = 2;
      return keep;
    }
    idx--;
  }
  if (c.work_budget) {                                                                  // zip:create deflates every file again
    uint64_t sum = 0;
    for (int i = EH_LANE; i < n; i += 64) sum += es[i].data_len;
    if (!codec_work(c, wave_sum64(sum))) return keep;
  }
  bptr out; uint64_t len;
  int rc = zip_create<true>(c, es, (uint32_t)n, &out, &len);                            // zip:create(Name, lists:reverse(NewFileSpec), [memory])
  if (rc == ZR_STOP) return keep;
  if (rc == ZR_UNSUP) { c.status = CASE_UNSUPPORTED; return keep; }
  if (rc == ZR_OK) {
    blk_store(c.bl, c.cur, (uint64_t)out, (uint32_t)len); c.nb = c.cur + 1;
    tr_aa(c, AT_archiver, AT_ok);                                                        // [{archiver, ok}, flatten(NewMeta) | Meta] :196
    wave_sync();
    c.pat_ret = 1;
    return keep;
  }
  blk_store(c.bl, c.cur, uni64(sd->archive.ptr), uni(sd->archive.len)); c.nb = c.cur + 1;
  wave_sync();
  c.pat_ret = 0;
  return mutator_restore(c, &sd->m, true);
}
// a file's evaluation ended: its pieces are the file's new bytes (prepare4sizer), the next file starts from the Mutator again
__device__ __noinline__ uint64_t ar_file_done(Ctx&, int em_field, bptr side) {
  EH_CTX;
  EH_G ArSide* sd = (EH_G ArSide*)side;
  EH_G ZipEntry* es = (EH_G ZipEntry*)uni64((uint64_t)sd->es);
  const int idx = (int)uni((uint32_t)sd->idx);
  uint64_t tot;
  bptr nb = gather_emits(c, em_field, &tot);
  c.pat_ret = -1;
  c.tr_base = uni(sd->tr_base0);
  if (!nb) return 0;
  c.nem = em_field;
  if (EH_LANE == 0) { es[idx].data = (uint64_t)nb; es[idx].data_len = (uint32_t)tot; sd->idx = idx - 1; }
  wave_sync();
  c.pat_ret = 0;
  return mutator_restore(c, &sd->m, false);
}

// The two container patterns behind one call each, so that the pattern machine (inlined into the kernel) only carries two calls.
// begin: c.pat_ret 2 = a payload is in bl[cur] and a frame was pushed; 1 = bl[cur..nb) is the pattern's result; 0 = not a
// container (go on with split/1 and mutate_once_loop); -1 = the case stops.
__device__ __noinline__ uint64_t pat_container_begin(Ctx&, uint32_t e_pri, uint32_t e_meta, int pat, EH_G PatFrame* frames, int nfr, uint32_t ip, int contpat) {
  EH_CTX;
  const uint64_t keep = ((uint64_t)e_pri << 32) | e_meta;
  if (pat == P_CP) {
    int r = cp_begin(c, e_pri, e_meta, frames, nfr, ip, contpat);
    c.pat_ret = r < 0 ? -1 : (r == 1 ? 2 : 0);
    return keep;
  }
  // list_to_binary([Bin|Rest]) -> one block
  c.pat_ret = -1;
  uint64_t tot = 0; for (int i = c.cur; i < c.nb; i++) tot += blk_load(c.bl, i).len;
  if (tot > 0xFFFFFFF0ull) { EH_SET_OVERFLOW(c, 308); return keep; }
  if (c.nb - c.cur > 1) {
    bptr all = ws_alloc_grow(c, tot);
    if (!all) return keep;
    uint64_t o = 0;
    for (int i = c.cur; i < c.nb; i++) { Blk x = blk_load(c.bl, i); wave_copy(all + o, (cbptr)x.ptr, x.len); o += x.len; }
    wave_sync();
    blk_store(c.bl, c.cur, (uint64_t)all, (uint32_t)tot); c.nb = c.cur + 1;
    wave_sync();
  }
  bptr side = ar_begin(c, e_pri, e_meta, ip, contpat);
  if (c.pat_ret != 1) return keep;
  return ar_step(c, e_pri, e_meta, side, frames, nfr);
}
// end of a payload's evaluation (the frame has been popped): c.pat_ret 2 = the next payload (ar) with a new frame; 1 = bl[cur..nb)
// is the result; 0 = go on with mutate_once_loop on bl[cur..nb) (split/1 done), c.pat_ip / c.pat_cont; -1 = the case stops.
__device__ __noinline__ uint64_t pat_container_end(Ctx&, uint32_t e_pri, uint32_t e_meta, int kind, int em_field, bptr side, EH_G PatFrame* frames, int nfr) {
  EH_CTX;
  uint64_t e;
  if (kind == P_CP) {
    e = cp_end(c, e_pri, e_meta, em_field, side);
    if (c.pat_ret < 0) return e;
    split_head(c);                                                                    // {This, LlN} = split({NewBin, Rest}) :254 (when changed, the pieces are NewBin again)
    return e;
  }
  e = ar_file_done(c, em_field, side);
  if (c.pat_ret < 0) return ((uint64_t)e_pri << 32) | e_meta;
  e = ar_step(c, (uint32_t)(e >> 32), (uint32_t)e, side, frames, nfr);
  if (c.pat_ret == 0) split_head(c);                                                  // the {error, _} clause :165-174
  return e;
}

EH_DEV void run_patterns(Ctx& c, LaneTab& lt, int pat) {
  int act = A_RUN_PAT, cont = C_EMIT, contpat = 0; uint32_t ip = 0;
  int guard = 0;
  EH_G PatFrame* frames = (EH_G PatFrame*)(c.aux + 1152); int nfr = 0;     // wrapper stack lives in slot memory
  while (act != A_DONE && c.status == CASE_OK) {
    if (++guard > 1000000) { EH_SET_OVERFLOW(c, 305); break; }
    switch (act) {
      case A_RUN_PAT:
        {                                                                             // [{pattern, once_dec | many_dec | burst} | Meta] :309,:326,:349; make_complex_pat's {pattern, Type} :356; nu: {pattern, no_muta} :390; co adds none
          const int pa = pat == P_OD ? AT_once_dec : pat == P_ND ? AT_many_dec : pat == P_BU ? AT_burst : pat == P_SK ? AT_skipper : pat == P_SZ ? AT_sizer : pat == P_CS ? AT_csum :
                         pat == P_AR ? AT_archiver : pat == P_CP ? AT_compressed : pat == P_NU ? AT_no_muta : -1;
          if (pa >= 0) tr_aa(c, AT_pattern, pa);
        }
        switch (pat) {
          case P_OD: cont = C_EMIT; act = A_MUTATE_ONCE; break;                       // :306-309
          case P_ND: cont = C_ND; act = A_MUTATE_ONCE; break;                         // :323-326
          case P_BU: cont = C_BU; act = A_MUTATE_ONCE; break;                         // :346-349
          case P_CO: pat = rng_erand(c.rng, 2) == 1 ? P_NU : P_OD; break;             // :378-384
          case P_NU: if (c.gen_pending) gen_force(c); split_head(c); emit_all(c); act = A_TERMINAL; break;   // :386-390
          default: {
            // make_complex_pat :351-357: the continuation pattern is drawn first, then Ip
            contpat = (int)rng_rand(c.rng, P_COUNT);                                  // rand_elem(patterns())
            cont = C_PAT;
            ip = rng_rand(c.rng, INITIAL_IP);
            if (c.gen_pending) { gen_force(c); if (c.status != CASE_OK) break; }      // uncons(Ll, false) calls a fun Ll
            if (c.cur >= c.nb) { c.status = CASE_CRASHED; break; }                    // uncons(Ll, false) -> false -> badarg
            Blk b = blk_load(c.bl, c.cur);
            cbptr H = (cbptr)b.ptr;
            if (pat == P_SK) {                                                        // mutate_once_skipper :146-161
              uint32_t len = rng_rand(c.rng, b.len / 2);
              if (c.trace) tr_kv_emit(TRK_SKIPPED, 0, 0, 0, 1, len, 0, 0);             // [{skipped, Len/8} | Meta] :154
              emit_ref(c, b.ptr, len);
              blk_store(c.bl, c.cur, b.ptr + len, b.len - len);
              wave_sync();
            } else if (pat == P_SZ) {                                                 // mutate_once_sizer :81-111
              SizerElem e;
              int r;
              {                                                                       // (out of work memory: borrow a larger area, pick again)
                const Rng rng0 = c.rng; const uint64_t mark = c.ws_used; int last_tier = 0;
                EH_PT0;
                for (;;) {
                  r = pick_simple_len(c, H, b.len, &e);
                  if (c.status != CASE_OVERFLOW || c.ovf_need == 0 || !ws_regrow(c, mark, &last_tier)) break;
                  c.rng = rng0;
                }
                EH_PT(c, 60);
              }
              if (r < 0) break;
              if (r == 0) tr_aa(c, AT_sizer, AT_failed);                              // [{sizer, failed} | Meta] :85
              if (r == 1) {
                if (c.trace) tr_kv_emit(TRK_SIZER, 2, e.size_bits / 8, e.big ? 1u : 0u, 3, e.len, e.a, e.b);   // [{sizer, Elem} | Meta] :97
                uint32_t nbytes = e.size_bits / 8;
                if ((uint64_t)e.a + nbytes + e.len > b.len) { c.status = CASE_CRASHED; break; }
                if (nfr >= MAX_FRAMES) { EH_SET_OVERFLOW(c, 306); break; }
                bptr fld = ws_alloc_grow(c, 16);
                if (!fld) break;
                emit_ref(c, b.ptr, e.a);                                              // H
                if (EH_LANE == 0) {
                  EH_G PatFrame& f = frames[nfr];
                  f.kind = P_SZ; f.em_field = c.nem; f.field = fld; f.size_bits = e.size_bits; f.big = e.big;
                  f.tail_ptr = b.ptr + e.a + nbytes + e.len; f.tail_len = b.len - (e.a + nbytes + e.len); f.crc = 0;
                }
                nfr++;
                emit_ref(c, (uint64_t)fld, nbytes);                                   // length field, filled when the inner evaluation ends
                blk_store(c.bl, c.cur, b.ptr + e.a + nbytes, e.len);                  // Blob
                wave_sync();
              }
            } else if (pat == P_CS) {                                                 // mutate_once_csum :115-144
              uint32_t iscrc, plen, blen;
              EH_PT0;
              int r = pick_csum(c, H, b.len, &iscrc, &plen, &blen);
              EH_PT(c, 59);
              if (r < 0) break;
              if (r == 0) tr_aa(c, AT_csum, AT_failed);                               // :119
              if (r == 1) {
                if (c.trace) tr_kv_emit(TRK_CSUM, 1, iscrc, 0, 2, plen, blen, 0);      // [{csum, Elem} | Meta] :131
                if (nfr >= MAX_FRAMES) { EH_SET_OVERFLOW(c, 307); break; }
                emit_ref(c, b.ptr, plen);                                             // P
                if (EH_LANE == 0) {
                  EH_G PatFrame& f = frames[nfr];
                  f.kind = P_CS; f.em_field = c.nem; f.crc = iscrc; f.field = nullptr; f.size_bits = 0; f.big = 0; f.tail_ptr = 0; f.tail_len = 0;
                }
                nfr++;
                blk_store(c.bl, c.cur, b.ptr + plen, blen);
                wave_sync();
              }
            } else if (pat == P_AR || pat == P_CP) {                                  // mutate_once_archiver :165-214, mutate_once_compressed :216-260
              EH_PT0;
              uint64_t e = pat_container_begin(c, lt.e_pri, lt.e_meta, pat, frames, nfr, ip, contpat);
              EH_PT(c, pat == P_CP ? 63 : 78);
              lt.e_pri = (uint32_t)(e >> 32); lt.e_meta = (uint32_t)e;
              if (c.pat_ret < 0) break;
              if (c.pat_ret == 2) { nfr++; act = A_LOOP; break; }                     // the rest of the chain on a payload: mutate_once_loop(Mutator, [], NextPat, Ip, Data, [])
              if (c.pat_ret == 1) { emit_all(c); act = A_TERMINAL; break; }           // [NewBin | {..}]
              tr_aa(c, pat == P_CP ? AT_compressed : AT_archiver, AT_failed);         // [{compressed, failed} | Meta] :259 / [{archiver, failed} | Meta] :169
            }
            split_head(c);
            act = A_LOOP;
            break;
          }
        }
        break;
      case A_MUTATE_ONCE:                                                             // mutate_once/4 :265-278
        if (!c.gen_pending && c.nb - c.cur == 1 && blk_load(c.bl, c.cur).len == 0) { tr_aa(c, AT_mutate_once, AT_empty_stopped); c.cur = c.nb; act = A_TERMINAL; break; }   // {Mutator, [{mutate_once, empty_stopped} | Meta]} :268-269 (a fun does not match [<<>>])
        ip = rng_rand(c.rng, INITIAL_IP);
        if (c.gen_pending) { gen_force(c); if (c.status != CASE_OK) break; }          // uncons(Ll, false) calls a fun Ll
        if (c.cur >= c.nb) { act = A_CONT; break; }                                   // Cont([], ...)
        split_head(c);
        act = A_LOOP;
        break;
      case A_LOOP: {                                                                  // mutate_once_loop/6 :281-296
        uint32_t n = rng_rand(c.rng, ip);
        if (n == 0 || c.nb - c.cur == 1) { mux_fuzzers(c, lt); act = A_CONT; }
        else { Blk b = blk_load(c.bl, c.cur); emit_ref(c, b.ptr, b.len); c.cur++; }
        break;
      }
      case A_CONT:
        switch (cont) {
          case C_EMIT: emit_all(c); act = A_TERMINAL; break;
          case C_ND:                                                                  // pat_many_dec_cont :313-321
            if (rng_occurs(c.rng, 4, 5)) { tr_aa(c, AT_pattern, AT_many_dec); act = A_MUTATE_ONCE; } else { emit_all(c); act = A_TERMINAL; }
            break;
          case C_BU: {                                                                // pat_burst_cont :331-344
            int n = 1;
            while (c.status == CASE_OK) {
              bool p = rng_occurs(c.rng, 4, 5);
              if (p || n < 2) { mux_fuzzers(c, lt); n++; } else { emit_all(c); act = A_TERMINAL; break; }
              if (++guard > 1000000) { EH_SET_OVERFLOW(c, 309); break; }
            }
            break;
          }
          case C_PAT: pat = contpat; act = A_RUN_PAT; break;
        }
        break;
      case A_TERMINAL: {
        // the innermost evaluation finished: unwind the sizer/csum wrappers inside out
        if (nfr == 0) { act = A_DONE; break; }
        wave_sync();
        PatFrame f = frames[--nfr];
        f.kind = (int)uni((uint32_t)f.kind); f.em_field = (int)uni((uint32_t)f.em_field); f.field = (bptr)uni64((uint64_t)f.field);
        f.size_bits = uni(f.size_bits); f.big = uni(f.big); f.tail_ptr = uni64(f.tail_ptr); f.tail_len = uni(f.tail_len); f.crc = uni(f.crc);
        if (f.kind == P_AR || f.kind == P_CP) {
          EH_PT0;
          uint64_t e = pat_container_end(c, lt.e_pri, lt.e_meta, f.kind, f.em_field, f.field, frames, nfr);
          EH_PT(c, f.kind == P_CP ? 69 : 79);
          lt.e_pri = (uint32_t)(e >> 32); lt.e_meta = (uint32_t)e;
          if (c.pat_ret < 0) break;
          ip = c.pat_ip; cont = C_PAT; contpat = c.pat_cont;
          if (c.pat_ret == 2) { nfr++; act = A_LOOP; }                                // ar: the next file's evaluation
          else if (c.pat_ret == 1) emit_all(c);                                       // [NewBin | Rest] ++ [..]: no continuation
          else { tr_aa(c, f.kind == P_CP ? AT_compressed : AT_archiver, AT_failed); act = A_LOOP; }   // unchanged / zip:create failed: mutate_once_loop(Mutator, [{compressed | archiver, failed} | Meta], NextPat, Ip, This, LlN) - the Meta from before the pattern (mutator_restore)
        } else if (f.kind == P_SZ) {
          // NewLen = size(NewBlob) = everything written after the length field  (:105-110)
          uint64_t tot = 0; for (int k = f.em_field + 1; k < c.nem; k++) tot += blk_load(c.em, k).len;
          // (the field piece itself is at em_field unless Size/8 was 0, which cannot happen)
          if (EH_LANE == 0) put_field(f.field, tot, f.size_bits, f.big != 0);
          wave_sync();
          emit_ref(c, f.tail_ptr, f.tail_len);                                        // TailBin
        } else {
          // NewC = recalc_csum(Type, NewBlob): gather the inner pieces, checksum, append  (:139-143)
          uint64_t tot = 0; for (int k = f.em_field; k < c.nem; k++) tot += blk_load(c.em, k).len;
          if (tot > 0xFFFFFFF0ull) { EH_SET_OVERFLOW(c, 310); break; }
          EH_PT0;
          bptr blob = ws_alloc_grow(c, tot + 16);
          if (!blob) break;
          uint64_t o = 0;
          for (int k = f.em_field; k < c.nem; k++) { Blk x = blk_load(c.em, k); wave_copy(blob + o, (cbptr)x.ptr, x.len); o += x.len; }
          wave_sync();
          EH_PT(c, 61);
          uint32_t cs = f.crc ? wave_crc32(blob, (uint32_t)tot) : wave_xor8(blob, (uint32_t)tot);
          EH_PT(c, 62);
          uint32_t cb = f.crc ? 4u : 1u;
          if (EH_LANE == 0) put_field(blob + tot, cs, cb * 8, true);
          wave_sync();
          c.nem = f.em_field;
          emit_ref(c, (uint64_t)blob, (uint32_t)tot + cb);
        }
        break;
      }
    }
  }
}

// =============================================================================================
// device: generators
// =============================================================================================
// finish/1 (erlamsa_gen.erl:43-51): with probability 1/(Len+1) a block of random bytes behind the stream, at bl[c.nb]
EH_DEV void gen_finish(Ctx& c, uint32_t len) {
  uint32_t n = rng_rand(c.rng, len + 1);
  if (n == len) {
    uint32_t bits = rng_range(c.rng, 1, 16);
    uint32_t nlen = rng_rand(c.rng, 1u << bits);
    if (nlen > 0) {                                                    // check_empty
      bptr dst = ws_alloc_grow(c, nlen);
      if (!dst) return;
      random_block_rev(c, dst, nlen);
      if (c.nb >= MAX_BLOCKS) { EH_SET_OVERFLOW(c, 312); return; }
      blk_store(c.bl, c.nb, (uint64_t)dst, nlen);
      c.nb++;
    }
  }
}
EH_DEV uint32_t rand_block_size(Ctx& c) {                              // :55-56
  const DevConfig& cfg = c.p->cfg;
  uint32_t r = rng_rand(c.rng, cfg.max_block_scaled);
  return r > cfg.min_block_scaled ? r : cfg.min_block_scaled;
}
EH_DEV void gen_direct(Ctx& c, cbptr in, uint32_t L) {       // erlamsa_gen.erl:152-164 (split_binary guard never holds)
  (void)rand_block_size(c);
  blk_store(c.bl, 0, (uint64_t)in, L);
  c.nb = 1;
  gen_finish(c, L);
  wave_sync();
}
// port_stream/2 forced (erlamsa_gen.erl:59-90) over corpus entry e: the blocks point into the arena (nothing is copied), the
// next block size is drawn after every full block, a short read is followed by eof and finish(Len).  Fills bl[0..nb).
EH_DEV void gen_stream(Ctx& c, uint32_t e) {
  const EH_G KParams& p = *c.p;
  uint64_t o0 = uni64(p.coff[e]), o1 = uni64(p.coff[e + 1]);
  cbptr in = p.corpus + o0;
  uint32_t L = (uint32_t)(o1 - o0), pos = 0;
  c.nb = 0;
  uint32_t wanted = rand_block_size(c);
  while (pos < L) {
    if (c.nb >= MAX_BLOCKS - 1) { EH_SET_OVERFLOW(c, 313); return; }
    uint32_t avail = L - pos;
    if (avail >= wanted) { blk_store(c.bl, c.nb++, (uint64_t)(in + pos), wanted); pos += wanted; wanted = rand_block_size(c); }
    else { blk_store(c.bl, c.nb++, (uint64_t)(in + pos), avail); pos = L; }
  }
  gen_finish(c, L);
  wave_sync();
}
// The file and jump generators hand the pattern a FUN (file_streamer :106-121, jump_streamer :136-150): only the paths are
// drawn when DataGen() runs; the streams are read, and their block sizes drawn, when the pattern's first uncons/2 calls the
// fun (erlamsa_utils.erl:93) - after the pattern's own first draws.  gen_force is that call.
__device__ __noinline__ void gen_force(Ctx&) {
  EH_CTX;
  const int kind = c.gen_pending;
  c.gen_pending = 0;
  if (kind == G_FILE) { gen_stream(c, c.gen_e1); return; }
  // jump_somewhere/2 :124-133
  uint64_t dptr[2]; uint32_t dlen[2];
  for (int k = 0; k < 2; k++) {
    gen_stream(c, k == 0 ? c.gen_e1 : c.gen_e2);
    if (c.status != CASE_OK) return;
    if (c.nb == 0) { c.status = CASE_CRASHED; return; }                // rand_elem([]) = [] ; size([]) -> badarg
    uint32_t i = rng_erand(c.rng, (uint32_t)c.nb) - 1;                 // rand_elem/1 (erlamsa_rnd.erl:128-132)
    Blk b = blk_load(c.bl, (int)i);
    dptr[k] = b.ptr; dlen[k] = b.len;
    wave_sync();
  }
  uint32_t s1 = rng_rand(c.rng, dlen[0]), s2 = rng_rand(c.rng, dlen[1]);
  uint32_t l1 = rng_erand(c.rng, dlen[0] - s1), l2 = rng_erand(c.rng, dlen[1] - s2);
  bptr dst = ws_alloc_grow(c, (uint64_t)l1 + l2);
  if (!dst) return;
  wave_copy(dst, (cbptr)dptr[0] + s1, l1);
  wave_copy(dst + l1, (cbptr)dptr[1] + s2, l2);
  wave_sync();
  blk_store(c.bl, 0, (uint64_t)dst, l1 + l2);                          // uncons(B) when is_binary(B) -> {B, []}
  c.nb = 1;
  wave_sync();
}
EH_DEV void gen_random(Ctx& c) {                                       // random_stream/1 :167-178
  const DevConfig& cfg = c.p->cfg;
  c.nb = 0;
  while (c.status == CASE_OK) {
    uint32_t n = rng_range(c.rng, 32, cfg.max_block_scaled);
    bptr dst = ws_alloc_grow(c, n);
    if (!dst) return;
    random_block_rev(c, dst, n);
    if (c.nb >= MAX_BLOCKS) { EH_SET_OVERFLOW(c, 311); return; }
    blk_store(c.bl, c.nb++, (uint64_t)dst, n);
    uint32_t ip = rng_range(c.rng, 1, 100);
    if (rng_rand(c.rng, ip) == 0) break;
  }
  wave_sync();
}

// =============================================================================================
// cooperative execution: chunk t of a posted loop (eh_device.h co_run / co_help)
// =============================================================================================
__device__ __noinline__ void co_exec(const EH_G CoJob* j, uint32_t t) {
  const uint32_t kind = uni(co_ld32(&j->kind));
  const uint64_t a0 = uni64(j->a[0]), a1 = uni64(j->a[1]), a2 = uni64(j->a[2]);
  if (kind == CO_COPY || kind == CO_EQUAL) {
    const uint32_t ch = uni((uint32_t)j->a[3]);
    const uint64_t off = (uint64_t)t * ch;
    const uint32_t len = a2 - off < ch ? (uint32_t)(a2 - off) : ch;
    if (kind == CO_COPY) wave_copy_raw((bptr)a0 + off, (cbptr)a1 + off, len);
    else if (!wave_equal_raw((cbptr)a0 + off, (cbptr)a1 + off, len)) { if (EH_LANE == 0) atomicAdd((EH_G unsigned long long*)&j->acc, 1ull); }
    return;
  }
  if (kind == CO_FBPASS) { fb_pass_chunk(j, t); return; }
}

// =============================================================================================
// the mutate kernel
// =============================================================================================
EH_DEV void sys_store(EH_G unsigned long long* p, unsigned long long v) {   // a store the host sees (page-locked host memory)
#ifdef HIPEMU
  *p = v;
#else
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
constexpr int CTR_OCC = 272;                 // counters [272, 274): 100 MHz ticks of the workgroups in cases, lingering for posted chunks ([7]: their lifetimes)
constexpr int CTR_STATUS = 264;              // counters [264, 270): cases of the batch by status ([8, 264) are the EH_PROF slots)
constexpr unsigned int CO_LINGER = 24;       // wavefronts of a pass that stay for posted chunks once the pass is out of tickets
// 2 wavefronts per SIMD = up to 256 VGPRs: at 4 (128 VGPRs) the scheduler loops and the candidate loop of base64_mutator
// reload spilled registers from scratch on every iteration (7.5 M instead of 1.6 M memory instructions for one
// b64-heavy case, twice the time, profiles/r03_summary.json "occupancy"); the longest cases set the duration of a pass.
#ifndef EH_WAVES_PER_SIMD
#define EH_WAVES_PER_SIMD 2
#endif
// The argument block lives in device memory: taking the address of a by-value kernel argument (c.p) made the
// compiler keep a ~700-byte private copy of it per lane.
__global__ void __launch_bounds__(64, EH_WAVES_PER_SIMD) eh_mutate_kernel(const KParams* __restrict__ pp_) {
  const EH_G KParams* pp = (const EH_G KParams*)pp_;
  const int l = EH_LANE;
  const EH_G KParams& p = *pp;
  Ctx& c = g_ctx;
  LaneTab lt;
  c.p = pp;
  c.work_budget = p.work_budget;
#ifdef EH_PROF
  // how many workgroups the device holds at once: earliest / latest workgroup start of the dispatch (100 MHz wall clock)
  if (l == 0) { unsigned long long t = __builtin_amdgcn_s_memrealtime(); atomicMin(&pp->prof[2 * 127], t ? t : 1ull); atomicMax(&pp->prof[2 * 127 + 1], t); }
#endif
  // this workgroup's slot: block tables + work area
  const uint32_t slot_id = blockIdx.x;
  bptr slot = p.slot_base + (uint64_t)slot_id * p.slot_stride;
  c.bl = (EH_G Blk*)slot;
  c.bl2 = c.bl + MAX_BLOCKS;
  c.em = c.bl2 + MAX_BLOCKS;
  c.aux = (bptr)(c.em + MAX_EMITS);
  bptr const trace0 = c.aux + AUX_BYTES;
  bptr const ws0 = trace0 + TRACE_CAP;

  // mode 0: the run state is shared by all cases
  Rng parent; int gen0 = 0; uint32_t pri0 = 0, meta0 = 0; int nfs0 = 0;
  if (p.mode == 0) {
    const EH_G RunState* rs = p.run;
    parent.a1 = rs->a1; parent.a2 = rs->a2; parent.a3 = rs->a3; parent.draws = 0;
    gen0 = rs->gen; nfs0 = rs->nfs;
    if (l < nfs0) {
      uint32_t name = rs->fs_name[l];
      pri0 = rs->fs_pri[l];
      meta0 = em_pack(rs->fs_score[l], name, name, name == M_SNAND ? (uint32_t)rs->snand_mask : 3u);
    }
  }

  // where the wave slots' time goes (eh_result_occupancy): this workgroup's life, its cases, its stay for posted chunks - 100 MHz ticks
  const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long case_ticks = 0, linger_ticks = 0;
  const uint64_t TICKET_BATCH = 4;                // cases claimed per atomic (one counter saturates at ~88 dequeues/us)
  uint64_t tk_next = 0, tk_end = 0;
  while (true) {
    if (tk_next == tk_end) {
      unsigned long long t = 0;
      if (l == 0) t = atomicAdd(p.ticket, (unsigned long long)TICKET_BATCH);
      tk_next = uni64(t); tk_end = tk_next + TICKET_BATCH;
    }
    if (p.board && uni(co_ld32(&p.board->open)) != 0) co_help(p.board, 2);   // chunks of heavy cases' loops, between two cases of my own
    uint64_t i = tk_next++;
    if (i >= p.n) {
#ifndef HIPEMU
      // Out of tickets: the pass ends with its heaviest cases, each on ONE wavefront, and what those post would find nobody between
      // two cases any more.  Up to CO_LINGER wavefronts of the pass stay for chunks - of any pass's cases - while the pass has
      // cases under way and some case on the device is posting; the others leave their slots to the workgroups of the next passes.
      if (p.board) {
        unsigned int mine = p.co_linger;
        if (l == 0) mine = (unsigned int)atomicAdd(p.ticket + 4, 1ull);
        if (uni(mine) < p.co_linger) {
          const unsigned long long lg0 = __builtin_amdgcn_s_memrealtime();
          EH_G CoBoard* bd = p.board;
          for (uint32_t spins = 0; spins < (1u << 24); spins++) {
            if (uni64(co_ld64(p.ticket + 3)) >= p.n || uni(co_ld32(&bd->posters)) == 0) break;
            if (uni(co_ld32(&bd->open)) != 0) co_help(bd, 8);
            else { for (int z = 0; z < 6; z++) __builtin_amdgcn_s_sleep(127); }   // ~20 us: posted loops last hundreds (every poll is a trip to the L2 of all these wavefronts)
          }
          linger_ticks = __builtin_amdgcn_s_memrealtime() - lg0;
        }
      }
#endif
      break;
    }
    c.co_posted = 0;
    const unsigned long long case_t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t tick0 = __builtin_readcyclecounter();
#ifndef HIPEMU
    c.t_case = tick0;
#else
    c.t_case = 0;                                                          // the emulator's lanes read their own clocks
#endif
    c.nchunk = 0; c.ws_peak = 0; c.ws_top = 0;
    c.trace = (p.flags & EH_FLAG_META_TRACE) ? trace0 : nullptr; c.ntrace = 0; c.tr_base = 0; c.m_aux = -1;
    c.ch_vstart[0] = 0; c.ch_vend[0] = p.work_cap; c.ch_base[0] = ws0; c.ch_tier[0] = 0; c.ch_area[0] = slot_id;
    ws_set_view(c, 0);
    for (int d = 0; d < LEX_LEVELS; d++) c.lex_ptr[d] = 0;
    unsigned long long base = 0; uint64_t total = 0;
    c.work = 0; c.depth = 0;
    c.status = CASE_OK; c.lastm = -1; c.nb = 0; c.cur = 0; c.nem = 0; c.ws_used = 0;
    c.r_kind = R_SAME; c.r_flush = 0; c.r_drop_next = 0; c.r2 = 0;
    if (l == 0) { ((EH_G StState*)c.aux)[0].count = 0; ((EH_G StState*)c.aux)[1].count = 0; for (int d = 0; d < LEX_LEVELS; d++) { EH_G LexCache& lc = ((EH_G LexCache*)(c.aux + AUX_LEXCACHE))[d]; lc.n = -1; lc.tcap = 0; } ((EH_G FoState*)(c.aux + 704))->has = 0; }
    wave_sync();
    int gen;
    Rng pr;
    if (p.mode == 0) {
      // ThreadSeed of case I = parent draws 3(I-1)+1..3(I-1)+3   (erlamsa_main.erl:179, erlamsa_rnd.erl:65)
      pr = parent;
      rng_skip(pr, 3 * (p.first_case + i - 1));
      gen = gen0; lt.e_pri = pri0; lt.e_meta = meta0; c.nfs = nfs0;
    } else {
      int mask;
      setup_run(p.cfg, p.seeds[3 * i], p.seeds[3 * i + 1], p.seeds[3 * i + 2], pr, gen, mask, lt.e_pri, lt.e_meta, c.nfs);
    }
#ifdef EH_PROF
#define EH_PH(k) do { uint64_t now_ = __builtin_readcyclecounter(); if (l == 0) { atomicAdd(&p.prof[2 * (64 + (k))], (unsigned long long)(now_ - ph0)); atomicAdd(&p.prof[2 * (64 + (k)) + 1], 1ull); } ph0 = now_; } while (0)
    uint64_t ph0 = __builtin_readcyclecounter();
#else
#define EH_PH(k) do {} while (0)
#endif
    int64_t t1 = (int64_t)rng_erand(pr, 99999), t2 = (int64_t)rng_erand(pr, 99999), t3 = (int64_t)rng_erand(pr, 99999);
    c.rng.draws = 0;
    rng_seed(c.rng, t1, t2, t3);                                        // worker: erlamsa_rnd:seed(ThreadSeed) :183

    uint64_t o0 = p.coff[p.corpus_first + i], o1 = p.coff[p.corpus_first + i + 1];
    o0 = uni64(o0); o1 = uni64(o1);
    EH_PH(0);
    c.gen_pending = 0;                                                                               // DataGen() :185
    if (gen == G_DIRECT) gen_direct(c, p.corpus + o0, (uint32_t)(o1 - o0));
    else if (gen == G_RANDOM) gen_random(c);
    else if (gen == G_FILE) { c.gen_e1 = rng_erand(c.rng, (uint32_t)p.n_paths) - 1; c.gen_pending = G_FILE; }          // file_streamer :108-110
    else { c.gen_e1 = rng_erand(c.rng, (uint32_t)p.n_paths) - 1; c.gen_e2 = rng_erand(c.rng, (uint32_t)p.n_paths) - 1; c.gen_pending = G_JUMP; }   // jump_streamer :138
    EH_PH(1);

    if (c.status == CASE_OK) {
      // choose_pattern_fun (erlamsa_patterns.erl:431-434) + choose_pri
      uint32_t r = rng_rand(c.rng, (uint32_t)p.cfg.pat_total);
      int pat = p.cfg.npat > 0 ? p.cfg.pat_id[p.cfg.npat - 1] : -1;
      for (int k = 0; k < p.cfg.npat; k++) {
        if (r == 0 || r < p.cfg.pat_pri[k]) { pat = p.cfg.pat_id[k]; break; }
        r -= p.cfg.pat_pri[k];
      }
      if (pat < 0) c.status = CASE_CRASHED; else run_patterns(c, lt, pat);  // Pat(Ll, CurMuta, Meta) :189
    }

    EH_PH(2);
    // ---- erlamsa_out:output/4: concatenate the written blocks into the output arena
    wave_sync();                                                        // (the last pieces were written by lane 0 just now)
    total = 0; base = 0;
    if (c.status == CASE_OK) for (int k = 0; k < c.nem; k++) total += blk_load(c.em, k).len;
    if (total > 0) {
      if (l == 0) base = atomicAdd(p.out_cursor, (unsigned long long)((total + 15) & ~15ull));
      base = uni64(base);
      if (base + total > p.out_cap) { c.status = CASE_ARENA_FULL; total = 0; }
    }
    if (total > 0) {
      uint64_t pos = base;
      for (int k = 0; k < c.nem; k++) { Blk b = blk_load(c.em, k); wave_copy(p.out + pos, (cbptr)b.ptr, b.len); pos += b.len; }
    }
    // the case's meta trace goes behind its output in the arena
    unsigned long long tbase = 0;
    if (c.trace && c.ntrace > 0) {
      wave_sync();
      if (l == 0) tbase = atomicAdd(p.out_cursor, (unsigned long long)((c.ntrace + 15u) & ~15u));
      tbase = uni64(tbase);
      if (tbase + c.ntrace > p.out_cap) { c.ntrace = 0; if (c.status == CASE_OK) { c.status = CASE_ARENA_FULL; total = 0; } }
      else wave_copy(p.out + tbase, c.trace, c.ntrace);
    }
    EH_PH(3);
#ifndef HIPEMU
    __builtin_amdgcn_s_setprio(0);                                      // (mux_fuzzers raises it for cases that run long)
#endif
    if (c.co_posted && l == 0) atomicAdd(&p.board->posters, 0xFFFFFFFFu);   // (- 1)
    if (l == 0) {
      atomicAdd(p.ticket + 3, 1ull);                                        // cases of the pass that are done (the lingering wavefronts watch it)
      atomicAdd(p.ticket + 5, (unsigned long long)total);                   // the batch's totals on the device (eh_result_summary: one small copy instead of n lengths and n statuses)
      atomicAdd(p.ticket + CTR_STATUS + (c.status >= 0 && c.status < 6 ? c.status : 1), 1ull);
    }
    // larger areas the case borrowed go back to the pool (the output has been copied out of them)
    wave_sync();
    if (c.nchunk > 0) ws_release_to(c, 0);
    if (l == 0) {
      p.out_off[i] = base; p.out_len[i] = total; p.status[i] = c.status;
      p.draws[i] = c.rng.draws; p.lastm[i] = c.status == CASE_OVERFLOW ? -c.ovf_line : c.lastm; p.cycles[i] = __builtin_readcyclecounter() - tick0;
      p.peak[i] = c.ws_peak + c.ws_top;
      if (p.flags & EH_FLAG_META_TRACE) { p.trace_off[i] = tbase; p.trace_len[i] = c.trace ? c.ntrace : 0u; }
    }
    case_ticks += __builtin_amdgcn_s_memrealtime() - case_t0;
    wave_sync();
  }
  // The last workgroup to leave writes the batch's totals into page-locked host memory: a host loop over many batches then needs no
  // copy call per batch (a hipMemcpy of a few bytes from a device full of persistent workgroups took the bench's host thread 60 - 90
  // ms per step, during which it launched nothing).  Every wavefront's counter updates are complete before its own increment
  // returns (the returned value is waited for with vmcnt(0)), so the one that sees all the others reads final values.
  wave_sync();
  unsigned long long left = 0;
  if (l == 0) {
    atomicAdd(p.ticket + 7, __builtin_amdgcn_s_memrealtime() - wg_t0);
    atomicAdd(p.ticket + CTR_OCC, case_ticks);
    atomicAdd(p.ticket + CTR_OCC + 1, linger_ticks);
  }
  if (l == 0) left = atomicAdd(p.ticket + 6, 1ull);
  if (uni64(left) + 1 == (unsigned long long)gridDim.x && p.summary_out) {
    if (l < 6) sys_store(p.summary_out + 3 + l, co_ld64(p.ticket + CTR_STATUS + l));
    if (l == 6) { sys_store(p.summary_out + 1, co_ld64(p.ticket + 5)); sys_store(p.summary_out + 2, p.n); }
    if (l == 7) { sys_store(p.summary_out + 10, co_ld64(p.ticket + 7)); sys_store(p.summary_out + 11, co_ld64(p.ticket + CTR_OCC)); sys_store(p.summary_out + 12, co_ld64(p.ticket + CTR_OCC + 1)); sys_store(p.summary_out + 13, (unsigned long long)gridDim.x); }
    wave_sync();
    if (l == 0) sys_store(p.summary_out + 9, p.batch_seq);
  }
}

// kernel-level self tests of the byte movers (driven by tests/test_gpu_primitives.py)
// =============================================================================================
// EH_FLAG_ORDERED_OUTPUT: compaction of the completion-ordered arena into case order
// =============================================================================================
// ord_off[i] = sum of out_len[0..i): one wavefront, 64 cases per step (n / 64 shuffle scans)
__global__ void __launch_bounds__(64) eh_order_scan_kernel(const uint64_t* out_len_, uint64_t* ord_off_, uint64_t n) {
  cqptr out_len = (cqptr)out_len_; qptr ord_off = (qptr)ord_off_;
  const int l = EH_LANE;
  uint64_t run = 0;
  for (uint64_t base = 0; base < n; base += 64) {
    uint64_t i = base + (uint64_t)l;
    uint64_t v = i < n ? out_len[i] : 0;
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      uint64_t t = ((uint64_t)(uint32_t)__shfl_up((int)(uint32_t)(inc >> 32), d) << 32) | (uint32_t)__shfl_up((int)(uint32_t)inc, d);
      if (l >= d) inc += t;
    }
    if (i < n) ord_off[i] = run + inc - v;
    run += uni64(((uint64_t)(uint32_t)__shfl((int)(uint32_t)(inc >> 32), 63) << 32) | (uint32_t)__shfl((int)(uint32_t)inc, 63));
  }
  if (l == 0) ord_off[n] = run;
}
// one wavefront per case (grid-stride): dst[ord_off[i] ..) = src[out_off[i] ..)
__global__ void __launch_bounds__(64) eh_order_gather_kernel(const uint8_t* src_, uint8_t* dst_, const uint64_t* out_off_, const uint64_t* out_len_,
                                                             const uint64_t* ord_off_, uint64_t n, uint64_t ord_base) {
  cbptr src = (cbptr)src_; bptr dst = (bptr)dst_; cqptr out_off = (cqptr)out_off_, out_len = (cqptr)out_len_, ord_off = (cqptr)ord_off_;
  for (uint64_t i = blockIdx.x; i < n; i += gridDim.x) {
    uint64_t len = uni64(out_len[i]), so = uni64(out_off[i]), d0 = uni64(ord_off[i]) - ord_base;
    uint64_t done = 0;
    while (done < len) { uint32_t c = len - done > 0x40000000ull ? 0x40000000u : (uint32_t)(len - done); wave_copy_raw(dst + d0 + done, src + so + done, c); done += c; }
  }
}

__global__ void __launch_bounds__(64) eh_test_copy_kernel(uint8_t* buf_, const uint32_t* jobs_, uint32_t njobs, uint32_t* eq_out_) {
  bptr buf = (bptr)buf_; cwptr jobs = (cwptr)jobs_; wptr eq_out = (wptr)eq_out_;
  g_ctx.p = nullptr;                                             // (not a case: nothing is posted for other wavefronts, co_run)
  // job = {kind, dst_off, src_off, n, plen}: kind 0 copy, 1 periodic fill, 2 equal
  for (uint32_t j = blockIdx.x; j < njobs; j += gridDim.x) {
    uint32_t kind = jobs[5 * j], d = jobs[5 * j + 1], s = jobs[5 * j + 2], n = jobs[5 * j + 3], pl = jobs[5 * j + 4];
    if (kind == 0) wave_copy_raw(buf + d, buf + s, n);
    else if (kind == 1) wave_fill_periodic(buf + d, buf + s, pl, n);
    else if (kind == 2) { bool e = wave_equal_raw(buf + d, buf + s, n); if (EH_LANE == 0) eq_out[j] = e ? 1u : 0u; }
    else {  // kind 3: mask window self check over buf[s, s+n) with window base d (multiple of 64)
      MaskWin<4> a, b; a.p = buf + s; a.L = n; b.p = buf + s; b.L = n;
      mw_load(a, d, LexCls()); mw_load_ref(b, d, LexCls());
      uint32_t bad = 0;
      for (int k = 0; k < 4; k++) bad += a.m[k] != b.m[k] ? 1u : 0u;
      bad += a.inrange != b.inrange ? 1u : 0u;
      MaskWin<4> c2, d2; c2.p = buf + s; c2.L = n; d2.p = buf + s; d2.L = n;
      mw_load(c2, d, DelimCls()); mw_load_ref(d2, d, DelimCls());
      for (int k = 0; k < 4; k++) bad += c2.m[k] != d2.m[k] ? 1u : 0u;
      bad = wave_sum(bad);
      if (EH_LANE == 0) eq_out[j] = bad;
    }
    wave_sync();
  }
}

// Self test of eh_zlib.h: op 0..2 compress in[0..n) as ZF_RAW / ZF_GZIP / ZF_ZLIB, op 4 / 5 = zlib:gunzip / zlib:inflate semantics
// (z_uncompress_size + z_uncompress_write); res[0] = bytes written, res[1] = 1 ok / 0 "raises".
__global__ void __launch_bounds__(64) eh_test_zlib_kernel(int op, const uint8_t* in_, uint64_t n, uint8_t* out_, uint64_t cap, uint8_t* scratch_, uint64_t* res_) {
  cbptr in = (cbptr)in_; bptr out = (bptr)out_, scratch = (bptr)scratch_; qptr res = (qptr)res_;
  g_ctx.p = nullptr;
  uint64_t len = 0; int ok = 1;
  if (op <= 2) { len = z_compress((EH_G ZDef*)scratch, op, in, n, out, cap); ok = len != 0; }
  else {
    EH_G ZInf* zi = (EH_G ZInf*)scratch; uint64_t off = 0;
    ok = z_uncompress_size(zi, op - 3, in, n, &len, &off);
    if (ok && len > cap) ok = 0;
    if (ok) ok = z_uncompress_write(zi, op - 3, in, n, off, out, len);
  }
  if (EH_LANE == 0) { res[0] = len; res[1] = (uint64_t)ok; }
}

// =============================================================================================
// host side
// =============================================================================================
static const MutaInfo MUTAS[M_COUNT] = {
    {"sgm", 10, 1}, {"js", 3, 1},  {"uw", 1, 1},   {"ui", 2, 1},  {"ab", 1, 1},  {"ad", 1, 1},  {"tr2", 1, 1}, {"td", 1, 1},
    {"num", 3, 1},  {"ts1", 2, 1}, {"tr", 2, 1},   {"ts2", 2, 1}, {"bd", 1, 1},  {"bei", 1, 1}, {"bed", 1, 1}, {"bf", 1, 1},
    {"bi", 1, 1},   {"ber", 1, 1}, {"br", 1, 1},   {"sp", 1, 1},  {"sr", 1, 1},  {"sd", 1, 1},  {"snand", 1, 1}, {"srnd", 1, 1},
    {"ld", 1, 1},   {"lds", 1, 1}, {"lr2", 1, 1},  {"lri", 1, 1}, {"lr", 1, 1},  {"ls", 1, 1},  {"lp", 1, 1},  {"lis", 1, 1},
    {"lrs", 1, 1},  {"ft", 2, 1},  {"fn", 1, 1},   {"fo", 2, 1},  {"len", 2, 1}, {"b64", 7, 1}, {"uri", 1, 1}, {"zip", 1, 1},
    {"nil", 0, 1}};
static const PatInfo PATS[P_COUNT] = {{"od", 1, 1}, {"nd", 2, 1}, {"bu", 1, 1}, {"sk", 2, 1}, {"sz", 2, 1},
                                      {"cs", 1, 1}, {"ar", 1, 1}, {"cp", 1, 1}, {"co", 0, 1}, {"nu", 0, 1}};

}  // namespace eh

using namespace eh;

// Host code fills KParams.  In the device pass of this file its pointer members are address-space qualified (eh_common.h), and
// the host functions are type-checked there too: dp(x) converts to whatever the member is.
template <class T> struct DevPtr { T* p; template <class U> operator U() const { return (U)p; } };
template <class T> static inline DevPtr<T> dp(T* p) { return DevPtr<T>{p}; }

// the work-area pool of a device (see pool_acquire)
struct DevPool {
  int device = 0; uint64_t work_cap = 0, big = 0, pool_bytes_opt = 0;
  int refs = 0;
  int ntiers = 0;                                       // tiers 1..ntiers
  uint8_t* base[POOL_TIERS + 1] = {}; uint64_t stride[POOL_TIERS + 1] = {}, cap[POOL_TIERS + 1] = {}; uint32_t cnt[POOL_TIERS + 1] = {};
  uint32_t* d_rings = nullptr; uint32_t* ring[POOL_TIERS + 1] = {}; unsigned long long* d_ctr = nullptr;
  CoBoard* d_board = nullptr;                           // cooperative execution: the board every context of the device posts on (eh_common.h)
};

struct eh_ctx {
  int device = 0;
  int cus = 0;
  std::string err;
  bool configured = false;
  DevConfig cfg;
  uint64_t max_case_bytes = 0, out_capacity_opt = 0, work_budget = 0;
  uint64_t fuse_stream_min = 16384, pool_bytes_opt = 0, dl_chunk = 256ull << 20;   // eh_options (ABI 4)
  uint32_t max_slots_opt = 0, flags = 0;
  int needs_paths = 0;                                  // corpus entries the configured generators need (file: 1, jump: 2)
  KParams* d_params = nullptr;                          // argument block of eh_mutate_kernel
  // eh_result_download: case-ordered chunks are gathered on the device into two bounce buffers; chunk k goes over PCIe
  // while chunk k+1 is gathered
  uint8_t* d_bounce[2] = {nullptr, nullptr}; uint64_t bounce_cap = 0, bounce_chunk = 0; hipStream_t dl_gather = nullptr, dl_copy = nullptr;
  hipEvent_t ev_g[2] = {nullptr, nullptr}, ev_c[2] = {nullptr, nullptr};
  uint8_t* d_out2 = nullptr; uint64_t out2_cap = 0;   // EH_FLAG_ORDERED_OUTPUT: second arena (case order)
  uint64_t* d_ord = nullptr; uint64_t ord_cap = 0;      // ordered offsets (n + 1)
  bool ordered = false;                                 // the last batch's results are in case order
  // corpus
  uint8_t* d_corpus = nullptr; uint64_t* d_coff = nullptr; bool own_corpus = false;
  uint64_t own_corpus_cap = 0, own_coff_cap = 0;         // capacity of the owned buffers: eh_corpus_upload reuses them when they fit
  uint64_t n_corpus = 0, corpus_bytes = 0;
  std::vector<uint64_t> h_coff;  // host copy of offsets (for totals)
  // work areas: a pool shared by every context of the device with the same sizes (DevPool below)
  DevPool* pool = nullptr;
  uint8_t* d_slots = nullptr; uint64_t slot_stride = 0, slot_cap = 0; uint32_t nslots = 0;   // a slot per workgroup of a batch
  uint64_t big_case_bytes = 0;
  // outputs
  uint8_t* d_out = nullptr; uint64_t out_cap = 0;
  uint64_t* d_off = nullptr; uint64_t* d_len = nullptr; int32_t* d_status = nullptr; uint64_t* d_draws = nullptr; int32_t* d_lastm = nullptr; uint64_t* d_cycles = nullptr; uint64_t* d_peak = nullptr; uint64_t* d_toff = nullptr; uint32_t* d_tlen = nullptr;
  uint64_t res_cap = 0;
  unsigned long long* d_counters = nullptr;  // [0] ticket, [1] out cursor, [2] input bytes, [3] cases done, [4] lingering wavefronts, [5] output bytes, [6] workgroups that left, [8, 264) EH_PROF, [264, 270) cases by status
  unsigned long long* h_sum = nullptr; uint64_t batch_seq = 0;   // page-locked: the last batch's totals, written by the kernel (KParams::summary_out)
  RunState* d_run = nullptr;
  int64_t* d_seeds = nullptr; uint64_t seeds_cap = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // request coalescing (eh_submit / eh_flush / eh_poll)
  std::recursive_mutex co_lock;
  std::condition_variable_any co_cv; bool co_collecting = false;   // a thread is waiting for / downloading the in-flight batch (lock released)
  bool co_internal = false;                             // co_launch / co_collect are calling the batch entry points themselves
  uint64_t co_next_ticket = 1, co_flush_cases = 4096, co_flush_bytes = 64ull << 20;
  std::vector<uint8_t> co_data; std::vector<uint64_t> co_off{0}; std::vector<int64_t> co_seeds; std::vector<uint64_t> co_tickets;   // pending batch
  bool co_inflight = false; std::vector<uint64_t> co_inflight_tickets;                  // launched, results still on the device
  struct CoResult { std::vector<uint8_t> out; int32_t status; };
  std::vector<uint64_t> co_cancelled;                                                   // in-flight tickets nobody will poll
  std::map<uint64_t, CoResult> co_done;                                                 // downloaded, not polled yet
  void* comm = nullptr; int comm_rank = 0, comm_n = 1;   // RCCL communicator of this context's device (eh_comm_init / eh_comm_init_local)
  hipStream_t last_stream = nullptr;
  hipStream_t own_stream = nullptr;                     // eh_stream: a stream of the context's own (non-blocking), made on first request
  uint64_t last_n = 0, last_in_bytes = 0;
  bool have_result = false;
};

#define HIPCHK(ctx, call)                                                                             \
  do {                                                                                                \
    hipError_t e_ = (call);                                                                           \
    if (e_ != hipSuccess) {                                                                           \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                 \
      return EH_E_HIP;                                                                                \
    }                                                                                                 \
  } while (0)

// A context that has coalesced requests pending or in flight (eh_submit .. eh_poll) belongs to the coalescer: a batch,
// a corpus or a configuration from outside would overwrite the buffers and options those requests run with.
#define CO_GUARD(ctx)                                                                                              \
  std::lock_guard<std::recursive_mutex> co_guard_((ctx)->co_lock);                                                  \
  if (!(ctx)->co_internal && ((ctx)->co_inflight || !(ctx)->co_tickets.empty())) {                                  \
    (ctx)->err = "coalesced requests are pending on this context: eh_flush and eh_poll them first, or use a context of its own"; \
    return EH_E_STATE;                                                                                              \
  }

namespace {

// erlamsa_utils:sort_by_priority/1 (erlamsa_utils.erl:113-117): lists:sort/2 with a strict '>' as the ordering fun — the
// order among equal priorities is whatever stdlib's merge sort gives (eh_otp_sort.h, the one implementation).
struct PItem { uint32_t pri; int id; };
typedef std::vector<PItem> PL;
static PL otp_sort_desc_strict(const PL& in) {
  otp::ListsSort<PItem> srt([](const PItem& a, const PItem& b) { return a.pri > b.pri; });
  return srt.sort(in);
}

static int lookup(const char* name, bool muta) {
  if (muta) { for (int i = 0; i < M_COUNT; i++) if (!strcmp(name, MUTAS[i].name)) return i; }
  else { for (int i = 0; i < P_COUNT; i++) if (!strcmp(name, PATS[i].name)) return i; }
  return -1;
}
// "-m"/"-p" list syntax: erlamsa_cmdparse:string_to_actions/3 (erlamsa_cmdparse.erl:232-257)
static int parse_actions(eh_ctx* ctx, const char* s, bool muta, std::vector<long>& pri /* -1 = not selected */) {
  int count = muta ? (int)M_COUNT : (int)P_COUNT;
  pri.assign(count, -1);
  if (!s) { for (int i = 0; i < count; i++) pri[i] = muta ? MUTAS[i].pri : PATS[i].pri; return EH_OK; }
  std::string str(s); size_t pos = 0;
  while (pos <= str.size()) {
    size_t e = str.find(',', pos); if (e == std::string::npos) e = str.size();
    std::string tok = str.substr(pos, e - pos); pos = e + 1;
    if (tok.empty()) continue;
    size_t eq = tok.find('=');
    std::string name = eq == std::string::npos ? tok : tok.substr(0, eq);
    int id = lookup(name.c_str(), muta);
    if (id < 0) { ctx->err = std::string("No such ") + (muta ? "mutations" : "patterns") + ": " + name; return EH_E_INVALID; }
    long p = muta ? MUTAS[id].pri : PATS[id].pri;
    if (eq != std::string::npos) { char* endp = nullptr; p = strtol(tok.c_str() + eq + 1, &endp, 10); if (*endp || p < 0 || p > 1000000) { ctx->err = "Invalid priority: " + tok; return EH_E_INVALID; } }
    pri[id] = p;
  }
  return EH_OK;
}

static int ensure_results(eh_ctx* ctx, uint64_t n) {
  if (n <= ctx->res_cap) return EH_OK;
  if (ctx->d_off) { (void)hipFree(ctx->d_off); (void)hipFree(ctx->d_len); (void)hipFree(ctx->d_status); (void)hipFree(ctx->d_draws); (void)hipFree(ctx->d_lastm); (void)hipFree(ctx->d_cycles); (void)hipFree(ctx->d_peak); (void)hipFree(ctx->d_toff); (void)hipFree(ctx->d_tlen); }
  HIPCHK(ctx, hipMalloc(&ctx->d_cycles, n * 8));
  HIPCHK(ctx, hipMalloc(&ctx->d_peak, n * 8));
  HIPCHK(ctx, hipMalloc(&ctx->d_toff, n * 8));
  HIPCHK(ctx, hipMalloc(&ctx->d_tlen, n * 4));
  HIPCHK(ctx, hipMalloc(&ctx->d_off, n * 8));
  HIPCHK(ctx, hipMalloc(&ctx->d_len, n * 8));
  HIPCHK(ctx, hipMalloc(&ctx->d_status, n * 4));
  HIPCHK(ctx, hipMalloc(&ctx->d_draws, n * 8));
  HIPCHK(ctx, hipMalloc(&ctx->d_lastm, n * 4));
  ctx->res_cap = n;
  return EH_OK;
}

// ---- the work-area pool ------------------------------------------------------------------------------------------
// One pool per (device, max_case_bytes, big_case_bytes, pool_bytes), shared by all contexts that ask for those sizes and
// freed with the last of them.  Tier 0 has one slot (block tables + max_case_bytes) for every wavefront the device can hold
// of eh_mutate_kernel — workgroups of any number of batches in flight, on any streams, take their slot from it — and
// tiers 1.. hold the larger areas (4x per tier up to big_case_bytes) a wavefront borrows for a case that outgrew its slot.
static std::mutex g_pool_lock;
static std::vector<DevPool*> g_pools;

static void pool_free(DevPool* pl) {
  (void)hipSetDevice(pl->device);
  (void)hipDeviceSynchronize();
  for (int t = 1; t <= pl->ntiers; t++) if (pl->base[t]) (void)hipFree(pl->base[t]);
  if (pl->d_rings) (void)hipFree(pl->d_rings);
  if (pl->d_ctr) (void)hipFree(pl->d_ctr);
  if (pl->d_board) (void)hipFree(pl->d_board);
  delete pl;
}
static void pool_release(eh_ctx* ctx) {
  if (!ctx->pool) return;
  std::lock_guard<std::mutex> g(g_pool_lock);
  DevPool* pl = ctx->pool; ctx->pool = nullptr;
  if (--pl->refs > 0) return;
  for (size_t i = 0; i < g_pools.size(); i++) if (g_pools[i] == pl) { g_pools.erase(g_pools.begin() + i); break; }
  pool_free(pl);
}
static int pool_acquire(eh_ctx* ctx, uint64_t work_cap, uint64_t big, uint64_t pool_bytes_opt) {
  if (ctx->pool && ctx->pool->work_cap == work_cap && ctx->pool->big == big && ctx->pool->pool_bytes_opt == pool_bytes_opt) return EH_OK;
  pool_release(ctx);
  std::lock_guard<std::mutex> g(g_pool_lock);
  for (DevPool* q : g_pools)
    if (q->device == ctx->device && q->work_cap == work_cap && q->big == big && q->pool_bytes_opt == pool_bytes_opt) { q->refs++; ctx->pool = q; return EH_OK; }
  DevPool* pl = new (std::nothrow) DevPool();
  if (!pl) return EH_E_NOMEM;
  pl->device = ctx->device; pl->work_cap = work_cap; pl->big = big; pl->pool_bytes_opt = pool_bytes_opt;
  pl->cap[0] = work_cap;
  // twice the area per tier (most cases that outgrow what they hold need less than twice as much); the last tier has `big`.
  // The pool's memory (pool_bytes; default: a quarter of the free memory, at most 64 GiB) is split over the tiers in the
  // proportions the bench workload asks for them (BASELINE configs[2], 4 MiB slots: by far most borrowers are fuse calls on
  // blocks of 0.1-1 MB, whose tables take 5-20 MB): 12 / 23 / 17 / 7 / 10 / 10 / 9 / 12 percent from the smallest tier up.
  // Every tier holds at least one area, at most 4096; a tier that cannot be allocated at all ends the ladder.
  // Round 4: the large tiers get three times what they had.  With 2 areas of 1 GiB and 2 of 512 MiB for six passes in flight the
  // heaviest cases of the passes queued for each other (eh_pool_stats of a bench run: ~130 waits of 1.8 G ticks each for the last
  // tier alone, 1.8 - 3.2 T ticks of sleeping wavefronts per run), which stretched exactly the cases that decide how long a pass lasts.
  uint32_t SHARE[POOL_TIERS] = {12, 23, 17, 7, 10, 10, 9, 12};
  if (const char* ev = getenv("EH_POOL_SHARES")) {                                 // (measurement knob: eight numbers, smallest tier first)
    uint32_t v[POOL_TIERS]; int k = 0;
    for (const char* q = ev; *q && k < POOL_TIERS; ) { char* end = nullptr; unsigned long x = strtoul(q, &end, 10); if (end == q) break; v[k++] = (uint32_t)x; q = *end ? end + 1 : end; }
    if (k == POOL_TIERS) { bool ok = true; for (int i = 0; i < k; i++) ok = ok && v[i] >= 1 && v[i] <= 1000; if (ok) memcpy(SHARE, v, sizeof(v)); }
  }
  size_t fr = 0, tot = 0;
  hipError_t e = hipSuccess;
  uint64_t pool_bytes = pool_bytes_opt;
  if (!pool_bytes) { pool_bytes = 16ull << 30; if (hipMemGetInfo(&fr, &tot) == hipSuccess) { pool_bytes = (uint64_t)fr / 4; if (pool_bytes > (64ull << 30)) pool_bytes = 64ull << 30; if (pool_bytes < (1ull << 30)) pool_bytes = 1ull << 30; } }
  uint64_t caps[POOL_TIERS]; int nt = 0;
  for (uint64_t cap = work_cap; cap < big && nt < POOL_TIERS; ) { cap = (cap * 2 < big && nt < POOL_TIERS - 1) ? cap * 2 : big; caps[nt++] = cap; }   // the last tier always has the full size
  uint32_t share_sum = 0;
  for (int k = 0; k < nt; k++) share_sum += SHARE[k == nt - 1 ? POOL_TIERS - 1 : k];
  for (int k = 0; k < nt; k++) {
    uint64_t stride_t = (caps[k] + 255) & ~255ull;
    uint64_t cnt = pool_bytes / share_sum * SHARE[k == nt - 1 ? POOL_TIERS - 1 : k] / stride_t; if (cnt < 1) cnt = 1; if (cnt > 4096) cnt = 4096;
    if (ctx->cus < 64 && cnt > 2) cnt = 2;                                         // (the emulator)
    int t = pl->ntiers + 1;
    for (; cnt >= 1; cnt /= 2) { e = hipMalloc(&pl->base[t], stride_t * cnt); if (e == hipSuccess) break; pl->base[t] = nullptr; }
    if (e != hipSuccess) break;
    pl->stride[t] = stride_t; pl->cap[t] = caps[k]; pl->cnt[t] = (uint32_t)cnt; pl->ntiers++;
  }
  uint64_t nring = 4;
  for (int t = 1; t <= pl->ntiers; t++) nring += pl->cnt[t];
  std::vector<uint32_t> init(nring);
  std::vector<unsigned long long> ctr(64, 0ull);
  if (hipMalloc(&pl->d_rings, nring * 4) != hipSuccess || hipMalloc(&pl->d_ctr, 64 * 8) != hipSuccess) { pool_free(pl); ctx->err = "work-area pool: out of device memory"; return EH_E_NOMEM; }
  uint64_t o = 0;
  for (int t = 1; t <= pl->ntiers; t++) {
    pl->ring[t] = pl->d_rings + o;
    for (uint32_t k = 0; k < pl->cnt[t]; k++) init[o + k] = k;
    ctr[2 * t] = 0; ctr[2 * t + 1] = pl->cnt[t];                                   // pop tickets, push tickets
    o += pl->cnt[t];
  }
  if (hipMemcpy(pl->d_rings, init.data(), nring * 4, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(pl->d_ctr, ctr.data(), 64 * 8, hipMemcpyHostToDevice) != hipSuccess) { pool_free(pl); ctx->err = "work-area pool: hipMemcpy failed"; return EH_E_HIP; }
  if (hipMalloc(&pl->d_board, sizeof(CoBoard)) != hipSuccess || hipMemset(pl->d_board, 0, sizeof(CoBoard)) != hipSuccess) { pool_free(pl); ctx->err = "work-area pool: no memory for the cooperation board"; return EH_E_NOMEM; }
  pl->refs = 1; g_pools.push_back(pl); ctx->pool = pl;
  return EH_OK;
}

// From which size a loop of a case is posted for other wavefronts, and in what chunks (KParams co_*).  A chunk costs its runner an
// agent-scope acquire and a release (microseconds, and the release writes back whatever is dirty in its XCD's L2), so chunks are
// hundreds of kilobytes; measurement knobs: EH_CO_COPY_MIN / EH_CO_COPY_CHUNK (bytes), EH_CO_FB_MIN / EH_CO_FB_CHUNK (positions).
static uint32_t co_env(const char* name, uint32_t dflt) { const char* v = getenv(name); if (!v || !*v) return dflt; unsigned long long x = strtoull(v, nullptr, 10); return x > 0 && x < (1ull << 31) ? (uint32_t)x : dflt; }
static void co_defaults(KParams* p, int cus) {
  const bool emu = cus < 64;                                                    // (the emulator's tests are small: they take the posted paths too)
  p->co_copy_min = co_env("EH_CO_COPY_MIN", emu ? (8u << 10) : (2u << 20));
  p->co_copy_chunk = co_env("EH_CO_COPY_CHUNK", emu ? (2u << 10) : (256u << 10));
  p->co_fb_min = co_env("EH_CO_FB_MIN", emu ? (16u << 10) : (512u << 10));
  p->co_fb_chunk = co_env("EH_CO_FB_CHUNK", emu ? (4u << 10) : (64u << 10)) & ~1023u;
  { const char* v = getenv("EH_CO_LINGER"); p->co_linger = v && *v ? (uint32_t)strtoul(v, nullptr, 10) : CO_LINGER; p->pad_co = 0; }
  if (p->co_copy_chunk < 1024) p->co_copy_chunk = 1024;
  if (p->co_fb_chunk < 1024) p->co_fb_chunk = 1024;
}

// Device memory for batches of up to `n` cases over `in_bytes` input bytes: result arrays, the work-area pool, output
// arena.  Buffers only ever grow, so after eh_reserve (or a first batch of the largest size) no launch allocates or
// frees — hipFree synchronises the whole device.
static int reserve(eh_ctx* ctx, uint64_t n, uint64_t in_bytes) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int rc = ensure_results(ctx, n ? n : 1);
  if (rc) return rc;
  uint64_t work_cap = ctx->max_case_bytes ? ctx->max_case_bytes : (8ull << 20);
  uint64_t big = ctx->big_case_bytes ? ctx->big_case_bytes : (32 * work_cap < (1024ull << 20) ? 32 * work_cap : (1024ull << 20));
  if (big < work_cap) big = work_cap;
  rc = pool_acquire(ctx, work_cap, big, ctx->pool_bytes_opt);
  if (rc) return rc;
  // a slot per workgroup of a batch: one workgroup per wavefront the device holds (EH_WAVES_PER_SIMD), or max_slots
  uint32_t want_slots = ctx->max_slots_opt ? ctx->max_slots_opt : (uint32_t)ctx->cus * 4u * EH_WAVES_PER_SIMD;
  if (want_slots > n) want_slots = (uint32_t)(n ? n : 1);
  uint64_t stride = (SLOT_TABLE_BYTES + work_cap + 255) & ~255ull;
  if (!ctx->d_slots || ctx->nslots < want_slots || ctx->slot_cap != work_cap) {
    if (ctx->d_slots) (void)hipFree(ctx->d_slots);
    ctx->d_slots = nullptr;
    HIPCHK(ctx, hipMalloc(&ctx->d_slots, stride * want_slots));
    ctx->nslots = want_slots; ctx->slot_cap = work_cap; ctx->slot_stride = stride;
  }
  uint64_t want_out = ctx->out_capacity_opt ? ctx->out_capacity_opt : (8 * (in_bytes ? in_bytes : ctx->corpus_bytes) + (2048ull << 20));
  if (!ctx->d_out || ctx->out_cap < want_out) {
    if (ctx->d_out) (void)hipFree(ctx->d_out);
    ctx->d_out = nullptr;
    HIPCHK(ctx, hipMalloc(&ctx->d_out, want_out));
    ctx->out_cap = want_out;
  }
  return EH_OK;
}

static int launch(eh_ctx* ctx, int mode, const int64_t seed[3], uint64_t first_case, uint64_t corpus_first, uint64_t n, hipStream_t st) {
  if (!ctx->configured || !ctx->d_corpus) { ctx->err = "configure and load a corpus first"; return EH_E_STATE; }
  if (corpus_first + n > ctx->n_corpus || (mode == 0 && first_case < 1)) { ctx->err = "case range outside the corpus"; return EH_E_INVALID; }
  // make_generator_fun (erlamsa_gen.erl:215-224) drops `file` without paths and `jump` with fewer than two: said aloud here
  if (ctx->n_corpus < (uint64_t)ctx->needs_paths || ctx->n_corpus > 0xFFFFFFFFull) { ctx->err = "generator file needs one corpus entry (path), jump two"; return EH_E_INVALID; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  uint64_t in_bytes = 0;
  if (!ctx->h_coff.empty()) in_bytes = ctx->h_coff[corpus_first + n] - ctx->h_coff[corpus_first];
  int rc = reserve(ctx, n, in_bytes);
  if (rc) return rc;
  if (!ctx->d_counters) { HIPCHK(ctx, hipMalloc(&ctx->d_counters, 4096)); HIPCHK(ctx, hipMalloc(&ctx->d_run, sizeof(RunState))); HIPCHK(ctx, hipMalloc(&ctx->d_params, sizeof(KParams))); }
  if (!ctx->h_sum) { HIPCHK(ctx, hipHostMalloc((void**)&ctx->h_sum, 128, hipHostMallocDefault)); memset(ctx->h_sum, 0, 128); }

  KParams p;
  memset(&p, 0, sizeof(p));
  p.corpus = dp(ctx->d_corpus); p.coff = dp(ctx->d_coff); p.corpus_first = corpus_first; p.n_paths = ctx->n_corpus; p.n = n; p.first_case = first_case;
  p.mode = mode; p.run = dp(ctx->d_run); p.seeds = dp(ctx->d_seeds); p.cfg = ctx->cfg;
  const DevPool* pl = ctx->pool;
  p.work_cap = pl->work_cap;
  p.work_budget = ctx->work_budget;                                            // 0 = no budget (the default)
  p.fuse_stream_min = ctx->fuse_stream_min;
  p.out = dp(ctx->d_out); p.out_cap = ctx->out_cap; p.out_cursor = dp(ctx->d_counters + 1);
  p.out_off = dp(ctx->d_off); p.out_len = dp(ctx->d_len); p.status = dp(ctx->d_status); p.draws = dp(ctx->d_draws); p.lastm = dp(ctx->d_lastm); p.cycles = dp(ctx->d_cycles); p.peak = dp(ctx->d_peak); p.trace_off = dp(ctx->d_toff); p.trace_len = dp(ctx->d_tlen); p.flags = ctx->flags;
  p.ticket = dp(ctx->d_counters); p.in_bytes = dp(ctx->d_counters + 2); p.prof = dp(ctx->d_counters + 8);   // counters [8, 264) = prof
  p.slot_base = dp(ctx->d_slots); p.slot_stride = ctx->slot_stride;
  p.ntiers = pl->ntiers; p.pool_ctr = dp(pl->d_ctr); p.pool_cap[0] = pl->work_cap;
  p.board = (ctx->flags & EH_FLAG_NO_COOP) ? dp((CoBoard*)nullptr) : dp(pl->d_board);
  co_defaults(&p, ctx->cus);
  p.summary_out = dp(ctx->h_sum); p.batch_seq = ++ctx->batch_seq;
  if (n == 0) { memset(ctx->h_sum, 0, 128); ctx->h_sum[9] = ctx->batch_seq; }      // (no mutate kernel: nothing else would write it)
  for (int t = 1; t <= pl->ntiers; t++) { p.pool_base[t] = dp(pl->base[t]); p.pool_stride[t] = pl->stride[t]; p.pool_cap[t] = pl->cap[t]; p.pool_cnt[t] = pl->cnt[t]; p.pool_ring[t] = dp(pl->ring[t]); }
  // persistent workgroups, each pulling cases from the ticket counter.  Batches in flight on several streams may
  // oversubscribe the device: the dispatcher starts a batch's workgroups as those of earlier ones leave.
  uint32_t grid0 = ctx->nslots < n ? ctx->nslots : (uint32_t)n;

  // counters, argument block and run state by ONE one-wavefront kernel (no fill / copy kernels of the runtime: see eh_prologue_kernel)
  hipLaunchKernelGGL(eh_prologue_kernel, dim3(1), dim3(64), 0, st, p, mode == 0 ? seed[0] : 0, mode == 0 ? seed[1] : 0, mode == 0 ? seed[2] : 0,
                     ctx->d_run, ctx->d_params, ctx->d_counters);
  HIPCHK(ctx, hipEventRecord(ctx->ev0, st));
  if (n > 0) hipLaunchKernelGGL(eh_mutate_kernel, dim3(grid0), dim3(64), 0, st, (const KParams*)ctx->d_params);
  HIPCHK(ctx, hipEventRecord(ctx->ev1, st));
  HIPCHK(ctx, hipGetLastError());
  ctx->ordered = false;
  if ((ctx->flags & EH_FLAG_ORDERED_OUTPUT) && n > 0) {
    // second arena + ordered offsets, then scan + gather on the same stream; the arenas swap roles so that
    // eh_result_device / eh_result_download see one contiguous case-ordered buffer
    if (!ctx->d_out2 || ctx->out2_cap < ctx->out_cap) {
      if (ctx->d_out2) (void)hipFree(ctx->d_out2);
      ctx->d_out2 = nullptr;
      HIPCHK(ctx, hipMalloc(&ctx->d_out2, ctx->out_cap));
      ctx->out2_cap = ctx->out_cap;
    }
    if (!ctx->d_ord || ctx->ord_cap < n + 1) {
      if (ctx->d_ord) (void)hipFree(ctx->d_ord);
      ctx->d_ord = nullptr;
      HIPCHK(ctx, hipMalloc(&ctx->d_ord, (n + 1) * 8));
      ctx->ord_cap = n + 1;
    }
    hipLaunchKernelGGL(eh_order_scan_kernel, dim3(1), dim3(64), 0, st, ctx->d_len, ctx->d_ord, n);
    uint32_t g = (uint32_t)ctx->cus * 32u; if (g > n) g = (uint32_t)n;
    hipLaunchKernelGGL(eh_order_gather_kernel, dim3(g), dim3(64), 0, st, ctx->d_out, ctx->d_out2, ctx->d_off, ctx->d_len, ctx->d_ord, n, 0ull);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_off, ctx->d_ord, n * 8, hipMemcpyDeviceToDevice, st));
    HIPCHK(ctx, hipGetLastError());
    uint8_t* t = ctx->d_out; ctx->d_out = ctx->d_out2; ctx->d_out2 = t;
    uint64_t tc = ctx->out_cap; ctx->out_cap = ctx->out2_cap; ctx->out2_cap = tc;
    ctx->ordered = true;
  }
  ctx->last_stream = st; ctx->last_n = n; ctx->last_in_bytes = in_bytes; ctx->have_result = true;
  return EH_OK;
}

}  // namespace

extern "C" {

uint32_t eh_abi_version(void) { return EH_ABI_VERSION; }
int eh_mutator_count(void) { return M_COUNT; }
const char* eh_mutator_name(int id) { return id >= 0 && id < M_COUNT ? MUTAS[id].name : nullptr; }
int eh_mutator_default_pri(int id) { return id >= 0 && id < M_COUNT ? MUTAS[id].pri : -1; }
int eh_mutator_on_gpu(int id) { return id >= 0 && id < M_COUNT ? MUTAS[id].on_gpu : 0; }
int eh_pattern_count(void) { return P_COUNT; }
const char* eh_pattern_name(int id) { return id >= 0 && id < P_COUNT ? PATS[id].name : nullptr; }
int eh_pattern_default_pri(int id) { return id >= 0 && id < P_COUNT ? PATS[id].pri : -1; }
int eh_pattern_on_gpu(int id) { return id >= 0 && id < P_COUNT ? PATS[id].on_gpu : 0; }
const char* eh_kernel_name(void) { return "eh_mutate_kernel"; }
int eh_meta_atom_count(void) { return AT_COUNT; }
const char* eh_meta_atom_name(int id) {
  static const char* const names[AT_COUNT] = {
#define EH_ATOM_NAME(n) #n,
      EH_ATOMS(EH_ATOM_NAME)
#undef EH_ATOM_NAME
  };
  return id >= 0 && id < AT_COUNT ? names[id] : nullptr;
}

const char* eh_strerror(int code) {
  switch (code) {
    case EH_OK: return "ok";
    case EH_E_INVALID: return "invalid argument";
    case EH_E_NODEVICE: return "no usable HIP device";
    case EH_E_HIP: return "HIP runtime error";
    case EH_E_NOMEM: return "out of memory";
    case EH_E_STATE: return "wrong call order";
    case EH_E_AGAIN: return "request not launched yet";
    case EH_E_UNSUPPORTED: return "mutator/pattern not available on the GPU in this build";
  }
  return "unknown error";
}
const char* eh_last_error(eh_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
uint64_t eh_last_error_copy(eh_ctx* ctx, char* buf, uint64_t cap) {
  if (!buf || !cap) return 0;
  if (!ctx) { buf[0] = 0; return 0; }
  std::lock_guard<std::recursive_mutex> g(ctx->co_lock);            // the coalescer's calls set the text under this lock
  uint64_t n = ctx->err.size() < cap - 1 ? ctx->err.size() : cap - 1;
  memcpy(buf, ctx->err.data(), n); buf[n] = 0;
  return n;
}

// Batches in flight on several HIP streams (one context each) only run side by side when every stream gets a hardware queue of its
// own; the runtime maps streams onto GPU_MAX_HW_QUEUES of them (default 4: with six passes in flight pairs of them serialise,
// 21.7 instead of 33.7 GB/s, profiles/r04_bench_q4.json).  The variable is read when the HIP runtime initialises, i.e. at the first
// HIP call of the process: set here, when the library is loaded, unless the host has chosen a value itself.  (A process that has
// used HIP before it loads this library - a torch that came first - keeps what it started with: INTEGRATION.md section 3.)
__attribute__((constructor)) static void eh_runtime_defaults() {
  // Only when the variable is not set, and only here - at load time, before this library has made a HIP call.  A host that loads the
  // library into a process with threads already running (a BEAM: setenv is not safe against a concurrent getenv) exports
  // GPU_MAX_HW_QUEUES=8 itself before it starts; then nothing is written (note 0).
  if (!getenv("GPU_MAX_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

int eh_create(int device, eh_ctx** out) {
  if (!out) return EH_E_INVALID;
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return EH_E_NODEVICE;
  eh_ctx* ctx = new (std::nothrow) eh_ctx();
  if (!ctx) return EH_E_NOMEM;
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete ctx; return EH_E_NODEVICE; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { delete ctx; return EH_E_NODEVICE; }
  ctx->cus = prop.multiProcessorCount;
  // (Until round 5 the kernel recursed - nested scheduler calls of b64 / sgm / js - and eh_create raised hipLimitStackSize to 6 KiB per lane
  // for the whole process.  The scheduler is a loop over explicit frames now (eh_device.h mux_fuzzers): the kernel's stack is the 1.8 KiB
  // the assembler adds up, and no limit of the process is touched.)
  // (the constant tables - AS183 powers, CRC-32, funny_unicode/0 - are initialised in the code object: eh_device.h, eh_zlib.h)
  if (hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
    eh_destroy(ctx);
    return EH_E_HIP;
  }
  *out = ctx;
  return EH_OK;
}

void eh_destroy(eh_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  if (ctx->comm) { ehcomm::Api* a = ehcomm::api(); if (a->h) (void)a->CommDestroy(ctx->comm); ctx->comm = nullptr; }
  if (ctx->own_corpus) { (void)hipFree(ctx->d_corpus); (void)hipFree(ctx->d_coff); }
  pool_release(ctx);
  (void)hipFree(ctx->d_slots); (void)hipFree(ctx->d_out); (void)hipFree(ctx->d_off); (void)hipFree(ctx->d_len);
  (void)hipFree(ctx->d_status); (void)hipFree(ctx->d_draws); (void)hipFree(ctx->d_lastm); (void)hipFree(ctx->d_cycles); (void)hipFree(ctx->d_peak); (void)hipFree(ctx->d_toff); (void)hipFree(ctx->d_tlen); (void)hipFree(ctx->d_counters); if (ctx->h_sum) (void)hipHostFree(ctx->h_sum);
  (void)hipFree(ctx->d_run); (void)hipFree(ctx->d_seeds);
  for (int k = 0; k < 2; k++) { if (ctx->d_bounce[k]) (void)hipFree(ctx->d_bounce[k]); if (ctx->ev_g[k]) (void)hipEventDestroy(ctx->ev_g[k]); if (ctx->ev_c[k]) (void)hipEventDestroy(ctx->ev_c[k]); }
  if (ctx->dl_gather) (void)hipStreamDestroy(ctx->dl_gather);
  if (ctx->dl_copy) (void)hipStreamDestroy(ctx->dl_copy);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  if (ctx->d_params) (void)hipFree(ctx->d_params);
  if (ctx->d_out2) (void)hipFree(ctx->d_out2);
  if (ctx->d_ord) (void)hipFree(ctx->d_ord);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  delete ctx;
}

static int co_collect(eh_ctx* ctx);
int eh_configure(eh_ctx* ctx, const eh_options* o) {
  if (!ctx || !o) return EH_E_INVALID;
  if (o->abi_version != EH_ABI_VERSION) { ctx->err = "eh_options.abi_version mismatch"; return EH_E_INVALID; }
  // Requests that are still PENDING run with the configuration in force when they are launched, so a new one is refused
  // (eh_flush them first).  A batch that is already in flight carries its configuration with it (launch() copies it into the
  // kernel's argument block) but owns the context's result buffers, which the next launch may re-size: it is collected first.
  std::lock_guard<std::recursive_mutex> co_guard_(ctx->co_lock);
  if (!ctx->co_internal && !ctx->co_tickets.empty()) {
    ctx->err = "coalesced requests are pending on this context: eh_flush them first, or use a context of its own";
    return EH_E_STATE;
  }
  if (!ctx->co_internal && ctx->co_inflight) { int rcc = co_collect(ctx); if (rcc) return rcc; }
  if (o->sequence_muta) {
    ctx->err = "sequence_muta (--consequtive-mutators, erlamsa_main.erl:223-235) chains the mutator scores from case to case: a batch of independent cases cannot keep that order; route this run to the BEAM path";
    return EH_E_UNSUPPORTED;
  }
  DevConfig cfg;
  memset(&cfg, 0, sizeof(cfg));
  std::vector<long> mp, pp;
  int rc = parse_actions(ctx, o->mutations, true, mp); if (rc) return rc;
  rc = parse_actions(ctx, o->patterns, false, pp); if (rc) return rc;
  for (int i = 0; i < M_COUNT; i++) if (mp[i] >= 0) {
    if (!MUTAS[i].on_gpu) { ctx->err = std::string("mutator '") + MUTAS[i].name + "' is not available on the GPU in this build"; return EH_E_UNSUPPORTED; }
    cfg.sel_name[cfg.nsel] = (uint8_t)i; cfg.sel_pri[cfg.nsel] = (uint32_t)mp[i]; cfg.nsel++;
  }
  // make_pattern (erlamsa_patterns.erl:416-428): foldl prepends => reversed table order, then sort_by_priority
  PL pl;
  for (int i = P_COUNT - 1; i >= 0; i--) if (pp[i] >= 0) {
    if (!PATS[i].on_gpu) { ctx->err = std::string("pattern '") + PATS[i].name + "' is not available on the GPU in this build"; return EH_E_UNSUPPORTED; }
    pl.push_back({(uint32_t)pp[i], i});
  }
  PL sp = otp_sort_desc_strict(pl);
  cfg.npat = (int)sp.size();
  for (size_t i = 0; i < sp.size(); i++) { cfg.pat_id[i] = (uint8_t)sp[i].id; cfg.pat_pri[i] = sp[i].pri; cfg.pat_total += (int)sp[i].pri; }
  if (cfg.npat == 0) { ctx->err = "no patterns selected"; return EH_E_INVALID; }
  // generators: table order of erlamsa_gen:generators/0 is random(1) ... direct(500)
  long gr = 1, gd = 500, gf = -1, gj = -1;
  if (o->generators) {
    gr = -1; gd = -1;
    std::string str(o->generators); size_t pos = 0;
    while (pos <= str.size()) {
      size_t e = str.find(',', pos); if (e == std::string::npos) e = str.size();
      std::string tok = str.substr(pos, e - pos); pos = e + 1;
      if (tok.empty()) continue;
      size_t eq = tok.find('=');
      std::string name = eq == std::string::npos ? tok : tok.substr(0, eq);
      long p = -1; if (eq != std::string::npos) p = strtol(tok.c_str() + eq + 1, nullptr, 10);
      if (name == "random") gr = p < 0 ? 1 : p; else if (name == "direct") gd = p < 0 ? 500 : p;
      else if (name == "file") gf = p < 0 ? 1000 : p; else if (name == "jump") gj = p < 0 ? 100 : p;      // Paths = the corpus entries
      else { ctx->err = "generator '" + name + "' is host-side I/O and not part of the GPU path"; return EH_E_UNSUPPORTED; }
    }
  }
  PL gl;
  if (gr >= 0) gl.push_back({(uint32_t)gr, G_RANDOM});                  // table order of erlamsa_gen:generators/0 :250-257
  if (gj >= 0) gl.push_back({(uint32_t)gj, G_JUMP});
  if (gd >= 0) gl.push_back({(uint32_t)gd, G_DIRECT});
  if (gf >= 0) gl.push_back({(uint32_t)gf, G_FILE});
  ctx->needs_paths = gj >= 0 ? 2 : (gf >= 0 ? 1 : 0);
  if (gl.empty()) { ctx->err = "No generators!"; return EH_E_INVALID; }
  PL sg = otp_sort_desc_strict(gl);
  cfg.ngen = (int)sg.size();
  for (size_t i = 0; i < sg.size(); i++) { cfg.gen_id[i] = (uint8_t)sg[i].id; cfg.gen_pri[i] = sg[i].pri; cfg.gen_total += (int)sg[i].pri; }
  double bs = o->blockscale == 0 ? 1.0 : o->blockscale;
  cfg.max_block_scaled = (uint32_t)llround(MAX_BLOCK_SIZE * bs);
  cfg.min_block_scaled = (uint32_t)llround(MIN_BLOCK_SIZE * bs);
  snprintf(cfg.ssrf_host, sizeof(cfg.ssrf_host), "%s", o->ssrf_host ? o->ssrf_host : "localhost");
  snprintf(cfg.ssrf_port, sizeof(cfg.ssrf_port), "%d", o->ssrf_port ? o->ssrf_port : 51234);
  ctx->cfg = cfg;
  ctx->work_budget = o->max_case_work;
  ctx->big_case_bytes = o->big_case_bytes; ctx->max_case_bytes = o->max_case_bytes; ctx->out_capacity_opt = o->out_capacity; ctx->max_slots_opt = o->max_slots; ctx->flags = o->flags;
  ctx->fuse_stream_min = o->fuse_stream_min ? o->fuse_stream_min : 16384;
  ctx->pool_bytes_opt = o->pool_bytes; ctx->dl_chunk = o->download_chunk_bytes ? o->download_chunk_bytes : (256ull << 20);
  ctx->configured = true;
  return EH_OK;
}

static int set_corpus(eh_ctx* ctx, uint8_t* d, uint64_t* doff, bool own, uint64_t n, uint64_t nbytes) {
  if (ctx->own_corpus) { (void)hipFree(ctx->d_corpus); (void)hipFree(ctx->d_coff); }
  ctx->d_corpus = d; ctx->d_coff = doff; ctx->own_corpus = own; ctx->n_corpus = n; ctx->corpus_bytes = nbytes;
  return EH_OK;
}
int eh_corpus_upload(eh_ctx* ctx, const uint8_t* data, const uint64_t* off, uint64_t n) {
  if (!ctx || !off || (!data && off[n] > 0)) return EH_E_INVALID;
  CO_GUARD(ctx);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  uint64_t nbytes = off[n];
  HIPCHK(ctx, hipDeviceSynchronize());                       // no launch may still read the previous corpus
  if (ctx->own_corpus && ctx->own_corpus_cap >= nbytes && ctx->own_coff_cap >= n + 1) {      // steady state of a service: no hipMalloc / hipFree
    if (nbytes) HIPCHK(ctx, hipMemcpy(ctx->d_corpus, data, nbytes, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_coff, off, (n + 1) * 8, hipMemcpyHostToDevice));
    ctx->h_coff.assign(off, off + n + 1);
    ctx->n_corpus = n; ctx->corpus_bytes = nbytes;
    return EH_OK;
  }
  uint8_t* d = nullptr; uint64_t* doff = nullptr;
  uint64_t cap_b = nbytes + nbytes / 2 + 4096, cap_o = n + n / 2 + 16;                         // room to grow
  hipError_t e = hipMalloc(&d, cap_b);
  if (e == hipSuccess) e = hipMalloc(&doff, cap_o * 8);
  if (e == hipSuccess && nbytes) e = hipMemcpy(d, data, nbytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(doff, off, (n + 1) * 8, hipMemcpyHostToDevice);
  if (e != hipSuccess) { if (d) (void)hipFree(d); if (doff) (void)hipFree(doff); HIPCHK(ctx, e); }
  ctx->h_coff.assign(off, off + n + 1);
  int rc = set_corpus(ctx, d, doff, true, n, nbytes);
  ctx->own_corpus_cap = cap_b; ctx->own_coff_cap = cap_o;
  return rc;
}
int eh_corpus_attach(eh_ctx* ctx, const void* d_data, const void* d_off, uint64_t n, uint64_t nbytes) {
  if (!ctx || !d_off) return EH_E_INVALID;
  CO_GUARD(ctx);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ctx->h_coff.resize(n + 1);
  HIPCHK(ctx, hipMemcpy(ctx->h_coff.data(), d_off, (n + 1) * 8, hipMemcpyDeviceToHost));
  if (ctx->h_coff[n] != nbytes) { ctx->err = "off[n] != nbytes"; return EH_E_INVALID; }
  return set_corpus(ctx, (uint8_t*)d_data, (uint64_t*)d_off, false, n, nbytes);
}

// ---- multi-GPU: the arena over RCCL (eh_comm.h) ------------------------------------------------------------------------------
#define NCCLCHK(ctx, a, call)                                                                                   \
  do {                                                                                                          \
    int r_ = (call);                                                                                            \
    if (r_ != 0) { (ctx)->err = std::string(#call) + ": " + ((a)->GetErrorString ? (a)->GetErrorString(r_) : "RCCL error"); return EH_E_HIP; } \
  } while (0)

// a device temporary of the collectives below: freed on every way out of the function
struct DevTmp { void* p = nullptr; ~DevTmp() { if (p) (void)hipFree(p); } };

// an owned corpus buffer for n entries / nbytes bytes (contents undefined); keeps the one it has when that is large enough
static int corpus_own_buffers(eh_ctx* ctx, uint64_t n, uint64_t nbytes) {
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipDeviceSynchronize());                       // no launch may still read the previous corpus
  if (ctx->own_corpus && ctx->own_corpus_cap >= nbytes && ctx->own_coff_cap >= n + 1) { ctx->n_corpus = n; ctx->corpus_bytes = nbytes; return EH_OK; }
  uint8_t* d = nullptr; uint64_t* doff = nullptr;
  const uint64_t cap_b = nbytes + 4096, cap_o = n + 16;
  hipError_t e = hipMalloc(&d, cap_b);
  if (e == hipSuccess) e = hipMalloc(&doff, cap_o * 8);
  if (e != hipSuccess) { if (d) (void)hipFree(d); if (doff) (void)hipFree(doff); HIPCHK(ctx, e); }
  int rc = set_corpus(ctx, d, doff, true, n, nbytes);
  ctx->own_corpus_cap = cap_b; ctx->own_coff_cap = cap_o;
  return rc;
}

int eh_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess && n > 0 ? n : 0; }
int eh_comm_unique_id(uint8_t id[128]) {
  if (!id) return EH_E_INVALID;
  ehcomm::Api* a = ehcomm::api();
  if (!a->h) return EH_E_UNSUPPORTED;
  ehcomm::UniqueId u;
  if (a->GetUniqueId(&u) != 0) return EH_E_HIP;
  memcpy(id, u.internal, 128);
  return EH_OK;
}
int eh_comm_init(eh_ctx* ctx, const uint8_t id[128], int rank, int nranks) {
  if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return EH_E_INVALID;
  ehcomm::Api* a = ehcomm::api();
  if (!a->h) { ctx->err = a->err; return EH_E_UNSUPPORTED; }
  if (ctx->comm) { (void)a->CommDestroy(ctx->comm); ctx->comm = nullptr; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  ehcomm::UniqueId u; memcpy(u.internal, id, 128);
  NCCLCHK(ctx, a, a->CommInitRank(&ctx->comm, nranks, u, rank));
  ctx->comm_rank = rank; ctx->comm_n = nranks;
  return EH_OK;
}
int eh_comm_init_local(eh_ctx** ctxs, int nctx) {
  if (!ctxs || nctx < 1 || nctx > 64) return EH_E_INVALID;
  for (int i = 0; i < nctx; i++) if (!ctxs[i]) return EH_E_INVALID;
  ehcomm::Api* a = ehcomm::api();
  if (!a->h) { ctxs[0]->err = a->err; return EH_E_UNSUPPORTED; }
  std::vector<int> devs(nctx); std::vector<ehcomm::Comm> comms(nctx, nullptr);
  for (int i = 0; i < nctx; i++) {
    devs[i] = ctxs[i]->device;
    for (int j = 0; j < i; j++) if (devs[j] == devs[i]) { ctxs[0]->err = "eh_comm_init_local: one context per device (contexts of one device share the corpus with eh_corpus_device / eh_corpus_attach)"; return EH_E_INVALID; }
    if (ctxs[i]->comm) { (void)a->CommDestroy(ctxs[i]->comm); ctxs[i]->comm = nullptr; }
  }
  NCCLCHK(ctxs[0], a, a->CommInitAll(comms.data(), nctx, devs.data()));
  for (int i = 0; i < nctx; i++) { ctxs[i]->comm = comms[i]; ctxs[i]->comm_rank = i; ctxs[i]->comm_n = nctx; }
  return EH_OK;
}
int eh_comm_destroy(eh_ctx* ctx) {
  if (!ctx) return EH_E_INVALID;
  if (ctx->comm) { ehcomm::Api* a = ehcomm::api(); if (a->h) (void)a->CommDestroy(ctx->comm); ctx->comm = nullptr; }
  ctx->comm_rank = 0; ctx->comm_n = 1;
  return EH_OK;
}

// One process per GPU.  Root hands over the arena (host pointers), the others pass NULL / 0; afterwards every rank's context holds
// the corpus as after eh_corpus_upload.  Three collectives on the context's stream: the sizes, the bytes, the offsets.
int eh_corpus_broadcast(eh_ctx* ctx, int root, const uint8_t* data, const uint64_t* off, uint64_t n) {
  if (!ctx) return EH_E_INVALID;
  CO_GUARD(ctx);
  if (!ctx->comm) { ctx->err = "eh_comm_init first"; return EH_E_STATE; }
  if (root < 0 || root >= ctx->comm_n) return EH_E_INVALID;
  const bool is_root = ctx->comm_rank == root;
  if (is_root && (!off || (!data && off[n] > 0))) return EH_E_INVALID;
  ehcomm::Api* a = ehcomm::api();
  HIPCHK(ctx, hipSetDevice(ctx->device));
  void* stv = nullptr; int rc = eh_stream(ctx, &stv); if (rc) return rc;
  hipStream_t st = (hipStream_t)stv;
  DevTmp hdr_mem;
  HIPCHK(ctx, hipMalloc(&hdr_mem.p, 16));
  uint64_t* d_hdr = (uint64_t*)hdr_mem.p;
  uint64_t hdr[2] = {is_root ? n : 0, is_root ? off[n] : 0};
  if (is_root) HIPCHK(ctx, hipMemcpy(d_hdr, hdr, 16, hipMemcpyHostToDevice));
  NCCLCHK(ctx, a, a->Broadcast(d_hdr, d_hdr, 2, ehcomm::kUint64, root, ctx->comm, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  HIPCHK(ctx, hipMemcpy(hdr, d_hdr, 16, hipMemcpyDeviceToHost));
  const uint64_t cn = hdr[0], nbytes = hdr[1];
  // (A rank that fails locally between the header and the payload - out of memory here - leaves its peers in the next collective:
  // there is no way to call them back.  The caller's answer to an error from this function is eh_comm_destroy on every rank,
  // ncclCommAbort's effect on the peers, and a fresh communicator; include/erlamsa_hip.h says so.)
  rc = corpus_own_buffers(ctx, cn, nbytes); if (rc) return rc;
  if (is_root) {
    if (nbytes) HIPCHK(ctx, hipMemcpy(ctx->d_corpus, data, nbytes, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_coff, off, (cn + 1) * 8, hipMemcpyHostToDevice));
  }
  if (nbytes) NCCLCHK(ctx, a, a->Broadcast(ctx->d_corpus, ctx->d_corpus, nbytes, ehcomm::kUint8, root, ctx->comm, st));
  NCCLCHK(ctx, a, a->Broadcast(ctx->d_coff, ctx->d_coff, cn + 1, ehcomm::kUint64, root, ctx->comm, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  ctx->h_coff.resize(cn + 1);
  if (is_root) ctx->h_coff.assign(off, off + cn + 1);
  else HIPCHK(ctx, hipMemcpy(ctx->h_coff.data(), ctx->d_coff, (cn + 1) * 8, hipMemcpyDeviceToHost));
  if (ctx->h_coff[cn] != nbytes) { ctx->err = "eh_corpus_broadcast: the offsets that arrived do not end at the arena's size"; return EH_E_HIP; }
  return EH_OK;
}

// Every rank gives ITS shard - the same number of entries and of bytes on every rank (BASELINE configs[4]: fixed-size seeds) -;
// afterwards every context holds the shards in rank order as one corpus.  In place: the shard is uploaded where it belongs and
// ncclAllGather fills in the others, so each xGMI link carries 1/nranks of the arena.
int eh_corpus_allgather(eh_ctx* ctx, const uint8_t* data, const uint64_t* off, uint64_t n_local) {
  if (!ctx || !off || (!data && off[n_local] > 0)) return EH_E_INVALID;
  CO_GUARD(ctx);
  if (!ctx->comm) { ctx->err = "eh_comm_init first"; return EH_E_STATE; }
  ehcomm::Api* a = ehcomm::api();
  HIPCHK(ctx, hipSetDevice(ctx->device));
  void* stv = nullptr; int rc = eh_stream(ctx, &stv); if (rc) return rc;
  hipStream_t st = (hipStream_t)stv;
  const int R = ctx->comm_n, me = ctx->comm_rank;
  const uint64_t B = off[n_local];
  // do all ranks bring the same shape?
  DevTmp sz_mem;
  HIPCHK(ctx, hipMalloc(&sz_mem.p, 16 * (size_t)R));
  uint64_t* d_sz = (uint64_t*)sz_mem.p;
  uint64_t mine[2] = {n_local, B};
  HIPCHK(ctx, hipMemcpy(d_sz + 2 * me, mine, 16, hipMemcpyHostToDevice));
  NCCLCHK(ctx, a, a->AllGather(d_sz + 2 * me, d_sz, 2, ehcomm::kUint64, ctx->comm, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  std::vector<uint64_t> sz(2 * (size_t)R);
  HIPCHK(ctx, hipMemcpy(sz.data(), d_sz, 16 * (size_t)R, hipMemcpyDeviceToHost));
  for (int r = 0; r < R; r++) if (sz[2 * r] != n_local || sz[2 * r + 1] != B) { ctx->err = "eh_corpus_allgather: shards differ in entries or bytes between ranks (use eh_corpus_broadcast)"; return EH_E_INVALID; }
  const uint64_t cn = n_local * (uint64_t)R, nbytes = B * (uint64_t)R;
  rc = corpus_own_buffers(ctx, cn, nbytes); if (rc) return rc;
  if (B) HIPCHK(ctx, hipMemcpy(ctx->d_corpus + B * me, data, B, hipMemcpyHostToDevice));
  std::vector<uint64_t> lo(n_local ? n_local : 1);
  for (uint64_t i = 0; i < n_local; i++) lo[i] = off[i] + B * me;
  if (n_local) HIPCHK(ctx, hipMemcpy(ctx->d_coff + n_local * me, lo.data(), n_local * 8, hipMemcpyHostToDevice));
  HIPCHK(ctx, hipMemcpy(ctx->d_coff + cn, &nbytes, 8, hipMemcpyHostToDevice));
  if (B) NCCLCHK(ctx, a, a->AllGather(ctx->d_corpus + B * me, ctx->d_corpus, B, ehcomm::kUint8, ctx->comm, st));
  if (n_local) NCCLCHK(ctx, a, a->AllGather(ctx->d_coff + n_local * me, ctx->d_coff, n_local, ehcomm::kUint64, ctx->comm, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  ctx->h_coff.resize(cn + 1);
  HIPCHK(ctx, hipMemcpy(ctx->h_coff.data(), ctx->d_coff, (cn + 1) * 8, hipMemcpyDeviceToHost));
  return EH_OK;
}

// One process, several GPUs (a BEAM node): the corpus loaded on ctxs[root] goes to the devices of all the others (grouped
// ncclBroadcast over the communicators of eh_comm_init_local).
int eh_corpus_broadcast_local(eh_ctx** ctxs, int nctx, int root) {
  if (!ctxs || nctx < 1 || root < 0 || root >= nctx) return EH_E_INVALID;
  for (int i = 0; i < nctx; i++) if (!ctxs[i] || !ctxs[i]->comm || ctxs[i]->comm_n != nctx || ctxs[i]->comm_rank != i) { if (ctxs[0]) ctxs[0]->err = "eh_comm_init_local on these contexts, in this order, first"; return EH_E_STATE; }
  eh_ctx* r = ctxs[root];
  if (!r->d_corpus) { r->err = "no corpus loaded on the root context"; return EH_E_STATE; }
  ehcomm::Api* a = ehcomm::api();
  const uint64_t cn = r->n_corpus, nbytes = r->corpus_bytes;
  std::vector<hipStream_t> sts(nctx);
  for (int i = 0; i < nctx; i++) {
    eh_ctx* x = ctxs[i];
    if (i != root) { int rc = corpus_own_buffers(x, cn, nbytes); if (rc) return rc; }
    void* stv = nullptr; int rc = eh_stream(x, &stv); if (rc) return rc;
    sts[i] = (hipStream_t)stv;
  }
  NCCLCHK(r, a, a->GroupStart());
  // (whatever fails inside the group, the group is ENDED before the function returns: an open group swallows every later call)
  int grc = EH_OK; eh_ctx* gbad = r; std::string gerr;
  for (int i = 0; i < nctx && grc == EH_OK; i++) {
    eh_ctx* x = ctxs[i];
    hipError_t he = hipSetDevice(x->device);
    if (he != hipSuccess) { grc = EH_E_HIP; gbad = x; gerr = std::string("hipSetDevice: ") + hipGetErrorString(he); break; }
    int nr = nbytes ? a->Broadcast(x->d_corpus, x->d_corpus, nbytes, ehcomm::kUint8, root, x->comm, sts[i]) : 0;
    if (nr == 0) nr = a->Broadcast(x->d_coff, x->d_coff, cn + 1, ehcomm::kUint64, root, x->comm, sts[i]);
    if (nr != 0) { grc = EH_E_HIP; gbad = x; gerr = std::string("ncclBroadcast: ") + (a->GetErrorString ? a->GetErrorString(nr) : "RCCL error"); }
  }
  const int ge = a->GroupEnd();
  if (grc != EH_OK) { gbad->err = gerr; if (gbad != ctxs[0]) ctxs[0]->err = gerr; return grc; }
  if (ge != 0) { r->err = std::string("ncclGroupEnd: ") + (a->GetErrorString ? a->GetErrorString(ge) : "RCCL error"); return EH_E_HIP; }
  for (int i = 0; i < nctx; i++) {
    eh_ctx* x = ctxs[i];
    HIPCHK(x, hipSetDevice(x->device));
    HIPCHK(x, hipStreamSynchronize(sts[i]));
    if (i != root) x->h_coff = r->h_coff;
  }
  return EH_OK;
}

int eh_reserve(eh_ctx* ctx, uint64_t max_cases) {
  if (!ctx) return EH_E_INVALID;
  if (!ctx->configured || !ctx->d_corpus) { ctx->err = "configure and load a corpus first"; return EH_E_STATE; }
  return reserve(ctx, max_cases, 0);
}

int eh_fuzz_batch(eh_ctx* ctx, const int64_t seed[3], uint64_t first_case, uint64_t corpus_first, uint64_t n, void* stream) {
  if (!ctx || !seed) return EH_E_INVALID;
  CO_GUARD(ctx);
  return launch(ctx, 0, seed, first_case, corpus_first, n, (hipStream_t)stream);
}
int eh_fuzz_calls(eh_ctx* ctx, const int64_t* seeds, uint64_t corpus_first, uint64_t n, void* stream) {
  if (!ctx || !seeds) return EH_E_INVALID;
  CO_GUARD(ctx);
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (n > ctx->seeds_cap) {
    if (ctx->d_seeds) (void)hipFree(ctx->d_seeds);
    ctx->d_seeds = nullptr;
    HIPCHK(ctx, hipMalloc(&ctx->d_seeds, n * 24));
    ctx->seeds_cap = n;
  }
  // synchronous copy: the caller's `seeds` may be freed or reused as soon as this call returns
  if (n) { HIPCHK(ctx, hipStreamSynchronize((hipStream_t)stream)); HIPCHK(ctx, hipMemcpy(ctx->d_seeds, seeds, n * 24, hipMemcpyHostToDevice)); }
  int64_t dummy[3] = {0, 0, 0};
  return launch(ctx, 1, dummy, 1, corpus_first, n, (hipStream_t)stream);
}
// ---- request coalescing -------------------------------------------------------------------------------------
// Brings the launched batch's results to the host (one download) and files them under their tickets.
static int co_collect(eh_ctx* ctx) {
  // (the caller holds co_lock exactly once.)  The wait for the batch and the download happen WITHOUT the lock, so that
  // other threads keep submitting to the next batch meanwhile; a second collector waits for the first.
  while (ctx->co_collecting) ctx->co_cv.wait(ctx->co_lock);
  if (!ctx->co_inflight) return EH_OK;
  ctx->co_collecting = true;
  const std::vector<uint64_t> tickets = ctx->co_inflight_tickets;
  const uint64_t n = tickets.size();
  ctx->co_lock.unlock();
  uint64_t in_b = 0, out_b = 0, nc = 0;
  std::vector<uint8_t> data; std::vector<uint64_t> off(n + 1); std::vector<int32_t> st(n ? n : 1);
  int rc = eh_result_totals(ctx, &in_b, &out_b, &nc);
  if (!rc && nc != n) { ctx->err = "coalescer: the context's last batch is not the one that was flushed"; rc = EH_E_STATE; }
  if (!rc) { data.resize(out_b ? out_b : 1); rc = eh_result_download(ctx, data.data(), data.size(), off.data(), st.data()); }
  ctx->co_lock.lock();
  if (!rc) {
    for (uint64_t i = 0; i < n; i++) {
      bool dropped = false;
      for (uint64_t t : ctx->co_cancelled) if (t == tickets[i]) dropped = true;
      if (dropped) continue;
      eh_ctx::CoResult r; r.out.assign(data.begin() + off[i], data.begin() + off[i + 1]); r.status = st[i];
      ctx->co_done.emplace(tickets[i], std::move(r));
    }
    ctx->co_cancelled.clear();
    ctx->co_inflight = false; ctx->co_inflight_tickets.clear();
  }
  ctx->co_collecting = false;
  ctx->co_cv.notify_all();
  return rc;
}
// co_lock held.  Launches the pending batch (after collecting the previous one: one result buffer per context).
static int co_launch(eh_ctx* ctx) {
  if (ctx->co_tickets.empty()) return EH_OK;
  int rc = co_collect(ctx);
  if (rc) return rc;
  if (ctx->co_tickets.empty()) return EH_OK;                        // (co_collect lets other threads in: one of them has launched the batch)
  const uint64_t n = ctx->co_tickets.size();
  ctx->co_internal = true;
  rc = eh_corpus_upload(ctx, ctx->co_data.data(), ctx->co_off.data(), n);
  if (!rc) rc = eh_fuzz_calls(ctx, ctx->co_seeds.data(), 0, n, nullptr);
  ctx->co_internal = false;
  if (rc) return rc;
  ctx->co_inflight = true; ctx->co_inflight_tickets.swap(ctx->co_tickets);
  ctx->co_tickets.clear(); ctx->co_data.clear(); ctx->co_off.assign(1, 0); ctx->co_seeds.clear();
  return EH_OK;
}
int eh_coalesce_limits(eh_ctx* ctx, uint64_t flush_cases, uint64_t flush_bytes) {
  if (!ctx || !flush_cases || !flush_bytes) return EH_E_INVALID;
  std::lock_guard<std::recursive_mutex> g(ctx->co_lock);
  ctx->co_flush_cases = flush_cases; ctx->co_flush_bytes = flush_bytes;
  return EH_OK;
}
int eh_submit(eh_ctx* ctx, const uint8_t* data, uint64_t len, const int64_t seed[3], uint64_t* ticket) {
  if (!ctx || !seed || !ticket || (!data && len)) return EH_E_INVALID;
  if (!ctx->configured) { ctx->err = "configure first"; return EH_E_STATE; }
  std::lock_guard<std::recursive_mutex> g(ctx->co_lock);
  const size_t d0 = ctx->co_data.size(), o0 = ctx->co_off.size(), s0 = ctx->co_seeds.size(), t0 = ctx->co_tickets.size();
  const uint64_t mine = ctx->co_next_ticket;
  try {
    ctx->co_data.insert(ctx->co_data.end(), data, data + len);
    ctx->co_off.push_back(ctx->co_data.size());
    ctx->co_seeds.insert(ctx->co_seeds.end(), seed, seed + 3);
    ctx->co_tickets.push_back(mine);
  } catch (const std::bad_alloc&) { ctx->co_data.resize(d0); ctx->co_off.resize(o0); ctx->co_seeds.resize(s0); ctx->co_tickets.resize(t0); return EH_E_NOMEM; }
  ctx->co_next_ticket++;                                  // (before any launch: co_collect lets other submitters in)
  if (ctx->co_tickets.size() >= ctx->co_flush_cases || ctx->co_data.size() >= ctx->co_flush_bytes) {
    int rc = co_launch(ctx);
    // a launch that fails takes this request back out (the caller gets no ticket for it); the others stay queued for the next flush
    if (rc) { std::string why = ctx->err; (void)eh_cancel(ctx, mine); ctx->err = why; return rc; }
  }
  *ticket = mine;
  return EH_OK;
}
int eh_cancel(eh_ctx* ctx, uint64_t ticket) {
  if (!ctx) return EH_E_INVALID;
  std::lock_guard<std::recursive_mutex> g(ctx->co_lock);
  if (ctx->co_done.erase(ticket)) return EH_OK;
  for (size_t i = 0; i < ctx->co_tickets.size(); i++) if (ctx->co_tickets[i] == ticket) {      // not launched yet: leaves the batch
    const uint64_t a = ctx->co_off[i], b = ctx->co_off[i + 1];
    ctx->co_data.erase(ctx->co_data.begin() + a, ctx->co_data.begin() + b);
    ctx->co_off.erase(ctx->co_off.begin() + i + 1);
    for (size_t k = i + 1; k < ctx->co_off.size(); k++) ctx->co_off[k] -= b - a;
    ctx->co_seeds.erase(ctx->co_seeds.begin() + 3 * i, ctx->co_seeds.begin() + 3 * i + 3);
    ctx->co_tickets.erase(ctx->co_tickets.begin() + i);
    return EH_OK;
  }
  for (uint64_t t : ctx->co_inflight_tickets) if (t == ticket) { ctx->co_cancelled.push_back(ticket); return EH_OK; }   // dropped when the batch is collected
  ctx->err = "unknown or already consumed ticket";
  return EH_E_INVALID;
}
int eh_flush(eh_ctx* ctx) {
  if (!ctx) return EH_E_INVALID;
  std::lock_guard<std::recursive_mutex> g(ctx->co_lock);
  return co_launch(ctx);
}
int eh_poll(eh_ctx* ctx, uint64_t ticket, uint8_t* out, uint64_t cap, uint64_t* out_len, int32_t* status) {
  if (!ctx || !out_len || !status) return EH_E_INVALID;
  std::lock_guard<std::recursive_mutex> g(ctx->co_lock);
  auto it = ctx->co_done.find(ticket);
  if (it == ctx->co_done.end()) {
    for (uint64_t t : ctx->co_tickets) if (t == ticket) return EH_E_AGAIN;
    bool launched = false;
    for (uint64_t t : ctx->co_inflight_tickets) if (t == ticket) launched = true;
    if (!launched) { ctx->err = "unknown or already consumed ticket"; return EH_E_INVALID; }
    int rc = co_collect(ctx);                                       // waits for the batch
    if (rc) return rc;
    it = ctx->co_done.find(ticket);
  }
  *out_len = it->second.out.size(); *status = it->second.status;
  if (it->second.out.size() > cap || (!out && !it->second.out.empty())) { ctx->err = "eh_poll: buffer too small"; return EH_E_INVALID; }
  if (!it->second.out.empty()) memcpy(out, it->second.out.data(), it->second.out.size());
  ctx->co_done.erase(it);
  return EH_OK;
}

int eh_sync(eh_ctx* ctx) {
  if (!ctx) return EH_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipStreamSynchronize(ctx->last_stream));
  return EH_OK;
}

int eh_stream(eh_ctx* ctx, void** stream) {
  if (!ctx || !stream) return EH_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  // (streams of different priorities - high / normal / low in turn over the contexts - were measured against the convoy of passes
  // launched together: 84.6 against 84.9 GB/s, profiles/r06_stream_priorities_and_slots.txt; not kept)
  if (!ctx->own_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking));
  *stream = (void*)ctx->own_stream;
  return EH_OK;
}

int eh_corpus_device(eh_ctx* ctx, const void** d_data, const void** d_off, uint64_t* n, uint64_t* nbytes) {
  if (!ctx) return EH_E_INVALID;
  if (!ctx->d_corpus) { ctx->err = "no corpus loaded"; return EH_E_STATE; }
  if (d_data) *d_data = ctx->d_corpus;
  if (d_off) *d_off = ctx->d_coff;
  if (n) *n = ctx->n_corpus;
  if (nbytes) *nbytes = ctx->corpus_bytes;
  return EH_OK;
}

int eh_host_alloc(void** p, uint64_t bytes) {
  if (!p) return EH_E_INVALID;
  *p = nullptr;
  return hipHostMalloc(p, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? EH_OK : EH_E_NOMEM;
}
void eh_host_free(void* p) { if (p) (void)hipHostFree(p); }

int eh_result_device(eh_ctx* ctx, const uint8_t** d_data, const uint64_t** d_off, const uint64_t** d_len, const int32_t** d_status, uint64_t* total) {
  if (!ctx) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  int rc = eh_sync(ctx); if (rc) return rc;
  if (d_data) *d_data = ctx->d_out;
  if (d_off) *d_off = ctx->d_off;
  if (d_len) *d_len = ctx->d_len;
  if (d_status) *d_status = ctx->d_status;
  if (total) {
    unsigned long long cur = 0;
    if (ctx->ordered) HIPCHK(ctx, hipMemcpy(&cur, ctx->d_ord + ctx->last_n, 8, hipMemcpyDeviceToHost));   // bytes of the compact, case-ordered buffer
    else HIPCHK(ctx, hipMemcpy(&cur, ctx->d_counters + 1, 8, hipMemcpyDeviceToHost));                     // bump cursor of the completion-ordered arena (16-byte granules)
    *total = cur > ctx->out_cap ? ctx->out_cap : cur;               // the cursor runs past the arena after EH_CASE_ARENA_FULL
  }
  return EH_OK;
}
int eh_result_totals(eh_ctx* ctx, uint64_t* in_bytes, uint64_t* out_bytes, uint64_t* n_cases) {
  if (!ctx) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  int rc = eh_sync(ctx); if (rc) return rc;
  uint64_t n = ctx->last_n;
  if (out_bytes) {
    std::vector<uint64_t> len(n ? n : 1);
    if (n) HIPCHK(ctx, hipMemcpy(len.data(), ctx->d_len, n * 8, hipMemcpyDeviceToHost));
    uint64_t s = 0; for (uint64_t i = 0; i < n; i++) s += len[i];
    *out_bytes = s;
  }
  if (in_bytes) *in_bytes = ctx->last_in_bytes;
  if (n_cases) *n_cases = n;
  return EH_OK;
}
// out[0] input bytes, out[1] output bytes, out[2] cases, out[3 + s] cases that ended with status s (0..5) - summed by the kernel,
// one copy of 2 KiB: what a host loop over many small batches reads per batch instead of n lengths and n statuses.
int eh_result_summary(eh_ctx* ctx, uint64_t* out /* 9 values */) {
  if (!ctx || !out) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  if (hipEventQuery(ctx->ev1) != hipSuccess) { (void)hipGetLastError(); int rc = eh_sync(ctx); if (rc) return rc; }   // (a batch that eh_batch_done has seen end needs no runtime call at all)
  // the kernel's last workgroup wrote them into page-locked memory, stamped with the batch's number: no copy call
  volatile unsigned long long* hs = ctx->h_sum;
  if (!hs || hs[9] != ctx->batch_seq) { ctx->err = "eh_result_summary: the batch's totals have not arrived in host memory"; return EH_E_HIP; }
  out[0] = ctx->last_in_bytes; out[1] = hs[1]; out[2] = ctx->last_n;
  for (int k = 0; k < 6; k++) out[3 + k] = hs[3 + k];
  return EH_OK;
}
int eh_result_occupancy(eh_ctx* ctx, uint64_t* out /* 5 values */) {
  if (!ctx || !out) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  if (hipEventQuery(ctx->ev1) != hipSuccess) { (void)hipGetLastError(); int rc = eh_sync(ctx); if (rc) return rc; }
  volatile unsigned long long* hs = ctx->h_sum;
  if (!hs || hs[9] != ctx->batch_seq) { ctx->err = "eh_result_occupancy: the batch's totals have not arrived in host memory"; return EH_E_HIP; }
  for (int k = 0; k < 4; k++) out[k] = ctx->last_n ? hs[10 + k] : 0;
  out[4] = (uint64_t)ctx->cus * 4u * EH_WAVES_PER_SIMD;
  return EH_OK;
}
int eh_result_download(eh_ctx* ctx, uint8_t* data, uint64_t cap, uint64_t* off, int32_t* status) {
  if (!ctx) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  int rc = eh_sync(ctx); if (rc) return rc;
  uint64_t n = ctx->last_n;
  std::vector<uint64_t> o(n ? n : 1), len(n ? n : 1);
  if (n) { HIPCHK(ctx, hipMemcpy(o.data(), ctx->d_off, n * 8, hipMemcpyDeviceToHost)); HIPCHK(ctx, hipMemcpy(len.data(), ctx->d_len, n * 8, hipMemcpyDeviceToHost)); }
  if (status && n) HIPCHK(ctx, hipMemcpy(status, ctx->d_status, n * 4, hipMemcpyDeviceToHost));
  uint64_t total = 0; for (uint64_t i = 0; i < n; i++) total += len[i];
  if (off) { uint64_t p = 0; for (uint64_t i = 0; i < n; i++) { off[i] = p; p += len[i]; } off[n] = p; }
  if (data) {
    if (total > cap) { ctx->err = "download buffer too small"; return EH_E_INVALID; }
    if (ctx->ordered) {                                          // already in case order and contiguous: one copy
      if (total) HIPCHK(ctx, hipMemcpy(data, ctx->d_out, total, hipMemcpyDeviceToHost));
      return EH_OK;
    }
    // completion-ordered arena -> case order: gather a chunk of consecutive cases into a bounce buffer on the device,
    // copy it out (full PCIe rate when `data` is pinned or registered host memory) while the next chunk is gathered
    if (total == 0) return EH_OK;
    uint64_t maxlen = 0; for (uint64_t i = 0; i < n; i++) if (len[i] > maxlen) maxlen = len[i];
    uint64_t chunk = ctx->dl_chunk < 64 ? 64 : ctx->dl_chunk;        // eh_options.download_chunk_bytes
    uint64_t want = maxlen > chunk ? maxlen : chunk;
    if (want > total) want = total;
    want = (want + 4095) & ~4095ull;
    if (!ctx->dl_gather) {
      HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->dl_gather, hipStreamNonBlocking));
      HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->dl_copy, hipStreamNonBlocking));
      for (int k = 0; k < 2; k++) { HIPCHK(ctx, hipEventCreate(&ctx->ev_g[k])); HIPCHK(ctx, hipEventCreate(&ctx->ev_c[k])); }
    }
    if (ctx->bounce_cap < want || ctx->bounce_chunk != chunk) {
      for (int k = 0; k < 2; k++) { if (ctx->d_bounce[k]) (void)hipFree(ctx->d_bounce[k]); ctx->d_bounce[k] = nullptr; }
      ctx->bounce_cap = 0;
      for (int k = 0; k < 2; k++) HIPCHK(ctx, hipMalloc(&ctx->d_bounce[k], want));
      ctx->bounce_cap = want; ctx->bounce_chunk = chunk;
    }
    if (!ctx->d_ord || ctx->ord_cap < n + 1) {
      if (ctx->d_ord) (void)hipFree(ctx->d_ord);
      ctx->d_ord = nullptr;
      HIPCHK(ctx, hipMalloc(&ctx->d_ord, (n + 1) * 8));
      ctx->ord_cap = n + 1;
    }
    std::vector<uint64_t> ord(n + 1);
    { uint64_t p = 0; for (uint64_t i = 0; i < n; i++) { ord[i] = p; p += len[i]; } ord[n] = p; }
    HIPCHK(ctx, hipMemcpy(ctx->d_ord, ord.data(), (n + 1) * 8, hipMemcpyHostToDevice));
    uint64_t a = 0; int k = 0;
    while (a < n) {
      uint64_t b = a + 1;
      while (b < n && ord[b + 1] - ord[a] <= ctx->bounce_cap) b++;
      const int buf = k & 1;
      if (k >= 2) HIPCHK(ctx, hipStreamWaitEvent(ctx->dl_gather, ctx->ev_c[buf], 0));     // the copy that last read this buffer
      uint64_t bytes = ord[b] - ord[a];
      if (bytes) {
        uint64_t cnt = b - a;
        uint32_t g = (uint32_t)ctx->cus * 32u; if (g > cnt) g = (uint32_t)cnt;
        hipLaunchKernelGGL(eh_order_gather_kernel, dim3(g), dim3(64), 0, ctx->dl_gather, (const uint8_t*)ctx->d_out, ctx->d_bounce[buf],
                           (const uint64_t*)(ctx->d_off + a), (const uint64_t*)(ctx->d_len + a), (const uint64_t*)(ctx->d_ord + a), cnt, ord[a]);
      }
      HIPCHK(ctx, hipEventRecord(ctx->ev_g[buf], ctx->dl_gather));
      HIPCHK(ctx, hipStreamWaitEvent(ctx->dl_copy, ctx->ev_g[buf], 0));
      if (bytes) HIPCHK(ctx, hipMemcpyAsync(data + ord[a], ctx->d_bounce[buf], bytes, hipMemcpyDeviceToHost, ctx->dl_copy));
      HIPCHK(ctx, hipEventRecord(ctx->ev_c[buf], ctx->dl_copy));
      a = b; k++;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->dl_copy));
    HIPCHK(ctx, hipGetLastError());
  }
  return EH_OK;
}
// erlamsa_out:file_writer/1 (erlamsa_out.erl:103-123): Tokens = re:split(Str, "%n"), Filename = the tokens joined by
// integer_to_list(N) (build_name/3) — every "%n" of the template becomes the case number.
static std::string build_name(const std::string& tmpl, uint64_t n) {
  const std::string num = std::to_string(n);
  std::string out; size_t pos = 0;
  for (;;) {
    size_t e = tmpl.find("%n", pos);
    if (e == std::string::npos) { out.append(tmpl, pos, std::string::npos); break; }
    out.append(tmpl, pos, e - pos); out += num; pos = e + 2;
  }
  return out;
}
int eh_result_write_files(eh_ctx* ctx, const char* name_template, uint64_t first_number, uint32_t threads, uint64_t* files_written, uint64_t* bytes_written, uint64_t* not_written) {
  if (!ctx || !name_template) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  uint64_t in_b = 0, total = 0, n = 0;
  int rc = eh_result_totals(ctx, &in_b, &total, &n);
  if (rc) return rc;
  // one case-ordered download (device gather + overlapped copies), then the files are written by a few host threads
  std::vector<uint64_t> off(n + 1); std::vector<int32_t> st(n ? n : 1);
  uint8_t* host = nullptr;
  bool pinned = total > 0 && hipHostMalloc((void**)&host, total, 0) == hipSuccess;
  if (total > 0 && !pinned) { host = (uint8_t*)malloc(total); if (!host) { ctx->err = "eh_result_write_files: out of host memory"; return EH_E_NOMEM; } }
  rc = eh_result_download(ctx, host, total, off.data(), st.data());
  std::atomic<uint64_t> next{0}, files{0}, bytes{0}, skipped{0}; std::atomic<int> failed{0};
  std::string first_error;
  std::mutex err_lock;
  if (!rc) {
    const std::string tmpl(name_template);
    auto work = [&]() {
      for (;;) {
        uint64_t i = next.fetch_add(1);
        if (i >= n || failed.load()) return;
        // the reference opens the file inside the worker, after the mutation: a case whose worker died or was killed
        // (statuses 1, 5) or that stopped at an engine limit (2, 3, 4) leaves no file
        if (st[i] != EH_CASE_OK) { skipped++; continue; }
        const std::string name = build_name(tmpl, first_number + i);
        FILE* f = fopen(name.c_str(), "wb");
        if (!f) { failed = 1; std::lock_guard<std::mutex> g(err_lock); if (first_error.empty()) first_error = "Error opening file '" + name + "'"; return; }
        const uint64_t len = off[i + 1] - off[i];
        if (len && fwrite(host + off[i], 1, len, f) != len) { failed = 1; std::lock_guard<std::mutex> g(err_lock); if (first_error.empty()) first_error = "short write to '" + name + "'"; }
        fclose(f);
        files++; bytes += len;
      }
    };
    uint32_t nt = threads ? threads : 8; if (nt > 64) nt = 64; if (nt > n) nt = (uint32_t)(n ? n : 1);
    std::vector<std::thread> ts;
    for (uint32_t t = 1; t < nt; t++) ts.emplace_back(work);
    work();
    for (auto& t : ts) t.join();
  }
  if (pinned) (void)hipHostFree(host); else free(host);
  if (files_written) *files_written = files.load();
  if (bytes_written) *bytes_written = bytes.load();
  if (not_written) *not_written = skipped.load();
  if (rc) return rc;
  if (failed.load()) { ctx->err = first_error; return EH_E_INVALID; }
  return EH_OK;
}
int eh_result_fetch(eh_ctx* ctx, uint64_t i, uint8_t* buf, uint64_t cap, uint64_t* out_len) {
  if (!ctx || !out_len) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  int rc = eh_sync(ctx); if (rc) return rc;
  if (i >= ctx->last_n) { ctx->err = "eh_result_fetch: case index outside the last batch"; return EH_E_INVALID; }
  uint64_t ol[2];
  HIPCHK(ctx, hipMemcpy(&ol[0], ctx->d_off + i, 8, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(&ol[1], ctx->d_len + i, 8, hipMemcpyDeviceToHost));
  *out_len = ol[1];
  if (ol[1] > cap || (!buf && ol[1])) { ctx->err = "eh_result_fetch: buffer too small"; return EH_E_INVALID; }
  if (ol[1]) HIPCHK(ctx, hipMemcpy(buf, ctx->d_out + ol[0], ol[1], hipMemcpyDeviceToHost));
  return EH_OK;
}
int eh_result_diag(eh_ctx* ctx, uint64_t* draws, int32_t* last_mutator) {
  if (!ctx) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  int rc = eh_sync(ctx); if (rc) return rc;
  uint64_t n = ctx->last_n;
  if (draws && n) HIPCHK(ctx, hipMemcpy(draws, ctx->d_draws, n * 8, hipMemcpyDeviceToHost));
  if (last_mutator && n) HIPCHK(ctx, hipMemcpy(last_mutator, ctx->d_lastm, n * 4, hipMemcpyDeviceToHost));
  return EH_OK;
}
int eh_result_cycles(eh_ctx* ctx, uint64_t* cycles) {
  if (!ctx || !cycles) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  int rc = eh_sync(ctx); if (rc) return rc;
  if (ctx->last_n) HIPCHK(ctx, hipMemcpy(cycles, ctx->d_cycles, ctx->last_n * 8, hipMemcpyDeviceToHost));
  return EH_OK;
}
int eh_result_meta(eh_ctx* ctx, uint64_t i, uint8_t* buf, uint64_t cap, uint64_t* n_events) {
  if (!ctx || !n_events) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  if (!(ctx->flags & EH_FLAG_META_TRACE)) { ctx->err = "eh_result_meta: configure with EH_FLAG_META_TRACE"; return EH_E_STATE; }
  int rc = eh_sync(ctx); if (rc) return rc;
  if (i >= ctx->last_n) { ctx->err = "eh_result_meta: case index outside the last batch"; return EH_E_INVALID; }
  uint64_t off = 0; uint32_t len = 0;
  HIPCHK(ctx, hipMemcpy(&off, ctx->d_toff + i, 8, hipMemcpyDeviceToHost));
  HIPCHK(ctx, hipMemcpy(&len, ctx->d_tlen + i, 4, hipMemcpyDeviceToHost));
  *n_events = len;                                                        // the trace's length whatever the buffer holds: (buf = NULL, cap = 0) asks for it
  const uint64_t take = buf ? (len < cap ? len : cap) : 0;                // "copied up to cap" (include/erlamsa_hip.h)
  // (with EH_FLAG_ORDERED_OUTPUT the traces stay in the completion-ordered arena, which is d_out2 after the swap)
  const uint8_t* src = (ctx->ordered ? ctx->d_out2 : ctx->d_out) + off;
  if (take) HIPCHK(ctx, hipMemcpy(buf, src, take, hipMemcpyDeviceToHost));
  return EH_OK;
}
int eh_result_peak(eh_ctx* ctx, uint64_t* peak) {
  if (!ctx || !peak) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  int rc = eh_sync(ctx); if (rc) return rc;
  if (ctx->last_n) HIPCHK(ctx, hipMemcpy(peak, ctx->d_peak, ctx->last_n * 8, hipMemcpyDeviceToHost));
  return EH_OK;
}
int eh_result_prof(eh_ctx* ctx, uint64_t* prof /* 256 values */) {
  if (!ctx || !prof) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  int rc = eh_sync(ctx); if (rc) return rc;
  HIPCHK(ctx, hipMemcpy(prof, ctx->d_counters + 8, 256 * 8, hipMemcpyDeviceToHost));
  return EH_OK;
}
// Self test hook: runs `njobs` byte-mover jobs ({kind,dst,src,n,plen} x uint32) on a caller
// supplied buffer image; jobs must touch disjoint destination ranges.  Returns the buffer.
int eh_selftest_movers(eh_ctx* ctx, uint8_t* buf, uint64_t buf_len, const uint32_t* jobs, uint32_t njobs, uint32_t* eq_out) {
  if (!ctx || !buf || !jobs) return EH_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  uint8_t* d = nullptr; uint32_t* dj = nullptr; uint32_t* de = nullptr;
  hipError_t e = hipMalloc(&d, buf_len);
  if (e == hipSuccess) e = hipMalloc(&dj, njobs * 20);
  if (e == hipSuccess) e = hipMalloc(&de, njobs * 4);
  if (e == hipSuccess) e = hipMemcpy(d, buf, buf_len, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(dj, jobs, njobs * 20, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(de, 0, njobs * 4);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(eh_test_copy_kernel, dim3(njobs < 256 ? njobs : 256), dim3(64), 0, 0, d, dj, njobs, de);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(buf, d, buf_len, hipMemcpyDeviceToHost);
  if (e == hipSuccess && eq_out) e = hipMemcpy(eq_out, de, njobs * 4, hipMemcpyDeviceToHost);
  if (d) (void)hipFree(d);
  if (dj) (void)hipFree(dj);
  if (de) (void)hipFree(de);
  HIPCHK(ctx, e);
  return EH_OK;
}
// Self test hook for the device deflate / inflate (eh_zlib.h): op 0 raw deflate, 1 zlib:gzip/1, 2 zlib:deflate(default), 4 zlib:gunzip/1,
// 5 zlib:inflate/2 as mutate_once_compressed/6 uses it.  *ok = 0 where the reference's call raises.
int eh_selftest_zlib(eh_ctx* ctx, int op, const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len, int32_t* ok) {
  if (!ctx || (!in && n) || !out || !out_len || !ok || op < 0 || op > 5 || op == 3 || (op <= 2 && cap < 64)) return EH_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  uint8_t* di = nullptr; uint8_t* dout = nullptr; uint8_t* ds = nullptr; uint64_t* dr = nullptr;
  hipError_t e = hipMalloc(&di, n + 16);
  if (e == hipSuccess) e = hipMalloc(&dout, cap + 16);
  if (e == hipSuccess) e = hipMalloc(&ds, sizeof(ZDef) > sizeof(ZInf) ? sizeof(ZDef) : sizeof(ZInf));
  if (e == hipSuccess) e = hipMalloc(&dr, 16);
  if (e == hipSuccess && n) e = hipMemcpy(di, in, n, hipMemcpyHostToDevice);
  uint64_t r[2] = {0, 0};
  if (e == hipSuccess) {
    hipLaunchKernelGGL(eh_test_zlib_kernel, dim3(1), dim3(64), 0, 0, op, (const uint8_t*)di, n, dout, cap, ds, dr);
    e = hipDeviceSynchronize();
  }
  if (e == hipSuccess) e = hipMemcpy(r, dr, 16, hipMemcpyDeviceToHost);
  if (e == hipSuccess && r[1] && r[0] <= cap && r[0]) e = hipMemcpy(out, dout, r[0], hipMemcpyDeviceToHost);
  if (di) (void)hipFree(di);
  if (dout) (void)hipFree(dout);
  if (ds) (void)hipFree(ds);
  if (dr) (void)hipFree(dr);
  HIPCHK(ctx, e);
  *out_len = r[0]; *ok = (int32_t)r[1];
  return EH_OK;
}
int eh_selftest_sort_by_priority(const uint32_t* pri, uint32_t n, uint32_t* perm) {
  if ((!pri || !perm) && n) return EH_E_INVALID;
  PL in;
  for (uint32_t i = 0; i < n; i++) in.push_back(PItem{pri[i], (int)i});
  PL out = otp_sort_desc_strict(in);
  for (uint32_t i = 0; i < n; i++) perm[i] = (uint32_t)out[i].id;
  return EH_OK;
}

int eh_pool_stats(eh_ctx* ctx, uint64_t* out /* 64 values */) {
  if (!ctx || !out) return EH_E_INVALID;
  if (!ctx->pool) { ctx->err = "no work-area pool yet (eh_reserve or a first batch creates it)"; return EH_E_STATE; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  uint64_t raw[64];
  HIPCHK(ctx, hipMemcpy(raw, ctx->pool->d_ctr, 64 * 8, hipMemcpyDeviceToHost));   // (a plain copy: fine while batches run)
  memcpy(out, raw, 40 * 8);
  out[40] = (uint64_t)ctx->pool->ntiers;
  for (int t = 0; t <= POOL_TIERS; t++) {
    out[41 + t] = t >= 1 && t <= ctx->pool->ntiers ? (uint64_t)ctx->pool->cnt[t] | (raw[40 + t] << 32) : 0;   // areas | the most that were out at once
    out[51 + t] = t <= ctx->pool->ntiers ? ctx->pool->cap[t] : 0;
  }
  out[61] = (uint64_t)ctx->pool->refs; out[62] = ctx->nslots; out[63] = 0;
  return EH_OK;
}
// Cooperative execution (eh_common.h CoBoard): out[0] loops posted, [1] chunks run by wavefronts between cases, [2] by the posting
// cases themselves, [3] cycles the posters waited for the last chunks, [4] loops that found the board full (run by their case alone).
int eh_coop_stats(eh_ctx* ctx, uint64_t* out /* 8 values */) {
  if (!ctx || !out) return EH_E_INVALID;
  if (!ctx->pool) { ctx->err = "no work-area pool yet (eh_reserve or a first batch creates it)"; return EH_E_STATE; }
  HIPCHK(ctx, hipSetDevice(ctx->device));
  HIPCHK(ctx, hipMemcpy(out, (const uint8_t*)ctx->pool->d_board + offsetof(CoBoard, stat), 8 * 8, hipMemcpyDeviceToHost));
  return EH_OK;
}
// Has the last batch of this context finished (results ready, the context free for the next batch)?  Never blocks: what a host
// that keeps several contexts busy polls to give the next batch to WHICHEVER context is free, instead of waiting for the oldest.
int eh_batch_done(eh_ctx* ctx, int* done) {
  if (!ctx || !done) return EH_E_INVALID;
  *done = 1;
  if (!ctx->have_result) return EH_OK;
  hipError_t e = hipEventQuery(ctx->ev1);
  if (e == hipSuccess) return EH_OK;
  if (e == hipErrorNotReady) { *done = 0; (void)hipGetLastError(); return EH_OK; }
  ctx->err = std::string("hipEventQuery: ") + hipGetErrorString(e);
  return EH_E_HIP;
}
int eh_last_kernel_ms(eh_ctx* ctx, float* ms) {
  if (!ctx || !ms) return EH_E_INVALID;
  if (!ctx->have_result) { ctx->err = "no batch has run"; return EH_E_STATE; }
  HIPCHK(ctx, hipEventSynchronize(ctx->ev1));
  HIPCHK(ctx, hipEventElapsedTime(ms, ctx->ev0, ctx->ev1));
  return EH_OK;
}

}  // extern "C"
