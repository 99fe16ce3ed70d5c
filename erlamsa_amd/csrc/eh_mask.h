// eh_mask.h — byte-class bit masks held in registers, for event-driven sequential automata.
//
// A lone wavefront retires roughly one dependent instruction every ~5 cycles, so an automaton that
// touches every byte (lexer, delimiter matcher) costs hundreds of cycles per byte.  Instead a
// 4 KiB window of the block is classified in parallel (one coalesced byte load per lane, one ballot
// per class) into 64-bit words — lane w keeps word w of every class in registers — and the
// automaton then JUMPS from event to event with readlane + ctz, never re-reading memory.
#pragma once
#include "eh_text.h"

namespace eh {

constexpr uint32_t MW_WORDS = 64;                 // words per window (one per lane)
constexpr uint32_t MW_STEP = 63 * 64;             // bytes consumed per window; word 63 is lookahead

EH_DEV uint64_t readlane64(uint64_t v, uint32_t lane_idx) {
  uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, (int)lane_idx);
  uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), (int)lane_idx);
  return ((uint64_t)hi << 32) | lo;
}

template <int K>
struct MaskWin {
  cbptr p; uint32_t L;
  uint32_t base;                                   // absolute offset of bit 0 of word 0 (multiple of 64)
  bool valid;
  uint64_t m[K];                                   // my word (lane = word index) of each class
  uint64_t inrange;                                // bits of my word that are < L
};

// cls(byte) -> bit k set if the byte belongs to class k.
// One coalesced byte load per lane and word, 16 loads in flight; one ballot per class and word.
template <int K, class Cls>
EH_DEV void mw_load(MaskWin<K>& w, uint32_t base, Cls cls) {
  const int l = EH_LANE;
  w.base = base; w.valid = true;
#pragma unroll
  for (int k = 0; k < K; k++) w.m[k] = 0;
  w.inrange = 0;
  for (uint32_t j0 = 0; j0 < MW_WORDS; j0 += 16) {
    if (base + 64u * j0 >= w.L) break;                           // the rest of the window is past the block
    uint32_t by[16];
#pragma unroll
    for (uint32_t u = 0; u < 16; u++) { uint32_t idx = base + 64u * (j0 + u) + (uint32_t)l; by[u] = idx < w.L ? (uint32_t)w.p[idx] : 256u; }
#pragma unroll
    for (uint32_t u = 0; u < 16; u++) {
      bool in = by[u] < 256u;
      uint32_t bits = in ? cls(by[u]) : 0u;
      unsigned long long inm = __ballot(in);
      if ((uint32_t)l == j0 + u) w.inrange = inm;
#pragma unroll
      for (int k = 0; k < K; k++) { unsigned long long mk = __ballot(((bits >> k) & 1u) != 0); if ((uint32_t)l == j0 + u) w.m[k] = mk; }
    }
  }
}
// reference implementation (one byte per lane, one ballot per class) used by the kernel self test
template <int K, class Cls>
EH_DEV void mw_load_ref(MaskWin<K>& w, uint32_t base, Cls cls) {
  const int l = EH_LANE;
  w.base = base; w.valid = true;
  for (int k = 0; k < K; k++) w.m[k] = 0;
  w.inrange = 0;
  for (uint32_t j = 0; j < MW_WORDS; j++) {
    uint32_t idx = base + 64u * j + (uint32_t)l;
    uint32_t bits = 0; bool in = idx < w.L;
    if (in) bits = cls((uint32_t)w.p[idx]);
    unsigned long long inm = __ballot(in);
    if ((uint32_t)l == j) w.inrange = inm;
    for (int k = 0; k < K; k++) { unsigned long long mk = __ballot(((bits >> k) & 1u) != 0); if ((uint32_t)l == j) w.m[k] = mk; }
  }
}
// smallest window-relative position >= from (from <= 4096) whose bit is set in `word` (the caller's
// per-lane word, any combination of classes); 4096 if none
EH_DEV uint32_t mw_next(uint64_t word, uint32_t from) {
  uint32_t wi = from >> 6;
  if (wi >= MW_WORDS) return 4096;
  uint64_t cur = readlane64(word, wi) >> (from & 63);
  if (cur) return from + (uint32_t)__builtin_ctzll(cur);
  unsigned long long nz = __ballot(word != 0);
  nz = wi >= 63 ? 0ull : (nz & ~((2ull << wi) - 1));
  if (!nz) return 4096;
  uint32_t j = (uint32_t)__builtin_ctzll(nz);
  return 64u * j + (uint32_t)__builtin_ctzll(readlane64(word, j));
}
EH_DEV bool mw_test(uint64_t word, uint32_t rel) { return (readlane64(word, rel >> 6) >> (rel & 63)) & 1ull; }

}  // namespace eh
