// eh_zlib.h — deflate / inflate on the device for the container paths of the reference: the `cp` pattern
// (erlamsa_patterns.erl:216-260: zlib:gunzip / zlib:inflate, mutate, zlib:gzip / zlib:deflate(default)), the `ar` pattern and the
// `zip` mutator (zip:foldl / zip:create, erlamsa_patterns.erl:165-214, erlamsa_mutations.erl:1149-1163).
//
// OTP's zlib module is a binding of zlib (the copy in erts/emulator/zlib: 1.2.8 - 1.2.11 for OTP 18 - 23, the releases the
// reference's CI runs).  The compressed bytes the reference writes are therefore zlib's deflate at level 6 (Z_DEFAULT_COMPRESSION),
// windowBits 15, memLevel 8, Z_DEFAULT_STRATEGY, fed with the whole input and Z_FINISH, and parity means reproducing that stream
// BYTE FOR BYTE: the same hash chains (3-byte hash, 32 K heads, chain limit 128 / 32 after a good match), lazy matching
// (max_lazy 16, nice 128, too-far rule for length-3 matches), block boundaries (16 383 symbols), the heap-built Huffman trees with
// zlib's tie breaking, its length-limiting fix-up, its run-length coding of the code lengths and its stored / static / dynamic
// choice.  What follows restates deflate.c's deflate_slow / longest_match / fill_window and trees.c in that sense (function by
// function, named in the comments); tests compare it with the image's libz 1.2.11 on the oracle side (tests/test_gpu_round3.py: test_device_zlib_against_libz,
// tests/hipemu/emu_zlib.py).
//
// Execution: ONE lane.  Match finding with lazy evaluation is a chain of data-dependent decisions and the container paths are rare
// (an input must really be a gzip / zlib / zip file); the wavefront's lane 0 runs these routines as scalar code while the other
// lanes wait at the next wave_sync().  Checksums over whole buffers (CRC-32, Adler-32) are wave-parallel (below and
// wave_crc32 in eh_engine.hip).  All state lives in a scratch block the caller hands in (ZDef: 330 KB, ZInf: 3 KB).
#pragma once
#include "eh_device.h"

namespace eh {

// ------------------------------------------------------------------------------------------------------------------------------
// constants of deflate.h / trees.c
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int Z_WSIZE = 32768, Z_WMASK = Z_WSIZE - 1, Z_HASH_MASK = 32767, Z_MIN_MATCH = 3, Z_MAX_MATCH = 258;
constexpr int Z_MIN_LOOKAHEAD = Z_MAX_MATCH + Z_MIN_MATCH + 1, Z_MAX_DIST = Z_WSIZE - Z_MIN_LOOKAHEAD, Z_TOO_FAR = 4096;
constexpr int Z_LIT_BUFSIZE = 16384;                         // 1 << (memLevel 8 + 6)
constexpr int Z_L_CODES = 286, Z_D_CODES = 30, Z_BL_CODES = 19, Z_HEAP_SIZE = 2 * Z_L_CODES + 1, Z_END_BLOCK = 256;
constexpr int Z_GOOD_MATCH = 8, Z_MAX_LAZY = 16, Z_NICE_MATCH = 128, Z_MAX_CHAIN = 128;        // configuration_table[6]

// CRC-32 (zlib polynomial, reflected) helpers (csum pattern, gzip and zip containers)
struct alignas(16) TabU32x256 { uint32_t v[256]; };
constexpr TabU32x256 crc32_table() {
  TabU32x256 t{};
  for (uint32_t i = 0; i < 256; i++) { uint32_t cc = i; for (int k = 0; k < 8; k++) cc = (cc & 1) ? 0xEDB88320u ^ (cc >> 1) : cc >> 1; t.v[i] = cc; }
  return t;
}
__constant__ TabU32x256 c_crc_table = crc32_table();       // (computed by the compiler, like the tables of eh_device.h)
EH_DEV uint32_t gf2_multmodp(uint32_t a, uint32_t b) {            // a(x)*b(x) mod p(x), reflected representation (x^0 = bit 31)
  uint32_t m = 1u << 31, p = 0;
  for (;;) {
    if (a & m) { p ^= b; if ((a & (m - 1)) == 0) break; }
    m >>= 1;
    b = (b & 1) ? (b >> 1) ^ 0xEDB88320u : b >> 1;
  }
  return p;
}
EH_DEV uint32_t gf2_xpow8n(uint64_t nbytes) {                      // x^(8*nbytes) mod p
  uint32_t r = 1u << 31, sq = 1u << 23;                            // sq = x^8
  while (nbytes) { if (nbytes & 1) r = gf2_multmodp(r, sq); sq = gf2_multmodp(sq, sq); nbytes >>= 1; }
  return r;
}
// erlang:crc32/1 of a contiguous buffer.  Every lane takes one contiguous piece (a multiple of 16 bytes, so that a lane's
// dwordx4 loads stay in step with its neighbours'), walks it four bytes per table step with the slicing tables staged in LDS
// (g_fuse_lds: free outside the fuse mutators - the callers are the csum / container patterns and the zip mutator), and the 64
// piece CRCs are combined with x^(8*len) shifts.  A whole result of some hundred MB behind a csum frame took a byte per table
// step out of constant memory before (profiles/r06_c4_tail.txt).
struct alignas(16) TabU32x1024 { uint32_t v[1024]; };
constexpr TabU32x1024 crc32_slices() {
  TabU32x1024 t{};
  for (uint32_t i = 0; i < 256; i++) { uint32_t cc = i; for (int k = 0; k < 8; k++) cc = (cc & 1) ? 0xEDB88320u ^ (cc >> 1) : cc >> 1; t.v[i] = cc; }
  for (int k = 1; k < 4; k++) for (uint32_t i = 0; i < 256; i++) { uint32_t q = t.v[(k - 1) * 256 + i]; t.v[k * 256 + i] = (q >> 8) ^ t.v[q & 0xFF]; }
  return t;
}
__constant__ TabU32x1024 c_crc_slices = crc32_slices();
EH_DEV uint32_t crc32_word(const uint32_t* T, uint32_t crc, uint32_t w) {
  crc ^= w;
  return T[768 + (crc & 255u)] ^ T[512 + ((crc >> 8) & 255u)] ^ T[256 + ((crc >> 16) & 255u)] ^ T[crc >> 24];
}
EH_DEV uint32_t wave_crc32(cbptr p, uint32_t n) {
  const int l = EH_LANE;
  uint32_t* T = g_fuse_lds;
  wave_sync();
  for (int i = l; i < 1024; i += 64) T[i] = c_crc_slices.v[i];
  wave_sync();
  const uint64_t chunk = ((((uint64_t)n + 63) / 64) + 15) & ~15ull;
  uint64_t a = (uint64_t)l * chunk, b = a + chunk; if (a > n) a = n; if (b > n) b = n;
  uint32_t crc = 0xFFFFFFFFu;
  uint64_t i = a;
  if (i + 16 <= b) {
    uint4 w = ldg16(p + i);
    for (; i + 32 <= b; i += 16) {
      const uint4 nx = ldg16(p + i + 16);                        // the next load is in flight while this one is folded in
      crc = crc32_word(T, crc, w.x); crc = crc32_word(T, crc, w.y); crc = crc32_word(T, crc, w.z); crc = crc32_word(T, crc, w.w);
      w = nx;
    }
    crc = crc32_word(T, crc, w.x); crc = crc32_word(T, crc, w.y); crc = crc32_word(T, crc, w.z); crc = crc32_word(T, crc, w.w);
    i += 16;
  }
  for (; i < b; i++) crc = T[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
  crc ^= 0xFFFFFFFFu;                                              // crc32 of my piece (0 for an empty one)
  // crc32_combine over the 64 pieces at once: piece k contributes crc_k * x^(8 * bytes behind it) (the left-to-right chain
  // ((c0 * x^l1 ^ c1) * x^l2 ^ c2) ... multiplied out; an empty piece has crc 0)
  uint32_t term = 0;
  if (a < n) term = b < n ? gf2_multmodp(gf2_xpow8n((uint64_t)n - b), crc) : crc;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) term ^= (uint32_t)__shfl_xor((int)term, d);
  wave_sync();
  return uni(term);
}
EH_DEV uint32_t fold_xor8(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }
EH_DEV uint32_t wave_xor8(cbptr p, uint32_t n) {
  const uint32_t l = (uint32_t)EH_LANE;
  uint32_t x = 0;
  const uint32_t nv = n >> 4;
  uint32_t i = l;
  for (; i + 192 < nv; i += 256) {
    const uint4 v0 = ldg16(p + 16 * (size_t)i), v1 = ldg16(p + 16 * (size_t)(i + 64));
    const uint4 v2 = ldg16(p + 16 * (size_t)(i + 128)), v3 = ldg16(p + 16 * (size_t)(i + 192));
    x ^= fold_xor8(v0) ^ fold_xor8(v1) ^ fold_xor8(v2) ^ fold_xor8(v3);
  }
  for (; i < nv; i += 64) x ^= fold_xor8(ldg16(p + 16 * (size_t)i));
  for (uint64_t j = ((uint64_t)nv << 4) + l; j < n; j += 64) x ^= p[j];
  x ^= x >> 16; x ^= x >> 8;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) x ^= (uint32_t)__shfl_xor((int)x, d);
  return uni(x) & 255u;
}

__constant__ uint8_t c_z_bl_order[Z_BL_CODES] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};   // bl_order (trees.c) = order (inflate.c)
struct ZTree { uint16_t fc[Z_HEAP_SIZE]; uint16_t dl[Z_HEAP_SIZE]; };     // ct_data: fc = Freq | Code, dl = Dad | Len (the unions of trees.c)
struct ZDef {
  uint32_t head[Z_WSIZE];      // most recent position with this hash; 0 = NIL (position 0 can never be a match, as in zlib)
  uint32_t prev[Z_WSIZE];      // link to the previous position with the same hash, indexed by position & wmask
  uint16_t d_buf[Z_LIT_BUFSIZE]; uint8_t l_buf[Z_LIT_BUFSIZE];
  ZTree lt, dt, bt;            // dyn_ltree, dyn_dtree, bl_tree (dt and bt use the first 61 / 39 entries)
  uint16_t st_lcode[288]; uint8_t st_llen[288];                 // static_ltree (static_dtree: 5 bits, bit-reversed code number)
  int32_t heap[Z_HEAP_SIZE]; uint8_t depth[Z_HEAP_SIZE]; uint16_t bl_count[16]; uint16_t next_code[16];
  int32_t heap_len, heap_max;
  uint32_t last_lit; uint64_t opt_len, static_len;
  // bit writer
  bptr out; uint64_t out_pos, out_cap; uint64_t bi_buf; int32_t bi_valid; int32_t overflow;
};

EH_DEV uint32_t z_bi_reverse(uint32_t code, int len) { uint32_t r = 0; do { r |= code & 1; code >>= 1; r <<= 1; } while (--len > 0); return r >> 1; }
EH_DEV void z_put_byte(EH_G ZDef& s, uint32_t b) { if (s.out_pos < s.out_cap) s.out[s.out_pos] = (uint8_t)b; else s.overflow = 1; s.out_pos++; }
EH_DEV void z_send_bits(EH_G ZDef& s, uint32_t value, int length) {
  s.bi_buf |= (uint64_t)value << s.bi_valid; s.bi_valid += length;
  while (s.bi_valid >= 8) { z_put_byte(s, (uint32_t)(s.bi_buf & 0xff)); s.bi_buf >>= 8; s.bi_valid -= 8; }
}
EH_DEV void z_bi_windup(EH_G ZDef& s) { if (s.bi_valid > 0) z_put_byte(s, (uint32_t)(s.bi_buf & 0xff)); s.bi_buf = 0; s.bi_valid = 0; }

// extra bits / bases of the length and distance codes (extra_lbits, extra_dbits, base_length, base_dist, _length_code, _dist_code of
// trees.c in closed form): lc = match length - 3, d = distance - 1
EH_DEV int z_length_code(uint32_t lc) { if (lc < 8) return (int)lc; if (lc == 255) return 28; int nb = 31 - __builtin_clz(lc); return 4 * (nb - 1) + (int)((lc - (1u << nb)) >> (nb - 2)); }
EH_DEV int z_extra_lbits(int code) { return code < 8 || code == 28 ? 0 : (code - 4) >> 2; }
EH_DEV uint32_t z_base_length(int code) { if (code < 8) return (uint32_t)code; if (code == 28) return 0; int e = (code - 4) >> 2; return ((4u + (uint32_t)(code & 3)) << e); }
EH_DEV int z_dist_code(uint32_t d) { if (d < 4) return (int)d; int nb = 31 - __builtin_clz(d); return 2 * nb + (int)((d >> (nb - 1)) & 1); }
EH_DEV int z_extra_dbits(int code) { return code < 4 ? 0 : (code >> 1) - 1; }
EH_DEV uint32_t z_base_dist(int code) { if (code < 4) return (uint32_t)code; int e = (code >> 1) - 1; return (2u + (uint32_t)(code & 1)) << e; }
EH_DEV int z_extra_blbits(int code) { return code < 16 ? 0 : code == 16 ? 2 : code == 17 ? 3 : 7; }

struct ZDesc { EH_G ZTree* tree; int elems, extra_base, max_length, kind; /* 0 l, 1 d, 2 bl */ int max_code; };
EH_DEV int z_desc_extra(const ZDesc& d, int n) { return d.kind == 0 ? z_extra_lbits(n - 257) : d.kind == 1 ? z_extra_dbits(n) : z_extra_blbits(n); }
EH_DEV int z_desc_static_len(const EH_G ZDef& s, const ZDesc& d, int n) { return d.kind == 0 ? s.st_llen[n] : 5; }   // bl_desc has no static tree

// trees.c: smaller(), pqdownheap()
EH_DEV bool z_smaller(const EH_G ZDef& s, const EH_G ZTree& t, int n, int m) { return t.fc[n] < t.fc[m] || (t.fc[n] == t.fc[m] && s.depth[n] <= s.depth[m]); }
EH_DEV void z_pqdownheap(EH_G ZDef& s, const EH_G ZTree& t, int k) {
  int v = s.heap[k], j = k << 1;
  while (j <= s.heap_len) {
    if (j < s.heap_len && z_smaller(s, t, s.heap[j + 1], s.heap[j])) j++;
    if (z_smaller(s, t, v, s.heap[j])) break;
    s.heap[k] = s.heap[j]; k = j; j <<= 1;
  }
  s.heap[k] = v;
}
// trees.c: gen_bitlen()
EH_DEV void z_gen_bitlen(EH_G ZDef& s, ZDesc& d) {
  EH_G ZTree& t = *d.tree; const int max_code = d.max_code, max_length = d.max_length;
  int h, n, m, bits, overflow = 0;
  for (bits = 0; bits <= 15; bits++) s.bl_count[bits] = 0;
  t.dl[s.heap[s.heap_max]] = 0;
  for (h = s.heap_max + 1; h < Z_HEAP_SIZE; h++) {
    n = s.heap[h];
    bits = t.dl[t.dl[n]] + 1;
    if (bits > max_length) { bits = max_length; overflow++; }
    t.dl[n] = (uint16_t)bits;
    if (n > max_code) continue;
    s.bl_count[bits]++;
    int xbits = n >= d.extra_base ? z_desc_extra(d, n) : 0;
    uint32_t f = t.fc[n];
    s.opt_len += (uint64_t)f * (uint32_t)(bits + xbits);
    if (d.kind != 2) s.static_len += (uint64_t)f * (uint32_t)(z_desc_static_len(s, d, n) + xbits);
  }
  if (overflow == 0) return;
  do {
    bits = max_length - 1;
    while (s.bl_count[bits] == 0) bits--;
    s.bl_count[bits]--; s.bl_count[bits + 1] += 2; s.bl_count[max_length]--;
    overflow -= 2;
  } while (overflow > 0);
  for (bits = max_length; bits != 0; bits--) {
    n = s.bl_count[bits];
    while (n != 0) {
      m = s.heap[--h];
      if (m > max_code) continue;
      if ((uint32_t)t.dl[m] != (uint32_t)bits) { s.opt_len += ((uint64_t)bits - t.dl[m]) * t.fc[m]; t.dl[m] = (uint16_t)bits; }
      n--;
    }
  }
}
// trees.c: gen_codes()
EH_DEV void z_gen_codes(EH_G ZTree& t, int max_code, const uint16_t* bl_count, uint16_t* next_code) {   // (next_code lives in the scratch block: no register arrays)
  uint32_t code = 0;
  for (int bits = 1; bits <= 15; bits++) { code = (code + bl_count[bits - 1]) << 1; next_code[bits] = (uint16_t)code; }
  for (int n = 0; n <= max_code; n++) { int len = t.dl[n]; if (len == 0) continue; t.fc[n] = (uint16_t)z_bi_reverse(next_code[len]++, len); }
}
// trees.c: build_tree()
__device__ __noinline__ void z_build_tree(EH_G ZDef& s, ZDesc& d) {
  EH_G ZTree& t = *d.tree; const int elems = d.elems;
  int n, m, max_code = -1, node;
  s.heap_len = 0; s.heap_max = Z_HEAP_SIZE;
  for (n = 0; n < elems; n++) {
    if (t.fc[n] != 0) { s.heap[++s.heap_len] = max_code = n; s.depth[n] = 0; }
    else t.dl[n] = 0;
  }
  while (s.heap_len < 2) {
    node = s.heap[++s.heap_len] = (max_code < 2 ? ++max_code : 0);
    t.fc[node] = 1; s.depth[node] = 0; s.opt_len--;
    if (d.kind != 2) s.static_len -= (uint64_t)z_desc_static_len(s, d, node);
  }
  d.max_code = max_code;
  for (n = s.heap_len / 2; n >= 1; n--) z_pqdownheap(s, t, n);
  node = elems;
  do {
    n = s.heap[1]; s.heap[1] = s.heap[s.heap_len--]; z_pqdownheap(s, t, 1);          // pqremove
    m = s.heap[1];
    s.heap[--s.heap_max] = n; s.heap[--s.heap_max] = m;
    t.fc[node] = (uint16_t)(t.fc[n] + t.fc[m]);
    s.depth[node] = (uint8_t)((s.depth[n] >= s.depth[m] ? s.depth[n] : s.depth[m]) + 1);
    t.dl[n] = t.dl[m] = (uint16_t)node;
    s.heap[1] = node++;
    z_pqdownheap(s, t, 1);
  } while (s.heap_len >= 2);
  s.heap[--s.heap_max] = s.heap[1];
  z_gen_bitlen(s, d);
  z_gen_codes(t, max_code, s.bl_count, s.next_code);
}
// trees.c: scan_tree() (send == false) and send_tree() (send == true)
__device__ __noinline__ void z_scan_send_tree(EH_G ZDef& s, EH_G ZTree& t, int max_code, bool send) {
  int prevlen = -1, curlen, nextlen = t.dl[0], count = 0, max_count = 7, min_count = 4;
  if (nextlen == 0) { max_count = 138; min_count = 3; }
  if (!send) t.dl[max_code + 1] = 0xffff;                                              // guard
  for (int n = 0; n <= max_code; n++) {
    curlen = nextlen; nextlen = t.dl[n + 1];
    if (++count < max_count && curlen == nextlen) continue;
    else if (count < min_count) {
      if (send) { do { z_send_bits(s, s.bt.fc[curlen], s.bt.dl[curlen]); } while (--count != 0); } else s.bt.fc[curlen] += (uint16_t)count;
    } else if (curlen != 0) {
      if (curlen != prevlen) { if (send) { z_send_bits(s, s.bt.fc[curlen], s.bt.dl[curlen]); count--; } else s.bt.fc[curlen]++; }
      if (send) { z_send_bits(s, s.bt.fc[16], s.bt.dl[16]); z_send_bits(s, (uint32_t)(count - 3), 2); } else s.bt.fc[16]++;
    } else if (count <= 10) {
      if (send) { z_send_bits(s, s.bt.fc[17], s.bt.dl[17]); z_send_bits(s, (uint32_t)(count - 3), 3); } else s.bt.fc[17]++;
    } else {
      if (send) { z_send_bits(s, s.bt.fc[18], s.bt.dl[18]); z_send_bits(s, (uint32_t)(count - 11), 7); } else s.bt.fc[18]++;
    }
    count = 0; prevlen = curlen;
    if (nextlen == 0) { max_count = 138; min_count = 3; }
    else if (curlen == nextlen) { max_count = 6; min_count = 3; }
    else { max_count = 7; min_count = 4; }
  }
}
EH_DEV void z_init_block(EH_G ZDef& s) {
  for (int n = 0; n < Z_L_CODES; n++) s.lt.fc[n] = 0;
  for (int n = 0; n < Z_D_CODES; n++) s.dt.fc[n] = 0;
  for (int n = 0; n < Z_BL_CODES; n++) s.bt.fc[n] = 0;
  s.lt.fc[Z_END_BLOCK] = 1;
  s.opt_len = s.static_len = 0; s.last_lit = 0;
}
// trees.c: _tr_tally(); true = the block is full
EH_DEV bool z_tally(EH_G ZDef& s, uint32_t dist, uint32_t lc) {
  s.d_buf[s.last_lit] = (uint16_t)dist; s.l_buf[s.last_lit++] = (uint8_t)lc;
  if (dist == 0) s.lt.fc[lc]++;
  else { dist--; s.lt.fc[z_length_code(lc) + 257]++; s.dt.fc[z_dist_code(dist)]++; }
  return s.last_lit == Z_LIT_BUFSIZE - 1;
}
// trees.c: compress_block() with the static (stat == true) or the dynamic trees
__device__ __noinline__ void z_compress_block(EH_G ZDef& s, bool stat) {
  for (uint32_t lx = 0; lx < s.last_lit; lx++) {
    uint32_t dist = s.d_buf[lx], lc = s.l_buf[lx];
    if (dist == 0) { if (stat) z_send_bits(s, s.st_lcode[lc], s.st_llen[lc]); else z_send_bits(s, s.lt.fc[lc], s.lt.dl[lc]); continue; }
    int code = z_length_code(lc), sym = code + 257;
    if (stat) z_send_bits(s, s.st_lcode[sym], s.st_llen[sym]); else z_send_bits(s, s.lt.fc[sym], s.lt.dl[sym]);
    int extra = z_extra_lbits(code);
    if (extra) z_send_bits(s, lc - z_base_length(code), extra);
    dist--;
    code = z_dist_code(dist);
    if (stat) z_send_bits(s, z_bi_reverse((uint32_t)code, 5), 5); else z_send_bits(s, s.dt.fc[code], s.dt.dl[code]);
    extra = z_extra_dbits(code);
    if (extra) z_send_bits(s, dist - z_base_dist(code), extra);
  }
  if (stat) z_send_bits(s, s.st_lcode[Z_END_BLOCK], s.st_llen[Z_END_BLOCK]); else z_send_bits(s, s.lt.fc[Z_END_BLOCK], s.lt.dl[Z_END_BLOCK]);
}
// trees.c: _tr_flush_block() (zlib 1.2.11; level 6, Z_DEFAULT_STRATEGY).  buf = nullptr when the block's start has slid out of
// zlib's window (block_start < 0): no stored block then.
__device__ __noinline__ void z_flush_block(EH_G ZDef& s, cbptr buf, uint64_t stored_len, int last) {
  ZDesc ld{&s.lt, Z_L_CODES, 257, 15, 0, 0}, dd{&s.dt, Z_D_CODES, 0, 15, 1, 0}, bd{&s.bt, Z_BL_CODES, 0, 7, 2, 0};
  z_build_tree(s, ld); z_build_tree(s, dd);
  z_scan_send_tree(s, s.lt, ld.max_code, false); z_scan_send_tree(s, s.dt, dd.max_code, false);     // build_bl_tree()
  z_build_tree(s, bd);
  const uint8_t* bl_order = c_z_bl_order;
  int max_blindex;
  for (max_blindex = Z_BL_CODES - 1; max_blindex >= 3; max_blindex--) if (s.bt.dl[bl_order[max_blindex]] != 0) break;
  s.opt_len += 3 * ((uint64_t)max_blindex + 1) + 5 + 5 + 4;
  uint64_t opt_lenb = (s.opt_len + 3 + 7) >> 3, static_lenb = (s.static_len + 3 + 7) >> 3;
  if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
  if (stored_len + 4 <= opt_lenb && buf != nullptr) {                                   // _tr_stored_block()
    z_send_bits(s, (0u << 1) + (uint32_t)last, 3);
    z_bi_windup(s);
    z_put_byte(s, (uint32_t)(stored_len & 0xff)); z_put_byte(s, (uint32_t)((stored_len >> 8) & 0xff));
    z_put_byte(s, (uint32_t)(~stored_len & 0xff)); z_put_byte(s, (uint32_t)((~stored_len >> 8) & 0xff));
    for (uint64_t i = 0; i < stored_len; i++) z_put_byte(s, buf[i]);
  } else if (static_lenb == opt_lenb) {
    z_send_bits(s, (1u << 1) + (uint32_t)last, 3);
    z_compress_block(s, true);
  } else {
    z_send_bits(s, (2u << 1) + (uint32_t)last, 3);
    const int lcodes = ld.max_code + 1, dcodes = dd.max_code + 1, blcodes = max_blindex + 1;      // send_all_trees()
    z_send_bits(s, (uint32_t)(lcodes - 257), 5); z_send_bits(s, (uint32_t)(dcodes - 1), 5); z_send_bits(s, (uint32_t)(blcodes - 4), 4);
    for (int rank = 0; rank < blcodes; rank++) z_send_bits(s, s.bt.dl[bl_order[rank]], 3);
    z_scan_send_tree(s, s.lt, lcodes - 1, true); z_scan_send_tree(s, s.dt, dcodes - 1, true);
    z_compress_block(s, false);
  }
  z_init_block(s);
  if (last) z_bi_windup(s);
}

// deflate.c: deflate_slow() + longest_match() + the bookkeeping of fill_window() over the whole input with Z_FINISH, in absolute
// positions: zlib's 64 KB window slides by 32 K whenever strstart reaches wsize + MAX_DIST with less than MIN_LOOKAHEAD ahead;
// `base` is the absolute position of its first byte.  Sliding changes nothing a match search can see (entries it clears are
// farther back than MAX_DIST) except that the window's own first byte cannot start a match - position 0 here, NIL in zlib.
// Raw deflate stream (no wrapper) of src[0..n) to s.out; lane 0 only.
__device__ __noinline__ void z_deflate_raw(EH_G ZDef& s, cbptr src, uint64_t n) {
  for (int i = 0; i < Z_WSIZE; i++) s.head[i] = 0;
  {                                                                                      // tr_static_init(): static_ltree
    uint16_t* blc = s.bl_count; uint16_t* next_code = s.next_code;
    for (int i = 0; i < 16; i++) blc[i] = 0;
    int k = 0;
    while (k <= 143) { s.st_llen[k++] = 8; blc[8]++; }
    while (k <= 255) { s.st_llen[k++] = 9; blc[9]++; }
    while (k <= 279) { s.st_llen[k++] = 7; blc[7]++; }
    while (k <= 287) { s.st_llen[k++] = 8; blc[8]++; }
    uint32_t code = 0;
    for (int bits = 1; bits <= 15; bits++) { code = (code + blc[bits - 1]) << 1; next_code[bits] = (uint16_t)code; }
    for (int m = 0; m <= 287; m++) { int len = s.st_llen[m]; s.st_lcode[m] = (uint16_t)z_bi_reverse(next_code[len]++, len); }
  }
  s.bi_buf = 0; s.bi_valid = 0;
  z_init_block(s);
  uint64_t strstart = 0, block_start = 0, base = 0, filled = 0;
  uint64_t match_start = 0, prev_match = 0;
  uint32_t match_length = Z_MIN_MATCH - 1, prev_length = Z_MIN_MATCH - 1;
  bool match_available = false;
  for (;;) {
    uint64_t lookahead = filled - strstart;
    if (lookahead < (uint64_t)Z_MIN_LOOKAHEAD) {                                         // fill_window(): everything that is left fits or the window slides
      while (filled < n) {
        if (strstart - base >= (uint64_t)(Z_WSIZE + Z_MAX_DIST)) base += Z_WSIZE;
        uint64_t more = 2 * (uint64_t)Z_WSIZE - (filled - base);
        if (more == 0) break;
        uint64_t take = n - filled < more ? n - filled : more;
        filled += take;
        if (filled - strstart >= (uint64_t)Z_MIN_LOOKAHEAD) break;
      }
      lookahead = filled - strstart;
      if (lookahead == 0) break;
    }
    uint32_t hash_head = 0;
    if (lookahead >= Z_MIN_MATCH) {                                                      // INSERT_STRING
      uint32_t h = (((uint32_t)src[strstart] << 10) ^ ((uint32_t)src[strstart + 1] << 5) ^ src[strstart + 2]) & Z_HASH_MASK;
      hash_head = s.prev[strstart & Z_WMASK] = s.head[h];
      s.head[h] = (uint32_t)strstart;
    }
    prev_length = match_length; prev_match = match_start; match_length = Z_MIN_MATCH - 1;
    if (hash_head != 0 && prev_length < (uint32_t)Z_MAX_LAZY && strstart - hash_head <= (uint64_t)Z_MAX_DIST && hash_head > base) {
      // longest_match()
      uint32_t chain_length = Z_MAX_CHAIN; uint32_t cur_match = hash_head;
      int best_len = (int)prev_length; int nice_match = Z_NICE_MATCH;
      const uint64_t limit = strstart - base > (uint64_t)Z_MAX_DIST ? strstart - Z_MAX_DIST : base;
      if (prev_length >= (uint32_t)Z_GOOD_MATCH) chain_length >>= 2;
      if ((uint64_t)nice_match > lookahead) nice_match = (int)lookahead;
      cbptr scan = src + strstart;
      // zlib compares up to MAX_MATCH bytes inside its window whatever the lookahead is (bytes beyond the input are whatever the
      // window holds) and clips the result to the lookahead afterwards; a candidate either differs inside the lookahead or reaches
      // nice_match <= lookahead, so comparing inside the input only gives the same choice
      const int maxcmp = lookahead < (uint64_t)Z_MAX_MATCH ? (int)lookahead : Z_MAX_MATCH;
      do {
        cbptr match = src + cur_match;
        int len = 0;
        if (best_len < maxcmp ? (match[best_len] == scan[best_len] && match[best_len - 1] == scan[best_len - 1] && match[0] == scan[0] && match[1] == scan[1])
                              : false) {
          len = 2;
          while (len < maxcmp && match[len] == scan[len]) len++;
        }
        if (len > best_len) {
          match_start = cur_match; best_len = len;
          if (len >= nice_match) break;
        }
      } while ((cur_match = s.prev[cur_match & Z_WMASK]) > limit && --chain_length != 0);
      match_length = (uint64_t)best_len <= lookahead ? (uint32_t)best_len : (uint32_t)lookahead;
      if (match_length <= 5 && (match_length == Z_MIN_MATCH && strstart - match_start > (uint64_t)Z_TOO_FAR)) match_length = Z_MIN_MATCH - 1;
    }
    if (prev_length >= Z_MIN_MATCH && match_length <= prev_length) {
      const uint64_t max_insert = strstart + lookahead - Z_MIN_MATCH;
      bool bflush = z_tally(s, (uint32_t)(strstart - 1 - prev_match), prev_length - Z_MIN_MATCH);
      prev_length -= 2;
      do {
        if (++strstart <= max_insert) {
          uint32_t h = (((uint32_t)src[strstart] << 10) ^ ((uint32_t)src[strstart + 1] << 5) ^ src[strstart + 2]) & Z_HASH_MASK;
          s.prev[strstart & Z_WMASK] = s.head[h]; s.head[h] = (uint32_t)strstart;
        }
      } while (--prev_length != 0);
      match_available = false; match_length = Z_MIN_MATCH - 1; strstart++;
      if (bflush) { z_flush_block(s, block_start >= base ? src + block_start : nullptr, strstart - block_start, 0); block_start = strstart; }
    } else if (match_available) {
      bool bflush = z_tally(s, 0, src[strstart - 1]);
      if (bflush) { z_flush_block(s, block_start >= base ? src + block_start : nullptr, strstart - block_start, 0); block_start = strstart; }
      strstart++;
    } else { match_available = true; strstart++; }
  }
  if (match_available) { (void)z_tally(s, 0, src[strstart - 1]); match_available = false; }
  z_flush_block(s, block_start >= base ? src + block_start : nullptr, strstart - block_start, 1);
}

// upper bound of the raw stream: deflateBound() for the default parameters, without the wrapper
EH_DEV uint64_t z_deflate_bound(uint64_t n) { return n + (n >> 12) + (n >> 14) + (n >> 25) + 13 + 64; }

// Adler-32 of a contiguous buffer, wave-parallel: a = 1 + sum(x_i), b = n + sum((n - i) * x_i)  (mod 65521)
EH_DEV uint32_t wave_adler32(cbptr p, uint64_t n) {
  // a = 1 + sum x_i, b = n + sum (n - i) x_i (mod 65521); a lane's 16 bytes at i0: (n - i0) * s - sum j x_j
  const uint32_t l = (uint32_t)EH_LANE;
  uint64_t a = 0, b = 0;
  const uint64_t nv = n >> 4;
  uint32_t r = (uint32_t)((n - 16 * (uint64_t)(l < nv ? l : 0)) % 65521u);          // (n - i0) mod 65521, stepped down by 1024 a round
  for (uint64_t i = l; i < nv; i += 64) {
    const uint4 v = ldg16(p + 16 * (size_t)i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t s = 0, t = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t b0 = w[q] & 255u, b1 = (w[q] >> 8) & 255u, b2 = (w[q] >> 16) & 255u, b3 = w[q] >> 24;
      s += b0 + b1 + b2 + b3;
      t += (uint32_t)(4 * q) * (b0 + b1 + b2 + b3) + b1 + 2 * b2 + 3 * b3;
    }
    a += s; b += (uint64_t)r * s + 65521u - t;                     // t <= 120 * 255 < 65521; 2^64 is out of reach for n < 2^32
    r = r >= 1024u ? r - 1024u : r + 65521u - 1024u;
  }
  for (uint64_t j = (nv << 4) + l; j < n; j += 64) { const uint64_t x = p[j]; a += x; b += ((n - j) % 65521u) * x; }
  a %= 65521u; b %= 65521u;
  uint32_t a32 = wave_sum((uint32_t)a), b32 = wave_sum((uint32_t)b);                    // 64 * 65520 fits
  a32 = (a32 + 1) % 65521u; b32 = (uint32_t)((b32 + n % 65521u) % 65521u);
  return (b32 << 16) | a32;
}

// ------------------------------------------------------------------------------------------------------------------------------
// inflate: inflate.c's observable behaviour for a whole input handed over at once - what is written before the input runs out
// (zlib:inflate/2 returns that much without an error), which inputs are data errors, where the stream ends.  Codes are decoded
// canonically, bit by bit (count / symbol arrays); a code is accepted when all its bits are there, like a table entry whose
// length does not exceed the bits held.
// ------------------------------------------------------------------------------------------------------------------------------
enum ZStatus : int { ZS_END = 0, ZS_TRUNC = 1, ZS_ERROR = 2 };
struct ZInf {
  cbptr in; uint64_t n, pos; uint64_t hold; int32_t bits;
  bptr out; uint64_t outn;                       // out == nullptr: count only
  uint16_t lcount[16], lsym[288], dcount[16], dsym[32], ccount[16], csym[19];
  uint16_t lens[19 + 288 + 32];                     // code length code lengths, then literal/length + distance lengths
  int32_t lmax, dmax, cmax;
  int32_t st, early;                                // results handed from lane 0 to the wavefront
};
EH_DEV bool zi_need(EH_G ZInf& z, int k) {                                                  // NEEDBITS: false = out of input
  while (z.bits < k) { if (z.pos >= z.n) return false; z.hold |= (uint64_t)z.in[z.pos++] << z.bits; z.bits += 8; }
  return true;
}
EH_DEV uint32_t zi_bits(EH_G ZInf& z, int k) { uint32_t v = (uint32_t)(z.hold & ((1ull << k) - 1)); z.hold >>= k; z.bits -= k; return v; }
// inflate_table(): 0 ok, -1 over-subscribed or incomplete set.  lens[0..n) -> count / symbol, *maxlen
EH_DEV int zi_table(const uint16_t* lens, int n, bool codes_type, uint16_t* count, uint16_t* sym, int32_t* maxlen) {
  uint16_t offs[16];
  for (int i = 0; i < 16; i++) count[i] = 0;
  for (int i = 0; i < n; i++) count[lens[i]]++;
  int max = 15; while (max >= 1 && count[max] == 0) max--;
  *maxlen = max;
  if (max == 0) return 0;                                                                // no symbols: decoding anything is an invalid code
  int left = 1;
  for (int len = 1; len <= 15; len++) { left <<= 1; left -= count[len]; if (left < 0) return -1; }
  if (left > 0 && (codes_type || max != 1)) return -1;
  offs[1] = 0; for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + count[len]);
  for (int i = 0; i < n; i++) if (lens[i] != 0) sym[offs[lens[i]]++] = (uint16_t)i;
  return 0;
}
// >= 0 symbol, -1 out of input, -2 invalid code
EH_DEV int zi_decode(EH_G ZInf& z, const uint16_t* count, const uint16_t* sym, int maxlen) {
  int code = 0, first = 0, index = 0;
  const int lim = maxlen == 0 ? 1 : maxlen;
  for (int len = 1; len <= lim; len++) {
    if (!zi_need(z, len)) return -1;
    code |= (int)((z.hold >> (len - 1)) & 1);
    int cnt = maxlen == 0 ? 0 : count[len];
    if (code - cnt < first) { (void)zi_bits(z, len); return sym[index + (code - first)]; }
    index += cnt; first += cnt; first <<= 1; code <<= 1;
  }
  return -2;
}
// the deflate blocks of a stream; z.in/pos at the first block header.  ZS_END: final block done (bits left in hold are dropped by
// the caller's BYTEBITS)
__device__ __noinline__ int zi_blocks(EH_G ZInf& z) {
  for (;;) {
    if (!zi_need(z, 3)) return ZS_TRUNC;
    int last = (int)zi_bits(z, 1), type = (int)zi_bits(z, 2);
    if (type == 3) return ZS_ERROR;                                                      // invalid block type
    if (type == 0) {
      (void)zi_bits(z, z.bits & 7);
      if (!zi_need(z, 32)) return ZS_TRUNC;
      uint32_t v = zi_bits(z, 32);
      if ((v & 0xffff) != ((v >> 16) ^ 0xffff)) return ZS_ERROR;                         // invalid stored block lengths
      uint32_t len = v & 0xffff;
      while (len) {                                                                      // (hold is empty here: bits == 0)
        if (z.pos >= z.n) return ZS_TRUNC;
        uint64_t c = z.n - z.pos < len ? z.n - z.pos : len;
        if (z.out) for (uint64_t i = 0; i < c; i++) z.out[z.outn + i] = z.in[z.pos + i];
        z.outn += c; z.pos += c; len -= (uint32_t)c;
      }
    } else {
      if (type == 1) {                                                                   // fixedtables()
        for (int i = 0; i < 144; i++) z.lens[i] = 8;
        for (int i = 144; i < 256; i++) z.lens[i] = 9;
        for (int i = 256; i < 280; i++) z.lens[i] = 7;
        for (int i = 280; i < 288; i++) z.lens[i] = 8;
        (void)zi_table(z.lens, 288, false, z.lcount, z.lsym, &z.lmax);
        for (int i = 0; i < 32; i++) z.lens[i] = 5;
        (void)zi_table(z.lens, 32, false, z.dcount, z.dsym, &z.dmax);
      } else {
        if (!zi_need(z, 14)) return ZS_TRUNC;
        int nlen = (int)zi_bits(z, 5) + 257, ndist = (int)zi_bits(z, 5) + 1, ncode = (int)zi_bits(z, 4) + 4;
        if (nlen > 286 || ndist > 30) return ZS_ERROR;                                   // too many length or distance symbols
        const uint8_t* order = c_z_bl_order;
        for (int i = 0; i < 19; i++) z.lens[i] = 0;
        for (int i = 0; i < ncode; i++) { if (!zi_need(z, 3)) return ZS_TRUNC; z.lens[order[i]] = (uint16_t)zi_bits(z, 3); }
        if (zi_table(z.lens, 19, true, z.ccount, z.csym, &z.cmax) != 0) return ZS_ERROR;  // invalid code lengths set
        int have = 0;
        while (have < nlen + ndist) {
          // (inflate.c looks the code up and needs code + extra bits together before it consumes anything; a truncated stream
          // stops either way and nothing has been written for this block yet)
          int sy = zi_decode(z, z.ccount, z.csym, z.cmax);
          if (sy == -1) return ZS_TRUNC;
          if (sy == -2) return ZS_ERROR;
          if (sy < 16) { z.lens[19 + have++] = (uint16_t)sy; continue; }
          int len = 0, copy;
          if (sy == 16) { if (!zi_need(z, 2)) return ZS_TRUNC; if (have == 0) return ZS_ERROR; len = z.lens[19 + have - 1]; copy = 3 + (int)zi_bits(z, 2); }
          else if (sy == 17) { if (!zi_need(z, 3)) return ZS_TRUNC; copy = 3 + (int)zi_bits(z, 3); }
          else { if (!zi_need(z, 7)) return ZS_TRUNC; copy = 11 + (int)zi_bits(z, 7); }
          if (have + copy > nlen + ndist) return ZS_ERROR;                                // invalid bit length repeat
          while (copy--) z.lens[19 + have++] = (uint16_t)len;
        }
        if (z.lens[19 + 256] == 0) return ZS_ERROR;                                       // missing end-of-block
        if (zi_table(z.lens + 19, nlen, false, z.lcount, z.lsym, &z.lmax) != 0) return ZS_ERROR;
        if (zi_table(z.lens + 19 + nlen, ndist, false, z.dcount, z.dsym, &z.dmax) != 0) return ZS_ERROR;
      }
      for (;;) {
        int sy = zi_decode(z, z.lcount, z.lsym, z.lmax);
        if (sy == -1) return ZS_TRUNC;
        if (sy == -2 || sy > 285) return ZS_ERROR;                                        // invalid literal/length code
        if (sy < 256) { if (z.out) z.out[z.outn] = (uint8_t)sy; z.outn++; continue; }
        if (sy == 256) break;
        int lcode = sy - 257;
        int eb = z_extra_lbits(lcode);
        if (!zi_need(z, eb)) return ZS_TRUNC;
        uint32_t len = (lcode == 28 ? 258u : 3u + z_base_length(lcode)) + (eb ? zi_bits(z, eb) : 0u);
        int dc = zi_decode(z, z.dcount, z.dsym, z.dmax);
        if (dc == -1) return ZS_TRUNC;
        if (dc == -2 || dc > 29) return ZS_ERROR;                                         // invalid distance code
        int db = z_extra_dbits(dc);
        if (!zi_need(z, db)) return ZS_TRUNC;
        uint64_t dist = 1u + z_base_dist(dc) + (db ? zi_bits(z, db) : 0u);
        if (dist > z.outn) return ZS_ERROR;                                               // invalid distance too far back
        if (z.out) for (uint32_t i = 0; i < len; i++) z.out[z.outn + i] = z.out[z.outn + i - dist];
        z.outn += len;
      }
    }
    if (last) return ZS_END;
  }
}
// (a length / distance pair whose bits run out half way has consumed its first part here while inflate.c would keep it for the
//  next call; the stream stops in both, with the same bytes written)

// zlib:gunzip/1 of OTP 20.1 - 23 (the OTP TARGET of this engine and of oracle/): inflateInit(Z, 16 + MAX_WBITS,
// reset), inflate(Z, Data), inflateEnd(Z).  With the `reset` end-of-stream behaviour the NIF calls inflateReset when a member has
// ended and input is left, so CONCATENATED MEMBERS ARE ALL DECODED (their data concatenated); bytes that are not another complete
// member are a data_error (a bad header at once, an unfinished one at inflateEnd, which raises data_error unless the last call
// saw the end of a stream).  So: success only for one or more complete members, each with correct CRC-32 and ISIZE, and nothing
// else; everything else is error:data_error for the caller (mutate_once_compressed then tries the zlib path, which fails on the
// gzip magic, and the block ends as {compressed, failed}).  Until round 4 this was the rule of OTP 18 - 20.0 (first member only,
// trailing bytes ignored).  Parses one member header at p, returns the offset of its deflate data or 0.
EH_DEV uint64_t zi_gzip_header(cbptr p, uint64_t n, const uint32_t* crc_table) {
  if (n < 10 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xe0)) return 0;
  uint32_t flags = p[3]; uint64_t pos = 10;
  if (flags & 0x04) { if (pos + 2 > n) return 0; uint64_t xl = p[pos] | ((uint64_t)p[pos + 1] << 8); pos += 2; if (pos + xl > n) return 0; pos += xl; }
  if (flags & 0x08) { while (pos < n && p[pos] != 0) pos++; if (pos >= n) return 0; pos++; }
  if (flags & 0x10) { while (pos < n && p[pos] != 0) pos++; if (pos >= n) return 0; pos++; }
  if (flags & 0x02) {
    if (pos + 2 > n) return 0;
    uint32_t crc = 0xFFFFFFFFu;
    for (uint64_t i = 0; i < pos; i++) crc = crc_table[(crc ^ p[i]) & 0xFF] ^ (crc >> 8);
    crc ^= 0xFFFFFFFFu;
    if ((crc & 0xffff) != (p[pos] | ((uint32_t)p[pos + 1] << 8))) return 0;              // header crc mismatch
    pos += 2;
  }
  return pos;
}


// ------------------------------------------------------------------------------------------------------------------------------
// wave-level entry points (every lane calls them; lane 0 runs the codecs, checksums are wave-parallel)
// ------------------------------------------------------------------------------------------------------------------------------
enum ZFormat : int { ZF_RAW = 0, ZF_GZIP = 1, ZF_ZLIB = 2 };
EH_DEV uint64_t z_wrap_bytes(int fmt) { return fmt == ZF_GZIP ? 18 : fmt == ZF_ZLIB ? 6 : 0; }
// zlib:gzip/1 (ZF_GZIP: deflateInit2(.., 16 + 15, 8, default): 1f 8b 08 00, mtime 0, xfl 0, OS 3 = Unix, CRC-32 and ISIZE little
// endian), zlib:deflateInit(Z, default) + deflate(Z, Data, finish) (ZF_ZLIB: 78 9c, Adler-32 big endian), or the bare stream of
// deflateInit(.., -15, ..) that zip:create uses (ZF_RAW).  dst needs z_deflate_bound(n) + z_wrap_bytes(fmt) bytes; returns the
// length written.
EH_DEV uint64_t z_compress(EH_G ZDef* sc, int fmt, cbptr src, uint64_t n, bptr dst, uint64_t cap) {
  uint32_t chk = fmt == ZF_GZIP ? wave_crc32(src, (uint32_t)n) : fmt == ZF_ZLIB ? wave_adler32(src, n) : 0u;
  if (EH_LANE == 0) {
    uint64_t h = 0;
    if (fmt == ZF_GZIP) { const uint8_t hd[10] = {0x1f, 0x8b, 8, 0, 0, 0, 0, 0, 0, 3}; for (int i = 0; i < 10; i++) dst[i] = hd[i]; h = 10; }
    else if (fmt == ZF_ZLIB) { dst[0] = 0x78; dst[1] = 0x9c; h = 2; }
    sc->out = dst + h; sc->out_pos = 0; sc->out_cap = cap - z_wrap_bytes(fmt); sc->overflow = 0;
    z_deflate_raw(*sc, src, n);
    bptr t = dst + h + sc->out_pos;
    if (!sc->overflow) {
      if (fmt == ZF_GZIP) { for (int i = 0; i < 4; i++) { t[i] = (uint8_t)(chk >> (8 * i)); t[4 + i] = (uint8_t)((uint32_t)n >> (8 * i)); } }
      else if (fmt == ZF_ZLIB) { for (int i = 0; i < 4; i++) t[i] = (uint8_t)(chk >> (24 - 8 * i)); }
    }
    sc->out_pos += z_wrap_bytes(fmt);
  }
  wave_sync();
  uint64_t total = uni64(sc->out_pos); uint32_t ovf = uni((uint32_t)sc->overflow);
  return ovf ? 0 : total;
}
// One decoding pass over in[off..n): out == nullptr counts.  Returns the ZStatus; zi->outn bytes (would be) written; *end = offset of
// the first byte behind the deflate data (after BYTEBITS) when the status is ZS_END.
EH_DEV int z_inflate_pass(EH_G ZInf* zi, cbptr in, uint64_t n, uint64_t off, bptr out, uint64_t* end) {
  if (EH_LANE == 0) {
    zi->in = in; zi->n = n; zi->pos = off; zi->hold = 0; zi->bits = 0; zi->out = out; zi->outn = 0;
    int st = zi_blocks(*zi);
    zi->st = st;
    zi->pos -= (uint64_t)(zi->bits >> 3);                                                // whole bytes still in hold belong to what follows
  }
  wave_sync();
  *end = uni64(zi->pos);
  return (int)uni((uint32_t)zi->st);
}
// What zlib:gunzip(Bin) (ZF_GZIP) or zlib:inflateInit(Z), zlib:inflate(Z, Bin) without inflateEnd (ZF_ZLIB) make of `in` in
// mutate_once_compressed/6 (erlamsa_patterns.erl:216-246).  Phase 1: z_uncompress_size -> 1 and *outn when there is a result
// (gzip: only a complete, so far well-formed member; zlib: also what a stream that just stops decodes to - inflate/2 returns it and
// nobody calls inflateEnd), 0 when the call raises (data_error, need_dictionary).  Phase 2: z_uncompress_write into a buffer of *outn
// bytes; checks the trailer (CRC-32 + ISIZE / Adler-32) and returns 0 when it is wrong.
EH_DEV int z_uncompress_size(EH_G ZInf* zi, int fmt, cbptr in, uint64_t n, uint64_t* outn, uint64_t* data_off) {
  *data_off = 0; *outn = 0;
  if (fmt == ZF_GZIP) {                                                                  // every member in turn (inflateReset): count
    uint64_t pos = 0, total = 0;
    do {
      if (EH_LANE == 0) zi->pos = zi_gzip_header(in + pos, n - pos, c_crc_table.v);
      wave_sync();
      const uint64_t off = uni64(zi->pos);
      if (off == 0) return 0;                                                            // not a member (or its header stops): data_error
      uint64_t end;
      const int st = z_inflate_pass(zi, in + pos, n - pos, off, nullptr, &end);
      if (st != ZS_END || end + 8 > n - pos) return 0;                                   // data error / an unfinished member: inflateEnd raises
      total += uni64(zi->outn);
      if (total > 0xFFFFFF00ull) { *outn = total; return 1; }                            // (the caller gives up on the size)
      pos += end + 8;
    } while (pos < n);
    *outn = total;
    return 1;
  }
  uint64_t off = 0; int early = -1;                                                      // early: 1 = empty result, 0 = raises
  if (EH_LANE == 0) {
    if (n < 2) early = 1;                                                                // HEAD needs 16 bits: nothing decoded, no error
    else {
      uint32_t b0 = in[0], b1 = in[1];
      if (((b0 << 8) + b1) % 31 != 0 || (b0 & 0x0f) != 8 || (b0 >> 4) + 8 > 15) early = 0;   // incorrect header check / unknown compression method / invalid window size
      else if (b1 & 0x20) early = n >= 6 ? 0 : 1;                                        // FDICT: {need_dictionary, Adler} once the id is there
      off = 2;
    }
    zi->early = early; zi->pos = off;
  }
  wave_sync();
  early = (int)uni((uint32_t)zi->early); off = uni64(zi->pos);
  *data_off = off;
  if (early >= 0) return early;
  uint64_t end;
  int st = z_inflate_pass(zi, in, n, off, nullptr, &end);
  *outn = uni64(zi->outn);
  if (st == ZS_ERROR) return 0;
  return 1;                                                                              // ZS_TRUNC: inflate/2 returns what it decoded, nobody calls inflateEnd
}
EH_DEV int z_uncompress_write(EH_G ZInf* zi, int fmt, cbptr in, uint64_t n, uint64_t data_off, bptr out, uint64_t outn) {
  uint64_t end;
  if (fmt == ZF_GZIP) {                                                                  // the members again, written one behind the other
    uint64_t pos = 0, total = 0;
    do {
      if (EH_LANE == 0) zi->pos = zi_gzip_header(in + pos, n - pos, c_crc_table.v);
      wave_sync();
      const uint64_t off = uni64(zi->pos);
      if (off == 0) return 0;
      const int st = z_inflate_pass(zi, in + pos, n - pos, off, out + total, &end);
      const uint64_t mlen = uni64(zi->outn);
      if (st != ZS_END || end + 8 > n - pos || total + mlen > outn) return 0;            // CHECK / LENGTH states starve: inflateEnd -> data_error
      const uint32_t crc = wave_crc32(out + total, (uint32_t)mlen);
      cbptr t = in + pos + end;
      uint32_t c0 = 0, l0 = 0; for (int i = 0; i < 4; i++) { c0 |= (uint32_t)uni(t[i]) << (8 * i); l0 |= (uint32_t)uni(t[4 + i]) << (8 * i); }
      if (c0 != crc || l0 != (uint32_t)mlen) return 0;                                   // incorrect data check / incorrect length check
      total += mlen; pos += end + 8;
    } while (pos < n);
    return total == outn ? 1 : 0;
  }
  if (outn == 0 && (n < 2 || (in[1] & 0x20))) return 1;
  int st = z_inflate_pass(zi, in, n, data_off, out, &end);
  if (st != ZS_END) return st == ZS_TRUNC ? 1 : 0;
  if (end + 4 > n) return 1;                                                             // the check value never arrives: no error, everything was written
  uint32_t ad = wave_adler32(out, outn);
  uint32_t a0 = 0; for (int i = 0; i < 4; i++) a0 = (a0 << 8) | (uint32_t)uni(in[end + i]);
  return a0 == ad ? 1 : 0;
}

}  // namespace eh
