// eh_zip.h — zip archives on the device: what zip:foldl/3 (prim_zip) hands the reference's funs and what zip:create/3 with
// [memory] writes back, for the `ar` pattern (erlamsa_patterns.erl:165-214) and the `zip` mutator
// (erlamsa_mutations.erl:1149-1163).  OTP's stdlib is not part of the reference tree; the semantics restated here (and, the same
// way, in oracle/oracle.cpp, namespace otpzip, where they are spelled out) are those of prim_zip.erl / zip.erl of OTP 18 - 23:
//   read:   end-of-central-directory record searched in the last 22, 44, 88, .. bytes; central directory walked entry by entry; a
//           file's bytes from its LOCAL header (method 0 / 8, compressed size), no CRC check, a deflate stream that just stops
//           yields what it decoded, a corrupt one kills the worker;
//   write:  local headers + data + central directory + end record; stored below 10 bytes and for .Z .zip .zoo .arc .lzh .arj,
//           raw deflate (eh_zlib.h, byte for byte zlib's) otherwise; the UNCOMPRESSED size written - and the number of bytes
//           taken from the new binary - is the size the entry had in the archive that was read.
// Archives with ZIP64 markers, encrypted or data-descriptor entries, directory entries or names that are empty / not ASCII end
// the case as EH_CASE_UNSUPPORTED: their outcome depends on corners of prim_zip this restatement does not pin.
// Header parsing is scalar work of lane 0; inflate / deflate / CRC-32 / copies are the routines of eh_zlib.h and eh_device.h.
#pragma once
#include "eh_zlib.h"

namespace eh {

enum ZipRc : int { ZR_OK = 0, ZR_ERROR = 1 /* {error, _} */, ZR_CRASH = 2 /* an error exception: the worker dies */, ZR_UNSUP = 3, ZR_STOP = -1 /* out of work memory: status set */ };
struct ZipEntry {
  uint64_t name; uint32_t name_len; uint32_t up;       // `up` x "../" goes in front of the name (mutate_zip_path/4)
  uint64_t data; uint32_t data_len; uint32_t usize;    // the file's bytes; file_info.size = the central directory's uncompressed size
  uint16_t time, date; uint32_t method;                // DOS time / date as read; method / crc / csz / lpos: filled by zip_create
  uint32_t crc, csz, lpos, pad;
};
struct ZipRd {
  cbptr a; uint64_t n; uint32_t entries, idx; uint64_t pos;
  int32_t rc; uint32_t method; uint64_t comp; uint32_t comp_len; uint32_t pad;   // lane 0 -> wavefront
  ZipEntry cur;
};
EH_DEV uint32_t zle16(cbptr p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
EH_DEV uint32_t zle32(cbptr p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
EH_DEV void zput16(bptr p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); }
EH_DEV void zput32(bptr p, uint32_t v) { zput16(p, v & 0xffff); zput16(p + 2, v >> 16); }
template <bool GROW> EH_DEV bptr zip_alloc(Ctx& c, uint64_t n) { return GROW ? ws_alloc_grow(c, n) : ws_alloc(c, n); }

// prim_zip:get_central_dir up to the first entry: get_end_of_central_dir (Sz, Sz + Sz, .. <= 16#ffff), eocd_and_comment_from_bin
EH_DEV int zip_open(EH_G ZipRd* r, cbptr a, uint64_t n) {
  // almost every block is not an archive: the wavefront looks for a signature first (eh_lex.h; a superset of the windows below)
  if (n > 0xFFFFFFF0ull || !has_zip_eocd(a, (uint32_t)n)) return ZR_ERROR;
  if (EH_LANE == 0) {
    int rc = ZR_ERROR;
    r->a = a; r->n = n; r->entries = 0; r->idx = 0; r->pos = 0;
    if (n >= 22) {
      uint64_t at = ~0ull;
      for (uint64_t sz = 22; sz <= 0xffff && at == ~0ull; sz += sz) {
        uint64_t start = n > sz ? n - sz : 0;
        for (uint64_t i = start; i + 22 <= n; i++) if (a[i] == 0x50 && a[i + 1] == 0x4b && a[i + 2] == 0x05 && a[i + 3] == 0x06) { at = i; break; }
        if (start == 0) break;
      }
      if (at != ~0ull) {
        cbptr e = a + at + 4;
        uint32_t entries = zle16(e + 6), off = zle32(e + 12), clen = zle16(e + 16);
        if (at + 22 + clen == n) {
          if (entries == 0xffff || off == 0xffffffffu) rc = ZR_UNSUP;
          else { r->entries = entries; r->pos = off; rc = ZR_OK; }
        }
      }
    }
    r->rc = rc;
  }
  wave_sync();
  return (int)uni((uint32_t)r->rc);
}
// get_cd_loop's next entry: the central-directory header and its name (what exists before the fun is called: a broken
// header fails the fold without the fun having run for this entry)
EH_DEV int zip_next_header(EH_G ZipRd* r) {
  if (EH_LANE == 0) {
    cbptr a = r->a; const uint64_t n = r->n;
    int rc = ZR_OK;
    if (r->pos + 46 > n || zle32(a + r->pos) != 0x02014b50u) rc = ZR_ERROR;                        // bad_central_directory
    else {
      cbptr h = a + r->pos;
      uint32_t gp = zle16(h + 8), fnl = zle16(h + 28), exl = zle16(h + 30), cml = zle16(h + 32), lho = zle32(h + 42);
      if (r->pos + 46 + fnl + exl + cml > n) rc = ZR_ERROR;
      else {
        EH_G ZipEntry& e = r->cur;
        e.name = (uint64_t)(h + 46); e.name_len = fnl; e.up = 0; e.time = (uint16_t)zle16(h + 12); e.date = (uint16_t)zle16(h + 14); e.usize = zle32(h + 24);
        e.data = 0; e.data_len = 0; e.method = 0; e.crc = 0; e.csz = 0; e.lpos = lho; e.pad = 0;
        r->pos += 46 + fnl + exl + cml; r->idx++;
        bool ascii = true; for (uint32_t i = 0; i < fnl; i++) if (h[46 + i] > 127) ascii = false;
        if (fnl == 0 || h[46 + fnl - 1] == '/' || (gp & 9) || zle32(h + 20) == 0xffffffffu || e.usize == 0xffffffffu || lho == 0xffffffffu || !ascii || (uint64_t)lho + 30 > n) rc = ZR_UNSUP;
      }
    }
    r->rc = rc;
  }
  wave_sync();
  return (int)uni((uint32_t)r->rc);
}
// ... and the entry's GetBin() (get_z_file / get_z_all), called from inside the fun.  The entry's bytes are the archive's own for
// a stored file and a fresh work-area block for a deflated one.
template <bool GROW> EH_DEV int zip_next_file(Ctx& c, EH_G ZipRd* r, EH_G ZipEntry* out) {
  if (EH_LANE == 0) {
    cbptr a = r->a; const uint64_t n = r->n;
    int rc = ZR_OK;
    cbptr l = a + r->cur.lpos;
    if (zle32(l) != 0x04034b50u) rc = ZR_ERROR;                                                   // bad_local_file_header
    else {
      uint32_t lgp = zle16(l + 6), method = zle16(l + 8), csz = zle32(l + 18), lfn = zle16(l + 26), lex = zle16(l + 28);
      uint64_t ds = (uint64_t)r->cur.lpos + 30 + lfn + lex;
      if ((lgp & 9) || ds > n) rc = ZR_UNSUP;
      else if (method != 0 && method != 8) rc = ZR_ERROR;                                         // throw({bad_file_header, _})
      else { uint64_t de = ds + csz > n ? n : ds + csz; r->method = method; r->comp = (uint64_t)(a + ds); r->comp_len = (uint32_t)(de - ds); }
    }
    r->cur.lpos = 0;
    r->rc = rc;
  }
  wave_sync();
  int rc = (int)uni((uint32_t)r->rc);
  if (rc != ZR_OK) return rc;
  const uint32_t method = uni(r->method); cbptr comp = (cbptr)uni64(r->comp); const uint32_t clen = uni(r->comp_len);
  uint64_t dptr = (uint64_t)comp, dlen = clen;
  if (method == 8) {                                                                              // inflateInit(Z, -MAX_WBITS), inflate, (catch) inflateEnd
    EH_G ZInf* zi = (EH_G ZInf*)zip_alloc<GROW>(c, sizeof(ZInf));
    if (!zi) return ZR_STOP;
    uint64_t end;
    int st = z_inflate_pass(zi, comp, clen, 0, nullptr, &end);
    if (st == ZS_ERROR) return ZR_CRASH;
    dlen = uni64(zi->outn);
    if (dlen > 0xFFFFFF00ull) { EH_SET_OVERFLOW(c, 320); return ZR_STOP; }
    bptr d = zip_alloc<GROW>(c, dlen + 16);
    if (!d) return ZR_STOP;
    (void)z_inflate_pass(zi, comp, clen, 0, d, &end);
    dptr = (uint64_t)d;
  }
  if (EH_LANE == 0) { *out = r->cur; out->data = dptr; out->data_len = (uint32_t)dlen; }
  wave_sync();
  return ZR_OK;
}
template <bool GROW> EH_DEV int zip_next(Ctx& c, EH_G ZipRd* r, EH_G ZipEntry* out) {
  int rc = zip_next_header(r);
  return rc != ZR_OK ? rc : zip_next_file<GROW>(c, r, out);
}
// zip:create(Name, Files, [memory]) over es[0..n) in order; *out / *len: the archive.
template <bool GROW> EH_DEV int zip_create(Ctx& c, EH_G ZipEntry* es, uint32_t n, bptr* out, uint64_t* len) {
  const int l = EH_LANE;
  // sizes first: what each entry takes from its binary, its method, whether the stream can be finished at all
  uint64_t bound = 22; uint32_t bad = 0;
  for (uint32_t i = 0; i < n; i++) {                                                              // (every lane computes the same)
    EH_G ZipEntry& e = es[i];
    const uint64_t U = e.usize, S = e.data_len, nl = (uint64_t)e.name_len + 3ull * e.up;
    cbptr nm = (cbptr)e.name;
    int dot = -1;                                                                                // filename:extension/1
    for (uint32_t k = 0; k < e.name_len; k++) { if (nm[k] == '.') dot = (int)k; else if (nm[k] == '/') dot = -1; }
    bool st_ext = false;
    if (dot >= 0) {
      const uint32_t el = e.name_len - (uint32_t)dot; cbptr x = nm + dot + 1;
      if (el == 2) st_ext = x[0] == 'Z';
      else if (el == 4) st_ext = (x[0] == 'z' && x[1] == 'i' && x[2] == 'p') || (x[0] == 'z' && x[1] == 'o' && x[2] == 'o') || (x[0] == 'a' && x[1] == 'r' && x[2] == 'c') ||
                                 (x[0] == 'l' && x[1] == 'z' && x[2] == 'h') || (x[0] == 'a' && x[1] == 'r' && x[2] == 'j');
    }
    e.method = (U < 10 || st_ext) ? 0u : 8u;
    uint64_t take = S < U ? S : U;
    if (nl > 0xffff) bad |= 2;
    if (U != 0) {
      if (e.method == 0) { if (S == 0) bad |= 1; }                                                // {read, U} -> eof: Output({write, eof}) exits
      else if (S != 0 && S <= 8192ull * ((U + 8191) / 8192 - 1)) bad |= 1;                        // `finish` never passed: deflateEnd -> data_error
    } else take = 0;
    if (e.method == 8 && S == 0) take = 0;
    e.csz = (uint32_t)take;                                                                       // (bytes to take; the compressed size replaces it below)
    bound += 30 + 46 + 2 * nl + (e.method == 8 ? z_deflate_bound(take) : take);
  }
  wave_sync();
  bad = uni(bad); bound = uni64(bound);
  if (bad & 2) return ZR_UNSUP;
  if (bad & 1) return ZR_ERROR;
  if (bound > 0xFFFFFF00ull) { EH_SET_OVERFLOW(c, 321); return ZR_STOP; }
  EH_G ZDef* zd = (EH_G ZDef*)zip_alloc<GROW>(c, sizeof(ZDef));
  if (!zd) return ZR_STOP;
  bptr dst = zip_alloc<GROW>(c, bound + 16);
  if (!dst) return ZR_STOP;
  uint64_t pos = 0;
  for (uint32_t i = 0; i < n; i++) {
    EH_G ZipEntry& e = es[i];
    const uint32_t take = uni(e.csz), method = uni(e.method), nl = uni(e.name_len), up = uni(e.up), U = uni(e.usize);
    cbptr data = (cbptr)uni64(e.data); cbptr nm = (cbptr)uni64(e.name);
    const uint32_t fnl = nl + 3 * up;
    bptr h = dst + pos; bptr body = h + 30 + fnl;
    uint32_t crc = 0, csz = 0;
    if (take > 0) {
      crc = wave_crc32(data, take);
      if (method == 8) { csz = (uint32_t)z_compress(zd, ZF_RAW, data, take, body, z_deflate_bound(take)); if (csz == 0) { EH_SET_OVERFLOW(c, 322); return ZR_STOP; } }
      else { wave_copy(body, data, take); csz = take; }
    }
    for (uint32_t k = l; k < fnl; k += 64) h[30 + k] = k < 3 * up ? (k % 3 == 2 ? (uint8_t)'/' : (uint8_t)'.') : nm[k - 3 * up];
    if (l == 0) {                                                                                 // local_file_header_from_info_method_name + the {pwrite, Pos0 + 14, <<CRC:32, CompSize:32>>}
      zput32(h, 0x04034b50u); zput16(h + 4, 20); zput16(h + 6, 0); zput16(h + 8, method); zput16(h + 10, e.time); zput16(h + 12, e.date);
      zput32(h + 14, crc); zput32(h + 18, csz); zput32(h + 22, U); zput16(h + 26, fnl); zput16(h + 28, 0);
      e.crc = crc; e.csz = csz; e.lpos = (uint32_t)pos;
    }
    wave_sync();
    pos += 30 + fnl + csz;
  }
  const uint64_t cd = pos;
  for (uint32_t i = 0; i < n; i++) {                                                              // put_central_dir: cd_file_header_from_lh_and_pos
    const EH_G ZipEntry& e = es[i];
    const uint32_t nl = uni(e.name_len), up = uni(e.up), fnl = nl + 3 * up;
    cbptr nm = (cbptr)uni64(e.name);
    bptr h = dst + pos;
    for (uint32_t k = l; k < fnl; k += 64) h[46 + k] = k < 3 * up ? (k % 3 == 2 ? (uint8_t)'/' : (uint8_t)'.') : nm[k - 3 * up];
    if (l == 0) {
      zput32(h, 0x02014b50u); zput16(h + 4, 20); zput16(h + 6, 20); zput16(h + 8, 0); zput16(h + 10, e.method); zput16(h + 12, e.time); zput16(h + 14, e.date);
      zput32(h + 16, e.crc); zput32(h + 20, e.csz); zput32(h + 24, e.usize); zput16(h + 28, fnl); zput16(h + 30, 0); zput16(h + 32, 0);
      zput16(h + 34, 0); zput16(h + 36, 0); zput32(h + 38, 0); zput32(h + 42, e.lpos);
    }
    pos += 46 + fnl;
  }
  if (l == 0) {                                                                                   // put_eocd
    bptr h = dst + pos;
    zput32(h, 0x06054b50u); zput16(h + 4, 0); zput16(h + 6, 0); zput16(h + 8, n); zput16(h + 10, n); zput32(h + 12, (uint32_t)(pos - cd)); zput32(h + 16, (uint32_t)cd); zput16(h + 20, 0);
  }
  pos += 22;
  wave_sync();
  *out = dst; *len = pos;
  return ZR_OK;
}

// zip_path_traversal/2 (erlamsa_mutations.erl:1149-1163): zip:foldl(fun mutate_zip_path/4, ..) draws rand(20) for every entry while
// the central directory is walked, then {ok, {_, Bin}} = zip:create(..) (a failing create is a badmatch: the worker dies)
__device__ __noinline__ int muta_zip(Ctx&) {
  EH_CTX;
  const Blk hb = blk_load(c.bl, c.cur);
  c.r_kind = R_SAME;
  if (!has_zip_eocd((cbptr)hb.ptr, hb.len)) return -1;                                   // no end record anywhere: {error, bad_eocd} (the common case, nothing allocated)
  EH_G ZipRd* rd = (EH_G ZipRd*)ws_alloc(c, sizeof(ZipRd));
  if (!rd) return 0;
  int rc = zip_open(rd, (cbptr)hb.ptr, hb.len);
  if (rc == ZR_UNSUP) { c.status = CASE_UNSUPPORTED; return 0; }
  if (rc != ZR_OK) return -1;
  const uint32_t n = uni(rd->entries);
  EH_G ZipEntry* es = (EH_G ZipEntry*)ws_alloc(c, (uint64_t)(n ? n : 1) * sizeof(ZipEntry));
  if (!es) return 0;
  for (uint32_t i = 0; i < n; i++) {
    rc = zip_next_header(rd);                                                                     // (a broken directory entry ends the fold before the fun runs)
    if (rc == ZR_UNSUP) { c.status = CASE_UNSUPPORTED; return 0; }
    if (rc != ZR_OK) return -1;
    uint32_t up = rng_rand(c.rng, 20);                                                            // mutate_zip_path/4: R first, then B()
    rc = zip_next_file<false>(c, rd, &es[i]);
    if (rc == ZR_STOP) return 0;
    if (rc == ZR_UNSUP) { c.status = CASE_UNSUPPORTED; return 0; }
    if (rc == ZR_CRASH) { c.status = CASE_CRASHED; return 0; }
    if (rc != ZR_OK) return -1;
    if (EH_LANE == 0) es[i].up = up;
    wave_sync();
  }
  bptr out; uint64_t len;
  rc = zip_create<false>(c, es, n, &out, &len);
  if (rc == ZR_STOP) return 0;
  if (rc == ZR_UNSUP) { c.status = CASE_UNSUPPORTED; return 0; }
  if (rc != ZR_OK) { c.status = CASE_CRASHED; return 0; }
  c.r_kind = R_NEW; c.r_ptr = out; c.r_len = (uint32_t)len;
  return +1;
}

}  // namespace eh
