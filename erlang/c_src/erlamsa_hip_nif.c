/* erlamsa_hip_nif.c — Erlang NIF shim over the C ABI of liberlamsa_hip.so (include/erlamsa_hip.h).
 *
 * Build on a host with OTP (this repository only compile-checks it against tests/stubs/erl_nif.h):
 *   cc -O2 -fPIC -shared -I$ERL_ROOT/usr/include -I../../include erlamsa_hip_nif.c \
 *      -L../../erlamsa_amd -lerlamsa_hip -o ../priv/erlamsa_hip_nif.so
 *
 *   open(Device)                                             -> {ok, Ctx} | {error, Reason}
 *   fuzz_batch_nif(Ctx, Opts, {A,B,C}, FirstCase, [binary()]) -> {ok, [{Status, binary()}]} | {error, Reason}
 *       one erlamsa_main:fuzzer/1 run, case I of the list = iteration FirstCase+I-1           (eh_fuzz_batch)
 *   fuzz_calls_nif(Ctx, Opts, [{A,B,C}], [binary()])          -> {ok, [{Status, binary()}]} | {error, Reason}
 *       case I is its own fuzzer/1 run with n = 1 and the I-th seed: erlamsa_app:fuzz/2      (eh_fuzz_calls)
 *   Status: enum eh_case_status.  Both run on a dirty I/O scheduler: one call = one GPU batch.
 *   submit_nif / flush_nif / poll_nif: request coalescing, see below.
 *   Several GPUs (ABI 7, include/erlamsa_hip.h "multi-GPU"), see the end of this file:
 *     device_count() / load_corpus_nif / fuzz_range_nif                       a corpus that stays loaded, case ranges of it
 *     comm_init_local_nif([Ctx]) / broadcast_local_nif([Ctx], Root)           one BEAM node, one context per GPU
 *     comm_unique_id_nif() / comm_init_nif(Ctx, Id, Rank, N) / corpus_broadcast_nif(Ctx, Root, Bins | none)   one node per GPU
 *
 * A context remembers the options it was last configured with: eh_configure runs again only when they change.
 */
#include <erl_nif.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "erlamsa_hip.h"

#define OPT_STR 1024
typedef struct {
  char muts[OPT_STR], pats[256], gens[128], host[64];
  int has_muts, has_pats, has_gens, has_host, port;
  double blockscale;
  uint64_t max_case_bytes, big_case_bytes, max_case_work;
  int sequence_muta, meta;
} opt_key;

static ErlNifResourceType* ctx_type;
typedef struct { eh_ctx* ctx; ErlNifMutex* lock; int configured; opt_key key; } ctx_res;
static void ctx_dtor(ErlNifEnv* env, void* obj) {
  (void)env; ctx_res* r = obj;
  if (r->ctx) eh_destroy(r->ctx);
  if (r->lock) enif_mutex_destroy(r->lock);
}

static int load(ErlNifEnv* env, void** priv, ERL_NIF_TERM info) {
  (void)priv; (void)info;
  ctx_type = enif_open_resource_type(env, NULL, "erlamsa_hip_ctx", ctx_dtor, ERL_NIF_RT_CREATE, NULL);
  return ctx_type ? 0 : 1;
}

static ERL_NIF_TERM mk_err_atom(ErlNifEnv* env, const char* a) { return enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_atom(env, a)); }
static ERL_NIF_TERM mk_error(ErlNifEnv* env, eh_ctx* c, int rc) {
  if (rc == EH_E_NOMEM) return mk_err_atom(env, "enomem");
  char msg[512]; msg[0] = 0;
  if (c) (void)eh_last_error_copy(c, msg, sizeof(msg));              /* copied under the engine's lock: pollers run beside submitters */
  return enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_string(env, msg[0] ? msg : eh_strerror(rc), ERL_NIF_LATIN1));
}

/* open(Device) -> {ok, Ctx} */
static ERL_NIF_TERM nif_open(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int dev; (void)argc;
  if (!enif_get_int(env, argv[0], &dev)) return enif_make_badarg(env);
  eh_ctx* c = NULL; int rc = eh_create(dev, &c);
  if (rc) return mk_error(env, NULL, rc);
  ctx_res* r = enif_alloc_resource(ctx_type, sizeof(*r));
  if (!r) { eh_destroy(c); return mk_err_atom(env, "enomem"); }
  memset(r, 0, sizeof(*r));
  r->ctx = c; r->lock = enif_mutex_create((char*)"erlamsa_hip_ctx");
  if (!r->lock) { enif_release_resource(r); return mk_err_atom(env, "enomem"); }
  ERL_NIF_TERM t = enif_make_resource(env, r); enif_release_resource(r);
  return enif_make_tuple2(env, enif_make_atom(env, "ok"), t);
}

/* 1 = present and read, 0 = key absent, -1 = present but not a string that fits */
static int get_str(ErlNifEnv* env, ERL_NIF_TERM map, const char* key, char* buf, unsigned n) {
  ERL_NIF_TERM v;
  if (!enif_get_map_value(env, map, enif_make_atom(env, key), &v)) return 0;
  return enif_get_string(env, v, buf, n, ERL_NIF_LATIN1) > 0 ? 1 : -1;
}
static int get_u64(ErlNifEnv* env, ERL_NIF_TERM map, const char* key, uint64_t* out) {
  ERL_NIF_TERM v; ErlNifUInt64 u;
  if (!enif_get_map_value(env, map, enif_make_atom(env, key), &v)) return 0;
  if (!enif_get_uint64(env, v, &u)) return -1;
  *out = u; return 1;
}

/* Opts: the strings erlamsa_mutations:tostring/1 / erlamsa_patterns:tostring/1 produce, plus the engine limits.
 * An option of the wrong type or an over-long string is a badarg, never a silent fall-back to the default. */
static int read_opts(ErlNifEnv* env, ERL_NIF_TERM map, opt_key* k) {
  ERL_NIF_TERM v; int rc;
  memset(k, 0, sizeof(*k)); k->blockscale = 1.0;
  if ((rc = get_str(env, map, "mutations", k->muts, sizeof(k->muts))) < 0) return 0;
  k->has_muts = rc;
  if ((rc = get_str(env, map, "patterns", k->pats, sizeof(k->pats))) < 0) return 0;
  k->has_pats = rc;
  if ((rc = get_str(env, map, "generators", k->gens, sizeof(k->gens))) < 0) return 0;   /* "direct=500,random=1", also file / jump: Paths = Bins */
  k->has_gens = rc;
  if ((rc = get_str(env, map, "ssrf_host", k->host, sizeof(k->host))) < 0) return 0;
  k->has_host = rc;
  if (enif_get_map_value(env, map, enif_make_atom(env, "ssrf_port"), &v) && !enif_get_int(env, v, &k->port)) return 0;
  if (enif_get_map_value(env, map, enif_make_atom(env, "blockscale"), &v) && !enif_get_double(env, v, &k->blockscale)) return 0;
  if (get_u64(env, map, "max_case_bytes", &k->max_case_bytes) < 0) return 0;
  if (get_u64(env, map, "big_case_bytes", &k->big_case_bytes) < 0) return 0;
  if (get_u64(env, map, "max_case_work", &k->max_case_work) < 0) return 0;
  /* erlamsa_main.erl:223-235: the engine refuses it (EH_E_UNSUPPORTED) and the caller stays on the BEAM path */
  if (enif_get_map_value(env, map, enif_make_atom(env, "sequence_muta"), &v)) k->sequence_muta = enif_compare(v, enif_make_atom(env, "true")) == 0;
  if (enif_get_map_value(env, map, enif_make_atom(env, "meta"), &v)) k->meta = enif_compare(v, enif_make_atom(env, "true")) == 0;   /* -M: keep every case's Meta list */
  return 1;
}

static int configure_if_changed(ctx_res* r, const opt_key* k) {
  if (r->configured && memcmp(&r->key, k, sizeof(*k)) == 0) return EH_OK;
  eh_options o; memset(&o, 0, sizeof(o)); o.abi_version = EH_ABI_VERSION;
  o.mutations = k->has_muts ? k->muts : NULL;        /* NULL = the reference's default table */
  o.patterns = k->has_pats ? k->pats : NULL;
  o.generators = k->has_gens ? k->gens : NULL;       /* NULL = what paths => [direct] leaves: direct=500, random=1 */
  o.ssrf_host = k->has_host ? k->host : NULL;
  o.ssrf_port = k->port; o.blockscale = k->blockscale;
  o.max_case_bytes = k->max_case_bytes; o.big_case_bytes = k->big_case_bytes; o.max_case_work = k->max_case_work;
  o.sequence_muta = (uint32_t)k->sequence_muta;
  o.flags = k->meta ? EH_FLAG_META_TRACE : 0;
  int rc = eh_configure(r->ctx, &o);
  r->configured = rc == EH_OK;
  if (rc == EH_OK) r->key = *k;
  return rc;
}

/* mode 0: argv = Ctx, Opts, Seed, FirstCase, Bins;  mode 1: argv = Ctx, Opts, Seeds, Bins */
static ERL_NIF_TERM run(ErlNifEnv* env, const ERL_NIF_TERM argv[], int mode) {
  ctx_res* r; unsigned n = 0, ns = 0; ErlNifUInt64 first = 1; opt_key k;
  ERL_NIF_TERM bins = argv[mode == 0 ? 4 : 3];
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r) || !enif_is_map(env, argv[1]) || !enif_get_list_length(env, bins, &n))
    return enif_make_badarg(env);
  if (!read_opts(env, argv[1], &k)) return enif_make_badarg(env);
  ErlNifSInt64 seed[3] = {0, 0, 0};
  int64_t* seeds = NULL; uint64_t* off = NULL; uint8_t* data = NULL; uint8_t* out = NULL; int32_t* status = NULL;
  ERL_NIF_TERM ret;
  if (mode == 0) {
    const ERL_NIF_TERM* st; int arity;
    if (!enif_get_tuple(env, argv[2], &arity, &st) || arity != 3 || !enif_get_uint64(env, argv[3], &first) || first < 1) return enif_make_badarg(env);
    for (int i = 0; i < 3; i++) if (!enif_get_int64(env, st[i], &seed[i])) return enif_make_badarg(env);
  } else {
    if (!enif_get_list_length(env, argv[2], &ns) || ns != n) return enif_make_badarg(env);
    seeds = malloc((size_t)(n ? n : 1) * 3 * sizeof(int64_t));
    if (!seeds) return mk_err_atom(env, "enomem");
    ERL_NIF_TERM head; unsigned i = 0;
    for (ERL_NIF_TERM l = argv[2]; enif_get_list_cell(env, l, &head, &l); i++) {
      const ERL_NIF_TERM* st; int arity; ErlNifSInt64 v;
      if (!enif_get_tuple(env, head, &arity, &st) || arity != 3) { free(seeds); return enif_make_badarg(env); }
      for (int j = 0; j < 3; j++) { if (!enif_get_int64(env, st[j], &v)) { free(seeds); return enif_make_badarg(env); } seeds[3 * i + j] = v; }
    }
  }
  /* pack the inputs: binaries are read-only and not retained past the call */
  off = malloc(((size_t)n + 1) * sizeof(uint64_t));
  if (!off) { ret = mk_err_atom(env, "enomem"); goto done; }
  {
    uint64_t total = 0; unsigned i = 0; ERL_NIF_TERM head; ErlNifBinary b;
    for (ERL_NIF_TERM l = bins; enif_get_list_cell(env, l, &head, &l); i++) {
      if (!enif_inspect_binary(env, head, &b)) { ret = enif_make_badarg(env); goto done; }
      off[i] = total; total += b.size;
    }
    off[n] = total;
    data = malloc(total ? total : 1);
    if (!data) { ret = mk_err_atom(env, "enomem"); goto done; }
    i = 0;
    for (ERL_NIF_TERM l = bins; enif_get_list_cell(env, l, &head, &l); i++) { enif_inspect_binary(env, head, &b); memcpy(data + off[i], b.data, b.size); }
  }
  enif_mutex_lock(r->lock);                      /* one batch at a time per context */
  {
    int rc = configure_if_changed(r, &k);
    if (!rc) rc = eh_corpus_upload(r->ctx, data, off, n);
    if (!rc) rc = mode == 0 ? eh_fuzz_batch(r->ctx, (const int64_t*)seed, first, 0, n, NULL) : eh_fuzz_calls(r->ctx, seeds, 0, n, NULL);
    uint64_t in_b = 0, out_b = 0, nc = 0;
    if (!rc) rc = eh_result_totals(r->ctx, &in_b, &out_b, &nc);
    if (!rc) {
      out = malloc(out_b ? out_b : 1); status = malloc(((size_t)n + 1) * sizeof(int32_t));
      if (!out || !status) rc = EH_E_NOMEM;
    }
    if (!rc) rc = eh_result_download(r->ctx, out, out_b, off, status);
    if (rc) { ret = mk_error(env, r->ctx, rc); enif_mutex_unlock(r->lock); goto done; }
  }
  enif_mutex_unlock(r->lock);
  ret = enif_make_list(env, 0);
  for (unsigned i = n; i-- > 0;) {
    ERL_NIF_TERM bin; size_t len = (size_t)(off[i + 1] - off[i]);
    unsigned char* p = enif_make_new_binary(env, len, &bin);
    if (!p) { ret = mk_err_atom(env, "enomem"); goto done; }
    memcpy(p, out + off[i], len);
    ret = enif_make_list_cell(env, enif_make_tuple2(env, enif_make_int(env, status[i]), bin), ret);
  }
  ret = enif_make_tuple2(env, enif_make_atom(env, "ok"), ret);
done:
  free(seeds); free(off); free(data); free(out); free(status);
  return ret;
}

/* Request coalescing (eh_submit / eh_flush / eh_poll): what a micro-batcher in erlamsa_fsupervisor calls.
 *   submit_nif(Ctx, Opts, {A,B,C}, Bin) -> {ok, Ticket}       flush_nif(Ctx) -> ok
 *   poll_nif(Ctx, Ticket) -> {ok, Status, Bin} | again | {error, Reason}   (dirty: waits for the ticket's batch) */
static ERL_NIF_TERM nif_submit(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  (void)argc; ctx_res* r; opt_key k; const ERL_NIF_TERM* st; int arity; ErlNifBinary b; ErlNifSInt64 v; int64_t seed[3];
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r) || !enif_is_map(env, argv[1]) || !read_opts(env, argv[1], &k) ||
      !enif_get_tuple(env, argv[2], &arity, &st) || arity != 3 || !enif_inspect_binary(env, argv[3], &b))
    return enif_make_badarg(env);
  for (int j = 0; j < 3; j++) { if (!enif_get_int64(env, st[j], &v)) return enif_make_badarg(env); seed[j] = v; }
  enif_mutex_lock(r->lock);
  uint64_t ticket = 0;
  int rc = EH_OK;
  /* other options than the previous request's: what is pending is launched with the options it came with; eh_configure then
     collects that batch (its results wait for poll_nif) before the options change */
  if (r->configured && memcmp(&r->key, &k, sizeof(k)) != 0) rc = eh_flush(r->ctx);
  if (!rc) rc = configure_if_changed(r, &k);
  if (!rc) rc = eh_submit(r->ctx, b.data, b.size, seed, &ticket);
  ERL_NIF_TERM ret = rc ? mk_error(env, r->ctx, rc) : enif_make_tuple2(env, enif_make_atom(env, "ok"), enif_make_uint64(env, ticket));
  enif_mutex_unlock(r->lock);
  return ret;
}
static ERL_NIF_TERM nif_flush(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  (void)argc; ctx_res* r;
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r)) return enif_make_badarg(env);
  enif_mutex_lock(r->lock);
  int rc = eh_flush(r->ctx);
  ERL_NIF_TERM ret = rc ? mk_error(env, r->ctx, rc) : enif_make_atom(env, "ok");
  enif_mutex_unlock(r->lock);
  return ret;
}
/* write_files_nif(Ctx, Template :: string(), FirstN) -> {ok, Files, Bytes, NotWritten} | {error, _}
 * erlamsa_out:file_writer/1 for a whole batch: every case of the context's last batch that ended ok goes to the file named
 * by Template with "%n" = case number (erlamsa_out.erl:103-123), written by host threads straight from one download. */
static ERL_NIF_TERM nif_write_files(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  (void)argc; ctx_res* r; char tmpl[1024]; ErlNifUInt64 first;
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r) || enif_get_string(env, argv[1], tmpl, sizeof(tmpl), ERL_NIF_LATIN1) <= 0 ||
      !enif_get_uint64(env, argv[2], &first)) return enif_make_badarg(env);
  uint64_t files = 0, bytes = 0, skipped = 0;
  enif_mutex_lock(r->lock);
  int rc = eh_result_write_files(r->ctx, tmpl, (uint64_t)first, 0, &files, &bytes, &skipped);
  ERL_NIF_TERM ret = rc ? mk_error(env, r->ctx, rc)
                        : enif_make_tuple4(env, enif_make_atom(env, "ok"), enif_make_uint64(env, files), enif_make_uint64(env, bytes), enif_make_uint64(env, skipped));
  enif_mutex_unlock(r->lock);
  return ret;
}
static ERL_NIF_TERM nif_poll(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  (void)argc; ctx_res* r; ErlNifUInt64 ticket;
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r) || !enif_get_uint64(env, argv[1], &ticket)) return enif_make_badarg(env);
  uint64_t len = 0; int32_t status = 0; size_t cap = 1 << 16; ERL_NIF_TERM ret;
  for (;;) {
    uint8_t* buf = malloc(cap);
    if (!buf) return mk_err_atom(env, "enomem");
    /* NOT under r->lock: eh_poll may wait for the whole batch in flight and download it, and submit_nif / flush_nif of other
       processes must go on filling the next batch meanwhile (the engine's coalescer calls are thread safe among themselves) */
    int rc = eh_poll(r->ctx, ticket, buf, cap, &len, &status);
    if (rc && rc != EH_E_AGAIN && !(rc == EH_E_INVALID && len > cap)) { ERL_NIF_TERM e = mk_error(env, r->ctx, rc); free(buf); return e; }
    if (rc == EH_E_AGAIN) { free(buf); return enif_make_atom(env, "again"); }
    if (rc == EH_E_INVALID && len > cap) { free(buf); cap = (size_t)len; continue; }   /* the ticket stays valid */
    ERL_NIF_TERM bin; unsigned char* p = enif_make_new_binary(env, (size_t)len, &bin);
    if (!p) { free(buf); return mk_err_atom(env, "enomem"); }
    memcpy(p, buf, (size_t)len); free(buf);
    ret = enif_make_tuple3(env, enif_make_atom(env, "ok"), enif_make_int(env, status), bin);
    return ret;
  }
}

static ERL_NIF_TERM nif_fuzz_batch(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) { (void)argc; return run(env, argv, 0); }
static ERL_NIF_TERM nif_fuzz_calls(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) { (void)argc; return run(env, argv, 1); }

/* ---- several GPUs (ABI 7) -----------------------------------------------------------------------------------------------------
 * The corpus stays loaded on the context(s); a process per device then runs its contiguous range of the case numbers
 * (erlamsa_hip:fuzz_batch_multi/2 = the shape of erlamsa_main:get_threading_mode/3, erlamsa_main.erl:95-108).  The arena reaches
 * the other GPUs over RCCL called from inside the library (csrc/eh_comm.h): the BEAM needs no HIP or RCCL binding. */
static int pack_bins(ErlNifEnv* env, ERL_NIF_TERM bins, unsigned n, uint8_t** data, uint64_t** off) {   /* 0 ok, 1 badarg, 2 enomem */
  *off = malloc(((size_t)n + 1) * sizeof(uint64_t)); *data = NULL;
  if (!*off) return 2;
  uint64_t total = 0; unsigned i = 0; ERL_NIF_TERM head; ErlNifBinary b;
  for (ERL_NIF_TERM l = bins; enif_get_list_cell(env, l, &head, &l); i++) { if (!enif_inspect_binary(env, head, &b)) return 1; (*off)[i] = total; total += b.size; }
  (*off)[n] = total;
  *data = malloc(total ? total : 1);
  if (!*data) return 2;
  i = 0;
  for (ERL_NIF_TERM l = bins; enif_get_list_cell(env, l, &head, &l); i++) { enif_inspect_binary(env, head, &b); memcpy(*data + (*off)[i], b.data, b.size); }
  return 0;
}
static int get_ctx_list(ErlNifEnv* env, ERL_NIF_TERM list, ctx_res** rs, eh_ctx** cs, unsigned* n) {
  if (!enif_get_list_length(env, list, n) || *n < 1 || *n > 64) return 0;
  ERL_NIF_TERM head; unsigned i = 0;
  for (ERL_NIF_TERM l = list; enif_get_list_cell(env, l, &head, &l); i++) { if (!enif_get_resource(env, head, ctx_type, (void**)&rs[i])) return 0; cs[i] = rs[i]->ctx; }
  return 1;
}
/* every context of a list, in the list's order (the same list from every caller: erlamsa_hip:multi_ctxs/0), so two callers cannot
 * hold one each and wait for the other's */
static void lock_all(ctx_res** rs, unsigned n) { for (unsigned i = 0; i < n; i++) enif_mutex_lock(rs[i]->lock); }
static void unlock_all(ctx_res** rs, unsigned n) { for (unsigned i = n; i-- > 0;) enif_mutex_unlock(rs[i]->lock); }
static ERL_NIF_TERM nif_device_count(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) { (void)argc; (void)argv; return enif_make_int(env, eh_device_count()); }
/* load_corpus_nif(Ctx, Bins) -> ok */
static ERL_NIF_TERM nif_load_corpus(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res* r; unsigned n = 0; uint8_t* data = NULL; uint64_t* off = NULL; (void)argc;
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r) || !enif_get_list_length(env, argv[1], &n)) return enif_make_badarg(env);
  int pr = pack_bins(env, argv[1], n, &data, &off);
  ERL_NIF_TERM ret = enif_make_atom(env, "ok");
  if (pr) ret = pr == 1 ? enif_make_badarg(env) : mk_err_atom(env, "enomem");
  else { enif_mutex_lock(r->lock); int rc = eh_corpus_upload(r->ctx, data, off, n); if (rc) ret = mk_error(env, r->ctx, rc); enif_mutex_unlock(r->lock); }
  free(data); free(off);
  return ret;
}
/* comm_init_local_nif([Ctx]) -> ok ;  broadcast_local_nif([Ctx], Root) -> ok */
static ERL_NIF_TERM nif_comm_init_local(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res* rs[64]; eh_ctx* cs[64]; unsigned n; (void)argc;
  if (!get_ctx_list(env, argv[0], rs, cs, &n)) return enif_make_badarg(env);
  lock_all(rs, n);                                         /* (no batch of another process runs on any of them meanwhile) */
  int rc = eh_comm_init_local(cs, (int)n);
  ERL_NIF_TERM ret = rc ? mk_error(env, cs[0], rc) : enif_make_atom(env, "ok");
  unlock_all(rs, n);
  return ret;
}
static ERL_NIF_TERM nif_broadcast_local(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res* rs[64]; eh_ctx* cs[64]; unsigned n; int root; (void)argc;
  if (!get_ctx_list(env, argv[0], rs, cs, &n) || !enif_get_int(env, argv[1], &root) || root < 0 || root >= (int)n) return enif_make_badarg(env);
  lock_all(rs, n);
  int rc = eh_corpus_broadcast_local(cs, (int)n, root);
  ERL_NIF_TERM ret = rc ? mk_error(env, cs[root], rc) : enif_make_atom(env, "ok");
  unlock_all(rs, n);
  return ret;
}
/* one node per GPU: comm_unique_id_nif() -> {ok, <<128 bytes>>} ; comm_init_nif(Ctx, Id, Rank, N) -> ok ; corpus_broadcast_nif(Ctx, Root, Bins | none) -> ok */
static ERL_NIF_TERM nif_comm_unique_id(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ERL_NIF_TERM bin; (void)argc; (void)argv;
  unsigned char* p = enif_make_new_binary(env, 128, &bin);
  int rc = p ? eh_comm_unique_id(p) : EH_E_NOMEM;
  return rc ? mk_error(env, NULL, rc) : enif_make_tuple2(env, enif_make_atom(env, "ok"), bin);
}
static ERL_NIF_TERM nif_comm_init(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res* r; ErlNifBinary id; int rank, n; (void)argc;
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r) || !enif_inspect_binary(env, argv[1], &id) || id.size != 128 || !enif_get_int(env, argv[2], &rank) || !enif_get_int(env, argv[3], &n)) return enif_make_badarg(env);
  enif_mutex_lock(r->lock);
  int rc = eh_comm_init(r->ctx, id.data, rank, n);
  ERL_NIF_TERM ret = rc ? mk_error(env, r->ctx, rc) : enif_make_atom(env, "ok");
  enif_mutex_unlock(r->lock);
  return ret;
}
static ERL_NIF_TERM nif_corpus_broadcast(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res* r; int root; unsigned n = 0; uint8_t* data = NULL; uint64_t* off = NULL; (void)argc;
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r) || !enif_get_int(env, argv[1], &root)) return enif_make_badarg(env);
  const int have = enif_is_list(env, argv[2]);
  if (have) { if (!enif_get_list_length(env, argv[2], &n)) return enif_make_badarg(env); int pr = pack_bins(env, argv[2], n, &data, &off); if (pr) { free(data); free(off); return pr == 1 ? enif_make_badarg(env) : mk_err_atom(env, "enomem"); } }
  enif_mutex_lock(r->lock);
  int rc = eh_corpus_broadcast(r->ctx, root, data, off, n);
  ERL_NIF_TERM ret = rc ? mk_error(env, r->ctx, rc) : enif_make_atom(env, "ok");
  enif_mutex_unlock(r->lock);
  free(data); free(off);
  return ret;
}
/* fuzz_range_nif(Ctx, Opts, {A,B,C}, FirstCase, CorpusFirst, N) -> {ok, [{Status, binary()}]}: N cases of the LOADED corpus */
static ERL_NIF_TERM nif_fuzz_range(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res* r; opt_key k; ErlNifUInt64 first, cfirst, n; const ERL_NIF_TERM* st; int arity; ErlNifSInt64 seed[3]; (void)argc;
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r) || !enif_is_map(env, argv[1]) || !read_opts(env, argv[1], &k) || !enif_get_tuple(env, argv[2], &arity, &st) || arity != 3 ||
      !enif_get_uint64(env, argv[3], &first) || first < 1 || !enif_get_uint64(env, argv[4], &cfirst) || !enif_get_uint64(env, argv[5], &n)) return enif_make_badarg(env);
  for (int i = 0; i < 3; i++) if (!enif_get_int64(env, st[i], &seed[i])) return enif_make_badarg(env);
  uint64_t* off = malloc(((size_t)n + 1) * sizeof(uint64_t)); int32_t* status = malloc(((size_t)n + 1) * sizeof(int32_t)); uint8_t* out = NULL;
  ERL_NIF_TERM ret;
  if (!off || !status) { ret = mk_err_atom(env, "enomem"); goto done; }
  enif_mutex_lock(r->lock);
  {
    int rc = configure_if_changed(r, &k);
    if (!rc) rc = eh_fuzz_batch(r->ctx, (const int64_t*)seed, first, cfirst, n, NULL);
    uint64_t in_b = 0, out_b = 0, nc = 0;
    if (!rc) rc = eh_result_totals(r->ctx, &in_b, &out_b, &nc);
    if (!rc) { out = malloc(out_b ? out_b : 1); if (!out) rc = EH_E_NOMEM; }
    if (!rc) rc = eh_result_download(r->ctx, out, out_b, off, status);
    if (rc) { ret = mk_error(env, r->ctx, rc); enif_mutex_unlock(r->lock); goto done; }
  }
  enif_mutex_unlock(r->lock);
  ret = enif_make_list(env, 0);
  for (uint64_t i = n; i-- > 0;) {
    ERL_NIF_TERM bin; size_t len = (size_t)(off[i + 1] - off[i]);
    unsigned char* p = enif_make_new_binary(env, len, &bin);
    if (!p) { ret = mk_err_atom(env, "enomem"); goto done; }
    memcpy(p, out + off[i], len);
    ret = enif_make_list_cell(env, enif_make_tuple2(env, enif_make_int(env, status[i]), bin), ret);
  }
  ret = enif_make_tuple2(env, enif_make_atom(env, "ok"), ret);
done:
  free(off); free(status); free(out);
  return ret;
}

/* meta_nif(Ctx, I) -> {ok, <<EventBytes>>}: the meta trace of case I of the context's last batch (eh_result_meta; the options must
 * carry meta => true, i.e. EH_FLAG_META_TRACE).  erlamsa_hip:meta_terms/1 decodes the bytes into the reference's own terms.
 * meta_atoms_nif() -> [atom()]: the atom table the ids in the bytes refer to (eh_meta_atom_name). */
static ERL_NIF_TERM nif_meta(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  ctx_res* r; ErlNifUInt64 i; (void)argc;
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r) || !enif_get_uint64(env, argv[1], &i)) return enif_make_badarg(env);
  ERL_NIF_TERM bin; uint64_t n = 0;
  static uint8_t dummy;
  enif_mutex_lock(r->lock);
  int rc = eh_result_meta(r->ctx, i, &dummy, 0, &n);           /* the length (ABI 8: min(len, cap) bytes are copied, EH_OK either way) */
  unsigned char* p = rc ? NULL : enif_make_new_binary(env, (size_t)n, &bin);
  if (!rc && !p) rc = EH_E_NOMEM;
  if (!rc) rc = eh_result_meta(r->ctx, i, p, n, &n);
  ERL_NIF_TERM ret = rc ? mk_error(env, r->ctx, rc) : enif_make_tuple2(env, enif_make_atom(env, "ok"), bin);
  enif_mutex_unlock(r->lock);
  return ret;
}
static ERL_NIF_TERM nif_meta_atoms(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  (void)argc; (void)argv;
  ERL_NIF_TERM l = enif_make_list(env, 0);
  for (int k = eh_meta_atom_count(); k-- > 0;) l = enif_make_list_cell(env, enif_make_atom(env, eh_meta_atom_name(k)), l);
  return l;
}

static ErlNifFunc funcs[] = {
  {"open", 1, nif_open, 0},
  {"meta_nif", 2, nif_meta, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"meta_atoms_nif", 0, nif_meta_atoms, 0},
  {"fuzz_batch_nif", 5, nif_fuzz_batch, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"fuzz_calls_nif", 4, nif_fuzz_calls, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"submit_nif", 4, nif_submit, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"flush_nif", 1, nif_flush, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"poll_nif", 2, nif_poll, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"write_files_nif", 3, nif_write_files, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"device_count", 0, nif_device_count, 0},
  {"load_corpus_nif", 2, nif_load_corpus, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"fuzz_range_nif", 6, nif_fuzz_range, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"comm_init_local_nif", 1, nif_comm_init_local, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"broadcast_local_nif", 2, nif_broadcast_local, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"comm_unique_id_nif", 0, nif_comm_unique_id, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"comm_init_nif", 4, nif_comm_init, ERL_NIF_DIRTY_JOB_IO_BOUND},
  {"corpus_broadcast_nif", 3, nif_corpus_broadcast, ERL_NIF_DIRTY_JOB_IO_BOUND},
};
ERL_NIF_INIT(erlamsa_hip, funcs, load, NULL, NULL, NULL)
