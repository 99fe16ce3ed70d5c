/* erlamsa_hip_nif.c — Erlang NIF shim over the C ABI of liberlamsa_hip.so (include/erlamsa_hip.h).
 *
 * NOT compiled in this repository's image (no erl_nif.h here); build on a host with OTP:
 *   cc -O2 -fPIC -shared -I$ERL_ROOT/usr/include -I../../include erlamsa_hip_nif.c \
 *      -L../../erlamsa_amd -lerlamsa_hip -o ../priv/erlamsa_hip_nif.so
 *
 * Exposes erlamsa_hip:fuzz_batch_nif(Opts :: map(), Seed :: {A,B,C}, FirstCase, [binary()])
 *   -> {ok, [{Status :: 0..5, binary()}]} | {error, Reason}   (status: enum eh_case_status)
 * It runs on a dirty I/O scheduler: one call = one GPU batch.
 */
#include <erl_nif.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "erlamsa_hip.h"

static ErlNifResourceType* ctx_type;
typedef struct { eh_ctx* ctx; } ctx_res;
static void ctx_dtor(ErlNifEnv* env, void* obj) { (void)env; ctx_res* r = obj; if (r->ctx) eh_destroy(r->ctx); }

static int load(ErlNifEnv* env, void** priv, ERL_NIF_TERM info) {
  (void)priv; (void)info;
  ctx_type = enif_open_resource_type(env, NULL, "erlamsa_hip_ctx", ctx_dtor, ERL_NIF_RT_CREATE, NULL);
  return ctx_type ? 0 : 1;
}

static ERL_NIF_TERM mk_error(ErlNifEnv* env, eh_ctx* c, int rc) {
  const char* msg = c ? eh_last_error(c) : eh_strerror(rc);
  return enif_make_tuple2(env, enif_make_atom(env, "error"), enif_make_string(env, msg && *msg ? msg : eh_strerror(rc), ERL_NIF_LATIN1));
}

/* open(Device) -> {ok, Ctx} */
static ERL_NIF_TERM nif_open(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  int dev; (void)argc;
  if (!enif_get_int(env, argv[0], &dev)) return enif_make_badarg(env);
  eh_ctx* c = NULL; int rc = eh_create(dev, &c);
  if (rc) return mk_error(env, NULL, rc);
  ctx_res* r = enif_alloc_resource(ctx_type, sizeof(*r)); r->ctx = c;
  ERL_NIF_TERM t = enif_make_resource(env, r); enif_release_resource(r);
  return enif_make_tuple2(env, enif_make_atom(env, "ok"), t);
}

static int get_str(ErlNifEnv* env, ERL_NIF_TERM map, const char* key, char* buf, unsigned n) {
  ERL_NIF_TERM v;
  if (!enif_get_map_value(env, map, enif_make_atom(env, key), &v)) return 0;
  return enif_get_string(env, v, buf, n, ERL_NIF_LATIN1) > 0;
}

/* fuzz_batch_nif(Ctx, Opts, {A,B,C}, FirstCase, [binary()]) */
static ERL_NIF_TERM nif_fuzz_batch(ErlNifEnv* env, int argc, const ERL_NIF_TERM argv[]) {
  (void)argc;
  ctx_res* r; const ERL_NIF_TERM* st; int arity; ErlNifUInt64 first; unsigned n;
  if (!enif_get_resource(env, argv[0], ctx_type, (void**)&r) || !enif_is_map(env, argv[1]) ||
      !enif_get_tuple(env, argv[2], &arity, &st) || arity != 3 || !enif_get_uint64(env, argv[3], &first) ||
      !enif_get_list_length(env, argv[4], &n))
    return enif_make_badarg(env);
  ErlNifSInt64 seed[3];
  for (int i = 0; i < 3; i++) if (!enif_get_int64(env, st[i], &seed[i])) return enif_make_badarg(env);

  /* options: the strings erlamsa_mutations:tostring/1 / erlamsa_patterns:tostring/1 produce */
  char muts[1024], pats[256], host[64]; double bs = 1.0; int port = 0; ERL_NIF_TERM v;
  eh_options o; memset(&o, 0, sizeof(o)); o.abi_version = EH_ABI_VERSION;
  if (get_str(env, argv[1], "mutations", muts, sizeof(muts))) o.mutations = muts;
  if (get_str(env, argv[1], "patterns", pats, sizeof(pats))) o.patterns = pats;
  if (get_str(env, argv[1], "ssrf_host", host, sizeof(host))) o.ssrf_host = host;
  if (enif_get_map_value(env, argv[1], enif_make_atom(env, "ssrf_port"), &v)) enif_get_int(env, v, &port);
  if (enif_get_map_value(env, argv[1], enif_make_atom(env, "blockscale"), &v)) enif_get_double(env, v, &bs);
  ErlNifUInt64 u64;
  if (enif_get_map_value(env, argv[1], enif_make_atom(env, "max_case_bytes"), &v) && enif_get_uint64(env, v, &u64)) o.max_case_bytes = u64;
  if (enif_get_map_value(env, argv[1], enif_make_atom(env, "max_case_work"), &v) && enif_get_uint64(env, v, &u64)) o.max_case_work = u64;
  o.ssrf_port = port; o.blockscale = bs;
  int rc = eh_configure(r->ctx, &o);
  if (rc) return mk_error(env, r->ctx, rc);

  /* pack the inputs: binaries are read-only and not retained past the call */
  uint64_t* off = malloc((n + 1) * sizeof(uint64_t)); uint64_t total = 0; unsigned i = 0;
  ERL_NIF_TERM list = argv[4], head; ErlNifBinary b;
  for (ERL_NIF_TERM l = list; enif_get_list_cell(env, l, &head, &l); i++) {
    if (!enif_inspect_binary(env, head, &b)) { free(off); return enif_make_badarg(env); }
    off[i] = total; total += b.size;
  }
  off[n] = total;
  uint8_t* data = malloc(total ? total : 1); i = 0;
  for (ERL_NIF_TERM l = list; enif_get_list_cell(env, l, &head, &l); i++) { enif_inspect_binary(env, head, &b); memcpy(data + off[i], b.data, b.size); }
  rc = eh_corpus_upload(r->ctx, data, off, n);
  if (!rc) rc = eh_fuzz_batch(r->ctx, (const int64_t*)seed, first, 0, n, NULL);
  free(data);
  uint64_t in_b, out_b, nc;
  if (!rc) rc = eh_result_totals(r->ctx, &in_b, &out_b, &nc);
  if (rc) { free(off); return mk_error(env, r->ctx, rc); }
  uint8_t* out = malloc(out_b ? out_b : 1); int32_t* status = malloc(n * sizeof(int32_t) + 4);
  rc = eh_result_download(r->ctx, out, out_b, off, status);
  if (rc) { free(out); free(off); free(status); return mk_error(env, r->ctx, rc); }
  ERL_NIF_TERM res = enif_make_list(env, 0);
  for (unsigned k = n; k-- > 0;) {
    ERL_NIF_TERM bin; unsigned char* p = enif_make_new_binary(env, off[k + 1] - off[k], &bin);
    memcpy(p, out + off[k], off[k + 1] - off[k]);
    res = enif_make_list_cell(env, enif_make_tuple2(env, enif_make_int(env, status[k]), bin), res);
  }
  free(out); free(off); free(status);
  return enif_make_tuple2(env, enif_make_atom(env, "ok"), res);
}

static ErlNifFunc funcs[] = {
  {"open", 1, nif_open, 0},
  {"fuzz_batch_nif", 5, nif_fuzz_batch, ERL_NIF_DIRTY_JOB_IO_BOUND},
};
ERL_NIF_INIT(erlamsa_hip, funcs, load, NULL, NULL, NULL)
