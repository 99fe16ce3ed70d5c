/* Minimal stand-in for OTP's <erl_nif.h>, for COMPILE CHECKS ONLY (tests/test_abi.py builds
 * erlang/c_src/erlamsa_hip_nif.c against it with -fsyntax-only).  This image has no Erlang/OTP; the
 * declarations below follow the documented public NIF API (erl_nif(3)) for exactly the functions the
 * shim uses.  Never link against this. */
#ifndef ERL_NIF_STUB_H
#define ERL_NIF_STUB_H
#include <stddef.h>
#include <stdint.h>
typedef uintptr_t ERL_NIF_TERM;
typedef struct enif_environment_t ErlNifEnv;
typedef struct enif_resource_type_t ErlNifResourceType;
typedef uint64_t ErlNifUInt64;
typedef int64_t ErlNifSInt64;
typedef struct { size_t size; unsigned char* data; void* ref_bin; void* spare[2]; } ErlNifBinary;
typedef void ErlNifResourceDtor(ErlNifEnv*, void*);
typedef enum { ERL_NIF_RT_CREATE = 1, ERL_NIF_RT_TAKEOVER = 2 } ErlNifResourceFlags;
typedef enum { ERL_NIF_LATIN1 = 1 } ErlNifCharEncoding;
#define ERL_NIF_DIRTY_JOB_CPU_BOUND 1
#define ERL_NIF_DIRTY_JOB_IO_BOUND 2
typedef struct { const char* name; unsigned arity; ERL_NIF_TERM (*fptr)(ErlNifEnv*, int, const ERL_NIF_TERM[]); unsigned flags; } ErlNifFunc;
ErlNifResourceType* enif_open_resource_type(ErlNifEnv*, const char* module_str, const char* name, ErlNifResourceDtor* dtor, ErlNifResourceFlags flags, ErlNifResourceFlags* tried);
void* enif_alloc_resource(ErlNifResourceType*, size_t);
void enif_release_resource(void*);
ERL_NIF_TERM enif_make_resource(ErlNifEnv*, void*);
int enif_get_resource(ErlNifEnv*, ERL_NIF_TERM, ErlNifResourceType*, void**);
int enif_get_int(ErlNifEnv*, ERL_NIF_TERM, int*);
int enif_get_int64(ErlNifEnv*, ERL_NIF_TERM, ErlNifSInt64*);
int enif_get_uint64(ErlNifEnv*, ERL_NIF_TERM, ErlNifUInt64*);
int enif_get_double(ErlNifEnv*, ERL_NIF_TERM, double*);
int enif_get_string(ErlNifEnv*, ERL_NIF_TERM, char*, unsigned, ErlNifCharEncoding);
int enif_get_tuple(ErlNifEnv*, ERL_NIF_TERM, int*, const ERL_NIF_TERM**);
int enif_get_list_length(ErlNifEnv*, ERL_NIF_TERM, unsigned*);
int enif_get_list_cell(ErlNifEnv*, ERL_NIF_TERM, ERL_NIF_TERM*, ERL_NIF_TERM*);
int enif_get_map_value(ErlNifEnv*, ERL_NIF_TERM map, ERL_NIF_TERM key, ERL_NIF_TERM* value);
int enif_is_map(ErlNifEnv*, ERL_NIF_TERM);
int enif_is_list(ErlNifEnv*, ERL_NIF_TERM);
int enif_compare(ERL_NIF_TERM lhs, ERL_NIF_TERM rhs);
int enif_inspect_binary(ErlNifEnv*, ERL_NIF_TERM, ErlNifBinary*);
unsigned char* enif_make_new_binary(ErlNifEnv*, size_t, ERL_NIF_TERM*);
ERL_NIF_TERM enif_make_atom(ErlNifEnv*, const char*);
ERL_NIF_TERM enif_make_int(ErlNifEnv*, int);
ERL_NIF_TERM enif_make_uint64(ErlNifEnv*, ErlNifUInt64);
ERL_NIF_TERM enif_make_string(ErlNifEnv*, const char*, ErlNifCharEncoding);
ERL_NIF_TERM enif_make_badarg(ErlNifEnv*);
ERL_NIF_TERM enif_make_list(ErlNifEnv*, unsigned cnt, ...);
ERL_NIF_TERM enif_make_list_cell(ErlNifEnv*, ERL_NIF_TERM, ERL_NIF_TERM);
ERL_NIF_TERM enif_make_tuple(ErlNifEnv*, unsigned cnt, ...);
#define enif_make_tuple2(env, a, b) enif_make_tuple(env, 2, a, b)
#define enif_make_tuple3(env, a, b, c) enif_make_tuple(env, 3, a, b, c)
#define enif_make_tuple4(env, a, b, c, d) enif_make_tuple(env, 4, a, b, c, d)
#define ERL_NIF_INIT(MOD, FUNCS, LOAD, RELOAD, UPGRADE, UNLOAD) \
  const ErlNifFunc* erl_nif_stub_funcs_##MOD(void) { (void)(LOAD); return FUNCS; }
/* erl_nif.h: thread API */
typedef struct ErlNifMutex_ ErlNifMutex;
ErlNifMutex* enif_mutex_create(char* name);
void enif_mutex_destroy(ErlNifMutex* mtx);
void enif_mutex_lock(ErlNifMutex* mtx);
void enif_mutex_unlock(ErlNifMutex* mtx);

#endif
