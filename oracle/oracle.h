// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product; only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
//
// C API of the CPU restatement of erlamsa's batch fuzz-case generation path
// (erlamsa_main:fuzzer/1 -> erlamsa_gen -> erlamsa_patterns -> erlamsa_mutations
//  -> erlamsa_rnd), see oracle/oracle.cpp for the per-function citations.
//
// PARITY UNPINNED: the reference cannot run on this image (no Erlang/OTP) and
// its own tests pin no byte-exact vectors; see DESIGN.md.
#pragma once
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct eo_config {
  // Dict keys of erlamsa_main:fuzzer/1 (erlamsa_main.erl:127-163)
  const char* mutations;   // "-m" syntax: "bd,bf=2,bi" ; NULL = default table
  const char* patterns;    // "-p" syntax: "od,nd,bu"   ; NULL = default table
  const char* generators;  // "direct=500,random=1" (also file, jump); NULL = direct mode default
  double blockscale;       // 1.0
  const char* ssrf_host;   // NULL = "localhost" (erlamsa_mutations.erl:698-703)
  int32_t ssrf_port;       // 0 = 51234
  // seeding mode:
  //  mode 0 (batch): one fuzzer/1 run with parent seed {s1,s2,s3}, n = ncases,
  //     case I (1-based, global index = first_case + i) gets corpus entry i.
  //  mode 1 (per-call): case i is its own fuzzer/1 run with n=1 and seed
  //     seeds[3*i..3*i+2]  (erlamsa_app:fuzz(Bin, #{seed => S})).
  int32_t mode;
  int64_t seed[3];
  uint64_t first_case;     // 1-based index of the first case of this batch (mode 0)
  const int64_t* seeds;    // mode 1
  uint64_t max_case_bytes; // engine cap mirrored here; 0 = unlimited
  uint64_t max_case_work;  // engine work budget mirrored here; 0 = unlimited
  double max_case_seconds; // wall-clock watchdog per case (the reference's maxrunningtime: the case's output is <<>>); 0 = none.
                           // Only bench.py's cpu_baseline leg sets it: parity tests never depend on time.
  // Paths of the `file` and `jump` generators (erlamsa_gen.erl:106-150): entry k = paths_data[paths_off[k] .. paths_off[k+1]).
  // NULL paths_off = the batch's own inputs.
  const uint8_t* paths_data;
  const uint64_t* paths_off;
  uint64_t paths_n;
} eo_config;

enum { EO_OK = 0, EO_CRASHED = 1, EO_OVERFLOW = 2, EO_UNSUPPORTED = 3, EO_BUDGET = 5, EO_TIMEOUT = 6 };

typedef struct eo_result {
  uint8_t* data;       // concatenated outputs
  uint64_t* off;       // n+1 offsets
  int32_t* status;     // n
  uint64_t* draws;     // n: PRNG draws consumed by the case worker (diagnostic)
  char* trace;         // optional '\n'-separated per-case meta trace (used/failed names)
  uint64_t trace_len;
} eo_result;

// returns 0 on success; on error returns nonzero and eo_last_error() explains.
int eo_fuzz_batch(const eo_config* cfg, const uint8_t* data, const uint64_t* off, uint64_t n,
                  int want_trace, eo_result* out);
void eo_free_result(eo_result* r);
const char* eo_last_error(void);

// building blocks exposed for unit tests ------------------------------------
// AS183: seeds with random:seed({a,b,c}) then writes `n` uniform() doubles.
void eo_rand_uniforms(int64_t a, int64_t b, int64_t c, uint64_t n, double* out);
// Runs ONE named mutator once on a single block with worker seed {a,b,c}
// (like the reference's eunit tests call Muta([Bin], [])).  Output = concatenation
// of the resulting block list; *nblocks = number of blocks; returns delta or
// INT32_MIN on crash.
int32_t eo_run_mutator(const char* name, int64_t a, int64_t b, int64_t c,
                       const uint8_t* in, uint64_t len, uint8_t** out, uint64_t* out_len,
                       uint32_t* nblocks);
// erlamsa_strlex:lex + unlex round trip; returns number of chunks, writes unlexed bytes.
int32_t eo_lex_roundtrip(const uint8_t* in, uint64_t len, uint8_t* out);
// kind 0: erlamsa_sgml fold_ast(parse(Bin)), counts = {N, NT, 0}; kind 1: erlamsa_json fold_ast(tokenize(Bin)),
// counts = {N, NT, NV}.  0 ok, -1 incorrect_sgml/incorrect_json, -2 crash.
int32_t eo_parse_fold(int32_t kind, const uint8_t* in, uint64_t len, uint8_t** out, uint64_t* out_len, int64_t* counts);
// order of lists:sort/2 with the strict '>' comparator of erlamsa_utils:sort_by_priority
// over `n` integer priorities; writes the permutation of input indices.
void eo_sort_by_priority(const int32_t* pri, uint32_t n, uint32_t* perm);
void eo_free(void* p);
// number of numerators for which reciprocal+FMA division differs from IEEE division (must be 0)
uint64_t eo_check_recip_div(void);

#ifdef __cplusplus
}
#endif
