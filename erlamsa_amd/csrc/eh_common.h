// eh_common.h — ids, tables and plain structs shared by the host API and the
// device kernels of liberlamsa_hip.so.
#pragma once
#include <stdint.h>

// Everything the kernels read and write besides their registers, the LDS and their stack is HBM (corpus arena, slots, pool areas,
// output arena, result arrays).  Device code says so in the pointer TYPE: address space 1, so loads and stores are global_*
// instructions.  A generic pointer makes them flat_*: an aperture check per access, and - what costs - a flat access counts in
// vmcnt AND lgkmcnt, so every LDS read of the per-case context (g_ctx) waits for the HBM traffic in flight.  Global converts to
// generic implicitly, never the other way round: a pointer into the LDS or the stack cannot end up in one of these types unseen.
// The host pass and the CPU emulator (tests/hipemu) see plain pointers.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIPEMU)
#define EH_G __attribute__((address_space(1)))
#else
#define EH_G
#endif

namespace eh {

typedef EH_G uint8_t* bptr;   typedef const EH_G uint8_t* cbptr;
typedef EH_G uint32_t* wptr;  typedef const EH_G uint32_t* cwptr;
typedef EH_G uint64_t* qptr;  typedef const EH_G uint64_t* cqptr;

// Mutator ids in the table order of erlamsa_mutations:mutations/1
// (reference src/erlamsa_mutations.erl:1291-1331).
enum MutaId : int {
  M_SGM, M_JS, M_UW, M_UI, M_AB, M_AD, M_TR2, M_TD, M_NUM, M_TS1, M_TR, M_TS2, M_BD, M_BEI, M_BED,
  M_BF, M_BI, M_BER, M_BR, M_SP, M_SR, M_SD, M_SNAND, M_SRND, M_LD, M_LDS, M_LR2, M_LRI, M_LR,
  M_LS, M_LP, M_LIS, M_LRS, M_FT, M_FN, M_FO, M_LEN, M_B64, M_URI, M_ZIP, M_NIL, M_COUNT
};
// Pattern ids in the table order of erlamsa_patterns:patterns/0 (erlamsa_patterns.erl:395-405).
enum PatId : int { P_OD, P_ND, P_BU, P_SK, P_SZ, P_CS, P_AR, P_CP, P_CO, P_NU, P_COUNT };
enum GenId : int { G_DIRECT = 0, G_RANDOM = 1, G_FILE = 2, G_JUMP = 3, G_COUNT = 4 };   // erlamsa_gen:generators/0 (stdin, genfuz: host I/O, not here)

enum CaseStatus : int { CASE_OK = 0, CASE_CRASHED = 1, CASE_OVERFLOW = 2, CASE_UNSUPPORTED = 3, CASE_ARENA_FULL = 4, CASE_BUDGET = 5 };

constexpr int MAX_FS = 64;          // mux_fuzzers list entries (one per lane of the wavefront)
constexpr int MAX_BLOCKS = 2048;    // block-list entries per case
constexpr int MAX_EMITS = 4096;     // deferred output pieces per case
constexpr int MAX_FRAMES = 16;      // nested sizer/csum wrappers
constexpr uint32_t TRACE_CAP = 32768; // meta-trace bytes kept per case (EH_FLAG_META_TRACE); the last byte is 0xFF when events were dropped

// ---- meta trace (ABI 7): the reference's Meta list of a case, element by element, in the order erlamsa_main.erl:58-70 prints it
// (lists:reverse(lists:flatten(Meta)), every element with ~p on a line of its own).  An event is a kind byte and its operands;
// atoms are ids into the table below (the 41 mutator codes first, in MutaId order), integers are LEB128 varints (zigzag where
// they can be negative).  eh_result_meta hands the bytes over as they are; eh_meta_atom_name names the atoms; the renderers are
// erlamsa_amd/meta.py and erlang/src/erlamsa_hip.erl meta_terms/1.
enum TraceKind : int {
  TRK_AA = 1,       // {Atom, Atom}            [a][b]                         {failed, sgm} {pattern, once_dec} {compressed, gzip}
  TRK_AI = 2,       // {Atom, Integer}         [a][zigzag varint]             {byte_drop, -1} {seq_repeat, 4096} {skipped_big, N}
  TRK_SIZER = 3,    // {sizer, {ok, Size, big | little, Len, A, B}}  [Size/8][big][varint Len][varint A][varint B]   erlamsa_patterns.erl:97
  TRK_CSUM = 4,     // {csum, {xor8 | crc32, Size, PLen, BLen}}      [crc][varint PLen][varint BLen]                  :131
  TRK_SKIPPED = 5,  // {skipped, Len/8}        [varint bytes]                 the float of a whole number :154
  TRK_ARCHIVER = 6, // {archiver, Name}        [varint n][n bytes]            the file name :183
};
#define EH_ATOMS(X)                                                                                                              \
  X(sgm) X(js) X(uw) X(ui) X(ab) X(ad) X(tr2) X(td) X(num) X(ts1) X(tr) X(ts2) X(bd) X(bei) X(bed) X(bf) X(bi) X(ber) X(br) X(sp)  \
  X(sr) X(sd) X(snand) X(srnd) X(ld) X(lds) X(lr2) X(lri) X(lr) X(ls) X(lp) X(lis) X(lrs) X(ft) X(fn) X(fo) X(len) X(b64) X(uri)  \
  X(zip) X(nil)                                                                                                                   \
  X(failed) X(used) X(pattern) X(skipped_big) X(once_dec) X(many_dec) X(burst) X(skipper) X(sizer) X(csum) X(archiver)            \
  X(compressed) X(no_muta) X(mutate_once) X(empty_stopped) X(decompressed) X(gzip) X(zlib) X(ok) X(success) X(json)               \
  X(base64_mutator) X(sed_utf8_widen) X(sed_utf8_insert) X(ascii_bad) X(ascii_delimeter) X(tree_dup) X(tree_del) X(muta_num)      \
  X(tree_swap_one) X(tree_stutter) X(tree_swap_two) X(byte_drop) X(byte_inc) X(byte_dec) X(byte_flip) X(byte_insert)              \
  X(byte_swap_random) X(byte_repeat) X(seq_perm) X(seq_repeat) X(seq_drop) X(seq_randmask) X(line_del) X(line_del_seq)            \
  X(line_dup) X(line_clone) X(line_repeat) X(line_swap) X(line_perm) X(list_ins) X(list_replace) X(fuse_this) X(fuse_next)        \
  X(fuse_old) X(muta_len) X(muta_zippath) X(nomutation) X(json_swap) X(json_dup) X(json_pump) X(json_repeat) X(json_insert)       \
  X(json_unserialize) X(json_innertext) X(null) X(bool) X(sgml_swap) X(sgml_dup) X(sgml_pump) X(sgml_repeat) X(sgml_insert2)      \
  X(sgml_permparams) X(sgml_breaktag) X(sgml_insert) X(sgml_xmlfeatures) X(xmlns) X(sgml_innertext)
enum AtomId : int {
#define EH_ATOM_ENUM(n) AT_##n,
  EH_ATOMS(EH_ATOM_ENUM)
#undef EH_ATOM_ENUM
  AT_COUNT
};
static_assert((int)AT_nil == (int)M_NIL && (int)AT_sgm == (int)M_SGM && (int)AT_b64 == (int)M_B64, "the mutator codes are the first atoms, in MutaId order");
constexpr int POOL_TIERS = 8;       // tiers of larger work areas a case can borrow from

// erlamsa.hrl:44-58
constexpr uint32_t INITIAL_IP = 24;
constexpr uint32_t AVG_BLOCK_SIZE = 2048;
constexpr uint32_t MIN_BLOCK_SIZE = 256;
constexpr uint32_t MAX_BLOCK_SIZE = 4096;
constexpr uint32_t ABSMAXHALF_BINARY_BLOCK = 500000;
constexpr uint32_t ABSMAX_BINARY_BLOCK = 1000000;
constexpr uint32_t SIZER_MAX_FIRST_BYTES = 512;
constexpr uint32_t PREAMBLE_MAX_BYTES = 32;

// A block (binary) of the case's lazy list: generic pointer + length.
struct Blk {
  uint64_t ptr;
  uint32_t len;
  uint32_t aux;
};

// Result of the per-run setup of erlamsa_main:fuzzer/1 (:134-158): parent PRNG state after the
// setup draws, generator choice, and the initial mux_fuzzers list in list order.
struct RunState {
  uint32_t a1, a2, a3;       // parent stream state after setup (next draws = ThreadSeeds)
  int32_t gen;               // GenId
  int32_t nfs;
  int32_t snand_mask;        // 0 nand, 1 or, 2 xor
  uint8_t fs_name[MAX_FS];   // MutaId per list position
  uint8_t fs_score[MAX_FS];
  uint32_t fs_pri[MAX_FS];
};

struct DevConfig {
  // selected mutators in TABLE order (make_mutator folds over the table): name id, pri
  int32_t nsel;
  uint8_t sel_name[MAX_FS];
  uint32_t sel_pri[MAX_FS];
  // patterns after erlamsa_utils:sort_by_priority (erlamsa_utils.erl:113-117)
  int32_t npat;
  int32_t pat_total;
  uint8_t pat_id[P_COUNT];
  uint32_t pat_pri[P_COUNT];
  // generators after sort_by_priority
  int32_t ngen;
  int32_t gen_total;
  uint8_t gen_id[G_COUNT];
  uint32_t gen_pri[G_COUNT];
  uint32_t max_block_scaled;  // round(MAX_BLOCK_SIZE * blockscale)
  uint32_t min_block_scaled;  // round(MIN_BLOCK_SIZE * blockscale)
  // SSRF endpoint strings pre-rendered on the host
  char ssrf_host[64];
  char ssrf_port[12];
};

// ---- cooperative execution of a heavy case's bulk loops (eh_device.h co_run / co_help) -------------------
// A case runs on ONE wavefront.  Loops over hundreds of kilobytes - block copies, the final concatenation, the streaming
// passes of eh_fuse2.h - are cut into chunks and posted on a board all contexts of the device share; wavefronts that are
// BETWEEN cases take chunks before they draw their next ticket, the poster takes chunks too and waits for the rest.
// word[j] = generation << 32 | chunks handed out (odd generation: open); a chunk is claimed by ONE atomic add, so its number and
// the job it belongs to are read together.
constexpr uint32_t CO_JOBS = 64, CO_ARGS = 12;
enum CoKind : uint32_t { CO_COPY = 1, CO_EQUAL = 2, CO_FBPASS = 3, CO_FBCOUNT = 4 };
struct CoJob {
  uint64_t a[CO_ARGS];          // arguments of the loop (kind-specific)
  unsigned long long acc;       // what the chunks add up to (members alive, mismatches found ...)
  uint32_t kind, nchunks, done, pad;
};
struct CoBoard {
  unsigned long long word[CO_JOBS];
  uint32_t owner[CO_JOBS];      // 1: a poster holds the entry
  uint32_t open;                // entries with chunks to hand out
  uint32_t posters;             // cases under way that have posted a loop: wavefronts whose pass has run out of tickets stay for chunks while there are any
  uint32_t pad[14];
  unsigned long long stat[8];   // [0] jobs posted, [1] chunks run by helpers, [2] by the posters themselves, [3] cycles posters waited, [4] jobs that found no free entry
  CoJob job[CO_JOBS];
};

struct KParams {
  cbptr corpus;
  cqptr coff;
  uint64_t corpus_first;
  uint64_t n_paths;           // entries of the whole corpus: the Paths of the file / jump generators
  uint64_t n;
  uint64_t first_case;        // 1-based case number of case 0 (mode 0)
  int32_t mode;               // 0 batch (one parent seed), 1 per-call seeds
  const EH_G RunState* run;        // mode 0
  const EH_G int64_t* seeds;       // mode 1: 3n
  DevConfig cfg;
  uint64_t work_cap;          // bytes of the linear work area of a tier-0 slot
  uint64_t work_budget;       // per-case byte budget (sum of block sizes handed to mutators)
  uint64_t fuse_stream_min;   // fuse/2 on la + lb >= this many bytes runs as the position-indexed refinement of eh_fuse2.h
  // outputs
  bptr out;
  uint64_t out_cap;
  EH_G unsigned long long* out_cursor;
  qptr out_off;
  qptr out_len;
  EH_G int32_t* status;
  qptr draws;
  EH_G int32_t* lastm;
  qptr cycles;           // per-case shader-clock ticks (diagnostic)
  qptr peak;             // per-case work-memory high-water mark in bytes (diagnostic)
  qptr trace_off;        // EH_FLAG_META_TRACE: where the case's trace sits in `out` ...
  wptr trace_len;        // ... and its length in bytes (TraceKind events, above)
  uint32_t flags;             // EH_FLAG_*
  EH_G unsigned long long* prof;   // EH_PROF builds: [2*k] cycles, [2*k+1] calls; k < 64 mutator fn, 64.. phases
  EH_G unsigned long long* ticket;
  EH_G unsigned long long* in_bytes;
  // Work areas.  Workgroup w of a batch owns slot w of the context (block tables + work_cap bytes of work area): a slot per
  // launched workgroup, so no wavefront ever waits for one.  A case that outgrows what it holds borrows a larger area from
  // a POOL all contexts of the device share (eh_engine.hip, DevPool; tiers 1..ntiers, twice the area per tier up to
  // big_case_bytes) and goes on there; only the mutator attempt that ran out of memory is repeated (eh_device.h, ws_regrow).
  // Every tier is a ring of free area indices: pool_ctr[2t] = pop tickets, pool_ctr[2t+1] = push tickets,
  // pool_ctr[20+t] = ticks wavefronts waited for an area of tier t, pool_ctr[30+t] = how many had to wait.
  bptr slot_base;
  uint64_t slot_stride;
  int32_t ntiers;                // tiers of the pool (1..ntiers)
  bptr pool_base[POOL_TIERS + 1];   // tier t: area k at pool_base[t] + k * pool_stride[t]
  uint64_t pool_stride[POOL_TIERS + 1];
  uint64_t pool_cap[POOL_TIERS + 1];    // work-area bytes of an area of tier t (pool_cap[0] == work_cap: the slot's own)
  uint32_t pool_cnt[POOL_TIERS + 1];
  wptr pool_ring[POOL_TIERS + 1];
  EH_G unsigned long long* pool_ctr;
  EH_G CoBoard* board;          // nullptr: every case does all its work itself (EH_FLAG_NO_COOP)
  EH_G unsigned long long* summary_out;  // page-locked HOST memory: the last workgroup to leave writes the batch's totals there (eh_result_summary reads them without a copy call)
  uint64_t batch_seq;           // ... stamped with this number
  uint32_t co_copy_min, co_copy_chunk;   // bytes: copies / compares of co_copy_min and more are posted in chunks of co_copy_chunk
  uint32_t co_fb_min, co_fb_chunk;       // positions: the same for the streaming passes of eh_fuse2.h (chunk: a multiple of 1024)
  uint32_t co_linger, pad_co;            // wavefronts of a pass that stay for posted chunks once the pass is out of tickets
};

struct MutaInfo { const char* name; int pri; int on_gpu; };
struct PatInfo { const char* name; int pri; int on_gpu; };

}  // namespace eh
