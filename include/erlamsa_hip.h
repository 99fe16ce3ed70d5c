/* erlamsa_hip.h — C ABI of liberlamsa_hip.so, the MI355X-native batch mutation
 * engine that sits behind erlamsa's own entry points.
 *
 * What each entry point replaces on the reference side (paths relative to the
 * reference repo's src/):
 *
 *   eh_configure        <- the option map read by erlamsa_main:fuzzer/1
 *                          (erlamsa_main.erl:127-163: seed, mutations, patterns,
 *                          generators, blockscale) and the "-m/-p/-g" list syntax of
 *                          erlamsa_cmdparse:string_to_actions/3 (erlamsa_cmdparse.erl:232-257);
 *                          SSRF endpoint = ETS global_config keys cm_host/cm_port
 *                          (erlamsa_mutations.erl:698-731)
 *   eh_corpus_upload /  <- the `input => Bin` key of erlamsa_utils:get_direct_fuzzing_opts/2
 *   eh_corpus_attach       (erlamsa_utils.erl:55-58), batched: N binaries as a packed
 *                          offset/length arena
 *   eh_fuzz_batch       <- erlamsa_main:fuzzer/1 with n = N, workers = 1
 *                          (erlamsa_main.erl:125-247): one parent seed, case I takes the I-th
 *                          gen_predictable_seed() of the parent stream (erlamsa_main.erl:179)
 *   eh_fuzz_calls       <- N independent erlamsa_app:fuzz(Bin, #{seed => S}) calls
 *                          (erlamsa_app.erl:255-263), i.e. what erlamsa_esi:call_fuzzer/3 ->
 *                          erlamsa_fsupervisor:get_fuzzing_output/1 does per HTTP request
 *                          (erlamsa_esi.erl:86-95, erlamsa_fsupervisor.erl:60-86)
 *   eh_result_*         <- the [binary()] returned by fuzzer/1 / the binary returned by
 *                          erlamsa_app:fuzz/2 (per-case status lets the shim rebuild
 *                          record_result/2's dropping of <<>> results, erlamsa_main.erl:120-122)
 *
 * Everything is plain pointers and sizes; no C++ or torch types cross this ABI.
 * All functions return 0 on success or a negative eh_error code; none throws.
 * A context is not thread-safe; use one context per calling thread / per GPU.
 */
#ifndef ERLAMSA_HIP_H
#define ERLAMSA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define EH_ABI_VERSION 8

typedef struct eh_ctx eh_ctx;

enum eh_error {
  EH_OK = 0,
  EH_E_INVALID = -1,      /* bad argument / unknown mutator or pattern name */
  EH_E_NODEVICE = -2,     /* no usable HIP device */
  EH_E_HIP = -3,          /* a HIP runtime call failed (see eh_last_error) */
  EH_E_NOMEM = -4,
  EH_E_STATE = -5,        /* call order: configure -> corpus -> fuzz -> result */
  EH_E_UNSUPPORTED = -6,  /* option names a mutator/pattern this build does not run on the GPU */
  EH_E_AGAIN = -7         /* eh_poll: the request has not been launched yet (eh_flush it) */
};

/* per-case status (eh_result_*) */
enum eh_case_status {
  EH_CASE_OK = 0,
  EH_CASE_CRASHED = 1,     /* the reference worker would have died (badmatch/badarith...):
                              output is <<>> (erlamsa_main.erl:211-220) */
  EH_CASE_OVERFLOW = 2,    /* exceeded max_case_bytes / block-table / arena capacity: output empty */
  EH_CASE_UNSUPPORTED = 3, /* a zip archive (pattern ar, mutator zip) with a feature whose outcome in OTP's prim_zip / zip this
                              build does not pin: ZIP64 markers, encrypted or data-descriptor entries, directory entries, an
                              empty or non-ASCII name, a local header outside the file (csrc/eh_zip.h).  Output empty; route the
                              case to BEAM.  gzip / zlib inputs (pattern cp) and ordinary zip archives run on the GPU */
  EH_CASE_ARENA_FULL = 4,  /* the output arena (eh_options.out_capacity) was exhausted; re-run the
                              case with a larger arena */
  EH_CASE_BUDGET = 5       /* exceeded max_case_work: the engine's deterministic stand-in for the
                              reference's maxrunningtime watchdog (erlamsa_main.erl:211-220), which
                              also yields <<>> */
};

typedef struct eh_options {
  uint32_t abi_version;      /* EH_ABI_VERSION */
  const char* mutations;     /* "bd,bf=2,num" (-m syntax); NULL = default table
                                (erlamsa_mutations.erl:1291-1331) */
  const char* patterns;      /* "od,nd,bu" (-p syntax); NULL = default (erlamsa_patterns.erl:395-405) */
  const char* generators;    /* "direct=500,random=1"; NULL = what paths=[direct] yields
                                (erlamsa_gen.erl:204-241,250-257).  Also "file" (file_streamer :106-121) and "jump"
                                (jump_streamer :136-150, jump_somewhere :124-133): their Paths are the entries of the
                                corpus (all of them, whatever sub-range a batch runs); a case draws its path(s), the
                                stream is cut into rand_block_size blocks on the device, and - like the reference's
                                fun - only when the pattern first looks at the list.  A launch with `file` needs one
                                corpus entry, with `jump` two (make_generator_fun would drop them: EH_E_INVALID here).
                                stdin / genfuz are host I/O and not part of the GPU path (EH_E_UNSUPPORTED) */
  double blockscale;         /* 0 => 1.0 (erlamsa_gen.erl:206) */
  const char* ssrf_host;     /* NULL => "localhost" */
  int32_t ssrf_port;         /* 0 => 51234 */
  uint64_t max_case_bytes;   /* work area of a slot; every workgroup (= wavefront) of a batch owns one; 0 => default (8 MiB).
                                A case that outgrows it goes on in larger areas borrowed from the device's work-area pool,
                                which contexts that ask for the same max_case_bytes, big_case_bytes and pool_bytes share */
  uint64_t out_capacity;     /* output arena bytes; 0 => 8 x batch input bytes + 2 GiB */
  uint64_t max_case_work;    /* OPTIONAL per-case work budget in bytes (sum over mutator attempts, failed ones
                                included, of block size x cost weight of the mutator: 8 for parsers and
                                per-byte-draw mutators, 64 for the fuse family and zip, 4 for num, 1 otherwise; plus 16 x
                                the list members of every fuse refinement round; plus 64 x the bytes the container
                                patterns cp / ar read, inflate and deflate);
                                0 => no budget (default): every case runs to completion */
  uint32_t max_slots;        /* slots = persistent workgroups (wavefronts) of one batch; 0 => one per wavefront the device holds.
                                Batches in flight on several streams may oversubscribe the device: a batch's workgroups
                                start as those of earlier batches leave */
  uint32_t flags;            /* EH_FLAG_* */
  uint64_t big_case_bytes;   /* largest work area of the pool.  A case that outgrows what it holds goes on in a larger area
                                borrowed from the pool (2x per tier, at most 8 tiers, the last one has this size; only the mutator
                                attempt that ran out of memory is repeated); a case that outgrows this too ends as
                                EH_CASE_OVERFLOW.  0 => 32 x max_case_bytes, at most 1 GiB; <= max_case_bytes => slots only */
  /* ABI 4: engine tuning that used to be environment variables of the library */
  uint64_t pool_bytes;       /* device memory of the work-area pool (all tiers together); 0 => a quarter of the memory that is
                                free when the pool is made, 1 .. 64 GiB.  Every tier holds at least one area */
  uint64_t download_chunk_bytes; /* bounce-buffer size of eh_result_download's device gather; 0 => 256 MiB */
  uint64_t fuse_stream_min;  /* erlamsa_fuse:fuse/2 on la + lb >= this many bytes runs as the position-indexed class
                                refinement (csrc/eh_fuse2.h) instead of the node-list refinement (csrc/eh_fuse.h);
                                results are identical.  0 => 16384 */
  /* ABI 7 */
  uint32_t sequence_muta;    /* the reference's --consequtive-mutators / sequence_muta (erlamsa_main.erl:223-235): the mutator
                                scores of case I are what case I-1 left behind - a serial chain over the cases of a run, which a
                                batch of independent cases cannot keep.  Non-zero => eh_configure refuses with EH_E_UNSUPPORTED
                                ("route this run to the BEAM path"); the shim (erlang/src/erlamsa_hip.erl) checks the same key */
  uint32_t reserved0;
} eh_options;

#define EH_FLAG_NO_COOP 64u        /* diagnostic: every case does all of its work on its own wavefront (no cooperative execution of
                                     the bulk loops of heavy cases, csrc/eh_common.h CoBoard; ABI 8).  Same bytes either way. */
#define EH_FLAG_ORDERED_OUTPUT 1u /* compact the output arena into case order after the batch */
#define EH_FLAG_META_TRACE 2u     /* keep every case's meta trace (eh_result_meta) */
#define EH_FLAG_SGML_NO_LANES 32u  /* diagnostic: the sgm tokenizer makes its tag attempts one after the other instead of 64 at a time, one per
                                     lane (csrc/eh_sgml.h); results are identical */
#define EH_FLAG_SGML_NO_REPLAY 16u /* diagnostic: the sgm tokenizer walks periodic documents tag by tag instead of replaying one period's
                                     tokens (csrc/eh_sgml.h); results are identical */
#define EH_FLAG_FUSE_NO_REDUCE 8u /* diagnostic: erlamsa_fuse:fuse/2 on large lists searches the lists as they are instead of copies with the
                                     long periodic stretches cut short (csrc/eh_fuse_red.h); results are identical */
#define EH_FLAG_FUSE_NO_LDS 4u    /* diagnostic: erlamsa_fuse:fuse/2 on small lists runs as the node-list refinement (csrc/eh_fuse.h)
                                     instead of the LDS-resident one (csrc/eh_fuse_lds.h); results are identical */

/* One context = one HIP device, its result buffers and slots.  eh_create touches no process-wide setting (until ABI 7 it raised the
 * device's hipLimitStackSize to 6 KiB per lane because the kernel recursed; the kernel's stack is static now). */
int eh_create(int device, eh_ctx** out);
void eh_destroy(eh_ctx* ctx);
int eh_configure(eh_ctx* ctx, const eh_options* opts);

/* Corpus = packed arena: data[off[i] .. off[i+1]) is seed binary i; off has n+1 entries. */
int eh_corpus_upload(eh_ctx* ctx, const uint8_t* data, const uint64_t* off, uint64_t n);
/* Same, but both arrays already live in this device's memory (e.g. after an RCCL
 * broadcast); the engine does not take ownership. */
int eh_corpus_attach(eh_ctx* ctx, const void* d_data, const void* d_off, uint64_t n, uint64_t nbytes);

/* ---- multi-GPU: the seed arena on every GPU of the node, over RCCL called from inside the library (csrc/eh_comm.h, ABI 7) ----
 * Cases are independent, so the mutation path never communicates; the one exchange is the arena at load time.
 * The reference's counterpart: every `--workers` scheduler reads the same input files (erlamsa_main.erl:90-108).  RCCL is reached
 * from here because the host north_star names - the BEAM - has no HIP or RCCL binding.  librccl.so is loaded on first use
 * (an RCCL the process has loaded already - a host that also uses torch.distributed - is shared, not loaded a second time;
 * environment EH_RCCL_LIB names another file); a single-GPU host never needs it.  EH_E_UNSUPPORTED when it cannot be loaded.
 *
 * One OS process per GPU (how bench.py is launched):
 *   rank 0: eh_comm_unique_id(id); the 128 bytes reach the other processes over the host's own channel (Erlang distribution, a
 *   file, torch.distributed's store).  Every rank: eh_comm_init(ctx, id, rank, nranks) - collective, returns when all have joined.
 *   eh_corpus_broadcast(ctx, root, data, off, n)    root passes the arena (host pointers), the others NULL / 0; afterwards every
 *                                                   context holds the corpus as after eh_corpus_upload (BASELINE configs[3]).
 *   eh_corpus_allgather(ctx, data, off, n_local)    every rank passes ITS shard, the same number of entries and bytes on all ranks
 *                                                   (configs[4]: fixed-size seeds); afterwards every context holds the shards in
 *                                                   rank order - each xGMI link carries 1/nranks of the arena.
 * One process, several GPUs (what one BEAM node is): eh_comm_init_local(ctxs, n) - one context per device, ncclCommInitAll - then
 *   eh_corpus_broadcast_local(ctxs, n, root) copies the corpus loaded on ctxs[root] to the devices of all the others.
 * Cases are then sharded by case number: rank r of W runs eh_fuzz_batch(first_case + a_r, corpus_first + a_r, n_r) for its
 * contiguous range (erlamsa_amd/shard.py case_range = the shape of erlamsa_main:get_threading_mode/3, erlamsa_main.erl:95-108);
 * results never depend on W. */
int eh_device_count(void);   /* HIP devices this process sees (0 when there is none or the runtime fails) */
int eh_comm_unique_id(uint8_t id[128]);
int eh_comm_init(eh_ctx* ctx, const uint8_t id[128], int rank, int nranks);
int eh_comm_init_local(eh_ctx** ctxs, int nctx);
int eh_comm_destroy(eh_ctx* ctx);
int eh_corpus_broadcast(eh_ctx* ctx, int root, const uint8_t* data, const uint64_t* off, uint64_t n);
int eh_corpus_allgather(eh_ctx* ctx, const uint8_t* data, const uint64_t* off, uint64_t n_local);
int eh_corpus_broadcast_local(eh_ctx** ctxs, int nctx, int root);

/* Device-side view of the loaded corpus (after eh_corpus_upload / eh_corpus_attach): lets further contexts of the same
 * device share one arena (eh_corpus_attach on them) instead of holding a copy each.  Any pointer may be NULL. */
int eh_corpus_device(eh_ctx* ctx, const void** d_data, const void** d_off, uint64_t* n, uint64_t* nbytes);

/* A HIP stream owned by the context (non-blocking, made on first request, destroyed with the context): pass it as the
 * `stream` of eh_fuzz_batch / eh_fuzz_calls to run the batches of several contexts side by side without the host
 * needing a HIP binding of its own (the BEAM has none: erlang/c_src uses this; so does bench.py). */
int eh_stream(eh_ctx* ctx, void** stream);

/* Page-locked host memory for eh_result_download / eh_corpus_upload at the full PCIe rate (hipHostMalloc / hipHostFree);
 * for hosts without a HIP binding. */
int eh_host_alloc(void** p, uint64_t bytes);
void eh_host_free(void* p);

/* Allocates all device memory batches of up to `max_cases` cases will need (result arrays, per-slot work
 * areas, output arena sized from eh_options.out_capacity or 8 x corpus bytes + 2 GiB), so that later
 * eh_fuzz_batch / eh_fuzz_calls launches never allocate or free.  Optional: the first batch does the same
 * on demand.  Reference counterpart: none (BEAM allocates per worker process); it exists because
 * erlamsa_fsupervisor-style services want a flat first-request latency. */
int eh_reserve(eh_ctx* ctx, uint64_t max_cases);

/* Case i of this call (0 <= i < n) mutates corpus entry corpus_first + i and is case number
 * first_case + i (1-based) of a fuzzer/1 run seeded with `seed`.  Results do not depend on how
 * a run is cut into calls, GPUs or streams.  `stream` is a hipStream_t (NULL = default
 * stream); the call enqueues work and returns, eh_result_* / eh_sync wait for it. */
int eh_fuzz_batch(eh_ctx* ctx, const int64_t seed[3], uint64_t first_case, uint64_t corpus_first, uint64_t n,
                  void* stream);
/* Case i is its own fuzzer/1 run (n = 1) with seed seeds[3i..3i+2] (host pointer). */
int eh_fuzz_calls(eh_ctx* ctx, const int64_t* seeds, uint64_t corpus_first, uint64_t n, void* stream);
int eh_sync(eh_ctx* ctx);
/* Non-blocking: *done = 1 when the context's last batch has finished (or none was launched).  A host that keeps several contexts
 * busy polls this to hand the next batch to whichever context is free (bench.py / erlamsa_amd/shard.py run_steps); no reference
 * counterpart (BEAM's receive ... after does the same for its worker processes, erlamsa_main.erl:211-220). */
int eh_batch_done(eh_ctx* ctx, int* done);

/* Request coalescing for services — what erlamsa_fsupervisor / erlamsa_esi do one request at a time
 * (erlamsa_esi.erl:86-95 call_fuzzer/3 -> erlamsa_fsupervisor:get_fuzzing_output/1, erlamsa_fsupervisor.erl:60-86: one
 * erlamsa_app:fuzz(Bin, #{seed => S}) per HTTP request).  Requests from any number of host threads are collected and
 * run as ONE eh_fuzz_calls batch; every request keeps the result it would have had alone (a case is a pure function of
 * its seed, its input and the configuration).
 *   eh_submit  copies the request into the pending batch and returns its ticket; when the pending batch reaches
 *              `flush_cases` requests or `flush_bytes` input bytes (eh_coalesce_limits, defaults 4096 / 64 MiB) it is
 *              launched at once.
 *   eh_flush   launches whatever is pending (what a service calls from its 200 us timer); no-op when nothing is.
 *   eh_poll    result of one request: EH_OK (out/out_len/status filled, the ticket is consumed), EH_E_AGAIN when the
 *              ticket is still pending (not flushed yet), EH_E_INVALID for an unknown or already consumed ticket or when
 *              `cap` is too small (out_len then says how much is needed and the ticket stays valid).  Waits for the
 *              batch the ticket was launched in.
 * The three calls are thread safe with respect to each other; they use the context's corpus slot and result buffers,
 * so a context used for coalescing is not used for eh_fuzz_batch at the same time.
 * eh_configure on such a context: refused (EH_E_STATE) while requests are pending - eh_flush them first, they keep the options they
 * were submitted under; a batch already in flight is collected (its results wait for eh_poll), then the options change. */
int eh_coalesce_limits(eh_ctx* ctx, uint64_t flush_cases, uint64_t flush_bytes);
int eh_submit(eh_ctx* ctx, const uint8_t* data, uint64_t len, const int64_t seed[3], uint64_t* ticket);
int eh_flush(eh_ctx* ctx);
int eh_poll(eh_ctx* ctx, uint64_t ticket, uint8_t* out, uint64_t cap, uint64_t* out_len, int32_t* status);
/* Gives a ticket up: a request that has not been launched leaves its batch, a launched one is dropped when the batch is
 * collected, a finished one is freed (a service whose client timed out or died calls this so that results do not pile up). */
int eh_cancel(eh_ctx* ctx, uint64_t ticket);

/* Device-side view of the last batch: out_data[out_off[i] .. out_off[i]+out_len[i]) is the
 * output of case i.  Pointers stay valid until the next eh_fuzz_* call on this context. */
int eh_result_device(eh_ctx* ctx, const uint8_t** d_data, const uint64_t** d_off, const uint64_t** d_len,
                     const int32_t** d_status, uint64_t* total_bytes);
/* Copies to host: `data` receives total_bytes (<= cap) bytes laid out in case order
 * regardless of EH_FLAG_ORDERED_OUTPUT; off has n+1 entries.  Any pointer may be NULL. */
int eh_result_download(eh_ctx* ctx, uint8_t* data, uint64_t cap, uint64_t* off, int32_t* status);
/* One case of the last batch: copies its output (<= cap bytes) to `buf` and stores its length; EH_E_INVALID with
 * *out_len set when cap is too small.  For consumers that want a few results of a large batch. */
int eh_result_fetch(eh_ctx* ctx, uint64_t i, uint8_t* buf, uint64_t cap, uint64_t* out_len);
/* Totals of the last batch (sum of input bytes read, output bytes written, cases). */
int eh_result_totals(eh_ctx* ctx, uint64_t* in_bytes, uint64_t* out_bytes, uint64_t* n_cases);
/* The same and the cases by status, summed on the device (ABI 8): out[0] input bytes, out[1] output bytes, out[2] cases,
 * out[3 + s] = cases with status s (EH_CASE_OK .. EH_CASE_BUDGET).  One 2 KiB copy whatever the batch size: for host loops over
 * many small batches (eh_result_totals copies every case's length). */
int eh_result_summary(eh_ctx* ctx, uint64_t* out);
/* Where the wave slots' time went in the last batch (diagnostic; written with the summary, no copy call), in ticks of the device's
 * 100 MHz clock summed over the batch's workgroups: out[0] from a workgroup's start to its end, out[1] of that inside cases, out[2]
 * staying for other cases' posted loops after the batch ran out of cases; out[3] = workgroups launched, out[4] = wavefronts of this
 * kernel the device holds at once (its wave slots).  Summed over the batches of a run and divided by (wall time x 1e8 x out[4]),
 * out[0] says how full the device was and out[1] how much of it did the cases' own work. */
int eh_result_occupancy(eh_ctx* ctx, uint64_t* out);
/* Per-case diagnostics of the last batch (device->host): PRNG draws consumed by the worker and
 * the id (index in eh_mutator_name) of the last mutator that fired, -1 if none; for an EH_CASE_OVERFLOW case, minus the
 * id of the capacity check that gave up (EH_SET_OVERFLOW sites in csrc/, a diagnostic).  May be NULL. */
int eh_result_diag(eh_ctx* ctx, uint64_t* draws, int32_t* last_mutator);

/* Per-case shader-clock ticks spent by the wavefront that ran the case (diagnostic). */
int eh_result_cycles(eh_ctx* ctx, uint64_t* cycles);

/* The file sink of erlamsa_out (erlamsa_out.erl:103-123, `-o "name-%n.ext"`): writes every case of the last batch that
 * ended EH_CASE_OK to the file named by the template with each "%n" replaced by the case number (first_number + i), as
 * build_name/3 does.  The results come to the host in one case-ordered download; `threads` host threads write the files
 * (0 => 8).  Cases with another status leave no file (the reference opens the file inside the worker, after the mutation)
 * and are counted in *not_written.  Any of the three counters may be NULL. */
int eh_result_write_files(eh_ctx* ctx, const char* name_template, uint64_t first_number, uint32_t threads,
                          uint64_t* files_written, uint64_t* bytes_written, uint64_t* not_written);

/* Meta trace of case i of the last batch (EH_FLAG_META_TRACE): the reference's Meta list of the case IN FULL - what erlamsa's -M /
 * meta logger prints (erlamsa_main.erl:58-70: lists:reverse(lists:flatten(Meta)), every element with ~p on a line of its own),
 * for the part of the list the batch path builds (from Pat(Ll, Muta, Meta) down; {nth, I}, the generator's, the output's and
 * {written, N} are the host's: erlang/src/erlamsa_hip.erl meta_terms/3 adds them): {pattern, once_dec | many_dec | burst |
 * skipper | sizer | csum | archiver | compressed | no_muta}, {sizer, Elem}, {csum, Elem}, {skipped, F}, {archiver, _},
 * {decompressed | compressed, _}, {mutate_once, empty_stopped}, every mutator's own entry ({byte_drop, D}, {seq_repeat, BSize},
 * {muta_num, 0 | 1}, {line_del, 1}, {fuse_this, D}, {tree_dup, 1}, {ascii_bad, D}, {muta_len, D}, {base64_mutator, D},
 * {uri, success}, {sgml_swap, 1} .. {sgml_innertext, 1}, {json_swap, 1} .. {json_innertext, _}, {failed, json} ...), {used, Name},
 * {failed, Name}, {skipped_big, Size} - nested scheduler calls included, in the order the reference prints them, with its quirk
 * that sgml_mutate / json_mutate drop the list when the block comes back unchanged (erlamsa_sgml.erl:748-749).
 * Encoding (csrc/eh_common.h TraceKind): a kind byte and its operands - 1 {Atom, Atom}: two atom ids; 2 {Atom, Int}: atom id,
 * zigzag LEB128; 3 sizer: Size/8, big, then LEB128 Len, A, B; 4 csum: crc32?, LEB128 PLen, BLen; 5 skipped: LEB128 bytes (printed
 * as a float); 6 {archiver, Name}: LEB128 n, n bytes.  Atom ids: eh_meta_atom_name (the mutator codes come first, in table order).
 * *n_events = BYTES of the trace; min(*n_events, cap) of them are copied into buf and the call returns EH_OK either way, so
 * (buf = NULL, cap = 0) asks for the length (ABI 8; until ABI 7 a buffer that was too small was EH_E_INVALID).  At most 32768 bytes per case are kept; the last byte is 0xFF when events
 * were dropped.  Renderers to ~p text: erlamsa_amd/meta.py, erlang/src/erlamsa_hip.erl. */
int eh_meta_atom_count(void);
const char* eh_meta_atom_name(int id);
int eh_result_meta(eh_ctx* ctx, uint64_t i, uint8_t* buf, uint64_t cap, uint64_t* n_events);

/* Per-case high-water mark of work memory in bytes (diagnostic; what sizes max_case_bytes and the pool's tiers). */
int eh_result_peak(eh_ctx* ctx, uint64_t* peak);

/* Profiling builds (-DEH_PROF) only: prof[2k] = ticks, prof[2k+1] = calls; k < 64 is a mutator
 * id, 64.. are phases (setup, generator, pattern+mutators, output copy).  256 values. */
int eh_result_prof(eh_ctx* ctx, uint64_t* prof);

/* Kernel self-test hook for the wave-level byte movers (tests only): jobs = njobs x
 * {kind (0 copy, 1 periodic fill, 2 equal), dst_off, src_off, n, plen} over the buffer image. */
int eh_selftest_movers(eh_ctx* ctx, uint8_t* buf, uint64_t buf_len, const uint32_t* jobs, uint32_t njobs, uint32_t* eq_out);
/* Self test of the device deflate / inflate behind the cp / ar patterns and the zip mutator (csrc/eh_zlib.h; replaces OTP's zlib
 * binding on this path: erlamsa_patterns.erl:216-246).  op 0 raw deflate stream, 1 zlib:gzip/1, 2 zlib:deflate(Z, Data, finish) after
 * deflateInit(Z, default), 4 zlib:gunzip/1, 5 zlib:inflate/2 without inflateEnd (a stream that just stops yields what was decoded).
 * *ok = 0 where the reference's call raises (data_error, need_dictionary) or `cap` is too small. */
int eh_selftest_zlib(eh_ctx* ctx, int op, const uint8_t* in, uint64_t n, uint8_t* out, uint64_t cap, uint64_t* out_len, int32_t* ok);

/* Host-side self test: erlamsa_utils:sort_by_priority/1 (erlamsa_utils.erl:113-117: lists:sort/2 with a strict '>') as the
 * engine's set-up orders patterns, generators and mutators - perm[k] = index of the k-th entry of the sorted list.  The oracle
 * and tests/pymodel.py carry their own restatements of OTP's merge sort; tests/test_oracle_otp.py diffs the three. */
int eh_selftest_sort_by_priority(const uint32_t* pri, uint32_t n, uint32_t* perm);

/* Work-area pool of this context's device (diagnostic), 64 values; t = tier 1 .. out[40]: out[2t] / out[2t+1] = areas
 * of tier t taken / returned since the pool was made (+ the tier's size for the latter), out[20+t] = shader-clock ticks
 * wavefronts waited for an area of tier t, out[30+t] = how many had to wait, out[40] = tiers, out[41+t] = areas of tier t (low 32 bits) and the most of them that
 * were wanted at the same time, taken or waited for (high 32 bits),
 * out[51+t] = bytes of an area of tier t (out[51] = the slots'), out[61] = contexts sharing the pool, out[62] = slots
 * (= workgroups of a batch) of this context. */
int eh_pool_stats(eh_ctx* ctx, uint64_t* out);
/* Cooperative execution of heavy cases (ABI 8): a case's loops over hundreds of kilobytes (block copies, the final concatenation,
 * the streaming passes of erlamsa_fuse on large lists) are cut into chunks that wavefronts BETWEEN two cases of their own run too.
 * out[0] loops posted, out[1] chunks run by such helpers, out[2] chunks run by the posting cases, out[3] shader cycles the posters
 * waited for the last chunks, out[4] loops that found the board full; out[5..7] reserved.  Counted per device since the pool exists. */
int eh_coop_stats(eh_ctx* ctx, uint64_t* out);

/* Elapsed GPU time of the mutate kernel of the last batch in ms (HIP events on the launch
 * stream), and its name for matching against a rocprofv3 kernel trace. */
int eh_last_kernel_ms(eh_ctx* ctx, float* ms);
const char* eh_kernel_name(void);

/* Introspection */
uint32_t eh_abi_version(void);
int eh_mutator_count(void);
const char* eh_mutator_name(int id);       /* table order of erlamsa_mutations:mutations/1 */
int eh_mutator_default_pri(int id);
int eh_mutator_on_gpu(int id);             /* 1 if this build runs it on the device */
int eh_pattern_count(void);
const char* eh_pattern_name(int id);
int eh_pattern_default_pri(int id);
int eh_pattern_on_gpu(int id);
const char* eh_strerror(int code);
const char* eh_last_error(eh_ctx* ctx);
/* The same text copied into caller memory (NUL-terminated, at most cap - 1 bytes; returns its length): for hosts that call
 * eh_submit / eh_flush / eh_poll from several threads, where the pointer eh_last_error returns may be rewritten meanwhile. */
uint64_t eh_last_error_copy(eh_ctx* ctx, char* buf, uint64_t cap);

#ifdef __cplusplus
}
#endif
#endif
