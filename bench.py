#!/usr/bin/env python3
"""bench.py — headline benchmark: mutated MB/s (+ cases/s) on the BASELINE config-3 corpus.

A "step" = one pass of the hot path (eh_fuzz_batch: generator -> pattern -> mux_fuzzers ->
mutators -> output arena) over the whole 64K x 4 KiB synthetic corpus, with the corpus arena
already resident in HBM.  Step k uses case numbers k*n+1 .. (k+1)*n of the same fuzzer/1 run, so
no step repeats another's work.  Multi-GPU: rank r holds the (RCCL-broadcast) arena and runs its
own contiguous range of case numbers — weak scaling, no data-path collective.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# (passes in flight run on separate HIP streams and want a hardware queue each: the library sets GPU_MAX_HW_QUEUES=8 itself when it is
# loaded - csrc/eh_engine.hip eh_runtime_defaults; only the N > 1 path, where torch starts the HIP runtime first, sets it here)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
DESC_BYTES = 24        # per-case descriptor: out_off, out_len, status/draws


def parity_windows(args, n_rows, passes_run):
    """Which cases the oracle leg runs: `--cpu-sample` S = rows 0..S/2 of THREE passes of the run - the first, one in the middle and
    one near the end of the passes the timed loop ran (0, 7, 19 for the driver's 5 + 20) -, so that the engine build that was timed
    is checked on case numbers from the whole run, not on its first pass only.  A run of fewer than three passes (or a corpus of
    its own: --config 5) takes what there is.  -> [(first_case, row0, rows)]"""
    want = [0, 7, 19] if passes_run >= 20 else sorted({0, passes_run // 3, max(0, passes_run - 1)})
    per = max(1, min(args.cpu_sample // 2 if len(want) > 1 else args.cpu_sample, n_rows))
    return [(p * n_rows + 1, 0, per) for p in want]


def cpu_baseline_leg(mat, seed, muts, pats, args, gpu_ref=None, windows=None):
    """oracle/ timed on the host: the `--cpu-sample` cases of parity_windows() (same corpus rows, options and work-area
    limit as the GPU run), in chunks of 8, handed to `--cpu-threads` worker threads (the ctypes call releases the GIL).
    Every case runs under a wall-clock watchdog of `--cpu-case-seconds` — the reference's own maxrunningtime semantics
    (erlamsa_main.erl:197-204: the worker is killed and the case yields <<>>; its CLI default is 30 s): the time of such a
    case counts, its output does not.  Reported: the aggregate rate over all threads and the threads actually used.
    The same outputs are the CHECKER of the engine build that was timed: gpu_ref holds status / length / draw count / SHA-1 of
    the engine's results for the same case numbers (taken from the device right before this leg); every case the oracle
    finished must agree, or the bench fails.  Engine-only statuses (2, 3) and cases cut by the watchdog are counted, not
    compared."""
    import hashlib
    import threading
    from erlamsa_amd import synth
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    po.lib()
    n = mat.shape[0]
    if windows is None:
        windows = [(1, 0, min(args.cpu_sample, n))]
    # work items: (index into the flat sample, first case number, first row, rows)
    items, flat = [], 0
    CH = 8
    for first_case, row0, rows in windows:
        for a in range(0, rows, CH):
            b = min(a + CH, rows)
            items.append((flat + a, first_case + a, row0 + a, b - a))
        flat += rows
    threads = max(1, args.cpu_threads or min(os.cpu_count() or 1, 64))
    whole = synth.as_arena(mat) if args.generators else None      # the Paths of file / jump: the whole corpus, whatever chunk a call runs
    lock = threading.Lock()
    state = {"next": 0, "cases": 0, "bytes": 0, "timeouts": 0, "checked": 0, "skipped": 0, "bad": []}
    t0 = time.perf_counter()

    def worker():
        while True:
            with lock:
                k = state["next"]
                if k >= len(items):
                    return
                state["next"] = k + 1
            a, case0, r0, cnt = items[k]
            b = a + cnt
            d, o = synth.as_arena(mat[r0:r0 + cnt])
            outs, st, dr, _ = po.fuzz_batch(d, o, seed=seed, mutations=muts, patterns=pats, generators=args.generators, paths=whole, first_case=case0,
                                           max_case_bytes=args.big_mib << 20, max_case_work=args.work_mib << 20,
                                           max_case_seconds=args.cpu_case_seconds)
            with lock:
                state["cases"] += b - a
                state["bytes"] += sum(len(x) for x in outs)
                state["timeouts"] += int((st == 6).sum())
                if gpu_ref is not None:
                    gst, gln, gdr, gsha = gpu_ref
                    for j in range(b - a):
                        i = a + j
                        if st[j] in (2, 3, 6) or gst[i] in (2, 3):
                            state["skipped"] += 1
                        elif int(st[j]) != int(gst[i]) or len(outs[j]) != int(gln[i]) or (st[j] == 0 and int(dr[j]) != int(gdr[i])) \
                                or hashlib.sha1(outs[j]).digest() != gsha[i]:
                            state["bad"].append(case0 + j)
                        else:
                            state["checked"] += 1

    ts = [threading.Thread(target=worker) for _ in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    ct = time.perf_counter() - t0
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), model)
    except OSError:
        pass
    if state["bad"]:
        raise RuntimeError("PARITY FAILURE: the engine's results of cases %s differ from the oracle's" % state["bad"][:16])
    return {"parity_checked": state["checked"], "parity_skipped": state["skipped"],
            "value": round(state["bytes"] / ct / 1e6, 3), "unit": "MB/s", "cores": threads, "kind": "port", "cpu_model": model,
            "host_cpus": os.cpu_count(),
            "cases_per_s": round(state["cases"] / ct, 2), "cases_cut_by_watchdog": state["timeouts"],
            "sample": "%d cases of the same run - %s - (same corpus rows, seed, mutators, patterns, work-area limit), oracle/ C++ restatement, "
                      "%d threads, %.1f s wall; per-case watchdog %.0f s (maxrunningtime semantics: time counted, output <<>>)"
                      % (state["cases"], ", ".join("%d..%d" % (f, f + r - 1) for f, _, r in windows), threads, ct, args.cpu_case_seconds)}


def kernel_source_sha1():
    """SHA-1 over the kernel's sources (erlamsa_amd/csrc/*, sorted by name): ties a profile under profiles/ to the build it is of"""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "erlamsa_amd", "csrc")
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()


def log(msg):
    sys.stderr.write("[bench %.1f] %s\n" % (time.time() % 100000, msg))
    sys.stderr.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cases", type=int, default=65536)
    ap.add_argument("--size", type=int, default=4096)
    ap.add_argument("--mutations", default=None, help="-m syntax; default: the reference's full default table (erlamsa_mutations.erl:1291-1331)")
    ap.add_argument("--patterns", default="od,nd,bu", help="-p syntax; 'default' = the reference's default table (all ten patterns, BASELINE configs[3])")
    ap.add_argument("--corpus", default="mixed", choices=["mixed", "uniform", "counter"],
                    help="mixed = BASELINE configs[2] (default); uniform = random bytes (configs[1] with --cases 1024 --size 256); counter = the "
                    "counter-hash corpus of synth.counter, written into the HBM arena by every rank for itself (no host staging, no broadcast)")
    ap.add_argument("--generators", default=None, help="-g syntax (direct, random, file, jump); default: what paths=[direct] leaves (direct=500,random=1). "
                    "file / jump stream the corpus entries as their Paths on the device (erlamsa_gen.erl:106-150)")
    ap.add_argument("--config", type=int, default=3, choices=[3, 5], help="3 = BASELINE configs[2] (the default, what the driver times); 5 = the shape of "
                    "configs[4]: 131072 x world seeds of 64 KiB from the counter-hash corpus, generator jump (cross-seed splices), mutators "
                    "ft,fn,fo,num,len, pattern sz, strong scaling (every rank holds the whole arena and runs its share of the cases); other flags override")
    ap.add_argument("--cpu-sample", type=int, default=2048, help="cases timed on the CPU oracle: the first N of the run (0 = skip)")
    ap.add_argument("--cpu-case-seconds", type=float, default=10.0, help="per-case wall-clock watchdog of the CPU oracle leg "
                    "(the reference's maxrunningtime; its CLI default is 30 s)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU oracle leg (0 = all host cores, at most 64)")
    ap.add_argument("--max-slots", type=int, default=1024, help="slots = persistent workgroups of one pass (0 = one per wavefront the device holds: "
                    "8 per CU = 2048); passes in flight oversubscribe the device.  1024 x 6 passes keeps ~45 GiB of the device free: with 2048 "
                    "(5 percent faster, profiles/r03_bench.json) the free memory fell below what the runtime wants for the queues' scratch and "
                    "two of six runs never left the set-up passes")
    ap.add_argument("--pool-gib", type=int, default=42, help="device memory of the work-area pool all contexts share (GiB), eh_options.pool_bytes; "
                    "0 = the library's own rule (a quarter of the free memory, at most 64 GiB)")
    ap.add_argument("--out-gib", type=int, default=27, help="output arena capacity per context (GiB)")
    ap.add_argument("--case-mib", type=int, default=4, help="work area of a slot (MiB), eh_options.max_case_bytes; every workgroup of a pass owns a slot, a case that "
                    "outgrows it borrows larger areas from the pool")
    ap.add_argument("--big-mib", type=int, default=1024, help="largest work area (MiB), eh_options.big_case_bytes: a case that outgrows its area "
                    "borrows areas of 2x, 4x, ... from the pool, up to this size")
    ap.add_argument("--budget-mib", type=int, default=64, help="after the headline run (no budget), repeat 3 steps with this per-case work "
                    "budget (eh_options.max_case_work) and report them under 'with_work_budget'; 0 = skip")
    ap.add_argument("--work-mib", type=int, default=0, help="optional per-case work budget (MiB), eh_options.max_case_work; "
                    "0 = off (default): every case runs to completion like under the reference's 30 s CLI watchdog")
    ap.add_argument("--pcie", type=int, default=1, help="1: after the timed steps, one extra pass whose outputs are downloaded to pinned host memory "
                    "(reported as 'pcie'); 0: skip")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak (default): every rank runs its own 64 K case numbers per step; "
                    "strong: a step is ONE run of --cases cases split over the ranks by shard.case_range (erlamsa_main.erl:95-108)")
    ap.add_argument("--extras-seconds", type=int, default=240, help="watchdog of the legs that run after the headline result exists (CPU oracle leg, "
                    "work-budget leg, PCIe leg): the JSON line is printed without a leg that has not come back by then")
    ap.add_argument("--engine-flags", type=int, default=0, help="eh_options.flags of every context (diagnostic: 64 = EH_FLAG_NO_COOP, every case does all "
                    "of its work on its own wavefront)")
    ap.add_argument("--inflight", type=int, default=7, help="passes in flight (engine contexts / HIP streams): the tail of a pass — a few "
                    "long single-wavefront cases — overlaps with the bulk of the next ones.  Every context owns its output arena (--out-gib) and its slots "
                    "(--max-slots x --case-mib); larger work areas come from one pool shared by all contexts (--pool-gib)")
    ap.add_argument("--setup-seconds", type=int, default=150, help="single-GPU runs execute in a child process; a child whose set-up passes (the first "
                    "dispatch on every HIP stream) have not finished after this many seconds is killed and the run repeated with "
                    "--inflight 3, then 1 (0 = no supervision)")
    ap.add_argument("--strong-leg", type=int, default=1, help="N > 1 with weak scaling: after the timed steps, a few steps of STRONG scaling (one run of --cases cases "
                    "split over the ranks by shard.case_range, what north_star's corpus sharding is) reported under 'strong_scaling_leg'; 0 = skip")
    ap.add_argument("--case-stats", type=int, default=1, help="1: collect per-case wave cycles and output lengths of the timed steps (one small D2H copy per "
                    "collected pass, outside no kernel's way) and report them under 'case_stats': how the bytes and the cycles are distributed over the cases")
    ap.add_argument("--profiled", type=int, default=0, help="1: the run is under rocprofv3 - no child process (--setup-seconds 0) and the process ends by "
                    "returning from main() instead of os._exit, so that the profiler's tool gets to write its output")
    pre, _ = ap.parse_known_args()
    if pre.config == 5:
        w5 = int(os.environ.get("WORLD_SIZE", "1"))
        ap.set_defaults(cases=131072 * w5, size=65536, corpus="counter", generators="jump", mutations="ft,fn,fo,num,len", patterns="sz", scaling="strong",
                        out_gib=8, cpu_sample=1024, budget_mib=0)
    args = ap.parse_args()
    if args.profiled:
        args.setup_seconds = 0

    # ---- supervision (single GPU only): the run proper happens in a child process, and the driver must get its JSON line.
    # A child that dies (round 3's driver run: "Memory access fault by GPU node-2" 3.6 s in, once, never reproduced) or that never
    # leaves its set-up passes is replaced: first by an identical one, then by more frugal ones (fewer passes in flight).
    if int(os.environ.get("WORLD_SIZE", "1")) == 1 and args.setup_seconds > 0 and not os.environ.get("EH_BENCH_CHILD"):
        import subprocess
        import threading
        # (seven passes in flight take 261 of the device's 288 GiB: a child that cannot get them is followed by round 5's six with the larger pool)
        ladder = ([], ["--inflight", "6", "--pool-gib", "60"], ["--inflight", "3", "--pool-gib", "60"], ["--inflight", "1", "--pool-gib", "60"])
        for attempt, extra in enumerate(ladder):
            env = dict(os.environ, EH_BENCH_CHILD="1")
            proc = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            marks = {"setup": None, "past": False, "stage": "start"}

            def pump(proc=proc, marks=marks):
                for ln in proc.stderr:
                    sys.stderr.write(ln)
                    sys.stderr.flush()
                    if ln.startswith("[bench "):
                        marks["stage"] = ln.split("]", 1)[1].strip()[:80]
                    if "reserve + set-up passes" in ln:
                        marks["setup"] = time.time()
                    if "warm-up steps" in ln:
                        marks["past"] = True
            th = threading.Thread(target=pump, daemon=True)
            th.start()
            stuck = False
            while proc.poll() is None:
                time.sleep(0.5)
                if marks["setup"] is not None and not marks["past"] and time.time() - marks["setup"] > args.setup_seconds:
                    stuck = True
                    proc.kill()
                    break
            out = proc.stdout.read()
            proc.wait()
            th.join(timeout=5)
            lines = [ln for ln in out.splitlines() if ln.startswith("{")]
            if lines and not stuck:                                     # a result is a result, whatever the child's exit code
                print(lines[-1], flush=True)
                sys.exit(0)
            nxt = ", repeating%s" % (" with " + " ".join(ladder[attempt + 1]) if ladder[attempt + 1] else " as configured") if attempt + 1 < len(ladder) else ""
            if stuck:
                log("set-up passes did not finish within %d s: child killed%s" % (args.setup_seconds, nxt))
            else:
                sys.stderr.write(out[-2000:])
                log("child ended without a result (exit code %s, last stage: %s)%s" % (proc.returncode, marks["stage"], nxt))
        sys.exit(1)

    if os.environ.get("EH_BENCH_SIMULATE"):                  # tests/test_bench_supervisor.py: a child that hangs in its set-up passes, or not
        sim = os.environ["EH_BENCH_SIMULATE"]
        if sim == "crash" or (sim == "crash_once" and not os.path.exists(os.environ["EH_BENCH_SIMULATE_FLAG"])):
            if sim == "crash_once":
                open(os.environ["EH_BENCH_SIMULATE_FLAG"], "w").close()
            log("corpus (simulated)")
            os.abort()                                           # what a GPU memory access fault does to the process
        log("reserve + set-up passes (simulated)")
        if sim == "hang" or (sim == "hang_once" and args.inflight != 3):
            time.sleep(3600)
        log("warm-up steps")
        print(json.dumps({"metric": "simulated", "inflight": args.inflight}), flush=True)
        return

    import numpy as np
    from erlamsa_amd import shard, synth
    rank, world, local = shard.rank_env()
    # A single-GPU run needs no torch at all: the corpus goes up through eh_corpus_upload, every context runs on its own HIP
    # stream (eh_stream), pinned host memory comes from eh_host_alloc.  torch is imported only for N > 1 - it is the RCCL binding
    # (broadcast of the arena, barrier, reduction of the result) - and then BEFORE the engine's library, so that both share one
    # HIP runtime.
    dist = torch = None
    # EH_BENCH_BACKEND=gloo: this very script's N > 1 path on CPU ranks with the emulator build of the engine (tests/test_bench_dry.py;
    # device pointers are host pointers there) - everything but RCCL itself
    on_gpu = os.environ.get("EH_BENCH_BACKEND", "nccl") == "nccl"
    if world > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")          # torch initialises the HIP runtime before the engine's library is loaded
        import torch
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        log("rank %d: process group (%s)" % (rank, "nccl = RCCL" if on_gpu else "gloo"))
        if on_gpu:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    import erlamsa_amd as ea
    from erlamsa_amd.engine import HostBuffer
    dev = None if torch is None else torch.device("cuda", local) if on_gpu else torch.device("cpu")

    n, size = args.cases, args.size
    # Measured set = the reference's full default mutator table (41 entries, default priorities) unless overridden.
    muts = args.mutations or ",".join("%s=%d" % (m, p) for m, p, _ in ea.mutator_table())
    pats = None if args.patterns == "default" else args.patterns     # "default": the reference's own table (erlamsa_patterns.erl:395-405), all ten at their priorities
    nmut_total = len(ea.mutator_table())

    # `--inflight` engine contexts, each on its own HIP stream: step k runs on context k % inflight, so
    # the long tail of one pass (a few MB-sized cases handled by single wavefronts) overlaps with
    # the next pass instead of idling the GPU.  Every step is still a complete, separate pass.
    nctx = max(1, min(args.inflight, args.steps))
    log("engine contexts (%d)" % nctx)
    engines = []
    for _ in range(nctx):
        e = ea.Engine(local if on_gpu else 0)
        e.configure(mutations=muts, patterns=pats, generators=args.generators, max_slots=args.max_slots, out_capacity=args.out_gib << 30,
                    max_case_bytes=args.case_mib << 20, max_case_work=args.work_mib << 20, big_case_bytes=args.big_mib << 20,
                    pool_bytes=args.pool_gib << 30, flags=args.engine_flags)
        engines.append(e)

    def sync():
        for e in engines:
            e.sync()
        if torch is not None and on_gpu:
            torch.cuda.synchronize()

    # ---- corpus: one arena in HBM, shared by all contexts of the device
    log("corpus")
    mat = None
    if world == 1:
        if args.corpus == "counter":
            mat = np.concatenate([synth.counter(range(r0, min(r0 + 4096, n)), size) for r0 in range(0, n, 4096)])
        else:
            mat = synth.mixed(n, size) if args.corpus == "mixed" else synth.uniform(n, size)
        engines[0].upload_corpus(*synth.as_arena(mat))
        if n * size > (1 << 30):
            mat = None                                              # (the CPU oracle leg wants a host copy: small runs only)
    else:
        # The arena reaches every GPU through the LIBRARY's own RCCL calls (include/erlamsa_hip.h "multi-GPU", csrc/eh_comm.h) - the path a
        # host without a HIP / RCCL binding (the BEAM) takes; torch.distributed only carries the 128 bytes of the unique id, the barriers
        # and the reduction of the result.  EH_BENCH_BACKEND=gloo: the same calls with tests/hipemu/fake_rccl.cpp behind EH_RCCL_LIB.
        def agree(ok):                                             # every rank takes the same path: all succeeded, or all fall back
            flags = [None] * world
            dist.all_gather_object(flags, bool(ok))
            return all(flags)
        transport = "library"
        try:
            uid = [ea.Engine.comm_unique_id() if rank == 0 else None]
            ok = True
        except ea.EngineError as ex:
            log("rank %d: %s" % (rank, ex)); uid, ok = [None], False
        if agree(ok):
            dist.broadcast_object_list(uid, src=0)
            try:
                engines[0].comm_init(uid[0], rank, world)
                ok = True
            except ea.EngineError as ex:
                log("rank %d: eh_comm_init: %s" % (rank, ex)); ok = False
            if not agree(ok):
                transport = "torch"
        else:
            transport = "torch"
        if transport == "library":
            try:
                if args.corpus == "counter":
                    # BASELINE configs[4]: every rank brings ITS shard (closed form of seed, row, byte - no file to read), all-gathered in place:
                    # each xGMI link carries 1 / world of the arena instead of one root feeding everybody
                    per = n // world
                    assert per * world == n, "--cases must be a multiple of the number of ranks for the all-gathered counter corpus"
                    shard_rows = np.concatenate([synth.counter(range(r0, min(r0 + 4096, (rank + 1) * per)), size) for r0 in range(rank * per, (rank + 1) * per, 4096)])
                    engines[0].corpus_allgather(*synth.as_arena(shard_rows))
                    del shard_rows
                else:
                    if rank == 0:
                        mat = synth.mixed(n, size) if args.corpus == "mixed" else synth.uniform(n, size)
                        engines[0].corpus_broadcast(0, *synth.as_arena(mat))
                    else:
                        engines[0].corpus_broadcast(0)
                ok = engines[0].n_corpus == n
            except ea.EngineError as ex:
                log("rank %d: the library's collective failed: %s" % (rank, ex)); ok = False
            if not agree(ok):
                transport = "torch"
        if transport == "torch":
            # fall-back (tests/test_bench_dry.py runs it as two CPU ranks, "weak-without-rccl"): the arena as a torch tensor, broadcast by
            # torch.distributed, attached to the context
            log("rank %d: the library's RCCL path is not available here - arena over torch.distributed" % rank)
            import torch
            arena = torch.empty(n * size, dtype=torch.uint8, device=dev)
            offs = torch.arange(n + 1, dtype=torch.int64, device=dev) * size
            if args.corpus == "counter":
                synth.counter_torch(arena, 0, n, size)
            else:
                if rank == 0:
                    mat = synth.mixed(n, size) if args.corpus == "mixed" else synth.uniform(n, size)
                    arena.copy_(torch.from_numpy(mat.reshape(-1)))
                shard.broadcast_corpus(arena, offs, src=0)
            if on_gpu:
                torch.cuda.synchronize()
            engines[0].attach_corpus(arena.data_ptr(), offs.data_ptr(), n, n * size)
            keep_alive = (arena, offs)
        assert engines[0].n_corpus == n
    for e in engines[1:]:
        e.share_corpus(engines[0])
    raw = [e.own_stream() for e in engines]
    seed = (1, 2, 3)
    # Context set-up, not a step: every context reserves its device memory for a full batch (eh_reserve) and
    # runs one untimed full-size pass on its own HIP stream, which makes the runtime allocate that queue's
    # scratch.  Otherwise the first pass of a context (its hipMallocs plus the queue's scratch) would land
    # inside the timed region whenever W is smaller than the number of contexts.
    log("reserve + set-up passes (%d contexts)" % nctx)
    for e in engines:
        e.reserve(n)
    # one context at a time: the first dispatch on a HIP stream makes the runtime allocate that hardware queue's scratch
    # (1.8 KiB of stack per lane for every wavefront slot of the device; 8 KiB while the kernel still recursed), which must not have to wait
    # for memory or wavefront slots that the persistent workgroups of five other passes are holding
    for k, (e, st) in enumerate(zip(engines, raw)):
        e.fuzz_batch(seed=seed, first_case=1, corpus_first=0, n=n, stream=st)
        e.sync()
        log("set-up pass %d of %d done" % (k + 1, nctx))
    arena_same = None
    if world > 1:
        # every rank must hold the bytes the others hold: the first 512 case numbers, which mutate rows 0..511 and - with file / jump -
        # rows anywhere in the arena, are run on EVERY rank and the digests compared (a collective that silently moved nothing, or
        # shards in another order, would show here)
        import hashlib
        e0 = engines[0]
        e0.fuzz_batch(seed=seed, first_case=1, corpus_first=0, n=min(512, n), stream=raw[0])
        e0.sync()
        ln0 = e0.lens()
        dig = hashlib.sha1(b"".join(hashlib.sha1(e0.fetch(i, int(ln0[i]))).digest() for i in range(len(ln0)))).digest()
        allv = [None] * world
        dist.all_gather_object(allv, dig)
        arena_same = all(v == allv[0] for v in allv)
        if not arena_same:
            raise RuntimeError("rank %d: the arena differs between ranks (results of the same case numbers differ)" % rank)
    log("warm-up steps")
    # rank r, step k -> case numbers ((k*world + r) * n) + 1 ... (shard.run_steps, the loop tests/test_dist_gloo.py drives too)
    strong = args.scaling == "strong"
    overflow_sites = {}
    names = [m for m, _, _ in ea.mutator_table()]

    cyc_pass, len_hist, occ_pass = [], [], []

    def on_result(k, e):                                      # which capacity check gave up, for the cases that end as EH_CASE_OVERFLOW
        st = e.status()
        occ_pass.append(e.occupancy())                          # (page-locked memory the kernel wrote: no copy call)
        if args.case_stats:
            cy = e.cycles()
            cy = np.where(cy > (1 << 60), 0, cy)                # (a case whose end stamp read lower than its start stamp: s_memtime is per-XCD and a wrapped difference is not a duration)
            cyc_pass.append((int(cy.sum()), int(cy.max()), int(cy.argmax()) + 1 + k * n))
            len_hist.append(np.sort(e.lens()))
        if (st == 2).any():
            _, lm = e.diag()
            for site in (-lm[st == 2]).tolist():
                overflow_sites[site] = overflow_sites.get(site, 0) + 1

    shard.run_steps(engines, raw, 0, args.warmup, rank, world, n, seed, strong=strong)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    log("timed steps")
    t0 = time.perf_counter()
    timed = shard.run_steps(engines, raw, args.warmup, args.steps, rank, world, n, seed, on_result=on_result, strong=strong)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    log("timed steps done: %.3f s per step" % (dt / max(args.steps, 1)))
    out_bytes, kern_ms, status_counts = timed["out_bytes"], timed["kernel_ms"], timed["status_counts"]

    my_cases = (shard.case_range(n, rank, world)[1] if strong else n) * args.steps
    dt_all, out_all, cases_all = shard.reduce_over_ranks(dt, out_bytes, my_cases, dist, dev)
    strong_leg = None
    if world > 1 and not strong and args.strong_leg and args.steps > 0:
        # the same corpus, strong scaling: ONE run of n cases per step, rank r takes case_range(n, r, world) of it (erlamsa_main.erl:95-108).
        # A rank's pass is 1 / world of a weak one, so world x as many passes go through a GPU per second and the leg needs world x as
        # many steps as the weak run before ramp and drain of the pipeline stop showing: 4 x contexts x world steps (at most 200),
        # whatever --steps says.  `value` stays the weak figure (the contract's definition for a path that shards into independent
        # units, and what a user with more cases than one pass runs); this is the labelled second.
        ks = max(2, min(4 * nctx * world, 200))
        sync(); dist.barrier(); sync()
        ts = time.perf_counter()
        sr = shard.run_steps(engines, raw, args.warmup + args.steps + 1000, ks, rank, world, n, seed, strong=True)
        sync(); dist.barrier(); sync()
        sdt = time.perf_counter() - ts
        sdt_all, sout_all, scases_all = shard.reduce_over_ranks(sdt, sr["out_bytes"], shard.case_range(n, rank, world)[1] * ks, dist, dev)
        strong_leg = {"steps": ks, "value": round(sout_all / sdt_all / 1e6, 1), "unit": "MB/s", "cases_per_s": round(scases_all / sdt_all, 1),
                      "ms_per_step": round(sdt_all / ks * 1e3, 3), "cases_per_step_all_ranks": n,
                      "what": "one run of %d cases per step split over %d ranks by contiguous case ranges; results are those of a 1-rank run (tests/test_comm_abi.py, tests/test_dist_gloo.py)" % (n, world)}

    if rank == 0:
        mbps = out_all / dt_all / 1e6
        nloc = shard.case_range(n, rank, world)[1] if strong else n
        in_bytes = float(nloc * size)
        avg_kern_s = float(np.mean(kern_ms)) / 1e3
        alg_bytes = in_bytes + out_bytes / args.steps + DESC_BYTES * nloc    # per launch (this rank)
        achieved = alg_bytes / avg_kern_s / 1e9
        # HBM-side bytes per launch cannot be measured from inside this process (PMC counters need rocprofv3 around it): the figure
        # of the newest committed counter run (profiles/rNN_summary.json, tools/profile_round.sh + collect_profiles.py) is attached
        # when it is a run of THIS kernel (same sources) on THIS workload with THESE engine options; otherwise traffic is null
        # and the file is only named.
        traffic, traffic_src = None, None
        ksha = kernel_source_sha1()
        try:
            import glob
            cand = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_summary.json")))
            with open(cand[-1]) as fh:
                ps = json.load(fh)
            wk = ps["workload_key"]
            mine = {"cases": n, "size": size, "max_case_work": args.work_mib << 20, "max_case_bytes": args.case_mib << 20, "mutators": muts, "patterns": pats,
                    "max_slots": args.max_slots, "pool_gib": args.pool_gib, "big_case_bytes": args.big_mib << 20, "kernel_source_sha1": ksha}
            if all(wk.get(k) == v for k, v in mine.items()):
                traffic = int(ps["traffic_bytes_per_launch"]["total_fetch_x2_plus_write"])
                traffic_src = "%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, per launch; a committed counter run of this kernel and workload, not measured in this process)" % os.path.relpath(cand[-1], ROOT)
            else:
                traffic_src = "%s is a counter run of another kernel build or configuration (%s): not attached" % (
                    os.path.relpath(cand[-1], ROOT), ", ".join(k for k, v in mine.items() if wk.get(k) != v))
        except (OSError, KeyError, ValueError, IndexError):
            pass
        res = {
            "metric": "mutated_MB_per_s", "value": round(mbps, 1), "unit": "MB/s",
            "cases_per_s": round(cases_all / dt_all, 1),
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt_all / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8 (byte edits) + f64 (AS183 draws)", "data": "synthetic",
            "config": {
                "workload": "%s: %d seeds x %d B %s, generator %s, patterns %s, "
                            "mutators %s (%d of the %d of the default table; not measured: %s)"
                            % ("BASELINE configs[2]" if (args.corpus, n, size, args.generators) == ("mixed", 65536, 4096, None) else
                               "the shape of BASELINE configs[4]" if args.config == 5 else "custom",
                               n, size, "mixed-binary corpus (50% random, 25% ASCII lines+numbers, 15% bracketed text, "
                               "10% length/CRC-framed)" if args.corpus == "mixed" else "uniform random bytes" if args.corpus == "uniform" else
                               "counter-hash corpus (uniform / text with numbers / length-framed / low-entropy rows, written on the device)",
                               args.generators or "direct=500/random=1", pats or "the default table (od,nd,bu,sk,sz,cs,ar,cp,co,nu at their default priorities)", muts, len(muts.split(",")), nmut_total,
                               ",".join(m for m, _, _ in ea.mutator_table() if m not in [x.split("=")[0] for x in muts.split(",")]) or "none"),
                "world_size_seen_by_torch_distributed": (dist.get_world_size() if dist is not None else 1),
                "arena_equal_on_all_ranks": arena_same,
                "arena_transport": (None if world == 1 else "torch.distributed broadcast + eh_corpus_attach (fall-back: the library's RCCL path failed here, see stderr)" if transport == "torch" else
                                    "eh_corpus_allgather (RCCL inside the library, per-rank shards)" if args.corpus == "counter" else "eh_corpus_broadcast (RCCL inside the library, root = rank 0)"),
                "seed": list(seed), "cases_per_step_per_gpu": n, "parallelism": "case-range sharding x%d, no collective on the mutation path" % world, "passes_in_flight": nctx, "context_setup": "eh_reserve + one untimed full-size pass per context/stream, one after the other, before the W warm-up steps",
                "max_case_bytes": args.case_mib << 20, "big_case_bytes": args.big_mib << 20, "max_case_work": args.work_mib << 20,
                "workgroups_per_pass": args.max_slots or "one per wavefront the device holds (8 per CU)", "pool_gib": args.pool_gib,
                "max_slots": args.max_slots, "engine_flags": args.engine_flags, "kernel_source_sha1": ksha,
                "host": "no torch in this process (corpus: eh_corpus_upload, streams: eh_stream, pinned memory: eh_host_alloc)" if world == 1 else "torch = the RCCL binding (arena broadcast, barrier, reduction)",
                "work_area_pool": engines[0].pool_stats(),
                "cooperative_execution": engines[0].coop_stats(),
            },
            "case_status": dict(zip(["ok", "crashed(reference worker dies)", "overflow(big_case_bytes)", "unsupported", "arena_full",
                                     "budget(max_case_work; reference analogue: maxrunningtime -> <<>>)"],
                                    [int(x) for x in status_counts])),
            "overflow_by_capacity_check": {
                "what": "the EH_CASE_OVERFLOW cases of the timed steps by the check that gave up (EH_SET_OVERFLOW sites in csrc/): 802 = a tree "
                        "stutter (mutator tr) whose result k^reps x |node| exceeds big_case_bytes — the reference builds the same binary until "
                        "its 256 MB process guard truncates it (erlamsa_mutations.erl:978-984); 102 = the largest work area is exhausted; "
                        "103 / 803 = a single result of 4 GiB or more (sr / tree)",
                "counts": {str(k): v for k, v in sorted(overflow_sites.items())}},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": ea.load_library().eh_kernel_name().decode(), "kernel_ms_avg": round(avg_kern_s * 1e3, 3),
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "launch": "one eh_fuzz_batch = 1 dispatch of the kernel; kernel_ms_avg from HIP events on the launch stream.  %d passes are "
                                   "in flight: a dispatch shares the device with the others for its whole duration, so the per-launch rate is "
                                   "about 1/%d of the device's" % (nctx, nctx),
                         "achieved_all_in_flight": round(alg_bytes / (dt_all / args.steps) / 1e9, 2)},
        }
        if strong_leg is not None:
            res["strong_scaling_leg"] = strong_leg
        # the host thread of the timed loop, per step: collecting a finished pass (totals, statuses: D2H copies), the case statistics,
        # eh_fuzz_batch itself, and waiting with every context busy
        res["host_loop_ms_per_step"] = {k: round(v * 1000.0 / (1 if k.endswith("_max") else max(args.steps, 1)), 2) for k, v in timed["host_loop_s"].items()}
        if args.case_stats and len_hist:
            # How the headline is made: the default table pumps (sr, lr, tr, sgm, fuse repeat data), so a small share
            # of the cases carries most of the bytes and of the wave cycles.
            al = np.concatenate(len_hist).astype(np.float64)
            al.sort()
            tot = float(al.sum()) or 1.0
            top1 = al[int(len(al) * 0.99):]
            small = al[al <= 65536]
            res["case_stats"] = {
                "what": "all %d cases of the timed steps" % len(al),
                "output_bytes_percentiles": {"p50": int(al[len(al) // 2]), "p90": int(al[int(len(al) * 0.9)]), "p99": int(al[int(len(al) * 0.99)]),
                                             "p99.9": int(al[int(len(al) * 0.999)]), "max": int(al[-1]), "mean": round(tot / len(al), 1)},
                "share_of_bytes_from_top_1pct_of_cases": round(float(top1.sum()) / tot, 4),
                "share_of_bytes_from_top_10pct_of_cases": round(float(al[int(len(al) * 0.9):].sum()) / tot, 4),
                "cases_with_output_le_64KiB": {"share_of_cases": round(len(small) / len(al), 4), "share_of_bytes": round(float(small.sum()) / tot, 4),
                                               "cases_per_s": round(len(small) / dt_all, 1), "MB_per_s": round(float(small.sum()) / dt_all / 1e6, 1)},
                "wave_cycles_per_pass": {"mean_sum_G": round(float(np.mean([c[0] for c in cyc_pass])) / 1e9, 1),
                                         "heaviest_case_Mcyc_mean_over_passes": round(float(np.mean([c[1] for c in cyc_pass])) / 1e6, 1),
                                         "heaviest_case_Mcyc_max": round(max(c[1] for c in cyc_pass) / 1e6, 1),
                                         "heaviest_case_number": max(cyc_pass, key=lambda c: c[1])[2],
                                         "note": "s_memtime cycles of the wavefront that ran the case, measured WITH the other passes in flight"},
            }
            # the stated second metric: cases per second of the cases whose output is at most 64 KiB - what a fuzzing user
            # feels, and what the pump cases that make the MB/s headline do not move.  Same timed region, same cases.
            res["cases_per_s_le_64KiB"] = res["case_stats"]["cases_with_output_le_64KiB"]["cases_per_s"]
        if occ_pass and dt > 0:
            # where the device's wave slots were during the timed steps (eh_result_occupancy): 100 MHz ticks of the passes' workgroups over
            # wall time x slots.  `held` = a workgroup sat in the slot; `in_cases` = it ran a case; the rest of `held` is the stay for other
            # cases' posted loops, ticket / set-up code and waiting for a work area; 1 - held = slots nobody held
            slots = occ_pass[0][4]; cap = dt * 1e8 * slots
            res["wave_slots"] = {"slots": slots, "held": round(sum(o[0] for o in occ_pass) / cap, 4), "in_cases": round(sum(o[1] for o in occ_pass) / cap, 4),
                                 "lingering_for_posted_loops": round(sum(o[2] for o in occ_pass) / cap, 4),
                                 "what": "share of (wall time of the timed steps x wave slots of the device); the passes in flight at the start of the timed steps are those of the warm-up (not counted), the last timed ones run to their end inside it"}
        if int(status_counts[4]) > 0:                             # EH_CASE_ARENA_FULL inside the timed steps: those cases' outputs are missing from `value`
            res["warning"] = "%d cases of the timed steps did not fit their pass's output arena (%d GiB): raise --out-gib; value counts the bytes that were produced" % (int(status_counts[4]), args.out_gib)
            log(res["warning"])
        # ---- legs that are not the headline: they run AFTER the result exists, under a watchdog that prints the line without them
        # if one of them does not come back (a hung leg must not cost the measured result)
        def leg_pcie():
            # PCIe-inclusive leg: one more pass whose outputs are also brought to host memory, case-ordered, through the
            # boundary call a host-side consumer uses (eh_result_download into pinned memory)
            cap = int(out_bytes / args.steps * 1.5) + (1 << 30)
            try:
                hbuf = HostBuffer(cap)
                kind = "pinned"
            except ea.EngineError:
                hbuf = np.empty(cap, dtype=np.uint8)
                kind = "pageable"
            e = engines[0]
            kx = args.warmup + args.steps + 100
            sync()
            tp = time.perf_counter()
            e.fuzz_batch(seed=seed, first_case=shard.weak_first_case(kx, rank, world, n), corpus_first=0, n=n, stream=raw[0])
            e.sync()
            tk = time.perf_counter()
            try:
                off, _ = e.download_into(hbuf.ptr if kind == "pinned" else hbuf.ctypes.data, cap)
                td = time.perf_counter()
                ob = int(off[-1])
                r = {"pcie_inclusive_MBps": round(ob / (td - tp) / 1e6, 1), "download_GBps": round(ob / (td - tk) / 1e9, 2), "out_bytes": ob,
                     "host_buffer": kind, "pass_s": round(tk - tp, 3), "download_s": round(td - tk, 3),
                     "note": "one pass + eh_result_download (device gather into case order, 2 bounce buffers, D2H overlapped), not overlapped with the next pass"}
                if nctx >= 2:
                    # ... and pipelined, the way a consumer would run it: while pass k goes over PCIe, pass k+1 runs on the device (two contexts)
                    hp = hbuf.ptr if kind == "pinned" else hbuf.ctypes.data
                    sync()
                    tq = time.perf_counter()
                    npipe, tot = 4, 0
                    engines[0].fuzz_batch(seed=seed, first_case=shard.weak_first_case(kx + 1, rank, world, n), corpus_first=0, n=n, stream=raw[0])
                    for j in range(npipe):
                        cur, nxt = engines[j % 2], engines[(j + 1) % 2]
                        if j + 1 < npipe:
                            nxt.fuzz_batch(seed=seed, first_case=shard.weak_first_case(kx + 2 + j, rank, world, n), corpus_first=0, n=n, stream=raw[(j + 1) % 2])
                        cur.sync()
                        o2, _ = cur.download_into(hp, cap)
                        tot += int(o2[-1])
                    tq = time.perf_counter() - tq
                    r["pipelined"] = {"passes": npipe, "pcie_inclusive_MBps": round(tot / tq / 1e6, 1), "s_per_pass": round(tq / npipe, 3),
                                      "note": "pass k+1 on the device while pass k is downloaded: bound by the slower of the two"}
                return r
            except ea.EngineError as ex:
                return {"error": str(ex)}

        def leg_budget():
            # second, labelled figure: the same steps under a per-case work budget (the round-1 configuration)
            for e in engines:
                e.configure(mutations=muts, patterns=pats, generators=args.generators, max_slots=args.max_slots, out_capacity=args.out_gib << 30,
                            max_case_bytes=args.case_mib << 20, max_case_work=args.budget_mib << 20, big_case_bytes=args.big_mib << 20,
                            pool_bytes=args.pool_gib << 30, flags=args.engine_flags)
            bsteps = min(2 * nctx, args.steps)
            sync()
            tb = time.perf_counter()
            br = shard.run_steps(engines, raw, args.warmup + args.steps, bsteps, rank, world, n, seed)
            sync()
            bdt = time.perf_counter() - tb
            bbytes, bstat = br["out_bytes"], br["status_counts"]
            return {"max_case_work": args.budget_mib << 20, "steps": bsteps, "value": round(bbytes / bdt / 1e6, 1), "unit": "MB/s",
                    "cases_per_s": round(n * bsteps / bdt, 1), "ms_per_step": round(bdt / bsteps * 1e3, 3),
                    "case_status_budget": int(bstat[5]), "case_status_overflow": int(bstat[2])}

        import threading
        state = {"leg": "", "done": False}
        lock = threading.Lock()

        def emit():
            with lock:
                if not state["done"]:
                    state["done"] = True
                    print(json.dumps(res), flush=True)

        def watchdog():
            deadline = time.time() + args.extras_seconds
            while time.time() < deadline:
                time.sleep(0.5)
                if state["done"]:
                    return
            res["extras_timed_out"] = "leg '%s' did not finish within %d s; the result above does not depend on it" % (state["leg"], args.extras_seconds)
            emit()
            if not args.profiled:
                os._exit(0)

        if world == 1:
            threading.Thread(target=watchdog, daemon=True).start()
            # the CPU oracle leg first: it is also the parity check of this very run
            reduced = None
            if args.cpu_sample > 0 and mat is None and args.corpus == "counter":
                # The arena is too large for a host copy (configs[4]: 8 GiB and more per GPU), and the oracle needs the Paths the cases draw
                # from: the parity sample runs the SAME kernel and options over the first rows of the arena as a corpus of their own
                # (the counter-hash rows are a closed form of their number) - same seed size, same generators, fewer paths.
                reduced = min(n, 4096)
                mat = np.concatenate([synth.counter(range(r0, min(r0 + 1024, reduced)), size) for r0 in range(0, reduced, 1024)])
                ep = ea.Engine(local if on_gpu else 0)
                ep.configure(mutations=muts, patterns=pats, generators=args.generators, max_slots=args.max_slots, out_capacity=2 << 30,
                             max_case_bytes=args.case_mib << 20, max_case_work=args.work_mib << 20, big_case_bytes=args.big_mib << 20, pool_bytes=args.pool_gib << 30, flags=args.engine_flags)
                ep.upload_corpus(*synth.as_arena(mat))
            if args.cpu_sample > 0 and mat is not None:
                state["leg"] = "cpu_baseline"
                import hashlib
                e0 = engines[0] if reduced is None else ep
                wins = parity_windows(args, mat.shape[0], (args.warmup + args.steps) if reduced is None else 1)
                log("parity sample: cases %s once more on context 0, alone on the device" % ", ".join("%d..%d" % (f, f + r - 1) for f, _, r in wins))
                st0, ln0, dr0, sha0 = [], [], [], []
                for first_case, row0, rows in wins:
                    e0.fuzz_batch(seed=seed, first_case=first_case, corpus_first=row0, n=rows, stream=raw[0] if reduced is None else 0)
                    e0.sync()
                    lw = e0.lens()[:rows].copy()
                    st0.append(e0.status()[:rows].copy()); ln0.append(lw); dr0.append(e0.diag()[0][:rows].copy())
                    sha0 += [hashlib.sha1(e0.fetch(i, int(lw[i]))).digest() for i in range(rows)]
                gpu_ref = (np.concatenate(st0), np.concatenate(ln0), np.concatenate(dr0), sha0)
                log("CPU oracle leg + parity check")
                cb = cpu_baseline_leg(mat, seed, muts, pats, args, gpu_ref, wins)
                res["parity_checked"] = cb.pop("parity_checked")
                res["parity"] = {"checked_bit_exact": res["parity_checked"], "not_compared": cb.pop("parity_skipped"),
                                 "corpus": "the run's own arena" if reduced is None else "rows 0..%d of the arena as a corpus of their own (same kernel, options, seed size and generators; the oracle needs the Paths on the host)" % (reduced - 1),
                                 "cases": [[f, f + r - 1] for f, _, r in wins],
                                 "what": "case numbers from three passes of this run (the first, one in the middle, one near the end of the timed steps), run once more on context 0 after the timed steps: status, length, "
                                         "PRNG draw count and SHA-1 of every output vs the oracle's; not compared = engine-only status (2, 3) or cut by the oracle leg's watchdog"}
                res["cpu_baseline"] = cb
            if args.pcie and args.steps > 0:
                state["leg"] = "pcie"
                log("PCIe leg")
                res["pcie"] = leg_pcie()
            if args.budget_mib > 0 and args.work_mib == 0:
                state["leg"] = "with_work_budget"
                log("work-budget leg")
                res["with_work_budget"] = leg_budget()
        emit()
        sys.stdout.flush()
        if world == 1 and not args.profiled:
            os._exit(0)                                              # (tearing six contexts down takes seconds the driver's clock would count)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
