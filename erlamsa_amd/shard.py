"""Case-range sharding for multi-GPU runs.

A case's output is a pure function of (parent seed, case number, its input, config), so ranks
never exchange data on the mutation path.  Rank r of W takes a contiguous range of case numbers,
the same shape as erlamsa_main:get_threading_mode/3 (reference src/erlamsa_main.erl:95-108) gives
its `--workers`, except that results do not depend on W (parity is defined against workers=1).
"""


def case_range(n_cases, rank, world):
    """Contiguous split of cases 0..n-1: the first n % world ranks take one extra case.
    Returns (first, count)."""
    base, rem = divmod(n_cases, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def weak_first_case(step, rank, world, cases_per_step):
    """bench.py's weak-scaling numbering: every (step, rank) pair gets a fresh block of case
    numbers of the same fuzzer/1 run; 1-based first case number."""
    return (step * world + rank) * cases_per_step + 1


def broadcast_corpus(arena, offsets, src=0):
    """RCCL (or gloo) broadcast of the packed seed arena + offset table from `src`."""
    import torch.distributed as dist
    dist.broadcast(arena, src=src)
    dist.broadcast(offsets, src=src)
    return arena, offsets


def rank_env(env=None):
    """(rank, world, local_rank) from the torch.distributed.run environment (1 process per GPU)."""
    import os
    env = os.environ if env is None else env
    return int(env.get("RANK", "0")), int(env.get("WORLD_SIZE", "1")), int(env.get("LOCAL_RANK", "0"))


def run_steps(engines, streams, first_step, steps, rank, world, n, seed, on_result=None, strong=False):
    """The step loop of bench.py (and of tests/test_dist_gloo.py): step k of this rank is one eh_fuzz_batch over the
    whole attached corpus with case numbers weak_first_case(k, rank, world, n).., on whichever context is free (eh_batch_done) and that
    context's stream (strong=True: step k is ONE run of n cases over all ranks — this rank takes case_range(n, rank, world)
    of it, the shape of erlamsa_main:get_threading_mode/3); a context's previous results are collected before it is reused.
    `streams` are raw stream handles (0 = the null stream).  Returns {"out_bytes", "kernel_ms": [...], "status_counts": int64[6]}; `on_result(step,
    engine)` is called at collection time (the engine still holds that step's results)."""
    import numpy as np
    nctx = len(engines)
    import time
    res = {"out_bytes": 0, "kernel_ms": [], "status_counts": np.zeros(6, dtype=np.int64),
           "host_loop_s": {"collect": 0.0, "collect_max": 0.0, "on_result": 0.0, "launch": 0.0, "launch_max": 0.0, "no_context_free": 0.0}}
    hl = res["host_loop_s"]                           # where the host thread's time goes (bench.py reports it next to ms_per_step)

    def collect_ctx(ci, k):
        t0 = time.perf_counter()
        e = engines[ci]
        _, ob, _, sc = e.summary()                    # waits for that context's batch; totals and statuses summed by the kernel: one small copy
        res["out_bytes"] += ob
        res["kernel_ms"].append(e.kernel_ms())        # HIP events recorded on the launch stream inside the library
        res["status_counts"] += sc
        t1 = time.perf_counter()
        if on_result is not None:
            on_result(k, e)
        t2 = time.perf_counter()
        hl["collect"] += t1 - t0; hl["collect_max"] = max(hl["collect_max"], t1 - t0); hl["on_result"] += t2 - t1

    # A step goes to WHICHEVER context is free.  Passes do not end in the order they were launched (a pass lasts as long as its
    # heaviest case, 0.75 - 2 s alone), and waiting for the oldest one leaves contexts - and, once their workgroups have left, the
    # device - idle behind one long tail.
    busy = {}                                          # context index -> step it runs
    free = list(range(nctx))
    for k in range(first_step, first_step + steps):
        tw0, c0 = time.perf_counter(), hl["collect"] + hl["on_result"]
        while not free:
            for ci in list(busy):
                if engines[ci].done():
                    collect_ctx(ci, busy.pop(ci)); free.append(ci)
            if not free:
                # small batches (configs[1]: 0.1 ms of kernel) end within the first polls: no sleep for 0.3 ms (a sleep is 0.06 ms
                # at best and was the step time of such runs), then half-millisecond naps (passes of the big workload last a second)
                if time.perf_counter() - tw0 > 0.0003:
                    time.sleep(0.0005)
        ci = free.pop(0)
        t0 = time.perf_counter()
        hl["no_context_free"] += (t0 - tw0) - (hl["collect"] + hl["on_result"] - c0)
        if strong:
            first, cnt = case_range(n, rank, world)
            engines[ci].fuzz_batch(seed=seed, first_case=k * n + first + 1, corpus_first=first, n=cnt, stream=streams[ci])
        else:
            engines[ci].fuzz_batch(seed=seed, first_case=weak_first_case(k, rank, world, n), corpus_first=0, n=n, stream=streams[ci])
        dt = time.perf_counter() - t0
        hl["launch"] += dt; hl["launch_max"] = max(hl["launch_max"], dt)
        busy[ci] = k
    for ci in sorted(busy, key=lambda c: busy[c]):
        collect_ctx(ci, busy[ci])
    return res


def reduce_over_ranks(dt, out_bytes, cases, dist=None, device=None):
    """bench.py's aggregation: MAX of the step-loop time over ranks, SUM of bytes and cases.  Returns floats."""
    if dist is None:
        return float(dt), float(out_bytes), float(cases)
    import torch
    tot = torch.tensor([dt, float(out_bytes), float(cases)], dtype=torch.float64, device=device)
    tmax = tot.clone()
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(tmax[0]), float(tot[1]), float(tot[2])
