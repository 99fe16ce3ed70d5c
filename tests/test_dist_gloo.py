"""N>1 path on CPU: two gloo ranks broadcast the seed arena, shard the case range, and the union
of what they compute equals the single-process result.  The per-rank 'engine' here is the oracle
(the checker) — the property under test is the sharding/broadcast logic that bench.py and the NIF
shim use, and that results are independent of the number of ranks."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch
    import torch.distributed as dist
    import pyoracle as po
    from erlamsa_amd import shard, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, size = 96, 128
    arena = torch.zeros(n * size, dtype=torch.uint8)
    offs = torch.zeros(n + 1, dtype=torch.int64)
    if rank == 0:
        arena.copy_(torch.from_numpy(synth.mixed(n, size).reshape(-1)))
        offs.copy_(torch.arange(n + 1, dtype=torch.int64) * size)
    shard.broadcast_corpus(arena, offs, src=0)
    first, cnt = shard.case_range(n, rank, world)
    o = offs.numpy().astype(np.uint64)
    sub_off = o[first:first + cnt + 1] - o[first]
    sub = arena.numpy()[int(o[first]):int(o[first + cnt])]
    outs, st, _, _ = po.fuzz_batch(sub if len(sub) else np.zeros(1, np.uint8), sub_off, seed=(1, 2, 3),
                                   mutations="bd,bf,bi,sr,num,ld", patterns="od,nd,bu", first_case=first + 1)
    gathered = [None] * world
    dist.all_gather_object(gathered, (first, outs, st.tolist()))
    if rank == 0:
        q.put((arena.numpy().copy(), offs.numpy().copy(), gathered))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_run_equals_single_process(world):
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    arena, offs, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    whole, st, _, _ = po.fuzz_batch(arena, offs.astype(np.uint64), seed=(1, 2, 3), mutations="bd,bf,bi,sr,num,ld", patterns="od,nd,bu")
    merged = [None] * len(whole)
    for first, outs, sts in gathered:
        for k, o in enumerate(outs):
            merged[first + k] = o
    assert merged == whole
