"""N>1 path on CPU: gloo ranks broadcast the seed arena and run bench.py's own step loop (shard.run_steps, shard.reduce_
over_ranks) over it with the ENGINE — the kernel code on the CPU wavefront emulator (tests/hipemu), where device
pointers are host pointers, so the broadcast torch tensors are attached exactly as bench.py attaches its HBM arena.
Rank r, step k runs case numbers weak_first_case(k, r, W, n)..; the union over ranks and steps must equal one
single-process oracle run over the same case numbers, independent of W.  The oracle is only the
checker here."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
MUTS, PATS, SEED = "bd,bf,bi,sr,num,ld,ab", "od,nd,bu", (1, 2, 3)
N, SIZE, STEPS = 24, 160, 2


def _worker(rank, world, port, emu_lib, q, strong=False, cfg=None):
    muts, pats, gens = cfg or (MUTS, PATS, None)
    os.environ["ERLAMSA_HIP_LIB"] = emu_lib
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import erlamsa_amd as ea
    from erlamsa_amd import shard, synth
    r, w, _ = shard.rank_env()
    dist.init_process_group("gloo", rank=r, world_size=w)
    arena = torch.zeros(N * SIZE, dtype=torch.uint8)
    offs = torch.zeros(N + 1, dtype=torch.int64)
    if r == 0:
        arena.copy_(torch.from_numpy(synth.mixed(N, SIZE).reshape(-1)))
        offs.copy_(torch.arange(N + 1, dtype=torch.int64) * SIZE)
    shard.broadcast_corpus(arena, offs, src=0)
    engines = []
    for _ in range(2):                                    # two contexts in flight, like bench.py --inflight 2
        e = ea.Engine(0)
        e.configure(mutations=muts, patterns=pats, generators=gens, max_case_bytes=1 << 20)
        e.attach_corpus(arena.data_ptr(), offs.data_ptr(), N, N * SIZE)
        engines.append(e)
    got = {}

    def keep(step, e):
        outs, st = e.download()
        got[step] = (outs, st.tolist())

    res = shard.run_steps(engines, [0, 0], 0, STEPS, r, w, N, SEED, on_result=keep, strong=strong)
    mine = shard.case_range(N, r, w)[1] if strong else N
    dt_all, out_all, cases_all = shard.reduce_over_ranks(1.0 + r, res["out_bytes"], mine * STEPS, dist, None)
    gathered = [None] * w
    dist.all_gather_object(gathered, (r, got, int(res["out_bytes"]), res["status_counts"].tolist()))
    if r == 0:
        q.put((arena.numpy().copy(), gathered, (dt_all, out_all, cases_all)))
    for e in engines:
        e.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_engine_run_equals_single_process_oracle(world):
    import torch.multiprocessing as mp
    import build_emu
    emu_lib = build_emu.build()
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    from erlamsa_amd import shard
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, emu_lib, q)) for r in range(world)]
    for p in procs:
        p.start()
    arena, gathered, (dt_all, out_all, cases_all) = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    offs = (np.arange(N + 1, dtype=np.uint64) * SIZE)
    total = 0
    for r, got, out_bytes, _ in gathered:
        assert sorted(got) == list(range(STEPS))
        for step, (outs, sts) in got.items():
            first = shard.weak_first_case(step, r, world, N)
            assert first == (step * world + r) * N + 1
            want, wst, _, _ = po.fuzz_batch(arena, offs, seed=SEED, mutations=MUTS, patterns=PATS, first_case=first)
            assert sts == wst.tolist()
            assert outs == want, "rank %d step %d" % (r, step)
            total += sum(len(o) for o in outs)
        assert out_bytes == sum(sum(len(o) for o in got[s][0]) for s in got)
    # the aggregation bench.py prints: max of the times, sums of bytes and cases
    assert dt_all == float(world) and out_all == float(total) and cases_all == float(world * N * STEPS)
    # no two (rank, step) pairs share a case number
    blocks = sorted(shard.weak_first_case(s, r, world, N) for r in range(world) for s in range(STEPS))
    assert blocks == [k * N + 1 for k in range(world * STEPS)]


def test_strong_scaling_split_gathers_to_one_run_in_case_order():
    """bench.py --scaling strong: a step is ONE run of N cases, rank r takes shard.case_range(N, r, W) of it; the ranks' outputs
    concatenated in rank order are the single-process oracle run of that step, case by case (erlamsa_main.erl:95-108)."""
    import torch.multiprocessing as mp
    import build_emu
    emu_lib = build_emu.build()
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, emu_lib, q, True)) for r in range(world)]
    for p in procs:
        p.start()
    arena, gathered, (dt_all, out_all, cases_all) = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    offs = (np.arange(N + 1, dtype=np.uint64) * SIZE)
    assert cases_all == float(N * STEPS)
    by_rank = {r: got for r, got, _, _ in gathered}
    for step in range(STEPS):
        want, wst, _, _ = po.fuzz_batch(arena, offs, seed=SEED, mutations=MUTS, patterns=PATS, first_case=step * N + 1)
        outs = [o for r in range(world) for o in by_rank[r][step][0]]
        sts = [x for r in range(world) for x in by_rank[r][step][1]]
        assert len(outs) == N and sts == wst.tolist() and outs == want, "step %d" % step


def test_c5_shape_jump_generator_over_the_whole_arena_on_every_rank():
    """BASELINE configs[4] in small: generator `jump` (cross-seed splices, erlamsa_gen.erl:124-150), the fuse family + num + len,
    pattern sz, one run split over 3 ranks (strong scaling).  A rank runs only its case range but draws its Paths from the WHOLE
    broadcast arena (the full arena on every GPU); the ranks' outputs in rank order are the single-process oracle run."""
    import torch.multiprocessing as mp
    import build_emu
    emu_lib = build_emu.build()
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pyoracle as po
    world = 3
    cfg = ("ft,fn,fo,num,len", "sz", "jump")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, emu_lib, q, True, cfg)) for r in range(world)]
    for p in procs:
        p.start()
    arena, gathered, (dt_all, out_all, cases_all) = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    offs = (np.arange(N + 1, dtype=np.uint64) * SIZE)
    by_rank = {r: got for r, got, _, _ in gathered}
    for step in range(STEPS):
        want, wst, _, _ = po.fuzz_batch(arena, offs, seed=SEED, mutations=cfg[0], patterns=cfg[1], generators=cfg[2], first_case=step * N + 1)
        outs = [o for r in range(world) for o in by_rank[r][step][0]]
        sts = [x for r in range(world) for x in by_rank[r][step][1]]
        assert len(outs) == N and sts == wst.tolist() and outs == want, "step %d" % step
        assert any(o not in arena.tobytes() for o in outs if o)           # splices of two entries, not copies of one
