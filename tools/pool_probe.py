#!/usr/bin/env python3
"""Passes in flight on the shared work-area pool: per-pass kernel ms, time per step and the pool's counters.
usage: tools/pool_probe.py INFLIGHT STEPS [MAX_SLOTS] [POOL_GIB] [N]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("HWQ", "8"))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import erlamsa_amd as ea
from erlamsa_amd import shard, synth

inflight, steps = int(sys.argv[1]), int(sys.argv[2])
max_slots = int(sys.argv[3]) if len(sys.argv) > 3 else 0
pool_gib = int(sys.argv[4]) if len(sys.argv) > 4 else 56
n = int(sys.argv[5]) if len(sys.argv) > 5 else 65536
mat = synth.mixed(65536, 4096)[:n]
dev = torch.device("cuda", 0)
arena = torch.from_numpy(mat.reshape(-1)).to(dev)
offs = (torch.arange(n + 1, dtype=torch.int64, device=dev) * 4096)
engines, streams = [], []
for _ in range(inflight):
    e = ea.Engine(0)
    e.configure(patterns="od,nd,bu", max_slots=max_slots, out_capacity=int(os.environ.get("OUT_GIB", "28")) << 30, max_case_bytes=int(os.environ.get("CASE_MIB", "16")) << 20,
                big_case_bytes=1 << 30, pool_bytes=pool_gib << 30)
    e.attach_corpus(arena.data_ptr(), offs.data_ptr(), n, n * 4096)
    engines.append(e); streams.append(torch.cuda.Stream(device=dev))
for e in engines:
    e.reserve(n)
print("pool:", engines[0].pool_stats(), "free GiB %.1f" % (torch.cuda.mem_get_info()[0] / 2**30), flush=True)
raw = [s.cuda_stream for s in streams]
t0 = time.perf_counter()
for e, st in zip(engines, raw):
    e.fuzz_batch(seed=(1, 2, 3), first_case=1, corpus_first=0, n=n, stream=st)
for e in engines:
    e.sync()
print("set-up passes (same cases on every context): %.2f s, kernel ms %s" % (time.perf_counter() - t0, [round(e.kernel_ms()) for e in engines]), flush=True)
ps0 = engines[0].pool_stats()
torch.cuda.synchronize()
t0 = time.perf_counter()
names = [m[0] for m in ea.mutator_table()]
def on_result(k, e):
    cyc = e.cycles().astype(np.float64); i = int(np.argmax(cyc)); st = e.status(); dr, lm = e.diag(); pk = e.peak()
    print("  step %d: kernel %.0f ms, heaviest case %d (number %d): %.0f Mcyc status %d draws %d last %s peak %.0f MiB; cases above 2 Gcyc: %d" % (
        k, e.kernel_ms(), i, k * n + i + 1, cyc[i] / 1e6, st[i], dr[i], names[lm[i]] if 0 <= lm[i] < len(names) else str(lm[i]), pk[i] / 2**20, int((cyc > 2e9).sum())), flush=True)
r = shard.run_steps(engines, raw, 1, steps, 0, 1, n, (1, 2, 3), on_result=on_result)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ps = engines[0].pool_stats()
print("inflight %d steps %d max_slots %d: %.3f s per step, %.1f GB/s, kernel ms %s" % (inflight, steps, max_slots, dt / steps, r["out_bytes"] / dt / 1e9, [round(x) for x in r["kernel_ms"]]))
print("status", r["status_counts"].tolist())
print("pool tiers MiB", [b >> 20 for b in ps["area_bytes"]], "areas", ps["areas"], "taken during the timed steps:", [a - b for a, b in zip(ps["taken"], ps0["taken"])], "waits", [a - b for a, b in zip(ps["waits"], ps0["waits"])],
      "wait Gticks", [round((a - b) / 1e9, 2) for a, b in zip(ps["wait_ticks"], ps0["wait_ticks"])])
