#!/usr/bin/env python3
"""Latency of the request coalescer (eh_submit / eh_flush / eh_poll, the erlamsa_fsupervisor / erlamsa_esi shape of SURVEY §8(f)-1)
for flushes of 1, 64 and 4 096 requests: p50 / p99 of submit-to-result per flush size, and of one request alone through
eh_fuzz_calls.  One JSON line.  Not run in round 3 (no GPU minutes were left when it was written); works on the emulator build.

  python tools/coalesce_latency.py [--reps 30] [--size 1024] [--mutations ...] [--patterns od,nd,bu]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth


def pct(v, p):
    v = sorted(v)
    return round(1e3 * v[min(len(v) - 1, int(p * len(v)))], 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--mutations", default=None)
    ap.add_argument("--patterns", default="od,nd,bu")
    ap.add_argument("--sizes", default="1,64,4096")
    a = ap.parse_args()
    rows = [bytes(r) for r in synth.mixed(4096, a.size, seed=9)]
    eng = ea.Engine(0)
    eng.configure(mutations=a.mutations, patterns=a.patterns, max_case_bytes=4 << 20)
    res = {"request_bytes": a.size, "reps": a.reps, "unit": "ms", "flush": {}}
    for nreq in [int(x) for x in a.sizes.split(",")]:
        eng.coalesce_limits(nreq + 1, 1 << 40)                  # nothing launches before the explicit flush
        lat = []
        for rep in range(a.reps + 2):
            t0 = time.perf_counter()
            tickets = [eng.submit(rows[(rep * 7 + i) % len(rows)], (rep + 1, i + 1, 3)) for i in range(nreq)]
            t1 = time.perf_counter()
            eng.flush()
            first = None
            for t in tickets:
                while True:
                    r = eng.poll(t)
                    if r is not None:
                        break
                if first is None:
                    first = time.perf_counter()
            t2 = time.perf_counter()
            if rep >= 2:                                         # two warm-up flushes (allocation of the context's buffers)
                lat.append((t1 - t0, first - t1, t2 - t0))
        res["flush"][str(nreq)] = {"submit_p50": pct([x[0] for x in lat], 0.5), "flush_to_first_result_p50": pct([x[1] for x in lat], 0.5),
                                   "flush_to_first_result_p99": pct([x[1] for x in lat], 0.99), "submit_to_all_results_p50": pct([x[2] for x in lat], 0.5),
                                   "submit_to_all_results_p99": pct([x[2] for x in lat], 0.99)}
    eng.close()
    e2 = ea.Engine(0)
    e2.configure(mutations=a.mutations, patterns=a.patterns, max_case_bytes=4 << 20)
    one = []
    for rep in range(a.reps + 2):
        data, off = ea.pack_corpus([rows[rep % len(rows)]])
        t0 = time.perf_counter()
        e2.upload_corpus(data, off)
        e2.fuzz_calls(np.array([[rep + 1, 2, 3]], dtype=np.int64))
        e2.download()
        if rep >= 2:
            one.append(time.perf_counter() - t0)
    e2.close()
    res["single_call_upload_run_download"] = {"p50": pct(one, 0.5), "p99": pct(one, 0.99)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
