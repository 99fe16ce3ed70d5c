#!/usr/bin/env python3
"""Writes tests/golden/bench_r03.npz: what the ORACLE (oracle/, the C++ restatement — not a BEAM run) gives for the bench
workload (BASELINE configs[2]: synth.mixed(65536, 4096), the reference's full default mutator table, patterns od,nd,bu, seed
{1,2,3}, case I = corpus row I-1, no work budget) on
  * rows 0 .. 4095 (cases 1 .. 4096), and
  * the heaviest cases of the whole pass by wavefront cycles (tests/golden/bench_heavy_cases.json, picked with
    eh_result_cycles on the MI355X: tools/survey_pass.py),
as (case index, status, draws, length, SHA-1).  tests/test_gpu_parity.py::test_bench_workload_full_table_vs_oracle compares
the engine's results of the same pass with it; tests/test_oracle_bench_golden.py re-derives a sample on the CPU.
usage: tests/golden/make_bench_golden.py [THREADS]"""
import hashlib, json, os, sys, threading, time
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pyoracle as po
from erlamsa_amd import synth

BIG = 1 << 30            # the bench's big_case_bytes, mirrored by the oracle's engine guard
PATS = "od,nd,bu"
SEED = (1, 2, 3)


def oracle_case(mat, i):
    d, o = synth.as_arena(mat[i:i + 1])
    outs, st, dr, _ = po.fuzz_batch(d, o, seed=SEED, patterns=PATS, first_case=i + 1, max_case_bytes=BIG)
    return int(st[0]), int(dr[0]), len(outs[0]), hashlib.sha1(outs[0]).digest()


def main():
    threads = int(sys.argv[1]) if len(sys.argv) > 1 else (os.cpu_count() or 8)
    mat = synth.mixed(65536, 4096)
    heavy = json.load(open(os.path.join(HERE, "bench_heavy_cases.json")))["cases"]
    idx = list(range(4096)) + [i for i in heavy if i >= 4096]
    res = {}
    lock = threading.Lock()
    po.lib()
    t0 = time.time()

    def run(todo, nthreads):
        def worker():
            while True:
                with lock:
                    if not todo:
                        return
                    i = todo.pop()
                r = oracle_case(mat, i)
                with lock:
                    res[i] = r
        ts = [threading.Thread(target=worker) for _ in range(nthreads)]
        [t.start() for t in ts]; [t.join() for t in ts]

    hv = set(heavy)
    run([i for i in reversed(idx) if i not in hv], threads)
    run([i for i in reversed(idx) if i in hv], max(1, min(3, threads)))     # the oracle needs several GB for a case with a 1 GB output
    idx = np.array(idx, dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "bench_r03.npz"), idx=idx,
                        status=np.array([res[i][0] for i in idx], dtype=np.int32), draws=np.array([res[i][1] for i in idx], dtype=np.uint64),
                        lens=np.array([res[i][2] for i in idx], dtype=np.uint64),
                        sha1=np.frombuffer(b"".join(res[i][3] for i in idx), dtype=np.uint8).reshape(len(idx), 20),
                        generator=np.array("oracle (C++ restatement) - NOT a BEAM run"))
    print("%d cases in %.0f s; status counts %s" % (len(idx), time.time() - t0, np.bincount([res[i][0] for i in idx], minlength=7).tolist()))


if __name__ == "__main__":
    main()
