#!/usr/bin/env python3
"""PCIe-inclusive rate of the boundary when it hands over HOST buffers (eh_corpus_upload + eh_fuzz_batch +
eh_result_download): the figure DESIGN.md quotes next to the HBM-resident `value` of bench.py.
Pageable host memory, one context, no overlap of transfer and compute."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import erlamsa_amd as ea
from erlamsa_amd import synth

n, size = 65536, 4096
muts = ",".join(m for m in ea.gpu_mutators() if m not in {"b64", "ft", "fn", "fo"})
mat = synth.mixed(n, size)
data, off = synth.as_arena(mat)
eng = ea.Engine(0)
eng.configure(mutations=muts, patterns="od,nd,bu", out_capacity=8 << 30)
res = []
for it in range(3):
    t0 = time.perf_counter()
    eng.upload_corpus(data, off)
    t1 = time.perf_counter()
    eng.fuzz_batch(seed=(1, 2, 3), first_case=1 + it * n)
    eng.sync()
    t2 = time.perf_counter()
    _, total, _ = eng.totals()
    buf = np.empty(max(total, 1), dtype=np.uint8)
    o = np.empty(n + 1, dtype=np.uint64)
    st = np.empty(n, dtype=np.int32)
    t3 = time.perf_counter()
    eng._chk(eng.lib.eh_result_download(eng.h, buf.ctypes.data, buf.size, o.ctypes.data, st.ctypes.data))
    t4 = time.perf_counter()
    res.append({"upload_s": t1 - t0, "kernel_s": t2 - t1, "download_s": t4 - t3, "out_bytes": int(total),
                "h2d_GBps": data.nbytes / (t1 - t0) / 1e9, "d2h_GBps": total / (t4 - t3) / 1e9,
                "end_to_end_out_GBps": total / ((t1 - t0) + (t2 - t1) + (t4 - t3)) / 1e9})
print(json.dumps({"workload": "65536 x 4096 B mixed, 35 mutators, od,nd,bu, host buffers in and out (pageable)", "passes": res}))
