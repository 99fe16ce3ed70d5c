#!/bin/bash
# round 6, GPU call 33: smaller passes, more of them in flight (same memory): the tail of a pass of 16 384 cases is shorter than that of 65 536, and the device is under-supplied (wave_slots.held 0.75)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06af; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --gpus 1 --pcie 0 --budget-mib 0 --cpu-sample 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-130 $O/bench_$name.json)"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); w=d.get("wave_slots") or {}; print("   held", w.get("held"), "in cases", w.get("in_cases"), "kernel ms", d["roofline"]["kernel_ms_avg"], "status", d.get("case_status"), d.get("warning"))
except Exception as e: print("   failed", e); print(open("$O/bench_$name.err").read()[-800:])
PY
}
run full_k7 --steps 20 --warmup 5 --max-slots 0
GPU_MAX_HW_QUEUES=16 run quarter_k16_q16 --cases 16384 --inflight 16 --out-gib 9 --steps 80 --warmup 20 --max-slots 0
GPU_MAX_HW_QUEUES=24 run quarter_k24_q24 --cases 16384 --inflight 24 --out-gib 8 --steps 80 --warmup 24 --max-slots 0
run quarter_k24_q8 --cases 16384 --inflight 24 --out-gib 8 --steps 80 --warmup 24 --max-slots 0
GPU_MAX_HW_QUEUES=16 run half_k14_q16 --cases 32768 --inflight 14 --out-gib 14 --steps 40 --warmup 14 --max-slots 0
