#!/bin/bash
# round 6: eight passes in flight once more - fewer slots per context, a smaller arena, more hardware queues (no source change: options and environment only)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06ak; mkdir -p $O
run() { name=$1; shift; env $ENVV timeout 600 python bench.py --gpus 1 --steps 24 --warmup 8 --pcie 0 --budget-mib 0 --cpu-sample 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-140 $O/bench_$name.json)"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); w=d.get("wave_slots") or {}; print("   held", w.get("held"), "in cases", w.get("in_cases"), "kernel ms", d["roofline"]["kernel_ms_avg"], "K", d["config"]["passes_in_flight"], "status", d["case_status"]["arena_full"], "waits", d["config"]["work_area_pool"]["waits"])
except Exception as e: print("   failed", e); print(open("$O/bench_$name.err").read()[-500:])
PY
}
ENVV="X=1" run k7
ENVV="GPU_MAX_HW_QUEUES=12" run k8_q12 --inflight 8 --max-slots 768 --pool-gib 40 --out-gib 26
ENVV="X=1" run k8_q8 --inflight 8 --max-slots 768 --pool-gib 40 --out-gib 26
ENVV="GPU_MAX_HW_QUEUES=12" run k7_q12
ENVV="GPU_MAX_HW_QUEUES=12" run k8_q12_s512 --inflight 8 --max-slots 512 --pool-gib 44 --out-gib 26
