#!/bin/bash
# round 6, GPU call 32: how many workgroups does the device hold when seven passes are in flight (pool counters 18 / 19 / 29 / 39)?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06ae; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-110 $O/bench_$name.json)"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); w=d.get("wave_slots") or {}; print("   held", w.get("held"), "in cases", w.get("in_cases"), "kernel ms", d["roofline"]["kernel_ms_avg"], "resident", d["config"]["work_area_pool"].get("workgroups_resident"))
except Exception as e: print("   failed", e); print(open("$O/bench_$name.err").read()[-600:])
PY
}
run s1024
run s2048 --max-slots 0
run s2048_k1 --max-slots 0 --inflight 1 --steps 6 --warmup 2
run s2048_k3 --max-slots 0 --inflight 3 --steps 12 --warmup 3
