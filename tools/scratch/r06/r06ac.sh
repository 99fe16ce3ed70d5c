#!/bin/bash
# round 6, GPU call 30: a quarter of the wave slots is held by nobody (wave_slots.held 0.74): workgroups per pass (1024 = half the device) and slot size, one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06ac; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-110 $O/bench_$name.json)"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); w=d.get("wave_slots") or {}; print("   held", w.get("held"), "in cases", w.get("in_cases"), "linger", w.get("lingering_for_posted_loops"), "kernel ms", d["roofline"]["kernel_ms_avg"], "inflight", d["config"]["passes_in_flight"], "wg", d["config"]["workgroups_per_pass"], "pool waits", d["config"]["work_area_pool"]["waits"])
except Exception as e: print("   failed", e)
PY
}
run default_1
run s2048_c2 --max-slots 2048 --case-mib 2
run s1536_c2 --max-slots 1536 --case-mib 2
run default_2
run s2048_c2_k6 --max-slots 2048 --case-mib 2 --inflight 6 --pool-gib 60
run s2048_c1 --max-slots 2048 --case-mib 1
run s2048_c2_b --max-slots 2048 --case-mib 2
run default_steps60 --steps 60 --warmup 8
