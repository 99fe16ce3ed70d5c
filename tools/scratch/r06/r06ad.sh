#!/bin/bash
# round 6, GPU call 31: the wave slots belong to the device (tier 0 of the pool), a pass may bring a workgroup for every slot: workgroups per pass, pool size, passes in flight - one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06ad; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py tests/test_gpu_round3.py -x -q -m gpu > $O/gputest.txt 2>&1; tail -2 $O/gputest.txt
run() { name=$1; shift; timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-110 $O/bench_$name.json)"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); w=d.get("wave_slots") or {}; print("   held", w.get("held"), "in cases", w.get("in_cases"), "linger", w.get("lingering_for_posted_loops"), "kernel ms", d["roofline"]["kernel_ms_avg"], "inflight", d["config"]["passes_in_flight"], "wg", d["config"]["workgroups_per_pass"], "pool waits", d["config"]["work_area_pool"]["waits"])
except Exception as e: print("   failed", e); print(open("$O/bench_$name.err").read()[-600:])
PY
}
run s1024
run s2048 --max-slots 0
run s2048_p60 --max-slots 0 --pool-gib 60
run s1536 --max-slots 1536
run s2048_k8_p48 --max-slots 0 --inflight 8 --pool-gib 48
run s2048_k6_p60 --max-slots 0 --inflight 6 --pool-gib 60
run s2048_b --max-slots 0
run s1024_b
