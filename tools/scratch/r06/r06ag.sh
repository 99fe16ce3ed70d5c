#!/bin/bash
# round 6, GPU call 34: does a heavy case run slower with passes in flight because it shares its SIMD with another wavefront, or because of the memory system?
# The same run with a build that lets ONE wavefront onto a SIMD (-DEH_WAVES_PER_SIMD=1: 1 024 wave slots): heaviest case per pass, in cycles.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06ag; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-110 $O/bench_$name.json)"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); w=d.get("wave_slots") or {}; print("   held", w.get("held"), "in cases", w.get("in_cases"), "kernel ms", d["roofline"]["kernel_ms_avg"], d["case_stats"]["wave_cycles_per_pass"])
except Exception as e: print("   failed", e); print(open("$O/bench_$name.err").read()[-600:])
PY
}
run two_per_simd X=1
run one_per_simd ERLAMSA_HIP_LIB=build/liberlamsa_hip_w1.so
