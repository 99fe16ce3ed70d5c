#!/bin/bash
# round 6: one and three passes in flight with the build that ships (the row "lone pass / fewer passes in flight" of DESIGN.md section 6)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06aj; mkdir -p $O
run() { name=$1; shift; timeout 600 python bench.py --gpus 1 --pcie 0 --budget-mib 0 --cpu-sample 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-140 $O/bench_$name.json)"; python - <<PY
import json
d=json.load(open("$O/bench_$name.json")); w=d.get("wave_slots") or {}; print("   held", w.get("held"), "in cases", w.get("in_cases"), "kernel ms", d["roofline"]["kernel_ms_avg"])
PY
}
run k1 --inflight 1 --steps 8 --warmup 2
run k3 --inflight 3 --steps 15 --warmup 4
run k5 --inflight 5 --steps 20 --warmup 5
run k7 --steps 20 --warmup 5
