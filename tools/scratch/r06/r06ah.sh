#!/bin/bash
# round 6, GPU call 36: cases a workgroup claims per atomic (EH_TICKET_BATCH; 4 so far): the cases behind a heavy one in its batch wait for it - the driver's command and configs[1], one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06ah; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-110 $O/bench_$name.json)"; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$name.json")); w=d.get("wave_slots") or {}; print("   held", w.get("held"), "in cases", w.get("in_cases"), "kernel ms", d["roofline"]["kernel_ms_avg"])
except Exception as e: print("   failed", e)
PY
}
c2() { name=$1; shift; env "$@" timeout 300 python bench.py --cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --case-stats 0 --inflight 6 --steps 2400 --warmup 60 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/c2_$name.json 2> $O/c2_$name.err; echo "c2 $name $(cut -c1-150 $O/c2_$name.json)"; }
for i in 1 2; do
  run b4_$i EH_TICKET_BATCH=4
  run b1_$i EH_TICKET_BATCH=1
  run b2_$i EH_TICKET_BATCH=2
done
c2 b4 EH_TICKET_BATCH=4
c2 b1 EH_TICKET_BATCH=1
c2 b4_again EH_TICKET_BATCH=4
c2 b1_again EH_TICKET_BATCH=1
