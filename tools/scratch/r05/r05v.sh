#!/bin/bash
# round 5, GPU call 27: workgroups per pass (a pass alone in its bulk phase can only fill max_slots of the device's 2 048 wave slots)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05v; mkdir -p $O
run() { # name slots poolgib
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --max-slots $2 --pool-gib $3 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "contexts", r["config"]["work_area_pool"]["contexts"], "waits", r["config"]["work_area_pool"]["waits"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  grep -c "Aborting\|child ended" $O/$1.err
}
run slots1024 1024 60
run slots1536 1536 60
run slots2048 2048 48
run slots1024_again 1024 60
