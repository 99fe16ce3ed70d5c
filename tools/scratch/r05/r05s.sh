#!/bin/bash
# round 5, GPU call 23: twelve half-sized passes with 256 workgroups each (scratch 128 MiB per queue instead of 256 - 512)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05s; mkdir -p $O
run() { # name cases inflight outgib slots steps warmup
  timeout 420 python bench.py --gpus 1 --cases $2 --inflight $3 --out-gib $4 --max-slots $5 --steps $6 --warmup $7 --pcie 0 --budget-mib 0 --cpu-sample 0 --setup-seconds 300 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "contexts", r["config"]["work_area_pool"]["contexts"], r.get("host_loop_ms_per_step"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  grep -c "Aborting" $O/$1.err
}
GPU_MAX_HW_QUEUES=12 run half12_q12_slots256 32768 12 15 256 40 12
GPU_MAX_HW_QUEUES=12 run half12_q12_slots384 32768 12 15 384 40 12
GPU_MAX_HW_QUEUES=10 run half10_q10_slots512 32768 10 15 512 40 10
