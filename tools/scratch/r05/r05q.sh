#!/bin/bash
# round 5, GPU call 21: is the wait in front of every launch the runtime's scratch memory (8 KiB per lane x 1024 workgroups = 0.5 GiB per
# queue: above HSA_SCRATCH_SINGLE_LIMIT, allocated per dispatch and given back after it)?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05q; mkdir -p $O
run() { # name cases inflight outgib slots steps warmup  (environment from the caller)
  timeout 420 python bench.py --gpus 1 --cases $2 --inflight $3 --out-gib $4 --max-slots $5 --steps $6 --warmup $7 --pcie 0 --budget-mib 0 --cpu-sample 0 --setup-seconds 300 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "contexts", r["config"]["work_area_pool"]["contexts"], r.get("host_loop_ms_per_step"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  grep -c "Aborting\|child ended" $O/$1.err
}
HSA_NO_SCRATCH_RECLAIM=1 run noreclaim_full6 65536 6 27 1024 20 6
HSA_SCRATCH_SINGLE_LIMIT=2147483648 HSA_SCRATCH_SINGLE_LIMIT_ASYNC=2147483648 run limit2g_full6 65536 6 27 1024 20 6
HSA_ENABLE_SCRATCH_ASYNC_RECLAIM=0 run noasync_full6 65536 6 27 1024 20 6
HSA_NO_SCRATCH_RECLAIM=1 GPU_MAX_HW_QUEUES=16 run noreclaim_half12 32768 12 15 512 40 12
HSA_NO_SCRATCH_RECLAIM=1 run noreclaim_full6_slots512 65536 6 27 512 20 6
