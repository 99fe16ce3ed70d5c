#!/bin/bash
# round 5, GPU call 19: where does the host thread of the step loop spend its time (smaller passes on more contexts were SLOWER per pass)?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05o; mkdir -p $O
run() { # name hwq cases inflight outgib slots steps warmup extra
  GPU_MAX_HW_QUEUES=$2 timeout 420 python bench.py --gpus 1 --cases $3 --inflight $4 --out-gib $5 --max-slots $6 --steps $7 --warmup $8 --pcie 0 --budget-mib 0 --cpu-sample 0 --setup-seconds 300 $9 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], r.get("host_loop_ms_per_step"), r.get("supervisor"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run full6 8 65536 6 27 1024 20 6
run full6_nostats 8 65536 6 27 1024 20 6 "--case-stats 0"
run half12_q16 16 32768 12 15 512 40 12
run half12_q16_nostats 16 32768 12 15 512 40 12 "--case-stats 0"
