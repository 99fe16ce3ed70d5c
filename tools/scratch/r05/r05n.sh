#!/bin/bash
# round 5, GPU call 18: smaller passes on more contexts - a context is held by its heaviest case; does halving the pass (and doubling the
# contexts, same memory) keep more bulk work queued?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05n; mkdir -p $O
run() { # name hwq cases inflight outgib slots steps warmup
  GPU_MAX_HW_QUEUES=$2 timeout 420 python bench.py --gpus 1 --cases $3 --inflight $4 --out-gib $5 --max-slots $6 --steps $7 --warmup $8 --pcie 0 --budget-mib 0 --cpu-sample 0 --setup-seconds 300 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "ms/step", r["ms_per_step"], "MB/s", r["value"], "cases/s", r["cases_per_s"], "kernel_ms", r["roofline"]["kernel_ms_avg"], r.get("warning"), r["case_stats"]["wave_cycles_per_pass"]["mean_sum_G"], r["case_stats"]["wave_cycles_per_pass"]["heaviest_case_Mcyc_mean_over_passes"], r.get("supervisor"))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
  tail -2 $O/$1.err
}
run half12_q16 16 32768 12 15 512 40 12
run quarter24_q24 24 16384 24 8 256 80 24
run half12_q8 8 32768 12 15 512 40 12
run full6_q16 16 65536 6 27 1024 20 6
