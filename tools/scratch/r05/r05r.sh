#!/bin/bash
# round 5, GPU call 22: how many areas of each pool tier are wanted at once, and does moving memory from the small tiers (never waited for)
# to the large ones (56 waits of 130 ms for a 1 GiB area per run) shorten the passes?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05r; mkdir -p $O
run() { # name
  timeout 420 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/$1.json 2> $O/$1.err
  python - $O/$1.json $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    p = r["config"]["work_area_pool"]
    print(sys.argv[2], "ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "areas", p["areas"], "peak", p["peak_wanted"], "waits", p["waits"], "wait_Gticks", [round(x / 1e9, 1) for x in p["wait_ticks"]], r["case_stats"]["wave_cycles_per_pass"]["heaviest_case_Mcyc_mean_over_passes"])
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run default
EH_POOL_SHARES=6,12,10,7,10,13,16,26 run big_tiers
EH_POOL_SHARES=8,16,12,8,11,12,13,20 run mid
