#!/bin/bash
# round 5, GPU call 16: s_setprio for cases that run long - does the last wavefront of a pass finish sooner?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05m; mkdir -p $O
for k in 1 2; do
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 > $O/bench$k.json 2> $O/bench$k.err
python - $O/bench$k.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "parity", r.get("parity_checked"), r["case_stats"]["wave_cycles_per_pass"])
PY
done
