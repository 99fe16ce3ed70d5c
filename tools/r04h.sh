#!/bin/bash
# round 4, eighth GPU call: more large work areas in the pool (the heaviest cases queued for 2 + 2 of them)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04h; mkdir -p $O
B="--cpu-sample 0 --budget-mib 0 --pcie 0"
for k in 1 2; do
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 $B > $O/b$k.log 2> $O/b$k.err; cut -c1-150 $O/b$k.log
python3 - <<PY
import json
d=json.loads(open("$O/b$k.log").read().strip().splitlines()[-1]); p=d["config"]["work_area_pool"]
print(d["ms_per_step"], d["roofline"]["kernel_ms_avg"], "areas", p["areas"], "waits", p["waits"], "wait Gticks", [round(x/1e9,1) for x in p["wait_ticks"]], d["case_status"])
PY
done
rocm-smi --showmeminfo vram | grep Used
