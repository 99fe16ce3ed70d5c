#!/bin/bash
# round 5, first GPU call: the shipped round-4 kernel; how step time depends on the passes in flight (step = b + T / K ?)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05a; mkdir -p $O
bash tools/gpu_probe.sh > $O/probe.txt 2>&1 || { cat $O/probe.txt; exit 9; }
for K in 6 3 1 2 4; do
  S=$((3 * K)); [ $S -lt 6 ] && S=6
  timeout 300 python bench.py --steps $S --warmup $K --inflight $K --cpu-sample 0 --pcie 0 --budget-mib 0 > $O/k$K.json 2> $O/k$K.err
  python - $O/k$K.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("K", r["config"]["passes_in_flight"], "ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], r.get("case_stats", {}).get("wave_cycles_per_pass"))
except Exception as ex:
    print("no result", sys.argv[1], ex)
PY
done
