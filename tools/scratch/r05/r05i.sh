#!/bin/bash
# round 5, GPU call 10: the driver's command twice more on a fresh box (run-to-run spread), with the round's counter summary attached
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05i; mkdir -p $O
for k in 1 2; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench$k.json 2> $O/bench$k.err
  python - $O/bench$k.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "traffic", r["roofline"]["traffic"], "parity", r.get("parity_checked"), r["case_stats"]["wave_cycles_per_pass"])
PY
done
