#!/bin/bash
# round 5, GPU call 6: the whole GPU suite (incl. the new configs[3] / configs[4] / gunzip / RCCL tests), the driver's bench command, worst passes
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/gputest.txt 2>&1; tail -5 $O/gputest.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "parity", r.get("parity_checked"), r.get("case_stats", {}).get("wave_cycles_per_pass"), r.get("pcie"), r.get("with_work_budget"))
PY
for k in 15 24 7 13 1; do timeout 120 python tools/r05_monsters.py $k 1 4 $O/m$k.json >> $O/passes.txt 2>&1; done; cat $O/passes.txt
