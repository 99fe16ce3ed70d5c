#!/bin/bash
# round 5, GPU call 11: the trace emitters out of line - is the scheduler loop back to what it cost before the full meta trace?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05j; mkdir -p $O
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "traffic", r["roofline"]["traffic"], "parity", r.get("parity_checked"), r["case_stats"]["wave_cycles_per_pass"])
PY
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "meta_trace" > $O/meta.txt 2>&1; tail -2 $O/meta.txt
