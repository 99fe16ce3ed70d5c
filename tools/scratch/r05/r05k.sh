#!/bin/bash
# round 5, GPU call 13: steps go to whichever context is free (eh_batch_done) instead of waiting for the oldest pass
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05k; mkdir -p $O
for k in 1 2; do
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench$k.json 2> $O/bench$k.err
python - $O/bench$k.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "traffic", r["roofline"]["traffic"], "parity", r.get("parity_checked"), r["case_stats"]["wave_cycles_per_pass"]["mean_sum_G"])
PY
done
timeout 200 python -m pytest tests/test_gpu_round4.py -q -m gpu -k "bench_script" > $O/t.txt 2>&1; tail -2 $O/t.txt
