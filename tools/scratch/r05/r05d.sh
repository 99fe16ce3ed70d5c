#!/bin/bash
# round 5, GPU call 4: attribute batches in the sgm tokenizer - the worst cases alone again, the worst passes again, the bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05d; mkdir -p $O
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/profile_alone.py @tools/scratch/r05c_cases.txt > $O/alone.txt 2>&1
grep "alone:\|slot 120\|slot  90" $O/alone.txt
for k in 11 12 22 24 15; do timeout 120 python tools/r05_monsters.py $k 1 5 $O/m$k.json >> $O/passes.txt 2>&1; done; cat $O/passes.txt
timeout 300 python bench.py --steps 18 --warmup 6 --cpu-sample 2048 --pcie 0 --budget-mib 0 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; tail -3 $O/bench.err
python - $O/bench.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "parity", r.get("parity_checked"), r.get("case_stats", {}).get("wave_cycles_per_pass"))
PY
