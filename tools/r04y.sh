#!/bin/bash
# round 4, GPU call 29: slots per pass now that the longest cases are a third as long; cycle survey of the build that ships
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04y; mkdir -p $O
bash tools/gpu_probe.sh || exit 0
for s in 768 1536; do
  timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --budget-mib 0 --pcie 0 --max-slots $s > $O/bench_s$s.log 2> $O/bench_s$s.err; echo "slots $s: $(cut -c1-150 $O/bench_s$s.log)"; grep -o '"kernel_ms_avg": [0-9.]*' $O/bench_s$s.log; grep -o '"pool_waits[^]]*]' $O/bench_s$s.log; grep -o '"arena_full[^,]*' $O/bench_s$s.log
done
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/survey_pass.py r04y > $O/survey.txt 2>&1; head -1 $O/survey.txt; grep -A12 "top cases" $O/survey.txt
