#!/bin/bash
# round 4, GPU call 23: box probe, then (healthy box) the lanes build: parity tests over the bench workload, bench, survey
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04t; mkdir -p $O
bash tools/gpu_probe.sh || exit 0
timeout 400 python -m pytest tests -q -m gpu -x -k "bench_workload_full or default_tables or sgml_json or golden or b64 or meta_trace" > $O/t2.txt 2>&1; tail -3 $O/t2.txt
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --budget-mib 0 --pcie 0 > $O/bench.log 2> $O/bench.err; cut -c1-160 $O/bench.log; grep -o '"kernel_ms_avg": [0-9.]*' $O/bench.log
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/survey_pass.py r04t > $O/survey.txt 2>&1; head -1 $O/survey.txt; grep "sgm \|sgm phases\|replays\|slot  85" $O/survey.txt; grep -A8 "top cases" $O/survey.txt
