#!/bin/bash
# round 4, GPU call 14: the memo of failed attribute-loop entries behind an LDS filter
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04n; mkdir -p $O
timeout 400 python -m pytest tests -q -m gpu -x -k "sgml or default_tables or bench_workload_full or golden or b64" > $O/t.txt 2>&1; tail -2 $O/t.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/profile_alone.py 0 43389 50785 14052 > $O/monsters.txt 2>&1; grep "alone\|sgm \|phase 2" $O/monsters.txt
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --budget-mib 0 --pcie 0 > $O/bench.log 2> $O/bench.err; cut -c1-160 $O/bench.log; grep -o '"kernel_ms_avg": [0-9.]*' $O/bench.log
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/survey_pass.py r04n > $O/survey.txt 2>&1; head -1 $O/survey.txt; grep "sgm \|sgm phases\|replays" $O/survey.txt; grep -A8 "top cases" $O/survey.txt
