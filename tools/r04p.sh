#!/bin/bash
# round 4, GPU call 19: tag attempts one per lane (sg_lane_attempt)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04p; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_round4.py -q -m gpu -x -k "sgml" > $O/t1.txt 2>&1; tail -6 $O/t1.txt
timeout 400 python -m pytest tests -q -m gpu -x -k "bench_workload_full or default_tables or sgml_json or golden or b64 or meta_trace" > $O/t2.txt 2>&1; tail -3 $O/t2.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/profile_alone.py 0 43389 50785 14052 60421 > $O/monsters.txt 2>&1; grep "alone\|sgm \|phase 2" $O/monsters.txt
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --budget-mib 0 --pcie 0 > $O/bench.log 2> $O/bench.err; cut -c1-160 $O/bench.log; grep -o '"kernel_ms_avg": [0-9.]*' $O/bench.log
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/survey_pass.py r04p > $O/survey.txt 2>&1; head -1 $O/survey.txt; grep "sgm \|sgm phases\|replays\|slot  85" $O/survey.txt; grep -A8 "top cases" $O/survey.txt
