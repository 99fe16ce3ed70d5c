#!/bin/bash
# round 4, eleventh GPU call: the tokenizer's replay of periodic documents
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04k; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_round4.py -q -m gpu -x -k "sgml or fuse_paths" > $O/t1.txt 2>&1; tail -15 $O/t1.txt
timeout 400 python -m pytest tests -q -m gpu -x -k "bench_workload_full or default_tables or sgml_json or golden or b64" > $O/t2.txt 2>&1; tail -3 $O/t2.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/profile_alone.py 0 43389 20050 13337 50785 > $O/monsters.txt 2>&1; grep "alone\|sgm \|slot  90" $O/monsters.txt
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --budget-mib 0 --pcie 0 > $O/bench.log 2> $O/bench.err; cut -c1-160 $O/bench.log; grep -o '"kernel_ms_avg": [0-9.]*' $O/bench.log
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/survey_pass.py r04k > $O/survey.txt 2>&1; head -2 $O/survey.txt; grep "ft \|fn \|sgm \|b64 \|ts1 \|ts2 \|tr  \|fuse_red\|sgm phases\|replays\|sum of the 1000\|percentiles" $O/survey.txt; grep -A12 "top cases" $O/survey.txt
