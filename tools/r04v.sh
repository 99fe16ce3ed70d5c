#!/bin/bash
# round 4, GPU call 25: lane batches from any text state (failing runs, giant tags), base64 decode by the wave: differentials, heaviest cases, bench, survey
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04v; mkdir -p $O
bash tools/gpu_probe.sh || exit 0
timeout 300 python tests/hipemu/emu_sgml_replay.py 6 2 3 > $O/sgml_diff.txt 2>&1; tail -4 $O/sgml_diff.txt
timeout 400 python -m pytest tests -q -m gpu -x -k "sgml or bench_workload_full or default_tables or golden or b64 or meta_trace or adversarial" > $O/t2.txt 2>&1; tail -3 $O/t2.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/profile_alone.py 0 43389 50785 14052 60421 63042 26265 > $O/monsters.txt 2>&1; grep "alone\|sgm \|phase 2\|slot  68\|slot  85" $O/monsters.txt
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --budget-mib 0 --pcie 0 > $O/bench.log 2> $O/bench.err; cut -c1-160 $O/bench.log; grep -o '"kernel_ms_avg": [0-9.]*' $O/bench.log
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/survey_pass.py r04v > $O/survey.txt 2>&1; head -1 $O/survey.txt; grep "sgm \|b64 \|sgm phases\|replays\|slot  85\|slot  68" $O/survey.txt; grep -A10 "top cases" $O/survey.txt
