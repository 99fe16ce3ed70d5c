#!/bin/bash
# round 6, GPU call 1: the round-start build - per-mutator cycle survey of two passes (EH_PROF build), the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06a; mkdir -p $O
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06a_p0 0 40 > $O/survey_pass0.txt 2>&1; head -3 $O/survey_pass0.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06a_p7 458752 40 > $O/survey_pass7.txt 2>&1; head -3 $O/survey_pass7.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
