#!/bin/bash
# round 6, GPU call 20: BASELINE configs[3] (all ten default patterns): per-mutator survey of passes 0 and 3, then the ten heaviest cases of pass 0 alone
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06s; mkdir -p $O
export ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so PATTERNS=default
timeout 300 python tools/survey_pass.py r06s_p0 0 40 > $O/survey_pass0.txt 2>&1; head -1 $O/survey_pass0.txt
timeout 300 python tools/survey_pass.py r06s_p3 196608 40 > $O/survey_pass3.txt 2>&1; head -1 $O/survey_pass3.txt
cases=$(grep "^  case " $O/survey_pass0.txt | head -10 | sed 's/^  case \([0-9]*\):.*/\1/' | tr '\n' ' ')
timeout 600 python tools/profile_alone.py 0 $cases > $O/alone.txt 2>&1; grep "^case" $O/alone.txt
