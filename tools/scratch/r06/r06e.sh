#!/bin/bash
# round 6, GPU call 5: which cases of a pass the posted loops shorten: per-case cycles of passes 0 and 7 with and without cooperation (EH_PROF build)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06e; mkdir -p $O
export ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so
for base in 0 458752; do
  ENGINE_FLAGS=64 timeout 300 python tools/survey_pass.py r06e_nocoop_$base $base 40 > $O/nocoop_$base.txt 2>&1; head -1 $O/nocoop_$base.txt
  timeout 300 python tools/survey_pass.py r06e_coop_$base $base 40 > $O/coop_$base.txt 2>&1; head -1 $O/coop_$base.txt
done
