#!/bin/bash
# round 6, GPU call 11: per-mutator cycles of pass 0 with the non-recursive scheduler (against r06i's survey of the recursive one)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06k; mkdir -p $O
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06k_p0 0 30 > $O/survey_pass0.txt 2>&1; head -1 $O/survey_pass0.txt
