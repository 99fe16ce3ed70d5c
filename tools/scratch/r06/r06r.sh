#!/bin/bash
# round 6, GPU call 19: per-mutator cycle survey of the build that ships (EH_PROF build of the final sources), passes 0 and 7
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06r; mkdir -p $O
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06r_p0 0 40 > $O/survey_pass0.txt 2>&1; head -1 $O/survey_pass0.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06r_p7 458752 40 > $O/survey_pass7.txt 2>&1; head -1 $O/survey_pass7.txt
