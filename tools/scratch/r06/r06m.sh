#!/bin/bash
# round 6, GPU call 13: per-mutator cycles of pass 0, recursive scheduler (2488b92) against the one without recursion, same box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06m; mkdir -p $O
ERLAMSA_HIP_LIB=build/liberlamsa_hip_rec_prof.so timeout 300 python tools/survey_pass.py r06m_rec 0 30 > $O/rec.txt 2>&1; head -1 $O/rec.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06m_flat 0 30 > $O/flat.txt 2>&1; head -1 $O/flat.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_rec_prof.so timeout 300 python tools/survey_pass.py r06m_rec2 0 30 > $O/rec2.txt 2>&1; head -1 $O/rec2.txt
