#!/bin/bash
# round 5, GPU call 3: where the sgm tokenizer's "between attempts" and "accepted tags" time goes (EH_PROF sub-slots 112-119)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05c; mkdir -p $O
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/profile_alone.py @tools/scratch/r05c_cases.txt > $O/alone.txt 2>&1
grep -c alone $O/alone.txt
