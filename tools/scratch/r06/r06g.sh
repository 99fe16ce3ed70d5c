#!/bin/bash
# round 6, GPU call 7: fr_occ rewritten (16 positions per lane): the heaviest cases of pass 0 alone (EH_PROF), a survey of pass 0, the bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06g; mkdir -p $O
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 400 python tools/profile_alone.py 0 43614 64577 20050 7692 52508 27694 32620 17327 60421 14052 > $O/alone.txt 2>&1; grep "^case" $O/alone.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06g_p0 0 30 > $O/survey_pass0.txt 2>&1; head -1 $O/survey_pass0.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
