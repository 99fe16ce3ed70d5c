#!/bin/bash
# round 6, GPU call 23: the driver's command with the build of call 22, then the default pass's sixteen heaviest cases alone (EH_PROF build)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06v; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
export ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so
timeout 300 python tools/survey_pass.py r06v_p0 0 40 > $O/survey_pass0.txt 2>&1; head -1 $O/survey_pass0.txt
cases=$(grep "^  case " $O/survey_pass0.txt | head -16 | sed 's/^  case \([0-9]*\):.*/\1/' | tr '\n' ' ')
timeout 900 python tools/profile_alone.py 0 $cases > $O/alone.txt 2>&1; grep "^case" $O/alone.txt
