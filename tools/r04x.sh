#!/bin/bash
# round 4, GPU call 27, 28: lanes stay inside the LDS window (giant tags back with the wave-wide machine): differential, heaviest cases, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04x; mkdir -p $O
bash tools/gpu_probe.sh || exit 0
timeout 300 python tests/hipemu/emu_sgml_replay.py 6 2 3 > $O/sgml_diff.txt 2>&1; tail -2 $O/sgml_diff.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/profile_alone.py 0 40457 43389 50785 63042 > $O/monsters.txt 2>&1; grep "alone\|phase 2" $O/monsters.txt | cut -c1-200
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --budget-mib 0 --pcie 0 > $O/bench.log 2> $O/bench.err; cut -c1-160 $O/bench.log; grep -o '"kernel_ms_avg": [0-9.]*' $O/bench.log; grep -o '"pool_waits[^]]*]' $O/bench.log
