#!/bin/bash
# round 4, fifth GPU call: what the heaviest cases of a pass spend their time in; more passes in flight
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04e; mkdir -p $O
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/profile_alone.py 0 28038 43389 20050 13337 22027 14052 50785 18716 > $O/monsters.txt 2>&1; cat $O/monsters.txt | head -120
timeout 200 python3 bench.py --gpus 1 --steps 16 --warmup 4 --inflight 8 --out-gib 24 --pool-gib 32 --max-slots 768 --cpu-sample 0 --budget-mib 0 --pcie 0 > $O/bench8.log 2> $O/bench8.err; cut -c1-160 $O/bench8.log; grep "timed steps done" $O/bench8.err
timeout 200 python3 bench.py --gpus 1 --steps 16 --warmup 4 --inflight 6 --cpu-sample 0 --budget-mib 0 --pcie 0 > $O/bench6.log 2> $O/bench6.err; cut -c1-160 $O/bench6.log; grep "timed steps done" $O/bench6.err
