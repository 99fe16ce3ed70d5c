#!/bin/bash
# round 6, GPU call 28: how many wavefronts of a pass linger for posted chunks (EH_CO_LINGER; 48 so far), the driver's command, one box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06aa; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-110 $O/bench_$name.json)"; }
for i in 1 2 3; do
  run l48_$i EH_CO_LINGER=48
  run l16_$i EH_CO_LINGER=16
  run l8_$i EH_CO_LINGER=8
  run l24_$i EH_CO_LINGER=24
done
