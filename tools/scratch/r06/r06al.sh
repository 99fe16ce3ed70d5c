#!/bin/bash
# round 6: 8 against 12 hardware queues with seven passes in flight (GPU_MAX_HW_QUEUES set by the host before the library loads), the driver's command, alternating on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06al; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-120 $O/bench_$name.json)"; }
for i in 1 2 3; do
  run q8_$i X=1
  run q12_$i GPU_MAX_HW_QUEUES=12
  run q16_$i GPU_MAX_HW_QUEUES=16
done
