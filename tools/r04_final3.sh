#!/bin/bash
# after tools/collect_profiles.py r04: the driver's command again, so that the line carries the counter run of this kernel build; BASELINE configs[3] on one GPU
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04z3; mkdir -p $O
bash tools/gpu_probe.sh || exit 0
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/bench_r04.log 2> $R/gpurun_out/bench_r04.err; cut -c1-220 $R/gpurun_out/bench_r04.log; grep -c "child" $R/gpurun_out/bench_r04.err; grep -o '"traffic": [0-9.e+]*' $R/gpurun_out/bench_r04.log
timeout 200 python3 bench.py --gpus 1 --patterns default --steps 12 --warmup 3 --budget-mib 0 --pcie 0 --cpu-sample 0 > $O/c4.log 2> $O/c4.err; cut -c1-200 $O/c4.log; grep -o '"kernel_ms_avg": [0-9.]*' $O/c4.log
