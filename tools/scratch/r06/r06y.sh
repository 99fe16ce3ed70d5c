#!/bin/bash
# round 6, GPU call 26: thresholds of the posted loops (env overrides, no rebuild) with the driver's command on one box: default (fb 512K/64K positions, copies 2 MiB/256 KiB) against lower thresholds
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06y; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-110 $O/bench_$name.json)"; }
run default_1 X=1
run fb256k EH_CO_FB_MIN=262144
run fb128k EH_CO_FB_MIN=131072 EH_CO_FB_CHUNK=32768
run default_2 X=1
run fb128k_c64k EH_CO_FB_MIN=131072
run copy512k EH_CO_COPY_MIN=524288 EH_CO_COPY_CHUNK=131072
run both EH_CO_FB_MIN=262144 EH_CO_COPY_MIN=1048576
run default_3 X=1
