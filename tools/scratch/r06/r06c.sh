#!/bin/bash
# round 6, GPU call 3: cooperative execution of big copies / compares / fuse2 passes (job board): suite, bench, the same without helpers
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06c; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; tail -2 $O/bench.err
timeout 600 python bench.py --gpus 1 --steps 6 --warmup 2 --inflight 1 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench_k1.json 2> $O/bench_k1.err; cut -c1-300 $O/bench_k1.json
timeout 600 python bench.py --gpus 1 --steps 12 --warmup 3 --inflight 3 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench_k3.json 2> $O/bench_k3.err; cut -c1-300 $O/bench_k3.json
