#!/bin/bash
# round 6, GPU call 2: HBM pointers typed address space 1 (global_* instead of flat_*): the GPU suite, then the driver's bench command
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06b; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench2.json 2> $O/bench2.err; cut -c1-300 $O/bench2.json
