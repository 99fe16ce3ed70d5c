#!/bin/bash
# round 5, GPU call 26: BASELINE configs[3] (all ten patterns) and configs[4]'s shape on the build that ships
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05u; mkdir -p $O
timeout 300 python bench.py --patterns default --steps 12 --warmup 6 --pcie 0 --budget-mib 0 > $O/c4.json 2> $O/c4.err; cut -c1-160 $O/c4.json
timeout 200 python bench.py --config 5 --cases 32768 --steps 40 --warmup 6 --pcie 0 > $O/c5.json 2> $O/c5.err; cut -c1-160 $O/c5.json
