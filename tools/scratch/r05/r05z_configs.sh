#!/bin/bash
# round 5: the other BASELINE configurations and the small rate tools (part of the first closing call, gpurun_out/r05z)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05z; mkdir -p $O
timeout 200 python bench.py --cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --inflight 1 --steps 200 --warmup 20 --pcie 0 --budget-mib 0 > $O/c2_inflight1.json 2> $O/c2.err; cut -c1-160 $O/c2_inflight1.json
timeout 300 python bench.py --patterns default --steps 12 --warmup 6 --pcie 0 --budget-mib 0 > $O/c4.json 2> $O/c4.err; cut -c1-160 $O/c4.json
timeout 300 python bench.py --config 5 --cases 32768 --steps 40 --warmup 6 --pcie 0 > $O/c5.json 2> $O/c5.err; cut -c1-160 $O/c5.json
timeout 200 python tools/zlib_rate.py > $O/zlib_rate.json 2>&1; cut -c1-200 $O/zlib_rate.json
timeout 200 python tools/coalesce_latency.py > $O/coalesce_latency.json 2>&1; cut -c1-200 $O/coalesce_latency.json
