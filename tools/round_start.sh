#!/bin/bash
# First GPU call of a round: everything the previous round could not run.  usage (from the repo root on the GPU box):
#   gpurun --timeout 1500 -- 'bash tools/round_start.sh r04'
tag=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x > gpurun_out/${tag}_gputest.txt 2>&1; tail -3 gpurun_out/${tag}_gputest.txt
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/${tag}_bench.log 2> gpurun_out/${tag}_bench.err; cut -c1-300 gpurun_out/${tag}_bench.log
timeout 200 python tools/zlib_rate.py > gpurun_out/${tag}_zlib_rate.json 2>&1; cut -c1-400 gpurun_out/${tag}_zlib_rate.json
timeout 200 python tools/coalesce_latency.py > gpurun_out/${tag}_coalesce_latency.json 2>&1; cut -c1-400 gpurun_out/${tag}_coalesce_latency.json
