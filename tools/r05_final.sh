#!/bin/bash
# round 5, the closing GPU call: the whole GPU suite, smoke, the driver's bench command, the profile of the round (kernel trace + four
# counter passes), the other BASELINE configurations, the small rate tools.   gpurun --timeout 2400 -- 'bash tools/r05_final.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05z; mkdir -p $O
bash tools/gpu_probe.sh > $O/probe.txt 2>&1 || { cat $O/probe.txt; exit 9; }
timeout 900 python -m pytest tests -q -m gpu > $O/gputest.txt 2>&1; tail -4 $O/gputest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-160 $O/bench.json
bash tools/profile_round.sh r05 > $O/profile_round.txt 2>&1; tail -12 $O/profile_round.txt
timeout 200 python bench.py --cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --inflight 1 --steps 200 --warmup 20 --pcie 0 --budget-mib 0 > $O/c2_inflight1.json 2> $O/c2.err; cut -c1-160 $O/c2_inflight1.json
timeout 300 python bench.py --patterns default --steps 12 --warmup 6 --pcie 0 --budget-mib 0 > $O/c4.json 2> $O/c4.err; cut -c1-160 $O/c4.json
timeout 300 python bench.py --config 5 --cases 32768 --steps 40 --warmup 6 --pcie 0 > $O/c5.json 2> $O/c5.err; cut -c1-160 $O/c5.json
timeout 200 python tools/zlib_rate.py > $O/zlib_rate.json 2>&1; cut -c1-200 $O/zlib_rate.json
timeout 200 python tools/coalesce_latency.py > $O/coalesce_latency.json 2>&1; cut -c1-200 $O/coalesce_latency.json
