#!/bin/bash
# round 4, ninth GPU call: the configurations that never had a GPU run - BASELINE configs[1] (C2), configs[3] on one GPU (C4: all ten
# default patterns), the shape of configs[4] (C5), each with a rocprofv3 kernel trace + stats; the device codecs' rate; the coalescer's latency
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r04i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --stats -f csv"
timeout 200 $P -d $O/prof_c2 -o c2 -- python $R/bench.py --cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --steps 400 --warmup 40 --inflight 1 --out-gib 1 --pool-gib 1 --budget-mib 0 --pcie 0 --cpu-sample 1024 > $O/c2_inflight1.log 2> $O/c2_inflight1.err; cut -c1-200 $O/c2_inflight1.log
timeout 200 python $R/bench.py --cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --steps 1200 --warmup 60 --inflight 6 --out-gib 1 --pool-gib 1 --budget-mib 0 --pcie 0 --cpu-sample 0 > $O/c2_inflight6.log 2> $O/c2_inflight6.err; cut -c1-200 $O/c2_inflight6.log
timeout 400 $P -d $O/prof_c4 -o c4 -- python $R/bench.py --patterns default --steps 12 --warmup 3 --budget-mib 0 --pcie 0 --cpu-sample 1024 > $O/c4.log 2> $O/c4.err; cut -c1-200 $O/c4.log
timeout 500 $P -d $O/prof_c5 -o c5 -- python $R/bench.py --config 5 --cases 32768 --steps 6 --warmup 2 --pcie 0 > $O/c5.log 2> $O/c5.err; cut -c1-200 $O/c5.log; tail -2 $O/c5.err
cd $R
timeout 200 python tools/zlib_rate.py > $O/zlib_rate.json 2> $O/zlib_rate.err; cut -c1-600 $O/zlib_rate.json
timeout 200 python tools/coalesce_latency.py > $O/coalesce_latency.json 2> $O/coalesce_latency.err; cut -c1-900 $O/coalesce_latency.json
find $O -name "*kernel_stats.csv" | head; for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -4 $f | cut -c1-200; done
