#!/bin/bash
# the two GPU tests added after the final sequence; kernel traces of BASELINE configs[1] and the shape of configs[4] on the build that ships
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04z4; mkdir -p $O
bash tools/gpu_probe.sh || exit 0
timeout 300 python -m pytest tests/test_gpu_round4.py -q -m gpu -k "base64 or replay" > $O/t.txt 2>&1; tail -2 $O/t.txt
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --stats -f csv"
timeout 100 $P -d $O/prof_c2 -o c2 -- python $R/bench.py --profiled 1 --cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --steps 400 --warmup 40 --inflight 1 --out-gib 1 --pool-gib 1 --budget-mib 0 --pcie 0 --cpu-sample 0 > $O/c2_prof.log 2> $O/c2_prof.err; cut -c1-160 $O/c2_prof.log
timeout 120 $P -d $O/prof_c5 -o c5 -- python $R/bench.py --profiled 1 --config 5 --cases 32768 --steps 6 --warmup 2 --pcie 0 > $O/c5_prof.log 2> $O/c5_prof.err; cut -c1-160 $O/c5_prof.log
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -3 $f | cut -c1-160; done
