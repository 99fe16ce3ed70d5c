#!/bin/bash
# The round's last GPU call, on the commit that ships: the GPU tests, the smoke run, the driver's exact bench command, the
# profiles of that build (tools/profile_round.sh: kernel trace + stats under the bench's own command, four PMC passes), kernel
# traces of the other BASELINE configurations, one run with the runtime's default number of hardware queues.
#   gpurun --timeout 2400 -- 'bash tools/r04_final.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04z; mkdir -p $O
timeout 900 python -m pytest tests -q -m gpu -x > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/bench_r04.log 2> $R/gpurun_out/bench_r04.err; cut -c1-220 $R/gpurun_out/bench_r04.log; grep -c "child" $R/gpurun_out/bench_r04.err
bash tools/profile_round.sh r04 2>&1 | tail -12
cd /tmp && export TMPDIR=/tmp
P="rocprofv3 --kernel-trace --stats -f csv"
timeout 150 $P -d $O/prof_c2 -o c2 -- python $R/bench.py --profiled 1 --cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --steps 400 --warmup 40 --inflight 1 --out-gib 1 --pool-gib 1 --budget-mib 0 --pcie 0 --cpu-sample 0 > $O/c2_prof.log 2> $O/c2_prof.err; cut -c1-160 $O/c2_prof.log
timeout 300 $P -d $O/prof_c4 -o c4 -- python $R/bench.py --profiled 1 --patterns default --steps 12 --warmup 3 --budget-mib 0 --pcie 0 --cpu-sample 0 > $O/c4_prof.log 2> $O/c4_prof.err; cut -c1-160 $O/c4_prof.log
timeout 300 $P -d $O/prof_c5 -o c5 -- python $R/bench.py --profiled 1 --config 5 --cases 32768 --steps 6 --warmup 2 --pcie 0 > $O/c5_prof.log 2> $O/c5_prof.err; cut -c1-160 $O/c5_prof.log
cd $R
GPU_MAX_HW_QUEUES=4 timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --budget-mib 0 --pcie 0 > $O/bench_q4.log 2> $O/bench_q4.err; cut -c1-160 $O/bench_q4.log
for f in $(find $O -name "*kernel_stats.csv"); do echo $f; head -3 $f | cut -c1-160; done
