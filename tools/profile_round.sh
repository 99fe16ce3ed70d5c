#!/bin/bash
# One profiling call on the GPU box: kernel trace + stats, then one --pmc pass each for FETCH_SIZE, WRITE_SIZE and the SQ
# set (never combined with trace domains).  usage: tools/profile_round.sh r02   -> gpurun_out/{prof_<tag>,pmc_*}
tag=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --cpu-sample 0 --budget-mib 0 --pcie 0"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_$tag -o $tag -- $B --steps 6 --warmup 3 > $R/gpurun_out/prof_${tag}_bench.log 2>&1
tail -1 $R/gpurun_out/prof_${tag}_bench.log | cut -c1-200
timeout 120 rocprofv3 --pmc FETCH_SIZE -f csv -d $R/gpurun_out/pmc_fetch -o p -- $B --steps 1 --warmup 0 > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 120 rocprofv3 --pmc WRITE_SIZE -f csv -d $R/gpurun_out/pmc_write -o p -- $B --steps 1 --warmup 0 > $R/gpurun_out/pmc_write.log 2>&1
timeout 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS -f csv -d $R/gpurun_out/pmc_sq -o p -- $B --steps 1 --warmup 0 > $R/gpurun_out/pmc_sq.log 2>&1
for f in pmc_fetch pmc_write pmc_sq; do tail -1 $R/gpurun_out/$f.log | cut -c1-120; done
find $R/gpurun_out/prof_$tag $R/gpurun_out/pmc_* -name "*.csv" | head -20
