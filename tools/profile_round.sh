#!/bin/bash
# One profiling call on the GPU box: kernel trace + stats of the bench's own command (passes in flight), then one --pmc pass
# each for FETCH_SIZE, WRITE_SIZE, the SQ set and the instruction-cache set, on ONE pass at a time (--inflight 1: the
# counters serialise dispatches anyway; never combined with trace domains).
# usage: tools/profile_round.sh r03   -> gpurun_out/{prof_<tag>,pmc_*}
tag=${1:-r05}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --profiled 1 --cpu-sample 0 --budget-mib 0 --pcie 0"
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_$tag -o $tag -- $B --steps 12 --warmup 3 > $R/gpurun_out/prof_${tag}_bench.log 2>&1
tail -1 $R/gpurun_out/prof_${tag}_bench.log | cut -c1-200
P="$B --inflight 1 --steps 1 --warmup 0"
timeout 150 rocprofv3 --pmc FETCH_SIZE -f csv -d $R/gpurun_out/pmc_fetch -o p -- $P > $R/gpurun_out/pmc_fetch.log 2>&1
timeout 150 rocprofv3 --pmc WRITE_SIZE -f csv -d $R/gpurun_out/pmc_write -o p -- $P > $R/gpurun_out/pmc_write.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS -f csv -d $R/gpurun_out/pmc_sq -o p -- $P > $R/gpurun_out/pmc_sq.log 2>&1
timeout 150 rocprofv3 --pmc SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_BUSY_CYCLES -f csv -d $R/gpurun_out/pmc_icache -o p -- $P > $R/gpurun_out/pmc_icache.log 2>&1
for f in pmc_fetch pmc_write pmc_sq pmc_icache; do tail -1 $R/gpurun_out/$f.log | cut -c1-120; done
find $R/gpurun_out/prof_$tag $R/gpurun_out/pmc_* -name "*.csv" | head -20
