#!/bin/bash
# round 5, the last GPU call: a host-side fix in csrc/eh_comm.h changed the sources' hash after the closing call (tools/r05_final.sh: the
# whole GPU suite, 88 passed) - the profile of the round and the driver's command once more, so that the counter summary is of the tree that ships.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05y2; mkdir -p $O
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
bash tools/profile_round.sh r05 > $O/profile_round.txt 2>&1; tail -3 $O/profile_round.txt
python tools/collect_profiles.py r05 > $O/collect.txt 2>&1; tail -2 $O/collect.txt
cp profiles/r05_summary.json profiles/r05_kernel_stats.csv profiles/r05_kernel_trace_mutate.csv profiles/r05_pmc_eh_mutate_kernel.csv profiles/r05_bench_under_rocprof.json $O/ 2>/dev/null
timeout 100 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-120 $O/bench.json
