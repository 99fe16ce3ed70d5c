#!/bin/bash
# round 5, the closing GPU call: the whole GPU suite, smoke, the driver's bench command, the profile of the round (kernel trace + four
# counter passes), the driver's command again with the counter summary of THIS build attached.   gpurun --timeout 2400 -- 'bash tools/r05_final.sh'
# (the other BASELINE configurations and the small rate tools: tools/scratch/r05/r05z_configs.sh, run once on the build before the trace
# emitters went out of line - same kernels outside the scheduler loop)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05y; mkdir -p $O
bash tools/gpu_probe.sh > $O/probe.txt 2>&1 || { cat $O/probe.txt; exit 9; }
timeout 900 python -m pytest tests -q -m gpu > $O/gputest.txt 2>&1; tail -4 $O/gputest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_before_profile.json 2> $O/bench1.err; cut -c1-120 $O/bench_before_profile.json
bash tools/profile_round.sh r05 > $O/profile_round.txt 2>&1; tail -3 $O/profile_round.txt
python tools/collect_profiles.py r05 > $O/collect.txt 2>&1; tail -2 $O/collect.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-120 $O/bench.json
cp profiles/r05_summary.json profiles/r05_kernel_stats.csv profiles/r05_kernel_trace_mutate.csv profiles/r05_pmc_eh_mutate_kernel.csv profiles/r05_bench_under_rocprof.json $O/ 2>/dev/null
