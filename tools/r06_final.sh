#!/bin/bash
# round 6, the closing GPU call: the whole GPU suite, smoke, the driver's bench command, the profile of the round (kernel trace + four
# counter passes), the driver's command again with the counter summary of THIS build attached, the other BASELINE configurations, the
# per-mutator survey of the final build.   gpurun --timeout 3400 -- 'bash tools/r06_final.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06final; mkdir -p $O
bash tools/gpu_probe.sh > $O/probe.txt 2>&1 || { cat $O/probe.txt; exit 9; }
timeout 900 python -m pytest tests -q -m gpu > $O/gputest.txt 2>&1; tail -4 $O/gputest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_before_profile.json 2> $O/bench1.err; cut -c1-120 $O/bench_before_profile.json
bash tools/profile_round.sh r06 > $O/profile_round.txt 2>&1; tail -3 $O/profile_round.txt
python tools/collect_profiles.py r06 > $O/collect.txt 2>&1; tail -2 $O/collect.txt
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-120 $O/bench.json
cp profiles/r06_summary.json profiles/r06_kernel_stats.csv profiles/r06_kernel_trace_mutate.csv profiles/r06_pmc_eh_mutate_kernel.csv profiles/r06_bench_under_rocprof.json $O/ 2>/dev/null
timeout 300 python bench.py --cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --case-stats 0 --inflight 6 --steps 2400 --warmup 60 --pcie 0 --budget-mib 0 > $O/c2.json 2> $O/c2.err; cut -c1-160 $O/c2.json
timeout 300 python bench.py --cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --case-stats 0 --inflight 1 --steps 400 --warmup 20 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/c2_inflight1.json 2> $O/c2b.err; cut -c1-160 $O/c2_inflight1.json
timeout 400 python bench.py --patterns default --steps 12 --warmup 6 --pcie 0 --budget-mib 0 > $O/c4.json 2> $O/c4.err; cut -c1-160 $O/c4.json
timeout 300 python bench.py --config 5 --cases 32768 --steps 40 --warmup 6 --pcie 0 > $O/c5.json 2> $O/c5.err; cut -c1-160 $O/c5.json
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06final_p0 0 40 > $O/survey_pass0.txt 2>&1; head -1 $O/survey_pass0.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06final_p7 458752 40 > $O/survey_pass7.txt 2>&1; head -1 $O/survey_pass7.txt
PATTERNS=default ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06final_c4_p0 0 40 > $O/survey_c4_pass0.txt 2>&1; head -1 $O/survey_c4_pass0.txt
