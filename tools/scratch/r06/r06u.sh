#!/bin/bash
# round 6, GPU call 22: configs[3] after the finders of the sz / cs patterns moved their tables to LDS (sorted right ends instead of 513 x 5 x 6 tests per offset; ballot masks
# instead of flag arrays in work memory): survey with the pattern-phase slots, configs[3] bench, the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06u; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu > $O/gputest.txt 2>&1; tail -2 $O/gputest.txt
PATTERNS=default ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06u_p0 0 40 > $O/survey_pass0.txt 2>&1; head -1 $O/survey_pass0.txt; grep "pattern phase" $O/survey_pass0.txt
timeout 900 python bench.py --patterns default --steps 12 --warmup 6 --pcie 0 --budget-mib 0 > $O/bench_c4.json 2> $O/bench_c4.err; cut -c1-200 $O/bench_c4.json
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
