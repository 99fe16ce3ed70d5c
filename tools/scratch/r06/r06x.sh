#!/bin/bash
# round 6, GPU call 25: the finders again (64 piece CRCs combined in parallel, prefix CRCs sixteen bytes a lane, the sizer's pick by scans): configs[3] survey + bench, driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06x; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round3.py -x -q -m gpu > $O/gputest.txt 2>&1; tail -2 $O/gputest.txt
PATTERNS=default ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06x_p0 0 40 > $O/survey_pass0.txt 2>&1; head -1 $O/survey_pass0.txt; grep "pattern phase" $O/survey_pass0.txt
timeout 900 python bench.py --patterns default --steps 12 --warmup 6 --pcie 0 --budget-mib 0 > $O/bench_c4.json 2> $O/bench_c4.err; cut -c1-200 $O/bench_c4.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json
