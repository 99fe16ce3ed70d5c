#!/bin/bash
# round 6, GPU call 21: configs[3] after the checksum rewrite (slicing CRC out of LDS, 16-byte xor8 / adler32): pattern-phase slots 116-124
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_round3.py tests/test_gpu_round6.py -x -q -m gpu > $O/gputest.txt 2>&1; tail -2 $O/gputest.txt
export PATTERNS=default
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06t_p0 0 40 > $O/survey_pass0.txt 2>&1; head -1 $O/survey_pass0.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 600 python tools/profile_alone.py 0 52880 35674 51034 55529 51614 20664 > $O/alone.txt 2>&1; grep "^case" $O/alone.txt
unset PATTERNS
timeout 900 python bench.py --patterns default --steps 12 --warmup 6 --pcie 0 --budget-mib 0 > $O/bench_c4.json 2> $O/bench_c4.err; cat $O/bench_c4.json | cut -c1-400
