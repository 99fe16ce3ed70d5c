#!/bin/bash
# round 6, GPU call 9: totals through page-locked memory (no copy call per step), host loop polls without sleeping; survey with the write accounting
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06i; mkdir -p $O
B="python bench.py --gpus 1 --pcie 0 --budget-mib 0 --cpu-sample 0"
show() { python - $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    cs = r.get("case_stats", {}).get("wave_cycles_per_pass", {})
    print(sys.argv[1], "MB/s", r["value"], "cases/s", r["cases_per_s"], "ms/step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "sumG", cs.get("mean_sum_G"), "heaviest", cs.get("heaviest_case_Mcyc_mean_over_passes"), r.get("host_loop_ms_per_step"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 400 $B --steps 20 --warmup 5 > $O/k6.json 2> $O/k6.err; show $O/k6.json; tail -2 $O/k6.err
timeout 400 $B --steps 20 --warmup 5 --case-stats 0 > $O/k6_nostats.json 2> $O/k6_nostats.err; show $O/k6_nostats.json
timeout 400 $B --steps 12 --warmup 3 --inflight 3 > $O/k3.json 2> $O/k3.err; show $O/k3.json
C2="--cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --case-stats 0"
timeout 300 $B $C2 --inflight 1 --steps 400 --warmup 20 > $O/c2_k1.json 2> $O/c2_k1.err; show $O/c2_k1.json
timeout 300 $B $C2 --inflight 6 --steps 2400 --warmup 60 > $O/c2_k6.json 2> $O/c2_k6.err; show $O/c2_k6.json
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/survey_pass.py r06i_p0 0 30 > $O/survey_pass0.txt 2>&1; head -1 $O/survey_pass0.txt; grep "work memory taken\|fuse on shortened" $O/survey_pass0.txt
