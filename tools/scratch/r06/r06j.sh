#!/bin/bash
# round 6, GPU call 10: nested scheduler calls without device recursion (static stack): GPU suite with the skip report, bench K = 6 / 3 / 1
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06j; mkdir -p $O
rm -f $O/skips.txt
EH_REPORT_SKIPS=$R/$O/skips.txt timeout 1200 python -m pytest tests -q -m gpu -x > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
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
timeout 400 $B --steps 12 --warmup 3 --inflight 3 > $O/k3.json 2> $O/k3.err; show $O/k3.json
timeout 400 $B --steps 8 --warmup 2 --inflight 1 > $O/k1.json 2> $O/k1.err; show $O/k1.json
timeout 400 $B --steps 24 --warmup 6 --inflight 8 --pool-gib 28 --out-gib 26 > $O/k8.json 2> $O/k8.err; show $O/k8.json; tail -2 $O/k8.err
