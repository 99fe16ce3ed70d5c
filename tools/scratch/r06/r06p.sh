#!/bin/bash
# round 6, GPU call 16: seven passes in flight (pool 42 GiB) against six (pool 60 GiB) with the driver's step counts, alternating on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06p; mkdir -p $O
B="python bench.py --gpus 1 --pcie 0 --budget-mib 0 --cpu-sample 0 --steps 20 --warmup 5"
show() { python - $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    cs = r.get("case_stats", {}).get("wave_cycles_per_pass", {})
    print(sys.argv[1], "MB/s", r["value"], "cases/s", r["cases_per_s"], "ms/step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "sumG", cs.get("mean_sum_G"), "heaviest", cs.get("heaviest_case_Mcyc_mean_over_passes"), "waits", r["config"]["work_area_pool"]["waits"], r.get("warning"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for i in 1 2; do
  timeout 400 $B > $O/k6_$i.json 2> $O/k6_$i.err; show $O/k6_$i.json
  timeout 400 $B --inflight 7 --pool-gib 42 > $O/k7_$i.json 2> $O/k7_$i.err; show $O/k7_$i.json; tail -1 $O/k7_$i.err | cut -c1-200
done
timeout 400 $B --inflight 7 --pool-gib 34 > $O/k7_pool34.json 2> $O/k7_pool34.err; show $O/k7_pool34.json
timeout 400 $B --inflight 8 --pool-gib 30 --out-gib 27 > $O/k8_pool30.json 2> $O/k8_pool30.err; show $O/k8_pool30.json; tail -1 $O/k8_pool30.err | cut -c1-200
