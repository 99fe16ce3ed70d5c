#!/bin/bash
# round 6, GPU call 12: the recursive scheduler (commit 2488b92, build/liberlamsa_hip_rec.so) against the one without recursion (the tree), same box, alternating
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06l; mkdir -p $O
B="python bench.py --gpus 1 --pcie 0 --budget-mib 0 --cpu-sample 0"
show() { python - $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    cs = r.get("case_stats", {}).get("wave_cycles_per_pass", {})
    print(sys.argv[1], "MB/s", r["value"], "ms/step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "sumG", cs.get("mean_sum_G"), "heaviest", cs.get("heaviest_case_Mcyc_mean_over_passes"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
for i in 1 2; do
  ERLAMSA_HIP_LIB=build/liberlamsa_hip_rec.so timeout 400 $B --steps 20 --warmup 5 > $O/rec_$i.json 2> $O/rec_$i.err; show $O/rec_$i.json
  timeout 400 $B --steps 20 --warmup 5 > $O/flat_$i.json 2> $O/flat_$i.err; show $O/flat_$i.json
done
ERLAMSA_HIP_LIB=build/liberlamsa_hip_rec.so timeout 400 $B --steps 8 --warmup 2 --inflight 1 > $O/rec_k1.json 2> $O/rec_k1.err; show $O/rec_k1.json
timeout 400 $B --steps 8 --warmup 2 --inflight 1 > $O/flat_k1.json 2> $O/flat_k1.err; show $O/flat_k1.json
