#!/bin/bash
# round 6, GPU call 6: lingering helpers (up to 48 wavefronts of a pass stay for chunks once the pass is out of tickets) + threshold sweep
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06f; mkdir -p $O
B="python bench.py --gpus 1 --pcie 0 --budget-mib 0 --cpu-sample 0"
show() { python - $1 <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    cs = r.get("case_stats", {}).get("wave_cycles_per_pass", {})
    print(sys.argv[1], "MB/s", r["value"], "ms/step", r["ms_per_step"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "sumG", cs.get("mean_sum_G"), "heaviest", cs.get("heaviest_case_Mcyc_mean_over_passes"), r["config"].get("cooperative_execution"))
except Exception as e:
    print(sys.argv[1], "failed", e)
PY
}
timeout 400 $B --steps 8 --warmup 2 --inflight 1 > $O/linger_k1.json 2> $O/linger_k1.err; show $O/linger_k1.json
timeout 400 $B --steps 12 --warmup 3 --inflight 3 > $O/linger_k3.json 2> $O/linger_k3.err; show $O/linger_k3.json
timeout 400 $B --steps 20 --warmup 5 > $O/linger_k6.json 2> $O/linger_k6.err; show $O/linger_k6.json
export EH_CO_FB_MIN=131072 EH_CO_FB_CHUNK=32768 EH_CO_COPY_MIN=1048576 EH_CO_COPY_CHUNK=131072
timeout 400 $B --steps 8 --warmup 2 --inflight 1 > $O/fine_k1.json 2> $O/fine_k1.err; show $O/fine_k1.json
timeout 400 $B --steps 12 --warmup 3 --inflight 3 > $O/fine_k3.json 2> $O/fine_k3.err; show $O/fine_k3.json
timeout 400 $B --steps 20 --warmup 5 > $O/fine_k6.json 2> $O/fine_k6.err; show $O/fine_k6.json
