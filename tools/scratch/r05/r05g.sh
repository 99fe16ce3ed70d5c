#!/bin/bash
# round 5, GPU call 7: more passes in flight within the same memory?  (step = b + T / K)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05g; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --steps 21 --warmup 7 --cpu-sample 0 --pcie 0 --budget-mib 0 "$@" > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "K", r["config"]["passes_in_flight"], "ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "status", r["case_status"], "waits", r["config"]["work_area_pool"]["waits"], r.get("warning"))
except Exception as ex:
    print("no result", sys.argv[2], ex)
PY
}
run k6 --inflight 6
run k7 --inflight 7 --pool-gib 42 --out-gib 27
run k8 --inflight 8 --pool-gib 36 --out-gib 26 --max-slots 768
