#!/bin/bash
# round 5, GPU call 14: with steps going to whichever context is free - seven or eight passes in flight?
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05l; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --pcie 0 --budget-mib 0 "$@" > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "K", r["config"]["passes_in_flight"], "ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "arena_full", r["case_status"]["arena_full"], "waits", r["config"]["work_area_pool"]["waits"])
except Exception as ex:
    print("no result", sys.argv[2], ex)
PY
}
run k7a --inflight 7 --pool-gib 54
run k7b --inflight 7 --pool-gib 46 --max-slots 768
run k8 --inflight 8 --pool-gib 40 --max-slots 768 --out-gib 27
