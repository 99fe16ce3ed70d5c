#!/bin/bash
# round 5, GPU call 8: seven passes in flight - how to split the memory between slots and the pool
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05h; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --steps 20 --warmup 5 --cpu-sample 0 --pcie 0 --budget-mib 0 "$@" > $O/$tag.json 2> $O/$tag.err
  python - $O/$tag.json $tag <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "K", r["config"]["passes_in_flight"], "ms/step", r["ms_per_step"], "MB/s", r["value"], "kernel_ms", r["roofline"]["kernel_ms_avg"], "arena_full", r["case_status"]["arena_full"], "waits", r["config"]["work_area_pool"]["waits"])
except Exception as ex:
    print("no result", sys.argv[2], ex)
PY
}
run a --inflight 7 --pool-gib 48 --max-slots 768
run b --inflight 7 --pool-gib 50 --max-slots 640
run c --inflight 7 --pool-gib 42 --max-slots 1024
nvidia-smi >/dev/null 2>&1; rocm-smi --showmeminfo vram 2>/dev/null | head -5
