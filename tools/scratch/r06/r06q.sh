#!/bin/bash
# round 6, GPU call 17: the driver's command with the new defaults (seven passes in flight, 42 GiB pool), all legs (oracle / parity, PCIe, budget)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06q; mkdir -p $O
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-200 $O/bench.json; grep -i "error\|Traceback\|killed\|repeating\|Elapsed" $O/bench.err | head
python - <<'PY'
import json
r=json.loads(open("gpurun_out/r06q/bench.json").read().strip().splitlines()[-1])
print({k:r.get(k) for k in ("value","cases_per_s","ms_per_step","parity_checked","cases_per_s_le_64KiB")}, r.get("parity",{}).get("not_compared"), r["config"]["passes_in_flight"], r.get("pcie",{}).get("pipelined"), list(r.get("with_work_budget",{}).items())[:3], r.get("extras_timed_out"))
PY
rocm-smi --showmeminfo vram | head -8
