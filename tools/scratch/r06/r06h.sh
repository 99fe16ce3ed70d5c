#!/bin/bash
# round 6, GPU call 8: stream priorities against the convoy of passes launched together; workgroups per pass; configs[1] with the totals summed on the device
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06h; mkdir -p $O
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
timeout 400 $B --steps 20 --warmup 5 > $O/base_k6.json 2> $O/base_k6.err; show $O/base_k6.json
EH_STREAM_PRIO=rr timeout 400 $B --steps 20 --warmup 5 > $O/prio_k6.json 2> $O/prio_k6.err; show $O/prio_k6.json
EH_STREAM_PRIO=rr timeout 400 $B --steps 20 --warmup 5 --max-slots 2048 --pool-gib 52 > $O/prio_slots2048_k6.json 2> $O/prio_slots2048_k6.err; show $O/prio_slots2048_k6.json
timeout 400 $B --steps 20 --warmup 5 --max-slots 2048 --pool-gib 52 > $O/slots2048_k6.json 2> $O/slots2048_k6.err; show $O/slots2048_k6.json
timeout 400 $B --steps 40 --warmup 5 > $O/base_k6_40steps.json 2> $O/base_k6_40steps.err; show $O/base_k6_40steps.json
C2="--cases 1024 --size 256 --corpus uniform --mutations bd,bf,bi --patterns od --case-stats 0"
timeout 300 $B $C2 --inflight 1 --steps 400 --warmup 20 > $O/c2_k1.json 2> $O/c2_k1.err; show $O/c2_k1.json
timeout 300 $B $C2 --inflight 6 --steps 1200 --warmup 60 > $O/c2_k6.json 2> $O/c2_k6.err; show $O/c2_k6.json
