#!/bin/bash
# round 6, GPU call 27: longest-expected-first ticket order (rows whose cases were expensive before start first) against case order, and fewer lingering wavefronts; the driver's command, one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06z; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-110 $O/bench_$name.json)"; python - <<PY
import json; d=json.load(open("$O/bench_$name.json")); print("   kernel ms", d["roofline"]["kernel_ms_avg"], "parity", d.get("parity_checked"), d["case_stats"]["wave_cycles_per_pass"]["heaviest_case_Mcyc_mean_over_passes"])
PY
}
run lpt_1 X=1
run caseorder_1 EH_LPT=0
run lpt_linger16 EH_CO_LINGER=16
run lpt_2 X=1
run caseorder_2 EH_LPT=0
run lpt_linger0 EH_CO_LINGER=0
timeout 600 python bench.py --gpus 1 --steps 40 --warmup 8 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench_lpt_steps40.json 2> $O/bench_lpt_steps40.err; echo "steps40 $(cut -c1-110 $O/bench_lpt_steps40.json)"
