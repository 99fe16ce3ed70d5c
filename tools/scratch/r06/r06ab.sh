#!/bin/bash
# round 6, GPU call 29: 24 lingering wavefronts per pass as the default + the wave-slot accounting (eh_result_occupancy) in the bench line; the driver's command, EH_CO_LINGER=48 beside it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_round6.py -x -q -m gpu > $O/gputest.txt 2>&1; tail -2 $O/gputest.txt
run() { name=$1; shift; env "$@" timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench_$name.json 2> $O/bench_$name.err; echo "$name $(cut -c1-110 $O/bench_$name.json)"; python - <<PY
import json; d=json.load(open("$O/bench_$name.json")); print("   ", d.get("wave_slots"))
PY
}
run default_1 X=1
run l48_1 EH_CO_LINGER=48
run default_2 X=1
run l48_2 EH_CO_LINGER=48
timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --inflight 1 --pcie 0 --budget-mib 0 --cpu-sample 0 > $O/bench_inflight1.json 2> $O/bench_inflight1.err; echo "inflight1 $(cut -c1-110 $O/bench_inflight1.json)"; python -c "import json; print(json.load(open(\"$O/bench_inflight1.json\")).get(\"wave_slots\"))"
