#!/bin/bash
# round 4, GPU call 26: lane batches with the next-stop table, search caps and back-off: differentials, heaviest cases, bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04w; mkdir -p $O
bash tools/gpu_probe.sh || exit 0
timeout 300 python tests/hipemu/emu_sgml_replay.py 6 2 3 > $O/sgml_diff.txt 2>&1; tail -4 $O/sgml_diff.txt
timeout 300 python tests/hipemu/emu_sgml_replay.py 4 7 1 > $O/sgml_diff2.txt 2>&1; tail -2 $O/sgml_diff2.txt
timeout 400 python -m pytest tests -q -m gpu -x -k "sgml or bench_workload_full or default_tables or golden or b64 or meta_trace or adversarial" > $O/t2.txt 2>&1; tail -3 $O/t2.txt
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/profile_alone.py 0 40457 50785 63042 36457 44522 64577 13337 7692 32620 52508 > $O/monsters.txt 2>&1; grep -v "slot 127\|slot  66" $O/monsters.txt | grep "alone\|calls\|phase 2" | cut -c1-150
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --budget-mib 0 --pcie 0 > $O/bench.log 2> $O/bench.err; cut -c1-160 $O/bench.log; grep -o '"kernel_ms_avg": [0-9.]*' $O/bench.log
