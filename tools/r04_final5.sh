#!/bin/bash
# the build with compile-time constant tables (no hipMemcpyToSymbol in eh_create; same instruction stream, tools/same_kernel_text.py): smoke, the driver's command, GPU tests that read every table
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04z5; mkdir -p $O
bash tools/gpu_probe.sh || exit 0
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/bench_r04.log 2> $R/gpurun_out/bench_r04.err; cut -c1-220 $R/gpurun_out/bench_r04.log; grep -c "child" $R/gpurun_out/bench_r04.err; grep -o '"traffic": [0-9.e+]*' $R/gpurun_out/bench_r04.log
timeout 120 python -m pytest tests -q -m gpu -k "golden or default_tables or bench_workload_full or utf or csum or container or zlib or primitives" > $O/t.txt 2>&1; tail -2 $O/t.txt
