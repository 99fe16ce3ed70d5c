#!/bin/bash
# The round's last GPU call, second edition (commit 38990f2 + docs): box probe, the GPU tests, the smoke run, the driver's exact
# bench command, the profiles of that build (tools/profile_round.sh: kernel trace + stats under the bench's own command, four
# PMC passes).  The other BASELINE configurations (C2 / C4 / C5, 4 hardware queues) were profiled on commit 9fa0133
# (tools/r04_final.sh); there was no GPU time left to repeat them.
#   gpurun --timeout 840 -- 'bash tools/r04_final2.sh'
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04z2; mkdir -p $O
bash tools/gpu_probe.sh || exit 0
timeout 600 python -m pytest tests -q -m gpu > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $R/gpurun_out/bench_r04.log 2> $R/gpurun_out/bench_r04.err; cut -c1-220 $R/gpurun_out/bench_r04.log; grep -c "child" $R/gpurun_out/bench_r04.err
bash tools/profile_round.sh r04 2>&1 | tail -12
