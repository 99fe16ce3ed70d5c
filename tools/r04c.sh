#!/bin/bash
# round 4, third GPU call: the LDS-resident fuse on the device - differential check, per-call cost, bench, survey
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04c; mkdir -p $O
timeout 300 python tests/hipemu/emu_fuse_lds.py 12 5 > $O/diff.txt 2>&1; tail -3 $O/diff.txt
for sz in 1024 4096 8192; do for m in ft fn; do
  ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 100 python tools/profile_fuse.py $sz 64 $m >> $O/fuse_lds.txt 2>&1
  FUSE_NO_LDS=1 ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 100 python tools/profile_fuse.py $sz 64 $m >> $O/fuse_nodes.txt 2>&1
done; done
grep -h "size" $O/fuse_lds.txt | head -20; echo ---; grep -h "size" $O/fuse_nodes.txt | head -20
timeout 300 python -m pytest tests -q -m gpu -x -k "fuse or bench_workload or default_tables or golden" > $O/gputest.txt 2>&1; tail -3 $O/gputest.txt
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; cut -c1-200 $O/bench.log; grep "timed steps done" $O/bench.err
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/survey_pass.py r04c > $O/survey.txt 2>&1; grep -A16 "fuse_lists calls" $O/survey.txt
