#!/bin/bash
# round 6, GPU call 24: the tree matcher's event loop with its decisions in scalar registers - A/B against the build before it on the same box:
# tree-heavy cases alone (EH_PROF builds: slot 82 = the matcher), then the driver's command old / new / old / new
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06w; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $O/gputest.txt 2>&1; tail -2 $O/gputest.txt
for v in oldtree new; do
  lib=build/liberlamsa_hip_prof.so; [ $v = oldtree ] && lib=build/liberlamsa_hip_prof_oldtree.so
  ERLAMSA_HIP_LIB=$lib timeout 600 python tools/profile_alone.py 0 1814 24770 7692 64577 27694 26534 > $O/alone_$v.txt 2>&1
  echo "== $v"; grep "^case\|slot  82\|slot  71" $O/alone_$v.txt
done
for i in 1 2; do
  ERLAMSA_HIP_LIB=build/liberlamsa_hip_oldtree.so timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 > $O/bench_old_$i.json 2> $O/bench_old_$i.err; echo "old $(cut -c1-120 $O/bench_old_$i.json)"
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --pcie 0 --budget-mib 0 > $O/bench_new_$i.json 2> $O/bench_new_$i.err; echo "new $(cut -c1-120 $O/bench_new_$i.json)"
done
