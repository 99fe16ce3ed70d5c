#!/bin/bash
# round 4, sixth GPU call: is a step bound by passes-in-flight / pass-duration or by the device?  (+ the cheaper fuse paths)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04f; mkdir -p $O
B="--cpu-sample 0 --budget-mib 0 --pcie 0"
timeout 100 python tests/hipemu/emu_fuse_lds.py 6 13 > $O/diff_lds.txt 2>&1; tail -1 $O/diff_lds.txt
timeout 100 python tests/hipemu/emu_fuse_red.py 2 17 1 > $O/diff_red.txt 2>&1; tail -1 $O/diff_red.txt
for m in ft fn; do ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 100 python tools/profile_fuse.py 4096 64 $m >> $O/fuse_lds.txt 2>&1; done
for m in ft; do ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 100 python tools/profile_fuse.py 524288 64 $m >> $O/fuse_big.txt 2>&1; done
grep -h -A1 "size" $O/fuse_lds.txt $O/fuse_big.txt | head -30
timeout 150 python3 bench.py --gpus 1 --steps 20 --warmup 5 $B > $O/b_std.log 2> $O/b_std.err; cut -c1-140 $O/b_std.log
timeout 150 python3 bench.py --gpus 1 --steps 40 --warmup 10 --cases 32768 --inflight 12 --out-gib 14 --max-slots 512 $B > $O/b_half12.log 2> $O/b_half12.err; cut -c1-140 $O/b_half12.log
timeout 150 python3 bench.py --gpus 1 --steps 20 --warmup 5 --max-slots 2048 --pool-gib 40 $B > $O/b_2048.log 2> $O/b_2048.err; cut -c1-140 $O/b_2048.log
timeout 150 python3 bench.py --gpus 1 --steps 20 --warmup 5 --inflight 7 --out-gib 26 --pool-gib 44 $B > $O/b_7.log 2> $O/b_7.err; cut -c1-140 $O/b_7.log
