#!/bin/bash
# round 4, fourth GPU call: the periodic-stretch reduction of large fuse calls on the device
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r04d; mkdir -p $O
timeout 200 python tests/hipemu/emu_fuse_red.py 3 7 1 > $O/diff1.txt 2>&1; tail -2 $O/diff1.txt
timeout 200 python tests/hipemu/emu_fuse_lds.py 8 9 > $O/diff_lds.txt 2>&1; tail -2 $O/diff_lds.txt
timeout 300 python - > $O/diff_big.txt 2>&1 <<'PY'
import sys
sys.path.insert(0, "tests/hipemu")
import emu_fuse_red as m
print(m.run(2, 11, big=8, with_oracle=False))
PY
tail -2 $O/diff_big.txt
for sz in 65536 524288; do for m in ft fn; do
  ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 100 python tools/profile_fuse.py $sz 64 $m >> $O/fuse_cut.txt 2>&1
  FUSE_NO_REDUCE=1 ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 100 python tools/profile_fuse.py $sz 64 $m >> $O/fuse_plain.txt 2>&1
done; done
grep -h -A1 "size" $O/fuse_cut.txt | head -30; echo ---; grep -h "size" $O/fuse_plain.txt | head -20
for sz in 4096; do for m in ft fn; do ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 100 python tools/profile_fuse.py $sz 64 $m >> $O/fuse_lds.txt 2>&1; done; done
grep -h -A1 "size" $O/fuse_lds.txt | head -14
timeout 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.log 2> $O/bench.err; cut -c1-200 $O/bench.log; grep "timed steps done" $O/bench.err
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/survey_pass.py r04d > $O/survey.txt 2>&1; head -3 $O/survey.txt; grep -A16 "fuse_lists calls" $O/survey.txt; grep "ft \|fn \|sgm \|b64 \|ts1 \|ts2 \|tr  " $O/survey.txt
