#!/bin/bash
# round 4, GPU call 24: base64 chunks decoded by the whole wavefront; the longest case of the workload alone
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04u; mkdir -p $O
bash tools/gpu_probe.sh || exit 0
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 200 python tools/profile_alone.py 0 28038 43389 > $O/monsters.txt 2>&1; grep -v "slot 127" $O/monsters.txt | head -60
timeout 300 python tests/hipemu/emu_b64.py 40 3 20 > $O/b64.txt 2>&1; tail -3 $O/b64.txt
