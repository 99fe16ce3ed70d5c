#!/bin/bash
# round 5, GPU call 2: the heaviest cases of the passes the bench runs, then each of the worst alone with the EH_PROF build
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r05b; mkdir -p $O
timeout 400 python tools/r05_monsters.py 0 25 4 $O/monsters.json > $O/passes.txt 2>&1; tail -130 $O/passes.txt | grep -v "^    " | head -30
python - $O/monsters.json > $O/worst.txt <<'PY'
import json, sys
r = sorted(json.load(open(sys.argv[1])), key=lambda x: -x["mcyc"])[:16]
for x in r: print(x["pass"] * 65536, x["row"])
PY
ERLAMSA_HIP_LIB=build/liberlamsa_hip_prof.so timeout 300 python tools/profile_alone.py @$O/worst.txt > $O/alone.txt 2>&1
grep -c alone $O/alone.txt
