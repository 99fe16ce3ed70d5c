#!/bin/bash
# round 6, last GPU call: the tree as committed - GPU suite, smoke, the driver's command
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r06ai; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu > $O/gputest.txt 2>&1; tail -2 $O/gputest.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-160 $O/bench.json
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['parity_checked'], d['roofline']['traffic'], d['roofline']['frac'], d['wave_slots']['held'], d['cpu_baseline']['value'])"
