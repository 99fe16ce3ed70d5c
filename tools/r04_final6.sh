#!/bin/bash
# the rest of the GPU suite on the build that ships (tools/r04_final5.sh ran the tests that read the constant tables), as far as the round's last GPU seconds reach
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r04z6; mkdir -p $O
timeout 138 python -m pytest tests -q -m gpu -k "not (golden or default_tables or bench_workload_full or utf or csum or container or zlib or primitives or bench_script or nccl)" > $O/t.txt 2>&1; tail -3 $O/t.txt
