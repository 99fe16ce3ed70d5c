#!/bin/bash
# usage: pmc_case.sh LIB TAG
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
export ERLAMSA_HIP_LIB=$1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_FLAT SQ_WAIT_INST_ANY -f csv -d $R/gpurun_out/pmc_$2 -o p -- python $R/tools/profile_alone.py 0 30172 > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_SMEM SQ_IFETCH SQC_ICACHE_MISSES SQC_ICACHE_REQ -f csv -d $R/gpurun_out/pmc2_$2 -o p -- python $R/tools/profile_alone.py 0 30172 > /dev/null 2>&1
python - <<PY
import csv,glob
for d in ("pmc_$2","pmc2_$2"):
    for f in glob.glob("$R/gpurun_out/%s/**/*counter_collection.csv"%d, recursive=True):
        agg={}
        for r in csv.DictReader(open(f)):
            if "eh_mutate" in r["Kernel_Name"]:
                agg[r["Counter_Name"]]=agg.get(r["Counter_Name"],0)+float(r["Counter_Value"])
        print("$2", {k:("%.3g"%v) for k,v in agg.items()})
PY
