#!/bin/bash
# instruction-fetch, scalar-cache and latency counters of k_run_moves (one rocprofv3 --pmc pass per group).  Usage: scripts/pmc_extra.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcx_$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive"
timeout 240 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL --output-format csv -d $OUT/p1 -o p1 -- $B "$@" > $OUT/b1.log 2>&1
timeout 240 rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL --output-format csv -d $OUT/p2 -o p2 -- $B "$@" > $OUT/b2.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p3 -o p3 -- $B "$@" > $OUT/b3.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p4 -o p4 -- $B "$@" > $OUT/b4.log 2>&1
timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/p5 -o p5 -- $B "$@" > $OUT/b5.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]; acc = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].split("(")[0].endswith("k_run_moves"):
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(acc): print("%-28s %16.0f  per move %10.2f" % (k, acc[k], acc[k] / 7.955e6))
PY
