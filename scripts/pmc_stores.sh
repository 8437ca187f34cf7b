#!/bin/bash
# where the stores of k_run_moves go: FLAT instructions that touch only LDS, L1->L2 write requests, L2 write-backs.  Usage: scripts/pmc_stores.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcs_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive"
timeout 240 rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $OUT/p1 -o p1 -- $B "$@" > $OUT/b1.log 2>&1
timeout 240 rocprofv3 --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --output-format csv -d $OUT/p2 -o p2 -- $B "$@" > $OUT/b2.log 2>&1
timeout 240 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_WRITEBACK_sum TCC_WRITE_sum --output-format csv -d $OUT/p3 -o p3 -- $B "$@" > $OUT/b3.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]; acc = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].split("(")[0].endswith("k_run_moves"):
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k in sorted(acc): print("%-36s %16.0f  per move %10.2f" % (k, acc[k], acc[k] / 7.955e6))
for i in range(1, 4):
    t = open(os.path.join(out, "b%d.log" % i)).read()
    if '"metric"' not in t: print("pass", i, "failed:", t[-400:])
PY
