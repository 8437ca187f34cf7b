#!/bin/bash
# where the L2's writes to memory come from (k_run_moves): write requests into the L2, write-backs by kind, evictions.  Usage: scripts/pmc_writes.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcw_$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive --secondary="
timeout 240 rocprofv3 --pmc TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum TCC_WRITE_sum TCC_WRITEBACK_sum --output-format csv -d $OUT/p1 -o p1 -- $B "$@" > $OUT/b1.log 2>&1
timeout 240 rocprofv3 --pmc TCC_NORMAL_WRITEBACK_sum TCC_ALL_TC_OP_WB_WRITEBACK_sum TCC_NORMAL_EVICT_sum TCC_REQ_sum --output-format csv -d $OUT/p2 -o p2 -- $B "$@" > $OUT/b2.log 2>&1
timeout 240 rocprofv3 --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --output-format csv -d $OUT/p3 -o p3 -- $B "$@" > $OUT/b3.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]; acc = {}; other = {}
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1]
        d = acc if k == "k_run_moves" else other
        key = r["Counter_Name"] if d is acc else (k, r["Counter_Name"])
        d[key] = d.get(key, 0.0) + float(r["Counter_Value"])
for k in sorted(acc): print("%-36s %16.0f  per move %10.2f" % (k, acc[k], acc[k] / 8.094e6))
for k in sorted(other):
    if other[k] > 1e6: print("  other kernel %-50s %14.0f" % (k, other[k]))
PY
tail -3 $OUT/b1.log | cut -c1-300
