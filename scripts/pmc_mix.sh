#!/bin/bash
# instruction mix and wait profile of k_run_moves per move.  Usage: scripts/pmc_mix.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/mix_$TAG; rm -rf $OUT; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/a -o a -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive "$@" > $OUT/a.log 2>&1
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_BRANCH TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum SQC_ICACHE_MISSES SQ_INSTS_SMEM --output-format csv -d $OUT/b -o b -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive "$@" > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections, json, re
acc = collections.defaultdict(float)
for tag in "ab":
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_run_moves" in r["Kernel_Name"]: acc[r["Counter_Name"]] += float(r["Counter_Value"])
line = [l for l in open("$OUT/a.log") if l.startswith("{")][-1]
d = json.loads(line); moves = 2 * int(re.search(r"(\d+) partition parts", d["config"]["workload"]).group(1)) * 1000
print("$TAG", {k: round(v / moves, 1) for k, v in sorted(acc.items())}, "value", round(d["value"] / 1e6, 1))
PY
