#!/bin/bash
# quick FETCH/WRITE + instruction counters for a bench variant.  Usage: scripts/pmc_quick.sh <tag> [bench args]
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcq_$TAG; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
timeout 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive "$@" > $OUT/b3.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -o pmc4 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive "$@" > $OUT/b4.log 2>&1
timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc1 -o pmc1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive "$@" > $OUT/b1.log 2>&1
python3 scripts/summarize_prof.py $OUT | python3 -c "
import sys, json
d = json.load(sys.stdin); p = d['pmc_k_run_moves_per_launch']
print('$TAG', {k: round(v / 7.86e6, 2) for k, v in p.items()}, 'per move; hbm bytes/launch', d.get('hbm_bytes_per_launch'))"
