#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box through gpurun). Usage: scripts/profile.sh <tag> [bench args...]
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS -d $OUT/pmc1 -o pmc1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $OUT/pmc2 -o pmc2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc4 -o pmc4 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline "$@" > $OUT/bench_pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
python3 scripts/summarize_prof.py $OUT
