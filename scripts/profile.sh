#!/bin/bash
# rocprofv3 passes for bench.py (run on the GPU box through gpurun).  Usage: scripts/profile.sh <tag> [bench args...]
# Pass 1: kernel trace + stats (per-kernel durations).  Passes 2-5: PMC counters, each in its own run
# (never combined with trace domains).  Summaries land in gpurun_out/prof_<tag>/; copy what should be judged to profiles/.
set -u
TAG=${1:-r01}; shift || true
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-inclusive --no-decompositions --secondary '' "$@" > $OUT/bench_trace.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS --output-format csv -d $OUT/pmc1 -o pmc1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive --no-decompositions --secondary '' "$@" > $OUT/bench_pmc1.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/pmc2 -o pmc2 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive --no-decompositions --secondary '' "$@" > $OUT/bench_pmc2.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -o pmc3 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive --no-decompositions --secondary '' "$@" > $OUT/bench_pmc3.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc4 -o pmc4 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-inclusive --no-decompositions --secondary '' "$@" > $OUT/bench_pmc4.log 2>&1
python3 scripts/summarize_prof.py $OUT
