#!/bin/bash
# A/B of (library, environment) pairs on one box: scripts/ab_paths.sh <reps> "<lib path relative to the repo root> [VAR=value ...]" ...
cd $GRAFT_REPO_ROOT
REPS=$1; shift
SPECS=("$@")
for r in $(seq $REPS); do
  for spec in "${SPECS[@]}"; do
    words=($spec); lib=${words[0]}
    env "${words[@]:1}" EMAT_LIB_PATH=$GRAFT_REPO_ROOT/$lib EMAT_ALLOW_STALE_LIB=1 python bench.py --no-cpu-baseline --no-inclusive --no-decompositions --secondary '' --steps 10 $EMAT_AB_ARGS 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$spec', '| rep $r', round(d['value'] / 1e6, 1), 'M moves/s', round(d['ms_per_step'], 2), 'ms/step kernel', round(d['roofline']['kernel_ms'], 2))"
  done
done
