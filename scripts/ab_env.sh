#!/bin/bash
# A/B of (library, environment) pairs on one box: scripts/ab_env.sh <reps> "<lib> [VAR=value ...]" ...   (library paths relative to delphy_amd/)
cd $GRAFT_REPO_ROOT
REPS=$1; shift
SPECS=("$@")
for r in $(seq $REPS); do
  for spec in "${SPECS[@]}"; do
    words=($spec); lib=${words[0]}
    env "${words[@]:1}" EMAT_LIB_PATH=$GRAFT_REPO_ROOT/delphy_amd/$lib python bench.py --no-cpu-baseline --no-inclusive --no-decompositions --secondary '' --steps 10 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$spec', '| rep $r', round(d['value'] / 1e6, 1), 'M moves/s', round(d['ms_per_step'], 2), 'ms/step kernel', round(d['roofline']['kernel_ms'], 2))"
  done
done
