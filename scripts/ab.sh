#!/bin/bash
# A/B timing of library variants on one box: scripts/ab.sh <reps> <lib1> <lib2> ...   (paths relative to delphy_amd/)
cd $GRAFT_REPO_ROOT
REPS=$1; shift
for r in $(seq $REPS); do
  for lib in "$@"; do
    EMAT_LIB_PATH=$GRAFT_REPO_ROOT/delphy_amd/$lib python bench.py --no-cpu-baseline --no-inclusive --steps 8 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('$lib', 'rep $r', round(d['value'] / 1e6, 1), 'M moves/s', round(d['ms_per_step'], 2), 'ms/step kernel', round(d['roofline']['kernel_ms'], 2))"
  done
done
