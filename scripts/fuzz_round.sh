#!/bin/bash
# A round of the randomised sweeps under other seeds and under stress settings (run on the GPU box).  Usage: scripts/fuzz_round.sh <first seed> <cases>
cd $GRAFT_REPO_ROOT
S=${1:-500}; C=${2:-40}
K="test_randomised_scenarios_move_for_move or test_randomised_cycles_with_the_tree_in_hbm or test_randomised_global_move_statistics"
run() { echo "== $*"; env "$@" EMAT_FUZZ_CASES=$C timeout 1500 python -m pytest tests/test_parity_gpu.py tests/test_device_tree_gpu.py -x -q -m gpu -k "$K" 2>&1 | tail -3; }
run EMAT_FUZZ_SEED=$S
run EMAT_FUZZ_SEED=$((S+1)) EMAT_CHUNKS=3 EMAT_LDS_MAX=3072
run EMAT_FUZZ_SEED=$((S+2)) EMAT_CHUNKS=5 EMAT_LDS_MAX=2048 EMAT_SLACK=1.0 EMAT_HEAP_PER_NODE=0
run EMAT_FUZZ_SEED=$((S+3)) EMAT_TREE_TIGHT=1
run EMAT_FUZZ_SEED=$((S+4)) EMAT_LDS_CLASSES=50,90,100
