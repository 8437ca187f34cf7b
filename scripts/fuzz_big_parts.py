"""Hand-run hunt: the randomised move-for-move sweep of tests/test_parity_gpu.py on LARGER trees cut into FEW parts (parts of hundreds to thousands of nodes:
prefix staging, lists in HBM, long candidate scans), which the suite's sweeps (trees of up to 320 tips) do not reach.
  python scripts/fuzz_big_parts.py SEED CASES [MAX_TIPS=3000] [MOVES=1500] [FIRST=0: skip the cases before this one]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from delphy_amd.scenarios import random_scenario
from helpers import run_parity
seed, cases = int(sys.argv[1]), int(sys.argv[2])
max_tips = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
moves = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
first = int(sys.argv[5]) if len(sys.argv) > 5 else 0
rng = np.random.default_rng(seed)
bad = 0
for case in range(cases):
    sc, nu_l, evo, what = random_scenario(rng, case, max_tips=max_tips)
    nparts = int(rng.integers(1, 14))
    t_step = sc.default_t_step() * float(rng.choice([0.25, 1.0, 4.0]))
    split_seed = int(rng.integers(1, 10**6))
    if case < first:
        continue
    try:
        run_parity(sc, nparts, moves, seed=split_seed, trace=moves, use_lds=bool(case % 5), t_step=t_step, nu_l=nu_l, evo=evo)
        print("ok  ", what, "| parts", nparts, flush=True)
    except Exception as ex:
        bad += 1
        print("FAIL", what, "| parts", nparts, "t_step %g seed %d:" % (t_step, split_seed), str(ex)[:400], flush=True)
print("failures:", bad)
