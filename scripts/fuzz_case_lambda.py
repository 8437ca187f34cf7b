"""Debugging aid: one case of tests/test_parity_gpu.py::test_randomised_global_move_statistics, move by move: the first move after which a part's
lambda_i (maintained incrementally by both sides) is not the same on the HIP engine and on the oracle BIT FOR BIT.
  python scripts/fuzz_case_lambda.py <seed> <case> <part> [moves=900]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
from helpers import random_scenario, split_parts, configure
from oracle_ffi import OracleEngine
seed, want, part = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
moves = int(sys.argv[4]) if len(sys.argv) > 4 else 900
rng = np.random.default_rng(seed)
for case in range(want + 1):
    sc, nu_l, evo, what = random_scenario(rng, case)
    nparts = int(min(max(1, sc.tree.num_nodes // 24), rng.integers(1, 14)))
    split_seed = int(rng.integers(1, 10**6))
print(what)
parts, incl, seeds, root_part, ref = split_parts(sc, nparts, split_seed)
gpu = d.EmatBackend(sc.num_sites, trace_moves=moves); orc = OracleEngine(sc.num_sites, trace_moves=moves)
configure(gpu, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
configure(orc, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
n = parts[part].num_nodes
shown = 0
for m in range(moves + 1):
    lg = np.asarray(gpu.part_derived(part, n)[0]); lo = np.asarray(orc.part_derived(part, n)[0])
    bad = np.nonzero(lg.view(np.uint64) != lo.view(np.uint64))[0]
    if len(bad):
        tr = orc.part_trace(part, moves)
        print("after %d moves: lambda_i differs at nodes %s (gpu - oracle: %s); the move before: %s" % (m, bad.tolist(), (lg[bad] - lo[bad]).tolist(), [float(x) for x in tr[m - 1]] if m else None))
        shown += 1
        if shown >= 4:
            break
    gpu.run_moves_per_part(1); gpu.synchronize(); orc.run_moves_per_part(1, threads=1)
else:
    print("lambda_i bit-identical through %d moves" % moves)
gpu.close(); orc.close()
