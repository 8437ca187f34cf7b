"""Debugging aid: one case of tests/test_parity_gpu.py::test_randomised_global_move_statistics by seed and case number, with move traces: the first
move of every part at which the HIP engine and the oracle differ.
  python scripts/fuzz_case_trace.py <seed> <case> [rounds=2] [moves=800]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
from helpers import random_scenario, split_parts, configure
from oracle_ffi import OracleEngine
seed, want = int(sys.argv[1]), int(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 2
moves = int(sys.argv[4]) if len(sys.argv) > 4 else 800
rng = np.random.default_rng(seed)
for case in range(want + 1):
    sc, nu_l, evo, what = random_scenario(rng, case)
    nparts = int(min(max(1, sc.tree.num_nodes // 24), rng.integers(1, 14)))
    split_seed = int(rng.integers(1, 10**6))
    if case != want:
        continue
    print(what)
    parts, incl, seeds, root_part, ref = split_parts(sc, nparts, split_seed)
    T = rounds * moves
    gpu = d.EmatBackend(sc.num_sites, trace_moves=T); orc = OracleEngine(sc.num_sites, trace_moves=T)
    configure(gpu, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
    configure(orc, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
    for r in range(rounds):
        gpu.run_moves_per_part(moves); gpu.synchronize(); orc.run_moves_per_part(moves, threads=4)
    for p in range(len(parts)):
        tg, to = gpu.part_trace(p, T), orc.part_trace(p, T)
        n = min(len(tg), len(to))
        bad = [i for i in range(n) if not (tg[i][0] == to[i][0] and tg[i][1] == to[i][1] and tg[i][2] == to[i][2])]
        worst = max((abs(tg[i][3] - to[i][3]) / max(1.0, abs(to[i][3])) for i in range(bad[0] if bad else n) if np.isfinite(tg[i][3]) and np.isfinite(to[i][3])), default=0.0)
        print("part %d%s: %d moves traced, first differing move %s, worst relative log_mh difference before it %.3g" % (p, " (root)" if p == root_part else "", n, bad[0] if bad else None, worst))
        if bad:
            for i in range(max(0, bad[0] - 3), min(n, bad[0] + 2)):
                print("    move %d: gpu %s | oracle %s" % (i, [float(x) for x in tg[i]], [float(x) for x in to[i]]))
    gpu.close(); orc.close()
