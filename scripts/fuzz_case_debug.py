"""Debugging aid: one case of tests/test_parity_gpu.py::test_randomised_global_move_statistics by seed and case number, part by part.
  python scripts/fuzz_case_debug.py <seed> <case>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
from helpers import random_scenario, split_parts, configure
from oracle_ffi import OracleEngine
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(want + 1):
    sc, nu_l, evo, what = random_scenario(rng, case)
    nparts = int(min(max(1, sc.tree.num_nodes // 24), rng.integers(1, 14)))
    split_seed = int(rng.integers(1, 10**6))
    if case != want:
        continue
    print(what, "| parts requested", nparts, "| pop", sc.pop.__dict__ if hasattr(sc.pop, "__dict__") else sc.pop, "| t_step", sc.default_t_step())
    parts, incl, seeds, root_part, ref = split_parts(sc, nparts, split_seed)
    gpu = d.EmatBackend(sc.num_sites); orc = OracleEngine(sc.num_sites)
    configure(gpu, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
    configure(orc, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
    for rounds in range(2):
        print("round", rounds, "totals gpu", gpu.totals(), "oracle", orc.totals())
        for p in range(len(parts)):
            n = parts[p].num_nodes
            g = gpu.part_derived(p, n); o = orc.part_derived(p, n)
            cg = gpu.part_coalescent(p); co = orc.part_coalescent(p)
            dk = float(np.max(np.abs(np.asarray(cg["k_bar_p"]) - np.asarray(co["k_bar_p"])))) if len(cg["k_bar_p"]) == len(co["k_bar_p"]) else -1
            print("  part %d%s nodes %d: log G %.10g / %.10g | prior %.17g / %.17g (diff %.6g) | cells %d / %d | max |dk_bar_p| %.3g" % (p, " (root)" if p == root_part else "", n, g[2], o[2], g[3], o[3], g[3] - o[3], len(cg["k_bar_p"]), len(co["k_bar_p"]), dk))
        gpu.run_moves_per_part(800); gpu.synchronize(); orc.run_moves_per_part(800, threads=4)
    print("after: totals gpu", gpu.totals(), "oracle", orc.totals())
    gpu.recalc_derived(); orc.recalc_derived()
    print("recomputed: totals gpu", gpu.totals(), "oracle", orc.totals())
    gpu.close(); orc.close()
