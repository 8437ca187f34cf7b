#!/usr/bin/env python3
"""Soak test on the GPU: many repartition -> moves -> reassemble cycles of the C++ run driver on a large tree, then the
whole tree is checked with the oracle's restatement of the reference's tree-integrity rules (phylo_tree.cpp:18-126) and
the engine's incremental log-posterior totals are compared with a from-scratch evaluation after every cycle.
Usage: scripts/stress_cycles.py [workload=C4] [cycles=20] [parts=8192] [max_part_nodes=100] [device_tree=0]
(EMAT_SLACK=1.05 EMAT_HEAP_PER_NODE=8 in the environment starves the list heaps, so that the out-of-space recoveries run too.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import configure
from oracle_ffi import OracleEngine

workload = sys.argv[1] if len(sys.argv) > 1 else "C4"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 20
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
limit = int(sys.argv[4]) if len(sys.argv) > 4 else 100
device_tree = len(sys.argv) > 5 and sys.argv[5] == "1"
sc = make_scenario(workload)
b = d.EmatBackend(sc.num_sites)
run = d.EmatRun(b, sc.tree, sc.ref, 777)
run.set_num_parts(parts); run.set_max_part_nodes(limit)
run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
if device_tree:
    run.set_device_tree(True)
t0 = time.perf_counter()
for cyc in range(cycles):
    run.repartition(); n, _ = run.num_parts()
    if not device_tree:
        run.push_params()
    run.run_moves(n * 1000); b.synchronize()
    g_inc, a_inc = b.totals()
    b.recalc_derived(); g_new, a_new = b.totals()
    assert abs(g_inc - g_new) <= 1e-7 * max(1.0, abs(g_new)) and abs(a_inc - a_new) <= 1e-7 * max(1.0, abs(a_new)), (cyc, g_inc, g_new, a_inc, a_new)
    stopped = sum(1 for p in range(n) if b.part_stats(p)["status"] != 0)
    assert stopped == 0
    run.reassemble()
    print("cycle %2d: %d parts, log_G %.3f, incremental vs from-scratch diff %.2e / %.2e, %.1f s" % (cyc, n, g_new, abs(g_inc - g_new), abs(a_inc - a_new), time.perf_counter() - t0), flush=True)
tree, ref = run.tree()
chk = OracleEngine(sc.num_sites)
sc.tree, sc.ref = tree, ref
configure(chk, sc, ref, [tree], [True], [1], 0)
rc, msg = chk.part_check(0)
assert rc == 0, msg
tips = tree.child0 == -1
assert tree.num_nodes == 2 * int(tips.sum()) - 1
print("whole tree after %d cycles: %d nodes, %d mutations, integrity OK" % (cycles, tree.num_nodes, len(tree.mut_site)))
chk.close(); run.close(); b.close()
