"""Whole cycles with the tree on the HOST (the adaptor of INTEGRATION sections 2-3: Run keeps its Phylo_tree), by phase.
  EMAT_VERBOSE=1 python scripts/host_tree_probe.py [cycles=6] [max_part_nodes=-1]     (reports per call on stderr)
  EMAT_VERBOSE=spans ...                                                             (host spans summed at exit)"""
import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 6
limit = int(sys.argv[2]) if len(sys.argv) > 2 else -1
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0
if threads:
    d.load_library().emat_set_host_threads(threads)
sc = make_scenario("C4")
b = d.EmatBackend(sc.num_sites)
run = d.EmatRun(b, sc.tree, sc.ref, 20261001)
run.set_num_parts(8192); run.set_max_part_nodes(limit); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
per = 50 * sc.tree.num_nodes
run.do_mcmc_steps(per, per)
t0 = time.perf_counter()
for c in range(cycles):
    t1 = time.perf_counter()
    run.repartition(); t2 = time.perf_counter()
    run.run_moves(per); b.synchronize(); t3 = time.perf_counter()
    run.reassemble(); t4 = time.perf_counter()
    print("cycle %d: repartition %.1f ms | moves %.1f ms | reassemble %.1f ms | parts %d" % (c, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, run.num_parts()[0]), flush=True)
print("ms per cycle", (time.perf_counter() - t0) / cycles * 1e3)
run.close(); b.close()
