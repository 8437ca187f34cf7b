import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
sc = make_scenario("C4")
b = d.EmatBackend(sc.num_sites)
run = d.EmatRun(b, sc.tree, sc.ref, 20261001)
run.set_num_parts(8192); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop); run.set_device_tree(True)
per = 50 * sc.tree.num_nodes
run.do_mcmc_steps(per, per)
os.environ["EMAT_VERBOSE"] = "1"
t0 = time.perf_counter(); run.do_mcmc_steps(3 * per, per); print("ms per cycle", (time.perf_counter() - t0) / 3 * 1e3)
