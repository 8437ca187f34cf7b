"""Host work of whole cycles: one backend against n shards of ONE process on device 0 (emat_multi, host exchange).  With all shards on one
GPU the kernels of the shards run one after the other, so wall time means nothing; what is compared is the host's named spans
(EMAT_VERBOSE=spans, summed over all handles of the process at exit):
  EMAT_VERBOSE=spans python scripts/multi_probe.py single [cycles=20]
  EMAT_VERBOSE=spans python scripts/multi_probe.py multi [cycles=20] [shards=8]"""
import sys, os, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
mode = sys.argv[1] if len(sys.argv) > 1 else "single"
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 20
shards = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sc = make_scenario("C4")
per = 50 * sc.tree.num_nodes
if mode == "single":
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, 20261001)
    run.set_num_parts(8192); run.set_max_part_nodes(-1); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop); run.set_device_tree(True)
    run.do_mcmc_steps(per, per)
    t0 = time.perf_counter(); run.do_mcmc_steps(cycles * per, per); dt = time.perf_counter() - t0
    print("single backend: %.2f ms per cycle over %d cycles" % (dt / cycles * 1e3, cycles))
    run.close(); b.close()
else:
    m = d.EmatMultiRun([0] * shards, sc.tree, sc.ref, 20261001, exchange="host")
    m.set_num_parts(8192); m._ck(m._lib.emat_multi_set_max_part_nodes(m._h, -1), "set_max_part_nodes"); m.set_hky(sc.mu, sc.kappa, sc.pi); m.set_pop_model(sc.pop)
    m.do_mcmc_steps(per, per)
    t0 = time.perf_counter(); m.do_mcmc_steps(cycles * per, per); dt = time.perf_counter() - t0
    print("%d shards on device 0 (host exchange): %.2f ms per cycle over %d cycles" % (shards, dt / cycles * 1e3, cycles))
    m.close()
