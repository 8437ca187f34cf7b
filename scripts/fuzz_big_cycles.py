"""Hand-run hunt: tests/test_device_tree_gpu.py::test_randomised_cycles_with_the_tree_in_hbm_equal_the_host_cycles on LARGER trees, MANY parts and with the
part-size limit on: whole cycles with the tree resident in HBM (partition, slabs and gather by kernels) against the same cycles with the tree on the host
(themselves checked against the oracle by the suite): same seeds => the same tree and reference sequence, bit for bit, after every cycle.
  python scripts/fuzz_big_cycles.py SEED CASES [MAX_TIPS=20000] [CYCLES=3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
import delphy_amd.engine as e
from delphy_amd.scenarios import Scenario, KAPPA, PI
from test_device_tree_gpu import _run, _same_tree
seed0, cases = int(sys.argv[1]), int(sys.argv[2])
max_tips = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
cycles = int(sys.argv[4]) if len(sys.argv) > 4 else 3
rng = np.random.default_rng(seed0)
bad = 0
for case in range(cases):
    tips = int(10 ** rng.uniform(2.0, np.log10(max_tips)))
    sites = int(rng.choice([80, 500, 3000, 29903]))
    span = float(rng.choice([60.0, 365.0, 900.0]))
    mu = float(10 ** rng.uniform(-3.4, -2.3)) / 365.0 * (30000.0 / max(sites, 300)) ** 0.5
    par = e.SynthParams(num_tips=tips, num_sites=sites, tip_span=span, pop_n0=float(10 ** rng.uniform(1.5, 3.0)), pop_growth=float(rng.choice([0.0, 2.0])) / 365.0,
                        mu=mu, gaps_per_tip=int(rng.integers(0, 4)), mean_gap_len=float(max(2.0, sites * 10 ** rng.uniform(-2.5, -1.0))), seed=int(rng.integers(1, 2**31)))
    par.pi, par.kappa = PI, KAPPA
    if rng.random() < 0.5:
        par.frac_uncertain_tips, par.tip_date_uncertainty = float(rng.uniform(0.05, 0.5)), float(rng.uniform(0.5, 10.0))
    tree, ref, tmax = e.make_synthetic_emat(par)
    while tree.mut_site.shape[0] > 30 * tips:
        mu /= 4.0; par.mu = mu
        tree, ref, tmax = e.make_synthetic_emat(par)
    kind = case % 3
    if kind == 0:
        pop = d.PopModel.exp(tmax, par.pop_n0, 0.0, 0.0)
    elif kind == 1:
        pop = d.PopModel.exp(tmax, par.pop_n0, float(rng.uniform(0.2, 3.0)) / 365.0, 1.0)
    else:
        x = np.unique(np.append(np.sort(tmax - span * 1.3 * rng.uniform(0.0, 1.0, int(rng.integers(2, 30)))), tmax))
        pop = d.PopModel.skygrid(x, np.log(par.pop_n0) + rng.normal(0.0, 0.4, x.shape[0]), log_linear=bool(case % 2))
    sc = Scenario("R%d" % case, tree, ref, tmax, mu, KAPPA, PI, pop, sites)
    parts = int(rng.choice([8, 60, 400, max(8, tips // 12), 65536]))
    limit = int(rng.choice([0, -1, 40]))
    seed = int(rng.integers(1, 10**6))
    per_cycle = int(rng.choice([5000, 10 * tips, 100 * tips]))
    what = "case %d (tips %d, sites %d, %d mutations, parts %d, limit %d, pop kind %d, seed %d, %d moves per cycle)" % (case, tips, sites, tree.mut_site.shape[0], parts, limit, kind, seed, per_cycle)
    bh, rh = _run(sc, seed, parts, False, max_part_nodes=limit)
    bd, rd = _run(sc, seed, parts, True, max_part_nodes=limit)
    try:
        for cycle in range(cycles):
            rh.do_mcmc_steps(per_cycle, per_cycle); rd.do_mcmc_steps(per_cycle, per_cycle)
            th, refh = rh.tree(); td, refd = rd.tree()
            _same_tree(th, td, "%s cycle %d" % (what, cycle))
            assert np.array_equal(refh, refd), (what, cycle)
        print("ok  ", what, "| parts made", rd.num_parts()[0], flush=True)
    except Exception as ex:
        bad += 1
        print("FAIL", what, str(ex)[:500], flush=True)
    finally:
        for r in (rh, rd): r.close()
        for b in (bh, bd): b.close()
print("failures:", bad)
