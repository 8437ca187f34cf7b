"""Hand-run hunt: tests/test_device_tree_gpu.py::test_default_device_tree_path_move_for_move_against_the_oracle over random scenarios -- the DEFAULT path of
a run (tree resident in HBM, parts cut by kernels, coalescent tables built by kernels, the part-size limit on or off), every pass compared move for move with
the oracle started from the parts, tables and RNG positions the device holds.
  python scripts/fuzz_default_path.py SEED CASES [MAX_TIPS=4000] [CYCLES=3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
import delphy_amd.engine as e
from delphy_amd.scenarios import Scenario, KAPPA, PI
from helpers import replay_device_parts_in_the_oracle
seed0, cases = int(sys.argv[1]), int(sys.argv[2])
max_tips = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
cycles = int(sys.argv[4]) if len(sys.argv) > 4 else 3
os.environ.pop("EMAT_TREE_HOST_COALESCENT", None)
rng = np.random.default_rng(seed0)
bad = 0
for case in range(cases):
    tips = int(10 ** rng.uniform(1.3, np.log10(max_tips)))
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
    parts = int(rng.choice([2, 8, 60, max(2, tips // 12), 65536]))
    limit = int(rng.choice([0, -1, 40]))
    seed = int(rng.integers(1, 10**6))
    per_part = int(rng.choice([100, 600, 2000]))
    what = "case %d (tips %d, sites %d, %d mutations, parts %d, limit %d, pop kind %d, seed %d, %d moves per part)" % (case, tips, sites, tree.mut_site.shape[0], parts, limit, kind, seed, per_part)
    b = d.EmatBackend(sc.num_sites, trace_moves=per_part + 1)
    run = d.EmatRun(b, sc.tree, sc.ref, seed)
    run.set_num_parts(parts); run.set_max_part_nodes(limit); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
    run.set_device_tree(True)
    cur_ref = sc.ref
    try:
        made = 0
        for cycle in range(cycles):
            run.repartition()
            made = run.num_parts()[0]
            replay_device_parts_in_the_oracle(sc, b, run, cur_ref, made * per_part + 7, per_part + 1)
            run.reassemble()
            _, cur_ref = run.tree()
        print("ok  ", what, "| parts made", made, flush=True)
    except Exception as ex:
        bad += 1
        print("FAIL", what, repr(ex)[:600], flush=True)
    finally:
        run.close(); b.close()
print("failures:", bad)
