"""Posterior equivalence of fine and coarse partitions ("statistical correctness of many tiny partitions").

The same tree, model and number of local moves per cycle (the reference's 50 x nodes), repartitioned every cycle, with
8 / 64 / many (~25 nodes each, the benchmark's density) parts.  All three chains sample the same posterior over trees
(parameters fixed: no global moves here), so after burn-in the marginals of log_G, the whole-tree coalescent prior, root
time, tree length and mutation count must agree; how fast they get there per move is what the frozen boundary nodes cost.
Usage (GPU box): python scripts/posterior_check.py [tips] [cycles]    -> gpurun_out/posterior_check.json + a table."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
from delphy_amd.scenarios import make_scenario


def chain(sc, num_parts, cycles, seed):
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, seed)
    run.set_num_parts(num_parts); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
    t_step = sc.default_t_step(); run.set_coalescent_t_step(t_step)
    nodes = sc.tree.num_nodes
    per_cycle = 50 * nodes
    tips = sc.tree.child0 == -1
    t_ref = float(np.max(sc.tree.t[tips]))
    rows = []; frozen = []; t0 = time.perf_counter()
    for c in range(cycles):
        run.repartition()
        n, _ = run.num_parts()
        frozen.append((n - 1) / nodes)                         # cut nodes: frozen (as tips) in the part above them for this cycle
        run.run_moves(per_cycle)
        G, _ = b.totals()
        prior = b.scalable_coalescent_log_prior(t_ref, t_step)
        nm = b.global_stats(1)[2]
        run.reassemble()
        tree, _ = run.tree()
        T = float(np.sum(tree.t[tree.parent >= 0] - tree.t[tree.parent[tree.parent >= 0]]))
        rows.append([G, prior, float(tree.t[tree.root]), T, nm])
    dt = time.perf_counter() - t0
    run.close(); b.close()
    return np.array(rows), n, float(np.mean(frozen)), dt, per_cycle


def summarize(x):
    """mean, sd and a standard error from batch means (8 batches) of the second half of the chain."""
    x = x[len(x) // 2:]
    bm = np.array([np.mean(c) for c in np.array_split(x, 8)])
    return float(np.mean(x)), float(np.std(x)), float(np.std(bm, ddof=1) / np.sqrt(len(bm)))


if __name__ == "__main__":
    tips = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    sc = make_scenario("C3", num_tips=tips, num_sites=29903, uncertain_tips=0.2)
    nodes = sc.tree.num_nodes
    names = ["log_G", "log_coalescent_prior", "root_time", "tree_length", "num_muts"]
    out = {"tips": tips, "nodes": nodes, "cycles": cycles, "configs": []}
    for num_parts in (16, 64, max(8, nodes // 24)):
        rows, n, frozen, dt, per_cycle = chain(sc, num_parts, cycles, 4242)
        cfg = {"parts_requested": num_parts, "parts": n, "frozen_fraction": frozen, "seconds": dt, "moves_per_cycle": per_cycle,
               "moves_per_s": cycles * per_cycle / dt, "stats": {nm: summarize(rows[:, j]) for j, nm in enumerate(names)},
               "first": rows[0].tolist(), "last": rows[-1].tolist()}
        out["configs"].append(cfg)
        print("parts %5d (requested %5d) frozen nodes %.1f%% | %.1f s, %.1f M moves/s inclusive" % (n, num_parts, 100 * frozen, dt, cycles * per_cycle / dt / 1e6), flush=True)
        for nm in names:
            m, s, se = cfg["stats"][nm]
            print("    %-22s mean %14.4f  sd %10.4f  se %9.4f" % (nm, m, s, se), flush=True)
    base = out["configs"][0]["stats"]
    print("z-scores against the coarsest chain (difference of means / combined batch-means standard error):")
    for cfg in out["configs"][1:]:
        z = {nm: (cfg["stats"][nm][0] - base[nm][0]) / max(1e-300, np.hypot(cfg["stats"][nm][2], base[nm][2])) for nm in names}
        cfg["z_vs_coarsest"] = z
        print("    parts %5d: " % cfg["parts"] + "  ".join("%s %+.2f" % (k, v) for k, v in z.items()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "posterior_check.json"), "w"), indent=1)
