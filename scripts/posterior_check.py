"""Posterior equivalence of fine and coarse partitions ("statistical correctness of many tiny partitions"), and what
a move buys in each: effective samples per second next to moves per second.

The same tree, model and number of local moves per cycle (the reference's 50 x nodes), repartitioned every cycle, cut into 8 parts
(the reference's own policy: parts = workers), 32 parts, and ~24 nodes per part (the benchmark's density).  All arms sample the same
posterior over trees (parameters fixed: no global moves here), so after burn-in the marginals of log_G, the whole-tree coalescent
prior, root time, tree length and mutation count must agree.  One sample per cycle; standard errors from the effective sample size
(Geyer's initial positive sequence on the autocorrelations), z-scores against the coarsest arm.
Usage (GPU box): python scripts/posterior_check.py [tips] [cycles] [burn_in]   -> gpurun_out/posterior_check.json + a table.
The arms run side by side as separate processes (each a few CUs wide); this process never touches the GPU."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

NAMES = ["log_G", "log_coalescent_prior", "root_time", "tree_length", "num_muts"]


def chain(tips, num_parts, cycles, seed, out_path):
    import delphy_amd as d
    from delphy_amd.scenarios import make_scenario
    sc = make_scenario("C3", num_tips=tips, num_sites=29903, uncertain_tips=0.2)
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, seed)
    run.set_num_parts(num_parts); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
    run.set_max_part_nodes(-1)       # the part-size limit on, as in bench.py's whole cycles (opt-in since round 6)
    run.set_device_tree(True)
    t_step = sc.default_t_step(); run.set_coalescent_t_step(t_step)
    nodes = sc.tree.num_nodes
    per_cycle = 50 * nodes
    tip_mask = sc.tree.child0 == -1
    t_ref = float(np.max(sc.tree.t[tip_mask]))
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
        par, c0, c1, t, root = b.tree_topology()
        T = float(np.sum(t[par >= 0] - t[par[par >= 0]]))
        rows.append([G, prior, float(t[root]), T, nm])
    dt = time.perf_counter() - t0
    run.close(); b.close()
    json.dump({"rows": rows, "parts": n, "parts_requested": num_parts, "frozen_fraction": float(np.mean(frozen)), "seconds": dt, "moves_per_cycle": per_cycle, "nodes": nodes}, open(out_path, "w"))


def ess(x):
    """Effective sample size: n / (1 + 2 sum of autocorrelations), the sum cut where consecutive pairs stop being positive (Geyer)."""
    x = np.asarray(x, np.float64); n = x.shape[0]
    x = x - x.mean()
    v = float(np.dot(x, x)) / n
    if v == 0.0:
        return float(n)
    f = np.fft.rfft(x, 2 * n)
    acf = np.fft.irfft(f * np.conj(f))[:n].real / (n * v)
    s = 0.0
    for k in range(1, n - 1, 2):
        pair = acf[k] + acf[k + 1]
        if pair <= 0.0:
            break
        s += pair
    return float(n / max(1.0, 1.0 + 2.0 * s))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--arm":
        chain(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
        sys.exit(0)
    tips = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 4400
    burn = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    nodes = 2 * tips - 1
    arms = [8, 32, max(8, nodes // 24)]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    procs = []
    for k, num_parts in enumerate(arms):
        path = os.path.join(ROOT, "gpurun_out", "posterior_arm_%d.json" % num_parts)
        procs.append((num_parts, path, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--arm", str(tips), str(num_parts), str(cycles), str(4242 + k), path])))
    out = {"tips": tips, "nodes": nodes, "cycles": cycles, "burn_in": burn, "configs": []}
    for num_parts, path, p in procs:
        assert p.wait() == 0, "arm %d failed" % num_parts
        a = json.load(open(path)); rows = np.array(a["rows"])[burn:]
        cfg = {k: a[k] for k in ("parts", "parts_requested", "frozen_fraction", "seconds", "moves_per_cycle")}
        cfg["retained"] = int(rows.shape[0]); cfg["moves_per_s"] = cycles * a["moves_per_cycle"] / a["seconds"]
        cfg["stats"] = {}
        for j, nm in enumerate(NAMES):
            e = ess(rows[:, j]); sd = float(np.std(rows[:, j], ddof=1))
            cfg["stats"][nm] = {"mean": float(np.mean(rows[:, j])), "sd": sd, "ess": e, "se": sd / np.sqrt(e), "ess_per_s": e / (a["seconds"] * rows.shape[0] / cycles),
                                "ess_per_million_moves": e / (rows.shape[0] * a["moves_per_cycle"] / 1e6)}
        out["configs"].append(cfg)
        print("parts %5d (requested %5d) frozen nodes %.1f%% | %d retained samples | %.1f s, %.2f M moves/s inclusive (three arms sharing the GPU)"
              % (cfg["parts"], num_parts, 100 * cfg["frozen_fraction"], cfg["retained"], a["seconds"], cfg["moves_per_s"] / 1e6), flush=True)
        for nm in NAMES:
            s = cfg["stats"][nm]
            print("    %-22s mean %14.4f  sd %10.4f  ESS %8.1f  se %9.4f  ESS/s %8.3f  ESS per 1e6 moves %8.4f" % (nm, s["mean"], s["sd"], s["ess"], s["se"], s["ess_per_s"], s["ess_per_million_moves"]), flush=True)
    base = out["configs"][0]["stats"]
    print("z-scores against the %d-part arm (difference of means / combined ESS-based standard error):" % out["configs"][0]["parts"])
    worst = 0.0
    for cfg in out["configs"][1:]:
        z = {nm: (cfg["stats"][nm]["mean"] - base[nm]["mean"]) / max(1e-300, float(np.hypot(cfg["stats"][nm]["se"], base[nm]["se"]))) for nm in NAMES}
        cfg["z_vs_coarsest"] = z
        worst = max(worst, max(abs(v) for v in z.values()))
        print("    parts %5d: " % cfg["parts"] + "  ".join("%s %+.2f" % (k, v) for k, v in z.items()))
    out["worst_abs_z"] = worst
    import delphy_amd as d
    out["emat_build_id"] = d.library_build_id()       # which device code sampled these chains (bench.py marks the block stale when it differs)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "posterior_check.json"), "w"), indent=1)
