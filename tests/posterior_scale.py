"""Posterior equivalence of fine and coarse partitions AT SCALE, and effective samples per second (run by hand on the GPU box; not
collected by pytest; results are committed under profiles/).  It lives under tests/ because its reference arm IS the oracle: the
reference's own policy of a part per worker (tools/delphy.cpp:130-132: here 8 parts on 8 host threads, oracle Subruns on parts cut
and gathered by the oracle's restatement of Run::repartition / reassemble), against which the engine's fine partitions on the GPU
are judged: 200 parts, and the benchmark's density of about 25 nodes per part.

All arms sample the same posterior over trees (model parameters fixed: no global moves on either side), one sample per cycle of the
reference's 50 x nodes local moves with a fresh partition.  Five summaries: log G, the whole-tree coalescent prior, root time, tree
length, mutation count.  Standard errors from the effective sample size (Geyer's initial positive sequence); every seed gives one
z-score per summary and fine arm against its own coarse arm; the seeds are independent, so the pooled z is the mean difference over
the root-sum-square of the standard errors.  Usage:
    python tests/posterior_scale.py [tips] [cycles] [burn_in] [seeds] [seed_base=7001]      -> gpurun_out/posterior_scale.json + a table
EMAT_POSTERIOR_COARSE_DIR=<dir>: the oracle arms are not run but read from <dir>/seed<k>_coarse.json(.gz) -- arms made earlier with
`--oracle <tips> 8 <cycles> <seed_base + 13 k> <path> <threads>` (they do not depend on the engine's build: round 6 ran the four
8 400-cycle arms in the build container, CPU only, and the device arms of the same length on the GPU box)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

NAMES = ["log_G", "log_coalescent_prior", "root_time", "tree_length", "num_muts"]


def _scenario(tips):
    from delphy_amd.scenarios import make_scenario
    return make_scenario("C3", num_tips=tips, num_sites=29903, uncertain_tips=0.2)


def _tree_summaries(tree):
    nonroot = np.arange(tree.num_nodes) != tree.root
    T = float(np.sum(tree.t[nonroot] - tree.t[tree.parent[nonroot]]))
    nm = int(tree.mut_offset[-1] - (tree.mut_offset[tree.root + 1] - tree.mut_offset[tree.root]))
    return float(tree.t[tree.root]), T, nm


def chain_gpu(tips, num_parts, cycles, seed, out_path, limit=-1):
    """The engine: tree resident in HBM, parts cut / moved / gathered by kernels.  `limit`: emat_run_set_max_part_nodes (-1 = the driver's default)."""
    import delphy_amd as d
    sc = _scenario(tips)
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, seed)
    run.set_num_parts(num_parts); run.set_max_part_nodes(limit); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
    run.set_device_tree(True)
    t_step = sc.default_t_step(); run.set_coalescent_t_step(t_step)
    nodes = sc.tree.num_nodes; per_cycle = 50 * nodes
    t_ref = float(np.max(sc.tree.t[sc.tree.child0 == -1]))
    rows, frozen, extra, largest, t0 = [], [], [], [], time.perf_counter()
    for c in range(cycles):
        run.repartition()
        n, _ = run.num_parts(); frozen.append((n - 1) / nodes)
        ps = run.partition_stats(); extra.append(ps["extra_cuts"]); largest.append(ps["largest_part_nodes"])     # the part-size limit of the run driver at work (emat_run_set_max_part_nodes)
        run.run_moves(per_cycle)
        G, _ = b.totals()
        prior = b.scalable_coalescent_log_prior(t_ref, t_step)
        nm = b.global_stats(1)[2]
        run.reassemble()
        par, c0, c1, t, root = b.tree_topology()
        rows.append([G, prior, float(t[root]), float(np.sum(t[par >= 0] - t[par[par >= 0]])), nm])
        if (c + 1) % 200 == 0 or c + 1 == cycles:             # what there is so far survives a cut-off run
            json.dump({"rows": rows, "parts": n, "parts_requested": num_parts, "frozen_fraction": float(np.mean(frozen)), "seconds": time.perf_counter() - t0, "moves_per_cycle": per_cycle,
                       "nodes": nodes, "engine": "gpu", "emat_build_id": d.library_build_id(), "max_part_nodes": ps["max_part_nodes"], "extra_cuts_per_cycle": float(np.mean(extra)),
                       "largest_part_nodes_max": int(np.max(largest))}, open(out_path + ".tmp", "w"))
            os.replace(out_path + ".tmp", out_path)
    run.close(); b.close()


def chain_oracle(tips, num_parts, cycles, seed, out_path, threads):
    """The reference's policy restated: few parts, one host thread each; oracle Subruns between the oracle's Run::repartition and
    reassemble.  The whole-tree coalescent prior is the oracle's Scalable_coalescent_prior on the reassembled tree."""
    from helpers import configure
    from oracle_ffi import OracleEngine, OracleRun
    sc = _scenario(tips)
    nodes = sc.tree.num_nodes; per_cycle = 50 * nodes
    t_step = sc.default_t_step()
    t_ref = float(np.max(sc.tree.t[sc.tree.child0 == -1]))
    orun = OracleRun(sc.tree, sc.ref, seed, num_parts)
    rng = np.random.default_rng(seed)
    rows, frozen, t0 = [], [], time.perf_counter()
    ref = sc.ref
    for c in range(cycles):
        orun.repartition()
        n, root_part = orun.num_parts(); frozen.append((n - 1) / nodes)
        _, ref = orun.tree()                                   # (normalize_root at the repartition may have re-referenced)
        parts = [orun.part(p)[0] for p in range(n)]
        orc = OracleEngine(sc.num_sites)
        configure(orc, sc, ref, parts, [p == root_part for p in range(n)], [int(x) for x in rng.integers(1, 2**62, n)], root_part, t_step)
        counts = np.full(n, per_cycle // n, np.int64); counts[: per_cycle % n] += 1
        orc.run_moves_counts(counts, threads=threads)
        G, _ = orc.totals()
        for p in range(n):
            orun.part_put(p, orc.part_download(p))
        orc.close()
        orun.reassemble(); orun.normalize_root()
        tree, ref = orun.tree()
        whole = OracleEngine(sc.num_sites)
        sc2 = type(sc)(sc.name, tree, ref, sc.t_max_tip, sc.mu, sc.kappa, sc.pi, sc.pop, sc.num_sites)
        configure(whole, sc2, ref, [tree], [True], [1], 0, t_step)
        prior = whole.scalable_log_prior(0, t_ref, t_step)
        whole.close()
        root_t, T, nm = _tree_summaries(tree)
        rows.append([G, prior, root_t, T, nm])
        if (c + 1) % 100 == 0 or c + 1 == cycles:
            json.dump({"rows": rows, "parts": n, "parts_requested": num_parts, "frozen_fraction": float(np.mean(frozen)), "seconds": time.perf_counter() - t0, "moves_per_cycle": per_cycle,
                       "nodes": nodes, "engine": "oracle on %d host threads" % threads}, open(out_path + ".tmp", "w"))
            os.replace(out_path + ".tmp", out_path)
    orun.close()


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


def _load(path):
    if not os.path.exists(path) and os.path.exists(path + ".gz"):
        import gzip
        return json.load(gzip.open(path + ".gz", "rt"))
    return json.load(open(path))


def summarise(path, burn, cycles):
    a = _load(path); rows = np.array(a["rows"])[burn:]
    cfg = {k: a[k] for k in ("parts", "parts_requested", "frozen_fraction", "seconds", "moves_per_cycle", "engine")}
    cfg["emat_build_id"] = a.get("emat_build_id")
    for k in ("max_part_nodes", "extra_cuts_per_cycle", "largest_part_nodes_max"):
        if k in a: cfg[k] = a[k]
    cfg["retained"] = int(rows.shape[0]); cfg["moves_per_s"] = len(a["rows"]) * a["moves_per_cycle"] / a["seconds"]
    cfg["stats"] = {}
    for j, nm in enumerate(NAMES):
        e = ess(rows[:, j]); sd = float(np.std(rows[:, j], ddof=1))
        cfg["stats"][nm] = {"mean": float(np.mean(rows[:, j])), "sd": sd, "ess": e, "se": sd / np.sqrt(e), "ess_per_s": e / (a["seconds"] * rows.shape[0] / len(a["rows"])),
                            "ess_per_million_moves": e / (rows.shape[0] * a["moves_per_cycle"] / 1e6)}
    return cfg


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--gpu":
        chain_gpu(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7]) if len(sys.argv) > 7 else -1); sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--oracle":
        chain_oracle(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], int(sys.argv[7])); sys.exit(0)
    only_summarise = len(sys.argv) > 1 and sys.argv[1] == "--summarise"      # (what the arms of a cut-off run left in gpurun_out/posterior_scale)
    if only_summarise:
        sys.argv.pop(1)
    tips = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
    cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 10400
    burn = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    seeds = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    seed_base = int(sys.argv[5]) if len(sys.argv) > 5 else 7001      # (another base = an independent replication of the whole comparison)
    coarse_cycles = int(os.environ.get("EMAT_POSTERIOR_COARSE_CYCLES", cycles))
    nodes = 2 * tips - 1
    fine = [200, max(8, nodes // 25)]
    out_dir = os.path.join(ROOT, "gpurun_out", "posterior_scale"); os.makedirs(out_dir, exist_ok=True)
    procs = []
    me = os.path.abspath(__file__)
    coarse_dir = os.environ.get("EMAT_POSTERIOR_COARSE_DIR")
    for s in range(seeds):
        seed = seed_base + 13 * s
        p = os.path.join(coarse_dir or out_dir, "seed%d_coarse.json" % s)
        procs.append((s, "coarse", p, None if (only_summarise or coarse_dir) else subprocess.Popen([sys.executable, me, "--oracle", str(tips), "8", str(coarse_cycles), str(seed), p, "8"])))
        for k, nparts in enumerate(fine):
            p = os.path.join(out_dir, "seed%d_fine%d.json" % (s, k))
            procs.append((s, "fine%d" % k, p, None if only_summarise else subprocess.Popen([sys.executable, me, "--gpu", str(tips), str(nparts), str(cycles), str(seed + 1 + k), p])))
    out = {"tips": tips, "nodes": nodes, "cycles": cycles, "coarse_cycles": coarse_cycles, "burn_in": burn, "seeds": seeds, "seed_base": seed_base, "arms": {}, "z_per_seed": [], "pooled": {},
           "coarse_arms_from": coarse_dir}
    res = {}
    for s, arm, path, p in procs:
        assert p is None or p.wait() == 0, (s, arm)
        res[(s, arm)] = summarise(path, burn, cycles)
        c = res[(s, arm)]
        print("seed %d %-7s parts %5d (frozen %.2f%%) %s | %d retained | %.0f s, %.2f M moves/s%s" % (s, arm, c["parts"], 100 * c["frozen_fraction"], c["engine"], c["retained"], c["seconds"], c["moves_per_s"] / 1e6,
              " | part-size limit %d nodes: %.1f cut nodes added per cycle, largest part %d" % (c["max_part_nodes"], c["extra_cuts_per_cycle"], c["largest_part_nodes_max"]) if "max_part_nodes" in c else ""), flush=True)
        for nm in NAMES:
            st = c["stats"][nm]
            print("      %-22s mean %14.4f sd %10.4f ESS %8.1f se %9.4f ESS/s %8.3f ESS per 1e6 moves %8.4f" % (nm, st["mean"], st["sd"], st["ess"], st["se"], st["ess_per_s"], st["ess_per_million_moves"]), flush=True)
    for (s, arm), c in res.items():
        out["arms"]["seed%d_%s" % (s, arm)] = c
    worst = 0.0
    for k in range(len(fine)):
        arm = "fine%d" % k
        pooled = {}
        for nm in NAMES:
            diffs = [res[(s, arm)]["stats"][nm]["mean"] - res[(s, "coarse")]["stats"][nm]["mean"] for s in range(seeds)]
            ses = [float(np.hypot(res[(s, arm)]["stats"][nm]["se"], res[(s, "coarse")]["stats"][nm]["se"])) for s in range(seeds)]
            zs = [dd / max(1e-300, se) for dd, se in zip(diffs, ses)]
            pooled[nm] = {"z_per_seed": zs, "mean_difference": float(np.mean(diffs)), "pooled_se": float(np.sqrt(np.sum(np.square(ses))) / seeds),
                          "pooled_z": float(np.mean(diffs) / max(1e-300, np.sqrt(np.sum(np.square(ses))) / seeds)), "sd_of_the_summary": float(np.mean([res[(s, "coarse")]["stats"][nm]["sd"] for s in range(seeds)]))}
            worst = max(worst, abs(pooled[nm]["pooled_z"]))
        out["pooled"]["%d_parts_vs_8" % res[(0, arm)]["parts_requested"]] = pooled
        print("requested %d parts against the 8-part oracle arm, %d seeds:" % (fine[k], seeds))
        for nm in NAMES:
            q = pooled[nm]
            print("      %-22s difference %+10.4f +- %8.4f (%.3f sd of the summary)  pooled z %+5.2f   per seed %s" % (nm, q["mean_difference"], q["pooled_se"], q["mean_difference"] / max(1e-300, q["sd_of_the_summary"]), q["pooled_z"], " ".join("%+.2f" % z for z in q["z_per_seed"])))
    out["worst_abs_pooled_z"] = worst
    dens = "fine%d" % (len(fine) - 1)
    out["ess_per_s_at_benchmark_density"] = {nm: float(np.mean([res[(s, dens)]["stats"][nm]["ess_per_s"] for s in range(seeds)])) for nm in NAMES}
    out["ess_per_s_reference_policy_oracle"] = {nm: float(np.mean([res[(s, "coarse")]["stats"][nm]["ess_per_s"] for s in range(seeds)])) for nm in NAMES}
    out["emat_build_id"] = res[(0, "fine0")]["emat_build_id"]
    # the run driver's part-size limit at work in the device arms (the oracle arm partitions by the reference's rule alone): the posterior-equivalence arm of that limit
    out["part_size_limit"] = {arm: {"max_part_nodes": res[(0, arm)].get("max_part_nodes"), "cut_nodes_added_per_cycle": float(np.mean([res[(s, arm)].get("extra_cuts_per_cycle", 0.0) for s in range(seeds)])),
                                    "largest_part_nodes": int(max(res[(s, arm)].get("largest_part_nodes_max", 0) for s in range(seeds)))} for arm in ("fine0", "fine1")}
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "posterior_scale.json"), "w"), indent=1)
    print("worst |pooled z| = %.2f" % worst)
