"""Does the run driver's part-size limit (emat_run_set_max_part_nodes: extra cut nodes drawn uniformly at random in oversized parts) leave the
posterior alone?  Device arms only, so that the runs can be long: the same tree, model, number of parts requested and moves per cycle; per seed one arm
under the reference's partition rule exactly (limit 0) and one arm per limit given (the driver's default, and a much tighter one that adds several times
as many cut nodes -- whatever the rule did to the posterior, that arm would show it first).  Summaries, effective sample sizes and pooled z as in
tests/posterior_scale.py.  Usage (GPU box):
    python tests/posterior_limit.py [tips=5000] [parts=200] [cycles=8400] [burn_in=400] [seeds=4] [limits=-1,48] [seed_base=9001]   -> gpurun_out/posterior_limit.json + a table"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from posterior_scale import NAMES, summarise  # noqa: E402

if __name__ == "__main__":
    a = sys.argv[1:]
    tips, parts, cycles, burn, seeds = (int(a[i]) if len(a) > i else v for i, v in enumerate((5000, 200, 8400, 400, 4)))
    limits = [int(x) for x in (a[5] if len(a) > 5 else "-1,48").split(",")]
    seed_base = int(a[6]) if len(a) > 6 else 9001      # (another base = an independent replication)
    out_dir = os.path.join(ROOT, "gpurun_out", "posterior_limit"); os.makedirs(out_dir, exist_ok=True)
    me = os.path.join(ROOT, "tests", "posterior_scale.py")
    procs = []
    for s in range(seeds):
        for k, lim in enumerate([0] + limits):
            p = os.path.join(out_dir, "seed%d_limit%d.json" % (s, lim))
            procs.append((s, lim, p, subprocess.Popen([sys.executable, me, "--gpu", str(tips), str(parts), str(cycles), str(seed_base + 17 * s + k), p, str(lim)])))
    res = {}
    for s, lim, path, p in procs:
        assert p.wait() == 0, (s, lim)
        c = res[(s, lim)] = summarise(path, burn, cycles)
        print("seed %d limit %3d: parts %5d (frozen %.2f%%) | %d retained | %.0f s | limit in effect %s nodes, %.1f cut nodes added per cycle, largest part %s" %
              (s, lim, c["parts"], 100 * c["frozen_fraction"], c["retained"], c["seconds"], c.get("max_part_nodes"), c.get("extra_cuts_per_cycle", 0.0), c.get("largest_part_nodes_max")), flush=True)
        for nm in NAMES:
            st = c["stats"][nm]
            print("      %-22s mean %14.4f sd %10.4f ESS %8.1f se %9.4f" % (nm, st["mean"], st["sd"], st["ess"], st["se"]), flush=True)
    out = {"tips": tips, "parts_requested": parts, "cycles": cycles, "burn_in": burn, "seeds": seeds, "seed_base": seed_base, "limits": limits, "emat_build_id": res[(0, 0)]["emat_build_id"], "pooled": {}, "arms": {}}
    for (s, lim), c in res.items():
        out["arms"]["seed%d_limit%d" % (s, lim)] = c
    worst = 0.0
    for lim in limits:
        pooled = {}
        for nm in NAMES:
            diffs = [res[(s, lim)]["stats"][nm]["mean"] - res[(s, 0)]["stats"][nm]["mean"] for s in range(seeds)]
            ses = [float(np.hypot(res[(s, lim)]["stats"][nm]["se"], res[(s, 0)]["stats"][nm]["se"])) for s in range(seeds)]
            pse = float(np.sqrt(np.sum(np.square(ses))) / seeds)
            pooled[nm] = {"z_per_seed": [dd / max(1e-300, se) for dd, se in zip(diffs, ses)], "mean_difference": float(np.mean(diffs)), "pooled_se": pse, "pooled_z": float(np.mean(diffs) / max(1e-300, pse)),
                          "sd_of_the_summary": float(np.mean([res[(s, 0)]["stats"][nm]["sd"] for s in range(seeds)]))}
            worst = max(worst, abs(pooled[nm]["pooled_z"]))
        out["pooled"]["limit_%d_vs_reference_rule" % lim] = dict(pooled, cut_nodes_added_per_cycle=float(np.mean([res[(s, lim)].get("extra_cuts_per_cycle", 0.0) for s in range(seeds)])),
                                                                  limit_in_effect=res[(0, lim)].get("max_part_nodes"))
        print("limit %d (in effect %s nodes, %.1f cut nodes added per cycle) against the reference's rule, %d seeds:" % (lim, res[(0, lim)].get("max_part_nodes"), out["pooled"]["limit_%d_vs_reference_rule" % lim]["cut_nodes_added_per_cycle"], seeds))
        for nm in NAMES:
            q = pooled[nm]
            print("      %-22s difference %+10.4f +- %8.4f (%.3f sd of the summary)  pooled z %+5.2f   per seed %s" % (nm, q["mean_difference"], q["pooled_se"], q["mean_difference"] / max(1e-300, q["sd_of_the_summary"]), q["pooled_z"], " ".join("%+.2f" % z for z in q["z_per_seed"])))
    out["worst_abs_pooled_z"] = worst
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "posterior_limit.json"), "w"), indent=1)
    print("worst |pooled z| = %.2f" % worst)
