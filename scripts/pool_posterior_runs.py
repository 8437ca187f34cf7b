"""Pools the runs of tests/posterior_scale.py made against the SAME oracle arms (profiles/r06/posterior_oracle_8_parts_8400_cycles: the coarse arms do not
depend on the engine's build and were made once) and writes the `replicates` block that bench.py quotes under mixing.at_scale.
  python scripts/pool_posterior_runs.py <newest run.json> <note for it> <older run.json:label> ...   -> prints the pooled table, rewrites the newest file in place
The device chains of all runs are independent (other seeds or other builds); the four oracle arms are shared, so their error is counted once."""
import json, sys
import numpy as np
newest_path, note = sys.argv[1], sys.argv[2]
older = [a.split(":", 1) for a in sys.argv[3:]]
runs = [(p, label, json.load(open(p))) for p, label in older] + [(newest_path, None, json.load(open(newest_path)))]
new = runs[-1][2]
names = ["log_G", "log_coalescent_prior", "root_time", "tree_length", "num_muts"]
seeds = new["seeds"]


def pooled(arm, name):
    dev, dev_se = [], []
    for _, _, o in runs:
        for s in range(seeds):
            st = o["arms"]["seed%d_%s" % (s, arm)]["stats"][name]; dev.append(st["mean"]); dev_se.append(st["se"])
    orc = [new["arms"]["seed%d_coarse" % s]["stats"][name] for s in range(seeds)]
    diff = np.mean(dev) - np.mean([x["mean"] for x in orc])
    se = np.sqrt(np.sum(np.square(dev_se)) / len(dev) ** 2 + np.sum(np.square([x["se"] for x in orc])) / seeds ** 2)
    return {"difference": float(diff), "se": float(se), "z": float(diff / se)}


pool = {label: {n: pooled(arm, n) for n in names} for arm, label in (("fine0", "about_270_parts_vs_8"), ("fine1", "about_470_parts_vs_8"))}
for k, v in pool.items():
    print(k, {n: (round(x["difference"], 3), round(x["se"], 3), round(x["z"], 2)) for n, x in v.items()})
new["coarse_arms_from"] = ("profiles/r06/posterior_oracle_8_parts_8400_cycles (the oracle under the reference's policy, 8 parts requested, two host threads per arm in the "
                           "build container, CPU only: made once, independent of the engine's build)")
new["ess_per_s_reference_policy_oracle"] = None
for k, v in new["arms"].items():
    if "coarse" in k:
        for st in v["stats"].values():
            st["ess_per_s"] = None
new["note"] = note
new["replicates"] = {
    "runs": [{"file": p, "build": o["emat_build_id"], "what": label, "worst_abs_pooled_z": o["worst_abs_pooled_z"]} for p, label, o in runs[:-1]]
            + [{"file": "this file", "build": new["emat_build_id"], "what": note, "worst_abs_pooled_z": new["worst_abs_pooled_z"]}],
    "pooled_over_the_runs": pool,
    "what": "%d runs of the matched comparison in round 6: %d independent device chains per fine arm against the same %d oracle arms of %d cycles (whose error is counted once)"
            % (len(runs), len(runs) * seeds, seeds, new.get("coarse_cycles", 0))}
json.dump(new, open(newest_path, "w"), indent=1)
