"""Whole cycles (repartition -> moves -> reassemble, tree resident in HBM) over more than one stencil period of the reference
(200 cycles, run.cpp:87-108), cycle by cycle: wall time, number of parts, largest part.  Usage:
  python scripts/steady_cycles.py [cycles=210] [max_part_nodes ...]      (-1 = the driver's default, 0 = the reference's rule)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import delphy_amd as d  # noqa: E402
from delphy_amd.scenarios import make_scenario  # noqa: E402


def run(sc, cycles, limit, parts=8192):
    per_cycle = 50 * sc.tree.num_nodes
    b = d.EmatBackend(sc.num_sites)
    r = d.EmatRun(b, sc.tree, sc.ref, 20261001)
    r.set_num_parts(parts); r.set_max_part_nodes(limit); r.set_hky(sc.mu, sc.kappa, sc.pi); r.set_pop_model(sc.pop); r.set_device_tree(True)
    r.do_mcmc_steps(per_cycle, per_cycle)
    rows = []
    t0 = time.perf_counter()
    for _ in range(cycles):
        t1 = time.perf_counter(); r.do_mcmc_steps(per_cycle, per_cycle); ms = (time.perf_counter() - t1) * 1e3
        st = r.partition_stats()
        rows.append((ms, st["num_parts"], st["largest_part_nodes"], st["extra_cuts"], b.last_run_ms()))
    dt = time.perf_counter() - t0
    lim = r.partition_stats()["max_part_nodes"]
    r.close(); b.close()
    a = np.array(rows)
    dec = [float(np.percentile(a[:, 0], q)) for q in (10, 50, 90, 100)]
    out = {"max_part_nodes": limit, "limit_in_effect": lim, "cycles": cycles, "moves_per_cycle": per_cycle, "moves_per_s": cycles * per_cycle / dt, "ms_per_cycle_mean": dt / cycles * 1e3,
           "ms_p10_p50_p90_max": dec, "p90_over_p10": dec[2] / dec[0],
           "by_decile_of_the_run": [{"cycles": "%d-%d" % (k, min(cycles, k + cycles // 10) - 1), "ms_mean": float(a[k: k + cycles // 10, 0].mean()), "pass_ms_mean": float(a[k: k + cycles // 10, 4].mean()),
                                     "parts_mean": float(a[k: k + cycles // 10, 1].mean()), "largest_part_max": int(a[k: k + cycles // 10, 2].max()), "extra_cuts_mean": float(a[k: k + cycles // 10, 3].mean())}
                                    for k in range(0, cycles, max(1, cycles // 10))]}
    return out


if __name__ == "__main__":
    cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 210
    limits = [int(x) for x in sys.argv[2:]] or [-1, 0]
    sc = make_scenario(os.environ.get("EMAT_WORKLOAD", "C4"))
    for lim in limits:
        print(json.dumps(run(sc, cycles, lim)), flush=True)
