#!/usr/bin/env python3
"""Initial-tree construction: wall time of emat_tree_build_usher_like on the device next to the oracle's restatement
of the reference's builder on one host core.  Usage: build_probe.py [C3] [tips]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from oracle_ffi import OracleBuild

name = sys.argv[1] if len(sys.argv) > 1 else "C3"
tips_n = int(sys.argv[2]) if len(sys.argv) > 2 else None
sc = make_scenario(name, num_tips=tips_n)
ob = OracleBuild(sc.ref); tips = ob.tip_descs_of(sc.tree)
b = d.EmatBackend(sc.num_sites); b.set_ref_sequence(sc.ref)
if tips.num_tips <= 30000:
    b.build_usher_like(tips, 1)      # warm-up (first launch, allocations)
t0 = time.perf_counter(); tree = b.build_usher_like(tips, 7); t_dev = time.perf_counter() - t0
skip_cpu = tips.num_tips > 30000 or os.environ.get("EMAT_PROBE_NO_ORACLE") == "1"
if not skip_cpu:
    t0 = time.perf_counter(); ot = ob.build_usher_like(tips, 7); t_cpu = time.perf_counter() - t0
rc, msg = ob.check(tree, tips)
print("%s: %d tips, %d sites, %d deltas in the descriptors | device %.2f s | oracle (one host core) %s | %d mutations in the built tree | closing checks: %s"
      % (name, tips.num_tips, sc.num_sites, tips.delta_site.shape[0], t_dev, "skipped" if skip_cpu else "%.2f s" % t_cpu, tree.mut_site.shape[0], "ok" if rc == 0 else msg))
b.close(); ob.close()
