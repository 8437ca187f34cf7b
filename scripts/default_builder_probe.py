"""Times the product's default initial-tree builder (emat_tree_build_default: host C++, no GPU) on the first N tips' worth of a workload and prints a digest
of the tree it makes (same descriptors + same seed => the same digest whatever was optimised).   python scripts/default_builder_probe.py [tips=20000] [workload=C4] [seed=8]"""
import sys, time, hashlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from oracle_ffi import OracleBuild
tips_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
name = sys.argv[2] if len(sys.argv) > 2 else "C4"
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sc = make_scenario(name, num_tips=tips_n)
ob = OracleBuild(sc.ref); tips = ob.tip_descs_of(sc.tree); ob.close()      # (the oracle only turns the scenario's tree into tip descriptors here)
b = d.EmatBackend(sc.num_sites, device=-1)
b.set_ref_sequence(sc.ref)
t0 = time.perf_counter(); c0 = time.process_time()
tree, ref, rep = b.build_default(tips, seed)
dt = time.perf_counter() - t0; dc = time.process_time() - c0
h = hashlib.sha256()
for f in ("parent", "child0", "child1", "t", "mut_offset", "mut_site", "mut_from", "mut_to", "mut_t", "miss_offset", "miss_start", "miss_end"):
    h.update(np.ascontiguousarray(getattr(tree, f)).tobytes())
h.update(np.ascontiguousarray(ref).tobytes())
print("%s, %d tips, seed %d: %.1f s (%.1f s of CPU), digest %s, report %s" % (name, tips_n, seed, dt, dc, h.hexdigest()[:16], rep))
b.close()
