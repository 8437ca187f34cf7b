"""The part-size limit's refinement on the host alone (no GPU): draws a run's partition for several epochs and limits, prints a digest of every
draw's cut nodes and the time per draw.  Used to check that a faster refine_stencil draws the very same cut nodes.
  python scripts/refine_probe.py [scenario=C4] [parts=8192] [draws=12]"""
import sys, time, hashlib
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
name = sys.argv[1] if len(sys.argv) > 1 else "C4"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
draws = int(sys.argv[3]) if len(sys.argv) > 3 else 12
sc = make_scenario(name)
for limit in (-1, 21, 30, 45, 200, -1):
    run = d.EmatRun(None, sc.tree, sc.ref, 20261001)
    run.set_num_parts(parts); run.set_max_part_nodes(limit)
    run.repartition(); run.reassemble()
    h = hashlib.sha256(); ts = []; n = 0
    for k in range(draws):
        if k % 3 == 0:
            run.repartition(); run.reassemble()      # another epoch: other streams, another pick of the stencil
        t0 = time.perf_counter()
        cuts = run.debug_redraw_partition()
        ts.append(time.perf_counter() - t0)
        h.update(np.ascontiguousarray(cuts, dtype=np.int32).tobytes()); n = len(cuts)
    print("limit %4d: %6d cut nodes, digest %s, median draw %.2f ms" % (limit, n, h.hexdigest()[:16], 1e3 * sorted(ts)[len(ts) // 2]), flush=True)
    run.close()
