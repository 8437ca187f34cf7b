#!/usr/bin/env python3
"""Wall-clock time of synchronised (isolated) passes of the benchmark workload: run_moves_per_part + synchronize, one at a time --
what a cycle's pass is -- next to back-to-back passes.  EMAT_VERBOSE=1 adds when every size class started and ended."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from delphy_amd.sharding import ShardedEngine
sc = make_scenario("C4")
eng = ShardedEngine(sc, num_parts=8192, seed=20261001)
eng.setup()
b = eng.backend
for _ in range(3):
    b.run_moves_per_part(1000)
b.synchronize()
iso = []
for _ in range(8):
    t0 = time.perf_counter(); b.run_moves_per_part(1000); b.synchronize(); iso.append((time.perf_counter() - t0) * 1e3)
t0 = time.perf_counter()
for _ in range(8):
    b.run_moves_per_part(1000)
b.synchronize(); b2b = (time.perf_counter() - t0) * 1e3 / 8
print("isolated passes: %s ms (median %.2f) | back to back %.2f ms per pass | main kernel %.2f ms" % (" ".join("%.1f" % x for x in iso), float(np.median(iso)), b2b, b.last_run_ms()))
eng.close()
