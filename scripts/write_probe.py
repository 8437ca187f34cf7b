"""One pass of C4 with or without the topology moves (argv[1] = 1 / 0): run under `rocprofv3 --pmc WRITE_SIZE` to see which
moves the HBM write traffic of k_run_moves comes from."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import split_parts, configure
topo = len(sys.argv) < 2 or sys.argv[1] != "0"
sc = make_scenario("C4")
parts, incl, seeds, root_part, ref = split_parts(sc, 8192, 20261001)
gpu = d.EmatBackend(sc.num_sites)
configure(gpu, sc, ref, parts, incl, seeds, root_part, topology=topo)
gpu.run_moves_per_part(1000); gpu.synchronize()
print("topology", topo, "parts", len(parts), "ms", gpu.last_run_ms(), flush=True)
gpu.close()
