"""Ad-hoc GPU probe: long parity runs + first timings (not part of the test suite)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import run_parity, split_parts, configure

def timing(name, tips, sites, nparts, moves, use_lds=True):
    t0 = time.time()
    sc = make_scenario(name, num_tips=tips, num_sites=sites)
    t1 = time.time()
    parts, incl, seeds, root_part, ref = split_parts(sc, nparts, 5)
    t2 = time.time()
    gpu = d.EmatBackend(sc.num_sites, use_lds=use_lds)
    configure(gpu, sc, ref, parts, incl, seeds, root_part)
    gpu.recalc_derived(); gpu.synchronize()
    t3 = time.time()
    gpu.run_moves_per_part(moves); gpu.synchronize(); ms1 = gpu.last_run_ms()
    gpu.run_moves_per_part(moves); gpu.synchronize(); ms = gpu.last_run_ms()
    sizes = np.array([p.num_nodes for p in parts])
    bad = [gpu.part_stats(p)["status"] for p in range(len(parts))]
    nb = sum(1 for b in bad if b != 0)
    st = gpu.part_stats(0)
    print("%s tips=%d parts=%d (nodes/part min %d med %d max %d) lds=%s: gen %.1fs part %.1fs setup %.1fs | %d moves/part: %.2f ms (first %.2f) -> %.2f M moves/s | bad parts %d | alg bytes/move %.0f"
          % (name, tips, len(parts), sizes.min(), np.median(sizes), sizes.max(), use_lds, t1 - t0, t2 - t1, t3 - t2, moves, ms, ms1, len(parts) * moves / ms / 1e3, nb,
             st["algorithmic_bytes"] / max(1, st["moves_done"])), flush=True)
    if nb:
        print("   first bad:", [(i, b) for i, b in enumerate(bad) if b][:5], gpu.last_error())
    gpu.close()

if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "long"):
        sc = make_scenario("C1", num_tips=100, num_sites=3000, uncertain_tips=0.3)
        t0 = time.time(); run_parity(sc, 1, 100000, trace=100000); print("long single root part 100k moves ok %.1fs" % (time.time() - t0), flush=True)
        sc = make_scenario("C2", num_tips=400, num_sites=4000, uncertain_tips=0.2)
        t0 = time.time(); run_parity(sc, 12, 50000, trace=50000); print("long C2 12 parts 50k moves/part ok %.1fs" % (time.time() - t0), flush=True)
        sc = make_scenario("C3", num_tips=2000, num_sites=29903)
        t0 = time.time(); run_parity(sc, 64, 20000, trace=20000); print("long C3-2k 64 parts 20k moves/part ok %.1fs" % (time.time() - t0), flush=True)
    if what in ("all", "time"):
        timing("C1", 100, 30000, 1, 20000)
        timing("C2", 1610, 18959, 64, 5000)
        timing("C3", 10000, 29903, 512, 2000)
        timing("C3", 10000, 29903, 512, 2000, use_lds=False)
        timing("C4", 100000, 29903, 4096, 1000)
        timing("C4", 100000, 29903, 8192, 1000)
        timing("C4", 100000, 29903, 8192, 1000, use_lds=False)
