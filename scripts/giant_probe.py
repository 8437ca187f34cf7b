"""Phase breakdown (the -DEMAT_PROFILE_PHASES library, EMAT_LIB_PATH) of the slowest chains of a WHOLE cycle -- after a few
repartitions the reference's partitioning rule has produced parts of several hundred nodes beside the typical two dozen."""
import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
sc = make_scenario("C4")
b = d.EmatBackend(sc.num_sites)
run = d.EmatRun(b, sc.tree, sc.ref, 20261001)
run.set_num_parts(8192); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop); run.set_device_tree(True)
per = 50 * sc.tree.num_nodes
run.do_mcmc_steps(3 * per, per)
lib = d.load_library()
lib.emat_debug_phase_ticks.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64)]
st = []
while True:
    try: st.append(b.part_stats(len(st)))
    except Exception: break
n = len(st)
dur = np.array([s["device_ticks"] for s in st], dtype=np.float64) / 1e5
nodes = np.array([s["num_nodes"] for s in st])
names = ["core:analyze+peel", "core:topology", "core:propose", "scans in LDS", "core:coal+accept+apply", "spr1:analyze+peel", "spr1:missing+seed_fill pre", "spr1:study pre+pick",
         "spr1:topology", "spr1:propose", "spr1:seed_fill post", "spr1:study post+alpha", "spr1:accept+apply", "regions (count)", "ALL simple moves", "ALL topology moves"]
buf = (C.c_int64 * 16)()
print("parts %d | chain ms: median %.2f p99 %.2f max %.2f | corr(duration, nodes) %.2f" % (n, np.median(dur), np.percentile(dur, 99), dur.max(), np.corrcoef(dur, nodes)[0, 1]))
sel = list(np.argsort(-nodes)[:3]) + list(np.argsort(-dur)[:3]) + list(np.argsort(dur)[n // 2: n // 2 + 1])
for p in sel:
    lib.emat_debug_phase_ticks(b.handle, int(p), buf); v = np.array(list(buf), dtype=np.float64)
    prop = st[p]["proposed"]; tot = v[14] + v[15]
    simple = max(1, prop[0] + prop[1] + prop[2]); topo = max(1, prop[3] + prop[4])
    print("part %d nodes %d dur %.2f ms proposed %s | ticks per simple move %.0f, per topology move %.0f | regions/study %.1f " % (p, nodes[p], dur[p], prop, v[14] / simple, v[15] / topo, v[13] / max(1, 2 * prop[4])))
    print("    " + " | ".join("%s %.1f%%" % (names[i], 100 * v[i] / tot) for i in list(range(0, 3)) + list(range(4, 13)) + [14, 15]))
    os.environ["EMAT_PHASE_EXTRA"] = "1"
    lib.emat_debug_phase_ticks(b.handle, int(p), buf)
    del os.environ["EMAT_PHASE_EXTRA"]
    e = np.array(list(buf), dtype=np.float64); k = max(1.0, e[0])
    print("    wave scans %d: missing intervals at X %.1f, site deltas %.1f, items %.1f, levels %.1f, in HBM %d, fell back to serial %d, sets not in LDS %d | scan ticks per scan %.0f | arena bytes per topology move: %.0f HBM, %.0f LDS"
          % (e[0], e[1] / k, e[2] / k, e[3] / k, e[4] / k, e[5], e[6], e[7], v[6] / k, e[8] / topo, e[9] / topo))
    print("    slab layout", b.debug_slab_layout(int(p)))
