#!/usr/bin/env python3
"""Which chains set the length of a pass, and what their slabs look like: per-part device time of one pass of the benchmark workload
next to the byte breakdown of the part's staged state (emat_debug_slab_layout) and how it runs (staged whole / prefix only / from HBM)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from delphy_amd.sharding import ShardedEngine

area = int(sys.argv[1]) if len(sys.argv) > 1 else 9648
sc = make_scenario("C4")
world, rank = int(os.environ.get("EMAT_PROBE_WORLD", "1")), int(os.environ.get("EMAT_PROBE_RANK", "0"))   # one GPU of N, emulated (scripts/scale_probe.py)
eng = ShardedEngine(sc, num_parts=8192, seed=20261001, rank=rank, world=world, device_tree=world > 1, allreduce=lambda a, op: a, allgather_bytes=lambda b: [b])
eng.setup()
b = eng.backend
b.run_moves_per_part(1000); b.synchronize()
t0 = np.array([b.part_stats(p)["device_ticks"] for p in range(eng.num_local_parts)])
p0 = [b.part_stats(p)["proposed"] for p in range(eng.num_local_parts)]
b.run_moves_per_part(1000); b.synchronize(); ms = b.last_run_ms()
st = [b.part_stats(p) for p in range(eng.num_local_parts)]
us = (np.array([s["device_ticks"] for s in st]) - t0) / 100.0
topo = np.array([s["proposed"][3] + s["proposed"][4] - q[3] - q[4] for s, q in zip(st, p0)])
lay = [b.debug_slab_layout(p) for p in range(eng.num_local_parts)]
used = np.array([l["header"] + l["nodes"] + l["cells"] + l["trace"] + l["heap_used"] for l in lay])
prefix = np.array([l["header"] + l["nodes"] + l["cells"] + l["trace"] for l in lay])
mode = np.where(used + 1024 <= area, 0, np.where(prefix <= area, 1, 2))     # 0 staged whole, 1 prefix only, 2 from HBM (approximation of the kernel's rule)
print("kernel %.2f ms | chain ms: median %.2f p90 %.2f p99 %.2f max %.2f | max / median %.2f" % (ms, np.median(us) / 1e3, np.percentile(us, 90) / 1e3, np.percentile(us, 99) / 1e3, us.max() / 1e3, us.max() / np.median(us)))
for m, name in enumerate(("staged whole", "prefix only", "from HBM")):
    sel = mode == m
    if sel.any():
        print("  %-12s %5d parts (%.1f%%): chain ms median %.2f p99 %.2f max %.2f | topology moves per chain %.1f" % (name, sel.sum(), 100 * sel.mean(), np.median(us[sel]) / 1e3, np.percentile(us[sel], 99) / 1e3, us[sel].max() / 1e3, topo[sel].mean()))
print("  corr(chain time, topology moves drawn) = %.2f, corr(chain time, staged bytes) = %.2f, corr(chain time, cells) = %.2f" % (np.corrcoef(us, topo)[0, 1], np.corrcoef(us, used)[0, 1], np.corrcoef(us, [l["num_cells"] for l in lay])[0, 1]))
print("  slowest chains:")
for i in np.argsort(-us)[:14]:
    l = lay[i]
    print("   part %5d: %.2f ms, %2d topology moves, nodes %3d, cells %3d (%d B), lists %d B, staged %d B -> %s" % (i, us[i] / 1e3, topo[i], l["nodes"] // 64, l["num_cells"], l["cells"], l["heap_used"], used[i], ("staged whole", "prefix only", "from HBM")[mode[i]]))
# a simple model: time = a + b * topology moves
A = np.vstack([np.ones_like(topo), topo]).T.astype(float)
coef, *_ = np.linalg.lstsq(A[mode == 0], us[mode == 0], rcond=None)
print("  staged-whole chains: time = %.2f ms + %.3f ms per topology move (1 000 moves per chain, %.1f topology moves on average)" % (coef[0] / 1e3, coef[1] / 1e3, topo.mean()))
eng.close()
