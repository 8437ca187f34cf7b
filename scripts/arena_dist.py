import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, ctypes as C
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from delphy_amd.sharding import ShardedEngine
sc = make_scenario("C4")
eng = ShardedEngine(sc, num_parts=8192, seed=20261001)
eng.setup()
eng.backend.run_moves_per_part(1000); eng.backend.synchronize()
n = eng.num_local_parts
out = np.zeros(n, np.int32)
lib = d.load_library(); lib.emat_debug_arena_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
assert lib.emat_debug_arena_bytes(eng.backend.handle, out.ctypes.data_as(C.POINTER(C.c_int32))) == 0
a = out[out >= 0]
st = [eng.backend.part_stats(p) for p in range(n)]
dur = np.array([s["device_ticks"] for s in st]) / 1e5
print("main-class parts", a.size, "| arena bytes: p1 %d p5 %d p10 %d p25 %d p50 %d" % tuple(np.percentile(a, [1, 5, 10, 25, 50])))
for lim in (512, 1024, 1536, 2048, 3072):
    m = (out >= 0) & (out < lim)
    print("arena < %4d: %5d parts (%.1f%%), their mean chain %.2f ms against %.2f ms of the rest" % (lim, m.sum(), 100.0 * m.sum() / a.size, dur[m].mean() if m.any() else 0, dur[(out >= lim)].mean()))
