python -m pytest tests -m gpu -x -q 2>&1 | tail -6
python - <<'PY'
import sys, time; sys.path.insert(0, "."); sys.path.insert(0, "tests")
import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from delphy_amd.sharding import ShardedEngine
sc = make_scenario("C4"); eng = ShardedEngine(sc, num_parts=8192, seed=20261001); eng.setup()
eng.backend.run_moves_per_part(100); eng.backend.synchronize()
for i in range(3):
    t0 = time.perf_counter(); T, M, nm = eng.global_stats(1); dt = time.perf_counter() - t0
print("C4 global stats in %.2f ms: num_muts %d, Ttwiddle %s" % (dt * 1e3, nm, T.tolist()))
eng.close()
PY
