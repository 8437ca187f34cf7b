"""What one GPU of N would do in a resident pass of bench.py's workload, measured on ONE GPU: rank r of N holds block r of the
same partition (tree on the device: no collective is needed to set a rank up), runs its pass alone, and the pass of the
N-GPU job lasts as long as its slowest rank's.  Usage: python scripts/scale_probe.py [workload] [parts] [moves] [N ...]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import delphy_amd as d  # noqa: E402,F401
from delphy_amd.scenarios import make_scenario
from delphy_amd.sharding import ShardedEngine

workload = sys.argv[1] if len(sys.argv) > 1 else "C4"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
moves = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
worlds = [int(x) for x in sys.argv[4:]] or [1, 2, 4, 8]
limit = int(os.environ.get("EMAT_PROBE_MAX_PART_NODES", "0"))           # the run driver's part-size limit (-1 = its default, 0 = the reference's rule)
total_moves = int(os.environ.get("EMAT_PROBE_TOTAL_MOVES", "0"))        # > 0: this many moves per pass over all parts (moves per part = total / parts)
sc = make_scenario(workload)
out = []
for world in worlds:
    per_rank = []
    for rank in range(world):
        eng = ShardedEngine(sc, num_parts=parts, seed=20261001, rank=rank, world=world, device=0, device_tree=True, max_part_nodes=limit,
                            allreduce=lambda a, op: a, allgather_bytes=lambda b: [b])
        eng.setup()
        if total_moves > 0:
            moves = max(1, total_moves // eng.total_parts)
        eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
        ms = []
        for _ in range(3):
            t0 = time.perf_counter()
            eng.backend.run_moves_per_part(moves); eng.backend.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
        sizes = np.array(eng.local_sizes)
        per_rank.append({"rank": rank, "parts": eng.num_local_parts, "ms": float(np.median(ms)), "max_part_nodes": int(sizes.max()), "has_root": bool(eng.local_root >= 0)})
        total = eng.total_parts
        eng.close()
    worst = max(p["ms"] for p in per_rank)
    row = {"world": world, "total_parts": total, "moves_per_part": moves, "max_part_nodes": limit, "pass_ms": worst, "moves_per_s": total * moves / worst * 1e3, "ranks": per_rank}
    print(json.dumps(row), flush=True)
    out.append(row)
base = out[0]["moves_per_s"] / out[0]["world"]
for r in out:
    print("N=%d  pass %.2f ms  %.1f M moves/s  x%.2f" % (r["world"], r["pass_ms"], r["moves_per_s"] / 1e6, r["moves_per_s"] / out[0]["moves_per_s"]))
