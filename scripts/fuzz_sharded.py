"""Hand-run hunt (one GPU, gloo): random scenarios through sharded cycles in WORLD processes sharing cuda:0, with the tree on
the host and with the tree in HBM, against the single-process run.  Usage: python scripts/fuzz_sharded.py SEED CASES [WORLD]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))   # (tests/: the gloo collectives of test_sharding_gloo)


def scenario(seed, case):
    from delphy_amd.scenarios import random_scenario
    rng = np.random.default_rng([seed, case])
    sc, nu_l, evo, what = random_scenario(rng, case * 4 + int(rng.integers(0, 4)), max_tips=700)   # (site rates / two partitions are not wired through ShardedEngine: the scenario's own model is used)
    parts = int(rng.choice([3, 5, 12, 40]))
    moves = int(rng.choice([600, 4000, 20 * sc.tree.num_nodes]))
    return sc, parts, moves, int(rng.integers(1, 10**6)), what


def worker(rank, world, port, out_dir, seed, case, device_tree):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from delphy_amd.sharding import ShardedEngine
    from test_sharding_gloo import gloo_collectives, _tree_fields
    ar, ag = gloo_collectives()
    sc, parts, moves, s, what = scenario(seed, case)
    res = {}
    try:
        eng = ShardedEngine(sc, num_parts=parts, seed=s, rank=rank, world=world, device=0, allreduce=ar, allgather_bytes=ag, device_tree=device_tree)
        for cyc in range(3):
            G, A = eng.cycle(moves)
            res["c%d" % cyc] = np.array([G, A])
        t, ref = eng.tree()
        res.update(_tree_fields(t)); res["root"] = np.array([t.root]); res["ref"] = ref
        eng.close()
    except Exception as ex:
        res["error"] = np.array([str(ex)])
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), **res)
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    from delphy_amd.sharding import ShardedEngine
    from test_sharding_gloo import _tree_fields
    seed, cases = int(sys.argv[1]), int(sys.argv[2])
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    bad = 0
    for case in range(cases):
        sc, parts, moves, s, what = scenario(seed, case)
        for device_tree in (False, True):
            single = ShardedEngine(sc, num_parts=parts, seed=s, device_tree=device_tree)
            try:
                tot = [single.cycle(moves) for _ in range(3)]
            except Exception as ex:
                print("SINGLE FAILED", what, "parts", parts, "device_tree", device_tree, ex, flush=True); bad += 1; single.close(); continue
            t, ref = single.tree()
            total_parts = single.total_parts
            single.close()
            if total_parts < world:
                continue   # fewer parts than processes is refused (tested elsewhere)
            out = tempfile.mkdtemp()
            mp.spawn(worker, args=(world, 36000 + (os.getpid() + case * 2 + int(device_tree)) % 2000, out, seed, case, device_tree), nprocs=world, join=True)
            ok = True
            zs = [np.load(os.path.join(out, "r%d.npz" % r)) for r in range(world)]
            if any("error" in z.files and "fewer parts than processes" in str(z["error"][0]) for z in zs):
                print("skip case %d %s: a later cycle's partition has fewer parts than processes (refused, loudly)" % (case, what), flush=True)
                continue
            for r in range(world):
                z = zs[r]
                if "error" in z.files:
                    ok = False; print("  rank", r, "error:", z["error"][0]); continue
                tol = 0.0 if device_tree else 1e-9
                for k, v in _tree_fields(t).items():
                    same = np.array_equal(z[k], v) if (device_tree or k not in ("t", "mut_t")) else np.allclose(z[k][np.abs(v) < 1e300], v[np.abs(v) < 1e300], rtol=1e-9, atol=0)
                    if not same: ok = False; print("  rank", r, "field", k, "differs")
                if int(z["root"][0]) != t.root or not np.array_equal(z["ref"], ref): ok = False; print("  rank", r, "root / ref differ")
                for cyc in range(3):
                    if not np.allclose(z["c%d" % cyc], np.array(tot[cyc]), rtol=1e-9, atol=0): ok = False; print("  rank", r, "cycle", cyc, "totals", z["c%d" % cyc], tot[cyc])
            print("%s case %d %s, %d parts (%d made), %d moves, device_tree %s" % ("ok  " if ok else "FAIL", case, what, parts, total_parts, moves, device_tree), flush=True)
            bad += 0 if ok else 1
    print("failures:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
