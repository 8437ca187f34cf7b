"""N > 1 path on CPU: two gloo ranks shard the parts and exchange the coalescent grid exactly as bench.py does
over RCCL; each rank's result must equal the single-process build."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from delphy_amd.scenarios import make_scenario
    from delphy_amd.sharding import ShardedEngine, block_range

    def allreduce(arr, op):
        t = torch.from_numpy(np.ascontiguousarray(arr).copy())
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op])
        return t.numpy()

    sc = make_scenario("C2", num_tips=400, num_sites=2000, uncertain_tips=0.2)
    eng = ShardedEngine(sc, num_parts=12, seed=77, rank=rank, world=world, device=-1, allreduce=allreduce)
    eng.setup()
    lo, hi = block_range(eng.total_parts, rank, world)
    assert (eng.part_lo, eng.part_hi) == (lo, hi) and eng.num_local_parts == hi - lo
    res = {"lo": lo, "hi": hi, "total": eng.total_parts}
    for p in range(eng.num_local_parts):
        c = eng.backend.part_coalescent(p)
        for k, v in c.items():
            res["%d_%s" % (lo + p, k)] = np.asarray(v)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **res)
    eng.close()
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    sys.path.insert(0, ROOT)
    from delphy_amd.scenarios import make_scenario
    from delphy_amd.sharding import ShardedEngine, block_range
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sc = make_scenario("C2", num_tips=400, num_sites=2000, uncertain_tips=0.2)
    single = ShardedEngine(sc, num_parts=12, seed=77, rank=0, world=1, device=-1)
    single.setup()
    covered = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        lo, hi = int(z["lo"]), int(z["hi"])
        assert (lo, hi) == block_range(single.total_parts, r, world) and int(z["total"]) == single.total_parts
        for p in range(lo, hi):
            ref = single.backend.part_coalescent(p)
            w = ref["num_active_parts"] >= 0
            assert np.array_equal(z["%d_num_active_parts" % p], ref["num_active_parts"])
            assert np.array_equal(z["%d_k_bar_p" % p], ref["k_bar_p"])                 # local quantity: identical
            assert np.array_equal(z["%d_popsize_bar" % p][w], ref["popsize_bar"][w])
            # the Gaussian means depend on k_bar, whose cross-rank sum is associated differently: equal to rounding
            assert np.allclose(z["%d_k_twiddle_bar_p" % p], ref["k_twiddle_bar_p"], rtol=0, atol=1e-9)
            assert np.allclose(z["%d_k_twiddle_bar" % p][w], ref["k_twiddle_bar"][w], rtol=0, atol=1e-8)
            covered += 1
    assert covered == single.total_parts
    single.close()


def test_block_range_partitions_everything():
    from delphy_amd.sharding import block_range
    for n in (1, 7, 8, 9, 7860):
        for w in (1, 2, 4, 8):
            spans = [block_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
