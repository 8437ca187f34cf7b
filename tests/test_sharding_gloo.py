"""N > 1 path on CPU: two gloo ranks shard the parts and exchange the coalescent grid exactly as bench.py does
over RCCL; each rank's result must equal the single-process build.  The second test runs the whole
multi-rank CYCLE of include/emat_host.h -- repartition, staged coalescent build, exchange of the parts, reassemble --
twice over on host-only handles (no moves: those need a GPU, tests/test_fullsize_gpu.py runs the same cycle with them)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gloo_collectives():
    def allreduce(arr, op):
        t = torch.from_numpy(np.ascontiguousarray(arr).copy())
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op])
        return t.numpy()

    def allgather_bytes(buf):
        world = dist.get_world_size()
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([buf.shape[0]], dtype=torch.int64))
        n = [int(x.item()) for x in sizes]
        padded = torch.zeros(max(n), dtype=torch.uint8); padded[: buf.shape[0]] = torch.from_numpy(buf)
        out = [torch.zeros(max(n), dtype=torch.uint8) for _ in range(world)]
        dist.all_gather(out, padded)
        return [o[: n[r]].numpy() for r, o in enumerate(out)]
    return allreduce, allgather_bytes


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from delphy_amd.scenarios import make_scenario
    from delphy_amd.sharding import ShardedEngine, block_range
    allreduce, allgather_bytes = gloo_collectives()

    sc = make_scenario("C2", num_tips=400, num_sites=2000, uncertain_tips=0.2)
    eng = ShardedEngine(sc, num_parts=12, seed=77, rank=rank, world=world, device=-1, allreduce=allreduce, allgather_bytes=allgather_bytes)
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


def _tree_fields(t):
    return {f: np.asarray(getattr(t, f)) for f in ("parent", "child0", "child1", "t", "t_min", "t_max", "mut_offset", "mut_site", "mut_from", "mut_to", "mut_t",
                                                   "miss_offset", "miss_start", "miss_end", "mfs_offset", "mfs_site", "mfs_state")}


def _cycle_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import delphy_amd as d
    from delphy_amd.scenarios import make_scenario
    from delphy_amd.sharding import ShardedEngine
    allreduce, allgather_bytes = gloo_collectives()
    sc = make_scenario("C3", num_tips=500, num_sites=3000)
    eng = ShardedEngine(sc, num_parts=16, seed=91, rank=rank, world=world, device=-1, allreduce=allreduce, allgather_bytes=allgather_bytes)
    res = {}
    for cyc in range(2):
        eng.repartition()                       # cut, upload the local block, staged coalescent build with three all-reduces
        assert eng.part_hi - eng.part_lo == eng.num_local_parts and 0 < eng.num_local_parts < eng.total_parts
        if cyc == 0:
            with pytest_raises(d.EmatError):    # the other rank's parts have not arrived: reassembling now must fail loudly
                eng.run.reassemble()
        eng.reassemble()                        # all-gather of the serialised parts, then the gather into the whole tree
        t, ref = eng.tree()
        for k, v in _tree_fields(t).items():
            res["c%d_%s" % (cyc, k)] = v
        res["c%d_root" % cyc] = np.array([t.root]); res["c%d_ref" % cyc] = ref
        res["c%d_parts" % cyc] = np.array([eng.total_parts, eng.part_lo, eng.part_hi, eng.root_part])
    with pytest_raises(d.EmatError):
        eng.run.unpack_parts(np.arange(100, dtype=np.uint8))   # garbage is rejected, not applied
    with pytest_raises(d.EmatError):
        eng.run.do_mcmc_steps(10, 10)                           # the single-process cycle refuses a sharded run
    np.savez(os.path.join(out_dir, "cycle%d.npz" % rank), **res)
    eng.close()
    dist.destroy_process_group()


class pytest_raises:
    def __init__(self, exc): self.exc = exc
    def __enter__(self): return self
    def __exit__(self, et, ev, tb):
        assert et is not None and issubclass(et, self.exc), "expected %s" % self.exc
        return True


def test_two_rank_repartition_exchange_reassemble_cycles(tmp_path):
    sys.path.insert(0, ROOT)
    from delphy_amd.scenarios import make_scenario
    from delphy_amd.sharding import ShardedEngine
    world = 2
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_cycle_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    sc = make_scenario("C3", num_tips=500, num_sites=3000)
    single = ShardedEngine(sc, num_parts=16, seed=91, device=-1)
    z = [np.load(os.path.join(str(tmp_path), "cycle%d.npz" % r)) for r in range(world)]
    for cyc in range(2):
        single.repartition(); single.reassemble()
        t, ref = single.tree()
        for r in range(world):
            assert int(z[r]["c%d_parts" % cyc][0]) == single.total_parts and int(z[r]["c%d_parts" % cyc][3]) == single.root_part
            assert int(z[r]["c%d_root" % cyc][0]) == t.root and np.array_equal(z[r]["c%d_ref" % cyc], ref)
            for k, v in _tree_fields(t).items():
                assert np.array_equal(z[r]["c%d_%s" % (cyc, k)], v), (cyc, r, k)      # no moves: every rank rebuilds exactly the single-process tree
        assert tuple(z[0]["c%d_parts" % cyc][1:3]) != tuple(z[1]["c%d_parts" % cyc][1:3])
    single.close()
