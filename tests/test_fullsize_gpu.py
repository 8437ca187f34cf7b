"""BASELINE.json-size runs (config C4: 100k tips) checked through size-independent properties:
  * the incrementally maintained log-G / log-prior of every part equal a from-scratch recomputation on the device
    (the reference's own debug invariant, Subrun::check_derived_quantities, subrun.cpp:28-56) AND the oracle's
    from-scratch values on the downloaded trees;
  * every sampled downloaded part passes the reference's tree-integrity asserts (phylo_tree.cpp:18-136) in the oracle;
  * the same seed gives bit-identical results twice, and the LDS-staged and HBM-resident paths agree bit for bit;
  * the whole cycle repartition -> moves -> reassemble through the host driver keeps the full tree valid.
"""
import os

import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from delphy_amd.sharding import ShardedEngine
from helpers import configure, rel_close
from oracle_ffi import OracleEngine

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def c4():
    return make_scenario("C4")


def _run(sc, use_lds, moves=300):
    eng = ShardedEngine(sc, num_parts=8192, seed=20261001, use_lds=use_lds)
    eng.setup()
    eng.backend.run_moves_per_part(moves)
    eng.backend.synchronize()
    return eng


def test_c4_incremental_totals_match_recomputation_and_oracle(c4):
    eng = _run(c4, True)
    b = eng.backend
    st = eng.local_stats()
    assert st["bad_parts"] == 0 and st["moves_done"] == eng.total_parts * 300
    assert sum(st["accepted"]) > 0.05 * st["moves_done"] and st["accepted"][3] + st["accepted"][4] > 0
    sample = list(range(0, eng.total_parts, 97)) + [eng.root_part]
    inc = {p: b.part_derived(p, eng.local_sizes[p]) for p in sample}
    G_inc, A_inc = b.totals()
    b.recalc_derived()
    G_re, A_re = b.totals()
    assert rel_close(G_inc, G_re, 1e-9) and rel_close(A_inc, A_re, 1e-9), (G_inc, G_re, A_inc, A_re)
    # oracle: integrity + from-scratch derived quantities of the downloaded parts
    trees = [b.part_download(p) for p in sample]
    orc = OracleEngine(c4.num_sites)
    orc.set_ref_sequence(eng_ref(eng)); orc.set_hky(c4.mu, c4.kappa, c4.pi); orc.set_flags(c4.t_max_tip)
    orc.upload_parts(trees, [p == eng.root_part for p in sample], [1] * len(sample))
    for i, p in enumerate(sample):
        rc, msg = orc.part_check(i)
        assert rc == 0, "part %d: %s" % (p, msg)
        lam_o, nm_o, G_o, _ = orc.part_derived(i, trees[i].num_nodes)
        lam_g, nm_g, G_g, _ = inc[p]
        assert np.array_equal(nm_o, nm_g)
        assert rel_close(lam_g, lam_o, 1e-9) and rel_close(G_g, G_o, 1e-9), (p, G_g, G_o)
    orc.close(); eng.close()


def eng_ref(eng):
    run = d.EmatRun(None, eng.sc.tree, eng.sc.ref, eng.seed)
    run.set_num_parts(eng.num_parts_requested)
    run.repartition()
    _, ref = run.tree()
    run.close()
    return ref


def test_c4_deterministic_and_lds_equals_hbm(c4):
    a = _run(c4, True, 200)
    b = _run(c4, False, 200)
    c = _run(c4, True, 200)
    for p in list(range(0, a.total_parts, 211)) + [a.root_part]:
        ta, tb, tc = a.backend.part_download(p), b.backend.part_download(p), c.backend.part_download(p)
        for f in ("parent", "child0", "child1", "t", "mut_offset", "mut_site", "mut_from", "mut_to", "mut_t", "miss_offset", "miss_start", "miss_end", "mfs_offset", "mfs_site", "mfs_state"):
            assert np.array_equal(getattr(ta, f), getattr(tb, f)), (p, f)
            assert np.array_equal(getattr(ta, f), getattr(tc, f)), (p, f)
    assert a.backend.totals() == b.backend.totals() == c.backend.totals()
    a.close(); b.close(); c.close()


def test_run_driver_cycles_on_device():
    sc = make_scenario("C3", num_tips=3000, num_sites=29903, uncertain_tips=0.1)
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, 5)
    run.set_num_parts(128)
    run.set_hky(sc.mu, sc.kappa, sc.pi)
    run.set_pop_model(sc.pop)
    run.do_mcmc_steps(3 * 128 * 500, 128 * 500)          # three repartition -> moves -> reassemble cycles
    tree, ref = run.tree()
    assert tree.num_nodes == sc.tree.num_nodes
    chk = OracleEngine(sc.num_sites)
    sc.tree, sc.ref = tree, ref
    configure(chk, sc, ref, [tree], [True], [1], 0)
    rc, msg = chk.part_check(0)
    assert rc == 0, msg
    # tips keep their dates' bounds and the tree still has the same tips
    tips = tree.child0 == -1
    assert np.all(tree.t[tips] >= tree.t_min[tips] - 1e-2) and np.all(tree.t[tips] <= tree.t_max[tips] + 1e-2)
    chk.close(); run.close(); b.close()


def test_trajectory_parity_at_c3_full_size():
    """Config C3 as a whole (10 000 tips, 29 903 sites, skygrid): 400 parts x 3000 moves, every move of every part compared
    with the oracle (kind, node, accept flag exact; log-MH within 1e-9), then trees, counters and RNG consumption."""
    from helpers import run_parity
    sc = make_scenario("C3")
    run_parity(sc, 400, 3000, seed=101, trace=3000)


def test_trajectory_parity_on_the_benchmark_workload():
    """The benchmark itself, checked move for move: config C4 (100 000 tips) cut exactly as bench.py cuts it, one pass of
    1 000 moves on every one of the ~7 955 parts, each part's trace, tree, counters and RNG consumption against the oracle."""
    from helpers import run_parity
    sc = make_scenario("C4")
    run_parity(sc, 8192, 1000, seed=20261001, trace=1000)


@pytest.mark.parametrize("num_parts,moves", [(8, 20000), (64, 5000)])
def test_c4_reference_partition_count(num_parts, moves):
    """Config C4 as BASELINE.json states it: the 100 000-tip tree cut into 8 subtrees (the reference's policy of as many
    parts as workers, tools/delphy.cpp:130-132; run.cpp:682-693) -- parts of ~25 000 nodes, nothing staged in LDS,
    unlimited candidate scans over a whole part (about six per part in 20 000 moves) -- and into 64; every move of every part against the oracle."""
    from helpers import run_parity
    sc = make_scenario("C4")
    st = run_parity(sc, num_parts, moves, seed=20261001, trace=moves)
    assert st["moves_done"] == moves and st["proposed"][4] > moves // 64     # SPR1 moves, ~1 % of them with an unlimited scan


def test_trajectory_parity_at_c2_full_size():
    """Config C2 exactly as SURVEY 8(d) states it (1 610 tips, 18 959 sites, exponential growth with a minimum population):
    64 parts x 20 000 moves, every move of every part against the oracle."""
    from helpers import run_parity
    sc = make_scenario("C2")
    assert sc.num_tips == 1610 and sc.num_sites == 18959
    st = run_parity(sc, 64, 20000, seed=103, trace=20000)
    assert st["moves_done"] == 20000


def test_c5_one_gpu_sampled_trajectories_and_whole_run_properties():
    """Config C5 (1 000 000 tips) on one GPU.  All ~80 000 parts run 300 moves; every 41st part (and the root part) is
    replayed by the oracle from the same coalescent tables and compared move for move; over ALL parts: every chain ran
    to completion, and the incrementally maintained totals equal a from-scratch recomputation on the device."""
    from helpers import compare_part, split_parts
    moves = 300
    sc = make_scenario("C5")
    assert sc.num_tips == 1000000
    parts, incl, seeds, root_part, ref = split_parts(sc, 81920, 20261005)
    assert len(parts) > 60000
    gpu = d.EmatBackend(sc.num_sites, trace_moves=moves)
    orc = OracleEngine(sc.num_sites, trace_moves=moves)
    try:
        configure(gpu, sc, ref, parts, incl, seeds, root_part)
        configure(orc, sc, ref, parts, incl, seeds, root_part)
        gpu.run_moves_per_part(moves)
        gpu.synchronize()
        sample = sorted(set(list(range(0, len(parts), 41)) + [root_part]))
        counts = np.zeros(len(parts), np.int64); counts[sample] = moves
        orc.run_moves_counts(counts, threads=8)
        for p in sample:
            compare_part(gpu, orc, p, parts[p].num_nodes, moves, 1e-9, moves)
        acc = np.zeros(5, np.int64)
        for p in range(len(parts)):
            st = gpu.part_stats(p)
            assert st["status"] == 0 and st["moves_done"] == moves, (p, st)
            acc += np.array(st["accepted"])
        assert acc.sum() > 0.05 * moves * len(parts) and acc[3] + acc[4] > 0
        G_inc, A_inc = gpu.totals()
        gpu.recalc_derived()
        G_re, A_re = gpu.totals()
        assert rel_close(G_inc, G_re, 1e-9) and rel_close(A_inc, A_re, 1e-9), (G_inc, G_re, A_inc, A_re)
    finally:
        gpu.close(); orc.close()


def test_c5_whole_cycles_with_the_tree_resident_in_hbm():
    """Config C5 through two whole cycles (repartition -> moves -> reassemble) without the tree ever leaving the device
    (SURVEY 8(f).2): 2 000 000 nodes cut into ~80 000 parts by kernels and gathered back; what comes back is a valid EMAT
    with the same tips, and a pass on a fresh partition keeps incremental and recomputed totals together."""
    sc = make_scenario("C5")
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, 20261005)
    run.set_num_parts(81920); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
    run.set_max_part_nodes(-1)       # the part-size limit bench.py's whole cycles opt into (the library's default is the reference's rule)
    run.set_device_tree(True)
    try:
        run.do_mcmc_steps(2 * 81920 * 100, 81920 * 100)
        tree, ref = run.tree()
        assert tree.num_nodes == sc.tree.num_nodes
        tips = tree.child0 == -1
        assert np.array_equal(tips, sc.tree.child0 == -1) and np.array_equal(tree.t_min, sc.tree.t_min) and np.array_equal(tree.t_max, sc.tree.t_max)
        assert not np.array_equal(tree.parent, sc.tree.parent) and not np.array_equal(tree.t, sc.tree.t)
        # structure: every non-root node hangs below an older parent that lists it as a child
        nonroot = np.arange(tree.num_nodes) != tree.root
        par = tree.parent[nonroot]
        assert np.all(par >= 0) and tree.parent[tree.root] == -1
        assert np.all((tree.child0[par] == np.flatnonzero(nonroot)) | (tree.child1[par] == np.flatnonzero(nonroot)))
        assert np.all(tree.t[par] <= tree.t[nonroot])
        # mutations stay on their branches, sorted in time
        nm = np.diff(tree.mut_offset)
        owner = np.repeat(np.arange(tree.num_nodes), nm)
        keep = owner != tree.root
        assert np.all(tree.mut_t[keep] >= tree.t[tree.parent[owner[keep]]]) and np.all(tree.mut_t[keep] <= tree.t[owner[keep]])
        assert np.all(tree.mut_from != tree.mut_to)
        run.repartition(); n, _ = run.num_parts(); run.run_moves(n * 50); b.synchronize()
        inc = b.totals(); b.recalc_derived(); rec = b.totals()
        assert rel_close(inc[0], rec[0], 1e-9) and rel_close(inc[1], rec[1], 1e-9), (inc, rec)
        run.reassemble()
    finally:
        run.close(); b.close()


def _sharded_gpu_worker(rank, world, port, out_dir, cycles, moves, device_tree=False):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)     # two ranks share cuda:0 here, which RCCL refuses: the collectives go over gloo
    from test_sharding_gloo import gloo_collectives, _tree_fields
    allreduce, allgather_bytes = gloo_collectives()
    sc = make_scenario("C3", num_tips=1500, num_sites=29903, uncertain_tips=0.1)
    eng = ShardedEngine(sc, num_parts=48, seed=97, rank=rank, world=world, device=0, allreduce=allreduce, allgather_bytes=allgather_bytes, device_tree=device_tree)
    res = {}
    for cyc in range(cycles):
        G, A = eng.cycle(moves)
        res["c%d_totals" % cyc] = np.array([G, A])
    t, ref = eng.tree()
    for k, v in _tree_fields(t).items():
        res[k] = v
    res["root"] = np.array([t.root]); res["ref"] = ref
    tips = t.child0 == -1
    res["grid_prior"] = np.array([eng.scalable_coalescent_log_prior(float(np.max(t.t[tips])))])
    eng.repartition()
    res["Ttwiddle_l"] = eng.Ttwiddle_l()
    np.savez(os.path.join(out_dir, "gpu_rank%d.npz" % rank), **res)
    eng.close()
    dist.destroy_process_group()


def test_two_process_cycles_on_one_gpu_match_the_single_process_run(tmp_path):
    """The multi-rank control flow with real moves: two processes (sharing the one GPU of the test box, collectives over
    gloo) run three cycles repartition -> staged coalescent build -> moves on their own parts -> exchange of the parts ->
    reassemble.  Both ranks must end with the same tree, and it must be the tree a single process computes: the chains
    are per part and seeded per part, so sharding changes nothing but the association of two cross-rank sums."""
    import torch.multiprocessing as mp
    from test_sharding_gloo import _tree_fields
    cycles, moves = 3, 48 * 400
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_sharded_gpu_worker, args=(2, port, str(tmp_path), cycles, moves), nprocs=2, join=True)
    z = [np.load(os.path.join(str(tmp_path), "gpu_rank%d.npz" % r)) for r in range(2)]
    sc = make_scenario("C3", num_tips=1500, num_sites=29903, uncertain_tips=0.1)
    single = ShardedEngine(sc, num_parts=48, seed=97)
    tot = [single.cycle(moves) for _ in range(cycles)]
    t, ref = single.tree()
    tips = t.child0 == -1
    grid = single.scalable_coalescent_log_prior(float(np.max(t.t[tips])))
    single.repartition()
    ttw = single.Ttwiddle_l()
    for r in range(2):
        assert rel_close(z[r]["Ttwiddle_l"], ttw, 1e-9)
        assert int(z[r]["root"][0]) == t.root and np.array_equal(z[r]["ref"], ref)
        for k, v in _tree_fields(t).items():
            if k in ("t", "mut_t"):
                fin = np.abs(v) < 1e300
                assert np.array_equal(fin, np.abs(z[r][k]) < 1e300) and rel_close(z[r][k][fin], v[fin], 1e-9), (r, k)
            else:
                assert np.array_equal(z[r][k], v), (r, k)
        for cyc in range(cycles):
            assert rel_close(z[r]["c%d_totals" % cyc], np.array(tot[cyc]), 1e-9), (r, cyc, z[r]["c%d_totals" % cyc], tot[cyc])
        assert rel_close(z[r]["grid_prior"], np.array([grid]), 1e-9)
    for k in _tree_fields(t):
        assert np.array_equal(z[0][k], z[1][k]), k       # the two ranks hold bit-identical trees
    single.close()


def test_two_process_cycles_with_the_tree_on_the_devices(tmp_path):
    """The same two processes with the whole tree resident in each one's HBM (SURVEY 8(f).2 + 8(e)): every rank cuts its own
    block of parts out of its own copy, and after the moves the ranks exchange node updates -- what their parts own and link --
    instead of part trees.  Chains, seeds and kernels are those of the single-process run with the tree in HBM, so both
    ranks must end with exactly its tree and reference sequence, cycle totals included."""
    import torch.multiprocessing as mp
    from test_sharding_gloo import _tree_fields
    cycles, moves = 3, 48 * 400
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_sharded_gpu_worker, args=(2, port, str(tmp_path), cycles, moves, True), nprocs=2, join=True)
    z = [np.load(os.path.join(str(tmp_path), "gpu_rank%d.npz" % r)) for r in range(2)]
    sc = make_scenario("C3", num_tips=1500, num_sites=29903, uncertain_tips=0.1)
    single = ShardedEngine(sc, num_parts=48, seed=97, device_tree=True)
    tot = [single.cycle(moves) for _ in range(cycles)]
    t, ref = single.tree()
    assert not np.array_equal(t.parent, sc.tree.parent)
    for r in range(2):
        assert int(z[r]["root"][0]) == t.root and np.array_equal(z[r]["ref"], ref)
        for k, v in _tree_fields(t).items():
            assert np.array_equal(z[r][k], v), (r, k)
        for cyc in range(cycles):
            assert rel_close(z[r]["c%d_totals" % cyc], np.array(tot[cyc]), 1e-9), (r, cyc)
    single.close()


def _rccl_worker(rank, port, out_dir):
    import torch
    import torch.distributed as dist
    from delphy_amd.sharding import _torch_collectives
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    allreduce, allgather_bytes = _torch_collectives("cuda:0")
    x = np.array([1.5, -2.25, 1e300, -1e-300])
    res = {"min": allreduce(x, "min"), "max": allreduce(x, "max"), "sum": allreduce(x, "sum"),
           "isum": allreduce(np.array([7, -3, 2**40], np.int64), "sum")}
    blob = np.frombuffer(os.urandom(100003), np.uint8).copy()
    got = allgather_bytes(blob)
    res["gather_ok"] = np.array([len(got) == 1 and np.array_equal(got[0], blob)])
    empty = allgather_bytes(np.zeros(0, np.uint8))
    res["empty_ok"] = np.array([len(empty) == 1 and empty[0].shape[0] == 0])
    from delphy_amd.sharding import _torch_device_allgather
    dev = _torch_device_allgather("cuda:0")(blob.shape[0], lambda t: t[: blob.shape[0]].copy_(torch.from_numpy(blob)))
    res["device_gather_ok"] = np.array([len(dev) == 1 and dev[0].is_cuda and np.array_equal(dev[0].cpu().numpy(), blob)])
    dist.barrier(); torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rccl.npz"), **res)
    dist.destroy_process_group()


def test_the_collectives_of_a_sharded_run_over_rccl(tmp_path):
    """What bench.py --gpus N and ShardedEngine put on the wire, through RCCL itself (backend "nccl") rather than gloo: f64 MIN /
    MAX / SUM, int64 SUM, the variable-length byte all-gather (empty contributions included) and the barrier.  One rank -- the
    test box has one GPU and RCCL refuses two ranks on a device -- so this pins dtype / operator support and the device
    plumbing, not the transport."""
    import torch.multiprocessing as mp
    port = 37500 + (os.getpid() % 2000)
    mp.spawn(_rccl_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    z = np.load(os.path.join(str(tmp_path), "rccl.npz"))
    x = np.array([1.5, -2.25, 1e300, -1e-300])
    for k in ("min", "max", "sum"):
        assert np.array_equal(z[k], x), k
    assert np.array_equal(z["isum"], np.array([7, -3, 2**40], np.int64))
    assert bool(z["gather_ok"][0]) and bool(z["empty_ok"][0]) and bool(z["device_gather_ok"][0])


def test_node_exchange_of_a_sharded_run_through_device_buffers():
    """The exchange after the moves of a sharded run with the tree in HBM, on DEVICE buffers: emat_tree_export_nodes writes into
    device memory (what a rank would hand to RCCL as it is), emat_tree_apply_nodes reads device memory (what the all-gather
    delivers).  Three ranks' worth of engines in this one process (the test box has one GPU), their buffers torch tensors on
    cuda:0, stepped through three cycles by hand exactly as ShardedEngine.reassemble does: every rank ends with the tree of
    the single-process run."""
    import torch
    from test_sharding_gloo import _tree_fields
    world, cycles, moves = 3, 3, 48 * 300
    sc = make_scenario("C3", num_tips=1500, num_sites=29903, uncertain_tips=0.1)
    single = ShardedEngine(sc, num_parts=48, seed=97, device_tree=True)
    for _ in range(cycles):
        single.cycle(moves)
    want, want_ref = single.tree()
    single.close()
    ident = lambda a, op: a
    engs = [ShardedEngine(sc, num_parts=48, seed=97, rank=r, world=world, device=0, allreduce=ident, allgather_bytes=lambda b: [b], device_tree=True) for r in range(world)]
    try:
        for cyc in range(cycles):
            for e in engs:
                e.repartition(); e.run.run_moves_sharded(moves)
            rds = [e.backend.tree_root_deltas() for e in engs]
            owners = [rd for rd in rds if rd is not None]
            assert len(owners) == 1
            site, frm, to = owners[0]
            for e in engs:
                e.backend.tree_gather_local(site, frm, to)
            bufs = []
            for e in engs:
                need = e.backend.tree_export_size()
                t = torch.empty(need, dtype=torch.uint8, device="cuda:0")
                assert e.backend.tree_export_nodes_into(t.data_ptr(), need) == need
                bufs.append(t)
            torch.cuda.synchronize()
            for r, e in enumerate(engs):
                for q in range(world):
                    if q != r:
                        e.backend.tree_apply_nodes_at(bufs[q].data_ptr(), int(bufs[q].numel()))
                e.backend.tree_reassemble_end()
                e.run.note_device_reassembled(site, to)
        for r, e in enumerate(engs):
            t, ref = e.tree()
            assert t.root == want.root and np.array_equal(ref, want_ref), r
            for k, v in _tree_fields(want).items():
                assert np.array_equal(_tree_fields(t)[k], v), (r, k)
    finally:
        for e in engs:
            e.close()
