"""One process, several backends (include/emat_host.h, emat_run_create_multi; delphy_amd/csrc/emat_multi.cpp): the C++ counterpart of
the reference's in-process seam, Run::run_local_moves handing its Subruns to a thread pool (run.cpp:682-693).  The 8-GPU node is the
driver's; what can be checked on a one-GPU box is (a) two backends that SHARE the device, their exchange going through host buffers
(RCCL takes one rank per device), and (b) one backend whose exchange goes through RCCL itself (ncclCommInitAll over one device,
ncclAllGather / ncclAllReduce on the buffers the kernels wrote) -- both against the single-backend run of the same seed: every
shard must end every cycle with exactly that tree."""
import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import make_scenario

FIELDS = ("parent", "child0", "child1", "t", "t_min", "t_max", "mut_offset", "mut_site", "mut_from", "mut_to", "mut_t",
          "miss_offset", "miss_start", "miss_end", "mfs_offset", "mfs_site", "mfs_state")


def test_a_multi_run_needs_devices():
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("CPU-side check")
    sc = make_scenario("C1", num_tips=20, num_sites=300)
    with pytest.raises(d.EmatError, match="NO_DEVICE"):
        d.EmatMultiRun([0, 1], sc.tree, sc.ref, 1)


def _single(sc, parts, seed, cycles, moves):
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, seed)
    run.set_num_parts(parts); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop); run.set_device_tree(True)
    out = []
    for _ in range(cycles):
        run.repartition()
        run.run_moves(parts * moves + 5)
        tot = b.totals()
        run.reassemble()
        out.append((run.tree(), tot))
    run.close(); b.close()
    return out


def _multi(sc, devices, exchange, parts, seed, cycles, moves):
    m = d.EmatMultiRun(devices, sc.tree, sc.ref, seed, exchange=exchange)
    m.set_num_parts(parts); m.set_hky(sc.mu, sc.kappa, sc.pi); m.set_pop_model(sc.pop)
    out = []
    try:
        for c in range(cycles):
            m.repartition()
            m.run_moves(parts * moves + 5)         # (the same count in both runs whatever the number of parts: the part-size limit makes it depend on the tree)
            m.check_derived(1.0)                       # the reference's paranoid check on every shard (subrun.cpp:28-56)
            tot = m.totals()
            m.reassemble()
            out.append(([m.tree(s) for s in range(m.num_shards)], tot))
        return out, m.exchange
    finally:
        m.close()


def _same(a, b, what):
    (ta, ra), (tb, rb) = a, b
    assert ta.root == tb.root and np.array_equal(ra, rb), what
    for f in FIELDS:
        x, y = getattr(ta, f), getattr(tb, f)
        assert x.shape == y.shape and np.array_equal(x, y), "%s: %s differs" % (what, f)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,exchange,cycles", [([0, 0], "host", 3), ([0, 0, 0], "auto", 3), ([0], "rccl", 3), ([0] * 8, "host", 5)])
def test_shards_of_one_process_equal_the_single_backend_run(devices, exchange, cycles):
    """(The last case is the shape of the driver's 8-GPU node: eight shards, five cycles, every shard's copy of the tree equal to the
    single-backend run's after every cycle -- on the one GPU of the test box, so through host buffers.)"""
    sc = make_scenario("C3", num_tips=3000, num_sites=29903, uncertain_tips=0.1)
    parts, seed, moves = 128, 17, 400
    want = _single(sc, parts, seed, cycles, moves)
    got, how = _multi(sc, devices, exchange, parts, seed, cycles, moves)
    assert how.startswith("RCCL") == (exchange == "rccl"), how
    for c in range(cycles):
        (tree1, tot1), (trees, tot) = want[c], got[c]
        for s, t in enumerate(trees):
            _same(t, tree1, "cycle %d shard %d" % (c, s))
        assert abs(tot[0] - tot1[0]) <= 1e-10 * abs(tot1[0]) and abs(tot[1] - tot1[1]) <= 1e-10 * abs(tot1[1]), (tot, tot1)   # shard sums associate differently
    assert not np.array_equal(want[-1][0][0].parent, sc.tree.parent)      # the cycles did change the tree


def _two_devices():
    import torch
    return torch.cuda.device_count() >= 2      # (counting devices does not initialise the GPU on this image)


@pytest.mark.gpu
@pytest.mark.parametrize("devices,exchange,cycles", [([0, 1], "rccl", 3), ([0, 1], "auto", 3), ([0, 1], "host", 3)])
def test_shards_on_two_devices_equal_the_single_backend_run(devices, exchange, cycles):
    """The first test that runs ncclCommInitAll with n > 1 and the grouped ncclAllGather / ncclAllReduce BETWEEN devices
    (emat_multi.cpp): collected on every box, skipped where there is one GPU (every lease so far), so that the first node with two runs
    it by itself.  Same statement as above: every shard ends every cycle with exactly the single-backend tree; `auto` must have picked
    RCCL (one shard per device), `host` exchanges the same records through host buffers."""
    if not _two_devices():
        pytest.skip("needs two GPUs (the test box has one)")
    sc = make_scenario("C3", num_tips=3000, num_sites=29903, uncertain_tips=0.1)
    parts, seed, moves = 128, 17, 400
    want = _single(sc, parts, seed, cycles, moves)
    got, how = _multi(sc, devices, exchange, parts, seed, cycles, moves)
    assert how.startswith("RCCL") == (exchange in ("rccl", "auto")), how
    for c in range(cycles):
        (tree1, tot1), (trees, tot) = want[c], got[c]
        for s, t in enumerate(trees):
            _same(t, tree1, "cycle %d shard %d" % (c, s))
        assert abs(tot[0] - tot1[0]) <= 1e-10 * abs(tot1[0]) and abs(tot[1] - tot1[1]) <= 1e-10 * abs(tot1[1]), (tot, tot1)


@pytest.mark.gpu
def test_do_mcmc_steps_of_a_multi_run():
    sc = make_scenario("C2", num_tips=800, num_sites=6000)
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, 5)
    run.set_num_parts(40); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop); run.set_device_tree(True)
    run.do_mcmc_steps(4 * 40 * 300, 40 * 300)
    want = run.tree(); run.close(); b.close()
    m = d.EmatMultiRun([0, 0], sc.tree, sc.ref, 5, exchange="host")
    m.set_num_parts(40); m.set_hky(sc.mu, sc.kappa, sc.pi); m.set_pop_model(sc.pop); m.set_paranoid(True)
    m.do_mcmc_steps(4 * 40 * 300, 40 * 300)
    for s in range(2):
        _same(m.tree(s), want, "shard %d" % s)
    m.close()
