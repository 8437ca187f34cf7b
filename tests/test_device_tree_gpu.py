"""the whole tree resident in HBM (emat_tree_* / emat_run_set_device_tree).  The checker is the host cycle
(Run::repartition / Run::reassemble restated in emat_run.cpp, itself checked against the oracle in test_host_driver.py and
test_fullsize_gpu.py): same seeds => same partitions, same slabs, same moves, same trees, bit for bit."""
import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import make_scenario

pytestmark = pytest.mark.gpu

FIELDS = ("parent", "child0", "child1", "t", "t_min", "t_max", "mut_offset", "mut_site", "mut_from", "mut_to", "mut_t",
          "miss_offset", "miss_start", "miss_end", "mfs_offset", "mfs_site", "mfs_state")


def _run(sc, seed, parts, device_tree, max_part_nodes=0):
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, seed)
    run.set_num_parts(parts)
    if max_part_nodes:
        run.set_max_part_nodes(max_part_nodes)
    run.set_hky(sc.mu, sc.kappa, sc.pi)
    run.set_pop_model(sc.pop)
    if device_tree:
        run.set_device_tree(True)
    return b, run


def _same_tree(ta, tb, what):
    assert ta.root == tb.root, what
    for f in FIELDS:
        a, b = getattr(ta, f), getattr(tb, f)
        assert a.shape == b.shape and np.array_equal(a, b), "%s: %s differs" % (what, f)


def test_slabs_cut_on_the_device_equal_the_host_encoded_ones():
    """After one repartition the parts on the device -- trees, coalescent windows, derived quantities, totals -- are the
    ones the host path uploads for the same seed."""
    sc = make_scenario("C3", num_tips=3000, num_sites=29903, uncertain_tips=0.1)
    bh, rh = _run(sc, 5, 96, False)
    bd, rd = _run(sc, 5, 96, True)
    rh.repartition(); rd.repartition()
    assert rh.num_parts() == rd.num_parts()
    n, root_part = rh.num_parts()
    bh.recalc_derived(); bd.recalc_derived()
    for p in range(n):
        th = bh.part_download(p)
        _same_tree(th, bd.part_download(p), "part %d" % p)
        ch, cd = bh.part_coalescent(p), bd.part_coalescent(p)
        for k in ch:
            assert np.array_equal(np.asarray(ch[k]), np.asarray(cd[k]), equal_nan=True), (p, k)
        dh, dd = bh.part_derived(p, th.num_nodes), bd.part_derived(p, th.num_nodes)
        for a, b in zip(dh, dd):
            assert np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True), p
    assert bh.totals() == bd.totals()
    for r in (rh, rd): r.close()
    for b in (bh, bd): b.close()


@pytest.mark.parametrize("max_part_nodes", [0, 40])
def test_cycles_with_the_tree_in_hbm_equal_the_host_cycles(max_part_nodes):
    """Five repartition -> moves -> reassemble cycles (new stencil pick and new seeds every cycle, root sequence changes
    folded into the reference): the tree that comes back from HBM is the host cycle's tree, and so is the reference."""
    sc = make_scenario("C3", num_tips=3000, num_sites=29903, uncertain_tips=0.1)
    bh, rh = _run(sc, 7, 128, False, max_part_nodes)
    bd, rd = _run(sc, 7, 128, True, max_part_nodes)
    per_cycle = 128 * 400
    for cycle in range(5):
        rh.do_mcmc_steps(per_cycle, per_cycle); rd.do_mcmc_steps(per_cycle, per_cycle)
        th, refh = rh.tree(); td, refd = rd.tree()
        _same_tree(th, td, "cycle %d" % cycle)
        assert np.array_equal(refh, refd), cycle
    assert not np.array_equal(th.parent, sc.tree.parent)          # the moves did re-hang the tree
    for r in (rh, rd): r.close()
    for b in (bh, bd): b.close()


def test_tree_round_trip_and_root_sequence_changes():
    """upload -> download is the identity; and over enough cycles of a small tree the root sequence does change, which the
    device folds into its reference exactly as Run::normalize_root does on the host."""
    sc = make_scenario("C1", num_tips=200, num_sites=3000, uncertain_tips=0.2)
    bh, rh = _run(sc, 11, 4, False)
    bd, rd = _run(sc, 11, 4, True)
    changed = False
    for cycle in range(12):
        rh.do_mcmc_steps(4 * 3000, 4 * 3000); rd.do_mcmc_steps(4 * 3000, 4 * 3000)
        th, refh = rh.tree(); td, refd = rd.tree()
        _same_tree(th, td, "cycle %d" % cycle)
        assert np.array_equal(refh, refd), cycle
        changed = changed or not np.array_equal(refd, sc.ref)
    assert changed, "the root sequence never changed: the test does not exercise the re-referencing"
    # back to the host: the run carries on from the downloaded tree
    rd.set_device_tree(False)
    rh.do_mcmc_steps(4 * 1000, 4 * 1000); rd.do_mcmc_steps(4 * 1000, 4 * 1000)
    th, refh = rh.tree(); td, refd = rd.tree()
    _same_tree(th, td, "after leaving HBM")
    for r in (rh, rd): r.close()
    for b in (bh, bd): b.close()
