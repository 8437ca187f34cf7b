"""the whole tree resident in HBM (emat_tree_* / emat_run_set_device_tree).  The checker is the host cycle
(Run::repartition / Run::reassemble restated in emat_run.cpp, itself checked against the oracle in test_host_driver.py and
test_fullsize_gpu.py): same seeds => same partitions, same slabs, same moves, same trees, bit for bit -- with the
coalescent tables built by the host's code (EMAT_TREE_HOST_COALESCENT=1).  By default they are built by kernels, whose
exp / log / cos differ from glibc's in the last place: those tables are compared with the host's within 1e-12, and whole
runs on them are checked through the oracle's from-scratch recomputation."""
import os

import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import configure, rel_close, replay_device_parts_in_the_oracle
from oracle_ffi import OracleEngine

pytestmark = pytest.mark.gpu

FIELDS = ("parent", "child0", "child1", "t", "t_min", "t_max", "mut_offset", "mut_site", "mut_from", "mut_to", "mut_t",
          "miss_offset", "miss_start", "miss_end", "mfs_offset", "mfs_site", "mfs_state")


def _run(sc, seed, parts, device_tree, max_part_nodes=0, host_coalescent=True):
    old = os.environ.get("EMAT_TREE_HOST_COALESCENT")
    os.environ["EMAT_TREE_HOST_COALESCENT"] = "1" if host_coalescent else "0"    # read when the backend is created
    try:
        b = d.EmatBackend(sc.num_sites)
    finally:
        if old is None:
            del os.environ["EMAT_TREE_HOST_COALESCENT"]
        else:
            os.environ["EMAT_TREE_HOST_COALESCENT"] = old
    run = d.EmatRun(b, sc.tree, sc.ref, seed)
    run.set_num_parts(parts)
    run.set_max_part_nodes(max_part_nodes)      # 0: the reference's partition rule exactly
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


def test_coalescent_tables_built_on_the_device():
    """k_gt_coal_*: every part's window of the grid against CoalBuilder's (same cells, same windows, lineage counts and
    activity counts exact, Gaussian draws and population integrals to 1e-12), the parts' RNG streams left at the same
    position, and the trees untouched by it."""
    for name, kw, parts in (("C3", dict(num_tips=3000, num_sites=29903, uncertain_tips=0.1), 96), ("C2", dict(num_tips=1610, num_sites=18959), 40),
                            ("C1", dict(num_tips=150, num_sites=3000, uncertain_tips=0.3), 3)):
        sc = make_scenario(name, **kw)
        bh, rh = _run(sc, 5, parts, False)
        bd, rd = _run(sc, 5, parts, True, host_coalescent=False)
        rh.repartition(); rd.repartition()
        n, root_part = rh.num_parts()
        assert (n, root_part) == rd.num_parts()
        bh.recalc_derived(); bd.recalc_derived()          # (puts the host path's parts on their slabs too)
        for p in range(n):
            _same_tree(bh.part_download(p), bd.part_download(p), "%s part %d" % (name, p))
            ch, cd = bh.part_coalescent(p), bd.part_coalescent(p)
            assert ch["t_ref"] == cd["t_ref"] and ch["t_step"] == cd["t_step"], (name, p)
            assert np.array_equal(ch["k_bar_p"], cd["k_bar_p"]), (name, p)
            assert np.array_equal(ch["num_active_parts"], cd["num_active_parts"]), (name, p)
            for k in ("k_twiddle_bar_p", "k_twiddle_bar", "popsize_bar"):
                a, b = np.asarray(ch[k]), np.asarray(cd[k])
                assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)), (name, p, k)
                ok = ~np.isnan(a)
                assert np.all(np.abs(a[ok] - b[ok]) <= 1e-12 * np.maximum(1.0, np.abs(a[ok]))), (name, p, k, np.max(np.abs(a[ok] - b[ok])))
            assert bh.part_stats(p)["rng_draws"] == bd.part_stats(p)["rng_draws"], (name, p)
        gh, gd = bh.totals(), bd.totals()
        assert gh[0] == gd[0] and rel_close(gh[1], gd[1], 1e-11), (name, gh, gd)
        for r in (rh, rd): r.close()
        for b in (bh, bd): b.close()


def test_whole_cycles_on_the_device_built_tables():
    """The default path end to end: six cycles, then the tree that comes back from HBM is a valid EMAT by the oracle's
    integrity check, the incremental totals of a fresh pass equal the from-scratch ones, and the tips are where they were."""
    sc = make_scenario("C3", num_tips=3000, num_sites=29903, uncertain_tips=0.1)
    b, run = _run(sc, 9, 128, True, host_coalescent=False)
    run.do_mcmc_steps(6 * 128 * 500, 128 * 500)
    tree, ref = run.tree()
    assert tree.num_nodes == sc.tree.num_nodes and not np.array_equal(tree.parent, sc.tree.parent)
    tips = tree.child0 == -1
    assert np.array_equal(tips, sc.tree.child0 == -1)
    assert np.array_equal(tree.t_min, sc.tree.t_min) and np.array_equal(tree.t_max, sc.tree.t_max)
    assert np.all(tree.t[tips] >= tree.t_min[tips] - 1e-2) and np.all(tree.t[tips] <= tree.t_max[tips] + 1e-2)
    chk = OracleEngine(sc.num_sites)
    sc.tree, sc.ref = tree, ref
    configure(chk, sc, ref, [tree], [True], [1], 0)
    rc, msg = chk.part_check(0)
    assert rc == 0, msg
    # one more pass: incremental log G / prior against the recomputation
    run.repartition(); run.run_moves(128 * 300); b.synchronize()
    inc = b.totals(); b.recalc_derived(); rec = b.totals()
    assert rel_close(inc[0], rec[0], 1e-9) and rel_close(inc[1], rec[1], 1e-9), (inc, rec)
    run.reassemble()
    chk.close(); run.close(); b.close()


@pytest.mark.parametrize("name,kw,parts", [("C3", dict(num_tips=3000, num_sites=29903, uncertain_tips=0.1), 128),
                                           ("C2", dict(num_tips=1610, num_sites=18959), 40),
                                           ("C1", dict(num_tips=150, num_sites=3000, uncertain_tips=0.3), 3)])
def test_default_device_tree_path_move_for_move_against_the_oracle(name, kw, parts):
    """The DEFAULT path (parts cut by kernels, coalescent tables built by kernels -- no EMAT_TREE_HOST_COALESCENT) compared
    move for move with the oracle over four cycles: before every pass the oracle is given the parts, tables and RNG
    positions the device holds, then both run the pass: traces, counters, RNG consumption, trees, totals."""
    sc = make_scenario(name, **kw)
    old = os.environ.pop("EMAT_TREE_HOST_COALESCENT", None)
    try:
        b = d.EmatBackend(sc.num_sites, trace_moves=600)
    finally:
        if old is not None:
            os.environ["EMAT_TREE_HOST_COALESCENT"] = old
    run = d.EmatRun(b, sc.tree, sc.ref, 9)
    run.set_num_parts(parts); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
    run.set_device_tree(True)
    ref = sc.ref
    try:
        for cycle in range(4):
            run.repartition()
            n = replay_device_parts_in_the_oracle(sc, b, run, ref, parts * 600 + 7, 600)
            assert n >= 2
            run.reassemble()
            _, ref = run.tree()
    finally:
        run.close(); b.close()


@pytest.mark.parametrize("name,kw,parts", [("C3", dict(num_tips=3000, num_sites=29903, uncertain_tips=0.1), 128),
                                           ("C1", dict(num_tips=200, num_sites=3000, uncertain_tips=0.2), 4)])
def test_device_tree_cycles_against_the_restated_reference_run(name, kw, parts):
    """The kernels that cut the HBM-resident tree and gather it back (k_gt_partition / _measure / _build / _gather) against
    oracle/orc_run.hpp -- the reference's Run::repartition / reassemble / normalize_root restated step by step: every part
    the device cuts is the oracle's subtree bit for bit (nodes, frozen cut nodes, synthetic sub-root lists), and the tree
    and reference sequence the device gathers after its moves are what the oracle's reassemble makes of the same parts."""
    from oracle_ffi import OracleRun
    sc = make_scenario(name, **kw)
    b, run = _run(sc, 7, parts, True, host_coalescent=False)
    orun = OracleRun(sc.tree, sc.ref, 7 ^ 0xD1B54A32D192ED03, parts)
    try:
        for cycle in range(5 if name == "C3" else 12):
            run.repartition(); orun.repartition()
            n, root_part = run.num_parts()
            assert (n, root_part) == orun.num_parts(), cycle
            for p in range(n):
                _same_tree(b.part_download(p), orun.part(p)[0], "cycle %d part %d as cut" % (cycle, p))
            run.run_moves(n * 500); b.synchronize()
            for p in range(n):
                orun.part_put(p, b.part_download(p))
            run.reassemble(); orun.reassemble()
            orun.normalize_root()               # the device folds what the root carries into the reference while it gathers
            (td, refd), (to, refo) = run.tree(), orun.tree()
            _same_tree(td, to, "cycle %d gathered" % cycle)
            assert np.array_equal(refd, refo), cycle
            rc, msg = orun.check()
            assert rc == 0, msg
    finally:
        run.close(); b.close(); orun.close()


def test_global_move_statistics_from_parts_cut_on_the_device():
    """SURVEY 8(f).1 on top of 8(f).2: calc_Ttwiddle_l, calc_num_muts_l, calc_Ttwiddle_beta_a / calc_num_muts_ab computed from
    the parts while they are on the device, against the oracle's whole-tree values on the tree that comes back from HBM."""
    sc = make_scenario("C3", num_tips=900, num_sites=6000, uncertain_tips=0.2)
    b, run = _run(sc, 61, 40, True, host_coalescent=False)
    for cyc in range(3):
        run.repartition()
        n, _ = run.num_parts()
        run.run_moves(n * 800)
        got_T, got_n = run.Ttwiddle_l(), b.num_muts_l()
        Tg, Mg, ng = b.global_stats(1)
        run.reassemble()
        whole, ref = run.tree()
        orc = OracleEngine(sc.num_sites)
        sc2 = make_scenario("C3", num_tips=900, num_sites=6000, uncertain_tips=0.2); sc2.tree, sc2.ref = whole, ref
        configure(orc, sc2, ref, [whole], [True], [1], 0)
        want_T, want_n = orc.Ttwiddle_l(0), orc.num_muts_l()
        To, Mo, no = orc.global_stats(1)
        orc.close()
        assert rel_close(got_T, want_T, 1e-9), (cyc, float(np.max(np.abs(got_T - want_T))))
        assert np.array_equal(got_n, want_n), cyc
        assert ng == no and np.array_equal(Mg, Mo) and rel_close(Tg, To, 1e-9), cyc
    run.close(); b.close()


def test_pools_and_heaps_grow_when_they_run_out(monkeypatch):
    """EMAT_TREE_TIGHT starts the cut-state pools and the three list heaps without any room: the kernels report the
    overflow, the host doubles the buffers and runs them again, and the cycles still equal the host cycles bit for bit."""
    monkeypatch.setenv("EMAT_TREE_TIGHT", "1")
    sc = make_scenario("C3", num_tips=1500, num_sites=29903, uncertain_tips=0.1)
    bh, rh = _run(sc, 13, 64, False)
    bd, rd = _run(sc, 13, 64, True)
    for cycle in range(4):
        rh.do_mcmc_steps(64 * 800, 64 * 800); rd.do_mcmc_steps(64 * 800, 64 * 800)
        th, refh = rh.tree(); td, refd = rd.tree()
        _same_tree(th, td, "cycle %d" % cycle)
        assert np.array_equal(refh, refd), cycle
    pool_regrows, heap_regrows, _ = bd.tree_counters()
    assert pool_regrows >= 1 and heap_regrows >= 1, (pool_regrows, heap_regrows)
    for r in (rh, rd): r.close()
    for b in (bh, bd): b.close()


def test_the_whole_tree_as_one_part_and_a_two_tip_tree():
    """Edge cases of the cutting: a single part (the cut point is the root, the synthetic sub-root lists are the root's own
    missations) and the smallest tree there is."""
    sc = make_scenario("C1", num_tips=60, num_sites=2000, uncertain_tips=0.5)
    bh, rh = _run(sc, 3, 1, False)
    bd, rd = _run(sc, 3, 1, True)
    for cycle in range(3):
        rh.do_mcmc_steps(4000, 4000); rd.do_mcmc_steps(4000, 4000)
        th, refh = rh.tree(); td, refd = rd.tree()
        _same_tree(th, td, "one part, cycle %d" % cycle)
        assert np.array_equal(refh, refd)
    for r in (rh, rd): r.close()
    for b in (bh, bd): b.close()
    sc = make_scenario("C1", num_tips=2, num_sites=500)
    bh, rh = _run(sc, 3, 1, False)
    bd, rd = _run(sc, 3, 1, True)
    for r in (rh, rd): r.set_coalescent_t_step(0.5)      # the root of a two-tip tree wanders by hundreds of such cells per cycle: the grid's room is outgrown and regrown
    rh.do_mcmc_steps(500, 500); rd.do_mcmc_steps(500, 500)
    th, refh = rh.tree(); td, refd = rd.tree()
    _same_tree(th, td, "two tips")
    assert np.array_equal(refh, refd)
    for r in (rh, rd): r.close()
    for b in (bh, bd): b.close()


def test_whole_tree_grid_prior_from_parts_cut_on_the_device():
    """Row a20 on top of 8(f).2: Scalable_coalescent_prior's log-prior from the parts on the device, against the oracle's
    whole-tree value on the tree that comes back."""
    sc = make_scenario("C3", num_tips=900, num_sites=6000, uncertain_tips=0.2)
    b, run = _run(sc, 67, 40, True, host_coalescent=False)
    t_step = sc.default_t_step()
    for cyc in range(2):
        run.repartition()
        n, _ = run.num_parts()
        run.run_moves(n * 600)
        got = b.scalable_coalescent_log_prior(sc.t_max_tip, t_step)
        run.reassemble()
        whole, ref = run.tree()
        orc = OracleEngine(sc.num_sites)
        sc2 = make_scenario("C3", num_tips=900, num_sites=6000, uncertain_tips=0.2); sc2.tree, sc2.ref = whole, ref
        configure(orc, sc2, ref, [whole], [True], [1], 0)
        want = orc.scalable_log_prior(0, sc.t_max_tip, t_step)
        orc.close()
        assert rel_close(got, want, 1e-9), (cyc, got, want)
    run.close(); b.close()


def test_out_of_space_recoveries_inside_cycles(monkeypatch):
    """List heaps sized with almost no slack (EMAT_SLACK / EMAT_HEAP_PER_NODE): in every pass a good share of the parts stops
    before a move for lack of room, is given more and runs the rest of its moves -- across tickets, on slabs that were cut
    on the device -- and the cycles still equal the host cycles bit for bit (a chain does not depend on where it paused)."""
    monkeypatch.setenv("EMAT_SLACK", "1.05")
    monkeypatch.setenv("EMAT_HEAP_PER_NODE", "8")
    sc = make_scenario("C3", num_tips=3000, num_sites=29903, uncertain_tips=0.1)
    bh, rh = _run(sc, 19, 128, False)
    bd, rd = _run(sc, 19, 128, True)
    for cycle in range(4):
        rh.do_mcmc_steps(128 * 1500, 128 * 1500); rd.do_mcmc_steps(128 * 1500, 128 * 1500)
        th, refh = rh.tree(); td, refd = rd.tree()
        _same_tree(th, td, "cycle %d" % cycle)
        assert np.array_equal(refh, refd), cycle
    for r in (rh, rd): r.close()
    for b in (bh, bd): b.close()


def _whole_tree_as_one_part(parent, c0, c1, root):
    """partition_tree's numbering for a single part: local 0 = root, children get consecutive indices when their parent is
    expanded, right child expanded first."""
    orig, kid0, kid1 = [root], [-1], [-1]
    work = [(root, 0)]
    while work:
        src, dst = work.pop()
        if c0[src] < 0:
            continue
        dl = len(orig); orig.append(int(c0[src])); dr = len(orig); orig.append(int(c1[src]))
        kid0 += [-1, -1]; kid1 += [-1, -1]
        kid0[dst], kid1[dst] = dl, dr
        work.append((int(c0[src]), dl)); work.append((int(c1[src]), dr))
    return np.array([0, len(orig)], np.int32), np.array(orig, np.int32), np.array(kid0, np.int32), np.array(kid1, np.int32)


def test_the_c_abi_of_the_resident_tree_directly():
    """emat_tree_upload / _get_topology / _repartition / _reassemble / _download without the run driver: the partition is
    made here, in Python; the part it yields is the one emat_part_upload makes of the same subtree; after the moves the tree
    that comes back equals that part, relabelled; malformed partitions are refused."""
    sc = make_scenario("C1", num_tips=120, num_sites=3000, uncertain_tips=0.3)
    old = os.environ.get("EMAT_TREE_HOST_COALESCENT"); os.environ["EMAT_TREE_HOST_COALESCENT"] = "1"
    try:
        bd = d.EmatBackend(sc.num_sites)
    finally:
        if old is None: del os.environ["EMAT_TREE_HOST_COALESCENT"]
        else: os.environ["EMAT_TREE_HOST_COALESCENT"] = old
    bh = d.EmatBackend(sc.num_sites)
    t_step = sc.default_t_step()
    for b in (bd, bh):
        b.set_ref_sequence(sc.ref); b.set_hky(sc.mu, sc.kappa, sc.pi); b.set_flags(sc.t_max_tip)
    bd.tree_upload(sc.tree)
    t0, r0 = bd.tree_download()
    _same_tree(t0, sc.tree, "upload -> download")
    assert np.array_equal(r0, sc.ref)
    parent, c0, c1, t, root = bd.tree_topology()
    assert root == sc.tree.root and np.array_equal(parent, sc.tree.parent) and np.array_equal(t, sc.tree.t)
    po, orig, k0, k1 = _whole_tree_as_one_part(parent, c0, c1, root)
    # malformed partitions
    with pytest.raises(d.EmatError, match="EMAT_ERR_INVALID_ARGUMENT"):
        bad = orig.copy(); bad[5] = bad[6]
        bd.tree_repartition(po, bad, k0, k1, 0, [7], sc.pop, t_step)
    with pytest.raises(d.EmatError, match="EMAT_ERR_INVALID_ARGUMENT"):
        bd.tree_repartition(np.array([0, len(orig) - 2], np.int32), orig[:-2], k0[:-2], k1[:-2], 0, [7], sc.pop, t_step)
    bd.tree_repartition(po, orig, k0, k1, 0, [7], sc.pop, t_step)
    # the same part through the host path: the subtree in the partition's numbering
    inv = np.empty(len(orig), np.int64); inv[orig] = np.arange(len(orig))
    sub = d.FlatTree.empty(len(orig), int(sc.tree.mut_offset[-1]), int(sc.tree.miss_offset[-1]), int(sc.tree.mfs_offset[-1]))
    sub.root = 0
    km = ki = kf = 0
    for s, o in enumerate(orig):
        sub.parent[s] = -1 if o == root else inv[sc.tree.parent[o]]
        sub.child0[s], sub.child1[s] = k0[s], k1[s]
        sub.t[s], sub.t_min[s], sub.t_max[s] = sc.tree.t[o], sc.tree.t_min[o], sc.tree.t_max[o]
        for a, b_, dst in (("mut_offset", ("mut_site", "mut_from", "mut_to", "mut_t"), "m"), ("miss_offset", ("miss_start", "miss_end"), "i"), ("mfs_offset", ("mfs_site", "mfs_state"), "f")):
            lo, hi = getattr(sc.tree, a)[o], getattr(sc.tree, a)[o + 1]
            k = {"m": km, "i": ki, "f": kf}[dst]
            for f in b_:
                getattr(sub, f)[k:k + hi - lo] = getattr(sc.tree, f)[lo:hi]
            if dst == "m": km += hi - lo
            elif dst == "i": ki += hi - lo
            else: kf += hi - lo
        sub.mut_offset[s + 1], sub.miss_offset[s + 1], sub.mfs_offset[s + 1] = km, ki, kf
    sub = sub.trimmed()
    bh.upload_parts([sub], [True], [7]); bh.build_coalescent_parts(sc.pop, 0, t_step)
    bh.recalc_derived(); bd.recalc_derived()
    _same_tree(bh.part_download(0), bd.part_download(0), "the part")
    assert bh.totals() == bd.totals()
    for b in (bh, bd):
        b.run_moves_per_part(4000); b.synchronize()
    ph = bh.part_download(0)
    _same_tree(ph, bd.part_download(0), "after the moves")
    site, frm, to = bd.tree_reassemble()
    whole, ref = bd.tree_download()
    # the part, relabelled back to the tree's node numbers, with the root's deltas folded into the reference
    assert whole.root == orig[ph.root]
    for s, o in enumerate(orig):
        assert whole.t[o] == ph.t[s]
        if ph.child0[s] >= 0:
            assert whole.child0[o] == orig[ph.child0[s]] and whole.child1[o] == orig[ph.child1[s]]
        lo, hi = ph.mut_offset[s], ph.mut_offset[s + 1]
        if s == ph.root:
            assert whole.mut_offset[o + 1] == whole.mut_offset[o]
            assert sorted(zip(ph.mut_site[lo:hi], ph.mut_from[lo:hi], ph.mut_to[lo:hi])) == sorted(zip(site, frm, to))
        else:
            wl, wh = whole.mut_offset[o], whole.mut_offset[o + 1]
            assert np.array_equal(whole.mut_site[wl:wh], ph.mut_site[lo:hi]) and np.array_equal(whole.mut_t[wl:wh], ph.mut_t[lo:hi])
        il, ih = ph.miss_offset[s], ph.miss_offset[s + 1]; wl, wh = whole.miss_offset[o], whole.miss_offset[o + 1]
        assert np.array_equal(whole.miss_start[wl:wh], ph.miss_start[il:ih]) and np.array_equal(whole.miss_end[wl:wh], ph.miss_end[il:ih])
    want_ref = sc.ref.copy(); want_ref[site] = to
    assert np.array_equal(ref, want_ref) and np.array_equal(frm, sc.ref[site])
    with pytest.raises(d.EmatError, match="EMAT_ERR_STATE"):
        bd.tree_reassemble()                  # nothing is out on slabs any more
    bd.close(); bh.close()


def test_cut_point_states_larger_than_the_small_kernel_holds():
    """A fast-evolving genome: the path from the root to a cut point carries hundreds of mutations, more net changes than the
    LDS of the first k_gt_measure pass holds; those parts go through its large variant, and the cycles still equal the host's."""
    from delphy_amd.engine import SynthParams, make_synthetic_emat
    from delphy_amd.scenarios import Scenario, PI, KAPPA
    p = SynthParams(num_tips=400, num_sites=12000, tip_span=365.0, pop_n0=365.0, pop_growth=0.0, mu=5e-3 / 365.0, gaps_per_tip=2, mean_gap_len=60.0, seed=20261099)
    p.pi = PI; p.kappa = KAPPA
    tree, ref, tmax = make_synthetic_emat(p)
    sc = Scenario("hot", tree, ref, tmax, p.mu, p.kappa, PI, d.PopModel.exp(tmax, 365.0, 0.0, 0.0), p.num_sites)
    bh, rh = _run(sc, 29, 12, False)
    bd, rd = _run(sc, 29, 12, True)
    for cycle in range(2):
        rh.do_mcmc_steps(12 * 600, 12 * 600); rd.do_mcmc_steps(12 * 600, 12 * 600)
        th, refh = rh.tree(); td, refd = rd.tree()
        _same_tree(th, td, "cycle %d" % cycle)
        assert np.array_equal(refh, refd), cycle
    assert bd.tree_counters()[2] >= 1, "no cut-point state outgrew the small kernel: the test does not reach the large one (%d mutations on the tree)" % tree.mut_offset[-1]
    for r in (rh, rd): r.close()
    for b in (bh, bd): b.close()


def _partition_in_python(parent, c0, c1, root, cuts):
    """tree_partitioning.h:88-135 / 196-239 written out once more, independently of both the host driver and the kernel."""
    is_cut = np.zeros(len(parent), bool); is_cut[list(cuts)] = True
    cut_of_part = list(cuts)
    if root in cuts: root_part = cuts.index(root)
    else: is_cut[root] = True; root_part = len(cuts); cut_of_part.append(root)
    off, orig, kid0, kid1 = [0], [], [], []
    for cut in cut_of_part:
        po, k0, k1 = [cut], [-1], [-1]
        work = [(cut, 0)]
        while work:
            src, dst = work.pop()
            if c0[src] < 0 or (is_cut[src] and src != cut):
                continue
            dl = len(po); po.append(int(c0[src])); dr = len(po); po.append(int(c1[src]))
            k0 += [-1, -1]; k1 += [-1, -1]
            k0[dst], k1[dst] = dl, dr
            work.append((int(c0[src]), dl)); work.append((int(c1[src]), dr))
        orig += po; kid0 += k0; kid1 += k1; off.append(len(orig))
    return len(cut_of_part), root_part, np.array(off, np.int32), np.array(orig, np.int32), np.array(kid0, np.int32), np.array(kid1, np.int32)


def test_partition_tree_on_the_device():
    """emat_tree_partition against partition_tree written out in Python: no cut at all (one part), random inner nodes as cuts
    (nested ones included), a stencil that names the root, and cut lists that are refused."""
    sc = make_scenario("C3", num_tips=2500, num_sites=2000)
    b = d.EmatBackend(sc.num_sites)
    b.set_ref_sequence(sc.ref); b.set_hky(sc.mu, sc.kappa, sc.pi); b.set_flags(sc.t_max_tip)
    b.tree_upload(sc.tree)
    parent, c0, c1, t, root = b.tree_topology()
    inner = np.flatnonzero(c0 >= 0)
    rng = np.random.default_rng(2)
    some = [int(x) for x in rng.choice(inner[inner != root], size=25, replace=False)]
    for cuts in ([], some, some[:7] + [root] + some[7:12]):
        got = b.tree_partition(cuts)
        want = _partition_in_python(parent, c0, c1, root, cuts)
        assert got[0] == want[0] and got[1] == want[1]
        for g, w in zip(got[2:], want[2:]):
            assert np.array_equal(g, w)
    with pytest.raises(d.EmatError, match="EMAT_ERR_INVALID_ARGUMENT"):
        b.tree_partition([some[0], some[0]])
    with pytest.raises(d.EmatError, match="EMAT_ERR_INVALID_ARGUMENT"):
        b.tree_partition([int(np.flatnonzero(c0 < 0)[0])])        # a tip is not a cut point
    b.close()


def test_randomised_cycles_with_the_tree_in_hbm_equal_the_host_cycles():
    """The cycle test above over a seeded sweep of what it holds constant: tree size (down to a dozen tips), genome length,
    mutation and gap density, tip-date uncertainty, population model, number of parts (one part up to as many as the tree
    yields), moves per cycle.  Same seeds => the tree and the reference that come back from HBM are the host cycle's."""
    import delphy_amd.engine as e
    from delphy_amd.scenarios import Scenario, KAPPA, PI
    rng = np.random.default_rng(int(os.environ.get("EMAT_FUZZ_SEED", "20261003")))   # EMAT_FUZZ_SEED / EMAT_FUZZ_CASES: longer hunts by hand
    for case in range(int(os.environ.get("EMAT_FUZZ_CASES", "14"))):
        tips = int(rng.integers(12, 900))
        sites = int(rng.choice([80, 500, 3000, 29903]))
        span = float(rng.choice([60.0, 365.0, 900.0]))
        mu = float(10 ** rng.uniform(-3.4, -2.3)) / 365.0 * (30000.0 / max(sites, 300)) ** 0.5
        par = e.SynthParams(num_tips=tips, num_sites=sites, tip_span=span, pop_n0=float(10 ** rng.uniform(1.5, 3.0)), pop_growth=float(rng.choice([0.0, 2.0])) / 365.0,
                            mu=mu, gaps_per_tip=int(rng.integers(0, 4)), mean_gap_len=float(max(2.0, sites * 10 ** rng.uniform(-2.5, -1.0))), seed=int(rng.integers(1, 2**31)))
        par.pi, par.kappa = PI, KAPPA
        if rng.random() < 0.5:
            par.frac_uncertain_tips, par.tip_date_uncertainty = float(rng.uniform(0.05, 0.5)), float(rng.uniform(0.5, 10.0))
        tree, ref, tmax = e.make_synthetic_emat(par)
        while tree.mut_site.shape[0] > 30 * tips:
            mu /= 4.0; par.mu = mu
            tree, ref, tmax = e.make_synthetic_emat(par)
        kind = case % 3
        if kind == 0:
            pop = d.PopModel.exp(tmax, par.pop_n0, 0.0, 0.0)
        elif kind == 1:
            pop = d.PopModel.exp(tmax, par.pop_n0, float(rng.uniform(0.2, 3.0)) / 365.0, 1.0)
        else:
            x = np.unique(np.append(np.sort(tmax - span * 1.3 * rng.uniform(0.0, 1.0, int(rng.integers(2, 30)))), tmax))
            pop = d.PopModel.skygrid(x, np.log(par.pop_n0) + rng.normal(0.0, 0.4, x.shape[0]), log_linear=bool(case % 2))
        sc = Scenario("R%d" % case, tree, ref, tmax, mu, KAPPA, PI, pop, sites)
        parts = int(rng.choice([1, 2, 7, 32, 4096]))
        seed = int(rng.integers(1, 10**6))
        per_cycle = int(rng.choice([500, 5000, 20 * tips]))
        what = "case %d (tips %d, sites %d, parts %d, pop kind %d, seed %d, %d moves per cycle)" % (case, tips, sites, parts, kind, seed, per_cycle)
        bh, rh = _run(sc, seed, parts, False)
        bd, rd = _run(sc, seed, parts, True)
        try:
            for cycle in range(4):
                rh.do_mcmc_steps(per_cycle, per_cycle); rd.do_mcmc_steps(per_cycle, per_cycle)
                th, refh = rh.tree(); td, refd = rd.tree()
                _same_tree(th, td, "%s cycle %d" % (what, cycle))
                assert np.array_equal(refh, refd), (what, cycle)
        except d.EmatError as ex:
            raise AssertionError("%s: %s" % (what, ex)) from ex
        finally:
            for r in (rh, rd): r.close()
            for b in (bh, bd): b.close()


@pytest.mark.parametrize("num_changes", [100, 700])
def test_gather_rereferences_missing_data_for_any_number_of_root_changes(num_changes):
    """emat_tree_gather_local folds the changes of the root sequence into the reference and re-references every node's missing
    data (Run::normalize_root -> Missation_map::ref_seq_changed): a missing site whose recorded state now equals the
    reference loses its record, one that had none (it followed the old reference) gains one with the old state.  The gather
    caches up to 256 changes in LDS and reads longer lists from HBM: both paths against a restatement in numpy, with changes
    injected through the C-ABI on a tree whose tips miss a fifth of the genome."""
    import delphy_amd.engine as e
    from delphy_amd.scenarios import Scenario, KAPPA, PI
    par = e.SynthParams(num_tips=150, num_sites=4000, tip_span=200.0, pop_n0=300.0, pop_growth=0.0, mu=2e-3 / 365.0, gaps_per_tip=4, mean_gap_len=200.0, seed=12)
    par.pi, par.kappa = PI, KAPPA
    tree, ref, tmax = e.make_synthetic_emat(par)
    sc = Scenario("gaps", tree, ref, tmax, par.mu, KAPPA, PI, d.PopModel.exp(tmax, 300.0, 0.0, 0.0), 4000)
    b = d.EmatBackend(sc.num_sites)
    b.set_ref_sequence(sc.ref); b.set_hky(sc.mu, sc.kappa, sc.pi); b.set_flags(sc.t_max_tip)
    b.tree_upload(sc.tree)
    parent, c0, c1, t, root = b.tree_topology()
    po, orig, k0, k1 = _whole_tree_as_one_part(parent, c0, c1, root)
    b.tree_repartition(po, orig, k0, k1, 0, [7], sc.pop, sc.default_t_step())
    rng = np.random.default_rng(num_changes)
    mutated = set(int(x) for x in sc.tree.mut_site)                     # keep clear of sites that carry mutations: the injected change is not one the moves made
    site = np.array(sorted(rng.choice([l for l in range(sc.num_sites) if l not in mutated], num_changes, replace=False)), np.int32)
    frm = sc.ref[site].astype(np.uint8)
    to = ((frm + rng.integers(1, 4, num_changes)) % 4).astype(np.uint8)
    assert b.tree_root_deltas()[0].shape[0] == 0                       # no moves were run
    b.tree_gather_local(site, frm, to)
    b.tree_reassemble_end()
    got, got_ref = b.tree_download()
    want_ref = sc.ref.copy(); want_ref[site] = to
    assert np.array_equal(got_ref, want_ref)
    for f in ("parent", "child0", "child1", "t", "mut_offset", "mut_site", "mut_from", "mut_to", "mut_t", "miss_offset", "miss_start", "miss_end"):
        assert np.array_equal(getattr(got, f), getattr(sc.tree, f)), f
    touched = 0
    change = {int(l): (int(a), int(c)) for l, a, c in zip(site, frm, to)}
    for n in range(sc.tree.num_nodes):
        iv = [(int(sc.tree.miss_start[k]), int(sc.tree.miss_end[k])) for k in range(sc.tree.miss_offset[n], sc.tree.miss_offset[n + 1])]
        fs = {int(sc.tree.mfs_site[k]): int(sc.tree.mfs_state[k]) for k in range(sc.tree.mfs_offset[n], sc.tree.mfs_offset[n + 1])}
        for l, (a, c) in change.items():
            if any(lo <= l < hi for lo, hi in iv):
                touched += 1
                if l in fs:
                    if fs[l] == c: del fs[l]
                else:
                    fs[l] = a
        lo, hi = got.mfs_offset[n], got.mfs_offset[n + 1]
        assert list(zip(got.mfs_site[lo:hi].tolist(), got.mfs_state[lo:hi].tolist())) == sorted(fs.items()), n
    assert touched > num_changes // 4
    b.close()




def test_children_mirror_is_current_when_reassemble_returns():
    """emat_tree_get_kids: the children of every node, the root and its time as the backend mirrors them -- after the upload, and
    after whole cycles of the run driver, where emat_tree_reassemble hands the links over before the gather of the lists has
    finished (k_gt_gather_links; gt_finish_gather): they must equal what emat_tree_get_topology and emat_tree_download report
    once the gather is through, and asking for them must not disturb the tree that is still being written."""
    sc = make_scenario("C3", num_tips=3000, num_sites=29903, uncertain_tips=0.1)
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, 23)
    run.set_num_parts(128); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop); run.set_device_tree(True)
    run.repartition(); run.reassemble()
    kids, root, t_root = b.tree_kids()
    assert np.array_equal(kids[:, 0], sc.tree.child0) and np.array_equal(kids[:, 1], sc.tree.child1) and root == sc.tree.root and t_root == sc.tree.t[root]
    changed = False
    for cycle in range(4):
        run.repartition(); run.run_moves(128 * 500 + 3); run.reassemble()
        kids, root, t_root = b.tree_kids()                     # (the lists may still be on their way)
        parent, c0, c1, t, root2 = b.tree_topology()           # (waits for them)
        assert root == root2 and t_root == t[root] and np.array_equal(kids[:, 0], c0) and np.array_equal(kids[:, 1], c1), cycle
        inner = np.flatnonzero(c0 >= 0)
        assert np.array_equal(parent[c0[inner]], inner) and np.array_equal(parent[c1[inner]], inner) and parent[root] == -1
        tree, _ = b.tree_download()
        assert np.array_equal(tree.child0, c0) and np.array_equal(tree.child1, c1) and np.array_equal(tree.t, t)
        changed = changed or not np.array_equal(c0, sc.tree.child0)
    assert changed                                                 # the cycles did re-hang subtrees
    run.close(); b.close()


def test_a_deferred_gather_that_fails_stays_failed_until_a_tree_is_uploaded():
    """emat_tree_reassemble returns once links, root and root changes are on the host; the gather of the lists is checked by whoever
    touches the tree next (gt_finish_gather).  If THAT fails, the tree on the device has new links and half-written lists: the failure
    must not be reported once and forgotten -- every emat_tree_* call fails with its text until emat_tree_upload replaces the tree
    (ADVICE round 5).  `debug_fail_gather` makes the next gather report an inconsistency after emat_tree_reassemble has returned."""
    sc = make_scenario("C3", num_tips=1500, num_sites=29903, uncertain_tips=0.1)
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, 29)
    run.set_num_parts(64); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop); run.set_device_tree(True)
    run.repartition(); run.run_moves(64 * 200); run.reassemble()
    good, good_ref = b.tree_download()                              # (a healthy cycle first)
    run.repartition(); run.run_moves(64 * 200)
    b.set_option("debug_fail_gather", 1)
    try:
        run.reassemble()                                            # returns: the links are there, the lists are on their way ...
    except d.EmatError as e:                                        # ... unless the root sequence changed in this cycle: then the gather is not postponed
        assert "inconsistent" in str(e)
    for attempt in range(3):                                        # ... and the gather's verdict is sticky
        with pytest.raises(d.EmatError, match="inconsistent|incomplete"):
            b.tree_topology() if attempt != 1 else b.tree_download()
    with pytest.raises(d.EmatError, match="incomplete"):
        b.tree_kids()
    with pytest.raises(d.EmatError, match="incomplete"):
        run.repartition()
    b.tree_upload(good)                                             # the way out
    parent, c0, c1, t, root = b.tree_topology()
    assert root == good.root and np.array_equal(c0, good.child0) and np.array_equal(t, good.t)
    # and an upload forgets the partition of the tree that was there: cutting without arrays must ask for a new one
    import ctypes as C
    n_parts, root_part = run.num_parts()
    seeds = np.ones(n_parts, np.uint64); m = sc.pop.c_struct()
    st = b._lib.emat_tree_repartition(b._h, n_parts, None, None, None, None, root_part, seeds.ctypes.data_as(C.POINTER(C.c_uint64)), C.byref(m), sc.default_t_step())
    assert st != 0 and b"emat_tree_partition first" in b._lib.emat_last_error(b._h)
    run.close(); b.close()
