"""SURVEY 8(f).4, initial-tree construction: emat_tree_build_usher_like (graft loop on the device) against the oracle's restatement
of the reference's build_usher_like_tree (oracle/orc_build.hpp, pinned to the reference's fix_up_missations cases): same
descriptors + same seed => the same tree, bit for bit -- topology, node times, every mutation with its time, every missation and
from-state; and the reference's own closing checks of the builder (tree integrity, every tip reproduces its descriptor)."""
import dataclasses

import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from oracle_ffi import OracleBuild

FIELDS = ("parent", "child0", "child1", "t", "t_min", "t_max", "mut_offset", "mut_site", "mut_from", "mut_to", "mut_t",
          "miss_offset", "miss_start", "miss_end", "mfs_offset", "mfs_site", "mfs_state")


def build_both(sc, seed):
    ob = OracleBuild(sc.ref)
    tips = ob.tip_descs_of(sc.tree)
    b = d.EmatBackend(sc.num_sites)
    try:
        b.set_ref_sequence(sc.ref)
        got = b.build_usher_like(tips, seed)
        want = ob.build_usher_like(tips, seed)
        assert got.root == want.root
        for f in FIELDS:
            x, y = getattr(got, f), getattr(want, f)
            assert x.shape == y.shape and np.array_equal(x, y), "%s differs (seed %d)" % (f, seed)
        rc, msg = ob.check(got, tips)
        assert rc == 0, msg
        return got, tips
    finally:
        b.close(); ob.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,kw", [("C1", dict(num_tips=12, num_sites=300)), ("C1", dict(num_tips=100, num_sites=30000, uncertain_tips=0.3)),
                                     ("C2", dict(num_tips=1610, num_sites=18959)), ("C3", dict(num_tips=3000, num_sites=29903, uncertain_tips=0.1))])
def test_device_builder_equals_the_restated_reference(name, kw):
    sc = make_scenario(name, **kw)
    for seed in (1, 20261001):
        tree, tips = build_both(sc, seed)
        assert tree.num_nodes == 2 * tips.num_tips - 1 and np.all(tree.child0[: tips.num_tips] == -1)


@pytest.mark.gpu
def test_c3_built_on_the_device_and_run():
    """Config C3 as a whole (10 000 tips, 29 903 sites) from its tip descriptors, equal to the oracle's tree bit for bit; the
    tree then goes through a cycle of local moves like any other."""
    sc = make_scenario("C3")
    tree, tips = build_both(sc, 7)
    assert tips.num_tips == 10000
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, tree, sc.ref, 3)
    run.set_num_parts(512); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
    run.set_device_tree(True); run.set_paranoid(True)
    run.do_mcmc_steps(512 * 200, 512 * 200)
    run.close(); b.close()


@pytest.mark.gpu
def test_descriptors_the_reference_rejects():
    sc = make_scenario("C1", num_tips=12, num_sites=300)
    ob = OracleBuild(sc.ref); tips = ob.tip_descs_of(sc.tree); ob.close()
    b = d.EmatBackend(sc.num_sites); b.set_ref_sequence(sc.ref)
    def broken(**kw):
        t = d.TipDescs(tips.t_min.copy(), tips.t_max.copy(), tips.delta_offset.copy(), tips.delta_site.copy(), tips.delta_to.copy(),
                       tips.miss_offset.copy(), tips.miss_start.copy(), tips.miss_end.copy())
        for k, f in kw.items():
            f(getattr(t, k))
        return t
    k = int(np.flatnonzero(np.diff(tips.delta_offset) > 0)[0]); j = int(tips.delta_offset[k])
    def to_ref(a): a[j] = sc.ref[tips.delta_site[j]]
    def out_of_range(a): a[j] = sc.num_sites
    def swap_dates(a): a[0] = tips.t_max[0] + 10.0
    for bad, what in ((broken(delta_to=to_ref), "equal 'from' and 'to'"), (broken(delta_site=out_of_range), "outside"), (broken(t_min=swap_dates), "t_min > t_max")):
        with pytest.raises(d.EmatError, match=what):
            b.build_usher_like(bad, 1)
    b.close()


def _default_built(name, seed, **kw):
    """A scenario whose tree is the reference's default initial tree for the scenario's tips (written against the root's sequence)."""
    sc = make_scenario(name, **kw)
    ob = OracleBuild(sc.ref)
    try:
        tips = ob.tip_descs_of(sc.tree)
        tree, ref, rep = ob.build_default(tips, seed)
        rc, msg = ob.check(tree, tips, ref=ref)
        assert rc == 0, msg
        return dataclasses.replace(sc, tree=tree, ref=ref), tips, tree, rep
    finally:
        ob.close()


def _default_built_by_the_product(sc, seed):
    """emat_tree_build_default (host C++ in the product) next to the oracle's restatement: same descriptors + same seed => the same tree,
    the same root sequence, the same parsimony report.  Returns the product's scenario."""
    ob = OracleBuild(sc.ref)
    b = d.EmatBackend(sc.num_sites, device=-1)          # host code: no device needed
    try:
        tips = ob.tip_descs_of(sc.tree)
        b.set_ref_sequence(sc.ref)
        got, gref, grep_ = b.build_default(tips, seed)
        want, wref, wrep = ob.build_default(tips, seed)
        assert got.root == want.root and np.array_equal(gref, wref) and grep_ == wrep, (grep_, wrep)
        for f in FIELDS:
            x, y = getattr(got, f), getattr(want, f)
            assert x.shape == y.shape and np.array_equal(x, y), "%s differs (seed %d)" % (f, seed)
        rc, msg = ob.check(got, tips, ref=gref)
        assert rc == 0, msg
        return dataclasses.replace(sc, tree=got, ref=gref), tips, grep_
    finally:
        b.close(); ob.close()


@pytest.mark.parametrize("name,kw,seed", [("C1", dict(num_tips=12, num_sites=300), 3), ("C1", dict(num_tips=2, num_sites=300), 1), ("C1", dict(num_tips=3, num_sites=300), 2),
                                          ("C1", dict(num_tips=150, num_sites=30000, uncertain_tips=0.3), 4), ("C2", dict(num_tips=1610, num_sites=18959), 5),
                                          ("C3", dict(num_tips=3000, num_sites=29903, uncertain_tips=0.1), 6), ("C4", dict(num_tips=4000), 8)])
def test_product_default_builder_equals_the_restated_reference(name, kw, seed):
    """SURVEY 8(f).4, the reference's default initial tree (utree.cpp) in the product: bit for bit the oracle's tree -- topology, every
    time, every mutation, missation and from-state, the root sequence -- and the reference's closing checks of a built tree."""
    sc2, tips, rep = _default_built_by_the_product(make_scenario(name, **kw), seed)
    assert sc2.tree.num_nodes == 2 * tips.num_tips - 1 and np.all(sc2.tree.child0[: tips.num_tips] == -1)
    assert rep["spr_deltas"] <= rep["refined_deltas"] <= rep["guide_deltas"]


def test_product_default_builder_on_randomised_descriptors():
    """Seeded random scenarios (tree size, genome length, mutation and gap density, tip-date uncertainty all vary)."""
    import os
    from delphy_amd.scenarios import random_scenario
    rng = np.random.default_rng(int(os.environ.get("EMAT_FUZZ_SEED", "20261010")))
    for case in range(int(os.environ.get("EMAT_FUZZ_CASES", "12"))):
        sc, _, _, what = random_scenario(rng, case, max_tips=600)
        try:
            _default_built_by_the_product(sc, 1000 + case)
        except AssertionError as e:
            raise AssertionError("%s: %s" % (what, e))


def test_product_default_builder_rejects_what_the_reference_rejects():
    sc = make_scenario("C1", num_tips=12, num_sites=300)
    ob = OracleBuild(sc.ref); tips = ob.tip_descs_of(sc.tree); ob.close()
    b = d.EmatBackend(sc.num_sites, device=-1)
    with pytest.raises(d.EmatError, match="emat_set_ref_sequence"):
        b.build_default(tips, 1)
    b.set_ref_sequence(sc.ref)
    k = int(np.flatnonzero(np.diff(tips.delta_offset) > 0)[0]); j = int(tips.delta_offset[k])
    bad = d.TipDescs(tips.t_min.copy(), tips.t_max.copy(), tips.delta_offset.copy(), tips.delta_site.copy(), tips.delta_to.copy(), tips.miss_offset.copy(), tips.miss_start.copy(), tips.miss_end.copy())
    bad.delta_to[j] = sc.ref[tips.delta_site[j]]
    with pytest.raises(d.EmatError, match="equal 'from' and 'to'"):
        b.build_default(bad, 1)
    b.close()


def test_the_references_default_builder_restated():
    """oracle/orc_utree.hpp (guide tree -> refinement rounds -> SPR refinement -> OLS rooting -> phylo tree; pinned to the reference's
    tests/utree_tests.cpp by oracle/orc_tests) on the tip descriptors of scenario trees: the reference's closing checks of a built
    tree pass, every stage is at least as parsimonious as the one before, the same stream gives the same tree."""
    for name, kw, seed in (("C1", dict(num_tips=12, num_sites=300), 3), ("C1", dict(num_tips=150, num_sites=30000, uncertain_tips=0.3), 4), ("C2", dict(num_tips=400, num_sites=18959), 5)):
        sc, tips, tree, rep = _default_built(name, seed, **kw)
        assert tree.num_nodes == 2 * tips.num_tips - 1 and np.all(tree.child0[: tips.num_tips] == -1)
        assert rep["spr_deltas"] <= rep["refined_deltas"] <= rep["guide_deltas"], rep
        # mutations above the root re-reference the root sequence away; all others are the unrooted tree's deltas
        assert tree.mut_site.shape[0] == rep["spr_deltas"], (tree.mut_site.shape[0], rep)
        _, _, again, _ = _default_built(name, seed, **kw)
        for f in FIELDS:
            assert np.array_equal(getattr(tree, f), getattr(again, f)), f


@pytest.mark.gpu
def test_chains_started_from_the_default_builders_tree():
    """The reference starts a run from build_initial_phylo_tree's tree (cmdline.cpp:437) -- here emat_tree_build_default's, equal to the
    oracle's; the local-move path on that starting state:
    derived quantities and the chains of every part, device against oracle, like test_parity_gpu's cases on simulated trees."""
    from helpers import run_parity
    sc, tips, rep = _default_built_by_the_product(make_scenario("C2", num_tips=700, num_sites=18959), 9)
    run_parity(sc, num_parts=12, moves_per_part=3000, seed=5, trace=300)


def test_builder_has_no_host_fallback():
    sc = make_scenario("C1", num_tips=12, num_sites=300)
    ob = OracleBuild(sc.ref); tips = ob.tip_descs_of(sc.tree); ob.close()
    b = d.EmatBackend(sc.num_sites, device=-1); b.set_ref_sequence(sc.ref)
    with pytest.raises(d.EmatError, match="NO_DEVICE"):
        b.build_usher_like(tips, 1)
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("blocks", ["1", "3", "16", "200"])
def test_any_number_of_workgroups_builds_the_same_tree(blocks, monkeypatch):
    """The graft loop runs on 1 .. one-per-CU workgroups that meet at a barrier of their own between the phases of a tip
    (EMAT_BUILD_BLOCKS overrides the default of one per 1 024 nodes): the tree does not depend on how many."""
    monkeypatch.setenv("EMAT_BUILD_BLOCKS", blocks)
    sc = make_scenario("C3", num_tips=1500, num_sites=29903, uncertain_tips=0.1)
    build_both(sc, 11)


@pytest.mark.gpu
def test_randomised_descriptors():
    """Seeded random scenarios (tree size, genome length, time span, mutation and gap density, tip-date uncertainty all vary):
    the device's tree equals the oracle's and passes the reference's closing checks.  EMAT_FUZZ_SEED / EMAT_FUZZ_CASES for longer hunts."""
    import os
    from delphy_amd.scenarios import random_scenario
    # (the second stream is round 4's catch: its case 4 has a tip with 537 deltas whose path to the root fits the grafting workgroup's LDS
    # staging while its deltas + the path's mutations do not -- the serial fall-back then read the path from the wrong buffer)
    streams = [(int(os.environ["EMAT_FUZZ_SEED"]), int(os.environ.get("EMAT_FUZZ_CASES", "16")))] if "EMAT_FUZZ_SEED" in os.environ else [(20261007, int(os.environ.get("EMAT_FUZZ_CASES", "16"))), (777001, 5)]
    for fuzz_seed, cases in streams:
        rng = np.random.default_rng(fuzz_seed)
        for case in range(cases):
            sc, _, _, what = random_scenario(rng, case, max_tips=600)
            try:
                build_both(sc, 1000 + case)
            except AssertionError as e:
                raise AssertionError("stream %d %s: %s" % (fuzz_seed, what, e))


@pytest.mark.gpu
def test_c4_built_on_the_device():
    """SURVEY 8(f).4 at the size the row names: config C4's 100 000 tips from their descriptors.  The oracle's restatement of the
    reference's builder is quadratic on one host core (38 s for 10 000 tips: hours here), so at full size the checks are the
    reference's own closing checks of the builder (phylo_tree.cpp:138-202: tree integrity, every tip reproduces its descriptor)
    and what the graft loop must conserve -- and the oracle is run on a 1/20 sample: the first 5 000 tip descriptors of the same
    configuration, whose tree must equal the device's bit for bit.  Time is printed and bounded loosely (round 4: the loop keeps
    positions and subtree sizes from tip to tip instead of recomputing them by pointer jumping, and the grafting thread works in LDS)."""
    import time
    sc = make_scenario("C4")
    ob = OracleBuild(sc.ref)
    tips = ob.tip_descs_of(sc.tree)
    assert tips.num_tips == 100000
    b = d.EmatBackend(sc.num_sites)
    try:
        b.set_ref_sequence(sc.ref)
        t0 = time.perf_counter(); tree = b.build_usher_like(tips, 7); dt = time.perf_counter() - t0
        print("C4 initial tree: %d tips in %.1f s on the device, %d mutations" % (tips.num_tips, dt, tree.mut_site.shape[0]))
        assert dt < 100.0, dt
        assert tree.num_nodes == 2 * tips.num_tips - 1 and np.all(tree.child0[: tips.num_tips] == -1) and np.all(tree.child0[tips.num_tips:] >= 0)
        rc, msg = ob.check(tree, tips)
        assert rc == 0, msg
        # the 1/20 sample against the oracle
        k = 5000
        d_hi, m_hi = int(tips.delta_offset[k]), int(tips.miss_offset[k])
        head = d.TipDescs(tips.t_min[:k].copy(), tips.t_max[:k].copy(), tips.delta_offset[: k + 1].copy(), tips.delta_site[:d_hi].copy(), tips.delta_to[:d_hi].copy(),
                          tips.miss_offset[: k + 1].copy(), tips.miss_start[:m_hi].copy(), tips.miss_end[:m_hi].copy())
        got = b.build_usher_like(head, 7)
        want = ob.build_usher_like(head, 7)
        assert got.root == want.root
        for f in FIELDS:
            x, y = getattr(got, f), getattr(want, f)
            assert x.shape == y.shape and np.array_equal(x, y), "%s differs on the 5 000-tip sample" % f
    finally:
        b.close(); ob.close()
