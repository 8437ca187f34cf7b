"""The reference's own expectations, as data (tests/golden/reference_expectations.json, generated from the reference's
tests/*.cpp by tests/golden/make_reference_expectations.py), against the CPU oracle -- and, on a GPU box, against the
device's population-model code through a test hook of the C-ABI."""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

import delphy_amd as d
import oracle_ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_expectations.json")))


def _pop_model(m):
    if m["kind"] == "const":
        return d.PopModel.const(m["pop"])
    if m["kind"] == "exp":
        return d.PopModel.exp(m["t0"], m["n0"], m["g"], m["min_pop"])
    return d.PopModel.skygrid(np.array(m["x"]), np.array(m["gamma"]), m["type"] == "log_linear")


def _valid(m):
    if m["kind"] == "const":
        return m["pop"] > 0
    if m["kind"] == "exp":
        return m["n0"] > 0 and m["min_pop"] >= 0
    return len(m["x"]) >= 1 and len(m["x"]) == len(m["gamma"]) and all(a < b for a, b in zip(m["x"], m["x"][1:]))


def test_pop_model_expectations_of_the_reference():
    L = oracle_ffi.lib()
    checked = 0
    for c in G["pop_model"]:
        if c["op"] == "construct":
            assert not _valid(c["model"]), c           # the inputs the reference rejects are exactly the ones the boundary rejects
            if c["model"]["kind"] != "skygrid":
                b = d.EmatBackend(10, device=-1)
                try:
                    b.set_ref_sequence(np.zeros(10, np.uint8)); b.set_hky(1e-3, 2.0, (0.25, 0.25, 0.25, 0.25))
                    t = d.FlatTree.empty(3, 0, 0, 0); t.root = 0
                    t.parent[:] = [-1, 0, 0]; t.child0[:] = [1, -1, -1]; t.child1[:] = [2, -1, -1]; t.t[:] = [0.0, 1.0, 1.0]
                    t.t_min[1:] = 1.0; t.t_max[1:] = 1.0
                    b.upload_parts([t], [True], [1])
                    with pytest.raises(d.EmatError):
                        b.build_coalescent_parts(_pop_model(c["model"]), 0, 0.1)
                finally:
                    b.close()
            continue
        pm = _pop_model(c["model"]).c_struct()
        a = c["args"]
        if c["op"] == "pop_at_time":
            got = L.orc_pop_at_time(C.byref(pm), a[0])
        elif c["op"] == "log_N":
            got = math.log(L.orc_pop_at_time(C.byref(pm), a[0]))
        elif c["op"] == "pop_integral":
            got = L.orc_pop_integral(C.byref(pm), a[0], a[1])
        else:
            got = L.orc_intensity_integral(C.byref(pm), a[0], a[1])
        assert abs(got - c["expected"]) <= max(c["tol"], 1e-12 * abs(c["expected"])), (c, got)
        checked += 1
    assert checked >= 110


def _iv_build(s):
    """The set a golden case describes: its initial intervals, then its inserts, through the oracle's Interval_set::insert."""
    return oracle_ffi.interval_op(0, s["initial"] + s["inserts"])


def _interval_cases(ops):
    """ops = (elements, contains, binary set op -> list, binary predicate): called per golden case; returns how many ran."""
    n = 0
    for c in G["interval_set"]:
        if c["op"] == "elements":
            assert _iv_build(c["set"]) == c["expected"], c                      # insert merges overlapping AND adjacent intervals
        elif c["op"] == "contains":
            assert ops["contains"](_iv_build(c["set"]), c["site"]) == c["expected"], c
        elif c["op"] == "slow_elements":
            assert [l for a, b in _iv_build(c["set"]) for l in range(a, b)] == c["expected"]
        elif c["op"] in ("merge", "intersect", "subtract"):
            assert ops[c["op"]](_iv_build(c["a"]), _iv_build(c["b"])) == c["expected"], c
        elif c["op"] == "is_subset_of":
            if "is_subset_of" not in ops:
                continue
            assert ops["is_subset_of"](_iv_build(c["a"]), _iv_build(c["b"])) == c["expected"], c
        n += 1
    return n


def test_interval_set_expectations_of_the_reference():
    io = oracle_ffi.interval_op
    n = _interval_cases({"contains": lambda A, l: io(5, A, [l]), "merge": lambda A, B: io(1, A, B), "intersect": lambda A, B: io(2, A, B),
                         "subtract": lambda A, B: io(3, A, B), "is_subset_of": lambda A, B: io(4, A, B)})
    assert n >= 38


@pytest.mark.gpu
def test_device_interval_algebra_against_the_reference_expectations():
    b = d.EmatBackend(100)
    try:
        n = _interval_cases({"contains": lambda A, l: b.debug_interval_op(5, A, [l]), "merge": lambda A, B: b.debug_interval_op(1, A, B),
                             "intersect": lambda A, B: b.debug_interval_op(2, A, B), "subtract": lambda A, B: b.debug_interval_op(3, A, B)})
        assert n >= 28
    finally:
        b.close()


def _tree_with_node_times(stage, num_tips):
    """A binary tree whose node times are the golden case's (the grid prior looks at times and tip flags only): coalescences
    are joined from the latest to the earliest, each taking the two latest lineages available."""
    n = 2 * num_tips - 1
    t = np.zeros(n)
    for k, v in stage["coalescence_times"].items(): t[int(k)] = v
    for k, v in stage["tip_times"].items(): t[int(k)] = v
    tree = d.FlatTree.empty(n, 0, 0, 0)
    tree.t[:] = t
    tips = list(range(num_tips - 1, n))
    tree.t_min[tips] = t[tips].astype(np.float32); tree.t_max[tips] = t[tips].astype(np.float32)
    pool = sorted(tips, key=lambda i: t[i])
    for c in sorted(range(num_tips - 1), key=lambda i: -t[i]):
        kids = [pool.pop(), pool.pop()]                      # the two latest lineages: both at or after t[c] for these data
        assert all(t[k] >= t[c] for k in kids)
        tree.child0[c], tree.child1[c] = kids
        tree.parent[kids[0]] = tree.parent[kids[1]] = c
        pool.append(c); pool.sort(key=lambda i: t[i])
    tree.root = pool[0]; tree.parent[tree.root] = -1
    return tree


def _grid_prior_through(engine_cls, stage, sc, **kw):
    tree = _tree_with_node_times(stage, sc["num_tips"])
    e = engine_cls(50, **kw)
    try:
        e.set_ref_sequence(np.zeros(50, np.uint8)); e.set_hky(1e-3, 2.0, (0.25, 0.25, 0.25, 0.25)); e.set_flags(0.0)
        e.upload_parts([tree], [True], [1])
        e.build_coalescent_parts(d.PopModel.const(sc["pop"]), 0, 1.0)
        if isinstance(e, d.EmatBackend):
            return e.scalable_coalescent_log_prior(sc["t_ref"], sc["t_step"])
        return e.scalable_log_prior(0, sc["t_ref"], sc["t_step"])
    finally:
        e.close()


def test_scalable_coalescent_expectations_of_the_reference():
    """scalable_coalescent_tests.cpp `log_prior`: the oracle's Scalable_coalescent_prior on trees carrying the case's node times."""
    sc = G["scalable_coalescent"]
    assert len(sc["stages"]) == 3
    for st in sc["stages"]:
        got = _grid_prior_through(oracle_ffi.OracleEngine, st, sc)
        assert abs(got - st["expected_log_prior"]) <= st["tol"], (st, got)


@pytest.mark.gpu
def test_device_scalable_coalescent_against_the_reference_expectations():
    """The same three expectations against emat_get_scalable_coalescent_log_prior (k_scalable_prior on the device)."""
    sc = G["scalable_coalescent"]
    for st in sc["stages"]:
        got = _grid_prior_through(d.EmatBackend, st, sc)
        assert abs(got - st["expected_log_prior"]) <= st["tol"], (st, got)


@pytest.mark.gpu
def test_device_population_models_against_the_reference_expectations():
    b = d.EmatBackend(100)
    try:
        checked = 0
        for c in G["pop_model"]:
            if c["op"] not in ("pop_at_time", "pop_integral", "log_N"):
                continue                                   # intensity_integral is not on the device path
            pm = _pop_model(c["model"])
            a = c["args"]
            if c["op"] == "pop_integral":
                got = b.debug_pop(pm, 1, [a[0]], [a[1]])[0]
            else:
                got = b.debug_pop(pm, 0, [a[0]], [0.0])[0]
                if c["op"] == "log_N":
                    got = math.log(got)
            assert abs(got - c["expected"]) <= max(c["tol"], 1e-12 * abs(c["expected"])), (c, got)
            checked += 1
        assert checked >= 80
    finally:
        b.close()


def _fixture_tree(q, all_times_equal):
    n = len(q["parent"])
    tree = d.FlatTree.empty(n, 0, 0, 0)
    tree.root = q["root"]; tree.parent[:] = q["parent"]
    tree.child0[:] = [k[0] for k in q["children"]]; tree.child1[:] = [k[1] for k in q["children"]]
    tree.t[:] = 0.0 if all_times_equal else q["t"]
    tips = [i for i in range(n) if q["children"][i][0] < 0]
    tree.t_min[tips] = tree.t[tips].astype(np.float32); tree.t_max[tips] = tree.t[tips].astype(np.float32)
    return tree


def _tree_queries_through(engine_cls, **kw):
    """phylo_tree_tests.cpp:365-525 as data: the three tables of the reference over its fixture tree (with its times, and with
    every time zero), asked of `engine_cls` query by query."""
    q = G["phylo_tree_queries"]
    for table, op, equal in (("find_MRCA_of", 0, False), ("find_MRCA_of_all_times_equal", 0, True), ("descends_from", 1, False)):
        rows = np.array(q[table], np.int32)
        assert rows.shape == (36, 3)
        e = engine_cls(50, **kw)
        try:
            e.set_ref_sequence(np.zeros(50, np.uint8)); e.set_hky(1e-3, 2.0, (0.25, 0.25, 0.25, 0.25)); e.set_flags(3.0)
            e.upload_parts([_fixture_tree(q, equal)], [True], [1])
            e.build_coalescent_parts(d.PopModel.const(10.0), 0, 1.0)
            got = e.debug_tree_query(0, op, rows[:, 0], rows[:, 1]) if isinstance(e, d.EmatBackend) else e.tree_query(0, op, rows[:, 0], rows[:, 1])
        finally:
            e.close()
        assert got.tolist() == rows[:, 2].tolist(), (table, [(r.tolist(), int(g)) for r, g in zip(rows, got) if r[2] != g])


def test_tree_query_expectations_of_the_reference():
    _tree_queries_through(oracle_ffi.OracleEngine)


@pytest.mark.gpu
def test_device_tree_queries_against_the_reference_expectations():
    """The same tables against the moves' own find_MRCA_of / descends_from on the device (emat_debug_tree_query)."""
    _tree_queries_through(d.EmatBackend)


# ---- tests/spr_move_tests.cpp as data: graft analysis, peel / apply, whole re-attachments, the history sampler ----------------------
import graft_golden as gg  # noqa: E402

SM = G["spr_move"]


def _with_fixture(engine_cls, fx, can_change_root, seed, fn):
    e = engine_cls(len(fx["ref_sequence"]))
    try:
        gg.configure_fixture(e, fx, can_change_root, seed)
        return fn(e)
    finally:
        e.close()


def _analyze_graft_expectations_through(engine_cls):
    """All nine analyze_graft_* tests of the reference (spr_move_tests.cpp:142-1433) and the tricky rooty graft's peel / closed-mutations /
    peel-and-reapply tests (:1435-1516): the fixture as the engine's only part, the engine's own analysis, the reference's numbers."""
    assert len(SM["analyze_graft"]) == 9 and len(SM["peel_apply"]) == 3
    for t in SM["analyze_graft"]:
        fx = SM["fixtures"][t["fixture"]]
        r = _with_fixture(engine_cls, fx, t["can_change_root"], 1, lambda e: e.debug_graft(0, t["X"], SM["mu_JC"], 0))
        gg.check_analysis(r["grafts"][0], t, "%s.%s (spr_move_tests.cpp:%d)" % (t["fixture"], t["test"], t["line"]))
    for t in SM["peel_apply"]:
        fx = SM["fixtures"][t["fixture"]]
        r, tree = _with_fixture(engine_cls, fx, t["can_change_root"], 1, lambda e: (e.debug_graft(0, t["X"], SM["mu_JC"], t["mode"]), e.part_download(0)))
        gg.check_peel_apply(r, tree, fx, t, "%s.%s (spr_move_tests.cpp:%d)" % (t["fixture"], t["test"], t["line"]))


def test_analyze_graft_expectations_of_the_reference():
    _analyze_graft_expectations_through(oracle_ffi.OracleEngine)


@pytest.mark.gpu
def test_device_graft_analysis_against_the_reference_expectations():
    """The same numbers against the moves' own device code (emat_debug_graft): analyze_graft, peel_graft, apply_graft run on lane 0 of a
    wavefront on the fixture's slab, as inside a chain -- not through the oracle."""
    _analyze_graft_expectations_through(d.EmatBackend)


def _full_spr_moves_through(engine_cls, seeds):
    """run_full_spr_move_test (spr_move_tests.cpp:1518-1641) over the reference's (X, SS, t) cases of all five fixtures: analyze the old graft,
    peel it, re-attach X above SS at t, propose a new graft from the random stream, apply it.  Then, as there: the tree is sound, log G moved
    by exactly new.delta_log_G - old.delta_log_G, the incrementally kept lambda_i / missing-site counts equal a recomputation, every tip
    still carries its sequence, and a fresh analysis of X finds the graft that was proposed."""
    device = engine_cls is d.EmatBackend
    done = 0
    for lst in SM["full_spr_move"]:
        fx = SM["fixtures"][lst["fixture"]]
        ref = fx["ref_sequence"]
        before = gg.tip_sequences(gg.fixture_tree(fx), ref)
        assert lst["seeds"] == 200
        for X, SS, t in lst["cases"]:
            for seed in range(seeds):
                def run(e):
                    if device:
                        e.recalc_derived()
                    r = e.debug_graft(0, X, SM["mu_JC"], 3, SS, t)
                    if device:
                        e.check_derived(1.0)        # Subrun::check_derived_quantities on the device: lambda_i, log G (1e-6), missing-site counts
                    else:
                        inc, scratch = e.part_log_G(0)
                        assert abs(inc - scratch) <= 1e-6, (lst["fixture"], X, SS, t, seed, inc, scratch)
                    return r, e.part_download(0), e.debug_graft(0, X, SM["mu_JC"], 0)
                r, tree, redux = _with_fixture(engine_cls, fx, True, seed + 12345, run)
                what = "%s X=%d SS=%d t=%g seed %d" % (lst["fixture"], X, SS, t, seed)
                assert len(r["grafts"]) == 2, what
                gg.same_grafts(redux["grafts"][0], r["grafts"][1])
                after = gg.tip_sequences(tree, ref)           # (also checks that every mutation starts from the state the sequence holds)
                for tip, seq in before.items():
                    assert np.array_equal(seq, after[tip]), (what, tip, seq, after[tip])
                chk = oracle_ffi.OracleEngine(len(ref))
                try:
                    chk.set_ref_sequence(np.asarray(ref, np.uint8)); chk.set_hky(1e-3, 2.0, (0.25, 0.25, 0.25, 0.25)); chk.set_flags(10.0)
                    chk.upload_parts([tree], [True], [1])
                    rc, msg = chk.part_check(0)               # assert_phylo_tree_integrity
                    assert rc == 0, (what, msg)
                finally:
                    chk.close()
                P = int(tree.parent[X])
                kids = {int(tree.child0[P]), int(tree.child1[P])}
                assert tree.t[P] == t and X in kids and (SS in kids or SS == P), what      # (SS == P: P slid along its own branch)
                done += 1
    return done


def test_full_spr_moves_of_the_reference_cases():
    assert _full_spr_moves_through(oracle_ffi.OracleEngine, 20) == 73 * 20


@pytest.mark.gpu
def test_device_full_spr_moves_of_the_reference_cases():
    """Spr_move::move + propose_new_graft + apply_graft as the device runs them inside an SPR move, on the reference's 73 cases."""
    assert _full_spr_moves_through(d.EmatBackend, 6) == 73 * 6


def _history_frequencies_through(engine_cls, num_histories):
    """sample_mutational_history's statistical test (spr_move_tests.cpp:1795-1961): histories of length T = 1 / mu_JC ending at random points
    of the simple fixture, from the start sequence TGCA; every history must be sorted, inside (t_end - T, t_end], and lead from the start
    sequence to the tree's sequence at its end point; mutations at the watched site (A at both ends everywhere) make a history `unusual`
    (any) or `super-unusual` (more than two), with the probabilities the reference derives, within its three binomial standard errors."""
    h = SM["sample_mutational_history"]
    fx = SM["fixtures"][h["fixture"]]
    L = len(fx["ref_sequence"])
    tree = gg.fixture_tree(fx)
    rng = np.random.default_rng(20261003)
    branch = rng.integers(0, tree.num_nodes - 1, num_histories).astype(np.int32)
    branch = np.where(branch >= fx["root"], branch + 1, branch).astype(np.int32)        # any node but the root
    lo, hi = tree.t[tree.parent[branch]], tree.t[branch]
    t_end = lo + (hi - lo) * rng.random(num_histories)
    T = h["mu_T"] / SM["mu_JC"]
    start = np.asarray(h["target_start_seq"], np.uint8)
    hist = _with_fixture(engine_cls, fx, True, 12345, lambda e: e.debug_sample_history(0, branch, t_end, start, T, SM["mu_JC"]))
    # the tree's sequence at (branch, t): the tip sequences machinery on a tree cut there
    def seq_at(b, t):
        path, n = [], int(b)
        while n >= 0:
            path.append(n); n = int(tree.parent[n])
        seq = np.array(fx["ref_sequence"], np.int64)
        for n in reversed(path):
            for k in range(int(tree.mut_offset[n]), int(tree.mut_offset[n + 1])):
                if n != b or tree.mut_t[k] <= t:
                    seq[int(tree.mut_site[k])] = tree.mut_to[k]
        return seq
    unusual = super_unusual = 0
    for i in range(num_histories):
        seq = start.astype(np.int64).copy()
        prev = -np.inf
        for fr, site, to, tm in hist[i]:
            assert prev <= tm and t_end[i] - T <= tm <= t_end[i], (i, hist[i])
            assert seq[site] == fr and fr != to, (i, hist[i]); seq[site] = to; prev = tm
        end = seq_at(branch[i], t_end[i])
        # sites missing at the end point have no defined state there; calc_site_state_at still answers for them from the lists above
        assert np.array_equal(seq, end), (i, seq, end, hist[i])
        at = sum(1 for m in hist[i] if m[1] == h["watched_site"])
        unusual += at > 0; super_unusual += at > 2
    pu, ps = unusual / num_histories, super_unusual / num_histories
    eu, es = math.sqrt(pu * (1 - pu) / num_histories), math.sqrt(ps * (1 - ps) / num_histories)
    assert abs(pu - h["expected_p_unusual"]) <= h["sigmas"] * eu, (pu, eu, h["expected_p_unusual"])
    assert abs(ps - h["expected_p_super_unusual"]) <= h["sigmas"] * es, (ps, es, h["expected_p_super_unusual"])
    return pu, ps


def test_history_sampler_frequencies_of_the_reference():
    _history_frequencies_through(oracle_ffi.OracleEngine, SM["sample_mutational_history"]["num_histories"])


@pytest.mark.gpu
def test_device_history_sampler_frequencies_of_the_reference():
    """The device's own sample_mutational_history + adjust_mutational_history (emat_debug_sample_history), 25 000 histories."""
    _history_frequencies_through(d.EmatBackend, SM["sample_mutational_history"]["num_histories"])


# ---- tests/phylo_tree_calc_tests.cpp as data: derived quantities and the global moves' sufficient statistics of its fixture -----------
PC = G["phylo_tree_calc"]


def _phylo_tree_calc_expectations_through(engine_cls):
    """calc_lambda_i, calc_num_sites_missing_at_every_node, calc_log_G_below_root, calc_log_root_prior (with states of probability zero),
    calc_num_muts / _beta_ab / _l and calc_Ttwiddle_beta_a on the reference's fixture (phylo_tree_calc_tests.cpp:14-116), each against the
    number the reference's test states (:248-284, 355-380, 381-439, 441-482, 497-505, 557-607)."""
    fx = PC["fixture"]
    n = len(fx["nodes"])
    def derived(includes_root, pi=None):
        f = dict(fx); f["evo"] = dict(fx["evo"])
        if pi is not None:
            f["evo"]["pi"] = pi
        def go(e):
            e.recalc_derived()
            lam, miss, log_G, _ = e.part_derived(0, n)
            stats = e.global_stats(2) if includes_root else None
            return lam, miss, log_G, stats, (e.num_muts_l() if includes_root else None)
        return _with_fixture(engine_cls, f, includes_root, 1, go)
    lam, miss, below, _, _ = derived(False)                                   # a part without the run's root: log G is the sum below its root
    assert np.all(np.abs(lam - np.array(PC["lambda_i"]["expected"])) <= PC["lambda_i"]["tol"]), (lam, PC["lambda_i"])
    assert miss.tolist() == PC["num_sites_missing"]["expected"]
    assert abs(below - PC["log_G_below_root"]["expected"]) <= PC["log_G_below_root"]["tol"], (below, PC["log_G_below_root"])
    lam2, miss2, with_root, (T, M, num), per_site = derived(True)
    assert np.array_equal(lam, lam2) and np.array_equal(miss, miss2)
    assert num == PC["num_muts"] and M.tolist() == PC["num_muts_beta_ab"] and per_site.tolist() == PC["num_muts_l"]
    assert np.all(np.abs(T - np.array(PC["Ttwiddle_beta_a"]["expected"])) <= PC["Ttwiddle_beta_a"]["tol"]), (T, PC["Ttwiddle_beta_a"])
    for case in PC["log_root_prior"]:
        _, _, g, _, _ = derived(True, case["pi"])
        prior = g - below if np.isfinite(g) else g                                # Subrun::calc_cur_log_G = root prior + sum below the root
        if case["expected"] == "-inf":
            assert prior == -np.inf, (case, prior)
        else:
            assert abs(prior - case["expected"]) <= case["tol"], (case, prior)


def test_phylo_tree_calc_expectations_of_the_reference():
    _phylo_tree_calc_expectations_through(oracle_ffi.OracleEngine)


@pytest.mark.gpu
def test_device_derived_quantities_against_the_reference_expectations():
    """The same numbers from the kernels: k_recalc_derived (lambda_i, missing-site counts, branch log-G sums, root prior),
    k_global_stats and k_num_muts_l on the fixture's slab."""
    _phylo_tree_calc_expectations_through(d.EmatBackend)


# ---- tests/tree_editing_tests.cpp as data: ten editing sessions, step by step -----------------------------------------------------------
def _tree_editing_expectations_through(engine_cls):
    """Every test of the reference's tree_editing_tests.cpp (:127-1115): slide up / down (a missation swallowing a mutation, mutations
    exactly on a branch end, the root sliding down), hop up, flip, hop down and the eight-step SPR -- the tree each test starts from, the
    session's steps through the engine's own editing primitives, then the reference's expected times, mutation lists (in its order where
    it states one), missations, parents and root; and lambda_i / missing-site counts kept up to date by the steps."""
    device = engine_cls is d.EmatBackend
    assert len(G["tree_editing"]) == 10
    for t in G["tree_editing"]:
        fx = t["tree"]
        what = "%s.%s (tree_editing_tests.cpp:%d)" % (t["fixture"], t["test"], t["line"])
        def run(e):
            if device:
                e.recalc_derived()
                e.debug_edit(0, t["X"], t["ops"])
                n = len(fx["nodes"])
                lam, miss, _, _ = e.part_derived(0, n)          # as the session's steps left them
                e.recalc_derived()
                lam2, miss2, _, _ = e.part_derived(0, n)        # from scratch (k_recalc_derived) on the edited tree
                assert np.all(np.abs(lam - lam2) <= 1e-6) and np.array_equal(miss, miss2), (what, lam, lam2, miss, miss2)
            else:
                dev, bad = e.debug_edit(0, t["X"], t["ops"])
                assert dev <= 1e-6 and bad == 0, (what, dev, bad)
            return e.part_download(0)
        tree = _with_fixture(engine_cls, fx, True, 1, run)
        ex = t["expect"]
        for node, want in ex["t"].items():
            assert abs(tree.t[int(node)] - want) <= 1e-6, (what, node, tree.t[int(node)], want)
        for node, want in ex["parent"].items():
            assert int(tree.parent[int(node)]) == want, (what, node, int(tree.parent[int(node)]), want)
        if "root" in ex:
            assert tree.root == ex["root"], what
        for node, want in ex["mutations"].items():
            n = int(node)
            a, b = int(tree.mut_offset[n]), int(tree.mut_offset[n + 1])
            got = [[int(tree.mut_from[k]), int(tree.mut_site[k]), int(tree.mut_to[k]), float(tree.mut_t[k])] for k in range(a, b)]
            if "ordered" in want:
                assert got == want["ordered"], (what, node, got, want)
            else:
                assert sorted(got) == sorted(want["unordered"]), (what, node, got, want)
        for node, want in ex["missations"].items():
            _, miss = gg.node_lists(tree, fx["ref_sequence"], int(node))
            assert miss == want, (what, node, miss, want)


def test_tree_editing_expectations_of_the_reference():
    _tree_editing_expectations_through(oracle_ffi.OracleEngine)


@pytest.mark.gpu
def test_device_tree_editing_against_the_reference_expectations():
    """The same ten sessions through the device's own edit_slide_P_along_branch / edit_do_hop_up / edit_flip / edit_hop_down
    (emat_debug_edit), the code Spr_move::move is made of on the GPU."""
    _tree_editing_expectations_through(d.EmatBackend)
