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
