"""The CPU oracle against the reference's own known-answer tests and golden vectors (no GPU)."""
import json
import os
import subprocess

import numpy as np

import oracle_ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_reproduces_reference_known_answer_tests():
    # oracle/orc_tests.cpp re-evaluates the fixtures and closed-form expectations of /root/reference/tests/*.cpp
    # (interval sets, missation maps, site deltas, phylo_tree_calc, spr_study regions, spr_move graft analyses and
    # their 200-1000-seed property tests, tree editing via full-move reversibility, coalescent priors, pop models)
    r = subprocess.run([os.path.join(ROOT, "oracle", "_build", "orc_tests")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    assert " 0 failed" in r.stdout


def test_gamma_q_against_scipy_golden_vectors():
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "gamma_q.json")))
    L = oracle_ffi.lib()
    for row in g["q"]:
        got = L.orc_gamma_q(row["a"], row["x"])
        assert abs(got - row["q"]) <= 1e-12 + 1e-10 * row["q"], row   # reference tolerance: 1e-12 abs (safe_gamma_math_tests.cpp)
    for row in g["q_inv"]:
        got = L.orc_gamma_q_inv(row["a"], row["q"])
        assert abs(got - row["x"]) <= 1e-9 * max(1.0, row["x"]), (row, got)
        assert abs(L.orc_gamma_q(row["a"], got) - row["q"]) <= 1e-11 * row["q"] + 1e-14


def check_gamma_reference_cases(gamma_q, gamma_q_inv):
    """Every point and assertion of the reference's test of its incomplete-gamma wrappers
    (/root/reference/tests/safe_gamma_math_tests.cpp:34-262, kept as data in tests/golden/gamma_reference_cases.json),
    applied to `gamma_q(a[], x[])` / `gamma_q_inv(a[], q[])`.  Beyond what the reference asserts, the sweeps are also held
    to scipy's values wherever the problem is well conditioned."""
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "gamma_reference_cases.json")))
    rows = g["q"]
    got = gamma_q([r["a"] for r in rows], [r["x"] for r in rows])
    for r, v in zip(rows, got):
        assert np.isfinite(v) and 0.0 <= v <= 1.0, (r, v)
        if r["kind"] == "near":
            assert abs(v - r["q"]) <= r["tol"], (r, v)                         # the reference's assertion
        assert abs(v - r["q"]) <= 1e-12 + 1e-9 * r["q"], (r, v)               # and scipy's value, relative in the tails
    rows = g["q_inv"]
    a = [r["a"] for r in rows]
    got = gamma_q_inv(a, [r["q"] for r in rows])
    back = gamma_q(a, [min(v, 1e300) for v in got])
    for r, v, q in zip(rows, got, back):
        if r["kind"] == "exact":
            assert v == np.inf, (r, v)
            continue
        assert np.isfinite(v) and v >= 0.0, (r, v)
        if r["kind"] == "near":
            assert abs(v - r["x"]) <= r["tol"], (r, v)
            continue
        if r["kind"] == "finite":
            assert v > 0.0, (r, v)
        tol = r["tol"] if r["kind"] == "q_back" else 1e-10 * r["q"] + 1e-15
        assert abs(q - r["q"]) <= tol, (r, v, q)                                # Q(a, Q_inv(a, Q)) = Q (InvExtremeQ)
        if 1.0 - r["q"] >= 1e-6:                                                # x itself, where 1 - Q still carries digits
            assert abs(v - r["x"]) <= 1e-8 * max(1.0, r["x"]), (r, v)
    rows = g["roundtrip"]
    a = [r["a"] for r in rows]
    got = gamma_q_inv(a, gamma_q(a, [r["x"] for r in rows]))
    for r, v in zip(rows, got):
        assert abs(v - r["x"]) <= r["tol"], (r, v)
    for r in g["log_integral"]:                                                # safe_log_gamma_integral = log(Q(x_min) - Q(x_max))
        hi, lo = gamma_q([r["a"], r["a"]], [r["x_min"], r["x_max"]])
        assert hi >= lo
        with np.errstate(divide="ignore"):
            v = float(np.log(hi - lo))
        if r["log"] == "-inf":
            assert v == -np.inf, (r, v)
        else:
            assert abs(v - r["log"]) <= r["tol"], (r, v)


def test_gamma_q_on_the_reference_tests_own_points():
    L = oracle_ffi.lib()
    check_gamma_reference_cases(lambda a, x: [L.orc_gamma_q(ai, xi) for ai, xi in zip(a, x)],
                                lambda a, q: [L.orc_gamma_q_inv(ai, qi) for ai, qi in zip(a, q)])
    # the sampler built on them stays inside its bounds (safe_gamma_math_tests.cpp:126-179 run inside oracle/orc_tests.cpp)


def test_rng_stream_known_answer():
    # Philox4x32-10 known-answer test (Random123 kat_vectors: counter = 0, key = 0)
    import ctypes as C
    out = (C.c_uint32 * 4)()
    oracle_ffi.lib().orc_rng_block(0, 0, out)
    assert [hex(x) for x in out] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
