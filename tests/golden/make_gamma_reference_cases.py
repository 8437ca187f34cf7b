"""Generates tests/golden/gamma_reference_cases.json: every (a, x) / (a, Q) point that the reference's own test of its
incomplete-gamma wrappers evaluates (tests/safe_gamma_math_tests.cpp:34-250), as DATA, with the value the reference
compares against.  The reference compares with Boost.Math 1.84 gamma_q (absent from /root/reference); scipy.special
(Cephes / Boost-derived igam) stands in for it here, as in make_gamma_golden.py.  Each row names the reference test it
comes from and the assertion the reference makes there:

  near      |got - expected| <= tol                  (BasicValues, LargeX, LargeA, NearMode, AtZero, LogIntegral*)
  roundtrip |Q_inv(a, Q(a, x)) - x| <= tol           (InvRoundTrip, InvExtremeQ)
  finite    not NaN, not inf (and > 0 for inverses)  (ComprehensiveSweep*)

`finite` rows still carry scipy's value, so the tests can hold oracle and device to it as well wherever it is
representable.  Run once: python tests/golden/make_gamma_reference_cases.py"""
import json
import math
import os

from scipy import special


def q(a, x):
    return float(special.gammaincc(a, x))


def geometric(first, last, factor):
    x = first
    while x <= last:
        yield x
        x *= factor


rows_q, rows_inv, rows_rt, rows_logint = [], [], [], []

# BasicValues / LargeX / LargeA / NearMode (:34-63)
for name, a, x, tol in (("BasicValues", 2.0, 5.0, 1e-12), ("BasicValues", 1.0, 1.0, 1e-12), ("BasicValues", 10.0, 5.0, 1e-12),
                        ("LargeX", 271.4, 6601.0, 1e-12), ("LargeA", 1000.0, 100.0, 1e-12),
                        ("NearMode", 271.4, 280.0, 1e-12), ("NearMode", 500.0, 1000.0, 1e-12)):
    rows_q.append({"test": name, "kind": "near", "a": a, "x": x, "q": q(a, x), "tol": tol})
for a in (1.0, 10.0, 100.0, 1000.0):   # AtZero (:65-69) and the x = 0 block of the sweep (:234-236)
    rows_q.append({"test": "AtZero", "kind": "near", "a": a, "x": 0.0, "q": 1.0, "tol": 1e-12})

# InvRoundTrip (:71-81)
for a in (0.5, 1.0, 2.0, 5.0, 10.0, 50.0, 100.0, 271.4, 500.0):
    for x in (0.1, 1.0, 5.0, 10.0, a / 2, a, a * 2, a * 10):
        if q(a, x) <= 0.0 or q(a, x) >= 1.0:
            continue   # the reference skips these too (:76)
        rows_rt.append({"test": "InvRoundTrip", "a": a, "x": x, "tol": 1e-6 * (x + 1)})

# InvExtremeQ (:83-95): x = Q_inv(a, Q) then Q(a, x) back, relative 1e-10 / absolute 1e-10
for a in (50.0, 100.0, 271.4, 500.0):
    for Q, rel in ((1e-300, True), (1.0 - 1e-15, False)):
        rows_inv.append({"test": "InvExtremeQ", "kind": "q_back", "a": a, "q": Q, "x": float(special.gammainccinv(a, Q)),
                         "tol": Q * 1e-10 if rel else 1e-10})
# InvBoundaries (:97-105)
for a in (1.0, 10.0, 100.0, 271.4):
    rows_inv.append({"test": "InvBoundaries", "kind": "exact", "a": a, "q": 0.0, "x": "inf"})
    rows_inv.append({"test": "InvBoundaries", "kind": "near", "a": a, "q": 1.0, "x": 0.0, "tol": 1e-10})

# SafeLogGammaIntegral* (:107-124)
for a, lo, hi in ((5.0, 1.0, 10.0), (271.4, 148.0, 6601.0)):
    rows_logint.append({"test": "LogIntegral", "a": a, "x_min": lo, "x_max": hi, "log": math.log(q(a, lo) - q(a, hi)), "tol": 1e-12})
rows_logint.append({"test": "LogIntegralAtBounds", "a": 5.0, "x_min": 1000.0, "x_max": 2000.0, "log": "-inf", "tol": 0.0})

# ComprehensiveSweepGammaQ (:181-245)
def sweep(a, x):
    rows_q.append({"test": "ComprehensiveSweepGammaQ", "kind": "finite", "a": a, "x": x, "q": q(a, x)})


for a in (0.01, 0.1, 0.5, 1.0, 1.5, 2.0, 3.0, 5.0, 10.0):
    for x in geometric(0.001, 1000, 1.7):
        sweep(a, x)
for a in (20.0, 50.0, 100.0, 150.0, 200.0):
    for x in geometric(0.01, 10000, 1.5):
        sweep(a, x)
for a in (250.0, 271.4, 300.0, 400.0, 500.0, 750.0, 1000.0):
    for ratio in (0.01, 0.1, 0.5, 0.8, 0.9, 0.95, 0.99, 1.0, 1.01, 1.05, 1.1, 1.2, 1.5, 2.0, 5.0, 10.0, 20.0, 50.0):
        sweep(a, a * ratio)
    for x in (1000.0, 5000.0, 10000.0, 50000.0, 100000.0, 0.001, 0.01, 0.1, 1.0, 10.0):
        sweep(a, x)
for a in (10.0, 10.5, 100.0, 100.5):
    for x in geometric(0.1, 1000, 2):
        sweep(a, x)

# ComprehensiveSweepGammaQInv (:247-262)
for a in (0.5, 1.0, 2.0, 5.0, 10.0, 50.0, 100.0, 200.0, 271.4, 300.0, 500.0, 1000.0):
    for Q in (1e-300, 1e-200, 1e-100, 1e-50, 1e-20, 1e-10, 1e-5, 0.001, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 0.999, 0.9999,
              1.0 - 1e-10, 1.0 - 1e-15):
        rows_inv.append({"test": "ComprehensiveSweepGammaQInv", "kind": "finite", "a": a, "q": Q, "x": float(special.gammainccinv(a, Q))})

out = {"source": "points of /root/reference/tests/safe_gamma_math_tests.cpp; expected values scipy.special %s" % __import__("scipy").__version__,
       "q": rows_q, "q_inv": rows_inv, "roundtrip": rows_rt, "log_integral": rows_logint}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "gamma_reference_cases.json"), "w"), indent=0)
print(len(rows_q), len(rows_inv), len(rows_rt), len(rows_logint))
