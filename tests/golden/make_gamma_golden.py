"""Generates tests/golden/gamma_q.json: golden vectors for the regularized upper incomplete gamma function
Q(a, x) and its inverse, which replace Boost.Math 1.84 gamma_q / gamma_q_inv (absent from /root/reference;
called at core/safe_gamma_math.h:46,68).  Source of truth: scipy.special.gammaincc / gammainccinv
(scipy 1.15.3), over the sweep ranges of the reference's own test (tests/safe_gamma_math_tests.cpp:181-250:
a = f*m + 1 with f = 0.8, m = 0..; x from tiny to far in the tail).  Run once: python tests/golden/make_gamma_golden.py"""
import json
import os

import numpy as np
from scipy import special

rows = []
for m in list(range(0, 12)) + [15, 20, 30, 50, 100]:
    a = 0.8 * m + 1
    for x in [0.0, 1e-12, 1e-6, 1e-3, 0.01, 0.1, 0.5, 1.0, 2.0, 5.0, 10.0, 20.0, 50.0, 100.0, 300.0, 700.0]:
        rows.append({"a": a, "x": x, "q": float(special.gammaincc(a, x))})
inv = []
for m in [0, 1, 2, 3, 5, 8, 12, 20, 50]:
    a = 0.8 * m + 1
    for q in [1e-12, 1e-9, 1e-6, 1e-3, 0.01, 0.1, 0.3, 0.5, 0.7, 0.9, 0.99, 0.999999]:
        inv.append({"a": a, "q": q, "x": float(special.gammainccinv(a, q))})
out = {"source": "scipy.special.gammaincc / gammainccinv, scipy %s" % __import__("scipy").__version__, "q": rows, "q_inv": inv}
json.dump(out, open(os.path.join(os.path.dirname(__file__), "gamma_q.json"), "w"), indent=0)
print(len(rows), len(inv))
