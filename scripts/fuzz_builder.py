#!/usr/bin/env python3
"""Longer hunts for the UShER-like builder: the cases of tests/test_initial_tree.py::test_randomised_descriptors, one line per case
(EMAT_FUZZ_SEED, EMAT_FUZZ_CASES, EMAT_FUZZ_FIRST: skip the cases before this one)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from delphy_amd.scenarios import random_scenario
from test_initial_tree import build_both
rng = np.random.default_rng(int(os.environ.get("EMAT_FUZZ_SEED", "20261007")))
first = int(os.environ.get("EMAT_FUZZ_FIRST", "0"))
for case in range(int(os.environ.get("EMAT_FUZZ_CASES", "16"))):
    sc, _, _, what = random_scenario(rng, case, max_tips=600)
    if case < first:
        continue
    print("case %d: %s" % (case, what), flush=True)
    build_both(sc, 1000 + case)
print("all cases ok", flush=True)
