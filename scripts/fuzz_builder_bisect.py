#!/usr/bin/env python3
"""Which tip is the first that the device's UShER-like builder places differently from the oracle: prefixes of one fuzz case
(EMAT_FUZZ_SEED, EMAT_FUZZ_CASE), bisected."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import delphy_amd as d
from delphy_amd.scenarios import random_scenario
from oracle_ffi import OracleBuild
from test_initial_tree import FIELDS
rng = np.random.default_rng(int(os.environ.get("EMAT_FUZZ_SEED", "20261007")))
want_case = int(os.environ.get("EMAT_FUZZ_CASE", "0"))
for case in range(want_case + 1):
    sc, _, _, what = random_scenario(rng, case, max_tips=600)
print(what, flush=True)
seed = 1000 + want_case
ob = OracleBuild(sc.ref); tips = ob.tip_descs_of(sc.tree)
b = d.EmatBackend(sc.num_sites); b.set_ref_sequence(sc.ref)
def head(k):
    dh, mh = int(tips.delta_offset[k]), int(tips.miss_offset[k])
    return d.TipDescs(tips.t_min[:k].copy(), tips.t_max[:k].copy(), tips.delta_offset[: k + 1].copy(), tips.delta_site[:dh].copy(), tips.delta_to[:dh].copy(),
                      tips.miss_offset[: k + 1].copy(), tips.miss_start[:mh].copy(), tips.miss_end[:mh].copy())
def same(k):
    h = head(k)
    try:
        got = b.build_usher_like(h, seed)
    except Exception as e:
        return False, str(e)[:120]
    want = ob.build_usher_like(h, seed)
    for f in FIELDS:
        x, y = getattr(got, f), getattr(want, f)
        if x.shape != y.shape or not np.array_equal(x, y):
            return False, f
    return True, ""
lo, hi = 2, tips.num_tips          # same(lo) assumed, same(hi) false
print("whole:", same(hi), flush=True)
while hi - lo > 1:
    mid = (lo + hi) // 2
    ok, why = same(mid)
    print("prefix %d: %s %s" % (mid, ok, why), flush=True)
    if ok: lo = mid
    else: hi = mid
X = hi - 1
print("first tip placed differently: X = %d: %d deltas, %d missing intervals, t in [%g, %g]" % (X, tips.delta_offset[X + 1] - tips.delta_offset[X], tips.miss_offset[X + 1] - tips.miss_offset[X], tips.t_min[X], tips.t_max[X]))
h = head(hi); got = b.build_usher_like(h, seed); want = ob.build_usher_like(h, seed)
n = hi; P = X + n - 1
for name, t in (("device", got), ("oracle", want)):
    print(name, "root", t.root, "| X parent", t.parent[X], "t_X", t.t[X], "| P parent", t.parent[P], "children", t.child0[P], t.child1[P], "t_P", t.t[P], "| muts on X", t.mut_offset[X + 1] - t.mut_offset[X], "on P", t.mut_offset[P + 1] - t.mut_offset[P])
