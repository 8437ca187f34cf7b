"""The seeded random scenarios of the GPU sweeps (delphy_amd.scenarios.random_scenario), checked where no GPU is needed: they are
reproducible, they cover every population model / site-rate / partition combination, and the oracle takes each of them (its
from-scratch log-posterior terms are finite and a few hundred moves run)."""
import numpy as np

from delphy_amd.scenarios import random_scenario
from helpers import configure, split_parts
from oracle_ffi import OracleEngine


def test_random_scenarios_are_reproducible_and_valid():
    seen = set()
    for case in range(12):
        a = random_scenario(np.random.default_rng([5, case]), case)
        b = random_scenario(np.random.default_rng([5, case]), case)
        sc, nu_l, evo, what = a
        assert what == b[3] and np.array_equal(sc.tree.parent, b[0].tree.parent) and np.array_equal(sc.tree.mut_t, b[0].tree.mut_t) and np.array_equal(sc.ref, b[0].ref)
        seen.add((sc.pop.kind, int(getattr(sc.pop, "skygrid_type", 0) or 0), nu_l is not None, evo is not None))
        t = sc.tree
        n = t.num_nodes
        assert n % 2 == 1 and t.parent[t.root] == -1 and int(np.sum(t.parent == -1)) == 1
        inner = t.child0 >= 0
        assert np.all(t.parent[t.child0[inner]] == np.flatnonzero(inner)) and np.all(t.parent[t.child1[inner]] == np.flatnonzero(inner))
        assert np.all(t.t[t.parent[np.arange(n) != t.root]] <= t.t[np.arange(n) != t.root])
        assert np.all(t.mut_from != t.mut_to) and np.all((t.mut_site >= 0) & (t.mut_site < sc.num_sites))
        parts, incl, seeds, root_part, ref = split_parts(sc, 3 if n > 80 else 1, 9)
        orc = OracleEngine(sc.num_sites)
        try:
            configure(orc, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
            G, A = orc.totals()
            assert np.isfinite(G) and np.isfinite(A), what
            orc.run_moves_per_part(300, threads=2)
            G, A = orc.totals()
            assert np.isfinite(G) and np.isfinite(A), what
        finally:
            orc.close()
    assert len({k[0] for k in seen}) >= 2 and any(k[2] for k in seen) and any(k[3] for k in seen)
