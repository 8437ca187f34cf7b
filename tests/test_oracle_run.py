"""SURVEY 8 row a23 against an INDEPENDENT restatement: the product's host driver (delphy_amd/csrc/emat_run.cpp: stencils,
partition_tree, subtrees with frozen cut nodes and synthetic sub-root lists, reassemble, normalize_root) compared bit for bit
with oracle/orc_run.hpp, which follows the reference's run.cpp:87-265 and tree_partitioning.h:88-239 step by step (one walk to
the root per sub-root, where the product carries cut-point states down the tree of cut points).  CPU only: the local moves
between repartition and reassemble are made by the oracle's Subruns and handed to both drivers."""
import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import make_scenario, random_scenario
from helpers import configure
from oracle_ffi import OracleEngine, OracleRun

FIELDS = ("parent", "child0", "child1", "t", "t_min", "t_max", "mut_offset", "mut_site", "mut_from", "mut_to", "mut_t",
          "miss_offset", "miss_start", "miss_end", "mfs_offset", "mfs_site", "mfs_state")


def same_tree(a, b, what):
    assert a.root == b.root, what
    for f in FIELDS:
        x, y = getattr(a, f), getattr(b, f)
        assert x.shape == y.shape and np.array_equal(x, y), "%s: %s differs" % (what, f)


def cycles(sc, seed, num_parts, n_cycles, moves_per_part):
    run = d.EmatRun(None, sc.tree, sc.ref, seed)
    run.set_num_parts(num_parts)
    run.set_max_part_nodes(0)        # the reference's partition rule exactly (the run driver's default also caps the part size)
    orun = OracleRun(sc.tree, sc.ref, seed ^ 0xD1B54A32D192ED03, num_parts)     # emat_run_create: the partition stream is SplitMix64(seed ^ this constant)
    root_changes = 0
    try:
        for cycle in range(n_cycles):
            run.repartition(); orun.repartition()
            n, root_part = run.num_parts()
            assert (n, root_part) == orun.num_parts(), cycle
            (_, ref), (_, oref) = run.tree(), orun.tree()
            assert np.array_equal(ref, oref), "cycle %d: reference sequence after normalize_root" % cycle
            parts, incl, seeds = [], [], []
            for p in range(n):
                tp, ip, sp = run.part(p)
                to, orig, cut = orun.part(p)
                same_tree(tp, to, "cycle %d part %d" % (cycle, p))
                assert ip == (p == root_part) and orig[to.root] == cut
                parts.append(tp); incl.append(ip); seeds.append(sp)
            # the local moves, by the oracle's Subruns; both drivers get the same trees back
            eng = OracleEngine(sc.num_sites)
            configure(eng, sc, ref, parts, incl, seeds, root_part)
            eng.run_moves_per_part(moves_per_part, threads=4)
            for p in range(n):
                t = eng.part_download(p)
                run.part_put(p, t); orun.part_put(p, t)
                if p == root_part and t.mut_offset[t.root + 1] > t.mut_offset[t.root]:
                    root_changes += 1
            eng.close()
            run.reassemble(); orun.reassemble()
            (tree, ref), (otree, oref) = run.tree(), orun.tree()
            same_tree(tree, otree, "cycle %d reassembled" % cycle)
            assert np.array_equal(ref, oref)
            rc, msg = orun.check()
            assert rc == 0, msg
        return root_changes
    finally:
        run.close(); orun.close()


@pytest.mark.parametrize("name,kw,parts,moves", [
    ("C1", dict(num_tips=150, num_sites=3000, uncertain_tips=0.3), 3, 3000),
    ("C1", dict(num_tips=100, num_sites=30000), 1, 2000),
    ("C2", dict(num_tips=600, num_sites=6000), 12, 1500),
    ("C3", dict(num_tips=3000, num_sites=29903, uncertain_tips=0.1), 128, 400),     # >= 64 small parts: the size at which the device takes over partition_tree
])
def test_partition_subtrees_and_reassembled_trees_equal_the_restated_reference(name, kw, parts, moves):
    sc = make_scenario(name, **kw)
    cycles(sc, 17, parts, 5, moves)


def test_root_sequence_changes_are_folded_into_the_reference_identically():
    """A small tree with little signal: mutations cross the root within a few cycles, the root part comes back with
    root "mutations", and normalize_root re-references every missation's from_states at the next repartition."""
    sc = make_scenario("C1", num_tips=40, num_sites=2000, uncertain_tips=0.3)
    assert cycles(sc, 5, 2, 14, 4000) > 0, "the root sequence never changed: the case is not exercised"


def test_randomised_scenarios():
    rng = np.random.default_rng(9000)
    for case in range(12):
        sc, _, _, what = random_scenario(rng, 4 * case)          # (constant population: the moves in between use the scenario's plain HKY model)
        parts = int(rng.integers(1, max(2, sc.tree.num_nodes // 24)))
        try:
            cycles(sc, 100 + case, parts, 3, 300)
        except AssertionError as e:
            raise AssertionError("%s, %d parts: %s" % (what, parts, e))
