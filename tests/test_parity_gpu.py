"""Parity of the HIP engine against the CPU oracle, through the C-ABI (include/emat_backend.h).

Bar (BASELINE.json north_star): mutation counts, site indices, states, interval endpoints, topology, move
decisions and RNG consumption bit-exact; log-posterior quantities and times within 1e-9 relative.
"""
import os

import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import assert_trees_match, configure, rel_close, run_parity, split_parts
from oracle_ffi import OracleEngine

pytestmark = pytest.mark.gpu


def test_derived_quantities_c1_small():
    sc = make_scenario("C1", num_tips=100, num_sites=3000)
    run_parity(sc, 4, 0)


def test_simple_moves_no_topology_small():
    sc = make_scenario("C1", num_tips=100, num_sites=3000, uncertain_tips=0.3)
    run_parity(sc, 4, 3000, topology=False, trace=3000)


def test_only_displacing_inner_nodes():
    sc = make_scenario("C1", num_tips=60, num_sites=1000)
    run_parity(sc, 3, 2000, only_displace=True, trace=2000)


def test_full_moves_c1_single_part_root():
    sc = make_scenario("C1", num_tips=100, num_sites=3000, uncertain_tips=0.3)
    st = run_parity(sc, 1, 4000, trace=4000)
    assert st["accepted"][3] + st["accepted"][4] > 0


def test_full_moves_c1_parts():
    sc = make_scenario("C1", num_tips=100, num_sites=30000)
    run_parity(sc, 4, 3000, trace=3000)


def test_full_moves_c2_exp_growth_parts():
    sc = make_scenario("C2", num_tips=400, num_sites=4000, uncertain_tips=0.2)
    run_parity(sc, 12, 2000, trace=2000)


def test_full_moves_skygrid_parts_hbm_resident():
    sc = make_scenario("C3", num_tips=600, num_sites=5000)
    run_parity(sc, 20, 1500, trace=1500, use_lds=False)


def test_full_moves_high_mutation_density():
    # many mutations per branch and dense missing data exercise the interval algebra and multi-hit branches
    sc = make_scenario("C1", num_tips=80, num_sites=400, seed=77)
    sc.mu = 3e-4
    import delphy_amd.engine as e
    tree, ref, tmax = e.make_synthetic_emat(e.SynthParams(num_tips=80, num_sites=400, mu=3e-4, gaps_per_tip=3, mean_gap_len=25, seed=77))
    sc.tree, sc.ref, sc.t_max_tip = tree, ref, tmax
    sc.pop = d.PopModel.exp(tmax, 365.0, 0.0, 0.0)
    run_parity(sc, 3, 3000, trace=3000)


def test_site_rate_heterogeneity():
    sc = make_scenario("C1", num_tips=100, num_sites=2000, uncertain_tips=0.2)
    nu = 0.25 + 1.5 * np.random.default_rng(5).random(2000)
    run_parity(sc, 4, 2000, trace=2000, nu_l=nu)


def test_site_rates_mostly_one_on_branches_of_more_than_32_mutations():
    """A branch reform gathers log(mu nu q) from the LDS table for sites of rate 1 and takes a logarithm for the others, keeping
    the 'needs a logarithm' flags of the first 32 mutations in a bit mask (reform_factors): a branch of more than 32 mutations
    whose first 32 hold a site with nu != 1 and whose later ones hold none used to shift the mask by >= 32."""
    import delphy_amd.engine as e
    par = e.SynthParams(num_tips=14, num_sites=6000, tip_span=200.0, pop_n0=2000.0, mu=6e-6, gaps_per_tip=1, mean_gap_len=40, seed=4242)
    par.pi, par.kappa = (0.31, 0.19, 0.21, 0.29), 5.0
    tree, ref, tmax = e.make_synthetic_emat(par)
    sc = make_scenario("C1", num_tips=14, num_sites=6000)
    sc.tree, sc.ref, sc.t_max_tip, sc.mu = tree, ref, tmax, par.mu
    sc.pop = d.PopModel.exp(tmax, 2000.0, 0.0, 0.0)
    nu = np.ones(6000)
    shaped = 0
    for n in range(tree.num_nodes):
        sites = tree.mut_site[tree.mut_offset[n]:tree.mut_offset[n + 1]]
        if sites.shape[0] > 32 and n != tree.root and sites[0] not in sites[32:] and np.all(nu[sites[32:]] == 1.0):
            nu[sites[0]] = 1.7; shaped += 1
    assert shaped >= 2, "the scenario must hold branches of more than 32 mutations (has %d)" % shaped
    run_parity(sc, 1, 3000, trace=3000, nu_l=nu, topology=False)
    run_parity(sc, 2, 1500, trace=1500, nu_l=nu)


def _variant_counts(sc, lds_max, monkeypatch):
    import ctypes as C
    if lds_max is None:
        monkeypatch.delenv("EMAT_LDS_MAX", raising=False)
    else:
        monkeypatch.setenv("EMAT_LDS_MAX", str(lds_max))
    parts, incl, seeds, root_part, ref = split_parts(sc, 4, 23)
    b = d.EmatBackend(sc.num_sites, trace_moves=200)
    configure(b, sc, ref, parts, incl, seeds, root_part, None)
    lib = d.load_library()
    lib.emat_debug_variant_counts.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    out = (C.c_int32 * 3)()
    assert lib.emat_debug_variant_counts(b.handle, out) == 0
    b.close()
    return list(out)


def test_code_variants_whole_prefix_hbm(monkeypatch):
    """The engine compiles its device code three times (whole slab in LDS / fixed-size prefix in LDS / HBM only) and
    picks per part; capping the staging area forces each variant in turn, and each must match the oracle."""
    sc = make_scenario("C1", num_tips=120, num_sites=4000)
    seen = {}
    for cap in (None, 32768, 16384, 12288, 10240, 8192, 6144, 4096, 2048, 512):
        counts = _variant_counts(sc, cap, monkeypatch)   # same decision as the kernel's, evaluated on the host
        for v in range(3):
            if counts[v] > 0 and v not in seen:
                seen[v] = cap
    assert sorted(seen) == [0, 1, 2], "staging caps tried do not exercise every variant: %s" % seen
    for v, cap in seen.items():
        if cap is None:
            monkeypatch.delenv("EMAT_LDS_MAX", raising=False)
        else:
            monkeypatch.setenv("EMAT_LDS_MAX", str(cap))
        run_parity(sc, 4, 3000, seed=23)


def test_two_site_partitions():
    """Two site partitions with their own mu / HKY tables (the reference's mpox set-up, run.cpp:400-435): per-partition
    tables are staged in LDS, the partition map is read per site."""
    sc = make_scenario("C1", num_tips=90, num_sites=6000)
    pi2 = np.array([0.1, 0.4, 0.3, 0.2])
    mu = np.array([sc.mu, 3.0 * sc.mu])
    pi = np.stack([np.asarray(sc.pi, np.float64), pi2])
    q = np.stack([d.hky_q_matrix(sc.kappa, sc.pi), d.hky_q_matrix(2.0, pi2)])
    pfs = (np.arange(sc.num_sites) // 500 % 2).astype(np.int32)
    run_parity(sc, 3, 4000, seed=31, evo=(mu, pi, q, pfs))


def test_skygrid_log_linear():
    sc = make_scenario("C3", num_tips=150, num_sites=5000, skygrid_log_linear=True)
    run_parity(sc, 4, 3000, seed=37)


def test_run_local_moves_remainder_goes_to_part_zero():
    """Run::run_local_moves (run.cpp:682-693): count / parts moves on every part, the remainder on part 0."""
    sc = make_scenario("C1", num_tips=80, num_sites=3000)
    run_parity(sc, 3, 1, seed=41, total_moves=3 * 1500 + 2, trace=0)


def test_run_moves_even_spreads_the_remainder_one_move_per_part():
    """emat_run_moves_even: the same number of moves on every part and one more on the first parts; each part's chain is the
    oracle's chain of that length."""
    sc = make_scenario("C1", num_tips=120, num_sites=3000)
    parts, incl, seeds, root_part, ref = split_parts(sc, 5, 43)
    gpu = d.EmatBackend(sc.num_sites)
    orc = OracleEngine(sc.num_sites)
    try:
        configure(gpu, sc, ref, parts, incl, seeds, root_part)
        configure(orc, sc, ref, parts, incl, seeds, root_part)
        gpu.run_moves_even(700, 3); gpu.synchronize()
        counts = [700 + (1 if p < 3 else 0) for p in range(len(parts))]
        orc.run_moves_counts(counts)
        for p in range(len(parts)):
            assert gpu.part_stats(p)["moves_done"] == counts[p]
            assert_trees_match(gpu.part_download(p), orc.part_download(p), 1e-9, "part %d" % p)
    finally:
        gpu.close(); orc.close()


def _stats_parity(sc, num_parts, moves, seed, evo=None, P=1):
    parts, incl, seeds, root_part, ref = split_parts(sc, num_parts, seed)
    gpu = d.EmatBackend(sc.num_sites)
    orc = OracleEngine(sc.num_sites)
    try:
        configure(gpu, sc, ref, parts, incl, seeds, root_part, None, evo=evo)
        configure(orc, sc, ref, parts, incl, seeds, root_part, None, evo=evo)
        for rounds in range(2):
            Tg, Mg, ng = gpu.global_stats(P)
            To, Mo, no = orc.global_stats(P)
            assert ng == no and np.array_equal(Mg, Mo), "mutation counts differ: %s vs %s" % (Mg.tolist(), Mo.tolist())
            assert rel_close(Tg, To, 1e-9), "Ttwiddle differs: %s vs %s" % (Tg.tolist(), To.tolist())
            assert Mg.sum() == ng and ng > 0 and np.all(Tg > 0)
            gpu.run_moves_per_part(moves); gpu.synchronize(); orc.run_moves_per_part(moves, threads=4)
    finally:
        gpu.close(); orc.close()


def test_global_move_statistics():
    """emat_get_global_stats (calc_Ttwiddle_beta_a, calc_num_muts_beta_ab, calc_num_muts on the device) against the oracle,
    before and after moves, with one and with two site partitions and per-site rates."""
    sc = make_scenario("C3", num_tips=400, num_sites=8000)
    _stats_parity(sc, 6, 1500, seed=43)
    pi2 = np.array([0.1, 0.4, 0.3, 0.2])
    evo = (np.array([sc.mu, 2.0 * sc.mu]), np.stack([np.asarray(sc.pi, np.float64), pi2]), np.stack([d.hky_q_matrix(sc.kappa, sc.pi), d.hky_q_matrix(2.0, pi2)]),
           (np.arange(sc.num_sites) // 700 % 2).astype(np.int32))
    _stats_parity(sc, 5, 1000, seed=47, evo=evo, P=2)


def test_randomised_global_move_statistics():
    """The reductions of SURVEY 8(f).1 over the random scenarios of the sweeps (site rates, two partitions, every population
    model, 1 to 13 parts), before and after moves."""
    from helpers import random_scenario
    rng = np.random.default_rng(int(os.environ.get("EMAT_FUZZ_SEED", "20261004")))
    for case in range(int(os.environ.get("EMAT_FUZZ_CASES", "12"))):
        sc, nu_l, evo, what = random_scenario(rng, case)
        nparts = int(min(max(1, sc.tree.num_nodes // 24), rng.integers(1, 14)))
        parts, incl, seeds, root_part, ref = split_parts(sc, nparts, int(rng.integers(1, 10**6)))
        gpu = d.EmatBackend(sc.num_sites); orc = OracleEngine(sc.num_sites)
        P = 2 if evo is not None else 1
        try:
            configure(gpu, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
            configure(orc, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
            scale = np.ones(2)
            for rounds in range(2):
                Tg, Mg, ng = gpu.global_stats(P)
                To, Mo, no = orc.global_stats(P)
                assert ng == no and np.array_equal(Mg, Mo), (what, Mg.tolist(), Mo.tolist())
                assert rel_close(Tg, To, 1e-9), (what, Tg.tolist(), To.tolist())
                # The totals are maintained INCREMENTALLY by both sides (as by the reference): a total that starts at -2.3e16 -- an exponential-growth
                # model whose population is 1e-12 in the grid's oldest cells -- and shrinks to -1e9 as the root moves keeps the ABSOLUTE rounding error of
                # its start, one unit in the last place of 2.3e16 = 4.0 (EMAT_FUZZ_SEED=6102, case 29, found in round 6; round 5's code gives the same
                # numbers; a recomputation agrees to the last digit).  So the tolerance is relative to the largest magnitude the total has had.
                tg, to = np.array(gpu.totals()), np.array(orc.totals())
                scale = np.maximum(scale, np.abs(to))
                assert np.all(np.abs(tg - to) <= 1e-9 * scale), (what, tg.tolist(), to.tolist())
                gpu.run_moves_per_part(800); gpu.synchronize(); orc.run_moves_per_part(800, threads=4)
            gpu.recalc_derived(); orc.recalc_derived()           # ... and recomputed from scratch the totals agree relative to what they ARE
            assert rel_close(np.array(gpu.totals()), np.array(orc.totals()), 1e-9), what
        except d.EmatError as ex:
            raise AssertionError("%s: %s" % (what, ex)) from ex
        finally:
            gpu.close(); orc.close()


def test_a_re_materialisation_in_mid_pass_keeps_what_the_moves_maintain():
    """What the moves maintain incrementally -- lambda_i, missing-site counts, log_G, the partial prior -- is recomputed when a Subrun is made
    and never again (subrun.cpp:17-26).  Until round 6 the engine recomputed it whenever a pass was interrupted to give a part more room: the
    same numbers up to rounding, and a chain that can tell.  The case that showed it (EMAT_FUZZ_SEED=6202, case 53 of the global-statistics
    sweep): the root part outgrows its coalescent grid in the second pass (401 -> 2 027 cells), every part is re-encoded, and in another
    part a node under which every site is missing below one child or the other -- d log G / dt = lambda - lambda exactly -- took the uniform
    branch of the bounded exponential in the reference's arithmetic and the other branch with a recomputed lambda_i two units in the last
    place off: one move skipped its acceptance draw and the chains parted.  Now the maintained values are carried over (PartHost, emat_backend.hip):
    two passes, every part's trace equal move for move, lambda_i equal BIT for bit."""
    from helpers import random_scenario
    rng = np.random.default_rng(6202)
    for case in range(54):
        sc, nu_l, evo, what = random_scenario(rng, case)
        nparts = int(min(max(1, sc.tree.num_nodes // 24), rng.integers(1, 14)))
        split_seed = int(rng.integers(1, 10**6))
    parts, incl, seeds, root_part, ref = split_parts(sc, nparts, split_seed)
    T = 1600
    gpu = d.EmatBackend(sc.num_sites, trace_moves=T); orc = OracleEngine(sc.num_sites, trace_moves=T)
    try:
        configure(gpu, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
        configure(orc, sc, ref, parts, incl, seeds, root_part, None, nu_l=nu_l, evo=evo)
        cells_before = len(gpu.part_coalescent(root_part)["k_bar_p"])
        for _ in range(2):
            gpu.run_moves_per_part(800); gpu.synchronize(); orc.run_moves_per_part(800, threads=4)
        assert len(gpu.part_coalescent(root_part)["k_bar_p"]) > cells_before + max(512, cells_before), "the root part's grid did not outgrow its slab: the case no longer interrupts a pass"
        for p in range(len(parts)):
            tg, to = gpu.part_trace(p, T), orc.part_trace(p, T)
            assert tg.shape == to.shape
            same = (tg[:, :3] == to[:, :3]).all(axis=1)
            assert same.all(), "part %d: first differing move %d: gpu %s oracle %s" % (p, int(np.argmin(same)), tg[int(np.argmin(same))], to[int(np.argmin(same))])
            n = parts[p].num_nodes
            lg, ng, _, _ = gpu.part_derived(p, n); lo, no, _, _ = orc.part_derived(p, n)
            assert np.array_equal(np.asarray(lg).view(np.uint64), np.asarray(lo).view(np.uint64)), "part %d: lambda_i differs in the last place" % p
            assert np.array_equal(ng, no)
    finally:
        gpu.close(); orc.close()


def test_parts_that_run_out_of_slab_space_are_regrown_and_finish(monkeypatch):
    """A part whose list heap is too small stops BEFORE a move (status 101, state intact); the engine must notice at the
    next synchronisation, give it more room and run the rest of its moves -- the caller sees a complete pass whose
    results still match the oracle move for move.  Forced here by uploading with no heap slack at all."""
    import delphy_amd.engine as e
    monkeypatch.setenv("EMAT_SLACK", "1.0")
    monkeypatch.setenv("EMAT_HEAP_PER_NODE", "0")
    sc = make_scenario("C1", num_tips=80, num_sites=400, seed=77)
    sc.mu = 3e-4
    tree, ref, tmax = e.make_synthetic_emat(e.SynthParams(num_tips=80, num_sites=400, mu=3e-4, gaps_per_tip=3, mean_gap_len=25, seed=77))
    sc.tree, sc.ref, sc.t_max_tip = tree, ref, tmax
    sc.pop = d.PopModel.exp(tmax, 365.0, 0.0, 0.0)
    run_parity(sc, 2, 4000, trace=4000)   # the trace survives the re-materialisation too


@pytest.mark.parametrize("use_lds", [True, False])
def test_containers_overflowing_inside_a_move_are_answered_with_more_room(monkeypatch, use_lds):
    """With no slack at all and 1 KB of reserve, single moves need more list heap than the check before the move holds back
    (status 102, INSIDE a move).  A staged leg never touched the HBM copy of its slab; a leg that runs on the HBM slab
    itself (here: staging switched off) took a copy of its starting state first.  Either way the part's starting state is
    what the host finds, it doubles the room and runs the moves again: a complete pass, still the oracle's chain."""
    import delphy_amd.engine as e
    monkeypatch.setenv("EMAT_SLACK", "1.0")
    monkeypatch.setenv("EMAT_HEAP_PER_NODE", "0")
    par = e.SynthParams(num_tips=60, num_sites=9000, tip_span=30.0, pop_n0=300.0, pop_growth=0.0, mu=6e-3 / 365.0, gaps_per_tip=1, mean_gap_len=60.0, seed=99)
    from delphy_amd.scenarios import Scenario, KAPPA, PI
    par.pi, par.kappa = PI, KAPPA
    tree, ref, tmax = e.make_synthetic_emat(par)
    sc = Scenario("tight", tree, ref, tmax, par.mu, KAPPA, PI, d.PopModel.exp(tmax, 300.0, 1.0 / 365.0, 1.0), 9000)
    run_parity(sc, 3, 3000, seed=5, trace=3000, use_lds=use_lds)


@pytest.mark.parametrize("use_lds", [True, False])
def test_root_grid_outgrowing_its_slab_room_stops_before_the_move_and_is_regrown(use_lds):
    """The reference's coalescent vectors grow without bound when the root moves into the past (very_scalable_coalescent.cpp:
    259-299); a slab holds room for a fixed number of cells.  With almost no signal in the data (60 sites, a handful of
    mutations) one root move can displace the root by the whole span of the tree -- hundreds of cells at this cell width.
    The moves that can do that ask before they change anything, the part stops with the move undone and its RNG rewound, the
    host re-materialises it with more cells and the move runs again: the chain is still the oracle's, move for move, for the
    staged part and for the part run from HBM (which has no earlier copy to fall back on)."""
    import delphy_amd.engine as e
    from delphy_amd.scenarios import Scenario, KAPPA, PI
    par = e.SynthParams(num_tips=120, num_sites=60, tip_span=30.0, pop_n0=400.0, pop_growth=0.0, mu=2e-5, gaps_per_tip=1, mean_gap_len=4.0, seed=3)
    par.pi, par.kappa = PI, KAPPA
    tree, ref, tmax = e.make_synthetic_emat(par)
    sc = Scenario("deep-root", tree, ref, tmax, par.mu, KAPPA, PI, d.PopModel.exp(tmax, 400.0, 0.0, 0.0), 60)
    t_step = sc.default_t_step() * 0.25
    for nparts in (1, 5):
        parts, incl, seeds, root_part, ref2 = split_parts(sc, nparts, 7)
        orc = OracleEngine(sc.num_sites)
        configure(orc, sc, ref2, parts, incl, seeds, root_part, t_step)
        cells0 = orc.part_coalescent(root_part)["k_bar_p"].shape[0]
        orc.run_moves_per_part(4000, threads=1)
        cells1 = orc.part_coalescent(root_part)["k_bar_p"].shape[0]
        orc.close()
        assert cells1 > cells0 + max(512, cells0), (cells0, cells1)   # past the room a freshly cut slab has: the device must have regrown it
        run_parity(sc, nparts, 4000, seed=7, trace=4000, use_lds=use_lds, t_step=t_step)


def test_randomised_scenarios_move_for_move():
    """A seeded sweep over what the fixed scenarios hold constant (delphy_amd.scenarios.random_scenario): tree size, genome length,
    mutation density, gap density, tip-date uncertainty, the population model (constant / exponential with a floor /
    skygrid staircase / skygrid log-linear with an irregular knot spacing), site-rate heterogeneity, one or two site
    partitions -- and here the coalescent cell width, the number of parts and LDS staging on / off.  Every case is compared
    move for move (trace) and quantity for quantity like the fixed ones.  (This sweep found the two capacity holes that
    DESIGN.md section 4 describes: a root grid outgrowing its slab and containers overflowing inside a move.)"""
    from helpers import random_scenario
    rng = np.random.default_rng(int(os.environ.get("EMAT_FUZZ_SEED", "20261002")))   # EMAT_FUZZ_SEED / EMAT_FUZZ_CASES: longer hunts by hand
    kinds = 0
    for case in range(int(os.environ.get("EMAT_FUZZ_CASES", "20"))):
        sc, nu_l, evo, what = random_scenario(rng, case)
        nparts = int(min(max(1, (sc.tree.num_nodes + 1) // 24), rng.integers(1, 14)))
        t_step = sc.default_t_step() * float(rng.choice([0.25, 1.0, 4.0]))
        seed = int(rng.integers(1, 10**6))
        try:
            run_parity(sc, nparts, 1500, seed=seed, trace=1500, use_lds=bool(case % 5), t_step=t_step, nu_l=nu_l, evo=evo)
        except Exception as ex:
            raise AssertionError("%s, parts %d, t_step %g, seed %d: %s" % (what, nparts, t_step, seed, ex)) from ex
        kinds |= 1 << (case % 4)
    assert kinds == 15


def test_device_gamma_q_against_scipy_golden_vectors():
    """The device's own gamma_q / gamma_q_inv (they replace Boost's, whose source is not in the reference tree) swept over
    the same scipy vectors and tolerances the oracle's are pinned to (tests/test_oracle_pinning.py; the reference compares
    at 1e-12 absolute, safe_gamma_math_tests.cpp:35-56)."""
    import json, os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gamma_q.json")))
    b = d.EmatBackend(100)
    try:
        rows = g["q"]
        got = b.debug_gamma(0, [r["a"] for r in rows], [r["x"] for r in rows])
        for r, v in zip(rows, got):
            assert abs(v - r["q"]) <= 1e-12 + 1e-10 * r["q"], (r, v)
        rows = g["q_inv"]
        a = np.array([r["a"] for r in rows])
        got = b.debug_gamma(1, a, [r["q"] for r in rows])
        back = b.debug_gamma(0, a, got)
        for r, v, q in zip(rows, got, back):
            assert abs(v - r["x"]) <= 1e-9 * max(1.0, r["x"]), (r, v)
            assert abs(q - r["q"]) <= 1e-11 * r["q"] + 1e-14, (r, v, q)
        # limits the reference tests (safe_gamma_math_tests.cpp:65-106): Q(a, 0) = 1, Q(a, inf) = 0, inverse of 1 and 0
        assert list(b.debug_gamma(0, [0.5, 3.0], [0.0, np.inf])) == [1.0, 0.0]
        assert list(b.debug_gamma(1, [0.5, 3.0], [1.0, 0.0])) == [0.0, np.inf]
    finally:
        b.close()


def test_device_gamma_q_on_the_reference_tests_own_points():
    """Every point and assertion of the reference's own test of its incomplete-gamma wrappers (safe_gamma_math_tests.cpp:34-262,
    kept as data in tests/golden/gamma_reference_cases.json), on the device's routines: Q down to 1e-300 and up to 1 - 1e-15,
    a up to 1000, x up to 1e5 -- the same checker the oracle passes in tests/test_oracle_pinning.py."""
    from test_oracle_pinning import check_gamma_reference_cases
    b = d.EmatBackend(100)
    try:
        check_gamma_reference_cases(lambda a, x: list(b.debug_gamma(0, a, x)), lambda a, q: list(b.debug_gamma(1, a, q)))
    finally:
        b.close()


def test_whole_tree_coalescent_prior_and_per_site_mutation_counts_from_the_parts():
    """Scalable_coalescent_prior::calc_log_prior (the whole-tree grid prior Run keeps beside the
    per-part ones, run.cpp:455-465) and calc_num_muts_l, computed on the device from the PARTS, against the oracle's
    restatement on the WHOLE tree -- before and after moves, for the three population models."""
    for name, kw, nparts in (("C1", dict(num_tips=150, num_sites=3000), 5), ("C2", dict(num_tips=400, num_sites=4000, uncertain_tips=0.2), 12),
                             ("C3", dict(num_tips=700, num_sites=5000), 24)):
        sc = make_scenario(name, **kw)
        parts, incl, seeds, root_part, ref = split_parts(sc, nparts, 53)
        gpu = d.EmatBackend(sc.num_sites)
        configure(gpu, sc, ref, parts, incl, seeds, root_part)
        try:
            for rounds in range(2):
                # reassemble what the device holds into one tree, through the product's own host driver
                run = d.EmatRun(None, sc.tree, sc.ref, 53)
                run.set_num_parts(nparts); run.repartition()
                for p in range(len(parts)):
                    run.part_put(p, gpu.part_download(p))
                run.reassemble()
                whole, ref2 = run.tree(); run.close()
                tips = whole.child0 == -1
                t_ref = float(np.max(whole.t[tips]))                     # calc_max_tip_time
                t_step = sc.default_t_step()
                orc = OracleEngine(sc.num_sites)
                sc2 = make_scenario(name, **kw); sc2.tree, sc2.ref = whole, ref2
                configure(orc, sc2, ref2, [whole], [True], [1], 0)
                want = orc.scalable_log_prior(0, t_ref, t_step)
                got = gpu.scalable_coalescent_log_prior(t_ref, t_step)
                assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), (name, rounds, got, want)
                # the staged (multi-GPU) form gives the same number
                _, _, first = gpu.scalable_coalescent_partial(t_ref, t_step, 0, 0)
                kb, logs, _ = gpu.scalable_coalescent_partial(t_ref, t_step, first - 3, 3 - first)
                assert gpu.scalable_coalescent_log_prior_from_grid(t_ref, t_step, first - 3, kb, logs) == pytest.approx(got, rel=1e-13)
                assert np.array_equal(gpu.num_muts_l(), orc.num_muts_l())
                assert gpu.num_muts_l().sum() == gpu.global_stats(1)[2]
                orc.close()
                gpu.run_moves_per_part(1500); gpu.synchronize()
        finally:
            gpu.close()


def test_global_stats_wait_for_regrown_parts(monkeypatch):
    """emat_get_global_stats right after emat_run_* (no emat_synchronize in between) with parts that run out of heap space
    mid-pass: the statistics must describe chains that finished ALL their moves (the getter finishes the pass first)."""
    import delphy_amd.engine as e
    monkeypatch.setenv("EMAT_SLACK", "1.0")
    monkeypatch.setenv("EMAT_HEAP_PER_NODE", "0")
    sc = make_scenario("C1", num_tips=80, num_sites=400, seed=77)
    sc.mu = 3e-4
    tree, ref, tmax = e.make_synthetic_emat(e.SynthParams(num_tips=80, num_sites=400, mu=3e-4, gaps_per_tip=3, mean_gap_len=25, seed=77))
    sc.tree, sc.ref, sc.t_max_tip = tree, ref, tmax
    sc.pop = d.PopModel.exp(tmax, 365.0, 0.0, 0.0)
    parts, incl, seeds, root_part, ref = split_parts(sc, 2, 11)
    gpu = d.EmatBackend(sc.num_sites); orc = OracleEngine(sc.num_sites)
    try:
        configure(gpu, sc, ref, parts, incl, seeds, root_part); configure(orc, sc, ref, parts, incl, seeds, root_part)
        gpu.run_moves_per_part(4000)
        Tg, Mg, ng = gpu.global_stats(1)            # no synchronize() before it
        orc.run_moves_per_part(4000, threads=2)
        To, Mo, no = orc.global_stats(1)
        assert ng == no and np.array_equal(Mg, Mo) and rel_close(Tg, To, 1e-9)
        assert all(gpu.part_stats(p)["moves_done"] == 4000 for p in range(len(parts)))
    finally:
        gpu.close(); orc.close()


def test_mid_size_parts_with_lists_outgrowing_the_lds_heap_room():
    """Parts of 60-100 nodes on a 29 903-site genome are staged in LDS with a 1 KB heap reserve; a move whose lists outgrow
    it mid-way drops the staged state and the leg is re-run from the (untouched) HBM copy.  Whether or not that happens in
    these 64 x 4000 moves, the result must be the oracle's move for move."""
    sc = make_scenario("C3", num_tips=2000, num_sites=29903, uncertain_tips=0.2)
    run_parity(sc, 64, 4000, seed=4242, trace=0)


def test_Ttwiddle_l_of_the_whole_tree_from_the_parts():
    """calc_Ttwiddle_l (phylo_tree_calc.cpp:176-222) needs the branch length BELOW every mutation, which
    crosses part boundaries: the run driver feeds the engine the length hanging below every boundary tip.  Against the
    oracle's restatement on the reassembled whole tree, before and after moves, with site-rate heterogeneity and two
    kinds of population model; the statistic does not depend on how the tree is cut."""
    for name, kw, nparts in (("C1", dict(num_tips=150, num_sites=3000), 5), ("C3", dict(num_tips=900, num_sites=6000, uncertain_tips=0.2), 40)):
        sc = make_scenario(name, **kw)
        b = d.EmatBackend(sc.num_sites)
        run = d.EmatRun(b, sc.tree, sc.ref, 61)
        run.set_num_parts(nparts); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
        try:
            for cyc in range(3):
                run.repartition()
                n, _ = run.num_parts()
                if cyc > 0:
                    run.run_moves(n * 800)
                got = run.Ttwiddle_l()
                run.reassemble()
                whole, ref = run.tree()
                orc = OracleEngine(sc.num_sites)
                sc2 = make_scenario(name, **kw); sc2.tree, sc2.ref = whole, ref
                configure(orc, sc2, ref, [whole], [True], [1], 0)
                want = orc.Ttwiddle_l(0)
                orc.close()
                assert rel_close(got, want, 1e-9), (name, cyc, float(np.max(np.abs(got - want))))
                assert np.all(want >= 0) and want.max() > 0
        finally:
            run.close(); b.close()


def test_a_pass_in_tickets_is_the_same_chain(monkeypatch):
    """EMAT_CHUNKS: every part's pass cut into 1, 3 and 7 tickets that hand the slab over through HBM -- move for move the
    oracle's chain each time (the trace, the tree, the counters and the RNG position do not know about the cuts)."""
    sc = make_scenario("C3", num_tips=1200, num_sites=8000, uncertain_tips=0.2)
    for chunks in ("1", "3", "7"):
        monkeypatch.setenv("EMAT_CHUNKS", chunks)
        run_parity(sc, 48, 1001, seed=53, trace=1001)


def test_check_derived_is_the_references_paranoid_check_on_the_device():
    """emat_check_derived (Subrun::check_derived_quantities, subrun.cpp:28-56; --v0-paranoid): after thousands of moves the
    incrementally maintained lambda_i / log_G / prior / missing-site counts of every part agree with a from-scratch
    recomputation within the reference's tolerances -- found without the oracle and without touching the state; with the
    tolerances shrunk far below round-off the same call fails and names the part; a run in paranoid mode cycles cleanly."""
    sc = make_scenario("C3", num_tips=2000, num_sites=29903, uncertain_tips=0.1)
    parts, incl, seeds, root_part, ref = split_parts(sc, 64, 5)
    b = d.EmatBackend(sc.num_sites)
    configure(b, sc, ref, parts, incl, seeds, root_part)
    b.run_moves_per_part(3000); b.synchronize()
    before = [b.part_derived(p, parts[p].num_nodes) for p in (0, root_part)]
    part, dev4 = b.check_derived(1.0)
    assert 0 <= part < len(parts) and dev4[3] == 0 and dev4[0] < 1e-8 and dev4[1] < 1e-6 and dev4[2] < 1e-5
    assert max(dev4[:3]) > 0.0                                   # round-off is there: the comparison is not of a value with itself
    after = [b.part_derived(p, parts[p].num_nodes) for p in (0, root_part)]
    for x, y in zip(before, after):                              # the check leaves the incremental values alone
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1]) and x[2] == y[2] and x[3] == y[3]
    with pytest.raises(d.EmatError, match="emat_check_derived: part"):
        b.check_derived(1e-12)
    b.close()
    b = d.EmatBackend(sc.num_sites)
    run = d.EmatRun(b, sc.tree, sc.ref, 3)
    run.set_num_parts(64); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
    run.set_device_tree(True); run.set_paranoid(True)
    run.do_mcmc_steps(3 * 64 * 800, 64 * 800)
    run.close(); b.close()


def test_recalc_derived_right_after_a_pass_waits_for_the_side_classes():
    """emat_run_* -> emat_recalc_derived -> emat_run_* -> emat_check_derived with no getter or synchronize in between.  The side
    classes of a pass (the root part, the giants) run on streams of their own that the engine's stream does not wait for at
    launch time; k_recalc_derived rewrites lambda / missing counts / totals of EVERY part, so it has to join them first and
    the next pass's side launches have to follow it.  Same calls with a synchronize after each give the result to compare with
    (the recomputation is deterministic, so the two engines must agree bit for bit)."""
    sc = make_scenario("C3", num_tips=1500, num_sites=12000, uncertain_tips=0.1)
    parts, incl, seeds, root_part, ref = split_parts(sc, 40, 9)
    a = d.EmatBackend(sc.num_sites); b = d.EmatBackend(sc.num_sites)
    try:
        configure(a, sc, ref, parts, incl, seeds, root_part); configure(b, sc, ref, parts, incl, seeds, root_part)
        for _ in range(3):
            a.run_moves_per_part(2500); a.recalc_derived(); a.run_moves_per_part(500)
            b.run_moves_per_part(2500); b.synchronize(); b.recalc_derived(); b.synchronize(); b.run_moves_per_part(500); b.synchronize()
        part, dev4 = a.check_derived(1.0)
        assert dev4[3] == 0
        assert a.totals() == b.totals()
        for p in (0, root_part, len(parts) - 1):
            assert_trees_match(a.part_download(p), b.part_download(p), 0.0, "part %d" % p)
            la, na, Ga, Aa = a.part_derived(p, parts[p].num_nodes); lb, nb, Gb, Ab = b.part_derived(p, parts[p].num_nodes)
            assert np.array_equal(la, lb) and np.array_equal(na, nb) and Ga == Gb and Aa == Ab
    finally:
        a.close(); b.close()


@pytest.mark.parametrize("release", ["auto", "full"])
def test_tickets_handed_over_across_xcds_are_the_same_chain(monkeypatch, release):
    """The default launch puts the tickets of a part on one XCD (ticket stride a multiple of the number of XCDs, which the engine
    probes) and hands the part over through write-through stores, without writing the XCD's L2 back.  That path is ONLY valid
    behind one L2: with EMAT_TICKET_XCD_SPREAD=1 (odd stride: every hand-over crosses from one XCD's L2 to another's) this test
    caught it reading a stale slab about once in twenty passes in round 4.  The engine therefore takes the plain agent-scope
    release whenever the stride does not keep a part on one XCD ("auto": what it decides by itself here; "full": EMAT_TICKET_RELEASE
    forces it everywhere), and every ticket checks that a cheaply handed-over part comes from its own XCD.  5 and 8 tickets per part
    and pass, on parts that are staged whole, by their prefix, and not at all (EMAT_LDS_MAX); repeated, since what it guards
    against is a race.  Every chain must be the oracle's, move for move."""
    monkeypatch.setenv("EMAT_TICKET_XCD_SPREAD", "1")
    if release == "full":
        monkeypatch.setenv("EMAT_TICKET_RELEASE", "full")
    sc = make_scenario("C3", num_tips=1500, num_sites=8000, uncertain_tips=0.2)
    for rep in range(3):
        for chunks, lds_max in (("5", None), ("8", "6144")):
            monkeypatch.setenv("EMAT_CHUNKS", chunks)
            if lds_max is None:
                monkeypatch.delenv("EMAT_LDS_MAX", raising=False)
            else:
                monkeypatch.setenv("EMAT_LDS_MAX", lds_max)
            run_parity(sc, 61, 1203, seed=59 + rep, trace=1203)     # 61 parts: an odd stride whatever the class sizes
