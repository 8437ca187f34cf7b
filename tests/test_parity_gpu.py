"""Parity of the HIP engine against the CPU oracle, through the C-ABI (include/emat_backend.h).

Bar (BASELINE.json north_star): mutation counts, site indices, states, interval endpoints, topology, move
decisions and RNG consumption bit-exact; log-posterior quantities and times within 1e-9 relative.
"""
import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import run_parity

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
