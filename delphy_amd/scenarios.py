"""Synthetic workloads of BASELINE.json / SURVEY section 8(d): configs C1..C5 (and scaled-down variants
for tests).  Everything is seeded; nothing here reads /root/reference."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .engine import FlatTree, PopModel, SynthParams, hky_q_matrix, make_synthetic_emat

# Stationary frequencies A, C, G, T.  Deliberately NOT symmetric: with pi_A = pi_T and pi_C = pi_G the HKY escape rates of A and T
# (and of C and G) coincide, so re-timing an A<->T mutation changes log G by exactly zero up to rounding noise, and the sign of
# that noise -- which differs between libm implementations -- decides whether the Metropolis step draws a uniform.  Such
# systematic ties would make move-for-move comparisons between two implementations meaningless.
PI = (0.31, 0.19, 0.21, 0.29)
KAPPA = 5.0


@dataclass
class Scenario:
    name: str
    tree: FlatTree
    ref: np.ndarray
    t_max_tip: float
    mu: float
    kappa: float
    pi: tuple
    pop: PopModel
    num_sites: int
    nu_l: Optional[np.ndarray] = None

    @property
    def num_tips(self) -> int:
        return (self.tree.num_nodes + 1) // 2

    def default_t_step(self) -> float:
        span = self.t_max_tip - float(self.tree.t[self.tree.root])
        return max(span / 400.0, 1.0 / 400.0)


def _skygrid(t_max_tip: float, span: float, n0: float, knots: int = 50, log_linear: bool = False) -> PopModel:
    x = np.array([t_max_tip - span * (knots - 1 - k) / (knots - 1) for k in range(knots)])
    gamma = np.array([math.log(n0) + 0.4 * math.sin(0.37 * k) for k in range(knots)])
    return PopModel.skygrid(x, gamma, log_linear)


def make_scenario(name: str, num_tips: Optional[int] = None, num_sites: Optional[int] = None, seed: Optional[int] = None,
                  uncertain_tips: float = 0.0, skygrid_log_linear: bool = False) -> Scenario:
    """name in {"C1","C2","C3","C4","C5"}; num_tips / num_sites override the config's size (tests use small ones)."""
    base = 20261001
    if name == "C1":
        p = SynthParams(num_tips=100, num_sites=30000, tip_span=365.0, pop_n0=365.0, pop_growth=0.0, mu=1e-3 / 365.0, gaps_per_tip=2, mean_gap_len=150.0, seed=base)
    elif name == "C2":
        p = SynthParams(num_tips=1610, num_sites=18959, tip_span=600.0, pop_n0=3 * 365.0, pop_growth=2.0 / 365.0, mu=1.2e-3 / 365.0, gaps_per_tip=1, mean_gap_len=57.0, seed=base + 1)
    elif name in ("C3", "C4", "C5"):
        tips = {"C3": 10000, "C4": 100000, "C5": 1000000}[name]
        span = {"C3": 365.0, "C4": 730.0, "C5": 730.0}[name]
        # exponential growth keeps the synthetic genealogy star-like, as pandemic-scale SARS-CoV-2 trees are; the scale of
        # N(t) is set so that the tree carries about one mutation per tip ("M ~ tips ... 3 tips"; C4: 95 748
        # mutations on 100 000 tips, mean branch 5.9 days at 30 substitutions per genome per year)
        p = SynthParams(num_tips=tips, num_sites=29903, tip_span=span, pop_n0=1500.0 * 365.0, pop_growth=6.0 / 365.0, mu=1e-3 / 365.0, gaps_per_tip=2, mean_gap_len=270.0,
                        seed=base + {"C3": 2, "C4": 3, "C5": 4}[name])
    else:
        raise ValueError(name)
    if num_tips is not None:
        p.num_tips = num_tips
    if num_sites is not None:
        p.num_sites = num_sites
        p.mean_gap_len = max(2.0, p.mean_gap_len * num_sites / {"C1": 30000, "C2": 18959}.get(name, 29903))
    if seed is not None:
        p.seed = seed
    p.pi = PI
    p.kappa = KAPPA
    if uncertain_tips > 0:
        p.frac_uncertain_tips = uncertain_tips
        p.tip_date_uncertainty = 5.0
    tree, ref, tmax = make_synthetic_emat(p)
    if name == "C1":
        pop = PopModel.exp(tmax, 365.0, 0.0, 0.0)            # constant population = Exp_pop_model with g = 0 (reference run.cpp:21)
    elif name == "C2":
        pop = PopModel.exp(tmax, 3 * 365.0, 2.0 / 365.0, 1.0)
    else:
        pop = _skygrid(tmax, p.tip_span * 1.2, 200.0 * 365.0, log_linear=skygrid_log_linear)   # N of the order of the generating model's over the sampled period
    return Scenario(name, tree, ref, tmax, p.mu, p.kappa, PI, pop, p.num_sites)


def random_scenario(rng, case, max_tips=320, density_cap=40):
    """A seeded random scenario for the sweeps: tree size, genome length, time span, mutation / gap density, tip-date
    uncertainty and population model all vary (the fixed configurations C1-C5 hold most of them constant)."""
    tips = int(rng.integers(12, max_tips))
    sites = int(rng.choice([60, 300, 2000, 9000]))
    span = float(rng.choice([30.0, 365.0, 1500.0]))
    mu = float(10 ** rng.uniform(-3.6, -2.0)) / 365.0 * (30000.0 / max(sites, 300)) ** 0.5
    par = SynthParams(num_tips=tips, num_sites=sites, tip_span=span, pop_n0=float(10 ** rng.uniform(1.5, 3.5)), pop_growth=float(rng.choice([0.0, 1.0, 5.0])) / 365.0,
                        mu=mu, gaps_per_tip=int(rng.integers(0, 5)), mean_gap_len=float(max(2.0, sites * 10 ** rng.uniform(-2.5, -0.8))), seed=int(rng.integers(1, 2**31)))
    par.pi, par.kappa = PI, KAPPA
    if rng.random() < 0.5:
        par.frac_uncertain_tips, par.tip_date_uncertainty = float(rng.uniform(0.05, 0.6)), float(rng.uniform(0.5, 20.0))
    tree, ref, tmax = make_synthetic_emat(par)
    while tree.mut_site.shape[0] > density_cap * tips:   # beyond that a move takes the device milliseconds
        mu /= 4.0; par.mu = mu
        tree, ref, tmax = make_synthetic_emat(par)
    kind = case % 4
    if kind == 0:
        pop = PopModel.exp(tmax, par.pop_n0, 0.0, 0.0)
    elif kind == 1:
        pop = PopModel.exp(tmax, par.pop_n0, float(rng.uniform(0.2, 4.0)) / 365.0, float(rng.choice([0.0, 1.0, par.pop_n0 / 50])))
    else:
        x = np.unique(np.append(np.sort(tmax - span * 1.3 * rng.uniform(0.0, 1.0, int(rng.integers(2, 40)))), tmax))
        pop = PopModel.skygrid(x, np.log(par.pop_n0) + rng.normal(0.0, 0.5, x.shape[0]), log_linear=(kind == 3))
    sc = Scenario("R%d" % case, tree, ref, tmax, mu, KAPPA, PI, pop, sites)
    nu_l = 0.2 + 1.8 * rng.random(sites) if case % 3 == 1 else None
    evo = None
    if case % 6 == 5:
        pi2 = rng.dirichlet([4.0, 4.0, 4.0, 4.0])
        evo = (np.array([mu, float(rng.uniform(0.3, 3.0)) * mu]), np.stack([np.asarray(PI, np.float64), pi2]),
               np.stack([hky_q_matrix(KAPPA, PI), hky_q_matrix(float(rng.uniform(1.0, 8.0)), pi2)]),
               (np.arange(sites) // max(1, sites // 7) % 2).astype(np.int32))
    what = "case %d (tips %d, sites %d, span %g, %d mutations, pop kind %d%s%s)" % (case, tips, sites, span, tree.mut_site.shape[0], kind,
                                                                                   ", site rates" if nu_l is not None else "", ", two partitions" if evo is not None else "")
    return sc, nu_l, evo, what
