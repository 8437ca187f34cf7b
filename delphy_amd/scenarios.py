"""Synthetic workloads of BASELINE.json / SURVEY section 8(d): configs C1..C5 (and scaled-down variants
for tests).  Everything is seeded; nothing here reads /root/reference."""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .engine import FlatTree, PopModel, SynthParams, make_synthetic_emat

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
