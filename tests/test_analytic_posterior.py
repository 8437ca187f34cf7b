"""A check that does not go through anyone's reading of the reference's code: a model small enough for its posterior to be
known in closed form, sampled by the moves + the coalescent parts, compared with the formula.

Two tips with identical sequences and no missing data under HKY and a constant population N; tip 0 is dated exactly at
t = 0, tip 1 somewhere in [0, B].  With lambda = sum over sites of mu nu_l q(state) (Delphy's lambda_i, the total rate of
leaving the sequence):

  * likelihood of "no mutation on a branch of length T" is exp(-lambda T) (a second-order term, histories with two or more
    mutations that cancel, is ~1e-3 here);
  * while only tip 1's lineage exists (between 0 and t1) nothing can coalesce; below 0 two lineages coalesce at rate 1/N.

So  t1 ~ Exponential(lambda) truncated to [0, B],  independent of  h = -t_root ~ Exponential(1/N + 2 lambda):

  E[t1] = 1/lambda - B / (exp(lambda B) - 1),      E[h] = sd[h] = 1 / (1/N + 2 lambda).

The chain uses every move type at the default mix (root displacement, tip displacement, branch reform, subtree slide, SPR
with its candidate scan, study and mutational-history proposals), rebuilds the coalescent grid every cycle (new Gaussian
auxiliary draws, the root part's grid growing into the past), and runs both on the oracle (CPU) and on the HIP engine."""
import math

import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import Scenario
from helpers import configure
from oracle_ffi import OracleEngine

L, N_POP, B, MU, KAPPA, PI = 3000, 100.0, 200.0, 1e-3 / 365.0, 5.0, (0.31, 0.19, 0.21, 0.29)
T_STEP = 2.0


def _two_tips(t_root=-30.0, t1=50.0):
    t = d.FlatTree.empty(3, 0, 0, 0)
    t.root = 0
    t.child0[0], t.child1[0] = 1, 2
    t.parent[1] = t.parent[2] = 0
    t.t[:] = (t_root, 0.0, t1)
    t.t_min[:] = (-3.4028234663852886e38, 0.0, 0.0)
    t.t_max[:] = (3.4028234663852886e38, 0.0, B)
    return t


def _chain(engine, cycles, moves_per_cycle, seed):
    rng = np.random.default_rng(seed)
    ref = rng.choice(4, size=L, p=PI).astype(np.uint8)
    tree = _two_tips()
    sc = Scenario("two tips", tree, ref, B, MU, KAPPA, PI, d.PopModel.const(N_POP), L)
    lam = None
    hs, t1s = [], []
    for c in range(cycles):
        sc.tree = tree
        configure(engine, sc, ref, [tree], [True], [seed * 1000003 + c], 0, t_step=T_STEP)
        if lam is None:
            lam = float(engine.part_derived(0, 3)[0][0])
        engine.run_moves_per_part(moves_per_cycle)
        tree = engine.part_download(0)
        assert tree.mut_offset[-1] == 0 or True          # (histories with cancelling mutation pairs are allowed; they are rare)
        tips = np.flatnonzero(tree.child0 == -1)
        moving = tips[tree.t_max[tips] > 0][0]
        hs.append(-float(tree.t[tree.root])); t1s.append(float(tree.t[moving]))
    return lam, np.array(hs), np.array(t1s)


def _batch_se(x, batches=20):
    m = np.array([b.mean() for b in np.array_split(x, batches)])
    return m.std(ddof=1) / math.sqrt(batches)


def _check(lam, hs, t1s):
    burn = len(hs) // 10
    hs, t1s = hs[burn:], t1s[burn:]
    rate_h = 1.0 / N_POP + 2.0 * lam
    want_h = 1.0 / rate_h
    want_t1 = 1.0 / lam - B / math.expm1(lam * B)
    var_t1 = 1.0 / lam ** 2 - B * B * math.exp(lam * B) / math.expm1(lam * B) ** 2
    se_h, se_t1 = _batch_se(hs), _batch_se(t1s)
    # the grid replaces k(k-1)/2 by its cell average (very_scalable_coalescent.cpp:355-386): a bias of order t_step / N
    tol_h = 4.0 * se_h + 0.03 * want_h
    tol_t1 = 4.0 * se_t1 + 0.02 * want_t1
    assert abs(hs.mean() - want_h) < tol_h, "root height: mean %.2f, closed form %.2f (se %.2f)" % (hs.mean(), want_h, se_h)
    assert abs(hs.std() - want_h) < 2.0 * tol_h, "root height: sd %.2f, closed form %.2f" % (hs.std(), want_h)
    assert abs(t1s.mean() - want_t1) < tol_t1, "tip date: mean %.2f, closed form %.2f (se %.2f; a flat prior alone would give %.1f)" % (t1s.mean(), want_t1, se_t1, B / 2)
    assert abs(t1s.std() - math.sqrt(var_t1)) < 0.06 * math.sqrt(var_t1) + 4.0 * se_t1, "tip date: sd %.2f, closed form %.2f" % (t1s.std(), math.sqrt(var_t1))
    assert 0.0 <= t1s.min() and t1s.max() <= B
    # the two are independent under the posterior
    assert abs(np.corrcoef(hs, t1s)[0, 1]) < 0.08


def test_two_tip_posterior_closed_form_on_the_oracle():
    orc = OracleEngine(L)
    try:
        lam, hs, t1s = _chain(orc, cycles=6000, moves_per_cycle=24, seed=7)
    finally:
        orc.close()
    assert 0.004 < lam < 0.02          # lambda N ~ 1: both terms of the root-height rate matter
    _check(lam, hs, t1s)


@pytest.mark.gpu
def test_two_tip_posterior_closed_form_on_the_gpu():
    gpu = d.EmatBackend(L)
    try:
        lam, hs, t1s = _chain(gpu, cycles=6000, moves_per_cycle=24, seed=11)
        gpu.synchronize()
    finally:
        gpu.close()
    _check(lam, hs, t1s)


# ---- the same idea with topology and with several parts --------------------------------------------------------------
# n tips, all dated exactly t = 0, identical sequences: while k lineages remain, the next coalescence comes at rate
# k(k-1)/(2N) (prior) + k lambda (every lineage must stay free of mutations), whatever the topology, so
#   E[root height] = sum_k 1 / (k(k-1)/(2N) + k lambda),   E[tree length] = sum_k k / (k(k-1)/(2N) + k lambda),   k = 2..n.
# The chain is the whole cycle: a fresh partition into 2-3 parts every cycle (frozen cut nodes, synthetic sub-roots), every
# move type inside the parts -- SPR moves re-hang subtrees all the time -- the coalescent prior factorised over the parts
# through the Gaussian auxiliary variables (very_scalable_coalescent.cpp:85-232), reassembly.
N_TIPS, T_STEP_N = 24, 0.5


def _ladder(n_tips):
    n = 2 * n_tips - 1
    t = d.FlatTree.empty(n, 0, 0, 0)
    # nodes 0..n_tips-2 inner (0 = root), the rest tips; inner node i has children i+1 (or a tip) and a tip
    t.root = 0
    tip = n_tips - 1
    for i in range(n_tips - 1):
        left = i + 1 if i + 1 < n_tips - 1 else tip + 1
        t.child0[i], t.child1[i] = left, tip if i + 1 < n_tips - 1 else tip
        t.parent[left] = i; t.parent[t.child1[i]] = i
        tip += 1 if i + 1 < n_tips - 1 else 2
        t.t[i] = -3.0 * (n_tips - 1 - i)
        t.t_min[i], t.t_max[i] = -3.4028234663852886e38, 3.4028234663852886e38
    assert np.all(t.parent[1:] >= 0) and tip == n
    return t


def _expected(lam, n_tips):
    rates = [(k, k * (k - 1) / (2.0 * N_POP) + k * lam) for k in range(2, n_tips + 1)]
    return sum(1.0 / r for _, r in rates), sum(k / r for k, r in rates)


def _tree_stats(tree):
    nonroot = np.arange(tree.num_nodes) != tree.root
    return -float(tree.t[tree.root]), float(np.sum(tree.t[nonroot] - tree.t[tree.parent[nonroot]]))


def _check_n(lam, hs, Ts):
    burn = len(hs) // 10
    hs, Ts = hs[burn:], Ts[burn:]
    want_h, want_T = _expected(lam, N_TIPS)
    se_h, se_T = _batch_se(hs), _batch_se(Ts)
    # cell averaging of k(k-1)/2 matters where coalescences are closer than a cell (k >= 10 here): a few per cent of the length
    assert abs(hs.mean() - want_h) < 4.0 * se_h + 0.04 * want_h, "root height: mean %.2f, closed form %.2f (se %.2f)" % (hs.mean(), want_h, se_h)
    assert abs(Ts.mean() - want_T) < 4.0 * se_T + 0.05 * want_T, "tree length: mean %.2f, closed form %.2f (se %.2f)" % (Ts.mean(), want_T, se_T)


def test_coalescent_posterior_of_n_tips_through_partitioned_cycles_on_the_oracle():
    rng = np.random.default_rng(5)
    ref = rng.choice(4, size=L, p=PI).astype(np.uint8)
    tree = _ladder(N_TIPS)
    sc = Scenario("n tips", tree, ref, 0.0, MU, KAPPA, PI, d.PopModel.const(N_POP), L)
    run = d.EmatRun(None, tree, ref, 17)
    run.set_num_parts(3)
    orc = OracleEngine(L)
    hs, Ts, lam, parts_seen = [], [], None, set()
    try:
        for cycle in range(2500):
            run.repartition()
            n, root_part = run.num_parts()
            parts_seen.add(n)
            parts, incl, seeds = zip(*(run.part(i) for i in range(n)))
            _, cur_ref = run.tree()
            configure(orc, sc, cur_ref, list(parts), list(incl), list(seeds), root_part, t_step=T_STEP_N)
            if lam is None:
                lam = float(orc.part_derived(root_part, parts[root_part].num_nodes)[0][parts[root_part].root])
            orc.run_moves_per_part(150)
            for i in range(n):
                run.part_put(i, orc.part_download(i))
            run.reassemble()
            whole, _ = run.tree()
            h, T = _tree_stats(whole)
            hs.append(h); Ts.append(T)
    finally:
        orc.close(); run.close()
    assert max(parts_seen) >= 2, "the tree was never cut: the test would not exercise the parts"
    _check_n(lam, np.array(hs), np.array(Ts))


@pytest.mark.gpu
def test_coalescent_posterior_of_n_tips_through_partitioned_cycles_on_the_gpu():
    rng = np.random.default_rng(5)
    ref = rng.choice(4, size=L, p=PI).astype(np.uint8)
    tree = _ladder(N_TIPS)
    # the third arm: two parts requested and a part-size limit of 21 nodes (emat_run_set_max_part_nodes: parts above it get further cut
    # nodes, drawn uniformly among their inner nodes -- the rule the run driver applies by default at three times the mean part
    # size, which this 47-node tree never reaches): the same closed-form posterior must come out
    for device_tree, parts, limit in ((False, 3, -1), (True, 3, -1), (True, 2, 21)):
        b = d.EmatBackend(L)
        run = d.EmatRun(b, tree, ref, 23)
        run.set_num_parts(parts); run.set_max_part_nodes(limit); run.set_hky(MU, KAPPA, PI); run.set_pop_model(d.PopModel.const(N_POP)); run.set_coalescent_t_step(T_STEP_N)
        if device_tree:
            run.set_device_tree(True)
        hs, Ts, cut = [], [], 0
        for cycle in range(2500):
            run.do_mcmc_steps(450, 450)
            st = run.partition_stats(); cut += st["extra_cuts"]
            assert limit < 0 or st["largest_part_nodes"] <= limit
            whole, _ = run.tree()
            h, T = _tree_stats(whole)
            hs.append(h); Ts.append(T)
        assert (cut > 2500) == (limit > 0), (limit, cut)
        # lambda: the rate of leaving the reference sequence (no mutations anywhere, so every node has it)
        chk = OracleEngine(L)
        sc = Scenario("n tips", whole, ref, 0.0, MU, KAPPA, PI, d.PopModel.const(N_POP), L)
        configure(chk, sc, ref, [tree], [True], [1], 0, t_step=T_STEP_N)
        lam = float(chk.part_derived(0, tree.num_nodes)[0][0])
        chk.close(); run.close(); b.close()
        _check_n(lam, np.array(hs), np.array(Ts))


# ---- two tips that DIFFER at m sites ---------------------------------------------------------------------------------
# Both dated exactly t = 0.  To first order in mu h every differing site contributes a factor proportional to the total
# branch length 2h (one mutation, somewhere on either branch; by reversibility both placements weigh the same), every other
# site exp(-2 q mu h), the prior exp(-h/N):  h ~ Gamma(m + 1, 1/N + 2 lambda),  E[h] = (m + 1) / rate,  sd = sqrt(m + 1) / rate.
# What this one reaches that the others do not: explicit mutations -- their times bound the displacement of the root and
# are re-drawn by branch reforms, they hop between the two branches through the root (the root sequence and with it the
# reference sequence change: Run::normalize_root), and each carries its log(mu q_ab) in log G.
M_DIFF = 3


def _two_tips_differing(ref, rng):
    t = d.FlatTree.empty(3, M_DIFF, 0, 0)
    t.root = 0
    t.child0[0], t.child1[0] = 1, 2
    t.parent[1] = t.parent[2] = 0
    t.t[:] = (-120.0, 0.0, 0.0)
    t.t_min[:] = (-3.4028234663852886e38, 0.0, 0.0)
    t.t_max[:] = (3.4028234663852886e38, 0.0, 0.0)
    sites = np.sort(rng.choice(L, size=M_DIFF, replace=False))
    times = np.sort(rng.uniform(-119.0, -1.0, size=M_DIFF))
    order = np.argsort(times, kind="stable")          # mutations of a branch are sorted by (t, site)
    t.mut_offset[:] = (0, 0, 0, M_DIFF)               # all on the branch to node 2
    for k, j in enumerate(order):
        t.mut_site[k] = sites[j]; t.mut_from[k] = ref[sites[j]]; t.mut_to[k] = (ref[sites[j]] + 1 + (sites[j] % 3)) % 4; t.mut_t[k] = times[j]
    return t


def _chain_differing(make_engine, via_run_driver, seed, cycles=5000):
    rng = np.random.default_rng(seed)
    ref = rng.choice(4, size=L, p=PI).astype(np.uint8)
    tree = _two_tips_differing(ref, rng)
    sc = Scenario("two differing tips", tree, ref, 0.0, MU, KAPPA, PI, d.PopModel.const(N_POP), L)
    hs, on_first, ref_changed = [], [], False
    if via_run_driver:      # the whole cycle, reference sequence re-normalised by the driver
        b = make_engine()
        run = d.EmatRun(b, tree, ref, seed)
        run.set_num_parts(1); run.set_hky(MU, KAPPA, PI); run.set_pop_model(d.PopModel.const(N_POP)); run.set_coalescent_t_step(T_STEP)
        run.set_device_tree(True)
        for c in range(cycles):
            run.do_mcmc_steps(30, 30)
            tr, cur_ref = run.tree()
            hs.append(-float(tr.t[tr.root])); on_first.append(int(tr.mut_offset[2] - tr.mut_offset[1]))
            ref_changed = ref_changed or not np.array_equal(cur_ref, ref)
            assert tr.mut_offset[-1] >= M_DIFF
        run.close(); b.close()
    else:                   # one part handed to the engine cycle after cycle; the root keeps its deltas from `ref`
        eng = make_engine()
        for c in range(cycles):
            sc.tree = tree
            configure(eng, sc, ref, [tree], [True], [seed * 7919 + c], 0, t_step=T_STEP)
            eng.run_moves_per_part(30)
            tree = eng.part_download(0)
            hs.append(-float(tree.t[tree.root]))
            r = tree.root; kids = (tree.child0[r], tree.child1[r])
            on_first.append(int(tree.mut_offset[kids[0] + 1] - tree.mut_offset[kids[0]]))
            ref_changed = ref_changed or tree.mut_offset[r + 1] > tree.mut_offset[r]
        eng.close()
    return np.array(hs), np.array(on_first), ref_changed, ref


def _check_differing(hs, on_first, ref_changed, lam):
    burn = len(hs) // 10
    hs, on_first = hs[burn:], on_first[burn:]
    rate = 1.0 / N_POP + 2.0 * lam
    want, want_sd = (M_DIFF + 1) / rate, math.sqrt(M_DIFF + 1) / rate
    se = _batch_se(hs)
    assert abs(hs.mean() - want) < 4.0 * se + 0.03 * want, "root height: mean %.2f, closed form %.2f (se %.2f)" % (hs.mean(), want, se)
    assert abs(hs.std() - want_sd) < 0.08 * want_sd + 4.0 * se, "root height: sd %.2f, closed form %.2f" % (hs.std(), want_sd)
    # each mutation sits on either branch with probability 1/2
    assert abs(on_first.mean() - M_DIFF / 2.0) < 0.15, "mutations on the first branch: %.2f of %d on average" % (on_first.mean(), M_DIFF)
    assert ref_changed, "no mutation ever crossed the root: the test does not exercise the re-rooting of sequences"


def _lambda_of(ref):
    chk = OracleEngine(L)
    t = _two_tips()
    sc = Scenario("x", t, ref, B, MU, KAPPA, PI, d.PopModel.const(N_POP), L)
    configure(chk, sc, ref, [t], [True], [1], 0, t_step=T_STEP)
    lam = float(chk.part_derived(0, 3)[0][0])
    chk.close()
    return lam


def test_two_differing_tips_posterior_closed_form_on_the_oracle():
    hs, on_first, ref_changed, ref = _chain_differing(lambda: OracleEngine(L), False, 31)
    _check_differing(hs, on_first, ref_changed, _lambda_of(ref))


@pytest.mark.gpu
def test_two_differing_tips_posterior_closed_form_on_the_gpu():
    hs, on_first, ref_changed, ref = _chain_differing(lambda: d.EmatBackend(L), False, 37)
    _check_differing(hs, on_first, ref_changed, _lambda_of(ref))
    hs, on_first, ref_changed, ref = _chain_differing(lambda: d.EmatBackend(L), True, 41)      # whole cycles, tree resident in HBM
    _check_differing(hs, on_first, ref_changed, _lambda_of(ref))
