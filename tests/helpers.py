"""Shared helpers for the parity tests (HIP engine vs CPU oracle)."""
import numpy as np

import delphy_amd as d
from oracle_ffi import OracleEngine


def split_parts(sc, num_parts, seed):
    """Partition the scenario's tree with the product's host driver (no GPU needed)."""
    run = d.EmatRun(None, sc.tree, sc.ref, seed)
    run.set_num_parts(num_parts)
    run.repartition()
    n, root_part = run.num_parts()
    parts, incl, seeds = [], [], []
    for i in range(n):
        t, r, s = run.part(i)
        parts.append(t); incl.append(r); seeds.append(s)
    _, ref = run.tree()
    run.close()
    return parts, incl, seeds, root_part, ref


def configure(engine, sc, ref, parts, incl, seeds, root_part, t_step=None, topology=True, only_displace=False, nu_l=None, evo=None):
    """`evo` = (mu[P], pi[P][4], q[P][4][4], partition_for_site[L]) replaces the scenario's single HKY partition."""
    engine.set_ref_sequence(ref)
    if evo is not None:
        mu, pi, q, pfs = evo
        engine.set_evo(mu, pi, q, nu_l if nu_l is not None else np.ones(sc.num_sites), pfs)
    else:
        engine.set_hky(sc.mu, sc.kappa, sc.pi, nu_l)
    engine.set_flags(sc.t_max_tip, only_displace, topology)
    engine.upload_parts(parts, incl, seeds)
    engine.build_coalescent_parts(sc.pop, root_part, t_step if t_step is not None else sc.default_t_step())


def rel_close(a, b, tol):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.all(np.abs(a - b) <= tol * np.maximum(1.0, np.maximum(np.abs(a), np.abs(b))))


def assert_trees_match(tg: d.FlatTree, to: d.FlatTree, tol=1e-9, what=""):
    """Topology, sites, states, interval endpoints bit-exact; times within `tol` relative."""
    assert tg.root == to.root, what
    for f in ("parent", "child0", "child1", "mut_offset", "mut_site", "mut_from", "mut_to", "miss_offset", "miss_start", "miss_end",
              "mfs_offset", "mfs_site", "mfs_state"):
        a, b = getattr(tg, f), getattr(to, f)
        assert a.shape == b.shape and np.array_equal(a, b), "%s: %s differs" % (what, f)
    assert np.array_equal(tg.t_min, to.t_min) and np.array_equal(tg.t_max, to.t_max), what
    assert rel_close(tg.t, to.t, tol), "%s: node times differ: max %g" % (what, np.max(np.abs(tg.t - to.t)))
    finite = np.abs(to.mut_t) < 1e300
    assert np.array_equal(finite, np.abs(tg.mut_t) < 1e300), what
    assert rel_close(tg.mut_t[finite], to.mut_t[finite], tol), "%s: mutation times differ" % what


def assert_traces_match(trg, tro, tol=1e-9, what="", abs_floor=0.0):
    """`abs_floor`: what rounding alone may put between two evaluations of a log MH ratio whose TERMS are large: the cells a root part appends to its
    coalescent grid get their population integral from the device's exp / expm1 on one side and glibc's on the other (one unit in the last place apart), and
    under an exponential-growth model whose partial prior is -9e7 a tip displacement's ratio of -0.02 came out 6e-9 apart (scripts/fuzz_big_parts.py 6800,
    case 445).  compare_part passes 16 units in the last place of the largest partial prior the part has held."""
    assert trg.shape == tro.shape, "%s: trace length %s vs %s" % (what, trg.shape, tro.shape)
    for i in range(trg.shape[0]):
        kg, ko = trg[i], tro[i]
        assert kg[0] == ko[0] and kg[1] == ko[1] and kg[2] == ko[2], "%s: move %d differs: gpu %s oracle %s" % (what, i, kg, ko)
        if np.isnan(ko[3]) or np.isnan(kg[3]):
            assert np.isnan(ko[3]) and np.isnan(kg[3]), "%s: move %d log_mh nan mismatch: %s %s" % (what, i, kg, ko)
        elif np.isinf(ko[3]):
            assert kg[3] == ko[3], "%s: move %d" % (what, i)
        else:
            assert abs(kg[3] - ko[3]) <= tol * max(1.0, abs(ko[3])) + abs_floor, "%s: move %d log_mh %r vs %r" % (what, i, kg[3], ko[3])


def compare_part(gpu, orc, p, num_nodes, trace, tol, expected_moves, totals_scale=(1.0, 1.0)):
    """Everything one part's chain leaves behind, HIP engine vs oracle: move trace, counters, RNG consumption, tree, derived
    quantities and coalescent cells.  `totals_scale`: the largest magnitudes log_G and the partial prior have had -- both sides maintain them
    INCREMENTALLY, as the reference does, so a total keeps the absolute rounding error of the largest value it ever held (EMAT_FUZZ_SEED=6200,
    case 49: a partial prior that starts at -1.7e21 under an exponential-growth model and ends at -1.2e14 is one unit in the last place of
    1.7e21 = 262144 apart on the two sides, and the oracle's own recomputation is 104112 away from its own running value)."""
    sg, so = gpu.part_stats(p), orc.part_stats(p)
    assert sg["status"] == 0, "part %d device status %d: %s" % (p, sg["status"], gpu.last_error())
    assert_traces_match(gpu.part_trace(p, trace), orc.part_trace(p, trace), tol, "part %d" % p,
                        abs_floor=16 * np.finfo(np.float64).eps * max(totals_scale[1], abs(float(orc.part_derived(p, num_nodes)[3]))))
    assert sg["moves_done"] == so["moves_done"] == expected_moves
    assert sg["proposed"] == so["proposed"] and sg["accepted"] == so["accepted"], "part %d counters %s vs %s" % (p, sg, so)
    assert sg["rng_draws"] == so["rng_draws"], "part %d rng draws %d vs %d" % (p, sg["rng_draws"], so["rng_draws"])
    assert_trees_match(gpu.part_download(p), orc.part_download(p), tol, "part %d" % p)
    lg, ng, Gg, Ag = gpu.part_derived(p, num_nodes)
    lo, no, Go, Ao = orc.part_derived(p, num_nodes)
    assert np.array_equal(ng, no)
    assert rel_close(lg, lo, 1e-9), "part %d lambda_i after moves" % p
    assert abs(Gg - Go) <= tol * max(1.0, abs(Go), totals_scale[0]) and abs(Ag - Ao) <= tol * max(1.0, abs(Ao), totals_scale[1]), \
        "part %d totals after moves: %r/%r %r/%r" % (p, Gg, Go, Ag, Ao)
    cg, co = gpu.part_coalescent(p), orc.part_coalescent(p)
    assert cg["k_bar_p"].shape == co["k_bar_p"].shape
    assert rel_close(cg["k_bar_p"], co["k_bar_p"], 1e-9) and rel_close(cg["k_twiddle_bar_p"], co["k_twiddle_bar_p"], 1e-9)


def run_parity(sc, num_parts, moves_per_part, seed=11, topology=True, only_displace=False, trace=200, use_lds=True, t_step=None, nu_l=None, tol=1e-9,
               evo=None, total_moves=None):
    """`total_moves`: drive both engines through run_local_moves(total) (reference Run::run_local_moves: count / parts each,
    remainder to part 0) instead of the same count on every part."""
    """Run the same seeded scenario through the HIP engine and the oracle and compare everything."""
    parts, incl, seeds, root_part, ref = split_parts(sc, num_parts, seed)
    gpu = d.EmatBackend(sc.num_sites, trace_moves=trace, use_lds=use_lds)
    orc = OracleEngine(sc.num_sites, trace_moves=trace)
    try:
        configure(gpu, sc, ref, parts, incl, seeds, root_part, t_step, topology, only_displace, nu_l, evo)
        configure(orc, sc, ref, parts, incl, seeds, root_part, t_step, topology, only_displace, nu_l, evo)
        # derived quantities from scratch
        scales = []
        for p in range(len(parts)):
            n = parts[p].num_nodes
            lg, ng, Gg, Ag = gpu.part_derived(p, n)
            lo, no, Go, Ao = orc.part_derived(p, n)
            scales.append((abs(float(Go)), abs(float(Ao))))
            assert np.array_equal(ng, no), "part %d num_sites_missing" % p
            assert rel_close(lg, lo, 1e-11), "part %d lambda_i max diff %g" % (p, np.max(np.abs(lg - lo)))
            assert rel_close(Gg, Go, tol) and rel_close(Ag, Ao, tol), "part %d log_G %r/%r prior %r/%r" % (p, Gg, Go, Ag, Ao)
        if moves_per_part > 0:
            if total_moves is not None:
                gpu.run_local_moves(total_moves)
                gpu.synchronize()
                orc.run_local_moves(total_moves, threads=4)
            else:
                gpu.run_moves_per_part(moves_per_part)
                gpu.synchronize()
                orc.run_moves_per_part(moves_per_part, threads=4)
            for p in range(len(parts)):
                expected_moves = moves_per_part if total_moves is None else total_moves // len(parts) + (total_moves - len(parts) * (total_moves // len(parts)) if p == 0 else 0)
                compare_part(gpu, orc, p, parts[p].num_nodes, trace, tol, expected_moves, scales[p])
            Gg, Ag = gpu.totals(); Go, Ao = orc.totals()
            assert abs(Gg - Go) <= tol * max(1.0, abs(Go), sum(s[0] for s in scales)) and abs(Ag - Ao) <= tol * max(1.0, abs(Ao), sum(s[1] for s in scales))
            # ... and recomputed from scratch they agree relative to what they ARE (the trees were compared above; this is the arithmetic of the recomputation)
            gpu.recalc_derived(); orc.recalc_derived()
            assert rel_close(np.array(gpu.totals()), np.array(orc.totals()), tol)
        return gpu.part_stats(0)
    finally:
        gpu.close(); orc.close()


def replay_device_parts_in_the_oracle(sc, b, run, ref, moves_total, trace, oracle_parts=None):
    """Hand the oracle exactly what the device starts a pass from -- every part's tree as the kernels cut it, the coalescent
    tables as the kernels built them, the position of every part's random stream -- run the pass on both, compare everything.
    `oracle_parts`: a list that receives the ORACLE's part trees after the pass (what its own moves made of them)."""
    n, root_part = run.num_parts()
    trees = [b.part_download(p) for p in range(n)]
    rngs = [b.part_rng(p) for p in range(n)]
    orc = OracleEngine(sc.num_sites, trace_moves=trace)
    try:
        orc.set_ref_sequence(ref); orc.set_hky(sc.mu, sc.kappa, sc.pi); orc.set_flags(sc.t_max_tip)
        orc.upload_parts(trees, [p == root_part for p in range(n)], [r["key"] for r in rngs])
        for p in range(n):
            tab = b.part_coalescent(p)
            # the device keeps the cells of the part's own time window; outside it the part has no lineages (k_bar_p = 0), the
            # shared vectors are reported as unknown (NaN / -1), and the reference's sum over ALL cells gets 0 from each such
            # cell whatever they hold: give them neutral values (a chain that wandered out there would differ from the device's)
            out = tab["num_active_parts"] < 0
            assert np.all(tab["k_bar_p"][out] == 0.0) and np.all(tab["k_twiddle_bar_p"][out] == 0.0)
            tab["k_twiddle_bar"] = np.where(out, 0.0, tab["k_twiddle_bar"]); tab["popsize_bar"] = np.where(out, 1.0, tab["popsize_bar"])
            tab["num_active_parts"] = np.where(out, 0, tab["num_active_parts"])
            assert not np.any(np.isnan(tab["k_twiddle_bar"])) and not np.any(np.isnan(tab["popsize_bar"]))
            orc.set_coalescent_part(sc.pop, p, p == root_part, tab, rngs[p])
        for p in range(n):       # from-scratch derived quantities on the tables both sides now share
            lg, ng, Gg, Ag = b.part_derived(p, trees[p].num_nodes)
            lo, no, Go, Ao = orc.part_derived(p, trees[p].num_nodes)
            assert np.array_equal(ng, no) and rel_close(lg, lo, 1e-11) and rel_close(Gg, Go, 1e-9) and rel_close(Ag, Ao, 1e-9), (p, Gg, Go, Ag, Ao)
        run.run_moves(moves_total); b.synchronize()
        counts = np.full(n, moves_total // n, np.int64); counts[: moves_total % n] += 1      # emat_run_moves spreads the remainder one move per part
        orc.run_moves_counts(counts, threads=4)
        for p in range(n):
            compare_part(b, orc, p, trees[p].num_nodes, trace, 1e-9, int(counts[p]))
            if oracle_parts is not None:
                oracle_parts.append(orc.part_download(p))
    finally:
        orc.close()
    return n


from delphy_amd.scenarios import random_scenario  # noqa: E402,F401  (the seeded random scenarios of the sweeps)
