"""ctypes binding of the CPU oracle (oracle/orc_capi.cpp).  TEST INFRASTRUCTURE ONLY: nothing under
delphy_amd/ imports this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do."""
import ctypes as C
import os

import numpy as np

from delphy_amd.engine import FlatTree, PopModel, _FlatTreeC, _PartStatsC, _PopModelC, _ptr, hky_q_matrix

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ROOT, "oracle", "_build", "libemat_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle library missing: run `make -C oracle`")
        L = C.CDLL(path)
        E = C.c_void_p
        i32, i64, u64, dbl = C.c_int32, C.c_int64, C.c_uint64, C.c_double
        P = C.POINTER
        sigs = {
            "orc_create": [C.c_int, C.c_int, P(E)], "orc_destroy": [E],
            "orc_set_ref_sequence": [E, P(C.c_uint8), C.c_int], "orc_set_evo": [E, C.c_int, P(dbl), P(dbl), P(dbl), P(dbl), P(C.c_int)],
            "orc_set_flags": [E, dbl, C.c_int, C.c_int], "orc_begin_upload": [E, C.c_int],
            "orc_part_upload": [E, C.c_int, P(_FlatTreeC), C.c_int, u64], "orc_end_upload": [E],
            "orc_build_coalescent_parts": [E, P(_PopModelC), C.c_int, dbl], "orc_recalc_derived": [E],
            "orc_set_coalescent_part": [E, P(_PopModelC), C.c_int, C.c_int, C.c_int, P(dbl), P(dbl), P(dbl), P(dbl), P(C.c_int), dbl, dbl, u64, u64, C.c_int],
            "orc_run_moves": [E, P(i64), C.c_int, C.c_int], "orc_get_totals": [E, P(dbl), P(dbl)], "orc_global_stats": [E, C.c_int, P(dbl), P(i64), P(i64)],
            "orc_part_get_sizes": [E, C.c_int, P(C.c_int), P(C.c_int), P(C.c_int), P(C.c_int)], "orc_part_download": [E, C.c_int, P(_FlatTreeC)],
            "orc_part_get_derived": [E, C.c_int, P(dbl), P(C.c_int), P(dbl), P(dbl)],
            "orc_part_get_coalescent": [E, C.c_int, P(C.c_int), P(dbl), P(dbl), P(dbl), P(dbl), P(C.c_int), P(dbl), P(dbl)],
            "orc_part_get_stats": [E, C.c_int, P(_PartStatsC)], "orc_part_get_trace": [E, C.c_int, P(C.c_int), P(dbl)],
            "orc_part_check": [E, C.c_int, C.c_char_p, C.c_int],
            "orc_Ttwiddle_l": [E, C.c_int, P(dbl)], "orc_num_muts_l": [E, P(C.c_int)], "orc_scalable_log_prior": [E, C.c_int, dbl, dbl, P(dbl)],
            "orc_tree_query": [E, C.c_int, C.c_int, C.c_int, P(C.c_int), P(C.c_int), P(C.c_int)],
            "orc_debug_graft": [E, C.c_int, C.c_int, dbl, C.c_int, C.c_int, dbl, P(dbl), C.c_int, P(C.c_int)], "orc_part_log_G": [E, C.c_int, P(dbl), P(dbl)], "orc_debug_edit": [E, C.c_int, C.c_int, C.c_int, P(C.c_int), P(C.c_int), P(dbl), P(dbl), P(C.c_int)],
            "orc_debug_sample_history": [E, C.c_int, C.c_int, P(C.c_int), P(dbl), P(C.c_uint8), dbl, dbl, P(C.c_int), P(dbl), C.c_int, P(C.c_int)],
        }
        for n, a in sigs.items():
            f = getattr(L, n); f.argtypes = a; f.restype = C.c_int
        L.orc_last_error.argtypes = [E]; L.orc_last_error.restype = C.c_char_p
        L.orc_gamma_q.argtypes = [dbl, dbl]; L.orc_gamma_q.restype = dbl
        L.orc_gamma_q_inv.argtypes = [dbl, dbl]; L.orc_gamma_q_inv.restype = dbl
        for n in ("orc_pop_at_time",):
            getattr(L, n).argtypes = [P(_PopModelC), dbl]; getattr(L, n).restype = dbl
        for n in ("orc_pop_integral", "orc_intensity_integral"):
            getattr(L, n).argtypes = [P(_PopModelC), dbl, dbl]; getattr(L, n).restype = dbl
        L.orc_interval_op.argtypes = [C.c_int, P(C.c_int), C.c_int, P(C.c_int), C.c_int, P(C.c_int), C.c_int]; L.orc_interval_op.restype = C.c_int
        L.orc_rng_block.argtypes = [u64, u64, P(C.c_uint32)]; L.orc_rng_block.restype = None
        _lib = L
    return _lib


class OracleEngine:
    """Same call sequence as delphy_amd.EmatBackend, executed by the CPU oracle."""

    def __init__(self, num_sites, trace_moves=0):
        self.L = lib()
        self.num_sites = num_sites
        self.h = C.c_void_p()
        assert self.L.orc_create(num_sites, trace_moves, C.byref(self.h)) == 0

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h); self.h = None

    def _ck(self, st, what):
        if st != 0:
            raise RuntimeError("%s failed: %s" % (what, self.L.orc_last_error(self.h).decode()))

    def set_ref_sequence(self, ref):
        ref = np.ascontiguousarray(ref, np.uint8)
        self._ck(self.L.orc_set_ref_sequence(self.h, _ptr(ref, C.c_uint8), ref.shape[0]), "set_ref_sequence")

    def set_evo(self, mu, pi, q, nu_l, pfs):
        mu = np.ascontiguousarray(mu, np.float64).reshape(-1); P = mu.shape[0]
        pi = np.ascontiguousarray(pi, np.float64).reshape(P * 4); q = np.ascontiguousarray(q, np.float64).reshape(P * 16)
        nu_l = np.ascontiguousarray(nu_l, np.float64); pfs = np.ascontiguousarray(pfs, np.int32)
        self._ck(self.L.orc_set_evo(self.h, P, _ptr(mu, C.c_double), _ptr(pi, C.c_double), _ptr(q, C.c_double), _ptr(nu_l, C.c_double), _ptr(pfs, C.c_int)), "set_evo")

    def set_hky(self, mu, kappa, pi, nu_l=None):
        nu = np.ones(self.num_sites) if nu_l is None else nu_l
        self.set_evo([mu], [pi], [hky_q_matrix(kappa, pi)], nu, np.zeros(self.num_sites, np.int32))

    def set_flags(self, t_max_tip, only_displacing_inner_nodes=False, topology_moves_enabled=True):
        self._ck(self.L.orc_set_flags(self.h, t_max_tip, int(only_displacing_inner_nodes), int(topology_moves_enabled)), "set_flags")

    def upload_parts(self, parts, includes_run_root, seeds):
        self._ck(self.L.orc_begin_upload(self.h, len(parts)), "begin_upload")
        for i, (t, r, s) in enumerate(zip(parts, includes_run_root, seeds)):
            v = t.c_view()
            self._ck(self.L.orc_part_upload(self.h, i, C.byref(v), int(r), int(s)), "part_upload")
        self.num_parts = len(parts)

    def build_coalescent_parts(self, pop: PopModel, root_part_index, t_step):
        m = pop.c_struct()
        self._ck(self.L.orc_build_coalescent_parts(self.h, C.byref(m), root_part_index, t_step), "build_coalescent_parts")

    def set_coalescent_part(self, pop: PopModel, part, includes_tree_root, tables: dict, rng: dict):
        """The part's coalescent arrays (as EmatBackend.part_coalescent returns them) and RNG position (EmatBackend.part_rng) from outside."""
        m = pop.c_struct()
        kb, kt, k, ps = (np.ascontiguousarray(tables[n], np.float64) for n in ("k_bar_p", "k_twiddle_bar_p", "k_twiddle_bar", "popsize_bar"))
        na = np.ascontiguousarray(tables["num_active_parts"], np.int32)
        self._ck(self.L.orc_set_coalescent_part(self.h, C.byref(m), part, int(includes_tree_root), kb.shape[0], _ptr(kb, C.c_double), _ptr(kt, C.c_double), _ptr(k, C.c_double),
                                                _ptr(ps, C.c_double), _ptr(na, C.c_int), tables["t_ref"], tables["t_step"], rng["counter"], rng["spare"], int(rng["has_spare"])), "set_coalescent_part")

    def debug_graft(self, part, X, mu_proposal, mode=0, new_sibling=0, new_t_P=0.0):
        """The oracle's Spr_move on one part, in the layout and with the modes of EmatBackend.debug_graft."""
        from delphy_amd.engine import decode_graft_output
        out = np.zeros(4096); n = C.c_int()
        self._ck(self.L.orc_debug_graft(self.h, part, X, mu_proposal, mode, new_sibling, new_t_P, _ptr(out, C.c_double), out.shape[0], C.byref(n)), "debug_graft")
        return decode_graft_output(out[: n.value], mode)

    def debug_sample_history(self, part, branch, t_end, start_seq, T, mu):
        branch = np.ascontiguousarray(branch, np.int32); t_end = np.ascontiguousarray(t_end, np.float64); seq = np.ascontiguousarray(start_seq, np.uint8)
        n = branch.shape[0]; cap = 64 * n + 1024
        counts = np.zeros(max(n, 1), np.int32); muts = np.zeros((cap, 4)); tot = C.c_int()
        self._ck(self.L.orc_debug_sample_history(self.h, part, n, _ptr(branch, C.c_int), _ptr(t_end, C.c_double), _ptr(seq, C.c_uint8), T, mu,
                                                 _ptr(counts, C.c_int), _ptr(muts, C.c_double), cap, C.byref(tot)), "debug_sample_history")
        out, k = [], 0
        for i in range(n):
            out.append([[int(m[1]), int(m[0]), int(m[2]), float(m[3])] for m in muts[k: k + counts[i]]]); k += int(counts[i])
        return out

    def debug_edit(self, part, X, ops):
        """One editing session as EmatBackend.debug_edit; returns (largest deviation of the kept lambda_i, nodes with a wrong missing-site count)."""
        kind = np.array([{"slide": 0, "hop_up": 1, "flip": 2, "hop_down": 3}[o[0]] for o in ops], np.int32)
        node = np.array([int(o[1]) if o[0] == "hop_down" else -1 for o in ops], np.int32)
        t = np.array([float(o[1]) if o[0] == "slide" else 0.0 for o in ops], np.float64)
        dev, bad = C.c_double(), C.c_int()
        self._ck(self.L.orc_debug_edit(self.h, part, X, kind.shape[0], _ptr(kind, C.c_int), _ptr(node, C.c_int), _ptr(t, C.c_double), C.byref(dev), C.byref(bad)), "debug_edit")
        return dev.value, bad.value

    def part_log_G(self, part):
        """(log G as maintained incrementally, log G recomputed from scratch) of one part."""
        a, b = C.c_double(), C.c_double()
        self._ck(self.L.orc_part_log_G(self.h, part, C.byref(a), C.byref(b)), "part_log_G")
        return a.value, b.value

    def recalc_derived(self):
        self._ck(self.L.orc_recalc_derived(self.h), "recalc_derived")

    def run_moves_per_part(self, moves, threads=1, paranoid=False):
        m = np.full(self.num_parts, moves, np.int64)
        self._ck(self.L.orc_run_moves(self.h, _ptr(m, C.c_int64), threads, int(paranoid)), "run_moves")

    def run_moves_counts(self, counts, threads=1, paranoid=False):
        """An explicit number of moves for every part (0 = leave the part alone)."""
        m = np.ascontiguousarray(counts, np.int64)
        assert m.shape[0] == self.num_parts
        self._ck(self.L.orc_run_moves(self.h, _ptr(m, C.c_int64), threads, int(paranoid)), "run_moves")

    def run_local_moves(self, count, threads=1, paranoid=False):
        sub = count // self.num_parts
        m = np.full(self.num_parts, sub, np.int64); m[0] = count - (self.num_parts - 1) * sub
        self._ck(self.L.orc_run_moves(self.h, _ptr(m, C.c_int64), threads, int(paranoid)), "run_moves")

    def totals(self):
        g, a = C.c_double(), C.c_double()
        self._ck(self.L.orc_get_totals(self.h, C.byref(g), C.byref(a)), "get_totals")
        return g.value, a.value

    def global_stats(self, num_partitions=1):
        T = np.zeros((num_partitions, 4)); M = np.zeros((num_partitions, 4, 4), np.int64); nm = C.c_int64()
        self._ck(self.L.orc_global_stats(self.h, num_partitions, _ptr(T, C.c_double), _ptr(M, C.c_int64), C.byref(nm)), "global_stats")
        return T, M, int(nm.value)

    def Ttwiddle_l(self, part):
        out = np.zeros(self.num_sites)
        self._ck(self.L.orc_Ttwiddle_l(self.h, part, _ptr(out, C.c_double)), "Ttwiddle_l")
        return out

    def num_muts_l(self):
        out = np.zeros(self.num_sites, np.int32)
        self._ck(self.L.orc_num_muts_l(self.h, _ptr(out, C.c_int)), "num_muts_l")
        return out

    def scalable_log_prior(self, part, t_ref, t_step):
        v = C.c_double()
        self._ck(self.L.orc_scalable_log_prior(self.h, part, t_ref, t_step, C.byref(v)), "scalable_log_prior")
        return v.value

    def tree_query(self, part, op, a, b):
        a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32); out = np.zeros_like(a); ip = C.POINTER(C.c_int)
        self._ck(self.L.orc_tree_query(self.h, part, op, a.shape[0], a.ctypes.data_as(ip), b.ctypes.data_as(ip), out.ctypes.data_as(ip)), "tree_query")
        return out

    def part_download(self, part):
        n, nm, ni, nf = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.L.orc_part_get_sizes(self.h, part, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf))
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view()
        self._ck(self.L.orc_part_download(self.h, part, C.byref(v)), "part_download")
        t.root = v.root
        return t.trimmed()

    def part_derived(self, part, num_nodes):
        lam = np.zeros(num_nodes); nm = np.zeros(num_nodes, np.int32); g, a = C.c_double(), C.c_double()
        self._ck(self.L.orc_part_get_derived(self.h, part, _ptr(lam, C.c_double), _ptr(nm, C.c_int), C.byref(g), C.byref(a)), "part_get_derived")
        return lam, nm, g.value, a.value

    def part_coalescent(self, part, cap=1 << 16):
        for cap in (cap, cap << 5):     # (a grid longer than that: a root that wandered for 100 000 moves over a fine grid)
            n = C.c_int(cap); kb, kt, k, ps = np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros(cap); na = np.zeros(cap, np.int32); tr, ts = C.c_double(), C.c_double()
            rc = self.L.orc_part_get_coalescent(self.h, part, C.byref(n), _ptr(kb, C.c_double), _ptr(kt, C.c_double), _ptr(k, C.c_double), _ptr(ps, C.c_double), _ptr(na, C.c_int), C.byref(tr), C.byref(ts))
            if rc == 0:
                break
        self._ck(rc, "part_get_coalescent")
        m = n.value
        return dict(k_bar_p=kb[:m], k_twiddle_bar_p=kt[:m], k_twiddle_bar=k[:m], popsize_bar=ps[:m], num_active_parts=na[:m], t_ref=tr.value, t_step=ts.value)

    def part_stats(self, part):
        s = _PartStatsC(); self.L.orc_part_get_stats(self.h, part, C.byref(s))
        return dict(status=s.status, num_nodes=s.num_nodes, moves_done=s.moves_done, proposed=list(s.proposed), accepted=list(s.accepted), rng_draws=s.rng_draws)

    def part_trace(self, part, cap):
        n = C.c_int(cap); tr = np.zeros((max(cap, 1), 4))
        self.L.orc_part_get_trace(self.h, part, C.byref(n), _ptr(tr, C.c_double))
        return tr[: n.value]

    def part_check(self, part):
        buf = C.create_string_buffer(512)
        rc = self.L.orc_part_check(self.h, part, buf, 512)
        return rc, buf.value.decode()


def interval_op(op, a, b=()):
    """The oracle's interval-set algebra on lists of [start, end): op 0 insert sequence, 1 merge, 2 intersect, 3 subtract ->
    list of pairs; 4 is_subset_of, 5 contains (b = [site]), 6 intersects -> bool."""
    A = np.ascontiguousarray(np.asarray(a, np.int32).reshape(-1)); B = np.ascontiguousarray(np.asarray(b, np.int32).reshape(-1))
    out = np.zeros(A.shape[0] + B.shape[0] + 2, np.int32)
    n = lib().orc_interval_op(op, _ptr(A, C.c_int), A.shape[0] // 2, _ptr(B, C.c_int), B.shape[0] if op == 5 else B.shape[0] // 2, _ptr(out, C.c_int), out.shape[0] // 2)
    assert n >= 0
    return out[: 2 * n].reshape(-1, 2).tolist() if op <= 3 else bool(out[0])


class OracleRun:
    """oracle/orc_run.hpp: the reference's Run::repartition / reassemble / normalize_root and tree_partitioning.h restated."""

    def __init__(self, tree: FlatTree, ref, seed, num_parts):
        self.L = lib()
        E = C.c_void_p
        if not hasattr(self.L, "_run_sigs"):
            i = C.c_int; P = C.POINTER
            for n, a in {"orc_run_create": [P(_FlatTreeC), P(C.c_uint8), i, C.c_uint64, i, P(E)], "orc_run_destroy": [E], "orc_run_repartition": [E],
                         "orc_run_num_parts": [E, P(i), P(i)], "orc_run_part_sizes": [E, i, P(i), P(i), P(i), P(i)],
                         "orc_run_part_get": [E, i, P(_FlatTreeC), P(i), P(i)], "orc_run_part_put": [E, i, P(_FlatTreeC)], "orc_run_reassemble": [E],
                         "orc_run_normalize_root": [E], "orc_run_tree_sizes": [E, P(i), P(i), P(i), P(i)], "orc_run_tree_get": [E, P(_FlatTreeC), P(C.c_uint8)],
                         "orc_run_tree_check": [E, C.c_char_p, i]}.items():
                f = getattr(self.L, n); f.argtypes = a; f.restype = C.c_int
            self.L.orc_run_last_error.argtypes = [E]; self.L.orc_run_last_error.restype = C.c_char_p
            self.L._run_sigs = True
        self.num_sites = int(np.asarray(ref).shape[0])
        ref = np.ascontiguousarray(ref, np.uint8)
        v = tree.c_view()
        self.h = E()
        assert self.L.orc_run_create(C.byref(v), _ptr(ref, C.c_uint8), self.num_sites, int(seed), int(num_parts), C.byref(self.h)) == 0

    def close(self):
        if self.h:
            self.L.orc_run_destroy(self.h); self.h = None

    def _ck(self, st, what):
        if st != 0:
            raise RuntimeError("%s failed: %s" % (what, self.L.orc_run_last_error(self.h).decode()))

    def repartition(self):
        self._ck(self.L.orc_run_repartition(self.h), "run_repartition")

    def num_parts(self):
        n, r = C.c_int(), C.c_int()
        self.L.orc_run_num_parts(self.h, C.byref(n), C.byref(r))
        return n.value, r.value

    def part(self, p):
        """(subtree the Subrun is made from, subtree node -> tree node, cut point)"""
        n, nm, ni, nf = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._ck(self.L.orc_run_part_sizes(self.h, p, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf)), "run_part_sizes")
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view(); orig = np.zeros(n.value, np.int32); cut = C.c_int()
        self._ck(self.L.orc_run_part_get(self.h, p, C.byref(v), _ptr(orig, C.c_int), C.byref(cut)), "run_part_get")
        t.root = v.root
        return t.trimmed(), orig, cut.value

    def part_put(self, p, subtree: FlatTree):
        v = subtree.c_view()
        self._ck(self.L.orc_run_part_put(self.h, p, C.byref(v)), "run_part_put")

    def reassemble(self):
        self._ck(self.L.orc_run_reassemble(self.h), "run_reassemble")

    def normalize_root(self):
        self._ck(self.L.orc_run_normalize_root(self.h), "run_normalize_root")

    def tree(self):
        n, nm, ni, nf = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.L.orc_run_tree_sizes(self.h, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf))
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view(); ref = np.zeros(self.num_sites, np.uint8)
        self._ck(self.L.orc_run_tree_get(self.h, C.byref(v), _ptr(ref, C.c_uint8)), "run_tree_get")
        t.root = v.root
        return t.trimmed(), ref

    def check(self):
        buf = C.create_string_buffer(512)
        rc = self.L.orc_run_tree_check(self.h, buf, 512)
        return rc, buf.value.decode()


class OracleBuild:
    """oracle/orc_build.hpp: the reference's build_usher_like_tree (with fix_up_missations, pseudo_date, randomize_mutation_times)
    restated, the tip descriptors of an existing tree, and the reference's closing checks of the builder."""

    def __init__(self, ref):
        from delphy_amd.engine import _TipDescsC
        self.L = lib()
        E = C.c_void_p
        if not hasattr(self.L, "_build_sigs"):
            i = C.c_int; P = C.POINTER
            for n, a in {"orc_build_create": [P(C.c_uint8), i, P(E)], "orc_build_destroy": [E],
                         "orc_build_descs_from_tree": [E, P(_FlatTreeC), P(i), P(i), P(i)],
                         "orc_build_descs_get": [E, P(C.c_float), P(C.c_float), P(i), P(i), P(C.c_uint8), P(i), P(i), P(i)],
                         "orc_build_usher_like": [E, P(_TipDescsC), C.c_uint64, P(i), P(i), P(i), P(i)],
                         "orc_build_default": [E, P(_TipDescsC), C.c_uint64, P(i), P(i), P(i), P(i), P(i)], "orc_build_tree_get": [E, P(_FlatTreeC)],
                         "orc_build_check": [E, P(_FlatTreeC), P(_TipDescsC), C.c_char_p, i], "orc_build_tree_ref": [E, P(C.c_uint8)],
                         "orc_build_check_with_ref": [E, P(_FlatTreeC), P(C.c_uint8), P(_TipDescsC), C.c_char_p, i]}.items():
                f = getattr(self.L, n); f.argtypes = a; f.restype = C.c_int
            self.L.orc_build_last_error.argtypes = [E]; self.L.orc_build_last_error.restype = C.c_char_p
            self.L._build_sigs = True
        ref = np.ascontiguousarray(ref, np.uint8)
        self.num_sites = int(ref.shape[0])
        self.h = E()
        assert self.L.orc_build_create(_ptr(ref, C.c_uint8), ref.shape[0], C.byref(self.h)) == 0

    def close(self):
        if self.h:
            self.L.orc_build_destroy(self.h); self.h = None

    def _ck(self, st, what):
        if st != 0:
            raise RuntimeError("%s failed: %s" % (what, self.L.orc_build_last_error(self.h).decode()))

    def tip_descs_of(self, tree: FlatTree):
        from delphy_amd.engine import TipDescs
        v = tree.c_view(); n, nd, ni = C.c_int(), C.c_int(), C.c_int()
        self._ck(self.L.orc_build_descs_from_tree(self.h, C.byref(v), C.byref(n), C.byref(nd), C.byref(ni)), "descs_from_tree")
        tmin, tmax = np.zeros(n.value, np.float32), np.zeros(n.value, np.float32)
        doff, dsite, dto = np.zeros(n.value + 1, np.int32), np.zeros(max(nd.value, 1), np.int32), np.zeros(max(nd.value, 1), np.uint8)
        moff, ms, me = np.zeros(n.value + 1, np.int32), np.zeros(max(ni.value, 1), np.int32), np.zeros(max(ni.value, 1), np.int32)
        self._ck(self.L.orc_build_descs_get(self.h, _ptr(tmin, C.c_float), _ptr(tmax, C.c_float), _ptr(doff, C.c_int), _ptr(dsite, C.c_int), _ptr(dto, C.c_uint8),
                                            _ptr(moff, C.c_int), _ptr(ms, C.c_int), _ptr(me, C.c_int)), "descs_get")
        return TipDescs(tmin, tmax, doff, dsite[: nd.value], dto[: nd.value], moff, ms[: ni.value], me[: ni.value])

    def build_usher_like(self, tips, seed) -> FlatTree:
        td = tips.c_struct(); n, nm, ni, nf = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self._ck(self.L.orc_build_usher_like(self.h, C.byref(td), int(seed), C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf)), "build_usher_like")
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view()
        self._ck(self.L.orc_build_tree_get(self.h, C.byref(v)), "build_tree_get")
        t.root = v.root
        return t.trimmed()

    def build_default(self, tips, seed):
        """oracle/orc_utree.hpp: the reference's default builder (build_initial_phylo_tree).  Returns (tree, ref, report): the tree is
        written against `ref`, the ROOT's sequence (the reference re-references there); report = dict(guide_deltas, refined_deltas,
        spr_deltas, rooting)."""
        td = tips.c_struct(); n, nm, ni, nf = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        rep = (C.c_int * 4)()
        self._ck(self.L.orc_build_default(self.h, C.byref(td), int(seed), C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf), rep), "build_default")
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view()
        self._ck(self.L.orc_build_tree_get(self.h, C.byref(v)), "build_tree_get")
        t.root = v.root
        ref = np.zeros(self.num_sites, np.uint8)
        self._ck(self.L.orc_build_tree_ref(self.h, _ptr(ref, C.c_uint8)), "build_tree_ref")
        return t.trimmed(), ref, dict(guide_deltas=rep[0], refined_deltas=rep[1], spr_deltas=rep[2], rooting="regression" if rep[3] == 0 else "midpoint")

    def check(self, tree: FlatTree, tips, ref=None):
        """`ref`: the sequence the tree is written against when that is not the one the descriptors are deltas to."""
        v = tree.c_view(); td = tips.c_struct(); buf = C.create_string_buffer(512)
        if ref is not None:
            ref = np.ascontiguousarray(ref, np.uint8); assert ref.shape[0] == self.num_sites
            rc = self.L.orc_build_check_with_ref(self.h, C.byref(v), _ptr(ref, C.c_uint8), C.byref(td), buf, 512)
            return rc, buf.value.decode()
        rc = self.L.orc_build_check(self.h, C.byref(v), C.byref(td), buf, 512)
        return rc, buf.value.decode()
