"""ctypes mirror of include/emat_backend.h and include/emat_host.h.

Names and argument meaning follow the reference interface each call replaces (`Subrun` as driven by
`Run`, /root/reference core/subrun.h:16-135 and core/run.cpp:110-293,610-693); see the headers for the
file:line of every entry point.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_LIB_NAME = "libemat_hip.so"


class EmatError(RuntimeError):
    pass


STATUS_NAMES = {
    0: "EMAT_OK", 1: "EMAT_ERR_INVALID_ARGUMENT", 2: "EMAT_ERR_NO_DEVICE", 3: "EMAT_ERR_HIP", 4: "EMAT_ERR_STATE",
    5: "EMAT_ERR_CAPACITY", 6: "EMAT_ERR_INTERNAL", 7: "EMAT_ERR_BUFFER_TOO_SMALL", 8: "EMAT_ERR_IO",
}


def library_path() -> str:
    return os.environ.get("EMAT_LIB_PATH") or os.path.join(_HERE, _LIB_NAME)


_DEVSRC = ("emat_backend.hip", "emat_device_core.hpp", "emat_device_moves.hpp", "emat_device_spr.hpp", "emat_slab.hpp", "emat_gtree_kernels.hpp", "emat_build.hpp", "Makefile")


def source_build_id() -> str:
    """The id csrc/Makefile would stamp into a library built from the sources as they are now (same files, same order)."""
    import hashlib
    h = hashlib.sha256()
    for f in _DEVSRC:
        h.update(open(os.path.join(_CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def library_build_id() -> str:
    """emat_build_id() of the loaded library: the device sources it was compiled from."""
    return load_library().emat_build_id().decode()


def build_library(force: bool = False) -> str:
    """Compile the HIP extension in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    path = library_path()
    if force and os.path.exists(path):
        os.remove(path)
    subprocess.run(["make", "-C", _CSRC], check=True, capture_output=not bool(os.environ.get("EMAT_VERBOSE_BUILD")))
    if not os.path.exists(path):
        raise EmatError("building %s failed" % _LIB_NAME)
    return path


class _FlatTreeC(C.Structure):
    _fields_ = [
        ("num_nodes", C.c_int32), ("root", C.c_int32),
        ("parent", C.POINTER(C.c_int32)), ("child0", C.POINTER(C.c_int32)), ("child1", C.POINTER(C.c_int32)),
        ("t", C.POINTER(C.c_double)), ("t_min", C.POINTER(C.c_float)), ("t_max", C.POINTER(C.c_float)),
        ("mut_offset", C.POINTER(C.c_int32)), ("mut_site", C.POINTER(C.c_int32)), ("mut_from", C.POINTER(C.c_uint8)),
        ("mut_to", C.POINTER(C.c_uint8)), ("mut_t", C.POINTER(C.c_double)),
        ("miss_offset", C.POINTER(C.c_int32)), ("miss_start", C.POINTER(C.c_int32)), ("miss_end", C.POINTER(C.c_int32)),
        ("mfs_offset", C.POINTER(C.c_int32)), ("mfs_site", C.POINTER(C.c_int32)), ("mfs_state", C.POINTER(C.c_uint8)),
        ("cap_muts", C.c_int32), ("cap_intervals", C.c_int32), ("cap_from_states", C.c_int32),
    ]


class _PopModelC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("p", C.c_double * 4), ("skygrid_type", C.c_int32), ("skygrid_num_knots", C.c_int32),
                ("skygrid_x", C.POINTER(C.c_double)), ("skygrid_gamma", C.POINTER(C.c_double))]


class _TipDescsC(C.Structure):
    _fields_ = [("num_tips", C.c_int32), ("t_min", C.POINTER(C.c_float)), ("t_max", C.POINTER(C.c_float)),
                ("delta_offset", C.POINTER(C.c_int32)), ("delta_site", C.POINTER(C.c_int32)), ("delta_to", C.POINTER(C.c_uint8)),
                ("miss_offset", C.POINTER(C.c_int32)), ("miss_start", C.POINTER(C.c_int32)), ("miss_end", C.POINTER(C.c_int32))]


class TipDescs:
    """The builder's input (include/emat_backend.h: emat_tip_descs; reference Tip_desc): per tip a date range, its differences
    from the reference sequence (CSR, ascending sites) and its missing intervals (CSR)."""

    def __init__(self, t_min, t_max, delta_offset, delta_site, delta_to, miss_offset, miss_start, miss_end):
        self.t_min = np.ascontiguousarray(t_min, np.float32); self.t_max = np.ascontiguousarray(t_max, np.float32)
        self.delta_offset = np.ascontiguousarray(delta_offset, np.int32); self.delta_site = np.ascontiguousarray(delta_site, np.int32)
        self.delta_to = np.ascontiguousarray(delta_to, np.uint8)
        self.miss_offset = np.ascontiguousarray(miss_offset, np.int32); self.miss_start = np.ascontiguousarray(miss_start, np.int32)
        self.miss_end = np.ascontiguousarray(miss_end, np.int32)
        self.num_tips = int(self.t_min.shape[0])

    def c_struct(self) -> "_TipDescsC":
        return _TipDescsC(self.num_tips, _ptr(self.t_min, C.c_float), _ptr(self.t_max, C.c_float), _ptr(self.delta_offset, C.c_int32), _ptr(self.delta_site, C.c_int32),
                          _ptr(self.delta_to, C.c_uint8), _ptr(self.miss_offset, C.c_int32), _ptr(self.miss_start, C.c_int32), _ptr(self.miss_end, C.c_int32))


class _ConfigC(C.Structure):
    _fields_ = [("device", C.c_int32), ("num_sites", C.c_int32), ("max_parts", C.c_int32), ("slab_slack", C.c_double),
                ("trace_moves", C.c_int32), ("use_lds", C.c_int32)]


class _PartStatsC(C.Structure):
    _fields_ = [("status", C.c_int32), ("num_nodes", C.c_int32), ("moves_done", C.c_int64), ("proposed", C.c_int64 * 5),
                ("accepted", C.c_int64 * 5), ("algorithmic_bytes", C.c_int64), ("rng_draws", C.c_int64), ("device_ticks", C.c_int64), ("algorithmic_write_bytes", C.c_int64)]


class _SynthParamsC(C.Structure):
    _fields_ = [("num_tips", C.c_int32), ("num_sites", C.c_int32), ("tip_span", C.c_double), ("tip_date_uncertainty", C.c_double),
                ("frac_uncertain_tips", C.c_double), ("pop_n0", C.c_double), ("pop_growth", C.c_double), ("mu", C.c_double),
                ("kappa", C.c_double), ("pi", C.c_double * 4), ("gaps_per_tip", C.c_int32), ("mean_gap_len", C.c_double), ("seed", C.c_uint64)]


def _ptr(a: np.ndarray, ct):
    return a.ctypes.data_as(C.POINTER(ct))


@dataclass
class FlatTree:
    """numpy owner of an `emat_flat_tree` (struct-of-arrays + CSR lists; include/emat_backend.h)."""
    root: int
    parent: np.ndarray
    child0: np.ndarray
    child1: np.ndarray
    t: np.ndarray
    t_min: np.ndarray
    t_max: np.ndarray
    mut_offset: np.ndarray
    mut_site: np.ndarray
    mut_from: np.ndarray
    mut_to: np.ndarray
    mut_t: np.ndarray
    miss_offset: np.ndarray
    miss_start: np.ndarray
    miss_end: np.ndarray
    mfs_offset: np.ndarray
    mfs_site: np.ndarray
    mfs_state: np.ndarray

    @property
    def num_nodes(self) -> int:
        return int(self.parent.shape[0])

    @staticmethod
    def empty(n: int, nm: int, ni: int, nf: int) -> "FlatTree":
        i32, f64, f32, u8 = np.int32, np.float64, np.float32, np.uint8
        return FlatTree(-1, np.full(n, -1, i32), np.full(n, -1, i32), np.full(n, -1, i32), np.zeros(n, f64), np.zeros(n, f32), np.zeros(n, f32),
                        np.zeros(n + 1, i32), np.zeros(max(nm, 1), i32), np.zeros(max(nm, 1), u8), np.zeros(max(nm, 1), u8), np.zeros(max(nm, 1), f64),
                        np.zeros(n + 1, i32), np.zeros(max(ni, 1), i32), np.zeros(max(ni, 1), i32),
                        np.zeros(n + 1, i32), np.zeros(max(nf, 1), i32), np.zeros(max(nf, 1), u8))

    def c_view(self) -> _FlatTreeC:
        v = _FlatTreeC()
        v.num_nodes = self.num_nodes
        v.root = self.root
        v.parent, v.child0, v.child1 = _ptr(self.parent, C.c_int32), _ptr(self.child0, C.c_int32), _ptr(self.child1, C.c_int32)
        v.t, v.t_min, v.t_max = _ptr(self.t, C.c_double), _ptr(self.t_min, C.c_float), _ptr(self.t_max, C.c_float)
        v.mut_offset, v.mut_site = _ptr(self.mut_offset, C.c_int32), _ptr(self.mut_site, C.c_int32)
        v.mut_from, v.mut_to, v.mut_t = _ptr(self.mut_from, C.c_uint8), _ptr(self.mut_to, C.c_uint8), _ptr(self.mut_t, C.c_double)
        v.miss_offset, v.miss_start, v.miss_end = _ptr(self.miss_offset, C.c_int32), _ptr(self.miss_start, C.c_int32), _ptr(self.miss_end, C.c_int32)
        v.mfs_offset, v.mfs_site, v.mfs_state = _ptr(self.mfs_offset, C.c_int32), _ptr(self.mfs_site, C.c_int32), _ptr(self.mfs_state, C.c_uint8)
        v.cap_muts, v.cap_intervals, v.cap_from_states = self.mut_site.shape[0], self.miss_start.shape[0], self.mfs_site.shape[0]
        return v

    def trimmed(self) -> "FlatTree":
        """Drop the padding of the variable-length arrays (after a download)."""
        n = self.num_nodes
        nm, ni, nf = int(self.mut_offset[n]), int(self.miss_offset[n]), int(self.mfs_offset[n])
        return FlatTree(self.root, self.parent, self.child0, self.child1, self.t, self.t_min, self.t_max,
                        self.mut_offset, self.mut_site[:nm].copy(), self.mut_from[:nm].copy(), self.mut_to[:nm].copy(), self.mut_t[:nm].copy(),
                        self.miss_offset, self.miss_start[:ni].copy(), self.miss_end[:ni].copy(),
                        self.mfs_offset, self.mfs_site[:nf].copy(), self.mfs_state[:nf].copy())

    @staticmethod
    def from_c_view(v: _FlatTreeC) -> "FlatTree":
        n = v.num_nodes
        def arr(p, cnt, dt):
            return np.ctypeslib.as_array(p, shape=(max(cnt, 1),)).astype(dt, copy=True)[:cnt] if cnt > 0 else np.zeros(0, dt)
        mo = arr(v.mut_offset, n + 1, np.int32)
        io = arr(v.miss_offset, n + 1, np.int32)
        fo = arr(v.mfs_offset, n + 1, np.int32)
        nm, ni, nf = int(mo[n]), int(io[n]), int(fo[n])
        def padded(a, dt):
            return a if a.shape[0] > 0 else np.zeros(1, dt)
        return FlatTree(v.root, arr(v.parent, n, np.int32), arr(v.child0, n, np.int32), arr(v.child1, n, np.int32),
                        arr(v.t, n, np.float64), arr(v.t_min, n, np.float32), arr(v.t_max, n, np.float32),
                        mo, padded(arr(v.mut_site, nm, np.int32), np.int32), padded(arr(v.mut_from, nm, np.uint8), np.uint8),
                        padded(arr(v.mut_to, nm, np.uint8), np.uint8), padded(arr(v.mut_t, nm, np.float64), np.float64),
                        io, padded(arr(v.miss_start, ni, np.int32), np.int32), padded(arr(v.miss_end, ni, np.int32), np.int32),
                        fo, padded(arr(v.mfs_site, nf, np.int32), np.int32), padded(arr(v.mfs_state, nf, np.uint8), np.uint8))


@dataclass
class PopModel:
    """Population model descriptor (reference core/pop_model.h): kind 0 const{pop}, 1 exp{t0,n0,g,min_pop}, 2 skygrid."""
    kind: int = 0
    p: Sequence[float] = (1.0, 0.0, 0.0, 0.0)
    skygrid_type: int = 1
    skygrid_x: Optional[np.ndarray] = None
    skygrid_gamma: Optional[np.ndarray] = None
    _keep: list = field(default_factory=list, repr=False)

    @staticmethod
    def const(pop: float) -> "PopModel":
        return PopModel(0, (pop, 0.0, 0.0, 0.0))

    @staticmethod
    def exp(t0: float, n0: float, g: float, min_pop: float = 0.0) -> "PopModel":
        return PopModel(1, (t0, n0, g, min_pop))

    @staticmethod
    def skygrid(x, gamma, log_linear: bool = False) -> "PopModel":
        return PopModel(2, (0.0, 0.0, 0.0, 0.0), 2 if log_linear else 1, np.ascontiguousarray(x, np.float64), np.ascontiguousarray(gamma, np.float64))

    def c_struct(self) -> _PopModelC:
        m = _PopModelC()
        m.kind = self.kind
        for i in range(4):
            m.p[i] = float(self.p[i])
        m.skygrid_type = self.skygrid_type
        if self.kind == 2:
            m.skygrid_num_knots = int(self.skygrid_x.shape[0])
            m.skygrid_x = _ptr(self.skygrid_x, C.c_double)
            m.skygrid_gamma = _ptr(self.skygrid_gamma, C.c_double)
        return m


@dataclass
class SynthParams:
    num_tips: int = 100
    num_sites: int = 30000
    tip_span: float = 365.0
    tip_date_uncertainty: float = 0.0
    frac_uncertain_tips: float = 0.0
    pop_n0: float = 365.0
    pop_growth: float = 0.0
    mu: float = 1e-3 / 365.0
    kappa: float = 5.0
    pi: Sequence[float] = (0.31, 0.19, 0.21, 0.29)
    gaps_per_tip: int = 2
    mean_gap_len: float = 150.0
    seed: int = 20261001


_lib = None


def load_library():
    """Load libemat_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise EmatError("%s is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback" % path)
    lib = C.CDLL(path)
    B, R, S = C.c_void_p, C.c_void_p, C.c_void_p
    i32, i64, u64, dbl = C.c_int32, C.c_int64, C.c_uint64, C.c_double
    P = C.POINTER
    sigs = {
        "emat_backend_create": [P(_ConfigC), P(B)], "emat_backend_destroy": [B],
        "emat_set_ref_sequence": [B, P(C.c_uint8), i32],
        "emat_set_evo": [B, i32, P(dbl), P(dbl), P(dbl), P(dbl), P(i32)],
        "emat_set_flags": [B, dbl, i32, i32],
        "emat_begin_upload": [B, i32], "emat_part_upload": [B, i32, P(_FlatTreeC), i32, u64], "emat_end_upload": [B],
        "emat_build_coalescent_parts": [B, P(_PopModelC), i32, dbl],
        "emat_coalescent_begin": [B, P(_PopModelC), i32, dbl, P(dbl), P(dbl)], "emat_coalescent_set_range": [B, dbl, dbl, P(i32)],
        "emat_coalescent_local_grid": [B, P(dbl), P(i32)], "emat_coalescent_sample": [B, P(dbl), P(i32), P(dbl)], "emat_coalescent_finish": [B, P(dbl)],
        "emat_run_local_moves": [B, i64], "emat_run_moves_per_part": [B, i64], "emat_synchronize": [B], "emat_recalc_derived": [B],
        "emat_get_totals": [B, P(dbl), P(dbl)], "emat_get_global_stats": [B, i32, P(dbl), P(i64), P(i64)],
        "emat_part_get_sizes": [B, i32, P(i32), P(i32), P(i32), P(i32)], "emat_part_download": [B, i32, P(_FlatTreeC)],
        "emat_part_get_derived": [B, i32, P(dbl), P(i32), P(dbl), P(dbl)], "emat_part_get_state_frequencies": [B, i32, P(i32), P(i32)],
        "emat_part_get_coalescent": [B, i32, P(i32), P(dbl), P(dbl), P(dbl), P(dbl), P(i32), P(dbl), P(dbl)],
        "emat_tree_build_usher_like": [B, P(_TipDescsC), u64], "emat_tree_build_default": [B, P(_TipDescsC), u64, P(i32)], "emat_tree_built_ref": [B, P(C.c_uint8)], "emat_tree_built_sizes": [B, P(i32), P(i32), P(i32), P(i32)], "emat_tree_built_get": [B, P(_FlatTreeC)],
        "emat_part_get_rng": [B, i32, P(u64), P(u64), P(u64), P(i32)], "emat_check_derived": [B, dbl, P(i32), P(dbl)], "emat_debug_slab_layout": [B, i32, P(C.c_uint32)],
        "emat_part_get_stats": [B, i32, P(_PartStatsC)], "emat_part_get_trace": [B, i32, P(i32), P(dbl)],
        "emat_last_run_ms": [B, P(dbl)], "emat_last_kernel_ms": [B, P(dbl), P(i32)],
        "emat_debug_gamma": [B, i32, i32, P(dbl), P(dbl), P(dbl)],
        "emat_debug_pop": [B, P(_PopModelC), i32, i32, P(dbl), P(dbl), P(dbl)], "emat_debug_interval_op": [B, i32, P(i32), i32, P(i32), i32, P(i32), P(i32)],
        "emat_debug_tree_query": [B, i32, i32, i32, P(i32), P(i32), P(i32)],
        "emat_set_option": [B, C.c_char_p, C.c_char_p], "emat_set_host_threads": [i32],
        "emat_debug_graft": [B, i32, i32, dbl, i32, i32, dbl, P(dbl), i32, P(i32)], "emat_debug_edit": [B, i32, i32, i32, P(i32), P(i32), P(dbl)],
        "emat_debug_sample_history": [B, i32, i32, P(i32), P(dbl), P(C.c_uint8), dbl, dbl, P(i32), P(dbl), i32, P(i32)],
        "emat_get_num_muts_l": [B, P(i32)], "emat_get_scalable_coalescent_log_prior": [B, dbl, dbl, P(dbl)],
        "emat_scalable_coalescent_partial": [B, dbl, dbl, i32, i32, P(dbl), P(dbl), P(i32)],
        "emat_scalable_coalescent_log_prior": [B, dbl, dbl, i32, i32, P(dbl), dbl, P(dbl)],
        "emat_synth_create": [P(_SynthParamsC), P(S)], "emat_synth_get": [S, P(_FlatTreeC), P(P(C.c_uint8)), P(dbl)],
        "emat_run_create": [B, P(_FlatTreeC), P(C.c_uint8), i32, u64, P(R)], "emat_run_destroy": [R],
        "emat_run_set_num_parts": [R, i32], "emat_run_set_max_part_nodes": [R, i32], "emat_run_partition_stats": [R, P(i32), P(i32), P(i32), P(i32)], "emat_run_debug_redraw_partition": [R, P(i32), P(i32)], "emat_run_follow_draws": [R, R], "emat_run_draw_partition": [R], "emat_run_set_hky": [R, dbl, dbl, P(dbl), P(dbl)], "emat_run_set_pop_model": [R, P(_PopModelC)],
        "emat_run_set_coalescent_t_step": [R, dbl], "emat_run_set_flags": [R, i32, i32],
        "emat_run_repartition": [R], "emat_run_num_parts": [R, P(i32), P(i32)],
        "emat_run_part_sizes": [R, i32, P(i32), P(i32), P(i32), P(i32)], "emat_run_part_get": [R, i32, P(_FlatTreeC), P(i32), P(u64)],
        "emat_run_part_put": [R, i32, P(_FlatTreeC)], "emat_run_push_params": [R], "emat_run_moves": [R, i64], "emat_run_reassemble": [R],
        "emat_run_set_shard": [R, i32, i32], "emat_run_shard_range": [R, P(i32), P(i32), P(i32)], "emat_run_coalescent_begin": [R, P(dbl), P(dbl)],
        "emat_run_moves_sharded": [R, i64], "emat_run_pack_local_parts": [R, P(C.c_uint8), u64, P(u64)], "emat_run_unpack_parts": [R, P(C.c_uint8), u64],
        "emat_run_moves_split": [B, i64, i64], "emat_run_get_Ttwiddle_l": [R, P(dbl)], "emat_run_Ttwiddle_ext": [R, P(dbl), P(i32), P(i32), P(dbl), i32, P(i32)],
        "emat_get_part_tree_lengths": [B, P(dbl)], "emat_Ttwiddle_l_partial": [B, P(i32), P(i32), P(dbl), P(dbl), P(dbl), P(dbl)],
        "emat_Ttwiddle_l_finish": [B, P(dbl), P(dbl), dbl, P(dbl)],
        "emat_run_do_mcmc_steps": [R, i64, i64], "emat_run_tree_sizes": [R, P(i32), P(i32), P(i32), P(i32)],
        "emat_run_tree_get": [R, P(_FlatTreeC), P(C.c_uint8)], "emat_run_t_max_tip": [R, P(dbl)],
        "emat_run_set_device_tree": [R, i32], "emat_run_moves_even": [B, i64, i32], "emat_debug_tree_counters": [B, P(i32)],
        "emat_tree_upload": [B, P(_FlatTreeC)], "emat_tree_get_sizes": [B, P(i32), P(i32), P(i32), P(i32)], "emat_tree_download": [B, P(_FlatTreeC), P(C.c_uint8)],
        "emat_tree_get_topology": [B, P(i32), P(i32), P(i32), P(dbl), P(i32)],
        "emat_tree_get_kids": [B, P(P(i32)), P(i32), P(i32), P(dbl)],
        "emat_tree_repartition": [B, i32, P(i32), P(i32), P(i32), P(i32), i32, P(u64), P(_PopModelC), dbl],
        "emat_tree_reassemble": [B, P(i32), P(i32), P(C.c_uint8), P(C.c_uint8), i32],
        "emat_tree_partition": [B, i32, P(i32), P(i32), P(i32), P(i32)], "emat_tree_get_partition": [B, P(i32), P(i32), P(i32), P(i32)],
        "emat_tree_repartition_range": [B, i32, P(i32), P(i32), P(i32), P(i32), i32, P(u64), P(_PopModelC), dbl, i32, i32],
        "emat_tree_get_root_deltas": [B, P(i32), P(i32), P(C.c_uint8), P(C.c_uint8), i32], "emat_tree_gather_local": [B, i32, P(i32), P(C.c_uint8), P(C.c_uint8)],
        "emat_tree_export_nodes": [B, P(C.c_uint8), u64, P(u64)], "emat_tree_apply_nodes": [B, P(C.c_uint8), u64], "emat_tree_reassemble_end": [B],
        "emat_run_note_device_reassembled": [R, i32, P(i32), P(C.c_uint8)], "emat_run_set_paranoid": [R, i32], "emat_run_set_reference_remainder": [R, i32],
    }
    M = C.c_void_p
    sigs.update({
        "emat_run_create_multi": [P(i32), i32, P(_ConfigC), P(_FlatTreeC), P(C.c_uint8), i32, u64, i32, P(M)], "emat_multi_destroy": [M],
        "emat_multi_set_num_parts": [M, i32], "emat_multi_set_max_part_nodes": [M, i32], "emat_multi_set_option": [M, C.c_char_p, C.c_char_p], "emat_multi_set_rccl_library": [C.c_char_p], "emat_multi_debug_rccl_load": [C.c_char_p, i32], "emat_multi_set_hky": [M, dbl, dbl, P(dbl), P(dbl)], "emat_multi_set_pop_model": [M, P(_PopModelC)],
        "emat_multi_set_coalescent_t_step": [M, dbl], "emat_multi_set_flags": [M, i32, i32], "emat_multi_set_paranoid": [M, i32],
        "emat_multi_repartition": [M], "emat_multi_run_moves": [M, i64], "emat_multi_check_derived": [M, dbl], "emat_multi_reassemble": [M],
        "emat_multi_get_totals": [M, P(dbl), P(dbl)], "emat_multi_do_mcmc_steps": [M, i64, i64],
        "emat_multi_tree_sizes": [M, P(i32), P(i32), P(i32), P(i32)], "emat_multi_tree_get": [M, i32, P(_FlatTreeC), P(C.c_uint8)],
    })
    for name, args in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.emat_last_error.argtypes = [B]
    lib.emat_last_error.restype = C.c_char_p
    if os.environ.get("EMAT_HOST_THREADS"):     # (the library reads no tuning from the environment: this mirror forwards it)
        lib.emat_set_host_threads(int(os.environ["EMAT_HOST_THREADS"]))
    lib.emat_build_id.argtypes = []
    lib.emat_build_id.restype = C.c_char_p
    lib.emat_run_last_error.argtypes = [R]
    lib.emat_run_last_error.restype = C.c_char_p
    lib.emat_multi_last_error.argtypes = [M]; lib.emat_multi_last_error.restype = C.c_char_p
    lib.emat_multi_exchange.argtypes = [M]; lib.emat_multi_exchange.restype = C.c_char_p
    lib.emat_multi_num_shards.argtypes = [M]; lib.emat_multi_num_shards.restype = C.c_int32
    lib.emat_multi_backend.argtypes = [M, i32]; lib.emat_multi_backend.restype = C.c_void_p
    lib.emat_multi_shard.argtypes = [M, i32]; lib.emat_multi_shard.restype = C.c_void_p
    lib.emat_synth_destroy.argtypes = [S]
    lib.emat_synth_destroy.restype = None
    _lib = lib
    return lib


def hky_q_matrix(kappa: float, pi: Sequence[float]) -> np.ndarray:
    """Normalised HKY rate matrix, q_ab = r_ab pi_b / (pi^T r pi) (reference core/evo_hky.cpp:7-50)."""
    pi = np.asarray(pi, np.float64)
    r = np.array([[0, 1, kappa, 1], [1, 0, 1, kappa], [kappa, 1, 0, 1], [1, kappa, 1, 0]], np.float64)
    rowv = np.array([sum(pi[a] * r[a][b] for a in range(4)) for b in range(4)])
    R = 0.0
    for b in range(4):
        R += rowv[b] * pi[b]
    q = np.zeros((4, 4))
    for a in range(4):
        for b in range(4):
            if a != b:
                q[a, b] = r[a, b] / R * pi[b]
                q[a, a] -= q[a, b]
    return q


def make_synthetic_emat(p: SynthParams):
    """Seeded synthetic EMAT.  Returns (FlatTree, ref_sequence uint8[L], t_max_tip)."""
    lib = load_library()
    sp = _SynthParamsC()
    for f in ("num_tips", "num_sites", "tip_span", "tip_date_uncertainty", "frac_uncertain_tips", "pop_n0", "pop_growth", "mu", "kappa",
              "gaps_per_tip", "mean_gap_len", "seed"):
        setattr(sp, f, getattr(p, f))
    for a in range(4):
        sp.pi[a] = float(p.pi[a])
    h = C.c_void_p()
    st = lib.emat_synth_create(C.byref(sp), C.byref(h))
    if st != 0:
        raise EmatError("emat_synth_create failed: %s" % STATUS_NAMES.get(st, st))
    try:
        v = _FlatTreeC()
        ref = C.POINTER(C.c_uint8)()
        tmax = C.c_double()
        lib.emat_synth_get(h, C.byref(v), C.byref(ref), C.byref(tmax))
        tree = FlatTree.from_c_view(v)
        refseq = np.ctypeslib.as_array(ref, shape=(p.num_sites,)).copy()
        return tree, refseq, float(tmax.value)
    finally:
        lib.emat_synth_destroy(h)


def decode_graft_output(v, mode: int) -> dict:
    """The doubles of emat_debug_graft (include/emat_backend.h) as {"status", "grafts": [graft...], "count_min_mutations",
    "count_closed_mutations", "closed_deltas"}; a graft is {"delta_log_G", "log_alpha_mut", "X", "S", "t_P", "branch_infos": [...]}
    with mutations as [from, site, to, t] and deltas as [site, from, to]."""
    pos = [0]
    def take():
        x = float(v[pos[0]]); pos[0] += 1; return x
    def graft():
        nbi = int(take()); g = {"delta_log_G": take(), "log_alpha_mut": take(), "X": int(take()), "S": int(take()), "t_P": take(), "branch_infos": []}
        for _ in range(nbi):
            b = {"A": int(take()), "B": int(take()), "is_open": take() != 0.0, "T_to_X": take(), "partial_lambda_at_A": take(), "partial_lambda_at_X": take()}
            b["warm_sites"] = [[int(take()), int(take())] for _ in range(int(take()))]
            b["hot_sites"] = [[int(take()), int(take())] for _ in range(int(take()))]
            muts = []
            for _ in range(int(take())):
                site, fr, to, t = int(take()), int(take()), int(take()), take()
                muts.append([fr, site, to, t])
            b["hot_muts_to_X"] = muts
            b["hot_deltas_to_X"] = [[int(take()), int(take()), int(take())] for _ in range(int(take()))]
            g["branch_infos"].append(b)
        return g
    out = {"status": int(take())}
    num = int(take())
    out["grafts"] = [graft()]
    if mode >= 1:
        out["count_min_mutations"] = int(take()); out["count_closed_mutations"] = int(take())
        out["closed_deltas"] = [[int(take()), int(take()), int(take())] for _ in range(int(take()))]
    if num == 2:
        out["grafts"].append(graft())
    assert pos[0] == len(v), "emat_debug_graft output not consumed exactly: %d of %d" % (pos[0], len(v))
    return out


# The library itself reads no tuning from the environment (emat_set_option per handle).  For A/B scripts and tests this mirror forwards
# EMAT_<NAME> variables of the process to every handle it creates -- the behaviour the library had built in until round 4.
OPTION_KEYS = ("slack", "heap_per_node", "lds_scratch", "lds_classes", "lds_max", "giants", "side_arena", "tree_host_coalescent", "ticket_taper", "chunks", "ticket_xcd_spread",
               "ticket_release", "ticket_weights", "single_ticket_parts", "parts_per_cu", "order_by_time", "build_blocks", "tree_tight", "fn_min_lists", "phase_extra", "no_uniform_sites")


def _forward_env_options(setter):
    for k in OPTION_KEYS:
        v = os.environ.get("EMAT_" + k.upper())
        if v is not None:
            setter(k, v)


class EmatBackend:
    """The engine behind include/emat_backend.h: one resident `Subrun` per partition part, on the GPU."""

    def __init__(self, num_sites: int, device: int = 0, trace_moves: int = 0, use_lds: bool = True, slab_slack: float = 0.0, max_parts: int = 0):
        self._lib = load_library()
        cfg = _ConfigC(device, num_sites, max_parts, slab_slack, trace_moves, 1 if use_lds else 0)
        self._h = C.c_void_p()
        st = self._lib.emat_backend_create(C.byref(cfg), C.byref(self._h))
        if st != 0:
            raise EmatError("emat_backend_create failed: %s (the engine needs a HIP device; there is no CPU fallback)" % STATUS_NAMES.get(st, st))
        self.num_sites = num_sites
        self._keep = []
        _forward_env_options(lambda k, v: self.set_option(k, v))

    def set_option(self, key: str, value):
        """A tuning / test option of this handle (include/emat_backend.h, emat_set_option), before the first launch."""
        self._ck(self._lib.emat_set_option(self._h, key.encode(), str(value).encode()), "emat_set_option(%s)" % key)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.emat_backend_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def _ck(self, st: int, what: str):
        if st != 0:
            msg = self._lib.emat_last_error(self._h)
            raise EmatError("%s: %s (%s)" % (what, STATUS_NAMES.get(st, st), msg.decode() if msg else ""))

    def set_ref_sequence(self, ref: np.ndarray):
        ref = np.ascontiguousarray(ref, np.uint8)
        self._ck(self._lib.emat_set_ref_sequence(self._h, _ptr(ref, C.c_uint8), ref.shape[0]), "emat_set_ref_sequence")

    def set_evo(self, mu, pi, q, nu_l, partition_for_site):
        mu = np.ascontiguousarray(mu, np.float64).reshape(-1)
        P = mu.shape[0]
        pi = np.ascontiguousarray(pi, np.float64).reshape(P * 4)
        q = np.ascontiguousarray(q, np.float64).reshape(P * 16)
        nu_l = np.ascontiguousarray(nu_l, np.float64)
        pfs = np.ascontiguousarray(partition_for_site, np.int32)
        self._ck(self._lib.emat_set_evo(self._h, P, _ptr(mu, C.c_double), _ptr(pi, C.c_double), _ptr(q, C.c_double), _ptr(nu_l, C.c_double), _ptr(pfs, C.c_int32)), "emat_set_evo")

    def set_hky(self, mu: float, kappa: float, pi, nu_l=None):
        nu = np.ones(self.num_sites) if nu_l is None else nu_l
        self.set_evo([mu], [pi], [hky_q_matrix(kappa, pi)], nu, np.zeros(self.num_sites, np.int32))

    def set_flags(self, t_max_tip: float, only_displacing_inner_nodes: bool = False, topology_moves_enabled: bool = True):
        self._ck(self._lib.emat_set_flags(self._h, t_max_tip, int(only_displacing_inner_nodes), int(topology_moves_enabled)), "emat_set_flags")

    def upload_parts(self, parts: Sequence[FlatTree], includes_run_root: Sequence[bool], seeds: Sequence[int]):
        self._ck(self._lib.emat_begin_upload(self._h, len(parts)), "emat_begin_upload")
        for i, (t, r, s) in enumerate(zip(parts, includes_run_root, seeds)):
            v = t.c_view()
            self._ck(self._lib.emat_part_upload(self._h, i, C.byref(v), int(r), int(s)), "emat_part_upload")
        self._ck(self._lib.emat_end_upload(self._h), "emat_end_upload")

    def build_coalescent_parts(self, pop: PopModel, root_part_index: int, t_step: float):
        m = pop.c_struct()
        self._ck(self._lib.emat_build_coalescent_parts(self._h, C.byref(m), root_part_index, t_step), "emat_build_coalescent_parts")

    # staged form for parts sharded over several processes / GPUs
    def coalescent_begin(self, pop: PopModel, root_part_index: int, t_step: float):
        m = pop.c_struct()
        lo, hi = C.c_double(), C.c_double()
        self._ck(self._lib.emat_coalescent_begin(self._h, C.byref(m), root_part_index, t_step, C.byref(lo), C.byref(hi)), "emat_coalescent_begin")
        return float(lo.value), float(hi.value)

    def coalescent_set_range(self, all_t_min: float, all_t_max: float) -> int:
        n = C.c_int32()
        self._ck(self._lib.emat_coalescent_set_range(self._h, all_t_min, all_t_max, C.byref(n)), "emat_coalescent_set_range")
        self._coal_cells = n.value
        return n.value

    def coalescent_local_grid(self):
        kb = np.zeros(self._coal_cells)
        na = np.zeros(self._coal_cells, np.int32)
        self._ck(self._lib.emat_coalescent_local_grid(self._h, _ptr(kb, C.c_double), _ptr(na, C.c_int32)), "emat_coalescent_local_grid")
        return kb, na

    def coalescent_sample(self, k_bar: np.ndarray, num_active: np.ndarray) -> np.ndarray:
        k_bar = np.ascontiguousarray(k_bar, np.float64)
        num_active = np.ascontiguousarray(num_active, np.int32)
        kt = np.zeros(self._coal_cells)
        self._ck(self._lib.emat_coalescent_sample(self._h, _ptr(k_bar, C.c_double), _ptr(num_active, C.c_int32), _ptr(kt, C.c_double)), "emat_coalescent_sample")
        return kt

    def coalescent_finish(self, k_twiddle_bar: np.ndarray):
        k = np.ascontiguousarray(k_twiddle_bar, np.float64)
        self._ck(self._lib.emat_coalescent_finish(self._h, _ptr(k, C.c_double)), "emat_coalescent_finish")

    def run_local_moves(self, count: int):
        self._ck(self._lib.emat_run_local_moves(self._h, count), "emat_run_local_moves")

    def run_moves_split(self, moves_per_part: int, extra_moves_part0: int):
        self._ck(self._lib.emat_run_moves_split(self._h, moves_per_part, extra_moves_part0), "emat_run_moves_split")

    def run_moves_per_part(self, moves: int):
        self._ck(self._lib.emat_run_moves_per_part(self._h, moves), "emat_run_moves_per_part")

    def synchronize(self):
        self._ck(self._lib.emat_synchronize(self._h), "emat_synchronize")

    # ---- the whole tree resident in HBM (include/emat_backend.h: emat_tree_*) ----
    def tree_upload(self, tree: FlatTree):
        v = tree.c_view()
        self._ck(self._lib.emat_tree_upload(self._h, C.byref(v)), "emat_tree_upload")

    def tree_topology(self):
        n, nm, ni, nf = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_tree_get_sizes(self._h, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf)), "emat_tree_get_sizes")
        parent, c0, c1 = (np.zeros(n.value, np.int32) for _ in range(3))
        t = np.zeros(n.value); root = C.c_int32()
        self._ck(self._lib.emat_tree_get_topology(self._h, _ptr(parent, C.c_int32), _ptr(c0, C.c_int32), _ptr(c1, C.c_int32), _ptr(t, C.c_double), C.byref(root)), "emat_tree_get_topology")
        return parent, c0, c1, t, int(root.value)

    def tree_kids(self):
        """(child0, child1) of every node, the root and the root's time, from the backend's own mirror (emat_tree_get_kids): current as
        soon as emat_tree_reassemble has returned, while the lists of the tree may still be on their way.  A copy: the mirror itself
        is only valid until the next reassemble."""
        kp = C.POINTER(C.c_int32)(); n = C.c_int32(); root = C.c_int32(); t_root = C.c_double()
        self._ck(self._lib.emat_tree_get_kids(self._h, C.byref(kp), C.byref(n), C.byref(root), C.byref(t_root)), "emat_tree_get_kids")
        kids = np.ctypeslib.as_array(kp, shape=(n.value, 2)).copy()
        return kids, int(root.value), float(t_root.value)

    def tree_repartition(self, part_offset, orig, kid0, kid1, root_part: int, seeds, pop: PopModel, t_step: float):
        po, og, k0, k1 = (np.ascontiguousarray(a, np.int32) for a in (part_offset, orig, kid0, kid1))
        sd = np.ascontiguousarray(seeds, np.uint64)
        m = pop.c_struct()
        self._ck(self._lib.emat_tree_repartition(self._h, int(po.shape[0]) - 1, _ptr(po, C.c_int32), _ptr(og, C.c_int32), _ptr(k0, C.c_int32), _ptr(k1, C.c_int32),
                                                 root_part, _ptr(sd, C.c_uint64), C.byref(m), t_step), "emat_tree_repartition")

    def tree_reassemble(self, capacity: Optional[int] = None):
        capacity = self.num_sites if capacity is None else capacity   # at most one change of the root sequence per site
        n = C.c_int32()
        site = np.zeros(capacity, np.int32); frm = np.zeros(capacity, np.uint8); to = np.zeros(capacity, np.uint8)
        self._ck(self._lib.emat_tree_reassemble(self._h, C.byref(n), _ptr(site, C.c_int32), _ptr(frm, C.c_uint8), _ptr(to, C.c_uint8), capacity), "emat_tree_reassemble")
        k = n.value
        return site[:k].copy(), frm[:k].copy(), to[:k].copy()

    def tree_partition(self, cut_nodes):
        """partition_tree on the device: (num_parts, root_part, part_offset, orig, kid0, kid1) for the given cut nodes."""
        cuts = np.ascontiguousarray(cut_nodes, np.int32)
        n, r = C.c_int32(), C.c_int32()
        sizes = np.zeros(cuts.shape[0] + 1, np.int32)
        self._ck(self._lib.emat_tree_partition(self._h, int(cuts.shape[0]), _ptr(cuts, C.c_int32), C.byref(n), C.byref(r), _ptr(sizes, C.c_int32)), "emat_tree_partition")
        off = np.zeros(n.value + 1, np.int32)
        self._ck(self._lib.emat_tree_get_partition(self._h, _ptr(off, C.c_int32), None, None, None), "emat_tree_get_partition")
        orig, k0, k1 = (np.zeros(int(off[-1]), np.int32) for _ in range(3))
        self._ck(self._lib.emat_tree_get_partition(self._h, None, _ptr(orig, C.c_int32), _ptr(k0, C.c_int32), _ptr(k1, C.c_int32)), "emat_tree_get_partition")
        assert np.array_equal(np.diff(off), sizes[:n.value])
        return n.value, r.value, off, orig, k0, k1

    def tree_root_deltas(self, capacity: Optional[int] = None):
        """(site, from, to) on the process that holds the root part, None elsewhere."""
        capacity = self.num_sites if capacity is None else capacity
        n = C.c_int32()
        site = np.zeros(capacity, np.int32); frm = np.zeros(capacity, np.uint8); to = np.zeros(capacity, np.uint8)
        self._ck(self._lib.emat_tree_get_root_deltas(self._h, C.byref(n), _ptr(site, C.c_int32), _ptr(frm, C.c_uint8), _ptr(to, C.c_uint8), capacity), "emat_tree_get_root_deltas")
        return None if n.value < 0 else (site[:n.value].copy(), frm[:n.value].copy(), to[:n.value].copy())

    def tree_gather_local(self, site, frm, to):
        site = np.ascontiguousarray(site, np.int32); frm = np.ascontiguousarray(frm, np.uint8); to = np.ascontiguousarray(to, np.uint8)
        self._ck(self._lib.emat_tree_gather_local(self._h, int(site.shape[0]), _ptr(site, C.c_int32), _ptr(frm, C.c_uint8), _ptr(to, C.c_uint8)), "emat_tree_gather_local")

    def tree_export_nodes(self) -> np.ndarray:
        need = C.c_uint64()
        self._ck(self._lib.emat_tree_export_nodes(self._h, None, 0, C.byref(need)), "emat_tree_export_nodes")
        buf = np.zeros(int(need.value), np.uint8)
        self._ck(self._lib.emat_tree_export_nodes(self._h, _ptr(buf, C.c_uint8), buf.shape[0], C.byref(need)), "emat_tree_export_nodes")
        return buf

    def tree_apply_nodes(self, buf: np.ndarray):
        buf = np.ascontiguousarray(buf, np.uint8)
        self._ck(self._lib.emat_tree_apply_nodes(self._h, _ptr(buf, C.c_uint8), buf.shape[0]), "emat_tree_apply_nodes")

    # the same exchange on raw addresses -- device memory (e.g. a torch tensor's data_ptr()) or host memory
    def tree_export_size(self) -> int:
        need = C.c_uint64()
        self._ck(self._lib.emat_tree_export_nodes(self._h, None, 0, C.byref(need)), "emat_tree_export_nodes")
        return int(need.value)

    def tree_export_nodes_into(self, address: int, capacity: int) -> int:
        need = C.c_uint64()
        self._ck(self._lib.emat_tree_export_nodes(self._h, C.cast(C.c_void_p(address), C.POINTER(C.c_uint8)), capacity, C.byref(need)), "emat_tree_export_nodes")
        return int(need.value)

    def tree_apply_nodes_at(self, address: int, nbytes: int):
        self._ck(self._lib.emat_tree_apply_nodes(self._h, C.cast(C.c_void_p(address), C.POINTER(C.c_uint8)), nbytes), "emat_tree_apply_nodes")

    def tree_reassemble_end(self):
        self._ck(self._lib.emat_tree_reassemble_end(self._h), "emat_tree_reassemble_end")

    def tree_download(self):
        n, nm, ni, nf = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_tree_get_sizes(self._h, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf)), "emat_tree_get_sizes")
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view()
        ref = np.zeros(self.num_sites, np.uint8)
        self._ck(self._lib.emat_tree_download(self._h, C.byref(v), _ptr(ref, C.c_uint8)), "emat_tree_download")
        t.root = v.root
        return t.trimmed(), ref

    def tree_counters(self):
        """(growths of the cut-state pools, growths of the list heaps, cut-point states that needed the large kernel) of the
        HBM-resident tree: testing aid."""
        out = np.zeros(3, np.int32)
        self._ck(self._lib.emat_debug_tree_counters(self._h, _ptr(out, C.c_int32)), "emat_debug_tree_counters")
        return int(out[0]), int(out[1]), int(out[2])

    def run_moves_even(self, moves_per_part: int, one_more_below: int):
        """`moves_per_part` moves on every part, one more on the parts [0, one_more_below) (not the reference's remainder rule)."""
        self._ck(self._lib.emat_run_moves_even(self._h, moves_per_part, one_more_below), "emat_run_moves_even")

    def recalc_derived(self):
        self._ck(self._lib.emat_recalc_derived(self._h), "emat_recalc_derived")

    def check_derived(self, tol_scale: float = 1.0):
        """The reference's check_derived_quantities on the device (raises EmatError when a part is off); returns (part, deviations)."""
        wp = C.c_int32(); w4 = (C.c_double * 4)()
        self._ck(self._lib.emat_check_derived(self._h, tol_scale, C.byref(wp), w4), "emat_check_derived")
        return wp.value, [float(x) for x in w4]

    def last_run_ms(self) -> float:
        ms = C.c_double()
        self._ck(self._lib.emat_last_run_ms(self._h, C.byref(ms)), "emat_last_run_ms")
        return float(ms.value)

    def last_kernel_ms(self):
        ms, n = C.c_double(), C.c_int32()
        self._ck(self._lib.emat_last_kernel_ms(self._h, C.byref(ms), C.byref(n)), "emat_last_kernel_ms")
        return float(ms.value), int(n.value)

    def totals(self):
        g, a = C.c_double(), C.c_double()
        self._ck(self._lib.emat_get_totals(self._h, C.byref(g), C.byref(a)), "emat_get_totals")
        return float(g.value), float(a.value)

    def global_stats(self, num_partitions: int = 1):
        """(Ttwiddle_beta_a [P][4], num_muts_beta_ab [P][4][4], num_muts) over the parts of this handle."""
        T = np.zeros((num_partitions, 4)); M = np.zeros((num_partitions, 4, 4), np.int64); nm = C.c_int64()
        self._ck(self._lib.emat_get_global_stats(self._h, num_partitions, T.ctypes.data_as(C.POINTER(C.c_double)), M.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(nm)),
                 "emat_get_global_stats")
        return T, M, int(nm.value)

    def num_muts_l(self) -> np.ndarray:
        """calc_num_muts_l over the parts of this handle (mutations per site)."""
        out = np.zeros(self.num_sites, np.int32)
        self._ck(self._lib.emat_get_num_muts_l(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))), "emat_get_num_muts_l")
        return out

    def scalable_coalescent_log_prior(self, t_ref: float, t_step: float) -> float:
        """Scalable_coalescent_prior::calc_log_prior of the whole tree, from the parts of this handle."""
        v = C.c_double()
        self._ck(self._lib.emat_get_scalable_coalescent_log_prior(self._h, t_ref, t_step, C.byref(v)), "emat_get_scalable_coalescent_log_prior")
        return float(v.value)

    def part_tree_lengths(self, num_parts: int) -> np.ndarray:
        out = np.zeros(num_parts)
        self._ck(self._lib.emat_get_part_tree_lengths(self._h, out.ctypes.data_as(C.POINTER(C.c_double))), "emat_get_part_tree_lengths")
        return out

    def Ttwiddle_l_partial(self, ext_offset, ext_node, ext_length):
        """(S, R, total tree length if this handle holds the run's root else 0) over the parts of this handle."""
        off = np.ascontiguousarray(ext_offset, np.int32); node = np.ascontiguousarray(ext_node, np.int32); val = np.ascontiguousarray(ext_length, np.float64)
        if node.shape[0] == 0:
            node = np.zeros(1, np.int32); val = np.zeros(1)
        S = np.zeros(self.num_sites); Rv = np.zeros(self.num_sites); T = C.c_double(0.0); dp = C.POINTER(C.c_double); ip = C.POINTER(C.c_int32)
        self._ck(self._lib.emat_Ttwiddle_l_partial(self._h, off.ctypes.data_as(ip), node.ctypes.data_as(ip), val.ctypes.data_as(dp), S.ctypes.data_as(dp), Rv.ctypes.data_as(dp), C.byref(T)),
                 "emat_Ttwiddle_l_partial")
        return S, Rv, float(T.value)

    def Ttwiddle_l_finish(self, S, R, tree_length: float) -> np.ndarray:
        S = np.ascontiguousarray(S, np.float64); R = np.ascontiguousarray(R, np.float64); out = np.zeros(self.num_sites); dp = C.POINTER(C.c_double)
        self._ck(self._lib.emat_Ttwiddle_l_finish(self._h, S.ctypes.data_as(dp), R.ctypes.data_as(dp), tree_length, out.ctypes.data_as(dp)), "emat_Ttwiddle_l_finish")
        return out

    def scalable_coalescent_partial(self, t_ref: float, t_step: float, first_cell: int, num_cells: int):
        """(partial k_bar grid over [first_cell, first_cell + num_cells), sum of -log N(t) over inner nodes, first cell needed)."""
        kb = np.zeros(max(num_cells, 1)); logs = C.c_double(); need = C.c_int32()
        self._ck(self._lib.emat_scalable_coalescent_partial(self._h, t_ref, t_step, first_cell, num_cells, kb.ctypes.data_as(C.POINTER(C.c_double)), C.byref(logs), C.byref(need)),
                 "emat_scalable_coalescent_partial")
        return kb[:num_cells], float(logs.value), int(need.value)

    def scalable_coalescent_log_prior_from_grid(self, t_ref: float, t_step: float, first_cell: int, k_bar_sum: np.ndarray, sum_neg_log_pop: float) -> float:
        kb = np.ascontiguousarray(k_bar_sum, np.float64); v = C.c_double()
        self._ck(self._lib.emat_scalable_coalescent_log_prior(self._h, t_ref, t_step, first_cell, kb.shape[0], kb.ctypes.data_as(C.POINTER(C.c_double)), sum_neg_log_pop, C.byref(v)),
                 "emat_scalable_coalescent_log_prior")
        return float(v.value)

    def debug_pop(self, pop: PopModel, op: int, a, b) -> np.ndarray:
        """Test hook: the device's pop_at_time (op 0) / pop_integral (op 1), point by point."""
        a = np.ascontiguousarray(a, np.float64); b = np.ascontiguousarray(b, np.float64); out = np.zeros_like(a)
        m = pop.c_struct(); dp = C.POINTER(C.c_double)
        self._ck(self._lib.emat_debug_pop(self._h, C.byref(m), op, a.shape[0], a.ctypes.data_as(dp), b.ctypes.data_as(dp), out.ctypes.data_as(dp)), "emat_debug_pop")
        return out

    def debug_interval_op(self, op: int, a, b):
        """Test hook: the device's interval-set algebra; sets as lists of [start, end)."""
        A = np.ascontiguousarray(np.asarray(a, np.int32).reshape(-1, 2)); Bv = np.ascontiguousarray(np.asarray(b, np.int32).reshape(-1))
        nb = Bv.shape[0] if op == 5 else Bv.shape[0] // 2
        out = np.zeros((A.shape[0] + nb + 1, 2), np.int32); n = C.c_int32(); ip = C.POINTER(C.c_int32)
        self._ck(self._lib.emat_debug_interval_op(self._h, op, A.ctypes.data_as(ip), A.shape[0], Bv.ctypes.data_as(ip), nb, out.ctypes.data_as(ip), C.byref(n)), "emat_debug_interval_op")
        return out[: n.value].tolist() if op <= 3 else bool(n.value)

    def debug_tree_query(self, part: int, op: int, a, b) -> np.ndarray:
        """Test hook: the moves' find_MRCA_of (op 0) / descends_from (op 1) on a resident part; -1 = no node."""
        a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32); out = np.zeros_like(a); ip = C.POINTER(C.c_int32)
        self._ck(self._lib.emat_debug_tree_query(self._h, part, op, a.shape[0], a.ctypes.data_as(ip), b.ctypes.data_as(ip), out.ctypes.data_as(ip)), "emat_debug_tree_query")
        return out

    def debug_graft(self, part: int, X: int, mu_proposal: float, mode: int = 0, new_sibling: int = 0, new_t_P: float = 0.0) -> dict:
        """Test hook: the moves' own graft analysis (mode 0), + peel (1), + apply (2), or a whole re-attachment with a proposed
        new graft (3) on a resident part; returns what emat_debug_graft writes out, decoded (decode_graft_output)."""
        out = np.zeros(4096); n = C.c_int32()
        self._ck(self._lib.emat_debug_graft(self._h, part, X, mu_proposal, mode, new_sibling, new_t_P, out.ctypes.data_as(C.POINTER(C.c_double)), out.shape[0], C.byref(n)), "emat_debug_graft")
        return decode_graft_output(out[: n.value], mode)

    def debug_edit(self, part: int, X: int, ops):
        """Test hook: one tree-editing session on node X; ops = [["slide", t] | ["hop_up"] | ["flip"] | ["hop_down", node], ...]."""
        kind = np.array([{"slide": 0, "hop_up": 1, "flip": 2, "hop_down": 3}[o[0]] for o in ops], np.int32)
        node = np.array([int(o[1]) if o[0] == "hop_down" else -1 for o in ops], np.int32)
        t = np.array([float(o[1]) if o[0] == "slide" else 0.0 for o in ops], np.float64)
        self._ck(self._lib.emat_debug_edit(self._h, part, X, kind.shape[0], _ptr(kind, C.c_int32), _ptr(node, C.c_int32), _ptr(t, C.c_double)), "emat_debug_edit")

    def debug_sample_history(self, part: int, branch, t_end, start_seq, T: float, mu: float):
        """Test hook: one sampled mutational history per (branch[i], t_end[i]) on a resident part; returns a list of histories, each a
        list of [from, site, to, t]."""
        branch = np.ascontiguousarray(branch, np.int32); t_end = np.ascontiguousarray(t_end, np.float64); seq = np.ascontiguousarray(start_seq, np.uint8)
        n = branch.shape[0]; cap = 64 * n + 1024
        counts = np.zeros(max(n, 1), np.int32); muts = np.zeros((cap, 4)); tot = C.c_int32()
        self._ck(self._lib.emat_debug_sample_history(self._h, part, n, _ptr(branch, C.c_int32), _ptr(t_end, C.c_double), _ptr(seq, C.c_uint8), T, mu,
                                                     _ptr(counts, C.c_int32), _ptr(muts, C.c_double), cap, C.byref(tot)), "emat_debug_sample_history")
        out, k = [], 0
        for i in range(n):
            out.append([[int(m[1]), int(m[0]), int(m[2]), float(m[3])] for m in muts[k: k + counts[i]]]); k += int(counts[i])
        return out

    def debug_gamma(self, mode: int, a, x_or_q) -> np.ndarray:
        """Test hook: the device's gamma_q (mode 0) / gamma_q_inv (mode 1), point by point."""
        a = np.ascontiguousarray(a, np.float64); x = np.ascontiguousarray(x_or_q, np.float64); out = np.zeros_like(a)
        dp = C.POINTER(C.c_double)
        self._ck(self._lib.emat_debug_gamma(self._h, mode, a.shape[0], a.ctypes.data_as(dp), x.ctypes.data_as(dp), out.ctypes.data_as(dp)), "emat_debug_gamma")
        return out

    def part_download(self, part: int) -> FlatTree:
        n, nm, ni, nf = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_part_get_sizes(self._h, part, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf)), "emat_part_get_sizes")
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view()
        self._ck(self._lib.emat_part_download(self._h, part, C.byref(v)), "emat_part_download")
        t.root = v.root
        return t.trimmed()

    def part_derived(self, part: int, num_nodes: int):
        lam = np.zeros(num_nodes)
        nmiss = np.zeros(num_nodes, np.int32)
        g, a = C.c_double(), C.c_double()
        self._ck(self._lib.emat_part_get_derived(self._h, part, _ptr(lam, C.c_double), _ptr(nmiss, C.c_int32), C.byref(g), C.byref(a)), "emat_part_get_derived")
        return lam, nmiss, float(g.value), float(a.value)

    def part_state_frequencies(self, part: int) -> np.ndarray:
        """Subrun::state_frequencies_of_ref_sequence_per_partition (subrun.h:45): counts[site partition, state] of the reference sequence."""
        n = C.c_int32(8)
        counts = np.zeros((8, 4), np.int32)
        self._ck(self._lib.emat_part_get_state_frequencies(self._h, part, C.byref(n), _ptr(counts, C.c_int32)), "emat_part_get_state_frequencies")
        return counts[: n.value].copy()

    def part_coalescent(self, part: int, cap: int = 1 << 16):
        for _ in range(2):      # (a grid that outgrew `cap` -- a root that wandered for 100 000 moves over a fine grid -- reports its length: once more with that)
            n = C.c_int32(cap)
            kb, kt, k, ps = np.zeros(cap), np.zeros(cap), np.zeros(cap), np.zeros(cap)
            na = np.zeros(cap, np.int32)
            tr, ts = C.c_double(), C.c_double()
            st = self._lib.emat_part_get_coalescent(self._h, part, C.byref(n), _ptr(kb, C.c_double), _ptr(kt, C.c_double), _ptr(k, C.c_double),
                                                    _ptr(ps, C.c_double), _ptr(na, C.c_int32), C.byref(tr), C.byref(ts))
            if st == 0 or n.value <= cap:
                break
            cap = n.value
        self._ck(st, "emat_part_get_coalescent")
        m = n.value
        return dict(k_bar_p=kb[:m], k_twiddle_bar_p=kt[:m], k_twiddle_bar=k[:m], popsize_bar=ps[:m], num_active_parts=na[:m], t_ref=tr.value, t_step=ts.value)

    def part_stats(self, part: int) -> dict:
        s = _PartStatsC()
        self._ck(self._lib.emat_part_get_stats(self._h, part, C.byref(s)), "emat_part_get_stats")
        return dict(status=s.status, num_nodes=s.num_nodes, moves_done=s.moves_done, proposed=list(s.proposed), accepted=list(s.accepted),
                    algorithmic_bytes=s.algorithmic_bytes, rng_draws=s.rng_draws, device_ticks=s.device_ticks, algorithmic_write_bytes=s.algorithmic_write_bytes)

    def build_usher_like(self, tips: "TipDescs", seed: int) -> "FlatTree":
        """the reference's UShER-like initial tree from tip descriptors (set_ref_sequence first); the graft loop runs on the device."""
        td = tips.c_struct()
        self._ck(self._lib.emat_tree_build_usher_like(self._h, C.byref(td), seed), "emat_tree_build_usher_like")
        n, nm, ni, nf = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_tree_built_sizes(self._h, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf)), "emat_tree_built_sizes")
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view()
        self._ck(self._lib.emat_tree_built_get(self._h, C.byref(v)), "emat_tree_built_get")
        t.root = v.root
        return t.trimmed()

    def build_default(self, tips: "TipDescs", seed: int):
        """the reference's DEFAULT initial tree (build_initial_phylo_tree: maximum-parsimony guide tree, nearest-first
        rebuilds, SPR refinement, regression rooting, dating) from tip descriptors (set_ref_sequence first).  Host code, as in the
        reference: works on a device = -1 handle.  Returns (tree, ref, report): the tree is written against `ref`, the ROOT's sequence."""
        td = tips.c_struct()
        rep = (C.c_int32 * 4)()
        self._ck(self._lib.emat_tree_build_default(self._h, C.byref(td), seed, rep), "emat_tree_build_default")
        n, nm, ni, nf = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_tree_built_sizes(self._h, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf)), "emat_tree_built_sizes")
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view()
        self._ck(self._lib.emat_tree_built_get(self._h, C.byref(v)), "emat_tree_built_get")
        t.root = v.root
        ref = np.zeros(self.num_sites, np.uint8)
        self._ck(self._lib.emat_tree_built_ref(self._h, ref.ctypes.data_as(C.POINTER(C.c_uint8))), "emat_tree_built_ref")
        return t.trimmed(), ref, dict(guide_deltas=rep[0], refined_deltas=rep[1], spr_deltas=rep[2], rooting="regression" if rep[3] == 0 else "midpoint")

    def main_class_mask(self, num_parts: int) -> np.ndarray:
        """Debugging aid: which resident parts run in the main launch (k_run_moves) rather than in a side class (k_run_moves_side)."""
        out = np.zeros(num_parts, np.int32)
        self._lib.emat_debug_arena_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        self._ck(self._lib.emat_debug_arena_bytes(self._h, out.ctypes.data_as(C.POINTER(C.c_int32))), "emat_debug_arena_bytes")
        return out >= 0

    def debug_slab_layout(self, part: int) -> dict:
        out = (C.c_uint32 * 8)()
        self._ck(self._lib.emat_debug_slab_layout(self._h, part, out), "emat_debug_slab_layout")
        return dict(zip(("header", "nodes", "cells", "trace", "heap_used", "heap_cap", "scratch", "num_cells"), [int(x) for x in out]))

    def part_rng(self, part: int) -> dict:
        """Where the part's Philox stream stands: key, blocks consumed, pending half block."""
        k, c, sp, hs = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_int32()
        self._ck(self._lib.emat_part_get_rng(self._h, part, C.byref(k), C.byref(c), C.byref(sp), C.byref(hs)), "emat_part_get_rng")
        return dict(key=int(k.value), counter=int(c.value), spare=int(sp.value), has_spare=bool(hs.value))

    def part_trace(self, part: int, cap: int) -> np.ndarray:
        n = C.c_int32(cap)
        tr = np.zeros((max(cap, 1), 4))
        self._ck(self._lib.emat_part_get_trace(self._h, part, C.byref(n), _ptr(tr, C.c_double)), "emat_part_get_trace")
        return tr[: n.value]

    def last_error(self) -> str:
        return self._lib.emat_last_error(self._h).decode()


class EmatRun:
    """Host-side driver above the boundary (include/emat_host.h): partitioning, repartition, reassemble."""

    def __init__(self, backend: Optional[EmatBackend], tree: FlatTree, ref_sequence: np.ndarray, seed: int):
        self._lib = load_library()
        self.backend = backend
        self._ref = np.ascontiguousarray(ref_sequence, np.uint8)
        self.num_sites = int(self._ref.shape[0])
        self._h = C.c_void_p()
        v = tree.c_view()
        st = self._lib.emat_run_create(backend.handle if backend is not None else None, C.byref(v), _ptr(self._ref, C.c_uint8), self.num_sites, int(seed), C.byref(self._h))
        if st != 0:
            raise EmatError("emat_run_create failed: %s" % STATUS_NAMES.get(st, st))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.emat_run_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st: int, what: str):
        if st != 0:
            msg = self._lib.emat_run_last_error(self._h)
            raise EmatError("%s: %s (%s)" % (what, STATUS_NAMES.get(st, st), msg.decode() if msg else ""))

    def set_num_parts(self, n: int):
        self._ck(self._lib.emat_run_set_num_parts(self._h, n), "emat_run_set_num_parts")

    def set_max_part_nodes(self, n: int):
        """Not in the reference: parts larger than n nodes get further, randomly drawn cut nodes at every repartition
        (0 = off, the reference's rule exactly: the default; -1 = three times the mean part size)."""
        self._ck(self._lib.emat_run_set_max_part_nodes(self._h, n), "emat_run_set_max_part_nodes")

    def partition_stats(self) -> dict:
        """The last repartition: number of parts, nodes of the largest, cut nodes added by the size limit, the limit in effect."""
        a, b, c, e = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_run_partition_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(e)), "emat_run_partition_stats")
        return {"num_parts": a.value, "largest_part_nodes": b.value, "extra_cuts": c.value, "max_part_nodes": e.value}

    def follow_draws(self, leader: "Optional[EmatRun]"):
        """This run takes the partition draws of `leader` (same process, seed, tree, settings) instead of drawing them again (emat_multi's shards)."""
        self._ck(self._lib.emat_run_follow_draws(self._h, leader._h if leader is not None else None), "emat_run_follow_draws")

    def draw_partition(self):
        """The draw of the coming cycle's partition (stencil refresh, pick, part-size limit) ahead of emat_run_repartition."""
        self._ck(self._lib.emat_run_draw_partition(self._h), "emat_run_draw_partition")

    def debug_redraw_partition(self) -> np.ndarray:
        """Test hook: the sorted cut nodes the last repartition's draw (same stencil, same refinement stream) gives on the tree as it is now."""
        n = C.c_int32(0)
        self._lib.emat_run_debug_redraw_partition(self._h, None, C.byref(n))        # (asks for the count)
        cuts = np.zeros(max(1, n.value), np.int32); n = C.c_int32(cuts.shape[0])
        self._ck(self._lib.emat_run_debug_redraw_partition(self._h, _ptr(cuts, C.c_int32), C.byref(n)), "emat_run_debug_redraw_partition")
        return cuts[: n.value].copy()

    def set_hky(self, mu: float, kappa: float, pi, nu_l=None):
        pi = np.ascontiguousarray(pi, np.float64)
        nu = None if nu_l is None else np.ascontiguousarray(nu_l, np.float64)
        self._ck(self._lib.emat_run_set_hky(self._h, mu, kappa, _ptr(pi, C.c_double), None if nu is None else _ptr(nu, C.c_double)), "emat_run_set_hky")

    def set_pop_model(self, pop: PopModel):
        m = pop.c_struct()
        self._ck(self._lib.emat_run_set_pop_model(self._h, C.byref(m)), "emat_run_set_pop_model")

    def set_coalescent_t_step(self, t_step: float):
        self._ck(self._lib.emat_run_set_coalescent_t_step(self._h, t_step), "emat_run_set_coalescent_t_step")

    def set_flags(self, only_displacing_inner_nodes: bool = False, topology_moves_enabled: bool = True):
        self._ck(self._lib.emat_run_set_flags(self._h, int(only_displacing_inner_nodes), int(topology_moves_enabled)), "emat_run_set_flags")

    def set_reference_remainder(self, on: bool = True):
        """The remainder of count / parts goes to part 0 as in Run::run_local_moves (default: one move each on the first parts)."""
        self._ck(self._lib.emat_run_set_reference_remainder(self._h, 1 if on else 0), "emat_run_set_reference_remainder")

    def set_paranoid(self, on: bool = True):
        self._ck(self._lib.emat_run_set_paranoid(self._h, 1 if on else 0), "emat_run_set_paranoid")

    def repartition(self):
        self._ck(self._lib.emat_run_repartition(self._h), "emat_run_repartition")

    def num_parts(self):
        n, r = C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_run_num_parts(self._h, C.byref(n), C.byref(r)), "emat_run_num_parts")
        return n.value, r.value

    def part(self, i: int):
        n, nm, ni, nf = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_run_part_sizes(self._h, i, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf)), "emat_run_part_sizes")
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view()
        incl, seed = C.c_int32(), C.c_uint64()
        self._ck(self._lib.emat_run_part_get(self._h, i, C.byref(v), C.byref(incl), C.byref(seed)), "emat_run_part_get")
        t.root = v.root
        return t.trimmed(), bool(incl.value), int(seed.value)

    def part_put(self, i: int, subtree: FlatTree):
        v = subtree.c_view()
        self._ck(self._lib.emat_run_part_put(self._h, i, C.byref(v)), "emat_run_part_put")

    def push_params(self):
        self._ck(self._lib.emat_run_push_params(self._h), "emat_run_push_params")

    def run_moves(self, count: int):
        self._ck(self._lib.emat_run_moves(self._h, count), "emat_run_moves")

    def reassemble(self):
        self._ck(self._lib.emat_run_reassemble(self._h), "emat_run_reassemble")

    def Ttwiddle_l(self) -> np.ndarray:
        """calc_Ttwiddle_l of the whole tree, computed from the parts on the device (single process)."""
        out = np.zeros(self.num_sites)
        self._ck(self._lib.emat_run_get_Ttwiddle_l(self._h, out.ctypes.data_as(C.POINTER(C.c_double))), "emat_run_get_Ttwiddle_l")
        return out

    def Ttwiddle_ext(self, tree_length_of_part: np.ndarray, num_local_parts: int):
        """(ext_offset, ext_node, ext_length) of this rank's parts for emat_Ttwiddle_l_partial."""
        tl = np.ascontiguousarray(tree_length_of_part, np.float64)
        cap = tl.shape[0] + 1
        off = np.zeros(num_local_parts + 1, np.int32); node = np.zeros(cap, np.int32); val = np.zeros(cap); cnt = C.c_int32()
        self._ck(self._lib.emat_run_Ttwiddle_ext(self._h, tl.ctypes.data_as(C.POINTER(C.c_double)), off.ctypes.data_as(C.POINTER(C.c_int32)), node.ctypes.data_as(C.POINTER(C.c_int32)),
                                                 val.ctypes.data_as(C.POINTER(C.c_double)), cap, C.byref(cnt)), "emat_run_Ttwiddle_ext")
        return off, node[: cnt.value], val[: cnt.value]

    # ---- a run sharded over several processes (include/emat_host.h) ----
    def set_shard(self, rank: int, world: int):
        self._ck(self._lib.emat_run_set_shard(self._h, rank, world), "emat_run_set_shard")

    def shard_range(self):
        lo, hi, lr = C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_run_shard_range(self._h, C.byref(lo), C.byref(hi), C.byref(lr)), "emat_run_shard_range")
        return lo.value, hi.value, lr.value

    def coalescent_begin(self):
        lo, hi = C.c_double(), C.c_double()
        self._ck(self._lib.emat_run_coalescent_begin(self._h, C.byref(lo), C.byref(hi)), "emat_run_coalescent_begin")
        return lo.value, hi.value

    def run_moves_sharded(self, count: int):
        self._ck(self._lib.emat_run_moves_sharded(self._h, count), "emat_run_moves_sharded")

    def pack_local_parts(self) -> np.ndarray:
        need = C.c_uint64()
        self._ck(self._lib.emat_run_pack_local_parts(self._h, None, 0, C.byref(need)), "emat_run_pack_local_parts")
        buf = np.zeros(int(need.value), np.uint8)
        self._ck(self._lib.emat_run_pack_local_parts(self._h, _ptr(buf, C.c_uint8), buf.shape[0], C.byref(need)), "emat_run_pack_local_parts")
        return buf

    def unpack_parts(self, buf: np.ndarray):
        buf = np.ascontiguousarray(buf, np.uint8)
        self._ck(self._lib.emat_run_unpack_parts(self._h, _ptr(buf, C.c_uint8), buf.shape[0]), "emat_run_unpack_parts")

    def set_device_tree(self, on: bool = True):
        """keep the authoritative tree in HBM; cycles then move only the partition and the topology."""
        self._ck(self._lib.emat_run_set_device_tree(self._h, int(on)), "emat_run_set_device_tree")

    def note_device_reassembled(self, site, to):
        site = np.ascontiguousarray(site, np.int32); to = np.ascontiguousarray(to, np.uint8)
        self._ck(self._lib.emat_run_note_device_reassembled(self._h, int(site.shape[0]), _ptr(site, C.c_int32), _ptr(to, C.c_uint8)), "emat_run_note_device_reassembled")

    def do_mcmc_steps(self, steps: int, local_moves_per_cycle: int = -1):
        self._ck(self._lib.emat_run_do_mcmc_steps(self._h, steps, local_moves_per_cycle), "emat_run_do_mcmc_steps")

    def tree(self):
        n, nm, ni, nf = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_run_tree_sizes(self._h, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf)), "emat_run_tree_sizes")
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view()
        ref = np.zeros(self.num_sites, np.uint8)
        self._ck(self._lib.emat_run_tree_get(self._h, C.byref(v), _ptr(ref, C.c_uint8)), "emat_run_tree_get")
        t.root = v.root
        return t.trimmed(), ref

    def t_max_tip(self) -> float:
        t = C.c_double()
        self._ck(self._lib.emat_run_t_max_tip(self._h, C.byref(t)), "emat_run_t_max_tip")
        return float(t.value)


class EmatMultiRun:
    """One run over several GPUs of ONE process (include/emat_host.h, emat_run_create_multi): n backends with the whole tree in the HBM
    of each, the exchanges of a cycle done in C++ over RCCL ("rccl"), through host buffers ("host"), or whichever applies ("auto")."""
    EXCHANGE = {"host": 0, "rccl": 1, "auto": 2}

    def __init__(self, devices: Sequence[int], tree: FlatTree, ref_sequence: np.ndarray, seed: int, exchange: str = "auto", trace_moves: int = 0, use_lds: bool = True):
        self._lib = load_library()
        self._ref = np.ascontiguousarray(ref_sequence, np.uint8)
        self.num_sites = int(self._ref.shape[0])
        dev = np.ascontiguousarray(devices, np.int32)
        cfg = _ConfigC(0, self.num_sites, 0, 0.0, trace_moves, 1 if use_lds else 0)
        self._h = C.c_void_p()
        v = tree.c_view()
        if os.environ.get("EMAT_RCCL_LIB"):
            self._lib.emat_multi_set_rccl_library(os.environ["EMAT_RCCL_LIB"].encode())
        st = self._lib.emat_run_create_multi(_ptr(dev, C.c_int32), int(dev.shape[0]), C.byref(cfg), C.byref(v), _ptr(self._ref, C.c_uint8), self.num_sites, int(seed),
                                             self.EXCHANGE[exchange], C.byref(self._h))
        if st != 0:
            raise EmatError("emat_run_create_multi failed: %s (the engine needs HIP devices; exchange \"rccl\" needs librccl.so and one device per shard)" % STATUS_NAMES.get(st, st))
        self.num_shards = int(self._lib.emat_multi_num_shards(self._h))
        _forward_env_options(lambda k, v: self._ck(self._lib.emat_multi_set_option(self._h, k.encode(), str(v).encode()), "emat_multi_set_option(%s)" % k))

    def close(self):
        if getattr(self, "_h", None):
            self._lib.emat_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, st: int, what: str):
        if st != 0:
            msg = self._lib.emat_multi_last_error(self._h)
            raise EmatError("%s: %s (%s)" % (what, STATUS_NAMES.get(st, st), msg.decode() if msg else ""))

    @property
    def exchange(self) -> str:
        return self._lib.emat_multi_exchange(self._h).decode()

    def backend_handle(self, shard: int):
        return self._lib.emat_multi_backend(self._h, shard)

    def set_num_parts(self, n: int):
        self._ck(self._lib.emat_multi_set_num_parts(self._h, n), "emat_multi_set_num_parts")

    def set_hky(self, mu: float, kappa: float, pi, nu_l=None):
        pi = np.ascontiguousarray(pi, np.float64)
        nu = None if nu_l is None else np.ascontiguousarray(nu_l, np.float64)
        self._ck(self._lib.emat_multi_set_hky(self._h, mu, kappa, _ptr(pi, C.c_double), None if nu is None else _ptr(nu, C.c_double)), "emat_multi_set_hky")

    def set_pop_model(self, pop: PopModel):
        m = pop.c_struct()
        self._ck(self._lib.emat_multi_set_pop_model(self._h, C.byref(m)), "emat_multi_set_pop_model")

    def set_coalescent_t_step(self, t_step: float):
        self._ck(self._lib.emat_multi_set_coalescent_t_step(self._h, t_step), "emat_multi_set_coalescent_t_step")

    def set_flags(self, only_displacing_inner_nodes: bool = False, topology_moves_enabled: bool = True):
        self._ck(self._lib.emat_multi_set_flags(self._h, int(only_displacing_inner_nodes), int(topology_moves_enabled)), "emat_multi_set_flags")

    def set_paranoid(self, on: bool = True):
        self._ck(self._lib.emat_multi_set_paranoid(self._h, int(on)), "emat_multi_set_paranoid")

    def repartition(self):
        self._ck(self._lib.emat_multi_repartition(self._h), "emat_multi_repartition")

    def run_moves(self, count: int):
        self._ck(self._lib.emat_multi_run_moves(self._h, count), "emat_multi_run_moves")

    def check_derived(self, tol_scale: float = 1.0):
        self._ck(self._lib.emat_multi_check_derived(self._h, tol_scale), "emat_multi_check_derived")

    def reassemble(self):
        self._ck(self._lib.emat_multi_reassemble(self._h), "emat_multi_reassemble")

    def totals(self):
        g, a = C.c_double(), C.c_double()
        self._ck(self._lib.emat_multi_get_totals(self._h, C.byref(g), C.byref(a)), "emat_multi_get_totals")
        return float(g.value), float(a.value)

    def do_mcmc_steps(self, steps: int, local_moves_per_cycle: int = -1):
        self._ck(self._lib.emat_multi_do_mcmc_steps(self._h, steps, local_moves_per_cycle), "emat_multi_do_mcmc_steps")

    def tree(self, shard: int = 0):
        n, nm, ni, nf = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        self._ck(self._lib.emat_multi_tree_sizes(self._h, C.byref(n), C.byref(nm), C.byref(ni), C.byref(nf)), "emat_multi_tree_sizes")
        t = FlatTree.empty(n.value, nm.value, ni.value, nf.value)
        v = t.c_view()
        ref = np.zeros(self.num_sites, np.uint8)
        self._ck(self._lib.emat_multi_tree_get(self._h, shard, C.byref(v), _ptr(ref, C.c_uint8)), "emat_multi_tree_get")
        t.root = v.root
        return t.trimmed(), ref
