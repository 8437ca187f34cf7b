"""`.dphy` files written from flat SoA trees (include/emat_dphy.h), verified and decoded against the REFERENCE's
schema: tests/golden/api_schema.json holds the vtable slots, types, defaults, alignments and struct layouts of its generated
header core/api_generated.h (flatc's output for core/api.fbs), extracted by tests/golden/make_api_schema.py -- the reader
below is driven by that table and knows no field number of its own, so writer and checker share no field table.  Also the file
layout of doc/dphy_file_format.md (version 3), and what becomes of times that float32 cannot hold (api.fbs:13-29)."""
import ctypes as C
import json
import os
import struct

import numpy as np

import delphy_amd as d
from delphy_amd.scenarios import make_scenario

dbl, i32, i64, u64 = C.c_double, C.c_int32, C.c_int64, C.c_uint64


class _PopModelC(C.Structure):
    _fields_ = d.engine._PopModelC._fields_


class DphyParams(C.Structure):
    _fields_ = [("step", i64), ("num_local_moves_per_global_move", i64), ("num_parts", i32), ("mu", dbl), ("mu_prior_alpha", dbl), ("mu_prior_beta", dbl),
                ("alpha", dbl), ("nu", C.POINTER(dbl)), ("hky_kappa", dbl), ("hky_pi", dbl * 4), ("pop_model", d.engine._PopModelC),
                ("pop_inv_n0_prior_alpha", dbl), ("pop_inv_n0_prior_beta", dbl), ("pop_g_prior_mu", dbl), ("pop_g_prior_scale", dbl), ("pop_g_min", dbl), ("pop_g_max", dbl),
                ("skygrid_tau", dbl), ("skygrid_tau_prior_alpha", dbl), ("skygrid_tau_prior_beta", dbl), ("skygrid_low_gamma_barrier_loc", dbl), ("skygrid_low_gamma_barrier_scale", dbl),
                ("skygrid_inv_nbar_prior_alpha", dbl), ("skygrid_inv_nbar_prior_beta", dbl),
                ("only_displacing_inner_nodes", i32), ("topology_moves_enabled", i32), ("repartitioning_enabled", i32), ("alpha_move_enabled", i32), ("mu_move_enabled", i32),
                ("final_pop_size_move_enabled", i32), ("pop_growth_rate_move_enabled", i32), ("skygrid_tau_move_enabled", i32), ("skygrid_low_gamma_barrier_enabled", i32),
                ("log_other_priors", dbl), ("log_coalescent_prior", dbl), ("log_G", dbl), ("total_branch_length", dbl)]


SCHEMA = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "api_schema.json")))


class FbError(AssertionError):
    pass


class Decoded:
    """One size-prefixed FlatBuffer verified and decoded against tests/golden/api_schema.json -- the vtable slots, types, defaults,
    alignments and struct layouts of the REFERENCE's generated header (core/api_generated.h), extracted by
    tests/golden/make_api_schema.py; nothing here knows a field number.  The checks are those of flatbuffers::Verifier
    (VerifyTableStart / VerifyField / VerifyOffset / VerifyVector / VerifyString / union verification), plus one the
    writer owes its readers: no vtable slot the schema does not know."""

    def __init__(self, data, root_table):
        self.b = bytes(data)
        n = len(self.b)
        self._req(n >= 8 and n % 8 == 0, "buffer length")
        (size,) = struct.unpack_from("<I", self.b, 0)
        self._req(size + 4 == n, "size prefix")
        self.value = self.table(4 + self._uoffset(4), root_table)

    def _req(self, ok, what):
        if not ok:
            raise FbError("FlatBuffer verification failed: " + what)

    def _u32(self, p):
        self._req(0 <= p and p + 4 <= len(self.b) and p % 4 == 0, "u32 at %d" % p)
        return struct.unpack_from("<I", self.b, p)[0]

    def _uoffset(self, p):
        o = self._u32(p)
        self._req(o > 0 and p + o < len(self.b), "offset at %d" % p)
        return o

    def struct_dtype(self, name):
        st = SCHEMA["structs"][name]
        dt = np.dtype({"names": [f["name"] for f in st["fields"]], "formats": ["<" + f["fmt"] if f["size"] > 1 else f["fmt"] for f in st["fields"]],
                       "offsets": [f["offset"] for f in st["fields"]], "itemsize": st["size"]})
        return dt, st["align"]

    def vector(self, pos, elem_size, elem_align):
        n = self._u32(pos)
        self._req((pos + 4) % elem_align == 0, "vector data alignment at %d" % pos)
        self._req(pos + 4 + n * elem_size <= len(self.b), "vector extent at %d" % pos)
        return n

    def string(self, pos):
        n = self.vector(pos, 1, 1)
        self._req(pos + 4 + n < len(self.b) and self.b[pos + 4 + n] == 0, "string terminator")
        return self.b[pos + 4: pos + 4 + n].decode()

    def table(self, pos, name):
        fields = SCHEMA["tables"][name]["fields"]
        self._req(pos % 4 == 0 and 0 <= pos and pos + 4 <= len(self.b), "table position")
        (so,) = struct.unpack_from("<i", self.b, pos)
        vt = pos - so
        self._req(0 <= vt and vt % 2 == 0 and vt + 4 <= len(self.b), "vtable position")
        vsize, tsize = struct.unpack_from("<HH", self.b, vt)
        self._req(vsize >= 4 and vsize % 2 == 0 and vt + vsize <= len(self.b) and pos + tsize <= len(self.b), "vtable / table extent")
        known = {f["vt"] for f in fields.values()}
        for slot in range(4, vsize, 2):
            (off,) = struct.unpack_from("<H", self.b, vt + slot)
            self._req(off == 0 or slot in known, "%s: vtable slot %d is not in the reference's schema" % (name, slot))
        out = {}
        for fname, f in fields.items():
            off = struct.unpack_from("<H", self.b, vt + f["vt"])[0] if f["vt"] + 2 <= vsize else 0
            if f["kind"] == "scalar":
                if off == 0:
                    out[fname] = f["default"]
                    continue
                p = pos + off
                self._req(off + f["size"] <= tsize and p % f["align"] == 0, "%s.%s: scalar extent / alignment" % (name, fname))
                out[fname] = struct.unpack_from("<" + f["fmt"], self.b, p)[0]
                continue
            if off == 0:
                out[fname] = None
                continue
            self._req(off + 4 <= tsize, "%s.%s: offset extent" % (name, fname))
            tgt = pos + off + self._uoffset(pos + off)
            if f["kind"] == "string":
                out[fname] = self.string(tgt)
            elif f["kind"] == "vector_of_scalars":
                n = self.vector(tgt, f["size"], f["size"])
                out[fname] = np.frombuffer(self.b, np.dtype("<" + f["fmt"]) if f["size"] > 1 else np.dtype(f["fmt"]), n, tgt + 4)
            elif f["kind"] == "vector_of_structs":
                dt, align = self.struct_dtype(f["struct"])
                n = self.vector(tgt, dt.itemsize, align)
                out[fname] = np.frombuffer(self.b, dt, n, tgt + 4)
            elif f["kind"] == "vector_of_tables":
                n = self.vector(tgt, 4, 4)
                out[fname] = [self.table(tgt + 4 + 4 * i + self._uoffset(tgt + 4 + 4 * i), f["table"]) for i in range(n)]
            elif f["kind"] == "table":
                out[fname] = self.table(tgt, f["table"])
            elif f["kind"] == "union":
                out[fname] = tgt      # resolved below, once its discriminator is known
        for fname, f in fields.items():
            if f["kind"] != "union":
                continue
            kind = out[f["type_field"]]
            members = {v: k for k, v in SCHEMA["unions"][f["union"]].items()}
            if kind == 0:
                self._req(out[fname] is None, "%s.%s: value without a type" % (name, fname))
            else:
                self._req(kind in members and out[fname] is not None, "%s.%s: union type %d" % (name, fname, kind))
                out[fname] = (members[kind], self.table(out[fname], members[kind]))
        return out


def test_the_schema_is_the_references():
    """tests/golden/api_schema.json is what make_api_schema.py extracts from the reference's generated header (checked where the
    reference is at hand), and it says what api.fbs says about a few fields one can read off the schema text."""
    ref_hdr = "/root/reference/core/api_generated.h"
    if os.path.exists(ref_hdr):
        import subprocess, sys, tempfile, shutil
        with tempfile.TemporaryDirectory() as tmp:
            shutil.copy(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_api_schema.py"), tmp)
            subprocess.run([sys.executable, os.path.join(tmp, "make_api_schema.py"), ref_hdr], check=True, capture_output=True)
            assert json.load(open(os.path.join(tmp, "api_schema.json"))) == SCHEMA
    P = SCHEMA["tables"]["Params"]["fields"]
    assert P["step"]["vt"] == 4 and P["mu_prior_alpha"]["vt"] == 4 + 2 * 38 and P["mu_prior_alpha"]["default"] == 1.0 and P["num_local_moves_per_global_move"]["default"] == -1
    assert P["pop_model"]["vt"] == 4 + 2 * 30 and P["pop_model_type"]["vt"] == 4 + 2 * 29 and P["mu_move_enabled"]["default"] == 1
    assert SCHEMA["structs"]["Node"]["size"] == 16 and SCHEMA["structs"]["Mutation"]["size"] == 16 and SCHEMA["structs"]["MissationInterval"]["size"] == 12
    assert [f["name"] for f in SCHEMA["structs"]["Mutation"]["fields"]] == ["branch", "site", "from", "to", "padding0", "t"]


def _call(fn, *args):
    n = u64()
    assert fn(*args, None, 0, C.byref(n)) == 0
    buf = (C.c_uint8 * n.value)()
    assert fn(*args, buf, n.value, C.byref(n)) == 0
    return bytes(buf)


def _lib():
    L = d.load_library()
    P = C.POINTER
    L.emat_dphy_params_defaults.argtypes = [P(DphyParams)]; L.emat_dphy_params_defaults.restype = None
    L.emat_dphy_tree_flatbuffer.argtypes = [P(d.engine._FlatTreeC), P(C.c_uint8), i32, P(C.c_uint8), u64, P(u64)]
    L.emat_dphy_tree_info_flatbuffer.argtypes = [P(d.engine._FlatTreeC), P(C.c_char_p), P(C.c_uint8), u64, P(u64)]
    L.emat_dphy_params_flatbuffer.argtypes = [P(DphyParams), i32, P(C.c_uint8), u64, P(u64)]
    L.emat_dphy_open.argtypes = [C.c_char_p, C.c_char_p, i32, C.c_char_p, i32, P(DphyParams), P(d.engine._FlatTreeC), P(C.c_char_p), P(C.c_void_p)]
    L.emat_dphy_write_state.argtypes = [C.c_void_p, P(d.engine._FlatTreeC), P(C.c_uint8), i32, P(DphyParams)]
    L.emat_dphy_close.argtypes = [C.c_void_p]
    return L


def _f32(x):
    """static_cast<float>(double) as api.cpp:60-72 does it: round to nearest even, overflow to +-inf."""
    with np.errstate(over="ignore"):
        return np.asarray(x, np.float64).astype(np.float32)


def _check_tree(buf, tree, ref):
    t = Decoded(buf, "Tree").value
    nodes, muts, miss = t["nodes"], t["mutations"], t["missation_intervals"]
    assert t["root_node"] == tree.root
    assert np.array_equal(nodes["parent"], tree.parent) and np.array_equal(nodes["left_child"], tree.child0) and np.array_equal(nodes["right_child"], tree.child1)
    assert np.array_equal(nodes["t"], _f32(tree.t))                                  # float32 times, api.fbs:13-18
    branch_of = np.repeat(np.arange(tree.num_nodes), np.diff(tree.mut_offset))
    assert np.array_equal(muts["branch"], branch_of) and np.array_equal(muts["site"], tree.mut_site)
    assert np.array_equal(muts["from"], tree.mut_from) and np.array_equal(muts["to"], tree.mut_to) and np.array_equal(muts["t"], _f32(tree.mut_t))
    assert np.all(muts["padding0"] == 0)
    assert np.array_equal(miss["branch"], np.repeat(np.arange(tree.num_nodes), np.diff(tree.miss_offset)))
    assert np.array_equal(miss["start_site"], tree.miss_start) and np.array_equal(miss["end_site"], tree.miss_end)
    assert np.array_equal(t["ref_seq"], ref)


def _check_params(buf, q, L):
    r = Decoded(buf, "Params").value
    same = ("step", "num_local_moves_per_global_move", "num_parts", "mu", "mu_prior_alpha", "mu_prior_beta", "alpha", "hky_kappa", "pop_inv_n0_prior_alpha", "pop_inv_n0_prior_beta",
            "pop_g_prior_mu", "pop_g_prior_scale", "pop_g_min", "pop_g_max", "skygrid_tau", "skygrid_tau_prior_alpha", "skygrid_tau_prior_beta", "skygrid_low_gamma_barrier_loc",
            "skygrid_low_gamma_barrier_scale", "skygrid_inv_nbar_prior_alpha", "skygrid_inv_nbar_prior_beta", "only_displacing_inner_nodes", "topology_moves_enabled",
            "repartitioning_enabled", "alpha_move_enabled", "mu_move_enabled", "final_pop_size_move_enabled", "pop_growth_rate_move_enabled", "skygrid_tau_move_enabled",
            "skygrid_low_gamma_barrier_enabled", "log_other_priors", "log_coalescent_prior", "log_G", "total_branch_length")
    for name in same:
        assert r[name.lower()] == getattr(q, name), name
    assert [r["hky_pi_" + a] for a in "acgt"] == list(q.hky_pi)
    assert r["log_posterior"] == q.log_G + q.log_coalescent_prior + q.log_other_priors            # run_to_api_params, api.cpp:293-297
    assert r["mpox_hack_enabled"] == 0 and r["mpox_mu"] == 0.0 and r["mpox_mu_star"] == 0.0
    kind, pm = r["pop_model"]
    if q.pop_model.kind == 2:
        n = q.pop_model.skygrid_num_knots
        assert kind == "SkygridPopModel" and pm["type"] == q.pop_model.skygrid_type
        assert np.array_equal(pm["x_k"], np.ctypeslib.as_array(q.pop_model.skygrid_x, (n,))) and np.array_equal(pm["gamma_k"], np.ctypeslib.as_array(q.pop_model.skygrid_gamma, (n,)))
        assert r["pop_t0"] == 0.0 and r["pop_n0"] == 0.0 and r["pop_g"] == 0.0          # the deprecated copies stay absent
    else:
        assert kind == "ExpPopModel" and [pm["t0"], pm["n0"], pm["g"], pm["min_pop"]] == list(q.pop_model.p)
        assert (r["pop_t0"], r["pop_n0"], r["pop_g"]) == (q.pop_model.p[0], q.pop_model.p[1], q.pop_model.p[2])
    if q.nu and not all(q.nu[l] == 1.0 for l in range(L)):
        assert np.array_equal(r["nu"], np.ctypeslib.as_array(q.nu, (L,)))
    else:
        assert r["nu"] is None


def test_tree_info_and_params_flatbuffers_round_trip():
    L = _lib()
    for name, kw in (("C2", dict(num_tips=300, num_sites=2000, uncertain_tips=0.3)), ("C3", dict(num_tips=200, num_sites=1500))):
        sc = make_scenario(name, **kw)
        v = sc.tree.c_view(); ref = np.ascontiguousarray(sc.ref, np.uint8)
        _check_tree(_call(L.emat_dphy_tree_flatbuffer, C.byref(v), ref.ctypes.data_as(C.POINTER(C.c_uint8)), sc.num_sites), sc.tree, ref)
        infos = Decoded(_call(L.emat_dphy_tree_info_flatbuffer, C.byref(v), None), "TreeInfo").value["node_infos"]
        assert len(infos) == sc.tree.num_nodes
        uncertain = 0
        for i, ni in enumerate(infos):
            tip = sc.tree.child0[i] == -1
            assert ni["name"] == ("TIP_%d" % i if tip else "")
            if tip and sc.tree.t_min[i] != sc.tree.t_max[i]:
                assert ni["has_uncertain_t"] == 1 and ni["t_min"] == sc.tree.t_min[i] and ni["t_max"] == sc.tree.t_max[i]
                uncertain += 1
            else:
                assert ni["has_uncertain_t"] == 0
        assert uncertain == int(np.sum((sc.tree.child0 == -1) & (sc.tree.t_min != sc.tree.t_max)))
        q = DphyParams(); L.emat_dphy_params_defaults(C.byref(q))
        q.step = 123456789012; q.num_parts = 7955; q.mu = sc.mu; q.hky_kappa = sc.kappa
        for a in range(4): q.hky_pi[a] = sc.pi[a]
        pm = sc.pop.c_struct(); q.pop_model = pm
        q.log_G = -1446268.79; q.log_coalescent_prior = -1137645.05; q.log_other_priors = -12.5; q.total_branch_length = 165757.5
        nu = np.ones(sc.num_sites)
        q.nu = nu.ctypes.data_as(C.POINTER(dbl))
        _check_params(_call(L.emat_dphy_params_flatbuffer, C.byref(q), sc.num_sites), q, sc.num_sites)     # all-one nu is omitted (api.cpp:218-221)
        nu[5] = 1.25
        _check_params(_call(L.emat_dphy_params_flatbuffer, C.byref(q), sc.num_sites), q, sc.num_sites)


def read_dphy(path):
    """The version-3 container of doc/dphy_file_format.md: (preamble fields, TreeInfo buffer, [(Tree buffer, Params buffer)], metadata
    JSON); checks the magic, the sentinel and the trailing offset that points at it."""
    b = open(path, "rb").read()
    assert b[:4] == b"DPHY" and struct.unpack_from("<I", b, 4)[0] == 3
    p = 8
    def rd_str(p):
        (n,) = struct.unpack_from("<I", b, p); return b[p + 4: p + 4 + n].decode(), p + 4 + n
    hdr = {}
    hdr["version"], p = rd_str(p); (hdr["build"],) = struct.unpack_from("<I", b, p); p += 4; hdr["commit"], p = rd_str(p)
    hdr["knee"], hdr["steps_per_sample"], hdr["alpha_on"], hdr["mpox"], hdr["mu_on"] = struct.unpack_from("<5I", b, p); p += 20
    (hdr["mu"],) = struct.unpack_from("<f", b, p); p += 4
    (n_info,) = struct.unpack_from("<I", b, p); p += 4
    info = b[p: p + n_info]; p += n_info
    states = []
    while True:
        (l1,) = struct.unpack_from("<I", b, p)
        if l1 == 0:
            sentinel = p; p += 4
            break
        (l2,) = struct.unpack_from("<I", b, p + 4); p += 8
        states.append((b[p: p + l1], b[p + l1: p + l1 + l2])); p += l1 + l2
    meta, p = rd_str(p)
    assert struct.unpack_from("<Q", b, p)[0] == sentinel and p + 8 == len(b)
    return hdr, info, states, meta


def test_dphy_file_layout(tmp_path):
    L = _lib()
    sc = make_scenario("C1", num_tips=40, num_sites=500)
    v = sc.tree.c_view(); ref = np.ascontiguousarray(sc.ref, np.uint8)
    q = DphyParams(); L.emat_dphy_params_defaults(C.byref(q)); q.mu = 2e-6; q.num_parts = 4
    pm = sc.pop.c_struct(); q.pop_model = pm
    path = os.path.join(str(tmp_path), "run.dphy").encode()
    w = C.c_void_p()
    assert L.emat_dphy_open(path, b"1.3.7", 2047, b"abc1234", 1000000, C.byref(q), C.byref(v), None, C.byref(w)) == 0
    for s in range(3):
        q.step = 1000000 * (s + 1)
        assert L.emat_dphy_write_state(w, C.byref(v), ref.ctypes.data_as(C.POINTER(C.c_uint8)), sc.num_sites, C.byref(q)) == 0
    assert L.emat_dphy_close(w) == 0
    hdr, info, states, meta = read_dphy(path)
    assert (hdr["version"], hdr["build"], hdr["commit"]) == ("1.3.7", 2047, "abc1234")
    assert (hdr["knee"], hdr["steps_per_sample"], hdr["alpha_on"], hdr["mpox"], hdr["mu_on"]) == (0, 1000000, 0, 0, 1) and hdr["mu"] == np.float32(2e-6)
    assert len(Decoded(info, "TreeInfo").value["node_infos"]) == sc.tree.num_nodes
    steps = []
    for tree_buf, params_buf in states:
        _check_tree(tree_buf, sc.tree, ref)
        steps.append(Decoded(params_buf, "Params").value["step"])
    assert steps == [1000000, 2000000, 3000000]
    assert json.loads(meta)["confidence"] == 90


def test_times_are_rounded_to_float32_as_the_reference_rounds_them():
    """api.fbs stores node and mutation times as float32; phylo_tree_to_api_tree converts with static_cast<float> (api.cpp:60-72):
    round to nearest, ties to even, and a root "mutation" at -DBL_MAX overflows to -inf.  Times chosen on and around ties."""
    L = _lib()
    sc = make_scenario("C1", num_tips=6, num_sites=40)
    tree = sc.tree
    ulp = 2.0 ** -23                                                   # float32 spacing in [1, 2)
    hard = np.array([1.0 + ulp / 2, 1.0 + 3 * ulp / 2, 1.0 + ulp / 2 + 2.0 ** -40, 737.123456789, -100000.3, 16777217.0, 0.1, -0.0, 1e-46, 3.4028235677973366e38])
    tree.t[: min(tree.num_nodes, hard.shape[0])] = hard[: min(tree.num_nodes, hard.shape[0])]
    if tree.mut_t.shape[0]:
        tree.mut_t[: min(tree.mut_t.shape[0], hard.shape[0])] = hard[: min(tree.mut_t.shape[0], hard.shape[0])][::-1]
        tree.mut_t[-1] = -1.7976931348623157e308                       # what a root delta carries (spr_move.cpp:422-425)
    v = tree.c_view(); ref = np.ascontiguousarray(sc.ref, np.uint8)
    buf = _call(L.emat_dphy_tree_flatbuffer, C.byref(v), ref.ctypes.data_as(C.POINTER(C.c_uint8)), sc.num_sites)
    _check_tree(buf, tree, ref)
    t = Decoded(buf, "Tree").value
    assert t["nodes"]["t"][0] == np.float32(1.0) and t["nodes"]["t"][1] == np.float32(1.0 + 2 * ulp) and t["nodes"]["t"][2] == np.float32(1.0 + ulp)   # ties to even, above a tie rounds up
    if tree.mut_t.shape[0]:
        assert np.isneginf(t["mutations"]["t"][-1])


def test_io_failures_are_reported(tmp_path):
    """A run file that cannot be opened or written in full is EMAT_ERR_IO (8), not EMAT_OK (/dev/full: every write fails at flush)."""
    L = _lib()
    sc = make_scenario("C1", num_tips=40, num_sites=500)
    v = sc.tree.c_view(); ref = np.ascontiguousarray(sc.ref, np.uint8)
    q = DphyParams(); L.emat_dphy_params_defaults(C.byref(q)); pm = sc.pop.c_struct(); q.pop_model = pm
    w = C.c_void_p()
    assert L.emat_dphy_open(os.path.join(str(tmp_path), "no", "such", "dir", "run.dphy").encode(), b"1", 1, b"c", 1, C.byref(q), C.byref(v), None, C.byref(w)) == 8
    if os.path.exists("/dev/full"):
        rc = L.emat_dphy_open(b"/dev/full", b"1", 1, b"c", 1, C.byref(q), C.byref(v), None, C.byref(w))
        if rc == 0:       # the preamble sat in stdio's buffer: the first flush finds out
            rc = L.emat_dphy_write_state(w, C.byref(v), ref.ctypes.data_as(C.POINTER(C.c_uint8)), sc.num_sites, C.byref(q))
            assert rc == 8
            assert L.emat_dphy_close(w) == 8
        else:
            assert rc == 8
