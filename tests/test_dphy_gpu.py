"""SURVEY 8(f).3 end to end on the GPU box: a run whose tree lives in HBM (parts cut, moved and gathered by kernels) written to a
`.dphy` file -- emat_tree_download -> emat_dphy_open / _write_state / _close (delphy_output.cpp:94-141, api.cpp:34-98) -- and read
back through the reference's own FlatBuffers contract (tests/golden/api_schema.json, from core/api_generated.h): every sample in the
file is the tree the device held at that cycle (times rounded to float32 as api.cpp:60-72 rounds them), and is the tree the ORACLE
makes of the same cycles -- its Subruns' moves on the same parts, put back together by its restatement of Run::reassemble."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import delphy_amd as d
from delphy_amd.scenarios import make_scenario
from helpers import replay_device_parts_in_the_oracle
from test_dphy_writer import Decoded, DphyParams, _check_tree, _f32, _lib, read_dphy

pytestmark = pytest.mark.gpu


def _close_in_float32(a, b):
    """Two float32 images of doubles that agree to 1e-9 relative: equal, or neighbours where the doubles straddle a rounding boundary."""
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return np.all((a == b) | (np.nextafter(a, b) == b))


def test_a_device_resident_run_written_to_a_dphy_file_is_the_oracles_run(tmp_path):
    from oracle_ffi import OracleRun
    L = _lib()
    sc = make_scenario("C3")                                   # 10 000 tips, 29 903 sites, HKY + skygrid
    assert sc.tree.num_nodes == 19999 and sc.num_sites == 29903
    parts = 400
    b = d.EmatBackend(sc.num_sites, trace_moves=300)
    run = d.EmatRun(b, sc.tree, sc.ref, 7)
    run.set_num_parts(parts); run.set_hky(sc.mu, sc.kappa, sc.pi); run.set_pop_model(sc.pop)
    run.set_max_part_nodes(0)        # the reference's partition rule exactly, as the restated Run applies it
    run.set_device_tree(True)
    orun = OracleRun(sc.tree, sc.ref, 7 ^ 0xD1B54A32D192ED03, parts)
    q = DphyParams(); L.emat_dphy_params_defaults(C.byref(q))
    q.mu = sc.mu; q.hky_kappa = sc.kappa; q.num_parts = parts
    for a in range(4): q.hky_pi[a] = sc.pi[a]
    pm = sc.pop.c_struct(); q.pop_model = pm
    path = os.path.join(str(tmp_path), "device_run.dphy").encode()
    v0 = sc.tree.c_view()
    w = C.c_void_p()
    assert L.emat_dphy_open(path, b"1.4.1", 2056, b"0000000", 1, C.byref(q), C.byref(v0), None, C.byref(w)) == 0
    device_trees, oracle_trees, totals = [], [], []
    ref = sc.ref
    try:
        for cycle in range(3):
            run.repartition(); orun.repartition()
            n, root_part = run.num_parts()
            assert (n, root_part) == orun.num_parts()
            moved = []
            replay_device_parts_in_the_oracle(sc, b, run, ref, n * 300 + 11, 300, oracle_parts=moved)   # device pass == oracle pass, move for move
            G, A = b.totals()
            for p in range(n):
                orun.part_put(p, moved[p])
            run.reassemble(); orun.reassemble(); orun.normalize_root()
            tree, ref = b.tree_download()                      # emat_tree_download: the gathered tree, straight from HBM
            tr, rr = run.tree()
            assert np.array_equal(ref, rr) and np.array_equal(tree.t, tr.t) and np.array_equal(tree.mut_site, tr.mut_site)
            q.step = (cycle + 1) * (n * 300 + 11); q.log_G = G; q.log_coalescent_prior = A
            q.total_branch_length = float(np.sum(tree.t[np.arange(tree.num_nodes) != tree.root] - tree.t[tree.parent[np.arange(tree.num_nodes) != tree.root]]))
            tv = tree.c_view(); r8 = np.ascontiguousarray(ref, np.uint8)
            assert L.emat_dphy_write_state(w, C.byref(tv), r8.ctypes.data_as(C.POINTER(C.c_uint8)), sc.num_sites, C.byref(q)) == 0
            device_trees.append((tree, r8)); oracle_trees.append(orun.tree()); totals.append((q.step, G, A))
        assert L.emat_dphy_close(w) == 0
    finally:
        run.close(); b.close(); orun.close()
    hdr, info, states, meta = read_dphy(path)
    assert hdr["version"] == "1.4.1" and hdr["build"] == 2056 and json.loads(meta)["confidence"] == 90
    infos = Decoded(info, "TreeInfo").value["node_infos"]
    assert len(infos) == sc.tree.num_nodes
    assert len(states) == 3
    changed = 0
    for k, (tree_buf, params_buf) in enumerate(states):
        tree, ref = device_trees[k]
        _check_tree(tree_buf, tree, ref)                       # every field, by the reference schema's names; float32 times
        t = Decoded(tree_buf, "Tree").value
        ot, oref = oracle_trees[k]
        assert t["root_node"] == ot.root and np.array_equal(t["ref_seq"], oref)
        assert np.array_equal(t["nodes"]["parent"], ot.parent) and np.array_equal(t["nodes"]["left_child"], ot.child0) and np.array_equal(t["nodes"]["right_child"], ot.child1)
        assert np.array_equal(t["mutations"]["branch"], np.repeat(np.arange(ot.num_nodes), np.diff(ot.mut_offset)))
        assert np.array_equal(t["mutations"]["site"], ot.mut_site) and np.array_equal(t["mutations"]["from"], ot.mut_from) and np.array_equal(t["mutations"]["to"], ot.mut_to)
        assert np.array_equal(t["missation_intervals"]["branch"], np.repeat(np.arange(ot.num_nodes), np.diff(ot.miss_offset)))
        assert np.array_equal(t["missation_intervals"]["start_site"], ot.miss_start) and np.array_equal(t["missation_intervals"]["end_site"], ot.miss_end)
        assert _close_in_float32(t["nodes"]["t"], _f32(ot.t)) and _close_in_float32(t["mutations"]["t"], _f32(ot.mut_t))
        r = Decoded(params_buf, "Params").value
        assert (r["step"], r["log_g"], r["log_coalescent_prior"]) == totals[k] and r["num_parts"] == parts
        changed += int(not np.array_equal(t["nodes"]["parent"], sc.tree.parent)) + int(not np.array_equal(t["nodes"]["t"], _f32(sc.tree.t)))
    assert changed == 6, "the samples do not differ from the starting tree: the moves left no trace in the file"
