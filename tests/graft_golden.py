"""The reference's SPR fixtures and expectations (tests/golden/reference_expectations.json, block "spr_move": generated from
/root/reference/tests/spr_move_tests.cpp by tests/golden/make_reference_expectations.py) turned into engine inputs and checks.
Used with the oracle on the CPU and, through the test hook emat_debug_graft, with the HIP engine's own device code."""
import json
import os

import numpy as np

import delphy_amd as d

HERE = os.path.dirname(os.path.abspath(__file__))


def spr_move_block():
    return json.load(open(os.path.join(HERE, "golden", "reference_expectations.json")))["spr_move"]


def fixture_tree(fx) -> d.FlatTree:
    """A fixture of the JSON as a flat tree: missations become maximal intervals plus from-state entries where they differ from
    the reference sequence (Missation_map, reference core/mutations.h:150-232)."""
    nodes, ref = fx["nodes"], fx["ref_sequence"]
    n = len(nodes)
    mo, ms, mf, mt, mtt = [0], [], [], [], []
    io, i0, i1 = [0], [], []
    fo, fs, fst = [0], [], []
    t = d.FlatTree.empty(n, 0, 0, 0)
    for i, nd in enumerate(nodes):
        t.parent[i] = nd["parent"]
        if nd["children"]:
            t.child0[i], t.child1[i] = nd["children"]
        t.t[i], t.t_min[i], t.t_max[i] = nd["t"], np.float32(nd["t_min"]), np.float32(nd["t_max"])
        for fr, site, to, tm in nd["mutations"]:
            ms.append(site); mf.append(fr); mt.append(to); mtt.append(tm)
        mo.append(len(ms))
        sites = sorted(l for l, _ in nd["missations"])
        for l in sites:
            if i1 and len(i1) > io[-1] and i1[-1] == l:
                i1[-1] = l + 1
            else:
                i0.append(l); i1.append(l + 1)
        io.append(len(i0))
        for l, a in sorted(nd["missations"]):
            if a != ref[l]:
                fs.append(l); fst.append(a)
        fo.append(len(fs))
    return d.FlatTree(fx["root"], t.parent, t.child0, t.child1, t.t, t.t_min, t.t_max,
                      np.asarray(mo, np.int32), np.asarray(ms, np.int32), np.asarray(mf, np.uint8), np.asarray(mt, np.uint8), np.asarray(mtt, np.float64),
                      np.asarray(io, np.int32), np.asarray(i0, np.int32), np.asarray(i1, np.int32),
                      np.asarray(fo, np.int32), np.asarray(fs, np.int32), np.asarray(fst, np.uint8))


def configure_fixture(engine, fx, can_change_root, seed=12345):
    """Model, flags and the fixture tree as the engine's only part (`can_change_root` is Spr_move's flag = Subrun::includes_run_root)."""
    ref = np.asarray(fx["ref_sequence"], np.uint8)
    evo = fx["evo"]
    engine.set_ref_sequence(ref)
    engine.set_evo(np.asarray(evo["mu"]), np.asarray(evo["pi"]), np.asarray(evo["q"]), np.asarray(evo["nu_l"]), np.asarray(evo["partition_for_site"], np.int32))
    t_max_tip = max(nd["t"] for nd in fx["nodes"])
    engine.set_flags(t_max_tip, False, True)
    engine.upload_parts([fixture_tree(fx)], [can_change_root], [seed])
    engine.build_coalescent_parts(d.PopModel.exp(t_max_tip, 10.0, 0.0, 0.0), 0, 0.5)   # the hooks never touch the coalescent prior; a launch wants one


def sites_of(intervals):
    return [l for s, e in intervals for l in range(s, e)]


def check_analysis(got, want, what):
    """One analyze_graft test of the reference against what an engine's analyze_graft found (`got` = a decoded graft)."""
    tol = want["tol"]
    assert len(got["branch_infos"]) == want["num_branch_infos"], what
    for wb in want["branch_infos"]:
        gb = got["branch_infos"][wb["index"]]
        w = "%s branch info %d" % (what, wb["index"])
        assert gb["A"] == wb["A"] and gb["B"] == wb["B"] and gb["is_open"] == wb["is_open"] and gb["T_to_X"] == wb["T_to_X"], (w, gb, wb)
        for f in ("partial_lambda_at_A", "partial_lambda_at_X"):
            assert abs(gb[f] - wb[f]) <= tol, (w, f, gb[f], wb[f])
        for f in ("warm_sites", "hot_sites"):
            c = wb[f]
            if "intervals" in c:
                assert gb[f] == c["intervals"], (w, f, gb[f], c)
            have = set(sites_of(gb[f]))
            assert set(c.get("contains", [])) <= have and not (set(c.get("not_contains", [])) & have), (w, f, gb[f], c)
        assert len(gb["hot_muts_to_X"]) == len(wb["hot_muts_to_X"]), (w, gb["hot_muts_to_X"], wb["hot_muts_to_X"])
        for gm, wm in zip(gb["hot_muts_to_X"], wb["hot_muts_to_X"]):          # ElementsAre: in order
            assert gm[:3] == wm[:3] and gm[3] == wm[3], (w, gm, wm)
        assert sorted(gb["hot_deltas_to_X"]) == sorted(wb["hot_deltas_to_X"]), (w, gb["hot_deltas_to_X"], wb["hot_deltas_to_X"])
    assert abs(got["log_alpha_mut"] - want["log_alpha_mut"]) <= tol, (what, got["log_alpha_mut"], want["log_alpha_mut"])
    assert abs(got["delta_log_G"] - want["delta_log_G"]) <= tol, (what, got["delta_log_G"], want["delta_log_G"])


def node_lists(tree: d.FlatTree, ref, node):
    """(sorted mutations [from, site, to, t], missations [site, from] in site order) of one node of a downloaded part."""
    a, b = int(tree.mut_offset[node]), int(tree.mut_offset[node + 1])
    muts = sorted([int(tree.mut_from[k]), int(tree.mut_site[k]), int(tree.mut_to[k]), float(tree.mut_t[k])] for k in range(a, b))
    over = {int(tree.mfs_site[k]): int(tree.mfs_state[k]) for k in range(int(tree.mfs_offset[node]), int(tree.mfs_offset[node + 1]))}
    miss = []
    for k in range(int(tree.miss_offset[node]), int(tree.miss_offset[node + 1])):
        for l in range(int(tree.miss_start[k]), int(tree.miss_end[k])):
            miss.append([l, over.get(l, int(ref[l]))])
    return muts, miss


def check_peel_apply(result, tree, fx, want, what):
    """A peel / closed-mutations / peel-and-reapply test of the reference against the decoded hook output and the part's tree afterwards."""
    if "count_closed_mutations" in want:
        assert result["count_closed_mutations"] == want["count_closed_mutations"], (what, result)
    if "closed_deltas" in want:
        assert sorted(result["closed_deltas"]) == sorted(want["closed_deltas"]), (what, result["closed_deltas"])
    for node, lists in want["nodes"].items():
        muts, miss = node_lists(tree, fx["ref_sequence"], int(node))
        if "mutations_unordered" in lists:
            assert muts == sorted(lists["mutations_unordered"]), (what, node, muts, lists["mutations_unordered"])
        if "missations" in lists:
            assert miss == lists["missations"], (what, node, miss, lists["missations"])


def tip_sequences(tree: d.FlatTree, ref):
    """{tip: (states with -1 where missing)}: the reference sequence with every list from the root down to the tip applied
    (view_of_sequence_at / reconstruct_missing_sites_at, reference phylo_tree_calc.cpp:19-56)."""
    out = {}
    for tip in range(tree.num_nodes):
        if tree.child0[tip] >= 0:
            continue
        path, n = [], tip
        while n >= 0:
            path.append(n); n = int(tree.parent[n])
        seq = np.array(ref, np.int64)
        missing = np.zeros(len(ref), bool)
        for n in reversed(path):
            for k in range(int(tree.mut_offset[n]), int(tree.mut_offset[n + 1])):
                l = int(tree.mut_site[k])
                if not missing[l]:
                    assert seq[l] == tree.mut_from[k], "mutation %d of node %d starts from another state than the sequence holds" % (k, n)
                seq[l] = tree.mut_to[k]
            for k in range(int(tree.miss_offset[n]), int(tree.miss_offset[n + 1])):
                missing[int(tree.miss_start[k]): int(tree.miss_end[k])] = True
        seq[missing] = -1
        out[tip] = seq
    return out


def same_grafts(a, b, tol=1e-6):
    """new_graft_redux against new_graft as the reference compares them (tests/spr_move_tests.cpp:512-548)."""
    assert a["X"] == b["X"] and a["S"] == b["S"] and abs(a["t_P"] - b["t_P"]) <= tol and len(a["branch_infos"]) == len(b["branch_infos"])
    for x, y in zip(a["branch_infos"], b["branch_infos"]):
        assert x["A"] == y["A"] and x["B"] == y["B"] and x["is_open"] == y["is_open"] and abs(x["T_to_X"] - y["T_to_X"]) <= tol
        assert abs(x["partial_lambda_at_A"] - y["partial_lambda_at_A"]) <= tol and abs(x["partial_lambda_at_X"] - y["partial_lambda_at_X"]) <= tol
        assert len(x["hot_muts_to_X"]) == len(y["hot_muts_to_X"])
        for m, n in zip(x["hot_muts_to_X"], y["hot_muts_to_X"]):
            assert m[:3] == n[:3] and abs(m[3] - n[3]) <= tol
        assert sorted(x["hot_deltas_to_X"]) == sorted(y["hot_deltas_to_X"])
    assert abs(a["delta_log_G"] - b["delta_log_G"]) <= tol and abs(a["log_alpha_mut"] - b["log_alpha_mut"]) <= tol
