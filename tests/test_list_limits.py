"""Per-node list counts are 16 bits wide inside the engine (ListRef, delphy_amd/csrc/emat_slab.hpp).  Nothing may ever be cut
to fit: a list longer than the boundary accepts is refused with EMAT_ERR_CAPACITY and a message that names the numbers --
at emat_part_upload, at emat_tree_upload, and when a repartition of the HBM-resident tree would give a part's sub-root a longer
synthetic delta list (run.cpp:141-153) than the device path holds.  Genomes longer than SARS-CoV-2's (the reference carries an
mpox set-up of 197 kb, run.cpp:400-435) are where that can happen.
"""
import numpy as np
import pytest

import delphy_amd as d

L = 200_000
NEG = -np.finfo(np.float64).max
FMAX = np.finfo(np.float32).max


def _ref():
    return (np.arange(L) % 4).astype(np.uint8)


def _cherry_with_subroot_deltas(n_deltas):
    """A three-node part (sub-root + two dated tips) whose sub-root carries `n_deltas` reference -> sub-root sequence deltas."""
    ref = _ref()
    t = d.FlatTree.empty(3, n_deltas, 0, 0)
    t.root = 0
    t.parent[:] = [-1, 0, 0]; t.child0[:] = [1, -1, -1]; t.child1[:] = [2, -1, -1]
    t.t[:] = [0.0, 10.0, 12.0]
    t.t_min[:] = [-FMAX, 10.0, 12.0]; t.t_max[:] = [FMAX, 10.0, 12.0]
    site = np.arange(n_deltas, dtype=np.int32) * 2
    t.mut_site[:n_deltas] = site; t.mut_from[:n_deltas] = ref[site]; t.mut_to[:n_deltas] = (ref[site] + 1) % 4; t.mut_t[:n_deltas] = NEG
    t.mut_offset[:] = [0, n_deltas, n_deltas, n_deltas]
    return t, ref


def test_a_70000_entry_subroot_delta_list_is_refused_at_upload_with_the_numbers():
    """Host-only handle (device = -1: staging works without a GPU): the upload of a part whose sub-root carries 70 000 deltas on a
    200 000-site genome fails loudly; 16 000 -- the documented limit -- is accepted."""
    b = d.EmatBackend(L, device=-1)
    try:
        big, ref = _cherry_with_subroot_deltas(70_000)
        b.set_ref_sequence(ref)
        with pytest.raises(d.EmatError, match=r"EMAT_ERR_CAPACITY.*70000 mutations.*at most 16000"):
            b.upload_parts([big], [False], [1])
        ok, _ = _cherry_with_subroot_deltas(16_000)
        b.upload_parts([ok], [False], [1])
        over, _ = _cherry_with_subroot_deltas(16_001)
        with pytest.raises(d.EmatError, match="EMAT_ERR_CAPACITY"):
            b.upload_parts([over], [False], [1])
    finally:
        b.close()


def _caterpillar(per_branch, depth):
    """root -> A1 -> ... -> A_depth, every A_k with a dated tip beside it and `per_branch` mutations at fresh sites on the branch
    above it: the sequence at A_depth differs from the root's at depth * per_branch sites."""
    ref = _ref()
    n = 2 * depth + 3                      # root, A_1..A_depth, one tip per inner node, two tips under A_depth
    t = d.FlatTree.empty(n, per_branch * depth, 0, 0)
    inner = list(range(depth + 1))         # 0 = root, k = A_k
    tip_of = [depth + 1 + k for k in range(depth + 1)]
    last_tip = 2 * depth + 2
    t.root = 0
    t.t_min[:] = -FMAX; t.t_max[:] = FMAX
    counts = np.zeros(n, np.int64)
    for k in range(depth + 1):
        t.t[k] = 10.0 * k
        t.child0[k] = tip_of[k]
        t.child1[k] = inner[k + 1] if k < depth else last_tip
        t.parent[t.child0[k]] = k; t.parent[t.child1[k]] = k
        if k > 0:
            counts[k] = per_branch
    for tip in tip_of + [last_tip]:
        t.t[tip] = 10.0 * depth + 20.0 + tip
        t.t_min[tip] = t.t_max[tip] = np.float32(t.t[tip]); t.t[tip] = float(t.t_min[tip])
    t.mut_offset[:] = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    for k in range(1, depth + 1):
        lo = int(t.mut_offset[k])
        site = np.arange((k - 1) * per_branch, k * per_branch, dtype=np.int32)
        t.mut_site[lo:lo + per_branch] = site; t.mut_from[lo:lo + per_branch] = ref[site]; t.mut_to[lo:lo + per_branch] = (ref[site] + 1) % 4
        t.mut_t[lo:lo + per_branch] = 10.0 * (k - 1) + 10.0 * (np.arange(per_branch) + 0.5) / per_branch    # sorted by (t, site)
    return t, ref, depth                   # A_depth is node `depth`


@pytest.mark.gpu
def test_a_cut_point_70000_changes_from_the_root_is_refused_by_the_resident_tree_path_and_by_the_upload():
    """The same on the GPU box, on both routes a part can take into the engine.  (a) The tree resident in HBM: every branch list is
    within the limit (14 000), but the cut point's sequence differs from the root's at 70 000 sites -- the repartition must refuse
    it (the cut-point state the kernels hold, and the 16-bit list counts of the slab it would have to write) instead of writing
    a slab with a wrapped count.  (b) The host route: emat_part_upload of that part with its 70 000-entry sub-root list."""
    tree, ref, cut = _caterpillar(14_000, 5)
    b = d.EmatBackend(L)
    try:
        b.set_ref_sequence(ref); b.set_hky(1e-3 / 365.0, 5.0, [0.31, 0.19, 0.21, 0.29]); b.set_flags(float(tree.t.max()))
        b.tree_upload(tree)
        n, root_part, po, orig, k0, k1 = b.tree_partition([cut])
        assert n == 2
        with pytest.raises(d.EmatError, match=r"emat_tree_repartition: part \d+: .*(cut point|16 000 entries)"):
            b.tree_repartition(po, orig, k0, k1, root_part, [3, 4], d.PopModel.const(365.0), 5.0)
        big, _ = _cherry_with_subroot_deltas(70_000)
        with pytest.raises(d.EmatError, match=r"EMAT_ERR_CAPACITY.*at most 16000"):
            b.upload_parts([big], [False], [1])
        # and a list that IS within the limit goes through both routes and runs
        small, ref2, cut2 = _caterpillar(300, 5)
        b2 = d.EmatBackend(L)
        try:
            b2.set_ref_sequence(ref2); b2.set_hky(1e-3 / 365.0, 5.0, [0.31, 0.19, 0.21, 0.29]); b2.set_flags(float(small.t.max()))
            b2.tree_upload(small)
            n, root_part, po, orig, k0, k1 = b2.tree_partition([cut2])
            b2.tree_repartition(po, orig, k0, k1, root_part, [3, 4], d.PopModel.const(365.0), 5.0)
            sub = b2.part_download(1 - root_part)
            assert sub.mut_offset[1] - sub.mut_offset[0] == 1500          # the cut point's synthetic delta list (run.cpp:141-153)
            b2.run_moves_per_part(500); b2.synchronize()
            part, dev4 = b2.check_derived(1.0)
            assert dev4[3] == 0
        finally:
            b2.close()
    finally:
        b.close()
