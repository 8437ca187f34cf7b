// flat_tree.hpp -- host-side owner of an `emat_flat_tree` image (struct-of-arrays + CSR lists).
//
// This is the boundary format of the engine: the reference's `Phylo_tree` (core/phylo_tree.h:14-64:
// AoS nodes, each owning a std::vector<Mutation> and a Missation_map) flattened into a handful of
// contiguous arrays that can be handed over the C-ABI and copied to HBM without pointer chasing.
#ifndef EMAT_FLAT_TREE_HPP_
#define EMAT_FLAT_TREE_HPP_

#include <cfloat>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/emat_backend.h"

namespace emat {

struct FlatTree {
  int32_t root = EMAT_NO_NODE;
  std::vector<int32_t> parent, child0, child1;
  std::vector<double> t;
  std::vector<float> t_min, t_max;
  std::vector<int32_t> mut_offset;   // [n+1]
  std::vector<int32_t> mut_site;
  std::vector<uint8_t> mut_from, mut_to;
  std::vector<double> mut_t;
  std::vector<int32_t> miss_offset;  // [n+1]
  std::vector<int32_t> miss_start, miss_end;
  std::vector<int32_t> mfs_offset;   // [n+1]
  std::vector<int32_t> mfs_site;
  std::vector<uint8_t> mfs_state;

  int32_t num_nodes() const { return (int32_t)parent.size(); }
  int32_t num_muts() const { return (int32_t)mut_site.size(); }
  int32_t num_intervals() const { return (int32_t)miss_start.size(); }
  int32_t num_from_states() const { return (int32_t)mfs_site.size(); }
  bool is_tip(int32_t n) const { return child0[n] == EMAT_NO_NODE; }

  void resize_nodes(int32_t n) {
    parent.assign(n, EMAT_NO_NODE); child0.assign(n, EMAT_NO_NODE); child1.assign(n, EMAT_NO_NODE);
    t.assign(n, 0.0); t_min.assign(n, -FLT_MAX); t_max.assign(n, FLT_MAX);
    mut_offset.assign(n + 1, 0); miss_offset.assign(n + 1, 0); mfs_offset.assign(n + 1, 0);
  }
  // Size every array for a download of the given shape.
  void allocate(int32_t n, int32_t nm, int32_t ni, int32_t nf) {
    resize_nodes(n);
    mut_site.assign(nm, 0); mut_from.assign(nm, 0); mut_to.assign(nm, 0); mut_t.assign(nm, 0.0);
    miss_start.assign(ni, 0); miss_end.assign(ni, 0);
    mfs_site.assign(nf, 0); mfs_state.assign(nf, 0);
  }
  // Non-owning C view (valid while *this is alive and unmodified).
  emat_flat_tree view() {
    emat_flat_tree v;
    v.num_nodes = num_nodes(); v.root = root;
    v.parent = parent.data(); v.child0 = child0.data(); v.child1 = child1.data();
    v.t = t.data(); v.t_min = t_min.data(); v.t_max = t_max.data();
    v.mut_offset = mut_offset.data(); v.mut_site = mut_site.data(); v.mut_from = mut_from.data();
    v.mut_to = mut_to.data(); v.mut_t = mut_t.data();
    v.miss_offset = miss_offset.data(); v.miss_start = miss_start.data(); v.miss_end = miss_end.data();
    v.mfs_offset = mfs_offset.data(); v.mfs_site = mfs_site.data(); v.mfs_state = mfs_state.data();
    v.cap_muts = num_muts(); v.cap_intervals = num_intervals(); v.cap_from_states = num_from_states();
    return v;
  }
  static FlatTree from_view(const emat_flat_tree& v) {
    FlatTree f;
    int32_t n = v.num_nodes;
    f.root = v.root;
    f.parent.assign(v.parent, v.parent + n); f.child0.assign(v.child0, v.child0 + n); f.child1.assign(v.child1, v.child1 + n);
    f.t.assign(v.t, v.t + n); f.t_min.assign(v.t_min, v.t_min + n); f.t_max.assign(v.t_max, v.t_max + n);
    f.mut_offset.assign(v.mut_offset, v.mut_offset + n + 1);
    int32_t nm = v.mut_offset[n];
    f.mut_site.assign(v.mut_site, v.mut_site + nm); f.mut_from.assign(v.mut_from, v.mut_from + nm);
    f.mut_to.assign(v.mut_to, v.mut_to + nm); f.mut_t.assign(v.mut_t, v.mut_t + nm);
    f.miss_offset.assign(v.miss_offset, v.miss_offset + n + 1);
    int32_t ni = v.miss_offset[n];
    f.miss_start.assign(v.miss_start, v.miss_start + ni); f.miss_end.assign(v.miss_end, v.miss_end + ni);
    f.mfs_offset.assign(v.mfs_offset, v.mfs_offset + n + 1);
    int32_t nf = v.mfs_offset[n];
    f.mfs_site.assign(v.mfs_site, v.mfs_site + nf); f.mfs_state.assign(v.mfs_state, v.mfs_state + nf);
    return f;
  }
};

// Structural validation of an uploaded image (the product's analogue of the reference's
// assert_tree_integrity, core/tree.h:371-412).  Returns "" when fine.
inline std::string validate_flat_tree(const emat_flat_tree& v, int32_t num_sites) {
  const int32_t n = v.num_nodes;
  if (n < 1) return "empty tree";
  if (v.root < 0 || v.root >= n) return "root out of range";
  if (v.parent[v.root] != EMAT_NO_NODE) return "root has a parent";
  if (v.mut_offset[0] != 0 || v.miss_offset[0] != 0 || v.mfs_offset[0] != 0) return "CSR offsets must start at 0";
  for (int32_t i = 0; i < n; ++i) {
    if (v.mut_offset[i + 1] < v.mut_offset[i] || v.miss_offset[i + 1] < v.miss_offset[i] || v.mfs_offset[i + 1] < v.mfs_offset[i])
      return "CSR offsets must be non-decreasing";
    const bool tip = v.child0[i] == EMAT_NO_NODE;
    if (tip != (v.child1[i] == EMAT_NO_NODE)) return "node with exactly one child";
    if (!tip) {
      for (int32_t c : {v.child0[i], v.child1[i]}) {
        if (c < 0 || c >= n || v.parent[c] != i) return "child/parent links inconsistent";
      }
    }
    if (i != v.root && (v.parent[i] < 0 || v.parent[i] >= n)) return "parent out of range";
    int32_t prev_end = -1;
    for (int32_t k = v.miss_offset[i]; k < v.miss_offset[i + 1]; ++k) {
      if (v.miss_start[k] < 0 || v.miss_end[k] > num_sites || v.miss_start[k] >= v.miss_end[k] || v.miss_start[k] <= prev_end)
        return "missation intervals must be sorted, disjoint, non-adjacent and in range";
      prev_end = v.miss_end[k];
    }
    for (int32_t k = v.mut_offset[i]; k < v.mut_offset[i + 1]; ++k) {
      if (v.mut_site[k] < 0 || v.mut_site[k] >= num_sites || v.mut_from[k] > 3 || v.mut_to[k] > 3 || v.mut_from[k] == v.mut_to[k])
        return "bad mutation";
    }
    int32_t prev_site = -1;
    for (int32_t k = v.mfs_offset[i]; k < v.mfs_offset[i + 1]; ++k) {
      if (v.mfs_site[k] <= prev_site || v.mfs_site[k] >= num_sites || v.mfs_state[k] > 3) return "bad missation from_state";
      prev_site = v.mfs_site[k];
    }
  }
  return "";
}

// No node may carry more than `max_list` mutations, missation intervals or from-states: the engine's per-node list counts are 16 bits
// wide (ListRef, emat_slab.hpp), so a longer list is refused at the boundary -- loudly, with the numbers -- never truncated.
// (A genome longer than SARS-CoV-2's can get there: a sub-root's synthetic delta list has up to one entry per site.)  "" when fine.
inline std::string flat_tree_list_limit(const emat_flat_tree& v, int32_t max_list) {
  for (int32_t i = 0; i < v.num_nodes; ++i) {
    const int32_t nm = v.mut_offset[i + 1] - v.mut_offset[i], ni = v.miss_offset[i + 1] - v.miss_offset[i], nf = v.mfs_offset[i + 1] - v.mfs_offset[i];
    if (nm > max_list || ni > max_list || nf > max_list)
      return "node " + std::to_string(i) + " carries " + std::to_string(nm) + " mutations, " + std::to_string(ni) + " missation intervals and " + std::to_string(nf) +
             " from-states: the engine holds at most " + std::to_string(max_list) + " entries per node and list (16-bit list counts)";
  }
  return "";
}

}  // namespace emat
#endif  // EMAT_FLAT_TREE_HPP_
