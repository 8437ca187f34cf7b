// orc_run.hpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Restatement of the part of the reference's `Run` that sits either side of the local moves:
//   tree_partitioning.h:88-135   details::make_partition_part
//   tree_partitioning.h:139-194  generate_random_partition_stencil (tree.h:324-365 randomized_traversal)
//   tree_partitioning.h:196-239  partition_tree
//   run.cpp:87-108               Run::refresh_partition_stencils
//   run.cpp:110-193              Run::repartition (subtrees with frozen cut nodes and the synthetic sub-root lists, :141-153)
//   run.cpp:195-256              Run::reassemble
//   run.cpp:258-265              Run::normalize_root  (phylo_tree.cpp:309-322 rereference_to_root_sequence, orc_calc.hpp)
// It follows the reference step by step ON PURPOSE -- one walk from every sub-root to the tree's root, hash-free maps in
// ascending site order -- and shares nothing with the product's host driver (delphy_amd/csrc/emat_run.cpp), which carries
// cut-point states down the tree of cut points, and nothing with the kernels that cut the HBM-resident tree.
//
// Parity unpinned in the reference: no reference test exercises tree_partitioning.* or Run::repartition / reassemble
// (.codecov.yml excludes run.cpp).  What pins this file is the reference's own integrity rules restated in
// orc_part_check (phylo_tree.cpp:18-136), which every part and every reassembled tree must pass (tests/test_oracle_run.py).
//
// Random choices: the reference draws them from the run's std::mt19937 through std::bernoulli_distribution(0.5) and
// std::uniform_int_distribution -- streams no other implementation can reproduce.  Here they come from a `Bit_source` with
// the two operations the reference uses (a fair coin, an index below n); the tests hand it the product's generator
// (SplitMix64: one 64-bit output per decision, coin = top bit, index = high half of the 128-bit product), so that the two
// implementations are asked the same questions in the same order and must return the same partitions.
#ifndef ORC_RUN_HPP_
#define ORC_RUN_HPP_

#include <limits>
#include <set>

#include "orc_calc.hpp"

namespace orc {

struct Bit_source {   // stands for the reference's absl::BitGenRef over the run's std::mt19937
  uint64_t s = 0;
  explicit Bit_source(uint64_t seed = 0) : s(seed) {}
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  bool coin() { return (next() >> 63) != 0; }                                                    // std::bernoulli_distribution{0.5}
  int below(int n) { return (int)(((unsigned __int128)next() * (uint64_t)n) >> 64); }            // std::uniform_int_distribution{0, n-1}
};

// tree_partitioning.h:31-54
struct Partition_part_node {
  Node_index parent = k_no_node;
  Node_index children[2] = {k_no_node, k_no_node};
  Node_index orig_tree_index = k_no_node;
  bool is_tip() const { return children[0] == k_no_node; }
};
struct Partition_part {
  Node_index cut_point = k_no_node;
  Node_index root = k_no_node;
  std::vector<Partition_part_node> nodes;
  int size() const { return (int)nodes.size(); }
  Node_index add_node() { nodes.emplace_back(); return (Node_index)nodes.size() - 1; }
};
struct Partition {
  std::vector<Partition_part> parts;
  int root_part_index = -1;
  std::vector<int> tree_index_to_partition_index;
};

// tree_partitioning.h:88-135
inline Partition_part make_partition_part(const Phylo_tree& src_tree, Node_index cut_point, const std::set<Node_index>& src_cut_points) {
  Partition_part part; part.cut_point = cut_point;
  struct Work_item { Node_index src_node, dst_node; };
  std::vector<Work_item> work_stack;
  Node_index root_dst_node = part.add_node();
  part.root = root_dst_node;
  work_stack.push_back({cut_point, root_dst_node});
  while (!work_stack.empty()) {
    auto [src_node, dst_node] = work_stack.back();
    work_stack.pop_back();
    part.nodes[dst_node].orig_tree_index = src_node;
    bool src_node_is_cut_point = src_cut_points.count(src_node) != 0;
    if (src_tree.at(src_node).is_tip() || (src_node_is_cut_point && src_node != cut_point)) {
      part.nodes[dst_node].children[0] = part.nodes[dst_node].children[1] = k_no_node;
    } else {
      Node_index dst_left = part.add_node();
      Node_index dst_right = part.add_node();
      part.nodes[dst_node].children[0] = dst_left; part.nodes[dst_node].children[1] = dst_right;
      part.nodes[dst_left].parent = dst_node;
      part.nodes[dst_right].parent = dst_node;
      work_stack.push_back({src_tree.at(src_node).children[0], dst_left});
      work_stack.push_back({src_tree.at(src_node).children[1], dst_right});
    }
  }
  return part;
}

// tree.h:324-365: (node, children_so_far) visits with the two children of every inner node in random order; the
// post-order visits are the ones with children_so_far == number of children
template <class F>
inline void randomized_post_order_traversal(const Phylo_tree& tree, Bit_source& bitgen, F&& visit /* returns false to stop */) {
  if (tree.size() == 0) return;
  std::vector<std::pair<Node_index, int>> work_stack;
  work_stack.push_back({tree.root, -1});
  while (!work_stack.empty()) {
    auto [node, children_so_far] = work_stack.back();
    work_stack.pop_back();
    if (children_so_far != -1) {
      if (children_so_far == tree.at(node).num_children()) { if (!visit(node)) return; }
    } else {
      int num_children = tree.at(node).num_children();
      work_stack.push_back({node, num_children});
      if (num_children != 0) {
        ORC_CHECK(num_children == 2);
        if (bitgen.coin()) {
          work_stack.push_back({tree.at(node).children[0], -1});
          work_stack.push_back({node, 1});
          work_stack.push_back({tree.at(node).children[1], -1});
          work_stack.push_back({node, 0});
        } else {
          work_stack.push_back({tree.at(node).children[1], -1});
          work_stack.push_back({node, 1});
          work_stack.push_back({tree.at(node).children[0], -1});
          work_stack.push_back({node, 0});
        }
      }
    }
  }
}

// tree_partitioning.h:139-194
inline std::vector<Node_index> generate_random_partition_stencil(const Phylo_tree& tree, int num_parts, Bit_source& bitgen) {
  std::vector<Node_index> part_infos;   // Partition_part_info{cut_point}
  std::vector<int> descendants(tree.size(), 0);
  long num_branches_left = tree.size();
  int num_parts_left = num_parts;
  randomized_post_order_traversal(tree, bitgen, [&](Node_index node) {
    if (node == tree.root) return false;
    if ((int)part_infos.size() == num_parts - 1) return false;
    descendants[node] = 1;
    if (tree.at(node).is_inner_node()) for (Node_index child : tree.at(node).children) descendants[node] += descendants[child];
    long min_subtree_size = std::max(10L, num_branches_left / (num_parts_left + 1));
    if (descendants[node] >= min_subtree_size) {
      bool is_allowed = true;
      if (is_allowed && (num_branches_left - (descendants[node] - 1)) < min_subtree_size) is_allowed = false;
      if (is_allowed && bitgen.coin()) is_allowed = false;
      if (is_allowed) {
        Node_index cut_point = node;
        int num_branches_in_part = descendants[cut_point] - 1;
        num_branches_left -= num_branches_in_part;
        ORC_CHECK(num_branches_left >= 0);
        part_infos.push_back(cut_point);
        descendants[cut_point] = 1;
        --num_parts_left;
      }
    }
    return true;
  });
  return part_infos;
}

// tree_partitioning.h:196-239
inline Partition partition_tree(const Phylo_tree& tree, const std::vector<Node_index>& stencil) {
  int root_part_index = (int)stencil.size();
  bool root_in_stencil = false;
  for (int i = 0; i != (int)stencil.size(); ++i)
    if (stencil[i] == tree.root) { root_in_stencil = true; root_part_index = i; break; }
  int num_partitions = (int)stencil.size() + (root_in_stencil ? 0 : 1);
  Partition partition; partition.parts.resize(num_partitions);
  std::set<Node_index> partition_cut_points(stencil.begin(), stencil.end());
  if (!root_in_stencil) { partition_cut_points.insert(tree.root); root_part_index = num_partitions - 1; }
  partition.root_part_index = root_part_index;
  for (int i = 0; i != num_partitions; ++i) {
    Node_index cut_point = (i == partition.root_part_index) ? tree.root : stencil[i];
    partition.parts[i] = make_partition_part(tree, cut_point, partition_cut_points);
  }
  partition.tree_index_to_partition_index.assign(tree.size(), -1);
  for (int i = 0; i != num_partitions; ++i)
    for (const auto& pn : partition.parts[i].nodes) partition.tree_index_to_partition_index[pn.orig_tree_index] = i;
  return partition;
}

// The stretch of Run between two global moves, without the Subruns themselves (they are orc::Subrun, driven separately).
struct Run {
  Phylo_tree tree;
  Bit_source bitgen;
  int num_parts = 1;
  std::vector<std::vector<Node_index>> partition_stencils;
  int repartitions_until_refresh = 0;
  Partition tree_partition;
  std::vector<Phylo_tree> subtrees;           // what the Subruns are constructed from (run.cpp:182-183)

  Run(Phylo_tree t, uint64_t seed, int parts) : tree(std::move(t)), bitgen(seed), num_parts(parts) {}

  // run.cpp:87-108.  The reference refreshes when step_ passes next_partition_stencil_refresh_step_ = step_ + 200 cycles' worth
  // of local moves; with one repartition per cycle that is every 200th repartition.
  void refresh_partition_stencils() {
    if (!partition_stencils.empty() && repartitions_until_refresh > 0) return;
    partition_stencils.clear();
    for (int i = 0; i != 10; ++i) partition_stencils.push_back(generate_random_partition_stencil(tree, num_parts, bitgen));
    repartitions_until_refresh = 200;
  }

  void normalize_root() { rereference_to_root_sequence(tree); }   // run.cpp:258-265 (the state-frequency bookkeeping is not on the path)

  // run.cpp:110-193
  void repartition() {
    refresh_partition_stencils();
    --repartitions_until_refresh;
    int partition_stencils_index = bitgen.below((int)partition_stencils.size());
    const auto& partition_stencil = partition_stencils.at(partition_stencils_index);
    tree_partition = partition_tree(tree, partition_stencil);

    ORC_CHECK(tree.at_root().missations.from_states.empty());
    normalize_root();

    subtrees.clear();
    for (const auto& partition_part : tree_partition.parts) {
      Node_index subroot = partition_part.cut_point;
      Site_deltas subroot_seq_deltas = deltas_ref_to_loc(tree, tree.node_loc(subroot));   // view_of_sequence_at(tree_, subroot).deltas()
      const auto& ref_seq = tree.ref_sequence;

      Phylo_tree subtree(partition_part.size());
      subtree.ref_sequence = ref_seq;
      subtree.root = partition_part.root;                                                  // copy_topology
      for (int i = 0; i < partition_part.size(); ++i) {
        subtree.at(i).parent = partition_part.nodes[i].parent;
        subtree.at(i).children[0] = partition_part.nodes[i].children[0];
        subtree.at(i).children[1] = partition_part.nodes[i].children[1];
      }

      // Missations above the subroot logically come before the mutations above it, so all their from_states match the reference
      Missation_map root_missation_map;
      root_missation_map.intervals = reconstruct_missing_sites_at(tree, subroot);
      root_missation_map.from_states.clear();

      Mutation_list root_mutations;
      for (const auto& [l, d] : subroot_seq_deltas) {
        if (!root_missation_map.contains(l)) root_mutations.push_back(Mutation{ref_seq[l], l, d.to, -std::numeric_limits<double>::max()});
      }
      sort_mutations(root_mutations);

      for (int partition_node = 0; partition_node < partition_part.size(); ++partition_node) {   // index_order_traversal
        Node_index node = partition_part.nodes[partition_node].orig_tree_index;
        Node_index subtree_node = partition_node;
        const auto& mutation_list = (node == subroot) ? root_mutations : tree.at(node).mutations;
        const auto& missation_map = (node == subroot) ? root_missation_map : tree.at(node).missations;
        subtree.at(subtree_node).t = tree.at(node).t;
        if (subtree.at(subtree_node).is_tip() && !tree.at(node).is_tip()) {
          // a "tip" of the subtree that is a frozen inner node of the tree
          subtree.at(subtree_node).t_min = (float)tree.at(node).t;
          subtree.at(subtree_node).t_max = (float)tree.at(node).t;
        } else {
          subtree.at(subtree_node).t_min = tree.at(node).t_min;
          subtree.at(subtree_node).t_max = tree.at(node).t_max;
        }
        subtree.at(subtree_node).mutations = mutation_list;
        subtree.at(subtree_node).missations = missation_map;
      }
      subtrees.push_back(std::move(subtree));
    }
  }

  // run.cpp:195-256, given the trees the Subruns hold after their moves
  void reassemble(const std::vector<Phylo_tree>& subrun_trees) {
    ORC_CHECK(subrun_trees.size() == tree_partition.parts.size());
    int num_partitions = (int)tree_partition.parts.size();
    for (int i = 0; i != num_partitions; ++i) {
      const auto& partition_part = tree_partition.parts[i];
      const auto& subtree = subrun_trees[i];
      ORC_CHECK(subtree.size() == partition_part.size());
      ORC_CHECK(subtree.at_root().missations.from_states.empty());
      for (Node_index subnode : post_order(subtree)) {
        Node_index node = partition_part.nodes[subnode].orig_tree_index;
        tree.at(node).t = subtree.at(subnode).t;
        if (subnode != subtree.root) {
          tree.at(node).mutations = subtree.at(subnode).mutations;
          tree.at(node).missations = subtree.at(subnode).missations;
        }
        if (subtree.at(subnode).is_inner_node()) {   // (not tree.at(node).is_inner_node(): cut points are handled as sub-roots only)
          Node_index subleft = subtree.at(subnode).children[0], subright = subtree.at(subnode).children[1];
          Node_index left = partition_part.nodes[subleft].orig_tree_index, right = partition_part.nodes[subright].orig_tree_index;
          tree.at(node).children[0] = left; tree.at(node).children[1] = right;
          tree.at(left).parent = node;
          tree.at(right).parent = node;
        }
      }
      if (i == tree_partition.root_part_index) {   // subrun.includes_run_root()
        Node_index subroot = subtree.root;
        Node_index subroot_in_main = partition_part.nodes[subroot].orig_tree_index;
        tree.root = subroot_in_main;
        tree.at_root().parent = k_no_node;
        tree.at_root().mutations = subtree.at(subroot).mutations;
        tree.at_root().missations = subtree.at(subroot).missations;
        ORC_CHECK(tree.ref_sequence == subtree.ref_sequence);
      }
    }
  }
};

}  // namespace orc
#endif  // ORC_RUN_HPP_
