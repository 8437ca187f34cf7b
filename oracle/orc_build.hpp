// orc_build.hpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Initial-tree construction, SURVEY.md 8(f).4: the reference's UShER-like builder and what it leans on, restated:
//   phylo_tree.cpp:138-202   assert_phylo_tree_matches_tip_descs   (the builder's own closing check)
//   phylo_tree.cpp:414-507   fix_up_missations
//   phylo_tree.cpp:567-575   randomize_mutation_times              (randomize_branch_mutation_times is in orc_calc.hpp)
//   phylo_tree.cpp:796-1049  build_usher_like_tree
//   dates.cpp:63-82          pseudo_date
//   tree.h:243-270           traversal (entering / between / leaving visits)
// Pinned to the reference's own cases of fix_up_missations (tests/phylo_tree_tests.cpp:539-760, re-evaluated with its fixtures and
// expected lists in orc_tests.cpp).  The builder itself has no known-answer test in the reference: what pins it is the check the
// reference runs at its end -- every tip's sequence and missing sites reproduce its descriptor, and the tree passes the
// integrity rules -- restated here and run on every tree the tests build.
//
// Random numbers: the reference draws from its std::mt19937 through absl::Uniform with four kinds of interval; here they come from
// the engine's counter-based stream (orc::Rng), one 64-bit draw each, in the reference's order of draws; hash-map iterations of the
// reference (the deltas of the new tip's branch) run in ascending site order.
#ifndef ORC_BUILD_HPP_
#define ORC_BUILD_HPP_

#include <set>

#include "orc_spr.hpp"

namespace orc {

struct Tip_desc {   // phylo_tree.h:137-143 (names are not on the path)
  float t_min = +42.0f, t_max = -42.0f;
  std::vector<Seq_delta> seq_deltas;
  Missation_map missations;
};

// absl::Uniform(absl::IntervalClosedClosed, bitgen, lo, hi): one draw mapped onto [0, 1]
inline double uniform_cc(Rng& rng, double lo, double hi) { return lo + (hi - lo) * ((double)(rng.next64() >> 11) * (1.0 / 9007199254740991.0)); }

// tree.h:243-270: (node, children_so_far) for children_so_far = 0 .. number of children; a tip is visited once, entering and leaving at once
template <class F> inline void traversal(const Phylo_tree& tree, F&& visit) {
  if (tree.size() == 0) return;
  std::vector<std::pair<Node_index, int>> work_stack;
  work_stack.push_back({tree.root, -1});
  while (!work_stack.empty()) {
    auto [node, children_so_far] = work_stack.back();
    work_stack.pop_back();
    if (children_so_far != -1) visit(node, children_so_far);
    else {
      const int num_children = tree.at(node).num_children();
      work_stack.push_back({node, num_children});
      for (int i = num_children - 1; i >= 0; --i) {
        work_stack.push_back({tree.at(node).children[i], -1});
        work_stack.push_back({node, i});
      }
    }
  }
}

// Sequence_overlay over the reference sequence: only the sites that differ are kept (sequence_overlay.h)
struct Seq_overlay {
  const std::vector<State>* base;
  std::map<Site_index, State> deltas;
  explicit Seq_overlay(const std::vector<State>& b) : base(&b) {}
  State get(Site_index l) const { auto it = deltas.find(l); return it != deltas.end() ? it->second : (*base)[l]; }
  void set(Site_index l, State s) { if (s == (*base)[l]) deltas.erase(l); else deltas[l] = s; }
};

// phylo_tree.cpp:414-507
inline void fix_up_missations(Phylo_tree& tree) {
  const auto& ref_seq = tree.ref_sequence;
  // First, bubble up any common missations from the leaves to the root
  for (Node_index node : post_order(tree)) {
    if (tree.at(node).is_inner_node()) {
      Node_index L = tree.at(node).children[0], R = tree.at(node).children[1];
      if (interval_sets_intersect(tree.at(L).missations.intervals, tree.at(R).missations.intervals)) {
        if (tree.at(node).missations.empty()) {
          factor_out_common_missations(tree.at(L).missations, tree.at(R).missations, tree.at(node).missations);
        } else {
          Missation_map extra_node_missations;
          factor_out_common_missations(tree.at(L).missations, tree.at(R).missations, extra_node_missations);
          tree.at(node).missations = merge_missations_nondestructively(tree.at(node).missations, extra_node_missations);
        }
      }
    }
  }
  // Remove any redundant downstream missations
  std::set<Site_index> cur_missing_sites;
  Seq_overlay cur_seq(ref_seq);
  traversal(tree, [&](Node_index node, int children_so_far) {
    bool needs_fixing = false;
    if (children_so_far == 0) {
      for (const auto& m : tree.at(node).mutations) { ORC_CHECK(m.from == cur_seq.get(m.site)); cur_seq.set(m.site, m.to); }
      for (const auto& [s, e] : tree.at(node).missations.intervals.v)
        for (Site_index l = s; l != e; ++l) { if (cur_missing_sites.count(l)) needs_fixing = true; cur_missing_sites.insert(l); }
      if (needs_fixing) {
        Missation_map new_missations;
        for (Site_index l : cur_missing_sites)
          if (node == tree.root || !is_site_missing_at(tree, tree.at(node).parent, l)) new_missations.insert(l, cur_seq.get(l), ref_seq);
        tree.at(node).missations = std::move(new_missations);
      }
    }
    if (children_so_far == tree.at(node).num_children()) {
      for (auto it = tree.at(node).mutations.rbegin(); it != tree.at(node).mutations.rend(); ++it) { ORC_CHECK(it->to == cur_seq.get(it->site)); cur_seq.set(it->site, it->from); }
      for (const auto& [s, e] : tree.at(node).missations.intervals.v)
        for (Site_index l = s; l != e; ++l) { ORC_CHECK(cur_missing_sites.count(l)); cur_missing_sites.erase(l); }
    }
  });
  // Now reconstruct from_states
  Seq_overlay seq(ref_seq);
  traversal(tree, [&](Node_index node, int children_so_far) {
    if (children_so_far == 0) {
      tree.at(node).missations.from_states.clear();
      for (const auto& [l, s] : seq.deltas) if (tree.at(node).missations.contains(l)) tree.at(node).missations.from_states.insert({l, s});
      auto& muts = tree.at(node).mutations;
      muts.erase(std::remove_if(muts.begin(), muts.end(), [&](const Mutation& m) { return is_site_missing_at(tree, node, m.site); }), muts.end());
      for (const auto& m : muts) { ORC_CHECK(m.from == seq.get(m.site)); seq.set(m.site, m.to); }
    }
    if (children_so_far == tree.at(node).num_children()) {
      for (auto it = tree.at(node).mutations.rbegin(); it != tree.at(node).mutations.rend(); ++it) { ORC_CHECK(it->to == seq.get(it->site)); seq.set(it->site, it->from); }
    }
  });
}

// dates.cpp:63-82
inline void pseudo_date(Phylo_tree& tree, Rng& rng) {
  for (Node_index node : post_order(tree)) {
    if (tree.at(node).is_inner_node()) {
      Node_index lc = tree.at(node).children[0], rc = tree.at(node).children[1];
      double est_t_left = tree.at(lc).t - (double)tree.at(lc).mutations.size() * 13.0;
      double est_t_right = tree.at(rc).t - (double)tree.at(rc).mutations.size() * 13.0;
      tree.at(node).t = std::min(est_t_left, est_t_right) - rng.uniform_co(0.5, 1.5);
    }
  }
}

// phylo_tree.cpp:567-575
inline void randomize_mutation_times(Phylo_tree& tree, Rng& rng) {
  for (Node_index node = 0; node < tree.size(); ++node)
    if (node != tree.root) tree.at(node).mutations = randomize_branch_mutation_times(tree, node, rng);
}

// phylo_tree.cpp:138-202: returns "" when every tip reproduces its descriptor, else what is wrong
inline std::string check_phylo_tree_matches_tip_descs(const Phylo_tree& tree, const std::vector<State>& orig_ref_sequence, const std::vector<Tip_desc>& tip_descs) {
  std::string err;
  auto fail = [&](const std::string& m) { if (err.empty()) err = m; };
  Seq_overlay cur_seq(tree.ref_sequence);
  Interval_set cur_missing_sites, scratch_sites;
  traversal(tree, [&](Node_index node, int children_so_far) {
    if (!err.empty()) return;
    if (children_so_far == 0) {
      for (const auto& m : tree.at(node).mutations) { if (m.from != cur_seq.get(m.site)) fail("mutation chain broken at node " + std::to_string(node)); cur_seq.set(m.site, m.to); }
      if (interval_sets_intersect(tree.at(node).missations.intervals, cur_missing_sites)) fail("missations of node " + std::to_string(node) + " overlap those above it");
      merge_interval_sets(scratch_sites, tree.at(node).missations.intervals, cur_missing_sites);
      std::swap(cur_missing_sites, scratch_sites);
    }
    if (children_so_far == tree.at(node).num_children()) {
      if (tree.at(node).is_tip()) {
        if (node >= (int)tip_descs.size()) { fail("tip index beyond the descriptors"); return; }
        if (!(cur_missing_sites == tip_descs[node].missations.intervals)) fail("tip " + std::to_string(node) + ": missing site intervals differ");
        Seq_overlay expected(orig_ref_sequence);
        for (const auto& sd : tip_descs[node].seq_deltas) { if (sd.from != expected.get(sd.site)) fail("descriptor of tip " + std::to_string(node) + " inconsistent"); expected.set(sd.site, sd.to); }
        const int L = (int)orig_ref_sequence.size();
        for (Site_index site = 0; site < L; ++site)
          if (!cur_missing_sites.contains(site) && cur_seq.get(site) != expected.get(site)) { fail("tip " + std::to_string(node) + ", site " + std::to_string(site) + " differs"); break; }
      }
      if (!tree.at(node).missations.intervals.empty()) {
        Interval_set old_missing_sites = std::move(cur_missing_sites);
        cur_missing_sites.clear();
        subtract_interval_sets(cur_missing_sites, old_missing_sites, tree.at(node).missations.intervals);
      }
      for (auto it = tree.at(node).mutations.rbegin(); it != tree.at(node).mutations.rend(); ++it) { if (it->to != cur_seq.get(it->site)) fail("mutation chain broken leaving node " + std::to_string(node)); cur_seq.set(it->site, it->from); }
    }
  });
  return err;
}

// phylo_tree.cpp:796-1049.  Tips are nodes 0 .. n-1 in descriptor order, the inner node made for tip X is X + n - 1.
inline Phylo_tree build_usher_like_tree(const std::vector<State>& ref_sequence, const std::vector<Tip_desc>& tip_descs, Rng& rng) {
  const int L = (int)ref_sequence.size();
  for (size_t tip = 0; tip < tip_descs.size(); ++tip) {   // the reference's input checks (:804-843)
    const auto& td = tip_descs[tip];
    for (const auto& m : td.seq_deltas) {
      ORC_CHECK(m.site >= 0 && m.site < L);
      ORC_CHECK(m.from == ref_sequence[m.site]);
      ORC_CHECK(m.from != m.to);
      ORC_CHECK(!td.missations.contains(m.site));
    }
    for (const auto& [s, e] : td.missations.intervals.v) ORC_CHECK(s >= 0 && s < L && e >= 0 && e < L + 1);
  }
  const int num_tips = (int)tip_descs.size();
  if (num_tips == 0) { Phylo_tree t; t.ref_sequence = ref_sequence; return t; }
  Phylo_tree tree(2 * num_tips - 1);
  tree.ref_sequence = ref_sequence;
  for (int tip = 0; tip < num_tips; ++tip) {
    const auto& td = tip_descs[tip];
    ORC_CHECK(td.t_min <= td.t_max);
    tree.at(tip).t_min = td.t_min; tree.at(tip).t_max = td.t_max;
    tree.at(tip).t = uniform_cc(rng, td.t_min, td.t_max);
  }
  ORC_CHECK(num_tips >= 2);
  {
    const int P = num_tips, A = 0, B = 1;
    tree.root = P;
    tree.at(P).parent = k_no_node;
    tree.at(P).children[0] = A; tree.at(P).children[1] = B;
    tree.at(A).parent = P; tree.at(B).parent = P;
    const double t_A = tree.at(A).t, t_B = tree.at(B).t;
    const double t_P = std::min(t_A - (double)tip_descs[A].seq_deltas.size() * 13.0, t_B - (double)tip_descs[B].seq_deltas.size() * 13.0) - 1.0;
    tree.at(P).t_min = -FLT_MAX; tree.at(P).t_max = +FLT_MAX; tree.at(P).t = t_P;
    for (const auto& m : tip_descs[A].seq_deltas) tree.at(A).mutations.push_back(Mutation{m.from, m.site, m.to, rng.uniform_oc(t_P, t_A)});
    sort_mutations(tree.at(A).mutations);
    tree.at(A).missations = tip_descs[A].missations;
    for (const auto& m : tip_descs[B].seq_deltas) tree.at(B).mutations.push_back(Mutation{m.from, m.site, m.to, rng.uniform_oc(t_P, t_B)});
    sort_mutations(tree.at(B).mutations);
    tree.at(B).missations = tip_descs[B].missations;
  }
  // Sequentially graft every tip where it implies the fewest additional mutations
  for (Node_index X = 2; X != num_tips; ++X) {
    const auto& tip_desc = tip_descs[X];
    const double t_X = tree.at(X).t;
    Site_deltas deltas_root_to_X;
    for (const auto& delta : tip_desc.seq_deltas) push_back_site_deltas(delta, deltas_root_to_X);
    const Interval_set missing_at_X = tip_desc.missations.intervals;
    tree.at(X).parent = k_no_node;
    Spr_study_builder builder(tree, k_no_node, tree.at(X).t, missing_at_X);
    builder.seed_fill_from(tree.root, (int)tree.at_root().mutations.size(), deltas_root_to_X, true);
    int all_min_muts = std::numeric_limits<int>::max();
    for (const auto& region : builder.result) all_min_muts = std::min(all_min_muts, region.min_muts);
    int chosen_region_idx = -1;
    double tot_min_T = 0.0;
    for (int i = 0; i != (int)builder.result.size(); ++i) {
      const auto& region = builder.result[i];
      if (region.min_muts == all_min_muts) {
        if (region.branch == tree.root) { chosen_region_idx = i; break; }   // always above the root if that is a possibility
        tot_min_T += region.t_max - region.t_min;
      }
    }
    if (chosen_region_idx == -1) {
      double so_far_min_T = 0.0;
      const double insertion_cum_t = rng.uniform_co(0.0, tot_min_T);
      for (int i = 0; i != (int)builder.result.size(); ++i) {
        const auto& region = builder.result[i];
        if (region.min_muts == all_min_muts) {
          so_far_min_T += region.t_max - region.t_min;
          if (insertion_cum_t <= so_far_min_T) { chosen_region_idx = i; break; }
        }
      }
    }
    ORC_CHECK(chosen_region_idx != -1);
    const Candidate_region chosen_region = builder.result[chosen_region_idx];
    const Node_index S = chosen_region.branch, G = tree.at(S).parent, P = X + num_tips - 1;
    double t_P;
    Site_deltas deltas_P_to_X;
    if (S == tree.root) {
      deltas_P_to_X = deltas_root_to_X;
      const double t_P_guess = t_X - (double)deltas_P_to_X.size() * 13.0;
      const double t_S = tree.at(S).t;
      t_P = std::min(t_P_guess, t_S) - 1.0;
      tree.root = P;
      tree.at(P).mutations.clear();
      std::swap(tree.at(P).mutations, tree.at(S).mutations);
    } else {
      t_P = rng.uniform_oo(chosen_region.t_min, chosen_region.t_max);
      deltas_P_to_X = deltas_root_to_X;
      displace_site_deltas_start_downwards(tree, deltas_P_to_X, tree.node_loc(tree.root), Phylo_tree_loc{S, t_P});
      const Node_index U = tree.at(G).sibling_of(S);
      tree.at(G).children[0] = P; tree.at(G).children[1] = U;
      tree.at(P).parent = G;
      ORC_CHECK(tree.at(U).parent == G);
      auto& mS = tree.at(S).mutations;
      size_t split = 0;
      while (split < mS.size() && !(mS[split].t > t_P)) ++split;
      tree.at(P).mutations.assign(mS.begin(), mS.begin() + split);
      mS.erase(mS.begin(), mS.begin() + split);
    }
    tree.at(P).t_min = -FLT_MAX; tree.at(P).t_max = +FLT_MAX; tree.at(P).t = t_P;
    tree.at(P).children[0] = X; tree.at(P).children[1] = S;
    tree.at(X).parent = P; tree.at(S).parent = P;
    for (const auto& [l, delta] : deltas_P_to_X) tree.at(X).mutations.push_back(Mutation{delta.from, l, delta.to, rng.uniform_oc(t_P, t_X)});
    sort_mutations(tree.at(X).mutations);
    tree.at(X).missations = tip_desc.missations;
  }
  fix_up_missations(tree);
  pseudo_date(tree, rng);
  randomize_mutation_times(tree, rng);
  return tree;
}

// Tip descriptors of an existing EMAT: what a MAPLE file of its tips would hold -- each tip's deltas against the reference
// sequence at the sites it has, its missing intervals, its date range.  (Test scenarios are generated as trees; this turns
// one into the builder's input.)
inline std::vector<Tip_desc> tip_descs_of(const Phylo_tree& t) {
  std::vector<Tip_desc> out;
  for (int n = 0; n < t.size(); ++n) if (t.at(n).is_tip()) {
    Tip_desc d; d.t_min = t.at(n).t_min; d.t_max = t.at(n).t_max;
    d.missations.intervals = reconstruct_missing_sites_at(t, n);
    for (auto& [l, dl] : deltas_ref_to_loc(t, t.node_loc(n))) if (!d.missations.intervals.contains(l)) d.seq_deltas.push_back(Seq_delta(l, t.ref_sequence[l], dl.to));
    out.push_back(std::move(d));
  }
  return out;
}

}  // namespace orc
#endif  // ORC_BUILD_HPP_
