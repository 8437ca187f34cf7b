// orc_utree.hpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// The reference's DEFAULT initial-tree builder (cmdline.cpp:437, Init_method::mp_plus_timing -> build_initial_phylo_tree), restated
// from core/utree.h and core/utree.cpp, function by function (the UShER-like builder of --v0-init-method
// old_usher_like is in orc_build.hpp):
//   utree.h:36-230, 395-487      Utree: arcs in mate pairs, the focus node, move_focus_to, split_edge
//   utree.h:327-373              annotated_arc_euler_tour, arc_euler_tour, utree_tips
//   utree.cpp:9-127              reset_focus, detach_tip, merge_through, remove_edge
//   utree.cpp:140-170            Relative_fitch_sets
//   utree.cpp:190-739            Utree_builder: add_tip, find_best_attachment_arc (branch and bound), split_best_arc_inserting_M,
//                                attach_tip / attach_subtree, strip_missing_deltas, init_fitch_X_for_tip / _for_subtree, ...
//   utree.cpp:744-755            build_guide_tree
//   utree.cpp:761-896            for_each_tip_in_nearest_first_order
//   utree.cpp:898-914            build_refined_tree
//   utree.cpp:920-1081           spr_refine (tip and subtree SPR)
//   utree.cpp:1085-1248          midpoint_root_utree
//   utree.cpp:1255-1464          ols_regression_root_utree
//   utree.cpp:1470-1731          gls_regression_root_utree
//   utree.cpp:1750-1890          utree_to_phylo_tree
//   utree.cpp:1892-1925          build_initial_phylo_tree
//   utree.cpp:2010-2258          assert_utree_integrity, assert_utree_matches_tip_descs
// Pinned to the reference's own tests of this module, tests/utree_tests.cpp, re-evaluated with its fixtures and expectations in
// orc_tests.cpp (every test that does not depend on the draws of a particular std::mt19937 stream: the structural ones literally,
// the randomised ones through the invariants they assert over many seeds).
//
// Random numbers: the reference draws from std::mt19937 through absl::Uniform<int> and std::bernoulli_distribution; here they come
// from orc::Rng, one 64-bit draw each, in the reference's order of draws.  Hash-map iterations of the reference (the site deltas
// of an arc) run in ascending site order here; where the reference's result depends on that order (which of an edge's deltas end up
// on which side of a new root, utree.cpp:1184-1188) this restatement is one of the orders the reference may take.
// Tip names are not on the path and are not carried.
#ifndef ORC_UTREE_HPP_
#define ORC_UTREE_HPP_

#include <array>
#include <cmath>
#include <functional>
#include <queue>

#include "orc_build.hpp"

namespace orc {

using Arc_index = int;
constexpr Arc_index k_no_arc = -1;

struct Uarc {                    // utree.h:42-48
  int target = -1;               // the node the arc points to; on the free list: the next free pair
  Site_deltas deltas;            // from origin to target
};
struct Unode {                   // utree.h:52-55
  std::array<Arc_index, 3> arcs = {k_no_arc, k_no_arc, k_no_arc};
  Arc_index arc_to_focus = k_no_arc;
};
enum class Arc_direction { entering, leaving };
struct Annotated_arc { Arc_index arc; Arc_direction direction; };

struct Utree {                   // utree.h:57-230
  std::vector<State> ref_sequence;
  Interval_set globally_missing_sites;
  std::vector<Uarc> arcs;
  std::vector<Unode> nodes;
  Arc_index arc_free_list_head = k_no_arc;
  int num_tips = 0;
  int num_inner_nodes_so_far = 0;
  Node_index focus = -1;
  Site_deltas deltas_ref_to_focus;

  static Utree make_empty(int num_tips) {                            // :73-85
    Utree tree;
    tree.num_tips = num_tips;
    const int num_nodes = std::max(1, 2 * num_tips - 1);
    const int num_arc_pairs = std::max(1, 2 * num_tips - 3 + 2);
    tree.nodes.resize(num_nodes);
    tree.arcs.resize(2 * num_arc_pairs);
    tree.arc_free_list_head = 0;
    for (int i = 0; i < 2 * num_arc_pairs; i += 2) tree.arcs[i].target = (i + 2 < 2 * num_arc_pairs) ? (i + 2) : k_no_arc;
    return tree;
  }
  Arc_index mate(Arc_index arc) const { return arc ^ 1; }
  Node_index origin(Arc_index arc) const { return arcs[mate(arc)].target; }
  Node_index target(Arc_index arc) const { return arcs[arc].target; }
  Arc_index find_arc(Node_index node, Node_index tgt) const {       // :97-102
    for (auto a : nodes[node].arcs) if (a != k_no_arc && arcs[a].target == tgt) return a;
    return k_no_arc;
  }
  int degree(Node_index node) const { int d = 0; for (auto a : nodes[node].arcs) if (a != k_no_arc) ++d; return d; }
  bool is_tip(Node_index node) const { return degree(node) == 1; }
  Node_index pick_random_tip(Rng& rng) const { return (Node_index)rng.uniform_int(num_tips); }
  Node_index pick_random_node(Rng& rng) const {                      // :120-126
    const int num_nodes = num_tips + num_inner_nodes_so_far;
    Node_index node;
    do { node = (Node_index)rng.uniform_int(num_nodes); } while (degree(node) == 0);
    return node;
  }
  int count_arc_deltas(Arc_index arc) const { return (int)arcs[arc].deltas.size(); }
  int count_deltas() const { int total = 0; for (Arc_index i = 0; i < (int)arcs.size(); i += 2) total += count_arc_deltas(i); return total; }
  Arc_index alloc_arc_pair() {                                       // :144-151
    ORC_CHECK(arc_free_list_head != k_no_arc);
    const Arc_index base = arc_free_list_head;
    arc_free_list_head = arcs[base].target;
    arcs[base].target = -1; arcs[base + 1].target = -1;
    return base;
  }
  void free_arc_pair(Arc_index arc) {                                // :154-160
    const Arc_index base = arc & ~1;
    arcs[base].deltas.clear(); arcs[base + 1].deltas.clear();
    arcs[base].target = arc_free_list_head;
    arc_free_list_head = base;
  }
  Arc_index add_arc(Node_index A, Node_index B) {                    // :164-181
    const Arc_index base = alloc_arc_pair();
    const Arc_index arc_AB = base, arc_BA = base + 1;
    arcs[arc_AB].target = B; arcs[arc_BA].target = A;
    auto wire = [&](Node_index node, Arc_index arc) {
      for (auto& a : nodes[node].arcs) if (a == k_no_arc) { a = arc; return; }
      ORC_CHECK(false && "no free arc slot");
    };
    wire(A, arc_AB); wire(B, arc_BA);
    return arc_AB;
  }
  std::vector<Annotated_arc> annotated_arc_euler_tour(Node_index source) const {   // utree.h:337-360 (the generator, run to its end)
    std::vector<Annotated_arc> out, stack;
    for (auto a : nodes[source].arcs) if (a != k_no_arc) { stack.push_back({mate(a), Arc_direction::leaving}); stack.push_back({a, Arc_direction::entering}); }
    while (!stack.empty()) {
      const Annotated_arc cur = stack.back(); stack.pop_back();
      out.push_back(cur);
      if (cur.direction == Arc_direction::entering) {
        const Node_index B = target(cur.arc);
        for (auto a : nodes[B].arcs) if (a != k_no_arc && a != mate(cur.arc)) { stack.push_back({mate(a), Arc_direction::leaving}); stack.push_back({a, Arc_direction::entering}); }
      }
    }
    return out;
  }
  void reset_focus(Node_index F) {                                   // utree.cpp:9-17
    focus = F;
    nodes[F].arc_to_focus = k_no_arc;
    for (auto [arc, direction] : annotated_arc_euler_tour(F)) if (direction == Arc_direction::entering) nodes[target(arc)].arc_to_focus = mate(arc);
  }
  Node_index detach_tip(Node_index X) {                              // utree.cpp:19-45
    ORC_CHECK(is_tip(X)); ORC_CHECK(degree(X) == 1); ORC_CHECK(focus != X); ORC_CHECK(num_tips >= 3);
    Arc_index arc_XM = k_no_arc;
    for (auto a : nodes[X].arcs) if (a != k_no_arc) { arc_XM = a; break; }
    ORC_CHECK(arc_XM != k_no_arc);
    const Node_index M = target(arc_XM);
    const Arc_index arc_MX = mate(arc_XM);
    for (auto& a : nodes[M].arcs) if (a == arc_MX) { a = k_no_arc; break; }
    for (auto& a : nodes[X].arcs) a = k_no_arc;
    nodes[X].arc_to_focus = k_no_arc;
    free_arc_pair(arc_XM);
    return M;
  }
  Arc_index merge_through(Node_index M) {                            // utree.cpp:47-113
    ORC_CHECK(degree(M) == 2); ORC_CHECK(focus != M); ORC_CHECK(nodes[M].arc_to_focus != k_no_arc);
    Arc_index arc_MA = k_no_arc, arc_MB = k_no_arc;
    for (auto a : nodes[M].arcs) if (a != k_no_arc) { if (arc_MA == k_no_arc) arc_MA = a; else arc_MB = a; }
    ORC_CHECK(arc_MA != k_no_arc && arc_MB != k_no_arc);
    const Node_index A = target(arc_MA), B = target(arc_MB);
    Site_deltas A_to_B = arcs[mate(arc_MA)].deltas;
    for (const auto& [site, delta] : arcs[arc_MB].deltas) push_back_site_deltas({site, delta.from, delta.to}, A_to_B);
    Site_deltas B_to_A;
    for (const auto& [site, delta] : A_to_B) B_to_A[site] = {delta.to, delta.from};
    const Arc_index arc_AB = alloc_arc_pair(), arc_BA = mate(arc_AB);
    arcs[arc_AB].target = B; arcs[arc_BA].target = A;
    arcs[arc_AB].deltas = std::move(A_to_B); arcs[arc_BA].deltas = std::move(B_to_A);
    const Arc_index arc_AM = find_arc(A, M), arc_BM = find_arc(B, M);
    ORC_CHECK(arc_AM != k_no_arc && arc_BM != k_no_arc);
    for (auto& a : nodes[A].arcs) if (a == arc_AM) { a = arc_AB; break; }
    for (auto& a : nodes[B].arcs) if (a == arc_BM) { a = arc_BA; break; }
    if (nodes[A].arc_to_focus == arc_AM) nodes[A].arc_to_focus = arc_AB;
    if (nodes[B].arc_to_focus == arc_BM) nodes[B].arc_to_focus = arc_BA;
    for (auto& a : nodes[M].arcs) a = k_no_arc;
    nodes[M].arc_to_focus = k_no_arc;
    free_arc_pair(arc_MA); free_arc_pair(arc_MB);
    return arc_AB;
  }
  void remove_edge(Node_index u, Node_index v) {                     // utree.cpp:115-127
    const Arc_index arc_uv = find_arc(u, v);
    ORC_CHECK(arc_uv != k_no_arc);
    const Arc_index arc_vu = mate(arc_uv);
    for (auto& a : nodes[u].arcs) if (a == arc_uv) { a = k_no_arc; break; }
    if (nodes[u].arc_to_focus == arc_uv) nodes[u].arc_to_focus = k_no_arc;
    for (auto& a : nodes[v].arcs) if (a == arc_vu) { a = k_no_arc; break; }
    if (nodes[v].arc_to_focus == arc_vu) nodes[v].arc_to_focus = k_no_arc;
    free_arc_pair(arc_uv);
  }
  // utree.h:431-487
  template <class Side> void split_edge(Arc_index arc_AB, Node_index M, Side site_delta_side) {
    const Arc_index arc_BA = mate(arc_AB);
    const Node_index A = origin(arc_AB), B = target(arc_AB);
    const Arc_index arc_AM = alloc_arc_pair(), arc_MA = mate(arc_AM);
    const Arc_index arc_MB = alloc_arc_pair(), arc_BM = mate(arc_MB);
    arcs[arc_AM].target = M; arcs[arc_MA].target = A; arcs[arc_MB].target = B; arcs[arc_BM].target = M;
    const Site_deltas old = arcs[arc_AB].deltas;     // (the reference iterates the live map: nothing below touches it)
    for (const auto& [site, delta] : old) {
      const Node_index side = site_delta_side(Seq_delta{site, delta.from, delta.to}, A, B);
      ORC_CHECK(side == A || side == B);
      if (side == A) { arcs[arc_AM].deltas[site] = {delta.from, delta.to}; arcs[arc_MA].deltas[site] = {delta.to, delta.from}; }
      else { arcs[arc_MB].deltas[site] = {delta.from, delta.to}; arcs[arc_BM].deltas[site] = {delta.to, delta.from}; }
    }
    nodes[M].arcs = {arc_MA, arc_MB, k_no_arc};
    for (auto& a : nodes[A].arcs) if (a == arc_AB) { a = arc_AM; break; }
    for (auto& a : nodes[B].arcs) if (a == arc_BA) { a = arc_BM; break; }
    if (nodes[A].arc_to_focus == arc_AB) { nodes[A].arc_to_focus = arc_AM; nodes[M].arc_to_focus = arc_MB; }
    if (nodes[B].arc_to_focus == arc_BA) { nodes[B].arc_to_focus = arc_BM; nodes[M].arc_to_focus = arc_MA; }
    free_arc_pair(arc_AB);
  }
  // utree.h:395-428
  template <class Pre, class Post> void move_focus_to(Node_index tgt, Pre pre_arc_hop, Post post_arc_hop) {
    if (tgt == focus) return;
    Node_index cur = tgt;
    Arc_index prev_saved = nodes[cur].arc_to_focus;
    nodes[cur].arc_to_focus = k_no_arc;
    while (prev_saved != k_no_arc) {
      const Node_index next_node = arcs[prev_saved].target;
      const Arc_index next_saved = nodes[next_node].arc_to_focus;
      nodes[next_node].arc_to_focus = mate(prev_saved);
      prev_saved = next_saved;
    }
    cur = focus;
    while (cur != tgt) {
      const Arc_index arc_R = nodes[cur].arc_to_focus;
      pre_arc_hop(arc_R);
      for (const auto& [site, delta] : arcs[arc_R].deltas) push_back_site_deltas({site, delta.from, delta.to}, deltas_ref_to_focus);
      post_arc_hop(arc_R);
      cur = arcs[arc_R].target;
    }
    focus = tgt;
  }
  template <class Pre> void move_focus_to(Node_index tgt, Pre pre) { move_focus_to(tgt, pre, [](Arc_index) {}); }
  void move_focus_to(Node_index tgt) { move_focus_to(tgt, [](Arc_index) {}, [](Arc_index) {}); }
};

// ---- integrity checks (utree.cpp:2010-2258): throw through ORC_CHECK; the check_ forms return the message ("" when fine) --------
inline void assert_utree_integrity(const Utree& tree) {
  const int num_nodes = tree.num_tips + tree.num_inner_nodes_so_far;
  if (num_nodes == 0) { ORC_CHECK(tree.focus == k_no_node); return; }
  if (num_nodes == 1) ORC_CHECK(tree.degree(0) == 0);
  else {
    for (Node_index i = 0; i < tree.num_tips; ++i) ORC_CHECK(tree.degree(i) == 1);
    for (Node_index i = tree.num_tips; i < num_nodes; ++i) ORC_CHECK(tree.degree(i) == 2 || tree.degree(i) == 3);
  }
  std::set<Arc_index> free_pairs;
  const int num_arcs = (int)tree.arcs.size();
  for (Arc_index cur = tree.arc_free_list_head; cur != k_no_arc; cur = tree.arcs[cur].target) {
    ORC_CHECK(cur >= 0 && cur < num_arcs);
    ORC_CHECK(cur == (cur & ~1));
    ORC_CHECK(free_pairs.insert(cur).second);
  }
  for (Arc_index base = 0; base < num_arcs; base += 2) {
    if (free_pairs.count(base)) continue;
    const Arc_index a = base, a_mate = tree.mate(a);
    const Node_index target_a = tree.target(a), origin_a = tree.origin(a);
    ORC_CHECK(target_a >= 0 && target_a < num_nodes);
    ORC_CHECK(origin_a >= 0 && origin_a < num_nodes);
    ORC_CHECK(origin_a != target_a);
    bool found = false; for (auto slot : tree.nodes[target_a].arcs) if (slot == a_mate) found = true;
    ORC_CHECK(found);
    found = false; for (auto slot : tree.nodes[origin_a].arcs) if (slot == a) found = true;
    ORC_CHECK(found);
    ORC_CHECK(tree.arcs[a].deltas.size() == tree.arcs[a_mate].deltas.size());
    for (const auto& [site, delta] : tree.arcs[a].deltas) {
      ORC_CHECK(!tree.globally_missing_sites.contains(site));
      auto it = tree.arcs[a_mate].deltas.find(site);
      ORC_CHECK(it != tree.arcs[a_mate].deltas.end());
      ORC_CHECK(it->second.from == delta.to);
      ORC_CHECK(it->second.to == delta.from);
    }
  }
  ORC_CHECK(tree.focus >= 0 && tree.focus < num_nodes);
  ORC_CHECK(tree.nodes[tree.focus].arc_to_focus == k_no_arc);
  std::vector<int> enter_count(num_nodes, 0);
  enter_count[tree.focus] = 1;
  for (auto [arc, direction] : tree.annotated_arc_euler_tour(tree.focus)) {
    if (direction == Arc_direction::entering) {
      const Node_index node = tree.target(arc);
      ORC_CHECK(node >= 0 && node < num_nodes);
      ++enter_count[node];
      ORC_CHECK(enter_count[node] <= 1);
    } else {
      const Node_index node = tree.origin(arc);
      ORC_CHECK(node >= 0 && node < num_nodes);
      ORC_CHECK(tree.nodes[node].arc_to_focus == arc);
    }
  }
  for (Node_index i = 0; i < num_nodes; ++i) ORC_CHECK(enter_count[i] == 1);
  const int L = (int)tree.ref_sequence.size();
  for (const auto& [site, delta] : tree.deltas_ref_to_focus) {
    ORC_CHECK(site >= 0 && site < L);
    ORC_CHECK(delta.from == tree.ref_sequence[site]);
    ORC_CHECK(delta.from != delta.to);
    ORC_CHECK(!tree.globally_missing_sites.contains(site));
  }
  for (Arc_index base = 0; base < num_arcs; base += 2) {
    if (free_pairs.count(base)) continue;
    for (const auto& [site, delta] : tree.arcs[base].deltas) { ORC_CHECK(site >= 0 && site < L); ORC_CHECK(delta.from != delta.to); }
  }
  Site_deltas ref_to_cur = tree.deltas_ref_to_focus;
  for (auto [arc, direction] : tree.annotated_arc_euler_tour(tree.focus)) {
    (void)direction;
    for (const auto& [site, delta] : tree.arcs[arc].deltas) {
      auto it = ref_to_cur.find(site);
      if (it != ref_to_cur.end()) ORC_CHECK(it->second.to == delta.from);
      else ORC_CHECK(tree.ref_sequence[site] == delta.from);
      push_back_site_deltas({site, delta.from, delta.to}, ref_to_cur);
    }
  }
  ORC_CHECK(ref_to_cur == tree.deltas_ref_to_focus);
}
inline void assert_utree_matches_tip_descs(const Utree& tree, const std::vector<Tip_desc>& tip_descs) {
  const int N = (int)tip_descs.size();
  ORC_CHECK(tree.num_tips == N);
  const int L = (int)tree.ref_sequence.size();
  for (Node_index i = 0; i < N; ++i) {
    const auto& td = tip_descs[i];
    ORC_CHECK(td.missations.intervals.is_valid(L));
    std::set<Site_index> seen;
    for (const auto& sd : td.seq_deltas) {
      ORC_CHECK(sd.site >= 0 && sd.site < L);
      ORC_CHECK(sd.from == tree.ref_sequence[sd.site]);
      ORC_CHECK(sd.from != sd.to);
      ORC_CHECK(!td.missations.intervals.contains(sd.site));
      ORC_CHECK(seen.insert(sd.site).second);
    }
  }
  if (N == 0) { ORC_CHECK(tree.globally_missing_sites.empty()); return; }
  Interval_set expected = tip_descs[0].missations.intervals;
  for (Node_index i = 1; i < N; ++i) expected = intersected(expected, tip_descs[i].missations.intervals);
  ORC_CHECK(tree.globally_missing_sites == expected);
  Site_deltas ref_to_cur = tree.deltas_ref_to_focus;
  auto check_tip = [&](Node_index tip) {
    const auto& td = tip_descs[tip];
    const auto& miss = td.missations.intervals;
    std::set<Site_index> sites;
    for (const auto& sd : td.seq_deltas) {
      auto it = ref_to_cur.find(sd.site);
      ORC_CHECK(it != ref_to_cur.end());
      ORC_CHECK(it->second.from == sd.from);
      ORC_CHECK(it->second.to == sd.to);
      sites.insert(sd.site);
    }
    for (const auto& [site, delta] : ref_to_cur) { (void)delta; if (!sites.count(site)) ORC_CHECK(miss.contains(site)); }
  };
  if (tree.focus < N) check_tip(tree.focus);
  for (auto [arc, direction] : tree.annotated_arc_euler_tour(tree.focus)) {
    for (const auto& [site, delta] : tree.arcs[arc].deltas) push_back_site_deltas({site, delta.from, delta.to}, ref_to_cur);
    if (direction == Arc_direction::entering) {
      const Node_index node = tree.target(arc);
      if (node < N) {
        const auto& miss = tip_descs[node].missations.intervals;
        for (const auto& [site, delta] : tree.arcs[arc].deltas) { (void)delta; ORC_CHECK(!miss.contains(site)); }
        check_tip(node);
      }
    }
  }
}
inline std::string check_utree_integrity(const Utree& tree) { try { assert_utree_integrity(tree); } catch (const std::exception& ex) { return ex.what(); } return ""; }
inline std::string check_utree_matches_tip_descs(const Utree& tree, const std::vector<Tip_desc>& tip_descs) {
  try { assert_utree_matches_tip_descs(tree, tip_descs); } catch (const std::exception& ex) { return ex.what(); } return "";
}

// ---- Relative_fitch_sets (utree.cpp:140-170) ------------------------------------------------------------------------------------
inline uint8_t to_seq_letter(State s) { return (uint8_t)(1u << s); }
struct Relative_fitch_sets {
  Site_deltas resolved_deltas_;                      // size-1 sites: {from: ref, to: f}
  std::map<Site_index, uint8_t> ambiguous_masks_;    // size-2/3 sites: bitmask
  Interval_set uninformative_sites_;                 // size-4 sites (incl. globally missing)
  void clear() { resolved_deltas_.clear(); ambiguous_masks_.clear(); uninformative_sites_.clear(); }
  bool contains(Site_index s, State state, State ref_state) const {
    if (uninformative_sites_.contains(s)) return true;
    if (auto it = ambiguous_masks_.find(s); it != ambiguous_masks_.end()) return (it->second & to_seq_letter(state)) != 0;
    if (auto it = resolved_deltas_.find(s); it != resolved_deltas_.end()) return state == it->second.to;
    return state == ref_state;
  }
  void on_ref_change(Site_index s, State old_state, State new_state) {
    if (uninformative_sites_.contains(s)) return;
    if (ambiguous_masks_.count(s)) return;
    pop_front_site_deltas({s, old_state, new_state}, resolved_deltas_);
  }
};

// ---- Utree_builder (utree.cpp:190-739) ------------------------------------------------------------------------------------------
struct Utree_builder {
  const std::vector<Tip_desc>& tip_descs_;
  Rng& rng_;
  Utree tree_;
  int tips_added_ = 0;
  int L_ = 0;
  double sqrt_6L_ = 0.0;
  Relative_fitch_sets fitch_X_;
  int n_mismatches_ = 0;
  Site_deltas M_to_X_deltas_;
  std::map<Site_index, State> m_overrides_;
  using Pq_entry = std::pair<int, Arc_index>;
  std::vector<Pq_entry> pq_;
  std::vector<Arc_index> best_arcs_;
  int component_abort_threshold_ = 0;
  std::vector<Node_index> component_dfs_stack_, component_nodes_;
  static constexpr int k_min_pruning_threshold = 2;

  Utree_builder(std::vector<State> ref_sequence, const std::vector<Tip_desc>& tip_descs, Rng& rng) : tip_descs_(tip_descs), rng_(rng) {   // :192-204
    const int N = (int)tip_descs.size();
    if (N > 0) tree_ = Utree::make_empty(N);
    tree_.ref_sequence = std::move(ref_sequence);
    L_ = (int)tree_.ref_sequence.size();
    sqrt_6L_ = std::sqrt(6.0 * L_);
  }
  Utree_builder(Utree tree, const std::vector<Tip_desc>& tip_descs, Rng& rng) : tip_descs_(tip_descs), rng_(rng), tree_(std::move(tree)) {   // :206-213
    tips_added_ = tree_.num_tips;
    L_ = (int)tree_.ref_sequence.size();
    sqrt_6L_ = std::sqrt(6.0 * L_);
  }
  Node_index alloc_inner_node() { return tree_.num_tips + tree_.num_inner_nodes_so_far++; }
  void add_tip(int X) {                                              // :221-246
    ORC_CHECK(X >= 0 && X < (int)tip_descs_.size());
    ORC_CHECK(tips_added_ < (int)tip_descs_.size());
    if (tips_added_ == 1) ORC_CHECK(X != tree_.focus);
    else if (tips_added_ >= 2) ORC_CHECK(tree_.degree(X) == 0);
    if (tips_added_ == 0) add_first_tip(X);
    else {
      update_globally_missing_sites(X);
      init_fitch_X_for_tip(X);
      auto [best_arc, best_cost] = find_best_attachment_arc();
      (void)best_cost;
      if (best_arc == k_no_arc) attach_tip_directly_to_isolated_focus(X);
      else { move_focus_updating_fitch_X(tree_.origin(best_arc)); attach_tip_to_focal_arc(X, best_arc, alloc_inner_node()); }
    }
    ++tips_added_;
  }
  Utree finish() { assert_utree_integrity(tree_); assert_utree_matches_tip_descs(tree_, tip_descs_); return std::move(tree_); }   // :249-253
  void move_focus_to(Node_index target) { tree_.move_focus_to(target); }
  int pruning_threshold(int cost) const {                            // :267-271
    const double sigma = cost / sqrt_6L_;
    const int threshold = (int)std::ceil(10.0 * sigma * (sigma + 5));
    return std::clamp(threshold, k_min_pruning_threshold, L_);
  }
  void add_first_tip(int X) {                                        // :277-287
    ORC_CHECK(tips_added_ == 0);
    const auto& tip_X = tip_descs_[X];
    tree_.focus = X;
    tree_.globally_missing_sites = tip_X.missations.intervals;
    for (const auto& sd : tip_X.seq_deltas) { ORC_CHECK(!tree_.globally_missing_sites.contains(sd.site)); tree_.deltas_ref_to_focus[sd.site] = {sd.from, sd.to}; }
  }
  void update_globally_missing_sites(int X) {                        // :294-312
    ORC_CHECK(tips_added_ >= 1);
    const auto& tip_X = tip_descs_[X];
    const auto& miss_X = tip_X.missations.intervals;
    if (!interval_set_is_subset_of(tree_.globally_missing_sites, miss_X)) {
      const Interval_set old = std::move(tree_.globally_missing_sites);
      tree_.globally_missing_sites = intersected(old, miss_X);
      for (const auto& sd : tip_X.seq_deltas)
        if (old.contains(sd.site) && !tree_.globally_missing_sites.contains(sd.site)) push_back_site_deltas(sd, tree_.deltas_ref_to_focus);
    }
  }
  State focus_state(Site_index s) const { auto it = tree_.deltas_ref_to_focus.find(s); return it != tree_.deltas_ref_to_focus.end() ? it->second.to : tree_.ref_sequence[s]; }
  void init_fitch_X_for_tip(int X) {                                 // :326-344
    ORC_CHECK(tips_added_ >= 1);
    const auto& tip_X = tip_descs_[X];
    const auto& miss_X = tip_X.missations.intervals;
    fitch_X_.clear(); n_mismatches_ = 0;
    fitch_X_.uninformative_sites_ = miss_X;
    for (const auto& sd : tip_X.seq_deltas) { ORC_CHECK(!miss_X.contains(sd.site)); fitch_X_.resolved_deltas_[sd.site] = {sd.from, sd.to}; }
    for (const auto& [site, delta] : tree_.deltas_ref_to_focus) if (!miss_X.contains(site)) push_front_site_deltas({site, delta.to, delta.from}, fitch_X_.resolved_deltas_);
    n_mismatches_ = (int)fitch_X_.resolved_deltas_.size();
  }
  void init_fitch_X_for_subtree(Node_index X) {                      // :351-415
    fitch_X_.clear(); n_mismatches_ = 0;
    const Node_index M = tree_.focus;
    const Arc_index arc_MX = tree_.find_arc(M, X);
    ORC_CHECK(arc_MX != k_no_arc);
    Arc_index arc_XD = k_no_arc, arc_XE = k_no_arc;
    for (auto a : tree_.nodes[X].arcs) { if (a == k_no_arc || tree_.target(a) == M) continue; if (arc_XD == k_no_arc) arc_XD = a; else arc_XE = a; }
    ORC_CHECK(arc_XD != k_no_arc && arc_XE != k_no_arc);
    const Node_index D = tree_.target(arc_XD), E = tree_.target(arc_XE);
    const Interval_set* miss_D = tree_.is_tip(D) ? &tip_descs_[D].missations.intervals : nullptr;
    const Interval_set* miss_E = tree_.is_tip(E) ? &tip_descs_[E].missations.intervals : nullptr;
    if (miss_D != nullptr && miss_E != nullptr) intersect_interval_sets(fitch_X_.uninformative_sites_, *miss_D, *miss_E);
    struct Mde { State m, d, e; };
    std::map<Site_index, Mde> mde;
    for (const auto& [s, delta] : tree_.arcs[arc_MX].deltas) mde[s] = {delta.from, delta.to, delta.to};
    for (const auto& [s, delta] : tree_.arcs[arc_XD].deltas) { if (auto it = mde.find(s); it != mde.end()) it->second.d = delta.to; else mde[s] = {delta.from, delta.to, delta.from}; }
    for (const auto& [s, delta] : tree_.arcs[arc_XE].deltas) { if (auto it = mde.find(s); it != mde.end()) it->second.e = delta.to; else mde[s] = {delta.from, delta.from, delta.to}; }
    for (const auto& [s, v] : mde) {
      if (fitch_X_.uninformative_sites_.contains(s)) continue;
      const bool d_missing = miss_D != nullptr && miss_D->contains(s);
      const bool e_missing = miss_E != nullptr && miss_E->contains(s);
      if (d_missing || e_missing || v.d == v.e) {
        const State f = d_missing ? v.e : v.d;
        if (v.m != f) { fitch_X_.resolved_deltas_[s] = {v.m, f}; ++n_mismatches_; }
      } else {
        fitch_X_.ambiguous_masks_[s] = to_seq_letter(v.d) | to_seq_letter(v.e);
        if (v.m != v.d && v.m != v.e) ++n_mismatches_;
      }
    }
  }
  std::pair<Arc_index, int> find_best_attachment_arc() {             // :421-482
    int best_cost = n_mismatches_;
    best_arcs_.clear();
    auto record = [&](int cost, Arc_index a) { if (cost < best_cost) { best_cost = cost; best_arcs_.clear(); } if (cost == best_cost) best_arcs_.push_back(a); };
    pq_.clear();
    const auto cmp = std::greater<>{};
    for (auto a : tree_.nodes[tree_.focus].arcs) if (a != k_no_arc) { const int cost = eval_focal_arc(a); record(cost, a); pq_.push_back({cost, a}); }
    std::make_heap(pq_.begin(), pq_.end(), cmp);
    while (!pq_.empty()) {
      std::pop_heap(pq_.begin(), pq_.end(), cmp);
      auto [priority, arc_R] = pq_.back();
      pq_.pop_back();
      if (priority > best_cost + pruning_threshold(best_cost)) break;
      move_focus_updating_fitch_X(tree_.target(arc_R));
      for (auto a : tree_.nodes[tree_.focus].arcs)
        if (a != k_no_arc && a != tree_.mate(arc_R)) { const int cost = eval_focal_arc(a); record(cost, a); pq_.push_back({cost, a}); std::push_heap(pq_.begin(), pq_.end(), cmp); }
    }
    if (best_arcs_.empty()) return {k_no_arc, best_cost};
    const int idx = rng_.uniform_int((int)best_arcs_.size());
    return {best_arcs_[idx], best_cost};
  }
  void attach_tip_directly_to_isolated_focus(int X) {                // :489-498
    ORC_CHECK(tips_added_ == 1);
    ORC_CHECK(tree_.degree(tree_.focus) == 0);
    const Arc_index arc_focus_X = tree_.add_arc(tree_.focus, X);
    for (const auto& [site, delta] : fitch_X_.resolved_deltas_) { tree_.arcs[arc_focus_X].deltas[site] = delta; tree_.arcs[tree_.mate(arc_focus_X)].deltas[site] = {delta.to, delta.from}; }
    tree_.nodes[X].arc_to_focus = tree_.mate(arc_focus_X);
  }
  State m_state(Site_index s) const { auto it = m_overrides_.find(s); return it != m_overrides_.end() ? it->second : focus_state(s); }
  void split_best_arc_inserting_M(Arc_index best_arc, Node_index M) {   // :512-547
    ORC_CHECK(best_arc != k_no_arc);
    ORC_CHECK(tree_.origin(best_arc) == tree_.focus);
    M_to_X_deltas_ = fitch_X_.resolved_deltas_;
    m_overrides_.clear();
    tree_.split_edge(best_arc, M, [&](Seq_delta sd, Node_index A, Node_index B) -> Node_index {
      const Site_index s = sd.site;
      auto place_on_A = [&]() {
        if (!fitch_X_.ambiguous_masks_.count(s)) pop_front_site_deltas({s, sd.from, sd.to}, M_to_X_deltas_);
        m_overrides_[s] = sd.to;
      };
      if (fitch_X_.uninformative_sites_.contains(s)) return rng_.coin() ? A : B;
      else if (fitch_X_.contains(s, sd.to, sd.from)) { place_on_A(); return A; }
      else if (fitch_X_.contains(s, sd.from, sd.from)) return B;
      else { const Node_index side = rng_.coin() ? A : B; if (side == A) place_on_A(); return side; }
    });
    ORC_CHECK(tree_.target(tree_.nodes[M].arc_to_focus) == tree_.focus);
  }
  void wire_M_X(Node_index M, Node_index X) {                        // :550-557
    const Arc_index arc_MX = tree_.add_arc(M, X);
    for (const auto& [site, delta] : M_to_X_deltas_) { tree_.arcs[arc_MX].deltas[site] = delta; tree_.arcs[tree_.mate(arc_MX)].deltas[site] = {delta.to, delta.from}; }
    tree_.nodes[X].arc_to_focus = tree_.mate(arc_MX);
  }
  void attach_tip_to_focal_arc(int X, Arc_index best_arc, Node_index M) { split_best_arc_inserting_M(best_arc, M); wire_M_X(M, X); }   // :561-564
  void attach_subtree_to_focal_arc(Node_index X, Arc_index best_arc, Node_index M, Arc_index arc_DE) {   // :569-608
    split_best_arc_inserting_M(best_arc, M);
    ORC_CHECK(tree_.nodes[M].arc_to_focus != k_no_arc);
    ORC_CHECK(tree_.target(tree_.nodes[M].arc_to_focus) == tree_.focus);
    const Node_index D = tree_.origin(arc_DE), E = tree_.target(arc_DE);
    tree_.split_edge(arc_DE, X, [&](Seq_delta sd, Node_index orig, Node_index dst) -> Node_index {
      const State m = m_state(sd.site), d = sd.from, e = sd.to;
      if (m == e) return orig;
      else if (m == d) return dst;
      else { M_to_X_deltas_[sd.site] = {m, d}; return dst; }
    });
    wire_M_X(M, X);
    ORC_CHECK(tree_.nodes[D].arc_to_focus == k_no_arc || tree_.nodes[E].arc_to_focus == k_no_arc);
    tree_.nodes[D].arc_to_focus = tree_.find_arc(D, X);
    tree_.nodes[E].arc_to_focus = tree_.find_arc(E, X);
  }
  void strip_missing_deltas(Arc_index arc) {                         // :619-643
    for (auto node : {tree_.origin(arc), tree_.target(arc)}) {
      if (!tree_.is_tip(node)) continue;
      const auto& miss = tip_descs_[node].missations.intervals;
      const Arc_index arc_from_node = (tree_.origin(arc) == node) ? arc : tree_.mate(arc);
      std::vector<std::pair<Site_index, Site_delta>> stripped;
      for (const auto& [s, d] : tree_.arcs[arc_from_node].deltas) if (miss.contains(s)) stripped.push_back({s, d});
      for (const auto& [s, d] : stripped) {
        if (node == tree_.focus) {
          n_mismatches_ += (int)fitch_X_.contains(s, d.from, d.from) - (int)fitch_X_.contains(s, d.to, d.from);
          fitch_X_.on_ref_change(s, d.from, d.to);
          push_back_site_deltas({s, d.from, d.to}, tree_.deltas_ref_to_focus);
        }
        tree_.arcs[arc_from_node].deltas.erase(s);
        tree_.arcs[tree_.mate(arc_from_node)].deltas.erase(s);
      }
    }
  }
  void move_focus_updating_fitch_X(Node_index target) {              // :648-657
    tree_.move_focus_to(target, [&](Arc_index a) {
      for (const auto& [site, delta] : tree_.arcs[a].deltas) {
        n_mismatches_ += (int)fitch_X_.contains(site, delta.from, delta.from) - (int)fitch_X_.contains(site, delta.to, delta.from);
        fitch_X_.on_ref_change(site, delta.from, delta.to);
      }
    });
  }
  void init_component_picker() {                                     // :659-670
    const int total_nodes = 2 * tree_.num_tips - 1;
    component_abort_threshold_ = (int)std::sqrt((double)total_nodes * std::log2((double)total_nodes));
  }
  Node_index pick_random_node_in_component(Node_index sink) {        // :676-702
    ORC_CHECK(tree_.nodes[sink].arc_to_focus == k_no_arc);
    component_nodes_.clear(); component_dfs_stack_.clear();
    for (auto a : tree_.nodes[sink].arcs) if (a != k_no_arc) component_dfs_stack_.push_back(tree_.target(a));
    while (!component_dfs_stack_.empty()) {
      const Node_index V = component_dfs_stack_.back(); component_dfs_stack_.pop_back();
      component_nodes_.push_back(V);
      if ((int)component_nodes_.size() > component_abort_threshold_) {
        while (true) {
          const Node_index S = tree_.pick_random_node(rng_);
          Node_index cur = S;
          while (tree_.nodes[cur].arc_to_focus != k_no_arc) cur = tree_.target(tree_.nodes[cur].arc_to_focus);
          if (cur == sink) return S;
        }
      }
      for (auto a : tree_.nodes[V].arcs) if (a != k_no_arc && a != tree_.nodes[V].arc_to_focus) component_dfs_stack_.push_back(tree_.target(a));
    }
    ORC_CHECK(!component_nodes_.empty());
    return component_nodes_[rng_.uniform_int((int)component_nodes_.size())];
  }
  int eval_focal_arc(Arc_index a) const {                            // :708-718
    ORC_CHECK(tree_.origin(a) == tree_.focus);
    int savings = 0;
    for (const auto& [site, delta] : tree_.arcs[a].deltas) if (!fitch_X_.contains(site, delta.from, delta.from) && fitch_X_.contains(site, delta.to, delta.from)) ++savings;
    return n_mismatches_ - savings;
  }
};

inline Utree build_guide_tree(std::vector<State> ref_sequence, const std::vector<Tip_desc>& tip_descs, Rng& rng) {   // utree.cpp:744-755
  Utree_builder builder(std::move(ref_sequence), tip_descs, rng);
  for (int k = 0; k < (int)tip_descs.size(); ++k) builder.add_tip(k);
  return builder.finish();
}

// utree.cpp:761-896
inline void for_each_tip_in_nearest_first_order(const Utree& guide_tree, Rng& rng, const std::function<void(Node_index, Node_index)>& callback) {
  const int N = guide_tree.num_tips;
  if (N == 0) return;
  if (N == 1) { callback(0, k_no_node); return; }
  struct Arc_nearest { Node_index tip = k_no_node; int dist = 0; };
  std::vector<Arc_nearest> arc_nearest(guide_tree.arcs.size());
  const Node_index R = 0;
  for (auto [arc_X_to_P, direction] : guide_tree.annotated_arc_euler_tour(R)) {      // pass 1 (post-order)
    if (direction != Arc_direction::leaving) continue;
    const Node_index X = guide_tree.origin(arc_X_to_P);
    const Arc_index arc_P_to_X = guide_tree.mate(arc_X_to_P);
    const int deltas_P_X = guide_tree.count_arc_deltas(arc_P_to_X);
    Node_index closest = k_no_node; int d_X_T = std::numeric_limits<int>::max();
    for (auto a : guide_tree.nodes[X].arcs) { if (a == k_no_arc || a == arc_X_to_P) continue; if (arc_nearest[a].dist < d_X_T) { d_X_T = arc_nearest[a].dist; closest = arc_nearest[a].tip; } }
    if (closest == k_no_node) arc_nearest[arc_P_to_X] = {X, deltas_P_X}; else arc_nearest[arc_P_to_X] = {closest, deltas_P_X + d_X_T};
  }
  for (auto [arc_P_to_X, direction] : guide_tree.annotated_arc_euler_tour(R)) {      // pass 2 (pre-order)
    if (direction != Arc_direction::entering) continue;
    const Node_index P = guide_tree.origin(arc_P_to_X);
    const Arc_index arc_X_to_P = guide_tree.mate(arc_P_to_X);
    const int deltas_X_P = guide_tree.count_arc_deltas(arc_X_to_P);
    Node_index closest = k_no_node; int d_P_T = std::numeric_limits<int>::max();
    for (auto a : guide_tree.nodes[P].arcs) { if (a == k_no_arc || a == arc_P_to_X) continue; if (arc_nearest[a].dist < d_P_T) { d_P_T = arc_nearest[a].dist; closest = arc_nearest[a].tip; } }
    if (closest == k_no_node) arc_nearest[arc_X_to_P] = {P, deltas_X_P}; else arc_nearest[arc_X_to_P] = {closest, deltas_X_P + d_P_T};
  }
  struct Pq_entry { int dist; Arc_index arc; Node_index closest_prev_tip; int d_closest_prev_tip; bool operator>(const Pq_entry& o) const { return dist > o.dist; } };
  std::priority_queue<Pq_entry, std::vector<Pq_entry>, std::greater<Pq_entry>> pq;
  const Node_index S = guide_tree.pick_random_tip(rng);
  callback(S, k_no_node);
  for (auto a : guide_tree.nodes[S].arcs) if (a != k_no_arc) pq.push({arc_nearest[a].dist, a, S, 0});
  while (!pq.empty()) {
    const auto [dist, arc, H, d_I_H] = pq.top();
    pq.pop();
    const Node_index T = arc_nearest[arc].tip;
    const int d_T_I = dist;
    callback(T, H);
    Arc_index arc_into_N = arc;
    Node_index Nn = guide_tree.target(arc);
    int d_N_I = guide_tree.count_arc_deltas(arc);
    while (Nn != T) {
      const int d_N_H = d_N_I + d_I_H, d_N_T = d_T_I - d_N_I;
      const Node_index branch_closest = (d_N_T <= d_N_H) ? T : H;
      const int branch_d = (d_N_T <= d_N_H) ? d_N_T : d_N_H;
      Arc_index arc_out_of_N = k_no_arc;
      for (auto a : guide_tree.nodes[Nn].arcs) {
        if (a == k_no_arc || a == guide_tree.mate(arc_into_N)) continue;
        if (arc_nearest[a].tip == T) arc_out_of_N = a; else pq.push({arc_nearest[a].dist, a, branch_closest, branch_d});
      }
      ORC_CHECK(arc_out_of_N != k_no_arc);
      arc_into_N = arc_out_of_N;
      Nn = guide_tree.target(arc_out_of_N);
      d_N_I += guide_tree.count_arc_deltas(arc_out_of_N);
    }
  }
}

inline Utree build_refined_tree(const Utree& guide_tree, const std::vector<Tip_desc>& tip_descs, Rng& rng) {   // utree.cpp:898-914
  Utree_builder builder(guide_tree.ref_sequence, tip_descs, rng);
  for_each_tip_in_nearest_first_order(guide_tree, rng, [&](Node_index tip, Node_index closest_prev_tip) {
    if (closest_prev_tip != k_no_node) builder.move_focus_to(closest_prev_tip);
    builder.add_tip(tip);
  });
  return builder.finish();
}

// utree.cpp:920-1081
inline void spr_refine(Utree& tree, const std::vector<Tip_desc>& tip_descs, Rng& rng) {
  const int N = tree.num_tips;
  if (N <= 2) return;
  Utree_builder builder(std::move(tree), tip_descs, rng);
  Utree& t = builder.tree_;
  const int max_attempts = 30 * N;
  builder.init_component_picker();
  int consecutive_non_improvements = 0;
  int cur_deltas = t.count_deltas();
  for (int attempt = 0; attempt < max_attempts; ++attempt) {
    Node_index M;
    do { M = t.pick_random_node(rng); } while (t.degree(M) != 3);
    const auto m_arcs = t.nodes[M].arcs;
    const Arc_index arc_MX = m_arcs[rng.uniform_int(3)];
    ORC_CHECK(arc_MX != k_no_arc);
    const Node_index X = t.target(arc_MX);
    Arc_index arc_MP = k_no_arc, arc_MQ = k_no_arc;
    for (auto a : m_arcs) { if (a == arc_MX) continue; if (arc_MP == k_no_arc) arc_MP = a; else arc_MQ = a; }
    const Node_index P = t.target(arc_MP);
    const int d_MX = t.count_arc_deltas(arc_MX), d_MP = t.count_arc_deltas(arc_MP), d_MQ = t.count_arc_deltas(arc_MQ);
    int old_cost = 0, best_cost = 0;
    Arc_index best_arc = k_no_arc;
    if (t.is_tip(X)) {                                                // ---- tip SPR
      if (t.focus == X) t.move_focus_to(M);
      t.detach_tip(X);
      if (t.focus == M) t.move_focus_to(P);
      const Arc_index arc_PQ = t.merge_through(M);
      t.move_focus_to(t.origin(arc_PQ));
      builder.init_fitch_X_for_tip(X);
      builder.strip_missing_deltas(arc_PQ);
      const int d_PQ = t.count_arc_deltas(arc_PQ);
      old_cost = d_MX + d_MP + d_MQ - d_PQ;
      best_arc = arc_PQ;
      best_cost = builder.eval_focal_arc(arc_PQ);
      if (best_cost >= old_cost) {
        Node_index S;
        do { S = t.pick_random_node(rng); } while (S == X);
        builder.move_focus_updating_fitch_X(S);
        auto [found_arc, found_cost] = builder.find_best_attachment_arc();
        if (found_cost < best_cost) { best_arc = found_arc; best_cost = found_cost; }
      }
      builder.move_focus_updating_fitch_X(t.origin(best_arc));
      builder.attach_tip_to_focal_arc(X, best_arc, M);
    } else {                                                          // ---- subtree SPR
      ORC_CHECK(t.degree(X) == 3);
      const Arc_index arc_XM = t.mate(arc_MX);
      Arc_index arc_XD = k_no_arc, arc_XE = k_no_arc;
      for (auto a : t.nodes[X].arcs) { if (a == k_no_arc || a == arc_XM) continue; if (arc_XD == k_no_arc) arc_XD = a; else arc_XE = a; }
      const Node_index D = t.target(arc_XD);
      const int d_XD = t.count_arc_deltas(arc_XD), d_XE = t.count_arc_deltas(arc_XE);
      t.move_focus_to(M);
      builder.init_fitch_X_for_subtree(X);
      t.remove_edge(M, X);
      const Arc_index arc_DX = t.mate(arc_XD);
      ORC_CHECK(t.nodes[D].arc_to_focus == arc_DX);
      ORC_CHECK(t.nodes[X].arc_to_focus == k_no_arc);
      t.nodes[D].arc_to_focus = k_no_arc;
      t.nodes[X].arc_to_focus = arc_XD;
      const Arc_index arc_DE = t.merge_through(X);
      builder.strip_missing_deltas(arc_DE);
      builder.move_focus_updating_fitch_X(P);
      const Arc_index arc_PQ = t.merge_through(M);
      builder.strip_missing_deltas(arc_PQ);
      const int d_PQ = t.count_arc_deltas(arc_PQ), d_DE = t.count_arc_deltas(arc_DE);
      old_cost = d_MX + d_MP + d_MQ + d_XD + d_XE - d_PQ - d_DE;
      builder.move_focus_updating_fitch_X(P);
      best_arc = arc_PQ;
      best_cost = builder.eval_focal_arc(arc_PQ);
      if (best_cost >= old_cost) {
        const Node_index S = builder.pick_random_node_in_component(P);
        builder.move_focus_updating_fitch_X(S);
        auto [found_arc, found_cost] = builder.find_best_attachment_arc();
        if (found_cost < best_cost) { best_arc = found_arc; best_cost = found_cost; }
      }
      builder.move_focus_updating_fitch_X(t.origin(best_arc));
      builder.attach_subtree_to_focal_arc(X, best_arc, M, arc_DE);
    }
    const int delta_change = best_cost - old_cost;
    cur_deltas += delta_change;
    consecutive_non_improvements = (delta_change < 0) ? 0 : consecutive_non_improvements + 1;
    if (consecutive_non_improvements >= N) break;
  }
  tree = builder.finish();
}

// ---- rooting (utree.cpp:1085-1731) ----------------------------------------------------------------------------------------------
enum class Rooting_method { regression, midpoint };
struct Rooting_info { Node_index root; Rooting_method method; double r2; double lambda; double t_MRCA; };
inline double tip_mid_date(const Tip_desc& td) { return (double)(td.t_min + td.t_max) / 2.0; }   // (float sum first, as the reference's static_cast<double>(t_min + t_max) / 2.0)

inline std::pair<Node_index, int> farthest_node_from(const Utree& tree, Node_index start) {   // :1085-1103
  Node_index best_node = start; int best_dist = 0, cur_dist = 0;
  for (auto [arc, direction] : tree.annotated_arc_euler_tour(start)) {
    const int arc_deltas = tree.count_arc_deltas(arc);
    if (direction == Arc_direction::entering) { cur_dist += arc_deltas; if (cur_dist >= best_dist) { best_dist = cur_dist; best_node = tree.target(arc); } }
    else cur_dist -= arc_deltas;
  }
  return {best_node, best_dist};
}
inline Rooting_info midpoint_root_utree(Utree& tree, const std::vector<Tip_desc>& tip_descs) {   // :1122-1248
  const int N = tree.num_tips;
  constexpr double lambda_fallback = 1.0 / 30.0;
  if (N == 0) return {k_no_node, Rooting_method::midpoint, 0.0, lambda_fallback, 0.0};
  if (N == 1) return {0, Rooting_method::midpoint, 0.0, lambda_fallback, tip_mid_date(tip_descs[0])};
  auto [u, ignore] = farthest_node_from(tree, 0);
  (void)ignore;
  auto [v, D] = farthest_node_from(tree, u);
  ORC_CHECK(tree.is_tip(u)); ORC_CHECK(tree.is_tip(v));
  const double lambda_rough = 1.0 / 30.0;
  const double t_u = tip_mid_date(tip_descs[u]), t_v = tip_mid_date(tip_descs[v]);
  const double Md = (double)D;
  constexpr double k_min_root_branch_length = 14.0;
  const double t_R = std::min((t_u + t_v) / 2.0 - Md / (2.0 * lambda_rough), std::min(t_u, t_v) - k_min_root_branch_length);
  const double c = (t_u - t_R) / ((t_u - t_R) + (t_v - t_R));
  const int n_u_total = (int)std::lround(c * D);
  ORC_CHECK(n_u_total >= 0 && n_u_total <= D);
  tree.move_focus_to(v);
  int cum_dist = 0, n_u = 0;
  Node_index cur = u;
  Arc_index root_arc = k_no_arc;
  while (cur != v) {
    const Arc_index arc = tree.nodes[cur].arc_to_focus;
    ORC_CHECK(arc != k_no_arc);
    const int arc_deltas = tree.count_arc_deltas(arc);
    if (cum_dist + arc_deltas >= n_u_total) { root_arc = arc; n_u = n_u_total - cum_dist; break; }
    cum_dist += arc_deltas;
    cur = tree.target(arc);
  }
  ORC_CHECK(root_arc != k_no_arc);
  const Node_index R = tree.num_tips + tree.num_inner_nodes_so_far;
  tree.num_inner_nodes_so_far += 1;
  int deltas_assigned = 0;
  tree.split_edge(root_arc, R, [&](Seq_delta, Node_index A, Node_index B) -> Node_index { const Node_index side = (deltas_assigned < n_u) ? A : B; ++deltas_assigned; return side; });
  const double Nd = (double)N;
  double sum_t = 0.0;
  for (Node_index tip = 0; tip < N; ++tip) sum_t += tip_mid_date(tip_descs[tip]);
  const double mean_t = sum_t / Nd;
  int cur_dist = 0;
  double sum_m = 0.0, sum_m2 = 0.0, sum_dt2 = 0.0, sum_m_dt = 0.0;
  for (auto [arc, direction] : tree.annotated_arc_euler_tour(R)) {
    const int arc_deltas = tree.count_arc_deltas(arc);
    if (direction == Arc_direction::entering) {
      cur_dist += arc_deltas;
      const Node_index node = tree.target(arc);
      if (tree.is_tip(node)) {
        const double m_i = (double)cur_dist, dt_i = tip_mid_date(tip_descs[node]) - mean_t;
        sum_m += m_i; sum_m2 += m_i * m_i; sum_dt2 += dt_i * dt_i; sum_m_dt += m_i * dt_i;
      }
    } else cur_dist -= arc_deltas;
  }
  const double mean_m = sum_m / Nd, var_t = sum_dt2 / Nd, var_m = sum_m2 / Nd - mean_m * mean_m, cov_mt = sum_m_dt / Nd;
  const double r2 = (var_m > 0.0 && var_t > 0.0) ? (cov_mt * cov_mt) / (var_m * var_t) : 0.0;
  if (var_t > 0.0 && cov_mt > 0.0) { const double lambda = cov_mt / var_t; return {R, Rooting_method::midpoint, r2, lambda, mean_t - mean_m / lambda}; }
  return {R, Rooting_method::midpoint, r2, lambda_fallback, mean_t - mean_m / lambda_fallback};
}

// The two DFS passes that both regression rooters share (utree.cpp:1330-1388, 1585-1631): per-arc subtree statistics, first for the
// arcs pointing away from node 0 (leaves first), then for the arcs pointing toward it (root first)
template <class Stats, class TipStats, class Shift, class Combine>
inline std::vector<Stats> two_pass_arc_stats(const Utree& tree, TipStats tip_stats, Shift shift, Combine combine) {
  std::vector<Stats> st(tree.arcs.size());
  const Node_index F = 0;
  for (auto [arc_X_to_P, direction] : tree.annotated_arc_euler_tour(F)) {
    if (direction != Arc_direction::leaving) continue;
    const Node_index X = tree.origin(arc_X_to_P);
    const Arc_index arc_P_to_X = tree.mate(arc_X_to_P);
    if (tree.is_tip(X)) st[arc_P_to_X] = tip_stats(X);
    else { Stats combined{}; for (auto a : tree.nodes[X].arcs) if (a != k_no_arc && a != arc_X_to_P) combined = combine(combined, shift(st[a], tree.count_arc_deltas(a))); st[arc_P_to_X] = combined; }
  }
  for (auto [arc_P_to_X, direction] : tree.annotated_arc_euler_tour(F)) {
    if (direction != Arc_direction::entering) continue;
    const Node_index P = tree.origin(arc_P_to_X);
    const Arc_index arc_X_to_P = tree.mate(arc_P_to_X);
    if (tree.is_tip(P)) st[arc_X_to_P] = tip_stats(P);
    else { Stats combined{}; for (auto a : tree.nodes[P].arcs) if (a != k_no_arc && a != arc_P_to_X) combined = combine(combined, shift(st[a], tree.count_arc_deltas(a))); st[arc_X_to_P] = combined; }
  }
  return st;
}

inline Rooting_info ols_regression_root_utree(Utree& tree, const std::vector<Tip_desc>& tip_descs, Rng& rng) {   // :1255-1464
  const int N = tree.num_tips;
  const double Nd = (double)N;
  if (N <= 2) return midpoint_root_utree(tree, tip_descs);
  double sum_t = 0.0;
  for (Node_index tip = 0; tip < N; ++tip) sum_t += tip_mid_date(tip_descs[tip]);
  const double mean_t = sum_t / Nd;
  auto dt_of = [&](Node_index tip) { return tip_mid_date(tip_descs[tip]) - mean_t; };
  double sum_dt2 = 0.0;
  for (Node_index tip = 0; tip < N; ++tip) { const double dt = dt_of(tip); sum_dt2 += dt * dt; }
  const double var_t = sum_dt2 / Nd;
  if (var_t <= 0.0) return midpoint_root_utree(tree, tip_descs);
  struct Ols_stats { int n = 0; double sum_dt = 0.0, sum_m = 0.0, sum_m_dt = 0.0, sum_m2 = 0.0; };
  auto shift = [](const Ols_stats& s, int D) -> Ols_stats { const double Dd = (double)D; return {s.n, s.sum_dt, Dd * s.n + s.sum_m, Dd * s.sum_dt + s.sum_m_dt, Dd * Dd * s.n + 2 * Dd * s.sum_m + s.sum_m2}; };
  auto combine = [](const Ols_stats& a, const Ols_stats& b) -> Ols_stats { return {a.n + b.n, a.sum_dt + b.sum_dt, a.sum_m + b.sum_m, a.sum_m_dt + b.sum_m_dt, a.sum_m2 + b.sum_m2}; };
  const auto ols_stats = two_pass_arc_stats<Ols_stats>(tree, [&](Node_index tip) -> Ols_stats { return {1, dt_of(tip), 0, 0, 0}; }, shift, combine);
  double best_r2 = -1.0;
  std::vector<std::pair<Arc_index, int>> best_candidates;
  for (Arc_index arc_A_to_B = 0; arc_A_to_B < (int)tree.arcs.size(); arc_A_to_B += 2) {
    if (ols_stats[arc_A_to_B].n == 0 && ols_stats[tree.mate(arc_A_to_B)].n == 0) continue;
    const int D = tree.count_arc_deltas(arc_A_to_B);
    const auto& s_AB = ols_stats[arc_A_to_B]; const auto& s_BA = ols_stats[tree.mate(arc_A_to_B)];
    for (int k = 0; k <= D; ++k) {
      const Ols_stats root_stats = combine(shift(s_BA, k), shift(s_AB, D - k));
      const double cov_mt = root_stats.sum_m_dt / Nd;
      if (cov_mt <= 0.0) continue;
      const double mean_m = root_stats.sum_m / Nd, var_m = root_stats.sum_m2 / Nd - mean_m * mean_m;
      if (var_m <= 0.0) continue;
      const double r2 = (cov_mt * cov_mt) / (var_m * var_t);
      if (r2 > best_r2) { best_r2 = r2; best_candidates.clear(); }
      if (r2 == best_r2) best_candidates.push_back({arc_A_to_B, k});
    }
  }
  if (best_candidates.empty()) return midpoint_root_utree(tree, tip_descs);
  const auto [best_arc, best_k] = best_candidates[rng.uniform_int((int)best_candidates.size())];
  const int best_D = tree.count_arc_deltas(best_arc);
  const Ols_stats best_root_stats = combine(shift(ols_stats[tree.mate(best_arc)], best_k), shift(ols_stats[best_arc], best_D - best_k));
  const Node_index R = tree.num_tips + tree.num_inner_nodes_so_far;
  tree.num_inner_nodes_so_far += 1;
  int deltas_assigned = 0;
  const int bk = best_k;
  tree.split_edge(best_arc, R, [&](Seq_delta, Node_index A, Node_index B) -> Node_index { const Node_index side = (deltas_assigned < bk) ? A : B; ++deltas_assigned; return side; });
  const double cov_mt = best_root_stats.sum_m_dt / Nd, lambda = cov_mt / var_t, mean_m = best_root_stats.sum_m / Nd;
  return {R, Rooting_method::regression, best_r2, lambda, mean_t - mean_m / lambda};
}

inline Rooting_info gls_regression_root_utree(Utree& tree, const std::vector<Tip_desc>& tip_descs, Rng& rng) {   // :1470-1731
  const int N = tree.num_tips;
  const double Nd = (double)N;
  if (N <= 2) return midpoint_root_utree(tree, tip_descs);
  double sum_t = 0.0;
  for (Node_index tip = 0; tip < N; ++tip) sum_t += tip_mid_date(tip_descs[tip]);
  const double mean_t = sum_t / Nd;
  auto dt_of = [&](Node_index tip) { return tip_mid_date(tip_descs[tip]) - mean_t; };
  double sum_dt2 = 0.0;
  for (Node_index tip = 0; tip < N; ++tip) { const double dt = dt_of(tip); sum_dt2 += dt * dt; }
  const double var_t = sum_dt2 / Nd;
  if (var_t <= 0.0) return midpoint_root_utree(tree, tip_descs);
  const double epsilon = 0.05 * tree.count_deltas() / Nd;
  struct Gls_stats { double s11 = 0.0, sd1 = 0.0, sm1 = 0.0, sdd = 0.0, smd = 0.0, smm = 0.0; };   // 1W1, dtW1, mW1, dtWdt, mWdt, mWm
  auto shift = [epsilon](Gls_stats s, int z) -> Gls_stats {
    const double zd = (double)z, sigma_sq = zd + epsilon;
    if (s.s11 >= 0.0) {
      const double gamma = 1.0 / (1.0 + sigma_sq * s.s11);
      const double shifted_sm1 = s.sm1 + zd * s.s11;
      return {s.s11 * gamma, s.sd1 * gamma, shifted_sm1 * gamma, s.sdd - sigma_sq * s.sd1 * s.sd1 * gamma,
              (s.smd + zd * s.sd1) - sigma_sq * s.sd1 * shifted_sm1 * gamma, (s.smm + 2.0 * zd * s.sm1 + zd * zd * s.s11) - sigma_sq * shifted_sm1 * shifted_sm1 * gamma};
    }
    const double dt_X = s.sd1, inv = 1.0 / sigma_sq;
    return {inv, dt_X * inv, zd * inv, dt_X * dt_X * inv, zd * dt_X * inv, zd * zd * inv};
  };
  auto combine = [](const Gls_stats& a, const Gls_stats& b) -> Gls_stats {
    ORC_CHECK(a.s11 >= 0); ORC_CHECK(b.s11 >= 0);
    return {a.s11 + b.s11, a.sd1 + b.sd1, a.sm1 + b.sm1, a.sdd + b.sdd, a.smd + b.smd, a.smm + b.smm};
  };
  const auto gls_stats = two_pass_arc_stats<Gls_stats>(tree, [&](Node_index tip) -> Gls_stats { Gls_stats g; g.s11 = -1; g.sd1 = dt_of(tip); return g; }, shift, combine);
  double best_chi2 = std::numeric_limits<double>::infinity();
  std::vector<std::pair<Arc_index, int>> best_candidates;
  for (Arc_index arc_A_to_B = 0; arc_A_to_B < (int)tree.arcs.size(); arc_A_to_B += 2) {
    if (gls_stats[arc_A_to_B].s11 == 0.0 && gls_stats[tree.mate(arc_A_to_B)].s11 == 0.0) continue;
    const int D = tree.count_arc_deltas(arc_A_to_B);
    const auto& s_AB = gls_stats[arc_A_to_B]; const auto& s_BA = gls_stats[tree.mate(arc_A_to_B)];
    for (int k = 0; k <= D; ++k) {
      const Gls_stats r = combine(shift(s_BA, k), shift(s_AB, D - k));
      const double denom = r.sdd * r.s11 - r.sd1 * r.sd1;
      if (denom <= 0.0) continue;
      const double alpha = (r.smd * r.s11 - r.sm1 * r.sd1) / denom;
      if (alpha <= 0.0) continue;
      const double beta = (r.sm1 - alpha * r.sd1) / r.s11;
      const double chi2 = r.smm - alpha * r.smd - beta * r.sm1;
      if (chi2 < best_chi2) { best_chi2 = chi2; best_candidates.clear(); }
      if (chi2 == best_chi2) best_candidates.push_back({arc_A_to_B, k});
    }
  }
  if (best_candidates.empty()) return midpoint_root_utree(tree, tip_descs);
  const auto [best_arc, best_k] = best_candidates[rng.uniform_int((int)best_candidates.size())];
  const int best_D = tree.count_arc_deltas(best_arc);
  const Gls_stats b = combine(shift(gls_stats[tree.mate(best_arc)], best_k), shift(gls_stats[best_arc], best_D - best_k));
  const Node_index R = tree.num_tips + tree.num_inner_nodes_so_far;
  tree.num_inner_nodes_so_far += 1;
  int deltas_assigned = 0;
  const int bk = best_k;
  tree.split_edge(best_arc, R, [&](Seq_delta, Node_index A, Node_index B) -> Node_index { const Node_index side = (deltas_assigned < bk) ? A : B; ++deltas_assigned; return side; });
  int cur_dist = 0;
  double sum_m = 0.0, sum_m2 = 0.0, sum_m_dt = 0.0;
  for (auto [arc, direction] : tree.annotated_arc_euler_tour(R)) {
    const int arc_deltas = tree.count_arc_deltas(arc);
    if (direction == Arc_direction::entering) {
      cur_dist += arc_deltas;
      if (tree.is_tip(tree.target(arc))) { const double m = (double)cur_dist, dt = dt_of(tree.target(arc)); sum_m += m; sum_m2 += m * m; sum_m_dt += m * dt; }
    } else cur_dist -= arc_deltas;
  }
  const double mean_m = sum_m / Nd, cov_mt = sum_m_dt / Nd, var_m = sum_m2 / Nd - mean_m * mean_m;
  const double r2 = (var_m > 0.0 && var_t > 0.0) ? (cov_mt * cov_mt) / (var_m * var_t) : 0.0;
  const double best_denom = b.sdd * b.s11 - b.sd1 * b.sd1;
  const double alpha = (b.smd * b.s11 - b.sm1 * b.sd1) / best_denom;
  const double beta = (b.sm1 - alpha * b.sd1) / b.s11;
  return {R, Rooting_method::regression, r2, alpha, mean_t - beta / alpha};
}

// ---- utree_to_phylo_tree (utree.cpp:1750-1890) and the whole pipeline (:1892-1925) -------------------------------------------------
inline Phylo_tree utree_to_phylo_tree(Utree& utree, const Rooting_info& rooting_info, const std::vector<Tip_desc>& tip_descs, Rng& rng) {
  const int N = utree.num_tips;
  const Node_index root = rooting_info.root;
  const double lambda = rooting_info.lambda, t_root = rooting_info.t_MRCA;
  if (N == 0) return Phylo_tree{0};
  utree.move_focus_to(root);
  auto closing_checks = [&](Phylo_tree& phylo_tree) {
    const std::string m1 = check_phylo_tree_integrity(phylo_tree);
    if (!m1.empty()) throw std::runtime_error("utree_to_phylo_tree: " + m1);
    const std::string m2 = check_phylo_tree_matches_tip_descs(phylo_tree, utree.ref_sequence, tip_descs);
    if (!m2.empty()) throw std::runtime_error("utree_to_phylo_tree: " + m2);
  };
  if (N == 1) {
    Phylo_tree phylo_tree{1};
    phylo_tree.ref_sequence = utree.ref_sequence;
    phylo_tree.root = root;
    auto& nd = phylo_tree.at(root);
    nd.parent = k_no_node;
    ORC_CHECK(root < (int)tip_descs.size());
    nd.t_min = tip_descs[root].t_min; nd.t_max = tip_descs[root].t_max;
    nd.t = std::clamp(t_root, (double)nd.t_min, (double)nd.t_max);
    nd.missations = tip_descs[root].missations;
    for (const auto& [site, delta] : utree.deltas_ref_to_focus) nd.mutations.push_back(Mutation{delta.from, site, delta.to, nd.t});
    rereference_to_root_sequence(phylo_tree);
    closing_checks(phylo_tree);
    return phylo_tree;
  }
  ORC_CHECK(utree.num_inner_nodes_so_far == N - 1);
  Phylo_tree phylo_tree{2 * N - 1};
  phylo_tree.ref_sequence = utree.ref_sequence;
  phylo_tree.root = root;
  const double min_branch_length = 0.1;
  {
    auto& root_node = phylo_tree.at(root);
    root_node.parent = k_no_node; root_node.t = t_root;
    root_node.t_min = -std::numeric_limits<float>::max(); root_node.t_max = +std::numeric_limits<float>::max();
  }
  auto add_child = [&](Node_index P, Node_index X) { auto& p = phylo_tree.at(P); if (p.children[0] == k_no_node) p.children[0] = X; else { ORC_CHECK(p.children[1] == k_no_node); p.children[1] = X; } };
  int m_X = 0;
  for (auto [arc, direction] : utree.annotated_arc_euler_tour(root)) {
    const int arc_deltas = utree.count_arc_deltas(arc);
    if (direction == Arc_direction::entering) {
      const Node_index P = utree.origin(arc), X = utree.target(arc);
      auto& node_X = phylo_tree.at(X);
      m_X += arc_deltas;
      node_X.parent = P;
      add_child(P, X);
      const double t_X_est = t_root + (double)m_X / lambda;
      if (utree.is_tip(X)) {
        ORC_CHECK(X < (int)tip_descs.size());
        node_X.t_min = tip_descs[X].t_min; node_X.t_max = tip_descs[X].t_max;
        node_X.t = std::clamp(t_X_est, (double)tip_descs[X].t_min, (double)tip_descs[X].t_max);
        node_X.missations = tip_descs[X].missations;
      } else {
        node_X.t = t_X_est;
        node_X.t_min = -std::numeric_limits<float>::max(); node_X.t_max = +std::numeric_limits<float>::max();
      }
      for (const auto& [site, delta] : utree.arcs[arc].deltas) node_X.mutations.push_back(Mutation{delta.from, site, delta.to, node_X.t});
    } else {
      const Node_index X = utree.origin(arc);
      auto& node_X = phylo_tree.at(X);
      m_X -= arc_deltas;
      if (!utree.is_tip(X)) {
        const double t_children_min = std::min(phylo_tree.at(node_X.children[0]).t, phylo_tree.at(node_X.children[1]).t);
        node_X.t = std::min(node_X.t, t_children_min - min_branch_length);
      }
    }
  }
  auto& root_node = phylo_tree.at(root);
  for (const auto& [site, delta] : utree.deltas_ref_to_focus) root_node.mutations.push_back(Mutation{delta.from, site, delta.to, root_node.t});
  const double t_children_min = std::min(phylo_tree.at(root_node.children[0]).t, phylo_tree.at(root_node.children[1]).t);
  root_node.t = std::min(root_node.t, t_children_min - min_branch_length);
  fix_up_missations(phylo_tree);
  randomize_mutation_times(phylo_tree, rng);
  rereference_to_root_sequence(phylo_tree);
  closing_checks(phylo_tree);
  return phylo_tree;
}

struct Initial_tree_report { int guide_deltas = 0, refined_rounds = 0, refined_deltas = 0, spr_deltas = 0; Rooting_info rooting{k_no_node, Rooting_method::midpoint, 0, 0, 0}; };
inline Phylo_tree build_initial_phylo_tree(std::vector<State> ref_sequence, const std::vector<Tip_desc>& tip_descs, Rng& rng, Initial_tree_report* report = nullptr) {
  Utree utree = build_guide_tree(std::move(ref_sequence), tip_descs, rng);
  constexpr int k_max_refinement_rounds = 5;
  int prev_deltas = utree.count_deltas();
  if (report) report->guide_deltas = prev_deltas;
  for (int round = 1; round <= k_max_refinement_rounds; ++round) {
    Utree refined = build_refined_tree(utree, tip_descs, rng);
    const int refined_deltas = refined.count_deltas();
    if (refined_deltas >= prev_deltas) break;
    prev_deltas = refined_deltas;
    utree = std::move(refined);
    if (report) report->refined_rounds = round;
  }
  if (report) report->refined_deltas = prev_deltas;
  spr_refine(utree, tip_descs, rng);
  if (report) report->spr_deltas = utree.count_deltas();
  const Rooting_info rooting_info = ols_regression_root_utree(utree, tip_descs, rng);
  if (report) report->rooting = rooting_info;
  return utree_to_phylo_tree(utree, rooting_info, tip_descs, rng);
}

}  // namespace orc
#endif  // ORC_UTREE_HPP_
