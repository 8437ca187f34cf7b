// orc_calc.hpp -- CPU ORACLE (test infrastructure, NOT product code).
// Restates reference core/phylo_tree_calc.{h,cpp}, core/site_deltas.{h,cpp} and the tree helpers of
// core/phylo_tree.cpp that the local-move path uses.
#ifndef ORC_CALC_HPP_
#define ORC_CALC_HPP_

#include <array>
#include "orc_core.hpp"

namespace orc {

// ---- site deltas (reference core/site_deltas.h:13-154) ------------------------------------------
// The reference keeps these in an absl::flat_hash_map; every order-sensitive use is followed by a
// sort or is order-independent, so the oracle iterates in ascending site order.
struct Site_delta { State from; State to; bool operator==(const Site_delta& o) const { return from == o.from && to == o.to; } };
using Site_deltas = std::map<Site_index, Site_delta>;

inline void push_front_site_deltas(Seq_delta dz, Site_deltas& sd) {   // site_deltas.h:43-65
  auto [it, inserted] = sd.try_emplace(dz.site, Site_delta{dz.from, dz.to});
  if (!inserted) {
    ORC_CHECK(dz.to == it->second.from);
    it->second.from = dz.from;
    if (it->second.from == it->second.to) sd.erase(it);
  }
}
inline void pop_front_site_deltas(Seq_delta dz, Site_deltas& sd) { push_front_site_deltas(dz.inverse(), sd); }  // :69-83
inline void push_back_site_deltas(Seq_delta dz, Site_deltas& sd) {    // site_deltas.h:88-110
  auto [it, inserted] = sd.try_emplace(dz.site, Site_delta{dz.from, dz.to});
  if (!inserted) {
    ORC_CHECK(dz.from == it->second.to);
    it->second.to = dz.to;
    if (it->second.from == it->second.to) sd.erase(it);
  }
}
inline void pop_back_site_deltas(Seq_delta dz, Site_deltas& sd) { push_back_site_deltas(dz.inverse(), sd); }
inline void append_site_deltas(Site_deltas& x_to_y, const Site_deltas& y_to_z) {   // :150-154
  for (auto& [l, d] : y_to_z) push_back_site_deltas({l, d.from, d.to}, x_to_y);
}

// ---- tree helpers (reference core/phylo_tree.cpp:204-312, 579-644) ------------------------------
inline Node_index find_MRCA_of(const Phylo_tree& tree, Node_index P, Node_index Q) {   // :204-280
  if (P == k_no_node) return P;
  if (Q == k_no_node) return Q;
  while (P != Q) {
    double tP = tree.at(P).t, tQ = tree.at(Q).t;
    if (tP > tQ) { P = tree.at(P).parent; ORC_CHECK(P != k_no_node); }
    else if (tP < tQ) { Q = tree.at(Q).parent; ORC_CHECK(Q != k_no_node); }
    else if (tree.at(P).is_tip()) { P = tree.at(P).parent; ORC_CHECK(P != k_no_node); }
    else if (tree.at(Q).is_tip()) { Q = tree.at(Q).parent; ORC_CHECK(Q != k_no_node); }
    else {
      std::vector<Node_index> aP, aQ;
      for (auto c = P; c != k_no_node; c = tree.at(c).parent) aP.push_back(c);
      for (auto c = Q; c != k_no_node; c = tree.at(c).parent) aQ.push_back(c);
      Node_index cand = tree.root;
      while (!aP.empty() && !aQ.empty() && aP.back() == aQ.back()) { cand = aP.back(); aP.pop_back(); aQ.pop_back(); }
      return cand;
    }
  }
  return P;
}
inline Phylo_tree_loc find_MRCA_of(const Phylo_tree& tree, Phylo_tree_loc p, Phylo_tree_loc q) {   // :282-290
  if (p.branch == q.branch) return p.t < q.t ? p : q;
  Node_index A = find_MRCA_of(tree, p.branch, q.branch);
  return {A, std::min(p.t, std::min(q.t, tree.at(A).t))};
}
inline bool descends_from(const Phylo_tree& tree, Node_index X, Node_index A) {   // :292-299
  if (A == k_no_node) return true;
  for (auto cur = X; cur != k_no_node; cur = tree.at(cur).parent) {
    if (cur == A) return true;
    if (tree.at(cur).t < tree.at(A).t) return false;
  }
  return false;
}
inline bool descends_from(const Phylo_tree& tree, Phylo_tree_loc x, Phylo_tree_loc a) {   // :301-307
  if (x.branch == a.branch) return a.t <= x.t;
  return descends_from(tree, x.branch, a.branch);
}
inline void rereference_to_root_sequence(Phylo_tree& tree) {   // :309-322
  for (auto& m : tree.at_root().mutations) tree.ref_sequence[m.site] = m.to;
  for (int n = 0; n < tree.size(); ++n)
    if (!tree.at(n).missations.empty())
      for (auto& m : tree.at_root().mutations) tree.at(n).missations.ref_seq_changed(m.site, m.from, m.to);
  tree.at_root().mutations.clear();
}
// phylo_tree.cpp:579-644.  The "complicated" path iterates a hash set in the reference; the oracle
// iterates sites in ascending order.
inline Mutation_list randomize_branch_mutation_times(const Phylo_tree& tree, Branch_index X, Rng& rng) {
  const auto& old = tree.at(X).mutations;
  if (X == tree.root) return old;
  double t_X = tree.at(X).t, t_P = tree.at_parent_of(X).t;
  std::map<Site_index, int> counts;
  bool complicated = false;
  for (auto& m : old) if (++counts[m.site] > 1) complicated = true;
  Mutation_list out;
  if (!complicated) {
    for (auto& m : old) out.push_back(Mutation{m.from, m.site, m.to, rng.uniform_oc(t_P, t_X)});
  } else {
    for (auto& [l, cnt] : counts) {
      std::vector<double> ts;
      for (auto& m : old) if (m.site == l) ts.push_back(rng.uniform_oc(t_P, t_X));
      std::sort(ts.begin(), ts.end());
      size_t k = 0;
      for (auto& m : old) if (m.site == l) out.push_back(Mutation{m.from, m.site, m.to, ts[k++]});
    }
  }
  sort_mutations(out);
  return out;
}

// ---- phylo_tree_calc (reference core/phylo_tree_calc.{h,cpp}) -----------------------------------
inline Site_deltas deltas_ref_to_loc(const Phylo_tree& tree, Phylo_tree_loc x) {   // view_of_sequence_at :19-35
  Site_deltas sd;
  for (auto cur = x.branch; cur != k_no_node; cur = tree.at(cur).parent) {
    auto& ms = tree.at(cur).mutations;
    for (auto it = ms.rbegin(); it != ms.rend(); ++it) if (it->t <= x.t) push_front_site_deltas(*it, sd);
  }
  return sd;
}
inline std::vector<State> view_of_sequence_at(const Phylo_tree& tree, Phylo_tree_loc x) {
  auto seq = tree.ref_sequence;
  for (auto& [l, d] : deltas_ref_to_loc(tree, x)) { ORC_CHECK(d.from == tree.ref_sequence[l]); seq[l] = d.to; }
  return seq;
}
inline std::vector<State> view_of_sequence_at(const Phylo_tree& tree, Node_index X) { return view_of_sequence_at(tree, tree.node_loc(X)); }

inline Interval_set reconstruct_missing_sites_at(const Phylo_tree& tree, Node_index node) {   // :41-56
  Interval_set so_far, scratch;
  for (auto cur = node; cur != k_no_node; cur = tree.at(cur).parent) {
    merge_interval_sets(scratch, so_far, tree.at(cur).missations.intervals);
    std::swap(so_far, scratch);
  }
  return so_far;
}
inline bool is_site_missing_at(const Phylo_tree& tree, Node_index node, Site_index l) {   // :58-65
  for (auto cur = node; cur != k_no_node; cur = tree.at(cur).parent) if (tree.at(cur).missations.contains(l)) return true;
  return false;
}
inline std::vector<int> calc_num_sites_missing_at_every_node(const Phylo_tree& tree) {   // :67-76
  std::vector<int> r(tree.size());
  for (auto n : pre_order(tree)) r[n] = (n == tree.root ? 0 : r[tree.at(n).parent]) + tree.at(n).missations.num_sites();
  return r;
}
inline void recalc_num_sites_missing_upstream(const Phylo_tree& tree, Node_index node, Node_index ancestor, std::vector<int>& nm) {  // :78-93
  int cur_m = nm[node];
  for (auto cur = node; cur != ancestor; cur = tree.at(cur).parent) {
    if (cur != tree.root) { int pm = cur_m - tree.at(cur).missations.num_sites(); nm[tree.at(cur).parent] = pm; cur_m = pm; }
  }
}
using State_freqs = std::vector<std::array<int, 4>>;
inline State_freqs calc_state_frequencies_per_partition_of(const std::vector<State>& seq, const Global_evo_model& evo) {  // :95-106
  State_freqs r(evo.num_partitions(), std::array<int, 4>{0, 0, 0, 0});
  for (int l = 0; l < (int)seq.size(); ++l) ++r[evo.partition_for_site[l]][seq[l]];
  return r;
}
inline State calc_site_state_at(const Phylo_tree& tree, Phylo_tree_loc q, Site_index l) {   // :108-118
  for (auto cur = q.branch; cur != k_no_node; cur = tree.at(cur).parent) {
    auto& ms = tree.at(cur).mutations;
    for (auto it = ms.rbegin(); it != ms.rend(); ++it) {
      if (it->t > q.t) continue;
      if (it->site == l) return it->to;
    }
  }
  return tree.ref_sequence[l];
}
inline std::vector<double> calc_cum_Q_l_for_sequence(const std::vector<State>& seq, const Global_evo_model& evo) {  // :379-388
  std::vector<double> c(seq.size() + 1, 0.0);
  double so_far = 0.0;
  for (int l = 0; l < (int)seq.size(); ++l) { so_far += evo.mu_l(l) * evo.nu_l[l] * evo.q_l_a(l, seq[l]); c[l + 1] = so_far; }
  return c;
}
inline double calc_lambda_for_sequence(const std::vector<State>& seq, const Global_evo_model& evo) {  // :390-399
  double lam = 0.0;
  for (int l = 0; l < (int)seq.size(); ++l) lam += evo.mu_l(l) * evo.nu_l[l] * evo.q_l_a(l, seq[l]);
  return lam;
}
// phylo_tree_calc.h:121-138
inline double calc_delta_lambda_across_missations(const Global_evo_model& evo, const std::vector<State>& ref,
                                                  const std::vector<double>& cumQ, const Missation_map& mi) {
  double r = 0.0;
  for (auto& [s, e] : mi.intervals.v) r -= cumQ[e] - cumQ[s];
  for (auto& [l, from] : mi.from_states) r -= evo.mu_l(l) * evo.nu_l[l] * (evo.q_l_a(l, from) - evo.q_l_a(l, ref[l]));
  return r;
}
// phylo_tree_calc.h:140-155
inline double calc_delta_lambda_across_branch(const Global_evo_model& evo, const std::vector<State>& ref,
                                              const std::vector<double>& cumQ, const Mutation_list& muts, const Missation_map& mi) {
  double r = 0.0;
  for (auto& m : muts) r += evo.mu_l(m.site) * evo.nu_l[m.site] * (evo.q_l_a(m.site, m.to) - evo.q_l_a(m.site, m.from));
  r += calc_delta_lambda_across_missations(evo, ref, cumQ, mi);
  return r;
}
inline double calc_lambda_at_node(const Phylo_tree& tree, Node_index node, const Global_evo_model& evo, const std::vector<double>& cumQ) {  // :406-418
  double r = cumQ.back();
  for (auto cur = node; cur != k_no_node; cur = tree.at(cur).parent)
    r += calc_delta_lambda_across_branch(evo, tree.ref_sequence, cumQ, tree.at(cur).mutations, tree.at(cur).missations);
  return r;
}
inline std::vector<double> calc_lambda_i(const Phylo_tree& tree, const Global_evo_model& evo, const std::vector<double>& cumQ) {  // :420-436
  std::vector<double> r(tree.size());
  double lam_ref = cumQ.back();
  for (auto n : pre_order(tree)) {
    double lp = (n == tree.root) ? lam_ref : r[tree.at(n).parent];
    r[n] = lp + calc_delta_lambda_across_branch(evo, tree.ref_sequence, cumQ, tree.at(n).mutations, tree.at(n).missations);
  }
  return r;
}
inline void recalc_lambda_i_upstream(const Phylo_tree& tree, Node_index node, Node_index ancestor, const Global_evo_model& evo,
                                     std::vector<double>& lambda_i, const std::vector<double>& cumQ) {   // :438-456
  double le = lambda_i[node];
  for (auto cur = node; cur != ancestor; cur = tree.at(cur).parent) {
    if (cur != tree.root) {
      double lp = le - calc_delta_lambda_across_branch(evo, tree.ref_sequence, cumQ, tree.at(cur).mutations, tree.at(cur).missations);
      lambda_i[tree.at(cur).parent] = lp; le = lp;
    }
  }
}
inline double calc_log_root_prior(const Phylo_tree& tree, const Global_evo_model& evo, const State_freqs& ref_freqs) {   // :467-504
  auto f = ref_freqs;
  auto& root = tree.at_root();
  for (auto& m : root.mutations) { int p = evo.partition_for_site[m.site]; --f[p][m.from]; ++f[p][m.to]; }
  for (auto& [s, e] : root.missations.intervals.v) for (int l = s; l != e; ++l) --f[evo.partition_for_site[l]][tree.ref_sequence[l]];
  for (auto& [l, from] : root.missations.from_states) { int p = evo.partition_for_site[l]; ++f[p][tree.ref_sequence[l]]; --f[p][from]; }
  double r = 0.0;
  for (int p = 0; p < evo.num_partitions(); ++p) {
    auto& pi = evo.partition_evo_model[p].pi_a;
    for (int a = 0; a < 4; ++a) {
      if (pi[a] != 0.0) r += f[p][a] * std::log(pi[a]);
      else if (f[p][a] != 0) return -std::numeric_limits<double>::infinity();
    }
  }
  return r;
}
inline double calc_log_root_prior(const Phylo_tree& tree, const Global_evo_model& evo) {
  return calc_log_root_prior(tree, evo, calc_state_frequencies_per_partition_of(tree.ref_sequence, evo));
}
// phylo_tree_calc.h:185-206
inline double calc_branch_log_G(double t_P, double t_X, double lambda_X, const Global_evo_model& evo, const Mutation_list& muts) {
  double r = -lambda_X * (t_X - t_P);
  for (auto it = muts.rbegin(); it != muts.rend(); ++it) {
    auto& m = *it; int l = m.site;
    r -= evo.mu_l(l) * evo.nu_l[l] * (evo.q_l_a(l, m.from) - evo.q_l_a(l, m.to)) * (m.t - t_P);
    r += std::log(evo.mu_l(l) * evo.nu_l[l] * evo.q_l_ab(l, m.from, m.to));
  }
  return r;
}
inline double calc_branch_log_G(const Phylo_tree& tree, Branch_index X, double lambda_X, const Global_evo_model& evo, const State_freqs& rf) {  // :545-558
  if (X == tree.root) return calc_log_root_prior(tree, evo, rf);
  return calc_branch_log_G(tree.at_parent_of(X).t, tree.at(X).t, lambda_X, evo, tree.at(X).mutations);
}
inline double calc_log_G_below_root(const Phylo_tree& tree, const Global_evo_model& evo, const std::vector<double>& lambda_i, const State_freqs& rf) {  // :515-543
  double r = 0.0;
  for (int n = 0; n < tree.size(); ++n) if (n != tree.root) r += calc_branch_log_G(tree, n, lambda_i[n], evo, rf);
  return r;
}
inline double calc_log_G_below_root(const Phylo_tree& tree, const Global_evo_model& evo) {
  auto cumQ = calc_cum_Q_l_for_sequence(tree.ref_sequence, evo);
  return calc_log_G_below_root(tree, evo, calc_lambda_i(tree, evo, cumQ), calc_state_frequencies_per_partition_of(tree.ref_sequence, evo));
}
inline double calc_path_log_G(const Phylo_tree& tree, Node_index A, Node_index B, const Global_evo_model& evo,
                              const std::vector<double>& lambda_i, const State_freqs& rf) {   // :560-575
  double r = 0.0;
  for (auto cur = B; cur != A; cur = tree.at(cur).parent) r += calc_branch_log_G(tree, cur, lambda_i[cur], evo, rf);
  return r;
}
inline int calc_num_muts(const Phylo_tree& tree) {   // :577-585
  int n = 0;
  for (int i = 0; i < tree.size(); ++i) if (i != tree.root) n += (int)tree.at(i).mutations.size();
  return n;
}
// Total branch length below every node (first loop of calc_T_l_a / calc_Ttwiddle_l, :130-142, :179-189), children before parents.
inline std::vector<double> calc_T_below_node(const Phylo_tree& tree) {
  std::vector<double> T(tree.size(), 0.0);
  std::vector<std::pair<Node_index, int>> stack; stack.push_back({tree.root, 0});
  while (!stack.empty()) {
    auto& [node, k] = stack.back();
    if (tree.at(node).is_tip() || k == 2) {
      double sum = 0.0;
      if (!tree.at(node).is_tip()) for (Node_index ch : tree.at(node).children) sum += (tree.at(ch).t - tree.at(node).t) + T[ch];
      T[node] = sum;
      stack.pop_back();
    } else { Node_index ch = tree.at(node).children[k]; ++k; stack.push_back({ch, 0}); }
  }
  return T;
}
inline std::vector<std::array<double, 4>> calc_T_l_a(const Phylo_tree& tree) {   // :130-174
  auto T_below_node = calc_T_below_node(tree);
  std::vector<std::array<double, 4>> T_l_a(tree.num_sites(), std::array<double, 4>{0.0, 0.0, 0.0, 0.0});
  const double T = T_below_node[tree.root];
  for (int l = 0; l < tree.num_sites(); ++l) T_l_a[l][tree.ref_sequence[l]] = T;
  for (int node = 0; node < tree.size(); ++node) {
    for (const auto& m : tree.at(node).mutations) {
      double T_below_mut = T_below_node[node] + (node == tree.root ? 0.0 : tree.at(node).t - m.t);
      T_l_a[m.site][m.from] -= T_below_mut;
      T_l_a[m.site][m.to] += T_below_mut;
    }
    double T_below_miss = T_below_node[node] + (node == tree.root ? 0.0 : tree.at(node).t - tree.at_parent_of(node).t);
    for (const auto& iv : tree.at(node).missations.intervals.v) for (int l = iv.first; l < iv.second; ++l) T_l_a[l][tree.ref_sequence[l]] -= T_below_miss;
    for (const auto& [l, from] : tree.at(node).missations.from_states) { T_l_a[l][tree.ref_sequence[l]] += T_below_miss; T_l_a[l][from] -= T_below_miss; }
  }
  return T_l_a;
}
inline std::vector<double> calc_Ttwiddle_l(const Phylo_tree& tree, const Global_evo_model& evo) {   // :176-222
  auto T_below_node = calc_T_below_node(tree);
  std::vector<double> Ttwiddle_l(tree.num_sites(), 0.0);
  const double T = T_below_node[tree.root];
  for (int l = 0; l < tree.num_sites(); ++l) Ttwiddle_l[l] = evo.q_l_a(l, tree.ref_sequence[l]) * T;
  for (int node = 0; node < tree.size(); ++node) {
    for (const auto& m : tree.at(node).mutations) {
      double T_below_mut = T_below_node[node] + (node == tree.root ? 0.0 : tree.at(node).t - m.t);
      Ttwiddle_l[m.site] -= evo.q_l_a(m.site, m.from) * T_below_mut;
      Ttwiddle_l[m.site] += evo.q_l_a(m.site, m.to) * T_below_mut;
    }
    double T_below_miss = T_below_node[node] + (node == tree.root ? 0.0 : tree.at(node).t - tree.at_parent_of(node).t);
    for (const auto& iv : tree.at(node).missations.intervals.v) for (int l = iv.first; l < iv.second; ++l) Ttwiddle_l[l] -= evo.q_l_a(l, tree.ref_sequence[l]) * T_below_miss;
    for (const auto& [l, from] : tree.at(node).missations.from_states) {
      Ttwiddle_l[l] += evo.q_l_a(l, tree.ref_sequence[l]) * T_below_miss;   // undo ref_seq assumption
      Ttwiddle_l[l] -= evo.q_l_a(l, from) * T_below_miss;                   // apply correct from_state
    }
  }
  return Ttwiddle_l;
}
inline std::vector<int> calc_num_muts_l(const Phylo_tree& tree) {   // :612-622
  std::vector<int> r(tree.num_sites(), 0);
  for (int i = 0; i < tree.size(); ++i) if (i != tree.root) for (const auto& m : tree.at(i).mutations) ++r[m.site];   // "mutations" above the root are deltas from the reference sequence
  return r;
}
// ---- sufficient statistics of the global moves (phylo_tree_calc.cpp:288-369, 577-610) ----------------------------
// Ttwiddle^beta_a = sum_{l in beta} nu_l T^(l)_a: the nu-weighted time every site of partition beta spends in state a
// over all branches below the root, sites missing on a branch excluded.  Follows the reference's enter / exit traversal.
typedef std::vector<std::array<double, 4>> Partition_state_times;
inline Partition_state_times calc_Ttwiddle_beta_a(const Phylo_tree& tree, const Global_evo_model& evo) {
  const int P = (int)evo.partition_evo_model.size();
  Partition_state_times T(P, std::array<double, 4>{0, 0, 0, 0}), n(P, std::array<double, 4>{0, 0, 0, 0});
  for (int l = 0; l < (int)tree.ref_sequence.size(); ++l) n[evo.partition_for_site[l]][tree.ref_sequence[l]] += evo.nu_l[l];
  auto enter = [&](Node_index node) {   // n: state at parent -> state at node
    const auto& nd = tree.at(node);
    for (const auto& iv : nd.missations.intervals.v) for (int l = iv.first; l != iv.second; ++l) n[evo.partition_for_site[l]][tree.ref_sequence[l]] -= evo.nu_l[l];
    for (const auto& [l, from] : nd.missations.from_states) { int b = evo.partition_for_site[l]; n[b][tree.ref_sequence[l]] += evo.nu_l[l]; n[b][from] -= evo.nu_l[l]; }
    for (const auto& m : nd.mutations) { int b = evo.partition_for_site[m.site]; n[b][m.from] -= evo.nu_l[m.site]; n[b][m.to] += evo.nu_l[m.site]; }
    if (node != tree.root) {
      const double t_P = tree.at(nd.parent).t, len = nd.t - t_P;
      for (int b = 0; b < P; ++b) for (int a = 0; a < 4; ++a) T[b][a] += n[b][a] * len;
      for (auto it = nd.mutations.rbegin(); it != nd.mutations.rend(); ++it) {
        int b = evo.partition_for_site[it->site];
        T[b][it->to] -= evo.nu_l[it->site] * (it->t - t_P);
        T[b][it->from] += evo.nu_l[it->site] * (it->t - t_P);
      }
    }
  };
  auto leave = [&](Node_index node) {   // n: state at node -> state at parent
    const auto& nd = tree.at(node);
    for (const auto& m : nd.mutations) { int b = evo.partition_for_site[m.site]; n[b][m.to] -= evo.nu_l[m.site]; n[b][m.from] += evo.nu_l[m.site]; }
    for (const auto& iv : nd.missations.intervals.v) for (int l = iv.first; l != iv.second; ++l) n[evo.partition_for_site[l]][tree.ref_sequence[l]] += evo.nu_l[l];
    for (const auto& [l, from] : nd.missations.from_states) { int b = evo.partition_for_site[l]; n[b][tree.ref_sequence[l]] -= evo.nu_l[l]; n[b][from] += evo.nu_l[l]; }
  };
  if (tree.size() == 0) return T;
  std::vector<std::pair<Node_index, int>> stack;   // traversal(tree): (node, children visited so far)
  stack.push_back({tree.root, 0});
  enter(tree.root);
  while (!stack.empty()) {
    auto& [node, k] = stack.back();
    if (tree.at(node).is_tip() || k == 2) { leave(node); stack.pop_back(); }
    else { Node_index ch = tree.at(node).children[k]; ++k; stack.push_back({ch, 0}); enter(ch); }
  }
  return T;
}
typedef std::vector<std::array<std::array<long long, 4>, 4>> Partition_mut_counts;
inline Partition_mut_counts calc_num_muts_beta_ab(const Phylo_tree& tree, const Global_evo_model& evo) {   // :599-610
  Partition_mut_counts r(evo.partition_evo_model.size());
  for (auto& x : r) for (auto& y : x) y = {0, 0, 0, 0};
  for (int i = 0; i < tree.size(); ++i) if (i != tree.root) for (const auto& m : tree.at(i).mutations) ++r[evo.partition_for_site[m.site]][m.from][m.to];
  return r;
}
inline double calc_max_tip_time(const Phylo_tree& tree) {   // :636-644
  double t = -std::numeric_limits<double>::infinity();
  for (int i = 0; i < tree.size(); ++i) if (tree.at(i).is_tip() && tree.at(i).t_max > t) t = tree.at(i).t_max;
  return t;
}
inline double calc_T(const Phylo_tree& tree) {   // :120-128
  double T = 0.0;
  for (int n = 0; n < tree.size(); ++n) if (n != tree.root) T += tree.at(n).t - tree.at_parent_of(n).t;
  return T;
}

// ---- site deltas between tree locations (reference core/site_deltas.cpp:7-101) --------------------
inline void displace_site_deltas_start_upwards(const Phylo_tree& tree, Site_deltas& sd, Phylo_tree_loc x, Phylo_tree_loc a) {   // :7-38
  ORC_CHECK(a.branch != k_no_node);
  auto G = tree.at(a.branch).parent;
  for (auto cur = x.branch; cur != G; cur = tree.at(cur).parent) {
    auto& ms = tree.at(cur).mutations;
    for (auto it = ms.rbegin(); it != ms.rend(); ++it) if (a.t <= it->t && it->t <= x.t) push_front_site_deltas(*it, sd);
  }
}
inline void displace_site_deltas_start_downwards(const Phylo_tree& tree, Site_deltas& sd, Phylo_tree_loc a, Phylo_tree_loc x) {   // :40-80
  ORC_CHECK(a.branch != k_no_node);
  auto G = tree.at(a.branch).parent;
  std::vector<Branch_index> br;
  for (auto cur = x.branch; cur != G; cur = tree.at(cur).parent) br.push_back(cur);
  for (auto it = br.rbegin(); it != br.rend(); ++it)
    for (auto& m : tree.at(*it).mutations) if (a.t <= m.t && m.t <= x.t) pop_front_site_deltas(m, sd);
}
inline Site_deltas calc_site_deltas_between(const Phylo_tree& tree, Phylo_tree_loc x, Phylo_tree_loc y) {   // :82-101
  Site_deltas sd;
  auto a = find_MRCA_of(tree, x, y);
  if (y != a) displace_site_deltas_start_upwards(tree, sd, y, a);
  if (x != a) displace_site_deltas_start_downwards(tree, sd, a, x);
  return sd;
}
inline Site_deltas calc_site_deltas_between(const Phylo_tree& tree, Node_index X, Node_index Y) {
  return calc_site_deltas_between(tree, tree.node_loc(X), tree.node_loc(Y));
}

// ---- integrity checks (reference core/tree.h:371-412, core/phylo_tree.cpp:18-136), returning a
//      message instead of aborting so that tests can report ------------------------------------------
inline std::string check_phylo_tree_integrity(const Phylo_tree& tree) {
  char buf[256];
  if (tree.size() == 0) return tree.root == k_no_node ? "" : "empty tree with a root";
  if (tree.root < 0 || tree.root >= tree.size()) return "bad root";
  if (tree.at(tree.root).parent != k_no_node) return "root has a parent";
  std::vector<char> seen(tree.size(), 0);
  auto order = pre_order(tree);
  for (auto n : order) {
    if (seen[n]) return "node visited twice";
    seen[n] = 1;
    auto& nd = tree.at(n);
    if (nd.is_inner_node()) {
      if (nd.children[1] == k_no_node) return "inner node with one child";
      for (int k = 0; k < 2; ++k) {
        int c = nd.children[k];
        if (c < 0 || c >= tree.size() || tree.at(c).parent != n) { std::snprintf(buf, sizeof buf, "child link broken at %d", n); return buf; }
        if (!(nd.t - 1e-2 <= tree.at(c).t + 1e-2)) { std::snprintf(buf, sizeof buf, "child %d earlier than parent %d", c, n); return buf; }
      }
      if (nd.t_min != -FLT_MAX || nd.t_max != FLT_MAX) { std::snprintf(buf, sizeof buf, "inner node %d has finite t bounds", n); return buf; }
    }
    if (!(nd.t_min - 1e-2 <= nd.t + 1e-2 && nd.t - 1e-2 <= nd.t_max + 1e-2)) { std::snprintf(buf, sizeof buf, "node %d t out of bounds", n); return buf; }
    if (!nd.missations.intervals.is_valid(tree.num_sites())) { std::snprintf(buf, sizeof buf, "node %d invalid interval set", n); return buf; }
    for (auto& [l, s] : nd.missations.from_states) {
      if (!nd.missations.contains(l)) { std::snprintf(buf, sizeof buf, "node %d from_state outside intervals", n); return buf; }
      if (s == tree.ref_sequence[l]) { std::snprintf(buf, sizeof buf, "node %d from_state equals ref", n); return buf; }
    }
  }
  for (auto s : seen) if (!s) return "unreachable node";
  // mutation + missation consistency by DFS with a running sequence (phylo_tree.cpp:18-111)
  std::vector<State> cur = tree.ref_sequence;
  std::vector<int> missing_depth(tree.num_sites(), 0);
  struct Frame { Node_index n; int k; };
  std::vector<Frame> st{{tree.root, 0}};
  while (!st.empty()) {
    auto& fr = st.back();
    auto& nd = tree.at(fr.n);
    if (fr.k == 0) {
      double min_t = fr.n != tree.root ? tree.at(nd.parent).t : k_neg_dbl_max;
      for (auto& [s, e] : nd.missations.intervals.v) for (int l = s; l < e; ++l) {
        if (missing_depth[l]) { std::snprintf(buf, sizeof buf, "node %d: site %d already missing upstream", fr.n, l); return buf; }
        if (nd.missations.get_from_state(l, tree.ref_sequence) != cur[l]) { std::snprintf(buf, sizeof buf, "node %d: missation from_state wrong at site %d", fr.n, l); return buf; }
        ++missing_depth[l];
      }
      for (auto& m : nd.mutations) {
        if (m.from == m.to || m.site < 0 || m.site >= tree.num_sites()) { std::snprintf(buf, sizeof buf, "node %d: bad mutation", fr.n); return buf; }
        if (!(min_t <= m.t && m.t <= nd.t)) { std::snprintf(buf, sizeof buf, "node %d: mutation time out of order/range (%g not in [%g,%g])", fr.n, m.t, min_t, nd.t); return buf; }
        min_t = m.t;
        if (m.from != cur[m.site]) { std::snprintf(buf, sizeof buf, "node %d: mutation from-state mismatch at site %d", fr.n, m.site); return buf; }
        if (missing_depth[m.site]) { std::snprintf(buf, sizeof buf, "node %d: mutation on missing site %d", fr.n, m.site); return buf; }
        cur[m.site] = m.to;
      }
    }
    if (nd.is_inner_node() && fr.k < 2) { Node_index c = nd.children[fr.k]; ++fr.k; st.push_back({c, 0}); continue; }
    if (nd.is_inner_node()) {
      if (interval_sets_intersect(tree.at(nd.children[0]).missations.intervals, tree.at(nd.children[1]).missations.intervals)) {
        std::snprintf(buf, sizeof buf, "node %d: missations on both children", fr.n); return buf;
      }
    }
    for (auto it = nd.mutations.rbegin(); it != nd.mutations.rend(); ++it) cur[it->site] = it->from;
    for (auto& [s, e] : nd.missations.intervals.v) for (int l = s; l < e; ++l) --missing_depth[l];
    st.pop_back();
  }
  return "";
}

}  // namespace orc
#endif  // ORC_CALC_HPP_
