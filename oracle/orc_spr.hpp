// orc_spr.hpp -- CPU ORACLE (test infrastructure, NOT product code).
// Restates reference core/spr_study.{h,cpp}, core/tree_editing.{h,cpp}, core/spr_move.{h,cpp}.
#ifndef ORC_SPR_HPP_
#define ORC_SPR_HPP_

#include "orc_calc.hpp"

namespace orc {

// =================================================================================================
// SPR study (reference core/spr_study.h:17-171, core/spr_study.cpp:9-549)
// =================================================================================================
struct Candidate_region {
  Branch_index branch; int mut_idx; double t_min; double t_max; int min_muts;
  double log_W_over_Wmax = 0.0, W_over_Wmax = 0.0;
  bool is_above_root() const { return t_min == k_neg_dbl_max; }
};

struct Spr_study_builder {
  const Phylo_tree* tree;
  Branch_index cur_branch = k_no_node;
  int cur_mut_idx = -1;
  int cur_muts_from_start = 0;
  Site_deltas cur_to_X_deltas;
  const Interval_set* missing_at_X;
  Node_index X = k_no_node;
  double t_X = std::numeric_limits<double>::max();
  int max_muts_from_start = std::numeric_limits<int>::max();
  struct Work_item { Branch_index target_branch; int target_mut_idx; bool is_backtracking; };
  std::vector<Work_item> work_stack;
  std::vector<Candidate_region> result;

  Spr_study_builder(const Phylo_tree& tree_, Node_index X_, double t_X_, const Interval_set& missing)
      : tree(&tree_), missing_at_X(&missing), X(X_), t_X(t_X_) {}

  bool cur_region_in_scope() const { return cur_branch != X && cur_muts_from_start <= max_muts_from_start; }
  double region_t_min(Branch_index b, int mi) const {   // spr_study.h:90-95
    if (b == tree->root) return k_neg_dbl_max;
    if (mi == 0) return tree->branch_begin_t(b);
    return tree->at(b).mutations.at(mi - 1).t;
  }
  double region_t_max(Branch_index b, int mi) const {   // spr_study.h:96-101
    if (b == tree->root) return tree->at(b).t;
    auto& muts = tree->at(b).mutations;
    if (mi == (int)muts.size()) return tree->branch_end_t(b);
    return muts.at(mi).t;
  }
  bool is_next_to_cur_region(Branch_index tb, int tmi) const {   // spr_study.h:136-151
    if (cur_branch == k_no_node || tb == k_no_node) return true;
    if (cur_branch == tb) return std::abs(tmi - cur_mut_idx) == 1;
    if (tb == tree->at(cur_branch).parent) return cur_mut_idx == 0 && tmi == (int)tree->at(tb).mutations.size();
    if (cur_branch == tree->at(tb).parent) return cur_mut_idx == (int)tree->at(cur_branch).mutations.size() && tmi == 0;
    return false;
  }
  void add_forward_movement(Branch_index tb, int tmi) {   // spr_study.h:153-158
    ORC_CHECK(is_next_to_cur_region(tb, tmi));
    work_stack.push_back({cur_branch, cur_mut_idx, true});
    work_stack.push_back({tb, tmi, false});
  }
  void move_to_neighbor(Branch_index tb, int tmi, bool backtracking) {   // spr_study.cpp:43-91
    ORC_CHECK(is_next_to_cur_region(tb, tmi));
    if (cur_branch != k_no_node && tb == cur_branch) {
      auto& muts = tree->at(cur_branch).mutations;
      if (tmi == cur_mut_idx + 1) {
        auto& m = muts.at(cur_mut_idx);
        if (!missing_at_X->contains(m.site)) { pop_front_site_deltas(m, cur_to_X_deltas); cur_muts_from_start += backtracking ? -1 : +1; }
      } else if (tmi == cur_mut_idx - 1) {
        auto& m = muts.at(tmi);
        if (!missing_at_X->contains(m.site)) { push_front_site_deltas(m, cur_to_X_deltas); cur_muts_from_start += backtracking ? -1 : +1; }
      } else ORC_CHECK(false);
    }
    cur_branch = tb; cur_mut_idx = tmi;
  }
  void visit_cur_region() {   // spr_study.cpp:93-101
    result.push_back(Candidate_region{cur_branch, cur_mut_idx, region_t_min(cur_branch, cur_mut_idx),
                                      region_t_max(cur_branch, cur_mut_idx), (int)cur_to_X_deltas.size()});
  }
  void seed_neighbors_except(Branch_index ob, int omi) {   // spr_study.cpp:103-128
    auto maybe = [&](int nb, int nmi) { if (nb == ob && nmi == omi) return; add_forward_movement(nb, nmi); };
    if (cur_branch != tree->root) {
      if (cur_mut_idx > 0) maybe(cur_branch, cur_mut_idx - 1);
      else { auto pb = tree->at(cur_branch).parent; maybe(pb, (int)tree->at(pb).mutations.size()); }
    }
    if (cur_mut_idx < (int)tree->at(cur_branch).mutations.size()) maybe(cur_branch, cur_mut_idx + 1);
    else if (tree->at(cur_branch).is_inner_node()) for (int k = 0; k < 2; ++k) maybe(tree->at(cur_branch).children[k], 0);
  }
  void do_pending_work() {   // spr_study.cpp:26-41
    while (!work_stack.empty()) {
      auto [tb, tmi, bt] = work_stack.back(); work_stack.pop_back();
      auto ob = cur_branch; auto omi = cur_mut_idx;
      move_to_neighbor(tb, tmi, bt);
      if (!bt && cur_region_in_scope()) { visit_cur_region(); seed_neighbors_except(ob, omi); }
    }
  }
  void account_for_Xs_detachment(bool can_change_root) {   // spr_study.cpp:130-209
    auto erase_marked = [&]() { result.erase(std::remove_if(result.begin(), result.end(), [](auto& r) { return r.branch == -1; }), result.end()); };
    if (X == k_no_node) {
      if (!can_change_root) { for (auto& r : result) if (r.branch == tree->root) r.branch = -1; erase_marked(); }
      return;
    }
    ORC_CHECK(X != tree->root);
    auto P = tree->at(X).parent; auto S = tree->at(P).sibling_of(X);
    int nGP = (int)tree->at(P).mutations.size();
    for (auto& region : result) {
      if (!can_change_root) {
        if (region.branch == tree->root) { ORC_CHECK(region.is_above_root()); region.branch = -1; continue; }
        else ORC_CHECK(!region.is_above_root());
      }
      if (region.branch != S && region.branch != P) continue;
      if (P != tree->root) {
        if (region.branch == S) {
          if (region.mut_idx == 0) region.t_min = region_t_min(P, nGP);
          region.mut_idx += nGP;
        } else if (region.branch == P) {
          if (region.mut_idx == nGP) region.branch = -1; else region.branch = S;
        }
      } else {
        if (!can_change_root) { if (region.branch == P) region.branch = -1; }
        else {
          if (region.branch == S && region.mut_idx == (int)tree->at(S).mutations.size()) { region.mut_idx += nGP; region.t_min = k_neg_dbl_max; }
          else region.branch = -1;
        }
      }
    }
    erase_marked();
  }
  void remove_regions_in_Xs_future() {   // spr_study.cpp:211-224
    for (auto& r : result) { if (r.t_min >= t_X) r.branch = -1; else if (r.t_max > t_X) r.t_max = t_X; }
    result.erase(std::remove_if(result.begin(), result.end(), [](auto& r) { return r.branch == -1; }), result.end());
  }
  void seed_fill_from(Branch_index ib, int imi, Site_deltas init, bool can_change_root) {   // spr_study.cpp:9-24
    ORC_CHECK(work_stack.empty()); ORC_CHECK(cur_branch == k_no_node);
    cur_to_X_deltas = std::move(init);
    add_forward_movement(ib, imi);
    do_pending_work();
    account_for_Xs_detachment(can_change_root);
    remove_regions_in_Xs_future();
  }
};

struct Spr_study {
  const Phylo_tree* tree;
  double lambda_X, mu, annealing_factor, t_X, t_max_tip;
  std::vector<Candidate_region> candidate_regions;
  double log_Wmax = 0.0, sum_W_over_Wmax = 0.0;

  struct Root_region_params { double f, t_S, s_min, s_max, x_min, x_max; int m; };
  Root_region_params root_params(const Candidate_region& region) const {
    Root_region_params p;
    p.f = annealing_factor; p.m = region.min_muts;
    p.t_S = tree->at(region.branch).t;
    p.s_min = std::abs(t_X - p.t_S);
    double t_early = std::min(t_X, p.t_S), tree_span = t_max_tip - t_early;
    ORC_CHECK(tree_span >= 0.0);
    p.s_max = p.s_min + 20.0 * tree_span;
    p.x_min = lambda_X * p.f * p.s_min; p.x_max = lambda_X * p.f * p.s_max;
    return p;
  }
  Spr_study(Spr_study_builder&& b, double lambda_X_, double f_, double t_X_, double t_max_tip_)   // spr_study.cpp:226-385
      : tree(b.tree), lambda_X(lambda_X_), annealing_factor(f_), t_X(t_X_), t_max_tip(t_max_tip_), candidate_regions(std::move(b.result)) {
    mu = lambda_X / (tree->num_sites() - b.missing_at_X->num_sites());
    const double f = annealing_factor;
    for (auto& region : candidate_regions) {
      double t_min = region.t_min, t_max = region.t_max; int m = region.min_muts;
      if (!region.is_above_root()) {
        double t_prime = 0.5 * (t_min + t_max);
        region.log_W_over_Wmax = std::log(f * lambda_X * (t_max - t_min)) +
            f * (-lambda_X * (t_X - t_prime) + m * std::log(mu * (t_X - t_prime) / 3));
      } else {
        auto p = root_params(region);
        if (p.x_max < 0.01) {
          double alpha = f * m + 1;
          region.log_W_over_Wmax = -M_LN2 + std::log(f * lambda_X) + f * m * std::log(mu / 3)
              + alpha * std::log(p.s_max) + std::log1p(-std::pow(p.s_min / p.s_max, alpha)) - std::log(alpha);
        } else {
          region.log_W_over_Wmax = -M_LN2 + f * m * std::log(mu / (3 * lambda_X * f)) + std::lgamma(f * m + 1)
              + safe_log_gamma_integral(f * m + 1, p.x_min, p.x_max);
        }
      }
    }
    ORC_CHECK(!candidate_regions.empty());
    log_Wmax = candidate_regions[0].log_W_over_Wmax;
    for (auto& r : candidate_regions) log_Wmax = std::max(log_Wmax, r.log_W_over_Wmax);
    sum_W_over_Wmax = 0.0;
    for (auto& r : candidate_regions) { r.log_W_over_Wmax -= log_Wmax; r.W_over_Wmax = std::exp(r.log_W_over_Wmax); sum_W_over_Wmax += r.W_over_Wmax; }
  }
  int pick_nexus_region(Rng& rng) const {   // spr_study.cpp:404-422
    double r = rng.uniform_co(0.0, sum_W_over_Wmax);
    for (int i = 0; i < (int)candidate_regions.size(); ++i) {
      if (candidate_regions[i].W_over_Wmax >= r) return i;
      r -= candidate_regions[i].W_over_Wmax;
    }
    return 0;
  }
  double pick_time_in_region(int idx, Rng& rng) const {   // spr_study.cpp:424-471
    auto& region = candidate_regions[idx];
    if (!region.is_above_root()) return rng.uniform_oc(region.t_min, region.t_max);
    auto p = root_params(region);
    double rand_s;
    if (p.x_max < 0.01) {
      double alpha = p.f * p.m + 1;
      double U = rng.u01_oo();
      double smin_a = std::pow(p.s_min, alpha), smax_a = std::pow(p.s_max, alpha);
      rand_s = std::pow(smin_a + U * (smax_a - smin_a), 1.0 / alpha);
    } else {
      rand_s = safe_sample_truncated_gamma(p.f * p.m + 1, lambda_X * p.f, p.s_min, p.s_max, rng);
    }
    double rand_t = 0.5 * (t_X + p.t_S - rand_s);
    return std::max(region.t_min, std::min(region.t_max, rand_t));
  }
  int find_region(Branch_index b, double t) const {   // spr_study.cpp:474-484
    for (int i = 0; i < (int)candidate_regions.size(); ++i) {
      auto& r = candidate_regions[i];
      if (r.branch == b && r.t_min < t && t <= r.t_max) return i;
    }
    return -1;
  }
  double log_alpha_in_region(int idx, double t) const {   // spr_study.cpp:486-549
    auto& region = candidate_regions[idx];
    double log_p_region = region.log_W_over_Wmax - std::log(sum_W_over_Wmax);
    if (!region.is_above_root()) return log_p_region - std::log(region.t_max - region.t_min);
    auto p = root_params(region);
    double s = t_X - t + p.t_S - t;
    if (s > p.s_max + 1e-6) return -std::numeric_limits<double>::infinity();
    if (p.x_max < 0.01) {
      double alpha = p.f * p.m + 1;
      return log_p_region + M_LN2 + std::log(alpha) + (alpha - 1) * std::log(s) + -alpha * std::log(p.s_max)
          + -std::log1p(-std::pow(p.s_min / p.s_max, alpha));
    }
    return log_p_region + M_LN2 + std::log(lambda_X * p.f) + p.f * p.m * std::log(lambda_X * p.f * s) + -lambda_X * p.f * s
        + -std::lgamma(p.f * p.m + 1) - safe_log_gamma_integral(p.f * p.m + 1, p.x_min, p.x_max);
  }
};

// =================================================================================================
// Tree editing (reference core/tree_editing.cpp:7-302)
// =================================================================================================
struct Tree_editing_session {
  Phylo_tree* tree; Node_index X; const Global_evo_model* evo;
  std::vector<double>* lambda_i; const std::vector<double>* cumQ; std::vector<int>* num_missing;
  Site_deltas deltas_nexus_to_X;

  Tree_editing_session(Phylo_tree& t, Node_index X_, const Global_evo_model& e, std::vector<double>& li,
                       const std::vector<double>& cq, std::vector<int>& nm)
      : tree(&t), X(X_), evo(&e), lambda_i(&li), cumQ(&cq), num_missing(&nm) {
    ORC_CHECK(X != tree->root);
    for (auto& m : tree->at(X).mutations) push_back_site_deltas(m, deltas_nexus_to_X);
    tree->at(X).mutations.clear();
  }
  double rate(const Mutation& m, State s) const { return evo->mu_l(m.site) * evo->nu_l.at(m.site) * evo->q_l_a(m.site, s); }
  double dlam(const Mutation& m, State minus, State plus) const {
    return evo->mu_l(m.site) * evo->nu_l.at(m.site) * (-evo->q_l_a(m.site, minus) + evo->q_l_a(m.site, plus));
  }
  void slide_P_along_branch(double new_t_P) {   // tree_editing.cpp:31-112
    auto P = tree->at(X).parent;
    ORC_CHECK(!tree->at(P).is_tip());
    if (P == tree->root) return slide_root(new_t_P);
    double old_t_P = tree->at(P).t;
    auto S = tree->at(P).sibling_of(X);
    auto& mP = tree->at(P).mutations; auto& mS = tree->at(S).mutations;
    if (new_t_P < old_t_P) {
      auto first = std::find_if(mP.begin(), mP.end(), [&](const Mutation& m) { return m.t >= new_t_P; });
      if (first != mP.end()) {
        std::reverse(mS.begin(), mS.end());
        for (auto it = mP.end(); it != first;) {
          --it; const Mutation m = *it;
          if (!tree->at(S).missations.contains(m.site)) mS.push_back(m);
          else tree->at(S).missations.set_from_state(m.site, m.from, tree->ref_sequence);
          if (!tree->at(X).missations.contains(m.site)) push_front_site_deltas(m, deltas_nexus_to_X);
          else tree->at(X).missations.set_from_state(m.site, m.from, tree->ref_sequence);
          lambda_i->at(P) += dlam(m, m.to, m.from);
        }
        std::reverse(mS.begin(), mS.end());
        mP.erase(first, mP.end());
      }
    } else {
      auto last = std::find_if(mS.begin(), mS.end(), [&](const Mutation& m) { return m.t > new_t_P; });
      if (last != mS.begin()) {
        for (auto it = mS.begin(); it != last; ++it) {
          const Mutation m = *it;
          mP.push_back(m);
          if (!tree->at(X).missations.contains(m.site)) push_front_site_deltas({m.site, m.to, m.from}, deltas_nexus_to_X);
          else tree->at(X).missations.set_from_state(m.site, m.to, tree->ref_sequence);
          lambda_i->at(P) += dlam(m, m.from, m.to);
        }
        mS.erase(mS.begin(), last);
      }
    }
    tree->at(P).t = new_t_P;
  }
  void slide_root(double new_t_P) {   // tree_editing.cpp:114-158
    auto P = tree->at(X).parent;
    ORC_CHECK(P == tree->root);
    double old_t_P = tree->at(P).t;
    auto S = tree->at(P).sibling_of(X);
    auto& mP = tree->at(P).mutations; auto& mS = tree->at(S).mutations;
    if (new_t_P > old_t_P) {
      auto last = std::find_if(mS.begin(), mS.end(), [&](const Mutation& m) { return m.t > new_t_P; });
      if (last != mS.begin()) {
        Site_deltas ref_to_root;
        for (auto& m : mP) push_back_site_deltas(m, ref_to_root);
        for (auto it = mS.begin(); it != last; ++it) {
          const Mutation m = *it;
          push_back_site_deltas(m, ref_to_root);
          if (!tree->at(X).missations.contains(m.site)) push_front_site_deltas({m.site, m.to, m.from}, deltas_nexus_to_X);
          else tree->at(X).missations.set_from_state(m.site, m.to, tree->ref_sequence);
          lambda_i->at(P) += dlam(m, m.from, m.to);
        }
        mP.clear();
        for (auto& [l, d] : ref_to_root) mP.push_back(Mutation{d.from, l, d.to, k_neg_dbl_max});
        mS.erase(mS.begin(), last);
      }
    }
    tree->at(P).t = new_t_P;
  }
  void hop_up() { do_hop_up(X); }
  void do_hop_up(Node_index X_) {   // tree_editing.cpp:164-231 (X_ shadows this->X on purpose)
    ORC_CHECK(X_ != tree->root);
    auto P = tree->at(X_).parent;
    ORC_CHECK(!tree->at(P).is_tip()); ORC_CHECK(P != tree->root); ORC_CHECK(tree->at(P).mutations.empty());
    auto G = tree->at(P).parent;
    ORC_CHECK(tree->at(P).t == tree->at(G).t);
    auto U = tree->at(G).sibling_of(P);
    auto S = tree->at(P).sibling_of(X_);
    if (!tree->at(P).missations.empty()) {
      tree->at(X_).missations = merge_missations_nondestructively(tree->at(X_).missations, tree->at(P).missations);
      tree->at(S).missations = merge_missations_nondestructively(tree->at(S).missations, tree->at(P).missations);
      tree->at(P).missations.clear();
    }
    std::swap(tree->at(P).mutations, tree->at(G).mutations);
    std::swap(tree->at(P).missations, tree->at(G).missations);
    ORC_CHECK(tree->at(G).missations.empty());
    if (interval_sets_intersect(tree->at(S).missations.intervals, tree->at(U).missations.intervals))
      factor_out_common_missations(tree->at(S).missations, tree->at(U).missations, tree->at(G).missations);
    if (G == tree->root) { tree->root = P; tree->at(P).parent = k_no_node; }
    else {
      auto GG = tree->at(G).parent; auto GU = tree->at(GG).sibling_of(G);
      tree->at(GG).children[0] = P; tree->at(GG).children[1] = GU;
      tree->at(P).parent = GG;
      ORC_CHECK(tree->at(GU).parent == GG);
    }
    tree->at(P).children[0] = X_; tree->at(P).children[1] = G;
    ORC_CHECK(tree->at(X_).parent == P);
    tree->at(G).parent = P;
    tree->at(G).children[0] = S; tree->at(G).children[1] = U;
    tree->at(S).parent = G;
    ORC_CHECK(tree->at(U).parent == G);
    lambda_i->at(P) = lambda_i->at(G);
    num_missing->at(P) = num_missing->at(G);
    lambda_i->at(G) = lambda_i->at(P) + calc_delta_lambda_across_missations(*evo, tree->ref_sequence, *cumQ, tree->at(G).missations);
    num_missing->at(G) = num_missing->at(P) + tree->at(G).missations.num_sites();
  }
  void flip() {   // tree_editing.cpp:233-278
    ORC_CHECK(X != tree->root);
    auto P = tree->at(X).parent;
    ORC_CHECK(!tree->at(P).is_tip()); ORC_CHECK(P != tree->root); ORC_CHECK(tree->at(P).mutations.empty());
    auto G = tree->at(P).parent;
    ORC_CHECK(tree->at(P).t == tree->at(G).t);
    auto U = tree->at(G).sibling_of(P);
    auto S = tree->at(P).sibling_of(X);
    if (!tree->at(P).missations.empty()) {
      tree->at(S).missations = merge_missations_nondestructively(tree->at(S).missations, tree->at(P).missations);
      tree->at(X).missations = merge_missations_nondestructively(tree->at(X).missations, tree->at(P).missations);
      tree->at(P).missations.clear();
    }
    if (interval_sets_intersect(tree->at(X).missations.intervals, tree->at(U).missations.intervals))
      factor_out_common_missations(tree->at(X).missations, tree->at(U).missations, tree->at(P).missations);
    tree->at(G).children[0] = S; tree->at(G).children[1] = P;
    tree->at(S).parent = G;
    ORC_CHECK(tree->at(P).parent == G);
    tree->at(P).children[0] = X; tree->at(P).children[1] = U;
    ORC_CHECK(tree->at(X).parent == P);
    tree->at(U).parent = P;
    lambda_i->at(P) = lambda_i->at(G) + calc_delta_lambda_across_missations(*evo, tree->ref_sequence, *cumQ, tree->at(P).missations);
    num_missing->at(P) = num_missing->at(G) + tree->at(P).missations.num_sites();
  }
  void hop_down(Node_index SS) {   // tree_editing.cpp:280-292
    ORC_CHECK(X != tree->root);
    auto P = tree->at(X).parent;
    ORC_CHECK(SS != tree->root);
    auto U = tree->at(SS).parent;
    ORC_CHECK(tree->at(U).parent == P);
    auto UU = tree->at(U).sibling_of(SS);
    ORC_CHECK(tree->at(U).mutations.empty());
    do_hop_up(UU);
  }
  void end() {   // tree_editing.cpp:294-302
    ORC_CHECK(tree->at(X).mutations.empty());
    if (!deltas_nexus_to_X.empty()) {
      double mut_t = 0.5 * (tree->at(X).t + tree->at_parent_of(X).t);
      for (auto& [l, d] : deltas_nexus_to_X) tree->at(X).mutations.push_back(Mutation{d.from, l, d.to, mut_t});
    }
  }
};

// =================================================================================================
// Mutational-history sampling (reference core/spr_move.cpp:1158-1439)
// =================================================================================================
inline State choose_different_state(State s, Rng& rng) {   // :1158-1162
  int delta = 1 + rng.uniform_int(k_num_states - 1);
  return (State)((s + delta) % k_num_states);
}
inline Mutation_list sample_mutational_history(Site_index L, double T, double mu, const Site_deltas& deltas, Rng& rng) {   // :1164-1370
  Mutation_list result;
  std::vector<State> to_states; std::vector<double> mut_times;
  if (!deltas.empty()) {
    K_truncated_poisson_distribution num_muts_ge1(mu * T, 1);
    for (auto& [l, delta] : deltas) {
      int n = 0;
      while (true) {
        n = num_muts_ge1(rng);
        to_states.clear();
        State s = delta.from;
        for (int i = 0; i < n; ++i) { s = choose_different_state(s, rng); to_states.push_back(s); }
        if (s == delta.to) break;
      }
      mut_times.clear();
      for (int i = 0; i < n; ++i) mut_times.push_back(rng.uniform_co(-T, 0.0));
      std::sort(mut_times.begin(), mut_times.end());
      State prev = delta.from;
      for (int i = 0; i < n; ++i) { ORC_CHECK(0 <= l && l < L); result.push_back(Mutation{prev, l, to_states[i], mut_times[i]}); prev = to_states[i]; }
    }
  }
  double muT = mu * T;
  double p_0 = std::exp(-muT), p_1 = muT * p_0;
  double log_one_minus_p_tricky = (muT < 1e-4) ? -0.5 * muT * muT : -muT - std::log1p(-p_1);
  int l = 0;
  if (L * muT * muT < 2e-6) l = L;
  while (l < L) {
    double u = rng.exponential(-log_one_minus_p_tricky);
    if (!(u >= 0 && u < L)) break;
    l += (int)std::floor(u);
    if (l >= L) break;
    if (deltas.count(l)) { ++l; continue; }
    int n = K_truncated_poisson_distribution(mu * T, 2)(rng);
    to_states.clear();
    State s = sA;
    for (int i = 0; i < n; ++i) { s = choose_different_state(s, rng); to_states.push_back(s); }
    if (s == sA) {
      mut_times.clear();
      for (int i = 0; i < n; ++i) mut_times.push_back(rng.uniform_co(-T, 0.0));
      std::sort(mut_times.begin(), mut_times.end());
      State prev = sA;
      for (int i = 0; i < n; ++i) { result.push_back(Mutation{prev, l, to_states[i], mut_times[i]}); prev = to_states[i]; }
      ++l;
    }
  }
  sort_mutations(result);
  return result;
}
inline Mutation_list sample_unconstrained_mutational_history(Site_index L, double T, double mu, Rng& rng) {   // :1372-1407
  std::map<Site_index, State> cur_state;
  Mutation_list traj;
  double t = 0.0;
  while (true) {
    t -= rng.exponential(mu * L);
    if (t <= -T) break;
    int l = rng.uniform_int(L);
    auto [it, ins] = cur_state.try_emplace(l, sA);
    State s = it->second;
    State ns = choose_different_state(s, rng);
    traj.push_back(Mutation{ns, l, s, t});
    it->second = ns;
  }
  std::reverse(traj.begin(), traj.end());
  return traj;
}
inline void adjust_mutational_history(Mutation_list& history, const Site_deltas& site_deltas, const Phylo_tree& tree, Phylo_tree_loc end_loc) {   // :1409-1439
  std::map<Site_index, State> end_states;
  for (auto it = history.rbegin(); it != history.rend(); ++it) {
    auto& m = *it;
    m.t += end_loc.t;
    if (!site_deltas.count(m.site)) {
      State end_state;
      auto [e, ins] = end_states.try_emplace(m.site, sA);
      if (!ins) end_state = e->second; else end_state = e->second = calc_site_state_at(tree, end_loc, m.site);
      int delta = end_state - sA;
      m.from = (State)((m.from + delta) % k_num_states);
      m.to = (State)((m.to + delta) % k_num_states);
    }
  }
}

// =================================================================================================
// SPR graft analysis and application (reference core/spr_move.h:28-150, core/spr_move.cpp:9-1156)
// =================================================================================================
struct Spr_graft {
  Node_index X, S; double t_P;
  enum { k_branch_info_P_X = 0, k_branch_info_P_S = 1, k_branch_info_S_P_X = 2 };
  struct Branch_info {
    Node_index A = k_no_node, B = k_no_node; bool is_open = false; double T_to_X = 0.0;
    double partial_lambda_at_A = 0.0, partial_lambda_at_X = 0.0;
    Interval_set warm_sites, hot_sites;
    Mutation_list hot_muts_to_X;
    Site_deltas hot_deltas_to_X;
  };
  std::vector<Branch_info> branch_infos;
  double delta_log_G = 0.0, log_alpha_mut = 0.0;
};

struct Spr_move {
  Phylo_tree* tree; double mu_proposal; bool can_change_root; const Global_evo_model* evo;
  std::vector<double>* lambda_i; const std::vector<double>* cumQ; std::vector<int>* num_missing;

  Spr_move(Phylo_tree& t, double mu_p, bool ccr, const Global_evo_model& e, std::vector<double>& li,
           const std::vector<double>& cq, std::vector<int>& nm)
      : tree(&t), mu_proposal(mu_p), can_change_root(ccr), evo(&e), lambda_i(&li), cumQ(&cq), num_missing(&nm) {}

  bool rooty(Node_index X) const { return tree->at(X).parent == tree->root; }
  double dq(const Mutation& m, State minus, State plus) const {
    return evo->mu_l(m.site) * evo->nu_l.at(m.site) * (-evo->q_l_a(m.site, minus) + evo->q_l_a(m.site, plus));
  }
  Spr_graft analyze_graft(Node_index X) const { auto g = start_graft_analysis(X); finish_graft_analysis(g); return g; }
  Spr_graft propose_new_graft(Node_index X, Rng& rng) const {
    auto g = start_graft_analysis(X); propose_new_graft_mutations(g, rng); finish_graft_analysis(g); return g;
  }
  Spr_graft start_graft_analysis(Node_index X) const { return rooty(X) ? start_rooty_graft_analysis(X) : start_inner_graft_analysis(X); }
  void propose_new_graft_mutations(Spr_graft& g, Rng& rng) const { if (rooty(g.X)) propose_new_rooty_graft_mutations(g, rng); else propose_new_inner_graft_mutations(g, rng); }
  void finish_graft_analysis(Spr_graft& g) const { if (rooty(g.X)) finish_rooty_graft_analysis(g); else finish_inner_graft_analysis(g); }
  void peel_graft(const Spr_graft& g) { ORC_CHECK(g.X != tree->root); if (rooty(g.X)) peel_rooty_graft(g); else peel_inner_graft(g); }
  void apply_graft(const Spr_graft& g) { ORC_CHECK(g.X != tree->root); if (rooty(g.X)) apply_rooty_graft(g); else apply_inner_graft(g); }
  int count_min_mutations(const Spr_graft& g) const {
    if (rooty(g.X)) return (int)g.branch_infos[Spr_graft::k_branch_info_S_P_X].hot_deltas_to_X.size();
    int r = 0; for (auto& bi : g.branch_infos) if (!bi.is_open) r += (int)bi.hot_deltas_to_X.size(); return r;
  }
  int count_closed_mutations(const Spr_graft& g) const {
    if (rooty(g.X)) return (int)g.branch_infos[Spr_graft::k_branch_info_S_P_X].hot_muts_to_X.size();
    int r = 0; for (auto& bi : g.branch_infos) if (!bi.is_open) r += (int)bi.hot_muts_to_X.size(); return r;
  }
  Site_deltas summarize_closed_mutations(const Spr_graft& g) const {
    if (rooty(g.X)) return g.branch_infos[Spr_graft::k_branch_info_S_P_X].hot_deltas_to_X;
    Site_deltas r; for (auto& bi : g.branch_infos) if (!bi.is_open) append_site_deltas(r, bi.hot_deltas_to_X); return r;
  }

  // ---- rooty grafts (spr_move.cpp:91-547) ----
  Spr_graft start_rooty_graft_analysis(Node_index X) const {   // :91-205
    ORC_CHECK(X != tree->root);
    double t_X = tree->at(X).t; auto P = tree->at(X).parent;
    ORC_CHECK(P == tree->root);
    double t_P = tree->at(P).t; auto S = tree->at(P).sibling_of(X); double t_S = tree->at(S).t;
    ORC_CHECK(can_change_root);
    auto& miss_P = tree->at(P).missations; auto& miss_X = tree->at(X).missations; auto& miss_S = tree->at(S).missations;
    auto& muts_X = tree->at(X).mutations; auto& muts_S = tree->at(S).mutations;
    Spr_graft g; g.X = X; g.S = S; g.t_P = t_P;
    g.branch_infos.resize(3);
    auto& PX = g.branch_infos[Spr_graft::k_branch_info_P_X];
    PX.A = P; PX.B = X; PX.is_open = true; PX.T_to_X = t_X - t_P;
    PX.partial_lambda_at_A = -1 * calc_delta_lambda_across_missations(*evo, tree->ref_sequence, *cumQ, miss_S);
    PX.warm_sites = miss_S.intervals; PX.hot_sites = PX.warm_sites;
    PX.partial_lambda_at_X = PX.partial_lambda_at_A;
    for (auto& m : muts_X) if (PX.hot_sites.contains(m.site)) { PX.hot_muts_to_X.push_back(m); PX.partial_lambda_at_X += dq(m, m.from, m.to); }
    auto& PS = g.branch_infos[Spr_graft::k_branch_info_P_S];
    PS.A = P; PS.B = S; PS.is_open = true; PS.T_to_X = t_S - t_P;
    PS.partial_lambda_at_A = -1 * calc_delta_lambda_across_missations(*evo, tree->ref_sequence, *cumQ, miss_X);
    PS.warm_sites = miss_X.intervals; PS.hot_sites = PS.warm_sites;
    PS.partial_lambda_at_X = PS.partial_lambda_at_A;
    for (auto& m : muts_S) if (PS.hot_sites.contains(m.site)) { PS.hot_muts_to_X.push_back(m); PS.partial_lambda_at_X += dq(m, m.from, m.to); }
    auto& SPX = g.branch_infos[Spr_graft::k_branch_info_S_P_X];
    SPX.A = S; SPX.B = P; SPX.is_open = false; SPX.T_to_X = (t_S - t_P) + (t_X - t_P);
    SPX.partial_lambda_at_X = lambda_i->at(X) - PX.partial_lambda_at_X;
    SPX.partial_lambda_at_A = lambda_i->at(S) - PS.partial_lambda_at_X;
    SPX.hot_sites.insert({0, tree->num_sites()});
    subtract_interval_sets(SPX.warm_sites, SPX.hot_sites, miss_P.intervals);
    subtract_interval_sets(SPX.hot_sites, SPX.warm_sites, miss_X.intervals);
    subtract_interval_sets(SPX.warm_sites, SPX.hot_sites, miss_S.intervals);
    SPX.hot_sites = SPX.warm_sites;
    for (auto it = muts_S.rbegin(); it != muts_S.rend(); ++it) {
      auto& m = *it;
      if (SPX.hot_sites.contains(m.site)) {
        Mutation rm{m.to, m.site, m.from, t_P - (m.t - t_P)};
        SPX.hot_muts_to_X.push_back(rm); push_back_site_deltas(rm, SPX.hot_deltas_to_X);
      }
    }
    for (auto& m : muts_X) if (SPX.hot_sites.contains(m.site)) { SPX.hot_muts_to_X.push_back(m); push_back_site_deltas(m, SPX.hot_deltas_to_X); }
    return g;
  }
  void propose_new_rooty_graft_mutations(Spr_graft& g, Rng& rng) const {   // :207-244
    auto X = g.X; auto P = tree->at(X).parent; auto S = tree->at(P).sibling_of(X);
    for (int idx = 0; idx < (int)g.branch_infos.size(); ++idx) {
      auto& bi = g.branch_infos[idx];
      ORC_CHECK(!bi.is_open || bi.hot_deltas_to_X.empty());
      if (!bi.hot_sites.empty()) {
        auto nm = bi.is_open ? sample_unconstrained_mutational_history(tree->num_sites(), bi.T_to_X, mu_proposal, rng)
                             : sample_mutational_history(tree->num_sites(), bi.T_to_X, mu_proposal, bi.hot_deltas_to_X, rng);
        if (!nm.empty()) {
          nm.erase(std::remove_if(nm.begin(), nm.end(), [&](const Mutation& m) { return !bi.hot_sites.contains(m.site); }), nm.end());
          auto path_end = (idx == Spr_graft::k_branch_info_P_S) ? tree->node_loc(S) : tree->node_loc(X);
          adjust_mutational_history(nm, bi.hot_deltas_to_X, *tree, path_end);
        }
        bi.hot_muts_to_X = std::move(nm);
        if (bi.is_open) {
          bi.partial_lambda_at_A = bi.partial_lambda_at_X;
          for (auto it = bi.hot_muts_to_X.rbegin(); it != bi.hot_muts_to_X.rend(); ++it) bi.partial_lambda_at_A += dq(*it, it->to, it->from);
        }
      }
    }
  }
  void finish_rooty_graft_analysis(Spr_graft& g) const {   // :246-316
    auto X = g.X; double t_X = tree->at(X).t; auto P = tree->at(X).parent; double t_P = tree->at(P).t;
    auto S = tree->at(P).sibling_of(X); double t_S = tree->at(S).t;
    auto& PX = g.branch_infos[Spr_graft::k_branch_info_P_X];
    auto& PS = g.branch_infos[Spr_graft::k_branch_info_P_S];
    auto& SPX = g.branch_infos[Spr_graft::k_branch_info_S_P_X];
    g.delta_log_G = 0.0;
    g.delta_log_G += calc_branch_log_G(t_P, t_X, PX.partial_lambda_at_X, *evo, PX.hot_muts_to_X);
    g.delta_log_G += calc_branch_log_G(t_P, t_S, PS.partial_lambda_at_X, *evo, PS.hot_muts_to_X);
    Mutation_list on_PS, on_PX;
    for (auto it = SPX.hot_muts_to_X.rbegin(); it != SPX.hot_muts_to_X.rend(); ++it)
      if (it->t < t_P) on_PS.push_back(Mutation{it->to, it->site, it->from, t_P + (t_P - it->t)});
    for (auto& m : SPX.hot_muts_to_X) if (m.t >= t_P) on_PX.push_back(m);
    g.delta_log_G += calc_branch_log_G(t_P, t_X, SPX.partial_lambda_at_X, *evo, on_PX);
    g.delta_log_G += calc_branch_log_G(t_P, t_S, SPX.partial_lambda_at_A, *evo, on_PS);
    for (auto& m : PX.hot_muts_to_X) g.delta_log_G += std::log(evo->pi_l_a(m.site, m.from) / evo->pi_l_a(m.site, m.to));
    for (auto& m : PS.hot_muts_to_X) g.delta_log_G += std::log(evo->pi_l_a(m.site, m.from) / evo->pi_l_a(m.site, m.to));
    for (auto& m : on_PS) g.delta_log_G += std::log(evo->pi_l_a(m.site, m.from) / evo->pi_l_a(m.site, m.to));
    g.log_alpha_mut = 0.0;
    for (auto& bi : g.branch_infos) {
      int L = bi.hot_sites.num_sites(); double T = bi.T_to_X; int M = (int)bi.hot_muts_to_X.size();
      g.log_alpha_mut += -mu_proposal * L * T + M * std::log(mu_proposal / 3);
      if (!bi.is_open) {
        int d = (int)bi.hot_deltas_to_X.size();
        double P_AC = -0.25 * std::expm1(-4. / 3. * mu_proposal * T);
        g.log_alpha_mut -= (L - d) * std::log1p(-3 * P_AC) + d * std::log(P_AC);
      }
    }
  }
  void peel_rooty_graft(const Spr_graft& g) {   // :318-431
    ORC_CHECK(can_change_root);
    auto X = g.X; double t_X = tree->at(X).t; auto P = tree->at(X).parent;
    ORC_CHECK(P == tree->root);
    double t_P = tree->at(P).t; auto S = tree->at(P).sibling_of(X);
    auto& muts_P = tree->at(P).mutations; auto& muts_X = tree->at(X).mutations; auto& muts_S = tree->at(S).mutations;
    auto& miss_X = tree->at(X).missations; auto& miss_S = tree->at(S).missations;
    auto& PX = g.branch_infos[Spr_graft::k_branch_info_P_X];
    auto& PS = g.branch_infos[Spr_graft::k_branch_info_P_S];
    auto& SPX = g.branch_infos[Spr_graft::k_branch_info_S_P_X];
    Site_deltas ref_to_root;
    for (auto& m : muts_P) push_back_site_deltas(m, ref_to_root);
    for (auto& m : muts_X) if (PX.hot_sites.contains(m.site)) { push_back_site_deltas(m, ref_to_root); miss_S.set_from_state(m.site, m.to, tree->ref_sequence); }
    for (auto& m : muts_S) if (PS.hot_sites.contains(m.site)) { push_back_site_deltas(m, ref_to_root); miss_X.set_from_state(m.site, m.to, tree->ref_sequence); }
    for (auto& m : muts_S) if (SPX.hot_sites.contains(m.site)) push_back_site_deltas(m, ref_to_root);
    muts_X.clear(); muts_S.clear(); muts_P.clear();
    double t_mut_X = 0.5 * (t_P + t_X);
    for (auto& [l, d] : SPX.hot_deltas_to_X) muts_X.push_back(Mutation{d.from, l, d.to, t_mut_X});
    for (auto& [l, d] : ref_to_root) { ORC_CHECK(tree->ref_sequence[l] == d.from); muts_P.push_back(Mutation{d.from, l, d.to, k_neg_dbl_max}); }
    lambda_i->at(P) = calc_lambda_at_node(*tree, P, *evo, *cumQ);
  }
  void apply_rooty_graft(const Spr_graft& g) {   // :433-547
    ORC_CHECK(can_change_root);
    auto X = g.X; double t_X = tree->at(X).t; auto P = tree->at(X).parent;
    ORC_CHECK(P == tree->root);
    double t_P = tree->at(P).t; auto S = tree->at(P).sibling_of(X); double t_S = tree->at(S).t;
    auto& muts_P = tree->at(P).mutations; auto& muts_X = tree->at(X).mutations; auto& muts_S = tree->at(S).mutations;
    auto& miss_X = tree->at(X).missations; auto& miss_S = tree->at(S).missations;
    auto& PX = g.branch_infos[Spr_graft::k_branch_info_P_X];
    auto& PS = g.branch_infos[Spr_graft::k_branch_info_P_S];
    auto& SPX = g.branch_infos[Spr_graft::k_branch_info_S_P_X];
    ORC_CHECK(muts_S.empty());
    muts_X.clear();
    Site_deltas ref_to_root;
    for (auto& m : muts_P) push_back_site_deltas(m, ref_to_root);
    muts_P.clear();
    for (auto it = PX.hot_muts_to_X.rbegin(); it != PX.hot_muts_to_X.rend(); ++it) {
      auto& m = *it; muts_X.push_back(m);
      push_back_site_deltas({m.site, m.to, m.from}, ref_to_root);
      miss_S.set_from_state(m.site, m.from, tree->ref_sequence);
    }
    for (auto it = PS.hot_muts_to_X.rbegin(); it != PS.hot_muts_to_X.rend(); ++it) {
      auto& m = *it; muts_S.push_back(m);
      push_back_site_deltas({m.site, m.to, m.from}, ref_to_root);
      miss_X.set_from_state(m.site, m.from, tree->ref_sequence);
    }
    for (auto& m : SPX.hot_muts_to_X) {
      if (m.t > t_P) muts_X.push_back(m);
      else { muts_S.push_back(Mutation{m.to, m.site, m.from, t_P + (t_P - m.t)}); push_back_site_deltas({m.site, m.from, m.to}, ref_to_root); }
    }
    sort_mutations(muts_X); sort_mutations(muts_S);
    muts_P.clear();
    for (auto& [l, d] : ref_to_root) { ORC_CHECK(tree->ref_sequence[l] == d.from); muts_P.push_back(Mutation{d.from, l, d.to, k_neg_dbl_max}); }
    clamp_mutation_times(muts_X, t_P, t_X); clamp_mutation_times(muts_S, t_P, t_S);
    lambda_i->at(P) = lambda_i->at(X) - calc_delta_lambda_across_branch(*evo, tree->ref_sequence, *cumQ, tree->at(X).mutations, tree->at(X).missations);
  }

  // ---- inner grafts (spr_move.cpp:582-1069) ----
  Spr_graft start_inner_graft_analysis(Node_index X) const {   // :582-738
    ORC_CHECK(X != tree->root);
    double t_X = tree->at(X).t; auto P = tree->at(X).parent;
    ORC_CHECK(P != tree->root);
    double t_P = tree->at(P).t; auto S = tree->at(P).sibling_of(X);
    Spr_graft g; g.X = X; g.S = S; g.t_P = t_P;
    {
      g.branch_infos.emplace_back();
      auto& PX = g.branch_infos.back();
      PX.A = P; PX.B = X; PX.is_open = false; PX.T_to_X = t_X - t_P;
      PX.warm_sites.insert({0, tree->num_sites()});
      subtract_interval_sets(PX.hot_sites, PX.warm_sites, tree->at(S).missations.intervals);
    }
    Missation_map sliding = tree->at(S).missations;
    {
      auto& PX = g.branch_infos[0];
      PX.partial_lambda_at_A = lambda_i->at(X);
      auto& mx = tree->at(X).mutations;
      for (auto it = mx.rbegin(); it != mx.rend(); ++it) PX.partial_lambda_at_A += dq(*it, it->to, it->from);
    }
    double next_partial_lambda_at_B = -1 * calc_delta_lambda_across_missations(*evo, tree->ref_sequence, *cumQ, sliding);
    g.branch_infos[0].partial_lambda_at_A -= next_partial_lambda_at_B;
    auto cur = P; auto parent = tree->at(cur).parent; auto sibling = tree->at(parent).sibling_of(cur);
    double partial_lambda = next_partial_lambda_at_B;
    while (!sliding.empty()) {
      g.branch_infos.emplace_back();
      auto* bi = &g.branch_infos.back();
      bi->A = parent; bi->B = cur; bi->is_open = false; bi->T_to_X = t_X - tree->at(parent).t;
      bi->warm_sites = sliding.intervals;
      auto& mc = tree->at(cur).mutations;
      for (auto it = mc.rbegin(); it != mc.rend(); ++it) {
        auto& m = *it;
        if (sliding.contains(m.site)) { partial_lambda += dq(m, m.to, m.from); sliding.set_from_state(m.site, m.from, tree->ref_sequence); }
      }
      subtract_interval_sets(bi->hot_sites, bi->warm_sites, tree->at(sibling).missations.intervals);
      subtract_interval_sets(sliding.intervals, bi->warm_sites, bi->hot_sites);
      for (auto it = sliding.from_states.begin(); it != sliding.from_states.end();) {
        if (!sliding.intervals.contains(it->first)) it = sliding.from_states.erase(it); else ++it;
      }
      next_partial_lambda_at_B = -1 * calc_delta_lambda_across_missations(*evo, tree->ref_sequence, *cumQ, sliding);
      bi->partial_lambda_at_A = partial_lambda - next_partial_lambda_at_B;
      partial_lambda = next_partial_lambda_at_B;
      if (parent != tree->root) {
        cur = parent; parent = tree->at(cur).parent; sibling = tree->at(parent).sibling_of(cur);
      } else {
        if (!can_change_root) { bi->hot_sites = bi->warm_sites; bi->partial_lambda_at_A += partial_lambda; }
        else if (!sliding.empty()) {
          g.branch_infos.emplace_back();
          auto& fo = g.branch_infos.back();
          fo.A = k_no_node; fo.B = tree->root; fo.is_open = true; fo.T_to_X = t_X - tree->at(parent).t;
          fo.warm_sites = sliding.intervals; fo.hot_sites = fo.warm_sites; fo.partial_lambda_at_A = partial_lambda;
        }
        sliding.clear();
      }
    }
    for (int i = 0; i < (int)g.branch_infos.size(); ++i) {
      auto& bi_i = g.branch_infos[i];
      if (bi_i.B == tree->root) continue;
      auto& mb = tree->at(bi_i.B).mutations;
      for (auto it = mb.rbegin(); it != mb.rend(); ++it) {
        auto& m = *it;
        if (bi_i.warm_sites.contains(m.site)) {
          bool found = false;
          for (int j = i; j < (int)g.branch_infos.size(); ++j)
            if (g.branch_infos[j].hot_sites.contains(m.site)) { g.branch_infos[j].hot_muts_to_X.push_back(m); found = true; }
          ORC_CHECK(found);
        }
      }
    }
    for (auto& bi : g.branch_infos) {
      std::reverse(bi.hot_muts_to_X.begin(), bi.hot_muts_to_X.end());
      bi.partial_lambda_at_X = bi.partial_lambda_at_A;
      for (auto& m : bi.hot_muts_to_X) {
        if (!bi.is_open) push_back_site_deltas(m, bi.hot_deltas_to_X);
        bi.partial_lambda_at_X += dq(m, m.from, m.to);
      }
    }
    return g;
  }
  void propose_new_inner_graft_mutations(Spr_graft& g, Rng& rng) const {   // :740-785
    auto X = g.X; auto path_end = tree->node_loc(X);
    for (auto& bi : g.branch_infos) {
      if (bi.hot_sites.empty()) { ORC_CHECK(bi.hot_muts_to_X.empty()); continue; }
      auto nm = bi.is_open ? sample_unconstrained_mutational_history(tree->num_sites(), bi.T_to_X, mu_proposal, rng)
                           : sample_mutational_history(tree->num_sites(), bi.T_to_X, mu_proposal, bi.hot_deltas_to_X, rng);
      if (!nm.empty()) {
        nm.erase(std::remove_if(nm.begin(), nm.end(), [&](const Mutation& m) { return !bi.hot_sites.contains(m.site); }), nm.end());
        if (bi.B == X)
          nm.erase(std::remove_if(nm.begin(), nm.end(), [&](const Mutation& m) {
            if (bi.hot_deltas_to_X.count(m.site)) return false;
            return is_site_missing_at(*tree, X, m.site); }), nm.end());
        adjust_mutational_history(nm, bi.hot_deltas_to_X, *tree, path_end);
      }
      bi.hot_muts_to_X = std::move(nm);
      if (bi.is_open) {
        bi.partial_lambda_at_A = bi.partial_lambda_at_X;
        for (auto it = bi.hot_muts_to_X.rbegin(); it != bi.hot_muts_to_X.rend(); ++it) bi.partial_lambda_at_A += dq(*it, it->to, it->from);
      }
    }
  }
  void finish_inner_graft_analysis(Spr_graft& g) const {   // :787-836
    auto X = g.X; double t_X = tree->at(X).t;
    g.delta_log_G = 0.0;
    for (auto& bi : g.branch_infos) g.delta_log_G += calc_branch_log_G(t_X - bi.T_to_X, t_X, bi.partial_lambda_at_X, *evo, bi.hot_muts_to_X);
    if (g.branch_infos.back().is_open) {
      ORC_CHECK(can_change_root);
      for (auto& m : g.branch_infos.back().hot_muts_to_X) g.delta_log_G += std::log(evo->pi_l_a(m.site, m.from) / evo->pi_l_a(m.site, m.to));
    }
    g.log_alpha_mut = 0.0;
    for (auto& bi : g.branch_infos) {
      int L = bi.hot_sites.num_sites();
      if (bi.B == X) L = (tree->num_sites() - num_missing->at(X)) - (bi.warm_sites.num_sites() - bi.hot_sites.num_sites());
      double T = bi.T_to_X; int M = (int)bi.hot_muts_to_X.size();
      g.log_alpha_mut += -mu_proposal * L * T + M * std::log(mu_proposal / 3);
      if (!bi.is_open) {
        int d = (int)bi.hot_deltas_to_X.size();
        double P_AC = -0.25 * std::expm1(-4. / 3. * mu_proposal * T);
        g.log_alpha_mut -= (L - d) * std::log1p(-3 * P_AC) + d * std::log(P_AC);
      }
    }
  }
  void recalc_lambda_along_hot_path(const Spr_graft& g) {   // :943-950 and :1059-1066
    for (int i = 0; i + 1 < (int)g.branch_infos.size(); ++i) {
      auto A = g.branch_infos[i].A, B = g.branch_infos[i].B;
      lambda_i->at(A) = lambda_i->at(B) - calc_delta_lambda_across_branch(*evo, tree->ref_sequence, *cumQ, tree->at(B).mutations, tree->at(B).missations);
    }
  }
  void peel_inner_graft(const Spr_graft& g) {   // :838-953
    auto X = g.X; double t_X = tree->at(X).t; auto P = tree->at(X).parent; double t_P = tree->at(P).t;
    auto& muts_X = tree->at(X).mutations;
    auto& final_path = g.branch_infos.back();
    Site_deltas ref_to_root;
    if (final_path.is_open) for (auto& m : tree->at_root().mutations) push_back_site_deltas(m, ref_to_root);
    for (auto& bi : g.branch_infos) {
      if (bi.B == tree->root) continue;
      if (bi.B == X && !final_path.is_open) { tree->at(X).mutations.clear(); continue; }
      auto& mb = tree->at(bi.B).mutations;
      for (auto it = mb.rbegin(); it != mb.rend(); ++it) {
        auto& m = *it;
        if (m.site < 0) continue;   // already marked (cannot happen: each branch visited once; kept for safety)
        if (bi.warm_sites.contains(m.site)) {
          if (!final_path.is_open || !final_path.hot_sites.contains(m.site)) {
            for (auto cur = X; cur != bi.B; cur = tree->at(cur).parent) {
              auto parent = tree->at(cur).parent; auto sib = tree->at(parent).sibling_of(cur);
              tree->at(sib).missations.set_from_state(m.site, m.from, tree->ref_sequence);
            }
            m.site = -1;
          }
        }
      }
    }
    if (final_path.is_open) {
      for (auto bit = g.branch_infos.rbegin(); bit != g.branch_infos.rend(); ++bit) {
        auto& bi = *bit;
        if (bi.B == tree->root) continue;
        for (auto& m : tree->at(bi.B).mutations) {
          if (m.site < 0) continue;
          if (final_path.hot_sites.contains(m.site)) {
            for (auto cur = bi.B; cur != tree->root; cur = tree->at(cur).parent) {
              auto parent = tree->at(cur).parent; auto sib = tree->at(parent).sibling_of(cur);
              tree->at(sib).missations.set_from_state(m.site, m.to, tree->ref_sequence);
            }
            push_back_site_deltas(m, ref_to_root);
            m.site = -1;
          }
        }
      }
    }
    for (auto& bi : g.branch_infos) {
      if (bi.B == tree->root) continue;
      auto& mb = tree->at(bi.B).mutations;
      mb.erase(std::remove_if(mb.begin(), mb.end(), [](const Mutation& m) { return m.site == -1; }), mb.end());
    }
    double t_mut_X = 0.5 * (t_P + t_X);
    for (auto& bi : g.branch_infos) {
      if (bi.B == tree->root) continue;
      for (auto& [l, d] : bi.hot_deltas_to_X) muts_X.push_back(Mutation{d.from, l, d.to, t_mut_X});
    }
    if (final_path.is_open) {
      auto& mr = tree->at_root().mutations; mr.clear();
      for (auto& [l, d] : ref_to_root) { ORC_CHECK(tree->ref_sequence[l] == d.from); mr.push_back(Mutation{d.from, l, d.to, k_neg_dbl_max}); }
    }
    recalc_lambda_along_hot_path(g);
  }
  void apply_inner_graft(const Spr_graft& g) {   // :955-1069
    auto X = g.X;
    auto& muts_X = tree->at(X).mutations;
    auto& final_path = g.branch_infos.back();
    muts_X.clear();
    Site_deltas ref_to_root;
    if (final_path.is_open) for (auto& m : tree->at_root().mutations) push_back_site_deltas(m, ref_to_root);
    for (auto& bi : g.branch_infos) {
      if (bi.B == X) { tree->at(X).mutations = bi.hot_muts_to_X; continue; }
      if (!bi.is_open) {
        for (auto& m : bi.hot_muts_to_X) {
          for (auto cur = X; cur != bi.A; cur = tree->at(cur).parent) {
            auto parent = tree->at(cur).parent;
            if (tree->at(parent).t <= m.t && m.t < tree->at(cur).t) { tree->at(cur).mutations.push_back(m); break; }
            auto sib = tree->at(parent).sibling_of(cur);
            tree->at(sib).missations.set_from_state(m.site, m.to, tree->ref_sequence);
          }
        }
      } else {
        for (auto it = bi.hot_muts_to_X.rbegin(); it != bi.hot_muts_to_X.rend(); ++it) {
          auto& m = *it;
          for (auto cur = X; cur != tree->root; cur = tree->at(cur).parent) {
            auto parent = tree->at(cur).parent;
            if (tree->at(parent).t <= m.t && m.t < tree->at(cur).t) tree->at(cur).mutations.push_back(m);
            if (tree->at(parent).t <= m.t) {
              auto sib = tree->at(parent).sibling_of(cur);
              tree->at(sib).missations.set_from_state(m.site, m.from, tree->ref_sequence);
            }
          }
          push_back_site_deltas({m.site, m.to, m.from}, ref_to_root);
        }
      }
    }
    for (auto& bi : g.branch_infos) {
      if (!bi.is_open) {
        double t_A = tree->at(bi.A).t, t_B = tree->at(bi.B).t;
        sort_mutations(tree->at(bi.B).mutations);
        clamp_mutation_times(tree->at(bi.B).mutations, t_A, t_B);
      }
    }
    if (final_path.is_open) {
      auto& mr = tree->at_root().mutations; mr.clear();
      for (auto& [l, d] : ref_to_root) mr.push_back(Mutation{d.from, l, d.to, k_neg_dbl_max});
    }
    recalc_lambda_along_hot_path(g);
  }

  // ---- topological move (spr_move.cpp:1101-1156) ----
  void move(Node_index X, Node_index SS, double new_t_P) {
    ORC_CHECK(X != tree->root);
    auto P = tree->at(X).parent; auto G = tree->at(P).parent; auto S = tree->at(P).sibling_of(X);
    if (SS == P) SS = S;
    auto GG = tree->at(SS).parent;
    if (GG == P) GG = G;
    auto A = find_MRCA_of(*tree, G, GG);
    Tree_editing_session edit{*tree, X, *evo, *lambda_i, *cumQ, *num_missing};
    while (tree->at(P).parent != A) { edit.slide_P_along_branch(tree->at_parent_of(P).t); edit.hop_up(); }
    if (!(descends_from(*tree, S, SS) || descends_from(*tree, SS, S))) {
      ORC_CHECK(A != k_no_node);
      edit.slide_P_along_branch(tree->at(A).t);
      edit.flip();
    }
    std::vector<Node_index> branches_to_SS;
    auto Xs_sib = tree->at(P).sibling_of(X);
    for (auto cur = SS; cur != Xs_sib; cur = tree->at(cur).parent) branches_to_SS.push_back(cur);
    for (auto it = branches_to_SS.rbegin(); it != branches_to_SS.rend(); ++it) {
      auto Y = *it;
      ORC_CHECK(tree->at(SS).parent != P);
      edit.slide_P_along_branch(tree->at_parent_of(Y).t);
      edit.hop_down(Y);
    }
    ORC_CHECK(tree->at(P).sibling_of(X) == SS);
    edit.slide_P_along_branch(new_t_P);
    edit.end();
  }
};

}  // namespace orc
#endif  // ORC_SPR_HPP_
