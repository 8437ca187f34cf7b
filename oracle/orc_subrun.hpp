// orc_subrun.hpp -- CPU ORACLE (test infrastructure, NOT product code).
// Restates reference core/subrun.{h,cpp}: one Markov chain over one partition part.
#ifndef ORC_SUBRUN_HPP_
#define ORC_SUBRUN_HPP_

#include "orc_coalescent.hpp"
#include "orc_spr.hpp"

namespace orc {

enum Move_kind { k_inner_node_displace = 0, k_tip_displace = 1, k_branch_reform = 2, k_subtree_slide = 3, k_spr1 = 4 };

struct Move_trace_entry { double kind, node, accepted, log_mh; };

struct Subrun {
  Rng* rng;
  Phylo_tree tree;
  bool includes_run_root;
  double t_max_tip = std::numeric_limits<double>::quiet_NaN();
  Global_evo_model evo;
  Very_scalable_coalescent_prior_part* coalescent_prior_part = nullptr;
  bool only_displacing_inner_nodes = false;
  bool topology_moves_enabled = true;

  bool derived_valid = false;
  double log_G = 0.0;
  State_freqs ref_freqs;
  std::vector<double> lambda_i, ref_cum_Q_l;
  std::vector<int> num_sites_missing;
  double log_augmented_coalescent_prior = 0.0;

  // bookkeeping (not in the reference): per-kind counters and an optional move trace for parity tests
  int64_t proposed[5] = {0, 0, 0, 0, 0}, accepted[5] = {0, 0, 0, 0, 0};
  int64_t moves_done = 0;
  int trace_capacity = 0;
  std::vector<Move_trace_entry> trace;
  Move_trace_entry cur_trace{};

  Subrun(Rng& r, Phylo_tree t, bool incl_root, Global_evo_model e) : rng(&r), tree(std::move(t)), includes_run_root(incl_root), evo(std::move(e)) {}

  void set_evo(const Global_evo_model& e) { evo = e; derived_valid = false; }
  void set_coalescent_prior_part(Very_scalable_coalescent_prior_part* p) { coalescent_prior_part = p; derived_valid = false; }
  void invalidate_derived_quantities() { derived_valid = false; }

  double calc_cur_log_G() const {   // subrun.cpp:58-68
    double r = 0.0;
    if (includes_run_root) r += calc_log_root_prior(tree, evo, ref_freqs);
    r += calc_log_G_below_root(tree, evo, lambda_i, ref_freqs);
    return r;
  }
  void recalc_derived_quantities() {   // subrun.cpp:17-26
    ref_freqs = calc_state_frequencies_per_partition_of(tree.ref_sequence, evo);
    ref_cum_Q_l = calc_cum_Q_l_for_sequence(tree.ref_sequence, evo);
    lambda_i = calc_lambda_i(tree, evo, ref_cum_Q_l);
    log_G = calc_cur_log_G();
    num_sites_missing = calc_num_sites_missing_at_every_node(tree);
    log_augmented_coalescent_prior = coalescent_prior_part ? coalescent_prior_part->calc_partial_log_prior() : -1.0;
  }
  void validate_derived_quantities() { if (!derived_valid) { recalc_derived_quantities(); derived_valid = true; } }

  // subrun.cpp:28-56, as a report for tests instead of CHECKs
  std::string check_derived_quantities() {
    char buf[256];
    if (coalescent_prior_part) {
      double e = coalescent_prior_part->calc_partial_log_prior();
      if (!(std::abs(log_augmented_coalescent_prior - e) < 1e-5)) { std::snprintf(buf, sizeof buf, "log_aug_prior %.12g != %.12g", log_augmented_coalescent_prior, e); return buf; }
    }
    auto el = calc_lambda_i(tree, evo, ref_cum_Q_l);
    for (int n = 0; n < tree.size(); ++n)
      if (!(std::abs(lambda_i[n] - el[n]) / (double)tree.num_sites() < 1e-8)) { std::snprintf(buf, sizeof buf, "lambda_i[%d] %.12g != %.12g", n, lambda_i[n], el[n]); return buf; }
    double eg = calc_cur_log_G();
    if (!(std::abs(log_G - eg) < 1e-6)) { std::snprintf(buf, sizeof buf, "log_G %.12g != %.12g", log_G, eg); return buf; }
    if (num_sites_missing != calc_num_sites_missing_at_every_node(tree)) return "num_sites_missing mismatch";
    return "";
  }

  int pick_random_node() { return rng->uniform_int(tree.size()); }                       // :123-126
  int pick_random_inner_node() { while (true) { int r = pick_random_node(); if (tree.at(r).is_inner_node()) return r; } }
  int pick_random_tip() { while (true) { int r = pick_random_node(); if (tree.at(r).is_tip()) return r; } }

  void begin_move(int kind) { ++proposed[kind]; cur_trace = {(double)kind, -1.0, 0.0, std::numeric_limits<double>::quiet_NaN()}; }
  void note(int node, double log_mh, bool acc, int kind) { cur_trace.node = node; cur_trace.log_mh = log_mh; cur_trace.accepted = acc ? 1.0 : 0.0; if (acc) ++accepted[kind]; }

  void mcmc_sub_iteration() {   // subrun.cpp:98-121
    validate_derived_quantities();
    cur_trace = {-1.0, -1.0, 0.0, std::numeric_limits<double>::quiet_NaN()};
    if (only_displacing_inner_nodes) {
      inner_node_displace_move();
    } else {
      double total_weight = 15.0 + 15.0;
      if (topology_moves_enabled) total_weight += 1.0 + 1.0;
      double r = rng->uniform_co(0.0, total_weight);
      if (r < 7.5) inner_node_displace_move();
      else if (r < 15.0) tip_displace_move();
      else if (r < 30.0) branch_reform_move();
      else if (topology_moves_enabled) { if (r < 31.0) subtree_slide_move(); else spr1_move(); }
    }
    if ((int)trace.size() < trace_capacity) trace.push_back(cur_trace);
    ++moves_done;
  }

  void inner_node_displace_move() {   // subrun.cpp:148-232
    begin_move(k_inner_node_displace);
    if (tree.size() < 1) return;
    // NOTE: a 1-node subtree has no inner node and the reference would spin forever; partitions have >= 3 nodes.
    int node = pick_random_inner_node();
    cur_trace.node = node;
    if (node == tree.root && !includes_run_root) return;
    double t_min = -std::numeric_limits<double>::infinity();
    if (node != tree.root) { t_min = tree.at_parent_of(node).t; for (auto& m : tree.at(node).mutations) t_min = std::max(t_min, m.t); }
    double t_max = +std::numeric_limits<double>::infinity();
    for (int k = 0; k < 2; ++k) { int c = tree.at(node).children[k]; t_max = std::min(t_max, tree.at(c).t); for (auto& m : tree.at(c).mutations) t_max = std::min(t_max, m.t); }
    double lambda_at_node = lambda_i[node];
    double d_logG_dt = 0.0;
    if (node != tree.root) d_logG_dt += -lambda_at_node;
    for (int k = 0; k < 2; ++k) {
      int c = tree.at(node).children[k];
      double lambda_just_below = lambda_at_node + calc_delta_lambda_across_missations(evo, tree.ref_sequence, ref_cum_Q_l, tree.at(c).missations);
      d_logG_dt -= -lambda_just_below;
    }
    double old_t = tree.at(node).t, log_alpha_ratio = 0.0, new_t = old_t;
    if (node == tree.root) {
      double tree_span = t_max_tip - t_max;
      ORC_CHECK(tree_span >= 0.0);
      double delta_scale = std::min((1 / lambda_i.at(node)) / 2, tree_span);
      new_t = old_t + rng->gaussian(0.0, delta_scale);
      if (new_t < t_min || new_t > t_max) return;
      log_alpha_ratio = 0.0;
    } else {
      Bounded_exponential_distribution dist{d_logG_dt, t_min, t_max};
      new_t = dist(*rng);
      log_alpha_ratio = d_logG_dt * (new_t - old_t);
    }
    if (new_t == t_min || new_t == t_max) return;
    double delta_log_G = d_logG_dt * (new_t - old_t);
    double delta_log_prior = coalescent_prior_part->calc_delta_partial_log_prior_after_displace_coalescence(old_t, new_t);
    double log_mh = delta_log_G + delta_log_prior - log_alpha_ratio;
    bool acc = log_mh >= 0.0 || rng->uniform_co(0.0, 1.0) < std::exp(log_mh);
    note(node, log_mh, acc, k_inner_node_displace);
    if (acc) {
      coalescent_prior_part->coalescence_displaced(old_t, new_t);
      tree.at(node).t = new_t;
      log_G += d_logG_dt * (new_t - old_t);
      log_augmented_coalescent_prior += delta_log_prior;
    }
  }

  void tip_displace_move() {   // subrun.cpp:234-285
    begin_move(k_tip_displace);
    if (tree.size() < 1) return;
    int node = pick_random_tip();
    cur_trace.node = node;
    ORC_CHECK(node != tree.root);
    if (tree.at(node).t_min == tree.at(node).t_max) return;
    double t_min = std::max((double)tree.at(node).t_min, tree.at_parent_of(node).t);
    for (auto& m : tree.at(node).mutations) t_min = std::max(t_min, m.t);
    double t_max = (double)tree.at(node).t_max;
    double d_logG_dt = -lambda_i[node];
    double old_t = tree.at(node).t;
    Bounded_exponential_distribution dist{d_logG_dt, t_min, t_max};
    double new_t = dist(*rng);
    double log_alpha_ratio = d_logG_dt * (new_t - old_t);
    if (new_t == t_min || new_t == t_max) return;
    double delta_log_G = d_logG_dt * (new_t - old_t);
    double delta_log_prior = coalescent_prior_part->calc_delta_partial_log_prior_after_displace_tip(old_t, new_t);
    double log_mh = delta_log_G + delta_log_prior - log_alpha_ratio;
    bool acc = log_mh >= 0.0 || rng->uniform_co(0.0, 1.0) < std::exp(log_mh);
    note(node, log_mh, acc, k_tip_displace);
    if (acc) {
      coalescent_prior_part->tip_displaced(old_t, new_t);
      tree.at(node).t = new_t;
      log_G += d_logG_dt * (new_t - old_t);
      log_augmented_coalescent_prior += delta_log_prior;
    }
  }

  void branch_reform_move() {   // subrun.cpp:287-320
    begin_move(k_branch_reform);
    if (tree.size() < 3) return;
    int X = pick_random_node();
    cur_trace.node = X;
    if (X == tree.root) return;
    int P = tree.at(X).parent;
    int S = tree.at(P).sibling_of(X);
    double t_X = tree.at(X).t, t_P = tree.at(P).t;
    if (P == tree.root) spr_move_core(X, {S, t_P}, 1.0);   // falls through
    auto new_mutations = randomize_branch_mutation_times(tree, X, *rng);
    double delta_log_G = calc_branch_log_G(t_P, t_X, lambda_i.at(X), evo, new_mutations)
        - calc_branch_log_G(t_P, t_X, lambda_i.at(X), evo, tree.at(X).mutations);
    double log_mh = delta_log_G;
    bool acc = log_mh >= 0.0 || rng->uniform_co(0.0, 1.0) < std::exp(log_mh);
    note(X, log_mh, acc, k_branch_reform);
    if (acc) { tree.at(X).mutations = new_mutations; log_G += delta_log_G; }
  }

  void enumerate_descendant_branches_straddling(int P, double t, int X, std::vector<int>& out) const {   // subrun.cpp:325-350
    if (P == X) return;
    if (t <= tree.at(P).t) out.push_back(P);
    else if (tree.at(P).is_inner_node()) for (int k = 0; k < 2; ++k) enumerate_descendant_branches_straddling(tree.at(P).children[k], t, X, out);
  }

  void subtree_slide_move() {   // subrun.cpp:352-448
    begin_move(k_subtree_slide);
    if (tree.size() < 2) return;
    int X = pick_random_node();
    cur_trace.node = X;
    if (X == tree.root) return;
    int P = tree.at(X).parent, S = tree.at(P).sibling_of(X);
    double t_early = (P == tree.root) ? std::min(tree.at(X).t, tree.at(S).t) : tree.at(tree.root).t;
    double tree_span = t_max_tip - t_early;
    ORC_CHECK(tree_span >= 0.0);
    double delta_scale = std::min((1 / lambda_i.at(X)) / 2, tree_span);
    double delta_t = rng->gaussian(0.0, delta_scale);
    double old_P_t = tree.at(P).t, new_P_t = old_P_t + delta_t;
    if (delta_t < 0.0) {
      if (P != tree.root && new_P_t < tree.at_parent_of(P).t) {
        int GG = tree.at(P).parent, SS = P;
        while (new_P_t < tree.at(GG).t) { SS = GG; GG = tree.at(GG).parent; if (GG == k_no_node) break; }
        std::vector<int> branches;
        enumerate_descendant_branches_straddling(SS, old_P_t, X, branches);
        double a_o2n = 1.0, a_n2o = 1.0 / (double)branches.size();
        spr_move_core(X, {SS, new_P_t}, a_n2o / a_o2n);
      } else spr_move_core(X, {S, new_P_t}, 1.0);
    } else {
      if (new_P_t > tree.at(X).t) return;
      if (new_P_t > tree.at(S).t) {
        std::vector<int> branches;
        enumerate_descendant_branches_straddling(P, new_P_t, X, branches);
        if (branches.empty()) return;
        int bi = rng->uniform_int((int)branches.size());
        int SS = branches[bi];
        double a_o2n = 1.0 / (double)branches.size(), a_n2o = 1.0;
        spr_move_core(X, {SS, new_P_t}, a_n2o / a_o2n);
      } else spr_move_core(X, {S, new_P_t}, 1.0);
    }
  }

  void spr1_move() {   // subrun.cpp:492-675
    begin_move(k_spr1);
    if (tree.size() < 2) return;
    double chooser = rng->uniform_co(0.0, 1.0);
    int limit = chooser < 0.01 ? std::numeric_limits<int>::max() : 1;
    double mu_JC = lambda_i.at(tree.root) / (tree.num_sites() - num_sites_missing.at(tree.root));
    double annealing_factor = 0.8;
    int X;
    do { X = pick_random_node(); } while (tree.root == X);
    cur_trace.node = X;
    if (lambda_i.at(X) == 0.0) return;
    double t_X = tree.at(X).t;
    int P = tree.at(X).parent;
    double old_t_P = tree.at(P).t;
    int old_S = tree.at(P).sibling_of(X);
    int old_G = tree.at(P).parent;
    bool pruning_changes_root = P == tree.root;
    if (pruning_changes_root && !includes_run_root) return;
    Spr_move spr{tree, mu_JC, includes_run_root, evo, lambda_i, ref_cum_Q_l, num_sites_missing};
    auto old_graft = spr.analyze_graft(X);
    spr.peel_graft(old_graft);
    int old_min_muts = spr.count_min_mutations(old_graft);
    auto old_deltas = spr.summarize_closed_mutations(old_graft);
    auto missing_at_X = reconstruct_missing_sites_at(tree, X);
    Spr_study_builder pre_builder{tree, X, t_X, missing_at_X};
    pre_builder.max_muts_from_start = limit;
    pre_builder.seed_fill_from(old_S, 0, std::move(old_deltas), includes_run_root);
    Spr_study pre_study{std::move(pre_builder), lambda_i.at(X), annealing_factor, t_X, t_max_tip};
    int new_region = pre_study.pick_nexus_region(*rng);
    int new_S = pre_study.candidate_regions[new_region].branch;
    ORC_CHECK(new_S != P);
    double new_t_P = pre_study.pick_time_in_region(new_region, *rng);
    double log_alpha_o2n = pre_study.log_alpha_in_region(new_region, new_t_P);
    double t_new_S = tree.at(new_S).t;
    int new_G = tree.at(new_S).parent;
    if (new_G == P) new_G = old_G;
    double t_new_G = (new_G == k_no_node) ? k_neg_dbl_max : tree.at(new_G).t;
    if (new_t_P == t_X || new_t_P == t_new_S || new_t_P == t_new_G) { spr.apply_graft(old_graft); return; }
    spr.move(X, new_S, new_t_P);
    auto new_graft = spr.propose_new_graft(X, *rng);
    ORC_CHECK(tree.at(X).parent == P);
    int new_min_muts = spr.count_min_mutations(new_graft);
    auto new_deltas = spr.summarize_closed_mutations(new_graft);
    Spr_study_builder post_builder{tree, X, t_X, missing_at_X};
    post_builder.max_muts_from_start = limit;
    post_builder.seed_fill_from(new_S, 0, std::move(new_deltas), includes_run_root);
    Spr_study post_study{std::move(post_builder), lambda_i.at(X), annealing_factor, t_X, t_max_tip};
    int old_region = post_study.find_region(old_S, old_t_P);
    ORC_CHECK(old_region != -1);
    double log_alpha_n2o = post_study.log_alpha_in_region(old_region, old_t_P);
    ORC_CHECK(new_min_muts == pre_study.candidate_regions[new_region].min_muts);
    ORC_CHECK(old_min_muts == post_study.candidate_regions[old_region].min_muts);
    double d_prior = coalescent_prior_part->calc_delta_partial_log_prior_after_displace_coalescence(old_t_P, new_t_P);
    double log_mh = (new_graft.delta_log_G - new_graft.log_alpha_mut) - (old_graft.delta_log_G - old_graft.log_alpha_mut)
        + log_alpha_n2o - log_alpha_o2n + d_prior;
    bool acc = log_mh >= 0.0 || rng->uniform_co(0.0, 1.0) < std::exp(log_mh);
    note(X, log_mh, acc, k_spr1);
    if (acc) {
      spr.apply_graft(new_graft);
      log_G -= old_graft.delta_log_G; log_G += new_graft.delta_log_G;
      log_augmented_coalescent_prior += d_prior;
      coalescent_prior_part->coalescence_displaced(old_t_P, new_t_P);
    } else {
      spr.move(X, old_S, old_t_P);
      spr.apply_graft(old_graft);
    }
  }

  void spr_move_core(int X, Phylo_tree_loc new_nexus, double alpha_ratio) {   // subrun.cpp:683-742
    if (X == tree.root) return;
    if (!includes_run_root) if (tree.at(X).parent == tree.root || new_nexus.branch == tree.root) return;
    double t_X = tree.at(X).t;
    int P = tree.at(X).parent;
    double old_t_P = tree.at(P).t;
    int old_S = tree.at(P).sibling_of(X);
    double new_t_P = new_nexus.t;
    if (new_t_P == t_X || new_t_P == tree.at(new_nexus.branch).t || (P != tree.root && new_t_P == tree.at_parent_of(P).t)) return;
    double mu_JC = lambda_i.at(tree.root) / (tree.num_sites() - num_sites_missing.at(tree.root));
    Spr_move spr{tree, mu_JC, includes_run_root, evo, lambda_i, ref_cum_Q_l, num_sites_missing};
    auto old_graft = spr.analyze_graft(X);
    spr.peel_graft(old_graft);
    spr.move(X, new_nexus.branch, new_nexus.t);
    auto new_graft = spr.propose_new_graft(X, *rng);
    double d_prior = coalescent_prior_part->calc_delta_partial_log_prior_after_displace_coalescence(old_t_P, new_nexus.t);
    double log_mh = (new_graft.delta_log_G - new_graft.log_alpha_mut) - (old_graft.delta_log_G - old_graft.log_alpha_mut)
        + std::log(alpha_ratio) + d_prior;
    bool acc = log_mh >= 0.0 || rng->uniform_co(0.0, 1.0) < std::exp(log_mh);
    if (cur_trace.kind == k_subtree_slide) note(X, log_mh, acc, k_subtree_slide);
    if (acc) {
      spr.apply_graft(new_graft);
      log_G -= old_graft.delta_log_G; log_G += new_graft.delta_log_G;
      log_augmented_coalescent_prior += d_prior;
      coalescent_prior_part->coalescence_displaced(old_t_P, new_nexus.t);
    } else {
      spr.move(X, old_S, old_t_P);
      spr.apply_graft(old_graft);
    }
  }
};

}  // namespace orc
#endif  // ORC_SUBRUN_HPP_
