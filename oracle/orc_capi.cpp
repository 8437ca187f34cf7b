// orc_capi.cpp -- CPU ORACLE (test infrastructure, NOT product code).
// C entry points over the oracle, shaped like include/emat_backend.h (prefix orc_ instead of
// emat_) so that tests/ can run the same scenario through both and compare.  Also used by
// bench.py's cpu_baseline leg (kind = "port").
#include <cstring>
#include <thread>

#include "../include/emat_backend.h"
#include "orc_build.hpp"
#include "orc_utree.hpp"
#include "orc_run.hpp"
#include "orc_subrun.hpp"

using namespace orc;

namespace {

Phylo_tree tree_from_flat(const emat_flat_tree& v, const std::vector<State>& ref) {
  Phylo_tree t(v.num_nodes);
  t.root = v.root;
  t.ref_sequence = ref;
  for (int i = 0; i < v.num_nodes; ++i) {
    auto& nd = t.at(i);
    nd.parent = v.parent[i]; nd.children[0] = v.child0[i]; nd.children[1] = v.child1[i];
    nd.t = v.t[i]; nd.t_min = v.t_min[i]; nd.t_max = v.t_max[i];
    for (int k = v.mut_offset[i]; k < v.mut_offset[i + 1]; ++k) nd.mutations.push_back(Mutation{v.mut_from[k], v.mut_site[k], v.mut_to[k], v.mut_t[k]});
    for (int k = v.miss_offset[i]; k < v.miss_offset[i + 1]; ++k) nd.missations.intervals.v.push_back({v.miss_start[k], v.miss_end[k]});
    for (int k = v.mfs_offset[i]; k < v.mfs_offset[i + 1]; ++k) nd.missations.from_states[v.mfs_site[k]] = v.mfs_state[k];
  }
  return t;
}

struct Part {
  Rng rng;
  std::unique_ptr<Subrun> subrun;
};

struct Engine {
  int L = 0;
  std::vector<State> ref;
  Global_evo_model evo;
  double t_max_tip = 0.0;
  bool only_displacing_inner_nodes = false, topology_moves_enabled = true;
  std::vector<std::unique_ptr<Part>> parts;
  std::vector<Very_scalable_coalescent_prior_part> coal_parts;
  std::shared_ptr<const Pop_model> pop_model;
  int trace_moves = 0;
  std::string last_error;
};

std::shared_ptr<const Pop_model> make_pop_model(const emat_pop_model& pm) {
  switch (pm.kind) {
    case EMAT_POP_CONST: return std::make_shared<Const_pop_model>(pm.p[0]);
    case EMAT_POP_EXP: return std::make_shared<Exp_pop_model>(pm.p[0], pm.p[1], pm.p[2], pm.p[3]);
    case EMAT_POP_SKYGRID:
      return std::make_shared<Skygrid_pop_model>(
          std::vector<double>(pm.skygrid_x, pm.skygrid_x + pm.skygrid_num_knots),
          std::vector<double>(pm.skygrid_gamma, pm.skygrid_gamma + pm.skygrid_num_knots),
          (Skygrid_pop_model::Type)pm.skygrid_type);
  }
  throw std::invalid_argument("bad pop model kind");
}

}  // namespace

#define ORC_TRY try {
#define ORC_CATCH } catch (const std::exception& ex) { e->last_error = ex.what(); return EMAT_ERR_INTERNAL; } return EMAT_OK;

extern "C" {

typedef struct Engine orc_engine;

int orc_create(int num_sites, int trace_moves, orc_engine** out) {
  auto* e = new Engine; e->L = num_sites; e->trace_moves = trace_moves; *out = e; return EMAT_OK;
}
int orc_destroy(orc_engine* e) { delete e; return EMAT_OK; }
const char* orc_last_error(orc_engine* e) { return e->last_error.c_str(); }

int orc_set_ref_sequence(orc_engine* e, const uint8_t* ref, int num_sites) {
  ORC_TRY
  ORC_CHECK(num_sites == e->L);
  e->ref.assign(ref, ref + num_sites);
  ORC_CATCH
}
int orc_set_evo(orc_engine* e, int P, const double* mu, const double* pi, const double* q, const double* nu_l, const int* partition_for_site) {
  ORC_TRY
  Global_evo_model g;
  g.partition_for_site.assign(partition_for_site, partition_for_site + e->L);
  g.nu_l.assign(nu_l, nu_l + e->L);
  g.partition_evo_model.resize(P);
  for (int p = 0; p < P; ++p) {
    g.partition_evo_model[p].mu = mu[p];
    for (int a = 0; a < 4; ++a) { g.partition_evo_model[p].pi_a[a] = pi[4 * p + a]; for (int b = 0; b < 4; ++b) g.partition_evo_model[p].q_ab[a][b] = q[16 * p + 4 * a + b]; }
  }
  e->evo = g;
  for (auto& pt : e->parts) if (pt && pt->subrun) pt->subrun->set_evo(e->evo);
  ORC_CATCH
}
int orc_set_flags(orc_engine* e, double t_max_tip, int only_displacing_inner_nodes, int topology_moves_enabled) {
  e->t_max_tip = t_max_tip; e->only_displacing_inner_nodes = only_displacing_inner_nodes != 0; e->topology_moves_enabled = topology_moves_enabled != 0;
  for (auto& pt : e->parts) if (pt && pt->subrun) {
    pt->subrun->t_max_tip = t_max_tip; pt->subrun->only_displacing_inner_nodes = e->only_displacing_inner_nodes; pt->subrun->topology_moves_enabled = e->topology_moves_enabled;
  }
  return EMAT_OK;
}
int orc_begin_upload(orc_engine* e, int num_parts) {
  e->coal_parts.clear(); e->parts.clear(); e->parts.resize(num_parts); return EMAT_OK;
}
int orc_part_upload(orc_engine* e, int part_id, const emat_flat_tree* subtree, int includes_run_root, uint64_t seed) {
  ORC_TRY
  auto pt = std::make_unique<Part>();
  pt->rng.key = seed; pt->rng.counter = 0; pt->rng.spare = 0; pt->rng.has_spare = false;
  pt->subrun = std::make_unique<Subrun>(pt->rng, tree_from_flat(*subtree, e->ref), includes_run_root != 0, e->evo);
  pt->subrun->t_max_tip = e->t_max_tip;
  pt->subrun->only_displacing_inner_nodes = e->only_displacing_inner_nodes;
  pt->subrun->topology_moves_enabled = e->topology_moves_enabled;
  pt->subrun->trace_capacity = e->trace_moves;
  e->parts.at(part_id) = std::move(pt);
  ORC_CATCH
}
int orc_end_upload(orc_engine*) { return EMAT_OK; }

int orc_build_coalescent_parts(orc_engine* e, const emat_pop_model* pm, int root_part_index, double t_step) {
  ORC_TRY
  e->pop_model = make_pop_model(*pm);
  std::vector<const Phylo_tree*> subtrees; std::vector<Rng*> prngs;
  for (auto& pt : e->parts) { subtrees.push_back(&pt->subrun->tree); prngs.push_back(&pt->rng); }
  e->coal_parts = make_very_scalable_coalescent_prior_parts(subtrees, root_part_index, e->pop_model, prngs, t_step);
  for (size_t i = 0; i < e->parts.size(); ++i) e->parts[i]->subrun->set_coalescent_prior_part(&e->coal_parts[i]);
  ORC_CATCH
}

// One part's coalescent arrays and RNG position handed in from outside (tests: the tables the DEVICE built for the part, read
// back through emat_part_get_coalescent / emat_part_get_rng), instead of built here: the chain that follows is then a function
// of the same inputs as the device's.  Cells outside the window the device keeps arrive as NaN: a chain that reads one shows.
int orc_set_coalescent_part(orc_engine* e, const emat_pop_model* pm, int part_id, int includes_tree_root, int num_cells,
                            const double* k_bar_p, const double* k_tw_p, const double* k_tw, const double* popsize_bar, const int* num_active,
                            double t_ref, double t_step, uint64_t rng_counter, uint64_t rng_spare, int rng_has_spare) {
  ORC_TRY
  ORC_CHECK(part_id >= 0 && part_id < (int)e->parts.size());
  if (e->coal_parts.size() != e->parts.size()) {
    ORC_CHECK(e->coal_parts.empty());   // (pointers into the vector are held by the subruns: sized once)
    e->coal_parts.resize(e->parts.size());
    e->pop_model = make_pop_model(*pm);
  }
  auto& cp = e->coal_parts[part_id]; auto& pt = *e->parts[part_id];
  cp.pop_model = e->pop_model; cp.subtree = &pt.subrun->tree; cp.prng = &pt.rng; cp.includes_tree_root = includes_tree_root != 0;
  cp.k_bar_p.assign(k_bar_p, k_bar_p + num_cells); cp.k_twiddle_bar_p.assign(k_tw_p, k_tw_p + num_cells);
  cp.k_twiddle_bar.assign(k_tw, k_tw + num_cells); cp.popsize_bar.assign(popsize_bar, popsize_bar + num_cells);
  cp.num_active_parts.assign(num_active, num_active + num_cells);
  cp.t_ref = t_ref; cp.t_step = t_step;
  pt.rng.counter = rng_counter; pt.rng.spare = rng_spare; pt.rng.has_spare = rng_has_spare != 0;
  pt.subrun->set_coalescent_prior_part(&cp);
  ORC_CATCH
}

int orc_recalc_derived(orc_engine* e) {
  ORC_TRY
  for (auto& pt : e->parts) { pt->subrun->invalidate_derived_quantities(); pt->subrun->validate_derived_quantities(); }
  ORC_CATCH
}

// moves_per_part[i] moves on part i, spread over `num_threads` host threads (parts are independent,
// reference run.cpp:682-693).  `paranoid` != 0 re-derives everything after every move
// (Subrun::check_derived_quantities + assert_phylo_tree_integrity, subrun.cpp:28-56,120).
int orc_run_moves(orc_engine* e, const int64_t* moves_per_part, int num_threads, int paranoid) {
  ORC_TRY
  int P = (int)e->parts.size();
  if (num_threads < 1) num_threads = 1;
  std::vector<std::string> errors(num_threads);
  auto work = [&](int tid) {
    try {
      for (int p = tid; p < P; p += num_threads) {
        auto& sr = *e->parts[p]->subrun;
        for (int64_t i = 0; i < moves_per_part[p]; ++i) {
          sr.mcmc_sub_iteration();
          if (paranoid) {
            auto msg = check_phylo_tree_integrity(sr.tree);
            if (msg.empty()) msg = sr.check_derived_quantities();
            if (!msg.empty()) throw std::runtime_error("part " + std::to_string(p) + " move " + std::to_string(i) + ": " + msg);
          }
        }
      }
    } catch (const std::exception& ex) { errors[tid] = ex.what(); }
  };
  if (num_threads == 1) work(0);
  else { std::vector<std::thread> th; for (int t = 0; t < num_threads; ++t) th.emplace_back(work, t); for (auto& x : th) x.join(); }
  for (auto& s : errors) if (!s.empty()) throw std::runtime_error(s);
  ORC_CATCH
}

/* Sufficient statistics of the global moves summed over the parts: every branch of the whole tree is a non-root branch
 * of exactly one part (a cut node is a tip of the part above it and the root of the part below), and a part's root
 * lists carry the state at the cut node, so the sums equal the whole-tree values (Run::calc_cur_Ttwiddle_beta_a,
 * calc_cur_num_muts_ab, run.cpp:445-453). */
int orc_global_stats(orc_engine* e, int P, double* Ttwiddle_beta_a /*[P][4]*/, int64_t* num_muts_beta_ab /*[P][4][4]*/, int64_t* num_muts) {
  ORC_TRY
  ORC_CHECK(P == (int)e->evo.partition_evo_model.size());
  for (int i = 0; i < 4 * P; ++i) Ttwiddle_beta_a[i] = 0.0;
  for (int i = 0; i < 16 * P; ++i) num_muts_beta_ab[i] = 0;
  int64_t nm = 0;
  for (auto& pt : e->parts) {
    const Phylo_tree& tree = pt->subrun->tree;
    auto T = calc_Ttwiddle_beta_a(tree, e->evo);
    auto M = calc_num_muts_beta_ab(tree, e->evo);
    for (int b = 0; b < P; ++b) for (int a = 0; a < 4; ++a) { Ttwiddle_beta_a[4 * b + a] += T[b][a]; for (int c = 0; c < 4; ++c) num_muts_beta_ab[16 * b + 4 * a + c] += M[b][a][c]; }
    nm += calc_num_muts(tree);
  }
  if (num_muts) *num_muts = nm;
  ORC_CATCH
}

/* calc_Ttwiddle_l (phylo_tree_calc.cpp:176-222) on the tree of ONE part (upload the whole tree as a single part). */
int orc_Ttwiddle_l(orc_engine* e, int part_id, double* out /*[L]*/) {
  ORC_TRY
  auto v = calc_Ttwiddle_l(e->parts.at(part_id)->subrun->tree, e->evo);
  for (int l = 0; l < e->L; ++l) out[l] = v[l];
  ORC_CATCH
}
/* calc_num_muts_l summed over the parts (the deltas above a part's root are not mutations). */
int orc_num_muts_l(orc_engine* e, int* out /*[L]*/) {
  ORC_TRY
  for (int l = 0; l < e->L; ++l) out[l] = 0;
  for (auto& pt : e->parts) { auto v = calc_num_muts_l(pt->subrun->tree); for (int l = 0; l < e->L; ++l) out[l] += v[l]; }
  ORC_CATCH
}
/* Run::calc_cur_log_coalescent_prior (run.cpp:455-465) on the tree of ONE part (upload the whole tree as a single part):
 * a Scalable_coalescent_prior built as Run builds it (run.cpp:40, 52-58), every node displaced to its time in index order. */
int orc_scalable_log_prior(orc_engine* e, int part_id, double t_ref, double t_step, double* out) {
  ORC_TRY
  const Phylo_tree& tree = e->parts.at(part_id)->subrun->tree;
  ORC_CHECK(e->pop_model != nullptr);
  Scalable_coalescent_prior prior(e->pop_model, tree.size(), t_ref, t_step);
  for (int n = 0; n < tree.size(); ++n) { if (tree.at(n).is_tip()) prior.mark_as_tip(n); else prior.mark_as_coalescence(n); }
  for (int n = 0; n < tree.size(); ++n) { if (tree.at(n).is_tip()) prior.displace_tip(n, tree.at(n).t); else prior.displace_coalescence(n, tree.at(n).t); }
  *out = prior.calc_log_prior();
  ORC_CATCH
}

/* find_MRCA_of (op 0) / descends_from (op 1) on the tree of one part, query by query; -1 = k_no_node (phylo_tree.cpp:204-280, 292-299) */
int orc_tree_query(orc_engine* e, int part_id, int op, int n, const int* a, const int* b, int* out) {
  ORC_TRY
  const Phylo_tree& tree = e->parts.at(part_id)->subrun->tree;
  for (int i = 0; i < n; ++i) out[i] = op == 0 ? (int)find_MRCA_of(tree, (Node_index)a[i], (Node_index)b[i]) : (descends_from(tree, (Node_index)a[i], (Node_index)b[i]) ? 1 : 0);
  ORC_CATCH
}

/* The oracle's Spr_move on one part, with what it found written out in the layout of emat_debug_graft (include/emat_backend.h):
 * mode 0 analyze_graft(X); 1 + peel_graft; 2 + apply_graft; 3 analyze, peel, move(X, new_sibling, new_t_P), propose_new_graft, apply. */
int orc_debug_graft(orc_engine* e, int part_id, int X, double mu_proposal, int mode, int new_sibling, double new_t_P, double* out, int out_cap, int* out_len) {
  ORC_TRY
  Part& pt = *e->parts.at(part_id);
  Subrun& sr = *pt.subrun;
  sr.ref_freqs = calc_state_frequencies_per_partition_of(sr.tree.ref_sequence, sr.evo);
  sr.ref_cum_Q_l = calc_cum_Q_l_for_sequence(sr.tree.ref_sequence, sr.evo);
  sr.lambda_i = calc_lambda_i(sr.tree, sr.evo, sr.ref_cum_Q_l);
  sr.num_sites_missing = calc_num_sites_missing_at_every_node(sr.tree);
  if (mode == 3) sr.log_G = sr.calc_cur_log_G();
  Spr_move spr{sr.tree, mu_proposal, sr.includes_run_root, sr.evo, sr.lambda_i, sr.ref_cum_Q_l, sr.num_sites_missing};
  int n = 0;
  auto put = [&](double v) { if (n < out_cap) out[n] = v; ++n; };
  auto put_graft = [&](const Spr_graft& g) {
    put((double)g.branch_infos.size()); put(g.delta_log_G); put(g.log_alpha_mut); put((double)g.X); put((double)g.S); put(g.t_P);
    for (auto& b : g.branch_infos) {
      put((double)b.A); put((double)b.B); put(b.is_open ? 1.0 : 0.0); put(b.T_to_X); put(b.partial_lambda_at_A); put(b.partial_lambda_at_X);
      put((double)b.warm_sites.v.size()); for (auto& [s0, s1] : b.warm_sites.v) { put((double)s0); put((double)s1); }
      put((double)b.hot_sites.v.size()); for (auto& [s0, s1] : b.hot_sites.v) { put((double)s0); put((double)s1); }
      put((double)b.hot_muts_to_X.size()); for (auto& m : b.hot_muts_to_X) { put((double)m.site); put((double)m.from); put((double)m.to); put(m.t); }
      put((double)b.hot_deltas_to_X.size()); for (auto& [l, d] : b.hot_deltas_to_X) { put((double)l); put((double)d.from); put((double)d.to); }
    }
  };
  put(0.0); put(mode == 3 ? 2.0 : 1.0);
  Spr_graft g0 = spr.analyze_graft(X);
  put_graft(g0);
  if (mode >= 1) {
    spr.peel_graft(g0);
    put((double)spr.count_min_mutations(g0)); put((double)spr.count_closed_mutations(g0));
    auto d = spr.summarize_closed_mutations(g0);
    put((double)d.size()); for (auto& [l, sd] : d) { put((double)l); put((double)sd.from); put((double)sd.to); }
  }
  if (mode == 2) spr.apply_graft(g0);
  if (mode == 3) {
    spr.move(X, new_sibling, new_t_P);
    Spr_graft g1 = spr.propose_new_graft(X, pt.rng);
    put_graft(g1);
    spr.apply_graft(g1);
    sr.log_G -= g0.delta_log_G; sr.log_G += g1.delta_log_G;
  }
  *out_len = n;
  ORC_CATCH
}
/* sample_mutational_history + adjust_mutational_history per (branch[i], t_end[i]), in the layout of emat_debug_sample_history */
int orc_debug_sample_history(orc_engine* e, int part_id, int n, const int* branch, const double* t_end, const uint8_t* start_seq, double T, double mu,
                             int* counts, double* muts, int muts_cap, int* num_muts) {
  ORC_TRY
  Part& pt = *e->parts.at(part_id);
  const Phylo_tree& tree = pt.subrun->tree;
  const int L = tree.num_sites();
  int written = 0;
  for (int i = 0; i < n; ++i) {
    Phylo_tree_loc end_loc{branch[i], t_end[i]};
    Site_deltas deltas;
    for (int l = 0; l < L; ++l) { State es = calc_site_state_at(tree, end_loc, l); if (es != (State)start_seq[l]) deltas.insert({l, Site_delta{(State)start_seq[l], es}}); }
    auto h = sample_mutational_history(L, T, mu, deltas, pt.rng);
    adjust_mutational_history(h, deltas, tree, end_loc);
    counts[i] = (int)h.size();
    for (auto& m : h) { if (written < muts_cap) { double* o = muts + 4 * (size_t)written; o[0] = m.site; o[1] = m.from; o[2] = m.to; o[3] = m.t; } ++written; }
  }
  *num_muts = written;
  ORC_CATCH
}
/* One Tree_editing_session on node X of one part, steps as in emat_debug_edit (0 slide(t), 1 hop_up, 2 flip, 3 hop_down(node)), then end();
 * *lambda_dev = largest |lambda_i kept by the session - lambda_i recomputed|, *missing_bad = nodes whose missing-site count is off. */
int orc_debug_edit(orc_engine* e, int part_id, int X, int n_ops, const int* op_kind, const int* op_node, const double* op_t, double* lambda_dev, int* missing_bad) {
  ORC_TRY
  Subrun& sr = *e->parts.at(part_id)->subrun;
  sr.ref_cum_Q_l = calc_cum_Q_l_for_sequence(sr.tree.ref_sequence, sr.evo);
  sr.lambda_i = calc_lambda_i(sr.tree, sr.evo, sr.ref_cum_Q_l);
  sr.num_sites_missing = calc_num_sites_missing_at_every_node(sr.tree);
  {
    Tree_editing_session edit{sr.tree, X, sr.evo, sr.lambda_i, sr.ref_cum_Q_l, sr.num_sites_missing};
    for (int i = 0; i < n_ops; ++i) {
      if (op_kind[i] == 0) edit.slide_P_along_branch(op_t[i]);
      else if (op_kind[i] == 1) edit.hop_up();
      else if (op_kind[i] == 2) edit.flip();
      else edit.hop_down(op_node[i]);
    }
    edit.end();
  }
  auto li = calc_lambda_i(sr.tree, sr.evo, sr.ref_cum_Q_l);
  auto nm = calc_num_sites_missing_at_every_node(sr.tree);
  double dev = 0.0; int bad = 0;
  for (int n = 0; n < sr.tree.size(); ++n) { dev = std::max(dev, std::fabs(li[n] - sr.lambda_i[n])); if (nm[n] != sr.num_sites_missing[n]) ++bad; }
  *lambda_dev = dev; *missing_bad = bad;
  ORC_CATCH
}
/* log G of one part as it stands (incrementally maintained) and recomputed from scratch, without touching the coalescent prior */
int orc_part_log_G(orc_engine* e, int part_id, double* incremental, double* from_scratch) {
  ORC_TRY
  Subrun& sr = *e->parts.at(part_id)->subrun;
  *incremental = sr.log_G;
  auto li = sr.lambda_i;
  sr.ref_freqs = calc_state_frequencies_per_partition_of(sr.tree.ref_sequence, sr.evo);
  sr.ref_cum_Q_l = calc_cum_Q_l_for_sequence(sr.tree.ref_sequence, sr.evo);
  sr.lambda_i = calc_lambda_i(sr.tree, sr.evo, sr.ref_cum_Q_l);
  *from_scratch = sr.calc_cur_log_G();
  sr.lambda_i = li;
  ORC_CATCH
}

int orc_get_totals(orc_engine* e, double* log_G, double* log_aug) {
  ORC_TRY
  double g = 0.0, a = 0.0;
  for (auto& pt : e->parts) { pt->subrun->validate_derived_quantities(); g += pt->subrun->log_G; a += pt->subrun->log_augmented_coalescent_prior; }
  if (log_G) *log_G = g; if (log_aug) *log_aug = a;
  ORC_CATCH
}

static void tree_sizes(const Phylo_tree& t, int* num_nodes, int* num_muts, int* num_intervals, int* num_from_states) {
  int nm = 0, ni = 0, nf = 0;
  for (int i = 0; i < t.size(); ++i) { nm += (int)t.at(i).mutations.size(); ni += t.at(i).missations.num_intervals(); nf += (int)t.at(i).missations.from_states.size(); }
  *num_nodes = t.size(); *num_muts = nm; *num_intervals = ni; *num_from_states = nf;
}
static void tree_to_flat(const Phylo_tree& t, emat_flat_tree* out) {
  ORC_CHECK(out->num_nodes == t.size());
  out->root = t.root;
  int km = 0, ki = 0, kf = 0;
  out->mut_offset[0] = out->miss_offset[0] = out->mfs_offset[0] = 0;
  for (int i = 0; i < t.size(); ++i) {
    auto& nd = t.at(i);
    out->parent[i] = nd.parent; out->child0[i] = nd.children[0]; out->child1[i] = nd.children[1];
    out->t[i] = nd.t; out->t_min[i] = nd.t_min; out->t_max[i] = nd.t_max;
    for (auto& m : nd.mutations) { ORC_CHECK(km < out->cap_muts); out->mut_site[km] = m.site; out->mut_from[km] = m.from; out->mut_to[km] = m.to; out->mut_t[km] = m.t; ++km; }
    for (auto& [s, en] : nd.missations.intervals.v) { ORC_CHECK(ki < out->cap_intervals); out->miss_start[ki] = s; out->miss_end[ki] = en; ++ki; }
    for (auto& [l, s] : nd.missations.from_states) { ORC_CHECK(kf < out->cap_from_states); out->mfs_site[kf] = l; out->mfs_state[kf] = s; ++kf; }
    out->mut_offset[i + 1] = km; out->miss_offset[i + 1] = ki; out->mfs_offset[i + 1] = kf;
  }
}
int orc_part_get_sizes(orc_engine* e, int part_id, int* num_nodes, int* num_muts, int* num_intervals, int* num_from_states) {
  tree_sizes(e->parts.at(part_id)->subrun->tree, num_nodes, num_muts, num_intervals, num_from_states);
  return EMAT_OK;
}
int orc_part_download(orc_engine* e, int part_id, emat_flat_tree* out) {
  ORC_TRY
  tree_to_flat(e->parts.at(part_id)->subrun->tree, out);
  ORC_CATCH
}
int orc_part_get_derived(orc_engine* e, int part_id, double* lambda_i, int* num_missing, double* log_G, double* log_aug) {
  ORC_TRY
  auto& sr = *e->parts.at(part_id)->subrun;
  sr.validate_derived_quantities();
  if (lambda_i) std::copy(sr.lambda_i.begin(), sr.lambda_i.end(), lambda_i);
  if (num_missing) std::copy(sr.num_sites_missing.begin(), sr.num_sites_missing.end(), num_missing);
  if (log_G) *log_G = sr.log_G;
  if (log_aug) *log_aug = sr.log_augmented_coalescent_prior;
  ORC_CATCH
}
int orc_part_get_coalescent(orc_engine* e, int part_id, int* num_cells, double* k_bar_p, double* k_tw_p, double* k_tw,
                            double* popsize_bar, int* num_active, double* t_ref, double* t_step) {
  ORC_TRY
  auto& cp = e->coal_parts.at(part_id);
  int n = (int)cp.k_bar_p.size();
  ORC_CHECK(*num_cells >= n);
  *num_cells = n;
  for (int i = 0; i < n; ++i) {
    if (k_bar_p) k_bar_p[i] = cp.k_bar_p[i];
    if (k_tw_p) k_tw_p[i] = cp.k_twiddle_bar_p[i];
    if (k_tw) k_tw[i] = cp.k_twiddle_bar[i];
    if (popsize_bar) popsize_bar[i] = cp.popsize_bar[i];
    if (num_active) num_active[i] = cp.num_active_parts[i];
  }
  if (t_ref) *t_ref = cp.t_ref;
  if (t_step) *t_step = cp.t_step;
  ORC_CATCH
}
int orc_part_get_stats(orc_engine* e, int part_id, emat_part_stats* out) {
  auto& pt = *e->parts.at(part_id);
  std::memset(out, 0, sizeof *out);
  out->num_nodes = pt.subrun->tree.size();
  out->moves_done = pt.subrun->moves_done;
  for (int k = 0; k < 5; ++k) { out->proposed[k] = pt.subrun->proposed[k]; out->accepted[k] = pt.subrun->accepted[k]; }
  out->rng_draws = (int64_t)pt.rng.counter;
  return EMAT_OK;
}
int orc_part_get_trace(orc_engine* e, int part_id, int* num_moves, double* trace) {
  auto& tr = e->parts.at(part_id)->subrun->trace;
  int n = std::min<int>(*num_moves, (int)tr.size());
  for (int i = 0; i < n; ++i) { trace[4 * i] = tr[i].kind; trace[4 * i + 1] = tr[i].node; trace[4 * i + 2] = tr[i].accepted; trace[4 * i + 3] = tr[i].log_mh; }
  *num_moves = n;
  return EMAT_OK;
}
// Returns 0 and an empty message when the part passes the reference's debug invariants.
int orc_part_check(orc_engine* e, int part_id, char* msg, int msg_cap) {
  auto& sr = *e->parts.at(part_id)->subrun;
  std::string m;
  try { m = check_phylo_tree_integrity(sr.tree); if (m.empty()) { sr.validate_derived_quantities(); m = sr.check_derived_quantities(); } }
  catch (const std::exception& ex) { m = ex.what(); }
  std::snprintf(msg, msg_cap, "%s", m.c_str());
  return m.empty() ? 0 : 1;
}

// scalar helpers exposed for golden-vector tests
double orc_gamma_q(double a, double x) { return gamma_q(a, x); }
double orc_gamma_q_inv(double a, double q) { return safe_gamma_q_inv(a, q); }
double orc_pop_at_time(const emat_pop_model* pm, double t) { return make_pop_model(*pm)->pop_at_time(t); }
double orc_pop_integral(const emat_pop_model* pm, double a, double b) { return make_pop_model(*pm)->pop_integral(a, b); }
double orc_intensity_integral(const emat_pop_model* pm, double a, double b) { return make_pop_model(*pm)->intensity_integral(a, b); }
/* Interval-set algebra of the oracle on plain arrays (pairs start, end).  op 0: insert the pairs of `a` one by one into an
 * empty set (Interval_set::insert, interval_set.h:96-125); 1 merge, 2 intersect, 3 subtract (a, b already valid sets);
 * 4 is_subset_of, 5 contains (site b[0]), 6 interval_sets_intersect -> out[0] = 0 / 1.  Returns the number of pairs written. */
int orc_interval_op(int op, const int* a, int na, const int* b, int nb, int* out, int cap) {
  Interval_set A, B, R;
  if (op == 0) { for (int i = 0; i < na; ++i) A.insert(Site_interval{a[2 * i], a[2 * i + 1]}); R = A; }
  else {
    for (int i = 0; i < na; ++i) A.v.push_back({a[2 * i], a[2 * i + 1]});
    if (op != 5) for (int i = 0; i < nb; ++i) B.v.push_back({b[2 * i], b[2 * i + 1]});
    if (op == 1) merge_interval_sets(R, A, B);
    else if (op == 2) intersect_interval_sets(R, A, B);
    else if (op == 3) subtract_interval_sets(R, A, B);
    else { if (cap < 1) return -1; out[0] = op == 4 ? interval_set_is_subset_of(A, B) : op == 5 ? A.contains(b[0]) : interval_sets_intersect(A, B); return 1; }
  }
  if ((int)R.v.size() > cap) return -1;
  for (size_t i = 0; i < R.v.size(); ++i) { out[2 * i] = R.v[i].first; out[2 * i + 1] = R.v[i].second; }
  return (int)R.v.size();
}

void orc_rng_block(uint64_t key, uint64_t counter, uint32_t out[4]) { Rng::philox4x32_10(counter, key, out); }


// ---- orc_run.hpp: Run::repartition / reassemble / normalize_root and tree_partitioning.h -----------------------------------------
struct orc_run { Run run; std::vector<Phylo_tree> subrun_trees; std::string last_error; };
#define RUN_TRY try {
#define RUN_CATCH } catch (const std::exception& ex) { r->last_error = ex.what(); return EMAT_ERR_INTERNAL; } return EMAT_OK;

int orc_run_create(const emat_flat_tree* tree, const uint8_t* ref, int num_sites, uint64_t seed, int num_parts, orc_run** out) {
  try {
    std::vector<State> rs(ref, ref + num_sites);
    *out = new orc_run{Run(tree_from_flat(*tree, rs), seed, num_parts), {}, {}};
  } catch (const std::exception&) { return EMAT_ERR_INTERNAL; }
  return EMAT_OK;
}
int orc_run_destroy(orc_run* r) { delete r; return EMAT_OK; }
const char* orc_run_last_error(orc_run* r) { return r->last_error.c_str(); }
int orc_run_repartition(orc_run* r) {
  RUN_TRY
  r->run.repartition();
  r->subrun_trees = r->run.subtrees;      // until orc_run_part_put says otherwise the Subruns made no move
  RUN_CATCH
}
int orc_run_num_parts(orc_run* r, int* num_parts, int* root_part) { *num_parts = (int)r->run.tree_partition.parts.size(); *root_part = r->run.tree_partition.root_part_index; return EMAT_OK; }
// the stencil in use is not kept by Run; the cut points of the parts are (Partition_part_info)
int orc_run_part_sizes(orc_run* r, int p, int* num_nodes, int* num_muts, int* num_intervals, int* num_from_states) {
  RUN_TRY
  tree_sizes(r->run.subtrees.at(p), num_nodes, num_muts, num_intervals, num_from_states);
  RUN_CATCH
}
// the subtree a Subrun is constructed from, the map subtree node -> tree node, and the part's cut point
int orc_run_part_get(orc_run* r, int p, emat_flat_tree* out, int* orig_tree_index, int* cut_point) {
  RUN_TRY
  tree_to_flat(r->run.subtrees.at(p), out);
  const auto& part = r->run.tree_partition.parts.at(p);
  if (orig_tree_index) for (int i = 0; i < part.size(); ++i) orig_tree_index[i] = part.nodes[i].orig_tree_index;
  if (cut_point) *cut_point = part.cut_point;
  RUN_CATCH
}
// the tree a Subrun holds after its moves
int orc_run_part_put(orc_run* r, int p, const emat_flat_tree* subtree) {
  RUN_TRY
  r->subrun_trees.at(p) = tree_from_flat(*subtree, r->run.tree.ref_sequence);
  RUN_CATCH
}
int orc_run_reassemble(orc_run* r) {
  RUN_TRY
  r->run.reassemble(r->subrun_trees);
  RUN_CATCH
}
int orc_run_normalize_root(orc_run* r) {
  RUN_TRY
  r->run.normalize_root();
  RUN_CATCH
}
int orc_run_tree_sizes(orc_run* r, int* num_nodes, int* num_muts, int* num_intervals, int* num_from_states) {
  tree_sizes(r->run.tree, num_nodes, num_muts, num_intervals, num_from_states); return EMAT_OK;
}
int orc_run_tree_get(orc_run* r, emat_flat_tree* out, uint8_t* ref_sequence) {
  RUN_TRY
  tree_to_flat(r->run.tree, out);
  if (ref_sequence) std::copy(r->run.tree.ref_sequence.begin(), r->run.tree.ref_sequence.end(), ref_sequence);
  RUN_CATCH
}
// the reference's integrity rules (phylo_tree.cpp:18-136) on the whole tree the Run holds
int orc_run_tree_check(orc_run* r, char* msg, int msg_cap) {
  std::string m;
  try { m = check_phylo_tree_integrity(r->run.tree); } catch (const std::exception& ex) { m = ex.what(); }
  std::snprintf(msg, msg_cap, "%s", m.c_str());
  return m.empty() ? 0 : 1;
}

// ---- orc_build.hpp: the UShER-like initial-tree builder ---------------------------------------------------------------------------
struct orc_build { std::vector<State> ref; std::vector<Tip_desc> descs; Phylo_tree tree; std::string last_error; };
#define BUILD_TRY try {
#define BUILD_CATCH } catch (const std::exception& ex) { r->last_error = ex.what(); return EMAT_ERR_INTERNAL; } return EMAT_OK;
static std::vector<Tip_desc> descs_from_c(const emat_tip_descs& td, const std::vector<State>& ref) {
  std::vector<Tip_desc> out(td.num_tips);
  for (int i = 0; i < td.num_tips; ++i) {
    out[i].t_min = td.t_min[i]; out[i].t_max = td.t_max[i];
    for (int k = td.delta_offset[i]; k < td.delta_offset[i + 1]; ++k) out[i].seq_deltas.push_back(Seq_delta(td.delta_site[k], ref.at(td.delta_site[k]), td.delta_to[k]));
    for (int k = td.miss_offset[i]; k < td.miss_offset[i + 1]; ++k) out[i].missations.intervals.v.push_back({td.miss_start[k], td.miss_end[k]});
  }
  return out;
}
int orc_build_create(const uint8_t* ref, int num_sites, orc_build** out) { *out = new orc_build; (*out)->ref.assign(ref, ref + num_sites); return EMAT_OK; }
int orc_build_destroy(orc_build* r) { delete r; return EMAT_OK; }
const char* orc_build_last_error(orc_build* r) { return r->last_error.c_str(); }
// descriptors of the tips of a tree (tips in node order): sizes first (null arrays), then the arrays
int orc_build_descs_from_tree(orc_build* r, const emat_flat_tree* tree, int* num_tips, int* num_deltas, int* num_intervals) {
  BUILD_TRY
  r->descs = tip_descs_of(tree_from_flat(*tree, r->ref));
  int nd = 0, ni = 0;
  for (auto& d : r->descs) { nd += (int)d.seq_deltas.size(); ni += d.missations.num_intervals(); }
  *num_tips = (int)r->descs.size(); *num_deltas = nd; *num_intervals = ni;
  BUILD_CATCH
}
int orc_build_descs_get(orc_build* r, float* t_min, float* t_max, int* delta_offset, int* delta_site, uint8_t* delta_to, int* miss_offset, int* miss_start, int* miss_end) {
  BUILD_TRY
  int kd = 0, ki = 0; delta_offset[0] = miss_offset[0] = 0;
  for (size_t i = 0; i < r->descs.size(); ++i) {
    auto& d = r->descs[i];
    t_min[i] = d.t_min; t_max[i] = d.t_max;
    for (auto& s : d.seq_deltas) { delta_site[kd] = s.site; delta_to[kd] = s.to; ++kd; }
    for (auto& [a, b] : d.missations.intervals.v) { miss_start[ki] = a; miss_end[ki] = b; ++ki; }
    delta_offset[i + 1] = kd; miss_offset[i + 1] = ki;
  }
  BUILD_CATCH
}
int orc_build_usher_like(orc_build* r, const emat_tip_descs* td, uint64_t seed, int* num_nodes, int* num_muts, int* num_intervals, int* num_from_states) {
  BUILD_TRY
  r->descs = descs_from_c(*td, r->ref);
  Rng rng; rng.key = seed;
  r->tree = build_usher_like_tree(r->ref, r->descs, rng);
  tree_sizes(r->tree, num_nodes, num_muts, num_intervals, num_from_states);
  BUILD_CATCH
}
// the reference's DEFAULT builder (orc_utree.hpp: guide tree -> refinement rounds -> SPR refinement -> OLS rooting -> phylo tree);
// report[0..3] = deltas of the guide tree, after the refinement rounds, after SPR refinement, the rooting method (0 regression, 1 midpoint)
int orc_build_default(orc_build* r, const emat_tip_descs* td, uint64_t seed, int* num_nodes, int* num_muts, int* num_intervals, int* num_from_states, int* report) {
  BUILD_TRY
  r->descs = descs_from_c(*td, r->ref);
  Rng rng; rng.key = seed;
  Initial_tree_report rep;
  r->tree = build_initial_phylo_tree(r->ref, r->descs, rng, &rep);
  if (report) { report[0] = rep.guide_deltas; report[1] = rep.refined_deltas; report[2] = rep.spr_deltas; report[3] = rep.rooting.method == Rooting_method::regression ? 0 : 1; }
  tree_sizes(r->tree, num_nodes, num_muts, num_intervals, num_from_states);
  BUILD_CATCH
}
int orc_build_tree_get(orc_build* r, emat_flat_tree* out) {
  BUILD_TRY
  tree_to_flat(r->tree, out);
  BUILD_CATCH
}
// the reference sequence the last built tree is written against: the default builder re-references its tree to the root sequence
// (phylo_tree.cpp:309-322), the UShER-like one leaves the sequence it was given
int orc_build_tree_ref(orc_build* r, uint8_t* out) {
  BUILD_TRY
  ORC_CHECK(r->tree.ref_sequence.size() == r->ref.size());
  std::copy(r->tree.ref_sequence.begin(), r->tree.ref_sequence.end(), out);
  BUILD_CATCH
}
// orc_build_check for a tree written against `tree_ref` rather than the sequence the descriptors are deltas to
int orc_build_check_with_ref(orc_build* r, const emat_flat_tree* tree, const uint8_t* tree_ref, const emat_tip_descs* td, char* msg, int msg_cap) {
  std::string m;
  try {
    auto t = tree_from_flat(*tree, std::vector<State>(tree_ref, tree_ref + r->ref.size()));
    m = check_phylo_tree_integrity(t);
    if (m.empty()) m = check_phylo_tree_matches_tip_descs(t, r->ref, descs_from_c(*td, r->ref));
  } catch (const std::exception& ex) { m = ex.what(); }
  std::snprintf(msg, msg_cap, "%s", m.c_str());
  return m.empty() ? 0 : 1;
}
// the reference's closing checks of the builder on ANY tree (e.g. the device's): integrity rules + every tip reproduces its descriptor
int orc_build_check(orc_build* r, const emat_flat_tree* tree, const emat_tip_descs* td, char* msg, int msg_cap) {
  std::string m;
  try {
    auto t = tree_from_flat(*tree, r->ref);
    m = check_phylo_tree_integrity(t);
    if (m.empty()) m = check_phylo_tree_matches_tip_descs(t, r->ref, descs_from_c(*td, r->ref));
  } catch (const std::exception& ex) { m = ex.what(); }
  std::snprintf(msg, msg_cap, "%s", m.c_str());
  return m.empty() ? 0 : 1;
}
}  // extern "C"
