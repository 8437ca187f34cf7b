// orc_coalescent.hpp -- CPU ORACLE (test infrastructure, NOT product code).
// Restates reference core/very_scalable_coalescent.cpp:14-459 and core/scalable_coalescent.cpp:34-251.
#ifndef ORC_COALESCENT_HPP_
#define ORC_COALESCENT_HPP_

#include <deque>
#include "orc_calc.hpp"

namespace orc {

static inline double square(double x) { return x * x; }

namespace vsc {
inline int cell_for(double t, double t_ref, double t_step) { return (int)std::floor((t_ref - t) / t_step); }   // :14-16
inline double cell_ubound(int cell, double t_ref, double t_step) { return t_ref - t_step * cell; }              // :18-20
inline double cell_lbound(int cell, double t_ref, double t_step) { return cell_ubound(cell, t_ref, t_step) - t_step; }
// very_scalable_coalescent.cpp:37-79
inline void add_interval(double t_start, double t_end, double delta_k, std::vector<double>& k, double t_ref, double t_step) {
  if (t_start < t_end) std::swap(t_start, t_end);
  int cell_start = cell_for(t_start, t_ref, t_step);
  ORC_CHECK(cell_start >= 0 && cell_start < (int)k.size());
  int cell_end = (int)k.size() - 1;
  if (t_end != cell_lbound(cell_end, t_ref, t_step)) {
    cell_end = cell_for(t_end, t_ref, t_step);
    ORC_CHECK(cell_end >= 0 && cell_end < (int)k.size());
  }
  if (cell_start == cell_end) {
    k[cell_start] += delta_k * (t_start - t_end) / t_step;
  } else {
    k[cell_start] += delta_k * (t_start - cell_lbound(cell_start, t_ref, t_step)) / t_step;
    k[cell_end] += delta_k * (cell_ubound(cell_end, t_ref, t_step) - t_end) / t_step;
    for (int i = cell_start + 1; i < cell_end; ++i) k[i] += delta_k;
  }
}
}  // namespace vsc

struct Very_scalable_coalescent_prior_part {
  std::shared_ptr<const Pop_model> pop_model;
  const Phylo_tree* subtree = nullptr;
  Rng* prng = nullptr;
  bool includes_tree_root = false;
  std::vector<double> k_bar_p, k_twiddle_bar_p, k_twiddle_bar, popsize_bar;
  std::vector<int> num_active_parts;
  double t_ref = 0.0, t_step = 1.0;

  void ensure_space(double t) {   // :259-299
    if (includes_tree_root) {
      int max_cell = vsc::cell_for(t, t_ref, t_step);
      for (int i = (int)popsize_bar.size(); i <= max_cell; ++i) {
        double lo = vsc::cell_lbound(i, t_ref, t_step), hi = vsc::cell_ubound(i, t_ref, t_step);
        popsize_bar.push_back(pop_model->pop_integral(lo, hi) / t_step);
        num_active_parts.push_back(1);
      }
      for (int i = (int)k_bar_p.size(); i <= max_cell; ++i) {
        double sigma = std::sqrt(popsize_bar[i] / t_step);
        double ktw = prng->gaussian(0.0, sigma);
        k_bar_p.push_back(1.0);
        k_twiddle_bar_p.push_back(ktw);
        k_twiddle_bar.push_back(ktw);
      }
    }
    int cell = vsc::cell_for(t, t_ref, t_step);
    ORC_CHECK(cell >= 0 && cell < (int)k_bar_p.size());
  }
  void coalescence_displaced(double old_t, double new_t) {   // :301-308
    ensure_space(new_t);
    vsc::add_interval(old_t, new_t, old_t <= new_t ? -1.0 : +1.0, k_bar_p, t_ref, t_step);
  }
  void tip_displaced(double old_t, double new_t) {           // :328-335
    ensure_space(new_t);
    vsc::add_interval(old_t, new_t, old_t <= new_t ? +1.0 : -1.0, k_bar_p, t_ref, t_step);
  }
  double cell_term(int i, double new_k, double old_k) const {
    return t_step / popsize_bar[i] * (
        +0.5 * (square(new_k) - square(old_k)) * num_active_parts[i]
        - (k_twiddle_bar_p[i] * num_active_parts[i] - k_twiddle_bar[i] + 0.5) * (new_k - old_k));
  }
  double calc_delta_partial_log_prior_on_add_interval(double min_t, double max_t, double delta_k) {   // :388-459
    { int c = vsc::cell_for(max_t, t_ref, t_step); ORC_CHECK(c >= 0 && c < (int)k_bar_p.size()); }
    ensure_space(min_t);
    double d = 0.0;
    if (min_t == max_t) return 0.0;
    ORC_CHECK(min_t < max_t);
    int cell_start = vsc::cell_for(max_t, t_ref, t_step);
    int cell_end = vsc::cell_for(min_t, t_ref, t_step);
    ORC_CHECK(0 <= cell_start && cell_start <= cell_end && cell_end < (int)k_bar_p.size());
    if (cell_start == cell_end) {
      int i = cell_start;
      double old_k = k_bar_p[i], new_k = old_k + delta_k * (max_t - min_t) / t_step;
      d -= cell_term(i, new_k, old_k);
    } else {
      int i = cell_start;
      double dt_start = max_t - vsc::cell_lbound(cell_start, t_ref, t_step);
      double dt_end = vsc::cell_ubound(cell_end, t_ref, t_step) - min_t;
      double old_k = k_bar_p[i], new_k = old_k + delta_k * dt_start / t_step;
      d -= cell_term(i, new_k, old_k);
      for (++i; i < cell_end; ++i) { old_k = k_bar_p[i]; new_k = old_k + delta_k; d -= cell_term(i, new_k, old_k); }
      old_k = k_bar_p[i]; new_k = old_k + delta_k * dt_end / t_step;
      d -= cell_term(i, new_k, old_k);
    }
    return d;
  }
  double calc_delta_partial_log_prior_after_displace_coalescence(double old_t, double new_t) {   // :310-326
    double d = (old_t <= new_t) ? calc_delta_partial_log_prior_on_add_interval(old_t, new_t, -1.0)
                                : calc_delta_partial_log_prior_on_add_interval(new_t, old_t, +1.0);
    d -= std::log(pop_model->pop_at_time(new_t) / pop_model->pop_at_time(old_t));
    return d;
  }
  double calc_delta_partial_log_prior_after_displace_tip(double old_t, double new_t) {   // :337-353
    return (old_t <= new_t) ? calc_delta_partial_log_prior_on_add_interval(old_t, new_t, +1.0)
                            : calc_delta_partial_log_prior_on_add_interval(new_t, old_t, -1.0);
  }
  double calc_partial_log_prior() const {   // :355-386
    double r = 0.0;
    for (int i = 0; i < (int)k_bar_p.size(); ++i)
      r -= t_step / popsize_bar[i] * (
          +0.5 * square(k_bar_p[i]) * num_active_parts[i]
          - (k_twiddle_bar_p[i] * num_active_parts[i] - k_twiddle_bar[i] + 0.5) * k_bar_p[i]);
    for (int n = 0; n < subtree->size(); ++n)
      if (subtree->at(n).is_inner_node()) r -= std::log(pop_model->pop_at_time(subtree->at(n).t));
    return r;
  }
};

// very_scalable_coalescent.cpp:85-232.  Gaussian draws come from each part's own RNG stream (the
// reference uses the part's std::mt19937, shared with its Subrun: run.cpp:112-114,182,287).
inline std::vector<Very_scalable_coalescent_prior_part> make_very_scalable_coalescent_prior_parts(
    const std::vector<const Phylo_tree*>& subtrees, int root_partition_index,
    std::shared_ptr<const Pop_model> pop_model, std::vector<Rng*>& prngs, double t_step) {
  std::vector<Very_scalable_coalescent_prior_part> result;
  if (subtrees.empty()) return result;
  int P = (int)subtrees.size();
  struct Info { double t_min, t_max; std::vector<double> k_bar_p, k_tw_p; };
  std::vector<Info> infos(P);
  for (int i = 0; i < P; ++i) {
    infos[i].t_min = std::numeric_limits<double>::max();
    infos[i].t_max = -std::numeric_limits<double>::max();
    auto& st = *subtrees[i];
    for (int n = 0; n < st.size(); ++n) {
      bool tip = st.at(n).is_tip();
      infos[i].t_min = std::min(infos[i].t_min, tip ? (double)st.at(n).t_min : st.at(n).t);
      infos[i].t_max = std::max(infos[i].t_max, tip ? (double)st.at(n).t_max : st.at(n).t);
    }
  }
  double all_t_min = infos[0].t_min, all_t_max = infos[0].t_max;
  for (auto& in : infos) { all_t_min = std::min(all_t_min, in.t_min); all_t_max = std::max(all_t_max, in.t_max); }
  infos[root_partition_index].t_min = all_t_min;
  double t_ref = all_t_max;
  int num_cells = vsc::cell_for(all_t_min, t_ref, t_step) + 1;
  std::vector<int> num_active(num_cells, 0);
  for (auto& in : infos) {
    int fc = vsc::cell_for(in.t_max, t_ref, t_step), lc = vsc::cell_for(in.t_min, t_ref, t_step);
    ORC_CHECK(0 <= fc && fc <= lc && lc < num_cells);
    for (int c = fc; c <= lc; ++c) num_active[c] += 1;
    in.k_bar_p.assign(lc + 1, 0.0);
    in.k_tw_p.assign(lc + 1, 0.0);
  }
  ORC_CHECK(num_active.back() != 0);
  for (int i = 0; i < P; ++i) {
    auto& st = *subtrees[i];
    for (int n = 0; n < st.size(); ++n)
      if (n != st.root) vsc::add_interval(st.at_parent_of(n).t, st.at(n).t, +1.0, infos[i].k_bar_p, t_ref, t_step);
  }
  vsc::add_interval(vsc::cell_lbound(num_cells - 1, t_ref, t_step), subtrees[root_partition_index]->at_root().t,
                    +1.0, infos[root_partition_index].k_bar_p, t_ref, t_step);
  std::vector<double> k_bar(num_cells, 0.0);
  for (auto& in : infos) for (int i = 0; i < (int)in.k_bar_p.size(); ++i) k_bar[i] += in.k_bar_p[i];
  std::vector<double> popsize_bar(num_cells, 0.0);
  for (int i = 0; i < num_cells; ++i)
    popsize_bar[i] = pop_model->pop_integral(vsc::cell_lbound(i, t_ref, t_step), vsc::cell_ubound(i, t_ref, t_step)) / t_step;
  for (int p = 0; p < P; ++p) {
    auto& in = infos[p];
    int fc = vsc::cell_for(in.t_max, t_ref, t_step), lc = vsc::cell_for(in.t_min, t_ref, t_step);
    for (int i = 0; i < (int)in.k_tw_p.size(); ++i) {
      if (fc <= i && i <= lc) {
        double mu = in.k_bar_p[i] - k_bar[i] / num_active[i];
        double sigma = std::sqrt(popsize_bar[i] / (num_active[i] * t_step));
        in.k_tw_p[i] = prngs[p]->gaussian(mu, sigma);
      } else in.k_tw_p[i] = 0.0;
    }
  }
  std::vector<double> k_tw(num_cells, 0.0);
  for (auto& in : infos) for (int i = 0; i < (int)in.k_tw_p.size(); ++i) k_tw[i] += in.k_tw_p[i];
  for (int p = 0; p < P; ++p) {
    Very_scalable_coalescent_prior_part part;
    part.pop_model = pop_model; part.subtree = subtrees[p]; part.prng = prngs[p];
    part.includes_tree_root = (p == root_partition_index);
    part.t_ref = t_ref; part.t_step = t_step;
    part.k_bar_p = std::move(infos[p].k_bar_p); part.k_twiddle_bar_p = std::move(infos[p].k_tw_p);
    part.k_twiddle_bar = k_tw; part.popsize_bar = popsize_bar; part.num_active_parts = num_active;
    result.push_back(std::move(part));
  }
  return result;
}

// ---- whole-tree grid prior (reference core/scalable_coalescent.cpp:34-251) ----------------------
struct Scalable_coalescent_prior {
  std::shared_ptr<const Pop_model> pop_model;
  struct Node_info { double t; bool is_tip; };
  std::vector<Node_info> node_infos;
  std::deque<double> k_bars, popsize_bars;
  double t_ref, t_step;
  int cells_before_t_ref = 0;

  Scalable_coalescent_prior(std::shared_ptr<const Pop_model> pm, int num_nodes, double t_ref_, double t_step_)
      : pop_model(std::move(pm)), node_infos(num_nodes, Node_info{t_ref_, false}), t_ref(t_ref_), t_step(t_step_) {}
  int cell_for(double t) const { return (int)std::floor((t - t_ref) / t_step) + cells_before_t_ref; }
  double cell_lbound(int c) const { return t_ref + ((c - cells_before_t_ref) * t_step); }
  double cell_ubound(int c) const { return cell_lbound(c) + t_step; }
  void reset(double ts) { t_step = ts; for (auto& ni : node_infos) ni.t = t_ref; k_bars.clear(); popsize_bars.clear(); cells_before_t_ref = 0; }
  void mark_as_tip(int n) { node_infos.at(n).is_tip = true; }
  void mark_as_coalescence(int n) { node_infos.at(n).is_tip = false; }
  void ensure_space(double t) {   // :48-86
    int cell = cell_for(t), tot = (int)k_bars.size();
    if (cell < 0) {
      int n = -cell; double t_max_new = cell_lbound(0);
      for (int i = 0; i < n; ++i) {
        k_bars.push_front(1.0);
        double t_min_new = t_max_new - t_step;
        double pb = pop_model->pop_integral(t_min_new, t_max_new) / t_step;
        if (pb == 0.0) pb = 1e-100;
        popsize_bars.push_front(pb);
        t_max_new = t_min_new;
      }
      cells_before_t_ref += n;
    } else if (cell >= tot) {
      int n = cell - tot + 1; double t_min_new = cell_ubound(tot - 1);
      for (int i = 0; i < n; ++i) {
        k_bars.push_back(0.0);
        double t_max_new = t_min_new + t_step;
        double pb = pop_model->pop_integral(t_min_new, t_max_new) / t_step;
        if (pb == 0.0) pb = 1e-100;
        popsize_bars.push_back(pb);
        t_min_new = t_max_new;
      }
    }
  }
  void add_interval(double ts, double te, double dk) {   // :88-116
    if (ts > te) std::swap(ts, te);
    ensure_space(ts); ensure_space(te);
    int cs = cell_for(ts), ce = cell_for(te);
    if (cs == ce) k_bars[cs] += dk * (te - ts) / t_step;
    else {
      k_bars[cs] += dk * (cell_ubound(cs) - ts) / t_step;
      k_bars[ce] += dk * (te - cell_lbound(ce)) / t_step;
      for (int i = cs + 1; i < ce; ++i) k_bars[i] += dk;
    }
  }
  void displace_tip(int n, double new_t) {               // :118-127
    double old_t = node_infos[n].t;
    if (old_t <= new_t) add_interval(old_t, new_t, +1.0); else add_interval(new_t, old_t, -1.0);
    node_infos[n].t = new_t;
  }
  void displace_coalescence(int n, double new_t) {       // :129-138
    double old_t = node_infos[n].t;
    if (old_t <= new_t) add_interval(old_t, new_t, -1.0); else add_interval(new_t, old_t, +1.0);
    node_infos[n].t = new_t;
  }
  double calc_log_prior() const {                         // :163-187
    double r = 0.0;
    for (size_t c = 0; c < k_bars.size(); ++c) r -= t_step * k_bars[c] * (k_bars[c] - 1) / (2.0 * popsize_bars[c]);
    for (auto& ni : node_infos) if (!ni.is_tip) r -= std::log(pop_model->pop_at_time(ni.t));
    return r;
  }
  double calc_delta_log_prior_after_displace_coalescence(int i, double new_t) {   // :189-251
    double d = 0.0, old_t = node_infos.at(i).t;
    if (old_t == new_t) return 0.0;
    bool adding = new_t < old_t;
    double min_t = std::min(old_t, new_t), max_t = std::max(old_t, new_t);
    ensure_space(old_t); ensure_space(new_t);
    int cs = cell_for(min_t), ce = cell_for(max_t);
    double sgn = adding ? +1.0 : -1.0;
    auto term = [&](int c, double dk) { double k = k_bars[c]; return t_step * ((k + dk) * (k + dk - 1) - k * (k - 1)) / (2 * popsize_bars[c]); };
    if (cs == ce) d -= term(cs, sgn * (max_t - min_t) / t_step);
    else {
      d -= term(cs, sgn * (cell_ubound(cs) - min_t) / t_step);
      for (int c = cs + 1; c < ce; ++c) d -= term(c, sgn);
      d -= term(ce, sgn * (max_t - cell_lbound(ce)) / t_step);
    }
    d -= std::log(pop_model->pop_at_time(new_t) / pop_model->pop_at_time(old_t));
    return d;
  }
};

}  // namespace orc
#endif  // ORC_COALESCENT_HPP_
