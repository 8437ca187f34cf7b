// emat_host_model.hpp -- host-side pieces of the engine that the reference also keeps on the host:
//   * the counter-based RNG shared with the kernels (one Philox4x32-10 block per draw);
//   * population models (reference core/pop_model.cpp:18-145, 181-204, 247-330, 525-560);
//   * make_very_scalable_coalescent_prior_parts (reference core/very_scalable_coalescent.cpp:85-232),
//     which Run::reset_very_scalable_coalescent_parts (core/run.cpp:277-293) executes on the host at every
//     push of global parameters.
// Product code (NOT the test oracle): it feeds the HIP engine's slabs.
#ifndef EMAT_HOST_MODEL_HPP_
#define EMAT_HOST_MODEL_HPP_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <stdexcept>
#include <vector>

#include "flat_tree.hpp"
#include "host_parallel.hpp"

namespace emat {

struct HostRng {
  uint64_t key = 0, counter = 0;
  static void philox(uint64_t ctr, uint64_t key, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0, k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    for (int r = 0; r < 10; ++r) {
      uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
      uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3; k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
  uint64_t spare = 0; bool has_spare = false;   // second 64-bit half of the last block, not yet consumed
  uint64_t next64() {
    if (has_spare) { has_spare = false; return spare; }
    uint32_t w[4]; philox(counter++, key, w);
    spare = (uint64_t)w[2] | ((uint64_t)w[3] << 32); has_spare = true;
    return (uint64_t)w[0] | ((uint64_t)w[1] << 32);
  }
  double gaussian(double mean, double sigma) {
    uint64_t a = next64(), b = next64();
    double u1 = ((double)(a >> 11) + 1.0) * 0x1.0p-53, u2 = (double)(b >> 11) * 0x1.0p-53;
    return mean + sigma * (std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586476925 * u2));
  }
};

struct HostPopModel {
  int kind = EMAT_POP_CONST;
  double p[4] = {1, 0, 0, 0};
  double t_c = std::numeric_limits<double>::quiet_NaN();
  int skygrid_type = 1;
  std::vector<double> x, gamma;

  static HostPopModel from_c(const emat_pop_model& m) {
    HostPopModel h; h.kind = m.kind;
    for (int i = 0; i < 4; ++i) h.p[i] = m.p[i];
    if (m.kind == EMAT_POP_CONST) { if (!(m.p[0] > 0.0)) throw std::invalid_argument("Population size should be positive"); }
    else if (m.kind == EMAT_POP_EXP) {
      if (!(m.p[1] > 0.0)) throw std::invalid_argument("Initial effective population size should be positive");
      if (m.p[3] < 0.0) throw std::invalid_argument("Minimum effective population size should be non-negative");
      if (m.p[3] > 0.0 && m.p[2] != 0.0) h.t_c = m.p[0] + std::log(m.p[3] / m.p[1]) / m.p[2];
    } else if (m.kind == EMAT_POP_SKYGRID) {
      if (m.skygrid_num_knots < 2) throw std::invalid_argument("Skygrid_pop_model needs at least two knots");
      h.skygrid_type = m.skygrid_type;
      h.x.assign(m.skygrid_x, m.skygrid_x + m.skygrid_num_knots);
      h.gamma.assign(m.skygrid_gamma, m.skygrid_gamma + m.skygrid_num_knots);
      for (size_t i = 0; i + 1 < h.x.size(); ++i) if (!(h.x[i] < h.x[i + 1])) throw std::invalid_argument("Skygrid_pop_model needs strictly increasing knot times");
    } else throw std::invalid_argument("unknown population model kind");
    return h;
  }
  int interval(double t) const { return (int)(std::lower_bound(x.begin(), x.end(), t) - x.begin()); }
  double exp_int(double a, double b) const { return p[1] / p[2] * std::exp(p[2] * (a - p[0])) * std::expm1(p[2] * (b - a)); }
  double pop_integral(double a, double b) const {
    if (kind == EMAT_POP_CONST) return (b - a) * p[0];
    if (kind == EMAT_POP_EXP) {
      const double t0 = p[0], n0 = p[1], g = p[2], mp = p[3];
      if (mp == 0.0) return g == 0.0 ? (b - a) * n0 : exp_int(a, b);
      if (g == 0.0) return (b - a) * std::max(mp, n0);
      if (g > 0.0) {
        if (b <= t_c) return (b - a) * mp;
        if (a >= t_c) return exp_int(a, b);
        return (t_c - a) * mp + n0 / g * std::exp(g * (t_c - t0)) * std::expm1(g * (b - t_c));
      }
      if (a >= t_c) return (b - a) * mp;
      if (b <= t_c) return exp_int(a, b);
      return n0 / g * std::exp(g * (a - t0)) * std::expm1(g * (t_c - a)) + (b - t_c) * mp;
    }
    const int M = (int)x.size() - 1;
    int ka = interval(a), kb = interval(b);
    int kka = std::max(ka - 1, 0), kkb = std::min(kb, M);
    double bias = -std::numeric_limits<double>::infinity();
    for (int k = kka; k <= kkb; ++k) bias = std::max(bias, gamma[k]);
    double result = 0.0;
    for (int k = ka; k <= kb; ++k) {
      double lo = k > 0 ? std::max(a, x[k - 1]) : a, hi = k <= M ? std::min(b, x[k]) : b;
      if (k == 0) result += std::exp(-bias + gamma[0]) * (hi - lo);
      else if (k == M + 1) result += std::exp(-bias + gamma[M]) * (hi - lo);
      else if (skygrid_type == 1) result += std::exp(-bias + gamma[k]) * (hi - lo);
      else if (gamma[k] == gamma[k - 1]) result += std::exp(-bias + gamma[k]) * (hi - lo);
      else {
        double c_lo = (lo - x[k - 1]) / (x[k] - x[k - 1]), c_hi = (hi - x[k - 1]) / (x[k] - x[k - 1]);
        double G_lo = (1 - c_lo) * gamma[k - 1] + c_lo * gamma[k], G_hi = (1 - c_hi) * gamma[k - 1] + c_hi * gamma[k];
        double D = G_hi - G_lo;
        result += (D == 0.0) ? std::exp(-bias + G_lo) * (hi - lo) : std::exp(-bias + G_lo) * (hi - lo) * (std::expm1(D) / D);
      }
    }
    return std::exp(std::log(result) + bias);
  }
};

// One part's share of the augmented coalescent prior, restricted to the window of cells it can touch.
struct HostCoalPart {
  int cell_first = 0;                 // first stored cell
  int n_cells_total = 0;              // length of the logical k_bar_p vector (last_cell + 1)
  std::vector<double> k_bar_p, k_twiddle_bar_p, k_twiddle_bar, popsize_bar;   // window [cell_first, n_cells_total)
  std::vector<int32_t> num_active_parts;
  double t_ref = 0.0, t_step = 1.0;
};

namespace coal_detail {
inline int cell_for(double t, double t_ref, double t_step) { return (int)std::floor((t_ref - t) / t_step); }
inline double cell_ubound(int c, double t_ref, double t_step) { return t_ref - t_step * c; }
inline double cell_lbound(int c, double t_ref, double t_step) { return cell_ubound(c, t_ref, t_step) - t_step; }
inline void add_interval(double ts, double te, double dk, std::vector<double>& k, double t_ref, double t_step) {   // cpp:37-79
  if (ts < te) std::swap(ts, te);
  int cs = cell_for(ts, t_ref, t_step);
  int ce = (int)k.size() - 1;
  if (te != cell_lbound(ce, t_ref, t_step)) ce = cell_for(te, t_ref, t_step);
  if (cs < 0 || ce >= (int)k.size() || cs > ce) throw std::runtime_error("coalescent grid: interval outside the part's cells");
  if (cs == ce) k[cs] += dk * (ts - te) / t_step;
  else {
    k[cs] += dk * (ts - cell_lbound(cs, t_ref, t_step)) / t_step;
    k[ce] += dk * (cell_ubound(ce, t_ref, t_step) - te) / t_step;
    for (int i = cs + 1; i < ce; ++i) k[i] += dk;
  }
}
}  // namespace coal_detail

// very_scalable_coalescent.cpp:85-232, split into the stages between which a multi-GPU run exchanges data
// (SURVEY 8e): (1) local time range -> all-reduce min/max; (2) local k_bar / num_active_parts contributions ->
// all-reduce sum; (3) Gaussian k_twiddle_bar_p draws from each part's own stream -> all-reduce sum of
// k_twiddle_bar; (4) window extraction.  A single process simply runs the four stages back to back.
struct CoalBuilder {
  HostPopModel pop;
  double t_step = 1.0;
  std::vector<const FlatTree*> trees;
  std::vector<HostRng*> rngs;
  int root_local = -1;               // index (in `trees`) of the part holding the run's root, or -1 if it lives on another rank
  std::vector<double> tmin, tmax;
  std::vector<double> tmax_exact;    // latest node time of each part in full precision (tips' t_min / t_max are floats)
  double t_ref = 0.0, all_min = 0.0;
  int num_cells = 0;
  std::vector<int> fc, lc;           // the reference's [first_cell, last_cell] of each part: where it is "active" and draws k_twiddle_bar_p
  std::vector<int> wf;               // first cell each part STORES: <= fc (see local_grid)
  std::vector<std::vector<double>> kbar_p, ktw_p;
  std::vector<double> popsize, k_bar, k_tw;
  std::vector<int32_t> num_active;

  void add_part(const FlatTree* t, HostRng* rng, bool is_root) { if (is_root) root_local = (int)trees.size(); trees.push_back(t); rngs.push_back(rng); }
  // stage 1
  void local_range(double& lo, double& hi) {
    using namespace coal_detail;
    const int P = (int)trees.size();
    tmin.assign(P, std::numeric_limits<double>::max()); tmax.assign(P, -std::numeric_limits<double>::max());
    tmax_exact.assign(P, -std::numeric_limits<double>::max());
    for (int p = 0; p < P; ++p) {
      const FlatTree& st = *trees[p];
      for (int n = 0; n < st.num_nodes(); ++n) {
        bool tip = st.is_tip(n);
        tmin[p] = std::min(tmin[p], tip ? (double)st.t_min[n] : st.t[n]);
        tmax[p] = std::max(tmax[p], tip ? (double)st.t_max[n] : st.t[n]);
        tmax_exact[p] = std::max(tmax_exact[p], st.t[n]);
      }
    }
    lo = std::numeric_limits<double>::max(); hi = -std::numeric_limits<double>::max();
    for (int p = 0; p < P; ++p) { lo = std::min(lo, tmin[p]); hi = std::max(hi, tmax[p]); }
  }
  // stage 2: with the global range known, this rank's contributions to k_bar and num_active_parts
  int set_range(double all_t_min, double all_t_max) {
    using namespace coal_detail;
    all_min = all_t_min; t_ref = all_t_max;
    if (root_local >= 0) tmin[root_local] = all_t_min;
    num_cells = cell_for(all_t_min, t_ref, t_step) + 1;
    return num_cells;
  }
  void local_grid(std::vector<double>& k_bar_local, std::vector<int32_t>& num_active_local) {
    using namespace coal_detail;
    const int P = (int)trees.size();
    k_bar_local.assign(num_cells, 0.0); num_active_local.assign(num_cells, 0);
    fc.assign(P, 0); lc.assign(P, 0); wf.assign(P, 0); kbar_p.assign(P, {}); ktw_p.assign(P, {});
    parallel_for(P, [&](int p) {   // per-part lineage counts: independent
      fc[p] = cell_for(tmax[p], t_ref, t_step); lc[p] = cell_for(tmin[p], t_ref, t_step);
      if (!(0 <= fc[p] && fc[p] <= lc[p] && lc[p] < num_cells)) throw std::runtime_error("coalescent grid: bad cell range");
      // The reference keeps every part's vectors from cell 0, so nothing stops a lineage from ending LATER than the
      // part's float-derived t_max: a frozen cut-point tip has t_max = (float)t, which may lie below its exact time t,
      // and when a cell boundary falls in between, the branch above it (and a displaced parent) reaches cell
      // first_cell - 1.  The stored window therefore starts at the cell of the part's latest EXACT node time when
      // that is earlier in the grid; activity counts and draws stay on the reference's [first_cell, last_cell].
      wf[p] = std::min(fc[p], std::max(0, cell_for(tmax_exact[p], t_ref, t_step)));
      kbar_p[p].assign(lc[p] + 1, 0.0); ktw_p[p].assign(lc[p] + 1, 0.0);
      const FlatTree& st = *trees[p];
      for (int n = 0; n < st.num_nodes(); ++n) if (n != st.root) add_interval(st.t[st.parent[n]], st.t[n], +1.0, kbar_p[p], t_ref, t_step);
      if (p == root_local) add_interval(cell_lbound(num_cells - 1, t_ref, t_step), st.t[st.root], +1.0, kbar_p[p], t_ref, t_step);
    });
    // reductions in part order (outside a part's window its counts are exactly zero: skipping them changes nothing)
    for (int p = 0; p < P; ++p) {
      for (int c = fc[p]; c <= lc[p]; ++c) num_active_local[c] += 1;
      for (int i = wf[p]; i <= lc[p]; ++i) k_bar_local[i] += kbar_p[p][i];   // the reference adds every stored cell (cpp:183-188)
      for (int i = 0; i < wf[p]; ++i) if (kbar_p[p][i] != 0.0) throw std::runtime_error("coalescent grid: lineage outside the part's window");
    }
  }
  // stage 3: draws, given the GLOBAL k_bar and num_active_parts; returns this rank's contribution to k_twiddle_bar
  void sample(const std::vector<double>& k_bar_global, const std::vector<int32_t>& num_active_global, std::vector<double>& k_tw_local) {
    using namespace coal_detail;
    const int P = (int)trees.size();
    k_bar = k_bar_global; num_active = num_active_global;
    if (num_active.back() == 0) throw std::runtime_error("coalescent grid: inactive final cell");
    popsize.assign(num_cells, 0.0);
    parallel_for(num_cells, [&](int i) { popsize[i] = pop.pop_integral(cell_lbound(i, t_ref, t_step), cell_ubound(i, t_ref, t_step)) / t_step; }, 16);
    k_tw_local.assign(num_cells, 0.0);
    parallel_for(P, [&](int p) {   // every part draws from its own stream
      for (int i = fc[p]; i <= lc[p]; ++i) {
        double mu = kbar_p[p][i] - k_bar[i] / num_active[i];
        double sigma = std::sqrt(popsize[i] / (num_active[i] * t_step));
        ktw_p[p][i] = rngs[p]->gaussian(mu, sigma);
      }
    });
    for (int p = 0; p < P; ++p) for (int i = fc[p]; i <= lc[p]; ++i) k_tw_local[i] += ktw_p[p][i];   // part order: reproducible sums
  }
  // stage 4
  std::vector<HostCoalPart> finish(const std::vector<double>& k_tw_global) {
    const int P = (int)trees.size();
    k_tw = k_tw_global;
    std::vector<HostCoalPart> out(P);
    parallel_for(P, [&](int p) {
      HostCoalPart& cp = out[p];
      cp.cell_first = wf[p]; cp.n_cells_total = lc[p] + 1; cp.t_ref = t_ref; cp.t_step = t_step;
      cp.k_bar_p.assign(kbar_p[p].begin() + wf[p], kbar_p[p].end());
      cp.k_twiddle_bar_p.assign(ktw_p[p].begin() + wf[p], ktw_p[p].end());   // zero outside [first_cell, last_cell], as in the reference
      cp.k_twiddle_bar.assign(k_tw.begin() + wf[p], k_tw.begin() + lc[p] + 1);
      cp.popsize_bar.assign(popsize.begin() + wf[p], popsize.begin() + lc[p] + 1);
      cp.num_active_parts.assign(num_active.begin() + wf[p], num_active.begin() + lc[p] + 1);
    });
    return out;
  }
};

// Single-process form (what Run::reset_very_scalable_coalescent_parts does, run.cpp:277-293).
inline std::vector<HostCoalPart> make_coalescent_parts(const std::vector<const FlatTree*>& subtrees, int root_part,
                                                       const HostPopModel& pop, std::vector<HostRng*>& rngs, double t_step) {
  CoalBuilder b; b.pop = pop; b.t_step = t_step;
  for (size_t p = 0; p < subtrees.size(); ++p) b.add_part(subtrees[p], rngs[p], (int)p == root_part);
  if (subtrees.empty()) return {};
  double lo, hi; b.local_range(lo, hi);
  b.set_range(lo, hi);
  std::vector<double> kb, kt; std::vector<int32_t> na;
  b.local_grid(kb, na);
  b.sample(kb, na, kt);
  return b.finish(kt);
}

}  // namespace emat
#endif  // EMAT_HOST_MODEL_HPP_
