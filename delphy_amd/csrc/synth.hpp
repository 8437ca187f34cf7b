// synth.hpp -- seeded synthetic EMAT generator.
//
// The reference ships no datasets and no simulator, so the bench and the parity tests build their
// inputs here: a heterochronous coalescent tree, mutations dropped on its branches by a Gillespie
// process under HKY, per-tip gaps of missing data hoisted to the deepest branch they cover
// ("N-pruning", reference core/mutations.h:85-90, core/phylo_tree.cpp:57-111), all emitted directly
// in the flat struct-of-arrays format of include/emat_backend.h.
#ifndef EMAT_SYNTH_HPP_
#define EMAT_SYNTH_HPP_

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <unordered_map>
#include <utility>
#include <vector>

#include "flat_tree.hpp"

namespace emat {

struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
  double u01() { return (double)(next() >> 11) * 0x1.0p-53; }                       // [0,1)
  double u01_oc() { return ((double)(next() >> 11) + 1.0) * 0x1.0p-53; }            // (0,1]
  int below(int n) { return (int)(((unsigned __int128)next() * (uint64_t)n) >> 64); }
  double expo() { return -std::log(u01_oc()); }
  int poisson(double lam) {
    if (lam < 30.0) { double L = std::exp(-lam), p = 1.0; int k = 0; do { ++k; p *= u01_oc(); } while (p > L); return k - 1; }
    // normal approximation is plenty for a test generator
    double u1 = u01_oc(), u2 = u01();
    double z = std::sqrt(-2.0 * std::log(u1)) * std::cos(6.283185307179586 * u2);
    return std::max(0, (int)std::lround(lam + std::sqrt(lam) * z));
  }
};

struct SynthParams {
  int32_t num_tips = 100;
  int32_t num_sites = 30000;
  double tip_span = 365.0;          // tip dates ~ U[0, tip_span) days
  double tip_date_uncertainty = 0.0;  // half-width of [t_min, t_max] around each tip date for `frac_uncertain_tips`
  double frac_uncertain_tips = 0.0;
  double pop_n0 = 365.0;            // N(t0) * generation time, days (t0 = latest tip)
  double pop_growth = 0.0;          // per day
  double mu = 1e-3 / 365.0;         // substitutions / site / day
  double kappa = 5.0;
  double pi[4] = {0.31, 0.19, 0.21, 0.29};
  int32_t gaps_per_tip = 2;
  double mean_gap_len = 150.0;
  uint64_t seed = 20261001;
};

struct SynthResult {
  FlatTree tree;                    // the whole EMAT; ref_sequence == root sequence (no root mutations)
  std::vector<uint8_t> ref_sequence;
  double t_max_tip = 0.0;
};

namespace synth_detail {
using Ivs = std::vector<std::pair<int32_t, int32_t>>;
inline Ivs intersect(const Ivs& A, const Ivs& B) {
  Ivs out; size_t i = 0, j = 0;
  while (i < A.size() && j < B.size()) {
    int32_t s = std::max(A[i].first, B[j].first), e = std::min(A[i].second, B[j].second);
    if (s < e) out.push_back({s, e});
    if (A[i].second <= B[j].second) ++i; else ++j;
  }
  return out;
}
inline Ivs subtract(const Ivs& A, const Ivs& B) {
  Ivs out; size_t j = 0;
  for (auto [s, e] : A) {
    int32_t cs = s;
    while (j < B.size() && B[j].second <= cs) ++j;
    size_t k = j;
    while (k < B.size() && B[k].first < e) {
      if (B[k].first > cs) out.push_back({cs, B[k].first});
      cs = std::max(cs, B[k].second);
      ++k;
    }
    if (cs < e) out.push_back({cs, e});
  }
  return out;
}
inline bool contains(const Ivs& A, int32_t l) {
  auto it = std::upper_bound(A.begin(), A.end(), l, [](int32_t x, const std::pair<int32_t, int32_t>& iv) { return x < iv.first; });
  if (it == A.begin()) return false;
  --it; return l < it->second;
}
}  // namespace synth_detail

inline SynthResult make_synthetic_emat(const SynthParams& p) {
  using namespace synth_detail;
  SplitMix64 rng(p.seed);
  const int32_t n = p.num_tips, N = 2 * n - 1, L = p.num_sites;
  SynthResult R;
  FlatTree& T = R.tree;
  T.resize_nodes(N);

  // reference sequence
  R.ref_sequence.resize(L);
  for (int32_t l = 0; l < L; ++l) {
    double u = rng.u01(); int s = 0; double c = p.pi[0];
    while (u >= c && s < 3) { ++s; c += p.pi[s]; }
    R.ref_sequence[l] = (uint8_t)s;
  }

  // 1. tip dates
  std::vector<double> tip_t(n);
  for (int32_t i = 0; i < n; ++i) tip_t[i] = rng.u01() * p.tip_span;
  double t0 = *std::max_element(tip_t.begin(), tip_t.end());
  R.t_max_tip = t0;
  for (int32_t i = 0; i < n; ++i) {
    T.t[i] = tip_t[i];
    float lo = (float)tip_t[i], hi = (float)tip_t[i];
    if (p.frac_uncertain_tips > 0.0 && rng.u01() < p.frac_uncertain_tips) {
      lo = (float)(tip_t[i] - p.tip_date_uncertainty); hi = (float)(tip_t[i] + p.tip_date_uncertainty);
    }
    // keep t inside [t_min, t_max] after float rounding
    if ((double)lo > tip_t[i]) lo = std::nextafter(lo, -FLT_MAX);
    if ((double)hi < tip_t[i]) hi = std::nextafter(hi, FLT_MAX);
    T.t_min[i] = lo; T.t_max[i] = hi;
    R.t_max_tip = std::max(R.t_max_tip, (double)hi);
  }

  // 2. heterochronous coalescent, backwards in time from the latest tip
  std::vector<int32_t> order(n);
  for (int32_t i = 0; i < n; ++i) order[i] = i;
  std::sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return tip_t[a] > tip_t[b]; });
  std::vector<int32_t> active;
  int32_t next_tip = 0, next_inner = n;
  double t = tip_t[order[0]];
  active.push_back(order[next_tip++]);
  while (next_inner < N) {
    int k = (int)active.size();
    double t_next_tip = next_tip < n ? tip_t[order[next_tip]] : -1e300;
    double t_coal = -1e300;
    if (k >= 2) {
      double E = rng.expo() * 2.0 / ((double)k * (k - 1));   // target intensity
      if (p.pop_growth == 0.0) t_coal = t - E * p.pop_n0;
      else {
        // int_{t'}^{t} ds / N(s) = E  with N(s) = n0 exp(g (s - t0))
        double g = p.pop_growth;
        double v = std::exp(-g * (t - t0)) + p.pop_n0 * g * E;
        t_coal = v > 0.0 ? t0 - std::log(v) / g : -1e300;
      }
    }
    if (t_coal > t_next_tip) {
      t = t_coal;
      int i = rng.below(k); int32_t a = active[i]; active[i] = active.back(); active.pop_back();
      int j = rng.below(k - 1); int32_t b = active[j]; active[j] = active.back(); active.pop_back();
      int32_t u = next_inner++;
      T.t[u] = t; T.child0[u] = a; T.child1[u] = b; T.parent[a] = u; T.parent[b] = u;
      active.push_back(u);
    } else {
      t = t_next_tip;
      active.push_back(order[next_tip++]);
    }
  }
  T.root = N - 1;

  // 3. per-tip gaps, and M(node) = sites missing in the whole subtree below node (bottom-up intersections)
  std::vector<Ivs> M(N);
  for (int32_t i = 0; i < n; ++i) {
    Ivs g;
    for (int q = 0; q < p.gaps_per_tip; ++q) {
      int32_t len = 1 + (int32_t)std::floor(rng.expo() * p.mean_gap_len);
      int32_t s = rng.below(L); int32_t e = std::min(L, s + len);
      g.push_back({s, e});
    }
    std::sort(g.begin(), g.end());
    Ivs merged;
    for (auto iv : g) { if (!merged.empty() && iv.first <= merged.back().second) merged.back().second = std::max(merged.back().second, iv.second); else merged.push_back(iv); }
    M[i] = std::move(merged);
  }
  for (int32_t u = n; u < N; ++u) M[u] = intersect(M[T.child0[u]], M[T.child1[u]]);   // children created before parents

  // 4. mutations by DFS from the root with a running delta-from-ref map
  const double rmat[4][4] = {{0, 1, p.kappa, 1}, {1, 0, 1, p.kappa}, {p.kappa, 1, 0, 1}, {1, p.kappa, 1, 0}};
  std::vector<std::vector<std::pair<int32_t, std::pair<uint8_t, uint8_t>>>> node_muts(N);   // (site,(from,to)) in time order
  std::vector<std::vector<double>> node_mut_t(N);
  std::vector<std::vector<std::pair<int32_t, uint8_t>>> node_mfs(N);
  std::vector<Ivs> node_miss(N);
  std::unordered_map<int32_t, uint8_t> cur;   // site -> current state where != ref
  struct Frame { int32_t node; int stage; std::vector<std::pair<int32_t, uint8_t>> undo; };
  std::vector<Frame> stack;
  stack.push_back({T.root, 0, {}});
  while (!stack.empty()) {
    Frame& f = stack.back();
    int32_t u = f.node;
    if (f.stage == 0) {
      // missation at the start of this branch
      if (u == T.root) node_miss[u] = M[u]; else node_miss[u] = subtract(M[u], M[T.parent[u]]);
      if (!node_miss[u].empty())
        for (auto& kv : cur) if (contains(node_miss[u], kv.first)) node_mfs[u].push_back({kv.first, kv.second});
      std::sort(node_mfs[u].begin(), node_mfs[u].end());
      if (u != T.root) {
        double tp = T.t[T.parent[u]], len = T.t[u] - tp;
        int k = rng.poisson(p.mu * L * len);
        std::vector<double> ts(k);
        for (auto& x : ts) x = tp + len * rng.u01_oc();
        std::sort(ts.begin(), ts.end());
        for (int q = 0; q < k; ++q) {
          int32_t l = rng.below(L);
          if (contains(M[u], l)) continue;   // invisible: site missing in the whole subtree below
          auto it = cur.find(l);
          uint8_t from = it != cur.end() ? it->second : R.ref_sequence[l];
          double w[4], tot = 0.0;
          for (int b = 0; b < 4; ++b) { w[b] = (b == from) ? 0.0 : rmat[from][b] * p.pi[b]; tot += w[b]; }
          double x = rng.u01() * tot; uint8_t to = 0; double c = w[0];
          while ((x >= c || to == from) && to < 3) { ++to; c += w[to]; }
          if (to == from) to = (uint8_t)((from + 1) & 3);
          node_muts[u].push_back({l, {from, to}});
          node_mut_t[u].push_back(std::min(ts[q], T.t[u]));
          f.undo.push_back({l, from});
          if (to == R.ref_sequence[l]) cur.erase(l); else cur[l] = to;
        }
      }
      f.stage = 1;
      if (T.child0[u] != EMAT_NO_NODE) { int32_t c = T.child0[u]; stack.push_back({c, 0, {}}); }
      continue;
    }
    if (f.stage == 1) {
      f.stage = 2;
      if (T.child1[u] != EMAT_NO_NODE) { int32_t c = T.child1[u]; stack.push_back({c, 0, {}}); }
      continue;
    }
    for (auto it = f.undo.rbegin(); it != f.undo.rend(); ++it) { if (it->second == R.ref_sequence[it->first]) cur.erase(it->first); else cur[it->first] = it->second; }
    stack.pop_back();
  }

  // 5. pack CSR
  for (int32_t u = 0; u < N; ++u) {
    T.mut_offset[u + 1] = T.mut_offset[u] + (int32_t)node_muts[u].size();
    T.miss_offset[u + 1] = T.miss_offset[u] + (int32_t)node_miss[u].size();
    T.mfs_offset[u + 1] = T.mfs_offset[u] + (int32_t)node_mfs[u].size();
  }
  T.mut_site.reserve(T.mut_offset[N]);
  for (int32_t u = 0; u < N; ++u) {
    for (size_t q = 0; q < node_muts[u].size(); ++q) {
      T.mut_site.push_back(node_muts[u][q].first); T.mut_from.push_back(node_muts[u][q].second.first);
      T.mut_to.push_back(node_muts[u][q].second.second); T.mut_t.push_back(node_mut_t[u][q]);
    }
    for (auto& iv : node_miss[u]) { T.miss_start.push_back(iv.first); T.miss_end.push_back(iv.second); }
    for (auto& fs : node_mfs[u]) { T.mfs_site.push_back(fs.first); T.mfs_state.push_back(fs.second); }
  }
  return R;
}

}  // namespace emat
#endif  // EMAT_SYNTH_HPP_
