// orc_core.hpp -- CPU ORACLE (test infrastructure, NOT product code).
//
// Dependency-free C++17 restatement of Delphy's EMAT data model and scalar helpers, used
// only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg to check the HIP
// engine in delphy_amd/.  Nothing under delphy_amd/ may include or link this.
//
// Parity status: pinned against the reference's own known-answer tests (see oracle/orc_tests.cpp,
// which re-evaluates the fixtures and closed-form expectations of /root/reference/tests/*.cpp);
// the incomplete-gamma helpers replace Boost.Math 1.84 (absent from /root/reference) and are
// pinned against scipy.special golden vectors in tests/golden/gamma_q.json.
//
// Every block cites the reference file:line it follows.
#ifndef ORC_CORE_HPP_
#define ORC_CORE_HPP_

#include <algorithm>
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace orc {

[[noreturn]] inline void fail(const char* file, int line, const char* what) {
  std::fprintf(stderr, "ORACLE CHECK FAILED %s:%d: %s\n", file, line, what);
  throw std::runtime_error(std::string("oracle check failed: ") + what);
}
#define ORC_CHECK(cond) do { if (!(cond)) ::orc::fail(__FILE__, __LINE__, #cond); } while (0)

// ---- basic types (reference core/tree.h:33-39, core/sequence.h:155-241) -----------------------
using Node_index = int;
using Branch_index = int;
using Site_index = int;
constexpr Node_index k_no_node = -1;
using State = uint8_t;  // 0..3 = A, C, G, T (Real_seq_letter)
constexpr State sA = 0, sC = 1, sG = 2, sT = 3;
constexpr int k_num_states = 4;
constexpr double k_neg_dbl_max = -std::numeric_limits<double>::max();

// ---- counter-based RNG ------------------------------------------------------------------------
// The reference draws from std::mt19937 through Abseil distributions (subrun.cpp:110,125,200-205;
// spr_move.cpp:1160-1388), which cannot be reproduced on a GPU.  Oracle and HIP engine share this
// Philox4x32-10 stream instead: two 64-bit draws per 128-bit block, keyed per part, indexed by a block
// counter.  Parity is therefore exact on every discrete decision given the same seed.
struct Rng {
  uint64_t key = 0;
  uint64_t counter = 0;

  static inline uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
  static void philox4x32_10(uint64_t ctr, uint64_t key, uint32_t out[4]) {
    uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
    uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
    for (int r = 0; r < 10; ++r) {
      const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
      uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
      uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
      uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
  // Each Philox block yields two 64-bit draws; the second half is kept for the next draw.
  uint64_t spare = 0; bool has_spare = false;
  uint64_t next64() {
    if (has_spare) { has_spare = false; return spare; }
    uint32_t w[4];
    philox4x32_10(counter++, key, w);
    spare = (uint64_t)w[2] | ((uint64_t)w[3] << 32); has_spare = true;
    return (uint64_t)w[0] | ((uint64_t)w[1] << 32);
  }
  static double to_co(uint64_t a) { return (double)(a >> 11) * 0x1.0p-53; }                 // [0,1)
  static double to_oo(uint64_t a) { return ((double)(a >> 12) + 0.5) * 0x1.0p-52; }         // (0,1)
  static double to_oc(uint64_t a) { return ((double)(a >> 11) + 1.0) * 0x1.0p-53; }         // (0,1]
  double u01_co() { return to_co(next64()); }
  double u01_oo() { return to_oo(next64()); }
  double u01_oc() { return to_oc(next64()); }
  double uniform_co(double lo, double hi) { return lo + (hi - lo) * u01_co(); }   // absl::Uniform(lo, hi)
  double uniform_oc(double lo, double hi) { return lo + (hi - lo) * u01_oc(); }   // IntervalOpenClosed
  double uniform_oo(double lo, double hi) { return lo + (hi - lo) * u01_oo(); }   // IntervalOpenOpen
  int uniform_int(int n) { return (int)(((unsigned __int128)next64() * (uint64_t)n) >> 64); }   // [0, n)
  bool coin() { return (next64() >> 63) != 0; }
  double gaussian(double mean, double sigma) {  // Box-Muller on two 64-bit draws
    uint64_t a = next64(), b = next64();
    double u1 = to_oc(a), u2 = to_co(b);
    double r = std::sqrt(-2.0 * std::log(u1));
    return mean + sigma * (r * std::cos(6.283185307179586476925 * u2));
  }
  double exponential(double rate) { return -std::log(u01_oc()) / rate; }
  // Poisson(lambda) by sequential inversion on one uniform (replaces std::poisson_distribution in
  // reference distributions.h:152); lambda is mu*T per site, i.e. << 1 in practice.
  int poisson(double lambda) {
    double u = u01_co();
    double p = std::exp(-lambda), F = p;
    int k = 0;
    while (u >= F && k < 100000) { ++k; p *= lambda / k; F += p; }
    return k;
  }
};

// ---- interval sets (reference core/interval_set.h:28-224, 238-515) -----------------------------
using Site_interval = std::pair<Site_index, Site_index>;

struct Interval_set {
  std::vector<Site_interval> v;  // sorted by start, disjoint, non-adjacent, half-open

  bool operator==(const Interval_set& o) const { return v == o.v; }
  bool empty() const { return v.empty(); }
  void clear() { v.clear(); }
  int num_intervals() const { return (int)v.size(); }
  int num_sites() const { int r = 0; for (auto& [s, e] : v) r += e - s; return r; }

  // interval_set.h:130-138
  bool contains(Site_index l) const {
    auto it = std::upper_bound(v.begin(), v.end(), l,
                               [](Site_index x, const Site_interval& iv) { return x < iv.first; });
    if (it == v.begin()) return false;
    --it;
    return l < it->second;
  }
  // interval_set.h:96-125 (+ overlapping_intervals_closed :141-186): closed-interval overlap, so
  // touching intervals coalesce
  void insert(Site_interval iv) {
    auto first = std::lower_bound(v.begin(), v.end(), iv.first,
                                  [](const Site_interval& a, Site_index x) { return a.first < x; });
    if (first != v.begin()) {
      auto prec = std::prev(first);
      if (iv.first <= prec->second) first = prec;
    }
    auto last = std::upper_bound(v.begin(), v.end(), iv.second,
                                 [](Site_index x, const Site_interval& a) { return x < a.first; });
    if (first == last) { v.insert(first, iv); return; }
    if (last == std::next(first) && first->first <= iv.first && iv.second <= first->second) return;
    Site_index ns = std::min(iv.first, first->first);
    Site_index ne = std::max(iv.second, std::prev(last)->second);
    auto pos = v.erase(first, last);
    v.insert(pos, {ns, ne});
  }
  void insert(Site_index l) { insert({l, l + 1}); }
  // interval_set.h:192-207
  bool no_consecutive_intervals() const {
    for (size_t i = 1; i < v.size(); ++i) if (v[i].first <= v[i - 1].second) return false;
    return true;
  }
  // interval_set.h:210-219
  bool is_valid(Site_index num_sites) const {
    Site_index prev_end = -1;
    for (auto& [s, e] : v) {
      if (s < 0 || e > num_sites || s >= e || s <= prev_end) return false;
      prev_end = e;
    }
    return true;
  }
};

// interval_set.h:238-288
inline void merge_interval_sets(Interval_set& dst, const Interval_set& A, const Interval_set& B) {
  std::vector<Site_interval> out;
  bool inside = false;
  Site_index cs = 0, ce = 0;
  size_t ia = 0, ib = 0;
  while (!(ia == A.v.size() && ib == B.v.size())) {
    bool useA = (ia == A.v.size()) ? false : (ib == B.v.size()) ? true : (A.v[ia].first <= B.v[ib].first);
    auto [fs, fe] = useA ? A.v[ia] : B.v[ib];
    if (!inside) { cs = fs; ce = fe; (useA ? ia : ib)++; inside = true; }
    else if (fs <= ce) { ce = std::max(ce, fe); (useA ? ia : ib)++; }
    else { out.push_back({cs, ce}); inside = false; }
  }
  if (inside) out.push_back({cs, ce});
  dst.v = std::move(out);
}
// interval_set.h:301-337
inline void intersect_interval_sets(Interval_set& dst, const Interval_set& A, const Interval_set& B) {
  std::vector<Site_interval> out;
  size_t ia = 0, ib = 0;
  while (ia != A.v.size() && ib != B.v.size()) {
    auto [sa, ea] = A.v[ia]; auto [sb, eb] = B.v[ib];
    Site_index so = std::max(sa, sb), eo = std::min(ea, eb);
    if (so < eo) out.push_back({so, eo});
    if (ea <= eb) ++ia; else ++ib;
  }
  dst.v = std::move(out);
}
// interval_set.h:351-383
inline bool interval_sets_intersect(const Interval_set& A, const Interval_set& B) {
  size_t ia = 0, ib = 0;
  while (ia != A.v.size() && ib != B.v.size()) {
    auto [sa, ea] = A.v[ia]; auto [sb, eb] = B.v[ib];
    if (std::max(sa, sb) < std::min(ea, eb)) return true;
    if (ea <= eb) ++ia; else ++ib;
  }
  return false;
}
// interval_set.h:385-419
inline bool interval_set_is_subset_of(const Interval_set& A, const Interval_set& B) {
  size_t ia = 0, ib = 0;
  while (ia != A.v.size() && ib != B.v.size()) {
    auto [sa, ea] = A.v[ia]; auto [sb, eb] = B.v[ib];
    if (eb <= sa) ++ib;
    else if (sb <= sa && ea <= eb) ++ia;
    else return false;
  }
  return ia == A.v.size();
}
// interval_set.h:421-500   dst = A - B
inline void subtract_interval_sets(Interval_set& dst, const Interval_set& A, const Interval_set& B) {
  std::vector<Site_interval> out;
  if (A.v.empty()) { dst.v.clear(); return; }
  size_t ia = 0, ib = 0;
  Site_index cs = A.v[0].first, ce = A.v[0].second;
  auto nextA = [&]() { ++ia; if (ia != A.v.size()) { cs = A.v[ia].first; ce = A.v[ia].second; } };
  while (ia != A.v.size()) {
    if (ib == B.v.size()) { out.push_back({cs, ce}); nextA(); continue; }
    auto [bs, be] = B.v[ib];
    if (bs < cs) {
      if (be <= cs) ++ib;
      else if (be < ce) { cs = be; ++ib; }
      else nextA();
    } else if (bs < ce) {
      if (cs < bs) out.push_back({cs, bs});
      if (be < ce) { cs = be; ++ib; }
      else nextA();
    } else { out.push_back({cs, ce}); nextA(); }
  }
  dst.v = std::move(out);
}
inline Interval_set merged(const Interval_set& A, const Interval_set& B) { Interval_set r; merge_interval_sets(r, A, B); return r; }
inline Interval_set intersected(const Interval_set& A, const Interval_set& B) { Interval_set r; intersect_interval_sets(r, A, B); return r; }
inline Interval_set subtracted(const Interval_set& A, const Interval_set& B) { Interval_set r; subtract_interval_sets(r, A, B); return r; }

// ---- mutations (reference core/mutations.h:21-83) --------------------------------------------
struct Mutation {
  State from; Site_index site; State to; double t;
  bool operator==(const Mutation& o) const { return from == o.from && site == o.site && to == o.to && t == o.t; }
};
using Mutation_list = std::vector<Mutation>;
inline bool mutations_on_branch_less(const Mutation& a, const Mutation& b) {   // mutations.h:41-43
  return a.t < b.t || (a.t == b.t && a.site < b.site);
}
// NOTE: std::sort is not stable; ties on (t, site) with different from/to cannot occur on a
// valid branch except transiently, so we use stable_sort to make the order fully deterministic.
inline void sort_mutations(Mutation_list& ms) { std::stable_sort(ms.begin(), ms.end(), mutations_on_branch_less); }
inline void clamp_mutation_times(Mutation_list& ms, double lo, double hi) {      // mutations.h:55-60
  for (auto& m : ms) m.t = std::clamp(m.t, lo, hi);
}
struct Seq_delta {
  Site_index site; State from; State to;
  Seq_delta() = default;
  Seq_delta(Site_index s, State f, State t) : site(s), from(f), to(t) {}
  Seq_delta(const Mutation& m) : site(m.site), from(m.from), to(m.to) {}
  Seq_delta inverse() const { return {site, to, from}; }
};

// ---- missation maps (reference core/mutations.h:124-350) --------------------------------------
struct Missation_map {
  Interval_set intervals;
  std::map<Site_index, State> from_states;  // only sites whose state differs from ref_sequence

  bool operator==(const Missation_map& o) const { return intervals == o.intervals && from_states == o.from_states; }
  bool empty() const { return intervals.empty(); }
  void clear() { intervals.clear(); from_states.clear(); }
  int num_intervals() const { return intervals.num_intervals(); }
  int num_sites() const { return intervals.num_sites(); }
  bool contains(Site_index l) const { return intervals.contains(l); }
  State get_from_state(Site_index l, const std::vector<State>& ref) const {       // :190-197
    auto it = from_states.find(l);
    return it != from_states.end() ? it->second : ref[l];
  }
  void set_from_state(Site_index l, State from, const std::vector<State>& ref) {  // :198-210
    ORC_CHECK(l >= 0 && l < (int)ref.size());
    if (from != ref[l]) from_states[l] = from; else from_states.erase(l);
  }
  void insert(Site_index l, State from, const std::vector<State>& ref) {          // :185-189
    intervals.insert(l);
    set_from_state(l, from, ref);
  }
  void ref_seq_changed(Site_index l, State old_ref, State new_ref) {              // :212-232
    if (!intervals.contains(l)) return;
    auto it = from_states.find(l);
    if (it != from_states.end()) {
      if (it->second == new_ref) from_states.erase(it);
    } else if (old_ref != new_ref) {
      from_states[l] = old_ref;
    }
  }
};
// mutations.h:250-298
inline void factor_out_common_missations(const Missation_map& A, const Missation_map& B,
                                         Missation_map& rA, Missation_map& rB, Missation_map& rC) {
  intersect_interval_sets(rC.intervals, A.intervals, B.intervals);
  subtract_interval_sets(rA.intervals, A.intervals, rC.intervals);
  subtract_interval_sets(rB.intervals, B.intervals, rC.intervals);
  auto ia = A.from_states.begin(); auto ib = B.from_states.begin();
  while (ia != A.from_states.end() && ib != B.from_states.end()) {
    if (ia->first < ib->first) { rA.from_states.insert(*ia); ++ia; }
    else if (ib->first < ia->first) { rB.from_states.insert(*ib); ++ib; }
    else { rC.from_states.insert(*ia); ++ia; ++ib; }
  }
  for (; ia != A.from_states.end(); ++ia) rA.from_states.insert(*ia);
  for (; ib != B.from_states.end(); ++ib) rB.from_states.insert(*ib);
}
// mutations.h:300-312 (in-place form)
inline void factor_out_common_missations(Missation_map& A, Missation_map& B, Missation_map& common) {
  Missation_map rA, rB;
  factor_out_common_missations(A, B, rA, rB, common);
  A = std::move(rA); B = std::move(rB);
}
// mutations.h:315-336
inline Missation_map merge_missations_nondestructively(const Missation_map& A, const Missation_map& B) {
  Missation_map r;
  merge_interval_sets(r.intervals, A.intervals, B.intervals);
  r.from_states.insert(A.from_states.begin(), A.from_states.end());
  r.from_states.insert(B.from_states.begin(), B.from_states.end());
  return r;
}

// ---- tree (reference core/tree.h:76-226, core/phylo_tree.h:14-64) ------------------------------
struct Phylo_node {
  Node_index parent = k_no_node;
  Node_index children[2] = {k_no_node, k_no_node};
  float t_min = +42.0f, t_max = -42.0f;
  double t = 0.0;
  Mutation_list mutations;
  Missation_map missations;

  bool is_tip() const { return children[0] == k_no_node; }
  bool is_inner_node() const { return !is_tip(); }
  int num_children() const { return is_tip() ? 0 : 2; }
  Node_index sibling_of(Node_index X) const {       // tree.h:177-180
    ORC_CHECK(X == children[0] || X == children[1]);
    return X == children[0] ? children[1] : children[0];
  }
};
struct Phylo_tree_loc { Branch_index branch; double t; };
inline bool operator==(const Phylo_tree_loc& a, const Phylo_tree_loc& b) { return a.branch == b.branch && a.t == b.t; }
inline bool operator!=(const Phylo_tree_loc& a, const Phylo_tree_loc& b) { return !(a == b); }

struct Phylo_tree {
  Node_index root = k_no_node;
  std::vector<Phylo_node> nodes;
  std::vector<State> ref_sequence;

  Phylo_tree() = default;
  explicit Phylo_tree(int n) : nodes(n) {}
  int size() const { return (int)nodes.size(); }
  Site_index num_sites() const { return (int)ref_sequence.size(); }
  Phylo_node& at(Node_index i) { return nodes.at(i); }
  const Phylo_node& at(Node_index i) const { return nodes.at(i); }
  Phylo_node& at_parent_of(Node_index i) { return at(at(i).parent); }
  const Phylo_node& at_parent_of(Node_index i) const { return at(at(i).parent); }
  Phylo_node& at_root() { return at(root); }
  const Phylo_node& at_root() const { return at(root); }
  Phylo_tree_loc node_loc(Node_index n) const { return {n, at(n).t}; }
  double branch_begin_t(Branch_index b) const { return at_parent_of(b).t; }
  double branch_end_t(Branch_index b) const { return at(b).t; }
};

// Deterministic equivalents of the reference's coroutine traversals (tree.h:243-318): children
// are visited in stored order.
inline std::vector<Node_index> pre_order(const Phylo_tree& tree) {
  std::vector<Node_index> out, stack;
  if (tree.size() == 0) return out;
  stack.push_back(tree.root);
  while (!stack.empty()) {
    Node_index n = stack.back(); stack.pop_back();
    out.push_back(n);
    if (tree.at(n).is_inner_node()) { stack.push_back(tree.at(n).children[1]); stack.push_back(tree.at(n).children[0]); }
  }
  return out;
}
inline std::vector<Node_index> post_order(const Phylo_tree& tree) {
  std::vector<Node_index> out;
  if (tree.size() == 0) return out;
  std::vector<std::pair<Node_index, int>> stack;
  stack.push_back({tree.root, 0});
  while (!stack.empty()) {
    auto& [n, k] = stack.back();
    if (tree.at(n).is_tip() || k == 2) { out.push_back(n); stack.pop_back(); }
    else { Node_index c = tree.at(n).children[k]; ++k; stack.push_back({c, 0}); }
  }
  return out;
}

// ---- evolution model (reference core/evo_model.h:11-48, core/evo_hky.cpp:7-50) ------------------
struct Site_evo_model {
  double mu = 0.0;
  double pi_a[4] = {0, 0, 0, 0};
  double q_ab[4][4] = {{0}};
  double q_a(State a) const { return -q_ab[a][a]; }
};
struct Global_evo_model {
  std::vector<int> partition_for_site;
  std::vector<double> nu_l;
  std::vector<Site_evo_model> partition_evo_model;
  int num_partitions() const { return (int)partition_evo_model.size(); }
  double mu_l(Site_index l) const { return partition_evo_model[partition_for_site[l]].mu; }
  double pi_l_a(Site_index l, State a) const { return partition_evo_model[partition_for_site[l]].pi_a[a]; }
  double q_l_a(Site_index l, State a) const { return partition_evo_model[partition_for_site[l]].q_a(a); }
  double q_l_ab(Site_index l, State a, State b) const { return partition_evo_model[partition_for_site[l]].q_ab[a][b]; }
};
inline Global_evo_model make_single_partition_global_evo_model(Site_index L) {  // evo_model.cpp:7-13
  Global_evo_model e;
  e.partition_for_site.assign(L, 0);
  e.nu_l.assign(L, 1.0);
  e.partition_evo_model.assign(1, Site_evo_model{});
  return e;
}
inline Global_evo_model make_global_evo_model(std::vector<int> partition_for_site) {  // evo_model.cpp:15-24
  Global_evo_model e;
  int P = 1 + *std::max_element(partition_for_site.begin(), partition_for_site.end());
  e.nu_l.assign(partition_for_site.size(), 1.0);
  e.partition_for_site = std::move(partition_for_site);
  e.partition_evo_model.assign(P, Site_evo_model{});
  return e;
}
struct Hky_model {
  double mu = 0.0, kappa = 1.0;
  double pi_a[4] = {0.25, 0.25, 0.25, 0.25};
  // evo_hky.cpp:7-50: q_ab = r_ab pi_b / R with R = pi^T r pi
  Site_evo_model derive_site_evo_model() const {
    double r[4][4] = {{0, 1, kappa, 1}, {1, 0, 1, kappa}, {kappa, 1, 0, 1}, {1, kappa, 1, 0}};
    // Eigen evaluates (pi^T r) pi: first the row vector, then the dot product
    double rowv[4];
    for (int b = 0; b < 4; ++b) { rowv[b] = 0.0; for (int a = 0; a < 4; ++a) rowv[b] += pi_a[a] * r[a][b]; }
    double R = 0.0;
    for (int b = 0; b < 4; ++b) R += rowv[b] * pi_a[b];
    Site_evo_model m; m.mu = mu;
    for (int a = 0; a < 4; ++a) {
      m.pi_a[a] = pi_a[a];
      m.q_ab[a][a] = 0.0;
      for (int b = 0; b < 4; ++b) if (a != b) { m.q_ab[a][b] = r[a][b] / R * pi_a[b]; m.q_ab[a][a] -= m.q_ab[a][b]; }
    }
    return m;
  }
};

// ---- population models (reference core/pop_model.cpp:18-145, 181-204, 247-330, 525-560) ---------
struct Pop_model {
  virtual ~Pop_model() = default;
  virtual double pop_at_time(double t) const = 0;
  virtual double pop_integral(double a, double b) const = 0;
  virtual double intensity_integral(double a, double b) const = 0;
};
struct Const_pop_model : Pop_model {
  double pop;
  explicit Const_pop_model(double p) : pop(p) { if (p <= 0.0) throw std::invalid_argument("pop must be positive"); }
  double pop_at_time(double) const override { return pop; }
  double pop_integral(double a, double b) const override { return (b - a) * pop; }
  double intensity_integral(double a, double b) const override { return (b - a) / pop; }
};
struct Exp_pop_model : Pop_model {
  double t0, n0, g, min_pop, t_c;
  Exp_pop_model(double t0_, double n0_, double g_, double min_pop_) : t0(t0_), n0(n0_), g(g_), min_pop(min_pop_) {
    if (n0 <= 0.0) throw std::invalid_argument("pop_at_t0 must be positive");
    if (min_pop < 0.0) throw std::invalid_argument("min_pop must be non-negative");
    t_c = (min_pop > 0.0 && g != 0.0) ? t0 + std::log(min_pop / n0) / g : std::numeric_limits<double>::quiet_NaN();
  }
  double pop_at_time(double t) const override { return std::max(min_pop, n0 * std::exp((t - t0) * g)); }
  double unclamped_int(double a, double b) const { return n0 / g * std::exp(g * (a - t0)) * std::expm1(g * (b - a)); }
  double pop_integral(double a, double b) const override {   // pop_model.cpp:43-91
    ORC_CHECK(a <= b);
    if (min_pop == 0.0) return g == 0.0 ? (b - a) * n0 : unclamped_int(a, b);
    if (g == 0.0) return (b - a) * std::max(min_pop, n0);
    if (g > 0.0) {
      if (b <= t_c) return (b - a) * min_pop;
      if (a >= t_c) return unclamped_int(a, b);
      return (t_c - a) * min_pop + n0 / g * std::exp(g * (t_c - t0)) * std::expm1(g * (b - t_c));
    }
    if (a >= t_c) return (b - a) * min_pop;
    if (b <= t_c) return unclamped_int(a, b);
    return n0 / g * std::exp(g * (a - t0)) * std::expm1(g * (t_c - a)) + (b - t_c) * min_pop;
  }
  double unclamped_intensity(double a, double b) const { return -1.0 / (n0 * g) * std::exp(-g * (a - t0)) * std::expm1(-g * (b - a)); }
  double intensity_integral(double a, double b) const override {   // pop_model.cpp:93-145
    ORC_CHECK(a <= b);
    if (min_pop == 0.0) return g == 0.0 ? (b - a) / n0 : unclamped_intensity(a, b);
    if (g == 0.0) return (b - a) / std::max(min_pop, n0);
    double inv = 1.0 / min_pop;
    if (g > 0.0) {
      if (b <= t_c) return (b - a) * inv;
      if (a >= t_c) return unclamped_intensity(a, b);
      return (t_c - a) * inv - 1.0 / (n0 * g) * std::exp(-g * (t_c - t0)) * std::expm1(-g * (b - t_c));
    }
    if (a >= t_c) return (b - a) * inv;
    if (b <= t_c) return unclamped_intensity(a, b);
    return -1.0 / (n0 * g) * std::exp(-g * (a - t0)) * std::expm1(-g * (t_c - a)) + (b - t_c) * inv;
  }
};
struct Skygrid_pop_model : Pop_model {
  enum Type { k_staircase = 1, k_log_linear = 2 };
  std::vector<double> x, gamma, minus_gamma;
  Type type;
  Skygrid_pop_model(std::vector<double> x_, std::vector<double> g_, Type ty) : x(std::move(x_)), gamma(std::move(g_)), type(ty) {
    if (x.size() < 2) throw std::invalid_argument("Skygrid needs at least two knots");
    if (x.size() != gamma.size()) throw std::invalid_argument("Skygrid x/gamma size mismatch");
    for (size_t i = 0; i + 1 < x.size(); ++i) if (!(x[i] < x[i + 1])) throw std::invalid_argument("Skygrid knots must increase");
    minus_gamma.resize(gamma.size());
    for (size_t k = 0; k < gamma.size(); ++k) minus_gamma[k] = -gamma[k];
  }
  int M() const { return (int)x.size() - 1; }
  int interval_containing_t(double t) const {   // pop_model.cpp:551-560
    auto it = std::lower_bound(x.begin(), x.end(), t);
    if (it == x.begin()) return 0;
    if (it == x.end()) return M() + 1;
    return (int)(it - x.begin());
  }
  double log_N(double t) const {               // pop_model.cpp:181-200
    int k = interval_containing_t(t), m = M();
    if (k == 0) return gamma[0];
    if (k > m) return gamma[m];
    if (type == k_staircase) return gamma[k];
    double c = (t - x[k - 1]) / (x[k] - x[k - 1]);
    return (1 - c) * gamma[k - 1] + c * gamma[k];
  }
  double pop_at_time(double t) const override { return std::exp(log_N(t)); }
  double log_int_N_core(double a, double b, const std::vector<double>& ge) const {   // pop_model.cpp:247-330
    ORC_CHECK(a <= b);
    int m = M();
    int ka = interval_containing_t(a), kb = interval_containing_t(b);
    int kka = std::max(ka - 1, 0), kkb = std::min(kb, m);
    double bias = -std::numeric_limits<double>::infinity();
    for (int k = kka; k <= kkb; ++k) bias = std::max(bias, ge[k]);
    double result = 0.0;
    const double inf = std::numeric_limits<double>::infinity();
    for (int k = ka; k <= kb; ++k) {
      double lo = std::max(a, k > 0 ? x[k - 1] : -inf);
      double hi = std::min(b, k <= m ? x[k] : +inf);
      if (k == 0) result += std::exp(-bias + ge[0]) * (hi - lo);
      else if (k == m + 1) result += std::exp(-bias + ge[m]) * (hi - lo);
      else if (type == k_staircase) result += std::exp(-bias + ge[k]) * (hi - lo);
      else {
        if (ge[k] == ge[k - 1]) result += std::exp(-bias + ge[k]) * (hi - lo);
        else {
          double c_lo = (lo - x[k - 1]) / (x[k] - x[k - 1]);
          double c_hi = (hi - x[k - 1]) / (x[k] - x[k - 1]);
          double G_lo = (1 - c_lo) * ge[k - 1] + c_lo * ge[k];
          double G_hi = (1 - c_hi) * ge[k - 1] + c_hi * ge[k];
          double D = G_hi - G_lo;
          if (D == 0.0) result += std::exp(-bias + G_lo) * (hi - lo);
          else result += std::exp(-bias + G_lo) * (hi - lo) * (std::expm1(D) / D);
        }
      }
    }
    return std::log(result) + bias;
  }
  double pop_integral(double a, double b) const override { return std::exp(log_int_N_core(a, b, gamma)); }
  double intensity_integral(double a, double b) const override { return std::exp(log_int_N_core(a, b, minus_gamma)); }
};

// ---- proposal distributions (reference core/distributions.h:38-175) ----------------------------
struct Bounded_exponential_distribution {
  double lambda, a, b;
  Bounded_exponential_distribution(double l, double a_, double b_) : lambda(l), a(a_), b(b_) {
    ORC_CHECK(a <= b);
    ORC_CHECK(!(std::isinf(a) && std::isinf(b)));
    ORC_CHECK(!(lambda > 0.0 && std::isinf(b)));
    ORC_CHECK(!(lambda < 0.0 && std::isinf(a)));
  }
  double bound(double x) const { return std::clamp(x, a, b); }
  double operator()(Rng& rng) const {
    double u = rng.u01_oo();
    double ltr = lambda * (b - a);
    if (lambda == 0.0) return bound(a + u * (b - a));
    if (lambda > 0 && ltr > 100) return bound(b + std::log(u) / lambda);
    if (lambda < 0 && ltr < -100) return bound(a + std::log(u) / lambda);
    return bound(a + std::log1p(u * (std::exp(ltr) - 1)) / lambda);
  }
};
struct K_truncated_poisson_distribution {
  double lambda; int min_k; double normalization = 0.0, term_before_min_k = 0.0, max_k = 0.0;
  K_truncated_poisson_distribution(double l, int mk) : lambda(l), min_k(mk) {
    ORC_CHECK(lambda > 0.0); ORC_CHECK(min_k >= 0);
    if (min_k <= lambda) return;  // rejection sampling
    max_k = std::max(10.0 * min_k, 10.0 * lambda);
    double last_term = 1.0;
    double expm1_lambda = std::expm1(lambda);
    normalization = expm1_lambda;
    for (int k = 1; k < min_k; ++k) { last_term *= lambda / k; normalization -= last_term; }
    term_before_min_k = last_term;
    if (normalization <= 0.0 || std::fabs(normalization) < 1e-10 * expm1_lambda) {
      normalization = 0.0;
      double nlt = last_term;
      for (int k = min_k; k < max_k; ++k) { nlt *= lambda / k; normalization += nlt; }
    }
    ORC_CHECK(normalization > 0.0);
  }
  int operator()(Rng& rng) const {
    if (normalization == 0.0) {
      while (true) { int k = rng.poisson(lambda); if (k >= min_k) return k; }
    }
    double u = rng.uniform_co(0.0, normalization);
    double cum = 0.0; int k = min_k; double term = term_before_min_k;
    while (k < max_k) { term *= lambda / k; cum += term; if (cum > u) break; ++k; }
    return k;
  }
};

// ---- incomplete gamma (replaces Boost.Math 1.84 gamma_q / gamma_q_inv, reference
//      core/safe_gamma_math.h:45-139; algorithm: series for x < a+1, modified-Lentz continued
//      fraction otherwise [Numerical Recipes 3e s6.2]; inverse by Halley iterations on P or Q) --
// the continued fraction's value h: Q(a, x) = exp(-x + a log x - lgamma(a)) * h for x >= a + 1
inline double gamma_q_fraction(double a, double x) {
  const double FPMIN = 1e-300;
  double b = x + 1.0 - a, c = 1.0 / FPMIN, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; ++i) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b; if (std::fabs(d) < FPMIN) d = FPMIN;
    c = b + an / c; if (std::fabs(c) < FPMIN) c = FPMIN;
    d = 1.0 / d;
    double del = d * c; h *= del;
    if (std::fabs(del - 1.0) < 1e-16) break;
  }
  return h;
}
inline double gamma_q(double a, double x) {
  ORC_CHECK(a > 0.0 && x >= 0.0);
  if (x == 0.0) return 1.0;
  if (std::isinf(x)) return 0.0;
  const double lg = std::lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 100000; ++n) {
      ap += 1.0; del *= x / ap; sum += del;
      if (std::fabs(del) < std::fabs(sum) * 1e-17) break;
    }
    double P = sum * std::exp(-x + a * std::log(x) - lg);
    return 1.0 - P;
  }
  return std::exp(-x + a * std::log(x) - lg) * gamma_q_fraction(a, x);
}
inline double safe_gamma_q(double a, double x) { return gamma_q(a, x); }
// x such that Q(a, x) = q
inline double safe_gamma_q_inv(double a, double q) {
  ORC_CHECK(q >= 0.0 && q <= 1.0);
  if (q == 0.0) return std::numeric_limits<double>::infinity();
  if (q == 1.0) return 0.0;
  const double lg = std::lgamma(a);
  if (q < 1e-3) {
    // Far upper tail (the reference's own test asks for Q down to 1e-300, safe_gamma_math_tests.cpp:83-95,247-262):
    // Newton on log Q(a, x) = log q, with log Q = -x + a log x - lgamma(a) + log h taken from the continued fraction
    // without ever forming Q, and d/dx log Q = -density / Q = -1 / (x h).  log Q is monotone and concave (a >= 1) or
    // convex (a < 1) in x, so the steps close in on the root from one side after the first.
    const double lq = std::log(q);
    double x = a + 1.0, h = gamma_q_fraction(a, x), prev = 0.0;
    if (-x + a * std::log(x) - lg + std::log(h) > lq) {   // the root lies where the fraction converges
      for (int j = 0; j < 100; ++j) {
        double dx = (-x + a * std::log(x) - lg + std::log(h) - lq) * x * h;
        double xn = x + dx;
        if (xn < a + 1.0) xn = a + 1.0;
        bool done = std::fabs(xn - x) <= 1e-14 * xn || (j > 2 && std::fabs(dx) >= std::fabs(prev));
        x = xn; prev = dx;
        if (done) break;
        h = gamma_q_fraction(a, x);
      }
      return x;
    }
  }
  const double p = 1.0 - q;
  double x;
  // initial guess (NR3 invgammp)
  if (a > 1.0) {
    double pp = (p < 0.5) ? p : q;
    double t = std::sqrt(-2.0 * std::log(pp));
    double xg = (2.30753 + t * 0.27061) / (1.0 + t * (0.99229 + t * 0.04481)) - t;
    if (p < 0.5) xg = -xg;
    x = std::max(1e-3, a * std::pow(1.0 - 1.0 / (9.0 * a) - xg / (3.0 * std::sqrt(a)), 3));
  } else {
    double t = 1.0 - a * (0.253 + a * 0.12);
    if (p < t) x = std::pow(p / t, 1.0 / a);
    else x = 1.0 - std::log(1.0 - (p - t) / (1.0 - t));
  }
  const double a1 = a - 1.0;
  for (int j = 0; j < 60; ++j) {
    if (x <= 0.0) { x = 1e-300; }
    // f(x) = Q(a,x) - q (decreasing); f' = -x^{a-1} e^{-x} / Gamma(a)
    double err = gamma_q(a, x) - q;
    double tdens = std::exp(-x + a1 * std::log(x) - lg);   // density of Gamma(a,1)
    if (tdens == 0.0) break;
    // Halley on P(a,x) - p (NR3 invgammp): u = (P - p)/P' = -(Q - q)/dens
    double u = -err / tdens;
    double dx = u / (1.0 - 0.5 * std::min(1.0, u * (a1 / x - 1.0)));
    double xn = x - dx;
    if (xn <= 0.0) xn = 0.5 * x;
    if (std::fabs(xn - x) < 1e-15 * std::max(xn, 1e-300)) { x = xn; break; }
    x = xn;
  }
  return x;
}
inline double safe_log_gamma_integral(double a, double x_min, double x_max) {   // safe_gamma_math.h:82-90
  ORC_CHECK(x_min < x_max);
  double Q_hi = safe_gamma_q(a, x_min), Q_lo = safe_gamma_q(a, x_max);
  ORC_CHECK(Q_hi >= Q_lo);
  return std::log(Q_hi - Q_lo);
}
inline double safe_sample_truncated_gamma(double alpha, double beta, double lo, double hi, Rng& rng) {  // :112-139
  ORC_CHECK(alpha > 0.0 && beta > 0.0 && lo < hi);
  double y_lo = beta * lo, y_hi = beta * hi;
  double Q_hi = safe_gamma_q(alpha, y_lo), Q_lo = safe_gamma_q(alpha, y_hi);
  ORC_CHECK(Q_lo < Q_hi);
  double rand_Q = rng.uniform_oc(Q_lo, Q_hi);
  double y = safe_gamma_q_inv(alpha, rand_Q);
  double x = y / beta;
  return std::clamp(x, lo, hi);
}

}  // namespace orc
#endif  // ORC_CORE_HPP_
