// orc_tests.cpp -- pins the CPU ORACLE against the reference's own known-answer tests.
//
// The reference's GoogleTest suite cannot be built here (Abseil/Boost/Eigen/GoogleTest absent), but its
// expectations are fixtures + closed-form formulas, which this file re-evaluates against the oracle:
//   /root/reference/tests/interval_set_tests.cpp, missation_map_tests.cpp, site_deltas_tests.cpp,
//   phylo_tree_calc_tests.cpp:14-470, spr_study_tests.cpp:14-205, spr_move_tests.cpp:73-1793,
//   tree_editing_tests.cpp, very_scalable_coalescent_tests.cpp:11-182, scalable_coalescent_tests.cpp,
//   pop_model_tests.cpp:45-234,281-780, distributions_tests.cpp, utree_tests.cpp (in orc_utree_tests.hpp).
// Only fixture DATA and expected VALUES are restated (with the line they come from); no reference
// source is copied.  Invoked by tests/test_oracle_pinning.py; exit code = number of failed checks.
#include <cstdio>
#include <set>

#include "../delphy_amd/csrc/synth.hpp"
#include "orc_build.hpp"
#include "orc_subrun.hpp"

using namespace orc;

static int g_fail = 0, g_checks = 0;
static const char* g_test = "";
#define EXPECT(cond) do { ++g_checks; if (!(cond)) { ++g_fail; std::printf("FAIL [%s] %s:%d: %s\n", g_test, __FILE__, __LINE__, #cond); } } while (0)
#define EXPECT_NEAR(a, b, tol) do { ++g_checks; double _a = (a), _b = (b); if (!(std::fabs(_a - _b) <= (tol))) { ++g_fail; std::printf("FAIL [%s] %s:%d: %s=%.15g vs %s=%.15g (tol %g)\n", g_test, __FILE__, __LINE__, #a, _a, #b, _b, (double)(tol)); } } while (0)
#define TEST(name) static void name(); struct name##_reg { name##_reg() { tests().push_back({#name, name}); } } name##_inst; static void name()
struct TestEntry { const char* name; void (*fn)(); };
static std::vector<TestEntry>& tests() { static std::vector<TestEntry> t; return t; }

static const double NEG = k_neg_dbl_max;

// ---- shared evo model of the reference fixtures (phylo_tree_calc_tests.cpp:48-74, spr_move_tests.cpp:23-49)
static Global_evo_model fixture_evo(std::vector<int> part = {0, 1, 0, 1}, std::vector<double> nu = {0.2, 0.3, 0.4, 0.5}) {
  auto evo = make_global_evo_model(part);
  evo.nu_l = nu;
  double q0[4][4] = {{-0.6 - 0.7 - 0.8, 0.6, 0.7, 0.8}, {0.9, -0.9 - 1.0 - 1.1, 1.0, 1.1}, {1.2, 1.3, -1.2 - 1.3 - 1.4, 1.4}, {1.5, 1.6, 1.7, -1.5 - 1.6 - 1.7}};
  double q1[4][4] = {{-2.6 - 2.7 - 2.8, 2.6, 2.7, 2.8}, {2.9, -2.9 - 3.0 - 3.1, 3.0, 3.1}, {3.2, 3.3, -3.2 - 3.3 - 3.4, 3.4}, {3.5, 3.6, 3.7, -3.5 - 3.6 - 3.7}};
  evo.partition_evo_model[0].mu = 0.1; evo.partition_evo_model[1].mu = 1.1;
  double pi0[4] = {0.05, 0.15, 0.25, 0.55}, pi1[4] = {0.07, 0.17, 0.23, 0.53};
  for (int a = 0; a < 4; ++a) { evo.partition_evo_model[0].pi_a[a] = pi0[a]; evo.partition_evo_model[1].pi_a[a] = pi1[a];
    for (int b = 0; b < 4; ++b) { evo.partition_evo_model[0].q_ab[a][b] = q0[a][b]; evo.partition_evo_model[1].q_ab[a][b] = q1[a][b]; } }
  return evo;
}
static void set_inner(Phylo_tree& t, int n, int parent, int c0, int c1, double time) {
  t.at(n).parent = parent; t.at(n).children[0] = c0; t.at(n).children[1] = c1; t.at(n).t = time; t.at(n).t_min = -FLT_MAX; t.at(n).t_max = FLT_MAX;
}
static void set_tip(Phylo_tree& t, int n, int parent, double time) {
  t.at(n).parent = parent; t.at(n).t = time; t.at(n).t_min = (float)time; t.at(n).t_max = (float)time;
}
static void miss(Phylo_tree& t, int n, std::vector<std::pair<int, State>> ms) { for (auto& [l, s] : ms) t.at(n).missations.insert(l, s, t.ref_sequence); }

enum { r_ = 0, x_ = 1, a_ = 2, b_ = 3, c_ = 4 };
// The 5-node fixture (phylo_tree_calc_tests.cpp:14-116 / spr_study_tests.cpp:14-82 with ref AACA and root
// delta C2A; spr_move_tests.cpp:73-140 with ref ACAA and root delta C1A).
static Phylo_tree complex_tree(bool spr_move_variant) {
  Phylo_tree t(5);
  t.root = r_;
  t.ref_sequence = spr_move_variant ? std::vector<State>{sA, sC, sA, sA} : std::vector<State>{sA, sA, sC, sA};
  set_inner(t, r_, k_no_node, x_, c_, -1.0);
  t.at(r_).mutations = {Mutation{sC, spr_move_variant ? 1 : 2, sA, NEG}};
  miss(t, r_, {{3, sA}});
  set_inner(t, x_, r_, a_, b_, 0.0);
  t.at(x_).mutations = {Mutation{sA, 0, sT, -0.5}};
  miss(t, x_, {{2, sA}});
  set_tip(t, a_, x_, 1.0); t.at(a_).mutations = {Mutation{sT, 0, sC, 0.5}};
  set_tip(t, b_, x_, 2.0); t.at(b_).mutations = {Mutation{sA, 1, sG, 1.0}};
  set_tip(t, c_, r_, 3.0); t.at(c_).mutations = {Mutation{sA, 0, sT, 0.0}, Mutation{sT, 0, sG, 1.0}};
  miss(t, c_, {{1, sA}});
  return t;
}

// ---- brute-force per-site evaluator: the general form of the per-site sums that the reference tests
//      write out by hand (e.g. phylo_tree_calc_tests.cpp:381-439) ---------------------------------------
struct Brute { std::vector<double> lambda_i; double log_G_below_root; std::vector<int> num_missing; };
static Brute brute_force(const Phylo_tree& tree, const Global_evo_model& evo) {
  int L = tree.num_sites(), N = tree.size();
  Brute B; B.lambda_i.assign(N, 0.0); B.num_missing.assign(N, 0); B.log_G_below_root = 0.0;
  std::vector<std::vector<State>> seq(N); std::vector<std::vector<char>> missing(N);
  for (auto n : pre_order(tree)) {
    auto& nd = tree.at(n);
    seq[n] = (n == tree.root) ? tree.ref_sequence : seq[nd.parent];
    missing[n] = (n == tree.root) ? std::vector<char>(L, 0) : missing[nd.parent];
    for (auto& [s, e] : nd.missations.intervals.v) for (int l = s; l < e; ++l) missing[n][l] = 1;
    double t_P = (n == tree.root) ? 0.0 : tree.at(nd.parent).t;
    if (n != tree.root) {
      for (int l = 0; l < L; ++l) {
        if (missing[n][l]) continue;
        double rate = evo.mu_l(l) * evo.nu_l[l];
        State s = seq[nd.parent][l]; double t_prev = t_P;
        for (auto& m : nd.mutations) if (m.site == l) {
          B.log_G_below_root += -rate * evo.q_l_a(l, s) * (m.t - t_prev) + std::log(rate * evo.q_l_ab(l, m.from, m.to));
          s = m.to; t_prev = m.t;
        }
        B.log_G_below_root += -rate * evo.q_l_a(l, s) * (nd.t - t_prev);
      }
    }
    for (auto& m : nd.mutations) seq[n][m.site] = m.to;
    for (int l = 0; l < L; ++l) { if (missing[n][l]) ++B.num_missing[n]; else B.lambda_i[n] += evo.mu_l(l) * evo.nu_l[l] * evo.q_l_a(l, seq[n][l]); }
  }
  return B;
}

static Phylo_tree tree_from_flat(const emat::FlatTree& f, const std::vector<uint8_t>& ref) {
  Phylo_tree t(f.num_nodes());
  t.root = f.root; t.ref_sequence.assign(ref.begin(), ref.end());
  for (int i = 0; i < f.num_nodes(); ++i) {
    auto& nd = t.at(i);
    nd.parent = f.parent[i]; nd.children[0] = f.child0[i]; nd.children[1] = f.child1[i]; nd.t = f.t[i]; nd.t_min = f.t_min[i]; nd.t_max = f.t_max[i];
    for (int k = f.mut_offset[i]; k < f.mut_offset[i + 1]; ++k) nd.mutations.push_back(Mutation{f.mut_from[k], f.mut_site[k], f.mut_to[k], f.mut_t[k]});
    for (int k = f.miss_offset[i]; k < f.miss_offset[i + 1]; ++k) nd.missations.intervals.v.push_back({f.miss_start[k], f.miss_end[k]});
    for (int k = f.mfs_offset[i]; k < f.mfs_offset[i + 1]; ++k) nd.missations.from_states[f.mfs_site[k]] = f.mfs_state[k];
  }
  return t;
}
static Global_evo_model hky_evo(int L, double mu, double kappa, const double pi[4]) {
  auto evo = make_single_partition_global_evo_model(L);
  Hky_model h; h.mu = mu; h.kappa = kappa; for (int a = 0; a < 4; ++a) h.pi_a[a] = pi[a];
  evo.partition_evo_model[0] = h.derive_site_evo_model();
  return evo;
}

// =================================================================================================
// interval sets (interval_set_tests.cpp) + brute-force cross-check
// =================================================================================================
TEST(interval_set_insert_and_contains) {
  Interval_set s; s.insert(3); s.insert(5); s.insert(4);                  // coalesces to [3,6)
  EXPECT(s.v == (std::vector<Site_interval>{{3, 6}}));
  s.insert({10, 12}); s.insert({6, 8});                                    // touching -> merged (closed-interval overlap)
  EXPECT(s.v == (std::vector<Site_interval>{{3, 8}, {10, 12}}));
  s.insert({4, 5});                                                        // fully contained: unchanged
  EXPECT(s.v == (std::vector<Site_interval>{{3, 8}, {10, 12}}));
  s.insert({7, 11});                                                       // bridges
  EXPECT(s.v == (std::vector<Site_interval>{{3, 12}}));
  EXPECT(s.contains(3) && s.contains(11) && !s.contains(12) && !s.contains(2));
  EXPECT(s.num_sites() == 9 && s.num_intervals() == 1);
}
TEST(interval_set_algebra_vs_bitsets) {
  emat::SplitMix64 rng(7);
  for (int iter = 0; iter < 2000; ++iter) {
    const int U = 40;
    auto rnd = [&]() { Interval_set s; int k = rng.below(5); for (int i = 0; i < k; ++i) { int a = rng.below(U), len = 1 + rng.below(6); s.insert({a, std::min(U, a + len)}); } return s; };
    Interval_set A = rnd(), B = rnd();
    EXPECT(A.is_valid(U) && B.is_valid(U));
    auto bits = [&](const Interval_set& s) { std::vector<char> b(U, 0); for (auto& [x, y] : s.v) for (int l = x; l < y; ++l) b[l] = 1; return b; };
    auto from_bits = [&](const std::vector<char>& b) { Interval_set s; for (int l = 0; l < U; ++l) if (b[l]) s.insert(l); return s; };
    auto a = bits(A), b = bits(B);
    std::vector<char> un(U), in(U), di(U); bool any = false, sub = true;
    for (int l = 0; l < U; ++l) { un[l] = a[l] | b[l]; in[l] = a[l] & b[l]; di[l] = a[l] & !b[l]; any |= in[l]; if (a[l] && !b[l]) sub = false; }
    EXPECT(merged(A, B) == from_bits(un));
    EXPECT(intersected(A, B) == from_bits(in));
    EXPECT(subtracted(A, B) == from_bits(di));
    EXPECT(interval_sets_intersect(A, B) == any);
    EXPECT(interval_set_is_subset_of(A, B) == sub);
    EXPECT(merged(A, B).no_consecutive_intervals());
  }
}
// missation_map_tests.cpp: from_states only hold deltas from the reference; ref_seq_changed keeps that true
TEST(missation_map_semantics) {
  std::vector<State> ref{sA, sC, sG, sT, sA};
  Missation_map m; m.insert(1, sC, ref); m.insert(2, sA, ref); m.insert(4, sA, ref);
  EXPECT(m.intervals.v == (std::vector<Site_interval>{{1, 3}, {4, 5}}));
  EXPECT(m.from_states.size() == 1 && m.from_states.at(2) == sA);
  EXPECT(m.get_from_state(1, ref) == sC && m.get_from_state(2, ref) == sA);
  m.set_from_state(2, sG, ref); EXPECT(m.from_states.empty());
  m.ref_seq_changed(1, sC, sT); EXPECT(m.from_states.at(1) == sC);        // not a delta before, is one now
  m.ref_seq_changed(1, sT, sC); EXPECT(m.from_states.empty());
  m.ref_seq_changed(3, sT, sA); EXPECT(m.from_states.empty());            // site not missing: ignored
  Missation_map A, B, C; A.insert(1, sC, ref); A.insert(2, sA, ref); B.insert(2, sA, ref); B.insert(3, sT, ref);
  factor_out_common_missations(A, B, C);
  EXPECT(A.intervals.v == (std::vector<Site_interval>{{1, 2}}) && B.intervals.v == (std::vector<Site_interval>{{3, 4}}) && C.intervals.v == (std::vector<Site_interval>{{2, 3}}));
  EXPECT(C.from_states.size() == 1 && C.from_states.at(2) == sA && A.from_states.empty() && B.from_states.empty());
  auto M = merge_missations_nondestructively(A, C);
  EXPECT(M.intervals.v == (std::vector<Site_interval>{{1, 3}}) && M.from_states.at(2) == sA);
}
// missation_map_tests.cpp:17-124 with the reference's own cases (subtract_missations_nondestructively, :126-137, is not on the path)
static std::vector<std::pair<int, State>> elements_of(const Missation_map& m, const std::vector<State>& ref) {   // slow_elements(ref_seq)
  std::vector<std::pair<int, State>> out;
  for (const auto& iv : m.intervals.v) for (int l = iv.first; l < iv.second; ++l) out.push_back({l, m.get_from_state(l, ref)});
  return out;
}
TEST(missation_map_reference_cases) {
  { Missation_map m; EXPECT(m.empty()); }                                                   // empty
  { std::vector<State> ref{sA, sC, sG, sT};                                                 // simple
    Missation_map m; m.insert(0, sA, ref); m.insert(1, sG, ref);
    EXPECT(m.intervals.v == (std::vector<Site_interval>{{0, 2}}));
    EXPECT(m.from_states.size() == 1 && m.from_states.at(1) == sG); }
  { std::vector<State> ref(7, sA);                                                          // missation_map_basics
    Missation_map m; m.insert(4, sA, ref); m.insert(6, sC, ref); m.insert(2, sG, ref); m.insert(4, sA, ref);
    EXPECT(m.num_intervals() == 3 && m.num_sites() == 3);
    EXPECT(m.intervals.v == (std::vector<Site_interval>{{2, 3}, {4, 5}, {6, 7}}));
    EXPECT(m.from_states == (std::map<Site_index, State>{{2, sG}, {6, sC}}));
    EXPECT(m.get_from_state(2, ref) == sG && m.get_from_state(4, ref) == sA && m.get_from_state(6, ref) == sC);
    m.insert(3, sT, ref);
    EXPECT(m.num_intervals() == 2 && m.num_sites() == 4);
    EXPECT(m.intervals.v == (std::vector<Site_interval>{{2, 5}, {6, 7}}));
    EXPECT(m.from_states == (std::map<Site_index, State>{{2, sG}, {3, sT}, {6, sC}}));
    EXPECT(m.contains(3) && !m.contains(5));
    ref[2] = sG; m.ref_seq_changed(2, sA, sG);
    ref[3] = sC; m.ref_seq_changed(3, sA, sC);
    ref[4] = sT; m.ref_seq_changed(4, sA, sT);
    ref[5] = sC; m.ref_seq_changed(5, sA, sC);
    ref[6] = sC; m.ref_seq_changed(6, sA, sC);
    EXPECT(m.num_intervals() == 2 && m.num_sites() == 4);
    EXPECT(m.intervals.v == (std::vector<Site_interval>{{2, 5}, {6, 7}}));
    EXPECT(m.from_states == (std::map<Site_index, State>{{3, sT}, {4, sA}})); }
  { std::vector<State> ref{sT, sT, sT};                                                     // factor_out_common_missations
    Missation_map A, B, C; A.insert(0, sA, ref); A.insert(1, sC, ref); B.insert(0, sA, ref); B.insert(2, sG, ref);
    factor_out_common_missations(A, B, C);
    EXPECT(elements_of(A, ref) == (std::vector<std::pair<int, State>>{{1, sC}}));
    EXPECT(elements_of(B, ref) == (std::vector<std::pair<int, State>>{{2, sG}}));
    EXPECT(elements_of(C, ref) == (std::vector<std::pair<int, State>>{{0, sA}})); }
  { std::vector<State> ref{sA, sA, sA};                                                     // merge_missations_nondestructively
    Missation_map A, B; A.insert(0, sA, ref); A.insert(1, sC, ref); B.insert(2, sG, ref);
    auto M = merge_missations_nondestructively(A, B);
    EXPECT(elements_of(M, ref) == (std::vector<std::pair<int, State>>{{0, sA}, {1, sC}, {2, sG}})); }
}
// site_deltas_tests.cpp:115-277 and :297-314 with the reference's own sequences of pushes and pops: the start of a delta list is moved
// step by step along  ATA -A0T- -T1A- -A2G- -T0A- -A1C- ACG  and back
TEST(site_deltas_reference_push_pop_append) {
  using SD = std::map<Site_index, Site_delta>;
  auto is = [](const Site_deltas& d, const SD& want) { return SD(d.begin(), d.end()) == want; };
  { Site_deltas d;                                                                       // push_pop_front_site_deltas
    push_front_site_deltas({1, sA, sC}, d); EXPECT(is(d, {{1, {sA, sC}}}));
    push_front_site_deltas({0, sT, sA}, d); EXPECT(is(d, {{0, {sT, sA}}, {1, {sA, sC}}}));
    push_front_site_deltas({2, sA, sG}, d); EXPECT(is(d, {{0, {sT, sA}}, {1, {sA, sC}}, {2, {sA, sG}}}));
    push_front_site_deltas({1, sT, sA}, d); EXPECT(is(d, {{0, {sT, sA}}, {1, {sT, sC}}, {2, {sA, sG}}}));
    push_front_site_deltas({0, sA, sT}, d); EXPECT(is(d, {{1, {sT, sC}}, {2, {sA, sG}}}));
    pop_front_site_deltas({0, sA, sT}, d); EXPECT(is(d, {{0, {sT, sA}}, {1, {sT, sC}}, {2, {sA, sG}}}));
    pop_front_site_deltas({1, sT, sA}, d); EXPECT(is(d, {{0, {sT, sA}}, {1, {sA, sC}}, {2, {sA, sG}}}));
    pop_front_site_deltas({2, sA, sG}, d); EXPECT(is(d, {{0, {sT, sA}}, {1, {sA, sC}}}));
    pop_front_site_deltas({0, sT, sA}, d); EXPECT(is(d, {{1, {sA, sC}}}));
    pop_front_site_deltas({1, sA, sC}, d); EXPECT(d.empty()); }
  { Site_deltas d;                                                                       // push_pop_back_site_deltas
    push_back_site_deltas({0, sA, sT}, d); EXPECT(is(d, {{0, {sA, sT}}}));
    push_back_site_deltas({1, sT, sA}, d); EXPECT(is(d, {{0, {sA, sT}}, {1, {sT, sA}}}));
    push_back_site_deltas({2, sA, sG}, d); EXPECT(is(d, {{0, {sA, sT}}, {1, {sT, sA}}, {2, {sA, sG}}}));
    push_back_site_deltas({0, sT, sA}, d); EXPECT(is(d, {{1, {sT, sA}}, {2, {sA, sG}}}));
    push_back_site_deltas({1, sA, sC}, d); EXPECT(is(d, {{1, {sT, sC}}, {2, {sA, sG}}}));
    pop_back_site_deltas({1, sA, sC}, d); EXPECT(is(d, {{1, {sT, sA}}, {2, {sA, sG}}}));
    pop_back_site_deltas({0, sT, sA}, d); EXPECT(is(d, {{0, {sA, sT}}, {1, {sT, sA}}, {2, {sA, sG}}}));
    pop_back_site_deltas({2, sA, sG}, d); EXPECT(is(d, {{0, {sA, sT}}, {1, {sT, sA}}}));
    pop_back_site_deltas({1, sT, sA}, d); EXPECT(is(d, {{0, {sA, sT}}}));
    pop_back_site_deltas({0, sA, sT}, d); EXPECT(d.empty()); }
  { Site_deltas d1, d2;                                                                  // append_site_deltas
    d1.insert({0, {sA, sT}}); d1.insert({2, {sC, sG}});
    d2.insert({0, {sT, sA}}); d2.insert({1, {sG, sA}}); d2.insert({2, {sG, sT}});
    append_site_deltas(d1, d2);
    EXPECT(is(d1, {{1, {sG, sA}}, {2, {sC, sT}}})); }
}
// site_deltas_tests.cpp:14-113 (fixture: ref AAAA, no missing data) and :279-295, :316-405 with the reference's own table of deltas
static Phylo_tree site_deltas_fixture() {
  Phylo_tree t(5);
  t.root = r_; t.ref_sequence = {sA, sA, sA, sA};
  set_inner(t, r_, k_no_node, x_, c_, -1.0);
  set_inner(t, x_, r_, a_, b_, 0.0); t.at(x_).mutations = {Mutation{sA, 0, sT, -0.5}};
  set_tip(t, a_, x_, 1.0); t.at(a_).mutations = {Mutation{sT, 0, sC, 0.5}};
  set_tip(t, b_, x_, 2.0); t.at(b_).mutations = {Mutation{sA, 1, sG, 1.0}};
  set_tip(t, c_, r_, 3.0); t.at(c_).mutations = {Mutation{sA, 0, sT, 0.0}, Mutation{sT, 0, sG, 1.0}};
  return t;
}
TEST(site_deltas_reference_fixture_cases) {
  using SD = std::map<Site_index, Site_delta>;
  auto t = site_deltas_fixture();
  auto is = [](const Site_deltas& d, const SD& want) { return SD(d.begin(), d.end()) == want; };
  // displace_site_delta_starts: up to the root the deltas turn the reference into the node's sequence; down again they cancel
  const std::vector<State> seqs[5] = {{sA, sA, sA, sA}, {sT, sA, sA, sA}, {sC, sA, sA, sA}, {sT, sG, sA, sA}, {sG, sA, sA, sA}};   // r x a b c
  for (int i : {a_, b_, c_, x_}) {
    Site_deltas d;
    displace_site_deltas_start_upwards(t, d, t.node_loc(i), t.node_loc(r_));
    auto seq = t.ref_sequence;
    for (auto& [l, dl] : d) { EXPECT(dl.from != dl.to && dl.from == seq[l]); seq[l] = dl.to; }
    EXPECT(seq == seqs[i]);
    displace_site_deltas_start_downwards(t, d, t.node_loc(r_), t.node_loc(i));
    EXPECT(d.empty());
  }
  // calc_site_deltas_between, every ordered pair of nodes
  const SD none{};
  const SD want[5][5] = {   // [from][to], order r x a b c
      /* r */ {none, {{0, {sA, sT}}}, {{0, {sA, sC}}}, {{0, {sA, sT}}, {1, {sA, sG}}}, {{0, {sA, sG}}}},
      /* x */ {{{0, {sT, sA}}}, none, {{0, {sT, sC}}}, {{1, {sA, sG}}}, {{0, {sT, sG}}}},
      /* a */ {{{0, {sC, sA}}}, {{0, {sC, sT}}}, none, {{0, {sC, sT}}, {1, {sA, sG}}}, {{0, {sC, sG}}}},
      /* b */ {{{0, {sT, sA}}, {1, {sG, sA}}}, {{1, {sG, sA}}}, {{0, {sT, sC}}, {1, {sG, sA}}}, none, {{0, {sT, sG}}, {1, {sG, sA}}}},
      /* c */ {{{0, {sG, sA}}}, {{0, {sG, sT}}}, {{0, {sG, sC}}}, {{0, {sG, sT}}, {1, {sA, sG}}}, none}};
  for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) EXPECT(is(calc_site_deltas_between(t, i, j), want[i][j]));
  // a few tricky tree locations
  EXPECT(is(calc_site_deltas_between(t, Phylo_tree_loc{r_, -2.0}, Phylo_tree_loc{c_, 0.5}), {{0, {sA, sT}}}));
  EXPECT(is(calc_site_deltas_between(t, Phylo_tree_loc{c_, 0.5}, Phylo_tree_loc{x_, 0.0}), none));
  EXPECT(is(calc_site_deltas_between(t, Phylo_tree_loc{b_, 1.5}, Phylo_tree_loc{c_, 0.5}), {{1, {sG, sA}}}));
}
// site_deltas_tests.cpp: composition and cancellation
TEST(site_deltas_algebra) {
  Site_deltas d;
  push_back_site_deltas({3, sA, sC}, d); push_back_site_deltas({3, sC, sG}, d);
  EXPECT(d.size() == 1 && d.at(3) == (Site_delta{sA, sG}));
  push_back_site_deltas({3, sG, sA}, d); EXPECT(d.empty());
  push_front_site_deltas({5, sT, sA}, d); push_front_site_deltas({5, sC, sT}, d);
  EXPECT(d.at(5) == (Site_delta{sC, sA}));
  pop_front_site_deltas({5, sC, sT}, d); EXPECT(d.at(5) == (Site_delta{sT, sA}));
  pop_back_site_deltas({5, sT, sA}, d); EXPECT(d.empty());
  auto t = complex_tree(false);
  auto xa = calc_site_deltas_between(t, x_, a_);
  EXPECT(xa.size() == 1 && xa.at(0) == (Site_delta{sT, sC}));
  auto ca = calc_site_deltas_between(t, c_, a_);   // G... -> C...
  EXPECT(ca.size() == 1 && ca.at(0) == (Site_delta{sG, sC}));
  auto cb = calc_site_deltas_between(t, c_, b_);
  EXPECT(cb.size() == 2 && cb.at(0) == (Site_delta{sG, sT}) && cb.at(1) == (Site_delta{sA, sG}));
}

// =================================================================================================
// phylo_tree_calc on the 5-node fixture (phylo_tree_calc_tests.cpp:118-470)
// =================================================================================================
TEST(calc_fixture_sequences_and_missing) {
  auto t = complex_tree(false);
  EXPECT(check_phylo_tree_integrity(t).empty());
  auto S = [&](int n, double tt) { return view_of_sequence_at(t, Phylo_tree_loc{n, tt}); };
  using V = std::vector<State>;
  EXPECT(S(r_, -1.5) == (V{sA, sA, sA, sA})); EXPECT(S(x_, -0.8) == (V{sA, sA, sA, sA})); EXPECT(S(x_, 0.0) == (V{sT, sA, sA, sA}));
  EXPECT(S(a_, 0.2) == (V{sT, sA, sA, sA})); EXPECT(S(a_, 1.0) == (V{sC, sA, sA, sA})); EXPECT(S(b_, 2.0) == (V{sT, sG, sA, sA}));
  EXPECT(S(c_, -0.5) == (V{sA, sA, sA, sA})); EXPECT(S(c_, 0.5) == (V{sT, sA, sA, sA})); EXPECT(S(c_, 1.5) == (V{sG, sA, sA, sA}));
  EXPECT(reconstruct_missing_sites_at(t, r_).v == (std::vector<Site_interval>{{3, 4}}));
  EXPECT(reconstruct_missing_sites_at(t, x_).v == (std::vector<Site_interval>{{2, 4}}));
  EXPECT(reconstruct_missing_sites_at(t, b_).v == (std::vector<Site_interval>{{2, 4}}));
  EXPECT(reconstruct_missing_sites_at(t, c_).v == (std::vector<Site_interval>{{1, 2}, {3, 4}}));
  EXPECT(!is_site_missing_at(t, r_, 2) && is_site_missing_at(t, r_, 3) && is_site_missing_at(t, a_, 2) && is_site_missing_at(t, c_, 1) && !is_site_missing_at(t, c_, 2));
  EXPECT(calc_site_state_at(t, t.node_loc(a_), 0) == sC && calc_site_state_at(t, t.node_loc(b_), 1) == sG && calc_site_state_at(t, t.node_loc(a_), 3) == sA);
  EXPECT(calc_num_sites_missing_at_every_node(t) == (std::vector<int>{1, 2, 2, 2, 2}));
  EXPECT_NEAR(calc_T(t), 8.0, 1e-12);
  EXPECT(calc_num_muts(t) == 5);
}
// phylo_tree_calc_tests.cpp:248-284 (calc_Ttwiddle_beta_a), :441-469 (calc_num_muts, _ab, _beta_ab)
TEST(calc_fixture_global_move_statistics) {
  auto t = complex_tree(false);
  auto evo = fixture_evo();
  auto nu = [&](int l) { return evo.nu_l[l]; };
  double expected[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  // site 0, partition 0
  expected[0][sA] += 0.5 * nu(0); expected[0][sT] += 0.5 * nu(0);   // r->x branch up to / from A0T
  expected[0][sT] += 0.5 * nu(0); expected[0][sC] += 0.5 * nu(0);   // x->a branch to / from T0C
  expected[0][sT] += 2.0 * nu(0);                                   // x->b branch
  expected[0][sA] += 1.0 * nu(0); expected[0][sT] += 1.0 * nu(0); expected[0][sG] += 2.0 * nu(0);   // r->c branch: A0T, T0G
  // site 1, partition 1 (missing on r->c)
  expected[1][sA] += 1.0 * nu(1); expected[1][sA] += 1.0 * nu(1); expected[1][sA] += 1.0 * nu(1); expected[1][sG] += 1.0 * nu(1);
  // site 2, partition 0 (missing below r->x); site 3 missing everywhere
  expected[0][sA] += 4.0 * nu(2);
  auto T = calc_Ttwiddle_beta_a(t, evo);
  EXPECT(T.size() == 2);
  for (int b = 0; b < 2; ++b) for (int a = 0; a < 4; ++a) EXPECT_NEAR(T[b][a], expected[b][a], 1e-12);
  auto M = calc_num_muts_beta_ab(t, evo);
  long long em[2][4][4] = {};
  ++em[0][sA][sT]; ++em[0][sT][sC]; ++em[1][sA][sG]; ++em[0][sA][sT]; ++em[0][sT][sG];
  long long total = 0;
  for (int b = 0; b < 2; ++b) for (int a = 0; a < 4; ++a) for (int c2 = 0; c2 < 4; ++c2) { EXPECT(M[b][a][c2] == em[b][a][c2]); total += M[b][a][c2]; }
  EXPECT(total == calc_num_muts(t));
  // phylo_tree_calc_tests.cpp:471-482: A0T on r->x, T0C on x->a, A0T and T0G on r->c; A1G on x->b; C2A above the root is not a mutation
  EXPECT(calc_num_muts_l(t) == (std::vector<int>{4, 1, 0, 0}));
  // phylo_tree_calc_tests.cpp:286-313 (calc_T_l_a) and :315-326 (calc_Ttwiddle_l = sum_a q^(l)_a T^(l)_a)
  double eT[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  eT[0][sA] += 0.5; eT[0][sT] += 0.5; eT[0][sT] += 0.5; eT[0][sC] += 0.5; eT[0][sT] += 2.0; eT[0][sA] += 1.0; eT[0][sT] += 1.0; eT[0][sG] += 2.0;
  eT[1][sA] += 1.0; eT[1][sA] += 1.0; eT[1][sA] += 1.0; eT[1][sG] += 1.0;   // site 1 missing below the r->c attachment point
  eT[2][sA] += 4.0;                                                          // site 2 missing below r->x; site 3 missing everywhere
  auto T_l_a = calc_T_l_a(t);
  auto Ttw = calc_Ttwiddle_l(t, evo);
  for (int l = 0; l < 4; ++l) {
    double expect_tw = 0.0;
    for (int a2 = 0; a2 < 4; ++a2) { EXPECT_NEAR(T_l_a[l][a2], eT[l][a2], 1e-6); expect_tw += evo.q_l_a(l, (State)a2) * eT[l][a2]; }
    EXPECT_NEAR(Ttw[l], expect_tw, 1e-6);
  }
}
TEST(calc_fixture_log_G_and_lambda) {
  auto t = complex_tree(false);
  auto evo = fixture_evo();
  auto mu = [&](int l) { return evo.mu_l(l); }; auto nu = [&](int l) { return evo.nu_l[l]; };
  auto qa = [&](int l, State a) { return evo.q_l_a(l, a); }; auto qab = [&](int l, State a, State b) { return evo.q_l_ab(l, a, b); };
  // phylo_tree_calc_tests.cpp:381-439
  double expected = 0.0;
  expected += -mu(0) * nu(0) * qa(0, sA) * 0.5 + std::log(mu(0) * nu(0) * qab(0, sA, sT)) + -mu(0) * nu(0) * qa(0, sT) * 0.5
      + -mu(0) * nu(0) * qa(0, sT) * 0.5 + std::log(mu(0) * nu(0) * qab(0, sT, sC)) + -mu(0) * nu(0) * qa(0, sC) * 0.5
      + -mu(0) * nu(0) * qa(0, sT) * 2.0
      + -mu(0) * nu(0) * qa(0, sA) * 1.0 + std::log(mu(0) * nu(0) * qab(0, sA, sT)) + -mu(0) * nu(0) * qa(0, sT) * 1.0
      + std::log(mu(0) * nu(0) * qab(0, sT, sG)) + -mu(0) * nu(0) * qa(0, sG) * 2.0;
  expected += -mu(1) * nu(1) * qa(1, sA) * 1.0 + -mu(1) * nu(1) * qa(1, sA) * 1.0
      + -mu(1) * nu(1) * qa(1, sA) * 1.0 + std::log(mu(1) * nu(1) * qab(1, sA, sG)) + -mu(1) * nu(1) * qa(1, sG) * 1.0;
  expected += -mu(2) * nu(2) * qa(2, sA) * 4.0;
  EXPECT_NEAR(calc_log_G_below_root(t, evo), expected, 1e-9);
  auto B = brute_force(t, evo);
  EXPECT_NEAR(B.log_G_below_root, expected, 1e-9);
  auto cumQ = calc_cum_Q_l_for_sequence(t.ref_sequence, evo);
  // cum_Q over ref AACA
  EXPECT_NEAR(cumQ[4], mu(0) * nu(0) * qa(0, sA) + mu(1) * nu(1) * qa(1, sA) + mu(2) * nu(2) * qa(2, sC) + mu(3) * nu(3) * qa(3, sA), 1e-12);
  auto li = calc_lambda_i(t, evo, cumQ);
  // lambda at each node = sum over present sites of mu nu q_state
  EXPECT_NEAR(li[r_], mu(0) * nu(0) * qa(0, sA) + mu(1) * nu(1) * qa(1, sA) + mu(2) * nu(2) * qa(2, sA), 1e-12);
  EXPECT_NEAR(li[x_], mu(0) * nu(0) * qa(0, sT) + mu(1) * nu(1) * qa(1, sA), 1e-12);
  EXPECT_NEAR(li[a_], mu(0) * nu(0) * qa(0, sC) + mu(1) * nu(1) * qa(1, sA), 1e-12);
  EXPECT_NEAR(li[b_], mu(0) * nu(0) * qa(0, sT) + mu(1) * nu(1) * qa(1, sG), 1e-12);
  EXPECT_NEAR(li[c_], mu(0) * nu(0) * qa(0, sG) + mu(2) * nu(2) * qa(2, sA), 1e-12);
  for (int n = 0; n < 5; ++n) { EXPECT_NEAR(li[n], B.lambda_i[n], 1e-12); EXPECT_NEAR(calc_lambda_at_node(t, n, evo, cumQ), li[n], 1e-12); }
  // recalc upstream reproduces the table
  auto li2 = li; li2[x_] = li2[r_] = -1; recalc_lambda_i_upstream(t, a_, k_no_node, evo, li2, cumQ);
  EXPECT_NEAR(li2[x_], li[x_], 1e-12); EXPECT_NEAR(li2[r_], li[r_], 1e-12);
  // root prior (phylo_tree_calc_tests.cpp:341-379): sites 0,1,2 are A at the root, site 3 missing
  EXPECT_NEAR(calc_log_root_prior(t, evo), std::log(evo.pi_l_a(0, sA)) + std::log(evo.pi_l_a(1, sA)) + std::log(evo.pi_l_a(2, sA)), 1e-12);
  auto evo2 = evo;
  double p0[4] = {0.3, 0.7, 0.0, 0.0}, p1[4] = {0.3, 0.0, 0.7, 0.0};
  for (int a = 0; a < 4; ++a) { evo2.partition_evo_model[0].pi_a[a] = p0[a]; evo2.partition_evo_model[1].pi_a[a] = p1[a]; }
  EXPECT_NEAR(calc_log_root_prior(t, evo2), 3 * std::log(0.3), 1e-12);
  double z0[4] = {0.0, 0.3, 0.7, 0.0};
  for (int a = 0; a < 4; ++a) evo2.partition_evo_model[0].pi_a[a] = z0[a];
  EXPECT(std::isinf(calc_log_root_prior(t, evo2)) && calc_log_root_prior(t, evo2) < 0);
  // path log G = sum of branch terms
  auto rf = calc_state_frequencies_per_partition_of(t.ref_sequence, evo);
  EXPECT_NEAR(calc_path_log_G(t, r_, a_, evo, li, rf),
              calc_branch_log_G(t, a_, li[a_], evo, rf) + calc_branch_log_G(t, x_, li[x_], evo, rf), 1e-12);
}
// phylo_tree_calc_tests.cpp:139-143 (a reversion on the way to a tip "used to crash") and :773-781 (latest tip time, with an uncertain tip)
TEST(calc_reversion_and_max_tip_time) {
  auto t = complex_tree(false);
  t.at(a_).mutations = {Mutation{sT, 0, sA, 0.5}};   // r -- A0T -- x -- T0A -- a
  auto seq = view_of_sequence_at(t, a_);
  EXPECT(seq == (std::vector<State>{sA, sA, sA, sA}));   // (this variant of the fixture has reference AACA and the root delta C2A: AAAA at the root, as in the reference's test)
  auto u = complex_tree(false);
  EXPECT(calc_max_tip_time(u) == u.at(c_).t);
  u.at(c_).t_max = (float)(u.at(c_).t + 10.0);
  EXPECT(calc_max_tip_time(u) != u.at(c_).t && calc_max_tip_time(u) == (double)u.at(c_).t_max);
}
TEST(calc_state_frequencies) {   // phylo_tree_calc_tests.cpp:179-189
  std::vector<State> seq{sA, sC, sC, sG, sG, sG, sT, sT, sT, sT};
  auto evo = make_global_evo_model({0, 1, 0, 1, 2, 0, 1, 2, 3, 0});
  auto f = calc_state_frequencies_per_partition_of(seq, evo);
  EXPECT((f[0] == std::array<int, 4>{1, 1, 1, 1}) && (f[1] == std::array<int, 4>{0, 1, 1, 1}) && (f[2] == std::array<int, 4>{0, 0, 1, 1}) && (f[3] == std::array<int, 4>{0, 0, 0, 1}));
}
TEST(calc_vs_bruteforce_on_synthetic_trees) {
  for (int seed = 0; seed < 6; ++seed) {
    emat::SynthParams p; p.num_tips = 40 + 10 * seed; p.num_sites = 300; p.mu = 2e-4; p.gaps_per_tip = 3; p.mean_gap_len = 12; p.seed = 100 + seed;
    auto R = emat::make_synthetic_emat(p);
    auto t = tree_from_flat(R.tree, R.ref_sequence);
    auto msg = check_phylo_tree_integrity(t);
    if (!msg.empty()) std::printf("  integrity: %s\n", msg.c_str());
    EXPECT(msg.empty());
    auto evo = hky_evo(p.num_sites, p.mu, p.kappa, p.pi);
    for (int l = 0; l < p.num_sites; ++l) evo.nu_l[l] = 0.5 + (l % 7) * 0.25;
    auto cumQ = calc_cum_Q_l_for_sequence(t.ref_sequence, evo);
    auto li = calc_lambda_i(t, evo, cumQ);
    auto B = brute_force(t, evo);
    for (int n = 0; n < t.size(); ++n) EXPECT_NEAR(li[n], B.lambda_i[n], 1e-9 * std::fabs(B.lambda_i[n]) + 1e-12);
    EXPECT_NEAR(calc_log_G_below_root(t, evo), B.log_G_below_root, 1e-9 * std::fabs(B.log_G_below_root));
    EXPECT(calc_num_sites_missing_at_every_node(t) == B.num_missing);
  }
}

// =================================================================================================
// SPR study: bit-exact candidate regions (spr_study_tests.cpp:84-205)
// =================================================================================================
struct Reg { int branch, mut_idx; double t_min, t_max; int min_muts; };
static bool same_regions(const std::vector<Candidate_region>& got, std::vector<Reg> want) {
  if (got.size() != want.size()) { std::printf("  region count %zu != %zu\n", got.size(), want.size()); return false; }
  for (auto& g : got) {
    bool found = false;
    for (auto it = want.begin(); it != want.end(); ++it)
      if (it->branch == g.branch && it->mut_idx == g.mut_idx && it->t_min == g.t_min && it->t_max == g.t_max && it->min_muts == g.min_muts) { want.erase(it); found = true; break; }
    if (!found) { std::printf("  unexpected region b=%d mi=%d [%g,%g] m=%d\n", g.branch, g.mut_idx, g.t_min, g.t_max, g.min_muts); return false; }
  }
  return true;
}
TEST(spr_study_regions) {
  auto t = complex_tree(false);
  { Interval_set none; Spr_study_builder b{t, a_, 5.0, none}; EXPECT(b.result.empty()); }
  { auto sd = calc_site_deltas_between(t, x_, a_); auto mx = reconstruct_missing_sites_at(t, a_);
    Spr_study_builder b{t, a_, 1.5, mx}; b.seed_fill_from(b_, 0, sd, true);
    EXPECT(same_regions(b.result, {{b_, 1, -0.5, 1.0, 1}, {b_, 2, 1.0, 1.5, 2}, {b_, 0, -1.0, -0.5, 1}, {r_, 1, NEG, -1.0, 1}, {c_, 0, -1.0, 0.0, 1}, {c_, 1, 0.0, 1.0, 1}, {c_, 2, 1.0, 1.5, 1}})); }
  { auto sd = calc_site_deltas_between(t, x_, a_); auto mx = reconstruct_missing_sites_at(t, a_);
    Spr_study_builder b{t, a_, 1.5, mx}; b.seed_fill_from(b_, 0, sd, false);
    EXPECT(same_regions(b.result, {{b_, 1, -0.5, 1.0, 1}, {b_, 2, 1.0, 1.5, 2}, {b_, 0, -1.0, -0.5, 1}, {c_, 0, -1.0, 0.0, 1}, {c_, 1, 0.0, 1.0, 1}, {c_, 2, 1.0, 1.5, 1}})); }
  { auto sd = calc_site_deltas_between(t, x_, a_); auto mx = reconstruct_missing_sites_at(t, a_);
    Spr_study_builder b{t, a_, 1.5, mx}; b.max_muts_from_start = 1; b.seed_fill_from(b_, 0, sd, true);
    EXPECT(same_regions(b.result, {{b_, 1, -0.5, 1.0, 1}, {b_, 2, 1.0, 1.5, 2}, {b_, 0, -1.0, -0.5, 1}, {r_, 1, NEG, -1.0, 1}, {c_, 0, -1.0, 0.0, 1}})); }
  { auto sd = calc_site_deltas_between(t, c_, x_); auto mx = reconstruct_missing_sites_at(t, x_);
    Spr_study_builder b{t, x_, t.at(x_).t, mx}; b.seed_fill_from(c_, (int)t.at(c_).mutations.size(), sd, true);
    EXPECT(same_regions(b.result, {{c_, 3, NEG, 0.0, 1}})); }
  { auto sd = calc_site_deltas_between(t, x_, c_); auto mx = reconstruct_missing_sites_at(t, c_);
    Spr_study_builder b{t, c_, t.at(c_).t, mx}; b.seed_fill_from(x_, (int)t.at(x_).mutations.size(), sd, true);
    EXPECT(same_regions(b.result, {{x_, 2, NEG, 0.0, 1}, {b_, 0, 0.0, 1.0, 1}, {b_, 1, 1.0, 2.0, 1}, {a_, 0, 0.0, 0.5, 1}, {a_, 1, 0.5, 1.0, 1}})); }
  { Site_deltas sd{{0, Site_delta{sA, sT}}}; Interval_set mx; mx.insert({3, 4});
    Spr_study_builder b{t, k_no_node, 1.5, mx}; b.seed_fill_from(t.root, (int)t.at_root().mutations.size(), sd, true);
    EXPECT(same_regions(b.result, {{a_, 0, 0.0, 0.5, 0}, {a_, 1, 0.5, 1.0, 1}, {b_, 0, 0.0, 1.0, 0}, {b_, 1, 1.0, 1.5, 1}, {x_, 1, -0.5, 0.0, 0}, {x_, 0, -1.0, -0.5, 1},
                                   {r_, 1, NEG, -1.0, 1}, {c_, 0, -1.0, 0.0, 1}, {c_, 1, 0.0, 1.0, 0}, {c_, 2, 1.0, 1.5, 1}})); }
}
// study weights: formulas of spr_study.cpp:318-385 evaluated independently; sampling consistency
TEST(spr_study_weights_and_sampling) {
  auto t = complex_tree(false);
  auto sd = calc_site_deltas_between(t, x_, a_); auto mx = reconstruct_missing_sites_at(t, a_);
  Spr_study_builder b{t, a_, 1.5, mx}; b.seed_fill_from(b_, 0, sd, true);
  double lambda_X = 0.37, f = 0.8, t_X = 1.5, t_max_tip = 3.0;
  Spr_study st{std::move(b), lambda_X, f, t_X, t_max_tip};
  double mu = lambda_X / (4 - 2);
  double sum = 0.0;
  for (auto& r : st.candidate_regions) {
    double lw;
    if (!r.is_above_root()) { double tp = 0.5 * (r.t_min + r.t_max); lw = std::log(f * lambda_X * (r.t_max - r.t_min)) + f * (-lambda_X * (t_X - tp) + r.min_muts * std::log(mu * (t_X - tp) / 3)); }
    else {
      double t_S = t.at(r.branch).t, s_min = std::fabs(t_X - t_S), s_max = s_min + 20.0 * (t_max_tip - std::min(t_X, t_S));
      double xmin = lambda_X * f * s_min, xmax = lambda_X * f * s_max, al = f * r.min_muts + 1;
      lw = -M_LN2 + f * r.min_muts * std::log(mu / (3 * lambda_X * f)) + std::lgamma(al) + std::log(gamma_q(al, xmin) - gamma_q(al, xmax));
    }
    EXPECT_NEAR(r.log_W_over_Wmax + st.log_Wmax, lw, 1e-10);
    sum += std::exp(lw - st.log_Wmax);
  }
  EXPECT_NEAR(sum, st.sum_W_over_Wmax, 1e-12);
  // log_alpha integrates to 1 over the below-root regions plus the root region's mass
  double total = 0.0;
  for (int i = 0; i < (int)st.candidate_regions.size(); ++i) {
    auto& r = st.candidate_regions[i];
    if (!r.is_above_root()) total += std::exp(st.log_alpha_in_region(i, r.t_max)) * (r.t_max - r.t_min);
    else {   // numeric integral over t of exp(log_alpha)
      double t_S = t.at(r.branch).t, s_min = std::fabs(t_X - t_S), s_max = s_min + 20.0 * (t_max_tip - std::min(t_X, t_S));
      double lo = 0.5 * (t_X + t_S - s_max), hi = r.t_max; int n = 200000; double h = (hi - lo) / n, acc = 0.0;
      for (int k = 0; k < n; ++k) acc += std::exp(st.log_alpha_in_region(i, lo + (k + 0.5) * h)) * h;
      total += acc;
    }
  }
  EXPECT_NEAR(total, 1.0, 1e-5);
  Rng rng; rng.key = 99;
  for (int k = 0; k < 2000; ++k) {
    int ri = st.pick_nexus_region(rng); double tt = st.pick_time_in_region(ri, rng);
    auto& r = st.candidate_regions[ri];
    EXPECT(r.t_min <= tt && tt <= r.t_max);
    EXPECT(st.find_region(r.branch, tt) == ri || tt == r.t_min);
  }
}

// =================================================================================================
// SPR graft analysis on the reference fixtures (spr_move_tests.cpp:142-470, 1261-1516)
// =================================================================================================
static const double mu_JC = 0.125;
static double P_JC(State a, State b, double t) { return a == b ? 1.0 + 3. / 4 * std::expm1(-4. / 3. * mu_JC * t) : -1. / 4. * std::expm1(-4. / 3. * mu_JC * t); }
struct SprCtx {
  Phylo_tree tree; Global_evo_model evo; std::vector<double> cumQ, lambda_i; std::vector<int> nm;
  SprCtx(Phylo_tree t, Global_evo_model e) : tree(std::move(t)), evo(std::move(e)) { refresh(); }
  void refresh() { cumQ = calc_cum_Q_l_for_sequence(tree.ref_sequence, evo); lambda_i = calc_lambda_i(tree, evo, cumQ); nm = calc_num_sites_missing_at_every_node(tree); }
  Spr_move spr(bool ccr = true) { return Spr_move{tree, mu_JC, ccr, evo, lambda_i, cumQ, nm}; }
};
TEST(spr_move_analyze_graft_simple) {
  SprCtx C(complex_tree(true), fixture_evo());
  auto& evo = C.evo;
  auto mu = [&](int l) { return evo.mu_l(l); }; auto nu = [&](int l) { return evo.nu_l[l]; };
  auto qa = [&](int l, State a) { return evo.q_l_a(l, a); }; auto qab = [&](int l, State a, State b) { return evo.q_l_ab(l, a, b); };
  auto spr = C.spr();
  { auto g = spr.analyze_graft(a_);   // spr_move_tests.cpp:142-199
    EXPECT(g.branch_infos.size() == 1);
    auto& b0 = g.branch_infos[0];
    EXPECT(b0.A == x_ && b0.B == a_ && !b0.is_open && b0.T_to_X == 1.0);
    EXPECT_NEAR(b0.partial_lambda_at_A, mu(0) * nu(0) * qa(0, sT) + mu(1) * nu(1) * qa(1, sA), 1e-9);
    EXPECT_NEAR(b0.partial_lambda_at_X, mu(0) * nu(0) * qa(0, sC) + mu(1) * nu(1) * qa(1, sA), 1e-9);
    EXPECT(b0.hot_sites.contains(0) && b0.hot_sites.contains(1));
    EXPECT(b0.hot_muts_to_X == (Mutation_list{Mutation{sT, 0, sC, 0.5}}));
    EXPECT(b0.hot_deltas_to_X.size() == 1 && b0.hot_deltas_to_X.at(0) == (Site_delta{sT, sC}));
    EXPECT_NEAR(g.log_alpha_mut, -mu_JC * 0.5 + std::log(mu_JC / 3) + -mu_JC * 0.5 + -std::log(P_JC(sT, sC, 1.0)) + -mu_JC * 1.0 + -std::log(P_JC(sA, sA, 1.0)), 1e-9);
    EXPECT_NEAR(g.delta_log_G, -mu(0) * nu(0) * qa(0, sT) * 0.5 + std::log(mu(0) * nu(0) * qab(0, sT, sC)) + -mu(0) * nu(0) * qa(0, sC) * 0.5 + -mu(1) * nu(1) * qa(1, sA) * 1.0, 1e-9); }
  { auto g = spr.analyze_graft(b_);   // :201-258
    auto& b0 = g.branch_infos[0];
    EXPECT(g.branch_infos.size() == 1 && b0.A == x_ && b0.B == b_ && b0.T_to_X == 2.0);
    EXPECT_NEAR(b0.partial_lambda_at_X, mu(0) * nu(0) * qa(0, sT) + mu(1) * nu(1) * qa(1, sG), 1e-9);
    EXPECT(b0.hot_muts_to_X == (Mutation_list{Mutation{sA, 1, sG, 1.0}}));
    EXPECT_NEAR(g.log_alpha_mut, -mu_JC * 2.0 + -std::log(P_JC(sT, sT, 2.0)) + -mu_JC * 1.0 + std::log(mu_JC / 3) + -mu_JC * 1.0 + -std::log(P_JC(sA, sG, 2.0)), 1e-9);
    EXPECT_NEAR(g.delta_log_G, -mu(0) * nu(0) * qa(0, sT) * 2.0 + -mu(1) * nu(1) * qa(1, sA) * 1.0 + std::log(mu(1) * nu(1) * qab(1, sA, sG)) + -mu(1) * nu(1) * qa(1, sG) * 1.0, 1e-9); }
  { auto g = spr.analyze_graft(c_);   // :260-364 (rooty)
    EXPECT(g.branch_infos.size() == 3);
    auto& PX = g.branch_infos[0]; auto& PS = g.branch_infos[1]; auto& SPX = g.branch_infos[2];
    EXPECT(PX.A == r_ && PX.B == c_ && PX.is_open && PX.T_to_X == 4.0);
    EXPECT_NEAR(PX.partial_lambda_at_A, mu(2) * nu(2) * qa(2, sA), 1e-9); EXPECT_NEAR(PX.partial_lambda_at_X, mu(2) * nu(2) * qa(2, sA), 1e-9);
    EXPECT(PX.warm_sites.v == (std::vector<Site_interval>{{2, 3}}) && PX.hot_sites.v == PX.warm_sites.v && PX.hot_muts_to_X.empty() && PX.hot_deltas_to_X.empty());
    EXPECT(PS.A == r_ && PS.B == x_ && PS.is_open && PS.T_to_X == 1.0);
    EXPECT_NEAR(PS.partial_lambda_at_A, mu(1) * nu(1) * qa(1, sA), 1e-9);
    EXPECT(PS.warm_sites.v == (std::vector<Site_interval>{{1, 2}}) && PS.hot_muts_to_X.empty());
    EXPECT(SPX.A == x_ && SPX.B == r_ && !SPX.is_open && SPX.T_to_X == 5.0);
    EXPECT_NEAR(SPX.partial_lambda_at_A, mu(0) * nu(0) * qa(0, sT), 1e-9); EXPECT_NEAR(SPX.partial_lambda_at_X, mu(0) * nu(0) * qa(0, sG), 1e-9);
    EXPECT(SPX.hot_sites.v == (std::vector<Site_interval>{{0, 1}}));
    EXPECT(SPX.hot_muts_to_X == (Mutation_list{Mutation{sT, 0, sA, -1.5}, Mutation{sA, 0, sT, 0.0}, Mutation{sT, 0, sG, 1.0}}));
    EXPECT(SPX.hot_deltas_to_X.size() == 1 && SPX.hot_deltas_to_X.at(0) == (Site_delta{sT, sG}));
    EXPECT_NEAR(g.log_alpha_mut, -mu_JC * 0.5 + std::log(mu_JC / 3) + -mu_JC * 0.5 + -mu_JC * 1.0 + std::log(mu_JC / 3) + -mu_JC * 1.0 + std::log(mu_JC / 3) + -mu_JC * 2.0 + -std::log(P_JC(sT, sG, 5.0))
                + -mu_JC * 1.0 + -mu_JC * 4.0, 1e-9);
    EXPECT_NEAR(g.delta_log_G, -mu(0) * nu(0) * qa(0, sA) * 0.5 + std::log(mu(0) * nu(0) * qab(0, sA, sT)) + -mu(0) * nu(0) * qa(0, sT) * 0.5
                + -mu(0) * nu(0) * qa(0, sA) * 1.0 + std::log(mu(0) * nu(0) * qab(0, sA, sT)) + -mu(0) * nu(0) * qa(0, sT) * 1.0 + std::log(mu(0) * nu(0) * qab(0, sT, sG)) + -mu(0) * nu(0) * qa(0, sG) * 2.0
                + std::log(evo.pi_l_a(0, sA) / evo.pi_l_a(0, sT)) + -mu(1) * nu(1) * qa(1, sA) * 1.0 + -mu(2) * nu(2) * qa(2, sA) * 4.0, 1e-9); }
  { auto g = spr.analyze_graft(x_);   // :366-470 (rooty, the inner node x pruned from under the root)
    EXPECT(g.branch_infos.size() == 3);
    auto& PX = g.branch_infos[Spr_graft::k_branch_info_P_X]; auto& PS = g.branch_infos[Spr_graft::k_branch_info_P_S]; auto& SPX = g.branch_infos[Spr_graft::k_branch_info_S_P_X];
    EXPECT(PX.A == r_ && PX.B == x_ && PX.is_open && PX.T_to_X == 1.0);   // site 1 is only present in P->X
    EXPECT_NEAR(PX.partial_lambda_at_A, mu(1) * nu(1) * qa(1, sA), 1e-9); EXPECT_NEAR(PX.partial_lambda_at_X, mu(1) * nu(1) * qa(1, sA), 1e-9);
    EXPECT(PX.warm_sites.v == (std::vector<Site_interval>{{1, 2}}) && PX.hot_sites.v == PX.warm_sites.v && PX.hot_muts_to_X.empty() && PX.hot_deltas_to_X.empty());
    EXPECT(PS.A == r_ && PS.B == c_ && PS.is_open && PS.T_to_X == 4.0);   // site 2 is only present in P->S
    EXPECT_NEAR(PS.partial_lambda_at_A, mu(2) * nu(2) * qa(2, sA), 1e-9); EXPECT_NEAR(PS.partial_lambda_at_X, mu(2) * nu(2) * qa(2, sA), 1e-9);
    EXPECT(PS.warm_sites.v == (std::vector<Site_interval>{{2, 3}}) && PS.hot_sites.v == PS.warm_sites.v && PS.hot_muts_to_X.empty() && PS.hot_deltas_to_X.empty());
    EXPECT(SPX.A == c_ && SPX.B == r_ && !SPX.is_open && SPX.T_to_X == 5.0);   // site 0 all the way along S->P->X
    EXPECT_NEAR(SPX.partial_lambda_at_A, mu(0) * nu(0) * qa(0, sG), 1e-9); EXPECT_NEAR(SPX.partial_lambda_at_X, mu(0) * nu(0) * qa(0, sT), 1e-9);
    EXPECT(SPX.warm_sites.v == (std::vector<Site_interval>{{0, 1}}) && SPX.hot_sites.v == SPX.warm_sites.v);
    EXPECT(SPX.hot_muts_to_X == (Mutation_list{Mutation{sG, 0, sT, -3.0}, Mutation{sT, 0, sA, -2.0}, Mutation{sA, 0, sT, -0.5}}));
    EXPECT(SPX.hot_deltas_to_X.size() == 1 && SPX.hot_deltas_to_X.at(0) == (Site_delta{sG, sT}));
    EXPECT_NEAR(g.log_alpha_mut, -mu_JC * 2.0 + std::log(mu_JC / 3) + -mu_JC * 1.0 + std::log(mu_JC / 3) + -mu_JC * 1.0 + -mu_JC * 0.5 + std::log(mu_JC / 3) + -mu_JC * 0.5 + -std::log(P_JC(sG, sT, 5.0))
                + -mu_JC * 1.0 + -mu_JC * 4.0, 1e-9);
    EXPECT_NEAR(g.delta_log_G, -mu(0) * nu(0) * qa(0, sA) * 1.0 + std::log(mu(0) * nu(0) * qab(0, sA, sT)) + -mu(0) * nu(0) * qa(0, sT) * 1.0 + std::log(mu(0) * nu(0) * qab(0, sT, sG)) + -mu(0) * nu(0) * qa(0, sG) * 2.0
                + -mu(0) * nu(0) * qa(0, sA) * 0.5 + std::log(mu(0) * nu(0) * qab(0, sA, sT)) + -mu(0) * nu(0) * qa(0, sT) * 0.5
                + std::log(evo.pi_l_a(0, sA) / evo.pi_l_a(0, sG)) + -mu(1) * nu(1) * qa(1, sA) * 1.0 + -mu(2) * nu(2) * qa(2, sA) * 4.0, 1e-9); }
}
enum { tP = 0, tX = 1, tS = 2 };
static SprCtx tricky_rooty() {   // spr_move_tests.cpp:1205-1259
  Phylo_tree t(3);
  t.ref_sequence = {sA, sA, sA, sC, sC, sA, sC, sT, sT};
  t.root = tP;
  set_inner(t, tP, k_no_node, tX, tS, 0.0);
  miss(t, tP, {{7, sT}}); t.at(tP).mutations = {Mutation{sT, 8, sA, NEG}};
  set_tip(t, tX, tP, 3.0); miss(t, tX, {{3, sC}, {4, sC}}); t.at(tX).mutations = {Mutation{sA, 1, sC, 1.0}, Mutation{sA, 5, sC, 2.0}};
  set_tip(t, tS, tP, 4.0); miss(t, tS, {{1, sA}, {2, sA}}); t.at(tS).mutations = {Mutation{sC, 3, sG, 1.0}, Mutation{sC, 6, sG, 2.0}};
  auto evo = fixture_evo({0, 1, 0, 1, 0, 1, 0, 1, 0}, {0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9});
  return SprCtx(std::move(t), std::move(evo));
}
static std::vector<std::pair<int, State>> miss_elems(const Phylo_tree& t, int n) {
  std::vector<std::pair<int, State>> out;
  for (auto& [s, e] : t.at(n).missations.intervals.v) for (int l = s; l < e; ++l) out.push_back({l, t.at(n).missations.get_from_state(l, t.ref_sequence)});
  return out;
}
static bool same_muts_unordered(Mutation_list a, Mutation_list b) {
  auto key = [](const Mutation& m) { return std::make_tuple(m.site, m.from, m.to, m.t); };
  std::sort(a.begin(), a.end(), [&](auto& x, auto& y) { return key(x) < key(y); }); std::sort(b.begin(), b.end(), [&](auto& x, auto& y) { return key(x) < key(y); });
  return a == b;
}
TEST(spr_move_tricky_rooty_analyze_graft_X) {   // spr_move_tests.cpp:1261-1433
  auto C = tricky_rooty(); auto& evo = C.evo;
  auto mu = [&](int l) { return evo.mu_l(l); }; auto nu = [&](int l) { return evo.nu_l[l]; };
  auto qa = [&](int l, State a) { return evo.q_l_a(l, a); }; auto qab = [&](int l, State a, State b) { return evo.q_l_ab(l, a, b); };
  auto pi = [&](int l, State a) { return evo.pi_l_a(l, a); };
  auto spr = C.spr(true);
  auto g = spr.analyze_graft(tX);
  EXPECT(g.branch_infos.size() == 3);
  auto& PX = g.branch_infos[Spr_graft::k_branch_info_P_X]; auto& PS = g.branch_infos[Spr_graft::k_branch_info_P_S]; auto& SPX = g.branch_infos[Spr_graft::k_branch_info_S_P_X];
  EXPECT(PX.A == tP && PX.B == tX && PX.is_open && PX.T_to_X == 3.0);   // P->X: sites 1, 2
  EXPECT_NEAR(PX.partial_lambda_at_A, mu(1) * nu(1) * qa(1, sA) + mu(2) * nu(2) * qa(2, sA), 1e-9);
  EXPECT_NEAR(PX.partial_lambda_at_X, mu(1) * nu(1) * qa(1, sC) + mu(2) * nu(2) * qa(2, sA), 1e-9);
  EXPECT(PX.warm_sites.v == (std::vector<Site_interval>{{1, 3}}) && PX.hot_sites.v == PX.warm_sites.v);
  EXPECT(PX.hot_muts_to_X == (Mutation_list{Mutation{sA, 1, sC, 1.0}}) && PX.hot_deltas_to_X.empty());
  EXPECT(PS.A == tP && PS.B == tS && PS.is_open && PS.T_to_X == 4.0);   // P->S: sites 3, 4
  EXPECT_NEAR(PS.partial_lambda_at_A, mu(3) * nu(3) * qa(3, sC) + mu(4) * nu(4) * qa(4, sC), 1e-9);
  EXPECT_NEAR(PS.partial_lambda_at_X, mu(3) * nu(3) * qa(3, sG) + mu(4) * nu(4) * qa(4, sC), 1e-9);
  EXPECT(PS.warm_sites.v == (std::vector<Site_interval>{{3, 5}}) && PS.hot_sites.v == PS.warm_sites.v);
  EXPECT(PS.hot_muts_to_X == (Mutation_list{Mutation{sC, 3, sG, 1.0}}) && PS.hot_deltas_to_X.empty());
  EXPECT(SPX.A == tS && SPX.B == tP && !SPX.is_open && SPX.T_to_X == 7.0);   // S->P->X: sites 0, 5, 6, 8
  EXPECT_NEAR(SPX.partial_lambda_at_A, mu(0) * nu(0) * qa(0, sA) + mu(5) * nu(5) * qa(5, sA) + mu(6) * nu(6) * qa(6, sG) + mu(8) * nu(8) * qa(8, sA), 1e-9);
  EXPECT_NEAR(SPX.partial_lambda_at_X, mu(0) * nu(0) * qa(0, sA) + mu(5) * nu(5) * qa(5, sC) + mu(6) * nu(6) * qa(6, sC) + mu(8) * nu(8) * qa(8, sA), 1e-9);
  EXPECT(SPX.warm_sites.v == (std::vector<Site_interval>{{0, 1}, {5, 7}, {8, 9}}) && SPX.hot_sites.v == SPX.warm_sites.v);
  EXPECT(SPX.hot_muts_to_X == (Mutation_list{Mutation{sG, 6, sC, -2.0}, Mutation{sA, 5, sC, 2.0}}));
  EXPECT(SPX.hot_deltas_to_X.size() == 2 && SPX.hot_deltas_to_X.at(5) == (Site_delta{sA, sC}) && SPX.hot_deltas_to_X.at(6) == (Site_delta{sG, sC}));
  EXPECT_NEAR(g.log_alpha_mut,
              -mu_JC * 7.0 + -std::log(P_JC(sA, sA, 7.0))                                                   // site 0 (S->P->X)
              + -mu_JC * 1.0 + std::log(mu_JC / 3) + -mu_JC * 2.0                                            // site 1 (P->X)
              + -mu_JC * 3.0                                                                                 // site 2 (P->X)
              + -mu_JC * 1.0 + std::log(mu_JC / 3) + -mu_JC * 3.0                                            // site 3 (P->S)
              + -mu_JC * 4.0                                                                                 // site 4 (P->S)
              + -mu_JC * 4.0 + -mu_JC * 2.0 + std::log(mu_JC / 3) + -mu_JC * 1.0 + -std::log(P_JC(sA, sC, 7.0))   // site 5 (S->P->X)
              + -mu_JC * 2.0 + std::log(mu_JC / 3) + -mu_JC * 2.0 + -mu_JC * 3.0 + -std::log(P_JC(sG, sC, 7.0))   // site 6 (S->P->X)
              + -mu_JC * 4.0 + -mu_JC * 3.0 + -std::log(P_JC(sA, sA, 7.0)), 1e-9);                           // site 7 missing; site 8 (S->P->X)
  EXPECT_NEAR(g.delta_log_G,
              -mu(0) * nu(0) * qa(0, sA) * 4.0 + -mu(0) * nu(0) * qa(0, sA) * 3.0
              + -mu(1) * nu(1) * qa(1, sA) * 1.0 + std::log(mu(1) * nu(1) * qab(1, sA, sC)) + -mu(1) * nu(1) * qa(1, sC) * 2.0 + std::log(pi(1, sA) / pi(1, sC))
              + -mu(2) * nu(2) * qa(2, sA) * 3.0
              + -mu(3) * nu(3) * qa(3, sC) * 1.0 + std::log(mu(3) * nu(3) * qab(3, sC, sG)) + -mu(3) * nu(3) * qa(3, sG) * 3.0 + std::log(pi(3, sC) / pi(3, sG))
              + -mu(4) * nu(4) * qa(4, sC) * 4.0
              + -mu(5) * nu(5) * qa(5, sA) * 4.0 + -mu(5) * nu(5) * qa(5, sA) * 2.0 + std::log(mu(5) * nu(5) * qab(5, sA, sC)) + -mu(5) * nu(5) * qa(5, sC) * 1.0
              + -mu(6) * nu(6) * qa(6, sC) * 2.0 + std::log(mu(6) * nu(6) * qab(6, sC, sG)) + -mu(6) * nu(6) * qa(6, sG) * 2.0 + -mu(6) * nu(6) * qa(6, sC) * 3.0 + std::log(pi(6, sC) / pi(6, sG))
              + -mu(8) * nu(8) * qa(8, sA) * 4.0 + -mu(8) * nu(8) * qa(8, sA) * 3.0, 1e-9);
}
TEST(spr_move_tricky_rooty_peel_and_reapply) {
  { auto C = tricky_rooty(); auto spr = C.spr();
    EXPECT(check_phylo_tree_integrity(C.tree).empty());
    auto g = spr.analyze_graft(tX); spr.peel_graft(g);   // :1435-1466
    EXPECT(same_muts_unordered(C.tree.at(tP).mutations, {Mutation{sA, 1, sC, NEG}, Mutation{sC, 3, sG, NEG}, Mutation{sC, 6, sG, NEG}, Mutation{sT, 8, sA, NEG}}));
    EXPECT(miss_elems(C.tree, tP) == (std::vector<std::pair<int, State>>{{7, sT}}));
    EXPECT(same_muts_unordered(C.tree.at(tX).mutations, {Mutation{sA, 5, sC, 1.5}, Mutation{sG, 6, sC, 1.5}}));
    EXPECT(miss_elems(C.tree, tX) == (std::vector<std::pair<int, State>>{{3, sG}, {4, sC}}));
    EXPECT(C.tree.at(tS).mutations.empty());
    EXPECT(miss_elems(C.tree, tS) == (std::vector<std::pair<int, State>>{{1, sC}, {2, sA}}));
    EXPECT(spr.count_closed_mutations(g) == 2);   // :1468-1483
    auto cd = spr.summarize_closed_mutations(g);
    EXPECT(cd.size() == 2 && cd.at(5) == (Site_delta{sA, sC}) && cd.at(6) == (Site_delta{sG, sC}));
    EXPECT(check_phylo_tree_integrity(C.tree).empty()); }
  { auto C = tricky_rooty(); auto spr = C.spr();
    auto g = spr.analyze_graft(tX); spr.peel_graft(g); spr.apply_graft(g);   // :1485-1516
    EXPECT(same_muts_unordered(C.tree.at(tP).mutations, {Mutation{sT, 8, sA, NEG}}));
    EXPECT(same_muts_unordered(C.tree.at(tX).mutations, {Mutation{sA, 1, sC, 1.0}, Mutation{sA, 5, sC, 2.0}}));
    EXPECT(miss_elems(C.tree, tX) == (std::vector<std::pair<int, State>>{{3, sC}, {4, sC}}));
    EXPECT(same_muts_unordered(C.tree.at(tS).mutations, {Mutation{sC, 3, sG, 1.0}, Mutation{sC, 6, sG, 2.0}}));
    EXPECT(miss_elems(C.tree, tS) == (std::vector<std::pair<int, State>>{{1, sA}, {2, sA}})); }
}

// ---- property tests (spr_move_tests.cpp:472-562 and :1518-1641), on reference fixtures and synthetic trees
static double total_log_G(const Phylo_tree& t, const Global_evo_model& evo) { return calc_log_root_prior(t, evo) + calc_log_G_below_root(t, evo); }
static void run_propose_new_graft_test(SprCtx& C, int num_seeds, bool can_change_root) {
  Phylo_tree old_tree = C.tree;
  for (int seed = 0; seed < num_seeds; ++seed) {
    Rng rng; rng.key = seed + 12345;
    C.tree = old_tree; C.refresh();
    auto spr = C.spr(can_change_root);
    int X = C.tree.root; while (X == C.tree.root) X = rng.uniform_int(C.tree.size());
    if (!can_change_root && C.tree.at(X).parent == C.tree.root) continue;
    auto old_graft = spr.analyze_graft(X);
    double old_log_G = total_log_G(C.tree, C.evo);
    spr.peel_graft(old_graft);
    EXPECT(check_phylo_tree_integrity(C.tree).empty());
    auto new_graft = spr.propose_new_graft(X, rng);
    spr.apply_graft(new_graft);
    auto msg = check_phylo_tree_integrity(C.tree);
    if (!msg.empty()) std::printf("  seed %d X=%d: %s\n", seed, X, msg.c_str());
    EXPECT(msg.empty());
    auto redux = spr.analyze_graft(X);
    double new_log_G = total_log_G(C.tree, C.evo);
    EXPECT(redux.branch_infos.size() == new_graft.branch_infos.size());
    for (size_t i = 0; i < std::min(redux.branch_infos.size(), new_graft.branch_infos.size()); ++i) {
      auto& bi = new_graft.branch_infos[i]; auto& br = redux.branch_infos[i];
      EXPECT(bi.A == br.A && bi.B == br.B && bi.is_open == br.is_open);
      EXPECT_NEAR(br.partial_lambda_at_A, bi.partial_lambda_at_A, 1e-6); EXPECT_NEAR(br.partial_lambda_at_X, bi.partial_lambda_at_X, 1e-6);
      EXPECT(br.hot_muts_to_X.size() == bi.hot_muts_to_X.size());
      for (size_t j = 0; j < std::min(br.hot_muts_to_X.size(), bi.hot_muts_to_X.size()); ++j) {
        auto& m = bi.hot_muts_to_X[j]; auto& mr = br.hot_muts_to_X[j];
        EXPECT(m.from == mr.from && m.to == mr.to && m.site == mr.site); EXPECT_NEAR(m.t, mr.t, 1e-6);
      }
      EXPECT(br.hot_deltas_to_X == bi.hot_deltas_to_X);
    }
    EXPECT_NEAR(redux.delta_log_G, new_graft.delta_log_G, 1e-6);
    EXPECT_NEAR(redux.log_alpha_mut, new_graft.log_alpha_mut, 1e-6);
    EXPECT_NEAR(new_log_G, old_log_G - old_graft.delta_log_G + new_graft.delta_log_G, 1e-6);
    for (int n = 0; n < C.tree.size(); ++n) if (C.tree.at(n).is_tip()) {
      auto sn = view_of_sequence_at(C.tree, n), so = view_of_sequence_at(old_tree, n);
      auto mn = reconstruct_missing_sites_at(C.tree, n), mo = reconstruct_missing_sites_at(old_tree, n);
      EXPECT(mn == mo);
      for (int l = 0; l < C.tree.num_sites(); ++l) if (!mn.contains(l)) EXPECT(sn[l] == so[l]);
    }
    auto li = calc_lambda_i(C.tree, C.evo, C.cumQ);
    for (int n = 0; n < C.tree.size(); ++n) EXPECT_NEAR(C.lambda_i[n], li[n], 1e-6);
    EXPECT(C.nm == calc_num_sites_missing_at_every_node(C.tree));
  }
  C.tree = old_tree; C.refresh();
}
static void run_full_spr_move_test(SprCtx& C, int X, int SS, double new_t_P, Rng& rng, bool can_change_root = true) {
  C.refresh();
  auto spr = C.spr(can_change_root);
  Phylo_tree old_tree = C.tree;
  int P = C.tree.at(X).parent; double old_t_P = C.tree.at(P).t; int S = C.tree.at(P).sibling_of(X);
  int G = C.tree.at(P).parent; int GG = C.tree.at(SS).parent; if (GG == P) GG = G;
  auto lam_ok = [&](const char* where) { auto li = calc_lambda_i(C.tree, C.evo, C.cumQ); for (int n = 0; n < C.tree.size(); ++n) { ++g_checks; if (!(std::fabs(li[n] - C.lambda_i[n]) <= 1e-6)) { ++g_fail; std::printf("FAIL [%s] lambda_i[%d] %s: %.12g vs %.12g (X=%d SS=%d t=%g)\n", g_test, n, where, C.lambda_i[n], li[n], X, SS, new_t_P); } } };
  auto integ = [&](const char* where) { auto m = check_phylo_tree_integrity(C.tree); ++g_checks; if (!m.empty()) { ++g_fail; std::printf("FAIL [%s] integrity %s: %s (X=%d SS=%d t=%g)\n", g_test, where, m.c_str(), X, SS, new_t_P); } };
  auto old_graft = spr.analyze_graft(X);
  double old_log_G = total_log_G(C.tree, C.evo);
  spr.peel_graft(old_graft); integ("after peel"); lam_ok("after peel");
  spr.move(X, SS, new_t_P); integ("after move"); lam_ok("after move");
  auto new_graft = spr.propose_new_graft(X, rng);
  spr.apply_graft(new_graft); integ("after apply");
  double new_log_G = total_log_G(C.tree, C.evo);
  EXPECT_NEAR(new_log_G, old_log_G - old_graft.delta_log_G + new_graft.delta_log_G, 1e-6);
  EXPECT(C.tree.ref_sequence == old_tree.ref_sequence);
  lam_ok("after apply");
  EXPECT(C.nm == calc_num_sites_missing_at_every_node(C.tree));
  for (int n = 0; n < C.tree.size(); ++n) { if (n == P || n == X || n == S || n == SS) continue; EXPECT(C.tree.at(n).parent == old_tree.at(n).parent); }
  for (int n = 0; n < C.tree.size(); ++n) {
    if (n == G || n == GG || n == P) continue;
    std::set<int> a{C.tree.at(n).children[0], C.tree.at(n).children[1]}, b{old_tree.at(n).children[0], old_tree.at(n).children[1]};
    EXPECT(a == b);
  }
  for (int n = 0; n < C.tree.size(); ++n) {
    EXPECT(C.tree.at(n).is_tip() == old_tree.at(n).is_tip());
    bool up_S = n != S && descends_from(C.tree, S, n), up_SS = n != SS && descends_from(C.tree, SS, n);
    if (C.tree.at(n).is_tip() || !(up_S || up_SS)) {
      auto so = view_of_sequence_at(old_tree, n), sn = view_of_sequence_at(C.tree, n);
      auto mo = reconstruct_missing_sites_at(old_tree, n), mn = reconstruct_missing_sites_at(C.tree, n);
      EXPECT(mo == mn);
      for (int l = 0; l < C.tree.num_sites(); ++l) if (!mo.contains(l) && !mn.contains(l)) EXPECT(so[l] == sn[l]);
    }
  }
  spr.peel_graft(new_graft); integ("after re-peel"); lam_ok("after re-peel");
  spr.move(X, S, old_t_P); integ("after move back"); lam_ok("after move back");
  spr.apply_graft(old_graft); integ("after re-apply"); lam_ok("after re-apply");
  for (int n = 0; n < C.tree.size(); ++n) {
    EXPECT(view_of_sequence_at(C.tree, n) == view_of_sequence_at(old_tree, n));
    EXPECT(reconstruct_missing_sites_at(C.tree, n) == reconstruct_missing_sites_at(old_tree, n));
    EXPECT(C.tree.at(n).parent == old_tree.at(n).parent);
    std::set<int> a{C.tree.at(n).children[0], C.tree.at(n).children[1]}, b{old_tree.at(n).children[0], old_tree.at(n).children[1]};
    EXPECT(a == b);
  }
  EXPECT_NEAR(total_log_G(C.tree, C.evo), old_log_G, 1e-6);
  EXPECT(C.nm == calc_num_sites_missing_at_every_node(C.tree));
}
struct Case { int X, SS; double t; };
TEST(spr_move_simple_fixture_properties) {
  SprCtx C(complex_tree(true), fixture_evo());
  run_propose_new_graft_test(C, 1000, true);
  for (auto& c : std::vector<Case>{{a_, b_, 0.5}, {a_, x_, -0.5}, {a_, c_, 0.5}, {a_, r_, -1.5}, {b_, a_, 0.5}, {b_, x_, -0.5}, {b_, c_, 0.5}, {b_, r_, -1.5},
                                   {c_, x_, -0.5}, {c_, r_, -1.5}, {c_, a_, 0.5}, {c_, b_, 1.0}, {x_, c_, -0.4}, {x_, r_, -1.6}})   // :1643-1671
    for (int seed = 0; seed < 200; ++seed) { Rng rng; rng.key = seed + 12345; run_full_spr_move_test(C, c.X, c.SS, c.t, rng); }
}
static SprCtx precarious_without_root() {   // spr_move_tests.cpp:848-929
  enum { r = 0, x = 1, y = 2, a = 3, b = 4, c = 5, d = 6 };
  Phylo_tree t(7); t.root = r; t.ref_sequence = {sA, sC, sA, sA};
  set_inner(t, r, k_no_node, x, d, -1.0); miss(t, r, {{2, sA}, {3, sA}}); t.at(r).mutations = {Mutation{sC, 1, sA, NEG}};
  set_inner(t, x, r, y, c, 0.0); t.at(x).mutations = {Mutation{sA, 1, sC, -0.5}};
  set_inner(t, y, x, a, b, 1.0);
  set_tip(t, a, y, 3.0); t.at(a).mutations = {Mutation{sC, 1, sT, 2.0}};
  set_tip(t, b, y, 3.0); miss(t, b, {{1, sC}});
  set_tip(t, c, x, 3.0); miss(t, c, {{1, sC}});
  set_tip(t, d, r, 3.0);
  return SprCtx(std::move(t), fixture_evo());
}
TEST(spr_move_precarious_without_root_properties) {
  enum { r = 0, x = 1, y = 2, a = 3, b = 4, c = 5, d = 6 };
  auto C = precarious_without_root();
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  run_propose_new_graft_test(C, 1000, true);
  for (auto& k : std::vector<Case>{{a, b, 1.4}, {a, y, 0.6}, {a, c, 1.5}, {a, d, 1.0}, {a, r, -1.5}, {b, a, 1.4}, {b, y, 0.6}, {b, c, 1.5}, {b, d, 1.0}, {b, r, -1.5},
                                   {c, y, 0.6}, {c, x, -0.4}, {c, a, 2.0}, {c, b, 2.0}, {c, d, 1.0}, {c, r, -1.5}, {d, x, -0.4}, {d, r, -1.6}, {d, a, 2.0}, {d, b, 2.0}, {d, c, 1.5},
                                   {y, c, 0.4}, {y, x, -0.6}, {y, d, 0.3}, {y, r, -1.2}, {x, d, -0.4}, {x, r, -1.6}})   // :1722-1763
    for (int seed = 0; seed < 200; ++seed) { Rng rng; rng.key = seed + 12345; run_full_spr_move_test(C, k.X, k.SS, k.t, rng); }
}
static SprCtx precarious_with_root() {   // spr_move_tests.cpp:1041-1105
  Phylo_tree t(5); t.root = r_; t.ref_sequence = {sA, sC, sA, sA};
  set_inner(t, r_, k_no_node, x_, c_, -1.0); miss(t, r_, {{2, sA}, {3, sA}}); t.at(r_).mutations = {Mutation{sC, 1, sA, NEG}};
  set_inner(t, x_, r_, a_, b_, 0.0); t.at(x_).mutations = {Mutation{sA, 1, sC, -0.5}};
  set_tip(t, a_, x_, 1.0);
  set_tip(t, b_, x_, 1.0); miss(t, b_, {{1, sC}});
  set_tip(t, c_, r_, 1.0); miss(t, c_, {{1, sA}});
  return SprCtx(std::move(t), fixture_evo());
}
TEST(spr_move_precarious_with_root_properties) {
  auto C = precarious_with_root();
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  run_propose_new_graft_test(C, 1000, true);
  for (auto& k : std::vector<Case>{{a_, b_, 0.6}, {a_, x_, -0.4}, {a_, c_, 0.0}, {a_, r_, -1.5}, {b_, a_, 0.6}, {b_, x_, -0.4}, {b_, c_, 0.0}, {b_, r_, -1.5},
                                   {c_, x_, -0.6}, {c_, r_, -1.4}, {c_, a_, 0.5}, {c_, b_, 0.5}, {x_, c_, -0.6}, {x_, r_, -1.2}})   // :1765-1793
    for (int seed = 0; seed < 200; ++seed) { Rng rng; rng.key = seed + 12345; run_full_spr_move_test(C, k.X, k.SS, k.t, rng); }
}
// ---- closed-form analyze_graft_a of the two precarious-path fixtures (spr_move_tests.cpp:931-1035, 1107-1199)
TEST(spr_move_precarious_without_root_analyze_graft_a) {
  enum { r = 0, x = 1, y = 2, a = 3, b = 4, c = 5, d = 6 };
  auto C = precarious_without_root(); auto& evo = C.evo;
  auto mu = [&](int l) { return evo.mu_l(l); }; auto nu = [&](int l) { return evo.nu_l[l]; };
  auto qa = [&](int l, State s) { return evo.q_l_a(l, s); }; auto qab = [&](int l, State s, State t) { return evo.q_l_ab(l, s, t); };
  auto spr = C.spr(false);   // can_change_root = false (:932)
  auto g = spr.analyze_graft(a);
  EXPECT(g.branch_infos.size() == 3);
  auto& b0 = g.branch_infos[0]; auto& b1 = g.branch_infos[1]; auto& b2 = g.branch_infos[2];
  EXPECT(b0.A == y && b0.B == a && !b0.is_open && b0.T_to_X == 2.0);   // y->a: sites 0 and 1 warm, site 0 hot
  EXPECT_NEAR(b0.partial_lambda_at_A, mu(0) * nu(0) * qa(0, sA), 1e-9); EXPECT_NEAR(b0.partial_lambda_at_X, mu(0) * nu(0) * qa(0, sA), 1e-9);
  EXPECT(b0.warm_sites.contains(0) && b0.warm_sites.contains(1) && b0.hot_sites.contains(0) && !b0.hot_sites.contains(1));
  EXPECT(b0.hot_muts_to_X.empty() && b0.hot_deltas_to_X.empty());
  EXPECT(b1.A == x && b1.B == y && !b1.is_open && b1.T_to_X == 3.0);   // x->y: site 1 warm, not hot
  EXPECT_NEAR(b1.partial_lambda_at_A, 0.0, 1e-9); EXPECT_NEAR(b1.partial_lambda_at_X, 0.0, 1e-9);
  EXPECT(b1.warm_sites.v == (std::vector<Site_interval>{{1, 2}}) && b1.hot_sites.empty() && b1.hot_muts_to_X.empty() && b1.hot_deltas_to_X.empty());
  EXPECT(b2.A == r && b2.B == x && !b2.is_open && b2.T_to_X == 4.0);   // r->x: site 1 warm and hot
  EXPECT_NEAR(b2.partial_lambda_at_A, mu(1) * nu(1) * qa(1, sA), 1e-9); EXPECT_NEAR(b2.partial_lambda_at_X, mu(1) * nu(1) * qa(1, sT), 1e-9);
  EXPECT(b2.warm_sites.v == (std::vector<Site_interval>{{1, 2}}) && b2.hot_sites.v == b2.warm_sites.v);
  EXPECT(b2.hot_muts_to_X == (Mutation_list{Mutation{sA, 1, sC, -0.5}, Mutation{sC, 1, sT, 2.0}}));
  EXPECT(b2.hot_deltas_to_X.size() == 1 && b2.hot_deltas_to_X.at(1) == (Site_delta{sA, sT}));
  EXPECT_NEAR(g.log_alpha_mut, -mu_JC * 2.0 + -std::log(P_JC(sA, sA, 2.0))
              + -mu_JC * 0.5 + std::log(mu_JC / 3) + -mu_JC * 0.5 + -mu_JC * 1.0 + -mu_JC * 1.0 + std::log(mu_JC / 3) + -mu_JC * 1.0 + -std::log(P_JC(sA, sT, 4.0)), 1e-9);
  EXPECT_NEAR(g.delta_log_G, -mu(0) * nu(0) * qa(0, sA) * 2.0
              + -mu(1) * nu(1) * qa(1, sA) * 0.5 + std::log(mu(1) * nu(1) * qab(1, sA, sC)) + -mu(1) * nu(1) * qa(1, sC) * 0.5 + -mu(1) * nu(1) * qa(1, sC) * 1.0
              + -mu(1) * nu(1) * qa(1, sC) * 1.0 + std::log(mu(1) * nu(1) * qab(1, sC, sT)) + -mu(1) * nu(1) * qa(1, sT) * 1.0, 1e-9);
}
TEST(spr_move_precarious_with_root_analyze_graft_a) {
  auto C = precarious_with_root(); auto& evo = C.evo;
  auto mu = [&](int l) { return evo.mu_l(l); }; auto nu = [&](int l) { return evo.nu_l[l]; };
  auto qa = [&](int l, State s) { return evo.q_l_a(l, s); }; auto qab = [&](int l, State s, State t) { return evo.q_l_ab(l, s, t); };
  auto spr = C.spr(true);
  auto g = spr.analyze_graft(a_);
  EXPECT(g.branch_infos.size() == 3);
  auto& b0 = g.branch_infos[0]; auto& b1 = g.branch_infos[1]; auto& b2 = g.branch_infos[2];
  EXPECT(b0.A == x_ && b0.B == a_ && !b0.is_open && b0.T_to_X == 1.0);   // x->a: sites 0 and 1 warm, site 0 hot
  EXPECT_NEAR(b0.partial_lambda_at_A, mu(0) * nu(0) * qa(0, sA), 1e-9); EXPECT_NEAR(b0.partial_lambda_at_X, mu(0) * nu(0) * qa(0, sA), 1e-9);
  EXPECT(b0.warm_sites.contains(0) && b0.warm_sites.contains(1) && b0.hot_sites.contains(0) && !b0.hot_sites.contains(1));
  EXPECT(b0.hot_muts_to_X.empty() && b0.hot_deltas_to_X.empty());
  EXPECT(b1.A == r_ && b1.B == x_ && !b1.is_open && b1.T_to_X == 2.0);   // r->x: site 1 warm, not hot
  EXPECT(b1.partial_lambda_at_A == 0.0 && b1.partial_lambda_at_X == 0.0);
  EXPECT(b1.warm_sites.v == (std::vector<Site_interval>{{1, 2}}) && b1.hot_sites.empty() && b1.hot_muts_to_X.empty() && b1.hot_deltas_to_X.empty());
  EXPECT(b2.A == k_no_node && b2.B == r_ && b2.is_open && b2.T_to_X == 2.0);   // *->r: site 1 warm and hot, an open path
  EXPECT_NEAR(b2.partial_lambda_at_A, mu(1) * nu(1) * qa(1, sA), 1e-9); EXPECT_NEAR(b2.partial_lambda_at_X, mu(1) * nu(1) * qa(1, sC), 1e-9);
  EXPECT(b2.warm_sites.v == (std::vector<Site_interval>{{1, 2}}) && b2.hot_sites.v == b2.warm_sites.v);
  EXPECT(b2.hot_muts_to_X == (Mutation_list{Mutation{sA, 1, sC, -0.5}}) && b2.hot_deltas_to_X.empty());   // open path: no hot deltas
  EXPECT_NEAR(g.log_alpha_mut, -mu_JC * 1.0 + -std::log(P_JC(sA, sA, 1.0)) + -mu_JC * 0.5 + std::log(mu_JC / 3) + -mu_JC * 0.5 + -mu_JC * 1.0, 1e-9);
  EXPECT_NEAR(g.delta_log_G, -mu(0) * nu(0) * qa(0, sA) * 1.0
              + -mu(1) * nu(1) * qa(1, sA) * 0.5 + std::log(mu(1) * nu(1) * qab(1, sA, sC)) + -mu(1) * nu(1) * qa(1, sC) * 0.5 + -mu(1) * nu(1) * qa(1, sC) * 1.0
              + std::log(evo.pi_l_a(1, sA) / evo.pi_l_a(1, sC)), 1e-9);
}
// ---- the two superfluous-mutation fixtures (spr_move_tests.cpp:568-615, 719-783): a mutation and its reversal on the path of the graft
static SprCtx superfluous_at_root() {   // :568-615: r (AANN) with tips x (T at site 0 from -0.5) and s (T at site 0 from 1.0)
  enum { r = 0, x = 1, s = 2 };
  Phylo_tree t(3); t.root = r; t.ref_sequence = {sA, sC, sA, sA};
  set_inner(t, r, k_no_node, x, s, -1.0); miss(t, r, {{2, sA}, {3, sA}}); t.at(r).mutations = {Mutation{sC, 1, sA, NEG}};
  set_tip(t, x, r, 0.0); t.at(x).mutations = {Mutation{sA, 0, sT, -0.5}};
  set_tip(t, s, r, 3.0); t.at(s).mutations = {Mutation{sA, 0, sT, 1.0}};
  return SprCtx(std::move(t), fixture_evo());
}
static SprCtx superfluous_not_at_root() {   // :719-783: x carries A0T, both its tips a and b carry T0A back; c is plain
  Phylo_tree t(5); t.root = r_; t.ref_sequence = {sA, sC, sA, sA};
  set_inner(t, r_, k_no_node, x_, c_, -1.0); miss(t, r_, {{2, sA}, {3, sA}}); t.at(r_).mutations = {Mutation{sC, 1, sA, NEG}};
  set_inner(t, x_, r_, a_, b_, 0.0); t.at(x_).mutations = {Mutation{sA, 0, sT, -0.5}};
  set_tip(t, a_, x_, 1.0); t.at(a_).mutations = {Mutation{sT, 0, sA, 0.5}};
  set_tip(t, b_, x_, 2.0); t.at(b_).mutations = {Mutation{sT, 0, sA, 1.0}};
  set_tip(t, c_, r_, 3.0);
  return SprCtx(std::move(t), fixture_evo());
}
TEST(spr_move_superfluous_mutation_at_root) {
  enum { r = 0, x = 1, s = 2 };
  auto C = superfluous_at_root(); auto& evo = C.evo;
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  auto mu = [&](int l) { return evo.mu_l(l); }; auto nu = [&](int l) { return evo.nu_l[l]; };
  auto qa = [&](int l, State a) { return evo.q_l_a(l, a); }; auto qab = [&](int l, State a, State b) { return evo.q_l_ab(l, a, b); };
  { auto spr = C.spr(true);
    auto g = spr.analyze_graft(x);   // :617-713
    EXPECT(g.branch_infos.size() == 3);
    auto& PX = g.branch_infos[Spr_graft::k_branch_info_P_X]; auto& PS = g.branch_infos[Spr_graft::k_branch_info_P_S]; auto& SPX = g.branch_infos[Spr_graft::k_branch_info_S_P_X];
    EXPECT(PX.A == r && PX.B == x && PX.is_open && PX.T_to_X == 1.0 && PX.partial_lambda_at_A == 0.0 && PX.partial_lambda_at_X == 0.0);   // no site is present only in P->X
    EXPECT(PX.warm_sites.empty() && PX.hot_sites.empty() && PX.hot_muts_to_X.empty() && PX.hot_deltas_to_X.empty());
    EXPECT(PS.A == r && PS.B == s && PS.is_open && PS.T_to_X == 4.0 && PS.partial_lambda_at_A == 0.0 && PS.partial_lambda_at_X == 0.0);   // nor only in P->S
    EXPECT(PS.warm_sites.empty() && PS.hot_sites.empty() && PS.hot_muts_to_X.empty() && PS.hot_deltas_to_X.empty());
    EXPECT(SPX.A == s && SPX.B == r && !SPX.is_open && SPX.T_to_X == 5.0);   // sites 0 and 1 all the way along S->P->X
    EXPECT_NEAR(SPX.partial_lambda_at_A, mu(0) * nu(0) * qa(0, sT) + mu(1) * nu(1) * qa(1, sA), 1e-9);
    EXPECT_NEAR(SPX.partial_lambda_at_X, mu(0) * nu(0) * qa(0, sT) + mu(1) * nu(1) * qa(1, sA), 1e-9);
    EXPECT(SPX.warm_sites.v == (std::vector<Site_interval>{{0, 2}}) && SPX.hot_sites.v == SPX.warm_sites.v);
    EXPECT(SPX.hot_muts_to_X == (Mutation_list{Mutation{sT, 0, sA, -3.0}, Mutation{sA, 0, sT, -0.5}}));   // T->A then A->T: superfluous, so ...
    EXPECT(SPX.hot_deltas_to_X.empty());                                                                       // ... no net delta
    EXPECT_NEAR(g.log_alpha_mut, -mu_JC * 2.0 + std::log(mu_JC / 3) + -mu_JC * 2.0 + -mu_JC * 0.5 + std::log(mu_JC / 3) + -mu_JC * 0.5 + -std::log(P_JC(sT, sT, 5.0))
                + -mu_JC * 5.0 + -std::log(P_JC(sA, sA, 5.0)), 1e-9);
    EXPECT_NEAR(g.delta_log_G, -mu(0) * nu(0) * qa(0, sA) * 2.0 + std::log(mu(0) * nu(0) * qab(0, sA, sT)) + -mu(0) * nu(0) * qa(0, sT) * 2.0
                + -mu(0) * nu(0) * qa(0, sA) * 0.5 + std::log(mu(0) * nu(0) * qab(0, sA, sT)) + -mu(0) * nu(0) * qa(0, sT) * 0.5
                + std::log(evo.pi_l_a(0, sA) / evo.pi_l_a(0, sT))
                + -mu(1) * nu(1) * qa(1, sA) * 4.0 + -mu(1) * nu(1) * qa(1, sA) * 1.0, 1e-9); }
  run_propose_new_graft_test(C, 1000, true);   // :715-717
  for (auto& k : std::vector<Case>{{x, s, -0.5}, {x, r, -1.5}, {s, x, -0.5}, {s, r, -1.5}})   // :1673-1690
    for (int seed = 0; seed < 200; ++seed) { Rng rng; rng.key = seed + 12345; run_full_spr_move_test(C, k.X, k.SS, k.t, rng); }
}
TEST(spr_move_superfluous_mutation_not_at_root) {
  auto C = superfluous_not_at_root(); auto& evo = C.evo;
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  auto mu = [&](int l) { return evo.mu_l(l); }; auto nu = [&](int l) { return evo.nu_l[l]; };
  auto qa = [&](int l, State a) { return evo.q_l_a(l, a); }; auto qab = [&](int l, State a, State b) { return evo.q_l_ab(l, a, b); };
  { auto spr = C.spr(true);
    auto g = spr.analyze_graft(a_);   // :785-842
    EXPECT(g.branch_infos.size() == 1);
    auto& b0 = g.branch_infos[0];
    EXPECT(b0.A == x_ && b0.B == a_ && !b0.is_open && b0.T_to_X == 1.0);
    EXPECT_NEAR(b0.partial_lambda_at_A, mu(0) * nu(0) * qa(0, sT) + mu(1) * nu(1) * qa(1, sA), 1e-9);
    EXPECT_NEAR(b0.partial_lambda_at_X, mu(0) * nu(0) * qa(0, sA) + mu(1) * nu(1) * qa(1, sA), 1e-9);
    EXPECT(b0.warm_sites.contains(0) && b0.warm_sites.contains(1) && b0.hot_sites.contains(0) && b0.hot_sites.contains(1));
    EXPECT(b0.hot_muts_to_X == (Mutation_list{Mutation{sT, 0, sA, 0.5}}));
    EXPECT(b0.hot_deltas_to_X.size() == 1 && b0.hot_deltas_to_X.at(0) == (Site_delta{sT, sA}));
    EXPECT_NEAR(g.log_alpha_mut, -mu_JC * 0.5 + std::log(mu_JC / 3) + -mu_JC * 0.5 + -std::log(P_JC(sT, sA, 1.0)) + -mu_JC * 1.0 + -std::log(P_JC(sA, sA, 1.0)), 1e-9);
    EXPECT_NEAR(g.delta_log_G, -mu(0) * nu(0) * qa(0, sT) * 0.5 + std::log(mu(0) * nu(0) * qab(0, sT, sA)) + -mu(0) * nu(0) * qa(0, sA) * 0.5 + -mu(1) * nu(1) * qa(1, sA) * 1.0, 1e-9); }
  run_propose_new_graft_test(C, 1000, true);   // :844-846
  for (auto& k : std::vector<Case>{{a_, b_, 0.6}, {a_, x_, -0.4}, {a_, c_, 0.5}, {a_, r_, -1.5}, {b_, a_, 0.6}, {b_, x_, -0.4}, {b_, c_, 0.5}, {b_, r_, -1.5},
                                   {c_, x_, -0.4}, {c_, r_, -1.3}, {c_, a_, 0.5}, {c_, b_, 1.0}, {x_, c_, -0.4}, {x_, r_, -1.6}})   // :1692-1720
    for (int seed = 0; seed < 200; ++seed) { Rng rng; rng.key = seed + 12345; run_full_spr_move_test(C, k.X, k.SS, k.t, rng); }
}
TEST(spr_move_tricky_rooty_properties) { auto C = tricky_rooty(); run_propose_new_graft_test(C, 1000, true); }

// same properties on synthetic trees with random legal regraft targets
TEST(spr_move_properties_on_synthetic_trees) {
  for (int seed = 0; seed < 4; ++seed) {
    emat::SynthParams p; p.num_tips = 30; p.num_sites = 120; p.mu = 6e-4; p.gaps_per_tip = 3; p.mean_gap_len = 10; p.seed = 500 + seed;
    auto R = emat::make_synthetic_emat(p);
    SprCtx C(tree_from_flat(R.tree, R.ref_sequence), hky_evo(p.num_sites, p.mu, p.kappa, p.pi));
    for (int l = 0; l < p.num_sites; ++l) C.evo.nu_l[l] = 0.5 + (l % 5) * 0.3;
    C.refresh();
    run_propose_new_graft_test(C, 300, true);
    Rng pick; pick.key = 777 + seed;
    for (int it = 0; it < 400; ++it) {
      int N = C.tree.size();
      int X = C.tree.root; while (X == C.tree.root) X = pick.uniform_int(N);
      int SS = pick.uniform_int(N);
      if (descends_from(C.tree, SS, X)) continue;
      int P = C.tree.at(X).parent;
      if (SS == P) continue;
      int GG = C.tree.at(SS).parent; if (GG == P) GG = C.tree.at(P).parent;
      double hi = std::min(C.tree.at(X).t, C.tree.at(SS).t);
      double lo = (GG == k_no_node) ? hi - 30.0 : C.tree.at(GG).t;
      if (SS == C.tree.root) { lo = C.tree.at(SS).t - 30.0; hi = std::min(C.tree.at(X).t, C.tree.at(SS).t); }
      if (!(lo < hi)) continue;
      double t = lo + (hi - lo) * (0.05 + 0.9 * pick.u01_co());
      Rng rng; rng.key = 4242 + it;
      run_full_spr_move_test(C, X, SS, t, rng);
    }
  }
}

// =================================================================================================
// coalescent priors (very_scalable_coalescent_tests.cpp:11-182, scalable_coalescent_tests.cpp)
// =================================================================================================
TEST(vsc_add_interval_cells) {   // very_scalable_coalescent_tests.cpp:29-97: partial overlaps first/last, full middle
  std::vector<double> k(6, 0.0);
  vsc::add_interval(9.5, 7.25, +1.0, k, 10.0, 1.0);      // cells: 0=(9,10], 1=(8,9], 2=(7,8]
  EXPECT_NEAR(k[0], 0.5, 1e-12); EXPECT_NEAR(k[1], 1.0, 1e-12); EXPECT_NEAR(k[2], 0.75, 1e-12); EXPECT_NEAR(k[3], 0.0, 1e-12);
  vsc::add_interval(7.25, 9.5, -1.0, k, 10.0, 1.0);      // order of endpoints irrelevant
  for (double v : k) EXPECT_NEAR(v, 0.0, 1e-12);
  vsc::add_interval(8.9, 8.4, 2.0, k, 10.0, 1.0);
  EXPECT_NEAR(k[1], 1.0, 1e-12);
  std::fill(k.begin(), k.end(), 0.0);
  vsc::add_interval(10.0, 4.0, 1.0, k, 10.0, 1.0);       // t_end exactly at the last cell's lower bound
  for (double v : k) EXPECT_NEAR(v, 1.0, 1e-12);
  EXPECT(vsc::cell_for(9.999, 10.0, 1.0) == 0 && vsc::cell_for(9.0, 10.0, 1.0) == 1 && vsc::cell_for(10.0, 10.0, 1.0) == 0);
}
// very_scalable_coalescent_tests.cpp:11-97 with the reference's own numbers: cells (6,8] (4,6] (2,4] (0,2] for t_ref = 8, t_step = 2
TEST(vsc_reference_cells_and_add_interval_cases) {
  const double t_ref = 8.0, t_step = 2.0;
  EXPECT(vsc::cell_for(7.5, t_ref, t_step) == 0); EXPECT(vsc::cell_for(6.1, t_ref, t_step) == 0); EXPECT(vsc::cell_for(5.9, t_ref, t_step) == 1);
  EXPECT(vsc::cell_for(2.5, t_ref, t_step) == 2); EXPECT(vsc::cell_for(3.5, t_ref, t_step) == 2);
  EXPECT_NEAR(vsc::cell_lbound(0, t_ref, t_step), 6.0, 1e-6); EXPECT_NEAR(vsc::cell_ubound(0, t_ref, t_step), 8.0, 1e-6);
  EXPECT_NEAR(vsc::cell_lbound(1, t_ref, t_step), 4.0, 1e-6); EXPECT_NEAR(vsc::cell_ubound(1, t_ref, t_step), 6.0, 1e-6);
  struct Case { double a, b; double want[4]; };
  const Case cases[] = {{2.0, 4.0, {0.0, 0.0, 2.0, 0.0}},     // add_interval_single_whole_cell
                        {2.0, 6.0, {0.0, 2.0, 2.0, 0.0}},     // add_interval_two_whole_cells
                        {3.0, 3.5, {0.0, 0.0, 0.5, 0.0}},     // add_interval_single_partial_cell
                        {3.5, 5.0, {0.0, 1.0, 0.5, 0.0}},     // add_interval_two_partial_cells
                        {3.5, 7.0, {1.0, 2.0, 0.5, 0.0}}};    // add_interval_two_partial_cells_and_one_whole_cell
  for (const Case& c : cases) {
    std::vector<double> cells(4, 0.0);
    vsc::add_interval(c.a, c.b, +2.0, cells, t_ref, t_step);
    for (int i = 0; i < 4; ++i) EXPECT_NEAR(cells[i], c.want[i], 1e-15);
  }
}
// distributions_tests.cpp:12-66 as the reference runs it: bounds of the bounded exponential, and for the K-truncated Poisson the ratio of
// the counts of k = K and k = K + 1 (K = max(min_k, floor(lambda))) within three sigma of (K + 1) / lambda, for all twenty combinations
TEST(distributions_reference_cases) {
  Rng rng; rng.key = 12345;
  const double lambda = 2.3, inf = std::numeric_limits<double>::infinity();
  for (int i = 0; i < 1000; ++i) {
    { Bounded_exponential_distribution d{+lambda, -inf, 5.0}; EXPECT(d(rng) <= 5.0); }
    { Bounded_exponential_distribution d{-lambda, 3.0, +inf}; EXPECT(d(rng) >= 3.0); }
    { Bounded_exponential_distribution d{lambda, 2.0, 5.0}; double x = d(rng); EXPECT(x >= 2.0 && x <= 5.0); }
  }
  for (double lam : {0.01, 0.1, 1.0, 10.0}) for (int min_k : {0, 1, 2, 5, 20}) {
    const int K = std::max(min_k, (int)std::floor(lam));
    long count_K = 0, count_K1 = 0;
    K_truncated_poisson_distribution d{lam, min_k};
    for (int sample = 0; sample < 100000; ++sample) { int k = d(rng); EXPECT(k >= min_k); if (k == K) ++count_K; else if (k == K + 1) ++count_K1; }
    EXPECT(count_K >= 10); EXPECT(count_K1 >= 10);
    const double upper = (count_K + 3 * std::sqrt((double)count_K)) / (count_K1 - 3 * std::sqrt((double)count_K1));
    const double lower = (count_K - 3 * std::sqrt((double)count_K)) / (count_K1 + 3 * std::sqrt((double)count_K1));
    const double expected = 1.0 / (lam / (K + 1));
    EXPECT(lower < expected); EXPECT(upper > expected);
  }
}
static Phylo_tree ladder_tree(int n_tips, double dt) {   // tips at t=0, coalescences at -dt, -2dt, ...
  Phylo_tree t(2 * n_tips - 1);
  t.ref_sequence = {sA};
  for (int i = 0; i < n_tips; ++i) set_tip(t, i, k_no_node, 0.0);
  int prev = 0;
  for (int i = 1; i < n_tips; ++i) {
    int u = n_tips + i - 1;
    set_inner(t, u, k_no_node, prev, i, -dt * i);
    t.at(prev).parent = u; t.at(i).parent = u; prev = u;
  }
  t.root = prev;
  return t;
}
TEST(scalable_coalescent_matches_exact_kingman_for_fine_grid) {   // scalable_coalescent_tests.cpp `log_prior`
  int n = 10; double dt = 1.0, N0 = 7.0;
  auto t = ladder_tree(n, dt);
  auto pm = std::make_shared<Const_pop_model>(N0);
  Scalable_coalescent_prior prior(pm, t.size(), 0.0, 0.001);
  for (int i = 0; i < t.size(); ++i) { if (t.at(i).is_tip()) { prior.mark_as_tip(i); prior.displace_tip(i, t.at(i).t); } else { prior.mark_as_coalescence(i); prior.displace_coalescence(i, t.at(i).t); } }
  // exact: between coalescence i and i+1 (going back) there are k = n - i lineages for duration dt
  double expected = 0.0;
  for (int i = 0; i < n - 1; ++i) { int k = n - i; expected += -dt * k * (k - 1) / (2.0 * N0) - std::log(N0); }
  EXPECT_NEAR(prior.calc_log_prior(), expected, 1e-6);
  // delta == difference of full evaluations (scalable_coalescent_tests.cpp `delta_log_prior`)
  int node = n + 3; double new_t = t.at(node).t + 0.37;
  double before = prior.calc_log_prior();
  double delta = prior.calc_delta_log_prior_after_displace_coalescence(node, new_t);
  prior.displace_coalescence(node, new_t);
  EXPECT_NEAR(prior.calc_log_prior() - before, delta, 1e-8);
}
// scalable_coalescent_tests.cpp:10-58 `log_prior`, the reference's own case with its own numbers: 10 tips, 9 coalescences,
// constant population 20, t_ref = 0, one-day cells
TEST(scalable_coalescent_reference_log_prior_case) {
  const double pop = 20.0; const int num_tips = 10, num_nodes = 2 * num_tips - 1;
  auto pm = std::make_shared<Const_pop_model>(pop);
  Scalable_coalescent_prior prior(pm, num_nodes, 0.0, 1.0);
  for (int i = 0; i != num_tips - 1; ++i) prior.mark_as_coalescence(i);
  for (int i = num_tips - 1; i != num_nodes; ++i) prior.mark_as_tip(i);
  EXPECT_NEAR(prior.calc_log_prior(), -(num_tips - 1) * std::log(pop), 1e-8);
  const double tc[9] = {-50, -40, -40, -30, -30, -30, -30, -20, -10};
  for (int i = 0; i < 9; ++i) prior.displace_coalescence(i, tc[i]);
  double expected = 0.0 - (-40.0 - (-50.0)) * (2 * 1) / 2 / pop - (-30.0 - (-40.0)) * (4 * 3) / 2 / pop - (-20.0 - (-30.0)) * (8 * 7) / 2 / pop
      - (-10.0 - (-20.0)) * (9 * 8) / 2 / pop - (-00.0 - (-10.0)) * (10 * 9) / 2 / pop - (num_tips - 1) * std::log(pop);
  EXPECT_NEAR(prior.calc_log_prior(), expected, 1e-8);
  const double tt[5] = {-45.0, -35.0, -30.0, -30.0, -25.0};
  for (int k = 0; k < 5; ++k) prior.displace_tip(num_tips - 1 + k, tt[k]);
  expected = 0.0 - (-45.0 - (-50.0)) * (2 * 1) / 2 / pop - (-40.0 - (-45.0)) * (1 * 0) / 2 / pop - (-35.0 - (-40.0)) * (3 * 2) / 2 / pop
      - (-30.0 - (-35.0)) * (2 * 1) / 2 / pop - (-25.0 - (-30.0)) * (4 * 3) / 2 / pop - (-20.0 - (-25.0)) * (3 * 2) / 2 / pop
      - (-10.0 - (-20.0)) * (4 * 3) / 2 / pop - (-00.0 - (-10.0)) * (5 * 4) / 2 / pop - (num_tips - 1) * std::log(pop);
  EXPECT_NEAR(prior.calc_log_prior(), expected, 1e-8);
}
// scalable_coalescent_tests.cpp:60-110 `delta_log_prior`: exponential growth, t_step 0.17841, coalescence times applied in a
// shuffled order (the reference shuffles with std::random_device; every order must satisfy the same identity, three are tried)
TEST(scalable_coalescent_reference_delta_log_prior_case) {
  for (int perm = 0; perm < 3; ++perm) {
    auto pm = std::make_shared<Exp_pop_model>(0.0, 20.0, 0.1, 0.0);
    const int num_tips = 10, num_nodes = 2 * num_tips - 1;
    Scalable_coalescent_prior prior(pm, num_nodes, 0.0, 0.17841);
    for (int i = 0; i != num_tips - 1; ++i) prior.mark_as_coalescence(i);
    for (int i = num_tips - 1; i != num_nodes; ++i) prior.mark_as_tip(i);
    std::vector<double> ts{-50.0, -40.0, -40.0, -30.0, -30.0, -30.0, -30.0, -20.0, -10.0};
    Rng rng; rng.key = 900 + perm;
    for (int i = (int)ts.size() - 1; i > 0; --i) std::swap(ts[i], ts[rng.uniform_int(i + 1)]);
    double now = prior.calc_log_prior();
    for (int i = 0; i != num_tips - 1; ++i) {
      double delta = prior.calc_delta_log_prior_after_displace_coalescence(i, ts[i]);
      prior.displace_coalescence(i, ts[i]);
      double after = prior.calc_log_prior();
      EXPECT_NEAR(delta, after - now, 1e-8);
      now = after;
    }
    for (int i = 0; i != num_tips - 1; ++i) prior.displace_coalescence(i, 0.0);
    for (int i = 0; i != num_tips - 1; ++i) prior.displace_coalescence(i, -60.0);
    const double tt[5] = {-45.0, -35.0, -30.0, -30.0, -25.0};
    for (int k = 0; k < 5; ++k) prior.displace_tip(num_tips - 1 + k, tt[k]);
    now = prior.calc_log_prior();
    for (int i = 0; i != num_tips - 1; ++i) {
      double delta = prior.calc_delta_log_prior_after_displace_coalescence(i, ts[i]);
      prior.displace_coalescence(i, ts[i]);
      double after = prior.calc_log_prior();
      EXPECT_NEAR(delta, after - now, 1e-8);
      now = after;
    }
  }
}
TEST(vsc_parts_sum_to_whole_and_delta_consistency) {   // very_scalable_coalescent_tests.cpp:99-182
  emat::SynthParams p; p.num_tips = 60; p.num_sites = 50; p.seed = 31; p.tip_date_uncertainty = 5.0; p.frac_uncertain_tips = 0.3;
  auto R = emat::make_synthetic_emat(p);
  auto whole = tree_from_flat(R.tree, R.ref_sequence);
  auto pm = std::make_shared<Exp_pop_model>(R.t_max_tip, 300.0, 0.004, 0.0);
  // one part == whole tree: with a single active part the Gaussian coupling cancels in log-prior DIFFERENCES
  Rng rng; rng.key = 5;
  std::vector<const Phylo_tree*> st{&whole}; std::vector<Rng*> pr{&rng};
  auto parts = make_very_scalable_coalescent_prior_parts(st, 0, pm, pr, 3.0);
  auto& part = parts[0];
  double base = part.calc_partial_log_prior();
  Rng mv; mv.key = 17;
  for (int it = 0; it < 300; ++it) {
    int node = mv.uniform_int(whole.size());
    auto& nd = whole.at(node);
    double old_t = nd.t, new_t;
    if (nd.is_tip()) { if (nd.t_min == nd.t_max) continue; new_t = mv.uniform_co(std::max((double)nd.t_min, whole.at(nd.parent).t), nd.t_max); }
    else { double lo = node == whole.root ? old_t - 20.0 : whole.at(nd.parent).t; double hi = std::min(whole.at(nd.children[0]).t, whole.at(nd.children[1]).t); new_t = mv.uniform_co(lo, hi); }
    double delta = nd.is_tip() ? part.calc_delta_partial_log_prior_after_displace_tip(old_t, new_t) : part.calc_delta_partial_log_prior_after_displace_coalescence(old_t, new_t);
    double before = part.calc_partial_log_prior();   // after ensure_space possibly grew the grid
    if (nd.is_tip()) part.tip_displaced(old_t, new_t); else part.coalescence_displaced(old_t, new_t);
    nd.t = new_t;
    EXPECT_NEAR(part.calc_partial_log_prior() - before, delta, 1e-8);
  }
  (void)base;
  // k_bar_p from scratch equals the incrementally maintained one
  Rng rng2; rng2.key = 5; std::vector<Rng*> pr2{&rng2};
  auto fresh = make_very_scalable_coalescent_prior_parts(st, 0, pm, pr2, 3.0);
  size_t ncommon = std::min(fresh[0].k_bar_p.size(), part.k_bar_p.size());
  for (size_t i = 0; i < ncommon; ++i) EXPECT_NEAR(fresh[0].k_bar_p[i], part.k_bar_p[i], 1e-9);
}

// =================================================================================================
// population models: literal expectations (pop_model_tests.cpp:45-68, 88-234, 281-780)
// =================================================================================================
TEST(pop_models_literals) {
  Exp_pop_model e(0.0, 1.0, std::log(2.0), 0.0);   // doubles every day (pop_model_tests.cpp:45-68)
  EXPECT_NEAR(e.pop_at_time(0.0), 1.0, 1e-12); EXPECT_NEAR(e.pop_at_time(1.0), 2.0, 1e-12); EXPECT_NEAR(e.pop_at_time(-1.0), 0.5, 1e-12);
  EXPECT_NEAR(e.pop_integral(0.0, 1.0), 1.0 / std::log(2.0), 1e-12);
  EXPECT_NEAR(e.intensity_integral(0.0, 1.0), 0.5 / std::log(2.0), 1e-12);
  Exp_pop_model b(0.0, 1.0, std::log(2.0), 0.25);  // barrier at 1/4 => t_c = -2 (pop_model_tests.cpp:88-234)
  EXPECT_NEAR(b.t_c, -2.0, 1e-12);
  EXPECT_NEAR(b.pop_at_time(-3.0), 0.25, 1e-12); EXPECT_NEAR(b.pop_at_time(-1.0), 0.5, 1e-12);
  EXPECT_NEAR(b.pop_integral(-4.0, -3.0), 0.25, 1e-12);
  EXPECT_NEAR(b.pop_integral(-3.0, -1.0), 0.25 + (0.5 - 0.25) / std::log(2.0), 1e-12);
  EXPECT_NEAR(b.intensity_integral(-3.0, -1.0), 4.0 + (4.0 - 2.0) / std::log(2.0), 1e-12);
  Exp_pop_model dn(0.0, 1.0, -std::log(2.0), 0.25);   // shrinking: clamped for t >= 2
  EXPECT_NEAR(dn.t_c, 2.0, 1e-12);
  EXPECT_NEAR(dn.pop_integral(1.0, 3.0), (0.5 - 0.25) / std::log(2.0) + 0.25, 1e-12);
  Const_pop_model c(3.0);
  EXPECT_NEAR(c.pop_integral(1.0, 3.0), 6.0, 1e-12); EXPECT_NEAR(c.intensity_integral(1.0, 4.0), 1.0, 1e-12);
  // Skygrid with knots {1,2,4,8}, gamma {-4,7,3,1} (pop_model_tests.cpp:281-780)
  std::vector<double> xs{1, 2, 4, 8}, gs{-4, 7, 3, 1};
  Skygrid_pop_model st(xs, gs, Skygrid_pop_model::k_staircase), ll(xs, gs, Skygrid_pop_model::k_log_linear);
  EXPECT(st.interval_containing_t(0.5) == 0 && st.interval_containing_t(1.0) == 0 && st.interval_containing_t(1.5) == 1 && st.interval_containing_t(2.0) == 1 && st.interval_containing_t(8.0) == 3 && st.interval_containing_t(9.0) == 4);
  EXPECT_NEAR(st.log_N(0.0), -4, 1e-12); EXPECT_NEAR(st.log_N(1.5), 7, 1e-12); EXPECT_NEAR(st.log_N(3.0), 3, 1e-12); EXPECT_NEAR(st.log_N(5.0), 1, 1e-12); EXPECT_NEAR(st.log_N(100.0), 1, 1e-12);
  EXPECT_NEAR(ll.log_N(1.5), 0.5 * -4 + 0.5 * 7, 1e-12); EXPECT_NEAR(ll.log_N(3.0), 5.0, 1e-12); EXPECT_NEAR(ll.log_N(6.0), 2.0, 1e-12); EXPECT_NEAR(ll.log_N(0.0), -4, 1e-12);
  EXPECT_NEAR(st.pop_integral(0.0, 1.0), std::exp(-4.0), 1e-12);
  EXPECT_NEAR(st.pop_integral(1.5, 3.0), 0.5 * std::exp(7.0) + 1.0 * std::exp(3.0), 1e-9);
  EXPECT_NEAR(st.pop_integral(0.0, 10.0), std::exp(-4.0) + std::exp(7.0) + 2 * std::exp(3.0) + 4 * std::exp(1.0) + 2 * std::exp(1.0), 1e-9);
  EXPECT_NEAR(st.intensity_integral(1.5, 3.0), 0.5 * std::exp(-7.0) + 1.0 * std::exp(-3.0), 1e-12);
  // log-linear: int_a^b exp(g0 + (t-x0) s) dt with slope s
  auto seg = [](double g0, double g1, double x0, double x1, double a, double b) { double s = (g1 - g0) / (x1 - x0); return (std::exp(g0 + (b - x0) * s) - std::exp(g0 + (a - x0) * s)) / s; };
  EXPECT_NEAR(ll.pop_integral(1.25, 1.75), seg(-4, 7, 1, 2, 1.25, 1.75), 1e-9 * seg(-4, 7, 1, 2, 1.25, 1.75));
  EXPECT_NEAR(ll.pop_integral(0.5, 5.0), 0.5 * std::exp(-4.0) + seg(-4, 7, 1, 2, 1, 2) + seg(7, 3, 2, 4, 2, 4) + seg(3, 1, 4, 8, 4, 5), 1e-9 * std::exp(7.0));
  EXPECT_NEAR(ll.intensity_integral(2.5, 3.5), seg(-7, -3, 2, 4, 2.5, 3.5), 1e-12);
}
// HKY model: rows sum to zero, detailed balance pi_a q_ab = pi_b q_ba, mean rate 1 (evo_hky.cpp:7-50)
// ---- tree editing sessions: the reference's own ten cases (tests/tree_editing_tests.cpp:127-1115) ----------------
// Each case rebuilds the reference's fixture (same topology, times, mutations and missations), plays the same
// sequence of elementary edits and checks the same expectations, plus tree integrity, lambda_i and the per-node
// missing-site counts as the reference's cases do at their end.
struct EditCtx {
  Phylo_tree tree, old_tree; Global_evo_model evo; std::vector<double> cumQ, lambda_i; std::vector<int> num_missing;
  explicit EditCtx(Phylo_tree t) : tree(std::move(t)), evo(fixture_evo()) { rebase(); }
  void rebase() {   // after a per-case tweak of the fixture
    old_tree = tree;
    cumQ = calc_cum_Q_l_for_sequence(tree.ref_sequence, evo);
    lambda_i = calc_lambda_i(tree, evo, cumQ);
    num_missing = calc_num_sites_missing_at_every_node(tree);
  }
  Tree_editing_session session(Node_index X) { return Tree_editing_session(tree, X, evo, lambda_i, cumQ, num_missing); }
  void check_derived() {
    EXPECT(check_phylo_tree_integrity(tree).empty());
    for (int n = 0; n < tree.size(); ++n) EXPECT_NEAR(lambda_i[n], calc_lambda_at_node(tree, n, evo, cumQ), 1e-6);
    EXPECT(num_missing == calc_num_sites_missing_at_every_node(tree));
  }
  using Elems = std::vector<std::pair<int, State>>;
  Elems missing_at(Node_index n) const {   // Missation_map::slow_elements (mutations.h): every missing site with its from-state
    Elems r;
    for (auto& iv : tree.at(n).missations.intervals.v)
      for (int l = iv.first; l < iv.second; ++l) r.push_back({l, tree.at(n).missations.get_from_state(l, tree.ref_sequence)});
    return r;
  }
  bool muts_are(Node_index n, Mutation_list want) const { return tree.at(n).mutations == want; }
  bool same_muts(Node_index n) const { return tree.at(n).mutations == old_tree.at(n).mutations; }
  bool same_miss(Node_index n) const { return tree.at(n).missations == old_tree.at(n).missations; }
  bool same_t(Node_index n) const { return tree.at(n).t == old_tree.at(n).t; }
};
// tree_editing_tests.cpp:58-124
static Phylo_tree edit_simple_tree() {
  Phylo_tree t(5);
  t.root = r_; t.ref_sequence = {sA, sC, sA, sA};
  set_inner(t, r_, k_no_node, x_, c_, -1.0); t.at(r_).mutations = {Mutation{sC, 1, sA, NEG}};
  set_inner(t, x_, r_, a_, b_, 0.0); t.at(x_).mutations = {Mutation{sA, 0, sT, -0.5}}; miss(t, x_, {{2, sA}});
  set_tip(t, a_, x_, 1.0); t.at(a_).mutations = {Mutation{sT, 0, sC, 0.5}};
  set_tip(t, b_, x_, 2.0); t.at(b_).mutations = {Mutation{sA, 1, sG, 1.0}};
  set_tip(t, c_, r_, 3.0); t.at(c_).mutations = {Mutation{sA, 0, sG, 1.0}}; miss(t, c_, {{1, sA}});
  return t;
}
TEST(tree_editing_slide_up_simple) {   // :127-190
  EditCtx C(edit_simple_tree());
  auto e = C.session(a_); e.slide_P_along_branch(-1.0); e.end();
  EXPECT(C.same_t(a_)); EXPECT(C.same_t(b_)); EXPECT(C.same_t(c_)); EXPECT(C.same_t(r_));
  EXPECT_NEAR(C.tree.at(x_).t, -1.0, 1e-6);
  EXPECT(C.muts_are(a_, {Mutation{sA, 0, sC, 0.0}}));
  EXPECT(C.muts_are(b_, {Mutation{sA, 0, sT, -0.5}, Mutation{sA, 1, sG, 1.0}}));
  EXPECT(C.same_muts(c_)); EXPECT(C.tree.at(x_).mutations.empty()); EXPECT(C.same_muts(r_));
  for (int n : {a_, b_, c_, x_, r_}) EXPECT(C.same_miss(n));
  C.check_derived();
}
TEST(tree_editing_slide_up_missation_kills_mutation) {   // :192-258
  EditCtx C(edit_simple_tree());
  C.tree.at(a_).mutations = {}; C.tree.at(a_).missations.clear(); miss(C.tree, a_, {{0, sT}});
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  C.rebase();
  auto e = C.session(a_); e.slide_P_along_branch(-1.0); e.end();
  EXPECT(C.same_t(a_)); EXPECT(C.same_t(b_)); EXPECT(C.same_t(c_)); EXPECT(C.same_t(r_));
  EXPECT_NEAR(C.tree.at(x_).t, -1.0, 1e-6);
  EXPECT(C.tree.at(a_).mutations.empty());
  EXPECT(C.muts_are(b_, {Mutation{sA, 0, sT, -0.5}, Mutation{sA, 1, sG, 1.0}}));
  EXPECT(C.same_muts(c_)); EXPECT(C.tree.at(x_).mutations.empty()); EXPECT(C.same_muts(r_));
  EXPECT(C.missing_at(a_) == (EditCtx::Elems{{0, sA}}));
  for (int n : {b_, c_, x_, r_}) EXPECT(C.same_miss(n));
  C.check_derived();
}
TEST(tree_editing_slide_up_against_edge_mutation) {   // :836-899
  EditCtx C(edit_simple_tree());
  C.tree.at(x_).mutations = {Mutation{sA, 0, sT, C.tree.at(r_).t}};
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  C.rebase();
  auto e = C.session(a_); e.slide_P_along_branch(C.tree.at(r_).t); e.end();
  EXPECT(C.same_t(a_)); EXPECT(C.same_t(b_)); EXPECT(C.same_t(c_)); EXPECT(C.same_t(r_));
  EXPECT_NEAR(C.tree.at(x_).t, -1.0, 1e-6);
  EXPECT(C.muts_are(a_, {Mutation{sA, 0, sC, 0.0}}));
  EXPECT(C.muts_are(b_, {Mutation{sA, 0, sT, -1.0}, Mutation{sA, 1, sG, 1.0}}));
  EXPECT(C.same_muts(c_)); EXPECT(C.tree.at(x_).mutations.empty()); EXPECT(C.same_muts(r_));
  for (int n : {a_, b_, c_, x_, r_}) EXPECT(C.same_miss(n));
  C.check_derived();
}
// tree_editing_tests.cpp:262-345 (x at t = 0) and :676-760 (the SPR fixture: the same tree with x at t = 1)
namespace hf { enum { r = 0, x = 1, y = 2, a = 3, b = 4, c = 5, d = 6 }; }
static Phylo_tree edit_hop_flip_tree(double t_x) {
  using namespace hf;
  Phylo_tree t(7);
  t.root = r; t.ref_sequence = {sA, sC, sA, sT};
  set_inner(t, r, k_no_node, y, d, -1.0);
  set_inner(t, y, r, x, c, 0.0); t.at(y).mutations = {Mutation{sA, 2, sG, -0.75}, Mutation{sT, 3, sA, -0.25}};
  set_inner(t, x, y, a, b, t_x); miss(t, x, {{2, sG}});
  set_tip(t, a, x, 2.0); miss(t, a, {{0, sA}});
  set_tip(t, b, x, 3.0); t.at(b).mutations = {Mutation{sA, 3, sC, 2.0}}; miss(t, b, {{1, sC}});
  set_tip(t, c, y, 1.0); miss(t, c, {{1, sC}});
  set_tip(t, d, r, -0.5);
  return t;
}
static void expect_parents(EditCtx& C, std::vector<int> want) {
  for (int n = 0; n < C.tree.size(); ++n) EXPECT(C.tree.at(n).parent == want[n]);
}
static void expect_after_hop(EditCtx& C) {   // what :389-431 and :602-644 both expect
  using namespace hf;
  for (int n = 0; n < 7; ++n) EXPECT(C.same_t(n));
  for (int n : {a, b, c, d, r}) EXPECT(C.same_muts(n));
  EXPECT(C.tree.at(x).mutations == C.old_tree.at(y).mutations);
  EXPECT(C.tree.at(y).mutations == C.old_tree.at(x).mutations);
  EXPECT(C.missing_at(a) == (EditCtx::Elems{{0, sA}, {2, sG}}));
  EXPECT(C.missing_at(b) == (EditCtx::Elems{{2, sG}}));
  EXPECT(C.tree.at(c).missations.empty()); EXPECT(C.same_miss(d)); EXPECT(C.same_miss(r));
  EXPECT(C.tree.at(x).missations.empty());
  EXPECT(C.missing_at(y) == (EditCtx::Elems{{1, sC}}));
  //                r        x  y  a  b  c  d
  expect_parents(C, {k_no_node, r, x, x, y, y, r});
  EXPECT(C.tree.root == r);
  C.check_derived();
}
TEST(tree_editing_hop_up) {   // :347-431
  EditCtx C(edit_hop_flip_tree(0.0));
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  auto e = C.session(hf::a); e.hop_up(); e.end();
  expect_after_hop(C);
}
TEST(tree_editing_hop_down) {   // :560-644
  EditCtx C(edit_hop_flip_tree(0.0));
  auto e = C.session(hf::c); e.hop_down(hf::b); e.end();
  expect_after_hop(C);
}
TEST(tree_editing_flip) {   // :433-558
  using namespace hf;
  EditCtx C(edit_hop_flip_tree(0.0));
  auto e = C.session(b); e.flip(); e.end();
  for (int n = 0; n < 7; ++n) EXPECT(C.same_t(n));
  EXPECT(C.same_muts(a)); EXPECT(C.muts_are(b, {Mutation{sA, 3, sC, 1.5}}));
  for (int n : {c, d, x, y, r}) EXPECT(C.same_muts(n));   // an upstream flip does not swap the x and y mutations
  EXPECT(C.missing_at(a) == (EditCtx::Elems{{0, sA}, {2, sG}}));
  EXPECT(C.missing_at(b) == (EditCtx::Elems{{2, sG}}));
  EXPECT(C.tree.at(c).missations.empty()); EXPECT(C.same_miss(d)); EXPECT(C.same_miss(r));
  EXPECT(C.missing_at(x) == (EditCtx::Elems{{1, sC}}));
  EXPECT(C.tree.at(y).missations.empty());
  expect_parents(C, {k_no_node, y, r, y, x, x, r});
  EXPECT(C.tree.root == r);
  C.check_derived();
}
TEST(tree_editing_spr) {   // :762-834
  using namespace hf;
  EditCtx C(edit_hop_flip_tree(1.0));
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  auto e = C.session(a);
  e.slide_P_along_branch(C.tree.at(y).t);
  e.hop_up();
  e.slide_P_along_branch(C.tree.at(r).t);
  e.slide_P_along_branch(C.tree.at(y).t);
  e.hop_down(b);
  e.flip();
  e.slide_P_along_branch(0.5);
  e.end();
  for (int n : {a, b, c, d, y, r}) EXPECT(C.same_t(n));
  EXPECT(C.tree.at(x).t == 0.5);
  EXPECT(C.tree.at(a).mutations.empty());
  EXPECT(C.muts_are(b, {Mutation{sA, 3, sC, 2.0}}));
  EXPECT(C.tree.at(c).mutations.empty()); EXPECT(C.tree.at(x).mutations.empty());
  for (int n : {d, y, r}) EXPECT(C.same_muts(n));
  EXPECT(C.missing_at(a) == (EditCtx::Elems{{0, sA}, {2, sG}}));
  EXPECT(C.missing_at(b) == (EditCtx::Elems{{1, sC}, {2, sG}}));
  EXPECT(C.missing_at(c) == (EditCtx::Elems{{1, sC}}));
  EXPECT(C.tree.at(x).missations.empty());
  for (int n : {d, y, r}) EXPECT(C.same_miss(n));
  expect_parents(C, {k_no_node, y, r, x, y, x, r});
  EXPECT(C.tree.root == r);
  C.check_derived();
}
// tree_editing_tests.cpp:901-962
static Phylo_tree edit_slide_down_tree() {
  Phylo_tree t(5);
  t.root = r_; t.ref_sequence = {sA, sA, sA, sA};
  set_inner(t, r_, k_no_node, x_, c_, 0.0);
  set_inner(t, x_, r_, a_, b_, 1.0); t.at(x_).mutations = {Mutation{sA, 0, sT, 0.5}};
  set_tip(t, a_, x_, 4.0); t.at(a_).mutations = {Mutation{sA, 1, sG, 4.0}}; miss(t, a_, {{0, sT}});
  set_tip(t, b_, x_, 3.0); t.at(b_).mutations = {Mutation{sA, 1, sC, 1.5}, Mutation{sT, 0, sG, 2.5}};
  set_tip(t, c_, r_, 0.5);
  return t;
}
TEST(tree_editing_slide_down) {   // :964-1000 (tip b is the lower end of the branch the junction slides down)
  EditCtx C(edit_slide_down_tree());
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  auto e = C.session(a_); e.slide_P_along_branch(3.0); e.end();
  EXPECT(C.same_t(a_)); EXPECT(C.same_t(b_)); EXPECT(C.same_t(c_)); EXPECT(C.same_t(r_));
  EXPECT_NEAR(C.tree.at(x_).t, 3.0, 1e-6);
  EXPECT(C.muts_are(a_, {Mutation{sC, 1, sG, 3.5}}));
  EXPECT(C.tree.at(b_).mutations.empty()); EXPECT(C.same_muts(c_));
  EXPECT(C.muts_are(x_, {Mutation{sA, 0, sT, 0.5}, Mutation{sA, 1, sC, 1.5}, Mutation{sT, 0, sG, 2.5}}));
  EXPECT(C.same_muts(r_));
  EXPECT(C.missing_at(a_) == (EditCtx::Elems{{0, sG}}));
  for (int n : {b_, c_, x_, r_}) EXPECT(C.same_miss(n));
  C.check_derived();
}
TEST(tree_editing_slide_down_against_edge_mutation) {   // :1002-1047
  EditCtx C(edit_slide_down_tree());
  C.tree.at(b_).mutations = {Mutation{sA, 1, sC, 1.5}, Mutation{sT, 0, sG, C.tree.at(b_).t}};
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  C.rebase();
  auto e = C.session(a_); e.slide_P_along_branch(C.tree.at(b_).t); e.end();
  EXPECT(C.same_t(a_)); EXPECT(C.same_t(b_)); EXPECT(C.same_t(c_)); EXPECT(C.same_t(r_));
  EXPECT_NEAR(C.tree.at(x_).t, 3.0, 1e-6);
  EXPECT(C.muts_are(a_, {Mutation{sC, 1, sG, 3.5}}));
  EXPECT(C.tree.at(b_).mutations.empty()); EXPECT(C.same_muts(c_));
  EXPECT(C.muts_are(x_, {Mutation{sA, 0, sT, 0.5}, Mutation{sA, 1, sC, 1.5}, Mutation{sT, 0, sG, 3.0}}));
  EXPECT(C.same_muts(r_));
  EXPECT(C.missing_at(a_) == (EditCtx::Elems{{0, sG}}));
  for (int n : {b_, c_, x_, r_}) EXPECT(C.same_miss(n));
  C.check_derived();
}
TEST(tree_editing_slide_down_root) {   // :1049-1115 (X = x: its parent is the root, which slides down the root-c branch)
  EditCtx C(edit_slide_down_tree());
  C.tree.at(c_).t = 0.9; C.tree.at(c_).t_min = C.tree.at(c_).t_max = 0.9f;
  C.tree.at(x_).mutations = {Mutation{sC, 1, sA, 0.5}};
  C.tree.at(r_).mutations = {Mutation{sA, 0, sT, NEG}, Mutation{sA, 1, sC, NEG}};
  C.tree.at(c_).mutations = {Mutation{sC, 1, sA, 0.02}, Mutation{sA, 2, sG, 0.05}};
  C.tree.at(a_).mutations = {Mutation{sA, 1, sG, 4.0}};
  C.tree.at(b_).mutations = {Mutation{sA, 1, sC, 1.5}, Mutation{sT, 0, sG, 2.5}};
  EXPECT(check_phylo_tree_integrity(C.tree).empty());
  C.rebase();
  auto e = C.session(x_); e.slide_P_along_branch(0.5); e.end();
  EXPECT(C.same_t(a_)); EXPECT(C.same_t(b_)); EXPECT(C.same_t(c_)); EXPECT(C.same_t(x_));
  EXPECT_NEAR(C.tree.at(r_).t, 0.5, 1e-6);
  EXPECT(C.same_muts(a_)); EXPECT(C.same_muts(b_));
  EXPECT(C.tree.at(c_).mutations.empty());
  EXPECT(C.muts_are(x_, {Mutation{sG, 2, sA, 0.75}}));
  auto rm = C.tree.at(r_).mutations; sort_mutations(rm);   // the reference compares these unordered
  EXPECT(rm == (Mutation_list{Mutation{sA, 0, sT, NEG}, Mutation{sA, 2, sG, NEG}}));
  for (int n : {a_, b_, c_, x_, r_}) EXPECT(C.same_miss(n));
  C.check_derived();
}

TEST(hky_model_properties) {
  Hky_model h; h.mu = 1e-3; h.kappa = 5.0; double pi[4] = {0.3, 0.2, 0.2, 0.3}; for (int a = 0; a < 4; ++a) h.pi_a[a] = pi[a];
  auto m = h.derive_site_evo_model();
  double mean = 0.0;
  for (int a = 0; a < 4; ++a) { double row = 0.0; for (int b = 0; b < 4; ++b) { row += m.q_ab[a][b]; if (a != b) EXPECT_NEAR(pi[a] * m.q_ab[a][b], pi[b] * m.q_ab[b][a], 1e-15); } EXPECT_NEAR(row, 0.0, 1e-15); mean += pi[a] * m.q_a(a); }
  EXPECT_NEAR(mean, 1.0, 1e-14);
  EXPECT_NEAR(m.q_ab[sA][sG] / m.q_ab[sA][sC], 5.0 * pi[sG] / pi[sC], 1e-12);
}
// samplers (distributions_tests.cpp): support and first moments
TEST(distributions_support_and_means) {
  Rng rng; rng.key = 3;
  { Bounded_exponential_distribution d{2.0, 1.0, 3.0}; double s = 0; int n = 200000;
    for (int i = 0; i < n; ++i) { double x = d(rng); EXPECT(x >= 1.0 && x <= 3.0); s += x; }
    double Z = (std::exp(6.0) - std::exp(2.0)) / 2.0; double mean = ((3 * std::exp(6.0) - std::exp(2.0)) / 2.0 - (std::exp(6.0) - std::exp(2.0)) / 4.0) / Z;
    EXPECT_NEAR(s / n, mean, 5e-3); }
  { Bounded_exponential_distribution d{-500.0, 1.0, 3.0}; for (int i = 0; i < 1000; ++i) { double x = d(rng); EXPECT(x >= 1.0 && x < 1.1); } }
  { Bounded_exponential_distribution d{0.0, 1.0, 3.0}; double s = 0; for (int i = 0; i < 100000; ++i) s += d(rng); EXPECT_NEAR(s / 100000, 2.0, 1e-2); }
  { K_truncated_poisson_distribution d{0.05, 1}; double s = 0; int n = 200000; for (int i = 0; i < n; ++i) { int k = d(rng); EXPECT(k >= 1); s += k; }
    EXPECT_NEAR(s / n, 0.05 / (1 - std::exp(-0.05)), 5e-3); }
  { K_truncated_poisson_distribution d{0.05, 2}; for (int i = 0; i < 10000; ++i) EXPECT(d(rng) >= 2); }
  { K_truncated_poisson_distribution d{3.0, 1}; double s = 0; int n = 200000; for (int i = 0; i < n; ++i) { int k = d(rng); EXPECT(k >= 1); s += k; } EXPECT_NEAR(s / n, 3.0 / (1 - std::exp(-3.0)), 2e-2); }
  { double s = 0, s2 = 0; int n = 400000; for (int i = 0; i < n; ++i) { double z = rng.gaussian(1.0, 2.0); s += z; s2 += z * z; } EXPECT_NEAR(s / n, 1.0, 2e-2); EXPECT_NEAR(s2 / n - (s / n) * (s / n), 4.0, 5e-2); }
  // truncated gamma sampler stays in range and round-trips the inverse
  for (int i = 0; i < 2000; ++i) { double x = safe_sample_truncated_gamma(1.8, 0.3, 0.5, 40.0, rng); EXPECT(x >= 0.5 && x <= 40.0); }
  // the reference's own sampler cases (safe_gamma_math_tests.cpp:126-179): basic, large alpha, a sweep starting at the mean,
  // an unbounded upper end -- and one deep in the tail, where every uniform draw in Q is below 1e-100
  for (int i = 0; i < 100; ++i) { double x = safe_sample_truncated_gamma(5.0, 2.0, 1.0, 10.0, rng); EXPECT(x >= 1.0 && x <= 10.0); }
  for (int i = 0; i < 100; ++i) { double x = safe_sample_truncated_gamma(271.4, 1.0, 100.0, 500.0, rng); EXPECT(x >= 100.0 && x <= 500.0); }
  for (double alpha : {1.0, 5.0, 10.0, 50.0, 100.0}) for (double beta : {0.5, 1.0, 2.0, 5.0}) {
    double lo = alpha / beta, hi = lo + 10.0;
    for (int i = 0; i < 10; ++i) { double x = safe_sample_truncated_gamma(alpha, beta, lo, hi, rng); EXPECT(x >= lo && x <= hi); }
  }
  for (int i = 0; i < 100; ++i) { double x = safe_sample_truncated_gamma(5.0, 1.0, 1.0, std::numeric_limits<double>::infinity(), rng); EXPECT(x >= 1.0); }
  { double s = 0; int n = 20000;   // Gamma(11, 1) conditioned on [300, 400]: density ~ x^10 e^-x, log-density slope -r = 10/300 - 1, curvature -c = -10/300^2: mean = 300 + 1/r - 2c/r^3
    for (int i = 0; i < n; ++i) { double x = safe_sample_truncated_gamma(11.0, 1.0, 300.0, 400.0, rng); EXPECT(x > 300.0 && x < 400.0); s += x; }
    EXPECT_NEAR(s / n, 300.0 + 1.0 / (1.0 - 10.0 / 300.0) - 2.0 * 10.0 / (300.0 * 300.0) / std::pow(1.0 - 10.0 / 300.0, 3), 3e-2); }
  for (double a : {0.5, 1.0, 1.8, 4.2, 25.0}) for (double q : {1e-8, 1e-3, 0.2, 0.5, 0.9, 0.999}) { double x = safe_gamma_q_inv(a, q); EXPECT_NEAR(gamma_q(a, x), q, 1e-11 * std::max(1.0, 1.0 / q) * q + 1e-13); }
}
// mutational-history sampler: endpoint constraints (spr_move_tests.cpp:1795-2002)
TEST(sample_mutational_history_constraints) {
  Rng rng; rng.key = 11;
  Site_deltas deltas{{2, Site_delta{sA, sC}}, {7, Site_delta{sG, sT}}};
  int with_extra = 0;
  for (int it = 0; it < 20000; ++it) {
    auto h = sample_mutational_history(10, 2.0, 0.125, deltas, rng);
    std::map<int, State> cur; std::map<int, State> first;
    double prev_t = -1e300;
    for (auto& m : h) {
      EXPECT(m.t >= -2.0 && m.t < 0.0 && m.t >= prev_t); prev_t = m.t;
      State expect_from = cur.count(m.site) ? cur[m.site] : (deltas.count(m.site) ? deltas.at(m.site).from : sA);
      EXPECT(m.from == expect_from && m.from != m.to);
      cur[m.site] = m.to;
    }
    for (auto& [l, d] : deltas) EXPECT(cur.count(l) && cur[l] == d.to);
    for (auto& [l, s] : cur) if (!deltas.count(l)) { EXPECT(s == sA); ++with_extra; }
  }
  EXPECT(with_extra > 0);   // A->x->A excursions on delta-free sites do occur at mu T = 0.25
  // tiny mu*T: must terminate and return nothing (spr_move_tests.cpp:1963-2002)
  auto h = sample_mutational_history(30000, 1e-3, 1e-9, {}, rng); EXPECT(h.empty());
  auto u = sample_unconstrained_mutational_history(30000, 1e-3, 1e-9, rng); EXPECT(u.empty());
  // unconstrained: right-to-left Gillespie ends in A everywhere
  for (int it = 0; it < 2000; ++it) {
    auto hh = sample_unconstrained_mutational_history(10, 2.0, 0.125, rng);
    std::map<int, State> cur;
    for (auto& m : hh) { if (cur.count(m.site)) EXPECT(cur[m.site] == m.from); cur[m.site] = m.to; }
    for (auto& [l, s] : cur) EXPECT(s == sA);
  }
}

// spr_move_tests.cpp:1795-1961: on the simple fixture, 25 000 histories of length T = 1 / mu_JC ending at a random point of the tree,
// from the target start sequence TGCA: site 3 is A at both ends everywhere, so every mutation there is part of an A ~> A excursion --
// "unusual" (any mutation at site 3) with probability (p* - p_0) / p*, "super-unusual" (more than two) with (p* - p_0 - p_1 - p_2) / p*,
// where p_m = (mu T)^m e^(-mu T) (P^m)_AA / m!: 0.17828634147519676 and 0.04133406505439621 at mu T = 1 (the reference's numbers,
// :1947-1952), within three binomial standard errors as there (the generator differs, so the counts do, the frequencies must not).
TEST(sample_mutational_history_unusual_trajectory_frequencies) {
  SprCtx C(complex_tree(true), fixture_evo());
  const int L = C.tree.num_sites(), num_histories = 25000;
  const std::vector<State> target_start_seq{sT, sG, sC, sA};
  int num_unusual = 0, num_super_unusual = 0;
  // the theory, recomputed: (P^m)_AA for the jump chain of JC69 is 1/4 + 3/4 (-1/3)^m
  { double pstar = 0, p[3] = {0, 0, 0}, f = 1;
    for (int m = 0; m < 40; ++m) { if (m > 0) f *= m; double pm = std::exp(-1.0) / f * (0.25 + 0.75 * std::pow(-1.0 / 3.0, m)); pstar += pm; if (m < 3) p[m] = pm; }
    EXPECT_NEAR((pstar - p[0]) / pstar, 0.17828634147519676, 1e-9); EXPECT_NEAR((pstar - p[0] - p[1] - p[2]) / pstar, 0.04133406505439621, 1e-9); }
  for (int seed = 0; seed < num_histories; ++seed) {
    Rng rng; rng.key = seed + 12345;
    int branch = r_; while (branch == r_) branch = rng.uniform_int(C.tree.size());
    const double t_end = rng.uniform_co(C.tree.branch_begin_t(branch), C.tree.at(branch).t);
    Phylo_tree_loc end_loc{branch, t_end};
    const double T = 1.0 / mu_JC;
    auto end_seq = target_start_seq;
    for (int l = 0; l < L; ++l) end_seq[l] = calc_site_state_at(C.tree, end_loc, l);
    EXPECT(end_seq[3] == sA);
    Site_deltas deltas;
    for (int l = 0; l < L; ++l) if (target_start_seq[l] != end_seq[l]) deltas.insert({l, Site_delta{target_start_seq[l], end_seq[l]}});
    auto h = sample_mutational_history(L, T, mu_JC, deltas, rng);
    adjust_mutational_history(h, deltas, C.tree, end_loc);
    auto seq = target_start_seq; double prev_t = -1e300; int at3 = 0;
    for (auto& m : h) {
      EXPECT(m.t >= prev_t && m.t >= t_end - T && m.t <= t_end); prev_t = m.t;
      EXPECT(seq[m.site] == m.from); seq[m.site] = m.to;
      if (m.site == 3) ++at3;
    }
    EXPECT(seq == end_seq);
    if (at3 > 0) ++num_unusual;
    if (at3 > 2) ++num_super_unusual;
  }
  const double pu = num_unusual / (double)num_histories, eu = std::sqrt(pu * (1 - pu) / num_histories);
  const double ps = num_super_unusual / (double)num_histories, es = std::sqrt(ps * (1 - ps) / num_histories);
  std::printf("  unusual %.5f +- %.5f (expected 0.17829), super-unusual %.5f +- %.5f (expected 0.04133)\n", pu, eu, ps, es);
  EXPECT_NEAR(pu, 0.17828634147519676, 3 * eu);
  EXPECT_NEAR(ps, 0.04133406505439621, 3 * es);
}

// =================================================================================================
// whole Subrun chains: the reference's debug invariants after every move
// (Subrun::check_derived_quantities subrun.cpp:28-56 + assert_phylo_tree_integrity)
// =================================================================================================
static void run_chain(int tips, int sites, uint64_t seed, bool includes_root, int moves, int pop_kind) {
  emat::SynthParams p; p.num_tips = tips; p.num_sites = sites; p.mu = 4e-4; p.gaps_per_tip = 2; p.mean_gap_len = 15; p.seed = seed;
  p.tip_date_uncertainty = 3.0; p.frac_uncertain_tips = 0.25;
  auto R = emat::make_synthetic_emat(p);
  auto tree = tree_from_flat(R.tree, R.ref_sequence);
  auto evo = hky_evo(sites, p.mu, p.kappa, p.pi);
  Rng rng; rng.key = seed * 7 + 1;
  Subrun sr(rng, tree, includes_root, evo);
  sr.t_max_tip = R.t_max_tip;
  std::shared_ptr<const Pop_model> pm;
  if (pop_kind == 0) pm = std::make_shared<Exp_pop_model>(R.t_max_tip, 200.0, 0.0, 0.0);
  else if (pop_kind == 1) pm = std::make_shared<Exp_pop_model>(R.t_max_tip, 400.0, 0.003, 1.0);
  else { std::vector<double> xs, gs; for (int k = 0; k <= 10; ++k) { xs.push_back(R.t_max_tip - 400.0 + 40.0 * k); gs.push_back(std::log(250.0) + 0.3 * std::sin(k)); } pm = std::make_shared<Skygrid_pop_model>(xs, gs, pop_kind == 2 ? Skygrid_pop_model::k_staircase : Skygrid_pop_model::k_log_linear); }
  std::vector<const Phylo_tree*> st{&sr.tree}; std::vector<Rng*> pr{&rng};
  auto parts = make_very_scalable_coalescent_prior_parts(st, 0, pm, pr, 2.0);
  sr.set_coalescent_prior_part(&parts[0]);
  auto tips_before = std::vector<std::vector<State>>(); std::vector<Interval_set> miss_before;
  for (int n = 0; n < sr.tree.size(); ++n) if (sr.tree.at(n).is_tip()) { tips_before.push_back(view_of_sequence_at(sr.tree, n)); miss_before.push_back(reconstruct_missing_sites_at(sr.tree, n)); }
  for (int i = 0; i < moves; ++i) {
    sr.mcmc_sub_iteration();
    auto msg = check_phylo_tree_integrity(sr.tree);
    if (msg.empty()) msg = sr.check_derived_quantities();
    ++g_checks;
    if (!msg.empty()) { ++g_fail; std::printf("FAIL [%s] seed %llu move %d kind %g: %s\n", g_test, (unsigned long long)seed, i, sr.cur_trace.kind, msg.c_str()); return; }
  }
  // tip sequences at non-missing sites never change (reference assert_tip_sequences_compatible_with_original_ones)
  size_t k = 0;
  for (int n = 0; n < sr.tree.size(); ++n) if (sr.tree.at(n).is_tip()) {
    auto sn = view_of_sequence_at(sr.tree, n); auto mn = reconstruct_missing_sites_at(sr.tree, n);
    EXPECT(mn == miss_before[k]);
    bool same = true; for (int l = 0; l < sites; ++l) if (!mn.contains(l) && sn[l] != tips_before[k][l]) same = false;
    EXPECT(same); ++k;
  }
  int64_t tot_acc = 0; for (int q = 0; q < 5; ++q) tot_acc += sr.accepted[q];
  EXPECT(tot_acc > moves / 20);
  EXPECT(sr.accepted[k_spr1] + sr.accepted[k_subtree_slide] > 0);
}
TEST(subrun_chain_invariants_root_part) { for (int s = 0; s < 4; ++s) run_chain(24, 150, 900 + s, true, 6000, s); }
TEST(subrun_chain_invariants_non_root_part) { for (int s = 0; s < 3; ++s) run_chain(30, 150, 950 + s, false, 6000, s % 2); }


// =================================================================================================
// Initial-tree construction (orc_build.hpp): the reference's own fix_up_missations cases, with its fixtures and its
// expected lists (tests/phylo_tree_tests.cpp:539-760), then the UShER-like builder under the check the reference
// itself closes it with
// =================================================================================================
static Phylo_tree three_node_tree(std::vector<State> ref) {   // r = 0 at t = 0 with tips a = 1 (t = 1) and b = 2 (t = 2)
  Phylo_tree t(3); t.root = 0; t.ref_sequence = ref;
  set_inner(t, 0, k_no_node, 1, 2, 0.0); set_tip(t, 1, 0, 1.0); set_tip(t, 2, 0, 2.0);
  return t;
}
static Phylo_tree phylo_tree_tests_complex_tree() {   // phylo_tree_tests.cpp:80-165 (c carries one mutation A0G here)
  auto t = complex_tree(false);
  t.at(c_).mutations = {Mutation{sA, 0, sG, 1.0}};
  return t;
}
static std::string consistent_missations(const Phylo_tree& t) {   // assert_missation_consistency + the rest of the integrity rules
  return check_phylo_tree_integrity(t);
}
TEST(fix_up_missations_reference_cases) {
  {   // :539-550 trivial
    Phylo_tree t(1); t.ref_sequence = {sA, sA, sA, sA}; t.root = 0; set_tip(t, 0, k_no_node, 0.0);
    fix_up_missations(t);
    EXPECT(t.at(0).mutations.empty()); EXPECT(t.at(0).missations.empty());
  }
  {   // :552-599 a missation that repeats one above it goes
    auto t = three_node_tree({sT});
    miss(t, 0, {{0, sT}}); miss(t, 2, {{0, sT}});
    fix_up_missations(t);
    EXPECT(elements_of(t.at(0).missations, t.ref_sequence) == (std::vector<std::pair<int, State>>{{0, sT}}));
    EXPECT(t.at(1).missations.empty()); EXPECT(t.at(2).missations.empty());
    EXPECT(consistent_missations(t).empty());
  }
  {   // :601-652 a mutation below a missation of its site is dropped
    auto t = three_node_tree({sA, sA});
    miss(t, 0, {{0, sA}});
    t.at(1).mutations = {Mutation{sA, 0, sT, 0.5}};
    t.at(2).mutations = {Mutation{sA, 1, sC, 1.0}};
    fix_up_missations(t);
    EXPECT(t.at(0).mutations.empty());
    EXPECT(elements_of(t.at(0).missations, t.ref_sequence) == (std::vector<std::pair<int, State>>{{0, sA}}));
    EXPECT(t.at(1).mutations.empty()); EXPECT(t.at(1).missations.empty());
    EXPECT(t.at(2).mutations == (Mutation_list{Mutation{sA, 1, sC, 1.0}})); EXPECT(t.at(2).missations.empty());
    EXPECT(consistent_missations(t).empty());
  }
  {   // :654-701 a missation on both children moves up once
    auto t = three_node_tree({sA});
    miss(t, 1, {{0, sA}}); miss(t, 2, {{0, sA}});
    fix_up_missations(t);
    EXPECT(elements_of(t.at(0).missations, t.ref_sequence) == (std::vector<std::pair<int, State>>{{0, sA}}));
    EXPECT(t.at(1).missations.empty()); EXPECT(t.at(2).missations.empty());
    EXPECT(consistent_missations(t).empty());
  }
  {   // :703-730 merge up twice and drop the mutations that end up below a missation
    auto t = phylo_tree_tests_complex_tree();
    miss(t, a_, {{0, sT}}); miss(t, b_, {{0, sT}});
    fix_up_missations(t);
    EXPECT(t.at(a_).missations.empty()); EXPECT(t.at(a_).mutations.empty());
    EXPECT(t.at(b_).missations.empty()); EXPECT(t.at(b_).mutations == (Mutation_list{Mutation{sA, 1, sG, 1.0}}));
    EXPECT(elements_of(t.at(x_).missations, t.ref_sequence) == (std::vector<std::pair<int, State>>{{0, sA}, {2, sA}}));
    EXPECT(t.at(x_).mutations.empty());
  }
  {   // :732-760 two mutations on one logical branch stay two
    auto t = phylo_tree_tests_complex_tree();
    t.at(a_).missations.clear(); miss(t, a_, {{1, sA}});
    t.at(a_).mutations = {Mutation{sA, 0, sC, +0.5}};
    t.at(x_).mutations = {Mutation{sA, 1, sG, -0.5}};
    t.at(b_).mutations = {Mutation{sG, 1, sC, 1.0}};
    fix_up_missations(t);
    EXPECT(check_phylo_tree_integrity(t).empty());
    EXPECT(t.at(x_).mutations == (Mutation_list{Mutation{sA, 1, sG, -0.5}}));
    EXPECT(t.at(b_).mutations == (Mutation_list{Mutation{sG, 1, sC, 1.0}}));
  }
}

// phylo_tree_tests.cpp:365-525: find_MRCA_of and descends_from over every pair of {a, b, c, x, r, no node} of the fixture tree, with
// the fixture's times and with all times equal (where the walk can no longer be steered by time), plus the tree-location forms.
TEST(find_MRCA_of_and_descends_from_reference_tables) {
  const Node_index order[6] = {a_, b_, c_, x_, r_, k_no_node};
  const Node_index N = k_no_node;
  const Node_index mrca[6][6] = {{a_, x_, r_, x_, r_, N}, {x_, b_, r_, x_, r_, N}, {r_, r_, c_, r_, r_, N}, {x_, x_, r_, x_, r_, N}, {r_, r_, r_, r_, r_, N}, {N, N, N, N, N, N}};
  const bool desc[6][6] = {{1, 0, 0, 1, 1, 1}, {0, 1, 0, 1, 1, 1}, {0, 0, 1, 0, 1, 1}, {0, 0, 0, 1, 1, 1}, {0, 0, 0, 0, 1, 1}, {0, 0, 0, 0, 0, 1}};
  for (int equal_times = 0; equal_times < 2; ++equal_times) {
    auto t = phylo_tree_tests_complex_tree();
    if (equal_times) for (int n = 0; n < t.size(); ++n) { t.at(n).t = 0.0; if (t.at(n).is_tip()) t.at(n).t_min = t.at(n).t_max = 0.0; }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) EXPECT(find_MRCA_of(t, order[i], order[j]) == mrca[i][j]);
    if (!equal_times) for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) EXPECT(descends_from(t, order[i], order[j]) == desc[i][j]);
  }
  auto t = phylo_tree_tests_complex_tree();
  auto node_loc = [&](Node_index n) { return Phylo_tree_loc{n, t.at(n).t}; };
  EXPECT(find_MRCA_of(t, Phylo_tree_loc{a_, 1.0}, Phylo_tree_loc{a_, 1.0}) == node_loc(a_));
  EXPECT(find_MRCA_of(t, Phylo_tree_loc{a_, 1.0}, Phylo_tree_loc{a_, 0.5}) == (Phylo_tree_loc{a_, 0.5}));
  EXPECT(find_MRCA_of(t, Phylo_tree_loc{a_, 0.5}, Phylo_tree_loc{b_, 1.0}) == node_loc(x_));
  EXPECT(find_MRCA_of(t, Phylo_tree_loc{a_, 0.5}, Phylo_tree_loc{c_, 1.0}) == node_loc(r_));
  EXPECT(find_MRCA_of(t, Phylo_tree_loc{a_, 0.5}, Phylo_tree_loc{x_, 0.0}) == node_loc(x_));
  EXPECT(find_MRCA_of(t, Phylo_tree_loc{a_, 0.5}, Phylo_tree_loc{x_, -0.5}) == (Phylo_tree_loc{x_, -0.5}));
  EXPECT(find_MRCA_of(t, Phylo_tree_loc{a_, 0.5}, Phylo_tree_loc{r_, -1.0}) == node_loc(r_));
  EXPECT(find_MRCA_of(t, Phylo_tree_loc{a_, 0.5}, Phylo_tree_loc{r_, -1.5}) == (Phylo_tree_loc{r_, -1.5}));
  const struct { Phylo_tree_loc x, a; bool expect; } locs[] = {
      {{a_, 1.0}, {a_, 1.0}, true}, {{a_, 1.0}, {a_, 0.5}, true}, {{a_, 0.5}, {a_, 1.0}, false}, {{a_, 0.5}, {b_, 1.0}, false}, {{a_, 0.5}, {c_, 1.0}, false},
      {{a_, 0.5}, {x_, 0.0}, true}, {{a_, 0.5}, {x_, -0.5}, true}, {{a_, 0.5}, {r_, -1.0}, true}, {{a_, 0.5}, {r_, -1.5}, true}};
  for (auto& q : locs) EXPECT(descends_from(t, q.x, q.a) == q.expect);
}
// phylo_tree_tests.cpp:287-363, 527-537: what the integrity check accepts (the fixture; two mutations of one site on one branch) and
// every broken tree the reference's assert_mutation_consistency / assert_missation_consistency tests expect to die on.
TEST(phylo_tree_integrity_reference_cases) {
  EXPECT(check_phylo_tree_integrity(phylo_tree_tests_complex_tree()).empty());
  { auto t = phylo_tree_tests_complex_tree(); t.at(x_).mutations = {Mutation{sA, 0, sC, -0.75}, Mutation{sC, 0, sT, -0.25}}; EXPECT(check_phylo_tree_integrity(t).empty()); }
  const Mutation bad_muts[] = {
      Mutation{sA, 0, sA, -0.5},    // from == to
      Mutation{sA, -15, sT, -0.5},  // site < 0
      Mutation{sA, 15, sT, -0.5},   // site >= genome size
      Mutation{sA, 0, sT, -1.5},    // before the parent's time
      Mutation{sA, 0, sT, +1.5},    // after the node's time
      Mutation{sC, 0, sT, -0.5}};   // `from` is not the state above (the root has A at site 0)
  for (auto& m : bad_muts) { auto t = phylo_tree_tests_complex_tree(); t.at(x_).mutations = {m}; EXPECT(!check_phylo_tree_integrity(t).empty()); }
  { auto t = phylo_tree_tests_complex_tree(); t.at(a_).missations.clear(); miss(t, a_, {{2, sA}}); EXPECT(!check_phylo_tree_integrity(t).empty()); }   // below A2N of r->x
  { auto t = phylo_tree_tests_complex_tree(); t.at(x_).missations.clear(); miss(t, x_, {{0, sA}}); EXPECT(!check_phylo_tree_integrity(t).empty()); }   // A0T is on the same branch
  { auto t = phylo_tree_tests_complex_tree(); t.at(x_).missations.clear(); miss(t, x_, {{1, sA}}); t.at(c_).missations.clear(); EXPECT(!check_phylo_tree_integrity(t).empty()); }   // A1G on x->b below it
  { auto t = phylo_tree_tests_complex_tree(); t.at(x_).missations.clear(); miss(t, x_, {{2, sA}}); t.at(c_).missations.clear(); miss(t, c_, {{2, sA}}); EXPECT(!check_phylo_tree_integrity(t).empty()); }   // on every branch out of r
}

// sequence_overlay_tests.cpp:17-79, 198-208 -- the semantics the builder relies on (orc_build.hpp's Seq_overlay): reads fall through
// to the base sequence, a write that differs is one delta, a write that restores the base state is none.
TEST(sequence_overlay_reference_semantics) {
  { const std::vector<State> base; Seq_overlay o(base); EXPECT(o.deltas.empty()); EXPECT(o.base == &base); }
  { const std::vector<State> base{sA, sC, sG, sT}; const Seq_overlay o(base);
    EXPECT(o.get(0) == sA); EXPECT(o.get(1) == sC); EXPECT(o.get(2) == sG); EXPECT(o.get(3) == sT); EXPECT(o.deltas.empty()); }
  { const std::vector<State> base{sA, sA}; Seq_overlay o(base); o.set(1, sC);
    EXPECT(o.get(0) == sA); EXPECT(o.get(1) == sC); EXPECT(o.deltas.size() == 1); }
  { const std::vector<State> base{sA, sA}; Seq_overlay o(base); o.set(1, sC); o.set(1, base[1]);
    EXPECT(o.get(0) == sA); EXPECT(o.get(1) == sA); EXPECT(o.deltas.empty()); }
  { const std::vector<State> base{sA, sC, sG, sT}; Seq_overlay o(base); o.set(0, sT); o.set(3, sA);
    std::vector<State> m; for (int l = 0; l < 4; ++l) m.push_back(o.get(l));
    EXPECT(m == (std::vector<State>{sT, sC, sG, sA})); EXPECT(base == (std::vector<State>{sA, sC, sG, sT})); }
}

// tree_tests.cpp:88-162: the orders in which the path's traversals visit a three-node tree (root a = index 2 with children b = 1, c = 0)
// and visit nothing of an empty one.
TEST(tree_traversal_orders_of_the_reference) {
  { Phylo_tree t(0); EXPECT(pre_order(t).empty()); EXPECT(post_order(t).empty()); int visits = 0; traversal(t, [&](Node_index, int) { ++visits; }); EXPECT(visits == 0); }
  Phylo_tree t(3);
  const Node_index c = 0, b = 1, a = 2;
  t.ref_sequence = {sA}; t.root = a;
  set_inner(t, a, k_no_node, b, c, 0.0); set_tip(t, b, a, 1.0); set_tip(t, c, a, 1.0);
  EXPECT(pre_order(t) == (std::vector<Node_index>{a, b, c}));
  EXPECT(post_order(t) == (std::vector<Node_index>{b, c, a}));
  std::vector<std::pair<Node_index, int>> seen;
  traversal(t, [&](Node_index n, int children_so_far) { seen.push_back({n, children_so_far}); });
  EXPECT(seen == (std::vector<std::pair<Node_index, int>>{{a, 0}, {b, 0}, {a, 1}, {c, 0}, {a, 2}}));
}

TEST(build_usher_like_tree_reproduces_its_tip_descriptors) {
  for (int seed = 0; seed < 8; ++seed) {
    emat::SynthParams p; p.num_tips = 12 + 37 * seed; p.num_sites = seed % 2 ? 300 : 2000; p.mu = (seed % 3 ? 6e-4 : 2e-3) / 365.0 * 365.0 / 365.0; p.gaps_per_tip = seed % 4; p.mean_gap_len = 25;
    p.seed = 4100 + seed; if (seed >= 4) { p.tip_date_uncertainty = 6.0; p.frac_uncertain_tips = 0.4; }
    auto R = emat::make_synthetic_emat(p);
    auto src = tree_from_flat(R.tree, R.ref_sequence);
    auto descs = tip_descs_of(src);
    EXPECT((int)descs.size() == p.num_tips);
    Rng rng; rng.key = 77 + seed;
    auto t = build_usher_like_tree(src.ref_sequence, descs, rng);
    EXPECT(t.size() == 2 * p.num_tips - 1);
    auto msg = check_phylo_tree_integrity(t);
    if (!msg.empty()) std::printf("  seed %d integrity: %s\n", seed, msg.c_str());
    EXPECT(msg.empty());
    msg = check_phylo_tree_matches_tip_descs(t, src.ref_sequence, descs);
    if (!msg.empty()) std::printf("  seed %d: %s\n", seed, msg.c_str());
    EXPECT(msg.empty());
    // the builder grafts where the fewest mutations are needed: a loose sanity bound on the mutations it needs against the tree the tips came from (tips with a quarter of their sites missing are placed blindly there and cost later tips extra mutations)
    if (!(calc_num_muts(t) <= 3 * calc_num_muts(src) + 10)) std::printf("  seed %d: %d mutations against %d in the source tree\n", seed, calc_num_muts(t), calc_num_muts(src));
    EXPECT(calc_num_muts(t) <= 3 * calc_num_muts(src) + 10);
    // deterministic given the stream
    Rng rng2; rng2.key = 77 + seed;
    auto t2 = build_usher_like_tree(src.ref_sequence, descs, rng2);
    bool same = t2.root == t.root;
    for (int n = 0; n < t.size() && same; ++n) same = t.at(n).parent == t2.at(n).parent && t.at(n).t == t2.at(n).t && t.at(n).mutations == t2.at(n).mutations && t.at(n).missations == t2.at(n).missations;
    EXPECT(same);
  }
}

#include "orc_utree_tests.hpp"

int main(int argc, char** argv) {
  const char* only = argc > 1 ? argv[1] : nullptr;
  for (auto& t : tests()) {
    if (only && std::string(t.name).find(only) == std::string::npos) continue;
    g_test = t.name;
    int before = g_fail;
    try { t.fn(); } catch (const std::exception& ex) { ++g_fail; std::printf("FAIL [%s] exception: %s\n", t.name, ex.what()); }
    std::printf("%-55s %s\n", t.name, g_fail == before ? "ok" : "FAILED");
  }
  std::printf("%d checks, %d failed\n", g_checks, g_fail);
  return g_fail > 250 ? 250 : g_fail;
}
