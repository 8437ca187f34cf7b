// emat_device_moves.hpp -- the five local moves of reference core/subrun.cpp:98-742 on a part slab.
// NO include guard (see emat_device_core.hpp).

#include "emat_device_spr.hpp"

namespace emat {
namespace EMAT_DEV_NS {

#ifndef EMAT_HOT_BYTES
#define EMAT_HOT_BYTES 1536   // LDS block an SPR1 move sets aside for its candidate scans (sets searched per region, scan items, regions)
#endif
enum { k_inner_node_displace = 0, k_tip_displace = 1, k_branch_reform = 2, k_subtree_slide = 3, k_spr1 = 4 };

EMAT_D void begin_move(Ctx& c, int kind) { hdr_of(c)->proposed[kind]++; c.tr_kind = (double)kind; c.tr_node = -1.0; c.tr_acc = 0.0; c.tr_log_mh = __builtin_nan(""); }
EMAT_D void note_move(Ctx& c, int node, double log_mh, bool acc, int kind) { c.tr_node = (double)node; c.tr_log_mh = log_mh; c.tr_acc = acc ? 1.0 : 0.0; if (acc) hdr_of(c)->accepted[kind]++; }
template <bool kLeaf> EMAT_DF bool mh_accept_t(Ctx& c, double log_mh) { return log_mh >= 0.0 || t_uniform_co<kLeaf>(c, 0.0, 1.0) < t_exp<kLeaf>(log_mh); }
EMAT_D bool mh_accept(Ctx& c, double log_mh) { return mh_accept_t<false>(c, log_mh); }

// distributions.h:38-69
template <bool kLeaf = false> EMAT_DF double bounded_exponential(Ctx& c, double lambda, double a, double b) {
  double u = t_u01_oo<kLeaf>(c);
  double ltr = lambda * (b - a);
  double x;
  if (lambda == 0.0) x = a + u * (b - a);
  else if (lambda > 0 && ltr > 100) x = b + t_log<kLeaf>(u) / lambda;
  else if (lambda < 0 && ltr < -100) x = a + t_log<kLeaf>(u) / lambda;
  else x = a + t_log1p<kLeaf>(u * (t_exp<kLeaf>(ltr) - 1)) / lambda;
  return x < a ? a : (b < x ? b : x);
}
template <bool kLeaf = false> EMAT_DF int pick_random_node(Ctx& c) { return t_uniform_int<kLeaf>(c, hdr_of(c)->n_nodes); }

// subrun.cpp:683-742.  What the move keeps across its calls (the two grafts, the nodes and times it started from) lives in a
// frame of the scratch arena, read back through the context after every call -- like an SPR1 move's Spr1Frame -- instead of
// in registers that every call would make the compiler spill to private memory: with one lane active every private dword
// dirties a 64-byte line of its own.
struct CoreFrame {
  int X, new_branch, P, old_S;
  double new_t, alpha_ratio, old_t_P, d_prior, log_mh;
  Graft old_graft, new_graft;
};
#ifndef EMAT_CF
#define EMAT_CF(c) (*(CoreFrame*)(c).frame)
#endif
EMAT_NOTAIL EMAT_DN void spr_move_core(Ctx& c, int X, int new_branch, double new_t, double alpha_ratio) { EMAT_TIMED(2);
  if (X == hdr_of(c)->root) return;
  if (!c.includes_run_root) if (nodes_of(c)[X].parent == hdr_of(c)->root || new_branch == hdr_of(c)->root) return;
  {
    const double t_X = nodes_of(c)[X].t;
    const int P = nodes_of(c)[X].parent;
    const double new_t_P = new_t;
    if (new_t_P == t_X || new_t_P == nodes_of(c)[new_branch].t || (P != hdr_of(c)->root && new_t_P == nodes_of(c)[nodes_of(c)[P].parent].t)) return;
    if (coal_needs_cells(c, new_t_P)) { stop_for_cells(c, (int)c.tr_kind); return; }   // before the graft is peeled: nothing has changed yet
    CoreFrame* f = (CoreFrame*)sc_alloc(c, (uint32_t)sizeof(CoreFrame));
    if (c.failed) return;
    c.frame = (uint8_t*)f;
    f->X = X; f->new_branch = new_branch; f->P = P; f->old_S = sibling_of(c, P, X);
    f->new_t = new_t; f->alpha_ratio = alpha_ratio; f->old_t_P = nodes_of(c)[P].t;
  }
  c.mu_prop = nodes_of(c)[hdr_of(c)->root].lambda / (c.L - nodes_of(c)[hdr_of(c)->root].n_missing);
  EMAT_PHASE_BEGIN();
  analyze_graft(c, EMAT_CF(c).X, EMAT_CF(c).old_graft);
  peel_graft(c, EMAT_CF(c).old_graft);
  EMAT_PHASE(c, 0);
  spr_move_topology(c, EMAT_CF(c).X, EMAT_CF(c).new_branch, EMAT_CF(c).new_t);
  EMAT_PHASE(c, 1);
  propose_new_graft(c, EMAT_CF(c).X, EMAT_CF(c).new_graft);
  EMAT_PHASE(c, 2);
  if (c.failed) return;
  EMAT_CF(c).d_prior = coal_delta_displace_coalescence(c, EMAT_CF(c).old_t_P, EMAT_CF(c).new_t);
  {
    const CoreFrame& f = EMAT_CF(c);
    EMAT_CF(c).log_mh = (f.new_graft.delta_log_G - f.new_graft.log_alpha_mut) - (f.old_graft.delta_log_G - f.old_graft.log_alpha_mut) + m_log(f.alpha_ratio) + f.d_prior;
  }
  if (c.failed) return;
  const bool acc = mh_accept(c, EMAT_CF(c).log_mh);
  if (c.tr_kind == (double)k_subtree_slide) note_move(c, EMAT_CF(c).X, EMAT_CF(c).log_mh, acc, k_subtree_slide);
  if (acc) {
    apply_graft(c, EMAT_CF(c).new_graft);
    hdr_of(c)->log_G -= EMAT_CF(c).old_graft.delta_log_G; hdr_of(c)->log_G += EMAT_CF(c).new_graft.delta_log_G;
    hdr_of(c)->log_aug_prior += EMAT_CF(c).d_prior;
    coal_coalescence_displaced(c, EMAT_CF(c).old_t_P, EMAT_CF(c).new_t);
  } else {
    spr_move_topology(c, EMAT_CF(c).X, EMAT_CF(c).old_S, EMAT_CF(c).old_t_P);
    apply_graft(c, EMAT_CF(c).old_graft);
  }
  EMAT_PHASE(c, 4);
}

// The three simple moves are compiled twice: kRoot = true for the part that holds the run's root (its root node may be
// displaced, its coalescent grid may grow, branch reforms next to the root go through spr_move_core) and kRoot = false for
// every other part, where none of that can happen: those versions contain no root-only code, and their only calls are the
// out-of-line transcendentals (m_log, m_exp, ...).
// (-DEMAT_SIMPLE_MOVES_INLINE puts them inside run_chain_loop instead, which spares each simple move the whole-wave save
// of the VGPR a non-leaf function parks its return address in -- scripts/micro/wwm.hip: ~490 cycles per call with one lane
// active -- and costs 128 VGPRs and 544 B of scratch instead of 64 and 432: measured +0.3 %, not taken.)
#ifndef EMAT_SIMPLE_MOVE
#ifdef EMAT_SIMPLE_MOVES_INLINE
#define EMAT_SIMPLE_MOVE EMAT_NOTAIL EMAT_DF
#else
#define EMAT_SIMPLE_MOVE EMAT_NOTAIL EMAT_DN
#endif
#endif
// kLeaf (only with kRoot = false): the move as a LEAF function -- transcendentals inlined, draws that cannot call -- which returns true when it ran out of
// numbers computed ahead and must be run again out of line (mcmc_sub_iteration); nothing is committed in that case (the check sits between the last draw,
// mh_accept's, and the first store to the part's state).
template <bool kRoot, bool kLeaf = false> EMAT_SIMPLE_MOVE bool inner_node_displace_move(Ctx& c) { EMAT_TIMED(2);   // subrun.cpp:148-232
  begin_move(c, k_inner_node_displace);
  int node;
  { EMAT_TIMED(2);   /* inner_displace: pick an inner node */
    int guard = 0; do { node = pick_random_node<kLeaf>(c); } while (is_tip(c, node) && !(kLeaf && c.rng_short) && guard++ < (1 << 26)); }
  if (kLeaf && c.rng_short) return true;
  c.tr_node = (double)node;
  const int root = hdr_of(c)->root;
  if (!kRoot && node == root) return kLeaf && c.rng_short;   // node == root && !includes_run_root
  const NodeRec nd = nodes_of(c)[node];
  double t_min = -k_inf;
  if (!kRoot || node != root) {
    t_min = nodes_of(c)[nd.parent].t;
    const MutRec* m = muts_of(c, node);
    for (int i = 0; i < (int)nd.muts.cnt; ++i) t_min = t_min > m[i].t ? t_min : m[i].t;
  }
  double t_max = k_inf;
  const int ch[2] = {nd.child0, nd.child1};
  const double lambda_at_node = nd.lambda;
  double d_logG_dt = 0.0;
  if (!kRoot || node != root) d_logG_dt += -lambda_at_node;
  for (int k = 0; k < 2; ++k) {
    const int cc = ch[k];
    t_max = t_max < nodes_of(c)[cc].t ? t_max : nodes_of(c)[cc].t;
    const MutRec* m = muts_of(c, cc);
    const int nm = nmuts(c, cc);
    for (int i = 0; i < nm; ++i) t_max = t_max < m[i].t ? t_max : m[i].t;
    c.bytes += 64 + 16 * nm + 24 * (int)nodes_of(c)[cc].miss.cnt;
  }
  { EMAT_TIMED(2);   /* inner_displace: delta_lambda_across_node_missations of both children */
  for (int k = 0; k < 2; ++k) {
    double lambda_just_below = lambda_at_node + delta_lambda_across_node_missations(c, ch[k]);
    d_logG_dt -= -lambda_just_below;
  } }
  c.bytes += 2 * 64 + 16 * (int)nd.muts.cnt;
  const double old_t = nd.t;
  double log_alpha_ratio = 0.0, new_t = old_t;
  if (kRoot && node == root) {
    double tree_span = c.t_max_tip - t_max;
    EMAT_CHECK(c, tree_span >= 0.0);
    double ds = (1 / nd.lambda) / 2;
    double delta_scale;   // std::min(ds, tree_span)
    if (tree_span < ds) delta_scale = tree_span; else delta_scale = ds;
    new_t = old_t + gaussian(c, 0.0, delta_scale);
    if (new_t < t_min || new_t > t_max) return false;
    log_alpha_ratio = 0.0;
  } else {
    EMAT_TIMED(2);   /* inner_displace: bounded_exponential */
    new_t = bounded_exponential<kLeaf>(c, d_logG_dt, t_min, t_max);
    log_alpha_ratio = d_logG_dt * (new_t - old_t);
  }
  if (new_t == t_min || new_t == t_max) return kLeaf && c.rng_short;
  if (kRoot) { if (coal_needs_cells(c, new_t)) { stop_for_cells(c, k_inner_node_displace); return false; } }   // nothing has changed yet
  double delta_log_G = d_logG_dt * (new_t - old_t);
  double delta_log_prior = coal_delta_displace_coalescence<kRoot, kLeaf>(c, old_t, new_t);
  if (c.failed) return kLeaf && c.rng_short;
  double log_mh = delta_log_G + delta_log_prior - log_alpha_ratio;
  bool acc;
  { EMAT_TIMED(2);   /* inner_displace: mh_accept + note_move */
  acc = mh_accept_t<kLeaf>(c, log_mh);
  if (kLeaf && c.rng_short) return true;   // (the last draw: nothing of the part has been written yet)
  note_move(c, node, log_mh, acc, k_inner_node_displace); }
  if (acc) {
    EMAT_TIMED(2);   /* inner_displace: accepted: coal_coalescence_displaced + updates */
    coal_coalescence_displaced<kRoot>(c, old_t, new_t);
    nodes_of(c)[node].t = new_t;
    hdr_of(c)->log_G += d_logG_dt * (new_t - old_t);
    hdr_of(c)->log_aug_prior += delta_log_prior;
  }
  return false;
}

template <bool kRoot, bool kLeaf = false> EMAT_SIMPLE_MOVE bool tip_displace_move(Ctx& c) { EMAT_TIMED(2);   // subrun.cpp:234-285
  begin_move(c, k_tip_displace);
  int node;
  { int guard = 0; do { node = pick_random_node<kLeaf>(c); } while (!is_tip(c, node) && !(kLeaf && c.rng_short) && guard++ < (1 << 26)); }
  if (kLeaf && c.rng_short) return true;
  c.tr_node = (double)node;
  const NodeRec nd = nodes_of(c)[node];
  if (nd.t_min == nd.t_max) return kLeaf && c.rng_short;
  double t_min;   // std::max(a, b) = a < b ? b : a
  if ((double)nd.t_min < nodes_of(c)[nd.parent].t) t_min = nodes_of(c)[nd.parent].t; else t_min = (double)nd.t_min;
  const MutRec* m = muts_of(c, node);
  for (int i = 0; i < (int)nd.muts.cnt; ++i) t_min = t_min > m[i].t ? t_min : m[i].t;
  const double t_max = (double)nd.t_max;
  const double d_logG_dt = -nd.lambda;
  const double old_t = nd.t;
  c.bytes += 2 * 64 + 16 * (int)nd.muts.cnt;
  double new_t = bounded_exponential<kLeaf>(c, d_logG_dt, t_min, t_max);
  double log_alpha_ratio = d_logG_dt * (new_t - old_t);
  if (new_t == t_min || new_t == t_max) return kLeaf && c.rng_short;
  double delta_log_G = d_logG_dt * (new_t - old_t);
  double delta_log_prior = coal_delta_displace_tip<kRoot>(c, old_t, new_t);
  if (c.failed) return kLeaf && c.rng_short;
  double log_mh = delta_log_G + delta_log_prior - log_alpha_ratio;
  bool acc = mh_accept_t<kLeaf>(c, log_mh);
  if (kLeaf && c.rng_short) return true;   // (the last draw: nothing of the part has been written yet)
  note_move(c, node, log_mh, acc, k_tip_displace);
  if (acc) {
    coal_tip_displaced<kRoot>(c, old_t, new_t);
    nodes_of(c)[node].t = new_t;
    hdr_of(c)->log_G += d_logG_dt * (new_t - old_t);
    hdr_of(c)->log_aug_prior += delta_log_prior;
  }
  return false;
}

// phylo_tree.cpp:579-644; result in scratch
template <bool kLeaf = false> EMAT_DF SVec<MutRec> randomize_branch_mutation_times(Ctx& c, int X) { EMAT_TIMED(2);
  const int n = nmuts(c, X);
  SVec<MutRec> out = sc_vec<MutRec>(c, n);
  if (c.failed) return out;
  const MutRec* old = muts_of(c, X);
  if (X == hdr_of(c)->root) { for (int i = 0; i < n; ++i) push(c, out, old[i]); return out; }
  const double t_X = nodes_of(c)[X].t, t_P = nodes_of(c)[nodes_of(c)[X].parent].t;
  bool complicated = false;
  for (int i = 0; i < n && !complicated; ++i) for (int j = i + 1; j < n; ++j) if (old[i].site == old[j].site) { complicated = true; break; }
  if (!complicated) {
    for (int i = 0; i < n; ++i) { MutRec r = make_mut(old[i].from, old[i].site, old[i].to, t_uniform_oc<kLeaf>(c, t_P, t_X)); r.pad = (uint16_t)i; push(c, out, r); }   // pad: where it came from (reform_factors)
  } else {
    // distinct sites in ascending order; per site, fresh sorted times assigned in the old order
    int prev_site = -1;
    while (true) {
      int l = 0x7fffffff;
      for (int i = 0; i < n; ++i) if (old[i].site > prev_site && old[i].site < l) l = old[i].site;
      if (l == 0x7fffffff) break;
      prev_site = l;
      int first = out.n, k = 0;
      for (int i = 0; i < n; ++i) if (old[i].site == l) { MutRec r = make_mut(old[i].from, old[i].site, old[i].to, t_uniform_oc<kLeaf>(c, t_P, t_X)); r.pad = (uint16_t)i; push(c, out, r); ++k; }
      // sort just the times of this site's block
      for (int a = first + 1; a < first + k && a < out.n; ++a) { double x = out.p[a].t; int b = a - 1; while (b >= first && out.p[b].t > x) { out.p[b + 1].t = out.p[b].t; --b; } out.p[b + 1].t = x; }
    }
  }
  sort_muts(out.p, out.n);
  return out;
}

// What a mutation contributes to a branch's log G besides its time (phylo_tree_calc.h:185-206): A = mu nu (q_from - q_to), the
// factor of (t - t_P), and B = log(mu nu q_from,to).  A branch reform evaluates the reference's sum twice -- over the branch's
// mutations as they are and as re-timed -- and both lists hold the SAME mutations (randomize_branch_mutation_times only draws
// new times): the factors are gathered once, all per-site loads of the branch in one round trip (they used to be one L2 round trip
// and one logarithm per mutation and list, in sequence), and each sum then runs in its own order over the same operands -- bit
// for bit the reference's two numbers.  A re-timed mutation carries the index of the one it came from in its `pad` field.
struct ReformFactors { double* A; double* B; };
template <bool kLeaf = false> EMAT_DF ReformFactors reform_factors(Ctx& c, const MutRec* m, int n) { EMAT_TIMED(2);
  ReformFactors f; f.A = (double*)sc_alloc(c, (uint32_t)n * 16u); f.B = f.A + n;
  if (c.failed) return f;
  uint32_t need_log = 0u; bool all_logs = false;   // which entries still hold the logarithm's argument (bit j for j < 32; beyond: decided again per entry)
  for (int j0 = 0; j0 < n; j0 += 4) {
    int l[4]; int pa[4]; double nu[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) l[k] = m[j0 + k < n ? j0 + k : n - 1].site;
#pragma unroll
    for (int k = 0; k < 4; ++k) { pa[k] = site_part(c, l[k]); nu[k] = site_nu(c, l[k]); }     // eight independent loads in flight
#pragma unroll
    for (int k = 0; k < 4; ++k) if (j0 + k < n) {
      const MutRec& mm = m[j0 + k];
      const double mn = mu_of(c)[pa[k]] * nu[k];
      const double* q = q_of(c) + pa[k] * 16;
      f.A[j0 + k] = mn * (-q[(int)mm.from * 5] - -q[(int)mm.to * 5]);      // mu nu (q_a(from) - q_a(to)), q_a = -q_aa
      if (c.have_logq && nu[k] == 1.0) f.B[j0 + k] = emat_lds_logq[pa[k] * 16 + (int)mm.from * 4 + (int)mm.to];
      else { f.B[j0 + k] = mn * q[(int)mm.from * 4 + (int)mm.to]; need_log |= 1u << ((j0 + k) & 31); if (j0 + k >= 32) all_logs = true; }   // the logarithm's argument; taken below
    }
  }
  if (all_logs) { for (int j = 0; j < n; ++j) if (!(c.have_logq && site_nu(c, m[j].site) == 1.0)) f.B[j] = t_log<kLeaf>(f.B[j]); }
  else if (need_log != 0u) { const int n32 = n < 32 ? n : 32; for (int j = 0; j < n32; ++j) if ((need_log >> j) & 1u) f.B[j] = t_log<kLeaf>(f.B[j]); }   // entries from 32 on need none (all_logs would be set): never shift by >= 32
  return f;
}
template <bool kRoot, bool kLeaf = false> EMAT_SIMPLE_MOVE bool branch_reform_move(Ctx& c) { EMAT_TIMED(2);   // subrun.cpp:287-320
  begin_move(c, k_branch_reform);
  if (hdr_of(c)->n_nodes < 3) return false;
  const int X = pick_random_node<kLeaf>(c);
  c.tr_node = (double)X;
  if (kLeaf && c.rng_short) return true;
  if (X == hdr_of(c)->root) return false;
  const int P = nodes_of(c)[X].parent;
  const int S = sibling_of(c, P, X);
  const double t_X = nodes_of(c)[X].t, t_P = nodes_of(c)[P].t;
  // in a part without the run's root, spr_move_core returns at once for a branch next to the subroot (subrun.cpp:689-697)
  if (kRoot) { if (P == hdr_of(c)->root) { spr_move_core(c, X, S, t_P, 1.0); if (c.failed) return false; } }
  const double lam = nodes_of(c)[X].lambda;
  const int n = nmuts(c, X);
  if (n == 0) {
    // nothing to re-time (six branches in ten at C4): the reference's two branch_log_G are one and the same number, no random
    // number is drawn, and the empty list replaces itself
    const double g = -lam * (t_X - t_P), delta_log_G = g - g;
    c.bytes += 2 * 64;
    const bool acc = mh_accept_t<kLeaf>(c, delta_log_G);
    if (kLeaf && c.rng_short) return true;
    note_move(c, X, delta_log_G, acc, k_branch_reform);
    if (acc) hdr_of(c)->log_G += delta_log_G;
    return false;
  }
#ifdef EMAT_X_NO_FACTORS
  SVec<MutRec> nm = randomize_branch_mutation_times<kLeaf>(c, X);
  if (c.failed) return kLeaf && c.rng_short;
  const double delta_log_G = branch_log_G(c, t_P, t_X, lam, nm.p, nm.n) - branch_log_G(c, t_P, t_X, lam, muts_of(c, X), nmuts(c, X));
#else
  const ReformFactors f = reform_factors<kLeaf>(c, muts_of(c, X), n);
  SVec<MutRec> nm = randomize_branch_mutation_times<kLeaf>(c, X);
  if (c.failed) return kLeaf && c.rng_short;
  double g_new = -lam * (t_X - t_P), g_old = g_new;
  { const MutRec* m = nm.p; for (int i = nm.n - 1; i >= 0; --i) { const int j = (int)m[i].pad; g_new -= f.A[j] * (m[i].t - t_P); g_new += f.B[j]; } }
  { const MutRec* m = muts_of(c, X); for (int i = n - 1; i >= 0; --i) { g_old -= f.A[i] * (m[i].t - t_P); g_old += f.B[i]; } }
  const double delta_log_G = g_new - g_old;
#endif
  c.bytes += 2 * 64 + 2 * 16 * nm.n;
  double log_mh = delta_log_G;
  bool acc = mh_accept_t<kLeaf>(c, log_mh);
  if (kLeaf && c.rng_short) return true;   // (the last draw: nothing of the part has been written yet)
  note_move(c, X, log_mh, acc, k_branch_reform);
  if (acc) {
    for (int i = 0; i < nm.n; ++i) nm.p[i].pad = 0;
    list_assign<MutRec>(c, nodes_of(c)[X].muts, nm.p, nm.n); hdr_of(c)->log_G += delta_log_G; c.bytes += 16 * nm.n; c.bytes_w += 16 * nm.n;
  }
  return false;
}

// subrun.cpp:325-350, iterative with an explicit stack in scratch
EMAT_DN SVec<int> enumerate_descendant_branches_straddling(Ctx& c, int P, double t, int X) {
  SVec<int> out; out.n = 0;
  // results grow up from the bottom of a scratch span, the DFS stack grows down from its top
  ScSpan span = sc_span(c, 512);
  out.p = (int*)span.lo; out.cap = 0;
  int* stack_base = (int*)span.hi;
  int sp = 0;
  auto room = [&](int er, int es) -> bool {
    if (span.lo + (size_t)(out.n + er) * 4 + 16 <= span.hi - (size_t)(sp + es) * 4) return true;
    if (!span.lds) return false;
    ScSpan big = sc_span_hbm(c);   // outgrew the LDS arena: migrate
    if (big.lo + (size_t)(out.n + er) * 4 + 16 > big.hi - (size_t)(sp + es) * 4) return false;
    int* no = (int*)big.lo; int* nb = (int*)big.hi;
    for (int i = 0; i < out.n; ++i) no[i] = out.p[i];
    for (int i = 1; i <= sp; ++i) nb[-i] = stack_base[-i];
    span = big; out.p = no; stack_base = nb;
    return true;
  };
  if (!room(0, 1)) { EMAT_FAIL(c, k_part_overflow); return out; }
  stack_base[-(++sp)] = P;
  while (sp > 0 && !c.failed) {
    int n = stack_base[-sp]; --sp;
    if (n == X) continue;
    if (t <= nodes_of(c)[n].t) { if (!room(1, 0)) { EMAT_FAIL(c, k_part_overflow); break; } out.p[out.n++] = n; }
    else if (!is_tip(c, n)) {
      if (!room(0, 2)) { EMAT_FAIL(c, k_part_overflow); break; }
      stack_base[-(++sp)] = nodes_of(c)[n].child1;   // child0 is processed first, as in the recursive original
      stack_base[-(++sp)] = nodes_of(c)[n].child0;
    }
  }
  out.cap = out.n;
  sc_span_commit(c, span, (uint32_t)out.n * 4u);
  return out;
}

EMAT_NOTAIL EMAT_DN void subtree_slide_move(Ctx& c) { EMAT_TIMED(2);   // subrun.cpp:352-448
  begin_move(c, k_subtree_slide);
  if (hdr_of(c)->n_nodes < 2) return;
  const int X = pick_random_node(c);
  c.tr_node = (double)X;
  const int root = hdr_of(c)->root;
  if (X == root) return;
  const int P = nodes_of(c)[X].parent, S = sibling_of(c, P, X);
  double t_early = (P == root) ? (nodes_of(c)[X].t < nodes_of(c)[S].t ? nodes_of(c)[X].t : nodes_of(c)[S].t) : nodes_of(c)[root].t;
  if (P == root) { if (nodes_of(c)[S].t < nodes_of(c)[X].t) t_early = nodes_of(c)[S].t; else t_early = nodes_of(c)[X].t; }   // std::min(tX, tS)
  double tree_span = c.t_max_tip - t_early;
  EMAT_CHECK(c, tree_span >= 0.0);
  double ds = (1 / nodes_of(c)[X].lambda) / 2;
  double delta_scale = tree_span < ds ? tree_span : ds;
  double delta_t = gaussian(c, 0.0, delta_scale);
  const double old_P_t = nodes_of(c)[P].t, new_P_t = old_P_t + delta_t;
  if (delta_t < 0.0) {
    if (P != root && new_P_t < nodes_of(c)[nodes_of(c)[P].parent].t) {
      int GG = nodes_of(c)[P].parent, SS = P;
      while (new_P_t < nodes_of(c)[GG].t) { SS = GG; GG = nodes_of(c)[GG].parent; if (GG == k_no_node) break; }
      SVec<int> branches = enumerate_descendant_branches_straddling(c, SS, old_P_t, X);
      if (c.failed) return;
      double a_o2n = 1.0, a_n2o = 1.0 / (double)branches.n;
      spr_move_core(c, X, SS, new_P_t, a_n2o / a_o2n);
    } else spr_move_core(c, X, S, new_P_t, 1.0);
  } else {
    if (new_P_t > nodes_of(c)[X].t) return;
    if (new_P_t > nodes_of(c)[S].t) {
      SVec<int> branches = enumerate_descendant_branches_straddling(c, P, new_P_t, X);
      if (c.failed) return;
      if (branches.n == 0) return;
      int bi = uniform_int(c, branches.n);
      int SS = branches.p[bi];
      double a_o2n = 1.0 / (double)branches.n, a_n2o = 1.0;
      spr_move_core(c, X, SS, new_P_t, a_n2o / a_o2n);
    } else spr_move_core(c, X, S, new_P_t, 1.0);
  }
}

// subrun.cpp:492-675 in three stretches on lane 0; between them the whole wave scans for candidate regions and weighs them
// (wave_scan_and_study).  A stretch that ends the move clears c.phase; one that parks it sets c.phase and c.svc.
EMAT_D void spr1_park(Ctx& c, int phase) { c.phase = (uint8_t)phase; c.svc = 1; }
EMAT_NOTAIL EMAT_DN void spr1_move_begin(Ctx& c) { EMAT_TIMED(2);
  begin_move(c, k_spr1);
  c.phase = 0;
  if (hdr_of(c)->n_nodes < 2) return;
  const double chooser = uniform_co(c, 0.0, 1.0);
#ifdef EMAT_X_NO_UNLIMITED_SCANS   // timing experiment only (a different chain): what the 1 % of scans without a limit cost the slowest chains
  const int limit = 1; (void)chooser;
#else
  const int limit = chooser < 0.01 ? 0x7fffffff : 1;
#endif
  const int root0 = hdr_of(c)->root;
  c.mu_prop = nodes_of(c)[root0].lambda / (c.L - nodes_of(c)[root0].n_missing);
  int X;
  { int guard = 0; do { X = pick_random_node(c); } while (hdr_of(c)->root == X && guard++ < (1 << 26)); }
  c.tr_node = (double)X;
  if (nodes_of(c)[X].lambda == 0.0) return;
  const int P = nodes_of(c)[X].parent;
  const bool pruning_changes_root = P == hdr_of(c)->root;
  if (pruning_changes_root && !c.includes_run_root) return;
  EMAT_PHASE_BEGIN();
  Spr1Frame* frp = (Spr1Frame*)sc_alloc(c, (uint32_t)sizeof(Spr1Frame));
  if (c.failed) return;
  Spr1Frame& fr = *frp;
  c.frame = (uint8_t*)frp;
  fr.X = X; fr.t_X = nodes_of(c)[X].t; fr.P = P; fr.old_t_P = nodes_of(c)[P].t; fr.old_S = sibling_of(c, P, X); fr.old_G = nodes_of(c)[P].parent;
  fr.limit = limit; fr.f = 0.8 /* annealing factor */; fr.t_max_tip = c.t_max_tip; fr.can_change_root = c.includes_run_root;
  // a part that leaves much of its staging area free (the large parts of a side class, whose scans run to hundreds of items)
  // sets aside half of what is free instead of the fixed block
  {
    const uint32_t a0 = (c.a_top + 15u) & ~15u, free_b = c.a_end > a0 ? c.a_end - a0 : 0u;
    uint32_t hot_b = EMAT_HOT_BYTES;
    if (free_b / 2 > hot_b) hot_b = (free_b / 2 < 32768u ? free_b / 2 : 32768u) & ~15u;
    fr.hot = (limit == 1) ? sc_reserve_hot(c, hot_b) : HotBlock{nullptr, 0};
  }
  analyze_graft(c, X, fr.old_graft);
  peel_graft(c, fr.old_graft);
  EMAT_PHASE(c, 5);
  if (c.failed) return;
  fr.old_min_muts = count_min_mutations(c, fr.old_graft);
  int extra = 4;
  if (limit != 1) { extra = 8; for (int n = 0; n < hdr_of(c)->n_nodes; ++n) if (c.includes_run_root || n != hdr_of(c)->root) extra += nmuts(c, n); }
  fr.extra = extra;
  fr.deltas = summarize_closed_mutations(c, fr.old_graft, extra);
  fr.missing_at_X = reconstruct_missing_sites_at(c, X);
  fr.n_missing_at_X = iv_num_sites(fr.missing_at_X.p, fr.missing_at_X.n);
  fr.lambda_X = nodes_of(c)[X].lambda;
  fr.init_branch = fr.old_S;
  if (c.failed) return;
  spr1_park(c, 1);   // -> scan from the old sibling, study; resumes in spr1_move_propose
}
EMAT_NOTAIL EMAT_DN void spr1_move_propose(Ctx& c) { EMAT_TIMED(2);
  Spr1Frame& fr = *(Spr1Frame*)c.frame;
  c.phase = 0;
  if (c.failed) return;
  EMAT_PHASE_BEGIN();
  const Study& pre = fr.study;
  const int X = fr.X, P = fr.P;
  int new_region; double new_t_P;
  { EMAT_TIMED(2);   /* spr1_propose: pick region, pick time, log_alpha_in_region */
  new_region = study_pick_nexus_region(c, pre);
  new_t_P = study_pick_time_in_region(c, pre, new_region);
  fr.log_alpha_o2n = study_log_alpha_in_region(c, pre, new_region, new_t_P); }
  const int new_S = pre.regions.p[new_region].branch;
  EMAT_CHECK(c, new_S != P);
  fr.pre_new_region_min_muts = pre.regions.p[new_region].min_muts;   // the second scan reuses the regions' storage
  fr.new_S = new_S; fr.new_t_P = new_t_P;
  const double t_new_S = nodes_of(c)[new_S].t;
  int new_G = nodes_of(c)[new_S].parent;
  if (new_G == P) new_G = fr.old_G;
  const double t_new_G = (new_G == k_no_node) ? k_neg_dbl_max : nodes_of(c)[new_G].t;
  if (c.failed) return;
  EMAT_PHASE(c, 7);
  if (new_t_P == fr.t_X || new_t_P == t_new_S || new_t_P == t_new_G) { apply_graft(c, fr.old_graft); return; }
  if (coal_needs_cells(c, new_t_P)) { apply_graft(c, fr.old_graft); if (!c.failed) stop_for_cells(c, k_spr1); return; }   // the old graft goes back on, as after a rejection
  spr_move_topology(c, X, new_S, new_t_P);
  EMAT_PHASE(c, 8);
  propose_new_graft(c, X, fr.new_graft);
  EMAT_PHASE(c, 9);
  if (c.failed) return;
  EMAT_CHECK(c, nodes_of(c)[X].parent == P);
  fr.new_min_muts = count_min_mutations(c, fr.new_graft);
  fr.deltas = summarize_closed_mutations(c, fr.new_graft, fr.extra);
  fr.init_branch = new_S;
  if (c.failed) return;
  spr1_park(c, 2);   // -> scan from the new sibling, study; resumes in spr1_move_finish
}
EMAT_NOTAIL EMAT_DN void spr1_move_finish(Ctx& c) { EMAT_TIMED(2);
  Spr1Frame& fr = *(Spr1Frame*)c.frame;
  c.phase = 0;
  if (c.failed) return;
  EMAT_PHASE_BEGIN();
  const Study& post = fr.study;
  const int X = fr.X;
  const int old_region = study_find_region(post, fr.old_S, fr.old_t_P);
  EMAT_CHECK(c, old_region != -1);
  if (c.failed) return;
  const double log_alpha_n2o = study_log_alpha_in_region(c, post, old_region, fr.old_t_P);
  EMAT_CHECK(c, fr.new_min_muts == fr.pre_new_region_min_muts);
  EMAT_CHECK(c, fr.old_min_muts == post.regions.p[old_region].min_muts);
  EMAT_PHASE(c, 11);
  const double d_prior = coal_delta_displace_coalescence(c, fr.old_t_P, fr.new_t_P);
  if (c.failed) return;
  const double log_mh = (fr.new_graft.delta_log_G - fr.new_graft.log_alpha_mut) - (fr.old_graft.delta_log_G - fr.old_graft.log_alpha_mut)
      + log_alpha_n2o - fr.log_alpha_o2n + d_prior;
  const bool acc = mh_accept(c, log_mh);
  note_move(c, X, log_mh, acc, k_spr1);
  if (acc) {
    apply_graft(c, fr.new_graft);
    hdr_of(c)->log_G -= fr.old_graft.delta_log_G; hdr_of(c)->log_G += fr.new_graft.delta_log_G;
    hdr_of(c)->log_aug_prior += d_prior;
    coal_coalescence_displaced(c, fr.old_t_P, fr.new_t_P);
  } else {
    EMAT_TIMED(2);   /* spr1_finish: rejected: move back + apply the old graft */
    spr_move_topology(c, X, fr.old_S, fr.old_t_P);
    apply_graft(c, fr.old_graft);
  }
  EMAT_PHASE(c, 12);
}

// ---- slab housekeeping ----------------------------------------------------------------------------------
// Squeeze the garbage out of the list heap by staging all live lists in the (idle) scratch region.
EMAT_DN bool compact_heap(Ctx& c) {
  uint32_t live = 0;
  const int n = hdr_of(c)->n_nodes;
  for (int i = 0; i < n; ++i) {
    live += (((uint32_t)nodes_of(c)[i].muts.cnt * 16u) + 15u) & ~15u;
    live += (((uint32_t)nodes_of(c)[i].miss.cnt * 8u) + 15u) & ~15u;
    live += (((uint32_t)nodes_of(c)[i].mfs.cnt * 8u) + 15u) & ~15u;
  }
  if (live > hdr_of(c)->scratch_end - hdr_of(c)->scratch_begin) return false;
  uint8_t* stage = c.G + hdr_of(c)->scratch_begin;
  uint32_t w = 0;
  for (int i = 0; i < n; ++i) {
    ListRef* refs[3] = {&nodes_of(c)[i].muts, &nodes_of(c)[i].miss, &nodes_of(c)[i].mfs};
    const uint32_t es[3] = {16u, 8u, 8u};
    for (int k = 0; k < 3; ++k) {
      uint32_t bytes = (uint32_t)refs[k]->cnt * es[k];
      const uint64_t* src = (const uint64_t*)heap_at(c, refs[k]->off); uint64_t* dst = (uint64_t*)(stage + w);
      for (uint32_t q = 0; q < bytes / 8; ++q) dst[q] = src[q];
      w += (bytes + 15u) & ~15u;
    }
  }
  uint32_t top = hdr_of(c)->heap_begin, r = 0;
  for (int i = 0; i < n; ++i) {
    ListRef* refs[3] = {&nodes_of(c)[i].muts, &nodes_of(c)[i].miss, &nodes_of(c)[i].mfs};
    const uint32_t es[3] = {16u, 8u, 8u};
    for (int k = 0; k < 3; ++k) {
      uint32_t bytes = (uint32_t)refs[k]->cnt * es[k], padded = (bytes + 15u) & ~15u;
      const uint64_t* src = (const uint64_t*)(stage + r); uint64_t* dst = (uint64_t*)heap_at(c, top);
      for (uint32_t q = 0; q < bytes / 8; ++q) dst[q] = src[q];
      refs[k]->off = top; refs[k]->cap = list_cap_for(padded, es[k]);
      top += padded; r += padded;
    }
  }
  hdr_of(c)->heap_top = top;
  return true;
}

// Subrun::mcmc_sub_iteration (subrun.cpp:98-121).  Returns false when the part must stop.  An SPR1 move parks itself
// twice for the wave's scan + study (c.svc != 0 on return): the next call resumes it.
EMAT_NOTAIL EMAT_D bool mcmc_sub_iteration(Ctx& c) {
#ifdef EMAT_PROFILE_PHASES
  const long long _mv0 = clock64();
#endif
  if (c.phase == 1) spr1_move_propose(c);
  else if (c.phase == 2) spr1_move_finish(c);
  else {
  // space check BEFORE the move, so that a stop leaves a consistent state
  {
    uint32_t heap_size = hdr_of(c)->heap_end - hdr_of(c)->heap_begin, free_b = hdr_of(c)->heap_end - hdr_of(c)->heap_top;
    uint32_t reserve = heap_size / 4 > 1024u ? heap_size / 4 : 1024u;
    if (reserve > heap_size / 2) reserve = heap_size / 2;
    if (free_b < reserve) {
      if (!compact_heap(c) || hdr_of(c)->heap_end - hdr_of(c)->heap_top < reserve) { if (hdr_of(c)->status == 0) hdr_of(c)->status = k_part_need_space; return false; }
    }
  }
  sc_reset(c);
  c.mv_rng_ctr = c.rng_ctr; c.mv_rng_had_spare = c.rng_has_spare;
  c.tr_kind = -1.0; c.tr_node = -1.0; c.tr_acc = 0.0; c.tr_log_mh = __builtin_nan("");
  // -DEMAT_LEAF_SIMPLE=1: a part without the run's root runs its simple moves as LEAF functions (transcendentals inlined, draws that cannot call: no return address
  // to park, no whole-wave save and reload around every move); one that ran out of numbers computed ahead has committed nothing: its counters are put back, the
  // stream is rewound to the move's first draw, and the same move runs out of line.  Built and measured in round 6 -- the leaf versions contain no call and no
  // scratch access, all trajectory-parity tests pass on them -- and worth NOTHING: 476.4 against 476.7 M moves/s, HBM writes unchanged: off.
#ifndef EMAT_LEAF_SIMPLE
#define EMAT_LEAF_SIMPLE 0
#endif
  const bool leaf = EMAT_LEAF_SIMPLE && !c.includes_run_root;
  const int64_t bytes0 = c.bytes, bytes_w0 = c.bytes_w;
  auto redo = [&](int kind) { hdr_of(c)->proposed[kind]--; c.bytes = bytes0; c.bytes_w = bytes_w0; c.rng_short = false; rng_rewind_to_move_start(c); };
  if (c.only_displacing_inner_nodes) {
    if (c.includes_run_root) inner_node_displace_move<true>(c);
    else if (!leaf || inner_node_displace_move<false, true>(c)) { if (leaf) redo(k_inner_node_displace); inner_node_displace_move<false>(c); }
  } else {
    double total_weight = 15.0 + 15.0;
    if (c.topology_moves_enabled) total_weight += 1.0 + 1.0;
    double r = uniform_co(c, 0.0, total_weight);
    if (r < 7.5) {
      if (c.includes_run_root) inner_node_displace_move<true>(c);
      else if (!leaf || inner_node_displace_move<false, true>(c)) { if (leaf) { redo(k_inner_node_displace); (void)uniform_co(c, 0.0, total_weight); } inner_node_displace_move<false>(c); }
    } else if (r < 15.0) {
      if (c.includes_run_root) tip_displace_move<true>(c);
      else if (!leaf || tip_displace_move<false, true>(c)) { if (leaf) { redo(k_tip_displace); (void)uniform_co(c, 0.0, total_weight); } tip_displace_move<false>(c); }
    } else if (r < 30.0) {
      if (c.includes_run_root) branch_reform_move<true>(c);
      else if (!leaf || branch_reform_move<false, true>(c)) { if (leaf) { redo(k_branch_reform); (void)uniform_co(c, 0.0, total_weight); } branch_reform_move<false>(c); }
    }
    else if (c.topology_moves_enabled) { if (r < 31.0) subtree_slide_move(c); else spr1_move_begin(c); }
  }
  }
#ifdef EMAT_PROFILE_PHASES
  { const long long _dt = clock64() - _mv0; hdr_of(c)->phase_ticks[(c.tr_kind >= 3.0) ? 15 : 14] += _dt; if (c.tr_kind == 0.0) EMAT_COUNT(c, 10, _dt); else if (c.tr_kind == 1.0) EMAT_COUNT(c, 15, _dt); }   // (inner-node / tip displacements apart: reserved[10], [15])
#endif
  if (c.svc != 0 && !c.failed) return true;   // parked: the move is not over
  c.phase = 0; c.svc = 0;
  if (hdr_of(c)->status == k_part_need_cells) return false;   // stopped before the move changed anything (stop_for_cells): not a move
  if (hdr_of(c)->trace_len < hdr_of(c)->trace_cap) {
    double* tr = (double*)slab_at(c, hdr_of(c)->off_trace) + 4 * hdr_of(c)->trace_len;
    tr[0] = c.tr_kind; tr[1] = c.tr_node; tr[2] = c.tr_acc; tr[3] = c.tr_log_mh;
    hdr_of(c)->trace_len++;
  }
  hdr_of(c)->moves_done++;
  return !c.failed;
}

// The chain itself: `c.moves_left` sub-iterations.  Its own function, and its counter in the context, so that nothing is
// live in registers across a move: the moves clobber every register (no callee-saved saves, see the Makefile), and
// whatever their caller kept in registers would be spilled and reloaded around each of them.
// Returns with c.svc != 0 when the current move waits for the wave (the kernel serves it and calls again), else when the
// moves are done or the part had to stop.
EMAT_NOTAIL EMAT_DN void run_chain_loop(Ctx& c) {
  c.svc = 0;
  for (;;) {
    if (c.phase == 0) {
      if (c.moves_left <= 0) return;
      if (rng_wants_fill(c)) { c.svc = 2; return; }                 // the wave computes the next stretch of the stream (rng_fill), then calls again
      c.moves_left -= 1;
    }
    if (!mcmc_sub_iteration(c)) { c.moves_left = 0; c.phase = 0; c.svc = 0; return; }
    if (c.svc != 0) return;
  }
}

// ---- derived quantities of one part from scratch (Subrun::recalc_derived_quantities, subrun.cpp:17-26;
//      calc_lambda_i phylo_tree_calc.cpp:420-436; calc_num_sites_missing_at_every_node :67-76;
//      calc_log_G_below_root :515-543; calc_log_root_prior :467-504; calc_partial_log_prior
//      very_scalable_coalescent.cpp:355-386).  `ref_freqs` = state counts of the reference sequence per site
//      partition [P][4] (Subrun::state_frequencies_of_ref_sequence_per_partition_).  Executed by ONE lane;
//      the wave-parallel version lives in emat_kernels.hip. --------------------------------------------------
EMAT_DN double calc_log_root_prior(Ctx& c, const int32_t* ref_freqs, int P) {
  const int root = hdr_of(c)->root;
  // counts are adjusted on the fly instead of copying the table
  double result = 0.0;
  for (int p = 0; p < P; ++p) {
    for (int a = 0; a < 4; ++a) {
      int f = ref_freqs[p * 4 + a];
      const MutRec* m = muts_of(c, root);
      for (int i = 0; i < nmuts(c, root); ++i) if (site_part(c, m[i].site) == p) { if (m[i].from == a) --f; if (m[i].to == a) ++f; }
      const IvRec* iv = miss_of(c, root);
      for (int i = 0; i < (int)nodes_of(c)[root].miss.cnt; ++i) for (int l = iv[i].start; l < iv[i].end; ++l) if (site_part(c, l) == p && (int)c.ref[l] == a) --f;
      const FsRec* fs = mfs_of(c, root);
      for (int i = 0; i < (int)nodes_of(c)[root].mfs.cnt; ++i) if (site_part(c, fs[i].site) == p) { if ((int)c.ref[fs[i].site] == a) ++f; if ((int)fs[i].state == a) --f; }
      double pa = pi_of(c)[p * 4 + a];
      if (pa != 0.0) result += f * m_log(pa);
      else if (f != 0) return -k_inf;
    }
  }
  return result;
}

}  // namespace EMAT_DEV_NS
}  // namespace emat
