// emat_device_spr.hpp -- device-side SPR machinery of the EMAT local-move engine (gfx950).
//
// Covers SURVEY section 8(a) rows a10-a16: site-delta tracking, the candidate-region scan
// (reference core/spr_study.cpp:9-549), graft analysis / peel / apply for inner and rooty grafts
// (core/spr_move.cpp:91-1069), the topological move through elementary tree edits
// (core/spr_move.cpp:1101-1156, core/tree_editing.cpp:7-302) and JC69 mutational-history sampling
// (core/spr_move.cpp:1158-1439).  All temporaries live in the part's scratch region; all node lists
// live in the slab heap.  Unordered maps of the reference become site-sorted arrays (every
// order-sensitive use in the reference is followed by a (t, site) sort).
// NO include guard (see emat_device_core.hpp).

#include "emat_device_core.hpp"

namespace emat {
namespace EMAT_DEV_NS {

// ---- scratch missation map ("sliding_missations", spr_move.cpp:615-698) ------------------------------------
EMAT_DN SVec<FsRec> fsv_set_v(Ctx& c, SVec<FsRec> v, int l, int from) {   // Missation_map::set_from_state on a scratch list (header by value: see sd_push_front_v)
  int k = fs_lower_bound(v.p, v.n, l);
  bool present = (k < v.n && v.p[k].site == l);
  if (from != (int)c.ref[l]) {
    if (present) { v.p[k].state = (uint8_t)from; return v; }
    if (v.n >= v.cap) { EMAT_FAIL(c, k_part_overflow); return v; }
    for (int i = v.n; i > k; --i) v.p[i] = v.p[i - 1];
    v.p[k].site = l; v.p[k].state = (uint8_t)from; v.p[k].pad[0] = v.p[k].pad[1] = v.p[k].pad[2] = 0; v.n++;
  } else if (present) { for (int i = k; i + 1 < v.n; ++i) v.p[i] = v.p[i + 1]; v.n--; }
  return v;
}
EMAT_DF void fsv_set(Ctx& c, SVec<FsRec>& v, int l, int from) { v = fsv_set_v(c, v, l, from); }

// Rebuild the root's "mutations" (ref -> root deltas, t = -DBL_MAX) from a delta list, in site order.
EMAT_DN void set_root_muts_from_deltas(Ctx& c, int root, const SVec<SdRec> d) { EMAT_TIMED(1);
  ListRef& r = nodes_of(c)[root].muts;
  list_reserve<MutRec>(c, r, d.n);
  if (c.failed) return;
  MutRec* m = list_ptr<MutRec>(c, r);
  for (int i = 0; i < d.n; ++i) { EMAT_CHECK(c, c.ref[d.p[i].site] == d.p[i].from); m[i] = make_mut(d.p[i].from, d.p[i].site, d.p[i].to, k_neg_dbl_max); }
  set_list_cnt(c, r, d.n);
}
EMAT_DN SVec<SdRec> deltas_from_root_muts(Ctx& c, int root, int extra_cap) { EMAT_TIMED(1);
  SVec<SdRec> d = sc_vec<SdRec>(c, nmuts(c, root) + extra_cap + 1);
  const MutRec* m = muts_of(c, root);
  for (int i = 0; i < nmuts(c, root); ++i) sd_push_back(c, d, m[i].site, m[i].from, m[i].to);
  return d;
}
// Upper bound on the mutations met on the way from `from_node` to the root.  In a part that does not hold
// the run's root, the subroot's list (ref -> subroot deltas, possibly hundreds of entries) is frozen and never
// crossed by a move, so it is not counted.
EMAT_D int path_mut_count(Ctx& c, int from_node) {
  int s = 0;
  for (int cur = from_node; cur != k_no_node; cur = nodes_of(c)[cur].parent) if (c.includes_run_root || nodes_of(c)[cur].parent != k_no_node) s += nmuts(c, cur);
  return s;
}

// =================================================================================================
// Tree editing (tree_editing.cpp:7-302)
// =================================================================================================
struct Edit { int X; SVec<SdRec> deltas; };

// Node missations <- merge(node, other) (mutations.h:315-336)
EMAT_DN void node_merge_missations_from(Ctx& c, int dst, int other) {
  ScMark mark = sc_mark(c);
  int na = (int)nodes_of(c)[dst].miss.cnt, nb = (int)nodes_of(c)[other].miss.cnt;
  SVec<IvRec> iv = sc_vec<IvRec>(c, na + nb + 1);
  int fa = (int)nodes_of(c)[dst].mfs.cnt, fb = (int)nodes_of(c)[other].mfs.cnt;
  SVec<FsRec> fs = sc_vec<FsRec>(c, fa + fb + 1);
  if (!c.failed) {
    iv.n = iv_merge(iv.p, miss_of(c, dst), na, miss_of(c, other), nb);
    const FsRec* A = mfs_of(c, dst); const FsRec* B = mfs_of(c, other);
    int i = 0, j = 0;
    while (i < fa || j < fb) {
      if (j == fb || (i < fa && A[i].site < B[j].site)) fs.p[fs.n++] = A[i++];
      else if (i == fa || B[j].site < A[i].site) fs.p[fs.n++] = B[j++];
      else { fs.p[fs.n++] = A[i++]; ++j; }   // map::insert keeps the first
    }
    list_assign<IvRec>(c, nodes_of(c)[dst].miss, iv.p, iv.n);
    list_assign<FsRec>(c, nodes_of(c)[dst].mfs, fs.p, fs.n);
  }
  sc_release(c, mark);
}
// factor_out_common_missations(A, B, common) (mutations.h:250-312); `common` node's lists are replaced
EMAT_DN void node_factor_out_common(Ctx& c, int a, int b, int common) {
  ScMark mark = sc_mark(c);
  int na = (int)nodes_of(c)[a].miss.cnt, nb = (int)nodes_of(c)[b].miss.cnt;
  SVec<IvRec> ic = sc_vec<IvRec>(c, na + nb + 1), ia = sc_vec<IvRec>(c, 2 * (na + nb) + 2), ib = sc_vec<IvRec>(c, 2 * (na + nb) + 2);
  int fa = (int)nodes_of(c)[a].mfs.cnt, fb = (int)nodes_of(c)[b].mfs.cnt;
  SVec<FsRec> fca = sc_vec<FsRec>(c, fa + 1), fcb = sc_vec<FsRec>(c, fb + 1), fcc = sc_vec<FsRec>(c, fa + 1);
  if (!c.failed) {
    ic.n = iv_intersect(ic.p, miss_of(c, a), na, miss_of(c, b), nb);
    ia.n = iv_subtract(ia.p, miss_of(c, a), na, ic.p, ic.n);
    ib.n = iv_subtract(ib.p, miss_of(c, b), nb, ic.p, ic.n);
    const FsRec* A = mfs_of(c, a); const FsRec* B = mfs_of(c, b);
    int i = 0, j = 0;
    while (i < fa && j < fb) {
      if (A[i].site < B[j].site) fca.p[fca.n++] = A[i++];
      else if (B[j].site < A[i].site) fcb.p[fcb.n++] = B[j++];
      else { fcc.p[fcc.n++] = A[i++]; ++j; }
    }
    while (i < fa) fca.p[fca.n++] = A[i++];
    while (j < fb) fcb.p[fcb.n++] = B[j++];
    list_assign<IvRec>(c, nodes_of(c)[a].miss, ia.p, ia.n); list_assign<FsRec>(c, nodes_of(c)[a].mfs, fca.p, fca.n);
    list_assign<IvRec>(c, nodes_of(c)[b].miss, ib.p, ib.n); list_assign<FsRec>(c, nodes_of(c)[b].mfs, fcb.p, fcb.n);
    list_assign<IvRec>(c, nodes_of(c)[common].miss, ic.p, ic.n); list_assign<FsRec>(c, nodes_of(c)[common].mfs, fcc.p, fcc.n);
  }
  sc_release(c, mark);
}

EMAT_DN Edit edit_slide_root_v(Ctx& c, Edit e, double new_t_P) {   // tree_editing.cpp:114-158 (the session by value: see sd_push_front_v)
  const int X = e.X, P = nodes_of(c)[X].parent;
  EMAT_CHECK(c, P == hdr_of(c)->root);
  double old_t_P = nodes_of(c)[P].t;
  int S = sibling_of(c, P, X);
  if (new_t_P > old_t_P) {
    MutRec* mS = muts_of(c, S); int nS = nmuts(c, S);
    int last = 0; while (last < nS && !(mS[last].t > new_t_P)) ++last;
    if (last != 0) {
      ScMark mark = sc_mark(c);
      SVec<SdRec> r2r = deltas_from_root_muts(c, P, last);
      for (int i = 0; i < last && !c.failed; ++i) {
        MutRec m = mS[i];
        sd_push_back(c, r2r, m.site, m.from, m.to);
        if (!miss_contains(c, X, m.site)) sd_push_front(c, e.deltas, m.site, m.to, m.from);
        else miss_set_from_state(c, X, m.site, m.to);
        nodes_of(c)[P].lambda += dq(c, m.site, m.from, m.to);
      }
      set_root_muts_from_deltas(c, P, r2r);
      list_erase_prefix<MutRec>(c, nodes_of(c)[S].muts, last);
      sc_release(c, mark);
    }
  }
  nodes_of(c)[P].t = new_t_P;
  return e;
}
EMAT_DF void edit_slide_root(Ctx& c, Edit& e, double new_t_P) { e = edit_slide_root_v(c, e, new_t_P); }
EMAT_FN_SLIDE Edit edit_slide_P_along_branch_v(Ctx& c, Edit e, double new_t_P) {   // tree_editing.cpp:31-112
  const int X = e.X, P = nodes_of(c)[X].parent;
  EMAT_CHECK(c, !is_tip(c, P));
  if (P == hdr_of(c)->root) return edit_slide_root_v(c, e, new_t_P);
  double old_t_P = nodes_of(c)[P].t;
  int S = sibling_of(c, P, X);
  if (new_t_P < old_t_P) {
    MutRec* mP = muts_of(c, P); int nP = nmuts(c, P);
    int first = 0; while (first < nP && !(mP[first].t >= new_t_P)) ++first;
    if (first != nP) {
      // the tail of G-P moves (in time order) to the FRONT of P-S
      int k = nP - first;
      int kept = 0;
      for (int i = first; i < nP; ++i) if (!miss_contains(c, S, mP[i].site)) ++kept;
      int nS = nmuts(c, S);
      list_reserve<MutRec>(c, nodes_of(c)[S].muts, nS + kept);
      if (c.failed) return e;
      MutRec* mS = muts_of(c, S); mP = muts_of(c, P);
      for (int i = nS - 1; i >= 0; --i) mS[i + kept] = mS[i];
      int w = kept;
      for (int i = nP - 1; i >= first && !c.failed; --i) {
        MutRec m = mP[i];
        if (!miss_contains(c, S, m.site)) { mS[--w] = m; }
        else miss_set_from_state(c, S, m.site, m.from);
        if (!miss_contains(c, X, m.site)) sd_push_front(c, e.deltas, m.site, m.from, m.to);
        else miss_set_from_state(c, X, m.site, m.from);
        nodes_of(c)[P].lambda += dq(c, m.site, m.to, m.from);
      }
      set_list_cnt(c, nodes_of(c)[S].muts, nS + kept);
      set_list_cnt(c, nodes_of(c)[P].muts, first);
      (void)k;
    }
  } else {
    MutRec* mS = muts_of(c, S); int nS = nmuts(c, S);
    int last = 0; while (last < nS && !(mS[last].t > new_t_P)) ++last;
    if (last != 0) {
      list_reserve<MutRec>(c, nodes_of(c)[P].muts, nmuts(c, P) + last);
      if (c.failed) return e;
      mS = muts_of(c, S);
      for (int i = 0; i < last && !c.failed; ++i) {
        MutRec m = mS[i];
        MutRec* mP = muts_of(c, P); mP[nodes_of(c)[P].muts.cnt] = m; nodes_of(c)[P].muts.cnt++;
        if (!miss_contains(c, X, m.site)) sd_push_front(c, e.deltas, m.site, m.to, m.from);
        else miss_set_from_state(c, X, m.site, m.to);
        nodes_of(c)[P].lambda += dq(c, m.site, m.from, m.to);
      }
      list_erase_prefix<MutRec>(c, nodes_of(c)[S].muts, last);
    }
  }
  nodes_of(c)[P].t = new_t_P;
  return e;
}
EMAT_DF void edit_slide_P_along_branch(Ctx& c, Edit& e, double new_t_P) { e = edit_slide_P_along_branch_v(c, e, new_t_P); }
EMAT_FN_HOP void edit_do_hop_up(Ctx& c, int X) {   // tree_editing.cpp:164-231
  EMAT_CHECK(c, X != hdr_of(c)->root);
  const int P = nodes_of(c)[X].parent;
  EMAT_CHECK(c, !is_tip(c, P) && P != hdr_of(c)->root && nmuts(c, P) == 0);
  const int G = nodes_of(c)[P].parent;
  if (c.failed || G == k_no_node) { EMAT_FAIL(c, k_part_internal); return; }
  EMAT_CHECK(c, nodes_of(c)[P].t == nodes_of(c)[G].t);
  const int U = sibling_of(c, G, P);
  const int S = sibling_of(c, P, X);
  if (nodes_of(c)[P].miss.cnt != 0) {
    node_merge_missations_from(c, X, P);
    node_merge_missations_from(c, S, P);
    nodes_of(c)[P].miss.cnt = 0; nodes_of(c)[P].mfs.cnt = 0;
  }
  swap_lists(nodes_of(c)[P].muts, nodes_of(c)[G].muts);
  swap_lists(nodes_of(c)[P].miss, nodes_of(c)[G].miss);
  swap_lists(nodes_of(c)[P].mfs, nodes_of(c)[G].mfs);
  EMAT_CHECK(c, nodes_of(c)[G].miss.cnt == 0);
  if (iv_intersects(miss_of(c, S), (int)nodes_of(c)[S].miss.cnt, miss_of(c, U), (int)nodes_of(c)[U].miss.cnt)) node_factor_out_common(c, S, U, G);
  if (G == hdr_of(c)->root) { hdr_of(c)->root = P; nodes_of(c)[P].parent = k_no_node; }
  else {
    int GG = nodes_of(c)[G].parent, GU = sibling_of(c, GG, G);
    nodes_of(c)[GG].child0 = P; nodes_of(c)[GG].child1 = GU;
    nodes_of(c)[P].parent = GG;
  }
  nodes_of(c)[P].child0 = X; nodes_of(c)[P].child1 = G;
  nodes_of(c)[G].parent = P;
  nodes_of(c)[G].child0 = S; nodes_of(c)[G].child1 = U;
  nodes_of(c)[S].parent = G;
  nodes_of(c)[P].lambda = nodes_of(c)[G].lambda;
  nodes_of(c)[P].n_missing = nodes_of(c)[G].n_missing;
  nodes_of(c)[G].lambda = nodes_of(c)[P].lambda + delta_lambda_across_node_missations(c, G);
  nodes_of(c)[G].n_missing = nodes_of(c)[P].n_missing + iv_num_sites(miss_of(c, G), (int)nodes_of(c)[G].miss.cnt);
  c.bytes += 5 * 64; c.bytes_w += 160;   // (five node records re-linked: counted once, half of it as written)
}
EMAT_DN void edit_flip(Ctx& c, const Edit e) {   // tree_editing.cpp:233-278 (reads e.X only)
  const int X = e.X, P = nodes_of(c)[X].parent;
  EMAT_CHECK(c, !is_tip(c, P) && P != hdr_of(c)->root && nmuts(c, P) == 0);
  const int G = nodes_of(c)[P].parent;
  if (c.failed || G == k_no_node) { EMAT_FAIL(c, k_part_internal); return; }
  EMAT_CHECK(c, nodes_of(c)[P].t == nodes_of(c)[G].t);
  const int U = sibling_of(c, G, P);
  const int S = sibling_of(c, P, X);
  if (nodes_of(c)[P].miss.cnt != 0) {
    node_merge_missations_from(c, S, P);
    node_merge_missations_from(c, X, P);
    nodes_of(c)[P].miss.cnt = 0; nodes_of(c)[P].mfs.cnt = 0;
  }
  if (iv_intersects(miss_of(c, X), (int)nodes_of(c)[X].miss.cnt, miss_of(c, U), (int)nodes_of(c)[U].miss.cnt)) node_factor_out_common(c, X, U, P);
  nodes_of(c)[G].child0 = S; nodes_of(c)[G].child1 = P;
  nodes_of(c)[S].parent = G;
  nodes_of(c)[P].child0 = X; nodes_of(c)[P].child1 = U;
  nodes_of(c)[U].parent = P;
  nodes_of(c)[P].lambda = nodes_of(c)[G].lambda + delta_lambda_across_node_missations(c, P);
  nodes_of(c)[P].n_missing = nodes_of(c)[G].n_missing + iv_num_sites(miss_of(c, P), (int)nodes_of(c)[P].miss.cnt);
  c.bytes += 5 * 64; c.bytes_w += 160;   // (five node records re-linked: counted once, half of it as written)
}
EMAT_DF void edit_hop_down(Ctx& c, const Edit& e, int SS) {   // tree_editing.cpp:280-292
  const int P = nodes_of(c)[e.X].parent;
  EMAT_CHECK(c, SS != hdr_of(c)->root);
  const int U = nodes_of(c)[SS].parent;
  EMAT_CHECK(c, nodes_of(c)[U].parent == P && nmuts(c, U) == 0);
  if (c.failed) return;
  edit_do_hop_up(c, sibling_of(c, U, SS));
}
// Tree_editing_session's constructor (tree_editing.cpp:7-29): the mutations on P-X become the session's deltas (room for `cap`)
EMAT_DF void edit_begin(Ctx& c, Edit& e, int X, int cap) {
  e.X = X;
  e.deltas = sc_vec<SdRec>(c, cap);
  const MutRec* m = muts_of(c, X);
  for (int i = 0; i < nmuts(c, X); ++i) sd_push_back(c, e.deltas, m[i].site, m[i].from, m[i].to);
  nodes_of(c)[X].muts.cnt = 0;
}
// Tree_editing_session::end (tree_editing.cpp:294-302): the deltas go back on P-X as mutations at its midpoint
EMAT_DF void edit_end(Ctx& c, Edit& e) {
  if (c.failed) return;
  const int X = e.X;
  EMAT_CHECK(c, nmuts(c, X) == 0);
  if (e.deltas.n != 0) {
    double mut_t = 0.5 * (nodes_of(c)[X].t + nodes_of(c)[nodes_of(c)[X].parent].t);
    list_reserve<MutRec>(c, nodes_of(c)[X].muts, e.deltas.n);
    if (!c.failed) {
      MutRec* m = muts_of(c, X);
      for (int i = 0; i < e.deltas.n; ++i) m[i] = make_mut(e.deltas.p[i].from, e.deltas.p[i].site, e.deltas.p[i].to, mut_t);
      set_list_cnt(c, nodes_of(c)[X].muts, e.deltas.n);
    }
  }
}
// spr_move.cpp:1101-1156
EMAT_FN_TOPO void spr_move_topology(Ctx& c, int X, int SS, double new_t_P) { EMAT_TIMED(1);
  if (c.failed) return;
  EMAT_CHECK(c, X != hdr_of(c)->root);
  const int P = nodes_of(c)[X].parent, G = nodes_of(c)[P].parent, S = sibling_of(c, P, X);
  if (SS == P) SS = S;
  int GG = nodes_of(c)[SS].parent;
  if (GG == P) GG = G;
  const int A = find_MRCA_of(c, G, GG);
  ScMark mark = sc_mark(c);
  Edit e;
  edit_begin(c, e, X, nmuts(c, X) + path_mut_count(c, P) + path_mut_count(c, SS) + 8);
  int guard = 0;
  while (!c.failed && nodes_of(c)[P].parent != A && guard++ < (1 << 24)) {
    edit_slide_P_along_branch(c, e, nodes_of(c)[nodes_of(c)[P].parent].t);
    if (c.failed) break;
    edit_do_hop_up(c, X);
  }
  if (!c.failed && !(descends_from(c, S, SS) || descends_from(c, SS, S))) {
    EMAT_CHECK(c, A != k_no_node);
    if (!c.failed) { edit_slide_P_along_branch(c, e, nodes_of(c)[A].t); if (!c.failed) edit_flip(c, e); }
  }
  if (!c.failed) {
    // branches from SS up to (excluding) X's current sibling, walked top-down
    const int Xs_sib = sibling_of(c, P, X);
    int depth = 0;
    for (int cur = SS; cur != Xs_sib && cur != k_no_node; cur = nodes_of(c)[cur].parent) ++depth;
    for (int d = depth - 1; d >= 0 && !c.failed; --d) {
      int Y = SS; for (int k = 0; k < d; ++k) Y = nodes_of(c)[Y].parent;
      edit_slide_P_along_branch(c, e, nodes_of(c)[nodes_of(c)[Y].parent].t);
      if (c.failed) break;
      edit_hop_down(c, e, Y);
    }
  }
  if (!c.failed) {
    EMAT_CHECK(c, sibling_of(c, P, X) == SS);
    edit_slide_P_along_branch(c, e, new_t_P);
  }
  edit_end(c, e);
  sc_release(c, mark);
}

// =================================================================================================
// Mutational-history sampling (spr_move.cpp:1158-1439)
// =================================================================================================
EMAT_D int choose_different_state(Ctx& c, int s) { int delta = 1 + uniform_int(c, 3); return (s + delta) % 4; }

struct KTruncPoisson { double lambda; int min_k; double normalization, term_before_min_k, max_k; };   // distributions.h:77-175
EMAT_DF KTruncPoisson ktp_make(double lambda, int min_k) {   // (inlined into its two callers: a 40-byte struct returned by an out-of-line function travels through private memory)
  KTruncPoisson d; d.lambda = lambda; d.min_k = min_k; d.normalization = 0.0; d.term_before_min_k = 0.0; d.max_k = 0.0;
  if ((double)min_k <= lambda) return d;
  d.max_k = (10.0 * min_k > 10.0 * lambda) ? 10.0 * min_k : 10.0 * lambda;
  double last_term = 1.0, em1 = m_expm1(lambda);
  d.normalization = em1;
  for (int k = 1; k < min_k; ++k) { last_term *= lambda / k; d.normalization -= last_term; }
  d.term_before_min_k = last_term;
  if (d.normalization <= 0.0 || fabs(d.normalization) < 1e-10 * em1) {
    d.normalization = 0.0;
    double nlt = last_term;
    for (int k = min_k; k < d.max_k; ++k) { nlt *= lambda / k; d.normalization += nlt; }
  }
  return d;
}
// (the distribution's five numbers as scalar arguments: a struct of 40 bytes passed by value goes through the caller's private frame all the same)
// (Inlined into the trajectory sampler, and that into the history sampler, since round 6: three rejection rounds per constrained site, each a call of this and
// each call a whole-wave save and its reload -- out of line they cost 1.2 % of a pass)
EMAT_FN_KTP int ktp_sample_s(Ctx& c, double lambda, int min_k, double normalization, double term_before_min_k, double max_k) {
  KTruncPoisson d; d.lambda = lambda; d.min_k = min_k; d.normalization = normalization; d.term_before_min_k = term_before_min_k; d.max_k = max_k;
  if (d.normalization == 0.0) { int guard = 0; while (guard++ < (1 << 26)) { int k = poisson(c, d.lambda); if (k >= d.min_k) return k; } return d.min_k; }
  double u = uniform_co(c, 0.0, d.normalization);
  double cum = 0.0; int k = d.min_k; double term = d.term_before_min_k;
  while (k < d.max_k) { term *= d.lambda / k; cum += term; if (cum > u) break; ++k; }
  return k;
}
EMAT_DF int ktp_sample(Ctx& c, const KTruncPoisson& d) { return ktp_sample_s(c, d.lambda, d.min_k, d.normalization, d.term_before_min_k, d.max_k); }

// Appends to `out` (open-ended scratch vector).  States of a trajectory are drawn first (rejection on the end state), then
// its times, exactly as spr_move.cpp:1181-1227.  Both are staged IN PLACE, in the records they end up in (`to` and `t` of
// out.p[out.n ..]); a vector that started in the LDS arena and runs out of room moves to the part's HBM scratch.
constexpr int k_open_max = 1 << 20;   // most elements an open vector of a sampler takes of an arena (the arena's free space bounds it first)
EMAT_DF bool open_room(Ctx& c, SVec<MutRec>& out, int extra) {
  if (out.cap - out.n >= extra) return true;
  if (sc_open_migrate(c, out, k_open_max) && out.cap - out.n >= extra) return true;
  EMAT_FAIL(c, k_part_overflow); return false;
}
// (The vector's header by value, in and out, and the distribution as scalars: see sd_push_front_v.  The trajectory was accepted exactly when the
// vector came back longer: a trajectory has at least min_k >= 1 mutations.)
EMAT_FN_SST SVec<MutRec> sample_site_trajectory_v(Ctx& c, SVec<MutRec> out, int l, int from, int to, double d_lambda, int d_min_k, double d_normalization, double d_term_before_min_k, double d_max_k,
                                              double T, bool accept_only_if_match) {
  int n = 0; int s = from;
  int guard = 0;
  while (guard++ < (1 << 26)) { EMAT_TIMED(1);   /* site_trajectory: one rejection round (ktp_sample + states) */
    { EMAT_TIMED(1);   /* site_trajectory: ktp_sample_s */
    n = ktp_sample_s(c, d_lambda, d_min_k, d_normalization, d_term_before_min_k, d_max_k); }
    if (!open_room(c, out, n)) return out;
    MutRec* rec = out.p + out.n;
    s = from;
    for (int i = 0; i < n; ++i) { s = choose_different_state(c, s); rec[i].to = (uint8_t)s; }
    if (s == to) break;
    if (!accept_only_if_match) return out;   // caller restarts from scratch on its own terms
  }
  MutRec* rec = out.p + out.n;
  EMAT_TIMED(1);   /* site_trajectory: times, sort, records */
  for (int i = 0; i < n; ++i) rec[i].t = uniform_co(c, -T, 0.0);
  for (int i = 1; i < n; ++i) { double x = rec[i].t; int j = i - 1; while (j >= 0 && rec[j].t > x) { rec[j + 1].t = rec[j].t; --j; } rec[j + 1].t = x; }   // the times alone: the states keep their order
  int prev = from;
  for (int i = 0; i < n; ++i) { const int st = rec[i].to; rec[i] = make_mut((uint8_t)prev, l, (uint8_t)st, rec[i].t); prev = st; }
  out.n += n;
  return out;
}
EMAT_DF void sample_site_trajectory(Ctx& c, SVec<MutRec>& out, int l, int from, int to, const KTruncPoisson& dist, double T, bool accept_only_if_match, bool& accepted) {
  const SVec<MutRec> r = sample_site_trajectory_v(c, out, l, from, to, dist.lambda, dist.min_k, dist.normalization, dist.term_before_min_k, dist.max_k, T, accept_only_if_match);
  accepted = r.n > out.n;
  out = r;
}
// spr_move.cpp:1164-1370; result appended into a fresh open-ended scratch vector (caller trims)
EMAT_FN_SMH SVec<MutRec> sample_mutational_history(Ctx& c, int L, double T, double mu, const SVec<SdRec>& deltas) { EMAT_TIMED(1);
  // the LDS arena if it has room for the constrained sites (one mutation each, rarely three) and the L (mu T)^2 / 2 other sites
  // expected to be hit twice or more on a long branch; should more turn up, the vector moves to HBM (open_room)
  const double twice = 0.5 * (double)L * (mu * T) * (mu * T);
  const int want = 2 * deltas.n + 8 + (twice < 1e6 ? (int)(3.0 * twice + 6.0 * sqrt(3.0 * twice)) : (1 << 20));
  SVec<MutRec> out = sc_open<MutRec>(c, k_open_max, want);
  if (c.failed) return out;
#ifdef EMAT_X_TRIVIAL_CONSTRAINED   // timing experiment only (a different chain, parity gone): every constrained site gets ONE mutation at the branch's midpoint without a draw --
  // what the pass would take if the per-site rejection sampling cost nothing: the bound on what per-site streams spread over the wavefront could buy
  if (deltas.n != 0 && open_room(c, out, deltas.n)) { for (int i = 0; i < deltas.n; ++i) out.p[out.n++] = make_mut(deltas.p[i].from, deltas.p[i].site, deltas.p[i].to, -0.5 * T); }
#else
  if (deltas.n != 0) { EMAT_TIMED(1);   /* sample_history: constrained sites (ktp_make + one trajectory per delta) */
    KTruncPoisson ge1 = ktp_make(mu * T, 1);
    for (int i = 0; i < deltas.n && !c.failed; ++i) { EMAT_TIMED(1);   /* sample_history: ONE constrained site */
      bool acc; sample_site_trajectory(c, out, deltas.p[i].site, deltas.p[i].from, deltas.p[i].to, ge1, T, true, acc); }
  }
#endif
  double muT = mu * T;
  int l = 0;
  if ((double)L * muT * muT < 2e-6) l = L;
  // (the skip rate is only worked out when some site will be looked at: an exponential and a log1p that most calls never use)
  double log_one_minus_p_tricky = 0.0;
  if (l < L) { const double p_0 = m_exp(-muT), p_1 = muT * p_0; log_one_minus_p_tricky = (muT < 1e-4) ? -0.5 * muT * muT : -muT - m_log1p(-p_1); }
  int guard = 0;
  EMAT_TIMED_BLOCK(1, skip_timer);   /* sample_history: geometric skipping over the other sites */
  while (l < L && !c.failed && guard++ < (1 << 26)) {
    double u = exponential(c, -log_one_minus_p_tricky);
    if (!(u >= 0 && u < (double)L)) break;
    l += (int)floor(u);
    if (l >= L) break;
    if (sd_contains(deltas, l)) { ++l; continue; }
    KTruncPoisson ge2 = ktp_make(mu * T, 2);
    bool acc = false;
    sample_site_trajectory(c, out, l, 0, 0, ge2, T, false, acc);
    if (acc) ++l;
  }
  EMAT_TIMED_END(skip_timer);
  sort_muts(out.p, out.n);
  sc_trim(c, out);
  return out;
}
// spr_move.cpp:1372-1407
EMAT_DN SVec<MutRec> sample_unconstrained_mutational_history(Ctx& c, int L, double T, double mu) { EMAT_TIMED(1);
  // about mu L T mutations: the LDS arena if it has room for that many and their spread (else, or if more turn up, HBM scratch)
  const double expect = mu * (double)L * T;
  const int want = expect < 1e6 ? (int)(expect + 6.0 * sqrt(expect)) + 8 : (1 << 20);
  SVec<MutRec> out = sc_open<MutRec>(c, k_open_max, want);
  if (c.failed) return out;
  double t = 0.0;
  int guard = 0;
  while (!c.failed && guard++ < (1 << 26)) {
    t -= exponential(c, mu * (double)L);
    if (t <= -T) break;
    int l = uniform_int(c, L);
    // current state of site l going right-to-left = `from` of the most recently generated mutation on l, else A
    int s = 0;
    for (int i = out.n - 1; i >= 0; --i) if (out.p[i].site == l) { s = out.p[i].from; break; }
    int ns = choose_different_state(c, s);
    if (out.n == out.cap && !sc_open_migrate(c, out, k_open_max)) { EMAT_FAIL(c, k_part_overflow); break; }
    push(c, out, make_mut((uint8_t)ns, l, (uint8_t)s, t));
  }
  for (int i = 0, j = out.n - 1; i < j; ++i, --j) { MutRec tmp = out.p[i]; out.p[i] = out.p[j]; out.p[j] = tmp; }
  sc_trim(c, out);
  return out;
}
// spr_move.cpp:1409-1439
EMAT_FN_ADJ void adjust_mutational_history(Ctx& c, const SVec<MutRec> h, const SVec<SdRec> deltas, int end_branch, double end_t) { EMAT_TIMED(1);   // (headers by value: only the records change)
  for (int i = h.n - 1; i >= 0; --i) {
    MutRec& m = h.p[i];
    m.t += end_t;
    if (!sd_contains(deltas, m.site)) {
      int end_state = calc_site_state_at(c, end_branch, end_t, m.site);   // pure function of (tree, site): memoisation not needed
      int delta = end_state;
      m.from = (uint8_t)((m.from + delta) % 4);
      m.to = (uint8_t)((m.to + delta) % 4);
    }
  }
}

// =================================================================================================
// Graft analysis (spr_move.h:28-84, spr_move.cpp:91-1099)
// =================================================================================================
struct BranchInfo {
  int A, B; bool is_open; double T_to_X, pl_A, pl_X;
  SVec<IvRec> warm, hot;
  SVec<MutRec> hot_muts;
  SVec<SdRec> hot_deltas;
};
struct Graft { int X, S; double t_P; BranchInfo* bi; int nbi; double delta_log_G, log_alpha_mut; bool rooty; };
enum { k_PX = 0, k_PS = 1, k_SPX = 2 };

EMAT_D void bi_init(BranchInfo& b) { b.A = b.B = k_no_node; b.is_open = false; b.T_to_X = b.pl_A = b.pl_X = 0.0; b.warm.p = nullptr; b.warm.n = b.warm.cap = 0; b.hot = b.warm; b.hot_muts.p = nullptr; b.hot_muts.n = b.hot_muts.cap = 0; b.hot_deltas.p = nullptr; b.hot_deltas.n = b.hot_deltas.cap = 0; }
EMAT_D double log_alpha_mut_term(double mu_p, int L, double T, int M, bool is_open, int d) {   // spr_move.cpp:296-315, 809-835
  // (Two of the logarithms are multiplied by a count that is usually zero -- no hot mutation, no hot delta -- and a finite logarithm times
  // zero is a zero of the logarithm's sign, which is what is added instead.  The calls are skipped exactly then; an argument whose logarithm is not finite takes the reference's path.)
  const double a3 = mu_p / 3;
  double r = -mu_p * L * T + ((M != 0 || !(a3 > 0.0 && a3 < k_inf)) ? M * m_log(a3) : (a3 < 1.0 ? -0.0 : 0.0));   // (0 x a negative logarithm is -0.0)
  if (!is_open) {
    double P_AC = -0.25 * m_expm1(-4. / 3. * mu_p * T);
    r -= (L - d) * m_log1p(-3 * P_AC) + ((d != 0 || !(P_AC > 0.0 && P_AC < 1.0)) ? d * m_log(P_AC) : -0.0);
  }
  return r;
}

// ---- rooty grafts -----------------------------------------------------------------------------------------
EMAT_DN void start_rooty_graft_analysis(Ctx& c, int X, Graft& g) { EMAT_TIMED(1);   // spr_move.cpp:91-205
  g.X = X; g.rooty = true; g.delta_log_G = g.log_alpha_mut = 0.0;
  const int P = nodes_of(c)[X].parent, S = sibling_of(c, P, X);
  const double t_X = nodes_of(c)[X].t, t_P = nodes_of(c)[P].t, t_S = nodes_of(c)[S].t;
  g.S = S; g.t_P = t_P;
  EMAT_CHECK(c, P == hdr_of(c)->root && c.includes_run_root);
  g.bi = (BranchInfo*)sc_alloc(c, 3 * sizeof(BranchInfo)); g.nbi = 3;
  if (c.failed) { g.nbi = 0; return; }
  for (int i = 0; i < 3; ++i) bi_init(g.bi[i]);
  const MutRec* mX = muts_of(c, X); const int nX = nmuts(c, X);
  const MutRec* mS = muts_of(c, S); const int nS = nmuts(c, S);
  BranchInfo& PX = g.bi[k_PX];
  PX.A = P; PX.B = X; PX.is_open = true; PX.T_to_X = t_X - t_P;
  PX.pl_A = -1 * delta_lambda_across_node_missations(c, S);
  PX.warm = iv_copy_sc(c, miss_of(c, S), (int)nodes_of(c)[S].miss.cnt); PX.hot = PX.warm;
  PX.pl_X = PX.pl_A;
  PX.hot_muts = sc_vec<MutRec>(c, nX);
  for (int i = 0; i < nX; ++i) if (iv_contains(PX.hot.p, PX.hot.n, mX[i].site)) { push(c, PX.hot_muts, mX[i]); PX.pl_X += dq(c, mX[i].site, mX[i].from, mX[i].to); }
  BranchInfo& PS = g.bi[k_PS];
  PS.A = P; PS.B = S; PS.is_open = true; PS.T_to_X = t_S - t_P;
  PS.pl_A = -1 * delta_lambda_across_node_missations(c, X);
  PS.warm = iv_copy_sc(c, miss_of(c, X), (int)nodes_of(c)[X].miss.cnt); PS.hot = PS.warm;
  PS.pl_X = PS.pl_A;
  PS.hot_muts = sc_vec<MutRec>(c, nS);
  for (int i = 0; i < nS; ++i) if (iv_contains(PS.hot.p, PS.hot.n, mS[i].site)) { push(c, PS.hot_muts, mS[i]); PS.pl_X += dq(c, mS[i].site, mS[i].from, mS[i].to); }
  BranchInfo& SPX = g.bi[k_SPX];
  SPX.A = S; SPX.B = P; SPX.is_open = false; SPX.T_to_X = (t_S - t_P) + (t_X - t_P);
  SPX.pl_X = nodes_of(c)[X].lambda - PX.pl_X;
  SPX.pl_A = nodes_of(c)[S].lambda - PS.pl_X;
  {
    IvRec all; all.start = 0; all.end = c.L;
    SVec<IvRec> s1 = iv_subtract_sc(c, &all, 1, miss_of(c, P), (int)nodes_of(c)[P].miss.cnt);
    SVec<IvRec> s2 = iv_subtract_sc(c, s1.p, s1.n, miss_of(c, X), (int)nodes_of(c)[X].miss.cnt);
    SVec<IvRec> s3 = iv_subtract_sc(c, s2.p, s2.n, miss_of(c, S), (int)nodes_of(c)[S].miss.cnt);
    SPX.warm = s3; SPX.hot = s3;
  }
  SPX.hot_muts = sc_vec<MutRec>(c, nS + nX);
  SPX.hot_deltas = sc_vec<SdRec>(c, nS + nX);
  for (int i = nS - 1; i >= 0; --i) if (iv_contains(SPX.hot.p, SPX.hot.n, mS[i].site)) {
    MutRec rm = make_mut(mS[i].to, mS[i].site, mS[i].from, t_P - (mS[i].t - t_P));
    push(c, SPX.hot_muts, rm); sd_push_back(c, SPX.hot_deltas, rm.site, rm.from, rm.to);
  }
  for (int i = 0; i < nX; ++i) if (iv_contains(SPX.hot.p, SPX.hot.n, mX[i].site)) { push(c, SPX.hot_muts, mX[i]); sd_push_back(c, SPX.hot_deltas, mX[i].site, mX[i].from, mX[i].to); }
  c.bytes += 3 * 64 + 16 * (nX + nS) + 24 * ((int)nodes_of(c)[X].miss.cnt + (int)nodes_of(c)[S].miss.cnt + (int)nodes_of(c)[P].miss.cnt);
  return;
}
EMAT_D void filter_not_hot(SVec<MutRec>& v, const SVec<IvRec>& hot) { int w = 0; for (int i = 0; i < v.n; ++i) if (iv_contains(hot.p, hot.n, v.p[i].site)) v.p[w++] = v.p[i]; v.n = w; }
EMAT_D SVec<MutRec> sample_history_for(Ctx& c, const BranchInfo& bi) {
  return bi.is_open ? sample_unconstrained_mutational_history(c, c.L, bi.T_to_X, c.mu_prop)
                    : sample_mutational_history(c, c.L, bi.T_to_X, c.mu_prop, bi.hot_deltas);
}
EMAT_D void recompute_open_pl_A(Ctx& c, BranchInfo& bi) {   // spr_move.cpp:234-241, 776-783
  if (!bi.is_open) return;
  bi.pl_A = bi.pl_X;
  for (int i = bi.hot_muts.n - 1; i >= 0; --i) bi.pl_A += dq(c, bi.hot_muts.p[i].site, bi.hot_muts.p[i].to, bi.hot_muts.p[i].from);
}
EMAT_DN void propose_new_rooty_graft_mutations(Ctx& c, Graft& g) { EMAT_TIMED(1);   // spr_move.cpp:207-244
  const int X = g.X, P = nodes_of(c)[X].parent, S = sibling_of(c, P, X);
  for (int idx = 0; idx < g.nbi && !c.failed; ++idx) {
    BranchInfo& bi = g.bi[idx];
    EMAT_CHECK(c, !bi.is_open || bi.hot_deltas.n == 0);
    if (bi.hot.n != 0) {
      SVec<MutRec> nm = sample_history_for(c, bi);
      if (nm.n != 0) {
        filter_not_hot(nm, bi.hot);
        int end_branch = (idx == k_PS) ? S : X;
        adjust_mutational_history(c, nm, bi.hot_deltas, end_branch, nodes_of(c)[end_branch].t);
      }
      bi.hot_muts = nm;
      recompute_open_pl_A(c, bi);
    }
  }
}
EMAT_D double log_pi_ratio(const Ctx& c, const MutRec& m) { return m_log(pi_a(c, m.site, m.from) / pi_a(c, m.site, m.to)); }
EMAT_DN void finish_rooty_graft_analysis(Ctx& c, Graft& g) { EMAT_TIMED(1);   // spr_move.cpp:246-316
  if (c.failed) return;
  const int X = g.X, P = nodes_of(c)[X].parent, S = sibling_of(c, P, X);
  const double t_X = nodes_of(c)[X].t, t_P = nodes_of(c)[P].t, t_S = nodes_of(c)[S].t;
  BranchInfo& PX = g.bi[k_PX]; BranchInfo& PS = g.bi[k_PS]; BranchInfo& SPX = g.bi[k_SPX];
  g.delta_log_G = 0.0;
  g.delta_log_G += branch_log_G(c, t_P, t_X, PX.pl_X, PX.hot_muts.p, PX.hot_muts.n);
  g.delta_log_G += branch_log_G(c, t_P, t_S, PS.pl_X, PS.hot_muts.p, PS.hot_muts.n);
  ScMark mark = sc_mark(c);
  SVec<MutRec> on_PS = sc_vec<MutRec>(c, SPX.hot_muts.n), on_PX = sc_vec<MutRec>(c, SPX.hot_muts.n);
  for (int i = SPX.hot_muts.n - 1; i >= 0; --i) { const MutRec& m = SPX.hot_muts.p[i]; if (m.t < t_P) push(c, on_PS, make_mut(m.to, m.site, m.from, t_P + (t_P - m.t))); }
  for (int i = 0; i < SPX.hot_muts.n; ++i) if (SPX.hot_muts.p[i].t >= t_P) push(c, on_PX, SPX.hot_muts.p[i]);
  g.delta_log_G += branch_log_G(c, t_P, t_X, SPX.pl_X, on_PX.p, on_PX.n);
  g.delta_log_G += branch_log_G(c, t_P, t_S, SPX.pl_A, on_PS.p, on_PS.n);
  for (int i = 0; i < PX.hot_muts.n; ++i) g.delta_log_G += log_pi_ratio(c, PX.hot_muts.p[i]);
  for (int i = 0; i < PS.hot_muts.n; ++i) g.delta_log_G += log_pi_ratio(c, PS.hot_muts.p[i]);
  for (int i = 0; i < on_PS.n; ++i) g.delta_log_G += log_pi_ratio(c, on_PS.p[i]);
  sc_release(c, mark);
  g.log_alpha_mut = 0.0;
  for (int i = 0; i < g.nbi; ++i) {
    BranchInfo& bi = g.bi[i];
    g.log_alpha_mut += log_alpha_mut_term(c.mu_prop, iv_num_sites(bi.hot.p, bi.hot.n), bi.T_to_X, bi.hot_muts.n, bi.is_open, bi.hot_deltas.n);
  }
}
EMAT_DN void peel_rooty_graft(Ctx& c, const Graft& g) { EMAT_TIMED(1);   // spr_move.cpp:318-431
  if (c.failed) return;
  const int X = g.X, P = nodes_of(c)[X].parent, S = sibling_of(c, P, X);
  const double t_X = nodes_of(c)[X].t, t_P = nodes_of(c)[P].t;
  const BranchInfo& PX = g.bi[k_PX]; const BranchInfo& PS = g.bi[k_PS]; const BranchInfo& SPX = g.bi[k_SPX];
  ScMark mark = sc_mark(c);
  const int nX = nmuts(c, X), nS = nmuts(c, S);
  SVec<SdRec> r2r = deltas_from_root_muts(c, P, nX + 2 * nS);
  const MutRec* mX = muts_of(c, X); const MutRec* mS = muts_of(c, S);
  for (int i = 0; i < nX && !c.failed; ++i) if (iv_contains(PX.hot.p, PX.hot.n, mX[i].site)) { sd_push_back(c, r2r, mX[i].site, mX[i].from, mX[i].to); miss_set_from_state(c, S, mX[i].site, mX[i].to); }
  for (int i = 0; i < nS && !c.failed; ++i) if (iv_contains(PS.hot.p, PS.hot.n, mS[i].site)) { sd_push_back(c, r2r, mS[i].site, mS[i].from, mS[i].to); miss_set_from_state(c, X, mS[i].site, mS[i].to); }
  for (int i = 0; i < nS && !c.failed; ++i) if (iv_contains(SPX.hot.p, SPX.hot.n, mS[i].site)) sd_push_back(c, r2r, mS[i].site, mS[i].from, mS[i].to);
  nodes_of(c)[X].muts.cnt = 0; nodes_of(c)[S].muts.cnt = 0; nodes_of(c)[P].muts.cnt = 0;
  const double t_mut_X = 0.5 * (t_P + t_X);
  list_reserve<MutRec>(c, nodes_of(c)[X].muts, SPX.hot_deltas.n);
  if (!c.failed) {
    MutRec* m = muts_of(c, X);
    for (int i = 0; i < SPX.hot_deltas.n; ++i) m[i] = make_mut(SPX.hot_deltas.p[i].from, SPX.hot_deltas.p[i].site, SPX.hot_deltas.p[i].to, t_mut_X);
    set_list_cnt(c, nodes_of(c)[X].muts, SPX.hot_deltas.n);
  }
  set_root_muts_from_deltas(c, P, r2r);
  nodes_of(c)[P].lambda = calc_lambda_at_node(c, P);
  sc_release(c, mark);
}
EMAT_DN void apply_rooty_graft(Ctx& c, const Graft& g) { EMAT_TIMED(1);   // spr_move.cpp:433-547
  if (c.failed) return;
  const int X = g.X, P = nodes_of(c)[X].parent, S = sibling_of(c, P, X);
  const double t_X = nodes_of(c)[X].t, t_P = nodes_of(c)[P].t, t_S = nodes_of(c)[S].t;
  const BranchInfo& PX = g.bi[k_PX]; const BranchInfo& PS = g.bi[k_PS]; const BranchInfo& SPX = g.bi[k_SPX];
  EMAT_CHECK(c, nmuts(c, S) == 0);
  ScMark mark = sc_mark(c);
  nodes_of(c)[X].muts.cnt = 0;
  SVec<SdRec> r2r = deltas_from_root_muts(c, P, PX.hot_muts.n + PS.hot_muts.n + SPX.hot_muts.n);
  nodes_of(c)[P].muts.cnt = 0;
  for (int i = PX.hot_muts.n - 1; i >= 0 && !c.failed; --i) {
    const MutRec& m = PX.hot_muts.p[i];
    list_push<MutRec>(c, nodes_of(c)[X].muts, m);
    sd_push_back(c, r2r, m.site, m.to, m.from);
    miss_set_from_state(c, S, m.site, m.from);
  }
  for (int i = PS.hot_muts.n - 1; i >= 0 && !c.failed; --i) {
    const MutRec& m = PS.hot_muts.p[i];
    list_push<MutRec>(c, nodes_of(c)[S].muts, m);
    sd_push_back(c, r2r, m.site, m.to, m.from);
    miss_set_from_state(c, X, m.site, m.from);
  }
  for (int i = 0; i < SPX.hot_muts.n && !c.failed; ++i) {
    const MutRec& m = SPX.hot_muts.p[i];
    if (m.t > t_P) list_push<MutRec>(c, nodes_of(c)[X].muts, m);
    else { list_push<MutRec>(c, nodes_of(c)[S].muts, make_mut(m.to, m.site, m.from, t_P + (t_P - m.t))); sd_push_back(c, r2r, m.site, m.from, m.to); }
  }
  sort_muts(muts_of(c, X), nmuts(c, X)); sort_muts(muts_of(c, S), nmuts(c, S));
  set_root_muts_from_deltas(c, P, r2r);
  clamp_mut_times(muts_of(c, X), nmuts(c, X), t_P, t_X); clamp_mut_times(muts_of(c, S), nmuts(c, S), t_P, t_S);
  nodes_of(c)[P].lambda = nodes_of(c)[X].lambda - delta_lambda_across_branch(c, X);
  sc_release(c, mark);
}

// ---- inner grafts -----------------------------------------------------------------------------------------
EMAT_FN_START void start_inner_graft_analysis(Ctx& c, int X, Graft& g) { EMAT_TIMED(1);   // spr_move.cpp:582-738
  g.X = X; g.rooty = false; g.delta_log_G = g.log_alpha_mut = 0.0; g.nbi = 0; g.bi = nullptr;
  const int P = nodes_of(c)[X].parent;
  EMAT_CHECK(c, X != hdr_of(c)->root && P != hdr_of(c)->root);
  if (c.failed) return;
  const int S = sibling_of(c, P, X);
  const double t_X = nodes_of(c)[X].t, t_P = nodes_of(c)[P].t;
  g.S = S; g.t_P = t_P;
  int depth = 0, path_muts = 0;
  { EMAT_TIMED(1);   /* start_inner: walk to the part's root (depth, path_muts) */
  for (int cur = X; cur != k_no_node; cur = nodes_of(c)[cur].parent) { ++depth; if (c.includes_run_root || nodes_of(c)[cur].parent != k_no_node) path_muts += nmuts(c, cur); } }
  // The hot path rarely climbs more than two or three branches (it ends where the sibling's missing sites are used up), while
  // the path to the part's root is 5-15 long: room for four entries to start with, doubled when they run out (the
  // abandoned array stays in the arena until the move ends) -- 104 bytes per entry of an arena of a few KB.
  int bi_cap = depth + 2 < 4 ? depth + 2 : 4;
  EMAT_TIMED_BLOCK(1, setup_timer);   /* start_inner: setup (branch infos, PX, sliding missations, pl_A of PX) */
  g.bi = (BranchInfo*)sc_alloc(c, (uint32_t)bi_cap * (uint32_t)sizeof(BranchInfo));
  if (c.failed) return;
  auto bi_room = [&]() {
    if (g.nbi < bi_cap) return;
    if (bi_cap >= depth + 2) { EMAT_FAIL(c, k_part_overflow); return; }
    const int ncap = 2 * bi_cap < depth + 2 ? 2 * bi_cap : depth + 2;
    BranchInfo* nb = (BranchInfo*)sc_alloc(c, (uint32_t)ncap * (uint32_t)sizeof(BranchInfo));
    if (c.failed) return;
    for (int i = 0; i < g.nbi; ++i) nb[i] = g.bi[i];
    g.bi = nb; bi_cap = ncap;
  };
  {
    BranchInfo& PX = g.bi[g.nbi++]; bi_init(PX);
    PX.A = P; PX.B = X; PX.is_open = false; PX.T_to_X = t_X - t_P;
    PX.warm = sc_vec<IvRec>(c, 1); { IvRec all; all.start = 0; all.end = c.L; push(c, PX.warm, all); }
    PX.hot = iv_subtract_sc(c, PX.warm.p, PX.warm.n, miss_of(c, S), (int)nodes_of(c)[S].miss.cnt);
  }
  // sliding_missations = copy of S's missations
  SVec<IvRec> sl_iv = iv_copy_sc(c, miss_of(c, S), (int)nodes_of(c)[S].miss.cnt);
  SVec<FsRec> sl_fs = sc_vec<FsRec>(c, (int)nodes_of(c)[S].mfs.cnt + path_muts + 1);
  { const FsRec* f = mfs_of(c, S); for (int i = 0; i < (int)nodes_of(c)[S].mfs.cnt; ++i) push(c, sl_fs, f[i]); }
  {
    BranchInfo& PX = g.bi[0];
    PX.pl_A = nodes_of(c)[X].lambda;
    const MutRec* m = muts_of(c, X);
    for (int i = nmuts(c, X) - 1; i >= 0; --i) PX.pl_A += dq(c, m[i].site, m[i].to, m[i].from);
  }
  EMAT_TIMED_END(setup_timer);
  double next_pl_B;
  { EMAT_TIMED(1);   /* start_inner: first delta_lambda_across_missations */
  next_pl_B = -1 * delta_lambda_across_missations(c, sl_iv.p, sl_iv.n, sl_fs.p, sl_fs.n); }
  g.bi[0].pl_A -= next_pl_B;
  int cur = P, parent = nodes_of(c)[cur].parent, sibling = sibling_of(c, parent, cur);
  double partial_lambda = next_pl_B;
  { EMAT_TIMED(1);   /* start_inner: sliding-missations loop up the hot path */
  while (sl_iv.n != 0 && !c.failed) {
    bi_room(); if (c.failed) break;
    BranchInfo& bi = g.bi[g.nbi++]; bi_init(bi);
    bi.A = parent; bi.B = cur; bi.is_open = false; bi.T_to_X = t_X - nodes_of(c)[parent].t;
    bi.warm = sl_iv;
    const MutRec* mc = muts_of(c, cur);
    for (int i = nmuts(c, cur) - 1; i >= 0; --i) {
      if (iv_contains(sl_iv.p, sl_iv.n, mc[i].site)) { partial_lambda += dq(c, mc[i].site, mc[i].to, mc[i].from); fsv_set(c, sl_fs, mc[i].site, mc[i].from); }
    }
    bi.hot = iv_subtract_sc(c, bi.warm.p, bi.warm.n, miss_of(c, sibling), (int)nodes_of(c)[sibling].miss.cnt);
    SVec<IvRec> new_sl = iv_subtract_sc(c, bi.warm.p, bi.warm.n, bi.hot.p, bi.hot.n);
    sl_iv = new_sl;
    { int w = 0; for (int i = 0; i < sl_fs.n; ++i) if (iv_contains(sl_iv.p, sl_iv.n, sl_fs.p[i].site)) sl_fs.p[w++] = sl_fs.p[i]; sl_fs.n = w; }
    next_pl_B = -1 * delta_lambda_across_missations(c, sl_iv.p, sl_iv.n, sl_fs.p, sl_fs.n);
    bi.pl_A = partial_lambda - next_pl_B;
    partial_lambda = next_pl_B;
    c.bytes += 64 + 16 * nmuts(c, cur) + 24 * ((int)nodes_of(c)[sibling].miss.cnt + bi.warm.n);
    if (parent != hdr_of(c)->root) {
      cur = parent; parent = nodes_of(c)[cur].parent; sibling = sibling_of(c, parent, cur);
    } else {
      if (!c.includes_run_root) { bi.hot = bi.warm; bi.pl_A += partial_lambda; }
      else if (sl_iv.n != 0) {
        bi_room(); if (c.failed) break;
        BranchInfo& fo = g.bi[g.nbi++]; bi_init(fo);
        fo.A = k_no_node; fo.B = hdr_of(c)->root; fo.is_open = true; fo.T_to_X = t_X - nodes_of(c)[parent].t;
        fo.warm = sl_iv; fo.hot = sl_iv; fo.pl_A = partial_lambda;
      }
      sl_iv.n = 0; sl_fs.n = 0;
    }
  } }
  if (c.failed) return;
  { EMAT_TIMED(1);   /* start_inner: distribute hot mutations along the hot path */
  // distribute hot mutations along the hot path (spr_move.cpp:700-735): gather (mutation, owner), then split by owner
  struct Owned { MutRec m; int owner; int pad; };
  SVec<Owned> tmp = sc_vec<Owned>(c, path_muts + 1);
  for (int i = 0; i < g.nbi; ++i) {
    BranchInfo& bi_i = g.bi[i];
    if (bi_i.B == hdr_of(c)->root) continue;
    const MutRec* mb = muts_of(c, bi_i.B);
    for (int k = nmuts(c, bi_i.B) - 1; k >= 0; --k) {
      if (iv_contains(bi_i.warm.p, bi_i.warm.n, mb[k].site)) {
        // (the hot sets of the branch infos are pairwise disjoint -- hot_j is cut out of warm_j, warm_j+1 is what is left of it -- so the first
        // owner found is the only one: the reference's loop over all j >= i pushes the mutation exactly once too)
        bool found = false;
        for (int j = i; j < g.nbi; ++j) if (iv_contains(g.bi[j].hot.p, g.bi[j].hot.n, mb[k].site)) { Owned o; o.m = mb[k]; o.owner = j; o.pad = 0; push(c, tmp, o); found = true; break; }
        EMAT_CHECK(c, found);
      }
    }
  }
  if (tmp.n == 0) {   // no hot mutation anywhere on the path (most grafts of a sparsely mutated tree): every list is empty, nothing to allocate or scan
    for (int j = 0; j < g.nbi; ++j) { BranchInfo& bi = g.bi[j]; bi.hot_muts.p = nullptr; bi.hot_muts.n = bi.hot_muts.cap = 0; bi.hot_deltas.p = nullptr; bi.hot_deltas.n = bi.hot_deltas.cap = 0; bi.pl_X = bi.pl_A; }
  } else
  for (int j = 0; j < g.nbi && !c.failed; ++j) {
    BranchInfo& bi = g.bi[j];
    int cnt = 0; for (int k = 0; k < tmp.n; ++k) if (tmp.p[k].owner == j) ++cnt;
    bi.hot_muts = sc_vec<MutRec>(c, cnt);
    bi.hot_deltas = sc_vec<SdRec>(c, cnt);
    for (int k = tmp.n - 1; k >= 0; --k) if (tmp.p[k].owner == j) push(c, bi.hot_muts, tmp.p[k].m);   // reversed encounter order
    bi.pl_X = bi.pl_A;
    for (int k = 0; k < bi.hot_muts.n; ++k) {
      const MutRec& m = bi.hot_muts.p[k];
      if (!bi.is_open) sd_push_back(c, bi.hot_deltas, m.site, m.from, m.to);
      bi.pl_X += dq(c, m.site, m.from, m.to);
    }
  } }
  return;
}
EMAT_FN_PNIG void propose_new_inner_graft_mutations(Ctx& c, Graft& g) { EMAT_TIMED(1);   // spr_move.cpp:740-785
  const int X = g.X;
  for (int idx = 0; idx < g.nbi && !c.failed; ++idx) {
    BranchInfo& bi = g.bi[idx];
    if (bi.hot.n == 0) { EMAT_CHECK(c, bi.hot_muts.n == 0); continue; }
    SVec<MutRec> nm = sample_history_for(c, bi);
    if (nm.n != 0) { EMAT_TIMED(1);   /* propose_inner: filter + adjust the sampled history */
      filter_not_hot(nm, bi.hot);
      if (bi.B == X) {
        int w = 0;
        for (int i = 0; i < nm.n; ++i) {
          bool drop = !sd_contains(bi.hot_deltas, nm.p[i].site) && is_site_missing_at(c, X, nm.p[i].site);
          if (!drop) nm.p[w++] = nm.p[i];
        }
        nm.n = w;
      }
      adjust_mutational_history(c, nm, bi.hot_deltas, X, nodes_of(c)[X].t);
    }
    bi.hot_muts = nm;
    recompute_open_pl_A(c, bi);
  }
}
EMAT_FN_FINI void finish_inner_graft_analysis(Ctx& c, Graft& g) { EMAT_TIMED(1);   // spr_move.cpp:787-836
  if (c.failed || g.nbi == 0) return;
  const int X = g.X; const double t_X = nodes_of(c)[X].t;
  g.delta_log_G = 0.0;
  for (int i = 0; i < g.nbi; ++i) { BranchInfo& bi = g.bi[i]; g.delta_log_G += branch_log_G(c, t_X - bi.T_to_X, t_X, bi.pl_X, bi.hot_muts.p, bi.hot_muts.n); }
  BranchInfo& last = g.bi[g.nbi - 1];
  if (last.is_open) for (int i = 0; i < last.hot_muts.n; ++i) g.delta_log_G += log_pi_ratio(c, last.hot_muts.p[i]);
  g.log_alpha_mut = 0.0;
  for (int i = 0; i < g.nbi; ++i) {
    BranchInfo& bi = g.bi[i];
    int Ls = iv_num_sites(bi.hot.p, bi.hot.n);
    if (bi.B == X) Ls = (c.L - nodes_of(c)[X].n_missing) - (iv_num_sites(bi.warm.p, bi.warm.n) - iv_num_sites(bi.hot.p, bi.hot.n));
    g.log_alpha_mut += log_alpha_mut_term(c.mu_prop, Ls, bi.T_to_X, bi.hot_muts.n, bi.is_open, bi.hot_deltas.n);
  }
}
EMAT_D void recalc_lambda_along_hot_path(Ctx& c, const Graft& g) {   // spr_move.cpp:943-950, 1059-1066
  for (int i = 0; i + 1 < g.nbi; ++i) { int A = g.bi[i].A, B = g.bi[i].B; nodes_of(c)[A].lambda = nodes_of(c)[B].lambda - delta_lambda_across_branch(c, B); }
}
EMAT_D void erase_marked_muts(Ctx& c, int node) { MutRec* m = muts_of(c, node); int n = nmuts(c, node), w = 0; for (int i = 0; i < n; ++i) if (m[i].site != -1) m[w++] = m[i]; set_list_cnt(c, nodes_of(c)[node].muts, w); }
EMAT_FN_PEEL void peel_inner_graft(Ctx& c, const Graft& g) { EMAT_TIMED(1);   // spr_move.cpp:838-953
  if (c.failed || g.nbi == 0) return;
  const int X = g.X, P = nodes_of(c)[X].parent, root = hdr_of(c)->root;
  const double t_X = nodes_of(c)[X].t, t_P = nodes_of(c)[P].t;
  const BranchInfo& fin = g.bi[g.nbi - 1];
  ScMark mark = sc_mark(c);
  SVec<SdRec> r2r; r2r.p = nullptr; r2r.n = r2r.cap = 0;
  if (fin.is_open) r2r = deltas_from_root_muts(c, root, path_mut_count(c, X));
  for (int i = 0; i < g.nbi && !c.failed; ++i) {
    const BranchInfo& bi = g.bi[i];
    if (bi.B == root) continue;
    if (bi.B == X && !fin.is_open) { nodes_of(c)[X].muts.cnt = 0; continue; }
    MutRec* mb = muts_of(c, bi.B);
    for (int k = nmuts(c, bi.B) - 1; k >= 0 && !c.failed; --k) {
      MutRec& m = mb[k];
      if (m.site < 0) continue;
      if (iv_contains(bi.warm.p, bi.warm.n, m.site) && (!fin.is_open || !iv_contains(fin.hot.p, fin.hot.n, m.site))) {
        for (int cur = X; cur != bi.B; cur = nodes_of(c)[cur].parent) { int par = nodes_of(c)[cur].parent; miss_set_from_state(c, sibling_of(c, par, cur), m.site, m.from); }
        m.site = -1;
      }
    }
  }
  if (fin.is_open) {
    for (int i = g.nbi - 1; i >= 0 && !c.failed; --i) {
      const BranchInfo& bi = g.bi[i];
      if (bi.B == root) continue;
      MutRec* mb = muts_of(c, bi.B);
      for (int k = 0; k < nmuts(c, bi.B) && !c.failed; ++k) {
        MutRec& m = mb[k];
        if (m.site < 0) continue;
        if (iv_contains(fin.hot.p, fin.hot.n, m.site)) {
          for (int cur = bi.B; cur != root; cur = nodes_of(c)[cur].parent) { int par = nodes_of(c)[cur].parent; miss_set_from_state(c, sibling_of(c, par, cur), m.site, m.to); }
          sd_push_back(c, r2r, m.site, m.from, m.to);
          m.site = -1;
        }
      }
    }
  }
  for (int i = 0; i < g.nbi; ++i) if (g.bi[i].B != root) erase_marked_muts(c, g.bi[i].B);
  const double t_mut_X = 0.5 * (t_P + t_X);
  for (int i = 0; i < g.nbi && !c.failed; ++i) {
    const BranchInfo& bi = g.bi[i];
    if (bi.B == root) continue;
    for (int k = 0; k < bi.hot_deltas.n; ++k) list_push<MutRec>(c, nodes_of(c)[X].muts, make_mut(bi.hot_deltas.p[k].from, bi.hot_deltas.p[k].site, bi.hot_deltas.p[k].to, t_mut_X));
  }
  if (fin.is_open) set_root_muts_from_deltas(c, root, r2r);
  recalc_lambda_along_hot_path(c, g);
  sc_release(c, mark);
}
EMAT_FN_APPLY void apply_inner_graft(Ctx& c, const Graft& g) { EMAT_TIMED(1);   // spr_move.cpp:955-1069
  if (c.failed || g.nbi == 0) return;
  const int X = g.X, root = hdr_of(c)->root;
  const BranchInfo& fin = g.bi[g.nbi - 1];
  ScMark mark = sc_mark(c);
  nodes_of(c)[X].muts.cnt = 0;
  SVec<SdRec> r2r; r2r.p = nullptr; r2r.n = r2r.cap = 0;
  if (fin.is_open) r2r = deltas_from_root_muts(c, root, fin.hot_muts.n);
  for (int i = 0; i < g.nbi && !c.failed; ++i) {
    const BranchInfo& bi = g.bi[i];
    if (bi.B == X) { list_assign<MutRec>(c, nodes_of(c)[X].muts, bi.hot_muts.p, bi.hot_muts.n); continue; }
    if (!bi.is_open) {
      for (int k = 0; k < bi.hot_muts.n && !c.failed; ++k) {
        const MutRec m = bi.hot_muts.p[k];
        for (int cur = X; cur != bi.A; cur = nodes_of(c)[cur].parent) {
          int par = nodes_of(c)[cur].parent;
          if (nodes_of(c)[par].t <= m.t && m.t < nodes_of(c)[cur].t) { list_push<MutRec>(c, nodes_of(c)[cur].muts, m); break; }
          miss_set_from_state(c, sibling_of(c, par, cur), m.site, m.to);
        }
      }
    } else {
      for (int k = bi.hot_muts.n - 1; k >= 0 && !c.failed; --k) {
        const MutRec m = bi.hot_muts.p[k];
        for (int cur = X; cur != root; cur = nodes_of(c)[cur].parent) {
          int par = nodes_of(c)[cur].parent;
          if (nodes_of(c)[par].t <= m.t && m.t < nodes_of(c)[cur].t) list_push<MutRec>(c, nodes_of(c)[cur].muts, m);
          if (nodes_of(c)[par].t <= m.t) miss_set_from_state(c, sibling_of(c, par, cur), m.site, m.from);
        }
        sd_push_back(c, r2r, m.site, m.to, m.from);
      }
    }
  }
  for (int i = 0; i < g.nbi; ++i) {
    const BranchInfo& bi = g.bi[i];
    if (!bi.is_open) { sort_muts(muts_of(c, bi.B), nmuts(c, bi.B)); clamp_mut_times(muts_of(c, bi.B), nmuts(c, bi.B), nodes_of(c)[bi.A].t, nodes_of(c)[bi.B].t); }
  }
  if (fin.is_open) set_root_muts_from_deltas(c, root, r2r);
  recalc_lambda_along_hot_path(c, g);
  sc_release(c, mark);
}

// ---- dispatch (spr_move.cpp:9-89, 549-580, 1071-1099) -------------------------------------------------------
EMAT_D bool is_rooty(const Ctx& c, int X) { return nodes_of(c)[X].parent == hdr_of(c)->root; }
// The graft is written where the caller keeps it (an arena frame): no copy through the caller's private stack.
EMAT_D void analyze_graft(Ctx& c, int X, Graft& g) {
  if (is_rooty(c, X)) { start_rooty_graft_analysis(c, X, g); finish_rooty_graft_analysis(c, g); }
  else { start_inner_graft_analysis(c, X, g); finish_inner_graft_analysis(c, g); }
}
EMAT_D void propose_new_graft(Ctx& c, int X, Graft& g) {
  if (is_rooty(c, X)) { start_rooty_graft_analysis(c, X, g); if (!c.failed) propose_new_rooty_graft_mutations(c, g); finish_rooty_graft_analysis(c, g); }
  else { start_inner_graft_analysis(c, X, g); if (!c.failed) propose_new_inner_graft_mutations(c, g); finish_inner_graft_analysis(c, g); }
}
EMAT_D void peel_graft(Ctx& c, const Graft& g) { if (is_rooty(c, g.X)) peel_rooty_graft(c, g); else peel_inner_graft(c, g); }
EMAT_D void apply_graft(Ctx& c, const Graft& g) { if (is_rooty(c, g.X)) apply_rooty_graft(c, g); else apply_inner_graft(c, g); }
EMAT_D int count_min_mutations(const Ctx& c, const Graft& g) {
  if (g.rooty) return g.bi[k_SPX].hot_deltas.n;
  int r = 0; for (int i = 0; i < g.nbi; ++i) if (!g.bi[i].is_open) r += g.bi[i].hot_deltas.n; return r;
}
// summarize_closed_mutations: fresh scratch delta list with room for `extra` more entries
EMAT_FN_SUMM SVec<SdRec> summarize_closed_mutations(Ctx& c, const Graft& g, int extra) { EMAT_TIMED(1);
  int tot = 0;
  if (g.rooty) tot = g.bi[k_SPX].hot_deltas.n; else for (int i = 0; i < g.nbi; ++i) if (!g.bi[i].is_open) tot += g.bi[i].hot_deltas.n;
  SVec<SdRec> r = sc_vec<SdRec>(c, tot + extra + 1);
  if (g.rooty) { const SVec<SdRec>& d = g.bi[k_SPX].hot_deltas; for (int k = 0; k < d.n; ++k) sd_push_back(c, r, d.p[k].site, d.p[k].from, d.p[k].to); }
  else for (int i = 0; i < g.nbi; ++i) if (!g.bi[i].is_open) { const SVec<SdRec>& d = g.bi[i].hot_deltas; for (int k = 0; k < d.n; ++k) sd_push_back(c, r, d.p[k].site, d.p[k].from, d.p[k].to); }
  return r;
}

// =================================================================================================
// SPR candidate study (spr_study.h:17-171, spr_study.cpp:9-549)
// =================================================================================================
// 48 B, three 16-byte groups: the scan writes the first two with one wide store each, the study fills the third.
struct alignas(16) Region { int branch, mut_idx, min_muts, pad; double t_min, t_max; double logW, W; };
struct Study {
  SVec<Region> regions;
  double lambda_X, mu, f, t_X, t_max_tip, log_Wmax, sum_W;
};
// DFS work item in one 64-bit word (one load / one store per pop / push): branch in the high half, then
// mut_idx (list sizes are < 2^16) and the backtracking flag in bit 0.
typedef uint64_t WorkItem;
EMAT_DF WorkItem wi_make(int branch, int mut_idx, int backtracking) { return ((uint64_t)(uint32_t)branch << 32) | ((uint64_t)(uint32_t)mut_idx << 1) | (uint64_t)(backtracking & 1); }
EMAT_DF int wi_branch(WorkItem w) { return (int)(uint32_t)(w >> 32); }
EMAT_DF int wi_mut_idx(WorkItem w) { return (int)((uint32_t)w >> 1); }
EMAT_DF int wi_backtracking(WorkItem w) { return (int)(w & 1u); }

EMAT_D double region_t_min(Ctx& c, int b, int mi) { if (b == hdr_of(c)->root) return k_neg_dbl_max; if (mi == 0) return nodes_of(c)[nodes_of(c)[b].parent].t; return muts_of(c, b)[mi - 1].t; }
EMAT_D double region_t_max(Ctx& c, int b, int mi) { if (b == hdr_of(c)->root) return nodes_of(c)[b].t; if (mi == nmuts(c, b)) return nodes_of(c)[b].t; return muts_of(c, b)[mi].t; }

// seed_fill_from (spr_study.cpp:9-24).  `deltas` (cur -> X) is consumed; `missing_at_X` must outlive the call.
// Regions grow upwards from the scratch top while the DFS work stack grows downwards from the scratch end.
EMAT_DN SVec<Region> study_seed_fill(Ctx& c, int X, double t_X, const SVec<IvRec>& missing_at_X_in, int max_muts_from_start,
                                    int init_branch, int init_mut_idx, SVec<SdRec>& deltas_in, bool can_change_root, HotBlock hot) {
  SVec<Region> res; res.p = nullptr; res.n = 0; res.cap = 0;
  if (c.failed) return res;
  // Local scans (the 99% case) work out of the move's reserved LDS block when there is one: the two sets that are
  // searched for every region are copied to its front (a local scan never modifies them), regions and DFS stack
  // share the rest and migrate to the HBM arena if they outgrow it.  Without a block they start in what is left of
  // the LDS arena, or in HBM.
  SVec<IvRec> missing_at_X = missing_at_X_in;
  SVec<SdRec> deltas_local = deltas_in;
  const bool local_scan = (max_muts_from_start == 1);
  ScSpan span;
  uint32_t front = 0;
  bool in_block = false;   // (not `front != 0`: both sets are often empty -- no missing interval at X, no site delta -- and the block is there all the same)
  if (local_scan && hot.p != nullptr) {
    const uint32_t need = (((uint32_t)missing_at_X_in.n * (uint32_t)sizeof(IvRec) + 15u) & ~15u) + (((uint32_t)deltas_in.n * (uint32_t)sizeof(SdRec) + 15u) & ~15u);
    if (need + 512u <= hot.bytes) {
      IvRec* mi = (IvRec*)hot.p; for (int i = 0; i < missing_at_X_in.n; ++i) mi[i] = missing_at_X_in.p[i];
      front = ((uint32_t)missing_at_X_in.n * (uint32_t)sizeof(IvRec) + 15u) & ~15u;
      SdRec* di = (SdRec*)(hot.p + front); for (int i = 0; i < deltas_in.n; ++i) di[i] = deltas_in.p[i];
      missing_at_X.p = mi; deltas_local.p = di; deltas_local.cap = deltas_in.n;
      front = need; in_block = true;
    }
  }
  SVec<SdRec>& deltas = in_block ? deltas_local : deltas_in;
  if (in_block) { span.lo = hot.p + front; span.hi = hot.p + (hot.bytes & ~15u); span.lds = true; span.reserved = true; }
  else span = local_scan ? sc_span(c, 1024) : sc_span_hbm(c);
#ifdef EMAT_PROFILE_PHASES
  if (span.lds) hdr_of(c)->phase_ticks[3] += 1;   // scans that start in the LDS arena
#endif
  res.p = (Region*)span.lo;
  const int root = hdr_of(c)->root;
  if (local_scan && deltas.n < k_scan_max_deltas) {
    // ---- local scan (99 % of the scans): at most one counted mutation is crossed inside the scope, so the cur->X delta
    // set is never modified and its size follows from the INITIAL set (+1 if the crossed site is new, -1 if the
    // crossing cancels the entry, 0 if it only rewrites it; site_deltas.h:43-83).  That removes the need to undo
    // anything on the way back: a work item carries the state of the region that pushed it -- {branch, mut_idx,
    // pusher branch, pusher mut_idx | counted crossings so far << 16 | delta-set size << 18}, one 16-byte load or
    // store -- and the DFS visits the regions in exactly the order of the general algorithm below.
    int4* stack_top = (int4*)span.hi;   // items live at stack_top[-1], [-2], ...
    int sp = 0;
    auto room = [&](int extra_regions, int extra_items) -> bool {
      const uint8_t* lo = span.lo + (size_t)(res.n + extra_regions) * sizeof(Region);
      const uint8_t* hi = span.hi - (size_t)(sp + extra_items) * sizeof(int4);
      if (lo + 16 <= hi) return true;
      if (!span.lds) return false;
      ScSpan big = sc_span_hbm(c);
      if (big.lo + (size_t)(res.n + extra_regions) * sizeof(Region) + 16 > big.hi - (size_t)(sp + extra_items) * sizeof(int4)) return false;
      Region* nr = (Region*)big.lo; int4* nb = (int4*)big.hi;
      for (int i = 0; i < res.n; ++i) nr[i] = res.p[i];
      for (int i = 1; i <= sp; ++i) nb[-i] = stack_top[-i];
      span = big; res.p = nr; stack_top = nb;
      return true;
    };
    auto push_item = [&](int tb, int tmi, int pb, int pmi, int fs, int size) {
      if (!room(0, 1)) { EMAT_FAIL(c, k_part_overflow); return; }
      ++sp; stack_top[-sp] = make_int4(tb, tmi, pb, (pmi & 0xffff) | (fs << 16) | (size << 18));
    };
    push_item(init_branch, init_mut_idx, k_no_node, -1, 0, deltas.n);
    while (sp > 0 && !c.failed) {
      const int4 it = stack_top[-sp]; --sp;
      const int branch = it.x, mut_idx = it.y, pb = it.z, pmi = (int)(it.w & 0xffff);   // (16 unsigned bits: a list holds at most k_max_list_len = 65 535 entries; the first item's 0xffff is never looked at, its pusher being k_no_node)
      int fs = (it.w >> 16) & 3, size = (int)((uint32_t)it.w >> 18);
      // move_to_neighbor (spr_study.cpp:43-91)
      if (pb != k_no_node && branch == pb) {
        const MutRec* m = muts_of(c, branch);
        const bool down = (mut_idx == pmi + 1);
        if (!down && mut_idx != pmi - 1) EMAT_FAIL(c, k_part_internal);
        const MutRec mm = m[down ? pmi : mut_idx];
        if (!iv_contains(missing_at_X.p, missing_at_X.n, mm.site)) {
          if (fs == 0) {
            const int new_from = down ? (int)mm.to : (int)mm.from;
            const int kk = sd_lower_bound(deltas.p, deltas.n, mm.site);
            const bool present = kk < deltas.n && deltas.p[kk].site == mm.site;
            size = deltas.n + (present ? (new_from == (int)deltas.p[kk].to ? -1 : 0) : +1);
          }
          fs += 1;
        }
      }
      if (branch == X || fs > 1) continue;
      // visit_cur_region (spr_study.cpp:93-101): logW / W are the study's to fill; two 16-byte stores here
      if (!room(1, 0)) { EMAT_FAIL(c, k_part_overflow); break; }
      {
        Region* r = &res.p[res.n++];
        const int4 head = make_int4(branch, mut_idx, size, 0);
        const double2 times = make_double2(region_t_min(c, branch, mut_idx), region_t_max(c, branch, mut_idx));
        *(int4*)r = head; *(double2*)&r->t_min = times;
      }
      c.bytes += 64 + 16;
      // seed_neighbors_except (spr_study.cpp:103-128)
      const int nm = nmuts(c, branch);
      if (branch != root) {
        if (mut_idx > 0) { if (!(branch == pb && mut_idx - 1 == pmi)) push_item(branch, mut_idx - 1, branch, mut_idx, fs, size); }
        else { const int ub = nodes_of(c)[branch].parent, umi = nmuts(c, ub); if (!(ub == pb && umi == pmi)) push_item(ub, umi, branch, mut_idx, fs, size); }
      }
      if (mut_idx < nm) { if (!(branch == pb && mut_idx + 1 == pmi)) push_item(branch, mut_idx + 1, branch, mut_idx, fs, size); }
      else if (!is_tip(c, branch)) {
        const int c0 = nodes_of(c)[branch].child0, c1 = nodes_of(c)[branch].child1;
        if (!(c0 == pb && 0 == pmi)) push_item(c0, 0, branch, mut_idx, fs, size);
        if (!(c1 == pb && 0 == pmi)) push_item(c1, 0, branch, mut_idx, fs, size);
      }
    }
  } else {
  // ---- general scan (spr_study.cpp:9-128): the delta set is updated while walking and restored by "backtracking" items
  WorkItem* stack_base = (WorkItem*)span.hi;   // items live at stack_base[-1], [-2], ...
  int sp = 0;
  auto room = [&](int extra_regions, int extra_items) -> bool {
    const uint8_t* lo = span.lo + (size_t)(res.n + extra_regions) * sizeof(Region);
    const uint8_t* hi = span.hi - (size_t)(sp + extra_items) * sizeof(WorkItem);
    if (lo + 16 <= hi) return true;
    if (!span.lds) return false;
    ScSpan big = sc_span_hbm(c);
    if (big.lo + (size_t)(res.n + extra_regions) * sizeof(Region) + 16 > big.hi - (size_t)(sp + extra_items) * sizeof(WorkItem)) return false;
    Region* nr = (Region*)big.lo; WorkItem* nb = (WorkItem*)big.hi;
    for (int i = 0; i < res.n; ++i) nr[i] = res.p[i];
    for (int i = 1; i <= sp; ++i) nb[-i] = stack_base[-i];
    span = big; res.p = nr; stack_base = nb;
    return true;
  };
  int cur_branch = k_no_node, cur_mut_idx = -1, cur_from_start = 0;
  int cur_size = deltas.n;
  auto add_forward = [&](int tb, int tmi) {
    if (!room(0, 2)) { EMAT_FAIL(c, k_part_overflow); return; }
    stack_base[-(sp + 1)] = wi_make(cur_branch, cur_mut_idx, 1);
    stack_base[-(sp + 2)] = wi_make(tb, tmi, 0);
    sp += 2;
  };
  add_forward(init_branch, init_mut_idx);
  while (sp > 0 && !c.failed) {
    const WorkItem wi = stack_base[-sp]; --sp;
    struct { int branch, mut_idx, backtracking; } w = {wi_branch(wi), wi_mut_idx(wi), wi_backtracking(wi)};
    const int ob = cur_branch, omi = cur_mut_idx;
    // move_to_neighbor (spr_study.cpp:43-91)
    if (cur_branch != k_no_node && w.branch == cur_branch) {
      const MutRec* m = muts_of(c, cur_branch);
      const bool down = (w.mut_idx == cur_mut_idx + 1);
      if (!down && w.mut_idx != cur_mut_idx - 1) EMAT_FAIL(c, k_part_internal);
      const MutRec mm = m[down ? cur_mut_idx : w.mut_idx];
      if (!iv_contains(missing_at_X.p, missing_at_X.n, mm.site)) {
        if (down) sd_pop_front(c, deltas, mm.site, mm.from, mm.to); else sd_push_front(c, deltas, mm.site, mm.from, mm.to);
        cur_from_start += w.backtracking ? -1 : +1;
        cur_size = deltas.n;
      }
    }
    cur_branch = w.branch; cur_mut_idx = w.mut_idx;
    if (!w.backtracking && cur_branch != X && cur_from_start <= max_muts_from_start) {
      // visit_cur_region (spr_study.cpp:93-101)
      if (!room(1, 0)) { EMAT_FAIL(c, k_part_overflow); break; }
      {   // logW / W are the study's to fill (make_study); only the two leading 16-byte groups are written here
        Region* r = &res.p[res.n++];
        const int4 head = make_int4(cur_branch, cur_mut_idx, cur_size, 0);
        const double2 times = make_double2(region_t_min(c, cur_branch, cur_mut_idx), region_t_max(c, cur_branch, cur_mut_idx));
        *(int4*)r = head; *(double2*)&r->t_min = times;
      }
      c.bytes += 64 + 16;
      // seed_neighbors_except (spr_study.cpp:103-128)
      if (cur_branch != root) {
        if (cur_mut_idx > 0) { if (!(cur_branch == ob && cur_mut_idx - 1 == omi)) add_forward(cur_branch, cur_mut_idx - 1); }
        else { int pb = nodes_of(c)[cur_branch].parent, pmi = nmuts(c, pb); if (!(pb == ob && pmi == omi)) add_forward(pb, pmi); }
      }
      if (cur_mut_idx < nmuts(c, cur_branch)) { if (!(cur_branch == ob && cur_mut_idx + 1 == omi)) add_forward(cur_branch, cur_mut_idx + 1); }
      else if (!is_tip(c, cur_branch)) {
        int c0 = nodes_of(c)[cur_branch].child0, c1 = nodes_of(c)[cur_branch].child1;
        if (!(c0 == ob && 0 == omi)) add_forward(c0, 0);
        if (!(c1 == ob && 0 == omi)) add_forward(c1, 0);
      }
    }
  }
  }
  // account_for_Xs_detachment (spr_study.cpp:130-209) and remove_regions_in_Xs_future (:211-224), fused into one
  // read-modify-compact pass over the regions (each is loaded and stored once, as whole 16-byte groups)
  int w = 0;
  {
    const bool have_X = X != k_no_node;
    const int P = have_X ? nodes_of(c)[X].parent : k_no_node, S = have_X ? sibling_of(c, P, X) : k_no_node;
    const int nGP = have_X ? nmuts(c, P) : 0;
    const int nS = (have_X && P == root) ? nmuts(c, S) : 0;
    const double t_min_P_end = (have_X && P != root) ? region_t_min(c, P, nGP) : 0.0;
    for (int i = 0; i < res.n; ++i) {
      int4 head = *(const int4*)&res.p[i];
      double2 times = *(const double2*)&res.p[i].t_min;
      int branch = head.x, mut_idx = head.y;
      if (!can_change_root && branch == root) continue;
      if (have_X && (branch == S || branch == P)) {
        if (P != root) {
          if (branch == S) { if (mut_idx == 0) times.x = t_min_P_end; mut_idx += nGP; }
          else { if (mut_idx == nGP) continue; branch = S; }
        } else {
          if (!can_change_root) { if (branch == P) continue; }
          else {
            if (branch == S && mut_idx == nS) { mut_idx += nGP; times.x = k_neg_dbl_max; }
            else continue;
          }
        }
      }
      if (times.x >= t_X) continue;
      if (times.y > t_X) times.y = t_X;
      head.x = branch; head.y = mut_idx;
      *(int4*)&res.p[w] = head; *(double2*)&res.p[w].t_min = times;
      ++w;
    }
  }
  res.n = w; res.cap = w;
  sc_span_commit(c, span, (uint32_t)w * (uint32_t)sizeof(Region));
  return res;
}
struct RootRegionParams { double f, t_S, s_min, s_max, x_min, x_max; int m; };
EMAT_D RootRegionParams root_region_params(Ctx& c, const Study& st, const Region& r) {
  RootRegionParams p; p.f = st.f; p.m = r.min_muts; p.t_S = nodes_of(c)[r.branch].t;
  p.s_min = fabs(st.t_X - p.t_S);
  double t_early = st.t_X < p.t_S ? st.t_X : p.t_S;
  double tree_span = st.t_max_tip - t_early;
  EMAT_CHECK(c, tree_span >= 0.0);
  p.s_max = p.s_min + 20.0 * tree_span;
  p.x_min = st.lambda_X * p.f * p.s_min; p.x_max = st.lambda_X * p.f * p.s_max;
  return p;
}
EMAT_D double safe_log_gamma_integral(Ctx& c, double a, double x_min, double x_max) {   // safe_gamma_math.h:82-90
  EMAT_CHECK(c, x_min < x_max);
  double Q_hi = gamma_q(a, x_min), Q_lo = gamma_q(a, x_max);
  return m_log(Q_hi - Q_lo);
}
constexpr double k_ln2 = 0.693147180559945309417232121458176568;
// log-weight of a region above the root (spr_study.cpp:336-370): truncated-Gamma weight through lgamma and the incomplete
// gamma function, or the power law below x_max = 0.01.  Out of line: its register appetite (lgamma, pow, log1p and the
// continued fraction) would otherwise make every study -- all 64 lanes -- spill around a branch that only the part holding
// the run's root ever takes.
EMAT_DN double root_region_log_weight(Ctx& c, const Study& st, const Region& r) {
  const double f = st.f, lambda_X = st.lambda_X;
  const int m = r.min_muts;
  RootRegionParams p = root_region_params(c, st, r);
  if (p.x_max < 0.01) {
    double alpha = f * m + 1;
    return -k_ln2 + m_log(f * lambda_X) + f * m * m_log(st.mu / 3) + alpha * m_log(p.s_max) + m_log1p(-pow(p.s_min / p.s_max, alpha)) - m_log(alpha);
  }
  return -k_ln2 + f * m * m_log(st.mu / (3 * lambda_X * f)) + lgamma(f * m + 1) + safe_log_gamma_integral(c, f * m + 1, p.x_min, p.x_max);
}
EMAT_DN Study make_study(Ctx& c, SVec<Region> regions, int num_missing_at_X, double lambda_X, double f, double t_X, double t_max_tip) { EMAT_TIMED(1);   // spr_study.cpp:226-385
  Study st; st.regions = regions; st.lambda_X = lambda_X; st.f = f; st.t_X = t_X; st.t_max_tip = t_max_tip; st.log_Wmax = 0.0; st.sum_W = 0.0;
  st.mu = lambda_X / (c.L - num_missing_at_X);
  if (c.failed) return st;
  EMAT_CHECK(c, regions.n > 0);
  // pass 1: log-weights, with the running maximum (NaN-aware, as std::max(a, b) = (a < b) ? b : a seeded with the first)
  double log_Wmax = 0.0;
  for (int i = 0; i < regions.n; ++i) {
    const Region r = regions.p[i];   // head + times: two 16-byte loads
    const int m = r.min_muts;
    double logW;
    if (r.t_min != k_neg_dbl_max) {
      double t_prime = 0.5 * (r.t_min + r.t_max);
      logW = m_log(f * lambda_X * (r.t_max - r.t_min)) + f * (-lambda_X * (t_X - t_prime) + m * m_log(st.mu * (t_X - t_prime) / 3));
    } else logW = root_region_log_weight(c, st, r);
    regions.p[i].logW = logW;
    if (i == 0) log_Wmax = logW; else if (log_Wmax < logW) log_Wmax = logW;
  }
  // pass 2: normalise by the maximum, weights and their sum (one 16-byte store per region)
  if (regions.n > 0) {
    st.log_Wmax = log_Wmax;
    for (int i = 0; i < regions.n; ++i) {
      const double lw = regions.p[i].logW - log_Wmax, W = m_exp(lw);
      *(double2*)&regions.p[i].logW = make_double2(lw, W);
      st.sum_W += W;
    }
  }
  c.bytes += 2 * 48 * (int64_t)regions.n; c.bytes_w += 48 * (int64_t)regions.n;   // every region record written once and read once
  return st;
}
EMAT_D int study_pick_nexus_region(Ctx& c, const Study& st) {   // spr_study.cpp:404-422
  double r = uniform_co(c, 0.0, st.sum_W);
  for (int i = 0; i < st.regions.n; ++i) { if (st.regions.p[i].W >= r) return i; r -= st.regions.p[i].W; }
  return 0;
}
EMAT_FN_PICKT double study_pick_time_in_region(Ctx& c, const Study& st, int idx) {   // spr_study.cpp:424-471
  const Region& r = st.regions.p[idx];
  if (r.t_min != k_neg_dbl_max) return uniform_oc(c, r.t_min, r.t_max);
  RootRegionParams p = root_region_params(c, st, r);
  double rand_s;
  if (p.x_max < 0.01) {
    double alpha = p.f * p.m + 1;
    double U = u01_oo(c);
    double smin_a = pow(p.s_min, alpha), smax_a = pow(p.s_max, alpha);
    rand_s = pow(smin_a + U * (smax_a - smin_a), 1.0 / alpha);
  } else {   // safe_sample_truncated_gamma (safe_gamma_math.h:112-139)
    double alpha = p.f * p.m + 1, beta = st.lambda_X * p.f;
    double y_lo = beta * p.s_min, y_hi = beta * p.s_max;
    double Q_hi = gamma_q(alpha, y_lo), Q_lo = gamma_q(alpha, y_hi);
    EMAT_CHECK(c, Q_lo < Q_hi);
    double rand_Q = uniform_oc(c, Q_lo, Q_hi);
    double y = gamma_q_inv(alpha, rand_Q);
    double x = y / beta;
    rand_s = x < p.s_min ? p.s_min : (p.s_max < x ? p.s_max : x);
  }
  double rand_t = 0.5 * (st.t_X + p.t_S - rand_s);
  double lo = r.t_max < rand_t ? r.t_max : rand_t;
  return r.t_min > lo ? r.t_min : lo;
}
EMAT_D int study_find_region(const Study& st, int branch, double t) {   // spr_study.cpp:474-484
  for (int i = 0; i < st.regions.n; ++i) { const Region& r = st.regions.p[i]; if (r.branch == branch && r.t_min < t && t <= r.t_max) return i; }
  return -1;
}
EMAT_FN_LALPHA double study_log_alpha_in_region(Ctx& c, const Study& st, int idx, double t) {   // spr_study.cpp:486-549
  const Region& r = st.regions.p[idx];
  double log_p_region = r.logW - m_log(st.sum_W);
  if (r.t_min != k_neg_dbl_max) return log_p_region - m_log(r.t_max - r.t_min);
  RootRegionParams p = root_region_params(c, st, r);
  double s = st.t_X - t + p.t_S - t;
  if (s > p.s_max + 1e-6) return -k_inf;
  if (p.x_max < 0.01) {
    double alpha = p.f * p.m + 1;
    return log_p_region + k_ln2 + m_log(alpha) + (alpha - 1) * m_log(s) + -alpha * m_log(p.s_max) + -m_log1p(-pow(p.s_min / p.s_max, alpha));
  }
  return log_p_region + k_ln2 + m_log(st.lambda_X * p.f) + p.f * p.m * m_log(st.lambda_X * p.f * s) + -st.lambda_X * p.f * s
      + -lgamma(p.f * p.m + 1) - safe_log_gamma_integral(c, p.f * p.m + 1, p.x_min, p.x_max);
}

// =================================================================================================
// Wave-cooperative candidate ranking (all 64 lanes; the rest of a move runs on lane 0)
// =================================================================================================
// State of an SPR1 move across its stretches on lane 0 and the wave-wide work in between.  Lives in the scratch arena.
struct Spr1Frame {
  // what the scan + study service reads ...
  int X, init_branch, limit, n_missing_at_X;
  double t_X, lambda_X, f, t_max_tip;
  SVec<IvRec> missing_at_X;
  SVec<SdRec> deltas;
  HotBlock hot;
  bool can_change_root;
  // ... and what it leaves behind
  Study study;
  // the move's own variables that outlive a stretch
  int P, old_S, old_G, extra, old_min_muts, new_min_muts, new_S, pre_new_region_min_muts;
  double old_t_P, new_t_P, log_alpha_o2n;
  Graft old_graft, new_graft;
};
EMAT_DF int wave_lane() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
EMAT_DF double wave_bcast(double x, int src) { return __shfl(x, src, 64); }

// Spr_study's constructor (spr_study.cpp:226-385) with one region per lane: the log-weights (two logarithms each, an
// incomplete gamma function above the root) and the weights exp(logW - max) are evaluated side by side, with exactly the
// arithmetic of make_study; the maximum is order-independent, and the sum of the weights -- which is not -- is taken by
// lane 0 in region order, so every number equals the serial one bit for bit.
EMAT_D void wave_make_study(Ctx& c, Spr1Frame& fr, SVec<Region> regions) {
  const int lane = wave_lane();
  Study& st = fr.study;
  if (lane == 0) {
    st.regions = regions; st.lambda_X = fr.lambda_X; st.f = fr.f; st.t_X = fr.t_X; st.t_max_tip = fr.t_max_tip; st.log_Wmax = 0.0; st.sum_W = 0.0;
    st.mu = fr.lambda_X / (c.L - fr.n_missing_at_X);
    EMAT_CHECK(c, regions.n > 0);
  }
  __syncthreads();
  if (c.failed || regions.n <= 0) return;
  const double f = fr.f, lambda_X = fr.lambda_X, t_X = fr.t_X, mu = st.mu;
  // pass 1: log-weights; running maximum as std::max(a, b) = (a < b) ? b : a seeded with the first (NaN-aware)
  double my_max = -k_inf; bool any = false;
  for (int i = lane; i < regions.n; i += 64) {
    const Region r = regions.p[i];
    const int m = r.min_muts;
    double logW;
    if (r.t_min != k_neg_dbl_max) {
      double t_prime = 0.5 * (r.t_min + r.t_max);
      logW = m_log(f * lambda_X * (r.t_max - r.t_min)) + f * (-lambda_X * (t_X - t_prime) + m * m_log(mu * (t_X - t_prime) / 3));
    } else logW = root_region_log_weight(c, st, r);
    regions.p[i].logW = logW;
    if (logW == logW) { if (!any || my_max < logW) my_max = logW; any = true; }   // NaNs never replace the running maximum
  }
  for (int off = 32; off > 0; off >>= 1) { const double o = __shfl_down(my_max, off, 64); if (my_max < o) my_max = o; }
  __syncthreads();
  double log_Wmax = wave_bcast(my_max, 0);
  { const double first = regions.p[0].logW; if (first != first) log_Wmax = first; }   // a NaN in front stays (every later comparison is false)
  // pass 2: normalise by the maximum, weights
  for (int i = lane; i < regions.n; i += 64) {
    const double lw = regions.p[i].logW - log_Wmax, W = m_exp(lw);
    *(double2*)&regions.p[i].logW = make_double2(lw, W);
  }
  __syncthreads();
  if (lane == 0) {
    double sum_W = 0.0;
    for (int i = 0; i < regions.n; ++i) sum_W += regions.p[i].W;
    st.log_Wmax = log_Wmax; st.sum_W = sum_W;
    c.bytes += 2 * 48 * (int64_t)regions.n; c.bytes_w += 48 * (int64_t)regions.n;
  }
  __syncthreads();
}

// ---- the candidate scan, level by level ---------------------------------------------------------------------------------
// study_seed_fill's local scan (at most one counted mutation crossed: 99 % of the scans) is a depth-first walk over the
// tree of regions around the starting point; what a region contributes, and which neighbours it hands on to, depends only
// on the region and on the state its pusher handed over.  So the walk is done breadth-first instead, one region per lane
// and level (ballot + prefix popcount place each lane's up to three successors in the next level), and the depth-first
// ORDER -- which the study's sums and the region pick depend on -- is restored afterwards: subtree sizes bottom-up, then
// positions top-down (a region is followed by the subtree of its LAST-pushed successor first, as on a stack).  The
// region list that comes out is the serial one entry for entry.
struct alignas(16) ScanItem {
  int branch, mut_idx, pb, w;            // target region, pusher's branch, pusher's (mut_idx & 0xffff) | crossings << 16 | delta-set size << 18; after processing: the item's own
  int child_begin, nch_vis, sz, pos;     // successors [child_begin, child_begin + (nch_vis & 3)), visited flag in bit 8; subtree size; depth-first position
};
static_assert(sizeof(ScanItem) == 32, "ScanItem must be two 16-byte groups");

EMAT_D bool wave_local_scan(Ctx& c, Spr1Frame& fr) {
  const int lane = wave_lane();
  const uint64_t below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  const int X = fr.X, root = hdr_of(c)->root;
  SVec<IvRec> miss = fr.missing_at_X;
  SVec<SdRec> del = fr.deltas;
  if (del.n >= k_scan_max_deltas) return false;
  // the two sets every region is checked against go to the front of the move's LDS block
  uint32_t front = 0;
  bool in_block = false;   // (see study_seed_fill: empty sets still have their place in the block)
  if (fr.hot.p != nullptr) {
    const uint32_t b_miss = ((uint32_t)miss.n * (uint32_t)sizeof(IvRec) + 15u) & ~15u, b_del = ((uint32_t)del.n * (uint32_t)sizeof(SdRec) + 15u) & ~15u;
    if (b_miss + b_del + 512u <= fr.hot.bytes) {
      IvRec* mi = (IvRec*)fr.hot.p; SdRec* di = (SdRec*)(fr.hot.p + b_miss);
      for (int i = lane; i < miss.n; i += 64) mi[i] = miss.p[i];
      for (int i = lane; i < del.n; i += 64) di[i] = del.p[i];
      miss.p = mi; del.p = di; front = b_miss + b_del; in_block = true;
      __syncthreads();
    }
  }
  bool done = false;
  ScanItem* items = nullptr; int* lvl = nullptr; int total = 0, nlev = 0;
  uint8_t* hbm_lo = nullptr;   // lower end of the HBM span when the items sit at its top (regions then come from below)
  for (int attempt = 0; attempt < 2 && !done; ++attempt) {
    uint8_t* base; uint32_t bytes;
    if (attempt == 0) {
      if (fr.hot.p == nullptr || !in_block || fr.hot.bytes - front < 24u * 36u) continue;
      base = fr.hot.p + front; bytes = (fr.hot.bytes - front) & ~15u;
    } else {
      const uint32_t g0 = (c.sc_top + 15u) & ~15u, g1 = hdr_of(c)->scratch_end & ~15u;
      if (g1 < g0 + 8192u) return false;
      const uint32_t span = g1 - g0, skip = (span / 96u * 48u) & ~15u;   // regions (48 B each) below, items + levels (36 B each) above
      base = c.G + g0 + skip; bytes = span - skip; hbm_lo = c.G + g0;
    }
    const int cap = (int)(bytes / 36u);
    items = (ScanItem*)base; lvl = (int*)(base + (size_t)cap * sizeof(ScanItem));
    if (lane == 0) { ScanItem it; it.branch = fr.init_branch; it.mut_idx = 0; it.pb = k_no_node; it.w = 0xffff | (0 << 16) | (del.n << 18); it.child_begin = 0; it.nch_vis = 0; it.sz = 0; it.pos = 0; items[0] = it; }
    __syncthreads();
    total = 1; nlev = 0;
    int lo = 0, hi = 1; bool overflow = false;
    while (lo < hi && !overflow) {
      if (nlev >= cap) { overflow = true; break; }
      if (lane == 0) lvl[nlev] = lo;
      ++nlev;
      int new_total = total;
      for (int chunk = lo; chunk < hi && !overflow; chunk += 64) {
        const int k = chunk + lane;
        const bool act = k < hi;
        int nch = 0, tb[3], tmi[3], branch = 0, mut_idx = 0, fs = 0, size = 0; bool visited = false;
        if (act) {
          const int4 a4 = *(const int4*)&items[k];
          branch = a4.x; mut_idx = a4.y;
          const int pb = a4.z, pmi = (int)(a4.w & 0xffff);   // (unsigned: see study_seed_fill)
          fs = (a4.w >> 16) & 3; size = (int)((uint32_t)a4.w >> 18);
          // move_to_neighbor (spr_study.cpp:43-91)
          if (pb != k_no_node && branch == pb) {
            const MutRec* m = muts_of(c, branch);
            const bool down = (mut_idx == pmi + 1);
            if (!down && mut_idx != pmi - 1) EMAT_FAIL(c, k_part_internal);
            const MutRec mm = m[down ? pmi : mut_idx];
            if (!iv_contains(miss.p, miss.n, mm.site)) {
              if (fs == 0) {
                const int new_from = down ? (int)mm.to : (int)mm.from;
                const int kk = sd_lower_bound(del.p, del.n, mm.site);
                const bool present = kk < del.n && del.p[kk].site == mm.site;
                size = del.n + (present ? (new_from == (int)del.p[kk].to ? -1 : 0) : +1);
              }
              fs += 1;
            }
          }
          visited = !(branch == X || fs > 1);
          if (visited) {   // seed_neighbors_except (spr_study.cpp:103-128), in push order
            const int nm = nmuts(c, branch);
            if (branch != root) {
              if (mut_idx > 0) { if (!(branch == pb && mut_idx - 1 == pmi)) { tb[nch] = branch; tmi[nch] = mut_idx - 1; ++nch; } }
              else { const int ub = nodes_of(c)[branch].parent, umi = nmuts(c, ub); if (!(ub == pb && umi == pmi)) { tb[nch] = ub; tmi[nch] = umi; ++nch; } }
            }
            if (mut_idx < nm) { if (!(branch == pb && mut_idx + 1 == pmi)) { tb[nch] = branch; tmi[nch] = mut_idx + 1; ++nch; } }
            else if (!is_tip(c, branch)) {
              const int c0 = nodes_of(c)[branch].child0, c1 = nodes_of(c)[branch].child1;
              if (!(c0 == pb && 0 == pmi)) { tb[nch] = c0; tmi[nch] = 0; ++nch; }
              if (!(c1 == pb && 0 == pmi)) { tb[nch] = c1; tmi[nch] = 0; ++nch; }
            }
          }
        }
        const uint64_t b0 = __ballot(act && (nch & 1)), b1 = __ballot(act && (nch & 2));
        const int off = __popcll(b0 & below) + 2 * __popcll(b1 & below);
        const int tot = __popcll(b0) + 2 * __popcll(b1);
        if (new_total + tot > cap) { overflow = true; break; }
        if (act) {
          const int w_own = (mut_idx & 0xffff) | (fs << 16) | (size << 18);
          for (int j = 0; j < nch; ++j) {
            ScanItem* ch = &items[new_total + off + j];
            *(int4*)ch = make_int4(tb[j], tmi[j], branch, w_own);
            *((int4*)ch + 1) = make_int4(0, 0, 0, 0);
          }
          items[k].w = w_own;
          *((int4*)&items[k] + 1) = make_int4(new_total + off, nch | (visited ? 256 : 0), 0, 0);
        }
        new_total += tot;
      }
      __syncthreads();
      lo = hi; hi = new_total; total = new_total;
    }
    done = !overflow;
  }
#ifdef EMAT_PROFILE_PHASES
  if (lane == 0) { int64_t* ex = (int64_t*)hdr_of(c)->reserved; ex[0] += 1; ex[1] += miss.n; ex[2] += del.n; ex[3] += total; ex[4] += nlev; ex[5] += (hbm_lo != nullptr) ? 1 : 0; ex[6] += done ? 0 : 1; ex[7] += in_block ? 0 : 1; }
#endif
  if (!done || c.failed) return false;
  // subtree sizes, deepest level first
  for (int l = nlev - 1; l >= 0; --l) {
    const int lo = lvl[l], hi = l + 1 < nlev ? lvl[l + 1] : total;
    for (int k = lo + lane; k < hi; k += 64) {
      const int4 b4 = *((const int4*)&items[k] + 1);
      int sz = 0;
      if (b4.y & 256) { sz = 1; for (int j = 0; j < (b4.y & 3); ++j) sz += items[b4.x + j].sz; }
      items[k].sz = sz;
    }
    __syncthreads();
  }
  // depth-first positions, top level first: a region, then its successors' subtrees from the last pushed to the first
  for (int l = 0; l < nlev; ++l) {
    const int lo = lvl[l], hi = l + 1 < nlev ? lvl[l + 1] : total;
    for (int k = lo + lane; k < hi; k += 64) {
      const int4 b4 = *((const int4*)&items[k] + 1);
      if (!(b4.y & 256)) continue;
      int running = b4.w + 1;
      for (int j = (b4.y & 3) - 1; j >= 0; --j) { const int szc = items[b4.x + j].sz; if (szc > 0) { items[b4.x + j].pos = running; running += szc; } }
    }
    __syncthreads();
  }
  const int V = items[0].sz;
  // the region records, in depth-first order
  SVec<Region>* shared = &fr.study.regions;
  if (lane == 0) {
    SVec<Region> r; r.n = 0; r.cap = 0; r.p = nullptr;
    if (hbm_lo != nullptr) {   // the items occupy the top of the HBM span: the regions take its bottom
      if ((size_t)V * sizeof(Region) + 16 <= (size_t)((uint8_t*)items - hbm_lo)) { r.p = (Region*)hbm_lo; r.cap = V; c.sc_top = (uint32_t)(hbm_lo - c.G) + (((uint32_t)V * (uint32_t)sizeof(Region) + 15u) & ~15u); }
      else EMAT_FAIL(c, k_part_overflow);
    } else r = sc_vec<Region>(c, V);
    *shared = r;
  }
  __syncthreads();
  if (c.failed) return true;   // nothing more to do: the move is abandoned
  Region* res = shared->p;
  for (int k = lane; k < total; k += 64) {
    const ScanItem it = items[k];
    if (!(it.nch_vis & 256)) continue;
    Region* r = &res[it.pos];
    *(int4*)r = make_int4(it.branch, it.mut_idx, (int)((uint32_t)it.w >> 18), 0);
    *(double2*)&r->t_min = make_double2(region_t_min(c, it.branch, it.mut_idx), region_t_max(c, it.branch, it.mut_idx));
  }
  __syncthreads();
  // account_for_Xs_detachment (spr_study.cpp:130-209) and remove_regions_in_Xs_future (:211-224): one read-modify-compact
  // pass, 64 regions at a time, the survivors keeping their order (ballot + prefix popcount)
  int w = 0;
  {
    const bool can_change_root = fr.can_change_root;
    const int P = nodes_of(c)[X].parent, S = sibling_of(c, P, X);
    const int nGP = nmuts(c, P);
    const int nS = (P == root) ? nmuts(c, S) : 0;
    const double t_min_P_end = (P != root) ? region_t_min(c, P, nGP) : 0.0;
    const double t_X = fr.t_X;
    for (int chunk = 0; chunk < V; chunk += 64) {
      const int i = chunk + lane;
      bool keep = false; int4 head = make_int4(0, 0, 0, 0); double2 times = make_double2(0.0, 0.0);
      if (i < V) {
        head = *(const int4*)&res[i]; times = *(const double2*)&res[i].t_min;
        int branch = head.x, mut_idx = head.y;
        keep = true;
        if (!can_change_root && branch == root) keep = false;
        else if (branch == S || branch == P) {
          if (P != root) {
            if (branch == S) { if (mut_idx == 0) times.x = t_min_P_end; mut_idx += nGP; }
            else { if (mut_idx == nGP) keep = false; else branch = S; }
          } else {
            if (!can_change_root) { if (branch == P) keep = false; }
            else {
              if (branch == S && mut_idx == nS) { mut_idx += nGP; times.x = k_neg_dbl_max; }
              else keep = false;
            }
          }
        }
        if (keep) { if (times.x >= t_X) keep = false; else { if (times.y > t_X) times.y = t_X; head.x = branch; head.y = mut_idx; } }
      }
      const uint64_t m = __ballot(keep);
      if (keep) { Region* r = &res[w + __popcll(m & below)]; *(int4*)r = head; *(double2*)&r->t_min = times; }
      w += __popcll(m);
    }
  }
  __syncthreads();
  if (lane == 0) {
    shared->n = w;
    sc_trim(c, *shared);
    c.bytes += (int64_t)(64 + 16) * V;
  }
  __syncthreads();
  return true;
}

// The wave-wide part of an SPR1 move: candidate scan (study_seed_fill) and study of the regions it found.
EMAT_DN void wave_scan_and_study(Ctx& c) {
  Spr1Frame& fr = *(Spr1Frame*)c.frame;
  const int lane = wave_lane();
  SVec<Region>* shared = &fr.study.regions;
#ifdef EMAT_PROFILE_PHASES
  const long long ph0 = clock64();
#endif
  bool scanned = false;
#ifndef EMAT_SERIAL_SCAN
  if (fr.limit == 1) scanned = wave_local_scan(c, fr);
#endif
  if (!scanned) {   // the 1 % of scans without a limit on the mutations crossed (and local ones that found no room): serial, on lane 0
#if defined(EMAT_PROFILE_PHASES) && defined(EMAT_X_UNLIMITED_SCANS)
    const long long un0 = clock64();
#endif
    if (lane == 0) *shared = study_seed_fill(c, fr.X, fr.t_X, fr.missing_at_X, fr.limit, fr.init_branch, 0, fr.deltas, fr.can_change_root, fr.hot);
    __syncthreads();
#if defined(EMAT_PROFILE_PHASES) && defined(EMAT_X_UNLIMITED_SCANS)
    if (lane == 0) { EMAT_COUNT(c, 11, 1); EMAT_COUNT(c, 12, clock64() - un0); }
#endif
  }
#ifdef EMAT_PROFILE_PHASES
  if (lane == 0) { hdr_of(c)->phase_ticks[6] += clock64() - ph0; hdr_of(c)->phase_ticks[13] += shared->n; }
#endif
  const SVec<Region> regions = *shared;
  wave_make_study(c, fr, regions);
}

}  // namespace EMAT_DEV_NS
}  // namespace emat
