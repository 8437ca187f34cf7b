// emat_utree_host.hpp -- SURVEY.md 8(f).4, second half: the reference's DEFAULT initial tree (build_initial_phylo_tree, core/utree.cpp:1892-1925;
// --v0-init-method mp_plus_timing, cmdline.cpp:113-115, 437) behind emat_tree_build_default.  Host C++, as in the reference: the method is a
// sequence of randomised local searches on an UNROOTED tree -- every tip is attached where it costs the fewest mutations, found by a
// best-first search from wherever the previous tip went in (branch and bound on the Fitch cost), the tree is rebuilt in nearest-first
// order until that stops helping, refined by subtree-prune-and-regraft moves, rooted by the regression of divergence on sampling date,
// and dated from the fitted rate -- and runs once per run in seconds where the UShER-like builder (emat_build.hpp) is quadratic.
// Included at the end of emat_backend.hip (after emat_build_host.hpp, whose closing passes and checks it shares).
//
// Layout.  The unrooted tree is struct-of-arrays: arcs come in mate pairs (a ^ 1), arc a holds its target and the site deltas
// origin -> target as a site-sorted vector (the reference: a hash map per arc); a node holds up to three arc slots and the arc that
// leads toward the FOCUS, the one node whose sequence is known explicitly (as deltas against the reference sequence).  Everything the
// reference iterates in hash order is iterated in ascending site order here.  Random numbers: the engine's Philox stream, one 64-bit
// draw per decision, in the reference's order of decisions -- so that the oracle's restatement (oracle/orc_utree.hpp, pinned to the
// reference's own tests/utree_tests.cpp) builds the same tree bit for bit from the same descriptors and seed.
#ifndef EMAT_UTREE_HOST_HPP_
#define EMAT_UTREE_HOST_HPP_

#include <queue>

namespace {
namespace ut {

struct Sd { int32_t site; uint8_t from, to; };                       // one site's change
// A list of them, ascending site, one entry per site.  Up to two entries live in the object itself: nine arcs in ten of a large tree
// carry at most two deltas, and a search that expands thousands of arcs per placement is bound by the cache misses of reaching them
// (24 bytes per arc, mates side by side, instead of a vector header that points somewhere else).
class Sdv {
  static constexpr uint32_t k_in = 2;
  union { Sd in_[k_in]; Sd* out_; };
  uint32_t n_ = 0, cap_ = k_in;
  Sd* data_() { return cap_ > k_in ? out_ : in_; }
  const Sd* data_() const { return cap_ > k_in ? out_ : in_; }
  void grow_(uint32_t want) {
    if (want <= cap_) return;
    uint32_t nc = cap_ * 2 > want ? cap_ * 2 : want;
    Sd* np = (Sd*)std::malloc(sizeof(Sd) * nc);
    if (!np) throw std::bad_alloc();
    std::memcpy(np, data_(), sizeof(Sd) * n_);
    if (cap_ > k_in) std::free(out_);
    out_ = np; cap_ = nc;
  }
 public:
  using iterator = Sd*; using const_iterator = const Sd*;
  Sdv() {}
  Sdv(const Sdv& o) { grow_(o.n_); std::memcpy(data_(), o.data_(), sizeof(Sd) * o.n_); n_ = o.n_; }
  Sdv(Sdv&& o) noexcept { std::memcpy((void*)this, (const void*)&o, sizeof(Sdv)); o.n_ = 0; o.cap_ = k_in; }
  Sdv& operator=(const Sdv& o) { if (this != &o) { n_ = 0; grow_(o.n_); std::memcpy(data_(), o.data_(), sizeof(Sd) * o.n_); n_ = o.n_; } return *this; }
  Sdv& operator=(Sdv&& o) noexcept { if (this != &o) { if (cap_ > k_in) std::free(out_); std::memcpy((void*)this, (const void*)&o, sizeof(Sdv)); o.n_ = 0; o.cap_ = k_in; } return *this; }
  ~Sdv() { if (cap_ > k_in) std::free(out_); }
  iterator begin() { return data_(); } iterator end() { return data_() + n_; }
  const_iterator begin() const { return data_(); } const_iterator end() const { return data_() + n_; }
  size_t size() const { return n_; } bool empty() const { return n_ == 0; }
  void clear() { n_ = 0; }
  void push_back(const Sd& d) { grow_(n_ + 1); data_()[n_++] = d; }
  iterator insert(iterator pos, const Sd& d) { const size_t k = (size_t)(pos - data_()); grow_(n_ + 1); Sd* p = data_(); std::memmove(p + k + 1, p + k, sizeof(Sd) * (n_ - k)); p[k] = d; ++n_; return p + k; }
  iterator erase(iterator pos) { Sd* p = data_(); const size_t k = (size_t)(pos - p); std::memmove(p + k, p + k + 1, sizeof(Sd) * (n_ - k - 1)); --n_; return p + k; }
};
static_assert(sizeof(Sdv) == 24, "an arc's deltas: 24 bytes");
inline Sdv::iterator sd_at(Sdv& v, int site) { return std::lower_bound(v.begin(), v.end(), site, [](const Sd& d, int s) { return d.site < s; }); }
inline Sdv::const_iterator sd_at(const Sdv& v, int site) { return std::lower_bound(v.begin(), v.end(), site, [](const Sd& d, int s) { return d.site < s; }); }
inline const Sd* sd_find(const Sdv& v, int site) { auto it = sd_at(v, site); return it != v.end() && it->site == site ? &*it : nullptr; }
inline void broken(const char* what) { throw std::runtime_error(std::string("inconsistent site deltas (") + what + ")"); }
// v := v, then (site: from -> to)            (site_deltas.h:88-110)
inline void sd_then(Sdv& v, int site, uint8_t from, uint8_t to) {
  auto it = sd_at(v, site);
  if (it == v.end() || it->site != site) { v.insert(it, Sd{site, from, to}); return; }
  if (it->to != from) broken("appending");
  it->to = to;
  if (it->from == it->to) v.erase(it);
}
// v := (site: from -> to), then v            (site_deltas.h:43-65)
inline void sd_first(Sdv& v, int site, uint8_t from, uint8_t to) {
  auto it = sd_at(v, site);
  if (it == v.end() || it->site != site) { v.insert(it, Sd{site, from, to}); return; }
  if (it->from != to) broken("prepending");
  it->from = from;
  if (it->from == it->to) v.erase(it);
}
inline void sd_set(Sdv& v, int site, uint8_t from, uint8_t to) { auto it = sd_at(v, site); if (it != v.end() && it->site == site) { it->from = from; it->to = to; } else v.insert(it, Sd{site, from, to}); }
inline void sd_erase(Sdv& v, int site) { auto it = sd_at(v, site); if (it != v.end() && it->site == site) v.erase(it); }
inline Sdv sd_reversed(const Sdv& v) { Sdv r(v); for (auto& d : r) std::swap(d.from, d.to); return r; }
inline bool ivs_subset(const BIvs& A, const BIvs& B) { return b_iv_subtract(A, B).empty(); }

constexpr int32_t k_none = -1;
struct Tips {                                                       // the descriptors, as the C-ABI hands them over
  const emat_tip_descs* td; const std::vector<uint8_t>* ref;
  int n() const { return td->num_tips; }
  BIvs missing(int i) const { BIvs v; for (int k = td->miss_offset[i]; k < td->miss_offset[i + 1]; ++k) v.push_back({td->miss_start[k], td->miss_end[k]}); return v; }
  int n_deltas(int i) const { return td->delta_offset[i + 1] - td->delta_offset[i]; }
  Sd delta(int i, int k) const { const int q = td->delta_offset[i] + k; const int l = td->delta_site[q]; return Sd{l, (*ref)[(size_t)l], td->delta_to[q]}; }
  double mid_date(int i) const { return (double)(td->t_min[i] + td->t_max[i]) / 2.0; }   // (the sum in float, as the reference's static_cast<double>(t_min + t_max) / 2.0)
};

struct Tree {
  std::vector<uint8_t> ref;
  BIvs missing_everywhere;                                           // sites no tip added so far has
  std::vector<int32_t> tgt; std::vector<Sdv> dl;                     // per arc (a free pair keeps the next free pair in tgt[even])
  std::vector<std::array<int32_t, 3>> adj; std::vector<int32_t> toward_focus;   // per node
  int32_t free_head = k_none, n_tips = 0, n_inner = 0, focus = k_none;
  Sdv ref_to_focus;

  void init(int tips) {                                             // utree.h:73-85
    n_tips = tips;
    const int nn = std::max(1, 2 * tips - 1), pairs = std::max(1, 2 * tips - 3 + 2);
    adj.assign((size_t)nn, {k_none, k_none, k_none}); toward_focus.assign((size_t)nn, k_none);
    tgt.assign((size_t)(2 * pairs), k_none); dl.assign((size_t)(2 * pairs), Sdv());
    for (int i = 0; i < 2 * pairs; i += 2) tgt[(size_t)i] = i + 2 < 2 * pairs ? i + 2 : k_none;
    free_head = 0;
  }
  static int mate(int a) { return a ^ 1; }
  int to(int a) const { return tgt[(size_t)a]; }
  int from(int a) const { return tgt[(size_t)mate(a)]; }
  int degree(int v) const { int d = 0; for (int a : adj[(size_t)v]) d += a != k_none; return d; }
  bool leaf(int v) const { return degree(v) == 1; }
  int arc_between(int u, int v) const { for (int a : adj[(size_t)u]) if (a != k_none && to(a) == v) return a; return k_none; }
  int weight(int a) const { return (int)dl[(size_t)a].size(); }
  int total_weight() const { int s = 0; for (size_t a = 0; a < dl.size(); a += 2) s += (int)dl[a].size(); return s; }
  int grab_pair() { if (free_head == k_none) throw std::runtime_error("arc pool exhausted"); const int b = free_head; free_head = tgt[(size_t)b]; tgt[(size_t)b] = tgt[(size_t)b + 1] = k_none; return b; }
  void drop_pair(int a) { const int b = a & ~1; dl[(size_t)b].clear(); dl[(size_t)b + 1].clear(); tgt[(size_t)b] = free_head; free_head = b; }
  void plug(int v, int a) { for (int& s : adj[(size_t)v]) if (s == k_none) { s = a; return; } throw std::runtime_error("a node with four neighbours"); }
  void swap_slot(int v, int old_arc, int new_arc) { for (int& s : adj[(size_t)v]) if (s == old_arc) { s = new_arc; return; } }
  int link(int u, int v) { const int b = grab_pair(); tgt[(size_t)b] = v; tgt[(size_t)b + 1] = u; plug(u, b); plug(v, b + 1); return b; }   // utree.h:164-181
  int random_tip(HostRng& g) const { return (int)(((unsigned __int128)g.next64() * (uint64_t)n_tips) >> 64); }
  int random_node(HostRng& g) const { const int nn = n_tips + n_inner; int v; do { v = (int)(((unsigned __int128)g.next64() * (uint64_t)nn) >> 64); } while (degree(v) == 0); return v; }

  // Depth-first tour of the arcs from `src` (utree.h:337-360): every arc is reported once going in (enter) and its mate once coming
  // back (leave); the slots of a node are taken last first.
  template <class Enter, class Leave> void tour(int src, Enter enter, Leave leave) const {
    std::vector<int32_t> st;                                         // arc << 1 | leaving
    auto fan = [&](int v, int except) { for (int a : adj[(size_t)v]) if (a != k_none && a != except) { st.push_back((mate(a) << 1) | 1); st.push_back(a << 1); } };
    fan(src, k_none);
    while (!st.empty()) {
      const int w = st.back(); st.pop_back();
      const int a = w >> 1;
      if (w & 1) leave(a); else { enter(a); fan(to(a), mate(a)); }
    }
  }
  void refocus(int f) { focus = f; toward_focus[(size_t)f] = k_none; tour(f, [&](int a) { toward_focus[(size_t)to(a)] = mate(a); }, [](int) {}); }   // utree.cpp:9-17
  // The focus walks to `v` (utree.h:395-428): the arcs of the path are turned round, then crossed one by one; `before(arc)` sees every
  // arc before its deltas go into ref_to_focus.
  template <class Before> void walk_focus_to(int v, Before before) {
    if (v == focus) return;
    int carry = toward_focus[(size_t)v]; toward_focus[(size_t)v] = k_none;
    while (carry != k_none) { const int nx = to(carry), keep = toward_focus[(size_t)nx]; toward_focus[(size_t)nx] = mate(carry); carry = keep; }
    for (int cur = focus; cur != v;) {
      const int a = toward_focus[(size_t)cur];
      before(a);
      for (const Sd& d : dl[(size_t)a]) sd_then(ref_to_focus, d.site, d.from, d.to);
      cur = to(a);
    }
    focus = v;
  }
  void walk_focus_to(int v) { walk_focus_to(v, [](int) {}); }
  // A new node M on the edge of arc ab (utree.h:431-487); side(delta) says which half gets a delta (true: the half next to the origin)
  template <class Side> void split(int ab, int M, Side near_origin) {
    const int ba = mate(ab), A = from(ab), B = to(ab);
    const int am = grab_pair(), ma = am + 1, mb = grab_pair(), bm = mb + 1;
    tgt[(size_t)am] = M; tgt[(size_t)ma] = A; tgt[(size_t)mb] = B; tgt[(size_t)bm] = M;
    const Sdv old = dl[(size_t)ab];
    for (const Sd& d : old) {
      if (near_origin(d)) { dl[(size_t)am].push_back(d); dl[(size_t)ma].push_back(Sd{d.site, d.to, d.from}); }
      else { dl[(size_t)mb].push_back(d); dl[(size_t)bm].push_back(Sd{d.site, d.to, d.from}); }
    }
    adj[(size_t)M] = {ma, mb, k_none};
    swap_slot(A, ab, am); swap_slot(B, ba, bm);
    if (toward_focus[(size_t)A] == ab) { toward_focus[(size_t)A] = am; toward_focus[(size_t)M] = mb; }
    if (toward_focus[(size_t)B] == ba) { toward_focus[(size_t)B] = bm; toward_focus[(size_t)M] = ma; }
    drop_pair(ab);
  }
  int unhook_leaf(int X) {                                          // utree.cpp:19-45
    int xm = k_none; for (int a : adj[(size_t)X]) if (a != k_none) { xm = a; break; }
    const int M = to(xm);
    swap_slot(M, mate(xm), k_none);
    adj[(size_t)X] = {k_none, k_none, k_none}; toward_focus[(size_t)X] = k_none;
    drop_pair(xm);
    return M;
  }
  int splice_out(int M) {                                           // utree.cpp:47-113: M has two neighbours left and is not the focus
    int ma = k_none, mb = k_none; for (int a : adj[(size_t)M]) if (a != k_none) { if (ma == k_none) ma = a; else mb = a; }
    const int A = to(ma), B = to(mb);
    Sdv a_to_b = dl[(size_t)mate(ma)];
    for (const Sd& d : dl[(size_t)mb]) sd_then(a_to_b, d.site, d.from, d.to);
    const int ab = grab_pair(), ba = ab + 1;
    tgt[(size_t)ab] = B; tgt[(size_t)ba] = A;
    dl[(size_t)ba] = sd_reversed(a_to_b); dl[(size_t)ab] = std::move(a_to_b);
    const int am = arc_between(A, M), bm = arc_between(B, M);
    swap_slot(A, am, ab); swap_slot(B, bm, ba);
    if (toward_focus[(size_t)A] == am) toward_focus[(size_t)A] = ab;
    if (toward_focus[(size_t)B] == bm) toward_focus[(size_t)B] = ba;
    adj[(size_t)M] = {k_none, k_none, k_none}; toward_focus[(size_t)M] = k_none;
    drop_pair(ma); drop_pair(mb);
    return ab;
  }
  void cut(int u, int v) {                                          // utree.cpp:115-127
    const int uv = arc_between(u, v), vu = mate(uv);
    swap_slot(u, uv, k_none); if (toward_focus[(size_t)u] == uv) toward_focus[(size_t)u] = k_none;
    swap_slot(v, vu, k_none); if (toward_focus[(size_t)v] == vu) toward_focus[(size_t)v] = k_none;
    drop_pair(uv);
  }
};

// What the piece to attach (a tip, or the root of a pruned subtree) may be at every site, RELATIVE to the focus's sequence
// (utree.cpp:140-170): one state (`fixed`: only where it differs from the focus), two or three (`open`: a bit mask), or anything (`any`).
struct Fitch {
  Sdv fixed; std::vector<std::pair<int32_t, uint8_t>> open; BIvs any;
  void clear() { fixed.clear(); open.clear(); any.clear(); }
  const uint8_t* mask_at(int site) const { auto it = std::lower_bound(open.begin(), open.end(), site, [](const std::pair<int32_t, uint8_t>& p, int s) { return p.first < s; }); return it != open.end() && it->first == site ? &it->second : nullptr; }
  bool allows(int site, uint8_t state, uint8_t focus_state) const {
    if (b_iv_contains(any, site)) return true;
    if (const uint8_t* m = mask_at(site)) return (*m >> state) & 1;
    if (const Sd* d = sd_find(fixed, site)) return state == d->to;
    return state == focus_state;
  }
  void focus_changed(int site, uint8_t was, uint8_t is) { if (b_iv_contains(any, site) || mask_at(site)) return; sd_first(fixed, site, is, was); }   // (pop_front of was -> is)
  // The same sets in ABSOLUTE terms, frozen when they are set up: where the piece's state is known (`fixed` as it was then: its `to`
  // states), else the state of the node that was the focus then (`then_focus`: that node's deltas against the reference sequence), else
  // the reference sequence.  With these a search can price every arc it expands from where the focus stands, instead of walking the
  // focus to each arc in best-first order as the reference does (at 30 000 tips the focus crossed 3.8 arcs per arc expanded): the
  // costs, the order of expansion and the list of ties are the same numbers in the same order.
  Sdv then_fixed, then_focus;
  void freeze(const Sdv& ref_to_focus) { then_fixed = fixed; then_focus = ref_to_focus; }
  bool allows_abs(int site, uint8_t state, const std::vector<uint8_t>& ref) const {
    if (b_iv_contains(any, site)) return true;
    if (const uint8_t* m = mask_at(site)) return (*m >> state) & 1;
    if (const Sd* d = sd_find(then_fixed, site)) return state == d->to;
    if (const Sd* d = sd_find(then_focus, site)) return state == d->to;
    return state == ref[(size_t)site];
  }
};

struct Builder {                                                    // utree.cpp:190-739
  const Tips& tips; HostRng& rng; Tree T;
  int placed = 0, L = 0; double sqrt_6L = 0.0;
  Fitch fx; int cost_here = 0;                                       // ... and what attaching AT the focus would cost
  Sdv m_to_x; std::vector<std::pair<int32_t, uint8_t>> m_state;     // the new joint M: its deltas to the piece, and where it differs from the focus
  std::vector<uint64_t> heap64; std::vector<int> ties, cost_at, across;
  std::vector<uint8_t> ref_mask, abs_mask;                           // per site: 1 << the reference's state; and the states the piece allows there, laid out once per search (best_arc)
  int give_up_after = 0; std::vector<int> dfs, component;
  long long n_pops = 0, n_crossed = 0, n_searches = 0;               // (EMAT_VERBOSE: arcs the searches expanded, arcs the focus crossed)

  Builder(const Tips& t, HostRng& g) : tips(t), rng(g) { T.init(t.n()); T.ref = *t.ref; setup(); }
  Builder(Tree&& tree, const Tips& t, HostRng& g) : tips(t), rng(g), T(std::move(tree)) { placed = T.n_tips; setup(); }
  void setup() { L = (int)T.ref.size(); sqrt_6L = std::sqrt(6.0 * L); if (L >= (1 << 22)) throw std::runtime_error("the default builder's search keys hold costs below 2^22: genome too long"); ref_mask.resize((size_t)L); for (int l = 0; l < L; ++l) ref_mask[(size_t)l] = (uint8_t)(1u << T.ref[(size_t)l]); abs_mask.resize((size_t)L); }
  int new_inner() { return T.n_tips + T.n_inner++; }
  int slack(int cost) const { const double s = cost / sqrt_6L; return std::clamp((int)std::ceil(10.0 * s * (s + 5)), 2, L); }   // :267-271
  uint8_t focus_state(int site) const { const Sd* d = sd_find(T.ref_to_focus, site); return d ? d->to : T.ref[(size_t)site]; }

  void add(int X) {                                                 // :221-246
    if (placed == 0) {
      T.focus = X; T.missing_everywhere = tips.missing(X);
      for (int k = 0; k < tips.n_deltas(X); ++k) T.ref_to_focus.push_back(tips.delta(X, k));
    } else {
      { // a site that X is the first to have takes X's state everywhere (:294-312)
        const BIvs mx = tips.missing(X);
        if (!ivs_subset(T.missing_everywhere, mx)) {
          const BIvs was = std::move(T.missing_everywhere);
          T.missing_everywhere = b_iv_intersect(was, mx);
          for (int k = 0; k < tips.n_deltas(X); ++k) { const Sd d = tips.delta(X, k); if (b_iv_contains(was, d.site) && !b_iv_contains(T.missing_everywhere, d.site)) sd_then(T.ref_to_focus, d.site, d.from, d.to); }
        }
      }
      fitch_of_tip(X);
      const int best = best_arc().first;
      if (best == k_none) {                                           // the second tip: straight onto the first (:489-498)
        const int a = T.link(T.focus, X);
        T.dl[(size_t)a] = fx.fixed; T.dl[(size_t)Tree::mate(a)] = sd_reversed(fx.fixed);
        T.toward_focus[(size_t)X] = Tree::mate(a);
      } else { walk(T.from(best)); const int M = new_inner(); open_edge(best, M); hang(M, X); }
    }
    ++placed;
  }
  void fitch_of_tip(int X) {                                        // :326-344
    fx.clear();
    fx.any = tips.missing(X);
    for (int k = 0; k < tips.n_deltas(X); ++k) fx.fixed.push_back(tips.delta(X, k));
    for (const Sd& d : T.ref_to_focus) if (!b_iv_contains(fx.any, d.site)) sd_first(fx.fixed, d.site, d.to, d.from);
    cost_here = (int)fx.fixed.size();
    fx.freeze(T.ref_to_focus);
  }
  void fitch_of_subtree(int X) {                                    // :351-415: the focus is X's neighbour M; X's other neighbours D, E
    fx.clear(); cost_here = 0;
    const int M = T.focus, mx = T.arc_between(M, X);
    int xd = k_none, xe = k_none; for (int a : T.adj[(size_t)X]) if (a != k_none && T.to(a) != M) { if (xd == k_none) xd = a; else xe = a; }
    const int D = T.to(xd), E = T.to(xe);
    const bool d_leaf = T.leaf(D), e_leaf = T.leaf(E);
    const BIvs miss_d = d_leaf ? tips.missing(D) : BIvs(), miss_e = e_leaf ? tips.missing(E) : BIvs();
    if (d_leaf && e_leaf) fx.any = b_iv_intersect(miss_d, miss_e);
    struct Tri { int32_t site; uint8_t m, d, e; };
    std::vector<Tri> tri;                                             // states at M, D, E wherever any of the three arcs has a delta
    auto at = [&](int site) { return std::lower_bound(tri.begin(), tri.end(), site, [](const Tri& t, int s) { return t.site < s; }); };
    for (const Sd& q : T.dl[(size_t)mx]) tri.push_back(Tri{q.site, q.from, q.to, q.to});
    for (const Sd& q : T.dl[(size_t)xd]) { auto it = at(q.site); if (it != tri.end() && it->site == q.site) it->d = q.to; else tri.insert(it, Tri{q.site, q.from, q.to, q.from}); }
    for (const Sd& q : T.dl[(size_t)xe]) { auto it = at(q.site); if (it != tri.end() && it->site == q.site) it->e = q.to; else tri.insert(it, Tri{q.site, q.from, q.from, q.to}); }
    for (const Tri& t : tri) {
      if (b_iv_contains(fx.any, t.site)) continue;
      const bool no_d = d_leaf && b_iv_contains(miss_d, t.site), no_e = e_leaf && b_iv_contains(miss_e, t.site);
      if (no_d || no_e || t.d == t.e) { const uint8_t f = no_d ? t.e : t.d; if (t.m != f) { fx.fixed.push_back(Sd{t.site, t.m, f}); ++cost_here; } }
      else { fx.open.push_back({t.site, (uint8_t)((1u << t.d) | (1u << t.e))}); if (t.m != t.d && t.m != t.e) ++cost_here; }
    }
    fx.freeze(T.ref_to_focus);
  }
  int cost_on(int a) const {                                        // attaching in the middle of focal arc a (:708-718)
    int saved = 0;
    for (const Sd& d : T.dl[(size_t)a]) if (!fx.allows(d.site, d.from, d.from) && fx.allows(d.site, d.to, d.from)) ++saved;
    return cost_here - saved;
  }
  void walk(int v) {                                                // the focus moves, the Fitch sets follow (:648-657)
    T.walk_focus_to(v, [&](int a) {
      ++n_crossed;
      for (const Sd& d : T.dl[(size_t)a]) { cost_here += (int)fx.allows(d.site, d.from, d.from) - (int)fx.allows(d.site, d.to, d.from); fx.focus_changed(d.site, d.from, d.to); }
    });
  }
  std::pair<int, int> best_arc() {                                  // best-first over the arcs around the focus (:421-482); the focus stays where it is
    int best = cost_here; ties.clear(); ++n_searches;
    auto note = [&](int c, int a) { if (c < best) { best = c; ties.clear(); } if (c == best) ties.push_back(a); };
    // Fitch::allows_abs for every site at once: the mask of states the piece allows, in that function's order of precedence (anything where the piece
    // has no data, else its two or three states, else its known state, else the state of the node that was the focus when the sets were made, else the
    // reference's).  30 KB laid out once per search (a microsecond) instead of four binary searches per delta priced: the guide tree and the rebuilds,
    // whose searches price many deltas, take half the time (20 000 tips of C4: 1.7 -> 0.9 s); the refinement's searches mostly cross arcs without deltas (- 10 %).
    std::memcpy(abs_mask.data(), ref_mask.data(), (size_t)L);
    for (const Sd& d : fx.then_focus) abs_mask[(size_t)d.site] = (uint8_t)(1u << d.to);
    for (const Sd& d : fx.then_fixed) abs_mask[(size_t)d.site] = (uint8_t)(1u << d.to);
    for (const auto& o : fx.open) abs_mask[(size_t)o.first] = o.second;
    for (const auto& iv : fx.any) std::memset(abs_mask.data() + iv.first, 0xF, (size_t)(iv.second - iv.first));
    const uint8_t* const am = abs_mask.data();
    if (cost_at.size() < T.adj.size()) cost_at.resize(T.adj.size());   // what attaching AT a node would cost, for the nodes the search has reached (a tree: each once)
    if (across.size() < T.tgt.size()) across.resize(T.tgt.size());
    // The queue: (cost, arc) pairs in ascending order, as the reference's heap of pairs pops them -- every arc enters once, so the order of popping is
    // the order of the keys whatever the container: one 64-bit key per entry, kept COMPLEMENTED so that the entry to pop is the largest.  While the
    // frontier is small (it is ~ 100 entries on average at 20 000 tips) the entries are simply kept sorted, the next one at the back: a pop is a
    // pop_back and a push a branch-free binary search + a short memmove, where a binary heap's pop is seven levels of unpredictable branches (measured:
    // 3/4 of the refinement's time was this loop, ~ 160 cycles per pop).  A frontier that outgrows k_sorted_max turns into a heap for the rest of the
    // search: an ascending array, reversed, is a max-heap.  And the loop is bound by the cache misses of reaching the arcs it expands, so a pop reads
    // as little as it can and asks for it ahead of time: the low bits of a key carry what the pop needs to know about where its arc came from (below
    // the bits that decide the order), the far end's three slots are asked for when an arc is queued, the deltas of the arcs beyond when it is next in
    // line.  Together: 60 000 tips of C4 in 151-161 s instead of 212-216 (same tree, digest for digest: scripts/default_builder_probe.py).
    constexpr int64_t k_bias = 1 << 21;
#ifndef EMAT_UT_SORTED_MAX           // (-DEMAT_UT_SORTED_MAX=3 -DEMAT_UT_CARRY_BIAS=1 -DEMAT_UT_CARRY_NONE=3: a test build in which small inputs take the heap and the fall-back too)
#define EMAT_UT_SORTED_MAX 384
#define EMAT_UT_CARRY_BIAS 128
#define EMAT_UT_CARRY_NONE 511
#endif
    constexpr size_t k_sorted_max = EMAT_UT_SORTED_MAX;
    constexpr int k_carry_bias = EMAT_UT_CARRY_BIAS, k_carry_none = EMAT_UT_CARRY_NONE;   // low 9 bits of a key: (cost AT the arc's far end) - (cost ON the arc) + 128, so that a pop reads nothing about where its arc came from; 511: does not fit, cost_at / across hold it
    auto key = [&](int c, int a, int carry) { return ~((uint64_t)((int64_t)c + k_bias) << 41 | (uint64_t)(uint32_t)a << 9 | (uint64_t)carry); };
    // one look at an arc's deltas gives both what attaching on it would save and how the cost changes across it; most arcs of a large tree carry no delta at all
    auto reach = [&](int a, int origin, int c_origin) {
      int saved = 0, shift = 0;
      for (const Sd& d : T.dl[(size_t)a]) { const unsigned m = am[(size_t)d.site]; const bool f = (m >> d.from) & 1u, t = (m >> d.to) & 1u; saved += (!f && t); shift += (int)f - (int)t; }
      const int c = c_origin - saved;
      note(c, a);
      int carry = saved + shift + k_carry_bias;
      if (carry < 0 || carry >= k_carry_none) { carry = k_carry_none; cost_at[(size_t)origin] = c_origin; across[(size_t)a] = shift; }
      __builtin_prefetch(&T.adj[(size_t)T.tgt[(size_t)a]]);
      return key(c, a, carry);
    };
    heap64.clear();
    bool sorted_mode = true;
    int thr_of = std::numeric_limits<int>::min(), thr = 0;
    auto push = [&](uint64_t k) {
      if (sorted_mode) {
        if (heap64.size() < k_sorted_max) {
          const size_t n = heap64.size(); heap64.push_back(k);
          // the first position whose entry is > k (none is equal): a new arc costs about what the arc it hangs from did, which was the next in line a
          // moment ago -- measured at 60 000 tips: 2.9 entries from the back on average in a frontier of 170 -- so look there first, then bisect the rest
          uint64_t* const p = heap64.data(); size_t lo = n;
          for (int look = 0; look < 6 && lo > 0 && p[lo - 1] > k; ++look) --lo;
          if (lo > 0 && p[lo - 1] > k) {
            size_t len = lo; lo = 0;
            while (len > 0) { const size_t half = len >> 1; const bool go = p[lo + half] < k; lo = go ? lo + half + 1 : lo; len = go ? len - half - 1 : half; }
          }
          std::memmove(p + lo + 1, p + lo, (n - lo) * sizeof(uint64_t)); p[lo] = k;
          return;
        }
        std::reverse(heap64.begin(), heap64.end()); sorted_mode = false;
      }
      heap64.push_back(k); std::push_heap(heap64.begin(), heap64.end());
    };
    auto pop = [&]() { if (!sorted_mode) std::pop_heap(heap64.begin(), heap64.end()); const uint64_t k = ~heap64.back(); heap64.pop_back(); return k; };
    for (int a : T.adj[(size_t)T.focus]) if (a != k_none) push(reach(a, T.focus, cost_here));
    while (!heap64.empty()) {
      const uint64_t top = pop(); ++n_pops;
      if (!heap64.empty()) {   // (what is next in line now is most likely the next one popped: its far end's slots were asked for when it was queued; now the deltas beyond)
        const int v2 = T.tgt[(size_t)(uint32_t)(~(sorted_mode ? heap64.back() : heap64.front()) >> 9)];
        for (int a : T.adj[(size_t)v2]) if (a != k_none) __builtin_prefetch(&T.dl[(size_t)a]);
      }
      const int c_in = (int)((int64_t)(top >> 41) - k_bias), a_in = (int)(uint32_t)(top >> 9), carry = (int)(top & 511u);
      if (best != thr_of) { thr_of = best; thr = best + slack(best); }   // (a division and a ceil(): once per improvement, not once per pop)
      if (c_in > thr) break;
      const int c_v = carry != k_carry_none ? c_in + carry - k_carry_bias : cost_at[(size_t)T.from(a_in)] + across[(size_t)a_in];
      const int v = T.to(a_in);
      for (int a : T.adj[(size_t)v]) if (a != k_none && a != Tree::mate(a_in)) push(reach(a, v, c_v));
    }
    if (ties.empty()) return {k_none, best};
    return {ties[(size_t)(((unsigned __int128)rng.next64() * (uint64_t)ties.size()) >> 64)], best};
  }
  bool coin() { return (rng.next64() >> 63) != 0; }
  uint8_t joint_state(int site) const { auto it = std::lower_bound(m_state.begin(), m_state.end(), site, [](const std::pair<int32_t, uint8_t>& p, int s) { return p.first < s; }); return it != m_state.end() && it->first == site ? it->second : focus_state(site); }
  void open_edge(int a, int M) {                                    // M goes onto focal arc a, its sequence chosen delta by delta (:512-547)
    m_to_x = fx.fixed; m_state.clear();
    T.split(a, M, [&](const Sd& d) -> bool {
      auto take = [&]() { if (!fx.mask_at(d.site)) sd_first(m_to_x, d.site, d.to, d.from); m_state.push_back({d.site, d.to}); };   // (the deltas come in site order)
      if (b_iv_contains(fx.any, d.site)) return coin();
      if (fx.allows(d.site, d.to, d.from)) { take(); return true; }
      if (fx.allows(d.site, d.from, d.from)) return false;
      const bool near = coin(); if (near) take(); return near;
    });
  }
  void hang(int M, int X) { const int a = T.link(M, X); T.dl[(size_t)a] = m_to_x; T.dl[(size_t)Tree::mate(a)] = sd_reversed(m_to_x); T.toward_focus[(size_t)X] = Tree::mate(a); }   // :550-557
  void hang_subtree(int X, int a, int M, int de) {                  // :569-608
    open_edge(a, M);
    const int D = T.from(de), E = T.to(de);
    T.split(de, X, [&](const Sd& q) -> bool {
      const uint8_t m = joint_state(q.site);
      if (m == q.to) return true;
      if (m != q.from) sd_set(m_to_x, q.site, m, q.from);
      return false;
    });
    hang(M, X);
    T.toward_focus[(size_t)D] = T.arc_between(D, X); T.toward_focus[(size_t)E] = T.arc_between(E, X);
  }
  void forget_unseen(int a) {                                       // a merged edge must not claim changes at sites its leaf lacks (:619-643)
    for (int v : {T.from(a), T.to(a)}) {
      if (!T.leaf(v)) continue;
      const BIvs miss = tips.missing(v);
      const int out = T.from(a) == v ? a : Tree::mate(a);
      Sdv gone; for (const Sd& d : T.dl[(size_t)out]) if (b_iv_contains(miss, d.site)) gone.push_back(d);
      for (const Sd& d : gone) {
        if (v == T.focus) { cost_here += (int)fx.allows(d.site, d.from, d.from) - (int)fx.allows(d.site, d.to, d.from); fx.focus_changed(d.site, d.from, d.to); sd_then(T.ref_to_focus, d.site, d.from, d.to); }
        sd_erase(T.dl[(size_t)out], d.site); sd_erase(T.dl[(size_t)Tree::mate(out)], d.site);
      }
    }
  }
  int random_node_of_component(int sink) {                          // :676-702
    component.clear(); dfs.clear();
    for (int a : T.adj[(size_t)sink]) if (a != k_none) dfs.push_back(T.to(a));
    while (!dfs.empty()) {
      const int v = dfs.back(); dfs.pop_back(); component.push_back(v);
      if ((int)component.size() > give_up_after)
        for (;;) { const int s = T.random_node(rng); int c = s; while (T.toward_focus[(size_t)c] != k_none) c = T.to(T.toward_focus[(size_t)c]); if (c == sink) return s; }
      for (int a : T.adj[(size_t)v]) if (a != k_none && a != T.toward_focus[(size_t)v]) dfs.push_back(T.to(a));
    }
    return component[(size_t)(((unsigned __int128)rng.next64() * (uint64_t)component.size()) >> 64)];
  }
};

inline Tree guide_tree(const Tips& tips, HostRng& rng) { Builder b(tips, rng); for (int k = 0; k < tips.n(); ++k) b.add(k); return std::move(b.T); }   // :744-755

// Tips in the order "nearest to what is already there first", each with the closest earlier tip (utree.cpp:761-896)
template <class Visit> void nearest_first(const Tree& G, HostRng& rng, Visit visit) {
  struct Near { int tip = k_none, dist = 0; };
  std::vector<Near> near(G.tgt.size());                                // per arc: the nearest tip beyond it
  auto best_beyond = [&](int v, int except) { Near b{k_none, std::numeric_limits<int>::max()}; for (int a : G.adj[(size_t)v]) if (a != k_none && a != except && near[(size_t)a].dist < b.dist) b = near[(size_t)a]; return b; };
  G.tour(0, [](int) {}, [&](int up) { const int X = G.from(up), down = Tree::mate(up); const Near b = best_beyond(X, up); near[(size_t)down] = b.tip == k_none ? Near{X, G.weight(down)} : Near{b.tip, G.weight(down) + b.dist}; });
  G.tour(0, [&](int down) { const int P = G.from(down), up = Tree::mate(down); const Near b = best_beyond(P, down); near[(size_t)up] = b.tip == k_none ? Near{P, G.weight(up)} : Near{b.tip, G.weight(up) + b.dist}; }, [](int) {});
  struct Item { int dist, arc, prev_tip, d_prev; bool operator>(const Item& o) const { return dist > o.dist; } };
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> pq;
  const int S = G.random_tip(rng);
  visit(S, k_none);
  for (int a : G.adj[(size_t)S]) if (a != k_none) pq.push({near[(size_t)a].dist, a, S, 0});
  while (!pq.empty()) {
    const Item it = pq.top(); pq.pop();
    const int Tt = near[(size_t)it.arc].tip;
    visit(Tt, it.prev_tip);
    int into = it.arc, v = G.to(it.arc), d_v = G.weight(it.arc);
    while (v != Tt) {                                                   // along the way to T, the side branches now have T or the earlier tip nearest
      const int d_prev = d_v + it.d_prev, d_new = it.dist - d_v;
      const int who = d_new <= d_prev ? Tt : it.prev_tip, how_far = d_new <= d_prev ? d_new : d_prev;
      int on = k_none;
      for (int a : G.adj[(size_t)v]) { if (a == k_none || a == Tree::mate(into)) continue; if (near[(size_t)a].tip == Tt) on = a; else pq.push({near[(size_t)a].dist, a, who, how_far}); }
      into = on; v = G.to(on); d_v += G.weight(on);
    }
  }
}
inline Tree rebuilt_nearest_first(const Tree& G, const Tips& tips, HostRng& rng) {   // :898-914
  Builder b(tips, rng);
  nearest_first(G, rng, [&](int tip, int prev) { if (prev != k_none) b.T.walk_focus_to(prev); b.add(tip); });
  return std::move(b.T);
}

inline void spr_refine(Tree& tree, const Tips& tips, HostRng& rng) {   // utree.cpp:920-1081
  const int N = tree.n_tips;
  if (N <= 2) return;
  Builder b(std::move(tree), tips, rng);
  Tree& T = b.T;
  { const int nn = 2 * N - 1; b.give_up_after = (int)std::sqrt((double)nn * std::log2((double)nn)); }
  int idle = 0;
  for (int attempt = 0; attempt < 30 * N; ++attempt) {
    int M; do { M = T.random_node(rng); } while (T.degree(M) != 3);
    const auto slots = T.adj[(size_t)M];
    const int mx = slots[(size_t)(((unsigned __int128)rng.next64() * 3u) >> 64)];
    const int X = T.to(mx);
    int mp = k_none, mq = k_none; for (int a : slots) if (a != mx) { if (mp == k_none) mp = a; else mq = a; }
    const int P = T.to(mp);
    const int w_mx = T.weight(mx), w_mp = T.weight(mp), w_mq = T.weight(mq);
    int before = 0, best_cost = 0, best = k_none;
    if (T.leaf(X)) {
      if (T.focus == X) T.walk_focus_to(M);
      T.unhook_leaf(X);
      if (T.focus == M) T.walk_focus_to(P);
      const int pq = T.splice_out(M);
      T.walk_focus_to(T.from(pq));
      b.fitch_of_tip(X);
      b.forget_unseen(pq);
      before = w_mx + w_mp + w_mq - T.weight(pq);
      best = pq; best_cost = b.cost_on(pq);
      if (best_cost >= before) {
        int S; do { S = T.random_node(rng); } while (S == X);
        b.walk(S);
        const auto [a, c] = b.best_arc();
        if (c < best_cost) { best = a; best_cost = c; }
      }
      b.walk(T.from(best));
      b.open_edge(best, M); b.hang(M, X);
    } else {
      const int xm = Tree::mate(mx);
      int xd = k_none, xe = k_none; for (int a : T.adj[(size_t)X]) if (a != k_none && a != xm) { if (xd == k_none) xd = a; else xe = a; }
      const int D = T.to(xd);
      const int w_xd = T.weight(xd), w_xe = T.weight(xe);
      T.walk_focus_to(M);
      b.fitch_of_subtree(X);
      T.cut(M, X);
      T.toward_focus[(size_t)D] = k_none; T.toward_focus[(size_t)X] = xd;   // X's side of the cut is a tree of its own, hanging from D
      const int de = T.splice_out(X);
      b.forget_unseen(de);
      b.walk(P);
      const int pq = T.splice_out(M);
      b.forget_unseen(pq);
      before = w_mx + w_mp + w_mq + w_xd + w_xe - T.weight(pq) - T.weight(de);
      b.walk(P);
      best = pq; best_cost = b.cost_on(pq);
      if (best_cost >= before) {
        const int S = b.random_node_of_component(P);
        b.walk(S);
        const auto [a, c] = b.best_arc();
        if (c < best_cost) { best = a; best_cost = c; }
      }
      b.walk(T.from(best));
      b.hang_subtree(X, best, M, de);
    }
    idle = best_cost - before < 0 ? 0 : idle + 1;
    if (idle >= N) { if (verbose_reports()) fprintf(stderr, "[emat] spr_refine: stopped after %d attempts (%d without improvement)\n", attempt + 1, idle); break; }
  }
  if (verbose_reports()) fprintf(stderr, "[emat] spr_refine: %lld searches expanded %lld arcs, the focus crossed %lld arcs\n", b.n_searches, b.n_pops, b.n_crossed);
  tree = std::move(T);
}

struct Rooting { int root; bool by_regression; double r2, rate, t_root; };
inline std::pair<int, int> farthest(const Tree& T, int start) {     // utree.cpp:1085-1103
  int best = start, best_d = 0, d = 0;
  T.tour(start, [&](int a) { d += T.weight(a); if (d >= best_d) { best_d = d; best = T.to(a); } }, [&](int a) { d -= T.weight(a); });
  return {best, best_d};
}
// The root goes k deltas along arc `a` from its origin: a new node that splits the edge, the first k deltas (in site order) on the origin's side
inline int root_on(Tree& T, int a, int k) { const int R = T.n_tips + T.n_inner++; int given = 0; T.split(a, R, [&](const Sd&) { return given++ < k; }); return R; }
inline Rooting midpoint_root(Tree& T, const Tips& tips) {            // utree.cpp:1122-1248
  const int N = T.n_tips; const double fallback = 1.0 / 30.0;
  const int u = farthest(T, 0).first;
  const auto [v, D] = farthest(T, u);
  const double t_u = tips.mid_date(u), t_v = tips.mid_date(v), Dd = (double)D;
  const double t_R = std::min((t_u + t_v) / 2.0 - Dd / (2.0 * (1.0 / 30.0)), std::min(t_u, t_v) - 14.0);
  const double c = (t_u - t_R) / ((t_u - t_R) + (t_v - t_R));
  const int n_u = (int)std::lround(c * D);
  T.walk_focus_to(v);
  int so_far = 0, k = 0, on = k_none;
  for (int cur = u; cur != v;) { const int a = T.toward_focus[(size_t)cur]; const int w = T.weight(a); if (so_far + w >= n_u) { on = a; k = n_u - so_far; break; } so_far += w; cur = T.to(a); }
  if (on == k_none) throw std::runtime_error("midpoint rooting found no edge");
  const int R = root_on(T, on, k);
  double sum_t = 0.0; for (int i = 0; i < N; ++i) sum_t += tips.mid_date(i);
  const double Nd = (double)N, mean_t = sum_t / Nd;
  int d = 0; double s_m = 0.0, s_m2 = 0.0, s_dt2 = 0.0, s_mdt = 0.0;
  T.tour(R, [&](int a) { d += T.weight(a); const int x = T.to(a); if (T.leaf(x)) { const double m = (double)d, dt = tips.mid_date(x) - mean_t; s_m += m; s_m2 += m * m; s_dt2 += dt * dt; s_mdt += m * dt; } },
         [&](int a) { d -= T.weight(a); });
  const double mean_m = s_m / Nd, var_t = s_dt2 / Nd, var_m = s_m2 / Nd - mean_m * mean_m, cov = s_mdt / Nd;
  const double r2 = (var_m > 0.0 && var_t > 0.0) ? (cov * cov) / (var_m * var_t) : 0.0;
  if (var_t > 0.0 && cov > 0.0) { const double rate = cov / var_t; return {R, false, r2, rate, mean_t - mean_m / rate}; }
  return {R, false, r2, fallback, mean_t - mean_m / fallback};
}
inline Rooting regression_root(Tree& T, const Tips& tips, HostRng& rng) {   // ordinary least squares of divergence on date over every possible root (utree.cpp:1255-1464)
  const int N = T.n_tips; const double Nd = (double)N;
  if (N <= 2) return midpoint_root(T, tips);
  double sum_t = 0.0; for (int i = 0; i < N; ++i) sum_t += tips.mid_date(i);
  const double mean_t = sum_t / Nd;
  auto dt_of = [&](int i) { return tips.mid_date(i) - mean_t; };
  double s_dt2 = 0.0; for (int i = 0; i < N; ++i) { const double dt = dt_of(i); s_dt2 += dt * dt; }
  const double var_t = s_dt2 / Nd;
  if (var_t <= 0.0) return midpoint_root(T, tips);
  struct Mom { int n = 0; double dt = 0.0, m = 0.0, mdt = 0.0, m2 = 0.0; };                     // moments of the tips beyond an arc, distances measured from its origin's end
  auto moved = [](const Mom& s, int D) { const double d = (double)D; return Mom{s.n, s.dt, d * s.n + s.m, d * s.dt + s.mdt, d * d * s.n + 2 * d * s.m + s.m2}; };
  auto both = [](const Mom& a, const Mom& b) { return Mom{a.n + b.n, a.dt + b.dt, a.m + b.m, a.mdt + b.mdt, a.m2 + b.m2}; };
  std::vector<Mom> mom(T.tgt.size());
  auto gather = [&](int v, int except) { Mom s{}; for (int a : T.adj[(size_t)v]) if (a != k_none && a != except) s = both(s, moved(mom[(size_t)a], T.weight(a))); return s; };
  T.tour(0, [](int) {}, [&](int up) { const int X = T.from(up); mom[(size_t)Tree::mate(up)] = T.leaf(X) ? Mom{1, dt_of(X), 0, 0, 0} : gather(X, up); });
  T.tour(0, [&](int down) { const int P = T.from(down); mom[(size_t)Tree::mate(down)] = T.leaf(P) ? Mom{1, dt_of(P), 0, 0, 0} : gather(P, down); }, [](int) {});
  double best_r2 = -1.0; std::vector<std::pair<int, int>> cand;
  for (int a = 0; a < (int)T.tgt.size(); a += 2) {
    if (mom[(size_t)a].n == 0 && mom[(size_t)a + 1].n == 0) continue;
    const int D = T.weight(a);
    for (int k = 0; k <= D; ++k) {
      const Mom r = both(moved(mom[(size_t)a + 1], k), moved(mom[(size_t)a], D - k));
      const double cov = r.mdt / Nd; if (cov <= 0.0) continue;
      const double mean_m = r.m / Nd, var_m = r.m2 / Nd - mean_m * mean_m; if (var_m <= 0.0) continue;
      const double r2 = (cov * cov) / (var_m * var_t);
      if (r2 > best_r2) { best_r2 = r2; cand.clear(); }
      if (r2 == best_r2) cand.push_back({a, k});
    }
  }
  if (cand.empty()) return midpoint_root(T, tips);
  const auto [a, k] = cand[(size_t)(((unsigned __int128)rng.next64() * (uint64_t)cand.size()) >> 64)];
  const Mom r = both(moved(mom[(size_t)a + 1], k), moved(mom[(size_t)a], T.weight(a) - k));
  const int R = root_on(T, a, k);
  const double rate = (r.mdt / Nd) / var_t;
  return {R, true, best_r2, rate, mean_t - (r.m / Nd) / rate};
}

// The rooted, dated tree (utree.cpp:1750-1890): times from the fitted rate, tips clamped into their date ranges, every inner node at
// least a tenth of a day above its children; then the builder's closing passes and the move to the root's own sequence.
inline void to_rooted(Tree& T, const Rooting& ro, const Tips& tips, HostRng& rng, std::vector<BHostNode>& nodes, std::vector<uint8_t>& ref_out) {
  const int N = T.n_tips, R = ro.root;
  nodes.assign((size_t)(2 * N - 1), BHostNode());
  T.walk_focus_to(R);
  nodes[(size_t)R].t = ro.t_root;
  int m = 0;
  auto settle = [&](int X) { BHostNode& x = nodes[(size_t)X]; x.t = std::min(x.t, std::min(nodes[(size_t)x.c0].t, nodes[(size_t)x.c1].t) - 0.1); };
  T.tour(R, [&](int a) {
    const int P = T.from(a), X = T.to(a); BHostNode& x = nodes[(size_t)X];
    m += T.weight(a);
    x.parent = P; { BHostNode& p = nodes[(size_t)P]; if (p.c0 == EMAT_NO_NODE) p.c0 = X; else p.c1 = X; }
    const double est = ro.t_root + (double)m / ro.rate;
    if (T.leaf(X)) { x.t_min = tips.td->t_min[X]; x.t_max = tips.td->t_max[X]; x.t = std::clamp(est, (double)x.t_min, (double)x.t_max); x.miss = tips.missing(X); }
    else x.t = est;
    for (const Sd& d : T.dl[(size_t)a]) x.muts.push_back(BHostMut{x.t, d.site, d.from, d.to});
  }, [&](int a) { const int X = T.from(a); m -= T.weight(a); if (!T.leaf(X)) settle(X); });
  settle(R);
  // the root's sequence becomes the reference sequence (phylo_tree.cpp:309-322): no site that every tip lacks is among these deltas
  ref_out = T.ref;
  for (const Sd& d : T.ref_to_focus) ref_out[(size_t)d.site] = d.to;
  b_fix_up_missations(nodes, R, ref_out);
  for (int v = 0; v < 2 * N - 1; ++v) if (v != R) b_randomize_branch(nodes[(size_t)v], nodes[(size_t)nodes[(size_t)v].parent].t, rng);
}

}  // namespace ut

emat_status build_default_tree(emat_backend* h, const emat_tip_descs& td, uint64_t seed, int32_t* report) {
  HostRng rng; rng.key = seed;
  const ut::Tips tips{&td, &h->ref};
  const int N = td.num_tips;
  const bool verbose = verbose_reports();
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double>(now() - a).count(); };
  try {
    auto t0 = now();
    ut::Tree T = ut::guide_tree(tips, rng);
    int cost = T.total_weight();
    if (report) report[0] = cost;
    const double s_guide = since(t0); t0 = now();
    int rounds = 0;
    for (int round = 0; round < 5; ++round) {                          // utree.cpp:1899-1911
      ut::Tree again = ut::rebuilt_nearest_first(T, tips, rng);
      ++rounds;
      const int c = again.total_weight();
      if (c >= cost) break;
      cost = c; T = std::move(again);
    }
    if (report) report[1] = cost;
    const double s_rebuild = since(t0); t0 = now();
    ut::spr_refine(T, tips, rng);
    if (report) report[2] = T.total_weight();
    const double s_spr = since(t0); t0 = now();
    const ut::Rooting ro = ut::regression_root(T, tips, rng);
    if (report) report[3] = ro.by_regression ? 0 : 1;
    const double s_root = since(t0); t0 = now();
    std::vector<BHostNode> nodes; std::vector<uint8_t> ref;
    ut::to_rooted(T, ro, tips, rng, nodes, ref);
    if (verbose) fprintf(stderr, "[emat] build_default: %d tips | guide tree %.2f s | %d nearest-first rebuild(s) %.2f s | SPR refinement %.2f s | rooting (%s, r2 %.3f, %.3g mutations per day, root at %.1f) %.2f s | dating + closing passes %.2f s\n",
                         N, s_guide, rounds, s_rebuild, s_spr, ro.by_regression ? "regression" : "midpoint", ro.r2, ro.rate, ro.t_root, s_root, since(t0));
    FlatTree& f = h->built.tree;
    f = FlatTree(); f.resize_nodes(2 * N - 1); f.root = ro.root;
    for (int v = 0; v < 2 * N - 1; ++v) {
      const BHostNode& nd = nodes[(size_t)v];
      f.parent[v] = nd.parent; f.child0[v] = nd.c0; f.child1[v] = nd.c1; f.t[v] = nd.t; f.t_min[v] = nd.t_min; f.t_max[v] = nd.t_max;
      for (auto& mu : nd.muts) { f.mut_site.push_back(mu.site); f.mut_from.push_back(mu.from); f.mut_to.push_back(mu.to); f.mut_t.push_back(mu.t); }
      for (auto& iv : nd.miss) { f.miss_start.push_back(iv.first); f.miss_end.push_back(iv.second); }
      for (auto& fs : nd.mfs) { f.mfs_site.push_back(fs.first); f.mfs_state.push_back(fs.second); }
      f.mut_offset[v + 1] = (int32_t)f.mut_site.size(); f.miss_offset[v + 1] = (int32_t)f.miss_start.size(); f.mfs_offset[v + 1] = (int32_t)f.mfs_site.size();
    }
    h->built.ref = std::move(ref);
    h->built.valid = true;
  } catch (const std::exception& ex) { h->set_error(std::string("emat_tree_build_default: ") + ex.what()); return EMAT_ERR_INVALID_ARGUMENT; }
  return EMAT_OK;
}

}  // namespace

extern "C" {

emat_status emat_tree_build_default(emat_backend* h, const emat_tip_descs* tips, uint64_t seed, int32_t* report) {
  if (!h || !tips) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->have_ref) return fail(h, EMAT_ERR_STATE, "emat_set_ref_sequence must come first: the descriptors are deltas against it");
  h->built.valid = false;
  const std::string bad = validate_tip_descs(*tips, h->ref);
  if (!bad.empty()) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_build_default: " + bad);
  return build_default_tree(h, *tips, seed, report);
}
emat_status emat_tree_built_ref(emat_backend* h, uint8_t* ref_sequence) {
  if (!h || !ref_sequence) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->built.valid) return fail(h, EMAT_ERR_STATE, "no tree has been built");
  std::copy(h->built.ref.begin(), h->built.ref.end(), ref_sequence);
  return EMAT_OK;
}

}  // extern "C"
#endif  // EMAT_UTREE_HOST_HPP_
