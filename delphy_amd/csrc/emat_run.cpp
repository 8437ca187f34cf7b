// emat_run.cpp -- host-side driver above the engine boundary (include/emat_host.h).
//
// Restates, on the host and in plain C++, the part of reference `Run` that orchestrates the local-move
// hot path: random partition stencils (core/tree_partitioning.h:139-194), partition_tree (:196-239),
// Run::repartition (core/run.cpp:110-193), Run::normalize_root (:258-265), Run::push_global_params_to_subruns
// (:267-275), Run::run_local_moves (:682-693) and Run::reassemble (:195-256).  Global moves
// (run.cpp:695-1235) are out of scope and are not here.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

#include "../../include/emat_host.h"
#include "flat_tree.hpp"
#include "host_parallel.hpp"
#include "synth.hpp"

namespace emat {

struct HMut { double t; int32_t site; uint8_t from, to; };
struct HIv { int32_t start, end; };
struct HFs { int32_t site; uint8_t state; };
struct HNode {
  int32_t parent = EMAT_NO_NODE, c0 = EMAT_NO_NODE, c1 = EMAT_NO_NODE;
  float t_min = -FLT_MAX, t_max = FLT_MAX;
  double t = 0.0;
  std::vector<HMut> muts; std::vector<HIv> miss; std::vector<HFs> mfs;
  bool is_tip() const { return c0 == EMAT_NO_NODE; }
};
struct HTree {
  int32_t root = EMAT_NO_NODE;
  std::vector<HNode> nodes;
  static HTree from_view(const emat_flat_tree& v) {
    HTree t; t.root = v.root; t.nodes.resize(v.num_nodes);
    for (int i = 0; i < v.num_nodes; ++i) {
      HNode& n = t.nodes[i];
      n.parent = v.parent[i]; n.c0 = v.child0[i]; n.c1 = v.child1[i]; n.t_min = v.t_min[i]; n.t_max = v.t_max[i]; n.t = v.t[i];
      n.muts.reserve(v.mut_offset[i + 1] - v.mut_offset[i]); n.miss.reserve(v.miss_offset[i + 1] - v.miss_offset[i]); n.mfs.reserve(v.mfs_offset[i + 1] - v.mfs_offset[i]);
      for (int k = v.mut_offset[i]; k < v.mut_offset[i + 1]; ++k) n.muts.push_back({v.mut_t[k], v.mut_site[k], v.mut_from[k], v.mut_to[k]});
      for (int k = v.miss_offset[i]; k < v.miss_offset[i + 1]; ++k) n.miss.push_back({v.miss_start[k], v.miss_end[k]});
      for (int k = v.mfs_offset[i]; k < v.mfs_offset[i + 1]; ++k) n.mfs.push_back({v.mfs_site[k], v.mfs_state[k]});
    }
    return t;
  }
  FlatTree to_flat() const {
    FlatTree f; const int n = (int)nodes.size();
    f.resize_nodes(n); f.root = root;
    { size_t nm = 0, ni = 0, nf = 0; for (const HNode& nd : nodes) { nm += nd.muts.size(); ni += nd.miss.size(); nf += nd.mfs.size(); }
      f.mut_site.reserve(nm); f.mut_from.reserve(nm); f.mut_to.reserve(nm); f.mut_t.reserve(nm); f.miss_start.reserve(ni); f.miss_end.reserve(ni); f.mfs_site.reserve(nf); f.mfs_state.reserve(nf); }
    for (int i = 0; i < n; ++i) {
      const HNode& nd = nodes[i];
      f.parent[i] = nd.parent; f.child0[i] = nd.c0; f.child1[i] = nd.c1; f.t[i] = nd.t; f.t_min[i] = nd.t_min; f.t_max[i] = nd.t_max;
      for (auto& m : nd.muts) { f.mut_site.push_back(m.site); f.mut_from.push_back(m.from); f.mut_to.push_back(m.to); f.mut_t.push_back(m.t); }
      for (auto& iv : nd.miss) { f.miss_start.push_back(iv.start); f.miss_end.push_back(iv.end); }
      for (auto& fs : nd.mfs) { f.mfs_site.push_back(fs.site); f.mfs_state.push_back(fs.state); }
      f.mut_offset[i + 1] = (int32_t)f.mut_site.size(); f.miss_offset[i + 1] = (int32_t)f.miss_start.size(); f.mfs_offset[i + 1] = (int32_t)f.mfs_site.size();
    }
    return f;
  }
};

static bool iv_contains(const std::vector<HIv>& v, int l) {
  auto it = std::upper_bound(v.begin(), v.end(), l, [](int x, const HIv& iv) { return x < iv.start; });
  if (it == v.begin()) return false;
  --it; return l < it->end;
}
static std::vector<HIv> iv_merge(const std::vector<HIv>& A, const std::vector<HIv>& B) {   // interval_set.h:238-288
  std::vector<HIv> out; size_t ia = 0, ib = 0; bool inside = false; int cs = 0, ce = 0;
  while (!(ia == A.size() && ib == B.size())) {
    bool useA = (ia == A.size()) ? false : (ib == B.size()) ? true : (A[ia].start <= B[ib].start);
    HIv f = useA ? A[ia] : B[ib];
    if (!inside) { cs = f.start; ce = f.end; (useA ? ia : ib)++; inside = true; }
    else if (f.start <= ce) { ce = std::max(ce, f.end); (useA ? ia : ib)++; }
    else { out.push_back({cs, ce}); inside = false; }
  }
  if (inside) out.push_back({cs, ce});
  return out;
}

// A part of the partition: subtree <-> whole-tree node maps (tree_partitioning.h:31-54).
struct PartMap {
  int32_t cut_point = EMAT_NO_NODE;
  std::vector<int32_t> orig;      // subtree node -> whole-tree node
};

struct RunDriver {
  emat_backend* backend = nullptr;
  std::string last_error;
  HTree tree;
  std::vector<uint8_t> ref;
  int L = 0;
  uint64_t seed = 0;
  SplitMix64 bitgen{0};
  int num_parts = 1;
  int max_part_nodes = 0;   // 0 = the reference's partition rule exactly (the default, so that a drop-in Run reproduces the reference / oracle partition); opt-in, not in the reference (see refine_stencil): > 0 = parts larger than this are cut further at every repartition, -1 = three times the mean part size
  int last_num_parts = 0, last_largest_part = 0, last_extra_cuts = 0;   // of the last repartition (emat_run_partition_stats)
  int last_pick = -1; uint64_t last_refine_epoch = 0;                  // ... which stencil it picked, and the epoch its refinement's random stream was keyed with (emat_run_debug_redraw_partition)
  // model
  bool have_hky = false; double hky_mu = 0, hky_kappa = 1, hky_pi[4] = {0.25, 0.25, 0.25, 0.25};
  std::vector<double> nu_l;
  bool have_pop = false; emat_pop_model pop{}; std::vector<double> sky_x, sky_g;
  double t_step = 1.0; bool t_step_set = false;
  int only_displacing_inner_nodes = 0, topology_moves_enabled = 1;
  bool reference_remainder = false;   // the remainder of count / parts goes to part 0 as in Run::run_local_moves (run.cpp:683-689), instead of one move per part
  bool paranoid = false;   // Run::paranoid (run.h:220-224): check the incrementally maintained quantities of every part after every pass
  // partition state
  std::vector<std::vector<int32_t>> stencils; int64_t stencil_refresh_countdown = 0;
  std::vector<PartMap> parts; std::vector<FlatTree> subtrees; std::vector<uint64_t> part_seeds;   // parts stay flat (SoA + CSR) end to end
  int root_part = -1;
  uint64_t epoch = 0;
  bool parts_uploaded = false, model_pushed = false, coal_built = false;
  // A run sharded over several processes (one GPU each): every process holds the whole tree and cuts it
  // identically; the attached backend only gets the parts [part_lo, part_hi) (backend part id = part - part_lo).
  int shard_rank = 0, shard_world = 1, part_lo = 0, part_hi = 0;
  std::vector<uint64_t> part_epoch;   // epoch at which each part's subtree was last refreshed (downloaded or received)
  // The whole tree resident in HBM (emat_tree_* of the backend): this driver then only sees topology and
  // node times (tp_* below); `tree` is brought up to date on demand (ensure_host_tree).
  bool device_tree = false, device_tree_uploaded = false, host_tree_stale = false;
  bool partition_on_device = false;   // the current partition was made by emat_tree_partition: parts[p].orig / part_kids are filled on demand
  emat_status ensure_partition_on_host() {
    if (!partition_on_device || parts.empty() || !parts[0].orig.empty()) return EMAT_OK;
    const int P = (int)parts.size();
    std::vector<int32_t> off((size_t)P + 1);
    emat_status st = bk(emat_tree_get_partition(backend, off.data(), nullptr, nullptr, nullptr)); if (st) return st;
    std::vector<int32_t> orig((size_t)off[P]), k0((size_t)off[P]), k1((size_t)off[P]);
    st = bk(emat_tree_get_partition(backend, nullptr, orig.data(), k0.data(), k1.data())); if (st) return st;
    part_kids.assign(P, {});
    for (int p = 0; p < P; ++p) {
      parts[p].orig.assign(orig.begin() + off[p], orig.begin() + off[p + 1]);
      part_kids[p].resize((size_t)(off[p + 1] - off[p]));
      for (int s = 0; s < off[p + 1] - off[p]; ++s) part_kids[p][s] = {k0[off[p] + s], k1[off[p] + s]};
    }
    return EMAT_OK;
  }

  double t_max_tip() const { double t = -INFINITY; for (auto& n : tree.nodes) if (n.is_tip() && n.t_max > t) t = n.t_max; return t; }   // phylo_tree_calc.cpp:636-644

  // tree_partitioning.h:139-194
  std::vector<int32_t> generate_random_partition_stencil() {
    std::vector<int32_t> cuts;
    const int N = tp_n;   // needs sync_topology() / fetch_device_topology()
    std::vector<int> descendants(N, 0);
    long num_branches_left = N; int num_parts_left = num_parts;
    struct Item { int32_t node; int csf; };
    std::vector<Item> stack; stack.push_back({tp_root, -1});
    bool done = false;
    while (!stack.empty() && !done) {
      Item it = stack.back(); stack.pop_back();
      const int32_t nd_c0 = tp_kids[it.node].c0, nd_c1 = tp_kids[it.node].c1;
      const int nch = nd_c0 == EMAT_NO_NODE ? 0 : 2;
      if (it.csf == -1) {
        stack.push_back({it.node, nch});
        if (nch == 2) {
          if (bitgen.next() >> 63) { stack.push_back({nd_c0, -1}); stack.push_back({it.node, 1}); stack.push_back({nd_c1, -1}); stack.push_back({it.node, 0}); }
          else { stack.push_back({nd_c1, -1}); stack.push_back({it.node, 1}); stack.push_back({nd_c0, -1}); stack.push_back({it.node, 0}); }
        }
        continue;
      }
      if (it.csf != nch) continue;   // only post-order visits matter
      const int node = it.node;
      if (node == tp_root) break;
      if ((int)cuts.size() == num_parts - 1) break;
      descendants[node] = 1;
      if (nch == 2) descendants[node] += descendants[nd_c0] + descendants[nd_c1];
      long min_subtree_size = std::max(10L, num_branches_left / (num_parts_left + 1));
      if (descendants[node] >= min_subtree_size) {
        bool allowed = true;
        if (allowed && (num_branches_left - (descendants[node] - 1)) < min_subtree_size) allowed = false;
        if (allowed && (bitgen.next() >> 63)) allowed = false;
        if (allowed) {
          num_branches_left -= descendants[node] - 1;
          cuts.push_back(node);
          descendants[node] = 1;
          --num_parts_left;
        }
      }
    }
    return cuts;
  }

  // NOT in the reference: cut oversized parts further.  The reference sizes its parts for a handful of CPU threads; its stencils are
  // lists of cut NODES drawn every 200 cycles, and as the moves re-hang subtrees the parts those nodes delimit drift apart in
  // size -- by a few per cent at 8 parts of 25 000 nodes, but at 8 000 parts of 25 nodes a stencil that started balanced holds parts
  // of 700-2 000 nodes after a few cycles.  Every part performs the same number of moves per pass and the GPU runs all parts at
  // once, so the pass lasts as long as the chain of the largest part.  Parts above `limit` nodes therefore get further cut nodes,
  // drawn UNIFORMLY AT RANDOM among their inner nodes, round after round until no piece exceeds the limit.
  // Why that does not bias the sampler (the reference's own concern, run.cpp:88-93): a pass only moves nodes WITHIN a part of the
  // refined partition R, so which nodes a part owns, and which of them are inner nodes, is the same before and after the pass --
  // for the parts of R and for every coarser level they were cut from.  The rule reads nothing else (no subtree sizes, no times),
  // so the probability of drawing R from the tree before the pass and from the tree after it is the same: the pass is a mixture of
  // within-part kernels whose weights are constant on every set of trees it connects, which keeps each kernel's detailed balance.
  // (Cutting at the node that halves a part best -- what this function did until round 4 -- reads subtree sizes, which a pass
  // changes: a state-dependent choice of the kind the reference avoids by redrawing its stencils slowly.)
  // The draws come from a stream of their own (seed, epoch): the reference-rule stream `bitgen` sees the same sequence with the
  // limit on or off.
  int effective_max_part_nodes(size_t num_nodes) const {
    if (max_part_nodes >= 0) return max_part_nodes;
    const long mean = (long)num_nodes / std::max(1, num_parts);
    return (int)std::max(64L, 3 * mean);
  }
  std::vector<int32_t> refine_stencil(std::vector<int32_t> cuts) {   // needs sync_topology() / fetch_device_topology()
    const int N = tp_n;
    const int eff = effective_max_part_nodes((size_t)N);
    last_extra_cuts = 0;
    if (eff <= 0) return cuts;
    const int limit = std::max(eff, 21);
    // cut marks as a bit set (25 KB at 200 000 nodes: resident in every core's cache, where a byte per node costs a second miss per
    // visited node); tasks set bits of their own parts only, but share words: atomic OR, relaxed loads
    std::vector<std::atomic<uint32_t>> cut_bits(((size_t)N + 31) / 32);
    for (auto& w : cut_bits) w.store(0, std::memory_order_relaxed);
    auto mark_cut = [&](int32_t v) { cut_bits[(size_t)v >> 5].fetch_or(1u << (v & 31), std::memory_order_relaxed); };
    auto is_cut = [&](int32_t v) { return (cut_bits[(size_t)v >> 5].load(std::memory_order_relaxed) >> (v & 31)) & 1u; };
    for (int32_t c : cuts) mark_cut(c);
    mark_cut(tp_root);
    std::vector<int32_t> roots(cuts);
    if (std::find(roots.begin(), roots.end(), tp_root) == roots.end()) roots.push_back(tp_root);
    // Every part of the stencil on its own (host threads): walk it from its cut node, and if it is oversized cut it, then its
    // oversized pieces, and so on.  A walk stops at cut nodes, so a task only ever writes is_cut of nodes its own part owns.
    std::vector<std::vector<int32_t>> extra(roots.size());
    const uint64_t round_seed = seed ^ (0x9E3779B97F4A7C15ull * (epoch + 1)) ^ 0x5A17C0DEull;
    const Kids* const kids = tp_kids;
    parallel_for((int)roots.size(), [&](int ri) {
      static thread_local std::vector<int32_t> work, inner, stack;   // (8 000 tasks per cycle: no allocation in any of them)
      static thread_local std::vector<std::pair<uint64_t, int32_t>> keyed;
      work.clear(); work.push_back(roots[(size_t)ri]);
      SplitMix64 rng(round_seed ^ (0xD6E8FEB86659FD93ull * (uint64_t)(roots[(size_t)ri] + 1)));   // a stream per part: the result does not depend on the threads
      bool first = true;
      while (!work.empty()) {
        const int32_t c = work.back(); work.pop_back();
        // the piece below c: its size (a cut child counts as one node: it is a tip here) and its inner nodes other than c
        int size = 0; inner.clear(); stack.clear(); stack.push_back(c);
        while (!stack.empty()) {
          const int32_t v = stack.back(); stack.pop_back(); ++size;
          const Kids k = kids[v];   // (both children in one cache line: the walk is bound by misses on a 200 000-node tree)
          if (k.c0 == EMAT_NO_NODE || (v != c && is_cut(v))) continue;
          if (v != c) inner.push_back(v);
          stack.push_back(k.c0); stack.push_back(k.c1);
          __builtin_prefetch(&kids[k.c0]); __builtin_prefetch(&kids[k.c1]);   // (c1 is visited next, c0 after c1's whole subtree: its line is on its way by then)
        }
        if (first && size <= limit) return;   // the common case: one walk, nothing to do
        first = false;
        if (size <= limit || inner.empty()) continue;
        // as many new cut nodes as would make the pieces `limit` nodes on average, a uniformly drawn subset of the inner nodes.
        // The subset must not depend on the ORDER in which the walk met the nodes -- that order is the piece's topology, which the pass changes, while
        // the set is not -- so that the draw is a function of (what a pass leaves alone, the stream) alone and repeating it on the tree after the pass
        // gives the very same cut nodes (round 6: emat_run_debug_redraw_partition, tests/test_host_driver.py; until then the invariance held in
        // distribution only).  Every inner node gets a pseudo-random 64-bit key from (a salt drawn from the part's stream, its own index) and the `want`
        // smallest keys are taken: a uniform subset whatever the order of `inner`, found by selection in O(n) -- sorting the nodes and shuffling, the
        // first version of this, cost the refinement 0.45 ms per cycle at C4.
        const int want = std::min((int)inner.size(), std::max(1, (size + limit - 1) / limit - 1));
        const uint64_t salt = rng.next();
        keyed.clear();
        for (int32_t v : inner) { uint64_t z = salt ^ ((uint64_t)(uint32_t)v * 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; keyed.push_back({z ^ (z >> 31), v}); }
        std::nth_element(keyed.begin(), keyed.begin() + (want - 1), keyed.end());
        std::sort(keyed.begin(), keyed.begin() + want);   // (the few that were drawn, by key: the order in which their pieces are looked at next)
        for (int k = 0; k < want; ++k) { const int32_t v = keyed[(size_t)k].second; mark_cut(v); extra[(size_t)ri].push_back(v); work.push_back(v); }
        work.push_back(c);   // what is left above the new cut nodes may still be too large
      }
    }, 16, 32);
    for (auto& e : extra) { cuts.insert(cuts.end(), e.begin(), e.end()); last_extra_cuts += (int)e.size(); }
    return cuts;
  }

  // ---- the draw of a cycle's partition: refresh_partition_stencils (run.cpp:87-108), the pick, and the part-size limit's refinement.
  // Apart from the cut itself so that the run drivers of ONE process (emat_multi: a driver per GPU, all with the same seed, all bound to
  // draw the very same thing) draw once: the leader draws (emat_run_draw_partition, or implicitly at its repartition), a follower
  // (emat_run_follow_draws) takes the leader's cut nodes -- 8 000-14 000 integers behind a shared pointer -- instead of walking the tree
  // again on the same host cores (eight shards refined the same stencil eight times).  Needs tp_* (sync_topology /
  // fetch_device_topology).
  std::shared_ptr<const std::vector<int32_t>> drawn; uint64_t drawn_for_epoch = ~0ull;
  RunDriver* follow = nullptr;   // the driver whose draws this one takes (same process, same seed, same tree)
  void draw_partition() {
    if (stencils.empty() || stencil_refresh_countdown <= 0) {
      stencils.clear();
      for (int i = 0; i < 10; ++i) stencils.push_back(generate_random_partition_stencil());
      stencil_refresh_countdown = 200;
    }
    --stencil_refresh_countdown;
    last_pick = bitgen.below((int)stencils.size()); last_refine_epoch = epoch;
    drawn = std::make_shared<const std::vector<int32_t>>(refine_stencil(stencils[(size_t)last_pick]));
    drawn_for_epoch = epoch;
  }
  emat_status ensure_draw() {   // the cut nodes of the partition of this epoch in `drawn`
    if (follow) {
      if (follow->drawn_for_epoch != epoch || !follow->drawn) return fail(EMAT_ERR_STATE, "this run takes its partition draws from another one (emat_run_follow_draws), which has not drawn this cycle's yet (emat_run_draw_partition on the leader first)");
      drawn = follow->drawn; drawn_for_epoch = epoch;
      last_pick = follow->last_pick; last_refine_epoch = follow->last_refine_epoch; last_extra_cuts = follow->last_extra_cuts;
      return EMAT_OK;
    }
    if (drawn_for_epoch != epoch || !drawn) draw_partition();
    return EMAT_OK;
  }

  void note_partition_stats() { last_num_parts = (int)parts.size(); last_largest_part = 0; for (auto& pm : parts) last_largest_part = std::max(last_largest_part, (int)pm.orig.size()); }
  // tree_partitioning.h:88-135 and :196-239
  void partition_tree(const std::vector<int32_t>& stencil) {
    int root_idx = (int)stencil.size(); bool root_in = false;
    for (size_t i = 0; i < stencil.size(); ++i) if (stencil[i] == tp_root) { root_in = true; root_idx = (int)i; break; }
    const int P = (int)stencil.size() + (root_in ? 0 : 1);
    std::vector<char> is_cut((size_t)tp_n, 0);
    for (int32_t c : stencil) is_cut[c] = 1;
    if (!root_in) { is_cut[tp_root] = 1; root_idx = P - 1; }
    root_part = root_idx;
    parts.assign(P, PartMap{});
    part_kids.assign(P, {});
    parallel_for(P, [&](int i) {   // every part walks down from its own cut point: independent
      PartMap& pm = parts[i];
      pm.cut_point = (i == root_part) ? tp_root : stencil[i];
      struct W { int32_t src, dst; };
      std::vector<W> work; pm.orig.clear(); pm.orig.push_back(pm.cut_point);
      work.push_back({pm.cut_point, 0});
      // children get consecutive indices when their parent is expanded; right child expanded first (LIFO)
      std::vector<std::pair<int32_t, int32_t>> kids(1, {EMAT_NO_NODE, EMAT_NO_NODE});
      while (!work.empty()) {
        W w = work.back(); work.pop_back();
        const int32_t k0 = tp_kids[w.src].c0, k1 = tp_kids[w.src].c1;
        if (k0 == EMAT_NO_NODE || (is_cut[w.src] && w.src != pm.cut_point)) continue;
        int32_t dl = (int32_t)pm.orig.size(); pm.orig.push_back(k0);
        int32_t dr = (int32_t)pm.orig.size(); pm.orig.push_back(k1);
        kids.resize(pm.orig.size(), {EMAT_NO_NODE, EMAT_NO_NODE});
        kids[w.dst] = {dl, dr};
        work.push_back({k0, dl}); work.push_back({k1, dr});
      }
      part_kids[i] = std::move(kids);
    });
  }
  std::vector<std::vector<std::pair<int32_t, int32_t>>> part_kids;
  // Compact copy of the whole tree's topology (the node records carry three vectors each and are 100+ bytes apart:
  // walking them misses the cache at every step).  Rebuilt at the start of every repartition.
  // The partitioner's walks read a node's two children and nothing else: side by side, one cache line per visit (they are bound
  // by misses on a 200 000-node tree).  `tp_kids` points at this object's own copy (host-resident tree) or straight into the
  // backend's page-locked mirror of the device-resident tree, which every reassemble refreshes.
  struct Kids { int32_t c0, c1; };
  std::vector<int32_t> tp_parent; std::vector<Kids> tp_kids_own; const Kids* tp_kids = nullptr;
  int tp_n = 0; int32_t tp_root = EMAT_NO_NODE; double tp_root_t = 0.0;
  void sync_topology() {
    const int N = (int)tree.nodes.size();
    tp_parent.resize(N); tp_kids_own.resize(N); tp_root = tree.root; tp_root_t = tree.nodes[tree.root].t;
    parallel_for(N, [&](int v) { const HNode& nd = tree.nodes[v]; tp_parent[v] = nd.parent; tp_kids_own[v] = Kids{nd.c0, nd.c1}; }, 4096);
    tp_kids = tp_kids_own.data(); tp_n = N;
  }
  emat_status fetch_device_topology() {   // device-resident tree: what the backend mirrored at its last upload / reassemble
    static_assert(sizeof(Kids) == 2 * sizeof(int32_t), "Kids is a pair of int32");
    const int32_t* k = nullptr; int32_t n = 0;
    emat_status st = bk(emat_tree_get_kids(backend, &k, &n, &tp_root, &tp_root_t)); if (st) return st;
    tp_kids = (const Kids*)k; tp_n = n;
    return EMAT_OK;
  }
  // `tree` (and `ref`) as of the last reassemble, when the authoritative copy lives on the device
  emat_status ensure_host_tree() {
    if (!device_tree || !host_tree_stale) return EMAT_OK;
    int32_t nn, nm, ni, nf;
    emat_status st = bk(emat_tree_get_sizes(backend, &nn, &nm, &ni, &nf)); if (st) return st;
    FlatTree f; f.allocate(nn, nm, ni, nf);
    emat_flat_tree v = f.view();
    st = bk(emat_tree_download(backend, &v, ref.data())); if (st) return st;
    f.root = v.root;
    emat_flat_tree fv = f.view();
    tree = HTree::from_view(fv);
    host_tree_stale = false;
    return EMAT_OK;
  }

  // Run::normalize_root + rereference_to_root_sequence (run.cpp:258-265, phylo_tree.cpp:309-322)
  void normalize_root() {
    HNode& r = tree.nodes[tree.root];
    if (r.muts.empty()) return;
    for (auto& m : r.muts) ref[m.site] = m.to;
    for (auto& nd : tree.nodes) {
      if (nd.miss.empty()) continue;
      for (auto& m : r.muts) {
        if (!iv_contains(nd.miss, m.site)) continue;
        auto it = std::lower_bound(nd.mfs.begin(), nd.mfs.end(), m.site, [](const HFs& f, int l) { return f.site < l; });
        if (it != nd.mfs.end() && it->site == m.site) { if (it->state == m.to) nd.mfs.erase(it); }
        else if (m.from != m.to) nd.mfs.insert(it, HFs{m.site, m.from});
      }
    }
    r.muts.clear();
    model_pushed = false;   // the reference sequence (hence cum_Q) changed
  }

  // State at a cut point c: the sites missing at c (union of the missations from c up to the root) and the deltas
  // reference sequence -> sequence at c (reconstruct_missing_sites_at phylo_tree_calc.cpp:41-56, view_of_sequence_at
  // :19-35).  The reference recomputes both by walking from every subroot to the root; here they are carried down the
  // tree of cut points instead -- state(c) = state(nearest cut point above c) extended by the path between the two --
  // which gives the same sets at a cost proportional to the part depth rather than the tree depth.
  struct HFsPair { int32_t site; uint8_t from, to; };
  struct CutState { std::vector<HIv> miss; std::vector<HFsPair> deltas; };
  void cut_point_states(std::vector<CutState>& out) {
    const int P = (int)parts.size();
    out.assign(P, CutState{});
    std::vector<int32_t> part_of_node(tree.nodes.size(), -1);
    for (int p = 0; p < P; ++p) part_of_node[parts[p].cut_point] = p;
    std::vector<int> above(P, -1);   // part whose cut point is the nearest one above this part's cut point
    std::vector<std::vector<int32_t>> path(P);   // nodes strictly below `above`'s cut point down to this cut point, top-down
    parallel_for(P, [&](int p) {
      std::vector<int32_t> up;
      int32_t cur = parts[p].cut_point;
      up.push_back(cur);
      for (cur = tp_parent[cur]; cur != EMAT_NO_NODE; cur = tp_parent[cur]) {
        if (part_of_node[cur] >= 0) { above[p] = part_of_node[cur]; break; }
        up.push_back(cur);
      }
      path[p].assign(up.rbegin(), up.rend());
    });
    // levels of the forest of cut points: a part's state needs only the state of the part above it, so the parts of
    // one level are independent
    std::vector<std::vector<int>> levels;
    {
      std::vector<std::vector<int>> below(P); std::vector<int> frontier;
      for (int p = 0; p < P; ++p) if (above[p] >= 0) below[above[p]].push_back(p); else frontier.push_back(p);
      while (!frontier.empty()) {
        std::vector<int> next;
        for (int p : frontier) for (int q : below[p]) next.push_back(q);
        levels.push_back(std::move(frontier));
        frontier = std::move(next);
      }
    }
    for (const auto& level : levels) parallel_for((int)level.size(), [&](int li) {
      const int p = level[li];
      CutState& st = out[p];
      std::map<int32_t, std::pair<uint8_t, uint8_t>> deltas;
      if (above[p] >= 0) {
        const CutState& a = out[above[p]];
        st.miss = a.miss;
        for (const auto& d : a.deltas) deltas.emplace_hint(deltas.end(), d.site, std::make_pair(d.from, d.to));
      }
      for (int32_t node : path[p]) {
        const HNode& nd = tree.nodes[node];
        if (!nd.miss.empty()) st.miss = iv_merge(st.miss, nd.miss);
        for (const auto& m : nd.muts) {   // forward in time: push_back_site_deltas
          auto f = deltas.find(m.site);
          if (f == deltas.end()) deltas[m.site] = {m.from, m.to};
          else { if (f->second.second != m.from) throw std::runtime_error("inconsistent mutation chain above a subroot"); f->second.second = m.to; if (f->second.first == f->second.second) deltas.erase(f); }
        }
      }
      st.deltas.reserve(deltas.size());
      for (const auto& [l, d] : deltas) st.deltas.push_back(HFsPair{l, d.first, d.second});
    }, 8);
  }

  uint64_t seed_for_part(int p) const {   // a fresh RNG stream per part and cycle
    uint64_t z = seed ^ (0x9E3779B97F4A7C15ull * (epoch + 1)) ^ ((uint64_t)p << 32 | (uint64_t)p);
    SplitMix64 sm(z);
    return sm.next();
  }
  void build_subtrees() {   // run.cpp:131-184
    const int P = (int)parts.size();
    subtrees.clear(); subtrees.resize(P); part_seeds.assign(P, 0);
    std::vector<CutState> states;
    auto tc0 = std::chrono::steady_clock::now();
    cut_point_states(states);
    if (verbose_reports()) fprintf(stderr, "[emat_run] cut_point_states %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tc0).count());
    auto build_one = [&](int p) {
      const PartMap& pm = parts[p];
      const int n = (int)pm.orig.size();
      const int32_t subroot = pm.cut_point;
      const std::vector<HIv>& root_miss = states[p].miss;
      // the subroot's synthetic lists (run.cpp:141-153): missing sites at the cut, deltas reference -> sequence at the cut
      std::vector<HFsPair> root_muts;
      for (const auto& d : states[p].deltas) if (!iv_contains(root_miss, d.site)) root_muts.push_back(d);
      FlatTree st; st.resize_nodes(n); st.root = 0;
      size_t nm = 0, ni = 0, nf = 0;
      for (int s = 0; s < n; ++s) {
        const int32_t o = pm.orig[s];
        if (o == subroot) { nm += root_muts.size(); ni += root_miss.size(); }
        else { const HNode& on = tree.nodes[o]; nm += on.muts.size(); ni += on.miss.size(); nf += on.mfs.size(); }
      }
      st.mut_site.resize(nm); st.mut_from.resize(nm); st.mut_to.resize(nm); st.mut_t.resize(nm);
      st.miss_start.resize(ni); st.miss_end.resize(ni); st.mfs_site.resize(nf); st.mfs_state.resize(nf);
      size_t km = 0, ki = 0, kf = 0;
      for (int s = 0; s < n; ++s) {
        const int32_t o = pm.orig[s];
        const HNode& on = tree.nodes[o];
        const int32_t k0 = part_kids[p][s].first, k1 = part_kids[p][s].second;
        st.child0[s] = k0; st.child1[s] = k1;
        if (k0 != EMAT_NO_NODE) { st.parent[k0] = s; st.parent[k1] = s; }
        st.t[s] = on.t;
        if (k0 == EMAT_NO_NODE && !on.is_tip()) { st.t_min[s] = (float)on.t; st.t_max[s] = (float)on.t; }   // frozen boundary node (run.cpp:165-168)
        else { st.t_min[s] = on.t_min; st.t_max[s] = on.t_max; }
        // A frozen boundary "tip" whose float-rounded bounds do not bracket t would fail the t_min <= t <= t_max
        // convention by an ulp of float; the reference tolerates 1e-2 (phylo_tree.cpp:117-121).  Keep t exact.
        if (o == subroot) {
          for (const auto& iv : root_miss) { st.miss_start[ki] = iv.start; st.miss_end[ki] = iv.end; ++ki; }
          for (const auto& d : root_muts) { st.mut_site[km] = d.site; st.mut_from[km] = ref[d.site]; st.mut_to[km] = d.to; st.mut_t[km] = -std::numeric_limits<double>::max(); ++km; }
        } else {
          for (const auto& m : on.muts) { st.mut_site[km] = m.site; st.mut_from[km] = m.from; st.mut_to[km] = m.to; st.mut_t[km] = m.t; ++km; }
          for (const auto& iv : on.miss) { st.miss_start[ki] = iv.start; st.miss_end[ki] = iv.end; ++ki; }
          for (const auto& f : on.mfs) { st.mfs_site[kf] = f.site; st.mfs_state[kf] = f.state; ++kf; }
        }
        st.mut_offset[s + 1] = (int32_t)km; st.miss_offset[s + 1] = (int32_t)ki; st.mfs_offset[s + 1] = (int32_t)kf;
      }
      st.parent[0] = EMAT_NO_NODE;
      subtrees[p] = std::move(st);
      part_seeds[p] = seed_for_part(p);
    };
    parallel_for(P, build_one);
  }

  emat_status fail(emat_status st, const std::string& m) { last_error = m; return st; }
  emat_status bk(emat_status st) { if (st != EMAT_OK) last_error = std::string("backend: ") + emat_last_error(backend); return st; }

  emat_status push_model() {
    if (!backend) return EMAT_OK;
    if (!have_hky) return fail(EMAT_ERR_STATE, "emat_run_set_hky must be called first");
    emat_status st = bk(emat_set_ref_sequence(backend, ref.data(), L)); if (st) return st;
    // Hky_model::derive_site_evo_model (evo_hky.cpp:7-50)
    const double k = hky_kappa; const double* pi = hky_pi;
    double r[4][4] = {{0, 1, k, 1}, {1, 0, 1, k}, {k, 1, 0, 1}, {1, k, 1, 0}};
    double rowv[4]; for (int b = 0; b < 4; ++b) { rowv[b] = 0.0; for (int a = 0; a < 4; ++a) rowv[b] += pi[a] * r[a][b]; }
    double R = 0.0; for (int b = 0; b < 4; ++b) R += rowv[b] * pi[b];
    double q[16];
    for (int a = 0; a < 4; ++a) { q[a * 4 + a] = 0.0; for (int b = 0; b < 4; ++b) if (a != b) { q[a * 4 + b] = r[a][b] / R * pi[b]; q[a * 4 + a] -= q[a * 4 + b]; } }
    std::vector<int32_t> pfs(L, 0);
    std::vector<double> nu = nu_l.empty() ? std::vector<double>(L, 1.0) : nu_l;
    st = bk(emat_set_evo(backend, 1, &hky_mu, pi, q, nu.data(), pfs.data())); if (st) return st;
    st = bk(emat_set_flags(backend, t_max_tip(), only_displacing_inner_nodes, topology_moves_enabled)); if (st) return st;
    model_pushed = true;
    return EMAT_OK;
  }
  double default_t_step() const {   // Run keeps ~400 cells over the tree span (run.cpp:20, :734-747)
    double lo = device_tree && tp_n > 0 ? tp_root_t : tree.nodes[tree.root].t, hi = t_max_tip();
    double span = hi - lo; if (!(span > 0)) span = 1.0;
    return std::max(span / 400.0, 1.0 / 400.0);
  }
  emat_status build_coalescent() {   // Run::reset_very_scalable_coalescent_parts (run.cpp:277-293)
    if (!backend) return EMAT_OK;
    if (!have_pop) return fail(EMAT_ERR_STATE, "emat_run_set_pop_model must be called first");
    emat_pop_model pm = pop; pm.skygrid_x = sky_x.data(); pm.skygrid_gamma = sky_g.data();
    emat_status st = bk(emat_build_coalescent_parts(backend, &pm, root_part, t_step_set ? t_step : default_t_step())); if (st) return st;
    coal_built = true;
    return EMAT_OK;
  }
  void shard_block(int n) {   // contiguous block of the parts for this process (sizes differ by at most one)
    const int base = n / shard_world, rem = n % shard_world;
    part_lo = shard_rank * base + std::min(shard_rank, rem);
    part_hi = part_lo + base + (shard_rank < rem ? 1 : 0);
  }
  emat_status upload_parts() {
    if (!backend) return EMAT_OK;
    const int nloc = part_hi - part_lo;
    if (nloc <= 0) return fail(EMAT_ERR_STATE, "this rank holds no parts: fewer parts than processes");
    emat_status st = bk(emat_begin_upload(backend, nloc)); if (st) return st;
    std::atomic<int> bad{EMAT_OK};
    parallel_for(nloc, [&](int q) {   // emat_part_upload is safe to call concurrently for distinct parts
      const int p = part_lo + q;
      emat_flat_tree v = subtrees[p].view();
      emat_status s1 = emat_part_upload(backend, q, &v, p == root_part ? 1 : 0, part_seeds[p]);
      if (s1 != EMAT_OK) bad.store(s1);
    });
    if (bad.load() != EMAT_OK) return bk((emat_status)bad.load());
    st = bk(emat_end_upload(backend)); if (st) return st;
    parts_uploaded = true;
    return EMAT_OK;
  }
  // Bring the subtrees of the local parts up to date with the device.
  emat_status download_local_parts() {
    if (!(backend && parts_uploaded)) return EMAT_OK;
    int32_t nn0, nm0, ni0, nf0;
    emat_status st0 = bk(emat_part_get_sizes(backend, 0, &nn0, &nm0, &ni0, &nf0)); if (st0) return st0;   // one D2H of all slabs, before the threads start
    std::atomic<int> bad{EMAT_OK};
    parallel_for(part_hi - part_lo, [&](int q) {
      int32_t nn, nm, ni, nf;
      emat_status st = emat_part_get_sizes(backend, q, &nn, &nm, &ni, &nf); if (st) { bad.store(st); return; }
      FlatTree f; f.allocate(nn, nm, ni, nf);
      emat_flat_tree v = f.view();
      st = emat_part_download(backend, q, &v); if (st) { bad.store(st); return; }
      f.root = v.root;
      subtrees[part_lo + q] = std::move(f);
      part_epoch[part_lo + q] = epoch;
    });
    if (bad.load() != EMAT_OK) return bk((emat_status)bad.load());
    return EMAT_OK;
  }
  // ---- exchange format of part subtrees between processes: per part {int32 id, nodes, muts, intervals, from_states, root, 0, 0}
  //      followed by the FlatTree arrays, every array padded to 8 bytes -----------------------------------------------------
  static uint64_t pad8(uint64_t x) { return (x + 7u) & ~(uint64_t)7u; }
  static uint64_t packed_bytes(const FlatTree& t) {
    const uint64_t n = t.num_nodes(), m = t.num_muts(), i = t.num_intervals(), f = t.num_from_states();
    return 32 + 3 * pad8(4 * n) + 8 * n + 2 * pad8(4 * n) + 3 * pad8(4 * (n + 1)) + pad8(4 * m) + 2 * pad8(m) + 8 * m + 2 * pad8(4 * i) + pad8(4 * f) + pad8(f);
  }
  template <class T> static void put(uint8_t*& w, const std::vector<T>& v) { std::memcpy(w, v.data(), v.size() * sizeof(T)); w += pad8(v.size() * sizeof(T)); }
  template <class T> static bool get(const uint8_t*& r, const uint8_t* end, std::vector<T>& v, size_t count) {
    if ((uint64_t)(end - r) < pad8(count * sizeof(T))) return false;
    v.resize(count); std::memcpy(v.data(), r, count * sizeof(T)); r += pad8(count * sizeof(T)); return true;
  }
  emat_status pack_local_parts(uint8_t* buf, uint64_t cap, uint64_t* needed) {
    emat_status st = download_local_parts(); if (st) return st;
    uint64_t tot = 0;
    for (int p = part_lo; p < part_hi; ++p) tot += packed_bytes(subtrees[p]);
    if (needed) *needed = tot;
    if (!buf || cap < tot) return buf ? fail(EMAT_ERR_BUFFER_TOO_SMALL, "emat_run_pack_local_parts: buffer too small") : EMAT_OK;
    uint8_t* w = buf;
    for (int p = part_lo; p < part_hi; ++p) {
      const FlatTree& t = subtrees[p];
      int32_t hdr[8] = {p, t.num_nodes(), t.num_muts(), t.num_intervals(), t.num_from_states(), t.root, 0, 0};
      std::memcpy(w, hdr, 32); w += 32;
      put(w, t.parent); put(w, t.child0); put(w, t.child1); put(w, t.t); put(w, t.t_min); put(w, t.t_max);
      put(w, t.mut_offset); put(w, t.mut_site); put(w, t.mut_from); put(w, t.mut_to); put(w, t.mut_t);
      put(w, t.miss_offset); put(w, t.miss_start); put(w, t.miss_end); put(w, t.mfs_offset); put(w, t.mfs_site); put(w, t.mfs_state);
    }
    return EMAT_OK;
  }
  emat_status unpack_parts(const uint8_t* buf, uint64_t bytes) {
    const uint8_t* r = buf; const uint8_t* end = buf + bytes;
    while (r < end) {
      if (end - r < 32) return fail(EMAT_ERR_INVALID_ARGUMENT, "emat_run_unpack_parts: truncated part header");
      int32_t hdr[8]; std::memcpy(hdr, r, 32); r += 32;
      const int p = hdr[0]; const size_t n = (size_t)hdr[1], m = (size_t)hdr[2], i = (size_t)hdr[3], f = (size_t)hdr[4];
      if (p < 0 || p >= (int)subtrees.size() || hdr[1] != (int)parts[p].orig.size() || hdr[2] < 0 || hdr[3] < 0 || hdr[4] < 0) return fail(EMAT_ERR_INVALID_ARGUMENT, "emat_run_unpack_parts: part does not belong to the current partition");
      FlatTree t; t.root = hdr[5];
      bool ok = get(r, end, t.parent, n) && get(r, end, t.child0, n) && get(r, end, t.child1, n) && get(r, end, t.t, n) && get(r, end, t.t_min, n) && get(r, end, t.t_max, n)
             && get(r, end, t.mut_offset, n + 1) && get(r, end, t.mut_site, m) && get(r, end, t.mut_from, m) && get(r, end, t.mut_to, m) && get(r, end, t.mut_t, m)
             && get(r, end, t.miss_offset, n + 1) && get(r, end, t.miss_start, i) && get(r, end, t.miss_end, i) && get(r, end, t.mfs_offset, n + 1) && get(r, end, t.mfs_site, f) && get(r, end, t.mfs_state, f);
      if (!ok) return fail(EMAT_ERR_INVALID_ARGUMENT, "emat_run_unpack_parts: truncated part");
      emat_flat_tree v = t.view();
      if (!validate_flat_tree(v, L).empty()) return fail(EMAT_ERR_INVALID_ARGUMENT, "emat_run_unpack_parts: part " + std::to_string(p) + " is not a valid tree");
      subtrees[p] = std::move(t);
      part_epoch[p] = epoch;
    }
    return EMAT_OK;
  }

  // The same cycle with the whole tree resident in HBM: the host draws and applies the stencil on the topology alone
  // and hands the parts over as three int arrays; nodes, mutations and missations never leave the device.
  emat_status repartition_device() {
    const bool verbose = verbose_reports();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now();
    if (!backend) return fail(EMAT_ERR_NO_DEVICE, "a device-resident tree needs a backend");
    if (!have_pop) return fail(EMAT_ERR_STATE, "emat_run_set_pop_model must be called first");
    emat_status st;
    if (!device_tree_uploaded) {
      normalize_root();
      if (!model_pushed) { st = push_model(); if (st) return st; }
      FlatTree f = tree.to_flat();
      emat_flat_tree v = f.view();
      st = bk(emat_tree_upload(backend, &v)); if (st) return st;
      device_tree_uploaded = true; host_tree_stale = false;
    }
    if (!model_pushed) { st = push_model(); if (st) return st; }
    { EMAT_SPAN("run.repartition: fetch_device_topology"); st = fetch_device_topology(); if (st) return st; }
    auto t1 = now();
    HostLaps laps;
    std::vector<int32_t> part_off, orig, kid0, kid1;
    try {
      { emat_status ds = ensure_draw(); if (ds) return ds; }
      const std::vector<int32_t>& stencil = *drawn;
      laps.mark("run.repartition: stencil pick + refine_stencil");
      part_kids.clear();
      // partition_tree itself: on the device (one thread per part) unless the parts are few and large, where one host thread
      // per part is the better fit
      const size_t N = (size_t)tp_n;
      if (stencil.size() + 1 >= 64 && N / (stencil.size() + 1) <= 2048) {
        int32_t P = 0, rp = -1;
        std::vector<int32_t> psz(stencil.size() + 1, 0);
        emat_status st1 = bk(emat_tree_partition(backend, (int32_t)stencil.size(), stencil.data(), &P, &rp, psz.data())); if (st1) return st1;
        last_num_parts = P; last_largest_part = 0; for (int p = 0; p < P; ++p) last_largest_part = std::max(last_largest_part, (int)psz[(size_t)p]);
        laps.mark("run.repartition: emat_tree_partition + largest part");
        parts.assign((size_t)P, PartMap{});
        for (int p = 0; p < P; ++p) parts[p].cut_point = p < (int)stencil.size() ? stencil[p] : tp_root;
        root_part = rp; partition_on_device = true;
        laps.mark("run.repartition: part maps");
      } else { partition_tree(stencil); partition_on_device = false; note_partition_stats(); }
      ++epoch;
    } catch (const std::exception& ex) { return fail(EMAT_ERR_INTERNAL, ex.what()); }
    auto t2 = now();
    const int P = (int)parts.size();
    if (partition_on_device) {
      part_seeds.assign(P, 0);
      for (int p = 0; p < P; ++p) part_seeds[p] = seed_for_part(p);
      subtrees.clear();
      shard_block(P);
      if (part_hi <= part_lo) return fail(EMAT_ERR_STATE, "this rank holds no parts: fewer parts than processes");
      part_epoch.assign(P, 0);
      emat_pop_model pm = pop; pm.skygrid_x = sky_x.data(); pm.skygrid_gamma = sky_g.data();
      auto t3 = now();
      laps.mark("run.repartition: seeds, shard block");
      st = bk(emat_tree_repartition_range(backend, P, nullptr, nullptr, nullptr, nullptr, root_part, part_seeds.data(), &pm, t_step_set ? t_step : default_t_step(), part_lo, part_hi));
      if (st) return st;
      parts_uploaded = true; coal_built = true; host_tree_stale = true;
      if (verbose) fprintf(stderr, "[emat_run] repartition (device tree): upload / topology %.1f ms | stencil + emat_tree_partition %.1f ms | seeds %.1f ms | emat_tree_repartition %.1f ms\n",
                           ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
      return EMAT_OK;
    }
    part_off.assign(P + 1, 0);
    for (int p = 0; p < P; ++p) part_off[p + 1] = part_off[p] + (int32_t)parts[p].orig.size();
    orig.resize(part_off[P]); kid0.resize(part_off[P]); kid1.resize(part_off[P]);
    part_seeds.assign(P, 0);
    parallel_for(P, [&](int p) {
      const int b = part_off[p], n = (int)parts[p].orig.size();
      for (int s = 0; s < n; ++s) { orig[b + s] = parts[p].orig[s]; kid0[b + s] = part_kids[p][s].first; kid1[b + s] = part_kids[p][s].second; }
      part_seeds[p] = seed_for_part(p);
    }, 64);
    subtrees.clear();
    shard_block(P);
    if (part_hi <= part_lo) return fail(EMAT_ERR_STATE, "this rank holds no parts: fewer parts than processes");
    part_epoch.assign(P, 0);
    emat_pop_model pm = pop; pm.skygrid_x = sky_x.data(); pm.skygrid_gamma = sky_g.data();
    auto t3 = now();
    // (a sharded run: every process has the whole tree in its HBM and cuts it identically; it builds the slabs of its own block of parts only)
    st = bk(emat_tree_repartition_range(backend, P, part_off.data(), orig.data(), kid0.data(), kid1.data(), root_part, part_seeds.data(), &pm, t_step_set ? t_step : default_t_step(), part_lo, part_hi));
    if (st) return st;
    parts_uploaded = true; coal_built = true; host_tree_stale = true;
    if (verbose) fprintf(stderr, "[emat_run] repartition (device tree): upload / topology %.1f ms | stencil + partition_tree %.1f ms | flatten %.1f ms | emat_tree_repartition %.1f ms\n",
                         ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
    return EMAT_OK;
  }
  emat_status reassemble_device() {
    if (!parts_uploaded) return fail(EMAT_ERR_STATE, "repartition first");
    if (shard_world > 1) return fail(EMAT_ERR_STATE, "a sharded run with the tree on the devices gathers in steps, with the exchange between them (emat_tree_get_root_deltas ... emat_tree_reassemble_end, then emat_run_note_device_reassembled)");
    int32_t nd = 0;   // at most one change per site
    std::vector<int32_t> site(ref.size()); std::vector<uint8_t> from(ref.size()), to(ref.size());
    emat_status st = bk(emat_tree_reassemble(backend, &nd, site.data(), from.data(), to.data(), (int32_t)ref.size())); if (st) return st;
    for (int k = 0; k < nd; ++k) ref[site[k]] = to[k];
    parts_uploaded = false; host_tree_stale = true;
    return EMAT_OK;
  }

  // Run::run_local_moves (run.cpp:682-693) with the remainder of count / parts spread one move per part instead of all
  // on part 0 (emat_run_moves_even explains why)
  emat_status run_moves(int64_t count) {
    const int64_t P = shard_world > 1 ? (int64_t)(part_hi - part_lo) : (int64_t)parts.size(), sub = count / P;   // (a sharded run goes through emat_run_moves_sharded)
    if (reference_remainder) return bk(emat_run_local_moves(backend, count));
    return bk(emat_run_moves_even(backend, sub, (int32_t)(count - P * sub)));
  }

  emat_status repartition() {   // run.cpp:110-193 (+ refresh_partition_stencils :87-108)
    const bool verbose = verbose_reports();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now(), t1 = t0, t2 = t0, t3 = t0, t4 = t0, t5 = t0;
    if (device_tree) return repartition_device();
    partition_on_device = false;
    try {
      sync_topology();
      { emat_status ds = ensure_draw(); if (ds) return ds; }
      const std::vector<int32_t>& stencil = *drawn;
      part_kids.clear();
      t1 = now();
      partition_tree(stencil);
      note_partition_stats();
      if (!tree.nodes[tree.root].mfs.empty()) return fail(EMAT_ERR_INTERNAL, "root missations carry from_states");
      normalize_root();
      ++epoch;
      t2 = now();
      build_subtrees();
      t3 = now();
    } catch (const std::exception& ex) { return fail(EMAT_ERR_INTERNAL, ex.what()); }
    parts_uploaded = false; coal_built = false;
    shard_block((int)subtrees.size());
    part_epoch.assign(subtrees.size(), 0);
    if (backend) {
      emat_status st;
      if (!model_pushed) { st = push_model(); if (st) return st; }
      t4 = now();
      st = upload_parts(); if (st) return st;
      t5 = now();
      // a sharded run builds the coalescent parts in stages, with all-reduces the caller owns in between (emat_run_coalescent_begin)
      if (shard_world == 1) { st = build_coalescent(); if (st) return st; }
    }
    if (verbose) fprintf(stderr, "[emat_run] repartition: stencils %.1f ms | partition_tree + normalize_root %.1f ms | build_subtrees %.1f ms | push_model %.1f ms | upload_parts %.1f ms | build_coalescent %.1f ms\n",
                         ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t5, now()));
    return EMAT_OK;
  }

  emat_status reassemble() {   // run.cpp:195-256
    const bool verbose = verbose_reports();
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    auto t0 = now(), t1 = t0, t2 = t0;
    if (device_tree) return reassemble_device();
    try {
      if (backend && parts_uploaded) { emat_status st0 = download_local_parts(); if (st0) return st0; }
      t1 = now();
      if (shard_world > 1 && backend)
        for (size_t p = 0; p < subtrees.size(); ++p)
          if (part_epoch[p] != epoch) return fail(EMAT_ERR_STATE, "part " + std::to_string(p) + " of another rank was not received this cycle (emat_run_unpack_parts)");
      t2 = now();
      for (size_t p = 0; p < subtrees.size(); ++p) if ((size_t)subtrees[p].num_nodes() != parts[p].orig.size()) return fail(EMAT_ERR_INTERNAL, "subtree size changed");
      // Every node of the whole tree is a non-root node of exactly one part (the run's root: the root of the root part),
      // and that part alone writes its time, lists and child links; a cut node's parent link is written by the part
      // above it, as the parent of one of its children.  The parts therefore gather independently.
      parallel_for((int)subtrees.size(), [&](int p) {
        const PartMap& pm = parts[p]; const FlatTree& st = subtrees[p];
        for (int s = 0; s < st.num_nodes(); ++s) {
          const int32_t o = pm.orig[s]; HNode& on = tree.nodes[o];
          const bool owns = s != st.root || p == root_part;
          if (owns) {
            on.t = st.t[s];
            const int m0 = st.mut_offset[s], m1 = st.mut_offset[s + 1], i0 = st.miss_offset[s], i1 = st.miss_offset[s + 1], f0 = st.mfs_offset[s], f1 = st.mfs_offset[s + 1];
            on.muts.resize(m1 - m0); for (int k = m0; k < m1; ++k) on.muts[k - m0] = HMut{st.mut_t[k], st.mut_site[k], st.mut_from[k], st.mut_to[k]};
            on.miss.resize(i1 - i0); for (int k = i0; k < i1; ++k) on.miss[k - i0] = HIv{st.miss_start[k], st.miss_end[k]};
            on.mfs.resize(f1 - f0); for (int k = f0; k < f1; ++k) on.mfs[k - f0] = HFs{st.mfs_site[k], st.mfs_state[k]};
          }
          if (!st.is_tip(s)) {
            int32_t l = pm.orig[st.child0[s]], r = pm.orig[st.child1[s]];
            on.c0 = l; on.c1 = r; tree.nodes[l].parent = o; tree.nodes[r].parent = o;
          }
        }
        if (p == root_part) { const int32_t nr = pm.orig[st.root]; tree.root = nr; tree.nodes[nr].parent = EMAT_NO_NODE; }
      });
    } catch (const std::exception& ex) { return fail(EMAT_ERR_INTERNAL, ex.what()); }
    if (verbose) fprintf(stderr, "[emat_run] reassemble: D2H + decode %.1f ms | per-part download %.1f ms | gather %.1f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, now()));
    return EMAT_OK;
  }
};

}  // namespace emat

using namespace emat;

struct emat_synth { SynthResult res; };
struct emat_run { RunDriver d; };

extern "C" {

emat_status emat_synth_create(const emat_synth_params* p, emat_synth** out) {
  if (!p || !out || p->num_tips < 2 || p->num_sites < 1) return EMAT_ERR_INVALID_ARGUMENT;
  SynthParams sp;
  sp.num_tips = p->num_tips; sp.num_sites = p->num_sites; sp.tip_span = p->tip_span; sp.tip_date_uncertainty = p->tip_date_uncertainty;
  sp.frac_uncertain_tips = p->frac_uncertain_tips; sp.pop_n0 = p->pop_n0; sp.pop_growth = p->pop_growth; sp.mu = p->mu; sp.kappa = p->kappa;
  for (int a = 0; a < 4; ++a) sp.pi[a] = p->pi[a];
  sp.gaps_per_tip = p->gaps_per_tip; sp.mean_gap_len = p->mean_gap_len; sp.seed = p->seed;
  auto* s = new emat_synth;
  try { s->res = make_synthetic_emat(sp); } catch (...) { delete s; return EMAT_ERR_INTERNAL; }
  *out = s;
  return EMAT_OK;
}
void emat_synth_destroy(emat_synth* s) { delete s; }
emat_status emat_synth_get(emat_synth* s, emat_flat_tree* tree_view, const uint8_t** ref, double* t_max_tip) {
  if (!s) return EMAT_ERR_INVALID_ARGUMENT;
  if (tree_view) *tree_view = s->res.tree.view();
  if (ref) *ref = s->res.ref_sequence.data();
  if (t_max_tip) *t_max_tip = s->res.t_max_tip;
  return EMAT_OK;
}

emat_status emat_run_create(emat_backend* backend, const emat_flat_tree* tree, const uint8_t* ref, int32_t L, uint64_t seed, emat_run** out) {
  if (!tree || !ref || !out || L <= 0) return EMAT_ERR_INVALID_ARGUMENT;
  if (!validate_flat_tree(*tree, L).empty()) return EMAT_ERR_INVALID_ARGUMENT;
  auto* r = new emat_run;
  r->d.backend = backend; r->d.tree = HTree::from_view(*tree); r->d.ref.assign(ref, ref + L); r->d.L = L; r->d.seed = seed; r->d.bitgen = SplitMix64(seed ^ 0xD1B54A32D192ED03ull);
  *out = r;
  return EMAT_OK;
}
emat_status emat_run_destroy(emat_run* r) { delete r; return EMAT_OK; }
const char* emat_run_last_error(const emat_run* r) { return r ? r->d.last_error.c_str() : "null run"; }

emat_status emat_run_set_max_part_nodes(emat_run* r, int32_t n) { if (!r || n < -1) return EMAT_ERR_INVALID_ARGUMENT; r->d.max_part_nodes = n; return EMAT_OK; }
/* Test hook: the cut nodes the LAST repartition's draw (same stencil, same random stream of the refinement) gives on the tree AS IT IS NOW.
 * Between a repartition and the reassemble that follows a pass, the moves only re-hang and re-time nodes within parts; refine_stencil
 * reads nothing such a pass can change, so the draw on the tree after the pass must be the draw on the tree before it -- the premise of
 * the argument that the part-size limit leaves the sampler's stationary distribution alone (refine_stencil).  Changes no state. */
emat_status emat_run_debug_redraw_partition(emat_run* r, int32_t* cut_nodes, int32_t* num_cut_nodes) {
  if (!r || !num_cut_nodes) return EMAT_ERR_INVALID_ARGUMENT;
  RunDriver& d = r->d;
  if (d.follow) return d.fail(EMAT_ERR_STATE, "this run takes its draws from another one (emat_run_follow_draws): ask that one");
  if (d.last_pick < 0 || d.last_pick >= (int)d.stencils.size()) return d.fail(EMAT_ERR_STATE, "emat_run_repartition first");
  if (d.device_tree) { emat_status st = d.fetch_device_topology(); if (st) return st; } else d.sync_topology();
  const uint64_t epoch_now = d.epoch; const int extra_now = d.last_extra_cuts;
  d.epoch = d.last_refine_epoch;
  std::vector<int32_t> cuts;
  try { cuts = d.refine_stencil(d.stencils[(size_t)d.last_pick]); } catch (const std::exception& ex) { d.epoch = epoch_now; d.last_extra_cuts = extra_now; return d.fail(EMAT_ERR_INTERNAL, ex.what()); }
  d.epoch = epoch_now; d.last_extra_cuts = extra_now;
  std::sort(cuts.begin(), cuts.end());
  const int32_t cap = *num_cut_nodes; *num_cut_nodes = (int32_t)cuts.size();
  if (cap < (int32_t)cuts.size() || !cut_nodes) return d.fail(EMAT_ERR_BUFFER_TOO_SMALL, "emat_run_debug_redraw_partition: array too small");
  std::copy(cuts.begin(), cuts.end(), cut_nodes);
  return EMAT_OK;
}
emat_status emat_run_partition_stats(emat_run* r, int32_t* num_parts, int32_t* largest_part_nodes, int32_t* extra_cuts, int32_t* max_part_nodes_in_effect) {
  if (!r) return EMAT_ERR_INVALID_ARGUMENT;
  if (num_parts) *num_parts = r->d.last_num_parts;
  if (largest_part_nodes) *largest_part_nodes = r->d.last_largest_part;
  if (extra_cuts) *extra_cuts = r->d.last_extra_cuts;
  if (max_part_nodes_in_effect) *max_part_nodes_in_effect = r->d.effective_max_part_nodes(r->d.device_tree ? (size_t)r->d.tp_n : r->d.tree.nodes.size());
  return EMAT_OK;
}
emat_status emat_run_set_num_parts(emat_run* r, int32_t n) { if (!r || n < 1) return EMAT_ERR_INVALID_ARGUMENT; r->d.num_parts = n; r->d.stencils.clear(); return EMAT_OK; }
emat_status emat_run_set_hky(emat_run* r, double mu, double kappa, const double pi[4], const double* nu_l) {
  if (!r || !pi || !(mu >= 0) || !(kappa > 0)) return EMAT_ERR_INVALID_ARGUMENT;
  r->d.hky_mu = mu; r->d.hky_kappa = kappa; for (int a = 0; a < 4; ++a) r->d.hky_pi[a] = pi[a];
  if (nu_l) r->d.nu_l.assign(nu_l, nu_l + r->d.L); else r->d.nu_l.clear();
  r->d.have_hky = true; r->d.model_pushed = false;
  return EMAT_OK;
}
emat_status emat_run_set_pop_model(emat_run* r, const emat_pop_model* pm) {
  if (!r || !pm) return EMAT_ERR_INVALID_ARGUMENT;
  r->d.pop = *pm;
  if (pm->kind == EMAT_POP_SKYGRID) { r->d.sky_x.assign(pm->skygrid_x, pm->skygrid_x + pm->skygrid_num_knots); r->d.sky_g.assign(pm->skygrid_gamma, pm->skygrid_gamma + pm->skygrid_num_knots); }
  r->d.have_pop = true; r->d.coal_built = false;
  return EMAT_OK;
}
emat_status emat_run_set_coalescent_t_step(emat_run* r, double t_step) { if (!r || !(t_step > 0)) return EMAT_ERR_INVALID_ARGUMENT; r->d.t_step = t_step; r->d.t_step_set = true; return EMAT_OK; }
emat_status emat_run_set_flags(emat_run* r, int32_t odin, int32_t topo) { if (!r) return EMAT_ERR_INVALID_ARGUMENT; r->d.only_displacing_inner_nodes = odin; r->d.topology_moves_enabled = topo; r->d.model_pushed = false; return EMAT_OK; }

emat_status emat_run_set_device_tree(emat_run* r, int32_t on) {
  if (!r) return EMAT_ERR_INVALID_ARGUMENT;
  RunDriver& d = r->d;
  if (on) {
    if (!d.backend) return d.fail(EMAT_ERR_NO_DEVICE, "a device-resident tree needs a backend");
    if (d.parts_uploaded) return d.fail(EMAT_ERR_STATE, "reassemble first");
    d.device_tree = true; d.device_tree_uploaded = false;   // uploaded at the next repartition
    return EMAT_OK;
  }
  if (d.device_tree) {
    if (d.parts_uploaded) return d.fail(EMAT_ERR_STATE, "reassemble first");
    emat_status st = d.ensure_host_tree(); if (st) return st;
    d.device_tree = false; d.device_tree_uploaded = false;
  }
  return EMAT_OK;
}
emat_status emat_run_note_device_reassembled(emat_run* r, int32_t num_root_deltas, const int32_t* site, const uint8_t* to) {
  if (!r || num_root_deltas < 0 || (num_root_deltas > 0 && (!site || !to))) return EMAT_ERR_INVALID_ARGUMENT;
  RunDriver& d = r->d;
  if (!d.device_tree || !d.parts_uploaded) return d.fail(EMAT_ERR_STATE, "no device-resident parts are out");
  for (int k = 0; k < num_root_deltas; ++k) { if (site[k] < 0 || site[k] >= d.L || to[k] > 3) return EMAT_ERR_INVALID_ARGUMENT; d.ref[site[k]] = to[k]; }
  d.parts_uploaded = false; d.host_tree_stale = true;
  return EMAT_OK;
}
emat_status emat_run_repartition(emat_run* r) { if (!r) return EMAT_ERR_INVALID_ARGUMENT; return r->d.repartition(); }
emat_status emat_run_num_parts(emat_run* r, int32_t* n, int32_t* root_part) { if (!r) return EMAT_ERR_INVALID_ARGUMENT; if (n) *n = (int)r->d.parts.size(); if (root_part) *root_part = r->d.root_part; return EMAT_OK; }
emat_status emat_run_part_sizes(emat_run* r, int32_t p, int32_t* nn, int32_t* nm, int32_t* ni, int32_t* nf) {
  if (r && r->d.device_tree) return r->d.fail(EMAT_ERR_STATE, "with a device-resident tree the parts exist only on the device (emat_part_get_sizes / emat_part_download of the backend)");
  if (!r || p < 0 || p >= (int)r->d.subtrees.size()) return EMAT_ERR_INVALID_ARGUMENT;
  const FlatTree& t = r->d.subtrees[p];
  if (nn) *nn = t.num_nodes(); if (nm) *nm = t.num_muts(); if (ni) *ni = t.num_intervals(); if (nf) *nf = t.num_from_states();
  return EMAT_OK;
}
static emat_status copy_out(const FlatTree& t, emat_flat_tree* out) {
  const int n = t.num_nodes();
  if (out->num_nodes < n || out->cap_muts < t.num_muts() || out->cap_intervals < t.num_intervals() || out->cap_from_states < t.num_from_states()) return EMAT_ERR_BUFFER_TOO_SMALL;
  out->num_nodes = n; out->root = t.root;
  std::copy(t.parent.begin(), t.parent.end(), out->parent); std::copy(t.child0.begin(), t.child0.end(), out->child0); std::copy(t.child1.begin(), t.child1.end(), out->child1);
  std::copy(t.t.begin(), t.t.end(), out->t); std::copy(t.t_min.begin(), t.t_min.end(), out->t_min); std::copy(t.t_max.begin(), t.t_max.end(), out->t_max);
  std::copy(t.mut_offset.begin(), t.mut_offset.end(), out->mut_offset); std::copy(t.mut_site.begin(), t.mut_site.end(), out->mut_site);
  std::copy(t.mut_from.begin(), t.mut_from.end(), out->mut_from); std::copy(t.mut_to.begin(), t.mut_to.end(), out->mut_to); std::copy(t.mut_t.begin(), t.mut_t.end(), out->mut_t);
  std::copy(t.miss_offset.begin(), t.miss_offset.end(), out->miss_offset); std::copy(t.miss_start.begin(), t.miss_start.end(), out->miss_start); std::copy(t.miss_end.begin(), t.miss_end.end(), out->miss_end);
  std::copy(t.mfs_offset.begin(), t.mfs_offset.end(), out->mfs_offset); std::copy(t.mfs_site.begin(), t.mfs_site.end(), out->mfs_site); std::copy(t.mfs_state.begin(), t.mfs_state.end(), out->mfs_state);
  return EMAT_OK;
}
emat_status emat_run_part_get(emat_run* r, int32_t p, emat_flat_tree* out, int32_t* incl_root, uint64_t* seed) {
  if (!r || !out || p < 0 || p >= (int)r->d.subtrees.size()) return EMAT_ERR_INVALID_ARGUMENT;
  if (incl_root) *incl_root = p == r->d.root_part ? 1 : 0;
  if (seed) *seed = r->d.part_seeds[p];
  return copy_out(r->d.subtrees[p], out);
}
emat_status emat_run_part_put(emat_run* r, int32_t p, const emat_flat_tree* st) {
  if (!r || !st || p < 0 || p >= (int)r->d.subtrees.size()) return EMAT_ERR_INVALID_ARGUMENT;
  if (!validate_flat_tree(*st, r->d.L).empty() || st->num_nodes != (int)r->d.parts[p].orig.size()) return EMAT_ERR_INVALID_ARGUMENT;
  r->d.subtrees[p] = FlatTree::from_view(*st);
  return EMAT_OK;
}
emat_status emat_run_push_params(emat_run* r) {
  if (!r) return EMAT_ERR_INVALID_ARGUMENT;
  if (!r->d.backend) return EMAT_OK;
  emat_status st = r->d.push_model(); if (st) return st;
  if (!r->d.parts_uploaded) return r->d.fail(EMAT_ERR_STATE, "repartition first");
  return r->d.build_coalescent();   // run.cpp:267-275 rebuilds the coalescent parts at every push
}
emat_status emat_run_moves(emat_run* r, int64_t count) {
  if (!r || count < 0) return EMAT_ERR_INVALID_ARGUMENT;
  if (!r->d.backend) return r->d.fail(EMAT_ERR_NO_DEVICE, "no backend attached: the host driver never runs moves itself");
  if (!r->d.parts_uploaded) return r->d.fail(EMAT_ERR_STATE, "repartition first");
  return r->d.run_moves(count);
}
emat_status emat_run_reassemble(emat_run* r) { if (!r) return EMAT_ERR_INVALID_ARGUMENT; return r->d.reassemble(); }

/* Drivers of one process that are bound to draw the same partitions (same seed, same tree: emat_multi's shards) draw once: `follower` takes
 * the cut nodes `leader` drew for the same cycle instead of drawing them again.  The leader draws at its own emat_run_repartition, or ahead of
 * it with emat_run_draw_partition (so that leader and followers can then cut side by side).  NULL leader = draw for itself again. */
emat_status emat_run_follow_draws(emat_run* follower, emat_run* leader) {
  if (!follower || follower == leader) return EMAT_ERR_INVALID_ARGUMENT;
  RunDriver& f = follower->d;
  if (leader) {
    const RunDriver& l = leader->d;
    if (l.seed != f.seed || l.tree.nodes.size() != f.tree.nodes.size() || l.epoch != f.epoch || l.num_parts != f.num_parts || l.max_part_nodes != f.max_part_nodes)
      return f.fail(EMAT_ERR_INVALID_ARGUMENT, "emat_run_follow_draws: leader and follower must be runs of the same seed, tree, cycle and partition settings");
    if (l.follow) return f.fail(EMAT_ERR_INVALID_ARGUMENT, "emat_run_follow_draws: the leader itself follows another run");
  }
  f.follow = leader ? &leader->d : nullptr;
  return EMAT_OK;
}
emat_status emat_run_draw_partition(emat_run* r) {
  if (!r) return EMAT_ERR_INVALID_ARGUMENT;
  RunDriver& d = r->d;
  if (d.follow) return d.fail(EMAT_ERR_STATE, "emat_run_draw_partition: this run takes its draws from another one");
  if (d.drawn && d.drawn_for_epoch == d.epoch) return EMAT_OK;   // already drawn for the cycle to come
  try {
    if (d.device_tree && d.device_tree_uploaded) { emat_status st = d.fetch_device_topology(); if (st) return st; }
    else { if (d.device_tree) d.normalize_root(); d.sync_topology(); }   // (before the first upload: the host's copy is the tree)
    d.draw_partition();
  } catch (const std::exception& ex) { return d.fail(EMAT_ERR_INTERNAL, ex.what()); }
  return EMAT_OK;
}
emat_status emat_run_set_shard(emat_run* r, int32_t rank, int32_t world) {
  if (!r || world < 1 || rank < 0 || rank >= world) return EMAT_ERR_INVALID_ARGUMENT;
  r->d.shard_rank = rank; r->d.shard_world = world; r->d.parts_uploaded = false;
  return EMAT_OK;
}
emat_status emat_run_shard_range(emat_run* r, int32_t* part_lo, int32_t* part_hi, int32_t* local_root_part) {
  if (!r) return EMAT_ERR_INVALID_ARGUMENT;
  if (part_lo) *part_lo = r->d.part_lo;
  if (part_hi) *part_hi = r->d.part_hi;
  if (local_root_part) *local_root_part = (r->d.root_part >= r->d.part_lo && r->d.root_part < r->d.part_hi) ? r->d.root_part - r->d.part_lo : -1;
  return EMAT_OK;
}
emat_status emat_run_coalescent_begin(emat_run* r, double* local_t_min, double* local_t_max) {
  if (!r || !local_t_min || !local_t_max) return EMAT_ERR_INVALID_ARGUMENT;
  RunDriver& d = r->d;
  if (!d.backend) return d.fail(EMAT_ERR_NO_DEVICE, "no backend attached");
  if (!d.parts_uploaded) return d.fail(EMAT_ERR_STATE, "repartition first");
  if (!d.have_pop) return d.fail(EMAT_ERR_STATE, "emat_run_set_pop_model must be called first");
  emat_pop_model pm = d.pop; pm.skygrid_x = d.sky_x.data(); pm.skygrid_gamma = d.sky_g.data();
  const int local_root = (d.root_part >= d.part_lo && d.root_part < d.part_hi) ? d.root_part - d.part_lo : -1;
  emat_status st = d.bk(emat_coalescent_begin(d.backend, &pm, local_root, d.t_step_set ? d.t_step : d.default_t_step(), local_t_min, local_t_max));
  if (st == EMAT_OK) d.coal_built = true;   // the caller finishes the stages on the backend
  return st;
}
emat_status emat_run_moves_sharded(emat_run* r, int64_t count) {   // Run::run_local_moves (run.cpp:682-693) over ALL parts of the run
  if (!r || count < 0) return EMAT_ERR_INVALID_ARGUMENT;
  RunDriver& d = r->d;
  if (!d.backend) return d.fail(EMAT_ERR_NO_DEVICE, "no backend attached: the host driver never runs moves itself");
  if (!d.parts_uploaded) return d.fail(EMAT_ERR_STATE, "repartition first");
  const int64_t P = (int64_t)d.parts.size(), sub = count / P, rem = count - P * sub;
  // the remainder is spread one move per part over the first parts of the run (see emat_run_moves_even)
  return d.bk(emat_run_moves_even(d.backend, sub, (int32_t)std::max<int64_t>(0, std::min<int64_t>(rem - d.part_lo, d.part_hi - d.part_lo))));
}
// For calc_Ttwiddle_l: the whole-tree branch length hanging below every boundary tip of the LOCAL parts, from the lengths
// inside every part of the run and the tree of parts (a part's boundary tips are the cut nodes of the parts below it).
emat_status emat_run_Ttwiddle_ext(emat_run* r, const double* tree_length_of_part, int32_t* ext_offset, int32_t* ext_node, double* ext_length, int32_t capacity, int32_t* count) {
  if (!r || !tree_length_of_part || !ext_offset || !count || capacity < 0 || (capacity > 0 && (!ext_node || !ext_length))) return EMAT_ERR_INVALID_ARGUMENT;
  RunDriver& d = r->d;
  { emat_status st = d.ensure_partition_on_host(); if (st) return st; }
  const int P = (int)d.parts.size();
  if (P == 0 || (int)d.part_kids.size() != P) return d.fail(EMAT_ERR_STATE, "repartition first");
  std::vector<int32_t> part_of_cut(d.tree.nodes.size(), -1);
  for (int p = 0; p < P; ++p) part_of_cut[d.parts[p].cut_point] = p;
  // children of each part in the tree of parts: (subtree node, part below).  A part's tips stay tips whatever the moves
  // do, so the partition's own record of them (which also exists when the parts themselves live only on the device) serves.
  std::vector<std::vector<std::pair<int32_t, int32_t>>> kids(P);
  for (int p = 0; p < P; ++p) {
    const auto& pk = d.part_kids[p];
    for (int s = 1; s < (int)pk.size(); ++s) if (pk[s].first == EMAT_NO_NODE) { const int q = part_of_cut[d.parts[p].orig[s]]; if (q >= 0 && q != p) kids[p].push_back({s, q}); }
  }
  // total length below each part's cut node: its own branches plus everything below its boundary tips (children first)
  std::vector<double> below(P, -1.0);
  std::vector<std::pair<int, size_t>> stack; stack.push_back({d.root_part, 0});
  while (!stack.empty()) {
    auto& [p, k] = stack.back();
    if (k < kids[p].size()) { const int q = kids[p][k].second; ++k; stack.push_back({q, 0}); }
    else { double t = tree_length_of_part[p]; for (auto& kv : kids[p]) t += below[kv.second]; below[p] = t; stack.pop_back(); }
  }
  int32_t n = 0;
  for (int p = d.part_lo; p < d.part_hi; ++p) {
    ext_offset[p - d.part_lo] = n;
    for (auto& kv : kids[p]) { if (n < capacity) { ext_node[n] = kv.first; ext_length[n] = below[kv.second]; } ++n; }
  }
  ext_offset[d.part_hi - d.part_lo] = n;
  *count = n;
  return n <= capacity ? EMAT_OK : d.fail(EMAT_ERR_BUFFER_TOO_SMALL, "emat_run_Ttwiddle_ext: arrays too small");
}
// calc_Ttwiddle_l of the whole tree from the parts on the device (single process).
emat_status emat_run_get_Ttwiddle_l(emat_run* r, double* Ttwiddle_l) {
  if (!r || !Ttwiddle_l) return EMAT_ERR_INVALID_ARGUMENT;
  RunDriver& d = r->d;
  if (!d.backend) return d.fail(EMAT_ERR_NO_DEVICE, "no backend attached");
  if (!d.parts_uploaded) return d.fail(EMAT_ERR_STATE, "repartition first");
  if (d.shard_world > 1) return d.fail(EMAT_ERR_STATE, "a sharded run gathers the part lengths and sums S, R across ranks itself (see emat_backend.h)");
  const int P = (int)d.parts.size();
  std::vector<double> len(P);
  emat_status st = d.bk(emat_get_part_tree_lengths(d.backend, len.data())); if (st) return st;
  std::vector<int32_t> off(P + 1), node(P); std::vector<double> val(P); int32_t cnt = 0;
  st = emat_run_Ttwiddle_ext(r, len.data(), off.data(), node.data(), val.data(), P, &cnt); if (st) return st;
  std::vector<double> S(d.L), R(d.L); double T = 0.0;
  st = d.bk(emat_Ttwiddle_l_partial(d.backend, off.data(), node.data(), val.data(), S.data(), R.data(), &T)); if (st) return st;
  return d.bk(emat_Ttwiddle_l_finish(d.backend, S.data(), R.data(), T, Ttwiddle_l));
}

emat_status emat_run_pack_local_parts(emat_run* r, uint8_t* buf, uint64_t capacity, uint64_t* bytes_needed) {
  if (!r) return EMAT_ERR_INVALID_ARGUMENT;
  try { return r->d.pack_local_parts(buf, capacity, bytes_needed); } catch (const std::exception& ex) { return r->d.fail(EMAT_ERR_INTERNAL, ex.what()); }
}
emat_status emat_run_unpack_parts(emat_run* r, const uint8_t* buf, uint64_t bytes) {
  if (!r || (!buf && bytes)) return EMAT_ERR_INVALID_ARGUMENT;
  try { return r->d.unpack_parts(buf, bytes); } catch (const std::exception& ex) { return r->d.fail(EMAT_ERR_INTERNAL, ex.what()); }
}
emat_status emat_run_set_reference_remainder(emat_run* r, int32_t on) { if (!r) return EMAT_ERR_INVALID_ARGUMENT; r->d.reference_remainder = on != 0; return EMAT_OK; }
emat_status emat_run_set_paranoid(emat_run* r, int32_t on) { if (!r) return EMAT_ERR_INVALID_ARGUMENT; r->d.paranoid = on != 0; return EMAT_OK; }
emat_status emat_run_do_mcmc_steps(emat_run* r, int64_t steps, int64_t per_cycle) {   // run.cpp:622-657 minus global moves
  if (!r || steps < 0) return EMAT_ERR_INVALID_ARGUMENT;
  if (!r->d.backend) return r->d.fail(EMAT_ERR_NO_DEVICE, "no backend attached");
  if (r->d.shard_world > 1) return r->d.fail(EMAT_ERR_STATE, "a sharded run is cycled by its caller, who owns the collectives (see emat_host.h)");
  if (per_cycle <= 0) per_cycle = 50 * (int64_t)r->d.tree.nodes.size();
  int64_t done = 0;
  const bool verbose = verbose_reports();
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  while (done < steps) {
    const auto t0 = now();
    HostLaps laps;
    emat_status st = r->d.repartition(); if (st) return st;
    const auto t1 = now(); laps.mark("cycle: 1 repartition");
    int64_t k = std::min(per_cycle, steps - done);
    st = r->d.run_moves(k); if (st) return st;
    if (r->d.paranoid) { st = r->d.bk(emat_check_derived(r->d.backend, 1.0, nullptr, nullptr)); if (st) return st; }
    const auto t2 = now(); laps.mark("cycle: 2 run_moves (launch)");
    st = r->d.reassemble(); if (st) return st;
    laps.mark("cycle: 3 reassemble (waits for the moves)");
    done += k;
    if (verbose) fprintf(stderr, "[emat_run] cycle: repartition %.1f ms | launch of the moves %.1f ms | reassemble (waits for the moves) %.1f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, now()));
  }
  if (!r->d.device_tree) r->d.normalize_root();   // (a device-resident tree is normalised by every reassemble)
  return EMAT_OK;
}
emat_status emat_run_tree_sizes(emat_run* r, int32_t* nn, int32_t* nm, int32_t* ni, int32_t* nf) {
  if (!r) return EMAT_ERR_INVALID_ARGUMENT;
  { emat_status st = r->d.ensure_host_tree(); if (st) return st; }
  int m = 0, i = 0, f = 0; for (auto& nd : r->d.tree.nodes) { m += (int)nd.muts.size(); i += (int)nd.miss.size(); f += (int)nd.mfs.size(); }
  if (nn) *nn = (int)r->d.tree.nodes.size(); if (nm) *nm = m; if (ni) *ni = i; if (nf) *nf = f;
  return EMAT_OK;
}
emat_status emat_run_tree_get(emat_run* r, emat_flat_tree* out, uint8_t* ref) {
  if (!r || !out) return EMAT_ERR_INVALID_ARGUMENT;
  { emat_status st = r->d.ensure_host_tree(); if (st) return st; }
  if (ref) std::copy(r->d.ref.begin(), r->d.ref.end(), ref);
  return copy_out(r->d.tree.to_flat(), out);
}
emat_status emat_run_t_max_tip(emat_run* r, double* t) { if (!r || !t) return EMAT_ERR_INVALID_ARGUMENT; *t = r->d.t_max_tip(); return EMAT_OK; }

}  // extern "C"
