// emat_gtree_kernels.hpp -- the whole tree resident in HBM: cutting it into part slabs and gathering the parts back
// without a host round trip of nodes, mutations and missations.
//
// Reference: Run::repartition (core/run.cpp:110-193) copies every part's nodes out of the whole Phylo_tree into a
// Subrun's tree, giving each sub-root the sequence state at its cut point (reconstruct_missing_sites_at /
// view_of_sequence_at, core/phylo_tree_calc.cpp:19-56); Run::reassemble (core/run.cpp:195-256) copies them back.
// Here the whole tree lives in HBM as node arrays + three record heaps, a partition is three int arrays from the host
// (which only ever sees the topology), and one wavefront per part does the copying in both directions.
//
// Included by emat_backend.hip (needs emat_slab.hpp and k_wave).
#ifndef EMAT_GTREE_KERNELS_HPP_
#define EMAT_GTREE_KERNELS_HPP_

namespace emat {

struct GList { uint32_t off, cnt; };   // records [off, off + cnt) of the heap of its kind

// What a walk from a cut point to the root reads of every node on its way, in ONE 16-byte request instead of three (parent, and the
// headers of the node's missations and mutations): k_gt_measure makes 13 000 such walks side by side and is bound by the number of
// requests it issues.  A copy (k_gt_pack_climb), made after the tree's lists and links were last written.  Counts fit 16 bits: no
// list of the tree is longer than k_gt_max_list.
struct GClimb { int32_t parent; uint32_t miss_off, muts_off; uint16_t miss_cnt, muts_cnt; };
static_assert(sizeof(GClimb) == 16, "GClimb is one 16-byte load");

struct GTreeDev {
  int32_t n_nodes;
  GClimb* climb;                     // [n_nodes], see above
  int32_t* root;                     // [1]
  int32_t* parent; int32_t* c0; int32_t* c1;
  double* t; const float* t_min; const float* t_max;
  GList* muts; GList* miss; GList* mfs;   // per node
  MutRec* mut_heap; IvRec* iv_heap; FsRec* fs_heap;
  uint32_t mut_cap, iv_cap, fs_cap;
  uint32_t* tops;                    // [3] next free record of each heap (a gather rebuilds all three from zero)
};

struct GPartition {
  int32_t num_parts, root_part;
  const int32_t* part_off;           // [P + 1] into orig / kid0 / kid1
  const int32_t* orig;               // part-local node -> whole-tree node; local node 0 is the part's cut point
  const int32_t* kid0; const int32_t* kid1;   // part-local children (EMAT_NO_NODE for the part's tips)
  const int32_t* lpar;               // part-local parent (EMAT_NO_NODE for local node 0), or null when the coalescent tables come from the host
};

enum GStatus : int32_t { k_gt_ok = 0, k_gt_cut_state_overflow = 1, k_gt_pool_overflow = 2, k_gt_list_too_long = 3, k_gt_inconsistent = 4, k_gt_heap_overflow = 5, k_gt_root_deltas_overflow = 6 };

struct GMeasure {                    // per part, written by k_gt_measure
  int32_t n_nodes, num_muts;
  uint32_t content_bytes;            // heap bytes of the part's lists, every list rounded up to 16 (encode_slab's layout)
  uint32_t root_muts_off, root_muts_cnt, root_miss_off, root_miss_cnt;   // the cut point's state, in the pools
  int32_t status;
  double t_min, t_max, t_max_exact;  // CoalBuilder::local_range: tips count with their float bounds, t_max_exact is the latest node time in full precision
};

// The cut-point states of all parts, bump-allocated by k_gt_measure.  One counter per pool would take 13 000 atomic additions to the same
// word, one behind the other (0.3 ms of a 0.45-ms kernel): each pool is k_gt_pool_lanes regions with a counter each (64 bytes apart), part p
// takes from region p mod k_gt_pool_lanes.  `tops` [2 x k_gt_pool_lanes x 16]: the counters keep counting past a full region, so the fullest
// one tells the host how much room to give.
constexpr uint32_t k_gt_pool_lanes = 64, k_gt_pool_stride = 16;
struct GPools { MutRec* muts; IvRec* ivs; uint32_t mut_cap, iv_cap; uint32_t* tops; };

struct GPartDesc {                   // per part, from the host: geometry of the slab + what its header starts with
  uint32_t slab_bytes, heap_bytes, scratch_bytes;
  int32_t cell_cap, trace_cap;
  uint32_t flags;
  uint64_t rng_key, rng_counter, rng_spare; uint32_t rng_has_spare;
  int32_t cell_first, n_cells, n_cells_total;
  double t_ref, t_step;
  uint64_t cells_off;                // host tables: byte offset of the part's packed cells (4 double arrays [n_cells] then 1 int32 array [n_cells]);
                                     // device tables: index of the part's window in the k_bar_p / k_twiddle_bar_p pools
  int32_t coal_first_active;         // the reference's first_cell of the part (>= cell_first): where it is active and draws k_twiddle_bar_p
  int32_t pad;
};

struct GCoal {                       // the whole grid, built on the device (null pointers: the tables came from the host, packed in `cells`)
  int32_t num_cells; double t_ref, t_step;
  double* kbar_pool; double* ktw_pool;   // every part's window [cell_first, n_cells_total), back to back (GPartDesc::cells_off)
  double* k_bar; double* k_tw; double* popsize; int32_t* num_active;   // [num_cells]
  double* ts_over_pop;                 // [num_cells] t_step / popsize: with k_tw and num_active the SharedCells the moves read
  int32_t* status;
};

constexpr int k_gt_max_cut_intervals = 2048, k_gt_small_cut_intervals = 256;   // the large variant fills the 64 KB of static LDS a workgroup may declare
constexpr int k_gt_max_cut_deltas = 3968, k_gt_small_cut_deltas = 256;
constexpr int k_gt_max_root_deltas = 256;   // how many changes of the root sequence k_gt_gather caches in LDS (more are read from HBM)
constexpr uint32_t k_gt_max_list = k_max_list_upload;   // ListRef counts are 16 bits (same limit as emat_part_upload)

__device__ inline uint32_t gt_a16(uint32_t x) { return (x + 15u) & ~15u; }
__device__ inline bool gt_iv_contains(const IvRec* v, int n, int l) {   // sorted, disjoint, half-open
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (v[mid].start <= l) lo = mid + 1; else hi = mid; }
  return lo > 0 && l < v[lo - 1].end;
}
__device__ inline uint32_t wave_incl_scan_u32(uint32_t x, int lane) {
  for (int d = 1; d < k_wave; d <<= 1) { uint32_t y = __shfl_up(x, d, k_wave); if (lane >= d) x += y; }
  return x;
}
__device__ inline uint32_t wave_sum_u32(uint32_t x) {
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, k_wave);
  return __shfl(x, 0, k_wave);
}

// ---- partition_tree (tree_partitioning.h:88-135) on the device: one thread per part walks its part down from the cut point,
//      depth first, right child first -- the order in which the host's LIFO work list expands nodes -- handing out part-local
//      indices exactly as it does (two consecutive ones to the children of every node it expands).  No stack: in a binary
//      tree with parent links the way back up is known.  mode 0 counts the part's nodes, mode 1 writes the arrays.
// (Round 5: a thread keeps the left siblings it still has to visit on a stack of its own in LDS, with the local index each was given,
// and reads a node's two children as one pair (`kids`, kept current by every reassemble): one round trip to memory per node instead of
// the four of finding the way back up through parent links -- the kernel is as long as the walk of the largest part is deep in
// dependent loads.  A part deeper than the stack falls back to the parent links for the rest of its walk: what was on the stack is
// exactly what walking up finds again.)
constexpr int k_gt_part_stack = 48;
__global__ void __launch_bounds__(64) k_gt_partition(GTreeDev g, const int2* kids, const uint8_t* is_cut, const int32_t* cut_of_part, int num_parts, int mode, int32_t* sizes,
                                                     const int32_t* part_off, int32_t* orig, int32_t* kid0, int32_t* kid1, int32_t* lpar, int32_t* lidx, const int32_t* gate) {
  __shared__ int2 pending[k_gt_part_stack][64];   // [depth][thread]
  const int p = blockIdx.x * blockDim.x + threadIdx.x, me = threadIdx.x;
  if (p >= num_parts) return;
  if (gate && *gate != 0) return;   // (k_gt_part_offsets found the counts wrong: the offsets are not to be written through)
  const int32_t cut = cut_of_part[p];
  const int b = mode ? part_off[p] : 0;
  int cnt = 1, sp = 0;
  bool stacked = true;
  if (mode) { orig[b] = cut; lpar[b] = EMAT_NO_NODE; }
  int32_t cur = cut; int dst = 0;
  for (;;) {
    const int2 k = kids[cur];
    if (k.x != EMAT_NO_NODE && (cur == cut || !is_cut[cur])) {   // expanded: its children join the part
      const int dl = cnt, dr = cnt + 1; cnt += 2;
      if (mode) { orig[b + dl] = k.x; orig[b + dr] = k.y; kid0[b + dst] = dl; kid1[b + dst] = dr; lpar[b + dl] = dst; lpar[b + dr] = dst; lidx[k.x] = dl; lidx[k.y] = dr; }
      if (stacked) { if (sp < k_gt_part_stack) { pending[sp][me] = make_int2(k.x, dl); ++sp; } else stacked = false; }
      cur = k.y; dst = dr;
      continue;
    }
    if (mode) { kid0[b + dst] = EMAT_NO_NODE; kid1[b + dst] = EMAT_NO_NODE; }   // a tip of the part
    if (stacked) {
      if (sp == 0) break;
      --sp; cur = pending[sp][me].x; dst = pending[sp][me].y;
      continue;
    }
    bool done = false;
    for (;;) {   // back up to the nearest ancestor whose left subtree is still to do
      if (cur == cut) { done = true; break; }
      const int32_t par = g.parent[cur];
      const int2 pk = kids[par];
      if (cur == pk.y) { cur = pk.x; dst = mode ? lidx[cur] : 0; break; }
      cur = par;
    }
    if (done) break;
  }
  if (!mode) sizes[p] = cnt;
}

// What the host's partitioner reads of the tree every cycle: each node's two children side by side (one cache line per visit of
// its walks), 8 bytes a node instead of the 24 of parent, children and time.
__global__ void k_gt_pack_climb(GTreeDev g) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= g.n_nodes) return;
  const GList mi = g.miss[v], mu = g.muts[v];
  GClimb c; c.parent = g.parent[v]; c.miss_off = mi.off; c.muts_off = mu.off; c.miss_cnt = (uint16_t)mi.cnt; c.muts_cnt = (uint16_t)mu.cnt;
  g.climb[v] = c;
}
__global__ void k_gt_pack_kids(GTreeDev g, int2* kids) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < g.n_nodes) kids[v] = make_int2(g.c0[v], g.c1[v]);
}

// The parts' offsets from their sizes, between the two passes of k_gt_partition: one workgroup, so that the host need not take
// the sizes, add them up and send the sums back while the device waits.  status: 3 = an empty part, 2 = the sizes do not add up to
// `expect_total` (the cut nodes do not partition the tree); mode 1 and whatever else is queued behind it look at it first.
__global__ void __launch_bounds__(1024) k_gt_part_offsets(const int32_t* sizes, int num_parts, long long expect_total, int32_t* part_off, int32_t* status) {
  __shared__ long long part[1024];
  const int t = threadIdx.x, per = (num_parts + 1023) / 1024, b = t * per, e = min(num_parts, b + per);
  long long s = 0; bool bad = false;
  for (int i = b; i < e; ++i) { const int v = sizes[i]; if (v < 1) bad = true; s += v; }
  part[t] = s;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) { const long long v = t >= d ? part[t - d] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
  const long long total = part[1023];
  long long run = part[t] - s;
  if (total == expect_total) for (int i = b; i < e; ++i) { part_off[i] = (int32_t)run; run += sizes[i]; }
  if (t == 1023) { part_off[num_parts] = (int32_t)(total == expect_total ? total : 0); if (total != expect_total) atomicMax(status, 2); }
  if (bad) atomicMax(status, 3);
}

// ---- pass 1 of a repartition: the state at every cut point + how much list content every part holds ------------
// State at a cut point c = the sites missing at c (union of the missations from c up to the root) and the net changes
// reference sequence -> sequence at c, sorted by site, over the sites present at c.  The walk goes UP from c, so later
// mutations are met first: a site's entry keeps the `to` of the first mutation met and takes the `from` of every
// further one; entries that end with from == to cancel.
struct GCutDelta { int32_t site; uint8_t from, to; uint16_t pad; };

// Compiled twice: with room for a few hundred intervals and changes (6 KB of LDS: 26 workgroups per CU -- what bounds the kernel
// is how many pointer chases are in flight) for all parts, and with the full capacities for the parts that overflowed it
// (`part_list`, or null for all parts).
template <int kMaxIv, int kMaxDl>
__global__ void __launch_bounds__(k_wave) k_gt_measure(GTreeDev g, GPartition pt, GPools pools, const uint8_t* ref, GMeasure* out, const int32_t* part_list, const int32_t* gate) {
  __shared__ IvRec acc[2][kMaxIv];
  __shared__ GCutDelta dl[kMaxDl];
  __shared__ int sh[4];
  __shared__ uint32_t sh_off[2];
  if (gate && *gate != 0) return;   // queued behind a partition that turned out wrong (emat_tree_partition)
  const int p = part_list ? part_list[blockIdx.x] : (int)blockIdx.x, lane = threadIdx.x;
  const int base = pt.part_off[p], n = pt.part_off[p + 1] - base;
  if (lane == 0) {
    int status = k_gt_ok, cur_buf = 0, n_acc = 0, n_dl = 0;
    const int32_t tree_root = g.root[0];
    // (Asking for the next ancestor's headers and the first entries of this one's lists ahead of time -- one round trip per ancestor instead
    // of three -- made the kernel TWICE as slow, 0.97 against 0.48 ms: with 6 600 walks in flight it is bound by the number of requests, and
    // most ancestors have no missations and at most one mutation to fetch.)
    GClimb here{EMAT_NO_NODE, 0u, 0u, 0, 0};
    for (int32_t cur = pt.orig[base]; cur != EMAT_NO_NODE && status == k_gt_ok; cur = here.parent) {
      here = g.climb[cur];
      const GList mi{here.miss_off, here.miss_cnt}, mu{here.muts_off, here.muts_cnt};
      if (mi.cnt != 0) {   // union with merging of touching intervals (interval_set.h:238-288)
        const IvRec* B = g.iv_heap + mi.off; const IvRec* A = acc[cur_buf]; IvRec* O = acc[cur_buf ^ 1];
        uint32_t ia = 0, ib = 0; int no = 0; bool inside = false; int cs = 0, ce = 0;
        while (!(ia == (uint32_t)n_acc && ib == mi.cnt)) {
          const bool useA = (ia == (uint32_t)n_acc) ? false : (ib == mi.cnt) ? true : (A[ia].start <= B[ib].start);
          const IvRec f = useA ? A[ia] : B[ib];
          if (!inside) { cs = f.start; ce = f.end; if (useA) ++ia; else ++ib; inside = true; }
          else if (f.start <= ce) { ce = f.end > ce ? f.end : ce; if (useA) ++ia; else ++ib; }
          else { if (no >= kMaxIv) { status = k_gt_cut_state_overflow; break; } O[no++] = IvRec{cs, ce}; inside = false; }
        }
        if (inside) { if (no >= kMaxIv) status = k_gt_cut_state_overflow; else O[no++] = IvRec{cs, ce}; }
        cur_buf ^= 1; n_acc = no;
      }
      if (cur != tree_root) {   // what the root node carries are not events (they are folded into the reference at every gather)
        const MutRec* M = g.mut_heap + mu.off;
        for (int k = (int)mu.cnt - 1; k >= 0 && status == k_gt_ok; --k) {
          const MutRec m = M[k];
          int lo = 0, hi = n_dl;
          while (lo < hi) { int mid = (lo + hi) >> 1; if (dl[mid].site < m.site) lo = mid + 1; else hi = mid; }
          if (lo < n_dl && dl[lo].site == m.site) { if (dl[lo].from != m.to) status = k_gt_inconsistent; dl[lo].from = m.from; }
          else if (n_dl >= kMaxDl) status = k_gt_cut_state_overflow;
          else { for (int j = n_dl; j > lo; --j) dl[j] = dl[j - 1]; dl[lo] = GCutDelta{m.site, m.from, m.to, 0}; ++n_dl; }
        }
      } else if (mu.cnt != 0) status = k_gt_inconsistent;
    }
    // what the sub-root of the part carries (run.cpp:141-153): the missing sites, and the net changes over the present ones
    int n_keep = 0;
    const IvRec* A = acc[cur_buf];
    if (status == k_gt_ok) {
      for (int k = 0; k < n_dl; ++k) {
        const GCutDelta d = dl[k];
        if (d.from == d.to || gt_iv_contains(A, n_acc, d.site)) continue;
        if (d.from != ref[d.site]) { status = k_gt_inconsistent; break; }
        dl[n_keep++] = d;
      }
    }
    uint32_t om = 0, oi = 0;
    if (status == k_gt_ok) {
      const uint32_t region = (uint32_t)p % k_gt_pool_lanes, m_room = pools.mut_cap / k_gt_pool_lanes, i_room = pools.iv_cap / k_gt_pool_lanes;
      om = atomicAdd(&pools.tops[region * k_gt_pool_stride], (uint32_t)n_keep); oi = atomicAdd(&pools.tops[(k_gt_pool_lanes + region) * k_gt_pool_stride], (uint32_t)n_acc);
      if (om + (uint32_t)n_keep > m_room || oi + (uint32_t)n_acc > i_room) status = k_gt_pool_overflow;
      om += region * m_room; oi += region * i_room;
      if ((uint32_t)n_keep > k_gt_max_list || (uint32_t)n_acc > k_gt_max_list) status = k_gt_list_too_long;
    }
    sh[0] = status; sh[1] = n_keep; sh[2] = n_acc; sh[3] = cur_buf; sh_off[0] = om; sh_off[1] = oi;
    GMeasure m{}; m.n_nodes = n; m.root_muts_off = om; m.root_muts_cnt = (uint32_t)n_keep; m.root_miss_off = oi; m.root_miss_cnt = (uint32_t)n_acc; m.status = status;
    out[p] = m;
  }
  __syncthreads();
  const int status = sh[0], n_keep = sh[1], n_acc = sh[2];
  if (status != k_gt_ok) return;
  const IvRec* A = acc[sh[3]];
  const uint32_t pool_m = sh_off[0], pool_i = sh_off[1];
  for (int k = lane; k < n_keep; k += k_wave) { const GCutDelta d = dl[k]; MutRec r; r.t = -1.7976931348623157e308; r.site = d.site; r.from = d.from; r.to = d.to; r.pad = 0; pools.muts[pool_m + k] = r; }
  for (int k = lane; k < n_acc; k += k_wave) pools.ivs[pool_i + k] = A[k];
  // list content of the other nodes
  uint32_t content = 0, nm_tot = 0; int bad = 0;
  for (int s = 1 + lane; s < n; s += k_wave) {
    const int32_t o = pt.orig[base + s];
    const uint32_t nm = g.muts[o].cnt, ni = g.miss[o].cnt, nf = g.mfs[o].cnt;
    if (nm > k_gt_max_list || ni > k_gt_max_list || nf > k_gt_max_list) bad = 1;
    content += gt_a16(nm * 16u) + gt_a16(ni * 8u) + gt_a16(nf * 8u); nm_tot += nm;
  }
  content = wave_sum_u32(content); nm_tot = wave_sum_u32(nm_tot); bad = (int)wave_sum_u32((uint32_t)bad);
  // the part's time range (CoalBuilder::local_range): a tip of the PART counts with its float bounds -- a frozen boundary
  // node has t_min = t_max = (float)t (run.cpp:165-168)
  double lo = 1.7976931348623157e308, hi = -1.7976931348623157e308, hi_exact = -1.7976931348623157e308;
  for (int s = lane; s < n; s += k_wave) {
    const int32_t o = pt.orig[base + s];
    const double t = g.t[o];
    double a = t, b = t;
    if (pt.kid0[base + s] == EMAT_NO_NODE) { if (g.c0[o] != EMAT_NO_NODE) { a = (double)(float)t; b = a; } else { a = (double)g.t_min[o]; b = (double)g.t_max[o]; } }
    lo = a < lo ? a : lo; hi = b > hi ? b : hi; hi_exact = t > hi_exact ? t : hi_exact;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double l2 = __shfl_down(lo, off, k_wave), h2 = __shfl_down(hi, off, k_wave), e2 = __shfl_down(hi_exact, off, k_wave);
    lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; hi_exact = e2 > hi_exact ? e2 : hi_exact;
  }
  if (lane == 0) {
    out[p].t_min = lo; out[p].t_max = hi; out[p].t_max_exact = hi_exact;
    out[p].content_bytes = content + gt_a16((uint32_t)n_keep * 16u) + gt_a16((uint32_t)n_acc * 8u);
    out[p].num_muts = (int32_t)(nm_tot + (uint32_t)n_keep);
    if (bad) out[p].status = k_gt_list_too_long;
  }
}

// ---- pass 2: write every part's slab, byte for byte what encode_slab writes for the same part -----------------------
__global__ void __launch_bounds__(k_wave) k_gt_build(GTreeDev g, GPartition pt, GPools pools, const GMeasure* measure, const GPartDesc* desc, const uint8_t* cells, GCoal co,
                                                     uint8_t* slabs, const uint64_t* slab_off, int part_base) {
  // (a process that holds only the parts [part_base, part_base + gridDim.x) of the run builds only their slabs: its slab q is part part_base + q)
  const int q = blockIdx.x, p = part_base + q, lane = threadIdx.x;
  const int base = pt.part_off[p], n = pt.part_off[p + 1] - base;
  const GPartDesc d = desc[p]; const GMeasure me = measure[p];
  uint8_t* slab = slabs + slab_off[q];
  {   // everything but the scratch tail starts out zero
    uint4* z = (uint4*)slab; const uint32_t n16 = (d.slab_bytes - d.scratch_bytes) / 16u;
    for (uint32_t i = lane; i < n16; i += k_wave) z[i] = uint4{0u, 0u, 0u, 0u};
  }
  __syncthreads();
  SlabHeader* H = (SlabHeader*)slab;
  uint32_t off = (uint32_t)sizeof(SlabHeader);
  const uint32_t off_nodes = off; off += (uint32_t)n * (uint32_t)sizeof(NodeRec);
  const bool root_part = (d.flags & k_flag_includes_run_root) != 0;
  const uint32_t off_cells = off; off += gt_a16((uint32_t)d.cell_cap * (root_part ? k_cell_bytes_root : k_cell_bytes_own));
  const uint32_t off_trace = off; off += gt_a16((uint32_t)d.trace_cap * 32u);
  const uint32_t heap_begin = off;
  NodeRec* N = (NodeRec*)(slab + off_nodes);
  uint32_t carry = 0;
  for (int s0 = 0; s0 < n; s0 += k_wave) {
    const int s = s0 + lane;
    uint32_t nm = 0, ni = 0, nf = 0; int32_t o = EMAT_NO_NODE;
    GList lm{0, 0}, li{0, 0}, lf{0, 0};
    if (s < n) {
      o = pt.orig[base + s];
      if (s == 0) { nm = me.root_muts_cnt; ni = me.root_miss_cnt; }
      else { lm = g.muts[o]; li = g.miss[o]; lf = g.mfs[o]; nm = lm.cnt; ni = li.cnt; nf = lf.cnt; }
    }
    const uint32_t bm = gt_a16(nm * 16u), bi = gt_a16(ni * 8u), bf = gt_a16(nf * 8u), bytes = bm + bi + bf;
    const uint32_t incl = wave_incl_scan_u32(bytes, lane);
    const uint32_t top = heap_begin + carry + incl - bytes;
    carry += __shfl(incl, k_wave - 1, k_wave);
    if (s < n) {
      NodeRec& r = N[s];
      const int32_t k0 = pt.kid0[base + s], k1 = pt.kid1[base + s];
      r.child0 = k0; r.child1 = k1;
      if (s == 0) r.parent = EMAT_NO_NODE;
      if (k0 != EMAT_NO_NODE) { N[k0].parent = s; N[k1].parent = s; }
      const double t = g.t[o];
      r.t = t;
      if (k0 == EMAT_NO_NODE && g.c0[o] != EMAT_NO_NODE) { r.t_min = (float)t; r.t_max = (float)t; }   // frozen boundary node (run.cpp:165-168)
      else { r.t_min = g.t_min[o]; r.t_max = g.t_max[o]; }
      // (k_gt_measure refused every part with a list above k_gt_max_list: status k_gt_list_too_long, reported by the host)
      r.muts.off = top; r.muts.cnt = (uint16_t)nm; r.muts.cap = list_cap_for(bm, 16u);
      r.miss.off = top + bm; r.miss.cnt = (uint16_t)ni; r.miss.cap = list_cap_for(bi, 8u);
      r.mfs.off = top + bm + bi; r.mfs.cnt = (uint16_t)nf; r.mfs.cap = list_cap_for(bf, 8u);
      const MutRec* sm = s == 0 ? pools.muts + me.root_muts_off : g.mut_heap + lm.off;
      const IvRec* si = s == 0 ? pools.ivs + me.root_miss_off : g.iv_heap + li.off;
      const FsRec* sf = g.fs_heap + lf.off;
      MutRec* dm = (MutRec*)(slab + top); IvRec* di = (IvRec*)(slab + top + bm); FsRec* df = (FsRec*)(slab + top + bm + bi);
      for (uint32_t k = 0; k < nm; ++k) dm[k] = sm[k];
      for (uint32_t k = 0; k < ni; ++k) di[k] = si[k];
      for (uint32_t k = 0; k < nf; ++k) df[k] = sf[k];
    }
  }
  if (co.kbar_pool) {   // coalescent window out of the grid built on the device (k_gt_coal_*)
    double* cb = (double*)(slab + off_cells);
    const int nc = d.n_cells, cap = d.cell_cap;
    const double* kb = co.kbar_pool + d.cells_off; const double* kt = co.ktw_pool + d.cells_off;
    for (int w = lane; w < nc; w += k_wave) {
      const int c = d.cell_first + w;
      cb[w] = kb[w]; cb[cap + w] = kt[w];
      if (!root_part) continue;          // the run-wide arrays are read from the grid itself (SharedCells); the root part keeps its own copies
      const double pb = co.popsize[c];
      cb[2 * cap + w] = co.k_tw[c]; cb[3 * cap + w] = pb;
      cb[4 * cap + w] = d.t_step / pb;
      ((int32_t*)(cb + 5 * cap))[w] = co.num_active[c];
    }
  } else {   // coalescent window as the host packed it (encode_slab's cell table)
    const double* src = (const double*)(cells + d.cells_off);
    double* cb = (double*)(slab + off_cells);
    const int nc = d.n_cells, cap = d.cell_cap;
    const int32_t* na = (const int32_t*)(src + 4 * (size_t)nc);
    for (int w = lane; w < nc; w += k_wave) {
      cb[w] = src[w]; cb[cap + w] = src[(size_t)nc + w];
      if (!root_part) continue;
      const double pb = src[3 * (size_t)nc + w];
      cb[2 * cap + w] = src[2 * (size_t)nc + w]; cb[3 * cap + w] = pb;
      cb[4 * cap + w] = d.t_step / pb;
      ((int32_t*)(cb + 5 * cap))[w] = na[w];
    }
  }
  if (lane == 0) {
    H->magic = k_slab_magic; H->slab_bytes = d.slab_bytes; H->n_nodes = n; H->root = 0;
    H->flags = d.flags; H->status = 0; H->rng_key = d.rng_key; H->rng_counter = d.rng_counter; H->rng_spare = d.rng_spare; H->rng_has_spare = d.rng_has_spare;
    H->off_nodes = off_nodes; H->off_cells = off_cells; H->off_trace = off_trace;
    H->heap_begin = heap_begin; H->heap_top = heap_begin + carry; H->heap_end = heap_begin + d.heap_bytes;
    H->scratch_begin = H->heap_end; H->scratch_end = H->scratch_begin + d.scratch_bytes;
    H->cell_first = d.cell_first; H->n_cells = d.n_cells; H->cell_cap = d.cell_cap; H->n_cells_total = d.n_cells_total;
    H->t_ref = d.t_ref; H->t_step = d.t_step;
    H->trace_cap = d.trace_cap; H->trace_len = 0;
  }
}

// ---- the coalescent grid on the device (Run::reset_very_scalable_coalescent_parts, run.cpp:277-293;
//      very_scalable_coalescent.cpp:85-232; staged on the host in emat_host_model.hpp's CoalBuilder) ---------------------------
// k_gt_coal_kbar: every part's lineage counts over its window of cells.  One lane per cell, each walking the part's nodes
// in index order -- the order in which the reference adds the intervals, so that every cell's sum is the reference's sum.
__device__ inline int gt_cell_for(double t, double t_ref, double t_step) { return (int)floor((t_ref - t) / t_step); }
__device__ inline double gt_cell_ubound(int c, double t_ref, double t_step) { return t_ref - t_step * c; }
__device__ inline double gt_cell_lbound(int c, double t_ref, double t_step) { return gt_cell_ubound(c, t_ref, t_step) - t_step; }
// what add_interval(ts, te, +1) (very_scalable_coalescent.cpp:37-79) adds to cell i of a part whose last cell is `last`
__device__ inline double gt_interval_share(double ts, double te, int i, int last, double t_ref, double t_step, int first_stored, int32_t* status) {
  if (ts < te) { const double x = ts; ts = te; te = x; }
  const int cs = gt_cell_for(ts, t_ref, t_step);
  int ce = last;
  if (te != gt_cell_lbound(ce, t_ref, t_step)) ce = gt_cell_for(te, t_ref, t_step);
  if (cs < first_stored || ce > last || cs > ce) { atomicMax(status, (int32_t)k_gt_inconsistent); return 0.0; }
  if (i < cs || i > ce) return 0.0;
  if (cs == ce) return (ts - te) / t_step;
  if (i == cs) return (ts - gt_cell_lbound(cs, t_ref, t_step)) / t_step;
  if (i == ce) return (gt_cell_ubound(ce, t_ref, t_step) - te) / t_step;
  return 1.0;
}
__global__ void __launch_bounds__(k_wave) k_gt_coal_kbar(GTreeDev g, GPartition pt, const GPartDesc* desc, GCoal co) {
  const int p = blockIdx.x, lane = threadIdx.x;
  const int base = pt.part_off[p], n = pt.part_off[p + 1] - base;
  const GPartDesc d = desc[p];
  const int first = d.cell_first, last = d.n_cells_total - 1;
  double* kb = co.kbar_pool + d.cells_off;
  for (int i = first + lane; i <= last; i += k_wave) {
    double k = 0.0;
    for (int s = 1; s < n; ++s) {
      const double share = gt_interval_share(g.t[pt.orig[base + pt.lpar[base + s]]], g.t[pt.orig[base + s]], i, last, co.t_ref, co.t_step, first, co.status);
      if (share != 0.0) k += share;   // (adding 0.0 would not change k either: kept out for speed)
    }
    if (p == pt.root_part) k += gt_interval_share(gt_cell_lbound(co.num_cells - 1, co.t_ref, co.t_step), g.t[pt.orig[base]], i, last, co.t_ref, co.t_step, first, co.status);
    kb[i - first] = k;
  }
}
// k_gt_coal_grid: one wavefront per cell of the whole grid -- sum of the parts' counts, number of active parts, mean
// population size.  Lane l adds up the parts l, l + 64, ... in ascending order and the 64 partial sums are folded in a
// fixed tree: the same bits on every run (the reference's plain part-order sum differs from it by rounding only).
__device__ inline double gt_wave_fold(double x) {
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, k_wave);
  return x;   // valid in lane 0
}
__global__ void __launch_bounds__(k_wave) EMAT_OCCUPANCY k_gt_coal_grid(int num_parts, const GPartDesc* desc, GCoal co, const PopTable* pop) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double k = 0.0; uint32_t active = 0;
  for (int p = lane; p < num_parts; p += k_wave) {
    const int first = desc[p].cell_first, fa = desc[p].coal_first_active, last = desc[p].n_cells_total - 1;
    if (c >= first && c <= last) k += co.kbar_pool[desc[p].cells_off + (uint64_t)(c - first)];
    if (c >= fa && c <= last) ++active;
  }
  k = gt_wave_fold(k); active = wave_sum_u32(active);
  if (lane != 0) return;
  co.k_bar[c] = k; co.num_active[c] = (int32_t)active;
  co.popsize[c] = dev::pop_integral(*pop, gt_cell_lbound(c, co.t_ref, co.t_step), gt_cell_ubound(c, co.t_ref, co.t_step)) / co.t_step;
  co.ts_over_pop[c] = co.t_step / co.popsize[c];
  if (c == co.num_cells - 1 && active == 0) atomicMax(co.status, (int32_t)k_gt_inconsistent);
}
// k_gt_coal_draw: every part draws k_twiddle_bar_p over its active cells from its own stream.  A Gaussian takes one whole
// Philox block (emat_device_core.hpp `gaussian`), so draw j of a fresh stream is block j: one lane per cell.
__global__ void __launch_bounds__(k_wave) k_gt_coal_draw(const GPartDesc* desc, GCoal co) {
  const int p = blockIdx.x, lane = threadIdx.x;
  const GPartDesc d = desc[p];
  const int first = d.cell_first, fa = d.coal_first_active, last = d.n_cells_total - 1;
  const double* kb = co.kbar_pool + d.cells_off; double* kt = co.ktw_pool + d.cells_off;
  for (int i = first + lane; i <= last; i += k_wave) {
    double v = 0.0;
    if (i >= fa) {
      const int na = co.num_active[i];
      const double mu = kb[i - first] - co.k_bar[i] / na;
      const double sigma = sqrt(co.popsize[i] / (na * co.t_step));
      uint32_t w[4];
      dev::philox4x32_10((uint64_t)(i - fa), d.rng_key, w);
      const uint64_t a = (uint64_t)w[0] | ((uint64_t)w[1] << 32), b = (uint64_t)w[2] | ((uint64_t)w[3] << 32);
      const double u1 = dev::to_oc(a), u2 = dev::to_co(b);
      v = mu + sigma * (sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925 * u2));
    }
    kt[i - first] = v;
  }
}
__global__ void __launch_bounds__(k_wave) k_gt_coal_ktw(int num_parts, const GPartDesc* desc, GCoal co) {
  const int c = blockIdx.x, lane = threadIdx.x;
  double k = 0.0;
  for (int p = lane; p < num_parts; p += k_wave) {
    const int first = desc[p].cell_first, fa = desc[p].coal_first_active, last = desc[p].n_cells_total - 1;
    if (c >= fa && c <= last) k += co.ktw_pool[desc[p].cells_off + (uint64_t)(c - first)];
  }
  k = gt_wave_fold(k);
  if (lane == 0) co.k_tw[c] = k;
}

// ---- gather: every part writes the nodes it owns back into the whole tree (run.cpp:195-256) ---------------------------
// Every node of the whole tree is a non-root node of exactly one part (the run's root: the root of the root part); that
// part writes its time, lists and child links, and the parent links of its children.  The three heaps are rebuilt from
// zero, each part reserving its share with one atomic per heap.  What the run's root carries afterwards -- the changes
// reference sequence -> root sequence -- is folded into the reference on the way (Run::normalize_root +
// rereference_to_root_sequence, run.cpp:258-265, phylo_tree.cpp:299-322): the root ends up without mutations, and the
// from-states of every missation that covers a changed site are rewritten (mutations.h:212-232).
struct GRootDelta { int32_t site; uint8_t from, to; uint16_t pad; };

// from-states of one node after the reference changed at the sites of R (sorted by site); writes them to `out` when it
// is not null; returns their number
__device__ inline uint32_t gt_rereferenced_from_states(const FsRec* fs, uint32_t nf, const IvRec* miss, uint32_t ni, const GRootDelta* R, int nR, FsRec* out) {
  if (ni == 0 || nR == 0) { if (out) for (uint32_t k = 0; k < nf; ++k) out[k] = fs[k]; return nf; }
  uint32_t k = 0, w = 0;
  for (int j = 0; j < nR; ++j) {
    const GRootDelta d = R[j];
    if (!gt_iv_contains(miss, (int)ni, d.site)) continue;
    while (k < nf && fs[k].site < d.site) { if (out) out[w] = fs[k]; ++w; ++k; }
    if (k < nf && fs[k].site == d.site) { if (fs[k].state != d.to) { if (out) out[w] = fs[k]; ++w; } ++k; }   // it now equals the reference: dropped
    else if (d.from != d.to) { if (out) { FsRec f{}; f.site = d.site; f.state = d.from; out[w] = f; } ++w; }      // it was the old reference state
  }
  while (k < nf) { if (out) out[w] = fs[k]; ++w; ++k; }
  return w;
}

__global__ void __launch_bounds__(k_wave) k_gt_gather(GTreeDev g, GPartition pt, const uint8_t* slabs, const uint64_t* slab_off, uint8_t* ref,
                                                      GRootDelta* root_deltas_out, int32_t* n_root_deltas_out, int32_t* status_out,
                                                      int part_base, const GRootDelta* R_in, int nR_in) {
  // `part_base`: slab q of this process is part part_base + q of the run.  `R_in` (or null): the changes of the root sequence,
  // when the part that holds the run's root lives in another process (which published them); else they are read off its slab.
  __shared__ GRootDelta R[k_gt_max_root_deltas];
  __shared__ uint32_t sh_base[3];
  __shared__ int sh_nR;
  const int q = blockIdx.x, p = part_base + q, lane = threadIdx.x;
  const int base = pt.part_off[p], n = pt.part_off[p + 1] - base;
  const uint8_t* slab = slabs + slab_off[q];
  const SlabHeader* H = (const SlabHeader*)slab;
  const NodeRec* N = (const NodeRec*)(slab + H->off_nodes);
  const bool is_root_part = p == pt.root_part;
  const GRootDelta* Rp = R;   // the changes of the root sequence: in LDS when they fit (nearly always a handful), else read where they lie
  if (R_in != nullptr) {
    if (nR_in > k_gt_max_root_deltas) Rp = R_in;
    else for (int k = lane; k < nR_in; k += k_wave) R[k] = R_in[k];
    if (lane == 0) sh_nR = nR_in;
  } else {
    const uint8_t* rs = slabs + slab_off[pt.root_part - part_base];
    const SlabHeader* Hr = (const SlabHeader*)rs;
    const NodeRec& rn = ((const NodeRec*)(rs + Hr->off_nodes))[Hr->root];
    const int nR = rn.muts.cnt;
    if (nR > k_gt_max_root_deltas) { if (lane == 0) atomicMax(status_out, (int32_t)k_gt_root_deltas_overflow); return; }
    const MutRec* M = (const MutRec*)(rs + rn.muts.off);
    for (int k = lane; k < nR; k += k_wave) { GRootDelta d{}; d.site = M[k].site; d.from = M[k].from; d.to = M[k].to; R[k] = d; }
    if (lane == 0) sh_nR = nR;
  }
  __syncthreads();
  const int nR = sh_nR;
  if (H->n_nodes != n || (!is_root_part && H->root != 0)) { if (lane == 0) atomicMax(status_out, (int32_t)k_gt_inconsistent); return; }
  const int local_root = H->root;
  // totals of the part -> its share of the three heaps
  uint32_t tm = 0, ti = 0, tf = 0;
  for (int s = lane; s < n; s += k_wave) {
    const bool is_lr = s == local_root;
    if (is_lr && !is_root_part) continue;
    const NodeRec& r = N[s];
    tm += is_lr ? 0u : r.muts.cnt; ti += r.miss.cnt;
    tf += gt_rereferenced_from_states((const FsRec*)(slab + r.mfs.off), r.mfs.cnt, (const IvRec*)(slab + r.miss.off), r.miss.cnt, Rp, nR, nullptr);
  }
  tm = wave_sum_u32(tm); ti = wave_sum_u32(ti); tf = wave_sum_u32(tf);
  if (lane == 0) {
    sh_base[0] = atomicAdd(&g.tops[0], tm); sh_base[1] = atomicAdd(&g.tops[1], ti); sh_base[2] = atomicAdd(&g.tops[2], tf);
    if (sh_base[0] + tm > g.mut_cap || sh_base[1] + ti > g.iv_cap || sh_base[2] + tf > g.fs_cap) { atomicMax(status_out, (int32_t)k_gt_heap_overflow); sh_base[0] = 0xffffffffu; }
  }
  __syncthreads();
  if (sh_base[0] == 0xffffffffu) return;
  uint32_t cm = sh_base[0], ci = sh_base[1], cf = sh_base[2];
  for (int s0 = 0; s0 < n; s0 += k_wave) {
    const int s = s0 + lane;
    uint32_t nm = 0, ni = 0, nf = 0; bool owns = false;
    if (s < n) {
      const bool is_lr = s == local_root;
      owns = !is_lr || is_root_part;
      if (owns) {
        const NodeRec& r = N[s];
        nm = is_lr ? 0u : r.muts.cnt; ni = r.miss.cnt;
        nf = gt_rereferenced_from_states((const FsRec*)(slab + r.mfs.off), r.mfs.cnt, (const IvRec*)(slab + r.miss.off), r.miss.cnt, Rp, nR, nullptr);
      }
    }
    const uint32_t im = wave_incl_scan_u32(nm, lane), ii = wave_incl_scan_u32(ni, lane), iff = wave_incl_scan_u32(nf, lane);
    const uint32_t om = cm + im - nm, oi = ci + ii - ni, of = cf + iff - nf;
    cm += __shfl(im, k_wave - 1, k_wave); ci += __shfl(ii, k_wave - 1, k_wave); cf += __shfl(iff, k_wave - 1, k_wave);
    if (s < n) {
      const NodeRec& r = N[s];
      const int32_t o = pt.orig[base + s];
      if (owns) {
        g.t[o] = r.t;
        g.muts[o] = GList{om, nm}; g.miss[o] = GList{oi, ni}; g.mfs[o] = GList{of, nf};
        const MutRec* sm = (const MutRec*)(slab + r.muts.off); const IvRec* si = (const IvRec*)(slab + r.miss.off);
        for (uint32_t k = 0; k < nm; ++k) g.mut_heap[om + k] = sm[k];
        for (uint32_t k = 0; k < ni; ++k) g.iv_heap[oi + k] = si[k];
        gt_rereferenced_from_states((const FsRec*)(slab + r.mfs.off), r.mfs.cnt, si, r.miss.cnt, Rp, nR, g.fs_heap + of);
      }
      if (r.child0 != EMAT_NO_NODE) {
        const int32_t l = pt.orig[base + r.child0], rr = pt.orig[base + r.child1];
        g.c0[o] = l; g.c1[o] = rr; g.parent[l] = o; g.parent[rr] = o;
      }
    }
  }
  if (is_root_part && lane == 0) {
    const int32_t nr = pt.orig[base + local_root];
    g.root[0] = nr; g.parent[nr] = EMAT_NO_NODE;
    for (int k = 0; k < nR; ++k) { ref[Rp[k].site] = Rp[k].to; root_deltas_out[k] = Rp[k]; }
    n_root_deltas_out[0] = nR;
  }
}

// The links alone -- children, parents, the root, and the packed children the host's partitioner reads -- ahead of k_gt_gather,
// which rewrites every list of the tree and takes ten times as long: the host draws the next partition from these while the
// lists are still on their way.  (k_gt_gather writes the same links again.)
__global__ void __launch_bounds__(k_wave) k_gt_gather_links(GTreeDev g, GPartition pt, const uint8_t* slabs, const uint64_t* slab_off, int part_base, int2* kids, double* root_t_out) {
  const int q = blockIdx.x, p = part_base + q, lane = threadIdx.x;
  const int base = pt.part_off[p], n = pt.part_off[p + 1] - base;
  const uint8_t* slab = slabs + slab_off[q];
  const SlabHeader* H = (const SlabHeader*)slab;
  if (H->n_nodes != n) return;   // (k_gt_gather reports it)
  const NodeRec* N = (const NodeRec*)(slab + H->off_nodes);
  for (int s = lane; s < n; s += k_wave) {
    const NodeRec& r = N[s];
    if (r.child0 == EMAT_NO_NODE) continue;
    const int32_t o = pt.orig[base + s], l = pt.orig[base + r.child0], rr = pt.orig[base + r.child1];
    g.c0[o] = l; g.c1[o] = rr; g.parent[l] = o; g.parent[rr] = o;
    kids[o] = make_int2(l, rr);
  }
  if (p == pt.root_part && lane == 0) {
    const int32_t nr = pt.orig[base + H->root];
    g.root[0] = nr; g.parent[nr] = EMAT_NO_NODE;
    root_t_out[0] = N[H->root].t;
  }
}

// ---- one run over several processes, every one with the whole tree in its HBM: after gathering its own parts a process
//      hands the nodes it owns to the others (compact per-node arrays + its three heap segments) and takes theirs -------------
// Writes as many of the changes as `out` holds (`cap`) and reports how many there are: the caller comes again with more room.
__global__ void __launch_bounds__(k_wave) k_gt_root_deltas(const uint8_t* slabs, const uint64_t* slab_off, int root_slab, GRootDelta* out, int cap, int32_t* n_out) {
  const uint8_t* rs = slabs + slab_off[root_slab];
  const SlabHeader* Hr = (const SlabHeader*)rs;
  const NodeRec& rn = ((const NodeRec*)(rs + Hr->off_nodes))[Hr->root];
  const int nR = rn.muts.cnt;
  const MutRec* M = (const MutRec*)(rs + rn.muts.off);
  for (int k = threadIdx.x; k < nR && k < cap; k += k_wave) { GRootDelta d{}; d.site = M[k].site; d.from = M[k].from; d.to = M[k].to; out[k] = d; }
  if (threadIdx.x == 0) n_out[0] = nR;
}
// One entry per node of every part the process ran: what the part OWNS of the node (time and lists: every node but the part's
// root, unless it is the run's root) and what it LINKS (children: every node that is inner within the part, its root
// included).  A cut node therefore appears twice, possibly in two processes: owned by the part above, linked by the one below.
constexpr uint32_t k_gt_export_owns = 1u << 30, k_gt_export_links = 1u << 31, k_gt_export_node_mask = (1u << 30) - 1u;
struct GNodeExport { uint32_t node_and_flags; int32_t c0, c1, pad; double t; GList muts, miss, mfs; };   // 48 B
__global__ void k_gt_export(GTreeDev g, const uint32_t* ids, int n, GNodeExport* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t id = ids[i]; const int32_t o = (int32_t)(id & k_gt_export_node_mask);
  GNodeExport e{}; e.node_and_flags = id; e.c0 = g.c0[o]; e.c1 = g.c1[o]; e.t = g.t[o]; e.muts = g.muts[o]; e.miss = g.miss[o]; e.mfs = g.mfs[o];
  out[i] = e;
}
// the lists of the nodes in `in` were appended to this process's heaps at bases `shift`
__global__ void k_gt_apply(GTreeDev g, const GNodeExport* in, int n, uint32_t shift_m, uint32_t shift_i, uint32_t shift_f, int32_t new_root) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && new_root != EMAT_NO_NODE) { g.root[0] = new_root; g.parent[new_root] = EMAT_NO_NODE; }
  if (i >= n) return;
  const GNodeExport e = in[i];
  const int32_t o = (int32_t)(e.node_and_flags & k_gt_export_node_mask);
  if (e.node_and_flags & k_gt_export_owns) {
    g.t[o] = e.t;
    g.muts[o] = GList{e.muts.off + shift_m, e.muts.cnt}; g.miss[o] = GList{e.miss.off + shift_i, e.miss.cnt}; g.mfs[o] = GList{e.mfs.off + shift_f, e.mfs.cnt};
  }
  if (e.node_and_flags & k_gt_export_links) { g.c0[o] = e.c0; g.c1[o] = e.c1; g.parent[e.c0] = o; g.parent[e.c1] = o; }
}

}  // namespace emat
#endif  // EMAT_GTREE_KERNELS_HPP_
