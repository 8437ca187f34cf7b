// emat_build.hpp -- SURVEY.md 8(f).4: initial-tree construction, the reference's UShER-like builder
// (build_usher_like_tree, core/phylo_tree.cpp:796-1049) behind emat_tree_build_usher_like.
//
// The builder grafts the tips one by one where they need the fewest new mutations.  Each placement is one UNLIMITED candidate
// scan from the root (Spr_study_builder::seed_fill_from, spr_study.cpp:9-128) -- the whole tree built so far -- which makes the
// loop O(tips x nodes): that loop runs on the device.  What the scan computes for a region (branch b, segment between its
// mutations) is the number of sites at which the sequence there differs from the new tip's, over the sites the tip has; that
// number changes by -1 / 0 / +1 per mutation crossed, so for a whole tree it is a per-branch sum (all nodes side by side)
// followed by a prefix sum down the tree, instead of the reference's depth-first walk with a hash map.  The walk's ORDER matters as
// well (among the regions that tie for the minimum the reference picks by cumulative length in visiting order, a floating-point
// sum): the visiting order is the pre-order with the second child first and a branch's segments in time order.  Every node's place in
// that order and the size of its subtree are KEPT from tip to tip (a graft inserts two places and enlarges its ancestors, which every
// node applies to itself from a five-word record of the graft); a subtree is then a stretch of places, so the per-branch changes go
// into a difference array over the places (+d where the subtree starts, -d where it ends) and ONE prefix scan gives every node its
// distance.  (Until round 4: two prefix sums down the tree by pointer jumping, log2(depth) rounds with a meeting each, per tip.)
// The tying regions are laid out in visiting order by a second scan, and the first workgroup -- one deciding thread, the others
// staging its inputs in LDS and doing what is parallel: the path to the root found by interval containment, the composition of
// the deltas site by site -- adds their lengths in order, draws, and makes the graft (three nodes change).
// One cooperative launch runs the whole loop without returning to the host: up to one workgroup of 1 024 threads per four CUs, all
// resident, meeting seven times per tip at a barrier of their own (BGrid::sync: one release and one acquire per workgroup).
//
// The O(nodes) passes after the loop -- fix_up_missations (phylo_tree.cpp:414-507), pseudo_date (dates.cpp:63-82),
// randomize_mutation_times (phylo_tree.cpp:567-644) -- are host C++ below: they run once, like the partitioning the reference
// also keeps on the host.  Random numbers: one Philox stream (the engine's), consumed in the reference's order of draws; its
// hash-map iterations run in ascending site order.  Product code: shares nothing with oracle/orc_build.hpp, which restates the
// same reference functions as the checker.
#ifndef EMAT_BUILD_HPP_
#define EMAT_BUILD_HPP_

namespace emat {

struct BDelta { int32_t site; uint8_t from, to; uint16_t pad; };   // one entry of a Site_deltas map, kept sorted by site

struct BuildDev {
  int32_t n_tips, L;
  const uint8_t* ref;
  const int32_t* d_off; const int32_t* d_site; const uint8_t* d_to;      // tips' deltas against the reference sequence, CSR, ascending site
  const int32_t* m_off; const int32_t* m_start; const int32_t* m_end;   // tips' missing intervals, CSR
  // the tree under construction (2 n - 1 nodes: tips 0 .. n-1, the inner node made for tip X is X + n - 1)
  int32_t* root; int32_t* parent; int32_t* c0; int32_t* c1; double* t; int32_t* sz;
  uint32_t* ml_off; int32_t* ml_cnt; MutRec* pool; uint32_t pool_cap; uint32_t* pool_top;
  // per-tip work arrays [2 n - 1]
  int32_t* delta; int32_t* vD[2]; int32_t* vP[2]; int32_t* anc[2]; int32_t* inv; int32_t* cnt; int32_t* off;
  // the regions that tie for the fewest mutations, in visiting order
  int32_t* tie_node; double* tie_tmin; double* tie_tmax; uint32_t tie_cap;
  // the grafting thread's own buffers
  int32_t* path; BDelta* sd; uint32_t sd_cap;
  uint64_t* rng;      // [4] key, counter, spare, has_spare
  int32_t* status;    // [2] 0 = ok | 1 mutation pool full | 2 delta buffer full | 3 tie list full | 4 inconsistent input | 5 a workgroup never arrived; [1] = tip at which it happened
  // what the workgroups of the launch share besides the tree: the meeting counter, the running minimum, one word per pointer-jumping
  // round, every workgroup's share of the tying regions, and the grafting thread's verdict on the tip
  unsigned long long* grid_counter; int32_t* gmin; int32_t* blk_sum; int32_t* blk_sum2; int32_t* stop_flag;
  int32_t* gdesc;     // [8] the previous graft, for phase A: the node it went above (EMAT_NO_NODE: none yet), that node's position and subtree size before, "swapped with its sibling", the sibling's size
  long long* prof;    // [8] 100 MHz ticks of the grafting thread: whole loop, waiting for the parallel phases, tie sums, path + deltas, links + sizes, mutations; pointer-jumping rounds; tying regions
};

struct BRng {   // the engine's stream (emat_device_core.hpp rng_next64 and friends), for the one thread that draws
  uint64_t key, ctr, spare; bool has_spare;
  __device__ uint64_t next64() {
    if (has_spare) { has_spare = false; return spare; }
    uint32_t w[4]; dev::philox4x32_10(ctr++, key, w);
    spare = (uint64_t)w[2] | ((uint64_t)w[3] << 32); has_spare = true;
    return (uint64_t)w[0] | ((uint64_t)w[1] << 32);
  }
  __device__ double uniform_co(double lo, double hi) { return lo + (hi - lo) * ((double)(next64() >> 11) * 0x1.0p-53); }
  __device__ double uniform_oc(double lo, double hi) { return lo + (hi - lo) * (((double)(next64() >> 11) + 1.0) * 0x1.0p-53); }
  __device__ double uniform_oo(double lo, double hi) { return lo + (hi - lo) * (((double)(next64() >> 12) + 0.5) * 0x1.0p-52); }
};

// The new tip X as the scan sees it: its deltas against the reference sequence and its missing intervals -- staged in LDS by every
// workgroup at the start of the tip when they fit (every mutation of the tree is looked up in them four times per tip), else in HBM.
struct BTip { const int32_t* dsite; const uint8_t* dto; int dn; const int32_t* mstart; const int32_t* mend; int mn; const uint8_t* ref; };
// state of the new tip at a site: its delta there, else the reference sequence
__device__ inline int b_tip_state(const BTip& x, int site) {
  int lo = 0, hi = x.dn;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (x.dsite[mid] < site) lo = mid + 1; else hi = mid; }
  return (lo < x.dn && x.dsite[lo] == site) ? (int)x.dto[lo] : (int)x.ref[site];
}
__device__ inline bool b_tip_missing(const BTip& x, int site) {
  int lo = 0, hi = x.mn;   // first interval with start > site
  while (lo < hi) { int mid = (lo + hi) >> 1; if (site < x.mstart[mid]) hi = mid; else lo = mid + 1; }
  return lo > 0 && site < x.mend[lo - 1];
}
// change of the distance to the new tip across one mutation, walking down (spr_study.cpp:43-91: only sites the tip has count)
__device__ inline int b_step(const BTip& x, const MutRec& m) {
  if (b_tip_missing(x, m.site)) return 0;
  const int st = b_tip_state(x, m.site);
  return ((int)m.to != st ? 1 : 0) - ((int)m.from != st ? 1 : 0);
}
// site_deltas.h:43-65 on a sorted array: put `from -> to` IN FRONT of the delta list
__device__ inline bool b_push_front(BDelta* sd, int& n, uint32_t cap, int site, int from, int to, bool& inconsistent) {
  int lo = 0, hi = n;
  while (lo < hi) { int mid = (lo + hi) >> 1; if (sd[mid].site < site) lo = mid + 1; else hi = mid; }
  if (lo < n && sd[lo].site == site) {
    if (to != (int)sd[lo].from) inconsistent = true;
    sd[lo].from = (uint8_t)from;
    if (sd[lo].from == sd[lo].to) { for (int i = lo; i + 1 < n; ++i) sd[i] = sd[i + 1]; --n; }
    return true;
  }
  if ((uint32_t)n >= cap) return false;
  for (int i = n; i > lo; --i) sd[i] = sd[i - 1];
  sd[lo].site = site; sd[lo].from = (uint8_t)from; sd[lo].to = (uint8_t)to; sd[lo].pad = 0; ++n;
  return true;
}

constexpr int k_build_threads = 1024;
// LDS staging: the new tip's descriptor (every workgroup), and for the graft step site deltas, path mutations / X's new mutations, path
// nodes.  Inputs that outgrow an area take the same steps on buffers in HBM.  -DEMAT_BUILD_TINY_STAGING shrinks the areas so that
// ordinary test inputs take every one of those fall-backs (scripts/build_variant.sh tiny_staging -DEMAT_BUILD_TINY_STAGING; the
// builder's parity tests must pass on it unchanged).
#ifdef EMAT_BUILD_TINY_STAGING
constexpr int k_x_deltas = 4, k_x_miss = 1;
constexpr int k_g_sd = 24, k_g_pm = 40, k_g_path = 8;
#else
constexpr int k_x_deltas = 2048, k_x_miss = 256;
constexpr int k_g_sd = 1024, k_g_pm = 1536, k_g_path = 1024;
#endif


// Inclusive sum over the 1 024 threads of a workgroup: a shuffle scan inside every wavefront, the sixteen wave totals scanned by
// the first wave, two meetings of the workgroup in all (the ten-step ladder through LDS this replaces had one after every step).
__device__ inline int b_block_scan_incl(int mine, int* s_wave /* [16] */, int tid) {
  const int lane = tid & 63, w = tid >> 6;
  int x = mine;
  for (int ofs = 1; ofs < 64; ofs <<= 1) { const int y = __shfl_up(x, ofs, 64); if (lane >= ofs) x += y; }
  if (lane == 63) s_wave[w] = x;
  __syncthreads();
  if (w == 0) { int t2 = lane < 16 ? s_wave[lane] : 0; for (int ofs = 1; ofs < 16; ofs <<= 1) { const int y = __shfl_up(t2, ofs, 64); if (lane >= ofs) t2 += y; } if (lane < 16) s_wave[lane] = t2; }
  __syncthreads();
  return x + (w > 0 ? s_wave[w - 1] : 0);
}

// Bitonic sort of up to 64 (key, payload) pairs, one per lane of ONE wavefront, in registers (compare-exchange through lane
// shuffles: no LDS round trips, no meeting of the workgroup); `descending` puts the largest key in lane 0.  Keys must be distinct
// (or equal only among padding, whose payloads do not matter).
__device__ inline void b_wave_sort(uint32_t& key, int& val, int n_pow2, bool descending) {
  const int lane = (int)(threadIdx.x & 63);
  for (int k2 = 2; k2 <= n_pow2; k2 <<= 1)
    for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
      const uint32_t ok = (uint32_t)__shfl_xor((int)key, j2, 64); const int ov = __shfl_xor(val, j2, 64);
      const bool lower = (lane & j2) == 0, up = ((lane & k2) == 0) != descending;      // `up`: this block sorts ascending
      const bool take_min = lower == up;
      const bool swap = take_min ? ok < key : ok > key;
      if (swap) { key = ok; val = ov; }
    }
}

// All workgroups of the launch meet here (they are all resident: a cooperative launch, hipLaunchCooperativeKernel, of at most one
// workgroup per CU -- the runtime refuses a grid it cannot hold at once): a counter that only grows --
// after the k-th meeting it stands at k x workgroups -- so nothing is ever reset.  The wait is bounded: a workgroup that never
// arrives (it cannot, short of a fault) makes the others give up and say so instead of hanging the device.
struct BGrid {
  unsigned long long* counter; int32_t* status; unsigned long long blocks, meetings;
  // One meeting of all workgroups.  Memory: what a workgroup stored before the meeting must be visible to every other afterwards, across
  // the eight XCDs' L2s.  ONE agent-scope release per workgroup (thread 0, after the workgroup's own barrier has ordered the other
  // threads' stores before it: the L2 write-back it issues is the XCD's, not the thread's) and ONE agent-scope acquire per workgroup
  // afterwards (the L1 it invalidates is the CU's, shared by all sixteen waves) -- instead of a full fence by every thread, which
  // made sixteen waves of every workgroup write back and invalidate in turn: scripts/micro/gridbar_xcc.hip, 64 workgroups over all
  // XCDs exchanging 16 KB each per round of two meetings: 6.5 us against 13 us (all waves acquire) and 36+ us (all threads fence);
  // no stale word in 2 000 rounds.  (Confining the launch to one XCD and reading through L2 would be 2.2 us, for an eighth of the CUs.)
  __device__ bool sync() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores have left the CU
    __syncthreads();
    ++meetings;
    if (blocks > 1) {
      __shared__ int s_ok;
      if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        atomicAdd(counter, 1ull);
        const unsigned long long target = meetings * blocks;
        // (bounded by wall-clock time -- a minute of the 100 MHz counter -- not by a number of polls: the grafting thread's serial
        // stretch on a tip with thousands of deltas is legitimately long)
        const uint64_t w0 = wall_clock64(); bool gave_up = false;
        // (polled with loads, not read-modify-writes: sixty-odd workgroups hammering one word with atomics queue up in front of the
        // arrival of the workgroup everybody is waiting for)
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) { __builtin_amdgcn_s_sleep(2); if (wall_clock64() - w0 > 6000000000ull) { gave_up = true; break; } }
        s_ok = gave_up ? 0 : 1;
        if (!s_ok) status[0] = 5;
      }
      __syncthreads();
      if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // what the other workgroups wrote before they arrived is visible from here on (this CU's L1 dropped)
      __syncthreads();
      return s_ok != 0;
    }
    return true;
  }
};

// Tips [first_tip, last_tip) are grafted onto the tree that holds tips 0 .. first_tip - 1 (the host sets up the first two).
__global__ void __launch_bounds__(k_build_threads) k_build_usher_graft(BuildDev b, int first_tip, int last_tip) {
  const int tid = threadIdx.x, NT = k_build_threads, n = b.n_tips;
  const int nb = (int)gridDim.x, blk = (int)blockIdx.x, gtid = blk * NT + tid, GT = nb * NT;
  __shared__ int s_carry, s_base, s_stop;
  __shared__ int s_wave[16];
  __shared__ int32_t s_xs[k_x_deltas]; __shared__ uint8_t s_xto[k_x_deltas]; __shared__ int32_t s_xms[k_x_miss]; __shared__ int32_t s_xme[k_x_miss];   // the new tip (BTip)
  __shared__ int s_cb[256];                                          // running totals of the workgroups' stretches (at most one workgroup per CU)
  // what the grafting thread works on, staged by its workgroup (step 5)
  __shared__ BDelta s_sd[k_g_sd]; __shared__ MutRec s_pm[k_g_pm]; __shared__ int s_path[k_g_path]; __shared__ int s_pcnt[k_g_path];
  __shared__ double s_len[k_build_threads]; __shared__ int s_gi[8]; __shared__ double s_gd[2];
  BGrid grid{b.grid_counter, b.status, (unsigned long long)nb, 0ull};
  BRng rng; rng.key = b.rng[0]; rng.ctr = b.rng[1]; rng.spare = b.rng[2]; rng.has_spare = b.rng[3] != 0;   // (the grafting thread's copy is the one that counts)
  const bool grafter = blk == 0 && tid == 0;
  for (int X = first_tip; X < last_tip; ++X) {
    const int nl = 2 * X - 1;                                      // nodes linked so far: tips 0 .. X-1 and inner nodes n .. n + X - 2
    auto node_of = [&](int i) { return i < X ? i : n + (i - X); };
    const int dx0 = b.d_off[X], dxn = b.d_off[X + 1] - dx0, mx0 = b.m_off[X], mxn = b.m_off[X + 1] - mx0;
    const double t_X = b.t[X];
    const int root = *b.root;
    BTip xt{b.d_site + dx0, b.d_to + dx0, dxn, b.m_start + mx0, b.m_end + mx0, mxn, b.ref};
    if (dxn <= k_x_deltas && mxn <= k_x_miss) {
      for (int k = tid; k < dxn; k += NT) { s_xs[k] = xt.dsite[k]; s_xto[k] = xt.dto[k]; }
      for (int k = tid; k < mxn; k += NT) { s_xms[k] = xt.mstart[k]; s_xme[k] = xt.mend[k]; }
      __syncthreads();
      xt.dsite = s_xs; xt.dto = s_xto; xt.mstart = s_xms; xt.mend = s_xme;
    }
    const long long t_tip0 = grafter ? (long long)wall_clock64() : 0ll;
    if (grafter) *b.gmin = 0x7fffffff;
    // (A) Positions in the visiting order and subtree sizes are kept from tip to tip: the previous graft put two nodes into the
    // order (its new inner node P in front of the subtree it was hung above, its tip behind that subtree), made every ancestor of P
    // two nodes larger and -- when the subtree was its parent's second child, which the new inner node is not -- swapped it with its
    // sibling's subtree (c0 = P, c1 = the sibling: the second child is visited first).  Every node brings ITSELF up to date from
    // the grafting thread's record of that graft, side by side.  (Until round 4 the positions were recomputed from scratch for every
    // tip, by pointer jumping over log2(depth) rounds with a meeting of all workgroups after each: eleven of a tip's twenty meetings.)
    // Then, per branch: how the distance to X changes across it, entered into a difference array over the visiting positions
    // (+d where the subtree starts, -d where it ends): the distance at a node is the running sum up to its position.
    int32_t* const pre_of = b.vP[0];
    int32_t* const E = b.vD[X & 1];                                  // zeroed while the previous tip was being placed
    {
      const int g_S = __hip_atomic_load(&b.gdesc[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int g_pS = b.gdesc[1], g_zS = b.gdesc[2], g_swap = b.gdesc[3], g_zU = b.gdesc[4];
      for (int i = gtid; i < nl; i += GT) {
        const int v = node_of(i);
        int p = pre_of[v], z = b.sz[v];
        if (g_S != EMAT_NO_NODE && v != X - 1 && v != n + X - 2) {     // (the previous graft's own two nodes were given their values by the grafting thread)
          if (v != g_S && p <= g_pS && g_pS < p + z) z += 2;            // an ancestor of the graft point
          if (!g_swap) { if (p >= g_pS + g_zS) p += 2; else if (p >= g_pS) p += 1; }
          else { if (p >= g_pS + g_zS + g_zU) p += 2; else if (p >= g_pS + g_zS) p -= g_zS; else if (p >= g_pS) p += g_zU + 1; }
          pre_of[v] = p; b.sz[v] = z;
        }
        const MutRec* m = b.pool + b.ml_off[v]; const int nm = b.ml_cnt[v];
        int d = 0;
        for (int k = 0; k < nm; ++k) d += b_step(xt, m[k]);
        b.delta[v] = d;
        if (v == root) atomicAdd(&E[0], dxn);                        // at the root the distance is the number of X's own deltas
        else if (d != 0) { atomicAdd(&E[p], d); atomicAdd(&E[p + z], -d); }
      }
    }
    if (!grid.sync()) return;
    // (B) running sum of the difference array: every workgroup scans its own stretch of the positions, 1 024 at a time, and
    // publishes its total; a node's distance is its stretch's running sum plus the totals of the stretches before
    const int chunk = ((nl + nb - 1) / nb + NT - 1) / NT * NT, p_lo = blk * chunk, p_hi = p_lo + chunk < nl ? p_lo + chunk : nl;
    int32_t* const PS = b.anc[0];
    {
      if (tid == 0) s_carry = 0;
      __syncthreads();
      for (int base = p_lo; base < p_hi; base += NT) {
        const int p = base + tid;
        const int mine = p < p_hi ? E[p] : 0;
        const int incl = b_block_scan_incl(mine, s_wave, tid);
        if (p < p_hi) PS[p] = s_carry + incl;
        __syncthreads();
        if (tid == NT - 1) s_carry += incl;
        __syncthreads();
      }
      if (tid == 0) b.blk_sum2[blk] = s_carry;
    }
    if (!grid.sync()) return;
    for (int q = tid; q < nb; q += NT) s_cb[q] = b.blk_sum2[q];
    __syncthreads();
    if (tid == 0) { int acc = 0; for (int q = 0; q < nb; ++q) { const int t2 = s_cb[q]; s_cb[q] = acc; acc += t2; } }
    __syncthreads();
    { int32_t* const En = b.vD[(X & 1) ^ 1]; const int ez = nl + 4 < 2 * n ? nl + 4 : 2 * n; for (int i = gtid; i < ez; i += GT) En[i] = 0; }   // the next tip's difference array (two nodes more)
    int32_t* const Dend = b.anc[1];                                  // distance to X at every node (the end of its branch)
    const int32_t* pre = pre_of;
    // (3) the fewest mutations any region offers.  A region of branch v is the stretch before its k-th mutation (k = 0 .. nm);
    // regions in X's future do not count, the one that straddles t_X ends there (spr_study.cpp:211-224)
    {
      int local = 0x7fffffff;
      for (int i = gtid; i < nl; i += GT) {
        const int v = node_of(i);
        { const int p = pre[v]; Dend[v] = PS[p] + s_cb[p / chunk]; }
        if (v == root) { if (dxn < local) local = dxn; continue; }   // the region above the root is always there
        const MutRec* m = b.pool + b.ml_off[v]; const int nm = b.ml_cnt[v];
        int D = Dend[v] - b.delta[v];
        double tmin = b.t[b.parent[v]];
        for (int k = 0; k <= nm; ++k) {
          if (!(tmin >= t_X) && D < local) local = D;
          if (k < nm) { D += b_step(xt, m[k]); tmin = m[k].t; }
        }
      }
      for (int ofs = 32; ofs > 0; ofs >>= 1) { const int o = __shfl_down(local, ofs, 64); if (o < local) local = o; }
      if ((tid & 63) == 0 && local != 0x7fffffff) atomicMin(b.gmin, local);
    }
    if (!grid.sync()) return;
    const int all_min = __hip_atomic_load(b.gmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool above_root = dxn == all_min;                          // "always pick above the root if that's a possibility" (:938-942)
    if (!above_root) {
      // (4) the tying regions in visiting order: count per node, exclusive scan over the visiting positions, write
      for (int i = gtid; i < nl; i += GT) {
        const int v = node_of(i);
        int c = 0;
        if (v != root) {
          const MutRec* m = b.pool + b.ml_off[v]; const int nm = b.ml_cnt[v];
          int D = Dend[v] - b.delta[v];
          double tmin = b.t[b.parent[v]];
          for (int k = 0; k <= nm; ++k) {
            if (!(tmin >= t_X) && D == all_min) ++c;
            if (k < nm) { D += b_step(xt, m[k]); tmin = m[k].t; }
          }
        }
        b.cnt[pre[v]] = c; b.inv[pre[v]] = v;
      }
      if (!grid.sync()) return;
      // every workgroup scans its own stretch of the visiting positions, NT at a time, and publishes its total
      if (tid == 0) s_carry = 0;
      __syncthreads();
      for (int base = p_lo; base < p_hi; base += NT) {
        const int p = base + tid;
        const int mine = p < p_hi ? b.cnt[p] : 0;
        const int incl = b_block_scan_incl(mine, s_wave, tid);
        if (p < p_hi) b.off[p] = s_carry + incl - mine;
        __syncthreads();
        if (tid == NT - 1) s_carry += incl;
        __syncthreads();
      }
      if (tid == 0) b.blk_sum[blk] = s_carry;
      if (!grid.sync()) return;
      if (tid == 0) { int base = 0; for (int q = 0; q < blk; ++q) base += b.blk_sum[q]; s_base = base; }
      __syncthreads();
      for (int p = p_lo + tid; p < p_hi; p += NT) {
        if (b.cnt[p] == 0) continue;
        const int v = b.inv[p];
        const MutRec* m = b.pool + b.ml_off[v]; const int nm = b.ml_cnt[v];
        int D = Dend[v] - b.delta[v], o = s_base + b.off[p];
        double tmin = b.t[b.parent[v]];
        for (int k = 0; k <= nm; ++k) {
          const double tmax = k < nm ? m[k].t : b.t[v];
          if (!(tmin >= t_X) && D == all_min) { if ((uint32_t)o < b.tie_cap) { b.tie_node[o] = v; b.tie_tmin[o] = tmin; b.tie_tmax[o] = tmax > t_X ? t_X : tmax; } ++o; }
          if (k < nm) { D += b_step(xt, m[k]); tmin = m[k].t; }
        }
      }
      if (!grid.sync()) return;
    }
    // (5) the first workgroup picks the region and makes the graft (:925-1030).  One thread decides -- the sums over the tying regions,
    // the draws and the composition of the deltas are sequential by definition -- but it works on LDS: the workgroup's other threads
    // stage what it reads (the tying regions' lengths, the mutations on the path from the root to the graft point, X's own deltas)
    // and carry what it wrote back out (X's new mutations).  Inputs too large for the staging areas take the same steps on the
    // buffers in HBM, as before round 4 (0.42 ms of a tip's 1.13 ms at C4 went into those single-lane HBM round trips).
    const int P = X + n - 1;
    const long long tg0 = grafter ? (long long)wall_clock64() : 0ll;
    int S = root; double t_P = 0.0;
    long long tg1 = 0;
    if (blk == 0) {
      if (grafter) {
        b.prof[1] += tg0 - t_tip0;
        int n_tie = 0; int stop = 0;
        if (!above_root) { for (int q = 0; q < nb; ++q) n_tie += b.blk_sum[q]; if ((uint32_t)n_tie > b.tie_cap) { b.status[0] = 3; b.status[1] = X; stop = 1; } }
        s_gi[0] = n_tie; s_gi[1] = stop; s_gi[2] = -1 /* chosen */; s_gi[3] = 0 /* found */;
        b.prof[7] += n_tie;
      }
      __syncthreads();
      const int n_tie = s_gi[0];
      if (s_gi[1] == 0 && !above_root) {
        // the total length of the tying regions, in visiting order, then the region the draw falls into: 1 024 lengths at a time
        double tot_min_T = 0.0;
        for (int base = 0; base < n_tie; base += NT) {
          if (base + tid < n_tie) s_len[tid] = b.tie_tmax[base + tid] - b.tie_tmin[base + tid];
          __syncthreads();
          if (grafter) { const int m = n_tie - base < NT ? n_tie - base : NT; for (int i = 0; i < m; ++i) tot_min_T += s_len[i]; }
          __syncthreads();
        }
        double insertion_cum_t = 0.0, so_far_min_T = 0.0;
        if (grafter) insertion_cum_t = rng.uniform_co(0.0, tot_min_T);
        for (int base = 0; base < n_tie; base += NT) {
          if (s_gi[3] != 0) break;                                      // (uniform: written before the previous barrier)
          if (base + tid < n_tie) s_len[tid] = b.tie_tmax[base + tid] - b.tie_tmin[base + tid];
          __syncthreads();
          if (grafter) { const int m = n_tie - base < NT ? n_tie - base : NT; for (int i = 0; i < m; ++i) { so_far_min_T += s_len[i]; if (insertion_cum_t <= so_far_min_T) { s_gi[2] = base + i; s_gi[3] = 1; break; } } }
          __syncthreads();
        }
        if (grafter) {
          tg1 = (long long)wall_clock64(); b.prof[2] += tg1 - tg0;
          const int chosen = s_gi[2];
          if (chosen < 0) { b.status[0] = 4; b.status[1] = X; s_gi[1] = 1; }
          else {
            S = b.tie_node[chosen];
            t_P = rng.uniform_oo(b.tie_tmin[chosen], b.tie_tmax[chosen]);
            s_gi[4] = S; s_gd[0] = t_P;
          }
        }
        __syncthreads();
      }
      if (!above_root && s_gi[1] == 0) {
        {
          // the path from the graft point up to the root: not walked (a chain of dependent loads as long as the tree is deep) but FOUND --
          // the ancestors of S are the nodes whose stretch of the visiting order contains S's place -- by the whole workgroup, and put
          // in order by their own places (the deeper, the later in the visiting order).  (Spreading the search over all workgroups
          // costs two more meetings and was measured no faster at C4.)
          const int S0 = s_gi[4], pS = pre_of[S0];
          uint32_t* const pk = (uint32_t*)s_len;
          if (tid == 0) s_gi[5] = 0;
          __syncthreads();
          for (int i0 = tid; i0 < nl; i0 += 8 * NT) {                   // eight nodes per thread at a time: sixteen loads in flight, not one round trip after another
            int vv[8], pp[8], zz[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int i = i0 + u * NT; vv[u] = node_of(i < nl ? i : nl - 1); }
#pragma unroll
            for (int u = 0; u < 8; ++u) { pp[u] = pre_of[vv[u]]; zz[u] = b.sz[vv[u]]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (i0 + u * NT < nl && pp[u] <= pS && pS < pp[u] + zz[u]) { const int slot = atomicAdd(&s_gi[5], 1); if (slot < k_g_path) { pk[slot] = (uint32_t)pp[u] + 1u; s_path[slot] = vv[u]; } }
          }
          __syncthreads();
          const int np = s_gi[5];
          if (np <= k_g_path) {
            int P2 = 2; while (P2 < np) P2 <<= 1;
            if (tid >= np && tid < P2) { pk[tid] = 0u; s_path[tid] = EMAT_NO_NODE; }
            __syncthreads();
            if (P2 <= 64) {                                              // (the usual case: one wavefront sorts in registers)
              if (tid < 64) { uint32_t key = tid < P2 ? pk[tid] : 0u; int val = tid < P2 ? s_path[tid] : EMAT_NO_NODE; b_wave_sort(key, val, P2, true); if (tid < P2) { pk[tid] = key; s_path[tid] = val; } }
              __syncthreads();
            } else
            for (int k2 = 2; k2 <= P2; k2 <<= 1)
              for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                if (tid < P2) { const int q = tid ^ j2; if (q > tid) { const uint32_t a1 = pk[tid], a2 = pk[q]; const bool down = (tid & k2) == 0; if ((a1 < a2) == down) { pk[tid] = a2; pk[q] = a1; const int t1 = s_path[tid]; s_path[tid] = s_path[q]; s_path[q] = t1; } } }
                __syncthreads();
              }
          } else if (grafter) { int k = 0; for (int v = S0; v != EMAT_NO_NODE; v = b.parent[v]) b.path[k++] = v; }   // deeper than the staging area: walked, into HBM
          __syncthreads();
          if (grafter) b.prof[6] += (long long)wall_clock64() - tg1;     // finding the path
        }
      }
      if (grafter && tg1 == 0) tg1 = (long long)wall_clock64();
      int stop = s_gi[1];
      BDelta* sd = b.sd; int nsd = dxn; bool bad = false, full = false;
      if (!stop && !above_root) {
        // deltas (S, t_P) -> X: X's own deltas against the root sequence, with the mutations on the way down from the root put in
        // front, inverted (site_deltas.cpp:40-80) -- root first, each branch's mutations in time order, up to t_P
        S = s_gi[4]; t_P = s_gd[0];
        const int np = s_gi[5];
        const double t_root = b.t[root];
        bool in_lds = np <= k_g_path;
        if (in_lds) {
          for (int i = tid; i < np; i += NT) {
            const MutRec* m = b.pool + b.ml_off[s_path[i]]; const int nm = b.ml_cnt[s_path[i]];
            int c = 0; for (int k = 0; k < nm; ++k) if (t_root <= m[k].t && m[k].t <= t_P) ++c;
            s_pcnt[i] = c;
          }
          __syncthreads();
          if (grafter) { int acc = 0; for (int i = np - 1; i >= 0; --i) { const int c = s_pcnt[i]; s_pcnt[i] = acc; acc += c; } s_gi[6] = acc; }
          __syncthreads();
          in_lds = s_gi[6] <= k_g_pm && dxn + s_gi[6] <= k_g_sd;
        }
        if (in_lds) {
          for (int i = tid; i < np; i += NT) {
            const MutRec* m = b.pool + b.ml_off[s_path[i]]; const int nm = b.ml_cnt[s_path[i]];
            int o = s_pcnt[i]; for (int k = 0; k < nm; ++k) if (t_root <= m[k].t && m[k].t <= t_P) s_pm[o++] = m[k];
          }
          for (int k = tid; k < dxn; k += NT) { BDelta d; d.site = b.d_site[dx0 + k]; d.from = b.ref[d.site]; d.to = b.d_to[dx0 + k]; d.pad = 0; s_sd[k] = d; }
          __syncthreads();
          sd = s_sd;
          const int T = s_gi[6];
          if (T > 0 && T <= 1024 && dxn < 1024 && T < k_g_path && dxn < k_g_path && b.L < (1 << 22)) {   // (keys and ranks live in s_len, s_path and s_pcnt)
            // What a run of push_front calls leaves at a site depends on that site's entry and on its own mutations, in path order,
            // alone: the sites are independent.  So: sort the path's mutations by (site, place on the path) -- a bitonic sort
            // in LDS --, let one thread per site run that site's calls (the same operations in the same order as the one-by-one
            // list edits, broken chains flagged the same way), and merge what is left with X's untouched deltas by rank.
            uint32_t* const keys = (uint32_t*)s_len; uint32_t* const rres = keys + 1024;      // s_len: 8 KB
            int* const cDex = s_path; int* const cRex = s_pcnt;                               // free once the path is staged
            int P2 = 2; while (P2 < T) P2 <<= 1;
            if (tid < P2) keys[tid] = tid < T ? ((uint32_t)s_pm[tid].site << 10) | (uint32_t)tid : 0xFFFFFFFFu;
            if (tid < 1024) rres[tid] = 0u;
            __syncthreads();
            if (P2 <= 64) {
              if (tid < 64) { uint32_t key = tid < P2 ? keys[tid] : 0xFFFFFFFFu; int val = 0; b_wave_sort(key, val, P2, false); if (tid < P2) keys[tid] = key; }
              __syncthreads();
            } else
            for (int k2 = 2; k2 <= P2; k2 <<= 1)
              for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                if (tid < P2) { const int q = tid ^ j2; if (q > tid) { const uint32_t a1 = keys[tid], a2 = keys[q]; const bool up = (tid & k2) == 0; if ((a1 > a2) == up) { keys[tid] = a2; keys[q] = a1; } } }
                __syncthreads();
              }
            if (tid == 0) s_gi[3] = 0;                                 // "a chain was broken"
            __syncthreads();
            if (tid < T && (tid == 0 || (keys[tid] >> 10) != (keys[tid - 1] >> 10))) {
              const int site = (int)(keys[tid] >> 10);
              int lo = 0, hi = dxn;
              while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_sd[mid].site < site) lo = mid + 1; else hi = mid; }
              bool has = lo < dxn && s_sd[lo].site == site;
              int from = has ? (int)s_sd[lo].from : 0, to = has ? (int)s_sd[lo].to : 0;
              if (has) s_sd[lo].pad = 1;                               // replaced by what this run leaves
              bool broke = false;
              for (int q = tid; q < T && (int)(keys[q] >> 10) == site; ++q) {
                const MutRec& m = s_pm[keys[q] & 1023u];
                if (has) { if ((int)m.from != from) broke = true; from = (int)m.to; if (from == to) has = false; }      // b_push_front(site, m.to, m.from) on an entry
                else { has = true; from = (int)m.to; to = (int)m.from; }                                               // ... and on none
              }
              if (broke) s_gi[3] = 1;
              rres[tid] = has ? (1u << 16) | ((uint32_t)from << 8) | (uint32_t)to : 0u;
            }
            __syncthreads();
            // ranks: kept deltas of X and kept run results, each in site order
            { const int keepD = tid < dxn && s_sd[tid].pad == 0 ? 1 : 0; const int incl = b_block_scan_incl(keepD, s_wave, tid); if (tid < dxn) cDex[tid] = incl - keepD; if (tid == dxn) cDex[dxn] = incl; __syncthreads(); }
            { const int keepR = tid < T && (rres[tid] >> 16) != 0u ? 1 : 0; const int incl = b_block_scan_incl(keepR, s_wave, tid); if (tid < T) cRex[tid] = incl - keepR; if (tid == T) cRex[T] = incl; __syncthreads(); }
            BDelta* const outp = (BDelta*)s_pm;                        // the path's mutations are spent
            const int totD = dxn < 1024 ? cDex[dxn] : 0;
            const int totR = T < 1024 ? cRex[T] : cRex[1023] + ((rres[1023] >> 16) != 0u ? 1 : 0);
            __syncthreads();
            if (tid < dxn && s_sd[tid].pad == 0) {
              const uint32_t want = (uint32_t)s_sd[tid].site << 10;
              int lo = 0, hi = T; while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < want) lo = mid + 1; else hi = mid; }
              outp[cDex[tid] + (lo < T ? cRex[lo] : totR)] = s_sd[tid];
            }
            if (tid < T && (rres[tid] >> 16) != 0u) {
              const int site = (int)(keys[tid] >> 10);
              int lo = 0, hi = dxn; while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_sd[mid].site < site) lo = mid + 1; else hi = mid; }
              BDelta d; d.site = site; d.from = (uint8_t)((rres[tid] >> 8) & 255u); d.to = (uint8_t)(rres[tid] & 255u); d.pad = 0;
              outp[cRex[tid] + (lo < dxn ? cDex[lo] : totD)] = d;
            }
            __syncthreads();
            nsd = totD + totR;
            for (int k = tid; k < nsd; k += NT) s_sd[k] = outp[k];
            if (s_gi[3] != 0) bad = true;
            __syncthreads();
          } else if (grafter) { for (int j = 0; j < T && !full; ++j) { if (!b_push_front(s_sd, nsd, (uint32_t)k_g_sd, s_pm[j].site, s_pm[j].to, s_pm[j].from, bad)) full = true; } }
        } else if (grafter) {
          for (int k = 0; k < dxn; ++k) { b.sd[k].site = b.d_site[dx0 + k]; b.sd[k].from = b.ref[b.d_site[dx0 + k]]; b.sd[k].to = b.d_to[dx0 + k]; b.sd[k].pad = 0; }
          // root first.  (The path, graft point first, is in s_path when it was FOUND -- np <= k_g_path -- and in b.path when it had to be
          // walked: until round 4's fuzz hunt this loop read b.path either way, i.e. stale nodes whenever a path that fitted the staging
          // area met a tip whose deltas + path mutations did not: a tip with 537 deltas, seed 777001 case 4.)
          const bool path_in_lds = np <= k_g_path;
          for (int i = np - 1; i >= 0 && !full; --i) {
            const int pv = path_in_lds ? s_path[i] : b.path[i];
            const MutRec* m = b.pool + b.ml_off[pv]; const int nm = b.ml_cnt[pv];
            for (int k = 0; k < nm; ++k) if (t_root <= m[k].t && m[k].t <= t_P) { if (!b_push_front(b.sd, nsd, b.sd_cap, m[k].site, m[k].to, m[k].from, bad)) { full = true; break; } }
          }
        }
        if (grafter) {
          const int G = b.parent[S];
          const int U = b.c0[G] == S ? b.c1[G] : b.c0[G];
          // what the graft does to the visiting order (phase A of the next tip): S's subtree makes room for P in front of it and for
          // X behind it; had it been G's second child (visited first), it now comes after its sibling's subtree
          { const int pS = pre_of[S], zS = b.sz[S], swap = b.c1[G] == S ? 1 : 0, zU = b.sz[U];
            b.gdesc[1] = pS; b.gdesc[2] = zS; b.gdesc[3] = swap; b.gdesc[4] = zU;
            pre_of[P] = swap ? pS + zU : pS; pre_of[X] = (swap ? pS + zU : pS) + 1 + zS; }
          b.c0[G] = P; b.c1[G] = U; b.parent[P] = G;
          int split = 0;                                               // the mutations of G-S before t_P now sit on G-P
          { const MutRec* m = b.pool + b.ml_off[S]; const int nm = b.ml_cnt[S]; while (split < nm && !(m[split].t > t_P)) ++split; }
          b.ml_off[P] = b.ml_off[S]; b.ml_cnt[P] = split; b.ml_off[S] += (uint32_t)split; b.ml_cnt[S] -= split;
          b.prof[3] += (long long)wall_clock64() - tg1;
        }
      } else if (!stop) {
        // above the root: X's deltas against the root sequence are the mutations it needs
        const bool in_lds = dxn <= k_g_sd;
        if (in_lds) { for (int k = tid; k < dxn; k += NT) { BDelta d; d.site = b.d_site[dx0 + k]; d.from = b.ref[d.site]; d.to = b.d_to[dx0 + k]; d.pad = 0; s_sd[k] = d; } sd = s_sd; }
        else if (grafter) for (int k = 0; k < dxn; ++k) { b.sd[k].site = b.d_site[dx0 + k]; b.sd[k].from = b.ref[b.d_site[dx0 + k]]; b.sd[k].to = b.d_to[dx0 + k]; b.sd[k].pad = 0; }
        if (grafter) {
          S = root;
          const double t_P_guess = t_X - (double)nsd * 13.0, t_S = b.t[S];
          t_P = (t_P_guess < t_S ? t_P_guess : t_S) - 1.0;
          *b.root = P; b.parent[P] = EMAT_NO_NODE;
          b.ml_off[P] = b.ml_off[S]; b.ml_cnt[P] = b.ml_cnt[S]; b.ml_cnt[S] = 0;      // the root's list moves up with the root (it is empty while building)
          { const int zS = b.sz[S]; b.gdesc[1] = 0; b.gdesc[2] = zS; b.gdesc[3] = 0; b.gdesc[4] = 0; pre_of[P] = 0; pre_of[X] = zS + 1; }   // P first, the old tree, X last
        }
      }
      __syncthreads();
      if (grafter) {
        const long long tg2 = (long long)wall_clock64();
        if (!stop) {
          if (full) { b.status[0] = 2; b.status[1] = X; stop = 1; }
          else if (bad) { b.status[0] = 4; b.status[1] = X; stop = 1; }
        }
        s_gi[7] = -1;                                                  // where X's mutations go in the pool (-1: nothing to copy)
        if (!stop) {
          b.t[P] = t_P; b.c0[P] = X; b.c1[P] = S; b.parent[X] = P; b.parent[S] = P;
          b.sz[X] = 1; b.sz[P] = b.sz[S] + 2;
          __hip_atomic_store(&b.gdesc[0], S, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // positions and ancestors' sizes follow at the start of the next tip (A)
          const long long tg3 = (long long)wall_clock64(); b.prof[4] += tg3 - tg2;
          // the mutations X needs, at random times on P-X, sorted by (t, site) (:1015-1021)
          const uint32_t o = *b.pool_top;
          if (o + (uint32_t)nsd > b.pool_cap) { b.status[0] = 1; b.status[1] = X; stop = 1; }
          else {
            const bool stage = nsd <= k_g_pm;                            // sorted in LDS, copied out by the whole workgroup
            MutRec* mx = stage ? s_pm : b.pool + o;
            for (int k = 0; k < nsd; ++k) {
              MutRec r; r.t = rng.uniform_oc(t_P, t_X); r.site = sd[k].site; r.from = sd[k].from; r.to = sd[k].to; r.pad = 0;
              int j = k - 1;                                             // stable insertion by (t, site)
              while (j >= 0 && (r.t < mx[j].t || (r.t == mx[j].t && r.site < mx[j].site))) { mx[j + 1] = mx[j]; --j; }
              mx[j + 1] = r;
            }
            b.ml_off[X] = o; b.ml_cnt[X] = nsd; *b.pool_top = o + (uint32_t)nsd;
            if (stage) { s_gi[7] = (int)o; s_gi[6] = nsd; }
          }
          b.prof[5] += (long long)wall_clock64() - tg3;
        }
        *b.stop_flag = stop ? X : 0;
        b.prof[0] += (long long)wall_clock64() - t_tip0;
      }
      __syncthreads();
      if (s_gi[7] >= 0) { MutRec* dst = b.pool + (uint32_t)s_gi[7]; const int cnt = s_gi[6]; for (int k = tid; k < cnt; k += NT) dst[k] = s_pm[k]; }
    }
    if (!grid.sync()) return;
    if (tid == 0) s_stop = __hip_atomic_load(b.stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_stop != 0) break;
  }
  if (grafter) { b.rng[1] = rng.ctr; b.rng[2] = rng.spare; b.rng[3] = rng.has_spare ? 1u : 0u; }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
struct BuiltTree { FlatTree tree; std::vector<uint8_t> ref; bool valid = false; };   // `ref`: the sequence the tree is written against (the default builder moves it to the root's)

struct BHostMut { double t; int32_t site; uint8_t from, to; };
struct BHostNode {
  int32_t parent = EMAT_NO_NODE, c0 = EMAT_NO_NODE, c1 = EMAT_NO_NODE;
  double t = 0.0; float t_min = -FLT_MAX, t_max = FLT_MAX;
  std::vector<BHostMut> muts; std::vector<std::pair<int32_t, int32_t>> miss; std::vector<std::pair<int32_t, uint8_t>> mfs;
};
using BIvs = std::vector<std::pair<int32_t, int32_t>>;
inline bool b_iv_contains(const BIvs& v, int l) {
  auto it = std::upper_bound(v.begin(), v.end(), l, [](int x, const std::pair<int32_t, int32_t>& iv) { return x < iv.first; });
  return it != v.begin() && l < std::prev(it)->second;
}
inline BIvs b_iv_merge(const BIvs& A, const BIvs& B) {        // union; touching intervals coalesce (interval_set.h:238-288)
  BIvs all(A); all.insert(all.end(), B.begin(), B.end());
  std::sort(all.begin(), all.end());
  BIvs out;
  for (auto& iv : all) { if (!out.empty() && iv.first <= out.back().second) out.back().second = std::max(out.back().second, iv.second); else out.push_back(iv); }
  return out;
}
inline BIvs b_iv_intersect(const BIvs& A, const BIvs& B) {
  BIvs out; size_t i = 0, j = 0;
  while (i < A.size() && j < B.size()) {
    int s = std::max(A[i].first, B[j].first), e = std::min(A[i].second, B[j].second);
    if (s < e) out.push_back({s, e});
    if (A[i].second <= B[j].second) ++i; else ++j;
  }
  return out;
}
inline BIvs b_iv_subtract(const BIvs& A, const BIvs& B) {
  BIvs out; size_t j = 0;
  for (auto [s, e] : A) {
    while (j < B.size() && B[j].second <= s) ++j;
    int cs = s;
    for (size_t k = j; k < B.size() && B[k].first < e; ++k) { if (B[k].first > cs) out.push_back({cs, B[k].first}); cs = std::max(cs, B[k].second); if (cs >= e) break; }
    if (cs < e) out.push_back({cs, e});
  }
  return out;
}

// fix_up_missations (phylo_tree.cpp:414-507) by its meaning: (1) what both children miss moves up to their parent, leaves first;
// (2) root first, a branch keeps only the missing sites that nothing above it already misses; (3) root first again, with the
// sequence carried along: mutations at sites missing at or above their branch go, and every missation's from-state is the state
// just above its branch where that differs from the reference sequence.
inline void b_fix_up_missations(std::vector<BHostNode>& N, int root, const std::vector<uint8_t>& ref) {
  std::vector<int> pre; pre.reserve(N.size());
  { std::vector<int> st{root}; while (!st.empty()) { int v = st.back(); st.pop_back(); pre.push_back(v); if (N[v].c0 != EMAT_NO_NODE) { st.push_back(N[v].c1); st.push_back(N[v].c0); } } }
  for (auto it = pre.rbegin(); it != pre.rend(); ++it) {           // children before parents
    BHostNode& nd = N[*it];
    if (nd.c0 == EMAT_NO_NODE) continue;
    BIvs common = b_iv_intersect(N[nd.c0].miss, N[nd.c1].miss);
    if (common.empty()) continue;
    N[nd.c0].miss = b_iv_subtract(N[nd.c0].miss, common); N[nd.c1].miss = b_iv_subtract(N[nd.c1].miss, common);
    nd.miss = b_iv_merge(nd.miss, common);
  }
  // (2) + (3) in one walk: the set missing above a branch and the sequence at its start, kept per depth
  struct Frame { int node; int stage; BIvs missing_above; };
  std::map<int32_t, uint8_t> seq;                                   // sites where the running sequence differs from `ref`
  auto state = [&](int l) { auto f = seq.find(l); return f != seq.end() ? f->second : ref[l]; };
  auto put = [&](int l, uint8_t s) { if (s == ref[l]) seq.erase(l); else seq[l] = s; };
  std::vector<Frame> st; st.push_back({root, 0, {}});
  while (!st.empty()) {
    Frame& f = st.back(); BHostNode& nd = N[f.node];
    if (f.stage == 0) {
      nd.miss = b_iv_subtract(nd.miss, f.missing_above);
      BIvs missing_here = b_iv_merge(f.missing_above, nd.miss);
      nd.mfs.clear();
      for (auto [s, e] : nd.miss) for (auto it = seq.lower_bound(s); it != seq.end() && it->first < e; ++it) nd.mfs.push_back({it->first, it->second});
      nd.muts.erase(std::remove_if(nd.muts.begin(), nd.muts.end(), [&](const BHostMut& m) { return b_iv_contains(missing_here, m.site); }), nd.muts.end());
      for (auto& m : nd.muts) { if (m.from != state(m.site)) throw std::runtime_error("mutation chain broken while fixing up missations"); put(m.site, m.to); }
      f.stage = 1;
      if (nd.c0 != EMAT_NO_NODE) { const int a = nd.c0, c = nd.c1; st.push_back({c, 0, missing_here}); st.push_back({a, 0, std::move(missing_here)}); }
      // (st may have reallocated: do not touch f / nd below)
    } else {
      for (auto it = nd.muts.rbegin(); it != nd.muts.rend(); ++it) put(it->site, it->from);
      st.pop_back();
    }
  }
}
// randomize_branch_mutation_times (phylo_tree.cpp:579-644): fresh uniform times on a branch; when a site mutates more than once
// on it, that site's times are drawn together, sorted, and handed out in the old order
inline void b_randomize_branch(BHostNode& nd, double t_P, HostRng& rng) {
  auto uni_oc = [&](double lo, double hi) { return lo + (hi - lo) * (((double)(rng.next64() >> 11) + 1.0) * 0x1.0p-53); };
  const double t_X = nd.t;
  std::map<int32_t, int> counts; bool complicated = false;
  for (auto& m : nd.muts) if (++counts[m.site] > 1) complicated = true;
  std::vector<BHostMut> out;
  if (!complicated) for (auto& m : nd.muts) out.push_back(BHostMut{uni_oc(t_P, t_X), m.site, m.from, m.to});
  else for (auto& [l, c] : counts) {
    std::vector<double> ts;
    for (auto& m : nd.muts) if (m.site == l) ts.push_back(uni_oc(t_P, t_X));
    std::sort(ts.begin(), ts.end());
    size_t k = 0;
    for (auto& m : nd.muts) if (m.site == l) out.push_back(BHostMut{ts[k++], m.site, m.from, m.to});
  }
  std::stable_sort(out.begin(), out.end(), [](const BHostMut& a, const BHostMut& b) { return a.t < b.t || (a.t == b.t && a.site < b.site); });
  nd.muts = std::move(out);
}

}  // namespace emat
#endif  // EMAT_BUILD_HPP_
