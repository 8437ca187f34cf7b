// emat_backend.hip -- HIP kernels and the C-ABI of the MI355X-native EMAT local-move engine.
//
// Boundary: include/emat_backend.h (each entry point cites the reference call it replaces).
// Execution model: one 64-lane wavefront (= one workgroup) per partition part.  A part's working set
// is one slab (emat_slab.hpp); `k_run_moves` streams it into LDS when it fits, runs the requested
// number of `Subrun::mcmc_sub_iteration` steps (reference core/subrun.cpp:98-121) there, and streams it
// back.  `k_recalc_derived` is the whole-part recomputation of reference core/subrun.cpp:17-26.
// There is no CPU fallback: without a HIP device every entry point fails with EMAT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <numeric>
#include <string>
#include <vector>
#include <map>
#include <stdexcept>

#include "../../include/emat_backend.h"
#ifdef EMAT_X_DIVERGENT
__device__ __forceinline__ uint32_t emat_opaque_zero() { uint32_t z; asm("v_mov_b32 %0, 0" : "=v"(z)); return z; }
#endif
// 1: every lane of the wave runs the chain, all with the same values (the chain's code does not depend on the lane);
// 0: lane 0 alone.  With one lane active every non-leaf device function saves and restores the INACTIVE lanes of the
// VGPR it parks its return address in (s_xor_saveexec + scratch_store / scratch_load + s_waitcnt vmcnt(0)); with all
// lanes active that save has nothing to store.
#ifndef EMAT_CHAIN_ON_ALL_LANES
#define EMAT_CHAIN_ON_ALL_LANES 0
#endif
#if defined(EMAT_PROFILE_PHASES) || defined(EMAT_COUNT_CALLS)
namespace emat {
constexpr int k_fn_replicas = 64;                          // (one table per workgroup index mod 64: eight thousand waves adding to ONE word per scope made the profiling build 5.6 x slower than the real one)
__device__ unsigned long long g_fn_ticks[k_fn_replicas * 3 * 2048][2];   // EMAT_TIMED scopes: [replica][header * 2048 + line][ticks, calls], all parts
__device__ unsigned g_fn_min_list_bytes = 0;              // ... or only the parts whose lists take at least this much (EMAT_FN_MIN_LISTS: the heavy parts near the root)
__shared__ int s_fn_count_me;
struct FnTimer {   // inclusive ticks and calls of the enclosing scope, keyed by (header, source line); lane 0 only
  int key; long long t0;
  __device__ FnTimer(int k) : key(k), t0(clock64()) {}
  __device__ void stop() { if (key >= 0) { this->~FnTimer(); key = -1; } }
  __device__ ~FnTimer() { if (key >= 0 && threadIdx.x == 0 && s_fn_count_me) { const long long dt = clock64() - t0; unsigned long long* e = g_fn_ticks[(blockIdx.x & (k_fn_replicas - 1)) * (3 * 2048) + key]; atomicAdd(&e[0], (unsigned long long)dt); atomicAdd(&e[1], 1ull); } }
};
}
#endif
#ifdef EMAT_PROFILE_PHASES
namespace emat { __device__ unsigned long long g_arena_site_bytes[2048][2]; }   // [source line & 2047][0 = LDS arena, 1 = HBM scratch], all parts
#endif
// The device code is compiled three times (see emat_device_core.hpp): `dev_lds` for parts whose whole persistent slab
// is staged in LDS, `dev_mix` for larger parts of which only header + nodes + cells are staged, and `dev` for parts
// that run entirely on their HBM slab (and for k_recalc_derived).
#define EMAT_DEV_NS dev_lds
#define EMAT_VARIANT_LDS 1
#include "emat_device_moves.hpp"
#undef EMAT_DEV_NS
#undef EMAT_VARIANT_LDS
#define EMAT_DEV_NS dev_mix
#define EMAT_VARIANT_LDS 2
#include "emat_device_moves.hpp"
#undef EMAT_DEV_NS
#undef EMAT_VARIANT_LDS
#define EMAT_DEV_NS dev
#define EMAT_VARIANT_LDS 0
#include "emat_device_moves.hpp"
#include "emat_host_model.hpp"
#include "flat_tree.hpp"
#include "host_parallel.hpp"

namespace emat {

// =================================================================================================
// Kernels
// =================================================================================================
struct KernelArgs {
  uint8_t* slabs;                 // all slabs, back to back
  const uint64_t* slab_off;       // [num_parts] byte offset of each part's slab
  const int32_t* order;           // [parts of this launch] part ids, largest first: workgroup b runs part order[b]
  int64_t* part_ticks;            // [2][num_parts] wall-clock ticks and start tick of each part's last run (occupancy timelines)
  const int32_t* ref_freqs;       // [P][4]
  const double* cum_nu;           // [L + 1][P][4]: nu-weighted counts of reference states per site partition over sites < k (k_global_stats)
  double* stats_out;              // [num_parts][k_stats_row]: per-part output of k_global_stats
  EvoTable evo;
  const PopTable* pop;
  SharedCells shared;             // the run-wide coalescent cell arrays (every part but the root part reads them here)
  RunFlags flags;
  int32_t num_parts;
  uint32_t lds_slab_bytes;        // capacity of the LDS staging area (0 = never stage)
  uint32_t lds_scratch_bytes;     // size of the per-part LDS scratch arena
  int64_t moves_per_part;
  const int64_t* moves_for_part;  // [num_parts] or null: per-part counts of a recovery launch (overrides moves_per_part / extra_moves_part0)
  int32_t* part_status;           // [num_parts] status every part ended its chain with (0 = ran to completion)
  int64_t extra_moves_part0;      // remainder of Run::run_local_moves goes to part 0 (run.cpp:683-689)
  int32_t one_more_below;         // parts [0, one_more_below) do one move more (emat_run_moves_even)
  // A pass cut into `chunks` tickets per part (EMAT_CHUNKS): workgroup b runs ticket b / class_stride of part order[b % class_stride];
  // ticket c of a part starts when ticket c - 1 has written the slab back (chunk_done[part] == c).  Shorter tickets pack the
  // slots better at the end of a pass; the chain -- one RNG stream per part, state in the slab -- is the same chain.
  int32_t chunks, class_count, class_stride, taper;
  int32_t* side_started;          // side classes of a synchronised pass: every workgroup counts itself in as it starts (k_wait_side_start), else nullptr
  int32_t single_below;           // the first `single_below` slots of the launch order (the parts expected to run longest) do their whole pass in ONE ticket, from the start of the pass: longest jobs first and unsplit, the many short chains -- in tickets -- fill in around them
  int32_t full_release;           // EMAT_TICKET_RELEASE=full: every ticket hands its part over with an agent-scope RELEASE (the path that needs no assumption about where workgroups run)
  int32_t cum_w[8];               // cumulative ticket weights (EMAT_TICKET_WEIGHTS) or zeros
  int32_t* chunk_done;            // [num_parts], zeroed before the launch
  // Room for a copy of every slab's persistent prefix, at the slab's own offset: a leg that runs on the HBM slab itself (part
  // not staged whole) saves it there first, so that a container overflowing INSIDE a move can be answered by putting the
  // leg's starting state back and asking the host for more room, as for a staged leg (which never touches the HBM copy).
  uint8_t* snaps;
};

constexpr int k_wave = 64;
constexpr int k_ticket_log = 8;   // tickets per part whose workgroup entry / exit ticks are kept (emat_debug_ticket_ticks)

__device__ inline void wave_copy16(uint8_t* dst, const uint8_t* src, uint32_t bytes, int lane) {
  const uint4* s = (const uint4*)src; uint4* d = (uint4*)dst;
  for (uint32_t i = lane; i < bytes / 16; i += k_wave) d[i] = s[i];
}
// The same copy with stores that are written through to where every XCD sees them (agent-scope atomics, 8 bytes each): what a
// ticket hands to the next one goes out this way, so that handing over needs no write-back of the XCD's whole L2 (below).
__device__ inline void wave_copy8_through(uint8_t* dst, const uint8_t* src, uint32_t bytes, int lane) {
  const uint64_t* s = (const uint64_t*)src; uint64_t* d = (uint64_t*)dst;
  for (uint32_t i = lane; i < bytes / 8; i += k_wave) __hip_atomic_store(&d[i], s[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ inline double wave_sum(double x) {
  for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, k_wave);
  return __shfl(x, 0, k_wave);
}

// Fills the model pointers of a context; HKY tables come from LDS when they were staged.
// `slab` = where slab offsets resolve for the generic code variant (the HBM slab); the LDS variants address header, nodes and
// lists through compile-time LDS addresses and only need `H` here, to pick up the RNG position and flags.
template <class CtxT> __device__ inline void init_ctx(CtxT& c, uint8_t* slab, uint8_t* gslab, const KernelArgs& a, const double* lds_tables, SlabHeader* H = nullptr) {
  c.S = slab; c.G = gslab; c.H = H != nullptr ? H : (SlabHeader*)slab; c.N = (NodeRec*)(slab + c.H->off_nodes);
  c.L = a.evo.num_sites;
  c.ref = (const __attribute__((address_space(1))) uint8_t*)a.evo.ref_sequence; c.part = (const __attribute__((address_space(1))) uint8_t*)a.evo.partition_for_site;
  c.nu = (const __attribute__((address_space(1))) double*)a.evo.nu_l; c.cumQ = (const __attribute__((address_space(1))) double*)a.evo.cum_Q_l;
  c.have_logq = lds_tables != nullptr;
  c.uniform_sites = a.evo.uniform_sites != 0;
  if (lds_tables) { c.mu = lds_tables; c.pi = lds_tables + k_max_lds_partitions; c.q = lds_tables + k_max_lds_partitions * 5; }
  else { c.mu = a.evo.mu; c.pi = a.evo.pi; c.q = a.evo.q; }
  c.pop = a.pop;
  c.sh_ktw = a.shared.k_twiddle_bar; c.sh_tsop = a.shared.ts_over_pop; c.sh_nact = a.shared.num_active_parts;
  c.t_max_tip = a.flags.t_max_tip;
  c.only_displacing_inner_nodes = a.flags.only_displacing_inner_nodes != 0;
  c.topology_moves_enabled = a.flags.topology_moves_enabled != 0;
  c.includes_run_root = (c.H->flags & k_flag_includes_run_root) != 0;
  c.rng_key = c.H->rng_key; c.rng_ctr = c.H->rng_counter; c.rng_spare = c.H->rng_spare; c.rng_has_spare = c.H->rng_has_spare != 0; c.rng_short = false; c.phase = 0; c.svc = 0; c.frame = nullptr;
  c.rng_base = c.rng_ctr - (uint64_t)k_rng_blocks;   // nothing computed ahead yet: the chain's first step asks the wave for it
  c.mu_prop = 0.0; c.sc_top = c.H->scratch_begin; c.A = nullptr; c.a_top = 0; c.a_end = 0; c.failed = false; c.bytes = 0; c.bytes_w = 0;
  c.tr_kind = -1.0; c.tr_node = -1.0; c.tr_acc = 0.0; c.tr_log_mh = 0.0;
}
__device__ inline const double* stage_tables(const KernelArgs& a, double* lds_tables, int lane) {
  if (a.evo.num_partitions > k_max_lds_partitions) return nullptr;
  const int P = a.evo.num_partitions;
  for (int i = lane; i < P; i += k_wave) lds_tables[i] = a.evo.mu[i];
  for (int i = lane; i < P * 4; i += k_wave) lds_tables[k_max_lds_partitions + i] = a.evo.pi[i];
  for (int i = lane; i < P * 16; i += k_wave) lds_tables[k_max_lds_partitions * 5 + i] = a.evo.q[i];
  for (int i = lane; i < P * 16; i += k_wave) { const double arg = a.evo.mu[i / 16] * 1.0 * a.evo.q[i]; emat_lds_logq[i] = arg > 0.0 ? dev::m_log(arg) : 0.0; }   // (the diagonal is never asked for)
  return lds_tables;
}

// ---- the hot path: `moves` sub-iterations on every part -------------------------------------------------
// One wavefront (= one workgroup) per part.  Every part is its own Markov chain with its own RNG stream, so the
// launch schedule does not change any result.  The chain itself is serial and runs on lane 0; all 64 lanes move the
// slab between HBM and LDS.  LDS layout: static [HKY tables], static [context], dynamic [staged slab][optional scratch arena]; all
// first three sit at compile-time offsets, which is what lets the `dev_lds` variant address them with DS instructions.
#ifndef EMAT_WAVES_PER_EU
#define EMAT_WAVES_PER_EU 4
#endif
// Every kernel that calls into the out-of-line device functions carries the same occupancy target: the register budget
// of a callee is the tightest one among its callers' targets only if ALL its callers have one (the debug kernels share
// `dev::` code with k_run_moves).  HIP's __launch_bounds__ second argument is ignored by the AMDGPU backend.
#define EMAT_OCCUPANCY __attribute__((amdgpu_waves_per_eu(EMAT_WAVES_PER_EU, EMAT_WAVES_PER_EU)))
// Room the list heap of an LDS-staged part keeps above its used size (measured at C4: 2048 -> 288, 1024 -> 296, 768 -> 287,
// 512 -> 282 M moves/s: less room leaves more LDS to the moves' scratch arena, too little sends parts through heap
// compactions and the HBM fall-back leg).
constexpr uint32_t k_lds_heap_room = 1024;

// Which XCD (accelerator complex) this wavefront runs on: hardware register XCC_ID (id 20, bits 3:0).  The eight XCDs of an
// MI355X have an L2 each, and what a ticket hands to its successor through write-through stores is only guaranteed to be seen by a
// successor behind the SAME L2 (a cross-XCD hand-over through that path was caught reading a stale slab by
// tests/test_parity_gpu.py::test_tickets_handed_over_across_xcds..., about once in twenty passes): the cheap hand-over is therefore
// only taken when the launch puts a part's tickets on one XCD -- checked, not assumed: the host verifies once per device how
// workgroups are dealt to XCDs (k_probe_xcc), and every ticket checks that its predecessor ran where it runs itself.
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15u); }
__global__ void k_probe_xcc(int32_t* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

template <bool kSide> __device__ __forceinline__ void run_moves_body(const KernelArgs& a) {
  const int lane = threadIdx.x;
  if (kSide && a.side_started != nullptr && lane == 0) __hip_atomic_fetch_add(a.side_started, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // resident: the main class may come
  double* lds_tables = (double*)emat_lds_tables;
  const bool tables_staged = stage_tables(a, lds_tables, lane) != nullptr;
  uint8_t* lds_slab = emat_lds;
  int* lds_flag = (int*)(emat_lds_ctx + k_lds_ctx_bytes - 16);   // spare tail of the context slot: lane 0 -> all lanes
  const int chunk = a.chunks > 1 ? (int)blockIdx.x / a.class_stride : 0, slot = a.chunks > 1 ? (int)blockIdx.x % a.class_stride : (int)blockIdx.x;
  if (a.chunks > 1 && slot >= a.class_count) return;   // padding: the stride is a multiple of 8 so that a part's tickets land on one XCD
  const bool single = a.chunks > 1 && slot < a.single_below;   // this part's pass is one ticket: its later tickets have nothing to do
  if (single && chunk > 0) return;
  const int part = a.order[slot];
  // Per-part words that one ticket of a part writes and the next reads (status, chain ticks): agent-scope atomics, which every
  // XCD sees without a cache write-back.
  if (lane == 0 && chunk < k_ticket_log) a.part_ticks[(size_t)(2 + 2 * chunk) * a.num_parts + part] = (int64_t)wall_clock64();   // the workgroup is resident from here ...
  auto st_status = [&](int32_t v) { __hip_atomic_store(&a.part_status[part], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto ld_status = [&]() -> int32_t { return __hip_atomic_load(&a.part_status[part], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto st_ticks = [&](int64_t v) { __hip_atomic_store(&a.part_ticks[part], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto ld_ticks = [&]() -> int64_t { return __hip_atomic_load(&a.part_ticks[part], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  uint8_t* gslab = a.slabs + a.slab_off[part];
  SlabHeader* gh = (SlabHeader*)gslab;
#ifndef EMAT_X_DYN_LDS_BY_TABLE
  // the staged variants address the dynamic LDS block at a constant (k_lds_dyn_base, emat_device_core.hpp): is it where they think it is?
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)emat_lds != k_lds_dyn_base) {
    if (lane == 0) { gh->fail_line = -3; st_status(k_part_internal); if (a.chunks > 1) __hip_atomic_store(&a.chunk_done[part], a.chunks, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
    return;
  }
#endif
  const int64_t pass_target = a.moves_for_part ? a.moves_for_part[part] : a.moves_per_part + (part == 0 ? a.extra_moves_part0 : 0) + (part < a.one_more_below ? 1 : 0);
  int64_t target = pass_target, done_at_start;
  if (a.chunks > 1) {
    if (chunk > 0) {   // the ticket before this one must have written the part's slab back
      // (workgroups are dispatched in index order, so that ticket is running or done: the wait cannot deadlock; it is
      // bounded all the same, and a part whose turn never came is reported, not waited for)
      if (lane == 0) {
        // (bounded by wall-clock time -- two minutes of the 100 MHz counter -- not by a number of polls: a forced ticket count on a
        // handful of huge parts that run out of HBM makes the earlier tickets of a part take seconds, found by the fuzz rounds)
        const uint64_t w0 = wall_clock64();
        bool waited = false, gave_up = false;
        int32_t seen;
        while (((seen = __hip_atomic_load(&a.chunk_done[part], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) & 0xff) < chunk) {
          __builtin_amdgcn_s_sleep(64); waited = true;
          if (wall_clock64() - w0 > 12000000000ull) { gave_up = true; break; }
        }
        // the predecessor handed the part over without writing its L2 back: it must have run behind this very L2
        if (!gave_up && (seen & 0x100) != 0 && ((seen >> 12) & 15) != xcc_id()) { gave_up = true; gh->fail_line = -(int32_t)__LINE__; }
        if (waited) atomicAdd((unsigned long long*)&a.chunk_done[((a.num_parts + 1) & ~1) + 2 * (blockIdx.x & 63)], (unsigned long long)(wall_clock64() - w0));   // (EMAT_VERBOSE: slot time spent waiting)
        // (a ticket that gives up says so to its successors as well: they would each wait out their own two minutes -- ADVICE round 4)
        if (gave_up) { st_status(k_part_internal); __hip_atomic_store(&a.chunk_done[part], a.chunks, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); *lds_flag = -1; } else *lds_flag = 0;
      }
      __syncthreads();
      if (*lds_flag == -1) return;
    } else if (lane == 0) __hip_atomic_store(&gh->pad0, (uint32_t)gh->moves_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // where the pass picked the part up (low bits: a pass is far shorter than 2^32 moves); written through like everything else a later ticket reads
    __syncthreads();
    // cumulative over the tickets so far.  The tickets of a part shrink in the ratio n : n-1 : ... : 1, so that the jobs that start
    // last -- the ones a pass ends with -- are the shortest (equal tickets: EMAT_TICKET_TAPER=0)
    if (single) target = pass_target;
    else if (a.cum_w[0] > 0) target = chunk + 1 == a.chunks ? pass_target : pass_target * a.cum_w[chunk] / a.cum_w[a.chunks - 1];   // a table of weights (the default for four tickets, or EMAT_TICKET_WEIGHTS="w1,w2,...")
    else if (a.taper) { const int64_t nn = a.chunks, done_w = (int64_t)(chunk + 1) * (2 * nn - chunk), all_w = nn * (nn + 1); target = chunk + 1 == a.chunks ? pass_target : pass_target * done_w / all_w; }
    else target = pass_target * (chunk + 1) / a.chunks;
    done_at_start = gh->moves_done - (int64_t)(uint32_t)((uint32_t)gh->moves_done - gh->pad0);
  } else done_at_start = gh->moves_done;
  const uint32_t area = a.lds_slab_bytes;
#if defined(EMAT_PROFILE_PHASES) || defined(EMAT_COUNT_CALLS)
  if (lane == 0) s_fn_count_me = (gh->heap_top - gh->heap_begin) >= g_fn_min_list_bytes ? 1 : 0;
#endif
  const bool can_stage = tables_staged && area != 0 && gh->off_nodes == (uint32_t)sizeof(SlabHeader);
  if (lane == 0 && chunk == 0) { st_ticks(0); a.part_ticks[a.num_parts + part] = (int64_t)wall_clock64(); }   // duration, start (emat_debug_part_ticks)
  // Up to two legs: a part whose USED state fits the staging area but whose heap capacity does not is staged whole with
  // an LDS-local heap limit; should its lists outgrow that, it is written back and finishes with its heap in HBM.
  bool allow_whole = true;
  bool plain_hbm_writes = false;   // some leg of this ticket changed the part's state in HBM with ordinary (cached) stores
  for (int leg = 0; leg < 2; ++leg) {
    __syncthreads();
    const uint32_t hbm_heap_end = gh->heap_end;
    uint32_t lds_heap_end = 0;
    if (can_stage && allow_whole) {
      // the list heap keeps k_lds_heap_room bytes to grow into (or its whole capacity, if smaller); what is left of the
      // area is the moves' LDS scratch arena -- without one, every temporary of an SPR move is an HBM round trip
      const uint32_t want = (gh->heap_top + k_lds_heap_room + 15u) & ~15u;
      if (hbm_heap_end <= area) lds_heap_end = hbm_heap_end < want ? hbm_heap_end : want;
      else if (want <= area) lds_heap_end = want;
    }
    const bool staged = lds_heap_end != 0;
    if (!staged) plain_hbm_writes = true;   // this leg edits lists (or everything) in the HBM slab itself, with ordinary stores
    const bool prefix = can_stage && !staged && gh->heap_begin <= area;
    // stage the persistent state (header, nodes, cells, trace, list heap) -- or, for a part too large for that, its
    // fixed-size prefix up to the list heap; scratch always stays in HBM
    const uint32_t staged_bytes = staged ? gh->heap_top : (prefix ? gh->heap_begin : 0u);
    if (staged_bytes) { wave_copy16(emat_lds_hdr, gslab, (uint32_t)sizeof(SlabHeader), lane); wave_copy16(lds_slab, gslab + sizeof(SlabHeader), staged_bytes - (uint32_t)sizeof(SlabHeader), lane); }
    // a leg that works on the HBM slab itself keeps a copy of what it found (header, nodes, cells, trace, lists in use)
    uint8_t* const snap = (!staged && a.snaps != nullptr) ? a.snaps + a.slab_off[part] : nullptr;
    if (snap != nullptr) wave_copy16(snap, gslab, (gh->heap_top + 15u) & ~15u, lane);
    __syncthreads();
    SlabHeader* const lds_hdr = (SlabHeader*)emat_lds_hdr;
    SlabHeader* H = (staged || prefix) ? lds_hdr : gh;
    // The root part is one chain like any other, but its moves walk long runs of coalescent cells (deep branches span
    // hundreds of cells): compute-bound, the longest chain of its launch.  Let its wave win issue arbitration on its SIMD.
    // (So are the parts of the side classes -- the giants of a partition, a few dozen among thousands: at the reference's
    // rule every part does the same number of moves, theirs cost two or three times a small part's, and the pass waits for them.)
    const bool raise_prio = kSide || (gh->flags & k_flag_includes_run_root) != 0;
    uint64_t tick0 = 0;
    if (lane == 0) {
      // The context lives in LDS, not in private memory: it is touched by almost every instruction.
      if (staged) {
        dev_lds::Ctx& c = *(dev_lds::Ctx*)(emat_lds_ctx);
        init_ctx(c, gslab, gslab, a, lds_tables, lds_hdr);
        H->heap_end = lds_heap_end;
        // whatever the part leaves unused of the staging area (plus the optional extra arena) serves as the first-level
        // scratch arena of its moves; scratch that does not fit goes to the part's HBM scratch region as before
        const uint32_t used = (lds_heap_end + 15u) & ~15u;
        c.A = lds_slab + (used - (uint32_t)sizeof(SlabHeader)); c.a_end = area + a.lds_scratch_bytes - used;
      } else if (prefix) {
        // the prefix leaves the rest of the staging area free: the moves' first-level arena, as for a staged part (before
        // round 2's end these parts -- 40-60 nodes, the slowest chains of a pass -- ran every candidate scan through HBM)
        dev_mix::Ctx& c = *(dev_mix::Ctx*)(emat_lds_ctx);
        init_ctx(c, gslab, gslab, a, lds_tables, lds_hdr);
        const uint32_t used = (gh->heap_begin + 15u) & ~15u;
        c.A = lds_slab + (used - (uint32_t)sizeof(SlabHeader)); c.a_end = area + a.lds_scratch_bytes - used;
      } else {
        dev::Ctx& c = *(dev::Ctx*)(emat_lds_ctx);
        init_ctx(c, gslab, gslab, a, tables_staged ? lds_tables : nullptr);
        if (area + a.lds_scratch_bytes > (uint32_t)sizeof(SlabHeader)) { c.A = lds_slab; c.a_end = area + a.lds_scratch_bytes - (uint32_t)sizeof(SlabHeader); }   // nothing of the part is staged: the whole dynamic block is arena
      }
      // (a later ticket of a part whose earlier one had to stop does nothing: the host gives the part more room and the rest of its moves)
      const bool stopped_before = chunk > 0 && ld_status() != 0;
      ((dev::Ctx*)(emat_lds_ctx))->moves_left = (H->status == 0 && !stopped_before) ? target - (H->moves_done - done_at_start) : 0;   // the three Ctx types share one layout
      tick0 = wall_clock64();
      if (raise_prio) __builtin_amdgcn_s_setprio(3);
    }
    __syncthreads();
    // The chain: stretches of moves on lane 0; whenever a move parks itself for work the whole wave shares (the candidate
    // scan and study of an SPR move), all 64 lanes do that work and lane 0 picks the move up again.
    for (;;) {
      if (EMAT_CHAIN_ON_ALL_LANES || lane == 0) {
        if (staged) dev_lds::run_chain_loop(*(dev_lds::Ctx*)(emat_lds_ctx + EMAT_OPQ));
        else if (prefix) dev_mix::run_chain_loop(*(dev_mix::Ctx*)(emat_lds_ctx + EMAT_OPQ));
        else dev::run_chain_loop(*(dev::Ctx*)(emat_lds_ctx + EMAT_OPQ));
      }
      __syncthreads();
      const int svc = ((const dev::Ctx*)(emat_lds_ctx))->svc;
      if (svc == 0) break;
      if (svc == 2) dev::rng_fill(*(dev::Ctx*)(emat_lds_ctx), lane);      // (the three Ctx types share one layout)
      else if (staged) dev_lds::wave_scan_and_study(*(dev_lds::Ctx*)(emat_lds_ctx));
      else if (prefix) dev_mix::wave_scan_and_study(*(dev_mix::Ctx*)(emat_lds_ctx));
      else dev::wave_scan_and_study(*(dev::Ctx*)(emat_lds_ctx));
      __syncthreads();
    }
    if (lane == 0) {
      if (raise_prio) __builtin_amdgcn_s_setprio(0);
      const dev::Ctx& c = *(const dev::Ctx*)(emat_lds_ctx);
      H->rng_counter = c.rng_ctr; H->rng_spare = c.rng_spare; H->rng_has_spare = c.rng_has_spare ? 1u : 0u;
      H->alg_bytes += c.bytes; H->alg_write16 += (uint32_t)((c.bytes_w + 8) >> 4);
      const int64_t dt = (int64_t)(wall_clock64() - tick0);
      H->device_ticks += dt;
      st_ticks(ld_ticks() + dt);
      if (!(chunk > 0 && ld_status() != 0)) st_status(H->status);   // an idle ticket leaves the earlier ticket's verdict alone
      int again = 0;
      if (staged) {
        H->heap_end = hbm_heap_end;
        if (H->status == k_part_need_space && lds_heap_end < hbm_heap_end && H->heap_top <= hbm_heap_end) { H->status = 0; st_status(0); again = 1; }
        // A list outgrew the LDS-local heap limit INSIDE a move: the staged state is lost, but nothing of this leg has
        // reached the HBM copy of the slab yet (moves only write HBM scratch), and the chain is a deterministic function
        // of that copy.  Drop the LDS state and run the leg again from HBM with the heap at its full capacity there.
        else if (H->status == k_part_overflow && lds_heap_end < hbm_heap_end) { st_status(0); again = 2; }
        // The root part's coalescent grid outgrew its capacity (the root wandered further into the past than the room the
        // host left): the same argument -- the HBM copy still is the consistent state this launch found -- lets the
        // host re-materialise the part with more cells and run its moves again (first leg only: a second leg follows a
        // write-back).
        else if (H->status == k_part_cell_overflow && leg == 0) { st_status(k_part_need_cells); again = 3; }
        // Anything else that overflowed inside a move of a staged leg (the heap at its full capacity, the scratch region):
        // the HBM copy is the leg's starting state just the same -- more room from the host, and the moves again.
        else if (H->status == k_part_overflow) { st_status(k_part_need_space); again = 3; }
      } else if (snap != nullptr && (H->status == k_part_overflow || H->status == k_part_cell_overflow)) {
        // the leg ran on the HBM slab: its starting state goes back in from the copy taken above
        st_status(H->status == k_part_cell_overflow ? k_part_need_cells : k_part_need_space); again = 4;
      }
      *lds_flag = again;
    }
    __syncthreads();
    const int again_all = *lds_flag;
    if (again_all < 2) {
      const uint32_t out_bytes = staged ? ((const SlabHeader*)emat_lds_hdr)->heap_top : (prefix ? staged_bytes : 0u);
      if (out_bytes && a.chunks > 1 && !single) { wave_copy8_through(gslab, emat_lds_hdr, (uint32_t)sizeof(SlabHeader), lane); wave_copy8_through(gslab + sizeof(SlabHeader), lds_slab, out_bytes - (uint32_t)sizeof(SlabHeader), lane); }
      else if (out_bytes) { wave_copy16(gslab, emat_lds_hdr, (uint32_t)sizeof(SlabHeader), lane); wave_copy16(gslab + sizeof(SlabHeader), lds_slab, out_bytes - (uint32_t)sizeof(SlabHeader), lane); }
    }
    if (again_all == 4) { wave_copy16(gslab, snap, (((const SlabHeader*)snap)->heap_top + 15u) & ~15u, lane); break; }
    if (again_all == 0 || again_all == 3) break;
    allow_whole = false;
  }
  if (lane == 0 && chunk < k_ticket_log) a.part_ticks[(size_t)(3 + 2 * chunk) * a.num_parts + part] = (int64_t)wall_clock64();   // ... to here
  if (a.chunks > 1 && !single) {   // hand the part to its next ticket: the slab is in HBM again
    __syncthreads();
    // An agent-scope release writes back every dirty line of this XCD's L2 -- among them the private-memory lines of the 500
    // other chains resident there -- once per ticket: half of a pass's HBM write traffic.  A ticket whose
    // part was staged has sent everything the next ticket reads through write-through stores (the slab image, status, ticks):
    // it only waits for those to be acknowledged.  One that edited lists in HBM directly keeps the full release.
    if (lane == 0) {
      if (plain_hbm_writes || a.full_release) __hip_atomic_store(&a.chunk_done[part], chunk + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_waitcnt(0); __hip_atomic_store(&a.chunk_done[part], (chunk + 1) | 0x100 | (xcc_id() << 12), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
  }
}
__global__ void __launch_bounds__(k_wave) EMAT_OCCUPANCY EMAT_NOTAIL k_run_moves(KernelArgs a) { run_moves_body<false>(a); }
// Same body under another name for the side launches of the size classes (the "giants", §4 of DESIGN.md), so that
// profiles keep the statistics of the main launch -- the one bench.py's roofline is about -- apart.
__global__ void __launch_bounds__(k_wave) EMAT_OCCUPANCY EMAT_NOTAIL k_run_moves_side(KernelArgs a) { run_moves_body<true>(a); }
// A side class holds a few workgroups of tens of KB of LDS each -- the root part's wants most of a CU -- and the main class 30 000 of
// ten KB: once the main class is on the device a large workgroup only fits when a CU happens to drain, which is at the tail of the
// pass.  Launch order alone does not settle who is placed first (two streams, two queues: a race the side classes lost in one process
// in three, rocprofv3 --kernel-trace: root part's kernel 27 ms instead of 9 in the first pass); and a side class that starts at the
// tail of a pass ends in the next one, starts again at ITS tail, and so on: every later pass of that process carries the lag, which the
// final synchronisation then waits for (0.9 ms per pass over ten passes -- the 400-against-418 M moves/s coin flip of rounds 3 and 4).
// So on a synchronised pass the engine's stream holds the main class back until every side workgroup has reported itself resident:
// microseconds on an idle device, and bounded (2 ms of the 100 MHz counter) in case something else holds the CUs.
__global__ void k_wait_side_start(const int32_t* started, int32_t expected) {
  if (threadIdx.x != 0) return;
  const uint64_t w0 = wall_clock64();
  while (__hip_atomic_load(started, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected && wall_clock64() - w0 < 200000ull) __builtin_amdgcn_s_sleep(8);
}

// ---- sufficient statistics of the global moves (calc_Ttwiddle_beta_a phylo_tree_calc.cpp:288-369, calc_num_muts_beta_ab
//      :599-610, calc_num_muts :577-585), one part per workgroup ---------------------------------------------------------
// n^beta_a(node) = nu-weighted number of sites of partition beta present in state a at the node.  Its change across a
// branch comes from the branch's missation intervals (two gathers into a prefix table over the reference sequence
// instead of the reference's site-by-site loop), from_states and mutations: lane-parallel over the nodes.  The
// values themselves follow from a pre-order accumulation (lane 0), and the statistics are sums over the non-root
// branches, again lane-parallel, reduced across lanes in a fixed order so that the result is reproducible.
constexpr int k_max_stats_partitions = 4;
constexpr int k_stats_row = k_max_stats_partitions * (4 + 16) + 1;   // Ttwiddle[P][4], num_muts[P][4][4], num_muts
__global__ void __launch_bounds__(k_wave) k_global_stats(KernelArgs a) {
  __shared__ double sh_T[k_wave][k_max_stats_partitions * 4];
  __shared__ int sh_M[k_wave][k_max_stats_partitions * 16];
  const int lane = threadIdx.x;
  const int part = blockIdx.x;
  uint8_t* slab = a.slabs + a.slab_off[part];
  dev::Ctx c;
  init_ctx(c, slab, slab, a, nullptr);
  const int P = a.evo.num_partitions, W = 4 * P;
  const int n = c.H->n_nodes, root = c.H->root;
  double* D = (double*)(slab + c.H->scratch_begin);   // [n][W]: per-branch change, then the value at the node
  // phase A: change of n^beta_a across every branch
  for (int i = lane; i < n; i += k_wave) {
    double* d = D + (size_t)i * W;
    for (int k = 0; k < W; ++k) d[k] = 0.0;
    const IvRec* iv = dev::miss_of(c, i);
    for (int j = 0; j < (int)c.N[i].miss.cnt; ++j) {
      const double* hi = a.cum_nu + (size_t)iv[j].end * W; const double* lo = a.cum_nu + (size_t)iv[j].start * W;
      for (int k = 0; k < W; ++k) d[k] -= hi[k] - lo[k];
    }
    const FsRec* fs = dev::mfs_of(c, i);
    for (int j = 0; j < (int)c.N[i].mfs.cnt; ++j) { const int l = fs[j].site, b = c.part[l]; d[4 * b + c.ref[l]] += c.nu[l]; d[4 * b + fs[j].state] -= c.nu[l]; }
    const MutRec* m = dev::muts_of(c, i);
    for (int j = 0; j < dev::nmuts(c, i); ++j) { const int l = m[j].site, b = c.part[l]; d[4 * b + m[j].from] -= c.nu[l]; d[4 * b + m[j].to] += c.nu[l]; }
  }
  __syncthreads();
  // phase B: pre-order accumulation from the reference sequence's counts (the last row of the prefix table)
  if (lane == 0) {
    const double* ref_row = a.cum_nu + (size_t)c.L * W;
    for (int k = 0; k < W; ++k) D[(size_t)root * W + k] += ref_row[k];
    int cur = root;
    while (cur != dev::k_no_node) {
      if (!dev::is_tip(c, cur)) {
        for (int kk = 0; kk < 2; ++kk) {
          const int ch = kk == 0 ? c.N[cur].child0 : c.N[cur].child1;
          for (int k = 0; k < W; ++k) D[(size_t)ch * W + k] += D[(size_t)cur * W + k];
        }
        cur = c.N[cur].child0;
      } else {
        int prev = cur; cur = c.N[cur].parent;
        while (cur != dev::k_no_node && c.N[cur].child1 == prev) { prev = cur; cur = c.N[cur].parent; }
        if (cur != dev::k_no_node) cur = c.N[cur].child1;
      }
    }
  }
  __syncthreads();
  // phase C: sums over the non-root branches
  for (int k = 0; k < W; ++k) sh_T[lane][k] = 0.0;
  for (int k = 0; k < 4 * W; ++k) sh_M[lane][k] = 0;
  for (int i = lane; i < n; i += k_wave) {
    if (i == root) continue;   // "mutations" above the root are deltas from the reference sequence
    const double t_P = c.N[c.N[i].parent].t, len = c.N[i].t - t_P;
    const double* d = D + (size_t)i * W;
    for (int k = 0; k < W; ++k) sh_T[lane][k] += d[k] * len;
    const MutRec* m = dev::muts_of(c, i);
    for (int j = dev::nmuts(c, i) - 1; j >= 0; --j) {
      const int l = m[j].site, b = c.part[l];
      sh_T[lane][4 * b + m[j].to] -= c.nu[l] * (m[j].t - t_P);
      sh_T[lane][4 * b + m[j].from] += c.nu[l] * (m[j].t - t_P);
      sh_M[lane][16 * b + 4 * m[j].from + m[j].to] += 1;
    }
  }
  __syncthreads();
  double* out = a.stats_out + (size_t)part * k_stats_row;
  for (int k = lane; k < W; k += k_wave) { double sum = 0.0; for (int r = 0; r < k_wave; ++r) sum += sh_T[r][k]; out[k] = sum; }
  for (int k = lane; k < 4 * W; k += k_wave) { long long sum = 0; for (int r = 0; r < k_wave; ++r) sum += sh_M[r][k]; out[k_max_stats_partitions * 4 + k] = (double)sum; }
  if (lane == 0) {
    long long nm = 0;
    for (int k = 0; k < 4 * W; ++k) for (int r = 0; r < k_wave; ++r) nm += sh_M[r][k];
    out[k_stats_row - 1] = (double)nm;
  }
}

// ---- whole-part derived quantities (Subrun::recalc_derived_quantities, subrun.cpp:17-26) ---------------------
// Lanes stride over the part's nodes: branch-local work (delta lambda across the branch, missing-site count,
// branch log-G, log N(t)) is embarrassingly parallel; only the pre-order accumulation of lambda_i /
// n_missing down the tree is a dependent chain, walked by lane 0 without a stack.
// kCheck = true is the reference's check_derived_quantities (subrun.cpp:28-56; forced in release builds by --v0-paranoid,
// run.h:220-224): the same recomputation, but into the part's scratch region, compared with what the moves maintained
// incrementally, and reported per part in check_out[part][4] = {max |d lambda_i| / L, |d log_G|, |d log prior|, number of
// nodes whose missing-site count differs}; the slab itself is left as it is.
template <bool kCheck> __device__ __forceinline__ void recalc_derived_body(const KernelArgs& a, double* check_out) {
  __shared__ __attribute__((aligned(16))) double lds_tables[k_lds_tables_bytes / 8];
  const int lane = threadIdx.x;
  const int part = blockIdx.x;
  uint8_t* slab = a.slabs + a.slab_off[part];
  const double* tables = stage_tables(a, lds_tables, lane);
  __syncthreads();
  dev::Ctx c;
  init_ctx(c, slab, slab, a, tables);
  const int n = c.H->n_nodes, root = c.H->root;
  // where the recomputed per-node values go: the node records themselves, or (check) two arrays in the scratch region
  double* lam_of = kCheck ? (double*)(slab + c.H->scratch_begin) : nullptr;
  int32_t* nmiss_of = kCheck ? (int32_t*)(slab + c.H->scratch_begin + (size_t)n * 8u) : nullptr;
  auto LAM = [&](int i) -> double& { return kCheck ? lam_of[i] : c.N[i].lambda; };
  auto NMISS = [&](int i) -> int32_t& { return kCheck ? nmiss_of[i] : c.N[i].n_missing; };
  // phase A: per-branch deltas
  for (int i = lane; i < n; i += k_wave) {
    LAM(i) = dev::delta_lambda_across_branch(c, i);
    NMISS(i) = dev::iv_num_sites(dev::miss_of(c, i), (int)c.N[i].miss.cnt);
  }
  __syncthreads();
  // phase B: pre-order prefix (phylo_tree_calc.cpp:420-436, :67-76)
  if (lane == 0) {
    int cur = root;
    LAM(root) = c.cumQ[c.L] + LAM(root);
    while (cur != dev::k_no_node) {
      if (!dev::is_tip(c, cur)) {
        for (int k = 0; k < 2; ++k) {
          int ch = k == 0 ? c.N[cur].child0 : c.N[cur].child1;
          LAM(ch) = LAM(cur) + LAM(ch);
          NMISS(ch) = NMISS(cur) + NMISS(ch);
        }
        cur = c.N[cur].child0;
      } else {
        // climb until we arrive from a first child, then step to its sibling
        int prev = cur; cur = c.N[cur].parent;
        while (cur != dev::k_no_node && c.N[cur].child1 == prev) { prev = cur; cur = c.N[cur].parent; }
        if (cur != dev::k_no_node) cur = c.N[cur].child1;
      }
    }
  }
  __syncthreads();
  // phase C: log G and the coalescent partial prior
  double acc_G = 0.0, acc_prior = 0.0;
  for (int i = lane; i < n; i += k_wave) {
    if (i != root) acc_G += dev::branch_log_G(c, c.N[c.N[i].parent].t, c.N[i].t, LAM(i), dev::muts_of(c, i), dev::nmuts(c, i));
    if (!dev::is_tip(c, i)) acc_prior -= log(dev::pop_at_time(*c.pop, c.N[i].t));
  }
  {
    // (lanes take different cells here: the shared arrays are read with ordinary loads, not through the scalar cache)
    const int cap = c.H->cell_cap, first = c.H->cell_first;
    const double* base = (const double*)(slab + c.H->off_cells);
    const double* kbar_p = base; const double* ktw_p = base + cap;
    for (int w = lane; w < c.H->n_cells; w += k_wave) {   // very_scalable_coalescent.cpp:355-386
      double na, tsop, ktw;
      if (c.includes_run_root) { na = (double)((const int32_t*)(base + 5 * cap))[w]; tsop = base[4 * cap + w]; ktw = base[2 * cap + w]; }
      else { na = (double)a.shared.num_active_parts[first + w]; tsop = a.shared.ts_over_pop[first + w]; ktw = a.shared.k_twiddle_bar[first + w]; }
      acc_prior -= tsop * (+0.5 * (kbar_p[w] * kbar_p[w]) * na - (ktw_p[w] * na - ktw + 0.5) * kbar_p[w]);
    }
  }
  acc_G = wave_sum(acc_G); acc_prior = wave_sum(acc_prior);
  double dev_lambda = 0.0; int bad_missing = 0;
  if (kCheck) {
    for (int i = lane; i < n; i += k_wave) {
      const double dl = fabs(c.N[i].lambda - lam_of[i]) / (double)c.L;
      if (!(dl <= dev_lambda)) dev_lambda = dl;                     // (a NaN on either side shows as NaN)
      if (c.N[i].n_missing != nmiss_of[i]) ++bad_missing;
    }
    for (int off = 32; off > 0; off >>= 1) {
      const double o = __shfl_down(dev_lambda, off, k_wave); if (!(o <= dev_lambda)) dev_lambda = o;
      bad_missing += __shfl_down(bad_missing, off, k_wave);
    }
  }
  if (lane == 0) {
    double lg = acc_G;
    if (c.includes_run_root) lg = dev::calc_log_root_prior(c, a.ref_freqs, a.evo.num_partitions) + acc_G;
    if (kCheck) {
      double* o = check_out + 4 * (size_t)part;
      o[0] = dev_lambda; o[1] = fabs(c.H->log_G - lg); o[2] = fabs(c.H->log_aug_prior - acc_prior); o[3] = (double)bad_missing;
      if (c.H->log_G == lg) o[1] = 0.0;                             // (-inf on both sides is agreement)
    } else {
      c.H->log_G = lg;
      c.H->log_aug_prior = acc_prior;
    }
  }
}
__global__ void __launch_bounds__(k_wave) k_recalc_derived(KernelArgs a) { recalc_derived_body<false>(a, nullptr); }
__global__ void __launch_bounds__(k_wave) k_check_derived(KernelArgs a, double* check_out) { recalc_derived_body<true>(a, check_out); }
// test hook (emat_debug_tree_query): the moves' own find_MRCA_of / descends_from on one part's slab, query by query
__global__ void __launch_bounds__(k_wave) k_debug_tree_query(KernelArgs a, int part, int op, const int32_t* qa, const int32_t* qb, int32_t* out, int n) {
  if (threadIdx.x != 0) return;
  uint8_t* slab = a.slabs + a.slab_off[part];
  dev::Ctx c;
  init_ctx(c, slab, slab, a, nullptr);
  for (int i = 0; i < n; ++i) out[i] = op == 0 ? dev::find_MRCA_of(c, qa[i], qb[i]) : (dev::descends_from(c, qa[i], qb[i]) ? 1 : 0);
}

// test hook (emat_debug_graft): the moves' own graft analysis, peel, re-attachment, proposal and apply on one part's slab, on
// lane 0 as inside a chain, with everything the analysis found written out as numbers (layout: emat_backend.h).
struct GraftOut { double* p; int cap; int n; };
__device__ inline void go_put(GraftOut& o, double v) { if (o.n < o.cap) o.p[o.n] = v; ++o.n; }
__device__ inline void go_graft(GraftOut& o, const dev::Graft& g) {
  go_put(o, (double)g.nbi); go_put(o, g.delta_log_G); go_put(o, g.log_alpha_mut); go_put(o, (double)g.X); go_put(o, (double)g.S); go_put(o, g.t_P);
  for (int i = 0; i < g.nbi; ++i) {
    const dev::BranchInfo& b = g.bi[i];
    go_put(o, (double)b.A); go_put(o, (double)b.B); go_put(o, b.is_open ? 1.0 : 0.0); go_put(o, b.T_to_X); go_put(o, b.pl_A); go_put(o, b.pl_X);
    go_put(o, (double)b.warm.n); for (int k = 0; k < b.warm.n; ++k) { go_put(o, (double)b.warm.p[k].start); go_put(o, (double)b.warm.p[k].end); }
    go_put(o, (double)b.hot.n); for (int k = 0; k < b.hot.n; ++k) { go_put(o, (double)b.hot.p[k].start); go_put(o, (double)b.hot.p[k].end); }
    go_put(o, (double)b.hot_muts.n); for (int k = 0; k < b.hot_muts.n; ++k) { const MutRec& m = b.hot_muts.p[k]; go_put(o, (double)m.site); go_put(o, (double)m.from); go_put(o, (double)m.to); go_put(o, m.t); }
    go_put(o, (double)b.hot_deltas.n); for (int k = 0; k < b.hot_deltas.n; ++k) { const dev::SdRec& d = b.hot_deltas.p[k]; go_put(o, (double)d.site); go_put(o, (double)d.from); go_put(o, (double)d.to); }
  }
}
__global__ void __launch_bounds__(k_wave) EMAT_OCCUPANCY k_debug_graft(KernelArgs a, int part, int X, double mu_proposal, int mode, int new_S, double new_t_P, double* out, int out_cap, int32_t* out_len) {
  __shared__ __attribute__((aligned(16))) double lds_tables[k_lds_tables_bytes / 8];
  const int lane = threadIdx.x;
  uint8_t* slab = a.slabs + a.slab_off[part];
  const double* tables = stage_tables(a, lds_tables, lane);
  __syncthreads();
  if (lane != 0) return;
  dev::Ctx c;
  init_ctx(c, slab, slab, a, tables);
  dev::sc_reset(c);
  c.mu_prop = mu_proposal;
  GraftOut o; o.p = out; o.cap = out_cap; o.n = 0;
  go_put(o, 0.0);                                   // [0]: the part's status afterwards
  go_put(o, mode == 3 ? 2.0 : 1.0);                 // [1]: grafts written out
  dev::Graft* g = (dev::Graft*)dev::sc_alloc(c, (uint32_t)(2 * sizeof(dev::Graft)));
  if (!c.failed) {
    dev::analyze_graft(c, X, g[0]);                 // Spr_move::analyze_graft (spr_move.cpp:9-36)
    if (!c.failed) go_graft(o, g[0]);
    if (mode >= 1 && !c.failed) {
      dev::peel_graft(c, g[0]);                     // Spr_move::peel_graft (:38-62)
      if (!c.failed) {
        go_put(o, (double)dev::count_min_mutations(c, g[0]));                      // count_min_mutations (spr_move.cpp:64-71)
        { int closed = 0; if (g[0].rooty) closed = g[0].bi[dev::k_SPX].hot_muts.n; else for (int i = 0; i < g[0].nbi; ++i) if (!g[0].bi[i].is_open) closed += g[0].bi[i].hot_muts.n;
          go_put(o, (double)closed); }                                             // count_closed_mutations (:73-89; not needed by a move)
        const dev::SVec<dev::SdRec> d = dev::summarize_closed_mutations(c, g[0], 0);      // summarize_closed_mutations (:1126-1156)
        go_put(o, (double)d.n); for (int k = 0; k < d.n; ++k) { go_put(o, (double)d.p[k].site); go_put(o, (double)d.p[k].from); go_put(o, (double)d.p[k].to); }
      }
    }
    if (mode == 2 && !c.failed) dev::apply_graft(c, g[0]);                         // Spr_move::apply_graft (:64-89)
    if (mode == 3 && !c.failed) {
      // the middle of an SPR move without its candidate study (subrun.cpp:580-640): re-attach at (new_S, new_t_P), propose the new
      // graft's mutations from the part's random stream, apply them, and account for the change of log G as an accepted move does
      dev::spr_move_topology(c, X, new_S, new_t_P);                                // Spr_move::move (spr_move.cpp:1071-1099)
      if (!c.failed) dev::propose_new_graft(c, X, g[1]);
      if (!c.failed) { go_graft(o, g[1]); dev::apply_graft(c, g[1]); }
      if (!c.failed) { c.H->log_G -= g[0].delta_log_G; c.H->log_G += g[1].delta_log_G; }
    }
  }
  c.H->rng_counter = c.rng_ctr; c.H->rng_spare = c.rng_spare; c.H->rng_has_spare = c.rng_has_spare ? 1u : 0u;
  out[0] = (double)(c.failed ? (c.H->status != 0 ? c.H->status : k_part_internal) : 0);
  *out_len = o.n;
}

// test hook (emat_debug_sample_history): the proposal's JC69 history sampler, history by history, as the reference's own statistical
// test drives it (tests/spr_move_tests.cpp:1795-1961): for history i the path ends at (branch[i], t_end[i]); the deltas are where
// `start_seq` differs from the tree's sequence there; sample_mutational_history + adjust_mutational_history; every mutation written out.
__global__ void __launch_bounds__(k_wave) EMAT_OCCUPANCY k_debug_sample_history(KernelArgs a, int part, int n, const int32_t* branch, const double* t_end, const uint8_t* start_seq,
                                                                                double T, double mu, int32_t* counts, double* muts, int muts_cap, int32_t* status) {
  __shared__ __attribute__((aligned(16))) double lds_tables[k_lds_tables_bytes / 8];
  const int lane = threadIdx.x;
  uint8_t* slab = a.slabs + a.slab_off[part];
  const double* tables = stage_tables(a, lds_tables, lane);
  __syncthreads();
  if (lane != 0) return;
  dev::Ctx c;
  init_ctx(c, slab, slab, a, tables);
  int written = 0;
  for (int i = 0; i < n && !c.failed; ++i) {
    dev::sc_reset(c);
    dev::SVec<dev::SdRec> deltas = dev::sc_vec<dev::SdRec>(c, c.L + 1);
    for (int l = 0; l < c.L && !c.failed; ++l) { const int e = dev::calc_site_state_at(c, branch[i], t_end[i], l); if (e != (int)start_seq[l]) dev::sd_push_back(c, deltas, l, (int)start_seq[l], e); }
    if (c.failed) break;
    dev::SVec<MutRec> h = dev::sample_mutational_history(c, c.L, T, mu, deltas);
    if (c.failed) break;
    dev::adjust_mutational_history(c, h, deltas, branch[i], t_end[i]);
    counts[i] = h.n;
    for (int k = 0; k < h.n; ++k, ++written) if (written < muts_cap) { double* o = muts + 4 * (size_t)written; o[0] = (double)h.p[k].site; o[1] = (double)h.p[k].from; o[2] = (double)h.p[k].to; o[3] = h.p[k].t; }
  }
  c.H->rng_counter = c.rng_ctr; c.H->rng_spare = c.rng_spare; c.H->rng_has_spare = c.rng_has_spare ? 1u : 0u;
  status[0] = c.failed ? (c.H->status != 0 ? c.H->status : k_part_internal) : 0;
  status[1] = written;
}

// test hook (emat_debug_edit): one tree-editing session on node X of one part's slab, step by step as the caller lists them
// (Tree_editing_session, reference tree_editing.cpp:7-302): 0 slide_P_along_branch(t), 1 hop_up, 2 flip, 3 hop_down(node); then end().
__global__ void __launch_bounds__(k_wave) EMAT_OCCUPANCY k_debug_edit(KernelArgs a, int part, int X, int n_ops, const int32_t* op_kind, const int32_t* op_node, const double* op_t, int32_t* status) {
  __shared__ __attribute__((aligned(16))) double lds_tables[k_lds_tables_bytes / 8];
  const int lane = threadIdx.x;
  uint8_t* slab = a.slabs + a.slab_off[part];
  const double* tables = stage_tables(a, lds_tables, lane);
  __syncthreads();
  if (lane != 0) return;
  dev::Ctx c;
  init_ctx(c, slab, slab, a, tables);
  dev::sc_reset(c);
  dev::Edit e;
  int cap = 8;
  for (int i = 0; i < c.H->n_nodes; ++i) cap += dev::nmuts(c, i);
  dev::edit_begin(c, e, X, cap);
  for (int i = 0; i < n_ops && !c.failed; ++i) {
    if (op_kind[i] == 0) dev::edit_slide_P_along_branch(c, e, op_t[i]);
    else if (op_kind[i] == 1) dev::edit_do_hop_up(c, X);
    else if (op_kind[i] == 2) dev::edit_flip(c, e);
    else dev::edit_hop_down(c, e, op_node[i]);
  }
  dev::edit_end(c, e);
  status[0] = c.failed ? (c.H->status != 0 ? c.H->status : k_part_internal) : 0;
}

// ---- compact copies of what the host reads most often, so that it does not have to download the slabs for them ----------
// Every part's 256-byte header (status, counters, log_G, log prior, RNG position) into one dense array.
__global__ void __launch_bounds__(k_wave) k_gather_headers(KernelArgs a, uint8_t* out) {
  const int part = blockIdx.x, lane = threadIdx.x;
  const uint4* src = (const uint4*)(a.slabs + a.slab_off[part]);
  uint4* dst = (uint4*)(out + (size_t)part * sizeof(SlabHeader));
  for (int i = lane; i < (int)(sizeof(SlabHeader) / 16); i += k_wave) dst[i] = src[i];
}

// calc_num_muts_l (phylo_tree_calc.cpp:612-622): mutations per site over the non-root branches of every part (integer
// atomics: the result does not depend on the order).
__global__ void __launch_bounds__(k_wave) k_num_muts_l(KernelArgs a, int32_t* out) {
  const int part = blockIdx.x, lane = threadIdx.x;
  uint8_t* slab = a.slabs + a.slab_off[part];
  dev::Ctx c;
  init_ctx(c, slab, slab, a, nullptr);
  const int n = c.H->n_nodes, root = c.H->root;
  for (int i = lane; i < n; i += k_wave) {
    if (i == root) continue;   // "mutations" above a part's root are deltas from the reference sequence
    const MutRec* m = dev::muts_of(c, i);
    for (int j = 0; j < dev::nmuts(c, i); ++j) atomicAdd(&out[m[j].site], 1);
  }
}

// ---- calc_Ttwiddle_l (phylo_tree_calc.cpp:176-222) over a partitioned tree ----------------------------------------------
// The reference starts every site at q_ref T_total and corrects it per mutation / missation with the branch length BELOW
// that point -- a quantity that crosses part boundaries.  A part gets, for every tip that is the cut node of a part below
// it, the total branch length hanging there (`ext`; the run driver computes it from k_part_lengths' sums and the tree of
// parts); with that its nodes' "length below" follow from one post-order pass, and the corrections are local again:
//   S[site]   += (q_to - q_from) T_below(mutation)                      and  (q_ref - q_from) T_below(branch) per from_state
//   D[start]  += T_below(branch),  D[end] -= T_below(branch)            per missing interval (the host's prefix sum over D
//                                                                       is R[l], and Ttwiddle_l = q_ref (T_total - R) + S)
// A part's root is skipped unless it is the run's root: a cut node's own branch belongs to the part above it.
__global__ void __launch_bounds__(k_wave) k_part_lengths(KernelArgs a, double* out) {
  const int part = blockIdx.x, lane = threadIdx.x;
  const uint8_t* slab = a.slabs + a.slab_off[part];
  const SlabHeader* H = (const SlabHeader*)slab; const NodeRec* N = (const NodeRec*)(slab + H->off_nodes);
  double acc = 0.0;
  for (int i = lane; i < H->n_nodes; i += k_wave) if (i != H->root) acc += N[i].t - N[N[i].parent].t;
  acc = wave_sum(acc);
  if (lane == 0) out[part] = acc;
}
__global__ void __launch_bounds__(k_wave) k_ttwiddle_l(KernelArgs a, const int32_t* ext_off, const int32_t* ext_node, const double* ext_val, double* S, double* D, double* t_root_out) {
  __shared__ __attribute__((aligned(16))) double lds_tables[k_lds_tables_bytes / 8];
  const int part = blockIdx.x, lane = threadIdx.x;
  uint8_t* slab = a.slabs + a.slab_off[part];
  const double* tables = stage_tables(a, lds_tables, lane);
  __syncthreads();
  dev::Ctx c;
  init_ctx(c, slab, slab, a, tables);
  const int n = c.H->n_nodes, root = c.H->root;
  double* Tw = (double*)(slab + c.H->scratch_begin);   // [n]: total branch length of the WHOLE tree below each node
  for (int i = lane; i < n; i += k_wave) Tw[i] = 0.0;
  __syncthreads();
  for (int k = ext_off[part] + lane; k < ext_off[part + 1]; k += k_wave) Tw[ext_node[k]] = ext_val[k];
  __syncthreads();
  if (lane == 0) {   // children before parents, without a stack
    int cur = root, from = dev::k_no_node;   // `from`: the node we arrived from (parent on the way down, a child on the way up)
    while (cur != dev::k_no_node) {
      const int par = c.N[cur].parent;
      int next;
      if (from == par && !dev::is_tip(c, cur)) next = c.N[cur].child0;
      else if (!dev::is_tip(c, cur) && from == c.N[cur].child0) next = c.N[cur].child1;
      else {   // a tip, or both children done: the node is complete
        if (cur != root) Tw[par] += (c.N[cur].t - c.N[par].t) + Tw[cur];
        next = cur == root ? dev::k_no_node : par;
      }
      from = cur; cur = next;
    }
    if (c.includes_run_root && t_root_out) *t_root_out = Tw[root];
  }
  __syncthreads();
  for (int i = lane; i < n; i += k_wave) {
    if (i == root && !c.includes_run_root) continue;
    const bool top = i == root;
    const double t_i = c.N[i].t, below = Tw[i];
    const MutRec* m = dev::muts_of(c, i);
    for (int j = 0; j < dev::nmuts(c, i); ++j) {
      const int l = m[j].site;
      const double T_below_mut = below + (top ? 0.0 : t_i - m[j].t);
      atomicAdd(&S[l], (dev::q_a(c, l, m[j].to) - dev::q_a(c, l, m[j].from)) * T_below_mut);
    }
    const double T_below_miss = below + (top ? 0.0 : t_i - c.N[c.N[i].parent].t);
    const IvRec* iv = dev::miss_of(c, i);
    for (int j = 0; j < (int)c.N[i].miss.cnt; ++j) { atomicAdd(&D[iv[j].start], T_below_miss); atomicAdd(&D[iv[j].end], -T_below_miss); }
    const FsRec* fs = dev::mfs_of(c, i);
    for (int j = 0; j < (int)c.N[i].mfs.cnt; ++j) { const int l = fs[j].site; atomicAdd(&S[l], (dev::q_a(c, l, c.ref[l]) - dev::q_a(c, l, fs[j].state)) * T_below_miss); }
  }
}

// ---- Scalable_coalescent_prior (scalable_coalescent.cpp:88-138, 163-187), one part per workgroup ----------------------
// The whole-tree grid prior is -sum_cells dt kbar (kbar - 1) / (2 Nbar) - sum_inner log N(t), where kbar_j, the mean number
// of lineages in cell j = [t_ref + j dt, t_ref + (j + 1) dt), j < 0, is 1 (the lineage above the root) plus, for every
// coalescence at t_i, the overlap of [t_i, t_ref] with the cell, minus the same for every tip.  That sum is additive over
// nodes, hence over parts: a cut node is the root of the part below it (counted there, as the coalescence it is) and a
// frozen tip of the part above it (which cannot tell it from a real tip and subtracts it), so the part below adds it once
// more.  With that, a part without the run's root contributes exactly nothing outside the cells its own nodes span, and
// the root part a constant -1 after them.  Lanes take cells; every lane walks the part's nodes in index order, so the
// partial sums are reproducible.  meta[part] = {first cell, number of cells, sum over inner nodes of -log N(t), status}.
__global__ void __launch_bounds__(k_wave) k_scalable_prior(KernelArgs a, double t_ref, double t_step, const uint64_t* out_off, const uint32_t* out_cap, double* out, double* meta) {
  __shared__ __attribute__((aligned(16))) double lds_tables[k_lds_tables_bytes / 8];
  const int part = blockIdx.x, lane = threadIdx.x;
  uint8_t* slab = a.slabs + a.slab_off[part];
  const double* tables = stage_tables(a, lds_tables, lane);
  __syncthreads();
  dev::Ctx c;
  init_ctx(c, slab, slab, a, tables);
  const int n = c.H->n_nodes, root = c.H->root;
  const bool root_part = c.includes_run_root;
  double tmin = dev::k_inf, tmax = -dev::k_inf, acc = 0.0;
  for (int i = lane; i < n; i += k_wave) {
    const double t = c.N[i].t;
    tmin = tmin < t ? tmin : t; tmax = tmax > t ? tmax : t;
    if (!dev::is_tip(c, i)) acc -= log(dev::pop_at_time(*c.pop, t));
  }
  for (int off = 32; off > 0; off >>= 1) { double o = __shfl_down(tmin, off, k_wave); tmin = tmin < o ? tmin : o; o = __shfl_down(tmax, off, k_wave); tmax = tmax > o ? tmax : o; }
  tmin = __shfl(tmin, 0, k_wave); tmax = __shfl(tmax, 0, k_wave);
  acc = wave_sum(acc);
  auto cell_of = [&](double t) { return (int)floor((t - t_ref) / t_step); };
  const int jlo = cell_of(tmin);
  int jhi = cell_of(tmax); if (jhi > -1) jhi = -1;   // cell 0 and later: no lineage of the tree lives after t_ref
  const int cnt = jhi >= jlo ? jhi - jlo + 1 : 0;
  double* row = out + out_off[part];
  const bool fits = (uint32_t)cnt <= out_cap[part];
  if (fits) {
    for (int j = jlo + lane; j <= jhi; j += k_wave) {
      const double ub_j = t_ref + (double)(j + 1) * t_step;
      double sum = 0.0;
      for (int i = 0; i < n; ++i) {
        const double t = c.N[i].t;
        const int cs = cell_of(t);
        if (j < cs) continue;
        double w = dev::is_tip(c, i) ? -1.0 : +1.0;
        if (i == root && !root_part) w += 1.0;
        sum += (j == cs) ? w * (ub_j - t) / t_step : w;   // add_interval: partial first cell, whole cells after it (:100-115)
      }
      row[j - jlo] = sum;
    }
  }
  if (lane == 0) { double* m = meta + (size_t)part * 4; m[0] = (double)jlo; m[1] = (double)cnt; m[2] = acc; m[3] = fits ? 0.0 : 1.0; }
}

// ---- test hook: the device's incomplete-gamma routines evaluated point by point (emat_debug_gamma) ----------------------
__global__ void __launch_bounds__(k_wave) EMAT_OCCUPANCY k_debug_gamma(const double* a, const double* x, double* out, int n, int mode) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = mode == 0 ? dev::gamma_q(a[i], x[i]) : dev::gamma_q_inv(a[i], x[i]);
}

// ---- test hooks: the device's population-model and interval-set routines on plain inputs (emat_debug_pop, _interval_op) --
__global__ void __launch_bounds__(k_wave) EMAT_OCCUPANCY k_debug_pop(PopTable pt, int op, const double* a, const double* b, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = op == 0 ? dev::pop_at_time(pt, a[i]) : dev::pop_integral(pt, a[i], b[i]);
}
__global__ void k_debug_interval_op(int op, const IvRec* A, int nA, const IvRec* B, int nB, int site, IvRec* out, int* n_out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (op == 1) *n_out = dev::iv_merge(out, A, nA, B, nB);
  else if (op == 2) *n_out = dev::iv_intersect(out, A, nA, B, nB);
  else if (op == 3) *n_out = dev::iv_subtract(out, A, nA, B, nB);
  else if (op == 5) *n_out = dev::iv_contains(A, nA, site) ? 1 : 0;
  else if (op == 6) *n_out = dev::iv_intersects(A, nA, B, nB) ? 1 : 0;
  else *n_out = -1;
}

}  // namespace emat
#include "emat_gtree_kernels.hpp"   // the whole tree in HBM: cutting it into part slabs and gathering the parts back
#include "emat_build.hpp"           // initial-tree construction: the graft loop as a kernel, the finishing passes on the host
namespace emat {

// =================================================================================================
// Host side
// =================================================================================================
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { set_error(std::string(#expr) + ": " + hipGetErrorString(_e)); return EMAT_ERR_HIP; } } while (0)

template <class T> struct DevBuf {
  T* p = nullptr; size_t n = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  // Room for `count` elements, contents undefined.  A buffer that must grow grows by a quarter more than asked: two dozen buffers are sized by the
  // number of parts, which creeps up from cycle to cycle within a stencil period (8 000 -> 14 000 at C4 with the part-size limit), and growing to the
  // exact need re-allocated a dozen of them EVERY cycle -- hipFree is 54 us a call (rocprofv3 --hip-trace of 40 whole cycles, round 6: 496 hipFree).
  hipError_t alloc(size_t count) {
    if (count > n) { if (p) (void)hipFree(p); p = nullptr; n = 0; const size_t want = count + (n_grown ? count / 4 : 0); hipError_t e = hipMalloc((void**)&p, std::max<size_t>(want, 1) * sizeof(T)); if (e != hipSuccess) return e; n = want; n_grown = true; }
    return hipSuccess;
  }
  bool n_grown = false;   // the first allocation is exact (most buffers are allocated once); every later one has room to spare
  hipError_t alloc_roomy(size_t count) { return count > n ? alloc(count + count / 4) : hipSuccess; }   // for buffers whose need creeps up from cycle to cycle: a quarter more than asked, so that most new maxima fit
  hipError_t upload(const T* src, size_t count) {
    hipError_t e = alloc(count); if (e != hipSuccess) return e;
    if (count) return hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice);
    return hipSuccess;
  }
};

// Host mirror of the device slabs in page-locked memory (grow-only): the per-cycle 64 MB H2D / D2H of C4 run at PCIe
// speed instead of being staged through a bounce buffer.  Not zero-filled: encode_slab initialises what it owns.
struct PinnedBytes {
  uint8_t* p = nullptr; size_t n = 0, cap = 0;
  ~PinnedBytes() { if (p) (void)hipHostFree(p); }
  uint8_t* data() { return p; }
  const uint8_t* data() const { return p; }
  size_t size() const { return n; }
  hipError_t resize(size_t bytes) {
    if (bytes > cap) {
      if (p) (void)hipHostFree(p);
      p = nullptr; cap = 0;
      const size_t want = bytes + bytes / 8;
      hipError_t e = hipHostMalloc((void**)&p, want, hipHostMallocDefault);
      if (e != hipSuccess) return e;
      cap = want;
    }
    n = bytes;
    return hipSuccess;
  }
};

struct PartHost {
  FlatTree tree;
  bool includes_run_root = false;
  HostRng rng;
  HostCoalPart coal;
  bool uploaded = false;
  // slab geometry
  uint64_t slab_off = 0;
  uint32_t slab_bytes = 0, scratch_bytes = 0;
  int32_t n_nodes = 0;             // (the tree itself may live only on the device: emat_tree_repartition)
  emat_part_stats stats{};
  std::vector<double> trace;       // the part's move trace so far (4 doubles per move), carried over re-materialisations
  double space_boost = 1.0;        // multiplier of the heap and scratch capacities; doubled when the part ran out of space
  int cell_boost = 1;              // multiplier of the room the root part's grid gets to grow into; quadrupled when it ran out
  // What the moves maintain INCREMENTALLY (lambda_i and the missing-site count of every node, log_G, the partial coalescent prior), kept across a
  // re-materialisation in the middle of a pass (finish_pass: some part ran out of slab space or grid cells).  The reference recomputes these when a Subrun
  // is made and never again (subrun.cpp:17-26); recomputing them half way gives the same numbers up to rounding -- and a chain that can tell: a node whose
  // d log G / dt cancels exactly (every site missing below one child or the other) takes the uniform branch of the bounded exponential with the maintained
  // lambda_i and the other branch with a recomputed one that is two units in the last place off (EMAT_FUZZ_SEED=6202, case 53, found in round 6).
  std::vector<double> kept_lambda; std::vector<int32_t> kept_n_missing; double kept_log_G = 0.0, kept_log_aug_prior = 0.0;
  bool derived_kept = false;       // set by finish_pass just before it re-materialises, consumed (and cleared) by materialize
};

// The whole tree in HBM (emat_gtree_kernels.hpp) with the host mirrors the partitioner and the coalescent builder need:
// topology and node times, a few MB per cycle instead of every list of every node.
struct GTreeHost {
  bool resident = false;            // emat_tree_upload was called
  bool parts_live = false;          // the slabs hold the parts of `partition` (between emat_tree_repartition and emat_tree_reassemble)
  int32_t n = 0;
  DevBuf<int32_t> parent, c0, c1, root; DevBuf<double> t; DevBuf<float> t_min, t_max; DevBuf<GList> muts, miss, mfs;
  DevBuf<MutRec> mut_heap; DevBuf<IvRec> iv_heap; DevBuf<FsRec> fs_heap; DevBuf<uint32_t> tops; DevBuf<int32_t> status;
  uint32_t used[3] = {0, 0, 0};     // records in use in the three heaps
  int32_t pool_regrows = 0, heap_regrows = 0, large_measures = 0;
  // current partition
  int32_t P = 0, root_part = -1, lo = 0, hi = 0;   // this process runs the parts [lo, hi) of the partition
  std::vector<int32_t> h_part_off, h_orig, h_kid0, h_kid1;   // host copies (h_orig / h_kid* only when the partition came from the host or was asked for)
  bool partition_on_device = false;   // made by emat_tree_partition
  DevBuf<int32_t> lidx;
  DevBuf<uint8_t> d_is_cut; DevBuf<int32_t> d_cut, d_sizes, d_part_status;   // emat_tree_partition's inputs and counts (kept: three allocations less per cycle)
  PinnedBytes pin_sizes, pin_measure;                                         // where its sizes + offsets, and the measures queued behind it, land
  hipEvent_t ev_sizes = nullptr, ev_measure = nullptr;
  bool measure_queued = false;      // k_gt_measure of the current partition was launched by emat_tree_partition, its results are on their way to pin_measure
  ~GTreeHost() { if (ev_sizes) (void)hipEventDestroy(ev_sizes); if (ev_measure) (void)hipEventDestroy(ev_measure); }
  DevBuf<GRootDelta> root_deltas_in;
  DevBuf<int32_t> part_off, orig, kid0, kid1, lpar;
  DevBuf<double> co_kbar, co_ktw, co_k_bar, co_k_tw, co_popsize, co_tsop; DevBuf<int32_t> co_num_active;   // the coalescent grid, when it is built on the device
  DevBuf<int32_t> measure_list;
  DevBuf<GMeasure> measure; DevBuf<MutRec> pool_muts; DevBuf<IvRec> pool_ivs; DevBuf<uint32_t> pool_tops;
  DevBuf<GPartDesc> desc; DevBuf<uint8_t> cells;
  DevBuf<GRootDelta> root_deltas; DevBuf<int32_t> n_root_deltas;
  // host mirrors
  std::vector<int32_t> h_parent, h_c0, h_c1; std::vector<double> h_t; std::vector<float> h_t_min, h_t_max; int32_t h_root = EMAT_NO_NODE;
  // Kept current by every reassemble: the children of every node, packed (pin_kids: n pairs), the root and its time -- what a cycle's
  // partitioner needs.  The arrays above follow only when somebody asks for them (gt_full_mirrors).
  DevBuf<int2> d_kids; PinnedBytes pin_kids; double h_root_t = 0.0; bool full_mirrors_stale = false;
  DevBuf<GClimb> climb; bool climb_current = false;   // GClimb records of every node (k_gt_pack_climb), remade before a measuring pass if lists or links were written since
  bool d_kids_current = false;      // d_kids holds every node's children (k_gt_gather_links only rewrites the inner nodes of the parts it sees)
  DevBuf<double> d_root_t; PinnedBytes pin_small;   // the root's time; { int32 root, int32 n_root_deltas, double t_root } on their way to the host
  // emat_tree_reassemble of a single process returns once topology and root are on the host: k_gt_gather may still be running.
  // Whoever touches the device-resident tree next (gt_require) waits for it and checks how it went (gt_finish_gather).
  bool gather_pending = false; std::vector<GRootDelta> gather_rd;
  emat_status gather_failed = EMAT_OK; std::string gather_failed_text;   // a deferred gather that failed: sticky until emat_tree_upload (gt_require)
  const int32_t* kids() const { return (const int32_t*)pin_kids.data(); }   // [2 v] = child0, [2 v + 1] = child1
  GTreeDev dev() {
    GTreeDev g{};
    g.n_nodes = n; g.climb = climb.p; g.root = root.p; g.parent = parent.p; g.c0 = c0.p; g.c1 = c1.p; g.t = t.p; g.t_min = t_min.p; g.t_max = t_max.p;
    g.muts = muts.p; g.miss = miss.p; g.mfs = mfs.p; g.mut_heap = mut_heap.p; g.iv_heap = iv_heap.p; g.fs_heap = fs_heap.p;
    g.mut_cap = (uint32_t)mut_heap.n; g.iv_cap = (uint32_t)iv_heap.n; g.fs_cap = (uint32_t)fs_heap.n; g.tops = tops.p;
    return g;
  }
  GPartition partition() { GPartition q{}; q.num_parts = P; q.root_part = root_part; q.part_off = part_off.p; q.orig = orig.p; q.kid0 = kid0.p; q.kid1 = kid1.p; q.lpar = lpar.p; return q; }
  GPools pools() { GPools q{}; q.muts = pool_muts.p; q.ivs = pool_ivs.p; q.mut_cap = (uint32_t)pool_muts.n; q.iv_cap = (uint32_t)pool_ivs.n; q.tops = pool_tops.p; return q; }
};

}  // namespace emat

using namespace emat;

struct emat_backend {
  emat_config cfg{};
  std::string last_error;
  int L = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  int num_cus = 0;
  int xcc_count = 0;                // XCDs that workgroups are dealt to round robin (probe_xcc_dealing); 0: not so, or unknown -> tickets always hand over with a full release
  // size classes: parts sorted by persistent size; class c stages up to class_lds[c] bytes per part and runs on its own stream
  static constexpr int k_max_classes = 3;
  hipStream_t class_stream[k_max_classes] = {};   // class 0 runs on `stream`; the others on streams shared by every handle of the device (side_stream)
  hipEvent_t ev_fork = nullptr, ev_join[k_max_classes] = {};
  int num_classes = 1; int class_begin[k_max_classes + 1] = {}; uint32_t class_lds[k_max_classes] = {};
  std::vector<int> class_of;        // per part
  std::vector<int64_t> expected_moves;   // per part: moves requested of it since its upload (apart from the part records: every launch adds to all of them)
  std::vector<int> cfg_class_pct{60};                // EMAT_LDS_CLASSES (tuning knob): percentiles of persistent size that close each class; the last
                                                     // class always extends to the largest part (its staging area is still that percentile's size)
  uint32_t cfg_lds_max = 96 * 1024;                  // EMAT_LDS_MAX (tuning knob): largest staging area; larger parts run out of HBM
  bool order_valid = false;         // d_order holds the current parts, largest first
  std::vector<int32_t> h_order;     // host copy of d_order
  bool last_launch_uniform = false; // the last launch ran the same number of moves on every part (its durations are comparable)
  bool cfg_order_by_time = false;   // option "order_by_time" (tuning knob): re-sort the launch order by measured durations at every synchronisation
  std::string cfg_ticket_weights;   // option "ticket_weights": "w1,w2,..." the tickets' ratio, as many numbers as tickets
  int cfg_build_blocks = 0;         // option "build_blocks": workgroups of the initial-tree builder's launch (0 = by tree size)
  bool cfg_debug_fail_gather = false;   // option "debug_fail_gather" (testing aid): the next deferred gather of the device-resident tree reports k_gt_inconsistent
  bool cfg_tree_tight = false;      // option "tree_tight" (testing aid): the device-resident tree gets no spare room, so that the growth paths run
  unsigned cfg_fn_min_lists = 0;    // option "fn_min_lists" (profiling builds): function timers count only parts whose lists take at least this many bytes
  bool cfg_phase_extra = false;     // option "phase_extra" (profiling builds): emat_debug_phase_ticks returns the scan and arena counters
                                    // (measured at C4: 292 vs 296 M moves/s -- with two parts per slot the slot that ran the longest part
                                    // still takes one more; off by default)
  bool pass_pending = false;        // a launch has not been checked for stopped parts yet (finish_pass)
  bool sides_in_flight = false;     // side-class launches that the engine's own stream has not been made to wait for yet (join_side_classes)
  bool sides_must_fork = true;      // the side streams have not seen what the engine's stream did since the last pass was checked
  emat_status fatal_status = EMAT_OK;   // a part stopped INSIDE a move: its tree is untrustworthy, and every run / getter keeps
  std::string fatal_message;            // failing with this until the parts are uploaded afresh (emat_begin_upload)
  double last_run_ms = 0.0;
  // model
  std::vector<uint8_t> ref, partition_for_site;
  std::vector<double> nu_l, cumQ, cum_nu, mu, pi, q;
  std::vector<int32_t> ref_freqs;
  bool uniform_sites = false;       // one site partition, every nu_l == 1.0: EvoTable::uniform_sites
  bool cfg_no_uniform_sites = false;   // option "no_uniform_sites" (A/B and tests): the moves read the per-site arrays even then
  int num_partitions = 0;
  RunFlags flags{0.0, 0, 1};
  bool have_ref = false, have_evo = false, have_pop = false, have_coal = false;
  HostPopModel pop;
  DevBuf<uint8_t> d_ref, d_part; DevBuf<double> d_nu, d_cumQ, d_cum_nu, d_stats, d_mu, d_pi, d_q, d_sky_x, d_sky_g; DevBuf<int32_t> d_ref_freqs; DevBuf<PopTable> d_pop;
  bool model_dirty = true;
  // parts
  std::vector<PartHost> parts;
  int uploads_expected = 0;
  int root_part = -1;
  PinnedBytes h_slabs;
  size_t slab_bytes_total = 0;      // bytes of all slabs on the device (h_slabs is brought to this size when somebody pulls)
  DevBuf<uint8_t> d_slabs, d_snaps; DevBuf<uint64_t> d_slab_off; DevBuf<int32_t> d_order, d_part_status; DevBuf<int64_t> d_part_ticks, d_moves_for_part;
  bool slabs_on_device = false;     // device slabs are materialised
  bool host_slabs_current = false;  // h_slabs mirrors the device
  bool derived_valid = false;
  uint32_t max_slab_bytes = 0;
  std::vector<uint32_t> persistent_bytes;   // per part: slab size without scratch
  std::vector<uint32_t> prefix_bytes;       // per part: header + nodes + cells + trace (what the prefix-staged variant keeps in LDS)
  std::vector<uint32_t> used_bytes;         // per part: prefix + list content (what a part staged whole brings into LDS)
  // SharedCells: host mirror (absolute cell index) and the device copy the kernels read
  std::vector<double> sh_ktw, sh_popsize, sh_tsop; std::vector<int32_t> sh_nact;
  bool grid_mirrors_on_device = false;   // the grid was built on the device (emat_tree_repartition) and the four vectors above have not been fetched yet
  DevBuf<double> d_sh_ktw, d_sh_tsop; DevBuf<int32_t> d_sh_nact;
  SharedCells shared_dev{nullptr, nullptr, nullptr, 0};   // what make_args hands the kernels (the HBM-resident tree points it at its own grid arrays)
  uint32_t cfg_side_arena = 1;              // EMAT_SIDE_ARENA (tuning knob; 0 = off): a part that would be left with less arena than this in the main area joins the giants' 8-per-CU class.  The default, 1 byte, moves exactly the parts that cannot be staged WHOLE there (0.5 % at C4): with their lists in HBM they were the last chains of every pass (19.6 ms where the rest was done by 19.9: pass 21.5 -> 20.1 ms); 1-4 KB moves hundreds and loses
  bool cfg_giants = true;                   // EMAT_GIANTS (tuning knob): parts that cannot even stage their prefix get a class of their own
  double cfg_heap_per_node = 64.0;  // EMAT_HEAP_PER_NODE: heap bytes per node on top of slack x content
  uint32_t cfg_lds_scratch = 0;     // EMAT_LDS_SCRATCH (tuning knob): per-part LDS scratch arena; 0 = all scratch in HBM (measured best at C4)
  bool host_only = false;           // cfg.device == -1: uploads / coalescent staging only, every launch fails with EMAT_ERR_NO_DEVICE
  std::unique_ptr<CoalBuilder> coal_builder;
  // dense copy of every part's slab header (k_gather_headers): what the scalar getters read instead of the slabs
  DevBuf<uint8_t> d_headers; std::vector<uint8_t> h_headers; bool headers_current = false;
  GTreeHost gt;                     // the whole tree, when it lives in HBM (emat_tree_upload)
  BuiltTree built;                  // what emat_tree_build_usher_like made, until it is fetched (emat_tree_built_get)
  bool cfg_taper = true;            // EMAT_TICKET_TAPER: tickets of a part shrink (10 : 6 : 3 : 1 for four tickets, else n : ... : 1) instead of being equal
  int cfg_chunks = 4;               // EMAT_CHUNKS (tuning knob): tickets per part and pass (main class; measured at C4 once a ticket's release no longer wrote the L2 back, equal tickets: 2 -> 378, 3 -> 384, 6 -> 382, 10 -> 379, 16 -> 365, 32 -> 322 M moves/s; tapered: 3 -> 390, 4 -> 392, 5 -> 388; before: 1 -> 311, 2 -> 338, 3 -> 340, 4 -> 331, 8 -> 301)
  bool cfg_ticket_spread = false;   // EMAT_TICKET_XCD_SPREAD=1 (tests): odd ticket stride, a part's tickets on different XCDs
  int cfg_single_ticket_parts = 0;        // EMAT_SINGLE_TICKET_PARTS (tuning knob): how many of the largest main-class parts run their pass as one ticket
  bool cfg_ticket_full_release = false;   // EMAT_TICKET_RELEASE=full: agent-scope release at every hand-over
  bool cfg_chunks_forced = false;   // option "chunks" was given: tickets also when the parts are fewer than the wave slots (tests)
  DevBuf<int32_t> d_chunk_done, d_side_started;
  int cfg_parts_per_cu = 0;         // EMAT_PARTS_PER_CU (tuning knob): workgroups of the main class per CU, instead of the percentile rule
  bool cfg_gt_host_coal = false;    // EMAT_TREE_HOST_COALESCENT=1: emat_tree_repartition builds the coalescent tables on the host (bit-identical to the host cycle; tests)

  void set_error(const std::string& s) { last_error = s; }
};

namespace {

uint32_t a16(uint32_t x) { return (x + 15u) & ~15u; }

// Encode one part into its slab (layout: emat_slab.hpp).
void encode_slab(const emat_backend& B, const PartHost& ph, uint8_t* slab, uint32_t slab_bytes, uint32_t heap_bytes, uint32_t scratch_bytes, int cell_cap, int trace_cap) {
  std::memset(slab, 0, slab_bytes - scratch_bytes);   // scratch (the slab's tail) is transient: never read before written
  const FlatTree& t = ph.tree;
  const int n = t.num_nodes();
  SlabHeader* H = (SlabHeader*)slab;
  H->magic = k_slab_magic; H->slab_bytes = slab_bytes; H->n_nodes = n; H->root = t.root;
  H->flags = ph.includes_run_root ? k_flag_includes_run_root : 0u;
  H->status = 0; H->rng_key = ph.rng.key; H->rng_counter = ph.rng.counter; H->rng_spare = ph.rng.spare; H->rng_has_spare = ph.rng.has_spare ? 1u : 0u;
  uint32_t off = sizeof(SlabHeader);
  H->off_nodes = off; off += (uint32_t)n * (uint32_t)sizeof(NodeRec);
  H->off_cells = off; off += a16((uint32_t)cell_cap * cell_bytes_for(ph.includes_run_root));
  H->off_trace = off; off += a16((uint32_t)trace_cap * 32u);
  H->heap_begin = off; H->heap_end = off + heap_bytes;
  H->scratch_begin = H->heap_end; H->scratch_end = H->scratch_begin + scratch_bytes;
  H->cell_first = ph.coal.cell_first; H->n_cells = (int)ph.coal.k_bar_p.size(); H->cell_cap = cell_cap; H->n_cells_total = ph.coal.n_cells_total;
  H->t_ref = ph.coal.t_ref; H->t_step = ph.coal.t_step;
  H->trace_cap = trace_cap; H->trace_len = std::min<int>(trace_cap, (int)(ph.trace.size() / 4));
  if (H->trace_len > 0) std::memcpy(slab + H->off_trace, ph.trace.data(), (size_t)H->trace_len * 32);
  NodeRec* N = (NodeRec*)(slab + H->off_nodes);
  uint32_t top = H->heap_begin;
  for (int i = 0; i < n; ++i) {
    NodeRec& r = N[i];
    r.parent = t.parent[i]; r.child0 = t.child0[i]; r.child1 = t.child1[i];
    r.t_min = t.t_min[i]; r.t_max = t.t_max[i]; r.t = t.t[i]; r.lambda = 0.0; r.n_missing = 0;
    int nm = t.mut_offset[i + 1] - t.mut_offset[i], ni = t.miss_offset[i + 1] - t.miss_offset[i], nf = t.mfs_offset[i + 1] - t.mfs_offset[i];
    // (counts were checked against k_max_list_len by the caller, materialize)
    r.muts.off = top; r.muts.cnt = (uint16_t)nm; r.muts.cap = list_cap_for(a16(nm * 16u), 16u);
    MutRec* m = (MutRec*)(slab + top);
    for (int k = 0; k < nm; ++k) { int s = t.mut_offset[i] + k; m[k].t = t.mut_t[s]; m[k].site = t.mut_site[s]; m[k].from = t.mut_from[s]; m[k].to = t.mut_to[s]; m[k].pad = 0; }
    top += a16(nm * 16u);
    r.miss.off = top; r.miss.cnt = (uint16_t)ni; r.miss.cap = list_cap_for(a16(ni * 8u), 8u);
    IvRec* iv = (IvRec*)(slab + top);
    for (int k = 0; k < ni; ++k) { int s = t.miss_offset[i] + k; iv[k].start = t.miss_start[s]; iv[k].end = t.miss_end[s]; }
    top += a16(ni * 8u);
    r.mfs.off = top; r.mfs.cnt = (uint16_t)nf; r.mfs.cap = list_cap_for(a16(nf * 8u), 8u);
    FsRec* fs = (FsRec*)(slab + top);
    for (int k = 0; k < nf; ++k) { int s = t.mfs_offset[i] + k; fs[k].site = t.mfs_site[s]; fs[k].state = t.mfs_state[s]; }
    top += a16(nf * 8u);
  }
  H->heap_top = top;
  double* cb = (double*)(slab + H->off_cells);
  const int nc = (int)ph.coal.k_bar_p.size();
  for (int w = 0; w < nc; ++w) {
    cb[w] = ph.coal.k_bar_p[w]; cb[cell_cap + w] = ph.coal.k_twiddle_bar_p[w];
    if (!ph.includes_run_root) continue;      // the run-wide arrays live once per device (SharedCells); the root part, which may append cells, keeps its own
    cb[2 * cell_cap + w] = ph.coal.k_twiddle_bar[w]; cb[3 * cell_cap + w] = ph.coal.popsize_bar[w];
    cb[4 * cell_cap + w] = ph.coal.t_step / ph.coal.popsize_bar[w];   // the factor every cell term starts with, divided once
    ((int32_t*)(cb + 5 * cell_cap))[w] = ph.coal.num_active_parts[w];
  }
  (void)B;
}

uint32_t heap_content_bytes(const FlatTree& t) {
  uint32_t b = 0;
  for (int i = 0; i < t.num_nodes(); ++i)
    b += a16((t.mut_offset[i + 1] - t.mut_offset[i]) * 16u) + a16((t.miss_offset[i + 1] - t.miss_offset[i]) * 8u) + a16((t.mfs_offset[i + 1] - t.mfs_offset[i]) * 8u);
  return b;
}

// Decode the device image of a part back into its host FlatTree + coalescent window + rng + stats.
void decode_slab(PartHost& ph, const uint8_t* slab, const double* shared_ktw, const double* shared_popsize, const int32_t* shared_nact) {
  const SlabHeader* H = (const SlabHeader*)slab;
  const NodeRec* N = (const NodeRec*)(slab + H->off_nodes);
  const int n = H->n_nodes;
  FlatTree& t = ph.tree;
  int nm = 0, ni = 0, nf = 0;
  for (int i = 0; i < n; ++i) { nm += N[i].muts.cnt; ni += N[i].miss.cnt; nf += N[i].mfs.cnt; }
  t.allocate(n, nm, ni, nf);
  t.root = H->root;
  int km = 0, ki = 0, kf = 0;
  for (int i = 0; i < n; ++i) {
    const NodeRec& r = N[i];
    t.parent[i] = r.parent; t.child0[i] = r.child0; t.child1[i] = r.child1; t.t[i] = r.t; t.t_min[i] = r.t_min; t.t_max[i] = r.t_max;
    const MutRec* m = (const MutRec*)(slab + r.muts.off);
    for (int k = 0; k < r.muts.cnt; ++k) { t.mut_site[km] = m[k].site; t.mut_from[km] = m[k].from; t.mut_to[km] = m[k].to; t.mut_t[km] = m[k].t; ++km; }
    const IvRec* iv = (const IvRec*)(slab + r.miss.off);
    for (int k = 0; k < r.miss.cnt; ++k) { t.miss_start[ki] = iv[k].start; t.miss_end[ki] = iv[k].end; ++ki; }
    const FsRec* fs = (const FsRec*)(slab + r.mfs.off);
    for (int k = 0; k < r.mfs.cnt; ++k) { t.mfs_site[kf] = fs[k].site; t.mfs_state[kf] = fs[k].state; ++kf; }
    t.mut_offset[i + 1] = km; t.miss_offset[i + 1] = ki; t.mfs_offset[i + 1] = kf;
  }
  ph.rng.counter = H->rng_counter; ph.rng.spare = H->rng_spare; ph.rng.has_spare = H->rng_has_spare != 0;
  { const double* tr = (const double*)(slab + H->off_trace); ph.trace.assign(tr, tr + (size_t)4 * H->trace_len); }
  const int nc = H->n_cells, cap = H->cell_cap;
  const double* cb = (const double*)(slab + H->off_cells);
  ph.coal.n_cells_total = H->n_cells_total;
  ph.coal.k_bar_p.assign(cb, cb + nc); ph.coal.k_twiddle_bar_p.assign(cb + cap, cb + cap + nc);
  if ((H->flags & k_flag_includes_run_root) != 0) {
    ph.coal.k_twiddle_bar.assign(cb + 2 * cap, cb + 2 * cap + nc); ph.coal.popsize_bar.assign(cb + 3 * cap, cb + 3 * cap + nc);
    const int32_t* na = (const int32_t*)(cb + 5 * cap); ph.coal.num_active_parts.assign(na, na + nc);
  } else if (shared_ktw != nullptr) {   // the window of the device's shared arrays (they do not change while the parts run)
    const int f = H->cell_first;
    ph.coal.k_twiddle_bar.assign(shared_ktw + f, shared_ktw + f + nc); ph.coal.popsize_bar.assign(shared_popsize + f, shared_popsize + f + nc);
    ph.coal.num_active_parts.assign(shared_nact + f, shared_nact + f + nc);
  }
}

emat_status fail(emat_backend* h, emat_status st, const std::string& msg) { h->set_error(msg); return st; }

// The streams the side launches of a pass fork onto: one small pool per device for the whole process.  The runtime maps
// streams onto a handful of hardware queues (four by default) and two streams on one queue run one after the other, so
// handles that come and go must not each bring streams of their own.  (The current device is the caller's.)
hipStream_t side_stream(int device, int i) {
  static std::mutex mu;
  static std::vector<std::array<hipStream_t, emat_backend::k_max_classes - 1>> pool;
  std::lock_guard<std::mutex> lock(mu);
  if ((int)pool.size() <= device) pool.resize((size_t)device + 1, std::array<hipStream_t, emat_backend::k_max_classes - 1>{});
  if (!pool[device][i] && hipStreamCreate(&pool[device][i]) != hipSuccess) return nullptr;
  return pool[device][i];
}

// Several handles may live in one process, one per GPU: every entry point that talks to the device selects its own first.
inline bool bind_device(emat_backend* h) { return h->host_only || hipSetDevice(h->cfg.device) == hipSuccess; }

emat_status sync_model_to_device(emat_backend* h) {
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (!h->model_dirty) return EMAT_OK;
  auto& B = *h;
  auto set_error = [&](const std::string& s) { B.set_error(s); };
  HIP_TRY(B.d_ref.upload(B.ref.data(), B.ref.size()));
  HIP_TRY(B.d_part.upload(B.partition_for_site.data(), B.partition_for_site.size()));
  HIP_TRY(B.d_nu.upload(B.nu_l.data(), B.nu_l.size()));
  HIP_TRY(B.d_cumQ.upload(B.cumQ.data(), B.cumQ.size()));
  HIP_TRY(B.d_cum_nu.upload(B.cum_nu.data(), B.cum_nu.size()));
  HIP_TRY(B.d_mu.upload(B.mu.data(), B.mu.size()));
  HIP_TRY(B.d_pi.upload(B.pi.data(), B.pi.size()));
  HIP_TRY(B.d_q.upload(B.q.data(), B.q.size()));
  HIP_TRY(B.d_ref_freqs.upload(B.ref_freqs.data(), B.ref_freqs.size()));
  HIP_TRY(B.d_sky_x.upload(B.pop.x.data(), B.pop.x.size()));
  HIP_TRY(B.d_sky_g.upload(B.pop.gamma.data(), B.pop.gamma.size()));
  PopTable pt{};
  pt.kind = B.pop.kind; pt.skygrid_type = B.pop.skygrid_type; pt.skygrid_num_knots = (int)B.pop.x.size();
  for (int i = 0; i < 4; ++i) pt.p[i] = B.pop.p[i];
  pt.t_c = B.pop.t_c; pt.skygrid_x = B.d_sky_x.p; pt.skygrid_gamma = B.d_sky_g.p;
  pt.skygrid_inv_dx = (B.pop.x.size() >= 2 && B.pop.x.back() > B.pop.x.front()) ? (double)(B.pop.x.size() - 1) / (B.pop.x.back() - B.pop.x.front()) : 0.0;
  HIP_TRY(B.d_pop.upload(&pt, 1));
  B.model_dirty = false;
  return EMAT_OK;
}

// reference_cum_Q and state frequencies of the reference sequence (phylo_tree_calc.cpp:379-388, :95-106)
void refresh_ref_derived(emat_backend* h) {
  const int L = h->L;
  h->cumQ.assign(L + 1, 0.0);
  double so_far = 0.0;
  for (int l = 0; l < L; ++l) {
    const int p = h->partition_for_site[l];
    so_far += h->mu[p] * h->nu_l[l] * (-h->q[p * 16 + h->ref[l] * 5]);
    h->cumQ[l + 1] = so_far;
  }
  {   // prefix table for k_global_stats: cum_nu[k][beta][a] = sum over sites l < k of partition beta with reference state a of nu_l
    const int W = 4 * h->num_partitions;
    h->cum_nu.assign((size_t)(L + 1) * W, 0.0);
    for (int l = 0; l < L; ++l) {
      const double* prev = &h->cum_nu[(size_t)l * W]; double* next = &h->cum_nu[(size_t)(l + 1) * W];
      for (int k = 0; k < W; ++k) next[k] = prev[k];
      next[4 * h->partition_for_site[l] + h->ref[l]] += h->nu_l[l];
    }
  }
  h->uniform_sites = h->num_partitions == 1 && !h->cfg_no_uniform_sites && std::all_of(h->nu_l.begin(), h->nu_l.end(), [](double x) { return x == 1.0; });
  h->ref_freqs.assign((size_t)h->num_partitions * 4, 0);
  for (int l = 0; l < L; ++l) ++h->ref_freqs[h->partition_for_site[l] * 4 + h->ref[l]];
  h->model_dirty = true;
}

KernelArgs make_args(emat_backend* h) {
  KernelArgs a{};
  a.slabs = h->d_slabs.p; a.slab_off = h->d_slab_off.p; a.order = h->d_order.p; a.part_status = h->d_part_status.p; a.moves_for_part = nullptr; a.part_ticks = h->d_part_ticks.p; a.ref_freqs = h->d_ref_freqs.p; a.cum_nu = h->d_cum_nu.p; a.stats_out = h->d_stats.p;
  a.evo.num_sites = h->L; a.evo.num_partitions = h->num_partitions; a.evo.uniform_sites = h->uniform_sites ? 1 : 0; a.evo.pad_ = 0;
  a.evo.ref_sequence = h->d_ref.p; a.evo.partition_for_site = h->d_part.p; a.evo.nu_l = h->d_nu.p; a.evo.cum_Q_l = h->d_cumQ.p;
  a.evo.mu = h->d_mu.p; a.evo.pi = h->d_pi.p; a.evo.q = h->d_q.p;
  a.pop = h->d_pop.p; a.shared = h->shared_dev; a.flags = h->flags; a.num_parts = (int)h->parts.size();
  a.lds_slab_bytes = 0; a.lds_scratch_bytes = 0; a.moves_per_part = 0; a.extra_moves_part0 = 0; a.one_more_below = 0; a.chunks = 1; a.class_count = 0; a.class_stride = 0; a.chunk_done = nullptr; a.snaps = nullptr;
  return a;
}

// Bring the host copies of all parts up to date with the device.
emat_status pull_from_device(emat_backend* h);
emat_status pull_headers(emat_backend* h);
emat_status materialize(emat_backend* h);
emat_status launch_moves(emat_backend* h, int64_t per_part, int64_t extra0, const std::vector<int64_t>* counts, int32_t one_more_below);

// After a launch: did every part run its chain to completion?  A part that ran out of list-heap or scratch space
// stops BEFORE a move with its state intact (status 101): it is given twice the room and the rest of its moves, up
// to four times.  Any other status means an invariant broke inside a move; that is reported, loudly, and the caller
// must not use the part's tree.
// Launch order for the NEXT passes from the durations measured in the last one: within each size class, the part that
// took longest goes first.  The dispatcher hands workgroups to free slots in index order, so this is longest-processing-
// time-first list scheduling with real times instead of the size proxy (correlation 0.44 at C4); a chain's cost changes
// slowly from pass to pass.  Called with the stream idle.
emat_status refresh_order_from_ticks(emat_backend* h) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  const size_t n = h->parts.size();
  if (!h->order_valid || n == 0 || !h->cfg_order_by_time) return EMAT_OK;
  std::vector<int64_t> ticks(n);
  HIP_TRY(hipMemcpy(ticks.data(), h->d_part_ticks.p, n * sizeof(int64_t), hipMemcpyDeviceToHost));
  std::vector<int32_t>& order = h->h_order;
  for (int c = 0; c < h->num_classes; ++c)
    std::stable_sort(order.begin() + h->class_begin[c], order.begin() + h->class_begin[c + 1], [&](int a, int b) { return ticks[a] > ticks[b]; });
  HIP_TRY(h->d_order.upload(order.data(), order.size()));
  return EMAT_OK;
}

// The side classes of a pass run on streams of their own and the engine's stream does not wait for them when the pass is launched
// (the next pass of the main class need not, and they are the longest chains of a pass): whoever is about to read or change
// slabs on the engine's stream joins them first.
emat_status join_side_classes(emat_backend* h) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  if (!h->sides_in_flight) return EMAT_OK;
  for (int c = 1; c < emat_backend::k_max_classes; ++c) if (h->class_stream[c]) HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join[c], 0));
  h->sides_in_flight = false;
  return EMAT_OK;
}

emat_status finish_pass(emat_backend* h) {
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (h->fatal_status != EMAT_OK) return fail(h, h->fatal_status, h->fatal_message);
  if (h->host_only || !h->pass_pending || !h->slabs_on_device) return EMAT_OK;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  const size_t n = h->parts.size();
  std::vector<int32_t> status(n);
  for (int round = 0; round < 5; ++round) {
    HostLaps laps;
    { emat_status js = join_side_classes(h); if (js) return js; }
    h->sides_must_fork = true;
    HIP_TRY(hipStreamSynchronize(h->stream));
    laps.mark("finish_pass: 1 wait for the moves");
    HIP_TRY(hipMemcpy(status.data(), h->d_part_status.p, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    laps.mark("finish_pass: 2 status D2H");
    h->pass_pending = false;
    size_t stopped = 0, fatal = n;
    for (size_t p = 0; p < n; ++p) if (status[p] != 0) { ++stopped; if (status[p] != k_part_need_space && status[p] != k_part_need_cells && fatal == n) fatal = p; }
    if (stopped == 0 && verbose_reports()) {   // what bounded the pass: the slowest chains next to the mean
      std::vector<int64_t> ticks(2 * n);
      HIP_TRY(hipMemcpy(ticks.data(), h->d_part_ticks.p, 2 * n * sizeof(int64_t), hipMemcpyDeviceToHost));
      std::vector<int> idx(n); std::iota(idx.begin(), idx.end(), 0);
      std::partial_sort(idx.begin(), idx.begin() + std::min<size_t>(4, n), idx.end(), [&](int a, int b) { return ticks[a] > ticks[b]; });
      double sum = 0; int64_t first = INT64_MAX, last = 0;
      for (size_t p = 0; p < n; ++p) { sum += (double)ticks[p]; first = std::min(first, ticks[n + p]); last = std::max(last, ticks[n + p] + ticks[p]); }
      float kms = 0.f; (void)hipEventElapsedTime(&kms, h->ev_start, h->ev_stop);
      if (h->d_chunk_done.n >= n + 130) {
        std::vector<int32_t> w(130); HIP_TRY(hipMemcpy(w.data(), h->d_chunk_done.p + n, 130 * sizeof(int32_t), hipMemcpyDeviceToHost));
        double wait = 0; for (int i = 0; i < 64; ++i) { unsigned long long v; memcpy(&v, &w[(n & 1) + 2 * i], 8); wait += (double)v; }   // (the counters start at the next even index)
        fprintf(stderr, "[emat] tickets waited %.2f s of slot time for their predecessors (%.1f %% of %d slots x pass)\n", wait / 1e8, 100.0 * wait / 1e5 / (kms * 16.0 * h->num_cus), 16 * h->num_cus);
      }
      fprintf(stderr, "[emat] pass: main class %.1f ms on the device (all chains: first start to last end %.1f ms) | chains: mean %.2f ms, slowest", kms, (last - first) / 1e5, sum / n / 1e5);
      for (size_t k = 0; k < std::min<size_t>(4, n); ++k) fprintf(stderr, " %.1f ms from %.1f (part %d, %d nodes%s)", ticks[idx[k]] / 1e5, (ticks[n + idx[k]] - first) / 1e5, idx[k], h->parts[idx[k]].n_nodes, idx[k] == h->root_part ? ", root part" : "");
      std::partial_sort(idx.begin(), idx.begin() + std::min<size_t>(3, n), idx.end(), [&](int a, int b) { return ticks[n + a] + ticks[a] > ticks[n + b] + ticks[b]; });
      fprintf(stderr, " | last to end:");
      for (size_t k = 0; k < std::min<size_t>(3, n); ++k) fprintf(stderr, " part %d (%d nodes, %u B) %.1f ms from %.1f", idx[k], h->parts[idx[k]].n_nodes, h->persistent_bytes[idx[k]], ticks[idx[k]] / 1e5, (ticks[n + idx[k]] - first) / 1e5);
      fprintf(stderr, "\n");
    }
    if (stopped == 0) { if (round == 0 && h->last_launch_uniform) (void)refresh_order_from_ticks(h); laps.mark("finish_pass: 3 status scan + refresh_order_from_ticks"); return EMAT_OK; }
    h->host_slabs_current = false; h->headers_current = false;
    emat_status st = pull_from_device(h); if (st) return st;
    if (fatal != n) {
      const SlabHeader* H = (const SlabHeader*)(h->h_slabs.data() + h->parts[fatal].slab_off);
      h->fatal_status = (status[fatal] == k_part_cell_overflow || status[fatal] == k_part_list_limit) ? EMAT_ERR_CAPACITY : EMAT_ERR_INTERNAL;
      h->fatal_message = "part " + std::to_string(fatal) + " stopped inside a move with status " + std::to_string(status[fatal]) + " (device source line " + std::to_string(H->fail_line) +
                         "); " + std::to_string(stopped) + " part(s) stopped in all";
      if (H->fail_line == -3)
        h->fatal_message += ": the dynamic LDS block of k_run_moves does not start where the device code was compiled to find it (k_lds_dyn_base): the library was not built with "
                            "-mllvm -amdgpu-lower-module-lds-strategy=module (csrc/Makefile)";
      else if (H->fail_line < 0)
        h->fatal_message += ": a ticket found that its predecessor had handed the part over on another XCD (the cheap hand-over is only valid behind one L2; the device did not "
                            "deal this launch's workgroups to its XCDs as probed): run with EMAT_TICKET_RELEASE=full";
      if (status[fatal] == k_part_list_limit)
        h->fatal_message += ": a per-node list would exceed " + std::to_string(k_max_list_len) + " entries (16-bit list counts, emat_slab.hpp): nothing was truncated, the run cannot continue";
      return fail(h, h->fatal_status, h->fatal_message);
    }
    if (round == 4) return fail(h, EMAT_ERR_CAPACITY, std::to_string(stopped) + " part(s) still out of slab space after four doublings");
    std::vector<int64_t> counts(n, 0);
    for (size_t p = 0; p < n; ++p) if (status[p] != 0) {
      PartHost& ph = h->parts[p];
      if (status[p] == k_part_need_cells) ph.cell_boost *= 4; else ph.space_boost *= 2.0;
      counts[p] = h->expected_moves[p] - ph.stats.moves_done;
      ph.stats.status = 0;
    }
    if (verbose_reports()) fprintf(stderr, "[emat] %zu part(s) ran out of slab space or grid cells: re-materialising with more room and running the rest of their moves\n", stopped);
    if (h->derived_valid) {   // (h_slabs is what pull_from_device just decoded: the slabs as the pass left them)
      parallel_for((int)n, [&](int p) {
        PartHost& ph = h->parts[(size_t)p];
        const uint8_t* slab = h->h_slabs.data() + ph.slab_off;
        const SlabHeader* H = (const SlabHeader*)slab; const NodeRec* N = (const NodeRec*)(slab + H->off_nodes);
        ph.kept_lambda.resize((size_t)H->n_nodes); ph.kept_n_missing.resize((size_t)H->n_nodes);
        for (int i = 0; i < H->n_nodes; ++i) { ph.kept_lambda[(size_t)i] = N[i].lambda; ph.kept_n_missing[(size_t)i] = N[i].n_missing; }
        ph.kept_log_G = H->log_G; ph.kept_log_aug_prior = H->log_aug_prior; ph.derived_kept = true;
      });
    }
    h->slabs_on_device = false; h->host_slabs_current = false; h->headers_current = false;   // every part is re-encoded from its decoded state (tree, RNG, cells, statistics, derived quantities)
    st = launch_moves(h, 0, 0, &counts, 0); if (st) return st;
  }
  return EMAT_OK;
}

emat_status pull_from_device_impl(emat_backend* h);
emat_status pull_from_device(emat_backend* h) {
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (h->fatal_status != EMAT_OK) return fail(h, h->fatal_status, h->fatal_message);
  if (h->host_only || !h->slabs_on_device) return EMAT_OK;
  if (h->pass_pending) { emat_status st = finish_pass(h); if (st) return st; }
  return pull_from_device_impl(h);
}
emat_status pull_from_device_impl(emat_backend* h) {
  if (h->host_only || !h->slabs_on_device || h->host_slabs_current) return EMAT_OK;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->grid_mirrors_on_device) {   // the run-wide cell arrays of a grid built on the device: decode_slab reads them
    const size_t nc = (size_t)h->shared_dev.num_cells;
    h->sh_ktw.resize(nc); h->sh_popsize.resize(nc); h->sh_tsop.resize(nc); h->sh_nact.resize(nc);
    HIP_TRY(hipMemcpy(h->sh_ktw.data(), h->shared_dev.k_twiddle_bar, nc * 8, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(h->sh_popsize.data(), h->gt.co_popsize.p, nc * 8, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(h->sh_tsop.data(), h->shared_dev.ts_over_pop, nc * 8, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(h->sh_nact.data(), h->shared_dev.num_active_parts, nc * 4, hipMemcpyDeviceToHost));
    h->grid_mirrors_on_device = false;
  }
  HIP_TRY(h->h_slabs.resize(h->slab_bytes_total));   // (grow-only; a repartition on the device does not touch the host mirror)
  HIP_TRY(hipMemcpy(h->h_slabs.data(), h->d_slabs.p, h->h_slabs.size(), hipMemcpyDeviceToHost));
  parallel_for((int)h->parts.size(), [&](int p) {
    PartHost& ph = h->parts[p];
    const uint8_t* slab = h->h_slabs.data() + ph.slab_off;
    const SlabHeader* H = (const SlabHeader*)slab;
    decode_slab(ph, slab, h->sh_ktw.empty() ? nullptr : h->sh_ktw.data(), h->sh_popsize.data(), h->sh_nact.data());
    ph.stats.status = H->status; ph.stats.num_nodes = H->n_nodes; ph.stats.moves_done = H->moves_done;
    for (int k = 0; k < 5; ++k) { ph.stats.proposed[k] = H->proposed[k]; ph.stats.accepted[k] = H->accepted[k]; }
    ph.stats.algorithmic_bytes = H->alg_bytes; ph.stats.algorithmic_write_bytes = 16 * (int64_t)H->alg_write16; ph.stats.rng_draws = (int64_t)H->rng_counter; ph.stats.device_ticks = H->device_ticks;
  });
  h->host_slabs_current = true;
  return EMAT_OK;
}

// Dense copy of the parts' headers: what emat_get_totals / emat_part_get_stats need is 256 bytes per part, not the slabs.
emat_status pull_headers(emat_backend* h) {
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (h->fatal_status != EMAT_OK) return fail(h, h->fatal_status, h->fatal_message);
  if (h->host_only || !h->slabs_on_device) return EMAT_OK;
  if (h->pass_pending) { emat_status st = finish_pass(h); if (st) return st; }
  if (h->headers_current) return EMAT_OK;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  const size_t n = h->parts.size();
  h->h_headers.resize(n * sizeof(SlabHeader));
  if (h->host_slabs_current) {
    for (size_t p = 0; p < n; ++p) std::memcpy(h->h_headers.data() + p * sizeof(SlabHeader), h->h_slabs.data() + h->parts[p].slab_off, sizeof(SlabHeader));
  } else {
    HIP_TRY(h->d_headers.alloc(n * sizeof(SlabHeader)));
    KernelArgs a = make_args(h);
    hipLaunchKernelGGL(k_gather_headers, dim3((unsigned)n), dim3(k_wave), 0, h->stream, a, h->d_headers.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(h->h_headers.data(), h->d_headers.p, n * sizeof(SlabHeader), hipMemcpyDeviceToHost));
  }
  h->headers_current = true;
  return EMAT_OK;
}
inline const SlabHeader* header_of(const emat_backend* h, size_t part) { return (const SlabHeader*)(h->h_headers.data() + part * sizeof(SlabHeader)); }

// Capacities of one part's slab: what a move may need on top of the part's present content.
struct SlabGeo { uint32_t heap, scratch; int cell_cap; uint32_t bytes; };
SlabGeo slab_geometry(const emat_backend* h, int n, int num_muts, uint32_t content, int nc, bool includes_run_root, double space_boost, int cell_boost = 1) {
  const double slack = h->cfg.slab_slack > 0 ? h->cfg.slab_slack : 3.0;
  const int trace_cap = h->cfg.trace_moves > 0 ? h->cfg.trace_moves : 0;
  SlabGeo g;
  g.heap = a16((uint32_t)(space_boost * std::max<double>(2048.0, content * slack + h->cfg_heap_per_node * n)));
  // worst case of one move: an unlimited SPR scan visits every (branch, inter-mutation segment) region of the
  // part (48 B each) with a DFS stack of up to 4 items per region (12 B each), next to two graft analyses
  const uint32_t regions_max = (uint32_t)n + (uint32_t)num_muts;
  g.scratch = a16((uint32_t)(space_boost * std::max<uint32_t>(8192u, 128u * regions_max + 4u * content + 256u * (uint32_t)n)));
  g.cell_cap = includes_run_root ? nc + cell_boost * std::max(512, nc) : nc;   // room for the root part's grid to grow into the past (a part that outgrows it stops with status 103 / 105)
  g.bytes = (uint32_t)sizeof(SlabHeader) + (uint32_t)n * (uint32_t)sizeof(NodeRec) + a16((uint32_t)g.cell_cap * cell_bytes_for(includes_run_root)) + a16((uint32_t)trace_cap * 32u) + g.heap + g.scratch;
  return g;
}
void place_slab(emat_backend* h, size_t p, const SlabGeo& g, uint64_t& off, uint32_t content_bytes) {
  if (h->used_bytes.size() <= p) h->used_bytes.resize(p + 1, 0u);
  h->used_bytes[p] = g.bytes - g.scratch - g.heap + content_bytes;
  PartHost& ph = h->parts[p];
  ph.slab_off = off; ph.slab_bytes = g.bytes; ph.scratch_bytes = g.scratch; off += g.bytes;
  h->persistent_bytes[p] = g.bytes - g.scratch;
  h->prefix_bytes[p] = g.bytes - g.scratch - g.heap;
  h->max_slab_bytes = std::max(h->max_slab_bytes, g.bytes);
}
// Size classes: which parts share a launch and an LDS staging area.  Class 0 has the largest area; the last class (the
// "main" one) holds the bulk of the parts.  Sets h->class_of / class_lds / class_begin; build_order lays the launch
// order out class by class.
void assign_size_classes(emat_backend* h) {
  const size_t n = h->parts.size();
  const bool verbose = verbose_reports();
  std::vector<uint32_t> v = h->persistent_bytes;
  const bool by_percentiles = h->cfg.use_lds && n != 0 && h->cfg_class_pct.size() > 1;
  if (by_percentiles || verbose) std::sort(v.begin(), v.end());   // (the default rule needs one order statistic: nth_element below)
  h->class_of.assign(n, 0);
  std::vector<uint32_t> areas;   // per class, descending
  const uint32_t lds_cu = 160u * 1024u, overhead = k_lds_static_bytes + (h->cfg.use_lds ? h->cfg_lds_scratch : 0u);
  auto area_for = [&](uint32_t k) {   // the staging area of a workgroup when k of them share a CU (LDS is allocated in 512-byte granules)
    const uint32_t share = (lds_cu / k) & ~511u;
    return share <= overhead ? 0u : std::min<uint32_t>((share - overhead) & ~15u, h->cfg_lds_max & ~15u);
  };
  if (!h->cfg.use_lds || n == 0) areas.push_back(0u);
  else if (h->cfg_class_pct.size() > 1) {
    // tuning knob EMAT_LDS_CLASSES="p1,p2,...": classes by rank of persistent size, class c closing at percentile p_c and
    // staging that percentile's size
    std::vector<std::pair<uint32_t, uint32_t>> asc;   // (largest persistent size of the class, staging bytes), ascending
    size_t lo = 0;
    for (size_t ci = 0; ci < h->cfg_class_pct.size(); ++ci) {
      const int pct = h->cfg_class_pct[ci];
      const bool last = ci + 1 == h->cfg_class_pct.size() || (int)asc.size() + 1 == emat_backend::k_max_classes;
      size_t hi = std::min(n, (n * (size_t)pct + 99) / 100);
      if (pct >= 100 || last) hi = n;
      if (hi <= lo) { if (last) break; continue; }
      uint32_t need = (v[(last ? std::min(n, (n * (size_t)pct + 99) / 100) : hi) - 1] + 511u) & ~511u;
      if (need > h->cfg_lds_max) need = h->cfg_lds_max & ~511u;   // larger parts: prefix-staged or HBM only
      asc.push_back({v[hi - 1], need});
      lo = hi;
      if (last) break;
    }
    if (asc.empty()) asc.push_back({v.back(), 0u});
    const int nc = (int)asc.size();
    for (int c = 0; c < nc; ++c) areas.push_back(asc[nc - 1 - c].second);
    for (size_t p = 0; p < n; ++p) { int c = 0; while (c + 1 < nc && h->persistent_bytes[p] > asc[c].first) ++c; h->class_of[p] = nc - 1 - c; }
  } else {
    // The default.  The percentile only says which parts MUST fit whole.  LDS is the resource that limits residency, so
    // take the most workgroups per CU (up to the 16 the VGPR budget allows) whose share of the 160 KiB still holds that
    // percentile, and give every workgroup its whole share: larger parts than asked for get staged whole at no cost in
    // occupancy, the rest stage their prefix.
    const int pct = h->cfg_class_pct.empty() ? 60 : h->cfg_class_pct[0];
    const size_t hi = pct >= 100 ? n : std::max<size_t>(1, std::min(n, (n * (size_t)pct + 99) / 100));
    if (!verbose) std::nth_element(v.begin(), v.begin() + (hi - 1), v.end());
    const uint32_t need = std::min<uint32_t>((v[hi - 1] + 511u) & ~511u, h->cfg_lds_max & ~511u);
    uint32_t main_area = 0;
    for (uint32_t k = 4u * EMAT_WAVES_PER_EU; k >= 1; --k) {   // 4 SIMDs x waves per SIMD allowed by the VGPR budget (one wave per workgroup)
      const uint32_t area = area_for(k);
      if (area == 0) continue;
      if (area >= need || k == 1) { main_area = area; break; }
    }
    if (h->cfg_parts_per_cu > 0 && area_for((uint32_t)h->cfg_parts_per_cu) != 0) main_area = area_for((uint32_t)h->cfg_parts_per_cu);   // EMAT_PARTS_PER_CU
    // Giants: a part whose fixed-size prefix (header, nodes, cells) does not fit the area would run entirely out of HBM,
    // at less than half the speed, and -- every part doing the same number of moves -- hold up the whole pass.  They
    // get launches of their own, with areas for 8 and for 1 workgroup per CU: each giant takes the smaller area if it
    // holds its prefix.  (One area sized for the largest giant, as in round 1, put every giant at one workgroup per CU,
    // and a partition that has drifted for a while holds hundreds of them: passes of 52 ms instead of 32 at C4.  More
    // than two side launches would need more concurrent streams than the runtime has hardware queues -- four by default,
    // GPU_MAX_HW_QUEUES -- and streams that share a queue run one after the other.)
    std::vector<uint32_t> ladder;
    if (h->cfg_giants) for (uint32_t k : {8u, 1u}) { const uint32_t a = area_for(k); if (a > main_area && (ladder.empty() || a > ladder.back())) ladder.push_back(a); }
    std::vector<int> rung_of(n, -1); std::vector<int> used(ladder.size(), 0);
    if (!ladder.empty())
      for (size_t p = 0; p < n; ++p) if (h->prefix_bytes[p] > main_area || (h->cfg_side_arena != 0 && p < h->used_bytes.size() && h->used_bytes[p] + k_lds_heap_room + h->cfg_side_arena > main_area)) {
        size_t r = 0; while (r + 1 < ladder.size() && ladder[r] < h->prefix_bytes[p]) ++r;
        rung_of[p] = (int)r; used[r] = 1;
      }
    std::vector<int> class_of_rung(ladder.size(), -1);
    for (int r = (int)ladder.size() - 1; r >= 0; --r) if (used[r] && (int)areas.size() + 1 < emat_backend::k_max_classes) { class_of_rung[r] = (int)areas.size(); areas.push_back(ladder[r]); }
    const int main_class = (int)areas.size();
    areas.push_back(main_area);
    for (size_t p = 0; p < n; ++p) h->class_of[p] = rung_of[p] >= 0 && class_of_rung[rung_of[p]] >= 0 ? class_of_rung[rung_of[p]] : main_class;
  }
  h->num_classes = (int)areas.size();
  std::vector<int> count(h->num_classes, 0);
  for (size_t p = 0; p < n; ++p) ++count[h->class_of[p]];
  h->class_begin[0] = 0;
  for (int c = 0; c < h->num_classes; ++c) { h->class_lds[c] = areas[c]; h->class_begin[c + 1] = h->class_begin[c] + count[c]; }
  if (verbose && n > 0) {
    fprintf(stderr, "[emat] parts %zu persistent bytes p50 %u p90 %u p99 %u max %u | classes:", n, v[n / 2], v[n * 9 / 10], v[n * 99 / 100], v.back());
    for (int c = 0; c < h->num_classes; ++c) fprintf(stderr, " [%d parts, LDS %u]", h->class_begin[c + 1] - h->class_begin[c], h->class_lds[c]);
    fprintf(stderr, "\n");
  }
}

// The run-wide coalescent cell arrays (SharedCells, emat_slab.hpp) from the host's copies of the parts: every part carries the
// run's values over its own window, and the windows agree where they overlap.
emat_status upload_shared_cells(emat_backend* h) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  int total = 0;
  for (const PartHost& ph : h->parts) total = std::max(total, ph.coal.cell_first + (int)ph.coal.k_twiddle_bar.size());
  const double nan = std::numeric_limits<double>::quiet_NaN();
  h->sh_ktw.assign((size_t)total, nan); h->sh_popsize.assign((size_t)total, nan); h->sh_tsop.assign((size_t)total, nan); h->sh_nact.assign((size_t)total, -1);
  for (const PartHost& ph : h->parts) {
    const HostCoalPart& c = ph.coal;
    for (size_t w = 0; w < c.k_twiddle_bar.size(); ++w) {
      const size_t i = (size_t)c.cell_first + w;
      if (ph.includes_run_root && h->sh_nact[i] >= 0) continue;   // (cells the root part appended carry its own values: never over another part's)
      h->sh_ktw[i] = c.k_twiddle_bar[w]; h->sh_popsize[i] = c.popsize_bar[w]; h->sh_tsop[i] = c.t_step / c.popsize_bar[w]; h->sh_nact[i] = c.num_active_parts[w];
    }
  }
  HIP_TRY(h->d_sh_ktw.upload(h->sh_ktw.data(), h->sh_ktw.size())); HIP_TRY(h->d_sh_tsop.upload(h->sh_tsop.data(), h->sh_tsop.size())); HIP_TRY(h->d_sh_nact.upload(h->sh_nact.data(), h->sh_nact.size()));
  h->shared_dev = SharedCells{h->d_sh_ktw.p, h->d_sh_tsop.p, h->d_sh_nact.p, total};
  h->grid_mirrors_on_device = false;
  return EMAT_OK;
}

// Encode all parts and push them to the device.
emat_status materialize(emat_backend* h) {
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (h->slabs_on_device) return EMAT_OK;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  if (!h->have_ref || !h->have_evo) return fail(h, EMAT_ERR_STATE, "set_ref_sequence and set_evo must precede running");
  if (!h->have_coal) return fail(h, EMAT_ERR_STATE, "emat_build_coalescent_parts must precede running");
  const int trace_cap = h->cfg.trace_moves > 0 ? h->cfg.trace_moves : 0;
  const bool may_keep = h->derived_valid;   // (a model or reference sequence set since then invalidates what finish_pass kept)
  uint64_t off = 0; h->max_slab_bytes = 0; h->persistent_bytes.assign(h->parts.size(), 0); h->prefix_bytes.assign(h->parts.size(), 0);
  std::vector<SlabGeo> geo(h->parts.size());
  for (size_t p = 0; p < h->parts.size(); ++p) {
    PartHost& ph = h->parts[p];
    // encode_slab writes 16-bit list counts: what reaches it was accepted at upload (<= k_max_list_upload) or came back from the
    // device (whose lists stop at k_max_list_len: k_part_list_limit) -- checked here all the same, so that no count is ever cut
    { emat_flat_tree v = ph.tree.view(); const std::string lim = flat_tree_list_limit(v, (int32_t)k_max_list_len); if (!lim.empty()) return fail(h, EMAT_ERR_CAPACITY, "part " + std::to_string(p) + ": " + lim); }
    geo[p] = slab_geometry(h, ph.tree.num_nodes(), ph.tree.num_muts(), heap_content_bytes(ph.tree), (int)ph.coal.k_bar_p.size(), ph.includes_run_root, ph.space_boost, ph.cell_boost);
    place_slab(h, p, geo[p], off, heap_content_bytes(ph.tree));
  }
  HIP_TRY(h->h_slabs.resize(off)); h->slab_bytes_total = off;
  std::vector<uint64_t> offs(h->parts.size());
  parallel_for((int)h->parts.size(), [&](int p) {
    PartHost& ph = h->parts[p];
    offs[p] = ph.slab_off;
    encode_slab(*h, ph, h->h_slabs.data() + ph.slab_off, ph.slab_bytes, geo[p].heap, geo[p].scratch, geo[p].cell_cap, trace_cap);
    // carry the statistics over re-materialisations
    SlabHeader* H = (SlabHeader*)(h->h_slabs.data() + ph.slab_off);
    H->moves_done = ph.stats.moves_done; for (int k = 0; k < 5; ++k) { H->proposed[k] = ph.stats.proposed[k]; H->accepted[k] = ph.stats.accepted[k]; }
    H->alg_bytes = ph.stats.algorithmic_bytes; H->alg_write16 = (uint32_t)(ph.stats.algorithmic_write_bytes / 16); H->device_ticks = ph.stats.device_ticks;
    if (may_keep && ph.derived_kept && (int)ph.kept_lambda.size() == H->n_nodes) {   // a re-materialisation in the middle of a pass: the maintained values go on as they are (PartHost)
      NodeRec* N = (NodeRec*)(h->h_slabs.data() + ph.slab_off + H->off_nodes);
      for (int i = 0; i < H->n_nodes; ++i) { N[i].lambda = ph.kept_lambda[(size_t)i]; N[i].n_missing = ph.kept_n_missing[(size_t)i]; }
      H->log_G = ph.kept_log_G; H->log_aug_prior = ph.kept_log_aug_prior;
    } else ph.derived_kept = false;
  });
  bool all_kept = !h->parts.empty();
  for (PartHost& ph : h->parts) { all_kept = all_kept && ph.derived_kept; ph.derived_kept = false; }
  assign_size_classes(h);
  h->order_valid = false;
  { emat_status st = upload_shared_cells(h); if (st) return st; }
  HIP_TRY(h->d_slabs.upload(h->h_slabs.data(), h->h_slabs.size()));
  HIP_TRY(h->d_slab_off.upload(offs.data(), offs.size()));
  { std::vector<int64_t> z((2 + 2 * k_ticket_log) * h->parts.size(), 0); HIP_TRY(h->d_part_ticks.upload(z.data(), z.size())); }   // + entry / exit ticks of the first k_ticket_log tickets of every part
  { std::vector<int32_t> z(h->parts.size(), 0); HIP_TRY(h->d_part_status.upload(z.data(), z.size())); }
  h->slabs_on_device = true; h->host_slabs_current = true; h->derived_valid = all_kept;
  return EMAT_OK;
}

// How this device deals the workgroups of a launch to its XCDs: the number of XCDs if workgroup b runs on XCD b mod that number
// (what MI300-class parts do in their default partition mode), 1 if there is one XCD, 0 if the pattern is anything else.  Asked
// once per device and process; the kernels' own per-ticket check (run_moves_body) is what a pass relies on.
// (On the caller's stream, never the null stream: once a process has used the null stream it holds one of the runtime's four hardware
// queues for good, and with two handles alive -- bench.py's engine and its run driver -- a main stream then shares a queue with a
// side stream and waits 15-19 ms per pass for that stream's kernel: whole cycles of 48 ms instead of 29.)
int probe_xcc_dealing(int device, hipStream_t stream) {
  static std::mutex mu; static int known[64]; static bool asked[64] = {};
  std::lock_guard<std::mutex> lock(mu);
  if (device < 0 || device >= 64) return 0;
  if (asked[device]) return known[device];
  asked[device] = true; known[device] = 0;
  const int nb = 2048;
  int32_t* d = nullptr; std::vector<int32_t> x((size_t)nb, -1);
  if (hipMalloc((void**)&d, nb * sizeof(int32_t)) != hipSuccess) return 0;
  hipLaunchKernelGGL(k_probe_xcc, dim3(nb), dim3(64), 0, stream, d);
  const bool ok = hipMemcpyAsync(x.data(), d, nb * sizeof(int32_t), hipMemcpyDeviceToHost, stream) == hipSuccess && hipStreamSynchronize(stream) == hipSuccess;
  (void)hipFree(d);
  if (!ok) { (void)hipGetLastError(); return 0; }
  int nx = 0; for (int v : x) nx = std::max(nx, v + 1);
  if (nx <= 0 || nx > 16) return 0;
  for (int b = 0; b < nb; ++b) if (x[(size_t)b] != (x[0] + b) % nx) nx = -1;
  if (verbose_reports()) fprintf(stderr, "[emat] device %d: workgroups are dealt to %s\n", device, nx > 0 ? (std::to_string(nx) + " XCD(s) round robin").c_str() : "the XCDs in no pattern the tickets can rely on: full releases");
  known[device] = nx > 0 ? nx : 0;
  return known[device];
}

emat_status launch_recalc(emat_backend* h) {
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  // the kernel rewrites lambda, missing-site counts and the totals of EVERY part, the parts of the side classes included: their
  // launches of the last pass must have ended, and their next ones must follow this one (every caller -- emat_recalc_derived
  // directly after emat_run_* as well -- goes through here)
  st = join_side_classes(h); if (st) return st;
  h->sides_must_fork = true;
  KernelArgs a = make_args(h);
  hipLaunchKernelGGL(k_recalc_derived, dim3((unsigned)h->parts.size()), dim3(k_wave), 0, h->stream, a);
  HIP_TRY(hipGetLastError());
  h->derived_valid = true; h->host_slabs_current = false; h->headers_current = false;
  return EMAT_OK;
}

// Launch order: one workgroup per part, largest persistent state first.  The hardware dispatcher hands workgroups to
// free slots in index order, which makes it a longest-processing-time-first list scheduler (size and duration of a
// part correlate at 0.85); packing lists of parts per workgroup on the host was measured slower twice.  The order only
// decides WHEN a part's chain runs; every chain is independent (own RNG stream, own slab).
// (`queued`: the copy goes onto the engine's stream behind whatever is running there -- for a caller that knows nothing in flight
// reads the order -- instead of waiting for the stream first.)
emat_status build_order(emat_backend* h, bool queued = false) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  const int n = (int)h->parts.size();
  // by class, then by persistent size, largest first, ties in index order: a stable radix sort (three passes of 12 bits) on
  // (class, ~size); the comparison sort this replaces took half a millisecond of every cycle at 13 000 parts
  std::vector<uint64_t> key((size_t)n), key2((size_t)n);
  std::vector<int32_t> order((size_t)n), order2((size_t)n);
  std::iota(order.begin(), order.end(), 0);
  auto sort_key = [&](int i) { return ((uint64_t)h->class_of[(size_t)i] << 32) | (uint64_t)(0xFFFFFFFFu - h->persistent_bytes[(size_t)i]); };   // 36 bits (k_max_classes <= 16)
  for (int i = 0; i < n; ++i) key[(size_t)i] = sort_key(i);
  for (int pass = 0; pass < 3; ++pass) {
    const int shift = 12 * pass;
    uint32_t count[4097] = {};
    for (int i = 0; i < n; ++i) ++count[((key[(size_t)i] >> shift) & 0xFFFu) + 1];
    for (int d = 0; d < 4096; ++d) count[d + 1] += count[d];
    for (int i = 0; i < n; ++i) { const uint32_t at = count[(key[(size_t)i] >> shift) & 0xFFFu]++; key2[at] = key[(size_t)i]; order2[at] = order[(size_t)i]; }
    key.swap(key2); order.swap(order2);
  }
  h->h_order = order;
  // (from the pageable member: measured in round 6 against a page-locked staging copy, which gained nothing here and made the
  // like copies of emat_tree_partition WAIT for the stream; h_order lives as long as the handle)
  if (queued) { HIP_TRY(h->d_order.alloc(order.size())); HIP_TRY(hipMemcpyAsync(h->d_order.p, h->h_order.data(), order.size() * sizeof(int32_t), hipMemcpyHostToDevice, h->stream)); }
  else { HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(h->d_order.upload(order.data(), order.size())); }
  h->order_valid = true;
  return EMAT_OK;
}

emat_status launch_moves(emat_backend* h, int64_t per_part, int64_t extra0, const std::vector<int64_t>* counts = nullptr, int32_t one_more_below = 0) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  if (h->parts.empty()) return fail(h, EMAT_ERR_STATE, "no parts uploaded");
  if (h->fatal_status != EMAT_OK) return fail(h, h->fatal_status, h->fatal_message);
  HostLaps laps;
  emat_status st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  laps.mark("launch_moves: 1 model + materialize");
  if (!h->derived_valid) { st = join_side_classes(h); if (st) return st; h->sides_must_fork = true; st = launch_recalc(h); if (st) return st; }
  laps.mark("launch_moves: 2 launch_recalc");
  const uint32_t lds_scratch = h->cfg.use_lds ? h->cfg_lds_scratch : 0u;
  // the dynamic block: the slab image beyond its header + the arena; tables, context and the header image are static LDS
  auto shmem_for = [&](uint32_t slab_area) { return (size_t)(slab_area > (uint32_t)sizeof(SlabHeader) ? slab_area - (uint32_t)sizeof(SlabHeader) : 0u) + lds_scratch; };
  for (int c = 0; c < h->num_classes; ++c)
    if (shmem_for(h->class_lds[c]) + k_lds_static_bytes + sizeof(SlabHeader) > 160 * 1024) return fail(h, EMAT_ERR_CAPACITY, "LDS request exceeds 160 KiB: lower EMAT_LDS_MAX or disable use_lds");
  if (!h->order_valid) { st = build_order(h); if (st) return st; }
  laps.mark("launch_moves: 3 build_order (sort, wait for the stream, H2D)");
  HIP_TRY(h->d_snaps.alloc(h->d_slabs.n));   // (as roomy as the slabs: grows when they do)
  laps.mark("launch_moves: 4 snapshot allocation");
  KernelArgs a = make_args(h);
  a.moves_per_part = per_part; a.extra_moves_part0 = extra0; a.one_more_below = one_more_below;
  a.lds_scratch_bytes = lds_scratch; a.snaps = h->d_snaps.p;
  if (counts) { st = join_side_classes(h); if (st) return st; h->sides_must_fork = true; HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(h->d_moves_for_part.upload(counts->data(), counts->size())); a.moves_for_part = h->d_moves_for_part.p; }
  else { h->expected_moves.resize(h->parts.size(), 0); for (size_t p = 0; p < h->parts.size(); ++p) h->expected_moves[p] += per_part + (p == 0 ? extra0 : 0) + ((int64_t)p < one_more_below ? 1 : 0); }
  h->pass_pending = true;
  h->last_launch_uniform = counts == nullptr;
  HIP_TRY(hipEventRecord(h->ev_start, h->stream));
  {
    const size_t sh_max = shmem_for(*std::max_element(h->class_lds, h->class_lds + h->num_classes));
    if (sh_max > 48 * 1024) {
      // the attribute belongs to the kernel on this device, not to the handle: raised when a request exceeds what any handle of
      // the process has asked for so far (two driver calls that used to sit in front of every pass)
      static std::mutex mu; static size_t granted[64] = {};
      std::lock_guard<std::mutex> lock(mu);
      size_t& g = granted[h->cfg.device & 63];
      if (sh_max > g) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_run_moves, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_max));
        HIP_TRY(hipFuncSetAttribute((const void*)k_run_moves_side, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh_max));
        g = sh_max;
      }
    }
    const int main_class = h->num_classes - 1;   // the last class holds the bulk of the parts
    // Order matters: a side class holds few, large workgroups (tens of KB of LDS each), which can only be placed while
    // the CUs are not yet packed with the 10 KB workgroups of the main class -- arriving second they would wait for
    // several neighbours to finish (measured: the root part started 10-15 ms into the pass).  So the side classes are
    // launched first, largest area first, each on a stream of its own, the main class last on the engine's stream.  The
    // engine's stream does not wait for them here (join_side_classes): a side class only ever follows its own previous launch
    // -- the parts of different classes have nothing to do with each other -- so that in back-to-back passes the main class of the
    // next pass starts when the main class of this one ends, and the side launches of the next pass, queued behind their
    // predecessors, find their LDS while this pass's main class drains.  They wait for the engine's stream only when it did
    // something since the last pass was checked (uploads, recalculation: sides_must_fork).  The timing events bracket the main class.
    // Tickets pay when there are more parts than wave slots (4 waves x 4 SIMDs per CU): with fewer, every part has a slot to
    // itself from the start and a second ticket could only wait behind its first while holding another slot.
    const int main_count = h->class_begin[h->num_classes] - h->class_begin[h->num_classes - 1];
    const int chunks = (counts == nullptr && per_part >= 4 * h->cfg_chunks && (h->cfg_chunks_forced || main_count > 4 * EMAT_WAVES_PER_EU * h->num_cus)) ? h->cfg_chunks : 1;
    if (chunks > 1) { HIP_TRY(h->d_chunk_done.alloc(h->parts.size() + 130)); HIP_TRY(hipMemsetAsync(h->d_chunk_done.p, 0, (h->parts.size() + 130) * sizeof(int32_t), h->stream)); }   // + 64 eight-byte counters of waiting time
    // (the counter the side workgroups of a synchronised pass report to, zeroed before the event the side streams wait for)
    int side_wgs = 0; for (int c = 0; c + 1 < h->num_classes; ++c) side_wgs += h->class_begin[c + 1] - h->class_begin[c];
    const bool hold_main = h->sides_must_fork && side_wgs > 0;
    if (hold_main) { HIP_TRY(h->d_side_started.alloc(1)); HIP_TRY(hipMemsetAsync(h->d_side_started.p, 0, sizeof(int32_t), h->stream)); }
    HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
    for (int c = 0; c < h->num_classes; ++c) {
      const int lo = h->class_begin[c], cnt = h->class_begin[c + 1] - lo;
      if (cnt <= 0) continue;
      KernelArgs b = a;
      b.order = a.order + lo; b.lds_slab_bytes = h->class_lds[c];
      // (only the main class: a side class has fewer workgroups than the device has room for, so all its tickets would be
      // resident at once and the waiting ones would sit on tens of KB of LDS each -- measured: cycles of 72 ms instead of 37)
      b.taper = h->cfg_taper ? 1 : 0;
      for (int k = 0; k < 8; ++k) b.cum_w[k] = 0;
      if (h->cfg_taper && chunks == 4) { b.cum_w[0] = 10; b.cum_w[1] = 16; b.cum_w[2] = 19; b.cum_w[3] = 20; }   // 10 : 6 : 3 : 1 (measured best of the ratios tried)
      if (const char* e = h->cfg_ticket_weights.empty() ? nullptr : h->cfg_ticket_weights.c_str()) { int acc = 0, k = 0; for (const char* q = e; *q && k < 8;) { acc += std::max(1, atoi(q)); b.cum_w[k++] = acc; while (*q && *q != ',') ++q; if (*q == ',') ++q; } if (k != chunks) for (int j = 0; j < 8; ++j) b.cum_w[j] = 0; }
      b.chunks = c == main_class ? chunks : 1; b.class_count = cnt; b.class_stride = (cnt + 7) & ~7; b.chunk_done = h->d_chunk_done.p;
      // Testing knobs.  EMAT_TICKET_XCD_SPREAD=1 makes the stride odd, so that consecutive tickets of a part land on DIFFERENT XCDs
      // (workgroups go round the eight XCDs by index): the hand-over then has to cross L2s, which the default placement avoids
      // but does not rely on -- what a ticket hands over goes out through agent-scope write-through stores and the next ticket
      // acquires at agent scope.  EMAT_TICKET_RELEASE=full takes the plain agent-scope release for every ticket.
      if (h->cfg_ticket_spread) b.class_stride = cnt | 1;
      // the cheap hand-over only where a part's tickets share an XCD: workgroups are dealt to the XCDs round robin (verified on this
      // device by probe_xcc_dealing, and re-checked by every ticket), so the stride must be a multiple of their number
      b.single_below = (c == main_class && chunks > 1) ? std::min(h->cfg_single_ticket_parts, cnt) : 0;
      b.full_release = (h->cfg_ticket_full_release || h->xcc_count <= 0 || b.class_stride % h->xcc_count != 0) ? 1 : 0;
      const unsigned grid = b.chunks > 1 ? (unsigned)(b.chunks * b.class_stride) : (unsigned)cnt;
      const bool side = c != main_class;
      hipStream_t sm = side ? h->class_stream[c + 1] : h->stream;
      if (side && h->sides_must_fork) HIP_TRY(hipStreamWaitEvent(sm, h->ev_fork, 0));
      b.side_started = (side && hold_main) ? h->d_side_started.p : nullptr;
      if (c == main_class && hold_main) { hipLaunchKernelGGL(k_wait_side_start, dim3(1), dim3(64), 0, sm, h->d_side_started.p, side_wgs); HIP_TRY(hipGetLastError()); }
      if (c == main_class) hipLaunchKernelGGL(k_run_moves, dim3(grid), dim3(k_wave), shmem_for(h->class_lds[c]), sm, b);
      else hipLaunchKernelGGL(k_run_moves_side, dim3(grid), dim3(k_wave), shmem_for(h->class_lds[c]), sm, b);
      HIP_TRY(hipGetLastError());
      if (side) { HIP_TRY(hipEventRecord(h->ev_join[c + 1], sm)); h->sides_in_flight = true; }
    }
    h->sides_must_fork = false;
  }
  HIP_TRY(hipEventRecord(h->ev_stop, h->stream));
  laps.mark("launch_moves: 5 expected moves, events, memsets, launches");
  h->host_slabs_current = false; h->headers_current = false;
  return EMAT_OK;
}

}  // namespace

// =================================================================================================
// C-ABI
// =================================================================================================
namespace { emat_status gt_finish_gather(emat_backend* h); }   // (emat_gtree_host.hpp, included at the end of this file)
extern "C" {

emat_status emat_backend_create(const emat_config* cfg, emat_backend** out) {
  if (!cfg || !out || cfg->num_sites <= 0) return EMAT_ERR_INVALID_ARGUMENT;
  if (cfg->device == -1) {   // host-only handle: can stage data and build coalescent parts, can never run a move
    auto h = std::make_unique<emat_backend>();
    h->cfg = *cfg; h->L = cfg->num_sites; h->host_only = true;
    *out = h.release();
    return EMAT_OK;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return EMAT_ERR_NO_DEVICE;
  if (cfg->device < 0 || cfg->device >= ndev) return EMAT_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(cfg->device) != hipSuccess) return EMAT_ERR_HIP;
  auto h = std::make_unique<emat_backend>();
  h->cfg = *cfg; h->L = cfg->num_sites;
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return EMAT_ERR_HIP; h->num_cus = prop.multiProcessorCount; }
  if (hipStreamCreate(&h->stream) != hipSuccess) return EMAT_ERR_HIP;
  h->xcc_count = probe_xcc_dealing(cfg->device, h->stream);
  for (hipEvent_t* e : {&h->ev_start, &h->ev_stop}) if (hipEventCreate(e) != hipSuccess) return EMAT_ERR_HIP;
  if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) return EMAT_ERR_HIP;
  for (int c = 1; c < emat_backend::k_max_classes; ++c) {
    if (!(h->class_stream[c] = side_stream(cfg->device, c - 1))) return EMAT_ERR_HIP;
    if (hipEventCreateWithFlags(&h->ev_join[c], hipEventDisableTiming) != hipSuccess) return EMAT_ERR_HIP;
  }
  *out = h.release();
  return EMAT_OK;
}
emat_status emat_backend_destroy(emat_backend* h) {
  if (!h) return EMAT_OK;
  if (h->host_only) { delete h; return EMAT_OK; }
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) { (void)hipStreamSynchronize(h->stream); (void)hipStreamDestroy(h->stream); }
  for (int c = 1; c < emat_backend::k_max_classes; ++c) {
    if (h->class_stream[c]) (void)hipStreamSynchronize(h->class_stream[c]);   // shared with the other handles of the device: never destroyed
    if (h->ev_join[c]) (void)hipEventDestroy(h->ev_join[c]);
  }
  for (hipEvent_t e : {h->ev_start, h->ev_stop, h->ev_fork}) if (e) (void)hipEventDestroy(e);
  HostSpans::instance().report();
  delete h;
  return EMAT_OK;
}
/* Tuning and test options of one handle (header: emat_set_option).  Until round 4 these were environment variables read at
 * emat_backend_create; a process that embeds the library sets them per handle instead, and nothing in its environment reaches them. */
emat_status emat_set_option(emat_backend* h, const char* key, const char* value) {
  if (!h || !key || !value) return EMAT_ERR_INVALID_ARGUMENT;
  if (strcmp(key, "debug_fail_gather") == 0) { h->cfg_debug_fail_gather = atoi(value) != 0; return EMAT_OK; }   // (a test hook that is armed in the middle of a run)
  if (h->slabs_on_device) return fail(h, EMAT_ERR_STATE, "emat_set_option: options are set before the first launch");
  const std::string k(key);
  const char* e = value;
  if (k == "slack") h->cfg.slab_slack = atof(e);
  else if (k == "heap_per_node") h->cfg_heap_per_node = atof(e);
  else if (k == "lds_scratch") h->cfg_lds_scratch = (uint32_t)atoi(e) & ~15u;
  else if (k == "lds_classes") {   // e.g. "60,90,99,100"
    h->cfg_class_pct.clear();
    for (const char* q = e; *q;) { h->cfg_class_pct.push_back(std::max(1, std::min(100, atoi(q)))); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
  }
  else if (k == "lds_max") h->cfg_lds_max = (uint32_t)atoi(e) & ~511u;
  else if (k == "giants") h->cfg_giants = atoi(e) != 0;
  else if (k == "side_arena") h->cfg_side_arena = (uint32_t)atoi(e);
  else if (k == "tree_host_coalescent") h->cfg_gt_host_coal = atoi(e) != 0;
  else if (k == "ticket_taper") h->cfg_taper = atoi(e) != 0;
  else if (k == "chunks") { h->cfg_chunks = std::max(1, std::min(64, atoi(e))); h->cfg_chunks_forced = true; }
  else if (k == "ticket_xcd_spread") h->cfg_ticket_spread = atoi(e) != 0;
  else if (k == "ticket_release") h->cfg_ticket_full_release = strcmp(e, "full") == 0;
  else if (k == "ticket_weights") h->cfg_ticket_weights = e;
  else if (k == "single_ticket_parts") h->cfg_single_ticket_parts = std::max(0, atoi(e));
  else if (k == "parts_per_cu") h->cfg_parts_per_cu = std::max(0, std::min(4 * EMAT_WAVES_PER_EU, atoi(e)));
  else if (k == "order_by_time") h->cfg_order_by_time = atoi(e) != 0;
  else if (k == "build_blocks") h->cfg_build_blocks = std::max(0, atoi(e));
  else if (k == "tree_tight") h->cfg_tree_tight = atoi(e) != 0;
  else if (k == "no_uniform_sites") { h->cfg_no_uniform_sites = atoi(e) != 0; if (h->have_evo && h->have_ref) refresh_ref_derived(h); }
  else if (k == "debug_fail_gather") h->cfg_debug_fail_gather = atoi(e) != 0;
  else if (k == "fn_min_lists") h->cfg_fn_min_lists = (unsigned)std::max(0, atoi(e));
  else if (k == "phase_extra") h->cfg_phase_extra = atoi(e) != 0;
  else return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_set_option: unknown option '" + k + "'");
  return EMAT_OK;
}
/* Size of the host thread pool of this library (per process; takes effect if called before the first parallel loop; 0 = default). */
emat_status emat_set_host_threads(int32_t n) { if (n < 0) return EMAT_ERR_INVALID_ARGUMENT; host_threads_override() = n; return EMAT_OK; }
const char* emat_last_error(const emat_backend* h) { return h ? h->last_error.c_str() : "null backend"; }
#ifndef EMAT_BUILD_ID
#define EMAT_BUILD_ID "unstamped"
#endif
const char* emat_build_id(void) { return EMAT_BUILD_ID; }

emat_status emat_set_ref_sequence(emat_backend* h, const uint8_t* ref, int32_t num_sites) {
  if (!h || !ref || num_sites != h->L) return EMAT_ERR_INVALID_ARGUMENT;
  for (int l = 0; l < num_sites; ++l) if (ref[l] > 3) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "reference sequence states must be 0..3");
  h->ref.assign(ref, ref + num_sites); h->have_ref = true;
  if (h->have_evo) refresh_ref_derived(h);
  h->derived_valid = false;
  return EMAT_OK;
}
emat_status emat_set_evo(emat_backend* h, int32_t P, const double* mu, const double* pi, const double* q, const double* nu_l, const int32_t* pfs) {
  if (!h || P <= 0 || P > 255 || !mu || !pi || !q || !nu_l || !pfs) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->have_ref) return fail(h, EMAT_ERR_STATE, "emat_set_ref_sequence must precede emat_set_evo");
  h->partition_for_site.resize(h->L);
  for (int l = 0; l < h->L; ++l) { if (pfs[l] < 0 || pfs[l] >= P) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "partition_for_site out of range"); h->partition_for_site[l] = (uint8_t)pfs[l]; }
  h->num_partitions = P;
  h->mu.assign(mu, mu + P); h->pi.assign(pi, pi + 4 * P); h->q.assign(q, q + 16 * P); h->nu_l.assign(nu_l, nu_l + h->L);
  h->have_evo = true;
  refresh_ref_derived(h);
  h->derived_valid = false;   // Subrun::set_evo invalidates derived quantities (subrun.h:29-30)
  return EMAT_OK;
}
emat_status emat_set_flags(emat_backend* h, double t_max_tip, int32_t only_displacing_inner_nodes, int32_t topology_moves_enabled) {
  if (!h) return EMAT_ERR_INVALID_ARGUMENT;
  h->flags.t_max_tip = t_max_tip; h->flags.only_displacing_inner_nodes = only_displacing_inner_nodes; h->flags.topology_moves_enabled = topology_moves_enabled;
  return EMAT_OK;
}

emat_status emat_begin_upload(emat_backend* h, int32_t num_parts) {
  if (!h || num_parts <= 0) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->cfg.max_parts > 0 && num_parts > h->cfg.max_parts) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "more parts than cfg.max_parts");
  if (h->stream) { (void)join_side_classes(h); h->sides_must_fork = true; (void)hipStreamSynchronize(h->stream); }   // (a pass still in flight is discarded with the parts, but not written over)
  h->coal_builder.reset();
  h->fatal_status = EMAT_OK; h->fatal_message.clear(); h->pass_pending = false;
  h->parts.clear(); h->parts.resize(num_parts);
  h->expected_moves.assign((size_t)num_parts, 0);
  h->uploads_expected = num_parts; h->root_part = -1; h->gt.parts_live = false;
  h->slabs_on_device = false; h->host_slabs_current = false; h->headers_current = false; h->have_coal = false; h->derived_valid = false;
  return EMAT_OK;
}
emat_status emat_part_upload(emat_backend* h, int32_t part_id, const emat_flat_tree* subtree, int32_t includes_run_root, uint64_t seed) {
  if (!h || !subtree || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->uploads_expected <= 0) return fail(h, EMAT_ERR_STATE, "emat_begin_upload must precede emat_part_upload");
  std::string msg = validate_flat_tree(*subtree, h->L);
  if (!msg.empty()) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "part " + std::to_string(part_id) + ": " + msg);
  msg = flat_tree_list_limit(*subtree, (int32_t)k_max_list_upload);
  if (!msg.empty()) return fail(h, EMAT_ERR_CAPACITY, "part " + std::to_string(part_id) + ": " + msg);
  PartHost& ph = h->parts[part_id];
  if (ph.uploaded) return fail(h, EMAT_ERR_STATE, "part uploaded twice");
  ph.tree = FlatTree::from_view(*subtree); ph.n_nodes = subtree->num_nodes;
  ph.includes_run_root = includes_run_root != 0;
  ph.rng.key = seed; ph.rng.counter = 0; ph.rng.spare = 0; ph.rng.has_spare = false;
  ph.uploaded = true; ph.stats = emat_part_stats{}; ph.space_boost = 1.0; ph.cell_boost = 1; ph.trace.clear();
  if (ph.includes_run_root) h->root_part = part_id;
  return EMAT_OK;
}
emat_status emat_end_upload(emat_backend* h) {
  if (!h) return EMAT_ERR_INVALID_ARGUMENT;
  for (auto& ph : h->parts) if (!ph.uploaded) return fail(h, EMAT_ERR_STATE, "emat_end_upload before every part was uploaded");
  h->uploads_expected = 0;
  return EMAT_OK;
}

emat_status emat_build_coalescent_parts(emat_backend* h, const emat_pop_model* pm, int32_t root_part_index, double t_step) {
  if (!h || !pm || !(t_step > 0.0)) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->parts.empty() || h->uploads_expected != 0) return fail(h, EMAT_ERR_STATE, "upload parts first");
  if (root_part_index < 0 || root_part_index >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  try {
    emat_status st = pull_from_device(h); if (st) return st;   // the trees (node times) may have moved on the device
    h->pop = HostPopModel::from_c(*pm);
    std::vector<const FlatTree*> trees; std::vector<HostRng*> rngs;
    for (auto& ph : h->parts) { trees.push_back(&ph.tree); rngs.push_back(&ph.rng); }
    auto cps = make_coalescent_parts(trees, root_part_index, h->pop, rngs, t_step);
    for (size_t p = 0; p < h->parts.size(); ++p) h->parts[p].coal = std::move(cps[p]);
  } catch (const std::exception& ex) { return fail(h, EMAT_ERR_INVALID_ARGUMENT, ex.what()); }
  h->have_pop = true; h->have_coal = true; h->model_dirty = true;
  h->slabs_on_device = false; h->host_slabs_current = false; h->headers_current = false; h->derived_valid = false;   // re-encode with the new cell tables
  return EMAT_OK;
}

// ---- staged form of emat_build_coalescent_parts for runs sharded over several GPUs ---------------
emat_status emat_coalescent_begin(emat_backend* h, const emat_pop_model* pm, int32_t root_part_index, double t_step, double* local_t_min, double* local_t_max) {
  if (!h || !pm || !(t_step > 0.0) || !local_t_min || !local_t_max) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->parts.empty() || h->uploads_expected != 0) return fail(h, EMAT_ERR_STATE, "upload parts first");
  if (root_part_index < -1 || root_part_index >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  try {
    emat_status st = pull_from_device(h); if (st) return st;
    h->pop = HostPopModel::from_c(*pm);
    h->coal_builder = std::make_unique<CoalBuilder>();
    h->coal_builder->pop = h->pop; h->coal_builder->t_step = t_step;
    for (size_t p = 0; p < h->parts.size(); ++p) h->coal_builder->add_part(&h->parts[p].tree, &h->parts[p].rng, (int)p == root_part_index);
    h->coal_builder->local_range(*local_t_min, *local_t_max);
  } catch (const std::exception& ex) { return fail(h, EMAT_ERR_INVALID_ARGUMENT, ex.what()); }
  return EMAT_OK;
}
emat_status emat_coalescent_set_range(emat_backend* h, double all_t_min, double all_t_max, int32_t* num_cells) {
  if (!h || !num_cells) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->coal_builder) return fail(h, EMAT_ERR_STATE, "emat_coalescent_begin first");
  *num_cells = h->coal_builder->set_range(all_t_min, all_t_max);
  return EMAT_OK;
}
emat_status emat_coalescent_local_grid(emat_backend* h, double* k_bar, int32_t* num_active) {
  if (!h || !k_bar || !num_active) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->coal_builder) return fail(h, EMAT_ERR_STATE, "emat_coalescent_begin first");
  try {
    std::vector<double> kb; std::vector<int32_t> na;
    h->coal_builder->local_grid(kb, na);
    std::copy(kb.begin(), kb.end(), k_bar); std::copy(na.begin(), na.end(), num_active);
  } catch (const std::exception& ex) { return fail(h, EMAT_ERR_INVALID_ARGUMENT, ex.what()); }
  return EMAT_OK;
}
emat_status emat_coalescent_sample(emat_backend* h, const double* k_bar, const int32_t* num_active, double* k_twiddle_bar_local) {
  if (!h || !k_bar || !num_active || !k_twiddle_bar_local) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->coal_builder) return fail(h, EMAT_ERR_STATE, "emat_coalescent_begin first");
  try {
    const int n = h->coal_builder->num_cells;
    std::vector<double> kt;
    h->coal_builder->sample(std::vector<double>(k_bar, k_bar + n), std::vector<int32_t>(num_active, num_active + n), kt);
    std::copy(kt.begin(), kt.end(), k_twiddle_bar_local);
  } catch (const std::exception& ex) { return fail(h, EMAT_ERR_INVALID_ARGUMENT, ex.what()); }
  return EMAT_OK;
}
emat_status emat_coalescent_finish(emat_backend* h, const double* k_twiddle_bar) {
  if (!h || !k_twiddle_bar) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->coal_builder) return fail(h, EMAT_ERR_STATE, "emat_coalescent_begin first");
  try {
    const int n = h->coal_builder->num_cells;
    auto cps = h->coal_builder->finish(std::vector<double>(k_twiddle_bar, k_twiddle_bar + n));
    for (size_t p = 0; p < h->parts.size(); ++p) h->parts[p].coal = std::move(cps[p]);
  } catch (const std::exception& ex) { return fail(h, EMAT_ERR_INVALID_ARGUMENT, ex.what()); }
  h->coal_builder.reset();
  h->have_pop = true; h->have_coal = true; h->model_dirty = true;
  h->slabs_on_device = false; h->host_slabs_current = false; h->headers_current = false; h->derived_valid = false;
  return EMAT_OK;
}

emat_status emat_run_local_moves(emat_backend* h, int64_t count) {
  if (!h || count < 0) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->parts.empty()) return fail(h, EMAT_ERR_STATE, "no parts uploaded");
  const int64_t P = (int64_t)h->parts.size();
  const int64_t sub = count / P;   // run.cpp:683-689
  return launch_moves(h, sub, count - P * sub);
}
emat_status emat_run_moves_per_part(emat_backend* h, int64_t moves_per_part) {
  if (!h || moves_per_part < 0) return EMAT_ERR_INVALID_ARGUMENT;
  return launch_moves(h, moves_per_part, 0);
}
emat_status emat_run_moves_split(emat_backend* h, int64_t moves_per_part, int64_t extra_moves_part0) {
  if (!h || moves_per_part < 0 || extra_moves_part0 < 0) return EMAT_ERR_INVALID_ARGUMENT;
  return launch_moves(h, moves_per_part, extra_moves_part0);
}
emat_status emat_run_moves_even(emat_backend* h, int64_t moves_per_part, int32_t one_more_below) {
  if (!h || moves_per_part < 0 || one_more_below < 0 || one_more_below > (int64_t)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  return launch_moves(h, moves_per_part, 0, nullptr, one_more_below);
}
emat_status emat_synchronize(emat_backend* h) {
  if (!h) return EMAT_ERR_INVALID_ARGUMENT;
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (h->host_only) return EMAT_OK;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  HIP_TRY(hipStreamSynchronize(h->stream));
  return finish_pass(h);
}
emat_status emat_recalc_derived(emat_backend* h) {
  if (!h) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->parts.empty()) return fail(h, EMAT_ERR_STATE, "no parts uploaded");
  return launch_recalc(h);
}

// The reference's Subrun::check_derived_quantities (subrun.cpp:28-56) for every part, on the device, without the oracle and
// without touching the state: `tol_scale` multiplies the reference's own tolerances (1e-8 per site on lambda_i, 1e-6 on
// log_G, 1e-5 on the augmented coalescent prior; missing-site counts exact).
emat_status emat_check_derived(emat_backend* h, double tol_scale, int32_t* worst_part, double* worst4) {
  if (!h || !(tol_scale > 0.0)) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->parts.empty()) return fail(h, EMAT_ERR_STATE, "no parts uploaded");
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = emat_synchronize(h); if (st) return st;
  if (!h->slabs_on_device || !h->derived_valid) return fail(h, EMAT_ERR_STATE, "emat_check_derived: nothing has been maintained incrementally yet (run moves first)");
  st = sync_model_to_device(h); if (st) return st;
  const size_t n = h->parts.size();
  for (auto& ph : h->parts) if ((uint64_t)ph.tree.num_nodes() * 12u > (uint64_t)ph.scratch_bytes) return fail(h, EMAT_ERR_CAPACITY, "scratch region too small for the check");
  DevBuf<double> d_out; HIP_TRY(d_out.alloc(4 * n));
  KernelArgs a = make_args(h);
  hipLaunchKernelGGL(k_check_derived, dim3((unsigned)n), dim3(k_wave), 0, h->stream, a, d_out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<double> out(4 * n);
  HIP_TRY(hipMemcpy(out.data(), d_out.p, out.size() * sizeof(double), hipMemcpyDeviceToHost));
  const double tol[4] = {1e-8 * tol_scale, 1e-6 * tol_scale, 1e-5 * tol_scale, 0.5};
  static const char* what[4] = {"lambda_i (per site)", "log_G", "log_augmented_coalescent_prior", "num_sites_missing (nodes that differ)"};
  int bad_part = -1, bad_q = -1; double worst_ratio = -1.0; int wp = 0;
  for (size_t p = 0; p < n; ++p) for (int q = 0; q < 4; ++q) {
    const double v = out[4 * p + q], ratio = v / tol[q];
    if (!(ratio <= worst_ratio)) { worst_ratio = ratio; wp = (int)p; }      // NaN counts as worst
    if (!(v < tol[q]) && bad_part < 0) { bad_part = (int)p; bad_q = q; }
  }
  if (worst_part) *worst_part = bad_part >= 0 ? bad_part : wp;
  if (worst4) for (int q = 0; q < 4; ++q) worst4[q] = out[4 * (size_t)(bad_part >= 0 ? bad_part : wp) + q];
  if (bad_part >= 0) {
    char buf[256]; snprintf(buf, sizeof buf, "emat_check_derived: part %d: incremental %s differs from its recomputation by %.3g (tolerance %.3g)", bad_part, what[bad_q], out[4 * (size_t)bad_part + bad_q], tol[bad_q]);
    return fail(h, EMAT_ERR_INTERNAL, buf);
  }
  return EMAT_OK;
}

emat_status emat_get_totals(emat_backend* h, double* log_G, double* log_aug) {
  if (!h) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->parts.empty()) return fail(h, EMAT_ERR_STATE, "no parts uploaded");
  emat_status st;
  if (!h->slabs_on_device || !h->derived_valid) { st = launch_recalc(h); if (st) return st; }
  st = pull_headers(h); if (st) return st;   // 256 bytes per part through a gather kernel; the slabs stay where they are
  double g = 0.0, a = 0.0;
  for (size_t p = 0; p < h->parts.size(); ++p) { const SlabHeader* H = header_of(h, p); g += H->log_G; a += H->log_aug_prior; }   // part order: reproducible
  if (log_G) *log_G = g;
  if (log_aug) *log_aug = a;
  return EMAT_OK;
}
emat_status emat_part_get_sizes(emat_backend* h, int32_t part_id, int32_t* nn, int32_t* nm, int32_t* ni, int32_t* nf) {
  if (!h || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = pull_from_device(h); if (st) return st;
  const FlatTree& t = h->parts[part_id].tree;
  if (nn) *nn = t.num_nodes(); if (nm) *nm = t.num_muts(); if (ni) *ni = t.num_intervals(); if (nf) *nf = t.num_from_states();
  return EMAT_OK;
}
emat_status emat_part_download(emat_backend* h, int32_t part_id, emat_flat_tree* out) {
  if (!h || !out || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = pull_from_device(h); if (st) return st;
  const FlatTree& t = h->parts[part_id].tree;
  const int n = t.num_nodes();
  if (out->num_nodes < n || out->cap_muts < t.num_muts() || out->cap_intervals < t.num_intervals() || out->cap_from_states < t.num_from_states())
    return fail(h, EMAT_ERR_BUFFER_TOO_SMALL, "emat_part_download: output arrays too small (see emat_part_get_sizes)");
  out->num_nodes = n; out->root = t.root;
  std::copy(t.parent.begin(), t.parent.end(), out->parent); std::copy(t.child0.begin(), t.child0.end(), out->child0); std::copy(t.child1.begin(), t.child1.end(), out->child1);
  std::copy(t.t.begin(), t.t.end(), out->t); std::copy(t.t_min.begin(), t.t_min.end(), out->t_min); std::copy(t.t_max.begin(), t.t_max.end(), out->t_max);
  std::copy(t.mut_offset.begin(), t.mut_offset.end(), out->mut_offset); std::copy(t.mut_site.begin(), t.mut_site.end(), out->mut_site);
  std::copy(t.mut_from.begin(), t.mut_from.end(), out->mut_from); std::copy(t.mut_to.begin(), t.mut_to.end(), out->mut_to); std::copy(t.mut_t.begin(), t.mut_t.end(), out->mut_t);
  std::copy(t.miss_offset.begin(), t.miss_offset.end(), out->miss_offset); std::copy(t.miss_start.begin(), t.miss_start.end(), out->miss_start); std::copy(t.miss_end.begin(), t.miss_end.end(), out->miss_end);
  std::copy(t.mfs_offset.begin(), t.mfs_offset.end(), out->mfs_offset); std::copy(t.mfs_site.begin(), t.mfs_site.end(), out->mfs_site); std::copy(t.mfs_state.begin(), t.mfs_state.end(), out->mfs_state);
  return EMAT_OK;
}
emat_status emat_part_get_derived(emat_backend* h, int32_t part_id, double* lambda_i, int32_t* num_missing, double* log_G, double* log_aug) {
  if (!h || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st;
  if (!h->slabs_on_device || !h->derived_valid) { st = launch_recalc(h); if (st) return st; }
  st = pull_from_device(h); if (st) return st;
  const uint8_t* slab = h->h_slabs.data() + h->parts[part_id].slab_off;
  const SlabHeader* H = (const SlabHeader*)slab; const NodeRec* N = (const NodeRec*)(slab + H->off_nodes);
  for (int i = 0; i < H->n_nodes; ++i) { if (lambda_i) lambda_i[i] = N[i].lambda; if (num_missing) num_missing[i] = N[i].n_missing; }
  if (log_G) *log_G = H->log_G;
  if (log_aug) *log_aug = H->log_aug_prior;
  return EMAT_OK;
}
emat_status emat_part_get_state_frequencies(emat_backend* h, int32_t part_id, int32_t* num_partitions, int32_t* counts) {
  if (!h || !num_partitions || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->have_ref || !h->have_evo) return fail(h, EMAT_ERR_STATE, "set_ref_sequence and set_evo first");
  if (h->gt.gather_pending) { emat_status st = gt_finish_gather(h); if (st) return st; }   // (a gather on its way may still move the reference sequence with the root's)
  const int P = h->num_partitions;
  if (*num_partitions < P || !counts) { *num_partitions = P; return fail(h, EMAT_ERR_BUFFER_TOO_SMALL, "emat_part_get_state_frequencies: room for fewer site partitions than the model has"); }
  *num_partitions = P;
  std::copy(h->ref_freqs.begin(), h->ref_freqs.begin() + (size_t)P * 4, counts);   // the host's copy of the table the kernels read (d_ref_freqs)
  return EMAT_OK;
}
emat_status emat_part_get_coalescent(emat_backend* h, int32_t part_id, int32_t* num_cells, double* k_bar_p, double* k_tw_p, double* k_tw,
                                     double* popsize_bar, int32_t* num_active, double* t_ref, double* t_step) {
  if (!h || !num_cells || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->have_coal) return fail(h, EMAT_ERR_STATE, "no coalescent parts built");
  emat_status st = pull_from_device(h); if (st) return st;
  const HostCoalPart& cp = h->parts[part_id].coal;
  const int n = cp.n_cells_total;
  if (*num_cells < n) { *num_cells = n; return fail(h, EMAT_ERR_BUFFER_TOO_SMALL, "emat_part_get_coalescent: arrays too small"); }
  *num_cells = n;
  // expand the window back to the logical vectors of the reference (zeros outside the window; the three
  // shared vectors are only known inside it)
  for (int i = 0; i < n; ++i) {
    int w = i - cp.cell_first; bool in = w >= 0 && w < (int)cp.k_bar_p.size();
    if (k_bar_p) k_bar_p[i] = in ? cp.k_bar_p[w] : 0.0;
    if (k_tw_p) k_tw_p[i] = in ? cp.k_twiddle_bar_p[w] : 0.0;
    if (k_tw) k_tw[i] = in ? cp.k_twiddle_bar[w] : __builtin_nan("");
    if (popsize_bar) popsize_bar[i] = in ? cp.popsize_bar[w] : __builtin_nan("");
    if (num_active) num_active[i] = in ? cp.num_active_parts[w] : -1;
  }
  if (t_ref) *t_ref = cp.t_ref;
  if (t_step) *t_step = cp.t_step;
  return EMAT_OK;
}
emat_status emat_debug_slab_layout(emat_backend* h, int32_t part_id, uint32_t* out8) {
  if (!h || !out8 || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->have_coal) return fail(h, EMAT_ERR_STATE, "no coalescent parts built");
  const PartHost& ph = h->parts[part_id];
  const int trace_cap = h->cfg.trace_moves > 0 ? h->cfg.trace_moves : 0;
  const uint32_t content = heap_content_bytes(ph.tree);
  const SlabGeo g = slab_geometry(h, ph.tree.num_nodes(), ph.tree.num_muts(), content, (int)ph.coal.k_bar_p.size(), ph.includes_run_root, ph.space_boost, ph.cell_boost);
  out8[0] = (uint32_t)sizeof(SlabHeader); out8[1] = (uint32_t)ph.tree.num_nodes() * (uint32_t)sizeof(NodeRec); out8[2] = a16((uint32_t)g.cell_cap * cell_bytes_for(ph.includes_run_root));
  out8[3] = a16((uint32_t)trace_cap * 32u); out8[4] = content; out8[5] = g.heap; out8[6] = g.scratch; out8[7] = (uint32_t)g.cell_cap;
  return EMAT_OK;
}
emat_status emat_part_get_rng(emat_backend* h, int32_t part_id, uint64_t* key, uint64_t* counter, uint64_t* spare, int32_t* has_spare) {
  if (!h || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only || !h->slabs_on_device) return fail(h, EMAT_ERR_STATE, "emat_part_get_rng: the parts are not on a device");
  emat_status st = pull_headers(h); if (st) return st;
  const SlabHeader* H = header_of(h, (size_t)part_id);
  if (key) *key = H->rng_key;
  if (counter) *counter = H->rng_counter;
  if (spare) *spare = H->rng_spare;
  if (has_spare) *has_spare = (int32_t)H->rng_has_spare;
  return EMAT_OK;
}
emat_status emat_part_get_stats(emat_backend* h, int32_t part_id, emat_part_stats* out) {
  if (!h || !out || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = pull_headers(h); if (st) return st;
  *out = h->parts[part_id].stats;
  if (h->slabs_on_device && !h->host_only) {
    const SlabHeader* H = header_of(h, (size_t)part_id);
    out->status = H->status; out->num_nodes = H->n_nodes; out->moves_done = H->moves_done;
    for (int k = 0; k < 5; ++k) { out->proposed[k] = H->proposed[k]; out->accepted[k] = H->accepted[k]; }
    out->algorithmic_bytes = H->alg_bytes; out->algorithmic_write_bytes = 16 * (int64_t)H->alg_write16; out->rng_draws = (int64_t)H->rng_counter; out->device_ticks = H->device_ticks;
    if (H->status != 0) h->set_error("part " + std::to_string(part_id) + " stopped with status " + std::to_string(H->status) + " at device line " + std::to_string(H->fail_line));
  }
  return EMAT_OK;
}
emat_status emat_part_get_trace(emat_backend* h, int32_t part_id, int32_t* num_moves, double* trace) {
  if (!h || !num_moves || !trace || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = pull_from_device(h); if (st) return st;
  if (!h->slabs_on_device) { *num_moves = 0; return EMAT_OK; }
  const uint8_t* slab = h->h_slabs.data() + h->parts[part_id].slab_off;
  const SlabHeader* H = (const SlabHeader*)slab;
  int n = std::min(*num_moves, H->trace_len);
  std::memcpy(trace, slab + H->off_trace, (size_t)n * 32);
  *num_moves = n;
  return EMAT_OK;
}
/* Sufficient statistics of the global moves over the parts of this handle (header: emat_get_global_stats). */
emat_status emat_get_global_stats(emat_backend* h, int32_t num_partitions, double* Ttwiddle_beta_a, int64_t* num_muts_beta_ab, int64_t* num_muts) {
  if (!h || !Ttwiddle_beta_a || !num_muts_beta_ab) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (h->parts.empty()) return fail(h, EMAT_ERR_STATE, "no parts uploaded");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = sync_model_to_device(h); if (st) return st;
  if (num_partitions != h->num_partitions) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "num_partitions does not match emat_set_evo");
  if (num_partitions > k_max_stats_partitions) return fail(h, EMAT_ERR_CAPACITY, "emat_get_global_stats supports at most 4 site partitions");
  st = materialize(h); if (st) return st;
  if (h->pass_pending) { st = finish_pass(h); if (st) return st; }   // statistics of chains that finished their moves, or a loud failure
  const int W = 4 * num_partitions;
  for (auto& ph : h->parts)   // the per-node table lives in the part's scratch region
    if ((uint64_t)ph.n_nodes * W * 8u > (uint64_t)ph.scratch_bytes) return fail(h, EMAT_ERR_CAPACITY, "scratch region too small for the statistics table");
  const size_t n = h->parts.size();
  if (h->d_stats.n < n * k_stats_row) { std::vector<double> z(n * k_stats_row, 0.0); HIP_TRY(h->d_stats.upload(z.data(), z.size())); }
  KernelArgs a = make_args(h);
  hipLaunchKernelGGL(k_global_stats, dim3((unsigned)n), dim3(k_wave), 0, h->stream, a);
  HIP_TRY(hipGetLastError());
  std::vector<double> rows(n * k_stats_row);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(rows.data(), h->d_stats.p, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
  for (int k = 0; k < W; ++k) Ttwiddle_beta_a[k] = 0.0;
  for (int k = 0; k < 4 * W; ++k) num_muts_beta_ab[k] = 0;
  int64_t nm = 0;
  for (size_t p = 0; p < n; ++p) {   // fixed order: reproducible sums
    const double* r = rows.data() + p * k_stats_row;
    for (int k = 0; k < W; ++k) Ttwiddle_beta_a[k] += r[k];
    for (int k = 0; k < 4 * W; ++k) num_muts_beta_ab[k] += (int64_t)r[k_max_stats_partitions * 4 + k];
    nm += (int64_t)r[k_stats_row - 1];
  }
  if (num_muts) *num_muts = nm;
  return EMAT_OK;
}
/* calc_num_muts_l on the device (header: emat_get_num_muts_l) */
emat_status emat_get_num_muts_l(emat_backend* h, int32_t* num_muts_l) {
  if (!h || !num_muts_l) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (h->parts.empty()) return fail(h, EMAT_ERR_STATE, "no parts uploaded");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  if (h->pass_pending) { st = finish_pass(h); if (st) return st; }
  DevBuf<int32_t> d_out;
  HIP_TRY(d_out.alloc((size_t)h->L));
  HIP_TRY(hipMemsetAsync(d_out.p, 0, (size_t)h->L * sizeof(int32_t), h->stream));
  KernelArgs a = make_args(h);
  hipLaunchKernelGGL(k_num_muts_l, dim3((unsigned)h->parts.size()), dim3(k_wave), 0, h->stream, a, d_out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(num_muts_l, d_out.p, (size_t)h->L * sizeof(int32_t), hipMemcpyDeviceToHost));
  return EMAT_OK;
}

/* calc_Ttwiddle_l on the device (header: emat_get_part_tree_lengths / emat_Ttwiddle_l_partial / emat_Ttwiddle_l_finish) */
emat_status emat_get_part_tree_lengths(emat_backend* h, double* tree_length_of_part) {
  if (!h || !tree_length_of_part) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (h->parts.empty()) return fail(h, EMAT_ERR_STATE, "no parts uploaded");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  if (h->pass_pending) { st = finish_pass(h); if (st) return st; }
  const size_t n = h->parts.size();
  DevBuf<double> d_out; HIP_TRY(d_out.alloc(n));
  KernelArgs a = make_args(h);
  hipLaunchKernelGGL(k_part_lengths, dim3((unsigned)n), dim3(k_wave), 0, h->stream, a, d_out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(tree_length_of_part, d_out.p, n * sizeof(double), hipMemcpyDeviceToHost));
  return EMAT_OK;
}
emat_status emat_Ttwiddle_l_partial(emat_backend* h, const int32_t* ext_offset, const int32_t* ext_node, const double* ext_length,
                                    double* S, double* R, double* tree_length_below_root) {
  if (!h || !ext_offset || !S || !R) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (h->parts.empty()) return fail(h, EMAT_ERR_STATE, "no parts uploaded");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  if (h->pass_pending) { st = finish_pass(h); if (st) return st; }
  const size_t n = h->parts.size(); const size_t L = (size_t)h->L;
  const int32_t n_ext = ext_offset[n];
  if (n_ext < 0 || (n_ext > 0 && (!ext_node || !ext_length))) return EMAT_ERR_INVALID_ARGUMENT;
  for (size_t p = 0; p < n; ++p) {
    if (ext_offset[p] > ext_offset[p + 1]) return EMAT_ERR_INVALID_ARGUMENT;
    for (int32_t k = ext_offset[p]; k < ext_offset[p + 1]; ++k) if (ext_node[k] < 0 || ext_node[k] >= h->parts[p].n_nodes) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "ext_node out of range");
    if ((uint64_t)h->parts[p].n_nodes * 8u > (uint64_t)h->parts[p].scratch_bytes) return fail(h, EMAT_ERR_CAPACITY, "scratch region too small for the length table");
  }
  DevBuf<int32_t> d_off, d_node; DevBuf<double> d_val, d_S, d_D, d_T;
  HIP_TRY(d_off.upload(ext_offset, n + 1)); HIP_TRY(d_node.upload(ext_node, (size_t)n_ext)); HIP_TRY(d_val.upload(ext_length, (size_t)n_ext));
  HIP_TRY(d_S.alloc(L)); HIP_TRY(d_D.alloc(L + 1)); HIP_TRY(d_T.alloc(1));
  HIP_TRY(hipMemsetAsync(d_S.p, 0, L * sizeof(double), h->stream)); HIP_TRY(hipMemsetAsync(d_D.p, 0, (L + 1) * sizeof(double), h->stream)); HIP_TRY(hipMemsetAsync(d_T.p, 0, sizeof(double), h->stream));
  KernelArgs a = make_args(h);
  hipLaunchKernelGGL(k_ttwiddle_l, dim3((unsigned)n), dim3(k_wave), 0, h->stream, a, d_off.p, d_node.p, d_val.p, d_S.p, d_D.p, d_T.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<double> D(L + 1);
  HIP_TRY(hipMemcpy(S, d_S.p, L * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(D.data(), d_D.p, (L + 1) * sizeof(double), hipMemcpyDeviceToHost));
  double run = 0.0;
  for (size_t l = 0; l < L; ++l) { run += D[l]; R[l] = run; }
  if (tree_length_below_root && h->root_part >= 0) HIP_TRY(hipMemcpy(tree_length_below_root, d_T.p, sizeof(double), hipMemcpyDeviceToHost));
  return EMAT_OK;
}
emat_status emat_Ttwiddle_l_finish(emat_backend* h, const double* S_sum, const double* R_sum, double tree_length, double* Ttwiddle_l) {
  if (!h || !S_sum || !R_sum || !Ttwiddle_l) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->have_ref || !h->have_evo) return fail(h, EMAT_ERR_STATE, "set_ref_sequence and set_evo first");
  for (int l = 0; l < h->L; ++l) {
    const double q_ref = -h->q[(size_t)h->partition_for_site[l] * 16 + (size_t)h->ref[l] * 5];   // q^(l)_a of the reference state
    Ttwiddle_l[l] = q_ref * (tree_length - R_sum[l]) + S_sum[l];
  }
  return EMAT_OK;
}

/* Scalable_coalescent_prior on the device (header: emat_scalable_coalescent_partial / _log_prior / emat_get_scalable_coalescent_log_prior) */
emat_status emat_scalable_coalescent_partial(emat_backend* h, double t_ref, double t_step, int32_t first_cell, int32_t num_cells,
                                             double* k_bar_partial, double* sum_neg_log_pop, int32_t* first_cell_needed) {
  if (!h || !(t_step > 0.0) || num_cells < 0 || (num_cells > 0 && !k_bar_partial)) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (h->parts.empty()) return fail(h, EMAT_ERR_STATE, "no parts uploaded");
  if (!h->have_pop) return fail(h, EMAT_ERR_STATE, "no population model: emat_build_coalescent_parts (or the staged form) first");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  if (h->pass_pending) { st = finish_pass(h); if (st) return st; }
  const size_t n = h->parts.size();
  // a part's nodes span about as many cells of this grid as of its very-scalable window (same cell width, other origin)
  std::vector<uint64_t> off(n); std::vector<uint32_t> cap(n); uint64_t tot = 0;
  const double ratio = h->parts[0].coal.t_step > 0.0 ? h->parts[0].coal.t_step / t_step : 1.0;
  for (size_t p = 0; p < n; ++p) {
    const PartHost& ph = h->parts[p];
    const size_t window = (size_t)std::max(0, ph.coal.n_cells_total - ph.coal.cell_first);   // (= k_bar_p.size(); the vectors themselves may live only on the device)
    const double cells_vs = (double)(ph.includes_run_root ? 2 * window + 512 : window);
    cap[p] = (uint32_t)std::min<double>(1e7, std::ceil(cells_vs * ratio) + 4.0);
    off[p] = tot; tot += cap[p];
  }
  DevBuf<uint64_t> d_off; DevBuf<uint32_t> d_cap; DevBuf<double> d_out, d_meta;
  HIP_TRY(d_off.upload(off.data(), n)); HIP_TRY(d_cap.upload(cap.data(), n)); HIP_TRY(d_out.alloc((size_t)tot)); HIP_TRY(d_meta.alloc(n * 4));
  KernelArgs a = make_args(h);
  hipLaunchKernelGGL(k_scalable_prior, dim3((unsigned)n), dim3(k_wave), 0, h->stream, a, t_ref, t_step, d_off.p, d_cap.p, d_out.p, d_meta.p);
  HIP_TRY(hipGetLastError());
  std::vector<double> rows((size_t)tot), meta(n * 4);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(rows.data(), d_out.p, rows.size() * sizeof(double), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(meta.data(), d_meta.p, meta.size() * sizeof(double), hipMemcpyDeviceToHost));
  int first_needed = 0;
  for (size_t p = 0; p < n; ++p) {
    if (meta[4 * p + 3] != 0.0) return fail(h, EMAT_ERR_CAPACITY, "part " + std::to_string(p) + " spans more grid cells than expected");
    if (meta[4 * p + 1] > 0.0) first_needed = std::min(first_needed, (int)meta[4 * p]);
  }
  if (first_cell_needed) *first_cell_needed = first_needed;
  for (int j = 0; j < num_cells; ++j) k_bar_partial[j] = 0.0;
  double logs = 0.0;
  for (size_t p = 0; p < n; ++p) {   // part order: reproducible sums
    const int jlo = (int)meta[4 * p], cnt = (int)meta[4 * p + 1];
    logs += meta[4 * p + 2];
    if (num_cells == 0) continue;
    if (cnt > 0 && (jlo < first_cell || jlo + cnt > first_cell + num_cells)) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "the cell range given does not cover every node (ask with num_cells = 0 for first_cell_needed)");
    const double* row = rows.data() + off[p];
    for (int k = 0; k < cnt; ++k) k_bar_partial[jlo - first_cell + k] += row[k];
    if (h->parts[p].includes_run_root) for (int j = jlo + cnt; j < first_cell + num_cells && j < 0; ++j) k_bar_partial[j - first_cell] -= 1.0;   // the root part's constant tail
  }
  if (sum_neg_log_pop) *sum_neg_log_pop = logs;
  return EMAT_OK;
}
emat_status emat_scalable_coalescent_log_prior(emat_backend* h, double t_ref, double t_step, int32_t first_cell, int32_t num_cells,
                                               const double* k_bar_partial_sum, double sum_neg_log_pop, double* log_prior) {
  if (!h || !(t_step > 0.0) || num_cells < 0 || (num_cells > 0 && !k_bar_partial_sum) || !log_prior) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->have_pop) return fail(h, EMAT_ERR_STATE, "no population model: emat_build_coalescent_parts (or the staged form) first");
  double r = 0.0;
  for (int k = 0; k < num_cells; ++k) {   // scalable_coalescent.cpp:163-187, cells in increasing time
    const int j = first_cell + k;
    if (j >= 0) break;
    const double lb = t_ref + (double)j * t_step, ub = lb + t_step;
    double popsize_bar = h->pop.pop_integral(lb, ub) / t_step;
    if (popsize_bar == 0.0) popsize_bar = 1e-100;   // the reference's stopgap (:62-64)
    const double kbar = 1.0 + k_bar_partial_sum[k];   // cells before t_ref start at one lineage (:56)
    r -= t_step * kbar * (kbar - 1) / (2.0 * popsize_bar);
  }
  *log_prior = r + sum_neg_log_pop;
  return EMAT_OK;
}
emat_status emat_get_scalable_coalescent_log_prior(emat_backend* h, double t_ref, double t_step, double* log_prior) {
  if (!h || !log_prior) return EMAT_ERR_INVALID_ARGUMENT;
  int32_t first = 0;
  emat_status st = emat_scalable_coalescent_partial(h, t_ref, t_step, 0, 0, nullptr, nullptr, &first); if (st) return st;
  std::vector<double> kb((size_t)(-first)); double logs = 0.0;
  st = emat_scalable_coalescent_partial(h, t_ref, t_step, first, -first, kb.data(), &logs, nullptr); if (st) return st;
  return emat_scalable_coalescent_log_prior(h, t_ref, t_step, first, -first, kb.data(), logs, log_prior);
}

/* test hook (header: emat_debug_gamma) */
emat_status emat_debug_gamma(emat_backend* h, int32_t mode, int32_t n, const double* a, const double* x_or_q, double* out) {
  if (!h || n < 0 || (mode != 0 && mode != 1) || (n > 0 && (!a || !x_or_q || !out))) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (n == 0) return EMAT_OK;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  DevBuf<double> da, dx, dout;
  HIP_TRY(da.upload(a, (size_t)n)); HIP_TRY(dx.upload(x_or_q, (size_t)n)); HIP_TRY(dout.upload(x_or_q, (size_t)n));
  hipLaunchKernelGGL(k_debug_gamma, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, da.p, dx.p, dout.p, n, mode);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return EMAT_OK;
}
/* test hooks (header: emat_debug_pop, emat_debug_interval_op) */
emat_status emat_debug_pop(emat_backend* h, const emat_pop_model* pm, int32_t op, int32_t n, const double* a, const double* b, double* out) {
  if (!h || !pm || n < 0 || (op != 0 && op != 1) || (n > 0 && (!a || !b || !out))) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (n == 0) return EMAT_OK;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  HostPopModel hp;
  try { hp = HostPopModel::from_c(*pm); } catch (const std::exception& ex) { return fail(h, EMAT_ERR_INVALID_ARGUMENT, ex.what()); }
  DevBuf<double> dx, dg, da, db, dout;
  HIP_TRY(dx.upload(hp.x.data(), hp.x.size())); HIP_TRY(dg.upload(hp.gamma.data(), hp.gamma.size()));
  HIP_TRY(da.upload(a, (size_t)n)); HIP_TRY(db.upload(b, (size_t)n)); HIP_TRY(dout.alloc((size_t)n));
  PopTable pt{};
  pt.kind = hp.kind; pt.skygrid_type = hp.skygrid_type; pt.skygrid_num_knots = (int)hp.x.size();
  for (int i = 0; i < 4; ++i) pt.p[i] = hp.p[i];
  pt.t_c = hp.t_c; pt.skygrid_x = dx.p; pt.skygrid_gamma = dg.p;
  pt.skygrid_inv_dx = (hp.x.size() >= 2 && hp.x.back() > hp.x.front()) ? (double)(hp.x.size() - 1) / (hp.x.back() - hp.x.front()) : 0.0;
  hipLaunchKernelGGL(k_debug_pop, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, h->stream, pt, (int)op, da.p, db.p, dout.p, (int)n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
  return EMAT_OK;
}
emat_status emat_debug_interval_op(emat_backend* h, int32_t op, const int32_t* a, int32_t na, const int32_t* b, int32_t nb, int32_t* out, int32_t* n_out) {
  if (!h || !n_out || na < 0 || nb < 0 || (na > 0 && !a) || (nb > 0 && !b) || !out) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  DevBuf<IvRec> dA, dB, dO; DevBuf<int> dn;
  HIP_TRY(dA.upload((const IvRec*)a, (size_t)na)); HIP_TRY(dB.upload((const IvRec*)b, (size_t)(op == 5 ? 0 : nb))); HIP_TRY(dO.alloc((size_t)(na + nb + 1))); HIP_TRY(dn.alloc(1));
  hipLaunchKernelGGL(k_debug_interval_op, dim3(1), dim3(64), 0, h->stream, (int)op, dA.p, (int)na, dB.p, (int)(op == 5 ? 0 : nb), op == 5 && nb > 0 ? b[0] : 0, dO.p, dn.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  int cnt = 0;
  HIP_TRY(hipMemcpy(&cnt, dn.p, sizeof(int), hipMemcpyDeviceToHost));
  if (cnt < 0) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_debug_interval_op: unknown op");
  *n_out = cnt;
  if (op <= 3 && cnt > 0) HIP_TRY(hipMemcpy(out, dO.p, (size_t)cnt * sizeof(IvRec), hipMemcpyDeviceToHost));
  return EMAT_OK;
}
/* test hook (header: emat_debug_tree_query) */
emat_status emat_debug_tree_query(emat_backend* h, int32_t part_id, int32_t op, int32_t n, const int32_t* a, const int32_t* b, int32_t* out) {
  if (!h || n < 0 || (op != 0 && op != 1) || (n > 0 && (!a || !b || !out))) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  const int nn = h->parts[part_id].n_nodes;
  for (int i = 0; i < n; ++i) if (a[i] < -1 || a[i] >= nn || b[i] < -1 || b[i] >= nn) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_debug_tree_query: node index out of range");
  if (n == 0) return EMAT_OK;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = emat_synchronize(h); if (st) return st;
  st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  DevBuf<int32_t> da, db, dout;
  HIP_TRY(da.upload(a, (size_t)n)); HIP_TRY(db.upload(b, (size_t)n)); HIP_TRY(dout.alloc((size_t)n));
  KernelArgs ka = make_args(h);
  hipLaunchKernelGGL(k_debug_tree_query, dim3(1), dim3(k_wave), 0, h->stream, ka, (int)part_id, (int)op, da.p, db.p, dout.p, (int)n);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out, dout.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return EMAT_OK;
}
/* test hook (header: emat_debug_graft) */
emat_status emat_debug_graft(emat_backend* h, int32_t part_id, int32_t X, double mu_proposal, int32_t mode, int32_t new_sibling, double new_t_P,
                             double* out, int32_t out_cap, int32_t* out_len) {
  if (!h || !out || !out_len || out_cap < 2 || mode < 0 || mode > 3) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  const int nn = h->parts[part_id].n_nodes;
  if (X < 0 || X >= nn || (mode == 3 && (new_sibling < 0 || new_sibling >= nn))) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_debug_graft: node index out of range");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = emat_synchronize(h); if (st) return st;
  st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  if (!h->derived_valid) { st = launch_recalc(h); if (st) return st; }   // the analysis starts from the nodes' lambda_i and missing-site counts, as a move does
  DevBuf<double> dout; DevBuf<int32_t> dlen;
  HIP_TRY(dout.alloc((size_t)out_cap)); HIP_TRY(dlen.alloc(1));
  KernelArgs ka = make_args(h);
  hipLaunchKernelGGL(k_debug_graft, dim3(1), dim3(k_wave), 0, h->stream, ka, (int)part_id, (int)X, mu_proposal, (int)mode, (int)new_sibling, new_t_P, dout.p, (int)out_cap, dlen.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out_len, dlen.p, sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(out, dout.p, (size_t)std::min(*out_len, out_cap) * sizeof(double), hipMemcpyDeviceToHost));
  if (mode >= 1) { h->host_slabs_current = false; h->headers_current = false; }   // the part's slab was edited on the device
  if (out[0] != 0.0) return fail(h, EMAT_ERR_INTERNAL, "emat_debug_graft: the device code stopped with part status " + std::to_string((int)out[0]));
  return *out_len > out_cap ? fail(h, EMAT_ERR_CAPACITY, "emat_debug_graft: out_cap too small") : EMAT_OK;
}
/* test hook (header: emat_debug_sample_history) */
emat_status emat_debug_sample_history(emat_backend* h, int32_t part_id, int32_t n, const int32_t* branch, const double* t_end, const uint8_t* start_seq, double T, double mu,
                                      int32_t* counts, double* muts, int32_t muts_cap, int32_t* num_muts) {
  if (!h || n < 0 || !branch || !t_end || !start_seq || !counts || !muts || muts_cap < 0 || !num_muts) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  const int nn = h->parts[part_id].n_nodes;
  for (int i = 0; i < n; ++i) if (branch[i] < 0 || branch[i] >= nn) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_debug_sample_history: node index out of range");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = emat_synchronize(h); if (st) return st;
  st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  if (!h->derived_valid) { st = launch_recalc(h); if (st) return st; }
  DevBuf<int32_t> db, dc, ds; DevBuf<double> dt, dm; DevBuf<uint8_t> dseq;
  HIP_TRY(db.upload(branch, (size_t)std::max(n, 1))); HIP_TRY(dt.upload(t_end, (size_t)std::max(n, 1))); HIP_TRY(dseq.upload(start_seq, (size_t)h->cfg.num_sites));
  HIP_TRY(dc.alloc((size_t)std::max(n, 1))); HIP_TRY(dm.alloc((size_t)std::max(muts_cap, 1) * 4)); HIP_TRY(ds.alloc(2));
  KernelArgs ka = make_args(h);
  hipLaunchKernelGGL(k_debug_sample_history, dim3(1), dim3(k_wave), 0, h->stream, ka, (int)part_id, (int)n, db.p, dt.p, dseq.p, T, mu, dc.p, dm.p, (int)muts_cap, ds.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  int32_t status[2];
  HIP_TRY(hipMemcpy(status, ds.p, sizeof status, hipMemcpyDeviceToHost));
  h->host_slabs_current = false; h->headers_current = false;   // the part's random stream moved on
  if (status[0] != 0) return fail(h, EMAT_ERR_INTERNAL, "emat_debug_sample_history: the device code stopped with part status " + std::to_string(status[0]));
  *num_muts = status[1];
  if (n > 0) HIP_TRY(hipMemcpy(counts, dc.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  if (status[1] > 0) HIP_TRY(hipMemcpy(muts, dm.p, (size_t)std::min(status[1], muts_cap) * 4 * sizeof(double), hipMemcpyDeviceToHost));
  return status[1] > muts_cap ? fail(h, EMAT_ERR_CAPACITY, "emat_debug_sample_history: muts_cap too small") : EMAT_OK;
}
/* test hook (header: emat_debug_edit) */
emat_status emat_debug_edit(emat_backend* h, int32_t part_id, int32_t X, int32_t n_ops, const int32_t* op_kind, const int32_t* op_node, const double* op_t) {
  if (!h || n_ops < 0 || (n_ops > 0 && (!op_kind || !op_node || !op_t))) return EMAT_ERR_INVALID_ARGUMENT;
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  const int nn = h->parts[part_id].n_nodes;
  if (X < 0 || X >= nn) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_debug_edit: node index out of range");
  for (int i = 0; i < n_ops; ++i) if (op_kind[i] < 0 || op_kind[i] > 3 || (op_kind[i] == 3 && (op_node[i] < 0 || op_node[i] >= nn))) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_debug_edit: bad step");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  emat_status st = emat_synchronize(h); if (st) return st;
  st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  if (!h->derived_valid) { st = launch_recalc(h); if (st) return st; }   // the session keeps lambda_i and the missing-site counts up to date from where they stand
  DevBuf<int32_t> dk, dn, ds; DevBuf<double> dt;
  HIP_TRY(dk.upload(op_kind, (size_t)std::max(n_ops, 1))); HIP_TRY(dn.upload(op_node, (size_t)std::max(n_ops, 1))); HIP_TRY(dt.upload(op_t, (size_t)std::max(n_ops, 1))); HIP_TRY(ds.alloc(1));
  KernelArgs ka = make_args(h);
  hipLaunchKernelGGL(k_debug_edit, dim3(1), dim3(k_wave), 0, h->stream, ka, (int)part_id, (int)X, (int)n_ops, dk.p, dn.p, dt.p, ds.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  int32_t status = 0;
  HIP_TRY(hipMemcpy(&status, ds.p, sizeof status, hipMemcpyDeviceToHost));
  h->host_slabs_current = false; h->headers_current = false;
  if (status != 0) return fail(h, EMAT_ERR_INTERNAL, "emat_debug_edit: the device code stopped with part status " + std::to_string(status));
  return EMAT_OK;
}
/* debugging aid (not part of the boundary): how many parts the next launch runs with each code variant
 * (out3 = {whole slab staged in LDS, prefix staged, HBM only}); mirrors the kernel's per-part decision (single class). */
emat_status emat_debug_variant_counts(emat_backend* h, int32_t* out3) {
  if (!h || !out3 || h->host_only) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  st = pull_from_device(h); if (st) return st;
  out3[0] = out3[1] = out3[2] = 0;
  const uint32_t area = h->class_lds[h->num_classes - 1];
  const bool tables = h->num_partitions <= k_max_lds_partitions;
  for (auto& ph : h->parts) {
    const SlabHeader* H = (const SlabHeader*)(h->h_slabs.data() + ph.slab_off);
    const bool can = tables && area != 0 && H->off_nodes == (uint32_t)sizeof(SlabHeader);
    if (can && (H->heap_end <= area || H->heap_top + k_lds_heap_room <= area)) ++out3[0]; else if (can && H->heap_begin <= area) ++out3[1]; else ++out3[2];
  }
  return EMAT_OK;
}
/* debugging aid (profiling builds): bytes the moves' arena handed out per allocating source line, [line & 2047][LDS, HBM] */
emat_status emat_debug_arena_sites(emat_backend* h, uint64_t* out_4096) {
  if (!h || !out_4096 || h->host_only) return EMAT_ERR_INVALID_ARGUMENT;
#ifdef EMAT_PROFILE_PHASES
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpyFromSymbol(out_4096, HIP_SYMBOL(::emat::g_arena_site_bytes), sizeof(unsigned long long) * 4096));
  return EMAT_OK;
#else
  return fail(h, EMAT_ERR_STATE, "built without -DEMAT_PROFILE_PHASES");
#endif
}
/* debugging aid (profiling builds): inclusive ticks and calls of the EMAT_TIMED scopes, [header * 2048 + line & 2047][ticks, calls]; read and cleared */
emat_status emat_debug_fn_ticks(emat_backend* h, uint64_t* out_12288) {
  if (!h || !out_12288 || h->host_only) return EMAT_ERR_INVALID_ARGUMENT;
#if defined(EMAT_PROFILE_PHASES) || defined(EMAT_COUNT_CALLS)
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<unsigned long long> z((size_t)12288 * ::emat::k_fn_replicas, 0);
  HIP_TRY(hipMemcpyFromSymbol(z.data(), HIP_SYMBOL(::emat::g_fn_ticks), sizeof(unsigned long long) * z.size()));
  for (int k = 0; k < 12288; ++k) { unsigned long long sum = 0; for (int r = 0; r < ::emat::k_fn_replicas; ++r) sum += z[(size_t)r * 12288 + k]; out_12288[k] = sum; }
  std::fill(z.begin(), z.end(), 0ull);
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(::emat::g_fn_ticks), z.data(), sizeof(unsigned long long) * z.size()));
  const unsigned min_lists = h->cfg_fn_min_lists;   // from the next pass on: only parts whose lists take at least this many bytes
  HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(::emat::g_fn_min_list_bytes), &min_lists, sizeof(min_lists)));
  return EMAT_OK;
#else
  return fail(h, EMAT_ERR_STATE, "built without -DEMAT_PROFILE_PHASES");
#endif
}
/* debugging aid: how much LDS arena the moves of every main-class part start with (bytes; -1 for parts of side classes) */
emat_status emat_debug_arena_bytes(emat_backend* h, int32_t* out_n) {
  if (!h || !out_n || h->host_only) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = sync_model_to_device(h); if (st) return st;
  st = materialize(h); if (st) return st;
  st = pull_from_device(h); if (st) return st;
  const uint32_t area = h->class_lds[h->num_classes - 1];
  for (size_t p = 0; p < h->parts.size(); ++p) {
    const SlabHeader* H = (const SlabHeader*)(h->h_slabs.data() + h->parts[p].slab_off);
    if (h->class_of[p] != h->num_classes - 1) { out_n[p] = -1; continue; }
    const uint32_t want = (H->heap_top + k_lds_heap_room + 15u) & ~15u;
    uint32_t used;
    if (H->heap_end <= area) used = std::min(H->heap_end, want); else if (want <= area) used = want; else used = (H->heap_begin + 15u) & ~15u;
    out_n[p] = used <= area ? (int32_t)(area - ((used + 15u) & ~15u)) : 0;
  }
  return EMAT_OK;
}
/* debugging aid (not part of the boundary): duration and start tick (100 MHz wall clock) of every part in the last pass */
emat_status emat_debug_part_ticks(emat_backend* h, int64_t* out_2n) {
  if (!h || !out_2n || h->host_only || !h->slabs_on_device) return EMAT_ERR_INVALID_ARGUMENT;
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  { emat_status js = join_side_classes(h); if (js) return js; }
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out_2n, h->d_part_ticks.p, sizeof(int64_t) * 2 * h->parts.size(), hipMemcpyDeviceToHost));
  return EMAT_OK;
}
/* debugging aid: workgroup entry and exit ticks (100 MHz) of the first 8 tickets of every part in the last pass, out[(2 * ticket + {0, 1}) * num_parts + part] */
emat_status emat_debug_ticket_ticks(emat_backend* h, int64_t* out_16n) {
  if (!h || !out_16n || h->host_only || !h->slabs_on_device) return EMAT_ERR_INVALID_ARGUMENT;
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  { emat_status js = join_side_classes(h); if (js) return js; }
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(out_16n, h->d_part_ticks.p + 2 * h->parts.size(), sizeof(int64_t) * 2 * k_ticket_log * h->parts.size(), hipMemcpyDeviceToHost));
  return EMAT_OK;
}
/* debugging aid (not part of the boundary): phase profile of a part, see EMAT_PROFILE_PHASES */
emat_status emat_debug_phase_ticks(emat_backend* h, int32_t part_id, int64_t* out16) {
  if (!h || !out16 || part_id < 0 || part_id >= (int)h->parts.size()) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = pull_from_device(h); if (st) return st;
  const SlabHeader* H = (const SlabHeader*)(h->h_slabs.data() + h->parts[part_id].slab_off);
#ifdef EMAT_PROFILE_PHASES
  for (int i = 0; i < 16; ++i) out16[i] = H->phase_ticks[i];
  if (h->cfg_phase_extra) for (int i = 0; i < 16; ++i) out16[i] = ((const int64_t*)H->reserved)[i];   // scan and arena counters instead
#else
  (void)H; for (int i = 0; i < 16; ++i) out16[i] = 0;   // phase counters exist only in -DEMAT_PROFILE_PHASES builds
#endif
  return EMAT_OK;
}
/* Duration of the k_run_moves launch of the last pass, from HIP events around that launch on its stream. */
emat_status emat_last_kernel_ms(emat_backend* h, double* ms, int32_t* num_parts_in_kernel) {
  if (!h || !ms) return EMAT_ERR_INVALID_ARGUMENT;
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  HIP_TRY(hipEventSynchronize(h->ev_stop));
  float f = 0.f;
  HIP_TRY(hipEventElapsedTime(&f, h->ev_start, h->ev_stop));
  *ms = f;
  if (num_parts_in_kernel) *num_parts_in_kernel = (int)h->parts.size();
  return EMAT_OK;
}
emat_status emat_last_run_ms(emat_backend* h, double* ms) {
  if (!h || !ms) return EMAT_ERR_INVALID_ARGUMENT;
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle");
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  HIP_TRY(hipEventSynchronize(h->ev_stop));
  float f = 0.f;
  HIP_TRY(hipEventElapsedTime(&f, h->ev_start, h->ev_stop));
  h->last_run_ms = f; *ms = f;
  return EMAT_OK;
}

}  // extern "C"

#include "emat_gtree_host.hpp"
#include "emat_build_host.hpp"
#include "emat_utree_host.hpp"
