// emat_device_core.hpp -- device-side building blocks of the EMAT local-move engine (gfx950).
//
// One wavefront owns one partition part and runs its Markov chain on the part's slab
// (emat_slab.hpp), which lives in LDS when it fits and in HBM otherwise; every function here
// takes a `Ctx` whose base pointer may be either.  The arithmetic follows the reference
// expression by expression (file:line cited at each function) so that results stay within 1e-9
// of the CPU path; integer outputs (sites, states, interval endpoints, region indices) are exact.
//
// This header is device code only.  It is NOT shared with oracle/ (which is an independent
// restatement used to check it).
// NO include guard: this header (through emat_device_moves.hpp) is included once per code variant by
// emat_backend.hip, with EMAT_DEV_NS naming the variant's namespace and EMAT_VARIANT_LDS selecting it:
//   EMAT_VARIANT_LDS = 1  the part's persistent slab is staged in LDS at a fixed offset of the workgroup's dynamic LDS
//                         block, so the slab base, header, node array and HKY tables are compile-time LDS addresses
//                         (ds_read / ds_write with immediate offsets, 32-bit address arithmetic);
//   EMAT_VARIANT_LDS = 2  only the slab's fixed-size prefix (header, nodes, coalescent cells, trace) is staged, at the
//                         same LDS offsets; the list heap stays in HBM (parts too large to stage whole);
//   EMAT_VARIANT_LDS = 0  the slab stays in HBM; base pointers are loaded from the context (generic addressing).
#ifndef EMAT_DEV_NS
#error "define EMAT_DEV_NS and EMAT_VARIANT_LDS before including the device headers"
#endif

#include <hip/hip_runtime.h>
#include <cfloat>
#include <cstdint>

#include "emat_slab.hpp"

#ifndef EMAT_DEVICE_COMMON_ONCE_
#define EMAT_DEVICE_COMMON_ONCE_
namespace emat {
// LDS of a workgroup: two STATIC objects -- the HKY tables and the context -- and the dynamic block [staged slab ...][scratch
// arena].  They are separate objects on purpose: stores into the slab through a computed index (a node's time, a list entry)
// then provably do not touch the context, so the compiler may keep context fields (RNG position, failure flag, byte counter,
// arena marks) in registers across them instead of reloading each from LDS after every such store -- inside one array of bytes
// every variable-index store may alias everything.
extern __shared__ __attribute__((aligned(16))) uint8_t emat_lds[];
constexpr uint32_t k_lds_tables_bytes = k_max_lds_partitions * (1 + 4 + 16) * 8;   // mu, pi, q per site partition
constexpr uint32_t k_lds_ctx_bytes = 288;
// The chain's random numbers come from a counter-based generator, so its NEXT blocks can be computed before they are wanted -- and by
// lanes the chain does not use: whenever the chain is about to run out, the wave's other lanes compute the Philox blocks of the next
// k_rng_blocks counters side by side (one block, two 64-bit draws, per lane) into this array, and the chain's draws become a 16-byte LDS
// read instead of ten rounds of 32 x 32 -> 64 multiplies on the scalar unit (a third of a simple move's scalar instructions; inlined
// at every draw, also a good part of the code the instruction cache has to hold).  Same counters, same numbers, same order.
// 0 switches it off (every draw computes its block).
#ifndef EMAT_RNG_BLOCKS
#define EMAT_RNG_BLOCKS 32
#endif
constexpr uint32_t k_rng_blocks = EMAT_RNG_BLOCKS;
static_assert(k_rng_blocks == 0 || (k_rng_blocks <= 64 && (k_rng_blocks & (k_rng_blocks - 1)) == 0), "one block per lane, a power of two");
constexpr uint32_t k_lds_rng_bytes = k_rng_blocks * 16;
constexpr uint32_t k_lds_static_bytes = k_lds_tables_bytes + k_lds_ctx_bytes + k_lds_rng_bytes + k_max_lds_partitions * 16 * 8;      // what every k_run_moves workgroup holds besides the slab image and the arena
__shared__ __attribute__((aligned(16))) uint8_t emat_lds_tables[k_lds_tables_bytes];
__shared__ __attribute__((aligned(16))) uint8_t emat_lds_ctx[k_lds_ctx_bytes];
__shared__ __attribute__((aligned(16))) uint32_t emat_lds_rng[k_rng_blocks ? k_rng_blocks * 4 : 4];
// log(mu q_ab) per site partition: what a mutation a -> b at a site of relative rate 1 contributes to its branch's log G besides its
// time (phylo_tree_calc.h:185-206 takes the logarithm per mutation and evaluation): sixteen numbers per partition, taken by the device's
// own log from the very product the moves would form (mu * 1.0 * q), when the tables are staged.
constexpr uint32_t k_lds_logq_bytes = k_max_lds_partitions * 16 * 8;
__shared__ __attribute__((aligned(16))) double emat_lds_logq[k_max_lds_partitions * 16];
// The staged slab's HEADER is a third static object, for the same reason: its fields (root, node count, the coalescent
// window, heap marks, counters) are read all the time and must not look clobbered by every store into nodes or lists.
// The dynamic block then starts with slab byte sizeof(SlabHeader): slab offset `off` lives at emat_lds + off - sizeof(SlabHeader).
__shared__ __attribute__((aligned(16))) uint8_t emat_lds_hdr[sizeof(SlabHeader)];
// Where the dynamic block starts in a k_run_moves workgroup: right after the five static objects above (the library is compiled with
// -amdgpu-lower-module-lds-strategy=module, which gives them fixed addresses; every one is a multiple of 16 bytes).  The out-of-line
// device functions of the staged variants use this CONSTANT instead of the symbol `emat_lds`, whose address LLVM makes them look up in a
// per-kernel table in constant memory (an s_load and its wait at the entry of every function that touches the slab); run_moves_body
// checks at its start that the dynamic block really is there and stops every part loudly if it is not.
constexpr uint32_t k_lds_dyn_base = k_lds_tables_bytes + k_lds_ctx_bytes + (k_rng_blocks ? k_rng_blocks * 16 : 16) + k_lds_logq_bytes + (uint32_t)sizeof(SlabHeader);
}  // namespace emat
// Profiling and call-counting builds declare one more static LDS object (s_fn_count_me, emat_backend.hip): the dynamic block then does not
// start at the constant, and run_moves_body's check would stop every part (fail_line -3).  Those builds take the symbol instead.
#if (defined(EMAT_PROFILE_PHASES) || defined(EMAT_COUNT_CALLS)) && !defined(EMAT_X_DYN_LDS_BY_TABLE)
#define EMAT_X_DYN_LDS_BY_TABLE
#endif
#ifdef EMAT_X_DYN_LDS_BY_TABLE   // (A/B: the symbol, as until round 5)
#define EMAT_DYN_LDS ::emat::emat_lds
#else
#define EMAT_DYN_LDS ((uint8_t*)(__attribute__((address_space(3))) uint8_t*)(uintptr_t)::emat::k_lds_dyn_base)
#endif
#define EMAT_D static __device__ inline
#define EMAT_DN static __device__ __noinline__
#define EMAT_DF static __device__ __forceinline__
// A function whose calls must not be marked as tail calls: its callees then qualify for LLVM's no-callee-saved-registers
// optimisation.  The top of the call tree carries it; below, calls keep the standard convention.
#define EMAT_NOTAIL __attribute__((disable_tail_calls))
// Thirteen functions of the topology moves are INLINED at their call sites since round 6 (-DEMAT_OUTL_<TAG> puts one back out of line for an A/B); two large
// ones stay out of line (-DEMAT_INL_<TAG> inlines one).  Measured one at a time and in combination at C4: KTP / SST -- the
// K-truncated Poisson draw and the trajectory of one site, three rejection rounds per constrained site -- + 1.2 % of a pass together; HOP / SLIDE -- tree_editing's
// hop and slide, once per level an SPR climbs -- + 0.5 % and + 0.3 %; FINI, PNIG, PEEL, PICKT, ADJ (finish_inner_graft_analysis, propose_new_inner_graft_mutations,
// peel_inner_graft, study_pick_time_in_region, adjust_mutational_history) +- 0.1 % each and together; with SMH, SUMM, APPLY, LALPHA (sample_mutational_history,
// summarize_closed_mutations, apply_inner_graft, study_log_alpha_in_region: - 0.2 ... - 0.5 % each ALONE) + 1.2 % all nine together -- what a call costs depends on what
// else is out of line around it.  START (start_inner_graft_analysis, four call sites) - 0.8 %, TOPO (spr_move_topology) - 0.4 %: they stay calls; so do the leaf helpers sd_push_back_v,
// find_MRCA_of, reconstruct_missing_sites_at, calc_site_state_at and edit_flip (- 0.1 ... - 0.5 % each inlined).
#ifdef EMAT_OUTL_KTP
#define EMAT_FN_KTP EMAT_DN
#else
#define EMAT_FN_KTP EMAT_DF
#endif
#ifdef EMAT_OUTL_SST
#define EMAT_FN_SST EMAT_DN
#else
#define EMAT_FN_SST EMAT_DF
#endif
#ifdef EMAT_OUTL_HOP
#define EMAT_FN_HOP EMAT_DN
#else
#define EMAT_FN_HOP EMAT_DF
#endif
#ifdef EMAT_OUTL_SLIDE
#define EMAT_FN_SLIDE EMAT_DN
#else
#define EMAT_FN_SLIDE EMAT_DF
#endif
#ifdef EMAT_OUTL_FINI
#define EMAT_FN_FINI EMAT_DN
#else
#define EMAT_FN_FINI EMAT_DF
#endif
#ifdef EMAT_OUTL_PNIG
#define EMAT_FN_PNIG EMAT_DN
#else
#define EMAT_FN_PNIG EMAT_DF
#endif
#ifdef EMAT_OUTL_PEEL
#define EMAT_FN_PEEL EMAT_DN
#else
#define EMAT_FN_PEEL EMAT_DF
#endif
#ifdef EMAT_OUTL_PICKT
#define EMAT_FN_PICKT EMAT_DN
#else
#define EMAT_FN_PICKT EMAT_DF
#endif
#ifdef EMAT_OUTL_ADJ
#define EMAT_FN_ADJ EMAT_DN
#else
#define EMAT_FN_ADJ EMAT_DF
#endif
#ifdef EMAT_OUTL_SMH
#define EMAT_FN_SMH EMAT_DN
#else
#define EMAT_FN_SMH EMAT_DF
#endif
#ifdef EMAT_OUTL_SUMM
#define EMAT_FN_SUMM EMAT_DN
#else
#define EMAT_FN_SUMM EMAT_DF
#endif
#ifdef EMAT_OUTL_APPLY
#define EMAT_FN_APPLY EMAT_DN
#else
#define EMAT_FN_APPLY EMAT_DF
#endif
#ifdef EMAT_OUTL_LALPHA
#define EMAT_FN_LALPHA EMAT_DN
#else
#define EMAT_FN_LALPHA EMAT_DF
#endif
#ifdef EMAT_INL_START
#define EMAT_FN_START EMAT_DF
#else
#define EMAT_FN_START EMAT_DN
#endif
#ifdef EMAT_INL_TOPO
#define EMAT_FN_TOPO EMAT_DF
#else
#define EMAT_FN_TOPO EMAT_DN
#endif
// -DEMAT_COUNT_CALLS: scripts/count_calls.py puts EMAT_CALLED(header) at the top of every device function of a copy of these
// headers; calls are counted per (header, line) in g_fn_ticks[..][1] and read with emat_debug_fn_ticks.
#ifdef EMAT_COUNT_CALLS
#define EMAT_CALLED(file_id) do { if (threadIdx.x == 0) atomicAdd(&::emat::g_fn_ticks[(blockIdx.x & 63) * (3 * 2048) + (file_id) * 2048 + (__LINE__ & 2047)][1], 1ull); } while (0)
#endif
#endif  // EMAT_DEVICE_COMMON_ONCE_

namespace emat {
namespace EMAT_DEV_NS {

constexpr double k_neg_dbl_max = -1.7976931348623157e308;
constexpr double k_inf = __builtin_huge_val();
constexpr int k_no_node = -1;

#ifndef EMAT_CONST_AS
#define EMAT_CONST_AS __attribute__((address_space(4)))
#endif
// A pointer every active lane agrees on, retyped to the constant address space: loads through it are scalar loads (s_load, the
// scalar cache) instead of vector loads.
template <class T> EMAT_DF const EMAT_CONST_AS T* uniform_const_ptr(const T* p) {
  const uint64_t a = (uint64_t)p;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return (const EMAT_CONST_AS T*)(((uint64_t)hi << 32) | lo);
}

// ---- per-wave context ----------------------------------------------------------------------------
struct Ctx {
  uint8_t* S;                 // slab base (LDS or HBM; generic address space)
  uint8_t* G;                 // base for scratch offsets: always the HBM copy of the slab (scratch is never staged in LDS)
  SlabHeader* H;
  NodeRec* N;
  // evolution model: per-site arrays in HBM, per-partition HKY tables in LDS when staged
  int L;
  // per-site model arrays: always in HBM, typed as global pointers so that they are read with global_load (own
  // counter, no LDS-aperture check) instead of generic flat_load
  const __attribute__((address_space(1))) uint8_t* ref;
  const __attribute__((address_space(1))) uint8_t* part;
  const __attribute__((address_space(1))) double* nu;
  const __attribute__((address_space(1))) double* cumQ;
  const double* mu;           // [P]
  const double* pi;           // [P][4]
  const double* q;            // [P][16]
  const PopTable* pop;
  double t_max_tip;
  bool only_displacing_inner_nodes;
  bool topology_moves_enabled;
  bool includes_run_root;
  bool uniform_sites;         // one site partition and nu_l == 1 everywhere (EvoTable::uniform_sites): site_part / site_nu answer without a load
  bool have_logq;             // emat_lds_logq is filled (the HKY tables are staged): sites of relative rate 1 take their log(mu q_ab) from it
  bool rng_has_spare;         // rng_spare holds the second 64-bit half of the last Philox block, not yet consumed
  // A move that wants work done by the whole wave (candidate scan and study of an SPR move) parks itself: `phase` says
  // where it resumes, `svc` what the wave is to do meanwhile, `frame` points at the move's state in the scratch arena.
  uint8_t phase, svc;
  // RNG (Philox4x32-10; one 128-bit block per draw)
  uint64_t rng_key, rng_ctr;
  uint64_t rng_base;          // emat_lds_rng holds the blocks of counters rng_base .. rng_base + k_rng_blocks - 1 (rng_fill)
  uint64_t rng_spare;
  uint8_t* frame;
  double mu_prop;             // effective JC69 rate of the current SPR move (subrun.cpp:502,710)
  // scratch: a small LDS arena first (A), the part's HBM scratch region as overflow (offsets from G)
  uint8_t* A;                 // LDS arena base (may be null)
  uint32_t a_top, a_end;      // byte offsets from A
  uint32_t sc_top;            // HBM arena bump pointer (byte offset from G)
  bool failed;
  bool mv_rng_had_spare;      // RNG position at the first draw of the current move: (mv_rng_ctr, mv_rng_had_spare), see stop_for_cells
  bool rng_short;             // a LEAF move asked for a number beyond what the wave has computed ahead (rng_next64_t<true>): it commits nothing and is run again out of line
  // statistics
  int64_t bytes;
  int64_t bytes_w;            // the part of `bytes` that is written (cells, re-timed lists, region records, re-hung nodes): roofline.algorithmic_write_bytes
  // trace of the current move
  double tr_log_mh; float tr_kind, tr_node, tr_acc;   // (kind, node and accept flag are small integers: floats hold them exactly)
  int64_t moves_left;         // moves of the current launch still to do (run_chain_loop keeps nothing in registers across a move)
  uint64_t mv_rng_ctr;
  // the run-wide cell arrays (SharedCells, emat_slab.hpp): absolute cell index; the root part has its own copies in its slab
  const double* sh_ktw; const double* sh_tsop; const int32_t* sh_nact;
};

static_assert(sizeof(Ctx) + 16 <= k_lds_ctx_bytes, "context outgrew its LDS slot (the last 16 bytes are the kernel's flag word)");
// Base pointers of the part's persistent state.
// -DEMAT_X_DIVERGENT (an experiment): what the chain would cost if its addresses were NOT wave-uniform,
// as they would be with several chains side by side in one wavefront -- every base address gets a zero the compiler cannot see
// through (a VGPR), so address arithmetic, loaded values and branches all leave the scalar unit.  Still one chain per wave.
#ifdef EMAT_X_DIVERGENT
#define EMAT_OPQ ::emat_opaque_zero()   // (defined before the first inclusion, in emat_backend.hip)
#else
#define EMAT_OPQ 0u
#endif
#if EMAT_VARIANT_LDS
EMAT_DF uint8_t* slab_at(const Ctx&, uint32_t off) { return EMAT_DYN_LDS + (off - (uint32_t)sizeof(SlabHeader)) + EMAT_OPQ; }   // slab byte `off` (beyond the header)
EMAT_DF SlabHeader* hdr_of(const Ctx&) { return (SlabHeader*)(emat_lds_hdr + EMAT_OPQ); }
EMAT_DF NodeRec* nodes_of(const Ctx&) { return (NodeRec*)(EMAT_DYN_LDS + EMAT_OPQ); }   // off_nodes == sizeof(SlabHeader), checked at launch
EMAT_DF const double* mu_of(const Ctx&) { return (const double*)emat_lds_tables; }
EMAT_DF const double* pi_of(const Ctx&) { return (const double*)emat_lds_tables + k_max_lds_partitions; }
EMAT_DF const double* q_of(const Ctx&) { return (const double*)emat_lds_tables + k_max_lds_partitions * 5; }
#if EMAT_VARIANT_LDS == 2
EMAT_DF uint8_t* heap_at(const Ctx& c, uint32_t off) { return c.G + off; }   // list offsets are slab-relative: same numbers, HBM base
#else
EMAT_DF uint8_t* heap_at(const Ctx& c, uint32_t off) { return slab_at(c, off); }
#endif
#else
EMAT_DF uint8_t* heap_at(const Ctx& c, uint32_t off) { return c.S + off; }
EMAT_DF uint8_t* slab_at(const Ctx& c, uint32_t off) { return c.S + off; }
EMAT_DF SlabHeader* hdr_of(const Ctx& c) { return c.H; }
EMAT_DF NodeRec* nodes_of(const Ctx& c) { return c.N; }
EMAT_DF const double* mu_of(const Ctx& c) { return c.mu; }
EMAT_DF const double* pi_of(const Ctx& c) { return c.pi; }
EMAT_DF const double* q_of(const Ctx& c) { return c.q; }
#endif

EMAT_D void fail_at(Ctx& c, int status, int line) {
  if (!c.failed) { c.failed = true; if (hdr_of(c)->status == 0) { hdr_of(c)->status = status; hdr_of(c)->fail_line = line; } }
}
#ifndef EMAT_FAIL
#define EMAT_FAIL(c, st) fail_at((c), (st), __LINE__)
#define EMAT_CHECK(c, cond) do { if (!(cond)) fail_at((c), ::emat::k_part_internal, __LINE__); } while (0)
#ifdef EMAT_PROFILE_PHASES
#define EMAT_SITE(line, hbm, v) atomicAdd(&::emat::g_arena_site_bytes[(line) & 2047][(hbm) ? 1 : 0], (unsigned long long)(v))
#define EMAT_COUNT(c, k, v) (((int64_t*)hdr_of(c)->reserved)[k] += (int64_t)(v))
#ifdef EMAT_X_UNLIMITED_SCANS   // probe variant: reserved[11], [12] count the scans without a limit and their ticks instead
#define EMAT_COUNT_TRIMS(c, k, v) do {} while (0)
#else
#define EMAT_COUNT_TRIMS(c, k, v) EMAT_COUNT(c, k, v)
#endif
#define EMAT_PHASE_BEGIN() long long _ph_t0 = clock64()
#define EMAT_PHASE(c, k) do { long long _t = clock64(); hdr_of(c)->phase_ticks[k] += _t - _ph_t0; _ph_t0 = _t; } while (0)
#define EMAT_TIMED(file_id) ::emat::FnTimer _fn_timer((file_id) * 2048 + (__LINE__ & 2047))
// a stretch that cannot be a block of its own (it declares what the rest of the function uses): EMAT_TIMED_BLOCK(file, name) ... EMAT_TIMED_END(name)
#define EMAT_TIMED_BLOCK(file_id, name) ::emat::FnTimer name((file_id) * 2048 + (__LINE__ & 2047))
#define EMAT_TIMED_END(name) name.stop()
#else
#define EMAT_TIMED(file_id) do {} while (0)
#define EMAT_TIMED_BLOCK(file_id, name) do {} while (0)
#define EMAT_TIMED_END(name) do {} while (0)
#define EMAT_SITE(line, hbm, v) do {} while (0)
#define EMAT_COUNT(c, k, v) do {} while (0)
#define EMAT_COUNT_TRIMS(c, k, v) do {} while (0)
#define EMAT_PHASE_BEGIN() do {} while (0)
#define EMAT_PHASE(c, k) do {} while (0)
#endif
#endif

// One copy of each transcendental per code variant: the OCML bodies are 60-150 instructions and were inlined at every call
// site, and the instruction cache of a CU pair is 64 KB for 32 resident chains that are all somewhere else in the code
// (measured: simple moves 15 % smaller, +1.5 % moves/s; taking the ten Philox rounds out of line as well costs more in
// calls than it saves in fetches).
#ifndef EMAT_INLINE_TRANSC   // bit mask: 1 log, 2 exp, 4 log1p, 8 expm1 inlined at their call sites instead of behind a call
#define EMAT_INLINE_TRANSC 0
#endif
#ifdef EMAT_X_FLOAT_TRANSC   // (experiment: what the chain would gain if the four cost a third -- single precision; parity is gone)
EMAT_DN double m_log(double x) { return (double)::logf((float)x); }
EMAT_DN double m_exp(double x) { return (double)::expf((float)x); }
EMAT_DN double m_log1p(double x) { return (double)::log1pf((float)x); }
EMAT_DN double m_expm1(double x) { return (double)::expm1f((float)x); }
#define EMAT_TRANSC_DEFINED
#endif
#ifdef EMAT_TRANSC_DEFINED
#elif EMAT_INLINE_TRANSC & 1
EMAT_DF double m_log(double x) { return ::log(x); }
#else
EMAT_DN double m_log(double x) { return ::log(x); }
#endif
#ifdef EMAT_TRANSC_DEFINED
#elif EMAT_INLINE_TRANSC & 2
EMAT_DF double m_exp(double x) { return ::exp(x); }
#else
EMAT_DN double m_exp(double x) { return ::exp(x); }
#endif
#ifdef EMAT_TRANSC_DEFINED
#elif EMAT_INLINE_TRANSC & 4
EMAT_DF double m_log1p(double x) { return ::log1p(x); }
#else
EMAT_DN double m_log1p(double x) { return ::log1p(x); }
#endif
#ifdef EMAT_TRANSC_DEFINED
#elif EMAT_INLINE_TRANSC & 8
EMAT_DF double m_expm1(double x) { return ::expm1(x); }
#else
EMAT_DN double m_expm1(double x) { return ::expm1(x); }
#endif
// ---- RNG: identical stream to the parity oracle (oracle/orc_core.hpp `Rng`) -------------------------
EMAT_D void philox4x32_10(uint64_t ctr, uint64_t key, uint32_t out[4]) {
  uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32), c2 = 0, c3 = 0;
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// (out of line: the rare draw beyond what the wave has computed ahead -- a long mutational history -- and builds without the buffer)
EMAT_DN uint64_t rng_next64_computed(Ctx& c) {
  uint32_t w[4];
  philox4x32_10(c.rng_ctr++, c.rng_key, w);
  c.rng_spare = (uint64_t)w[2] | ((uint64_t)w[3] << 32); c.rng_has_spare = true;
  return (uint64_t)w[0] | ((uint64_t)w[1] << 32);
}
// kLeaf: for a move compiled as a LEAF function (no call anywhere in it: no return address to park, hence no whole-wave save and reload around every
// move -- DESIGN.md section 8, round 6).  Its draws come from what the wave computed ahead; should that run out in the middle of the move (a long run
// of rejected node picks: about one move in 10^4) the draw says so and returns a harmless number, the move commits nothing, and the caller rewinds the
// stream and runs the same move out of line.
template <bool kLeaf> EMAT_DF uint64_t rng_next64_t(Ctx& c) {
  if (c.rng_has_spare) { c.rng_has_spare = false; return c.rng_spare; }
  if (k_rng_blocks != 0) {
    const uint64_t k = c.rng_ctr - c.rng_base;
    if (k < (uint64_t)k_rng_blocks) {
      const uint4 w = *(const uint4*)&emat_lds_rng[(uint32_t)k * 4u];
      c.rng_ctr += 1;
      c.rng_spare = (uint64_t)w.z | ((uint64_t)w.w << 32); c.rng_has_spare = true;
      return (uint64_t)w.x | ((uint64_t)w.y << 32);
    }
  }
  if (kLeaf) { c.rng_short = true; return 0x8000000000000000ull; }
  return rng_next64_computed(c);
}
EMAT_D uint64_t rng_next64(Ctx& c) { return rng_next64_t<false>(c); }
// All lanes: the blocks of the next k_rng_blocks counters (the caller synchronises the wave before and after).
EMAT_D void rng_fill(Ctx& c, int lane) {
  if (k_rng_blocks == 0) return;
  const uint64_t base = c.rng_ctr;
  if (lane < (int)k_rng_blocks) { uint32_t w[4]; philox4x32_10(base + (uint64_t)lane, c.rng_key, w); *(uint4*)&emat_lds_rng[(uint32_t)lane * 4u] = make_uint4(w[0], w[1], w[2], w[3]); }
  if (lane == 0) c.rng_base = base;
}
// the chain, between two moves: few enough blocks left that the next move might run out
#ifndef EMAT_RNG_MARGIN
#define EMAT_RNG_MARGIN 8     // blocks (two draws each) a move may use before it has to compute its own
#endif
EMAT_D bool rng_wants_fill(const Ctx& c) { return k_rng_blocks != 0 && c.rng_ctr - c.rng_base + (uint64_t)EMAT_RNG_MARGIN > (uint64_t)k_rng_blocks; }
// Rewind the stream to where the current move drew its first number (the spare half-block is recomputed, not stored).
EMAT_DN void rng_rewind_to_move_start(Ctx& c) {
  c.rng_ctr = c.mv_rng_ctr; c.rng_has_spare = c.mv_rng_had_spare;
  if (c.rng_has_spare) { uint32_t w[4]; philox4x32_10(c.rng_ctr - 1, c.rng_key, w); c.rng_spare = (uint64_t)w[2] | ((uint64_t)w[3] << 32); }
}
EMAT_D double to_co(uint64_t a) { return (double)(a >> 11) * 0x1.0p-53; }
EMAT_D double to_oo(uint64_t a) { return ((double)(a >> 12) + 0.5) * 0x1.0p-52; }
EMAT_D double to_oc(uint64_t a) { return ((double)(a >> 11) + 1.0) * 0x1.0p-53; }
EMAT_D double u01_co(Ctx& c) { return to_co(rng_next64(c)); }
EMAT_D double u01_oo(Ctx& c) { return to_oo(rng_next64(c)); }
EMAT_D double u01_oc(Ctx& c) { return to_oc(rng_next64(c)); }
EMAT_D double uniform_co(Ctx& c, double lo, double hi) { return lo + (hi - lo) * u01_co(c); }
EMAT_D double uniform_oc(Ctx& c, double lo, double hi) { return lo + (hi - lo) * u01_oc(c); }
EMAT_D int uniform_int(Ctx& c, int n) { return (int)__umul64hi(rng_next64(c), (uint64_t)n); }
template <bool kLeaf> EMAT_DF double t_u01_co(Ctx& c) { return to_co(rng_next64_t<kLeaf>(c)); }
template <bool kLeaf> EMAT_DF double t_u01_oo(Ctx& c) { return to_oo(rng_next64_t<kLeaf>(c)); }
template <bool kLeaf> EMAT_DF double t_uniform_co(Ctx& c, double lo, double hi) { return lo + (hi - lo) * t_u01_co<kLeaf>(c); }
template <bool kLeaf> EMAT_DF double t_uniform_oc(Ctx& c, double lo, double hi) { return lo + (hi - lo) * to_oc(rng_next64_t<kLeaf>(c)); }
template <bool kLeaf> EMAT_DF int t_uniform_int(Ctx& c, int n) { return (int)__umul64hi(rng_next64_t<kLeaf>(c), (uint64_t)n); }
// the four transcendentals: the shared out-of-line copy, or -- in a leaf move -- the same OCML routine inlined (the same numbers)
template <bool kLeaf> EMAT_DF double t_log(double x) { if constexpr (kLeaf) return ::log(x); else return m_log(x); }
template <bool kLeaf> EMAT_DF double t_exp(double x) { if constexpr (kLeaf) return ::exp(x); else return m_exp(x); }
template <bool kLeaf> EMAT_DF double t_log1p(double x) { if constexpr (kLeaf) return ::log1p(x); else return m_log1p(x); }
EMAT_D double gaussian(Ctx& c, double mean, double sigma) {
  uint64_t a = rng_next64(c), b = rng_next64(c);
  double u1 = to_oc(a), u2 = to_co(b);
  double r = sqrt(-2.0 * m_log(u1));
  return mean + sigma * (r * cos(6.283185307179586476925 * u2));
}
EMAT_D double exponential(Ctx& c, double rate) { return -m_log(u01_oc(c)) / rate; }
EMAT_D int poisson(Ctx& c, double lambda) {
  double u = u01_co(c);
  double p = m_exp(-lambda), F = p;
  int k = 0;
  while (u >= F && k < 100000) { ++k; p *= lambda / k; F += p; }
  return k;
}

// ---- scratch arenas (temporaries of one move) -----------------------------------------------------------
template <class T> struct SVec { T* p; int n; int cap; };
struct ScMark { uint32_t a, g; };
EMAT_D ScMark sc_mark(const Ctx& c) { ScMark m; m.a = c.a_top; m.g = c.sc_top; return m; }
EMAT_D void sc_release(Ctx& c, ScMark m) { c.a_top = m.a; c.sc_top = m.g; }
EMAT_D void sc_reset(Ctx& c) { c.a_top = 0; c.sc_top = hdr_of(c)->scratch_begin; }
EMAT_D bool sc_in_lds(const Ctx& c, const void* p) { return c.A != nullptr && (const uint8_t*)p >= c.A && (const uint8_t*)p < c.A + c.a_end; }
EMAT_D uint8_t* sc_alloc(Ctx& c, uint32_t bytes, int line = __builtin_LINE()) {
  uint32_t b = (bytes + 15u) & ~15u;
  if (c.a_top + b <= c.a_end) { uint8_t* p = c.A + c.a_top; c.a_top += b; EMAT_COUNT(c, 9, b); EMAT_SITE(line, 0, b); return p; }
  if (c.sc_top + b > hdr_of(c)->scratch_end) { EMAT_FAIL(c, k_part_overflow); return c.G + hdr_of(c)->scratch_begin; }
  EMAT_COUNT(c, 8, b); EMAT_SITE(line, 1, b);
  uint8_t* p = c.G + c.sc_top;
  c.sc_top += b;
  return p;
}
template <class T> EMAT_D SVec<T> sc_vec(Ctx& c, int cap, int line = __builtin_LINE()) {
  SVec<T> v; v.n = 0;
  uint32_t bytes = (uint32_t)cap * (uint32_t)sizeof(T);
  uint32_t b = (bytes + 15u) & ~15u;
  if (cap >= 0 && c.a_top + b <= c.a_end) { v.p = (T*)(c.A + c.a_top); v.cap = cap; c.a_top += b; EMAT_COUNT(c, 9, b); EMAT_SITE(line, 0, b); return v; }
  if (cap < 0 || c.sc_top + b > hdr_of(c)->scratch_end) { EMAT_FAIL(c, k_part_overflow); v.p = (T*)(c.G + hdr_of(c)->scratch_begin); v.cap = 0; return v; }
  EMAT_COUNT(c, 8, b); EMAT_SITE(line, 1, b);
  v.p = (T*)(c.G + c.sc_top); v.cap = cap; c.sc_top += b;
  return v;
}
template <class T> EMAT_DF void push(Ctx& c, SVec<T>& v, const T& x, int line = __builtin_LINE()) { if (v.n < v.cap) v.p[v.n++] = x; else fail_at(c, k_part_overflow, line); }   // reports the caller's line
// An open-ended vector: takes (up to `max_elems` of) the free space of one arena -- the LDS arena when it still
// has at least `min_lds_elems` elements of room, else the HBM arena -- and must be trimmed with sc_trim.
template <class T> EMAT_D SVec<T> sc_open(Ctx& c, int max_elems, int min_lds_elems) {
  SVec<T> v; v.n = 0;
  uint32_t a0 = (c.a_top + 15u) & ~15u;
  int room_a = c.a_end > a0 ? (int)((c.a_end - a0) / sizeof(T)) : 0;
  if (room_a >= min_lds_elems) { v.p = (T*)(c.A + a0); v.cap = room_a < max_elems ? room_a : max_elems; c.a_top = c.a_end; return v; }
  uint32_t g0 = (c.sc_top + 15u) & ~15u;
  int room_g = hdr_of(c)->scratch_end > g0 ? (int)((hdr_of(c)->scratch_end - g0) / sizeof(T)) : 0;
  v.p = (T*)(c.G + g0); v.cap = room_g < max_elems ? room_g : max_elems; c.sc_top = hdr_of(c)->scratch_end;
  if (v.cap <= 0) { v.cap = 0; EMAT_FAIL(c, k_part_overflow); }
  return v;
}
// An open vector in the LDS arena ran out of room: move it to the part's HBM scratch region (the rest of which it takes,
// up to `max_elems`) and give the LDS arena back.  False when it already was in HBM or HBM has no more room than it had.
template <class T> EMAT_DF bool sc_open_migrate(Ctx& c, SVec<T>& v, int max_elems) {
  if (!sc_in_lds(c, v.p)) return false;
  uint32_t g0 = (c.sc_top + 15u) & ~15u;
  int room_g = hdr_of(c)->scratch_end > g0 ? (int)((hdr_of(c)->scratch_end - g0) / sizeof(T)) : 0;
  if (room_g > max_elems) room_g = max_elems;
  if (room_g <= v.cap) return false;
  T* np = (T*)(c.G + g0);
  for (int i = 0; i < v.n; ++i) np[i] = v.p[i];
  c.a_top = (uint32_t)((uint8_t*)v.p - c.A);
  v.p = np; v.cap = room_g; c.sc_top = hdr_of(c)->scratch_end;
  return true;
}
// give back the unused tail of the most recent allocation in its arena
template <class T> EMAT_DF void sc_trim(Ctx& c, SVec<T>& v, int line = __builtin_LINE()) {
  uint32_t used = ((uint32_t)v.n * (uint32_t)sizeof(T) + 15u) & ~15u;
  if (sc_in_lds(c, v.p)) { c.a_top = (uint32_t)((uint8_t*)v.p - c.A) + used; EMAT_COUNT(c, 9, used); EMAT_SITE(line, 0, used); }
  else { c.sc_top = (uint32_t)((uint8_t*)v.p - c.G) + used; EMAT_COUNT(c, 8, used); EMAT_COUNT_TRIMS(c, 11, 1); EMAT_SITE(line, 1, used); }
  v.cap = v.n;
}
// Two containers growing towards each other inside one arena (results upwards from `lo`, a work stack downwards
// from `hi`); picks the LDS arena when it has at least `min_lds_bytes` free.
struct ScSpan { uint8_t* lo; uint8_t* hi; bool lds; bool reserved; };
// A block of the LDS arena set aside at the start of a topology move for its hottest temporaries (candidate-scan
// regions and DFS stack, the sets the scan searches per region), before the bulkier and colder graft analysis
// claims the arena.  Lives until the move's sc_reset; reused by consecutive scans of the move.
struct HotBlock { uint8_t* p; uint32_t bytes; };
EMAT_D HotBlock sc_reserve_hot(Ctx& c, uint32_t bytes) {
  HotBlock h; h.p = nullptr; h.bytes = 0;
  uint32_t a0 = (c.a_top + 15u) & ~15u;
  if (c.A != nullptr && c.a_end > a0 && c.a_end - a0 >= bytes) { h.p = c.A + a0; h.bytes = bytes; c.a_top = a0 + bytes; }
  return h;
}
EMAT_D ScSpan sc_span(Ctx& c, uint32_t min_lds_bytes) {
  ScSpan s;
  uint32_t a0 = (c.a_top + 15u) & ~15u;
  s.reserved = false;
  if (c.a_end > a0 && c.a_end - a0 >= min_lds_bytes) { s.lo = c.A + a0; s.hi = c.A + (c.a_end & ~15u); s.lds = true; return s; }
  uint32_t g0 = (c.sc_top + 15u) & ~15u;
  s.lo = c.G + g0; s.hi = c.G + (hdr_of(c)->scratch_end & ~15u); s.lds = false;
  if (s.hi < s.lo) s.hi = s.lo;
  return s;
}
EMAT_D ScSpan sc_span_hbm(Ctx& c) {
  ScSpan s; uint32_t g0 = (c.sc_top + 15u) & ~15u;
  s.reserved = false;
  s.lo = c.G + g0; s.hi = c.G + (hdr_of(c)->scratch_end & ~15u); s.lds = false;
  if (s.hi < s.lo) s.hi = s.lo;
  return s;
}
EMAT_D void sc_span_commit(Ctx& c, const ScSpan& s, uint32_t used_bytes, int line = __builtin_LINE()) {
  uint32_t u = (used_bytes + 15u) & ~15u;
  if (s.reserved) return;   // the block stays reserved
  if (s.lds) { c.a_top = (uint32_t)(s.lo - c.A) + u; EMAT_COUNT(c, 9, u); EMAT_SITE(line, 0, u); } else { c.sc_top = (uint32_t)(s.lo - c.G) + u; EMAT_COUNT(c, 8, u); EMAT_COUNT_TRIMS(c, 12, 1); EMAT_SITE(line, 1, u); }
}

// ---- persistent per-node lists in the slab heap ---------------------------------------------------------
EMAT_D uint32_t heap_alloc(Ctx& c, uint32_t bytes) {
  uint32_t b = (bytes + 15u) & ~15u;
  if (hdr_of(c)->heap_top + b > hdr_of(c)->heap_end) { EMAT_FAIL(c, k_part_overflow); return hdr_of(c)->heap_begin; }
  uint32_t off = hdr_of(c)->heap_top;
  hdr_of(c)->heap_top += b;
  return off;
}
template <class T> EMAT_D T* list_ptr(Ctx& c, const ListRef& r) { return (T*)heap_at(c, r.off); }
// Every write of a list's count goes through here: it must fit the capacity the list was given (list_reserve: at most
// k_max_list_len, or the part stops with k_part_list_limit), so no count is ever cut to 16 bits unseen.
EMAT_D void set_list_cnt(Ctx& c, ListRef& r, int n) { if ((uint32_t)n > (uint32_t)r.cap) { EMAT_FAIL(c, k_part_internal); return; } r.cnt = (uint16_t)n; }
template <class T> EMAT_D void list_reserve(Ctx& c, ListRef& r, int want) {
  if (want <= (int)r.cap) return;
  int nc = (int)r.cap * 2; if (nc < want) nc = want; if (nc < 4) nc = 4;
  if (nc > (int)k_max_list_len) { if (want > (int)k_max_list_len) { EMAT_FAIL(c, k_part_list_limit); return; } nc = (int)k_max_list_len; }   // a ListRef counts in 16 bits: stop, never wrap
  uint32_t off = heap_alloc(c, (uint32_t)nc * (uint32_t)sizeof(T));
  if (c.failed) return;
  T* dst = (T*)heap_at(c, off); const T* src = (const T*)heap_at(c, r.off);
  for (int i = 0; i < (int)r.cnt; ++i) dst[i] = src[i];
  r.off = off; r.cap = (uint16_t)nc;
}
template <class T> EMAT_D void list_push(Ctx& c, ListRef& r, const T& x) {
  list_reserve<T>(c, r, (int)r.cnt + 1);
  if (c.failed) return;
  list_ptr<T>(c, r)[r.cnt] = x; r.cnt++;
}
template <class T> EMAT_D void list_assign(Ctx& c, ListRef& r, const T* src, int n) {
  list_reserve<T>(c, r, n);
  if (c.failed) return;
  T* dst = list_ptr<T>(c, r);
  for (int i = 0; i < n; ++i) dst[i] = src[i];
  set_list_cnt(c, r, n);
}
template <class T> EMAT_D void list_erase_prefix(Ctx& c, ListRef& r, int k) {
  T* p = list_ptr<T>(c, r);
  for (int i = k; i < (int)r.cnt; ++i) p[i - k] = p[i];
  set_list_cnt(c, r, (int)r.cnt - k);
}
EMAT_D void swap_lists(ListRef& a, ListRef& b) { ListRef t = a; a = b; b = t; }

EMAT_D MutRec* muts_of(Ctx& c, int n) { return (MutRec*)heap_at(c, nodes_of(c)[n].muts.off); }
EMAT_D IvRec* miss_of(Ctx& c, int n) { return (IvRec*)heap_at(c, nodes_of(c)[n].miss.off); }
EMAT_D FsRec* mfs_of(Ctx& c, int n) { return (FsRec*)heap_at(c, nodes_of(c)[n].mfs.off); }
EMAT_D int nmuts(const Ctx& c, int n) { return (int)nodes_of(c)[n].muts.cnt; }
EMAT_D bool is_tip(const Ctx& c, int n) { return nodes_of(c)[n].child0 == k_no_node; }
EMAT_D int sibling_of(Ctx& c, int parent, int x) {
  EMAT_CHECK(c, x == nodes_of(c)[parent].child0 || x == nodes_of(c)[parent].child1);
  return x == nodes_of(c)[parent].child0 ? nodes_of(c)[parent].child1 : nodes_of(c)[parent].child0;
}
EMAT_D MutRec make_mut(uint8_t from, int site, uint8_t to, double t) { MutRec m; m.t = t; m.site = site; m.from = from; m.to = to; m.pad = 0; return m; }
EMAT_D bool mut_less(const MutRec& a, const MutRec& b) { return a.t < b.t || (a.t == b.t && a.site < b.site); }   // mutations.h:41-43
// stable insertion sort by (t, site): lists are tiny and nearly sorted
EMAT_D void sort_muts(MutRec* p, int n) {
  for (int i = 1; i < n; ++i) { MutRec x = p[i]; int j = i - 1; while (j >= 0 && mut_less(x, p[j])) { p[j + 1] = p[j]; --j; } p[j + 1] = x; }
}
EMAT_D void clamp_mut_times(MutRec* p, int n, double lo, double hi) {   // mutations.h:55-60
  for (int i = 0; i < n; ++i) { double t = p[i].t; p[i].t = t < lo ? lo : (hi < t ? hi : t); }
}

// ---- evolution model accessors (evo_model.h:35-47) ----------------------------------------------------------
// Site partition and relative rate of a site.  Both arrays live in HBM (L entries each); a chain asks for them once per mutation or
// from-state it touches, each time a dependent global load of several hundred cycles.  In the reference's default model -- one site
// partition, nu_l == 1 everywhere (run.h:256: alpha moves off) -- the answers are 0 and 1.0 and `mu * 1.0 * x` IS `mu * x` bit for bit,
// so the loads are skipped (round 6; option "no_uniform_sites" keeps them for A/B runs and for the parity tests of that path).
EMAT_DF int site_part(const Ctx& c, int l) { return c.uniform_sites ? 0 : (int)c.part[l]; }
EMAT_DF double site_nu(const Ctx& c, int l) { return c.uniform_sites ? 1.0 : c.nu[l]; }
EMAT_D double mu_nu(const Ctx& c, int l) { return mu_of(c)[site_part(c, l)] * site_nu(c, l); }
EMAT_D double q_a(const Ctx& c, int l, int a) { return -q_of(c)[site_part(c, l) * 16 + a * 5]; }
EMAT_D double q_ab(const Ctx& c, int l, int a, int b) { return q_of(c)[site_part(c, l) * 16 + a * 4 + b]; }
EMAT_D double pi_a(const Ctx& c, int l, int a) { return pi_of(c)[site_part(c, l) * 4 + a]; }
// mu nu (-q_minus + q_plus)
EMAT_D double dq(const Ctx& c, int l, int minus, int plus) { return mu_of(c)[site_part(c, l)] * site_nu(c, l) * (-q_a(c, l, minus) + q_a(c, l, plus)); }

// ---- interval-set algebra on raw sorted arrays (interval_set.h:130-138, 238-500) ---------------------------
EMAT_D bool iv_contains(const IvRec* v, int n, int l) {
  int lo = 0, hi = n;   // first interval with start > l
  while (lo < hi) { int mid = (lo + hi) >> 1; if (l < v[mid].start) hi = mid; else lo = mid + 1; }
  if (lo == 0) return false;
  return l < v[lo - 1].end;
}
EMAT_D int iv_num_sites(const IvRec* v, int n) { int r = 0; for (int i = 0; i < n; ++i) r += v[i].end - v[i].start; return r; }
// dst must have room for nA + nB intervals
EMAT_D int iv_merge(IvRec* dst, const IvRec* A, int nA, const IvRec* B, int nB) {
  int out = 0, ia = 0, ib = 0; bool inside = false; int cs = 0, ce = 0;
  while (!(ia == nA && ib == nB)) {
    bool useA = (ia == nA) ? false : (ib == nB) ? true : (A[ia].start <= B[ib].start);
    int fs = useA ? A[ia].start : B[ib].start, fe = useA ? A[ia].end : B[ib].end;
    if (!inside) { cs = fs; ce = fe; if (useA) ++ia; else ++ib; inside = true; }
    else if (fs <= ce) { ce = ce > fe ? ce : fe; if (useA) ++ia; else ++ib; }
    else { dst[out].start = cs; dst[out].end = ce; ++out; inside = false; }
  }
  if (inside) { dst[out].start = cs; dst[out].end = ce; ++out; }
  return out;
}
// dst must have room for nA + nB intervals
EMAT_D int iv_intersect(IvRec* dst, const IvRec* A, int nA, const IvRec* B, int nB) {
  int out = 0, ia = 0, ib = 0;
  while (ia != nA && ib != nB) {
    int so = A[ia].start > B[ib].start ? A[ia].start : B[ib].start;
    int eo = A[ia].end < B[ib].end ? A[ia].end : B[ib].end;
    if (so < eo) { dst[out].start = so; dst[out].end = eo; ++out; }
    if (A[ia].end <= B[ib].end) ++ia; else ++ib;
  }
  return out;
}
EMAT_D bool iv_intersects(const IvRec* A, int nA, const IvRec* B, int nB) {
  int ia = 0, ib = 0;
  while (ia != nA && ib != nB) {
    int so = A[ia].start > B[ib].start ? A[ia].start : B[ib].start;
    int eo = A[ia].end < B[ib].end ? A[ia].end : B[ib].end;
    if (so < eo) return true;
    if (A[ia].end <= B[ib].end) ++ia; else ++ib;
  }
  return false;
}
// dst = A - B; dst must have room for nA + nB intervals
EMAT_D int iv_subtract(IvRec* dst, const IvRec* A, int nA, const IvRec* B, int nB) {
  if (nA == 0) return 0;
  int out = 0, ia = 0, ib = 0;
  int cs = A[0].start, ce = A[0].end;
  while (ia != nA) {
    bool next = false;
    if (ib == nB) { dst[out].start = cs; dst[out].end = ce; ++out; next = true; }
    else {
      int bs = B[ib].start, be = B[ib].end;
      if (bs < cs) {
        if (be <= cs) ++ib;
        else if (be < ce) { cs = be; ++ib; }
        else next = true;
      } else if (bs < ce) {
        if (cs < bs) { dst[out].start = cs; dst[out].end = bs; ++out; }
        if (be < ce) { cs = be; ++ib; }
        else next = true;
      } else { dst[out].start = cs; dst[out].end = ce; ++out; next = true; }
    }
    if (next) { ++ia; if (ia != nA) { cs = A[ia].start; ce = A[ia].end; } }
  }
  return out;
}
// scratch-allocated results
EMAT_D SVec<IvRec> iv_subtract_sc(Ctx& c, const IvRec* A, int nA, const IvRec* B, int nB) {
  SVec<IvRec> r = sc_vec<IvRec>(c, nA + nB + 1);
  if (!c.failed) r.n = iv_subtract(r.p, A, nA, B, nB);
  sc_trim(c, r);
  return r;
}
EMAT_D SVec<IvRec> iv_copy_sc(Ctx& c, const IvRec* A, int nA) {
  SVec<IvRec> r = sc_vec<IvRec>(c, nA);
  if (!c.failed) { for (int i = 0; i < nA; ++i) r.p[i] = A[i]; r.n = nA; }
  return r;
}

// ---- from-state lists (sorted by site) and node missation maps (mutations.h:184-232) ---------------------------
EMAT_D int fs_lower_bound(const FsRec* v, int n, int l) { int lo = 0, hi = n; while (lo < hi) { int mid = (lo + hi) >> 1; if (v[mid].site < l) lo = mid + 1; else hi = mid; } return lo; }
EMAT_D bool miss_contains(Ctx& c, int node, int l) { return iv_contains(miss_of(c, node), (int)nodes_of(c)[node].miss.cnt, l); }
EMAT_D int miss_get_from_state(Ctx& c, int node, int l) {
  const FsRec* v = mfs_of(c, node); int n = (int)nodes_of(c)[node].mfs.cnt;
  int k = fs_lower_bound(v, n, l);
  return (k < n && v[k].site == l) ? (int)v[k].state : (int)c.ref[l];
}
EMAT_DN void miss_set_from_state(Ctx& c, int node, int l, int from) {
  ListRef& r = nodes_of(c)[node].mfs;
  FsRec* v = mfs_of(c, node); int n = (int)r.cnt;
  int k = fs_lower_bound(v, n, l);
  bool present = (k < n && v[k].site == l);
  if (from != (int)c.ref[l]) {
    if (present) { v[k].state = (uint8_t)from; return; }
    list_reserve<FsRec>(c, r, n + 1);
    if (c.failed) return;
    v = mfs_of(c, node);
    for (int i = n; i > k; --i) v[i] = v[i - 1];
    v[k].site = l; v[k].state = (uint8_t)from; v[k].pad[0] = v[k].pad[1] = v[k].pad[2] = 0;
    set_list_cnt(c, r, n + 1);
  } else if (present) {
    for (int i = k; i + 1 < n; ++i) v[i] = v[i + 1];
    set_list_cnt(c, r, n - 1);
  }
}

// ---- site deltas: sorted array of {site, from, to} (site_deltas.h:43-154) -----------------------------------------
struct SdRec { int32_t site; uint8_t from, to; uint16_t pad; };
EMAT_D int sd_lower_bound(const SdRec* v, int n, int l) { int lo = 0, hi = n; while (lo < hi) { int mid = (lo + hi) >> 1; if (v[mid].site < l) lo = mid + 1; else hi = mid; } return lo; }
EMAT_D bool sd_contains(const SVec<SdRec>& v, int l) { int k = sd_lower_bound(v.p, v.n, l); return k < v.n && v.p[k].site == l; }
EMAT_DF void sd_insert_at(Ctx& c, SVec<SdRec>& v, int k, int site, int from, int to) {
  if (v.n >= v.cap) { EMAT_FAIL(c, k_part_overflow); return; }
  for (int i = v.n; i > k; --i) v.p[i] = v.p[i - 1];
  v.p[k].site = site; v.p[k].from = (uint8_t)from; v.p[k].to = (uint8_t)to; v.p[k].pad = 0; v.n++;
}
EMAT_DF void sd_erase_at(SVec<SdRec>& v, int k) { for (int i = k; i + 1 < v.n; ++i) v.p[i] = v.p[i + 1]; v.n--; }
// (The out-of-line helpers that edit a scratch vector take its header BY VALUE and return it: a header whose address is handed to a
// function the compiler does not inline has to live in private memory -- scratch, one 64-byte line per dword with one lane active, and an
// L1 / L2 round trip per access where a register costs nothing.  The wrappers with the old signatures are inlined into their callers.)
EMAT_DN SVec<SdRec> sd_push_front_v(Ctx& c, SVec<SdRec> v, int site, int from, int to) {   // site_deltas.h:43-65
  int k = sd_lower_bound(v.p, v.n, site);
  if (k < v.n && v.p[k].site == site) {
    EMAT_CHECK(c, to == (int)v.p[k].from);
    v.p[k].from = (uint8_t)from;
    if (v.p[k].from == v.p[k].to) sd_erase_at(v, k);
  } else sd_insert_at(c, v, k, site, from, to);
  return v;
}
EMAT_DF void sd_push_front(Ctx& c, SVec<SdRec>& v, int site, int from, int to) { v = sd_push_front_v(c, v, site, from, to); }
EMAT_DF void sd_pop_front(Ctx& c, SVec<SdRec>& v, int site, int from, int to) { sd_push_front(c, v, site, to, from); }
EMAT_DN SVec<SdRec> sd_push_back_v(Ctx& c, SVec<SdRec> v, int site, int from, int to) {    // site_deltas.h:88-110
  int k = sd_lower_bound(v.p, v.n, site);
  if (k < v.n && v.p[k].site == site) {
    EMAT_CHECK(c, from == (int)v.p[k].to);
    v.p[k].to = (uint8_t)to;
    if (v.p[k].from == v.p[k].to) sd_erase_at(v, k);
  } else sd_insert_at(c, v, k, site, from, to);
  return v;
}
EMAT_DF void sd_push_back(Ctx& c, SVec<SdRec>& v, int site, int from, int to) { v = sd_push_back_v(c, v, site, from, to); }

// ---- genetic-likelihood calculus (phylo_tree_calc.h:121-206, phylo_tree_calc.cpp:41-118,406-456) ----------------
// (Issuing the gathers four intervals at a time, padded with repeats, was measured in round 4: 1.3 % slower on inner-node
// displacements -- a tip's two or three gaps are cheaper as the short serial loop the compiler makes of this.)
EMAT_D double delta_lambda_across_missations(Ctx& c, const IvRec* iv, int niv, const FsRec* fs, int nfs) {   // h:121-138
  double r = 0.0;
  for (int i = 0; i < niv; ++i) r -= c.cumQ[iv[i].end] - c.cumQ[iv[i].start];
  for (int i = 0; i < nfs; ++i) { int l = fs[i].site; r -= mu_of(c)[site_part(c, l)] * site_nu(c, l) * (q_a(c, l, fs[i].state) - q_a(c, l, c.ref[l])); }
  return r;
}
EMAT_D double delta_lambda_across_node_missations(Ctx& c, int node) {
  return delta_lambda_across_missations(c, miss_of(c, node), (int)nodes_of(c)[node].miss.cnt, mfs_of(c, node), (int)nodes_of(c)[node].mfs.cnt);
}
EMAT_D double delta_lambda_across_branch(Ctx& c, int node) {   // h:140-155
  double r = 0.0;
  const MutRec* m = muts_of(c, node); int nm = nmuts(c, node);
  for (int i = 0; i < nm; ++i) { int l = m[i].site; r += mu_of(c)[site_part(c, l)] * site_nu(c, l) * (q_a(c, l, m[i].to) - q_a(c, l, m[i].from)); }
  r += delta_lambda_across_node_missations(c, node);
  return r;
}
EMAT_DN double calc_lambda_at_node(Ctx& c, int node) {   // cpp:406-418
  double r = c.cumQ[c.L];
  for (int cur = node; cur != k_no_node; cur = nodes_of(c)[cur].parent) r += delta_lambda_across_branch(c, cur);
  return r;
}
// log(mu_l nu_l q_ab): from the table where the site's relative rate is exactly 1 (mu * 1.0 * q is mu * q bit for bit), else taken
EMAT_D double log_mu_nu_q(const Ctx& c, int l, int a, int b) {
  const int pa = site_part(c, l); const double nu = site_nu(c, l);
  if (c.have_logq && nu == 1.0) return emat_lds_logq[pa * 16 + a * 4 + b];
  return m_log(mu_of(c)[pa] * nu * q_of(c)[pa * 16 + a * 4 + b]);
}
EMAT_D double branch_log_G(const Ctx& c, double t_P, double t_X, double lambda_X, const MutRec* m, int nm) {   // h:185-206
  double r = -lambda_X * (t_X - t_P);
  for (int i = nm - 1; i >= 0; --i) {
    int l = m[i].site;
    r -= mu_of(c)[site_part(c, l)] * site_nu(c, l) * (q_a(c, l, m[i].from) - q_a(c, l, m[i].to)) * (m[i].t - t_P);
    r += log_mu_nu_q(c, l, m[i].from, m[i].to);
  }
  return r;
}
EMAT_DN int calc_site_state_at(Ctx& c, int branch, double t, int l) {   // cpp:108-118
  for (int cur = branch; cur != k_no_node; cur = nodes_of(c)[cur].parent) {
    const MutRec* m = muts_of(c, cur);
    for (int i = nmuts(c, cur) - 1; i >= 0; --i) { if (m[i].t > t) continue; if (m[i].site == l) return m[i].to; }
  }
  return c.ref[l];
}
EMAT_D bool is_site_missing_at(Ctx& c, int node, int l) {   // cpp:58-65
  for (int cur = node; cur != k_no_node; cur = nodes_of(c)[cur].parent) if (miss_contains(c, cur, l)) return true;
  return false;
}
// cpp:41-56; result in scratch
EMAT_DN SVec<IvRec> reconstruct_missing_sites_at(Ctx& c, int node) { EMAT_TIMED(0);
  int total = 0;
  for (int cur = node; cur != k_no_node; cur = nodes_of(c)[cur].parent) total += (int)nodes_of(c)[cur].miss.cnt;
  SVec<IvRec> a = sc_vec<IvRec>(c, total + 1), b = sc_vec<IvRec>(c, total + 1);
  if (c.failed) return a;
  IvRec* so_far = a.p; IvRec* other = b.p; int n = 0;
  for (int cur = node; cur != k_no_node; cur = nodes_of(c)[cur].parent) {
    const int cnt = (int)nodes_of(c)[cur].miss.cnt;
    if (cnt == 0) continue;   // (most inner nodes miss nothing of their own: the union with nothing is the set as it stands, not a copy of it)
    int k = iv_merge(other, so_far, n, miss_of(c, cur), cnt);
    IvRec* t = so_far; so_far = other; other = t; n = k;
  }
  SVec<IvRec> r; r.p = so_far; r.n = n; r.cap = total + 1;
  return r;
}
EMAT_D bool descends_from(Ctx& c, int X, int A) {   // phylo_tree.cpp:292-299
  if (A == k_no_node) return true;
  for (int cur = X; cur != k_no_node; cur = nodes_of(c)[cur].parent) {
    if (cur == A) return true;
    if (nodes_of(c)[cur].t < nodes_of(c)[A].t) return false;
  }
  return false;
}
EMAT_DN int find_MRCA_of(Ctx& c, int P, int Q) {   // phylo_tree.cpp:204-280
  if (P == k_no_node) return P;
  if (Q == k_no_node) return Q;
  int guard = 0;
  while (P != Q && guard++ < (1 << 28)) {
    double tP = nodes_of(c)[P].t, tQ = nodes_of(c)[Q].t;
    if (tP > tQ) { P = nodes_of(c)[P].parent; EMAT_CHECK(c, P != k_no_node); if (P == k_no_node) return Q; }
    else if (tP < tQ) { Q = nodes_of(c)[Q].parent; EMAT_CHECK(c, Q != k_no_node); if (Q == k_no_node) return P; }
    else if (is_tip(c, P)) { P = nodes_of(c)[P].parent; if (P == k_no_node) return Q; }
    else if (is_tip(c, Q)) { Q = nodes_of(c)[Q].parent; if (Q == k_no_node) return P; }
    else {
      // equal times, distinct inner nodes (rare): deepest common node of the two root paths
      int dP = 0, dQ = 0;
      for (int x = P; x != k_no_node; x = nodes_of(c)[x].parent) ++dP;
      for (int x = Q; x != k_no_node; x = nodes_of(c)[x].parent) ++dQ;
      int a = P, b = Q;
      while (dP > dQ) { a = nodes_of(c)[a].parent; --dP; }
      while (dQ > dP) { b = nodes_of(c)[b].parent; --dQ; }
      while (a != b) { a = nodes_of(c)[a].parent; b = nodes_of(c)[b].parent; }
      return a;
    }
  }
  return P;
}

// ---- population models (pop_model.cpp:18-145, 181-204, 247-330, 525-560) ------------------------------------------
// lower_bound on the knots: the first k with x[k] >= t, 0 .. M+1 (pop_model.cpp: std::ranges::lower_bound).  Instead of
// a binary search (a chain of dependent loads) the index is guessed from the mean knot spacing and then corrected by
// comparing with the neighbouring knots, which yields exactly the lower_bound result for any knot placement and
// costs two independent loads when the knots are evenly spaced (the usual Skygrid set-up).
EMAT_DF int skygrid_interval(const PopTable& p, double t) {
  const int n = p.skygrid_num_knots;
  double g = (t - p.skygrid_x[0]) * p.skygrid_inv_dx;
  int k = g > 0.0 ? (g < (double)n ? (int)g : n) : 0;   // NaN -> 0
  while (k > 0 && !(p.skygrid_x[k - 1] < t)) --k;
  while (k < n && p.skygrid_x[k] < t) ++k;
  return k;
}
EMAT_DF double skygrid_log_N(const PopTable& p, double t) {
  int k = skygrid_interval(p, t), M = p.skygrid_num_knots - 1;
  if (k == 0) return p.skygrid_gamma[0];
  if (k > M) return p.skygrid_gamma[M];
  if (p.skygrid_type == 1) return p.skygrid_gamma[k];
  double cc = (t - p.skygrid_x[k - 1]) / (p.skygrid_x[k] - p.skygrid_x[k - 1]);
  return (1 - cc) * p.skygrid_gamma[k - 1] + cc * p.skygrid_gamma[k];
}
EMAT_DF double pop_at_time(const PopTable& p, double t) {
  if (p.kind == 0) return p.p[0];
  if (p.kind == 1) { double v = p.p[1] * m_exp((t - p.p[0]) * p.p[2]); return p.p[3] > v ? p.p[3] : v; }
  return m_exp(skygrid_log_N(p, t));
}
// log(N(t_new) / N(t_old)) (very_scalable_coalescent.cpp:323).  For the skygrid N = exp(log_N), so the ratio is formed in
// log space directly; the other models go through pop_at_time's formula as the reference does.  Written for the lane that
// runs a chain, and read through the scalar unit.  The model's table is a few hundred bytes
// of read-only global memory that every wavefront of the device reads: as ordinary loads, its fields, the pointers to the
// knots and the knots themselves are a chain of a dozen dependent vector loads per time (measured: 16 % of all chain time
// at C4, section 8 of DESIGN.md).  Only one lane is active here, so the time and every address are uniform by construction;
// saying so (readfirstlane) and reading through the constant address space turns the chain into s_load from the scalar
// cache.  (Same operations in the same order as skygrid_log_N / pop_at_time.)
EMAT_DF double uniform_f64(double v) {
  const uint64_t a = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
EMAT_DF double skygrid_log_N_uniform(const EMAT_CONST_AS PopTable* p, double t) {   // skygrid_interval + skygrid_log_N
  const EMAT_CONST_AS double* x = uniform_const_ptr((const double*)p->skygrid_x);
  const EMAT_CONST_AS double* ga = uniform_const_ptr((const double*)p->skygrid_gamma);
  const int n = p->skygrid_num_knots, M = n - 1;
  const double g = (t - x[0]) * p->skygrid_inv_dx;
  int k = __builtin_amdgcn_readfirstlane(g > 0.0 ? (g < (double)n ? (int)g : n) : 0);   // NaN -> 0
  while (k > 0 && !(x[k - 1] < t)) --k;
  while (k < n && x[k] < t) ++k;
  if (k == 0) return ga[0];
  if (k > M) return ga[M];
  if (p->skygrid_type == 1) return ga[k];
  const double cc = (t - x[k - 1]) / (x[k] - x[k - 1]);
  return (1 - cc) * ga[k - 1] + cc * ga[k];
}
template <bool kLeaf = false> EMAT_DF double log_pop_ratio_uniform(const PopTable* pp, double t_new, double t_old) {
  const EMAT_CONST_AS PopTable* p = uniform_const_ptr(pp);
  const int kind = p->kind;
  if (kind == 0) return 0.0;
  t_new = uniform_f64(t_new); t_old = uniform_f64(t_old);
  if (kind == 2) return skygrid_log_N_uniform(p, t_new) - skygrid_log_N_uniform(p, t_old);
  const double t0 = p->p[0], n0 = p->p[1], gr = p->p[2], floor_n = p->p[3];
  double a = n0 * t_exp<kLeaf>((t_new - t0) * gr); a = floor_n > a ? floor_n : a;
  double b = n0 * t_exp<kLeaf>((t_old - t0) * gr); b = floor_n > b ? floor_n : b;
  return t_log<kLeaf>(a / b);
}
EMAT_D double exp_unclamped_int(const PopTable& p, double a, double b) { double n0 = p.p[1], g = p.p[2], t0 = p.p[0]; return n0 / g * m_exp(g * (a - t0)) * m_expm1(g * (b - a)); }
EMAT_DN double skygrid_log_int_N(const PopTable& p, double a, double b) {   // pop_model.cpp:247-330 with gamma_eff = gamma
  const double* x = p.skygrid_x; const double* ge = p.skygrid_gamma;
  int M = p.skygrid_num_knots - 1;
  int ka = skygrid_interval(p, a), kb = skygrid_interval(p, b);
  int kka = ka - 1 > 0 ? ka - 1 : 0, kkb = kb < M ? kb : M;
  double bias = -k_inf;
  for (int k = kka; k <= kkb; ++k) bias = bias > ge[k] ? bias : ge[k];
  double result = 0.0;
  for (int k = ka; k <= kb; ++k) {
    double lo = k > 0 ? (a > x[k - 1] ? a : x[k - 1]) : a;
    double hi = k <= M ? (b < x[k] ? b : x[k]) : b;
    if (k == 0) result += m_exp(-bias + ge[0]) * (hi - lo);
    else if (k == M + 1) result += m_exp(-bias + ge[M]) * (hi - lo);
    else if (p.skygrid_type == 1) result += m_exp(-bias + ge[k]) * (hi - lo);
    else if (ge[k] == ge[k - 1]) result += m_exp(-bias + ge[k]) * (hi - lo);
    else {
      double c_lo = (lo - x[k - 1]) / (x[k] - x[k - 1]), c_hi = (hi - x[k - 1]) / (x[k] - x[k - 1]);
      double G_lo = (1 - c_lo) * ge[k - 1] + c_lo * ge[k], G_hi = (1 - c_hi) * ge[k - 1] + c_hi * ge[k];
      double D = G_hi - G_lo;
      if (D == 0.0) result += m_exp(-bias + G_lo) * (hi - lo);
      else result += m_exp(-bias + G_lo) * (hi - lo) * (m_expm1(D) / D);
    }
  }
  return m_log(result) + bias;
}
EMAT_NOTAIL EMAT_DN double pop_integral(const PopTable& p, double a, double b) {
  if (p.kind == 0) return (b - a) * p.p[0];
  if (p.kind == 1) {   // pop_model.cpp:43-91
    double n0 = p.p[1], g = p.p[2], t0 = p.p[0], min_pop = p.p[3], t_c = p.t_c;
    if (min_pop == 0.0) return g == 0.0 ? (b - a) * n0 : exp_unclamped_int(p, a, b);
    if (g == 0.0) return (b - a) * (min_pop > n0 ? min_pop : n0);
    if (g > 0.0) {
      if (b <= t_c) return (b - a) * min_pop;
      if (a >= t_c) return exp_unclamped_int(p, a, b);
      return (t_c - a) * min_pop + n0 / g * m_exp(g * (t_c - t0)) * m_expm1(g * (b - t_c));
    }
    if (a >= t_c) return (b - a) * min_pop;
    if (b <= t_c) return exp_unclamped_int(p, a, b);
    return n0 / g * m_exp(g * (a - t0)) * m_expm1(g * (t_c - a)) + (b - t_c) * min_pop;
  }
  return m_exp(skygrid_log_int_N(p, a, b));
}

// ---- per-part coalescent prior (very_scalable_coalescent.cpp:14-79, 259-459) ----------------------------------------
// The part stores only its window of cells [cell_first, cell_first + n_cells): outside it k_bar_p is
// identically zero for the whole residency, so those cells contribute nothing (cpp:355-386).  The window starts at the
// cell of the part's latest node time in full precision or at the reference's first_cell, whichever comes first
// (emat_host_model.hpp, CoalBuilder::local_grid).
struct Cells {
  double* kbar_p; double* ktw_p;                                            // the part's own arrays, window index
  double* ktw; double* popsize; double* ts_over_pop; int32_t* nactive;      // root part only: its own copies, window index
  const EMAT_CONST_AS double* sh_ktw; const EMAT_CONST_AS double* sh_tsop; const EMAT_CONST_AS int32_t* sh_nact;   // every other part: the device's table, absolute index
  int first; bool local;
};
EMAT_D Cells cells_of(Ctx& c) {
  Cells k; const int cap = hdr_of(c)->cell_cap;
  double* base = (double*)slab_at(c, hdr_of(c)->off_cells);
  k.kbar_p = base; k.ktw_p = base + cap; k.first = hdr_of(c)->cell_first; k.local = c.includes_run_root;
  if (k.local) { k.ktw = base + 2 * cap; k.popsize = base + 3 * cap; k.ts_over_pop = base + 4 * cap; k.nactive = (int32_t*)(base + 5 * cap); k.sh_ktw = nullptr; k.sh_tsop = nullptr; k.sh_nact = nullptr; }
  else { k.ktw = nullptr; k.popsize = nullptr; k.ts_over_pop = nullptr; k.nactive = nullptr; k.sh_ktw = uniform_const_ptr(c.sh_ktw); k.sh_tsop = uniform_const_ptr(c.sh_tsop); k.sh_nact = uniform_const_ptr(c.sh_nact); }
  return k;
}
// (A fast path through the reciprocal of t_step, falling back to the division near cell boundaries, was measured in round 4: 6 %
// SLOWER on inner-node displacements -- the division is a dozen straight-line instructions, the shortcut a branch.)
// (A product with 1 / t_step here and in the two functions below -- guarded so that it floors like the quotient -- takes 60 vector
// instructions out of a displacement move and was measured at 464.0 against 464.5 M moves/s: the division stays, as the reference has it.)
EMAT_D int cell_for(const Ctx& c, double t) { return (int)floor((hdr_of(c)->t_ref - t) / hdr_of(c)->t_step); }
EMAT_D double cell_ubound(const Ctx& c, int cell) { return hdr_of(c)->t_ref - hdr_of(c)->t_step * cell; }
EMAT_D double cell_lbound(const Ctx& c, int cell) { return cell_ubound(c, cell) - hdr_of(c)->t_step; }
// cpp:259-299 (only the root part may grow, towards the past)
// `kGrow` = the caller may be running the part that holds the run's root.  Parts that do not can never append cells, and
// their simple moves are compiled with kGrow = false so that they contain no call at all (leaf functions: no return
// address or frame pointer to save).
EMAT_NOTAIL EMAT_DN void coal_grow(Ctx& c, int cell);
template <bool kGrow = true> EMAT_DF void coal_ensure_space(Ctx& c, double t) {
  int cell = cell_for(c, t);
  if (kGrow) { if (cell >= hdr_of(c)->n_cells_total && c.includes_run_root) coal_grow(c, cell); }   // rare: the root moved past the grid
  if (cell < hdr_of(c)->cell_first || cell >= hdr_of(c)->n_cells_total) EMAT_FAIL(c, k_part_internal);
}
EMAT_NOTAIL EMAT_DN void coal_grow(Ctx& c, int cell) {
  {
    Cells k = cells_of(c);
    while (hdr_of(c)->n_cells_total <= cell) {
      int i = hdr_of(c)->n_cells_total, w = i - hdr_of(c)->cell_first;
      if (w >= hdr_of(c)->cell_cap) { EMAT_FAIL(c, k_part_cell_overflow); return; }
      double popsize_bar_i = pop_integral(*c.pop, cell_lbound(c, i), cell_ubound(c, i)) / hdr_of(c)->t_step;
      double sigma = sqrt(popsize_bar_i / hdr_of(c)->t_step);
      double ktw = gaussian(c, 0.0, sigma);
      k.popsize[w] = popsize_bar_i; k.ts_over_pop[w] = hdr_of(c)->t_step / popsize_bar_i; k.nactive[w] = 1; k.kbar_p[w] = 1.0; k.ktw_p[w] = ktw; k.ktw[w] = ktw;
      hdr_of(c)->n_cells_total = i + 1; hdr_of(c)->n_cells = w + 1;
    }
  }
}
// The root part's grid grows towards the past when a move puts a coalescence beyond its last cell (coal_grow), and the
// slab holds room for `cell_cap` cells.  The moves that can do that ask BEFORE they change anything whether the new time
// fits; when it does not, the part stops as if the move had never started -- its bookkeeping undone, the RNG rewound to
// the move's first draw -- with k_part_need_cells, the host re-materialises the part with more cells, and the move runs
// again from the same stream position: the chain stays the chain the reference's unbounded vectors would give.
EMAT_DF bool coal_needs_cells(const Ctx& c, double t) {
  if (!c.includes_run_root) return false;
  const int cell = cell_for(c, t);
  return cell >= hdr_of(c)->n_cells_total && cell - hdr_of(c)->cell_first >= hdr_of(c)->cell_cap;
}
EMAT_DN void stop_for_cells(Ctx& c, int kind) {
  rng_rewind_to_move_start(c);
  hdr_of(c)->proposed[kind]--;
  c.failed = true;
  if (hdr_of(c)->status == 0) hdr_of(c)->status = k_part_need_cells;
}
// cpp:37-79 on k_bar_p
EMAT_DF void coal_add_interval(Ctx& c, double t_start, double t_end, double delta_k) {
  if (c.failed) return;
  if (t_start < t_end) { double t = t_start; t_start = t_end; t_end = t; }
  Cells k = cells_of(c);
  const int first = hdr_of(c)->cell_first;
  int cell_start = cell_for(c, t_start);
  int cell_end = hdr_of(c)->n_cells_total - 1;
  if (t_end != cell_lbound(c, cell_end)) cell_end = cell_for(c, t_end);
  if (cell_start < first || cell_end >= hdr_of(c)->n_cells_total || cell_start > cell_end) { EMAT_FAIL(c, k_part_internal); return; }
  const double ts = hdr_of(c)->t_step;
  if (cell_start == cell_end) k.kbar_p[cell_start - first] += delta_k * (t_start - t_end) / ts;
  else {
    k.kbar_p[cell_start - first] += delta_k * (t_start - cell_lbound(c, cell_start)) / ts;
    k.kbar_p[cell_end - first] += delta_k * (cell_ubound(c, cell_end) - t_end) / ts;
    for (int i = cell_start + 1; i < cell_end; ++i) k.kbar_p[i - first] += delta_k;
  }
  c.bytes += 8 * (int64_t)(cell_end - cell_start + 1); c.bytes_w += 8 * (int64_t)(cell_end - cell_start + 1);
}
EMAT_DF double coal_cell_term(const Ctx& c, const Cells& k, int w, double new_k, double old_k) {
  double na, tsop, ktw;   // num_active_parts, t_step / popsize_bar (the same double, divided when the cell was made), k_twiddle_bar
  if (k.local) { na = (double)k.nactive[w]; tsop = k.ts_over_pop[w]; ktw = k.ktw[w]; }
  else { const int i = __builtin_amdgcn_readfirstlane(k.first + w); na = (double)k.sh_nact[i]; tsop = k.sh_tsop[i]; ktw = k.sh_ktw[i]; }
  return tsop * (
      +0.5 * (new_k * new_k - old_k * old_k) * na
      - (k.ktw_p[w] * na - ktw + 0.5) * (new_k - old_k));
}
// cpp:388-459
template <bool kGrow = true> EMAT_DF double coal_delta_on_add_interval(Ctx& c, double min_t, double max_t, double delta_k) { EMAT_TIMED(0);
  { int cm = cell_for(c, max_t); if (cm < hdr_of(c)->cell_first || cm >= hdr_of(c)->n_cells_total) { EMAT_FAIL(c, k_part_internal); return 0.0; } }
  coal_ensure_space<kGrow>(c, min_t);
  if (c.failed) return 0.0;
  if (min_t == max_t) return 0.0;
  Cells k = cells_of(c);
  const int first = hdr_of(c)->cell_first; const double ts = hdr_of(c)->t_step;
  int cell_start = cell_for(c, max_t), cell_end = cell_for(c, min_t);
  double d = 0.0;
  if (cell_start == cell_end) {
    int w = cell_start - first;
    double old_k = k.kbar_p[w], new_k = old_k + delta_k * (max_t - min_t) / ts;
    d -= coal_cell_term(c, k, w, new_k, old_k);
  } else {
    int i = cell_start;
    double dt_start = max_t - cell_lbound(c, cell_start);
    double dt_end = cell_ubound(c, cell_end) - min_t;
    double old_k = k.kbar_p[i - first], new_k = old_k + delta_k * dt_start / ts;
    d -= coal_cell_term(c, k, i - first, new_k, old_k);
    for (++i; i < cell_end; ++i) { old_k = k.kbar_p[i - first]; new_k = old_k + delta_k; d -= coal_cell_term(c, k, i - first, new_k, old_k); }
    old_k = k.kbar_p[i - first]; new_k = old_k + delta_k * dt_end / ts;
    d -= coal_cell_term(c, k, i - first, new_k, old_k);
  }
  c.bytes += 36 * (int64_t)(cell_end - cell_start + 1);
  EMAT_COUNT(c, 13, cell_end - cell_start + 1); EMAT_COUNT(c, 14, 1);
  return d;
}
template <bool kGrow = true, bool kLeaf = false> EMAT_DF double coal_delta_displace_coalescence(Ctx& c, double old_t, double new_t) { EMAT_TIMED(0);   // cpp:310-326
  double d = (old_t <= new_t) ? coal_delta_on_add_interval<kGrow>(c, old_t, new_t, -1.0) : coal_delta_on_add_interval<kGrow>(c, new_t, old_t, +1.0);
  { EMAT_TIMED(0); d -= log_pop_ratio_uniform<kLeaf>(c.pop, new_t, old_t); }
  return d;
}
template <bool kGrow = true> EMAT_DF double coal_delta_displace_tip(Ctx& c, double old_t, double new_t) {           // cpp:337-353
  return (old_t <= new_t) ? coal_delta_on_add_interval<kGrow>(c, old_t, new_t, +1.0) : coal_delta_on_add_interval<kGrow>(c, new_t, old_t, -1.0);
}
template <bool kGrow = true> EMAT_DF void coal_coalescence_displaced(Ctx& c, double old_t, double new_t) {           // cpp:301-308
  coal_ensure_space<kGrow>(c, new_t);
  coal_add_interval(c, old_t, new_t, old_t <= new_t ? -1.0 : +1.0);
}
template <bool kGrow = true> EMAT_DF void coal_tip_displaced(Ctx& c, double old_t, double new_t) {                   // cpp:328-335
  coal_ensure_space<kGrow>(c, new_t);
  coal_add_interval(c, old_t, new_t, old_t <= new_t ? +1.0 : -1.0);
}

// ---- incomplete gamma (replaces Boost gamma_q / gamma_q_inv used at spr_study.cpp:368,463,544;
//      series / modified-Lentz continued fraction, inverse by Halley steps) --------------------------------------------
// the continued fraction's value h: Q(a, x) = exp(-x + a log x - lgamma(a)) * h for x >= a + 1
EMAT_DN double gamma_q_fraction(double a, double x) {
  const double FPMIN = 1e-300;
  double b = x + 1.0 - a, cc = 1.0 / FPMIN, d = 1.0 / b, h = d;
  for (int i = 1; i < 100000; ++i) {
    double an = -i * (i - a);
    b += 2.0;
    d = an * d + b; if (fabs(d) < FPMIN) d = FPMIN;
    cc = b + an / cc; if (fabs(cc) < FPMIN) cc = FPMIN;
    d = 1.0 / d;
    double del = d * cc; h *= del;
    if (fabs(del - 1.0) < 1e-16) break;
  }
  return h;
}
EMAT_DN double gamma_q(double a, double x) {
  if (x == 0.0) return 1.0;
  if (isinf(x)) return 0.0;
  const double lg = lgamma(a);
  if (x < a + 1.0) {
    double ap = a, sum = 1.0 / a, del = sum;
    for (int n = 0; n < 100000; ++n) { ap += 1.0; del *= x / ap; sum += del; if (fabs(del) < fabs(sum) * 1e-17) break; }
    return 1.0 - sum * m_exp(-x + a * m_log(x) - lg);
  }
  return m_exp(-x + a * m_log(x) - lg) * gamma_q_fraction(a, x);
}
EMAT_DN double gamma_q_inv(double a, double q) {
  if (q == 0.0) return k_inf;
  if (q == 1.0) return 0.0;
  const double lg = lgamma(a);
  if (q < 1e-3) {
    // Far upper tail (the reference's own test asks for Q down to 1e-300, safe_gamma_math_tests.cpp:83-95,247-262):
    // Newton on log Q(a, x) = log q, log Q = -x + a log x - lgamma(a) + log h taken from the continued fraction without
    // ever forming Q, and d/dx log Q = -density / Q = -1 / (x h); the steps close in on the root from one side.
    const double lq = m_log(q);
    double x = a + 1.0, h = gamma_q_fraction(a, x), prev = 0.0;
    if (-x + a * m_log(x) - lg + m_log(h) > lq) {   // the root lies where the fraction converges
      for (int j = 0; j < 100; ++j) {
        double dx = (-x + a * m_log(x) - lg + m_log(h) - lq) * x * h;
        double xn = x + dx;
        if (xn < a + 1.0) xn = a + 1.0;
        bool done = fabs(xn - x) <= 1e-14 * xn || (j > 2 && fabs(dx) >= fabs(prev));
        x = xn; prev = dx;
        if (done) break;
        h = gamma_q_fraction(a, x);
      }
      return x;
    }
  }
  const double p = 1.0 - q;
  double x;
  if (a > 1.0) {
    double pp = (p < 0.5) ? p : q;
    double t = sqrt(-2.0 * m_log(pp));
    double xg = (2.30753 + t * 0.27061) / (1.0 + t * (0.99229 + t * 0.04481)) - t;
    if (p < 0.5) xg = -xg;
    double v = a * pow(1.0 - 1.0 / (9.0 * a) - xg / (3.0 * sqrt(a)), 3.0);
    x = v > 1e-3 ? v : 1e-3;
  } else {
    double t = 1.0 - a * (0.253 + a * 0.12);
    if (p < t) x = pow(p / t, 1.0 / a);
    else x = 1.0 - m_log(1.0 - (p - t) / (1.0 - t));
  }
  const double a1 = a - 1.0;
  for (int j = 0; j < 60; ++j) {
    if (x <= 0.0) x = 1e-300;
    double err = gamma_q(a, x) - q;
    double tdens = m_exp(-x + a1 * m_log(x) - lg);
    if (tdens == 0.0) break;
    double u = -err / tdens;
    double w = u * (a1 / x - 1.0);
    double dx = u / (1.0 - 0.5 * (w < 1.0 ? w : 1.0));
    double xn = x - dx;
    if (xn <= 0.0) xn = 0.5 * x;
    double tol = 1e-15 * (xn > 1e-300 ? xn : 1e-300);
    if (fabs(xn - x) < tol) { x = xn; break; }
    x = xn;
  }
  return x;
}

}  // namespace EMAT_DEV_NS
}  // namespace emat
