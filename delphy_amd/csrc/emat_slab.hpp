// emat_slab.hpp -- layout of one partition part's working set ("slab") in HBM / LDS.
//
// Reference data model being flattened: Phylo_node{parent, children[2], t_min, t_max, t,
// vector<Mutation>, Missation_map} (core/phylo_tree.h:14-23), plus the Subrun's derived per-node
// arrays lambda_i_ and num_sites_missing_at_every_node_ (core/subrun.h:100-107) and the part's
// Very_scalable_coalescent_prior_part vectors (core/very_scalable_coalescent.h:47-56).
//
// A slab is ONE relocatable byte blob per part (all internal references are byte offsets), so that a
// workgroup can stream it into LDS with one coalesced copy, run its chain there, and stream it back:
//
//   [SlabHeader 256 B][NodeRec x n_nodes (64 B each)][cell table (2 arrays x cell_cap; the root part: 6 arrays)]
//   [trace ring][list heap: 16-B mutation / 8-B interval / 8-B from-state records][scratch]
//
// Plain C++ (no HIP types) because the host encoder and the kernels share it.
#ifndef EMAT_SLAB_HPP_
#define EMAT_SLAB_HPP_

#include <cstdint>

namespace emat {

constexpr uint32_t k_slab_magic = 0x454D4154u;  // "EMAT"

enum PartStatus : int32_t {
  k_part_ok = 0,
  k_part_need_space = 101,     // stopped BEFORE a move: heap reserve or scratch too small (state is consistent)
  k_part_overflow = 102,       // a container overflowed INSIDE a move (state is not trustworthy)
  k_part_cell_overflow = 103,  // root part needed more coalescent cells than its capacity
  k_part_internal = 104,       // an invariant that the reference CHECKs failed
  k_part_list_limit = 106,     // a per-node list would exceed what a ListRef can count (k_max_list_len): fatal, reported as EMAT_ERR_CAPACITY
                               // with the limit named -- never truncated
  k_part_need_cells = 105      // stopped BEFORE a move that would have grown the root part's grid past its capacity (state is consistent,
                               // the move undone and its RNG rewound: stop_for_cells); also reported for a 103 of a staged first leg, whose
                               // slab in HBM is still the state the launch found.  The host gives the part more cells and runs the rest.
};

enum SlabFlags : uint32_t { k_flag_includes_run_root = 1u };

// A list living in the slab heap.  `off` is a byte offset from the slab base.
struct ListRef {
  uint32_t off;
  uint16_t cnt;
  uint16_t cap;
};

// What the 16-bit counts allow.  Everything that writes a ListRef count goes through a check against these (set_list_cnt on the
// device, the encoders on the host and in k_gt_build); the boundary accepts lists up to k_max_list_upload entries per node, which
// leaves a list room to grow four-fold before the device stops the part with k_part_list_limit.
constexpr uint32_t k_max_list_len = 65535;      // ListRef::cnt / cap
constexpr uint32_t k_max_list_upload = 16000;   // per-node list length accepted by emat_part_upload / emat_tree_upload / a repartition
// Candidate-scan items pack {pusher's mutation index: 16 unsigned bits -- any index a list can have --, crossings: 2 bits, size of the
// site-delta set: 14 bits}; scans whose delta set is not far below 2^14 take the general (unpacked) algorithm instead.
constexpr int k_scan_max_deltas = 8000;
constexpr uint16_t list_cap_for(uint32_t padded_bytes, uint32_t elem_bytes) { return (uint16_t)(padded_bytes / elem_bytes > k_max_list_len ? k_max_list_len : padded_bytes / elem_bytes); }   // (constexpr: host and device)

struct MutRec {     // 16 B   (reference Mutation{from, site, to, t}, core/mutations.h:21-29)
  double t;
  int32_t site;
  uint8_t from;
  uint8_t to;
  uint16_t pad;
};
struct IvRec { int32_t start, end; };          // 8 B, half-open [start,end)  (core/interval_set.h:26)
struct FsRec { int32_t site; uint8_t state; uint8_t pad[3]; };   // 8 B, Missation_map::from_states entry

struct NodeRec {    // 64 B = one cache line per node
  int32_t parent;
  int32_t child0;
  int32_t child1;
  float t_min;
  float t_max;
  ListRef muts;       // MutRec[]
  ListRef miss;       // IvRec[]
  ListRef mfs;        // FsRec[]
  int32_t n_missing;  // num_sites_missing_at_every_node_[node]
  double t;
  double lambda;      // lambda_i_[node]
};
static_assert(sizeof(ListRef) == 8, "ListRef must be 8 bytes");
static_assert(sizeof(MutRec) == 16, "MutRec must be 16 bytes");
static_assert(sizeof(NodeRec) == 64, "NodeRec must be 64 bytes");

struct SlabHeader {
  uint32_t magic;
  uint32_t slab_bytes;         // total capacity of this slab
  int32_t n_nodes;
  int32_t root;
  uint32_t flags;
  int32_t status;              // PartStatus
  uint64_t rng_key;
  uint64_t rng_counter;
  double log_G;
  double log_aug_prior;
  // regions (byte offsets from slab base)
  uint32_t off_nodes;
  uint32_t off_cells;          // the part's own two double arrays [cell_cap] (k_bar_p, k_twiddle_bar_p); the root part: 5 double arrays then 1 int32 array
  uint32_t off_trace;
  uint32_t heap_begin, heap_top, heap_end;
  uint32_t scratch_begin, scratch_end;
  // coalescent window: stored cells are [cell_first, cell_first + n_cells)
  int32_t cell_first;
  int32_t n_cells;
  int32_t cell_cap;
  int32_t n_cells_total;       // length of the part's logical k_bar_p vector (= cell_first + n_cells)
  double t_ref;
  double t_step;
  // trace ring (tests)
  int32_t trace_cap;
  int32_t trace_len;
  // statistics
  int64_t moves_done;
  int64_t proposed[5];
  int64_t accepted[5];
  int64_t alg_bytes;
  int32_t fail_line;           // source line of the first failed device check (debugging aid)
  uint32_t alg_write16;        // the written part of alg_bytes, in units of 16 bytes (the header has no room for another 64-bit counter; heap compactions used to be counted here)
  int64_t device_ticks;        // wall_clock64() ticks (100 MHz) spent in the serial section of k_run_moves, cumulative
  uint64_t rng_spare;          // unconsumed second half of the last Philox block
  uint32_t rng_has_spare;
  uint32_t pad0;
#ifdef EMAT_PROFILE_PHASES
  int64_t phase_ticks[16];     // optional phase profile (builds with -DEMAT_PROFILE_PHASES), s_memtime ticks
  uint8_t reserved[128];
};
static_assert(sizeof(SlabHeader) == 512, "SlabHeader must be 512 bytes in profiling builds");
#else
};
// Every byte of the header is staged in LDS with the part, and LDS is what limits how many parts a CU holds.
static_assert(sizeof(SlabHeader) == 256, "SlabHeader must be 256 bytes");
#endif

// Shared, read-only model data in HBM (one copy per device).
struct EvoTable {               // reference Global_evo_model (core/evo_model.h:20-48)
  int32_t num_sites;
  int32_t num_partitions;
  int32_t uniform_sites;              // one site partition and nu_l == 1 at every site (the reference's default: run.h:256): the moves then read neither per-site array
  int32_t pad_;
  const uint8_t* ref_sequence;        // [L]
  const uint8_t* partition_for_site;  // [L]
  const double* nu_l;                 // [L]
  const double* cum_Q_l;              // [L+1]   ref_cum_Q_l_ (core/phylo_tree_calc.cpp:379-388)
  const double* mu;                   // [P]
  const double* pi;                   // [P][4]
  const double* q;                    // [P][4][4]
};
// Coalescent cell table of a part.  Of the six vectors of the reference's Very_scalable_coalescent_prior_part
// (very_scalable_coalescent.h:47-56) only two are the part's own -- k_bar_p, which its moves update, and k_twiddle_bar_p; the
// others (k_twiddle_bar, popsize_bar, num_active_parts, and t_step / popsize_bar kept beside them) are the same for every
// part of the run -- the reference shares them through pointers too -- and live ONCE per device (SharedCells), indexed by
// absolute cell and read through the scalar cache: a slab carries 16 bytes per cell instead of 44, which is what lets the
// parts with the longest time spans (60-120 cells: the slowest chains of a pass) keep an LDS arena for their topology moves.
// The part that holds the run's root is the exception: it may APPEND cells while it runs (ensure_space, cpp:259-299), with
// values only it knows, so it keeps all six arrays in its own slab, `cell_cap` long each.
constexpr uint32_t k_cell_bytes_own = 2 * 8;
constexpr uint32_t k_cell_bytes_root = 5 * 8 + 4;
inline uint32_t cell_bytes_for(bool includes_run_root) { return includes_run_root ? k_cell_bytes_root : k_cell_bytes_own; }
struct SharedCells {            // one per device; [num_cells] each, absolute cell index
  const double* k_twiddle_bar;
  const double* ts_over_pop;    // t_step / popsize_bar, divided once when the grid is built
  const int32_t* num_active_parts;
  int32_t num_cells;
};
constexpr int k_max_lds_partitions = 2;   // HKY tables of up to this many site partitions are staged in LDS

struct PopTable {               // reference Pop_model family (core/pop_model.h)
  int32_t kind;                 // emat_pop_model_kind
  int32_t skygrid_type;
  int32_t skygrid_num_knots;
  int32_t pad;
  double p[4];
  double t_c;                   // Exp_pop_model::t_c_
  double skygrid_inv_dx;        // (knots - 1) / (x_last - x_first): first guess of the knot interval of a time (0 if degenerate)
  const double* skygrid_x;
  const double* skygrid_gamma;
};

struct RunFlags {
  double t_max_tip;
  int32_t only_displacing_inner_nodes;
  int32_t topology_moves_enabled;
};

inline uint32_t align_up(uint32_t x, uint32_t a) { return (x + a - 1) / a * a; }

}  // namespace emat
#endif  // EMAT_SLAB_HPP_
