/*
 * emat_backend.h -- C-ABI of the MI355X-native EMAT local-move engine.
 *
 * This is the drop-in boundary for Delphy's per-iteration hot path.  The reference has no
 * FFI for this path; the seam is the C++ class `Subrun` as driven by `Run`
 * (reference core/subrun.h:16-135, core/run.cpp:110-293,610-693).  Every entry point below
 * names the reference interaction it replaces.  All functions return an `emat_status`
 * (0 = OK); nothing throws across the boundary; all pointers are plain host pointers and
 * the backend copies what it needs (the caller keeps ownership of its buffers).
 *
 * Conventions (reference core/tree.h:33-39, core/mutations.h:21-29, core/interval_set.h):
 *   - nodes are int32 indices local to the subtree, EMAT_NO_NODE = -1;
 *   - states are 0..3 = A,C,G,T (reference core/sequence.h `Real_seq_letter`);
 *   - mutations on a branch are sorted by (t, site); the subtree root's "mutations" are
 *     the deltas ref_sequence -> subroot sequence with t = -DBL_MAX;
 *   - missation intervals are sorted, disjoint, non-adjacent half-open [start,end).
 */
#ifndef EMAT_BACKEND_H_
#define EMAT_BACKEND_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMAT_NO_NODE (-1)

typedef enum emat_status {
  EMAT_OK = 0,
  EMAT_ERR_INVALID_ARGUMENT = 1,
  EMAT_ERR_NO_DEVICE = 2,        /* HIP device / extension missing: the product path never falls back to CPU */
  EMAT_ERR_HIP = 3,
  EMAT_ERR_STATE = 4,            /* call sequence violated (e.g. run before upload) */
  EMAT_ERR_CAPACITY = 5,         /* a part ran out of slab space; see emat_part_get_status */
  EMAT_ERR_INTERNAL = 6,
  EMAT_ERR_BUFFER_TOO_SMALL = 7,
  EMAT_ERR_IO = 8                /* a file could not be opened, written in full, or closed (include/emat_dphy.h) */
} emat_status;

/* Flat (struct-of-arrays, CSR) image of one `Phylo_tree` (reference core/phylo_tree.h:14-64).
 * Used in both directions.  For downloads the caller provides the arrays and their capacities. */
typedef struct emat_flat_tree {
  int32_t num_nodes;
  int32_t root;
  int32_t* parent;       /* [num_nodes] */
  int32_t* child0;       /* [num_nodes], EMAT_NO_NODE for tips */
  int32_t* child1;       /* [num_nodes] */
  double*  t;            /* [num_nodes] */
  float*   t_min;        /* [num_nodes] tips: date bounds; inner nodes -FLT_MAX */
  float*   t_max;        /* [num_nodes] tips: date bounds; inner nodes +FLT_MAX */
  /* mutations, CSR by node */
  int32_t* mut_offset;   /* [num_nodes+1] */
  int32_t* mut_site;     /* [num_muts] */
  uint8_t* mut_from;     /* [num_muts] */
  uint8_t* mut_to;       /* [num_muts] */
  double*  mut_t;        /* [num_muts] */
  /* missation intervals, CSR by node */
  int32_t* miss_offset;  /* [num_nodes+1] */
  int32_t* miss_start;   /* [num_intervals] */
  int32_t* miss_end;     /* [num_intervals] */
  /* missation from_states (only sites whose state != ref_sequence), CSR by node */
  int32_t* mfs_offset;   /* [num_nodes+1] */
  int32_t* mfs_site;     /* [num_from_states] */
  uint8_t* mfs_state;    /* [num_from_states] */
  /* capacities of the variable-length arrays (used by downloads only) */
  int32_t cap_muts;
  int32_t cap_intervals;
  int32_t cap_from_states;
} emat_flat_tree;

/* Population model descriptor (reference core/pop_model.h:12-241). */
typedef enum emat_pop_model_kind {
  EMAT_POP_CONST = 0,            /* Const_pop_model{pop}                                  p[0]=pop */
  EMAT_POP_EXP = 1,              /* Exp_pop_model{t0, pop_at_t0, growth_rate, min_pop}    p[0..3]  */
  EMAT_POP_SKYGRID = 2           /* Skygrid_pop_model{x[], gamma[], type}                          */
} emat_pop_model_kind;

typedef struct emat_pop_model {
  int32_t kind;
  double  p[4];
  int32_t skygrid_type;          /* 1 = staircase, 2 = log-linear (reference pop_model.h:149-185) */
  int32_t skygrid_num_knots;     /* M+1 */
  const double* skygrid_x;       /* [num_knots] strictly increasing */
  const double* skygrid_gamma;   /* [num_knots] log N at knots */
} emat_pop_model;

typedef struct emat_config {
  int32_t device;                /* HIP device ordinal; -1 = host-only handle (uploads and coalescent staging work,
                                    every launch returns EMAT_ERR_NO_DEVICE: there is no CPU fallback) */
  int32_t num_sites;             /* L */
  int32_t max_parts;             /* upper bound on parts resident at once (0 = grow on demand) */
  double  slab_slack;            /* >=1: per-part working-set capacity as a multiple of its content (0 = default 3.0) */
  int32_t trace_moves;           /* >0: record the first N moves of every part in a trace ring (tests) */
  int32_t use_lds;               /* 1 = stage part working sets in LDS when they fit (default), 0 = always HBM */
} emat_config;

typedef struct emat_backend emat_backend;

/* ---- life cycle ------------------------------------------------------------------------- */
/* replaces: Subrun construction/destruction as a set (reference run.cpp:131,182-183) */
emat_status emat_backend_create(const emat_config* cfg, emat_backend** out);
emat_status emat_backend_destroy(emat_backend* h);
/* Tuning and test options of one handle, by name, set after emat_backend_create and before the first launch (defaults are what
 * bench.py measures; an unknown name is refused).  The library reads NO tuning from the environment (only EMAT_VERBOSE, which
 * makes it narrate on stderr -- or, with the value "spans", account for the host's time per named stretch of a cycle and print the
 * table when a handle is destroyed): an embedding process decides per handle.  The Python mirror forwards EMAT_<NAME> variables for
 * A/B scripts.
 *   "lds_classes"   percentile of part sizes the staging area must hold whole, default "60"; a comma list makes size classes
 *   "lds_max"       largest staging area in bytes (default 98304);  "lds_scratch"  extra LDS scratch arena per part (default 0)
 *   "giants"        0 disables the side launches;  "side_arena"  bytes of arena a part must be left with in the main area
 *   "chunks"        tickets per part and pass in the main class (default 4);  "ticket_taper"  0 = equal tickets
 *   "ticket_weights" "w1,w2,...";  "single_ticket_parts"  how many of the largest parts run a pass unsplit;  "parts_per_cu"
 *   "slack", "heap_per_node"   list-heap capacity = content x slack + bytes per node (3.0 / 64)
 *   "order_by_time" 1 = re-sort the launch order by measured chain times at every synchronisation
 *   "tree_host_coalescent"  1 = emat_tree_repartition builds the coalescent tables with the host's code (bit-identical to the host cycle)
 *   testing aids: "ticket_xcd_spread" (a part's tickets on different XCDs), "ticket_release" ("full": plain agent-scope releases),
 *   "tree_tight" (no spare room in the device tree), "build_blocks" (workgroups of the initial-tree builder),
 *   "no_uniform_sites" (1: the moves read the per-site partition and rate arrays even when the model is the reference's default of one
 *   site partition and nu_l == 1 everywhere, where the answers are known without a load: the A/B of that short cut, round 6),
 *   "debug_fail_gather" (1: the next deferred gather of the device-resident tree reports an inconsistency; may be set at any time);
 *   profiling builds: "fn_min_lists", "phase_extra". */
emat_status emat_set_option(emat_backend* h, const char* key, const char* value);
/* Size of the library's host thread pool (per process, before its first parallel loop; 0 = default: min(cores, 16)). */
emat_status emat_set_host_threads(int32_t n);
const char* emat_last_error(const emat_backend* h);   /* human-readable text for the last failure */
/* First 16 hex digits of the SHA-256 of the sources the kernels of THIS library were compiled from (csrc/Makefile: DEVSRC, in
 * that order); "unstamped" for a build that bypassed the Makefile.  bench.py refuses a library whose id differs from the
 * sources beside it, and quotes a committed PMC profile only for the id the profile was measured on. */
const char* emat_build_id(void);

/* ---- shared, read-only inputs ----------------------------------------------------------- */
/* replaces: `subtree.ref_sequence = ref_seq` (reference run.cpp:139-140) */
emat_status emat_set_ref_sequence(emat_backend* h, const uint8_t* ref_sequence, int32_t num_sites);

/* replaces: Subrun::set_evo (reference subrun.h:29-30; evo_model.h:20-48).  q is row-major
 * q[beta][a][b] with q[a][a] = -sum_{b!=a} q[a][b].  Invalidates derived quantities. */
emat_status emat_set_evo(emat_backend* h, int32_t num_partitions, const double* mu /*[P]*/,
                         const double* pi /*[P][4]*/, const double* q /*[P][4][4]*/,
                         const double* nu_l /*[L]*/, const int32_t* partition_for_site /*[L]*/);

/* replaces: set_t_max_tip / set_only_displacing_inner_nodes / set_topology_moves_enabled
 * (reference subrun.h:24-40; pushed in run.cpp:267-275) */
emat_status emat_set_flags(emat_backend* h, double t_max_tip, int32_t only_displacing_inner_nodes,
                           int32_t topology_moves_enabled);

/* ---- parts ------------------------------------------------------------------------------ */
/* replaces: the loop that builds one Subrun per partition part (reference run.cpp:131-184).
 * Discards all resident parts, then expects exactly `num_parts` emat_part_upload calls
 * followed by emat_end_upload. */
emat_status emat_begin_upload(emat_backend* h, int32_t num_parts);
/* replaces: Subrun(bitgen, tree, includes_run_root, evo) (reference subrun.cpp:10-15).  `seed`
 * keys the part's counter-based RNG stream (the reference seeds one std::mt19937 per part,
 * run.cpp:112-114). */
/* Thread-safety: between emat_begin_upload and emat_end_upload, emat_part_upload may be called concurrently for
 * DISTINCT part ids; every other entry point expects one caller at a time per handle. */
emat_status emat_part_upload(emat_backend* h, int32_t part_id, const emat_flat_tree* subtree,
                             int32_t includes_run_root, uint64_t seed);
emat_status emat_end_upload(emat_backend* h);

/* replaces: Run::reset_very_scalable_coalescent_parts -> make_very_scalable_coalescent_prior_parts
 * + Subrun::set_coalescent_prior_part (reference run.cpp:277-293; very_scalable_coalescent.cpp:85-232).
 * Builds every part's k_bar_p / k_twiddle_bar_p and the shared k_twiddle_bar, popsize_bar,
 * num_active_parts from the resident subtrees.  Invalidates derived quantities. */
emat_status emat_build_coalescent_parts(emat_backend* h, const emat_pop_model* pop_model,
                                        int32_t root_part_index, double t_step);

/* Staged form of emat_build_coalescent_parts for a run whose parts are sharded over several GPUs (one
 * process per GPU).  Between the stages the caller all-reduces tiny vectors over RCCL:
 *   begin       -> this rank's [t_min, t_max]                       -> all-reduce MIN / MAX
 *   set_range   -> number of grid cells
 *   local_grid  -> this rank's k_bar / num_active_parts             -> all-reduce SUM   (very_scalable_coalescent.cpp:153-188)
 *   sample      -> this rank's share of k_twiddle_bar (Gaussian draws from each part's stream, :198-219) -> all-reduce SUM
 *   finish      -> every resident part gets its Very_scalable_coalescent_prior_part
 * `root_part_index` is -1 on ranks that do not hold the run's root part. */
emat_status emat_coalescent_begin(emat_backend* h, const emat_pop_model* pop_model, int32_t root_part_index, double t_step,
                                  double* local_t_min, double* local_t_max);
emat_status emat_coalescent_set_range(emat_backend* h, double all_t_min, double all_t_max, int32_t* num_cells);
emat_status emat_coalescent_local_grid(emat_backend* h, double* k_bar /*[num_cells]*/, int32_t* num_active_parts /*[num_cells]*/);
emat_status emat_coalescent_sample(emat_backend* h, const double* k_bar, const int32_t* num_active_parts, double* k_twiddle_bar_local /*[num_cells]*/);
emat_status emat_coalescent_finish(emat_backend* h, const double* k_twiddle_bar /*[num_cells]*/);

/* ---- the hot path ----------------------------------------------------------------------- */
/* replaces: Run::run_local_moves(count) (reference run.cpp:682-693): `count / num_parts` calls of
 * Subrun::mcmc_sub_iteration() on every part, the remainder going to part 0.  Asynchronous with
 * respect to the host; emat_synchronize (or any getter) waits. */
emat_status emat_run_local_moves(emat_backend* h, int64_t count);
/* Same, with an explicit number of moves per part (all parts the same). */
emat_status emat_run_moves_per_part(emat_backend* h, int64_t moves_per_part);
/* Same, plus `extra_moves_part0` more on this handle's part 0: the split of Run::run_local_moves when the run's parts are
 * spread over several handles (the handle holding the run's part 0 gets the remainder). */
emat_status emat_run_moves_split(emat_backend* h, int64_t moves_per_part, int64_t extra_moves_part0);
/* NOT the reference's rule: `moves_per_part` moves on every part and one more on the parts [0, one_more_below).  The
 * reference hands the whole remainder of count / parts to subrun 0 -- at most parts - 1 moves, nothing next to a subrun's
 * share at 8 threads, but six times a part's share at 8 000 parts, where that one chain then sets the length of the pass
 * (measured at C4: 62 ms instead of 34).  Spreading the remainder one move per part runs the same total. */
emat_status emat_run_moves_even(emat_backend* h, int64_t moves_per_part, int32_t one_more_below);
/* Waits for the launches issued so far and checks that every part ran its chain to completion.  The reference's containers
 * grow without bound; a slab has fixed room.  A part that runs out of it -- list heap, scratch, or the cells the root part's
 * coalescent grid grows into the past -- is re-materialised with more room (twice; four times the cells) and the rest of
 * its moves run, transparently, up to four times (then EMAT_ERR_CAPACITY): it either stopped BEFORE a move (the heap
 * reserve is checked there, and the moves that can grow the grid ask before they change anything), or a container
 * overflowed INSIDE a move and the state its leg started from was put back (the untouched HBM copy of a staged part, a
 * shadow copy for a part run on its HBM slab).  Either way the chain continues exactly as it would have with unbounded
 * containers.  A part that broke an invariant the reference CHECKs makes this -- and every getter, all of which
 * synchronise first -- fail with EMAT_ERR_INTERNAL naming the part and the device source line; the reference aborts in that
 * situation. */
emat_status emat_synchronize(emat_backend* h);

/* ---- The whole tree resident in HBM -------------------------------------------------------------
 * Run::repartition (core/run.cpp:110-193) copies every part out of the whole Phylo_tree into its Subrun's tree and
 * Run::reassemble (core/run.cpp:195-256) copies the parts back; through emat_part_upload / emat_part_download that is a
 * host round trip of every node, mutation and missation per cycle.  With the tree uploaded ONCE, a cycle needs from the
 * host only the partition itself -- which nodes form which part -- and moves topology + node times (a few MB) back:
 *
 *   emat_tree_upload            the whole tree -> HBM (node arrays + three record heaps); the root node must carry no
 *                               mutations (Run::normalize_root, run.cpp:258-265)
 *   emat_tree_get_topology      parent / children / node times / root, as of the last upload or reassemble: what
 *                               generate_random_partition_stencil and partition_tree (tree_partitioning.h:88-239) read
 *   emat_tree_repartition       run.cpp:110-193 + reset_very_scalable_coalescent_parts (run.cpp:277-293): the parts as
 *                               CSR over part-local nodes -- `orig` maps a part's node to the tree's node, local node 0
 *                               being the part's cut point, `kid0` / `kid1` are part-local children (EMAT_NO_NODE at the
 *                               part's tips).  One kernel computes the sequence state at every cut point
 *                               (phylo_tree_calc.cpp:19-56) and sizes the parts, a second writes the slabs -- byte for
 *                               byte what emat_part_upload + emat_build_coalescent_parts + the first launch would have
 *                               produced for the same parts and seeds.  The coalescent cell tables are built on the host
 *                               from topology + times.
 *   emat_tree_reassemble        run.cpp:195-256 + normalize_root: every part writes the nodes it owns back, the record
 *                               heaps are rebuilt, the changes of the root sequence are folded into the reference
 *                               sequence on the device (and reported, so that the caller's copy can follow).  It returns
 *                               when links, root and root changes are on the host (emat_tree_get_kids); the gather of the
 *                               lists may still be running, and the next emat_tree_* call that needs them waits for it
 *                               (that is also where a failure of the gather is reported)
 *   emat_tree_download          the whole tree (and its reference sequence) back as a flat tree
 *
 * Between emat_tree_repartition and emat_tree_reassemble the backend holds ordinary parts: every run / getter /
 * reduction entry point above works on them unchanged. */
emat_status emat_tree_upload(emat_backend* h, const emat_flat_tree* tree);
emat_status emat_tree_get_sizes(emat_backend* h, int32_t* num_nodes, int32_t* num_muts, int32_t* num_intervals, int32_t* num_from_states);
emat_status emat_tree_download(emat_backend* h, emat_flat_tree* out, uint8_t* ref_sequence /* [num_sites] or NULL */);
emat_status emat_tree_get_topology(emat_backend* h, int32_t* parent, int32_t* child0, int32_t* child1, double* t, int32_t* root);
/* What generate_random_partition_stencil and partition_tree (tree_partitioning.h:88-239) read of the tree and nothing more: every
 * node's two children side by side ((*kids)[2 v], (*kids)[2 v + 1]; EMAT_NO_NODE at the tips), the root and the root's time.
 * `*kids` points into the backend's own page-locked mirror, which every reassemble refreshes (1.6 MB at 200 000 nodes instead of
 * the 4 MB of emat_tree_get_topology, and no copy): valid until the next emat_tree_reassemble(_end) or emat_tree_upload. */
emat_status emat_tree_get_kids(emat_backend* h, const int32_t** kids, int32_t* num_nodes, int32_t* root, double* t_root);
/* partition_tree (tree_partitioning.h:88-135, 196-239) on the device, for callers that only have the cut nodes of a stencil:
 * one thread per part walks the part down from its cut point and numbers its nodes exactly as the reference's work list does.
 * Part i has cut node cut_nodes[i]; unless the stencil names the run's root, the root part comes last.  The arrays stay on
 * the device: emat_tree_repartition(_range) cuts THIS partition when it is given NULL for part_offset / orig / kid0 / kid1;
 * emat_tree_get_partition downloads them (any pointer may be NULL). */
emat_status emat_tree_partition(emat_backend* h, int32_t num_cuts, const int32_t* cut_nodes, int32_t* num_parts, int32_t* root_part, int32_t* part_sizes /* [num_cuts + 1] or NULL */);
emat_status emat_tree_get_partition(emat_backend* h, int32_t* part_offset, int32_t* orig, int32_t* kid0, int32_t* kid1);
emat_status emat_tree_repartition(emat_backend* h, int32_t num_parts, const int32_t* part_offset /* [num_parts + 1] */, const int32_t* orig,
                                  const int32_t* kid0, const int32_t* kid1, int32_t root_part, const uint64_t* seeds /* [num_parts] */,
                                  const emat_pop_model* pop_model, double t_step);
/* `site` / `from` / `to` [capacity] receive the changes of the reference sequence (old state, new state); any of the
 * output arguments may be NULL / 0 when the caller reads the sequence through emat_tree_download instead. */
emat_status emat_tree_reassemble(emat_backend* h, int32_t* num_root_deltas, int32_t* site, uint8_t* from, uint8_t* to, int32_t capacity);
/* One run over several processes, one GPU each, EVERY one with the whole tree in its HBM (the tree is a few tens of MB; what
 * is worth sharding is the moves).  Every process cuts the same partition and calls emat_tree_repartition_range with its own
 * block [part_lo, part_hi) of the parts (backend part id = part - part_lo): the sequence states at the cut points and the
 * coalescent grid are computed for the whole run by every process -- identically, no exchange -- and only the slabs of the
 * block are built.  After the moves:
 *   emat_tree_get_root_deltas     on the process that holds the root part (*num_root_deltas = -1 elsewhere); the caller
 *                                 hands the result to every process
 *   emat_tree_gather_local        every process: its own parts back into its copy of the tree (heaps rebuilt from zero)
 *   emat_tree_export_nodes        every process: what its parts own and link, as one buffer (buf = NULL: size only)
 *   emat_tree_apply_nodes         every process, once per OTHER process's buffer (an all-gather, the caller's)
 *   emat_tree_reassemble_end      every process: mirrors refreshed; every copy of the tree holds the same nodes again
 * (lists sit at different heap offsets in different processes, which nothing depends on).  `buf` of emat_tree_export_nodes /
 * emat_tree_apply_nodes may be host memory or device memory: with device buffers the exchange is one RCCL all-gather on
 * what the kernels wrote, and nothing but a 32-byte header and the per-node records (for validation) crosses PCIe.
 * Streams: the engine launches on a stream of its own.  emat_tree_export_nodes returns with its kernels finished (the buffer
 * may be handed to a collective on any stream); emat_tree_apply_nodes reads `buf` from the engine's stream as soon as it is
 * called, so the caller must have waited for whatever produced a DEVICE buffer (e.g. the all-gather on its own stream). */
emat_status emat_tree_repartition_range(emat_backend* h, int32_t num_parts, const int32_t* part_offset, const int32_t* orig, const int32_t* kid0, const int32_t* kid1,
                                        int32_t root_part, const uint64_t* seeds, const emat_pop_model* pop_model, double t_step, int32_t part_lo, int32_t part_hi);
emat_status emat_tree_get_root_deltas(emat_backend* h, int32_t* num_root_deltas, int32_t* site, uint8_t* from, uint8_t* to, int32_t capacity);
emat_status emat_tree_gather_local(emat_backend* h, int32_t num_root_deltas, const int32_t* site, const uint8_t* from, const uint8_t* to);
emat_status emat_tree_export_nodes(emat_backend* h, uint8_t* buf, uint64_t capacity, uint64_t* bytes_needed);
emat_status emat_tree_apply_nodes(emat_backend* h, const uint8_t* buf, uint64_t bytes);
emat_status emat_tree_reassemble_end(emat_backend* h);

/* Forces the from-scratch recomputation that Subrun::validate_derived_quantities() performs
 * after set_evo / set_coalescent_prior_part (reference subrun.cpp:17-26).  Called implicitly
 * by the run functions when the derived quantities are stale. */
emat_status emat_recalc_derived(emat_backend* h);

/* replaces: Subrun::check_derived_quantities (reference subrun.cpp:28-56), which debug builds run after every move and
 * --v0-paranoid forces in release builds (run.h:220-224, cmdline.cpp:177).  Every part's lambda_i, num_sites_missing, log_G
 * and augmented coalescent prior are recomputed from scratch ON THE DEVICE into scratch memory and compared with what the
 * moves maintained incrementally; the state is not touched, nothing but this library is involved.  `tol_scale` multiplies
 * the reference's own tolerances (|d lambda_i| / L < 1e-8, |d log_G| < 1e-6, |d prior| < 1e-5; missing-site counts exact).
 * Returns EMAT_OK, or EMAT_ERR_INTERNAL with emat_last_error naming the first offending part and quantity.
 * `*worst_part` / `worst4` (either may be NULL): that part -- or, when all is well, the part closest to a tolerance -- and
 * its four deviations {lambda per site, log_G, prior, nodes with a wrong missing-site count}. */
emat_status emat_check_derived(emat_backend* h, double tol_scale, int32_t* worst_part, double* worst4);

/* ---- results ---------------------------------------------------------------------------- */
/* replaces: sum of subrun.log_G() / subrun.log_augmented_coalescent_prior()
 * (reference run.cpp:340-348) */
emat_status emat_get_totals(emat_backend* h, double* log_G, double* log_augmented_coalescent_prior);

/* ---- sufficient statistics of the global moves ---------------------------- */
/* replaces: Run::calc_cur_Ttwiddle_beta_a / calc_cur_num_muts / calc_cur_num_muts_ab (reference run.cpp:445-453), i.e.
 * calc_Ttwiddle_beta_a (phylo_tree_calc.cpp:288-369), calc_num_muts_beta_ab (:599-610), calc_num_muts (:577-585), which
 * the host-side global moves (mu, HKY kappa / pi; run.cpp:781-1010) consume.  Computed on the device over the parts of
 * this handle and summed in part order, so that the trees need not be downloaded for them: every branch of the whole
 * tree is a non-root branch of exactly one part.  With parts spread over several handles (GPUs) the caller adds the
 * per-handle results.  Ttwiddle_beta_a[beta][a] = sum over sites l of partition beta of nu_l x (time site l spends in
 * state a over all branches, missing stretches excluded); num_muts_beta_ab[beta][a][b] counts mutations a -> b.
 * num_partitions must equal the value given to emat_set_evo and be <= 4 (EMAT_ERR_CAPACITY otherwise). */
emat_status emat_get_global_stats(emat_backend* h, int32_t num_partitions, double* Ttwiddle_beta_a /*[P][4]*/,
                                  int64_t* num_muts_beta_ab /*[P][4][4]*/, int64_t* num_muts /* may be NULL */);

/* replaces: calc_num_muts_l (reference phylo_tree_calc.cpp:612-622), consumed with calc_Ttwiddle_l by the site-rate
 * (alpha / nu_l) moves (run.cpp:1109-1110, 1184-1185): mutations per site over all branches of all parts of this handle
 * (the deltas above a part's root are not mutations).  With parts on several handles the caller adds the vectors. */
emat_status emat_get_num_muts_l(emat_backend* h, int32_t* num_muts_l /*[num_sites]*/);

/* replaces: calc_Ttwiddle_l (reference phylo_tree_calc.cpp:176-222; with calc_num_muts_l the input of the site-rate moves,
 * run.cpp:1109, 1184): Ttwiddle^(l) = sum_a q^(l)_a T^(l)_a, the escape-rate-weighted time site l spends in each state.
 * The reference corrects q_ref T_total per mutation / missation by the branch length BELOW that point, which crosses
 * part boundaries, so the computation is staged (the run driver, who knows the tree of parts, wraps it:
 * emat_run_get_Ttwiddle_l):
 *   emat_get_part_tree_lengths   sum of the branch lengths inside each part;
 *   emat_Ttwiddle_l_partial      given, for every part, the tips that are cut nodes of parts below and the whole-tree
 *                                branch length hanging at each (CSR: ext_offset[num_parts + 1], ext_node, ext_length), the
 *                                per-site sums S and R over this handle's parts and, on the handle holding the run's root,
 *                                the total tree length;
 *   emat_Ttwiddle_l_finish       Ttwiddle_l = q_ref (T_total - R) + S from the sums over ALL handles (all-reduce SUM). */
emat_status emat_get_part_tree_lengths(emat_backend* h, double* tree_length_of_part /*[num_parts]*/);
emat_status emat_Ttwiddle_l_partial(emat_backend* h, const int32_t* ext_offset, const int32_t* ext_node, const double* ext_length,
                                    double* S /*[num_sites]*/, double* R /*[num_sites]*/, double* tree_length_below_root /* may be NULL */);
emat_status emat_Ttwiddle_l_finish(emat_backend* h, const double* S_sum, const double* R_sum, double tree_length, double* Ttwiddle_l /*[num_sites]*/);

/* replaces: Run::calc_cur_log_coalescent_prior (reference run.cpp:455-465), i.e. Scalable_coalescent_prior::calc_log_prior
 * (scalable_coalescent.cpp:163-187) with every node displaced to its current time (:88-138): the whole-tree grid prior
 *   - sum_cells t_step kbar (kbar - 1) / (2 Nbar)  -  sum over inner nodes of log N(t),
 * under the population model last given to emat_build_coalescent_parts / emat_coalescent_begin.  `t_ref` is the time of
 * the latest tip (calc_max_tip_time, run.cpp:40) and cell j is [t_ref + j t_step, t_ref + (j + 1) t_step), j < 0.
 * The grid is additive over nodes and therefore over parts (a cut node counts once, as the coalescence it is), the prior
 * is not linear in it: with parts on several GPUs each rank asks for its partial grid over a common cell range
 * (`first_cell_needed`, all-reduce MIN, tells how far back that must reach; call with num_cells = 0 to get it), the
 * partial grids and log sums are all-reduced (SUM) and any rank evaluates the formula.  A single handle uses the
 * one-call form. */
emat_status emat_scalable_coalescent_partial(emat_backend* h, double t_ref, double t_step, int32_t first_cell, int32_t num_cells,
                                             double* k_bar_partial /*[num_cells], overwritten*/, double* sum_neg_log_pop /* may be NULL */,
                                             int32_t* first_cell_needed /* may be NULL */);
emat_status emat_scalable_coalescent_log_prior(emat_backend* h, double t_ref, double t_step, int32_t first_cell, int32_t num_cells,
                                               const double* k_bar_partial_sum /*[num_cells]*/, double sum_neg_log_pop, double* log_prior);
emat_status emat_get_scalable_coalescent_log_prior(emat_backend* h, double t_ref, double t_step, double* log_prior);

/* Sizes needed to download a part (so that the caller can size an emat_flat_tree). */
emat_status emat_part_get_sizes(emat_backend* h, int32_t part_id, int32_t* num_nodes,
                                int32_t* num_muts, int32_t* num_intervals, int32_t* num_from_states);
/* replaces: reading subrun.tree() in Run::reassemble (reference run.cpp:200-250) */
emat_status emat_part_download(emat_backend* h, int32_t part_id, emat_flat_tree* out);

/* replaces: subrun.lambda_i(), num_sites_missing_at_every_node(), log_G(),
 * log_augmented_coalescent_prior() (reference subrun.h:43-55).  Any pointer may be NULL. */
emat_status emat_part_get_derived(emat_backend* h, int32_t part_id, double* lambda_i /*[num_nodes]*/,
                                  int32_t* num_sites_missing /*[num_nodes]*/, double* log_G,
                                  double* log_augmented_coalescent_prior);

/* replaces: subrun.state_frequencies_of_ref_sequence_per_partition() (reference subrun.h:45-46; Run reads the root part's in
 * check_global_and_local_totals_match, run.cpp:354-355, and the HKY moves use the run's own copy, run.cpp:477): how many sites of
 * every site partition carry each state in the reference sequence the part is written against
 * (calc_state_frequencies_per_partition_of, phylo_tree_calc.cpp:95-106) -- the table the kernels evaluate the root prior from
 * (calc_log_root_prior, phylo_tree_calc.cpp:467-504), refreshed whenever the reference sequence follows the root sequence.
 * `*num_partitions` is in/out (capacity in rows of four, count out); counts[beta * 4 + a]. */
emat_status emat_part_get_state_frequencies(emat_backend* h, int32_t part_id, int32_t* num_partitions, int32_t* counts /*[num_partitions][4]*/);

/* Coalescent-part arrays of one part (reference very_scalable_coalescent.h:47-56), for the
 * cross-part exchange and for tests.  `*num_cells` is in/out (capacity in, length out). */
emat_status emat_part_get_coalescent(emat_backend* h, int32_t part_id, int32_t* num_cells,
                                     double* k_bar_p, double* k_twiddle_bar_p, double* k_twiddle_bar,
                                     double* popsize_bar, int32_t* num_active_parts,
                                     double* t_ref, double* t_step);

/* Where a part's random stream stands (reference: the Subrun's own bit generator, run.cpp:112-114, 182-183): Philox4x32-10 keyed
 * by `key`; `counter` blocks consumed; the unconsumed second 64-bit half of the last block if `has_spare`.  With it, the part's
 * tree and its coalescent arrays, a checker can replay the part's chain from exactly the state the device starts from --
 * also when the parts were cut and their tables built by kernels (emat_tree_repartition).  Any pointer may be NULL. */
emat_status emat_part_get_rng(emat_backend* h, int32_t part_id, uint64_t* key, uint64_t* counter, uint64_t* spare, int32_t* has_spare);

/* Per-part status and move counters. */
typedef struct emat_part_stats {
  int32_t status;                /* 0 = OK; otherwise an emat_status raised inside the kernel */
  int32_t num_nodes;
  int64_t moves_done;
  int64_t proposed[5];           /* inner_displace, tip_displace, branch_reform, subtree_slide, spr1 */
  int64_t accepted[5];
  int64_t algorithmic_bytes;     /* bytes the moves touched, counted with SURVEY section 8(d)'s per-record sizes */
  int64_t rng_draws;
  int64_t device_ticks;          /* 100 MHz wall-clock ticks this part's wavefront spent running moves (cumulative) */
  int64_t algorithmic_write_bytes; /* the part of algorithmic_bytes that is written: coalescent cells, re-timed mutation lists, region records, re-linked nodes */
} emat_part_stats;
emat_status emat_part_get_stats(emat_backend* h, int32_t part_id, emat_part_stats* out);

/* Trace of the first `cfg.trace_moves` moves of a part (tests): 4 doubles per move =
 * {move kind (0..4, -1 = no-op), node X, accepted (0/1), log_mh (NaN when the move exited early)}. */
emat_status emat_part_get_trace(emat_backend* h, int32_t part_id, int32_t* num_moves /*in/out*/,
                                double* trace /*[4*num_moves]*/);

/* Duration of the last emat_run_* launch measured with HIP events on the engine's own stream (milliseconds): the launch of the
 * main size class.  The few parts of the side classes run beside it on streams of their own and are waited for by whatever next
 * reads or changes the parts (emat_synchronize, emat_tree_reassemble, the getters), not by the next emat_run_* call. */
emat_status emat_last_run_ms(emat_backend* h, double* ms);
/* Duration of the dominant kernel (k_run_moves, the bulk size class) inside the last pass, and how many parts it ran. */
emat_status emat_last_kernel_ms(emat_backend* h, double* ms, int32_t* num_parts_in_kernel);

/* ---- test hooks (not part of the boundary) ------------------------------------------------- */
/* Evaluates, on the device and point by point, the engine's own implementations of the regularised upper incomplete gamma
 * function: mode 0: out[i] = Q(a[i], x_or_q[i]); mode 1: out[i] = the x with Q(a[i], x) = x_or_q[i].  They stand in for
 * boost::math::gamma_q / gamma_q_inv (Boost 1.84, reference safe_gamma_math.h:46,68; reached from spr_study.cpp:368,463,544),
 * whose source is not part of the reference tree; tests/test_parity_gpu.py sweeps them over tests/golden/gamma_q.json. */
emat_status emat_debug_gamma(emat_backend* h, int32_t mode, int32_t n, const double* a, const double* x_or_q, double* out);
/* The device's population-model routines, point by point (reference pop_model.cpp:18-145, 247-330): op 0: out[i] = N(a[i])
 * (pop_at_time); op 1: out[i] = integral of N over [a[i], b[i]] (pop_integral).  tests/ sweeps them over the reference's
 * own expectations (tests/golden/reference_expectations.json). */
emat_status emat_debug_pop(emat_backend* h, const emat_pop_model* pop_model, int32_t op, int32_t n, const double* a, const double* b, double* out);
/* The moves' own tree queries on one resident part, query by query (reference phylo_tree.cpp:204-280, 292-299): op 0: out[i] =
 * find_MRCA_of(a[i], b[i]); op 1: out[i] = descends_from(a[i], b[i]) (0 / 1); -1 stands for k_no_node.  tests/ runs them over the
 * reference's own table of cases (phylo_tree_tests.cpp:365-525). */
emat_status emat_debug_tree_query(emat_backend* h, int32_t part_id, int32_t op, int32_t n, const int32_t* a, const int32_t* b, int32_t* out);
/* The moves' own SPR graft machinery on one resident part, with what it found written out as numbers (reference Spr_move,
 * spr_move.h:86-150, spr_move.cpp:9-1156).  `mu_proposal` is the JC69 rate of the proposal (the reference's mu_JC); whether the root
 * sequence may change is the part's includes_run_root flag (the reference's can_change_root).
 *   mode 0: analyze_graft(X);  1: ... + peel_graft;  2: ... + apply_graft of the same graft (a round trip);
 *   mode 3: analyze_graft(X), peel_graft, Spr_move::move(X, new_sibling, new_t_P), propose_new_graft(X) from the part's random
 *           stream, apply_graft, and the part's log G updated by the two delta_log_G as an accepted move does (what the reference's
 *           run_full_spr_move_test steps through, tests/spr_move_tests.cpp:1518-1641).
 * out (doubles): [0] part status afterwards (0 = fine), [1] number of grafts that follow (1; 2 in mode 3: old, then new), each graft as
 *   nbi, delta_log_G, log_alpha_mut, X, S, t_P, then per branch info: A, B, is_open, T_to_X, partial_lambda_at_A, partial_lambda_at_X,
 *   n_warm, (start, end) x n_warm, n_hot, (start, end) x n_hot, n_hot_muts, (site, from, to, t) x n, n_hot_deltas, (site, from, to) x n;
 *   after the first graft, for mode >= 1: count_min_mutations, count_closed_mutations, n, (site, from, to) x n of summarize_closed_mutations.
 * *out_len = doubles needed (EMAT_ERR_CAPACITY when more than out_cap).  tests/ runs it over the reference's own fixtures and expected
 * values (tests/spr_move_tests.cpp:142-1516, as data in tests/golden/reference_expectations.json). */
emat_status emat_debug_graft(emat_backend* h, int32_t part_id, int32_t X, double mu_proposal, int32_t mode, int32_t new_sibling, double new_t_P,
                             double* out, int32_t out_cap, int32_t* out_len);
/* The proposal's JC69 mutational-history sampler on one resident part, history by history (reference sample_mutational_history +
 * adjust_mutational_history, spr_move.cpp:1164-1370, 1409-1439), driven as the reference's own statistical test drives it
 * (tests/spr_move_tests.cpp:1795-1961): history i ends at the point (branch[i], t_end[i]) of the tree and starts T earlier from `start_seq`
 * (num_sites states); its site deltas are where start_seq differs from the tree's sequence at that point.  counts[i] = its number of
 * mutations; muts = (site, from, to, t) per mutation, histories back to back; *num_muts = their total (EMAT_ERR_CAPACITY above muts_cap).
 * Random numbers come from the part's stream. */
emat_status emat_debug_sample_history(emat_backend* h, int32_t part_id, int32_t n, const int32_t* branch, const double* t_end, const uint8_t* start_seq, double T, double mu,
                                      int32_t* counts, double* muts, int32_t muts_cap, int32_t* num_muts);
/* One tree-editing session of the moves' own device code on node X of a resident part (reference Tree_editing_session,
 * tree_editing.cpp:7-302; what Spr_move::move is built from): the constructor, then step i of n_ops: op_kind[i] = 0
 * slide_P_along_branch(op_t[i]), 1 hop_up(), 2 flip(), 3 hop_down(op_node[i]); then end().  lambda_i and the missing-site counts of
 * the nodes are kept up to date by the steps, as in a move (emat_check_derived verifies them afterwards).  tests/ runs the ten
 * cases of the reference's tests/tree_editing_tests.cpp through it. */
emat_status emat_debug_edit(emat_backend* h, int32_t part_id, int32_t X, int32_t n_ops, const int32_t* op_kind, const int32_t* op_node, const double* op_t);
/* The device's interval-set algebra on two valid sets given as (start, end) pairs (reference interval_set.h:130-138, 238-500):
 * op 1 merge, 2 intersect, 3 subtract -> pairs in `out` (room for na + nb + 1 pairs), *n_out = their number; op 5 contains
 * (site b[0]), 6 sets intersect -> *n_out = 0 / 1. */
emat_status emat_debug_interval_op(emat_backend* h, int32_t op, const int32_t* a, int32_t na, const int32_t* b, int32_t nb, int32_t* out, int32_t* n_out);
/* How often the cut-state pools (out3[0]) and the list heaps (out3[1]) of the HBM-resident tree had to grow (with
 * EMAT_TREE_TIGHT set in the environment they start without any room, so that tests reach those paths), and how many
 * cut-point states needed the large variant of k_gt_measure (out3[2]). */
emat_status emat_debug_tree_counters(emat_backend* h, int32_t* out3);

/* ---- initial-tree construction ------------------------------------------------------------------------
 * replaces: build_usher_like_tree (reference core/phylo_tree.cpp:796-1049; --v0-init-method old_usher_like through
 * build_rough_initial_tree_from_maple, cmdline.cpp:88-121) with its closing passes fix_up_missations (:414-507), pseudo_date
 * (dates.cpp:63-82) and randomize_mutation_times (:567-644).
 * Tip descriptors (reference Tip_desc, phylo_tree.h:137-143, as io.cpp's MAPLE reader fills them): per tip its date range, its
 * differences from the reference sequence of emat_set_ref_sequence (ascending sites; `from` is the reference state) and its
 * missing intervals (sorted, disjoint, non-adjacent).  Tips become nodes 0 .. num_tips-1 in this order, the inner node created
 * for tip X is X + num_tips - 1.  The O(tips x nodes) graft loop runs on the device; the tree is a function of (descriptors,
 * seed).  Input that the reference rejects (site out of range, delta onto the reference state, a site both missing and changed,
 * t_min > t_max, fewer than two tips) is EMAT_ERR_INVALID_ARGUMENT with emat_last_error saying which tip. */
typedef struct emat_tip_descs {
  int32_t num_tips;
  const float*   t_min;          /* [num_tips] */
  const float*   t_max;          /* [num_tips] */
  const int32_t* delta_offset;   /* [num_tips+1] CSR into delta_site / delta_to */
  const int32_t* delta_site;
  const uint8_t* delta_to;
  const int32_t* miss_offset;    /* [num_tips+1] CSR into miss_start / miss_end */
  const int32_t* miss_start;
  const int32_t* miss_end;
} emat_tip_descs;
emat_status emat_tree_build_usher_like(emat_backend* h, const emat_tip_descs* tips, uint64_t seed);
/* The tree it made, as a flat tree (for emat_run_create / emat_tree_upload): sizes, then the arrays. */
emat_status emat_tree_built_sizes(emat_backend* h, int32_t* num_nodes, int32_t* num_muts, int32_t* num_intervals, int32_t* num_from_states);
emat_status emat_tree_built_get(emat_backend* h, emat_flat_tree* out);
/* replaces: build_initial_phylo_tree (reference core/utree.cpp:1892-1925), the reference's DEFAULT initial tree (--v0-init-method
 * mp_plus_timing, cmdline.cpp:113-115, 437): maximum-parsimony guide tree by branch-and-bound insertion (utree.cpp:190-755), up to five
 * rebuilds in nearest-first order (:761-914), SPR refinement of tips and subtrees (:920-1081), rooting by the least-squares regression
 * of divergence on sampling date with the midpoint fall-back (:1085-1464), dating from the fitted rate (:1750-1890), then
 * fix_up_missations and randomize_mutation_times.  Host code, as in the reference (no device needed: works on a device = -1 handle);
 * same descriptors + same seed => the same tree.  The tree is written against the ROOT's sequence, as the reference's is after
 * rereference_to_root_sequence (phylo_tree.cpp:309-322): fetch that sequence with emat_tree_built_ref and hand it on wherever the tree
 * goes (emat_run_create, emat_set_ref_sequence + emat_tree_upload).  report (may be NULL): [0..2] site deltas in the guide tree, after
 * the rebuilds, after SPR refinement; [3] 0 = rooted by regression, 1 = by the midpoint fall-back. */
emat_status emat_tree_build_default(emat_backend* h, const emat_tip_descs* tips, uint64_t seed, int32_t* report /*[4]*/);
emat_status emat_tree_built_ref(emat_backend* h, uint8_t* ref_sequence /*[num_sites]*/);

/* Byte breakdown of one part's slab as the backend lays it out (works on host-only handles too): header, node records, coalescent
 * cell table, trace ring, list-heap content, list-heap capacity, scratch, and the number of cells kept. */
emat_status emat_debug_slab_layout(emat_backend* h, int32_t part_id, uint32_t* out8);

#ifdef __cplusplus
}
#endif
#endif /* EMAT_BACKEND_H_ */
