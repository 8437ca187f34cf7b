/*
 * emat_host.h -- C-ABI of the host-side driver that sits ABOVE the engine boundary
 * (include/emat_backend.h): the part of Delphy's `Run` that the north star keeps on the host.
 *
 *   reference                                              here
 *   ---------                                              ----
 *   generate_random_partition_stencil, partition_tree      emat_run_repartition
 *     (core/tree_partitioning.h:139-239)
 *   Run::repartition (core/run.cpp:110-193)                emat_run_repartition
 *   Run::push_global_params_to_subruns (run.cpp:267-275)   emat_run_push_params
 *   Run::run_local_moves (run.cpp:682-693)                 emat_run_moves        -> emat_run_moves_even of the backend (remainder spread over the parts)
 *   Run::reassemble (run.cpp:195-256)                      emat_run_reassemble
 *   Run::do_mcmc_steps without the global moves            emat_run_do_mcmc_steps
 *     (run.cpp:622-657; global moves are out of scope)
 *
 * Also exports the seeded synthetic-EMAT generator used by the bench and the tests
 * (SURVEY section 8(d) configs C1..C5).
 */
#ifndef EMAT_HOST_H_
#define EMAT_HOST_H_

#include "emat_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic EMATs ------------------------------------------------------------------------ */
typedef struct emat_synth_params {
  int32_t num_tips;
  int32_t num_sites;
  double tip_span;              /* tip dates ~ U[0, tip_span) days */
  double tip_date_uncertainty;  /* half-width of [t_min, t_max] for the uncertain tips */
  double frac_uncertain_tips;
  double pop_n0;                /* N(t0) * generation time at the latest tip, days */
  double pop_growth;            /* per day */
  double mu;                    /* substitutions / site / day */
  double kappa;
  double pi[4];
  int32_t gaps_per_tip;
  double mean_gap_len;
  uint64_t seed;
} emat_synth_params;

typedef struct emat_synth emat_synth;
emat_status emat_synth_create(const emat_synth_params* p, emat_synth** out);
void emat_synth_destroy(emat_synth* s);
/* Pointers stay valid until emat_synth_destroy. */
emat_status emat_synth_get(emat_synth* s, emat_flat_tree* tree_view, const uint8_t** ref_sequence, double* t_max_tip);

/* ---- host tree container + partitioning (usable without any GPU) -------------------------------- */
typedef struct emat_run emat_run;

/* `backend` may be NULL: the driver then only partitions / reassembles (CPU tests, and feeding the
 * parity oracle with exactly the same subtrees as the HIP engine). */
emat_status emat_run_create(emat_backend* backend, const emat_flat_tree* tree, const uint8_t* ref_sequence,
                            int32_t num_sites, uint64_t seed, emat_run** out);
emat_status emat_run_destroy(emat_run* r);
const char* emat_run_last_error(const emat_run* r);

/* reference Run::set_num_parts (run.h:46-47) */
emat_status emat_run_set_num_parts(emat_run* r, int32_t num_parts);
/* NOT in the reference: at every repartition, parts of more than `max_nodes` nodes get further cut nodes, because on the GPU a pass
 * lasts as long as the chain of its largest part, and the parts of a stencil drift apart in size while it is in use (run.cpp:87-108
 * redraws stencils every 200 cycles only).  The extra cut nodes are drawn uniformly at random among the part's inner nodes -- a rule
 * that reads nothing a pass can change, so that it leaves the sampler's stationary distribution alone (emat_run.cpp, refine_stencil).
 * 0 (the DEFAULT since round 6) = off: the reference's rule exactly, so that a run that sets nothing reproduces the reference's (and the
 * oracle's) partition; -1 = three times the mean part size, at least 64 (what bench.py's `inclusive` and the posterior scripts opt into);
 * > 0 = that many nodes. */
emat_status emat_run_set_max_part_nodes(emat_run* r, int32_t max_nodes);
/* Test hook of the rule above: the (sorted) cut nodes that the last repartition's draw -- same stencil, same random stream of the
 * refinement -- gives on the tree as it is NOW.  A pass only moves nodes within parts and the rule reads nothing such a pass changes, so
 * after emat_run_reassemble the answer must be what it was right after emat_run_repartition (tests/test_host_driver.py): the premise of
 * the argument that the limit cannot bias the sampler.  `*num_cut_nodes`: capacity in, count out.  Changes no state. */
emat_status emat_run_debug_redraw_partition(emat_run* r, int32_t* cut_nodes, int32_t* num_cut_nodes);
/* The last repartition: how many parts, the node count of the largest, how many cut nodes the rule above added, the limit in effect. */
emat_status emat_run_partition_stats(emat_run* r, int32_t* num_parts, int32_t* largest_part_nodes, int32_t* extra_cuts, int32_t* max_part_nodes_in_effect);
/* HKY substitution model with per-site relative rates nu_l (NULL = all 1): reference Hky_model +
 * Run::derive_evo (evo_hky.cpp:7-50).  One site partition. */
emat_status emat_run_set_hky(emat_run* r, double mu, double kappa, const double pi[4], const double* nu_l);
emat_status emat_run_set_pop_model(emat_run* r, const emat_pop_model* pm);
emat_status emat_run_set_coalescent_t_step(emat_run* r, double t_step);
emat_status emat_run_set_flags(emat_run* r, int32_t only_displacing_inner_nodes, int32_t topology_moves_enabled);
/* How emat_run_moves / emat_run_do_mcmc_steps split `count` moves over P parts.  Default (0): count / P on every part and the
 * remainder one move each on the first parts -- NOT the reference's rule, which gives the whole remainder to subrun 0
 * (Run::run_local_moves, run.cpp:683-689): harmless with 8 parts, but with thousands of parts it makes part 0 the longest
 * chain of every pass by a factor of several.  1: the reference's rule, for runs that must reproduce its per-part move counts
 * (emat_run_local_moves of the backend).  Sharded runs (emat_run_moves_sharded) always spread. */
emat_status emat_run_set_reference_remainder(emat_run* r, int32_t on);
/* Run::set_paranoid (reference run.h:220-224; CLI flag --v0-paranoid, cmdline.cpp:177): emat_run_do_mcmc_steps then runs
 * emat_check_derived on every part after every pass of local moves and stops with its error if one is off. */
emat_status emat_run_set_paranoid(emat_run* r, int32_t on);

/* keep the authoritative tree in HBM (emat_tree_* of the backend).  From the next emat_run_repartition on,
 * a cycle moves only the partition to the device and topology + node times back: emat_run_repartition draws and applies
 * the stencil on the topology and calls emat_tree_repartition, emat_run_reassemble calls emat_tree_reassemble (which also
 * does Run::normalize_root's work).  emat_run_tree_get / emat_run_tree_sizes download the tree when asked;
 * emat_run_part_* (host copies of the parts) are not available.  Same seeds, same partitions, same trees as the host
 * cycle, bit for bit.  `on` = 0 brings the tree back to the host. */
emat_status emat_run_set_device_tree(emat_run* r, int32_t on);
/* A sharded run with the tree on the devices: emat_run_repartition cuts every process's own block of parts out of its own
 * copy of the tree (emat_tree_repartition_range); the gather goes through the backend's staged calls with the caller's
 * exchange between them (emat_backend.h: emat_tree_get_root_deltas ... emat_tree_reassemble_end; delphy_amd/sharding.py),
 * after which this tells the driver that the parts are back and which sites of the reference sequence changed. */
emat_status emat_run_note_device_reassembled(emat_run* r, int32_t num_root_deltas, const int32_t* site, const uint8_t* to);
/* Cut the tree into parts and (when a backend is attached) upload them and build their coalescent parts. */
emat_status emat_run_repartition(emat_run* r);
emat_status emat_run_num_parts(emat_run* r, int32_t* num_parts, int32_t* root_part_index);
/* Sizes / contents of part i as built by the last repartition (to feed another engine, e.g. the oracle). */
emat_status emat_run_part_sizes(emat_run* r, int32_t part, int32_t* num_nodes, int32_t* num_muts, int32_t* num_intervals, int32_t* num_from_states);
emat_status emat_run_part_get(emat_run* r, int32_t part, emat_flat_tree* out, int32_t* includes_run_root, uint64_t* seed);
/* Replace part i's subtree (e.g. with what another engine computed) before emat_run_reassemble. */
emat_status emat_run_part_put(emat_run* r, int32_t part, const emat_flat_tree* subtree);

emat_status emat_run_push_params(emat_run* r);
emat_status emat_run_moves(emat_run* r, int64_t count);
/* Gather the parts back into the whole tree; with a backend attached the parts are downloaded first. */
emat_status emat_run_reassemble(emat_run* r);
/* Several drivers in ONE process that are bound to draw the same partitions (same seed, same tree, same settings: the shards of
 * emat_run_create_multi) draw once: a follower takes the cut nodes its leader drew for the cycle instead of picking and refining the
 * stencil again on the same host cores.  The leader draws at its own emat_run_repartition, or ahead of it with emat_run_draw_partition,
 * after which leader and followers can cut side by side.  (Not in the reference, whose Run is one object; its refresh_partition_stencils
 * + pick, run.cpp:87-108, 127-129, is what is being shared.) */
emat_status emat_run_follow_draws(emat_run* follower, emat_run* leader /* NULL: draw for itself again */);
emat_status emat_run_draw_partition(emat_run* leader);
/* ---- one run over several processes, one GPU each ---------------------------------------------------
 * Every process creates the same run (same tree, same seed) with its own backend and calls emat_run_set_shard once.
 * The partition is then computed identically everywhere, but a process uploads only its contiguous block of parts
 * (emat_run_shard_range; backend part id = part - part_lo).  One cycle (reference run.cpp:622-657) is
 *
 *   emat_run_repartition                                     every rank, no communication
 *   emat_run_coalescent_begin -> all-reduce MIN / MAX        the staged form of emat_build_coalescent_parts
 *   emat_coalescent_set_range / _local_grid -> all-reduce SUM / _sample -> all-reduce SUM / _finish   (on the backend)
 *   emat_run_moves_sharded(count)                            count / parts moves on every part of the run, the remainder one move each on the first parts
 *   emat_run_pack_local_parts -> all-gather of the byte buffers -> emat_run_unpack_parts for every other rank's buffer
 *   emat_run_reassemble                                      every rank ends up with the same whole tree
 *   emat_get_totals on the backend -> all-reduce SUM (2 doubles)
 *
 * The collectives (RCCL over xGMI, or gloo in tests) belong to the caller: delphy_amd/sharding.py drives them through
 * torch.distributed.  emat_run_do_mcmc_steps is the single-process cycle and refuses a sharded run. */
emat_status emat_run_set_shard(emat_run* r, int32_t rank, int32_t world);
emat_status emat_run_shard_range(emat_run* r, int32_t* part_lo, int32_t* part_hi, int32_t* local_root_part /* -1: the root part lives on another rank */);
emat_status emat_run_coalescent_begin(emat_run* r, double* local_t_min, double* local_t_max);
emat_status emat_run_moves_sharded(emat_run* r, int64_t count);
/* Serialises this rank's parts as they are on its device now.  Call with buf = NULL to learn the size. */
emat_status emat_run_pack_local_parts(emat_run* r, uint8_t* buf, uint64_t capacity, uint64_t* bytes_needed);
emat_status emat_run_unpack_parts(emat_run* r, const uint8_t* buf, uint64_t bytes);

/* ---- one run over several GPUs of ONE process (delphy_amd/csrc/emat_multi.cpp) --------------------------------------------
 * replaces: the thread pool behind Run::run_local_moves (reference core/run.cpp:682-693: one task per Subrun, in-process) for a
 * C++ `Run` that owns several MI355X.  n backends, one per entry of `devices` (cfg->device is ignored), every one with the whole
 * tree in its HBM and a contiguous block of the partition's parts; the exchanges of a sharded cycle (above) are done inside, in
 * C++: an all-gather of what every shard's parts own -- emat_tree_export_nodes writes into the device buffer that
 * ncclAllGather sends over xGMI, emat_tree_apply_nodes reads what arrived -- and an ncclAllReduce of the two log-posterior
 * totals.  `exchange`: EMAT_EXCHANGE_RCCL (librccl.so is loaded at run time with dlopen: no link-time dependency),
 * EMAT_EXCHANGE_HOST (the same steps through host buffers: what two backends sharing one device must use, RCCL taking one
 * rank per device), or EMAT_EXCHANGE_AUTO (RCCL when every shard has a device of its own and the library loads).
 * One cycle = emat_multi_repartition, emat_multi_run_moves (returns with the kernels of every GPU in flight),
 * emat_multi_reassemble; emat_multi_do_mcmc_steps strings them together as Run::do_mcmc_steps does (run.cpp:622-657, without the
 * global moves).  The shards' own handles stay reachable for everything else (emat_multi_backend / emat_multi_shard: statistics of
 * the global moves, paranoid checks, part downloads).  Entry points are not thread-safe with respect to each other. */
typedef struct emat_multi emat_multi;
enum { EMAT_EXCHANGE_HOST = 0, EMAT_EXCHANGE_RCCL = 1, EMAT_EXCHANGE_AUTO = 2 };
emat_status emat_run_create_multi(const int32_t* devices, int32_t n, const emat_config* cfg, const emat_flat_tree* tree, const uint8_t* ref_sequence,
                                  int32_t num_sites, uint64_t seed, int32_t exchange, emat_multi** out);
emat_status emat_multi_destroy(emat_multi* m);
const char* emat_multi_last_error(const emat_multi* m);
const char* emat_multi_exchange(const emat_multi* m);          /* what the exchange goes through, in words */
int32_t emat_multi_num_shards(const emat_multi* m);
emat_backend* emat_multi_backend(emat_multi* m, int32_t shard);
emat_run* emat_multi_shard(emat_multi* m, int32_t shard);
emat_status emat_multi_set_num_parts(emat_multi* m, int32_t num_parts);
emat_status emat_multi_set_max_part_nodes(emat_multi* m, int32_t max_nodes);
emat_status emat_multi_set_option(emat_multi* m, const char* key, const char* value);   /* emat_set_option on every shard's backend */
/* Which RCCL to dlopen (name or path) instead of the usual names; process-wide, before emat_run_create_multi; NULL = the usual names. */
emat_status emat_multi_set_rccl_library(const char* name);
emat_status emat_multi_debug_rccl_load(char* err, int32_t err_cap);   /* test hook: EMAT_OK if RCCL loads with the present setting, else EMAT_ERR_HIP and why */
emat_status emat_multi_set_hky(emat_multi* m, double mu, double kappa, const double pi[4], const double* nu_l);
emat_status emat_multi_set_pop_model(emat_multi* m, const emat_pop_model* pm);
emat_status emat_multi_set_coalescent_t_step(emat_multi* m, double t_step);
emat_status emat_multi_set_flags(emat_multi* m, int32_t only_displacing_inner_nodes, int32_t topology_moves_enabled);
emat_status emat_multi_set_paranoid(emat_multi* m, int32_t on);
emat_status emat_multi_repartition(emat_multi* m);
emat_status emat_multi_run_moves(emat_multi* m, int64_t count);
emat_status emat_multi_check_derived(emat_multi* m, double tol_scale);
emat_status emat_multi_reassemble(emat_multi* m);
emat_status emat_multi_get_totals(emat_multi* m, double* log_G, double* log_augmented_coalescent_prior);
emat_status emat_multi_do_mcmc_steps(emat_multi* m, int64_t steps, int64_t local_moves_per_cycle);
emat_status emat_multi_tree_sizes(emat_multi* m, int32_t* num_nodes, int32_t* num_muts, int32_t* num_intervals, int32_t* num_from_states);
emat_status emat_multi_tree_get(emat_multi* m, int32_t shard, emat_flat_tree* out, uint8_t* ref_sequence /*[L]*/);

/* replaces: calc_Ttwiddle_l(tree_, evo_) (reference phylo_tree_calc.cpp:176-222, called at run.cpp:1109, 1184) while the parts
 * are on the device: the driver knows the tree of parts, which the staged engine calls need (emat_backend.h).  Single process;
 * a sharded run all-gathers emat_get_part_tree_lengths, calls emat_run_Ttwiddle_ext + emat_Ttwiddle_l_partial on every rank,
 * all-reduces S, R and the root's tree length, and finishes anywhere. */
emat_status emat_run_get_Ttwiddle_l(emat_run* r, double* Ttwiddle_l /*[num_sites]*/);
emat_status emat_run_Ttwiddle_ext(emat_run* r, const double* tree_length_of_part /*[all parts of the run]*/, int32_t* ext_offset /*[local parts + 1]*/,
                                  int32_t* ext_node, double* ext_length, int32_t capacity, int32_t* count);

/* repartition -> [push params, local moves, reassemble] per cycle of `local_moves_per_cycle` moves
 * (<= 0: 50 x nodes, the reference default run.cpp:669-672), repartitioning at every cycle boundary. */
emat_status emat_run_do_mcmc_steps(emat_run* r, int64_t steps, int64_t local_moves_per_cycle);

/* Whole tree out (sizes first). The reference sequence may have been re-referenced to the root
 * sequence (Run::normalize_root, run.cpp:258-265). */
emat_status emat_run_tree_sizes(emat_run* r, int32_t* num_nodes, int32_t* num_muts, int32_t* num_intervals, int32_t* num_from_states);
emat_status emat_run_tree_get(emat_run* r, emat_flat_tree* out, uint8_t* ref_sequence /*[L]*/);
emat_status emat_run_t_max_tip(emat_run* r, double* t_max_tip);

#ifdef __cplusplus
}
#endif
#endif /* EMAT_HOST_H_ */
