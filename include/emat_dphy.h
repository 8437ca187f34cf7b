/*
 * emat_dphy.h -- `.dphy` run files and their FlatBuffers payloads straight from the engine's flat SoA trees
 * (SURVEY section 8(f).3), so that what a GPU run produces can be opened by the reference's CLI tools and by
 * delphy-web without passing through a `Phylo_tree`.
 *
 *   reference                                                     here
 *   ---------                                                     ----
 *   phylo_tree_to_api_tree        (core/api.cpp:34-98)            emat_dphy_tree_flatbuffer
 *   phylo_tree_to_api_tree_info   (core/api.cpp:100-127)          emat_dphy_tree_info_flatbuffer
 *   run_to_api_params             (core/api.cpp:210-313)          emat_dphy_params_flatbuffer
 *   Delphy_output::output_preamble / output_state / output_epilog emat_dphy_open / emat_dphy_write_state / emat_dphy_close
 *                                 (core/delphy_output.cpp:94-141)
 *
 * File layout: doc/dphy_file_format.md, version 3.  Buffer schemas: core/api.fbs (tables Tree, TreeInfo / NodeInfo,
 * Params, ExpPopModel, SkygridPopModel; structs Node, Mutation, MissationInterval).  The buffers are size-prefixed, as
 * FlatBufferBuilder::FinishSizePrefixed leaves them; node times and mutation times are float32 in the schema
 * (api.fbs:13-29).  The encoder is written here from the published FlatBuffers wire format (no flatc, no runtime): it
 * lays every object out after the objects that refer to it, which the format allows (offsets to tables, vectors and
 * strings are unsigned and forward; a table's offset to its vtable is signed).
 *
 * Every function is host-only and needs no GPU.
 */
#ifndef EMAT_DPHY_H_
#define EMAT_DPHY_H_

#include "emat_backend.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Everything run_to_api_params reads off a Run.  Fields a GPU run does not model keep the reference's defaults
 * (emat_dphy_params_defaults: run.cpp:21-39). */
typedef struct emat_dphy_params {
  int64_t step;
  int64_t num_local_moves_per_global_move;   /* -1 = the reference's default (50 x nodes) */
  int32_t num_parts;
  double mu, mu_prior_alpha, mu_prior_beta;
  double alpha;
  const double* nu;                          /* [num_sites] or NULL (= all 1: the reference omits the vector then) */
  double hky_kappa, hky_pi[4];
  emat_pop_model pop_model;                  /* EMAT_POP_EXP or EMAT_POP_SKYGRID (a constant population is Exp with g = 0, run.cpp:21) */
  double pop_inv_n0_prior_alpha, pop_inv_n0_prior_beta, pop_g_prior_mu, pop_g_prior_scale, pop_g_min, pop_g_max;
  double skygrid_tau, skygrid_tau_prior_alpha, skygrid_tau_prior_beta, skygrid_low_gamma_barrier_loc, skygrid_low_gamma_barrier_scale;
  double skygrid_inv_nbar_prior_alpha, skygrid_inv_nbar_prior_beta;
  int32_t only_displacing_inner_nodes, topology_moves_enabled, repartitioning_enabled, alpha_move_enabled, mu_move_enabled;
  int32_t final_pop_size_move_enabled, pop_growth_rate_move_enabled, skygrid_tau_move_enabled, skygrid_low_gamma_barrier_enabled;
  double log_other_priors, log_coalescent_prior, log_G;   /* log_posterior is their sum, as in the reference */
  double total_branch_length;                /* calc_T of the tree */
} emat_dphy_params;
void emat_dphy_params_defaults(emat_dphy_params* p);

/* Each encoder writes a size-prefixed buffer into `buf` and its total length (prefix included: what the .dphy file
 * stores as the buffer's length) into `*bytes`; with buf = NULL or too small a capacity it only reports the length
 * (EMAT_ERR_BUFFER_TOO_SMALL in the latter case). */
emat_status emat_dphy_tree_flatbuffer(const emat_flat_tree* tree, const uint8_t* ref_sequence, int32_t num_sites,
                                      uint8_t* buf, uint64_t capacity, uint64_t* bytes);
/* `names` = one C string per node, or NULL: tips are then called "TIP_<i>", inner nodes get empty names. */
emat_status emat_dphy_tree_info_flatbuffer(const emat_flat_tree* tree, const char* const* names,
                                           uint8_t* buf, uint64_t capacity, uint64_t* bytes);
emat_status emat_dphy_params_flatbuffer(const emat_dphy_params* params, int32_t num_sites,
                                        uint8_t* buf, uint64_t capacity, uint64_t* bytes);

/* The file itself. */
typedef struct emat_dphy_writer emat_dphy_writer;
/* Errors: EMAT_ERR_IO when the file cannot be opened, a write or the final close falls short (every write is checked; a writer
 * that has failed once keeps answering EMAT_ERR_IO); EMAT_ERR_CAPACITY for a buffer whose length does not fit the format's
 * 32-bit length fields. */
emat_status emat_dphy_open(const char* path, const char* core_version, int32_t build_number, const char* commit,
                           int32_t steps_per_sample, const emat_dphy_params* params_for_flags,
                           const emat_flat_tree* tree, const char* const* names, emat_dphy_writer** out);
emat_status emat_dphy_write_state(emat_dphy_writer* w, const emat_flat_tree* tree, const uint8_t* ref_sequence, int32_t num_sites,
                                  const emat_dphy_params* params);
/* Writes the epilogue (sentinel, default metadata JSON, position of the sentinel) and closes the file. */
emat_status emat_dphy_close(emat_dphy_writer* w);

#ifdef __cplusplus
}
#endif
#endif /* EMAT_DPHY_H_ */
