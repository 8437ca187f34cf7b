// emat_dphy.cpp -- `.dphy` files and their FlatBuffers payloads from flat SoA trees (include/emat_dphy.h).
//
// Replaces, for trees that never were a Phylo_tree: phylo_tree_to_api_tree / _tree_info, run_to_api_params
// (reference core/api.cpp:34-127, 210-313) and Delphy_output (core/delphy_output.cpp:94-141).  Schemas: core/api.fbs;
// file layout: doc/dphy_file_format.md (version 3).
//
// The FlatBuffers encoder below is written from the wire format, not from the FlatBuffers library (absent here, and a
// builder that works back to front is not what a writer that already knows every size needs): a buffer is laid out
// FORWARD -- root table, then the objects it refers to, then theirs -- so every unsigned offset points ahead, as the
// format requires, and a table's signed offset to its vtable points back to the vtable written just before it.
// Alignment is relative to the start of the size-prefixed buffer and its length is a multiple of 8, as
// FlatBufferBuilder::FinishSizePrefixed leaves it.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/emat_dphy.h"

namespace {

struct Fb {
  std::vector<uint8_t> b;
  size_t pos() const { return b.size(); }
  void pad_to(size_t align, size_t bias = 0) { while ((b.size() + bias) % align) b.push_back(0); }
  template <class T> void put(T v) { const uint8_t* p = (const uint8_t*)&v; b.insert(b.end(), p, p + sizeof(T)); }
  void put_bytes(const void* p, size_t n) { const uint8_t* q = (const uint8_t*)p; b.insert(b.end(), q, q + n); }
  template <class T> void patch(size_t at, T v) { std::memcpy(b.data() + at, &v, sizeof(T)); }
  // an unsigned forward offset stored at `at`, to the object that starts at the current position
  void point_here(size_t at) { patch<uint32_t>(at, (uint32_t)(pos() - at)); }
  void begin() { b.clear(); put<uint32_t>(0); put<uint32_t>(0); }   // size prefix, offset to the root table
  void finish() { pad_to(8); patch<uint32_t>(0, (uint32_t)(b.size() - 4)); }
  // vector header: the elements start right after the 32-bit length and must be aligned to `elem_align`
  void begin_vector(uint32_t count, size_t elem_align) { pad_to(elem_align < 4 ? 4 : elem_align, 4); put<uint32_t>(count); }
  void string(const char* s) { const size_t n = std::strlen(s); pad_to(4); put<uint32_t>((uint32_t)n); put_bytes(s, n); put<uint8_t>(0); }
};

// One table: fields by schema id; scalars are given by value, references (to tables, vectors, strings written later) leave a
// slot that the caller patches with point_here once the target's position is known.
struct Field { int id; int size; uint8_t bytes[8]; };
struct Table {
  std::vector<Field> f;
  template <class T> void scalar(int id, T v) { Field x; x.id = id; x.size = (int)sizeof(T); std::memset(x.bytes, 0, 8); std::memcpy(x.bytes, &v, sizeof(T)); f.push_back(x); }
  void ref(int id) { scalar<uint32_t>(id, 0); }
  // Writes vtable + table; returns the table's position; slot_of[id] = position of that field's inline slot.
  size_t write(Fb& fb, std::vector<size_t>& slot_of) const {
    int max_id = -1; for (auto& x : f) max_id = x.id > max_id ? x.id : max_id;
    // inline layout: 8-byte fields first (the table is placed so that its offset 4 is 8-aligned), then 4, 2, 1
    std::vector<uint16_t> off(max_id + 1, 0);
    uint16_t cur = 4;
    for (int sz : {8, 4, 2, 1}) for (auto& x : f) if (x.size == sz) { off[x.id] = cur; cur = (uint16_t)(cur + sz); }
    const uint16_t table_bytes = cur, vt_bytes = (uint16_t)(4 + 2 * (max_id + 1));
    fb.pad_to(2);
    // vtable, then padding so that (table position + 4) is a multiple of 8
    size_t vt = fb.pos();
    while ((vt + vt_bytes + 4) % 8 != 0 || (vt + vt_bytes) % 4 != 0) { fb.put<uint8_t>(0); vt = fb.pos(); }
    fb.put<uint16_t>(vt_bytes); fb.put<uint16_t>(table_bytes);
    for (int i = 0; i <= max_id; ++i) fb.put<uint16_t>(off[i]);
    const size_t tp = fb.pos();
    fb.put<int32_t>((int32_t)(tp - vt));
    fb.b.resize(tp + table_bytes, 0);
    slot_of.assign(max_id + 1, 0);
    for (auto& x : f) { std::memcpy(fb.b.data() + tp + off[x.id], x.bytes, x.size); slot_of[x.id] = tp + off[x.id]; }
    return tp;
  }
};

emat_status deliver(const Fb& fb, uint8_t* buf, uint64_t cap, uint64_t* bytes) {
  if (bytes) *bytes = fb.b.size();
  if (!buf) return EMAT_OK;
  if (cap < fb.b.size()) return EMAT_ERR_BUFFER_TOO_SMALL;
  std::memcpy(buf, fb.b.data(), fb.b.size());
  return EMAT_OK;
}

bool tree_ok(const emat_flat_tree* t) { return t && t->num_nodes > 0 && t->parent && t->child0 && t->child1 && t->t && t->t_min && t->t_max && t->mut_offset && t->miss_offset; }

}  // namespace

extern "C" {

void emat_dphy_params_defaults(emat_dphy_params* p) {   // the reference's Run constructor (core/run.cpp:21-39)
  if (!p) return;
  std::memset(p, 0, sizeof *p);
  p->num_local_moves_per_global_move = -1; p->num_parts = 1;
  p->mu = 1e-3 / 365.0; p->mu_prior_alpha = 1.0; p->mu_prior_beta = 0.0; p->alpha = 10.0;
  p->hky_kappa = 1.0; for (int a = 0; a < 4; ++a) p->hky_pi[a] = 0.25;
  p->pop_model.kind = EMAT_POP_EXP; p->pop_model.p[0] = 0.0; p->pop_model.p[1] = 1000.0; p->pop_model.p[2] = 0.0; p->pop_model.p[3] = 1.0;
  p->pop_g_prior_mu = 0.001 / 365.0; p->pop_g_prior_scale = 30.701135 / 365.0; p->pop_g_min = -INFINITY; p->pop_g_max = +INFINITY;
  p->skygrid_tau = 1.0; p->skygrid_tau_prior_alpha = 0.001; p->skygrid_tau_prior_beta = 0.001;
  p->skygrid_low_gamma_barrier_loc = std::log(1.0); p->skygrid_low_gamma_barrier_scale = -std::log(0.70);
  p->topology_moves_enabled = 1; p->repartitioning_enabled = 1; p->mu_move_enabled = 1; p->final_pop_size_move_enabled = 1; p->pop_growth_rate_move_enabled = 1;
}

// table Tree { nodes:[Node]; mutations:[Mutation]; missation_intervals:[MissationInterval]; ref_seq:[RealSeqLetter]; root_node:int32; }
emat_status emat_dphy_tree_flatbuffer(const emat_flat_tree* tree, const uint8_t* ref, int32_t L, uint8_t* buf, uint64_t cap, uint64_t* bytes) {
  if (!tree_ok(tree) || !ref || L <= 0) return EMAT_ERR_INVALID_ARGUMENT;
  const int n = tree->num_nodes;
  const int nm = tree->mut_offset[n], ni = tree->miss_offset[n];
  Fb fb; fb.begin();
  Table t; t.ref(0); t.ref(1); t.ref(2); t.ref(3); t.scalar<int32_t>(4, tree->root);
  std::vector<size_t> slot;
  const size_t tp = t.write(fb, slot);
  fb.patch<uint32_t>(4, (uint32_t)(tp - 4));
  // struct Node { parent, left_child, right_child: int32; t: float32 }  (16 bytes)
  fb.begin_vector((uint32_t)n, 4); fb.patch<uint32_t>(slot[0], (uint32_t)(fb.pos() - 4 - slot[0]));
  for (int i = 0; i < n; ++i) { fb.put<int32_t>(tree->parent[i]); fb.put<int32_t>(tree->child0[i]); fb.put<int32_t>(tree->child1[i]); fb.put<float>((float)tree->t[i]); }
  // struct Mutation { branch, site: int32; from, to: uint8; t: float32 }  (16 bytes, two bytes of padding before t), by branch
  fb.begin_vector((uint32_t)nm, 4); fb.patch<uint32_t>(slot[1], (uint32_t)(fb.pos() - 4 - slot[1]));
  for (int i = 0; i < n; ++i) for (int k = tree->mut_offset[i]; k < tree->mut_offset[i + 1]; ++k) {
    fb.put<int32_t>(i); fb.put<int32_t>(tree->mut_site[k]); fb.put<uint8_t>(tree->mut_from[k]); fb.put<uint8_t>(tree->mut_to[k]); fb.put<uint16_t>(0); fb.put<float>((float)tree->mut_t[k]);
  }
  // struct MissationInterval { branch, start_site, end_site: int32 }  (12 bytes), by branch then start
  fb.begin_vector((uint32_t)ni, 4); fb.patch<uint32_t>(slot[2], (uint32_t)(fb.pos() - 4 - slot[2]));
  for (int i = 0; i < n; ++i) for (int k = tree->miss_offset[i]; k < tree->miss_offset[i + 1]; ++k) { fb.put<int32_t>(i); fb.put<int32_t>(tree->miss_start[k]); fb.put<int32_t>(tree->miss_end[k]); }
  // ref_seq: [RealSeqLetter] (uint8: A, C, G, T = 0..3, the engine's own coding)
  fb.begin_vector((uint32_t)L, 1); fb.patch<uint32_t>(slot[3], (uint32_t)(fb.pos() - 4 - slot[3]));
  fb.put_bytes(ref, (size_t)L);
  fb.finish();
  return deliver(fb, buf, cap, bytes);
}

// table TreeInfo { node_infos:[NodeInfo]; }  table NodeInfo { name:string (0); has_uncertain_t:bool (1); t_min:float32 (2); t_max:float32 (3); }
emat_status emat_dphy_tree_info_flatbuffer(const emat_flat_tree* tree, const char* const* names, uint8_t* buf, uint64_t cap, uint64_t* bytes) {
  if (!tree_ok(tree)) return EMAT_ERR_INVALID_ARGUMENT;
  const int n = tree->num_nodes;
  Fb fb; fb.begin();
  Table root; root.ref(0);
  std::vector<size_t> slot;
  const size_t tp = root.write(fb, slot);
  fb.patch<uint32_t>(4, (uint32_t)(tp - 4));
  fb.begin_vector((uint32_t)n, 4); fb.patch<uint32_t>(slot[0], (uint32_t)(fb.pos() - 4 - slot[0]));
  const size_t elems = fb.pos();
  fb.b.resize(fb.b.size() + (size_t)n * 4, 0);
  std::vector<size_t> name_slot(n);
  for (int i = 0; i < n; ++i) {
    Table ni; ni.ref(0);
    const bool tip = tree->child0[i] == EMAT_NO_NODE;
    if (tip && tree->t_min[i] != tree->t_max[i]) { ni.scalar<uint8_t>(1, 1); ni.scalar<float>(2, tree->t_min[i]); ni.scalar<float>(3, tree->t_max[i]); }   // api.cpp:113-117
    std::vector<size_t> s2;
    const size_t p = ni.write(fb, s2);
    fb.patch<uint32_t>(elems + (size_t)i * 4, (uint32_t)(p - (elems + (size_t)i * 4)));
    name_slot[i] = s2[0];
  }
  char tmp[32];
  for (int i = 0; i < n; ++i) {
    const char* nm = names ? (names[i] ? names[i] : "") : "";
    if (!names && tree->child0[i] == EMAT_NO_NODE) { std::snprintf(tmp, sizeof tmp, "TIP_%d", i); nm = tmp; }
    fb.pad_to(4);
    fb.point_here(name_slot[i]);
    fb.string(nm);
  }
  fb.finish();
  return deliver(fb, buf, cap, bytes);
}

// table Params (api.fbs; ids in comments), union PopModel { ExpPopModel = 1, SkygridPopModel = 2 } at ids 29 (type) / 30 (value)
emat_status emat_dphy_params_flatbuffer(const emat_dphy_params* q, int32_t L, uint8_t* buf, uint64_t cap, uint64_t* bytes) {
  if (!q || L <= 0) return EMAT_ERR_INVALID_ARGUMENT;
  const emat_pop_model& pm = q->pop_model;
  const bool is_exp = pm.kind == EMAT_POP_EXP || pm.kind == EMAT_POP_CONST, is_sky = pm.kind == EMAT_POP_SKYGRID;
  if (!is_exp && !is_sky) return EMAT_ERR_INVALID_ARGUMENT;
  if (is_sky && (pm.skygrid_num_knots < 1 || !pm.skygrid_x || !pm.skygrid_gamma)) return EMAT_ERR_INVALID_ARGUMENT;
  const double t0 = pm.kind == EMAT_POP_CONST ? 0.0 : pm.p[0], n0 = pm.kind == EMAT_POP_CONST ? pm.p[0] : pm.p[1], g = pm.kind == EMAT_POP_CONST ? 0.0 : pm.p[2], min_pop = pm.kind == EMAT_POP_CONST ? 0.0 : pm.p[3];
  bool nu_all_one = true;
  if (q->nu) for (int l = 0; l < L; ++l) if (q->nu[l] != 1.0) { nu_all_one = false; break; }
  Fb fb; fb.begin();
  Table t;
  t.scalar<int64_t>(0, q->step); t.scalar<int64_t>(1, q->num_local_moves_per_global_move); t.scalar<int32_t>(2, q->num_parts);
  t.scalar<double>(3, q->mu); t.scalar<double>(38, q->mu_prior_alpha); t.scalar<double>(39, q->mu_prior_beta); t.scalar<double>(4, q->alpha);
  if (q->nu && !nu_all_one) t.ref(5);                                                          // api.cpp:218-221
  t.scalar<double>(6, q->hky_kappa); for (int a = 0; a < 4; ++a) t.scalar<double>(7 + a, q->hky_pi[a]);
  t.scalar<uint8_t>(29, (uint8_t)(is_exp ? 1 : 2)); t.ref(30);
  t.scalar<double>(40, q->pop_inv_n0_prior_alpha); t.scalar<double>(41, q->pop_inv_n0_prior_beta);
  t.scalar<double>(42, q->pop_g_prior_mu); t.scalar<double>(43, q->pop_g_prior_scale); t.scalar<double>(44, q->pop_g_min); t.scalar<double>(45, q->pop_g_max);
  t.scalar<double>(31, q->skygrid_tau); t.scalar<double>(32, q->skygrid_tau_prior_alpha); t.scalar<double>(33, q->skygrid_tau_prior_beta);
  t.scalar<double>(36, q->skygrid_low_gamma_barrier_loc); t.scalar<double>(37, q->skygrid_low_gamma_barrier_scale);
  t.scalar<double>(46, q->skygrid_inv_nbar_prior_alpha); t.scalar<double>(47, q->skygrid_inv_nbar_prior_beta);
  if (is_exp) { t.scalar<double>(26, t0); t.scalar<double>(11, n0); t.scalar<double>(12, g); }   // deprecated copies, api.cpp:277-283
  t.scalar<uint8_t>(13, q->only_displacing_inner_nodes != 0); t.scalar<uint8_t>(14, q->topology_moves_enabled != 0); t.scalar<uint8_t>(15, q->repartitioning_enabled != 0);
  t.scalar<uint8_t>(16, q->alpha_move_enabled != 0); t.scalar<uint8_t>(25, q->mu_move_enabled != 0); t.scalar<uint8_t>(27, q->final_pop_size_move_enabled != 0);
  t.scalar<uint8_t>(28, q->pop_growth_rate_move_enabled != 0); t.scalar<uint8_t>(34, q->skygrid_tau_move_enabled != 0); t.scalar<uint8_t>(35, q->skygrid_low_gamma_barrier_enabled != 0);
  t.scalar<double>(17, q->log_G + q->log_coalescent_prior + q->log_other_priors); t.scalar<double>(18, q->log_other_priors);
  t.scalar<double>(19, q->log_coalescent_prior); t.scalar<double>(20, q->log_G); t.scalar<double>(21, q->total_branch_length);
  std::vector<size_t> slot;
  const size_t tp = t.write(fb, slot);
  fb.patch<uint32_t>(4, (uint32_t)(tp - 4));
  // the population model's own table, then the vectors
  size_t xk_slot = 0, gk_slot = 0;
  {
    Table pt; std::vector<size_t> s2;
    if (is_exp) { pt.scalar<double>(0, t0); pt.scalar<double>(1, n0); pt.scalar<double>(2, g); pt.scalar<double>(3, min_pop); }
    else { pt.scalar<int8_t>(0, (int8_t)(pm.skygrid_type == 1 ? 1 : 2)); pt.ref(1); pt.ref(2); }
    const size_t pp = pt.write(fb, s2);
    fb.patch<uint32_t>(slot[30], (uint32_t)(pp - slot[30]));
    if (is_sky) { xk_slot = s2[1]; gk_slot = s2[2]; }
  }
  if (q->nu && !nu_all_one) { fb.begin_vector((uint32_t)L, 8); fb.patch<uint32_t>(slot[5], (uint32_t)(fb.pos() - 4 - slot[5])); fb.put_bytes(q->nu, (size_t)L * 8); }
  if (is_sky) {
    fb.begin_vector((uint32_t)pm.skygrid_num_knots, 8); fb.patch<uint32_t>(xk_slot, (uint32_t)(fb.pos() - 4 - xk_slot)); fb.put_bytes(pm.skygrid_x, (size_t)pm.skygrid_num_knots * 8);
    fb.begin_vector((uint32_t)pm.skygrid_num_knots, 8); fb.patch<uint32_t>(gk_slot, (uint32_t)(fb.pos() - 4 - gk_slot)); fb.put_bytes(pm.skygrid_gamma, (size_t)pm.skygrid_num_knots * 8);
  }
  fb.finish();
  return deliver(fb, buf, cap, bytes);
}

// ---- the file (Delphy_output, delphy_output.cpp:94-141) ------------------------------------------------------------------
// Every write is checked: a full disk must not leave a truncated run file behind a row of EMAT_OKs.  Lengths are 32-bit in the
// format (doc/dphy_file_format.md), so a buffer that does not fit one is refused rather than truncated.
struct emat_dphy_writer { FILE* f; bool failed; };

static bool w_bytes(emat_dphy_writer* w, const void* p, size_t n) { if (w->failed) return false; if (n && std::fwrite(p, 1, n, w->f) != n) w->failed = true; return !w->failed; }
static bool w_u32(emat_dphy_writer* w, uint32_t v) { uint8_t b[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)}; return w_bytes(w, b, 4); }
static bool w_u64(emat_dphy_writer* w, uint64_t v) { return w_u32(w, (uint32_t)v) && w_u32(w, (uint32_t)(v >> 32)); }
static bool w_str(emat_dphy_writer* w, const char* s) { const size_t n = std::strlen(s); if (n > 0xffffffffull) { w->failed = true; return false; } return w_u32(w, (uint32_t)n) && w_bytes(w, s, n); }
static bool fits_u32(size_t n) { return n <= 0xffffffffull; }

emat_status emat_dphy_open(const char* path, const char* core_version, int32_t build_number, const char* commit, int32_t steps_per_sample,
                           const emat_dphy_params* q, const emat_flat_tree* tree, const char* const* names, emat_dphy_writer** out) {
  if (!path || !core_version || !commit || !q || !tree_ok(tree) || !out) return EMAT_ERR_INVALID_ARGUMENT;
  uint64_t n = 0;
  emat_status st = emat_dphy_tree_info_flatbuffer(tree, names, nullptr, 0, &n); if (st) return st;
  if (!fits_u32((size_t)n)) return EMAT_ERR_CAPACITY;
  std::vector<uint8_t> info((size_t)n);
  st = emat_dphy_tree_info_flatbuffer(tree, names, info.data(), n, &n); if (st) return st;
  FILE* f = std::fopen(path, "wb");
  if (!f) return EMAT_ERR_IO;
  emat_dphy_writer* w = new emat_dphy_writer{f, false};
  w_bytes(w, "DPHY", 4);
  w_u32(w, 3);                                            // save format version
  w_str(w, core_version); w_u32(w, (uint32_t)build_number); w_str(w, commit);
  w_u32(w, 0);                                            // knee index
  w_u32(w, (uint32_t)steps_per_sample);
  w_u32(w, q->alpha_move_enabled ? 1u : 0u); w_u32(w, 0u /* mpox hack: not modelled */); w_u32(w, q->mu_move_enabled ? 1u : 0u);
  { const float mu = (float)q->mu; uint32_t bits; std::memcpy(&bits, &mu, 4); w_u32(w, bits); }
  w_u32(w, (uint32_t)info.size()); w_bytes(w, info.data(), info.size());
  if (w->failed) { std::fclose(f); delete w; return EMAT_ERR_IO; }
  *out = w;
  return EMAT_OK;
}
emat_status emat_dphy_write_state(emat_dphy_writer* w, const emat_flat_tree* tree, const uint8_t* ref, int32_t L, const emat_dphy_params* q) {
  if (!w || !w->f) return EMAT_ERR_INVALID_ARGUMENT;
  if (w->failed) return EMAT_ERR_IO;
  uint64_t nt = 0, np = 0;
  emat_status st = emat_dphy_tree_flatbuffer(tree, ref, L, nullptr, 0, &nt); if (st) return st;
  st = emat_dphy_params_flatbuffer(q, L, nullptr, 0, &np); if (st) return st;
  if (!fits_u32((size_t)nt) || !fits_u32((size_t)np)) return EMAT_ERR_CAPACITY;
  std::vector<uint8_t> bt((size_t)nt), bp((size_t)np);
  st = emat_dphy_tree_flatbuffer(tree, ref, L, bt.data(), nt, &nt); if (st) return st;
  st = emat_dphy_params_flatbuffer(q, L, bp.data(), np, &np); if (st) return st;
  w_u32(w, (uint32_t)bt.size()); w_u32(w, (uint32_t)bp.size());
  w_bytes(w, bt.data(), bt.size()); w_bytes(w, bp.data(), bp.size());
  if (!w->failed && std::fflush(w->f) != 0) w->failed = true;
  return w->failed ? EMAT_ERR_IO : EMAT_OK;
}
emat_status emat_dphy_close(emat_dphy_writer* w) {
  if (!w) return EMAT_OK;
  bool ok = !w->failed;
  if (w->f) {
    const long end = std::ftell(w->f);
    if (end < 0) { w->failed = true; ok = false; }
    w_u32(w, 0);                                          // no more trees
    w_str(w, "{\"confidence\":90,\"topology\":0,\"presentation\":0,\"spacing\":0,\"colorBy\":0,\"burnin\":0,\"metadataPresent\":0,"
             "\"metadataText\":null,\"metadataFile\":null,\"metadataDelimiter\":null,\"selectedMDField\":-1,\"metadataColors\":{}}");
    w_u64(w, (uint64_t)(end < 0 ? 0 : end));
    ok = ok && !w->failed;
    if (std::fclose(w->f) != 0) ok = false;
  }
  delete w;
  return ok ? EMAT_OK : EMAT_ERR_IO;
}

}  // extern "C"
