// emat_gtree_host.hpp -- host side of the HBM-resident whole tree: upload / download, and the two halves
// of a cycle -- cutting the tree into part slabs and gathering the parts back -- as kernels (emat_gtree_kernels.hpp)
// around the few things that stay on the host: slab geometry, size classes, and the coalescent cell tables, which are
// built from a skeleton (topology + times) of every part with the same code as on the host path.
//
// Included at the end of emat_backend.hip.
#ifndef EMAT_GTREE_HOST_HPP_
#define EMAT_GTREE_HOST_HPP_

namespace {

const char* gt_status_text(int32_t s) {
  switch (s) {
    case k_gt_cut_state_overflow: return "the sequence state at a cut point is larger than the device path holds (k_gt_max_cut_intervals / k_gt_max_cut_deltas)";
    case k_gt_pool_overflow: return "cut-state pool overflow";
    case k_gt_list_too_long: return "a node list (the synthetic delta or missation list of a part's sub-root included) exceeds the 16 000 entries the engine accepts per node and list (16-bit list counts)";
    case k_gt_inconsistent: return "inconsistent mutation chain above a cut point, or a part whose node count changed";
    case k_gt_heap_overflow: return "list heap overflow";
    case k_gt_root_deltas_overflow: return "too many changes of the root sequence in one cycle";
    default: return "unknown";
  }
}

emat_status gt_finish_gather(emat_backend* h);
// `uploading`: emat_tree_upload, the one call that may follow a failed gather (it replaces the tree the gather left half written).
emat_status gt_require(emat_backend* h, bool need_resident, bool gather_may_run = false, bool uploading = false) {
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  if (uploading) return EMAT_OK;   // (whatever gather was pending or failed concerns the tree that is being replaced)
  if (need_resident && !h->gt.resident) return fail(h, EMAT_ERR_STATE, "emat_tree_upload first");
  // A deferred gather that failed (emat_tree_reassemble had already returned EMAT_OK with the new links) leaves the device-resident tree
  // with new links but half-written lists: every later emat_tree_* call reports that, with the first failure's text, until a tree is uploaded.
  if (h->gt.gather_failed != EMAT_OK) return fail(h, h->gt.gather_failed, "the device-resident tree is incomplete since a deferred gather failed (" + h->gt.gather_failed_text + "): emat_tree_upload a tree to go on");
  if (h->gt.gather_pending && !gather_may_run) return gt_finish_gather(h);
  return EMAT_OK;
}

// After a reassemble: the packed children, the root and its time come over at once (1.6 MB at 200 000 nodes, into page-locked
// memory); parent, children and times as separate arrays -- 4 MB more, which a cycle of the run driver never looks at -- on demand.
emat_status gt_fetch_mirrors(emat_backend* h) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  const size_t n = (size_t)G.n;
  HIP_TRY(G.d_kids.alloc(n)); HIP_TRY(G.pin_kids.resize(n * sizeof(int2)));
  hipLaunchKernelGGL(k_gt_pack_kids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, G.dev(), G.d_kids.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(G.pin_kids.data(), G.d_kids.p, n * sizeof(int2), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(&G.h_root, G.root.p, 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(&G.h_root_t, G.t.p + G.h_root, 8, hipMemcpyDeviceToHost));
  G.full_mirrors_stale = true; G.d_kids_current = true;
  return EMAT_OK;
}
// the walk records of k_gt_measure, when lists or links were written since they were last made (queued on the engine's stream)
emat_status gt_ensure_climb(emat_backend* h) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  if (G.climb_current && G.climb.n >= (size_t)G.n) return EMAT_OK;
  HIP_TRY(G.climb.alloc((size_t)G.n));
  hipLaunchKernelGGL(k_gt_pack_climb, dim3((unsigned)((G.n + 255) / 256)), dim3(256), 0, h->stream, G.dev());
  HIP_TRY(hipGetLastError());
  G.climb_current = true;
  return EMAT_OK;
}
emat_status gt_full_mirrors(emat_backend* h) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  if (!G.full_mirrors_stale) return EMAT_OK;
  const size_t n = (size_t)G.n;
  G.h_parent.resize(n); G.h_c0.resize(n); G.h_c1.resize(n); G.h_t.resize(n);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(G.h_parent.data(), G.parent.p, n * 4, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(G.h_t.data(), G.t.p, n * 8, hipMemcpyDeviceToHost));
  const int32_t* k = G.kids();
  for (size_t v = 0; v < n; ++v) { G.h_c0[v] = k[2 * v]; G.h_c1[v] = k[2 * v + 1]; }
  G.full_mirrors_stale = false;
  return EMAT_OK;
}

}  // namespace

extern "C" {

emat_status emat_tree_upload(emat_backend* h, const emat_flat_tree* tree) {
  if (!h || !tree) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, false, false, true); if (st) return st;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  const std::string msg = validate_flat_tree(*tree, h->L);
  if (!msg.empty()) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_upload: " + msg);
  { const std::string lim = flat_tree_list_limit(*tree, (int32_t)k_max_list_upload); if (!lim.empty()) return fail(h, EMAT_ERR_CAPACITY, "emat_tree_upload: " + lim); }
  const int n = tree->num_nodes;
  if (tree->mut_offset[tree->root + 1] != tree->mut_offset[tree->root])
    return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_upload: the root node must carry no mutations (fold them into the reference sequence first: Run::normalize_root)");
  const size_t nm = (size_t)tree->mut_offset[n], ni = (size_t)tree->miss_offset[n], nf = (size_t)tree->mfs_offset[n];
  std::vector<GList> lm(n), li(n), lf(n);
  std::vector<MutRec> rm(nm); std::vector<IvRec> ri(ni); std::vector<FsRec> rf(nf);
  for (int i = 0; i < n; ++i) {
    lm[i] = GList{(uint32_t)tree->mut_offset[i], (uint32_t)(tree->mut_offset[i + 1] - tree->mut_offset[i])};
    li[i] = GList{(uint32_t)tree->miss_offset[i], (uint32_t)(tree->miss_offset[i + 1] - tree->miss_offset[i])};
    lf[i] = GList{(uint32_t)tree->mfs_offset[i], (uint32_t)(tree->mfs_offset[i + 1] - tree->mfs_offset[i])};
  }
  for (size_t k = 0; k < nm; ++k) { MutRec r{}; r.t = tree->mut_t[k]; r.site = tree->mut_site[k]; r.from = tree->mut_from[k]; r.to = tree->mut_to[k]; rm[k] = r; }
  for (size_t k = 0; k < ni; ++k) ri[k] = IvRec{tree->miss_start[k], tree->miss_end[k]};
  for (size_t k = 0; k < nf; ++k) { FsRec r{}; r.site = tree->mfs_site[k]; r.state = tree->mfs_state[k]; rf[k] = r; }
  GTreeHost& G = h->gt;
  if (h->stream) HIP_TRY(hipStreamSynchronize(h->stream));
  G.n = n;
  HIP_TRY(G.parent.upload(tree->parent, n)); HIP_TRY(G.c0.upload(tree->child0, n)); HIP_TRY(G.c1.upload(tree->child1, n));
  HIP_TRY(G.t.upload(tree->t, n)); HIP_TRY(G.t_min.upload(tree->t_min, n)); HIP_TRY(G.t_max.upload(tree->t_max, n));
  HIP_TRY(G.root.upload(&tree->root, 1));
  HIP_TRY(G.muts.upload(lm.data(), n)); HIP_TRY(G.miss.upload(li.data(), n)); HIP_TRY(G.mfs.upload(lf.data(), n));
  // the moves create and destroy list records: room for twice the present content plus a record per node
  const bool tight = h->cfg_tree_tight;   // testing aid: no room at all, so that the growth paths run
  HIP_TRY(G.mut_heap.alloc(tight ? nm + 1 : 2 * nm + (size_t)n + 1024)); HIP_TRY(G.iv_heap.alloc(tight ? ni + 1 : 2 * ni + (size_t)n + 1024)); HIP_TRY(G.fs_heap.alloc(tight ? nf + 1 : 2 * nf + (size_t)n + 1024));
  if (nm) HIP_TRY(hipMemcpy(G.mut_heap.p, rm.data(), nm * sizeof(MutRec), hipMemcpyHostToDevice));
  if (ni) HIP_TRY(hipMemcpy(G.iv_heap.p, ri.data(), ni * sizeof(IvRec), hipMemcpyHostToDevice));
  if (nf) HIP_TRY(hipMemcpy(G.fs_heap.p, rf.data(), nf * sizeof(FsRec), hipMemcpyHostToDevice));
  G.used[0] = (uint32_t)nm; G.used[1] = (uint32_t)ni; G.used[2] = (uint32_t)nf;
  HIP_TRY(G.tops.upload(G.used, 3));
  { int32_t z = 0; HIP_TRY(G.status.upload(&z, 1)); }
  G.h_parent.assign(tree->parent, tree->parent + n); G.h_c0.assign(tree->child0, tree->child0 + n); G.h_c1.assign(tree->child1, tree->child1 + n);
  G.h_t.assign(tree->t, tree->t + n); G.h_t_min.assign(tree->t_min, tree->t_min + n); G.h_t_max.assign(tree->t_max, tree->t_max + n);
  G.h_root = tree->root; G.h_root_t = tree->t[tree->root]; G.full_mirrors_stale = false; G.d_kids_current = false; G.gather_pending = false; G.climb_current = false;
  G.gather_failed = EMAT_OK; G.gather_failed_text.clear();
  G.measure_queued = false; G.partition_on_device = false; G.P = 0;   // (a partition made of the tree that was here says nothing about this one: emat_tree_partition first)
  HIP_TRY(G.pin_kids.resize((size_t)n * sizeof(int2)));
  { int32_t* k = (int32_t*)G.pin_kids.data(); for (int v = 0; v < n; ++v) { k[2 * v] = tree->child0[v]; k[2 * v + 1] = tree->child1[v]; } }
  G.resident = true; G.parts_live = false;
  return EMAT_OK;
}

emat_status emat_tree_get_sizes(emat_backend* h, int32_t* nn, int32_t* nm, int32_t* ni, int32_t* nf) {
  if (!h) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  if (h->gt.parts_live) return fail(h, EMAT_ERR_STATE, "the parts are out on their slabs: emat_tree_reassemble first");
  if (nn) *nn = h->gt.n;
  if (nm) *nm = (int32_t)h->gt.used[0];
  if (ni) *ni = (int32_t)h->gt.used[1];
  if (nf) *nf = (int32_t)h->gt.used[2];
  return EMAT_OK;
}

emat_status emat_tree_download(emat_backend* h, emat_flat_tree* out, uint8_t* ref_sequence) {
  if (!h || !out) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  if (G.parts_live) return fail(h, EMAT_ERR_STATE, "the parts are out on their slabs: emat_tree_reassemble first");
  st = gt_full_mirrors(h); if (st) return st;
  const int n = G.n;
  if (out->num_nodes < n || out->cap_muts < (int32_t)G.used[0] || out->cap_intervals < (int32_t)G.used[1] || out->cap_from_states < (int32_t)G.used[2]) return EMAT_ERR_BUFFER_TOO_SMALL;
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<GList> lm(n), li(n), lf(n);
  std::vector<MutRec> rm(G.used[0]); std::vector<IvRec> ri(G.used[1]); std::vector<FsRec> rf(G.used[2]);
  HIP_TRY(hipMemcpy(lm.data(), G.muts.p, (size_t)n * sizeof(GList), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(li.data(), G.miss.p, (size_t)n * sizeof(GList), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(lf.data(), G.mfs.p, (size_t)n * sizeof(GList), hipMemcpyDeviceToHost));
  if (!rm.empty()) HIP_TRY(hipMemcpy(rm.data(), G.mut_heap.p, rm.size() * sizeof(MutRec), hipMemcpyDeviceToHost));
  if (!ri.empty()) HIP_TRY(hipMemcpy(ri.data(), G.iv_heap.p, ri.size() * sizeof(IvRec), hipMemcpyDeviceToHost));
  if (!rf.empty()) HIP_TRY(hipMemcpy(rf.data(), G.fs_heap.p, rf.size() * sizeof(FsRec), hipMemcpyDeviceToHost));
  out->num_nodes = n; out->root = G.h_root;
  std::copy(G.h_parent.begin(), G.h_parent.end(), out->parent); std::copy(G.h_c0.begin(), G.h_c0.end(), out->child0); std::copy(G.h_c1.begin(), G.h_c1.end(), out->child1);
  std::copy(G.h_t.begin(), G.h_t.end(), out->t); std::copy(G.h_t_min.begin(), G.h_t_min.end(), out->t_min); std::copy(G.h_t_max.begin(), G.h_t_max.end(), out->t_max);
  int32_t km = 0, ki = 0, kf = 0;
  out->mut_offset[0] = 0; out->miss_offset[0] = 0; out->mfs_offset[0] = 0;
  for (int i = 0; i < n; ++i) {   // the heaps hold the lists in the order the parts wrote them: back into node order
    if ((size_t)lm[i].off + lm[i].cnt > rm.size() || (size_t)li[i].off + li[i].cnt > ri.size() || (size_t)lf[i].off + lf[i].cnt > rf.size()) return fail(h, EMAT_ERR_INTERNAL, "emat_tree_download: list outside its heap");
    for (uint32_t k = 0; k < lm[i].cnt; ++k, ++km) { const MutRec& r = rm[lm[i].off + k]; out->mut_t[km] = r.t; out->mut_site[km] = r.site; out->mut_from[km] = r.from; out->mut_to[km] = r.to; }
    for (uint32_t k = 0; k < li[i].cnt; ++k, ++ki) { out->miss_start[ki] = ri[li[i].off + k].start; out->miss_end[ki] = ri[li[i].off + k].end; }
    for (uint32_t k = 0; k < lf[i].cnt; ++k, ++kf) { out->mfs_site[kf] = rf[lf[i].off + k].site; out->mfs_state[kf] = rf[lf[i].off + k].state; }
    out->mut_offset[i + 1] = km; out->miss_offset[i + 1] = ki; out->mfs_offset[i + 1] = kf;
  }
  if (ref_sequence) std::copy(h->ref.begin(), h->ref.end(), ref_sequence);
  return EMAT_OK;
}

emat_status emat_tree_get_topology(emat_backend* h, int32_t* parent, int32_t* child0, int32_t* child1, double* t, int32_t* root) {
  if (!h) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  const GTreeHost& G = h->gt;
  if (G.parts_live) return fail(h, EMAT_ERR_STATE, "the parts are out on their slabs: emat_tree_reassemble first");
  st = gt_full_mirrors(h); if (st) return st;
  const int n = G.n, chunk = 32768;   // (4 MB at 200 000 nodes, once per cycle: on the host threads)
  parallel_for((n + chunk - 1) / chunk, [&](int c) {
    const size_t b = (size_t)c * chunk, e = std::min<size_t>((size_t)n, b + chunk);
    if (parent) std::copy(G.h_parent.begin() + b, G.h_parent.begin() + e, parent + b);
    if (child0) std::copy(G.h_c0.begin() + b, G.h_c0.begin() + e, child0 + b);
    if (child1) std::copy(G.h_c1.begin() + b, G.h_c1.begin() + e, child1 + b);
    if (t) std::copy(G.h_t.begin() + b, G.h_t.begin() + e, t + b);
  }, 1);
  if (root) *root = G.h_root;
  return EMAT_OK;
}

emat_status emat_tree_get_kids(emat_backend* h, const int32_t** kids, int32_t* num_nodes, int32_t* root, double* t_root) {
  if (!h || !kids) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true, true); if (st) return st;   // (the mirror is current as soon as emat_tree_reassemble returns; the lists may still be on their way)
  const GTreeHost& G = h->gt;
  if (G.parts_live) return fail(h, EMAT_ERR_STATE, "the parts are out on their slabs: emat_tree_reassemble first");
  *kids = G.kids();
  if (num_nodes) *num_nodes = G.n;
  if (root) *root = G.h_root;
  if (t_root) *t_root = G.h_root_t;
  return EMAT_OK;
}

emat_status emat_tree_partition(emat_backend* h, int32_t num_cuts, const int32_t* cut_nodes, int32_t* num_parts, int32_t* root_part, int32_t* part_sizes) {
  if (!h || num_cuts < 0 || (num_cuts > 0 && !cut_nodes) || !num_parts || !root_part) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  if (G.parts_live) return fail(h, EMAT_ERR_STATE, "the parts are out on their slabs: emat_tree_reassemble first");
  const int n = G.n;
  EMAT_SPAN("tree_partition (all)");
  // partition_tree's rule for the run's root (tree_partitioning.h:196-239): a part of its own at the end unless the stencil names it
  // (Both arrays in ordinary memory.  Staging them in page-locked memory was tried in round 6 (a copy from pageable memory might
  // wait for the stream) and measured: page-locked host memory is read uncached by the CPU, and the validation loop below reads is_cut once per
  // cut node -- 45 -> 680 us for this stretch; the pageable copy never waited for the stream (the runtime stages it).  The copies below are
  // complete before this function returns: it waits for ev_sizes, which is recorded behind them.)
  std::vector<uint8_t> is_cut_v((size_t)n, 0);
  std::vector<int32_t> cut_v(cut_nodes, cut_nodes + num_cuts); cut_v.push_back(EMAT_NO_NODE);   // (room for the root's own part)
  uint8_t* const is_cut = is_cut_v.data(); int32_t* const cut_of_part = cut_v.data();
  int n_cut_parts = num_cuts;
  const int32_t* const kids = G.kids();
  int rp = -1;
  for (int i = 0; i < num_cuts; ++i) {
    const int32_t c = cut_nodes[i];
    if (c < 0 || c >= n || is_cut[c] || kids[2 * (size_t)c] == EMAT_NO_NODE) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_partition: cut nodes must be distinct inner nodes");
    is_cut[c] = 1;
    if (c == G.h_root && rp < 0) rp = i;
  }
  if (rp < 0) { is_cut[G.h_root] = 1; rp = num_cuts; cut_of_part[n_cut_parts++] = G.h_root; }
  const int P = n_cut_parts;
  if (h->cfg.max_parts > 0 && P > h->cfg.max_parts) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "more parts than cfg.max_parts");
  HostLaps laps;
  // Everything below is queued on the engine's stream without the host in between: the counts, the offsets from the counts
  // (k_gt_part_offsets), the arrays, and -- when the model on the device is current -- the measuring pass of the repartition that
  // is bound to follow (k_gt_measure needs nothing the host computes from this partition).  The host only waits for the sizes.
  if (!G.ev_sizes) { HIP_TRY(hipEventCreateWithFlags(&G.ev_sizes, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&G.ev_measure, hipEventDisableTiming)); }
  HIP_TRY(G.d_is_cut.alloc((size_t)n)); HIP_TRY(G.d_cut.alloc((size_t)P)); HIP_TRY(G.d_sizes.alloc((size_t)P)); HIP_TRY(G.d_part_status.alloc(1));
  const size_t total = (size_t)n + (size_t)P - 1;
  HIP_TRY(G.part_off.alloc((size_t)P + 1)); HIP_TRY(G.orig.alloc(total)); HIP_TRY(G.kid0.alloc(total)); HIP_TRY(G.kid1.alloc(total)); HIP_TRY(G.lpar.alloc(total)); HIP_TRY(G.lidx.alloc((size_t)n));
  HIP_TRY(G.pin_sizes.resize(((size_t)2 * P + 2) * sizeof(int32_t)));
  HIP_TRY(hipMemcpyAsync(G.d_is_cut.p, is_cut, (size_t)n, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(G.d_cut.p, cut_of_part, (size_t)P * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemsetAsync(G.d_part_status.p, 0, sizeof(int32_t), h->stream));
  const unsigned blocks = (unsigned)((P + 63) / 64);
  if (!G.d_kids_current) {   // (the first partition after an upload: the packed children every reassemble keeps current from then on)
    HIP_TRY(G.d_kids.alloc((size_t)n));
    hipLaunchKernelGGL(k_gt_pack_kids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, G.dev(), G.d_kids.p);
    HIP_TRY(hipGetLastError());
    G.d_kids_current = true;
  }
  hipLaunchKernelGGL(k_gt_partition, dim3(blocks), dim3(64), 0, h->stream, G.dev(), (const int2*)G.d_kids.p, (const uint8_t*)G.d_is_cut.p, (const int32_t*)G.d_cut.p, P, 0, G.d_sizes.p,
                     (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (int32_t*)nullptr, (const int32_t*)nullptr);
  hipLaunchKernelGGL(k_gt_part_offsets, dim3(1), dim3(1024), 0, h->stream, (const int32_t*)G.d_sizes.p, P, (long long)total, G.part_off.p, G.d_part_status.p);
  HIP_TRY(hipGetLastError());
  int32_t* pin = (int32_t*)G.pin_sizes.data();   // [P] sizes, [P + 1] offsets, status
  HIP_TRY(hipMemcpyAsync(pin, G.d_sizes.p, (size_t)P * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(pin + P, G.part_off.p, ((size_t)P + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(pin + 2 * P + 1, G.d_part_status.p, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipEventRecord(G.ev_sizes, h->stream));
  hipLaunchKernelGGL(k_gt_partition, dim3(blocks), dim3(64), 0, h->stream, G.dev(), (const int2*)G.d_kids.p, (const uint8_t*)G.d_is_cut.p, (const int32_t*)G.d_cut.p, P, 1, (int32_t*)nullptr,
                     (const int32_t*)G.part_off.p, G.orig.p, G.kid0.p, G.kid1.p, G.lpar.p, G.lidx.p, (const int32_t*)G.d_part_status.p);
  HIP_TRY(hipGetLastError());
  G.P = P; G.root_part = rp; G.partition_on_device = true; G.h_orig.clear(); G.h_kid0.clear();
  G.measure_queued = false;
  if (h->have_ref && h->have_evo && !h->model_dirty && h->d_ref.n >= (size_t)h->L) {
    HIP_TRY(G.measure.alloc((size_t)P));
    { const size_t room = h->cfg_tree_tight ? 1 : (size_t)64 * P + 4096;
      HIP_TRY(G.pool_muts.alloc(std::max<size_t>(G.pool_muts.n, room))); HIP_TRY(G.pool_ivs.alloc(std::max<size_t>(G.pool_ivs.n, room))); }
    HIP_TRY(G.pool_tops.alloc(2 * k_gt_pool_lanes * k_gt_pool_stride));
    HIP_TRY(G.pin_measure.resize((size_t)P * sizeof(GMeasure)));
    HIP_TRY(hipMemsetAsync(G.pool_tops.p, 0, 2 * k_gt_pool_lanes * k_gt_pool_stride * sizeof(uint32_t), h->stream));
    { emat_status cs = gt_ensure_climb(h); if (cs) return cs; }
    hipLaunchKernelGGL((k_gt_measure<k_gt_small_cut_intervals, k_gt_small_cut_deltas>), dim3((unsigned)P), dim3(k_wave), 0, h->stream, G.dev(), G.partition(), G.pools(), (const uint8_t*)h->d_ref.p, G.measure.p,
                       (const int32_t*)nullptr, (const int32_t*)G.d_part_status.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(G.pin_measure.data(), G.measure.p, (size_t)P * sizeof(GMeasure), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(G.ev_measure, h->stream));
    G.measure_queued = true;
  }
  laps.mark("tree_partition: 1 validation, uploads, launches");
  HIP_TRY(hipEventSynchronize(G.ev_sizes));
  laps.mark("tree_partition: 2 wait for sizes + offsets");
  const int32_t pst = pin[2 * P + 1];
  if (pst != 0) {
    G.P = 0; G.partition_on_device = false; G.measure_queued = false;
    HIP_TRY(hipStreamSynchronize(h->stream));   // (what was queued behind the counts saw the status too and did nothing)
    if (pst == 3) return fail(h, EMAT_ERR_INTERNAL, "emat_tree_partition: empty part");
    return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_partition: the cut nodes do not partition the tree (a cut below another part's tip?)");
  }
  G.h_part_off.assign(pin + P, pin + 2 * P + 1);
  *num_parts = P; *root_part = rp;
  if (part_sizes) std::copy(pin, pin + P, part_sizes);
  return EMAT_OK;
}

namespace { emat_status gt_partition_to_host(emat_backend* h) {   // the arrays of a partition made on the device, when the host wants them after all
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  if (!G.h_orig.empty() || G.P == 0) return EMAT_OK;
  const size_t total = (size_t)G.h_part_off[G.P];
  G.h_orig.resize(total); G.h_kid0.resize(total); G.h_kid1.resize(total);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(G.h_orig.data(), G.orig.p, total * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(G.h_kid0.data(), G.kid0.p, total * sizeof(int32_t), hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(G.h_kid1.data(), G.kid1.p, total * sizeof(int32_t), hipMemcpyDeviceToHost));
  return EMAT_OK;
} }

emat_status emat_tree_get_partition(emat_backend* h, int32_t* part_offset, int32_t* orig, int32_t* kid0, int32_t* kid1) {
  if (!h) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  GTreeHost& G = h->gt;
  if (G.P == 0) return fail(h, EMAT_ERR_STATE, "no partition yet");
  st = gt_partition_to_host(h); if (st) return st;
  if (part_offset) std::copy(G.h_part_off.begin(), G.h_part_off.end(), part_offset);
  if (orig) std::copy(G.h_orig.begin(), G.h_orig.end(), orig);
  if (kid0) std::copy(G.h_kid0.begin(), G.h_kid0.end(), kid0);
  if (kid1) std::copy(G.h_kid1.begin(), G.h_kid1.end(), kid1);
  return EMAT_OK;
}

emat_status emat_tree_repartition(emat_backend* h, int32_t num_parts, const int32_t* part_offset, const int32_t* orig, const int32_t* kid0, const int32_t* kid1,
                                  int32_t root_part, const uint64_t* seeds, const emat_pop_model* pm, double t_step) {
  return emat_tree_repartition_range(h, num_parts, part_offset, orig, kid0, kid1, root_part, seeds, pm, t_step, 0, num_parts);
}
emat_status emat_tree_repartition_range(emat_backend* h, int32_t num_parts, const int32_t* part_offset, const int32_t* orig, const int32_t* kid0, const int32_t* kid1,
                                        int32_t root_part, const uint64_t* seeds, const emat_pop_model* pm, double t_step, int32_t part_lo, int32_t part_hi) {
  const bool made_here = part_offset == nullptr && orig == nullptr && kid0 == nullptr && kid1 == nullptr;   // the partition of emat_tree_partition, already on the device
  if (!h || num_parts <= 0 || (!made_here && (!part_offset || !orig || !kid0 || !kid1)) || !seeds || !pm || !(t_step > 0.0) || root_part < 0 || root_part >= num_parts) return EMAT_ERR_INVALID_ARGUMENT;
  if (part_lo < 0 || part_hi > num_parts || part_lo >= part_hi) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  if (h->cfg.max_parts > 0 && num_parts > h->cfg.max_parts) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "more parts than cfg.max_parts");
  if (!h->have_ref || !h->have_evo) return fail(h, EMAT_ERR_STATE, "set_ref_sequence and set_evo must precede emat_tree_repartition");
  GTreeHost& G = h->gt;
  const int P = num_parts, n = G.n, lo = part_lo, hi = part_hi, nloc = hi - lo;   // this process runs the parts [lo, hi): local part q is part lo + q
  const bool verbose = verbose_reports();
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  if (made_here) {
    if (!G.partition_on_device || G.P != P || G.root_part != root_part) return fail(h, EMAT_ERR_STATE, "emat_tree_partition first (with no arrays given, its partition is the one that is cut)");
    part_offset = G.h_part_off.data();
  }
  // every node is a non-root node of exactly one part; the run's root is the root of the root part
  if (part_offset[0] != 0 || (int64_t)part_offset[P] != (int64_t)n + P - 1) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_repartition: the parts do not cover the tree");
  const size_t total = (size_t)part_offset[P];
  if (!made_here || h->cfg_gt_host_coal) { st = gt_full_mirrors(h); if (st) return st; }
  if (!made_here) {
    std::atomic<int> bad{0};
    parallel_for(P, [&](int p) {
      const int b = part_offset[p], np = part_offset[p + 1] - b;
      if (np < 1) { bad.store(1); return; }
      for (int s = 0; s < np; ++s) {
        const int32_t o = orig[b + s], k0 = kid0[b + s], k1 = kid1[b + s];
        if (o < 0 || o >= n || (k0 == EMAT_NO_NODE) != (k1 == EMAT_NO_NODE)) { bad.store(1); return; }
        if (k0 != EMAT_NO_NODE && (k0 <= 0 || k0 >= np || k1 <= 0 || k1 >= np || G.h_c0[o] == EMAT_NO_NODE)) { bad.store(1); return; }
      }
    }, 64);
    if (bad.load() || orig[part_offset[root_part]] != G.h_root) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_repartition: malformed partition");
    // every node is owned exactly once: as a non-root node of its part, the run's root as the root of the root part
    std::vector<uint8_t> owned((size_t)n, 0);
    for (int p = 0; p < P && !bad.load(); ++p)
      for (int s = (p == root_part ? 0 : 1), b = part_offset[p], np = part_offset[p + 1] - b; s < np; ++s) { if (owned[orig[b + s]]++) { bad.store(1); break; } }
    if (bad.load()) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_repartition: a node belongs to two parts");
  }
  EMAT_SPAN("tree_repartition (all)");
  {   // the population model is handed over with every repartition and is the same from cycle to cycle: only a changed one travels
    HostPopModel np;
    try { np = HostPopModel::from_c(*pm); } catch (const std::exception& ex) { return fail(h, EMAT_ERR_INVALID_ARGUMENT, ex.what()); }
    const bool same = h->have_pop && np.kind == h->pop.kind && np.skygrid_type == h->pop.skygrid_type && std::memcmp(np.p, h->pop.p, sizeof(np.p)) == 0 &&
                      std::memcmp(&np.t_c, &h->pop.t_c, sizeof(double)) == 0 && np.x == h->pop.x && np.gamma == h->pop.gamma;
    if (!same) { h->pop = std::move(np); h->model_dirty = true; }
    h->have_pop = true;
  }
  const bool measure_queued = made_here && G.measure_queued && !h->model_dirty;   // (queued by emat_tree_partition: the stream is busy with it, and nothing below needs it idle)
  G.measure_queued = false;
  if (!measure_queued && h->stream) HIP_TRY(hipStreamSynchronize(h->stream));
  { EMAT_SPAN("tree_repartition: sync_model_to_device"); st = sync_model_to_device(h); if (st) return st; }
  HostLaps laps;
  G.P = P; G.root_part = root_part; G.parts_live = false;
  const bool device_coal = !h->cfg_gt_host_coal;
  if (nloc != P && !device_coal) return fail(h, EMAT_ERR_STATE, "a block of the parts needs the coalescent tables built on the device (EMAT_TREE_HOST_COALESCENT builds them from all parts on the host)");
  G.lo = lo; G.hi = hi;
  if (!made_here) { G.h_part_off.assign(part_offset, part_offset + P + 1); G.h_orig.assign(orig, orig + total); G.h_kid0.assign(kid0, kid0 + total); G.h_kid1.assign(kid1, kid1 + total); G.partition_on_device = false; part_offset = G.h_part_off.data(); }
  if (!made_here) { HIP_TRY(G.part_off.upload(part_offset, (size_t)P + 1)); HIP_TRY(G.orig.upload(orig, total)); HIP_TRY(G.kid0.upload(kid0, total)); HIP_TRY(G.kid1.upload(kid1, total)); }
  if (made_here && !device_coal) { st = gt_partition_to_host(h); if (st) return st; orig = G.h_orig.data(); kid0 = G.h_kid0.data(); kid1 = G.h_kid1.data(); }   // the host builder wants skeletons
  if (device_coal && !made_here) {
    std::vector<int32_t> lpar(total, EMAT_NO_NODE);
    parallel_for(P, [&](int p) {
      const int b = part_offset[p], np = part_offset[p + 1] - b;
      for (int s = 0; s < np; ++s) if (kid0[b + s] != EMAT_NO_NODE) { lpar[b + kid0[b + s]] = s; lpar[b + kid1[b + s]] = s; }
    }, 64);
    HIP_TRY(G.lpar.upload(lpar.data(), total));
  }
  if (!measure_queued) {
    HIP_TRY(G.measure.alloc(P));
    { const size_t room = h->cfg_tree_tight ? 1 : (size_t)64 * P + 4096;
      HIP_TRY(G.pool_muts.alloc(std::max<size_t>(G.pool_muts.n, room))); HIP_TRY(G.pool_ivs.alloc(std::max<size_t>(G.pool_ivs.n, room))); }
    HIP_TRY(G.pool_tops.alloc(2 * k_gt_pool_lanes * k_gt_pool_stride));
  }
  const auto t1 = now(); laps.mark("tree_repartition: 01 allocations");
  auto launch_measure = [&]() -> emat_status {
    { emat_status cs = gt_ensure_climb(h); if (cs) return cs; }
    HIP_TRY(hipMemsetAsync(G.pool_tops.p, 0, 2 * k_gt_pool_lanes * k_gt_pool_stride * sizeof(uint32_t), h->stream));
    hipLaunchKernelGGL((k_gt_measure<k_gt_small_cut_intervals, k_gt_small_cut_deltas>), dim3((unsigned)P), dim3(k_wave), 0, h->stream, G.dev(), G.partition(), G.pools(), (const uint8_t*)h->d_ref.p, G.measure.p, (const int32_t*)nullptr, (const int32_t*)nullptr);
    HIP_TRY(hipGetLastError());
    return EMAT_OK;
  };
  // the parts whose cut-point state did not fit the small kernel's LDS: once more, with the full capacities
  auto remeasure_large = [&](const std::vector<int32_t>& list) -> emat_status {
    HIP_TRY(G.measure_list.upload(list.data(), list.size()));
    hipLaunchKernelGGL((k_gt_measure<k_gt_max_cut_intervals, k_gt_max_cut_deltas>), dim3((unsigned)list.size()), dim3(k_wave), 0, h->stream, G.dev(), G.partition(), G.pools(), (const uint8_t*)h->d_ref.p, G.measure.p, (const int32_t*)G.measure_list.p, (const int32_t*)nullptr);
    HIP_TRY(hipGetLastError());
    return EMAT_OK;
  };
  st = join_side_classes(h); if (st) return st;   // (side launches of a pass nobody gathered: the slabs are about to be rebuilt)
  h->sides_must_fork = true;
  if (!measure_queued) { st = launch_measure(); if (st) return st; }
  laps.mark("tree_repartition: 02 join sides + launch k_gt_measure");
  // The records of the parts.  With the coalescent tables built on the host they come first (the builder reads the parts' skeletons:
  // topology + times) and fill the time the device spends measuring; with the tables built on the device nothing on the host needs
  // them before the slabs are placed, and they are written in ONE pass over the records at the end, together with everything the
  // measures decide (a record is 600 bytes, thirteen thousand of them are 8 MB: three passes cost three times the cache misses).
  h->coal_builder.reset();
  h->fatal_status = EMAT_OK; h->fatal_message.clear(); h->pass_pending = false;
  if (!device_coal) h->parts.clear();   // (with the tables built on the device the records hold no vectors worth freeing and re-allocating 8 000 times per cycle)
  h->parts.resize(nloc);
  h->expected_moves.assign((size_t)nloc, 0);
  h->uploads_expected = 0; h->root_part = (root_part >= lo && root_part < hi) ? root_part - lo : -1;
  h->slabs_on_device = false; h->host_slabs_current = false; h->headers_current = false; h->have_coal = false; h->derived_valid = false;
  auto init_record = [&](int q) {
    const int p = lo + q;
    PartHost& ph = h->parts[q];
    const int b = part_offset[p], np = part_offset[p + 1] - b;
    FlatTree& t = ph.tree;
    if (!device_coal) t.resize_nodes(np);   // (with the tables built on the device the host never needs the part's tree: a pull decodes it from the slab)
    else if (t.num_nodes() != 0) t = FlatTree{};   // left over from a pull of the previous partition
    t.root = 0;
    for (int s = 0; s < np && !device_coal; ++s) {
      const int32_t o = orig[b + s], k0 = kid0[b + s], k1 = kid1[b + s];
      t.child0[s] = k0; t.child1[s] = k1;
      if (k0 != EMAT_NO_NODE) { t.parent[k0] = s; t.parent[k1] = s; }
      t.t[s] = G.h_t[o];
      if (k0 == EMAT_NO_NODE && G.h_c0[o] != EMAT_NO_NODE) { t.t_min[s] = (float)G.h_t[o]; t.t_max[s] = (float)G.h_t[o]; }
      else { t.t_min[s] = G.h_t_min[o]; t.t_max[s] = G.h_t_max[o]; }
    }
    ph.includes_run_root = p == root_part; ph.n_nodes = np;
    ph.rng.key = seeds[p]; ph.rng.counter = 0; ph.rng.spare = 0; ph.rng.has_spare = false;
    ph.uploaded = true; ph.stats = emat_part_stats{}; ph.space_boost = 1.0; ph.cell_boost = 1; ph.trace.clear();
  };
  if (!device_coal) parallel_for(nloc, init_record, 64);
  const auto t2 = now(); laps.mark("tree_repartition: 03 part records (host coalescent only)");
  if (!device_coal) try {
    std::vector<const FlatTree*> trees; std::vector<HostRng*> rngs;
    for (auto& ph : h->parts) { trees.push_back(&ph.tree); rngs.push_back(&ph.rng); }
    auto cps = make_coalescent_parts(trees, root_part, h->pop, rngs, t_step);
    for (int p = 0; p < P; ++p) h->parts[p].coal = std::move(cps[p]);   // (host tables: lo = 0, hi = P)
  } catch (const std::exception& ex) { return fail(h, EMAT_ERR_INVALID_ARGUMENT, ex.what()); }
  h->have_coal = true;
  const auto t3 = now();
  std::vector<GMeasure> me(P);
  bool large_done = false;
  for (int attempt = 0;; ++attempt) {
    if (attempt == 0 && measure_queued) {
      HIP_TRY(hipEventSynchronize(G.ev_measure));
      laps.mark("tree_repartition: 04 wait for k_gt_measure");
      std::memcpy(me.data(), G.pin_measure.data(), (size_t)P * sizeof(GMeasure));
    } else {
      HIP_TRY(hipStreamSynchronize(h->stream));
      laps.mark("tree_repartition: 04 wait for k_gt_measure");
      HIP_TRY(hipMemcpy(me.data(), G.measure.p, (size_t)P * sizeof(GMeasure), hipMemcpyDeviceToHost));
    }
    laps.mark("tree_repartition: 05 measures D2H");
    {
      std::vector<int32_t> large;
      for (int p = 0; p < P; ++p) if (me[p].status == k_gt_cut_state_overflow) large.push_back(p);
      if (!large.empty() && !large_done) {
        large_done = true; G.large_measures += (int32_t)large.size();
        st = remeasure_large(large); if (st) return st;
        HIP_TRY(hipStreamSynchronize(h->stream));
        HIP_TRY(hipMemcpy(me.data(), G.measure.p, (size_t)P * sizeof(GMeasure), hipMemcpyDeviceToHost));
      }
    }
    int32_t worst = k_gt_ok; int who = -1;
    for (int p = 0; p < P; ++p) if (me[p].status != k_gt_ok && (worst == k_gt_ok || me[p].status != k_gt_pool_overflow)) { worst = me[p].status; who = p; if (worst != k_gt_pool_overflow) break; }
    if (worst == k_gt_ok) break;
    if (worst != k_gt_pool_overflow || attempt == 4) return fail(h, worst == k_gt_cut_state_overflow || worst == k_gt_list_too_long ? EMAT_ERR_CAPACITY : EMAT_ERR_INTERNAL,
                                                                  "emat_tree_repartition: part " + std::to_string(who) + ": " + gt_status_text(worst));
    uint32_t tops[2] = {0u, 0u};
    {   // the atomics kept counting: the fullest region of either pool says what every region needs
      std::vector<uint32_t> all(2 * k_gt_pool_lanes * k_gt_pool_stride);
      HIP_TRY(hipMemcpy(all.data(), G.pool_tops.p, all.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
      for (uint32_t r = 0; r < k_gt_pool_lanes; ++r) { tops[0] = std::max(tops[0], all[r * k_gt_pool_stride] * k_gt_pool_lanes); tops[1] = std::max(tops[1], all[(k_gt_pool_lanes + r) * k_gt_pool_stride] * k_gt_pool_lanes); }
    }
    HIP_TRY(G.pool_muts.alloc((size_t)tops[0] * 2 + 4096)); HIP_TRY(G.pool_ivs.alloc((size_t)tops[1] * 2 + 4096));
    ++G.pool_regrows; large_done = false;
    st = launch_measure(); if (st) return st;
  }
  const auto t4 = now(); laps.mark("tree_repartition: 06 status scan of the measures");
  // geometry, placement, size classes: as for host-encoded parts
  const int trace_cap = h->cfg.trace_moves > 0 ? h->cfg.trace_moves : 0;
  uint64_t off = 0; h->max_slab_bytes = 0; h->persistent_bytes.assign(nloc, 0); h->prefix_bytes.assign(nloc, 0);
  std::vector<GPartDesc> desc(P); std::vector<uint64_t> offs(nloc);
  uint64_t cells_bytes = 0;
  GCoal co{};
  std::vector<int> first_active(P, 0), num_cells_of(P, 0);
  if (device_coal) {
    // CoalBuilder::local_range / set_range / local_grid's cell ranges from the parts' time ranges (plain arithmetic: the
    // same numbers as on the host); the tables themselves are filled by k_gt_coal_* below
    using namespace coal_detail;
    double all_min = std::numeric_limits<double>::max(), all_max = -std::numeric_limits<double>::max();
    for (int p = 0; p < P; ++p) { all_min = std::min(all_min, me[p].t_min); all_max = std::max(all_max, me[p].t_max); }
    co.t_ref = all_max; co.t_step = t_step; co.num_cells = cell_for(all_min, all_max, t_step) + 1;
    uint64_t pool = 0;
    for (int p = 0; p < P; ++p) {   // the grid is the whole run's: every process builds all of it (it has the whole tree), identically
      const int fc = cell_for(me[p].t_max, co.t_ref, t_step), lc = cell_for(p == root_part ? all_min : me[p].t_min, co.t_ref, t_step);
      if (!(0 <= fc && fc <= lc && lc < co.num_cells)) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "coalescent grid: bad cell range");
      const int wf = std::min(fc, std::max(0, cell_for(me[p].t_max_exact, co.t_ref, t_step)));
      first_active[p] = fc;
      num_cells_of[p] = lc - wf + 1;
      GPartDesc& d = desc[p];
      d.cell_first = wf; d.n_cells = lc - wf + 1; d.n_cells_total = lc + 1; d.t_ref = co.t_ref; d.t_step = t_step; d.coal_first_active = fc;
      d.rng_key = seeds[p]; d.cells_off = pool; pool += (uint64_t)(lc - wf + 1);
    }
    HIP_TRY(G.co_kbar.alloc(pool)); HIP_TRY(G.co_ktw.alloc(pool));
    HIP_TRY(G.co_k_bar.alloc(co.num_cells)); HIP_TRY(G.co_k_tw.alloc(co.num_cells)); HIP_TRY(G.co_popsize.alloc(co.num_cells)); HIP_TRY(G.co_num_active.alloc(co.num_cells)); HIP_TRY(G.co_tsop.alloc(co.num_cells));
    co.kbar_pool = G.co_kbar.p; co.ktw_pool = G.co_ktw.p; co.k_bar = G.co_k_bar.p; co.k_tw = G.co_k_tw.p; co.popsize = G.co_popsize.p; co.num_active = G.co_num_active.p; co.ts_over_pop = G.co_tsop.p;
    co.status = G.status.p;
    // the grid's four kernels need nothing but the cell ranges: they run while the host places the slabs (the descriptors travel
    // twice -- with the cell ranges now, complete before k_gt_build)
    HIP_TRY(G.desc.upload(desc.data(), (size_t)P));
    HIP_TRY(hipMemsetAsync(G.status.p, 0, sizeof(int32_t), h->stream));
    const unsigned cell_blocks = (unsigned)co.num_cells;
    hipLaunchKernelGGL(k_gt_coal_kbar, dim3((unsigned)P), dim3(k_wave), 0, h->stream, G.dev(), G.partition(), (const GPartDesc*)G.desc.p, co);
    hipLaunchKernelGGL(k_gt_coal_grid, dim3(cell_blocks), dim3(k_wave), 0, h->stream, P, (const GPartDesc*)G.desc.p, co, (const PopTable*)h->d_pop.p);
    hipLaunchKernelGGL(k_gt_coal_draw, dim3((unsigned)P), dim3(k_wave), 0, h->stream, (const GPartDesc*)G.desc.p, co);
    hipLaunchKernelGGL(k_gt_coal_ktw, dim3(cell_blocks), dim3(k_wave), 0, h->stream, P, (const GPartDesc*)G.desc.p, co);
    HIP_TRY(hipGetLastError());
  }
  laps.mark("tree_repartition: 07 cell ranges + pool allocations + coalescent launches");
  // where every slab goes: from the measures alone (a fresh record is the root part or not, and has no boosts)
  std::vector<SlabGeo> geo((size_t)nloc);
  if (h->used_bytes.size() < (size_t)nloc) h->used_bytes.resize((size_t)nloc, 0u);
  for (int q = 0; q < nloc; ++q) {
    const int p = lo + q;
    const int nc = device_coal ? num_cells_of[p] : (int)h->parts[q].coal.k_bar_p.size();
    const SlabGeo g = slab_geometry(h, me[p].n_nodes, me[p].num_muts, me[p].content_bytes, nc, p == root_part, 1.0);
    geo[(size_t)q] = g; offs[(size_t)q] = off; off += g.bytes;
    h->used_bytes[(size_t)q] = g.bytes - g.scratch - g.heap + me[p].content_bytes;
    h->persistent_bytes[(size_t)q] = g.bytes - g.scratch;
    h->prefix_bytes[(size_t)q] = g.bytes - g.scratch - g.heap;
    h->max_slab_bytes = std::max(h->max_slab_bytes, g.bytes);
    if (!device_coal) { desc[p].cells_off = cells_bytes; cells_bytes += (uint64_t)nc * 32u + (((uint64_t)nc * 4u + 7u) & ~(uint64_t)7u); }
  }
  laps.mark("tree_repartition: 08a slab geometry + placement");
  // the one pass over the records, and the descriptors the build kernel reads
  parallel_for(nloc, [&](int q) {
    const int p = lo + q;
    if (device_coal) init_record(q);
    PartHost& ph = h->parts[q];
    const SlabGeo& g = geo[(size_t)q];
    ph.slab_off = offs[(size_t)q]; ph.slab_bytes = g.bytes; ph.scratch_bytes = g.scratch;
    GPartDesc d = desc[p];
    if (device_coal) {
      HostCoalPart& c = ph.coal;
      if (!c.k_bar_p.empty()) c = HostCoalPart{};   // (tables decoded by a pull of the previous partition)
      c.cell_first = d.cell_first; c.n_cells_total = d.n_cells_total; c.t_ref = d.t_ref; c.t_step = d.t_step;
      ph.rng.counter = (uint64_t)(d.n_cells_total - first_active[p]);   // one Philox block per Gaussian draw: cells first_active .. last
    }
    const int nc = device_coal ? num_cells_of[p] : (int)ph.coal.k_bar_p.size();
    d.slab_bytes = g.bytes; d.heap_bytes = g.heap; d.scratch_bytes = g.scratch; d.cell_cap = g.cell_cap; d.trace_cap = trace_cap;
    d.flags = ph.includes_run_root ? k_flag_includes_run_root : 0u;
    d.rng_key = ph.rng.key; d.rng_counter = ph.rng.counter; d.rng_spare = ph.rng.spare; d.rng_has_spare = ph.rng.has_spare ? 1u : 0u;
    d.cell_first = ph.coal.cell_first; d.n_cells = nc; d.n_cells_total = ph.coal.n_cells_total; d.t_ref = ph.coal.t_ref; d.t_step = ph.coal.t_step;
    d.coal_first_active = device_coal ? first_active[p] : ph.coal.cell_first;
    desc[p] = d;
  }, 256);
  const auto t5 = now(); laps.mark("tree_repartition: 08b records + descriptors (one pass)");
  assign_size_classes(h);
  h->order_valid = false;
  const auto t6 = now(); laps.mark("tree_repartition: 09 size classes");
  std::vector<uint8_t> cells(cells_bytes);
  if (!device_coal) parallel_for(P, [&](int p) {
    const HostCoalPart& c = h->parts[p].coal; const size_t nc = c.k_bar_p.size();
    double* w = (double*)(cells.data() + desc[p].cells_off);
    std::copy(c.k_bar_p.begin(), c.k_bar_p.end(), w); std::copy(c.k_twiddle_bar_p.begin(), c.k_twiddle_bar_p.end(), w + nc);
    std::copy(c.k_twiddle_bar.begin(), c.k_twiddle_bar.end(), w + 2 * nc); std::copy(c.popsize_bar.begin(), c.popsize_bar.end(), w + 3 * nc);
    std::copy(c.num_active_parts.begin(), c.num_active_parts.end(), (int32_t*)(w + 4 * nc));
  }, 64);
  const auto t7 = now();
  HIP_TRY(G.desc.upload(desc.data(), (size_t)P)); HIP_TRY(G.cells.upload(cells.data(), cells.size()));
  HIP_TRY(h->d_slab_off.upload(offs.data(), offs.size()));
  laps.mark("tree_repartition: 10 descriptors + offsets H2D");
  HIP_TRY(h->d_slabs.alloc_roomy(off)); h->slab_bytes_total = off;
  HIP_TRY(h->d_part_ticks.alloc((2 + 2 * k_ticket_log) * (size_t)nloc)); HIP_TRY(hipMemsetAsync(h->d_part_ticks.p, 0, (2 + 2 * k_ticket_log) * (size_t)nloc * sizeof(int64_t), h->stream));
  HIP_TRY(h->d_part_status.alloc((size_t)nloc)); HIP_TRY(hipMemsetAsync(h->d_part_status.p, 0, (size_t)nloc * sizeof(int32_t), h->stream));
  laps.mark("tree_repartition: 11 slab allocation + memsets");
  hipLaunchKernelGGL(k_gt_build, dim3((unsigned)nloc), dim3(k_wave), 0, h->stream, G.dev(), G.partition(), G.pools(), (const GMeasure*)G.measure.p, (const GPartDesc*)G.desc.p,
                     (const uint8_t*)G.cells.p, co, h->d_slabs.p, (const uint64_t*)h->d_slab_off.p, lo);
  HIP_TRY(hipGetLastError());
  laps.mark("tree_repartition: 12 launch of k_gt_build");
  // the launch order of the pass to come, while the device builds the slabs (nothing in flight reads it: the side launches of the
  // last pass were joined above, and the copy is queued behind the kernels)
  st = build_order(h, true); if (st) return st;
  laps.mark("tree_repartition: 12b launch order (sort + queued H2D)");
  if (device_coal) {   // a lineage outside its part's window, or a grid whose last cell no part is active in: the host builder throws on both
    int32_t cst = 0;
    HIP_TRY(hipStreamSynchronize(h->stream));
    laps.mark("tree_repartition: 13 wait for the kernels");
    HIP_TRY(hipMemcpy(&cst, G.status.p, sizeof(cst), hipMemcpyDeviceToHost));
    if (cst != k_gt_ok) { h->slabs_on_device = false; h->parts.clear(); return fail(h, EMAT_ERR_INVALID_ARGUMENT, "coalescent grid: a lineage outside its part's cells, or an inactive final cell"); }
    // the run-wide cell arrays ARE the grid the kernels just built: the moves read them in place; the host's mirror of them (a few KB,
    // read when slabs are decoded) follows when a pull asks for it (pull_grid_mirrors)
    h->shared_dev = SharedCells{G.co_k_tw.p, G.co_tsop.p, G.co_num_active.p, co.num_cells};
    h->grid_mirrors_on_device = true;
    h->sh_ktw.clear(); h->sh_popsize.clear(); h->sh_tsop.clear(); h->sh_nact.clear();
  } else { emat_status st2 = upload_shared_cells(h); if (st2) return st2; h->grid_mirrors_on_device = false; }
  laps.mark("tree_repartition: 14 status of the grid");
  h->slabs_on_device = true; h->host_slabs_current = false; h->headers_current = false; h->derived_valid = false;
  G.parts_live = true;
  if (verbose) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    fprintf(stderr, "[emat] tree_repartition: checks + uploads %.1f ms | part records %.1f ms | host coalescent tables %.1f ms | wait for k_gt_measure %.1f ms | geometry %.1f ms | size classes %.1f ms | "
                    "packing %.1f ms | uploads + kernels %.1f ms\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), ms(t4, t5), ms(t5, t6), ms(t6, t7), ms(t7, now()));
  }
  return EMAT_OK;
}

// ---- gathering the parts back: in one go (emat_tree_reassemble), or in the steps between which several processes exchange
//      what each of them ran (emat_tree_get_root_deltas, emat_tree_gather_local, emat_tree_export_nodes / _apply_nodes, _end) ------
}  // extern "C"
namespace {

// the heaps must keep what they hold (the nodes gathered so far) when they grow
template <class T> emat_status gt_grow_keeping(emat_backend* h, DevBuf<T>& buf, size_t used, size_t want) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  if (want <= buf.n) return EMAT_OK;
  DevBuf<T> bigger;
  HIP_TRY(bigger.alloc(want));
  if (used) HIP_TRY(hipMemcpy(bigger.p, buf.p, used * sizeof(T), hipMemcpyDeviceToDevice));
  std::swap(buf.p, bigger.p); std::swap(buf.n, bigger.n);
  return EMAT_OK;
}

emat_status gt_root_deltas(emat_backend* h, std::vector<GRootDelta>& rd, bool& owner) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  owner = G.root_part >= G.lo && G.root_part < G.hi;
  rd.clear();
  if (!owner) return EMAT_OK;
  HIP_TRY(G.root_deltas.alloc(k_gt_max_root_deltas)); HIP_TRY(G.n_root_deltas.alloc(1));
  int32_t nd = 0;
  for (int attempt = 0; attempt < 2; ++attempt) {   // nearly always a handful; a cycle that changed more root sites than there is room gets the room and is read again
    HIP_TRY(hipMemsetAsync(G.n_root_deltas.p, 0, sizeof(int32_t), h->stream));
    hipLaunchKernelGGL(k_gt_root_deltas, dim3(1), dim3(k_wave), 0, h->stream, (const uint8_t*)h->d_slabs.p, (const uint64_t*)h->d_slab_off.p, G.root_part - G.lo, G.root_deltas.p, (int)G.root_deltas.n, G.n_root_deltas.p);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(&nd, G.n_root_deltas.p, sizeof(nd), hipMemcpyDeviceToHost));
    if ((size_t)nd <= G.root_deltas.n) break;
    HIP_TRY(G.root_deltas.alloc((size_t)nd));
  }
  rd.resize((size_t)nd);
  if (nd > 0) HIP_TRY(hipMemcpy(rd.data(), G.root_deltas.p, (size_t)nd * sizeof(GRootDelta), hipMemcpyDeviceToHost));
  return EMAT_OK;
}

// every local part writes the nodes it owns into this process's copy of the tree; the heaps are rebuilt from zero
// (in two halves: the launch, and the wait + check, which a single process postpones until somebody needs the tree: gt_finish_gather)
emat_status gt_launch_gather(emat_backend* h) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  G.climb_current = false;   // (lists and links are about to be rewritten)
  HIP_TRY(hipMemsetAsync(G.tops.p, 0, 3 * sizeof(uint32_t), h->stream));
  HIP_TRY(hipMemsetAsync(G.status.p, 0, sizeof(int32_t), h->stream));
  HIP_TRY(hipMemsetAsync(G.n_root_deltas.p, 0, sizeof(int32_t), h->stream));
  hipLaunchKernelGGL(k_gt_gather, dim3((unsigned)(G.hi - G.lo)), dim3(k_wave), 0, h->stream, G.dev(), G.partition(), (const uint8_t*)h->d_slabs.p, (const uint64_t*)h->d_slab_off.p,
                     h->d_ref.p, G.root_deltas.p, G.n_root_deltas.p, G.status.p, G.lo, (const GRootDelta*)G.root_deltas_in.p, (int)G.gather_rd.size());
  HIP_TRY(hipGetLastError());
  return EMAT_OK;
}
emat_status gt_finish_gather_once(emat_backend* h);
// The gather stays pending until it has SUCCEEDED; a failure is kept (gt_require) instead of being reported once from whichever call came next.
emat_status gt_finish_gather(emat_backend* h) {
  GTreeHost& G = h->gt;
  if (!G.gather_pending) return EMAT_OK;
  const emat_status st = gt_finish_gather_once(h);
  if (st == EMAT_OK) { G.gather_pending = false; return EMAT_OK; }
  G.gather_failed = st; G.gather_failed_text = h->last_error;
  return st;
}
emat_status gt_finish_gather_once(emat_backend* h) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  int32_t status = 0;
  for (int attempt = 0;; ++attempt) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(&status, G.status.p, sizeof(status), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(G.used, G.tops.p, sizeof(G.used), hipMemcpyDeviceToHost));
    if (h->cfg_debug_fail_gather && status == k_gt_ok) { h->cfg_debug_fail_gather = false; status = k_gt_inconsistent; }   // option "debug_fail_gather" (testing aid): the next gather reports an inconsistency
    if (status == k_gt_ok) break;
    if (status != k_gt_heap_overflow || attempt == 2) return fail(h, EMAT_ERR_INTERNAL, std::string("emat_tree_reassemble: ") + gt_status_text(status));
    // the atomics kept counting: G.used is what the heaps need (nothing of the old content is read by the gather)
    HIP_TRY(G.mut_heap.alloc((size_t)G.used[0] * 2 + 1024)); HIP_TRY(G.iv_heap.alloc((size_t)G.used[1] * 2 + 1024)); HIP_TRY(G.fs_heap.alloc((size_t)G.used[2] * 2 + 1024));
    ++G.heap_regrows;
    emat_status st = gt_launch_gather(h); if (st) return st;
  }
  if (!G.gather_rd.empty()) {   // the reference sequence moved with the root sequence: its derived tables follow
    for (const GRootDelta& d : G.gather_rd) h->ref[d.site] = d.to;
    HIP_TRY(hipMemcpy(h->d_ref.p, h->ref.data(), h->ref.size(), hipMemcpyHostToDevice));
    refresh_ref_derived(h);
  }
  return EMAT_OK;
}
emat_status gt_gather_local(emat_backend* h, const std::vector<GRootDelta>& rd, bool wait = true) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  const size_t room = std::max<size_t>(rd.size(), (size_t)k_gt_max_root_deltas);
  HIP_TRY(G.root_deltas.alloc(room)); HIP_TRY(G.n_root_deltas.alloc(1)); HIP_TRY(G.root_deltas_in.alloc(room));
  if (!rd.empty()) HIP_TRY(hipMemcpy(G.root_deltas_in.p, rd.data(), rd.size() * sizeof(GRootDelta), hipMemcpyHostToDevice));
  G.gather_rd = rd;
  emat_status st = gt_launch_gather(h); if (st) return st;
  G.gather_pending = true;
  return wait || !rd.empty() ? gt_finish_gather(h) : EMAT_OK;   // (a changed root sequence changes the host's tables too: not worth postponing)
}

emat_status gt_reassemble_end(emat_backend* h) {
  emat_status st = gt_fetch_mirrors(h); if (st) return st;
  h->gt.parts_live = false;
  return EMAT_OK;
}

emat_status gt_reassemble_begin(emat_backend* h) {
  GTreeHost& G = h->gt;
  if (!G.parts_live) return fail(h, EMAT_ERR_STATE, "emat_tree_repartition first");
  if ((int)h->parts.size() != G.hi - G.lo) return fail(h, EMAT_ERR_STATE, "the parts on the device are not the ones emat_tree_repartition made");
  emat_status st = finish_pass(h); if (st) return st;   // every chain ran to completion (or was given more room and finished)
  return materialize(h);                                // (a recovery leaves the parts decoded on the host: back onto their slabs)
}

}  // namespace
extern "C" {

emat_status emat_tree_reassemble(emat_backend* h, int32_t* num_root_deltas, int32_t* site, uint8_t* from, uint8_t* to, int32_t capacity) {
  if (!h || (capacity > 0 && (!site || !from || !to))) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  GTreeHost& G = h->gt;
  const bool verbose = verbose_reports();
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  if (G.parts_live && (G.lo != 0 || G.hi != G.P)) return fail(h, EMAT_ERR_STATE, "this process holds a block of the parts: the other processes' nodes must be exchanged (emat_tree_gather_local, _export_nodes, _apply_nodes, _reassemble_end)");
  HostLaps laps;
  st = gt_reassemble_begin(h); if (st) return st;
  const auto t1 = now(); laps.mark("tree_reassemble: 1 finish_pass (waits for the moves)");
  // One process holds every part.  What the host needs to draw the next partition -- the links, the root, the changes of the root
  // sequence -- is written and fetched first (three small kernels, one wait); the gather of every list follows and is NOT waited
  // for: it runs while the caller picks and refines its next stencil, and whoever touches the tree next finishes it (gt_require).
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  std::vector<GRootDelta> rd;
  {
    const size_t n = (size_t)G.n;
    HIP_TRY(G.root_deltas.alloc(k_gt_max_root_deltas)); HIP_TRY(G.n_root_deltas.alloc(1));
    HIP_TRY(G.d_kids.alloc(n)); HIP_TRY(G.pin_kids.resize(n * sizeof(int2))); HIP_TRY(G.d_root_t.alloc(1)); HIP_TRY(G.pin_small.resize(16));
    HIP_TRY(hipMemsetAsync(G.n_root_deltas.p, 0, sizeof(int32_t), h->stream));
    hipLaunchKernelGGL(k_gt_root_deltas, dim3(1), dim3(k_wave), 0, h->stream, (const uint8_t*)h->d_slabs.p, (const uint64_t*)h->d_slab_off.p, G.root_part - G.lo, G.root_deltas.p, (int)G.root_deltas.n, G.n_root_deltas.p);
    hipLaunchKernelGGL(k_gt_gather_links, dim3((unsigned)(G.hi - G.lo)), dim3(k_wave), 0, h->stream, G.dev(), G.partition(), (const uint8_t*)h->d_slabs.p, (const uint64_t*)h->d_slab_off.p, G.lo, G.d_kids.p, G.d_root_t.p);
    if (!G.d_kids_current) hipLaunchKernelGGL(k_gt_pack_kids, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, G.dev(), G.d_kids.p);   // (the first time: the tips' entries)
    HIP_TRY(hipGetLastError());
    uint8_t* small = G.pin_small.data();
    HIP_TRY(hipMemcpyAsync(G.pin_kids.data(), G.d_kids.p, n * sizeof(int2), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(small, G.root.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(small + 4, G.n_root_deltas.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(small + 8, G.d_root_t.p, 8, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    int32_t nd = 0;
    std::memcpy(&G.h_root, small, 4); std::memcpy(&nd, small + 4, 4); std::memcpy(&G.h_root_t, small + 8, 8);
    G.d_kids_current = true; G.full_mirrors_stale = true;
    if ((size_t)nd > G.root_deltas.n) { bool owner = false; st = gt_root_deltas(h, rd, owner); if (st) return st; }   // (more changes than there was room for: read again, with the room)
    else { rd.resize((size_t)nd); if (nd > 0) HIP_TRY(hipMemcpy(rd.data(), G.root_deltas.p, (size_t)nd * sizeof(GRootDelta), hipMemcpyDeviceToHost)); }
  }
  laps.mark("tree_reassemble: 2 root deltas, links, children mirror");
  st = gt_gather_local(h, rd, false); if (st) return st;
  const auto t2 = now(); laps.mark("tree_reassemble: 3 launch of k_gt_gather");
  G.parts_live = false;
  const int nd = (int)rd.size();
  if (verbose) fprintf(stderr, "[emat] tree_reassemble: wait for the moves + status check %.1f ms | root changes (%d) + k_gt_gather + reference tables %.1f ms | topology + times D2H %.1f ms\n",
                       ms(t0, t1), nd, ms(t1, t2), ms(t2, now()));
  if (num_root_deltas) *num_root_deltas = nd;
  if (nd > capacity) return capacity > 0 || num_root_deltas == nullptr ? fail(h, EMAT_ERR_BUFFER_TOO_SMALL, "emat_tree_reassemble: more root changes than the caller has room for (the tree itself is complete; emat_tree_download returns the reference sequence)") : EMAT_OK;
  for (int k = 0; k < nd; ++k) { site[k] = rd[k].site; from[k] = rd[k].from; to[k] = rd[k].to; }
  return EMAT_OK;
}

emat_status emat_tree_get_root_deltas(emat_backend* h, int32_t* num_root_deltas, int32_t* site, uint8_t* from, uint8_t* to, int32_t capacity) {
  if (!h || !num_root_deltas || (capacity > 0 && (!site || !from || !to))) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  st = gt_reassemble_begin(h); if (st) return st;
  std::vector<GRootDelta> rd; bool owner = false;
  st = gt_root_deltas(h, rd, owner); if (st) return st;
  *num_root_deltas = owner ? (int32_t)rd.size() : -1;
  if ((int)rd.size() > capacity) return fail(h, EMAT_ERR_BUFFER_TOO_SMALL, "emat_tree_get_root_deltas: arrays too small");
  for (size_t k = 0; k < rd.size(); ++k) { site[k] = rd[k].site; from[k] = rd[k].from; to[k] = rd[k].to; }
  return EMAT_OK;
}

emat_status emat_tree_gather_local(emat_backend* h, int32_t num_root_deltas, const int32_t* site, const uint8_t* from, const uint8_t* to) {
  if (!h || num_root_deltas < 0 || (num_root_deltas > 0 && (!site || !from || !to))) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  st = gt_reassemble_begin(h); if (st) return st;
  std::vector<GRootDelta> rd((size_t)num_root_deltas);
  for (int k = 0; k < num_root_deltas; ++k) { GRootDelta d{}; d.site = site[k]; d.from = from[k]; d.to = to[k]; rd[k] = d; }
  return gt_gather_local(h, rd);
}

// buffer: { uint32 magic, uint32 n_entries, uint32 tops[3], int32 new_root, uint32 pad[2] } GNodeExport[n_entries] MutRec[tops0] IvRec[tops1] FsRec[tops2]
emat_status emat_tree_export_nodes(emat_backend* h, uint8_t* buf, uint64_t capacity, uint64_t* bytes_needed) {
  if (!h || !bytes_needed) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  if (!G.parts_live) return fail(h, EMAT_ERR_STATE, "emat_tree_repartition and emat_tree_gather_local first");
  // one entry per node of every local part (k_gt_export's comment says what an entry carries)
  st = gt_partition_to_host(h); if (st) return st;
  std::vector<uint32_t> ids;
  for (int p = G.lo; p < G.hi; ++p) {
    const int b = G.h_part_off[p], np = G.h_part_off[p + 1] - b;
    for (int s = 0; s < np; ++s) {
      uint32_t id = (uint32_t)G.h_orig[b + s];
      if (s != 0 || p == G.root_part) id |= k_gt_export_owns;   // (the gather made sure a part that is not the root part still has local node 0 as its root)
      if (G.h_kid0[b + s] != EMAT_NO_NODE) id |= k_gt_export_links;
      ids.push_back(id);
    }
  }
  const uint64_t n = ids.size();
  const uint64_t need = 32 + n * sizeof(GNodeExport) + (uint64_t)G.used[0] * sizeof(MutRec) + (uint64_t)G.used[1] * sizeof(IvRec) + (uint64_t)G.used[2] * sizeof(FsRec);
  *bytes_needed = need;
  if (!buf) return EMAT_OK;
  if (capacity < need) return fail(h, EMAT_ERR_BUFFER_TOO_SMALL, "emat_tree_export_nodes: buffer too small");
  DevBuf<uint32_t> d_ids; DevBuf<GNodeExport> d_out;
  HIP_TRY(d_ids.upload(ids.data(), n)); HIP_TRY(d_out.alloc(n));
  hipLaunchKernelGGL(k_gt_export, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, G.dev(), (const uint32_t*)d_ids.p, (int)n, d_out.p);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  uint32_t hdr[8] = {0x454E4F44u, (uint32_t)n, G.used[0], G.used[1], G.used[2], 0u, 0u, 0u};
  int32_t new_root = EMAT_NO_NODE;
  if (G.root_part >= G.lo && G.root_part < G.hi) HIP_TRY(hipMemcpy(&new_root, G.root.p, sizeof(new_root), hipMemcpyDeviceToHost));
  std::memcpy(&hdr[5], &new_root, 4);
  // `buf` may be host memory or device memory (a buffer the caller hands to RCCL as it is): the copy kind is resolved by the runtime
  uint8_t* w = buf;
  HIP_TRY(hipMemcpy(w, hdr, 32, hipMemcpyDefault)); w += 32;
  if (n) HIP_TRY(hipMemcpy(w, d_out.p, n * sizeof(GNodeExport), hipMemcpyDefault));
  w += n * sizeof(GNodeExport);
  if (G.used[0]) HIP_TRY(hipMemcpy(w, G.mut_heap.p, (size_t)G.used[0] * sizeof(MutRec), hipMemcpyDefault));
  w += (size_t)G.used[0] * sizeof(MutRec);
  if (G.used[1]) HIP_TRY(hipMemcpy(w, G.iv_heap.p, (size_t)G.used[1] * sizeof(IvRec), hipMemcpyDefault));
  w += (size_t)G.used[1] * sizeof(IvRec);
  if (G.used[2]) HIP_TRY(hipMemcpy(w, G.fs_heap.p, (size_t)G.used[2] * sizeof(FsRec), hipMemcpyDefault));
  return EMAT_OK;
}

emat_status emat_tree_apply_nodes(emat_backend* h, const uint8_t* buf, uint64_t bytes) {
  if (!h || !buf || bytes < 32) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  GTreeHost& G = h->gt;
  if (!G.parts_live) return fail(h, EMAT_ERR_STATE, "emat_tree_repartition and emat_tree_gather_local first");
  // `buf` may be host or device memory (what an all-gather on device buffers delivers): the header and the entries are checked on
  // the host either way (a few MB at most), the three heap segments go where they belong without touching it
  bool on_device = false;
  { hipPointerAttribute_t at{}; if (hipPointerGetAttributes(&at, buf) == hipSuccess) on_device = at.type == hipMemoryTypeDevice; else (void)hipGetLastError(); }
  uint32_t hdr[8];
  HIP_TRY(hipMemcpy(hdr, buf, 32, hipMemcpyDefault));
  const uint64_t n = hdr[1], m0 = hdr[2], m1 = hdr[3], m2 = hdr[4];
  int32_t new_root; std::memcpy(&new_root, &hdr[5], 4);
  if (hdr[0] != 0x454E4F44u || bytes != 32 + n * sizeof(GNodeExport) + m0 * sizeof(MutRec) + m1 * sizeof(IvRec) + m2 * sizeof(FsRec)) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_apply_nodes: not a buffer of emat_tree_export_nodes");
  if (new_root != EMAT_NO_NODE && (new_root < 0 || new_root >= G.n)) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_apply_nodes: root out of range");
  std::vector<GNodeExport> staged;
  if (on_device && n) { staged.resize(n); HIP_TRY(hipMemcpy(staged.data(), buf + 32, n * sizeof(GNodeExport), hipMemcpyDeviceToHost)); }
  const GNodeExport* e = on_device ? staged.data() : (const GNodeExport*)(buf + 32);
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t o = e[i].node_and_flags & k_gt_export_node_mask;
    const bool owns = (e[i].node_and_flags & k_gt_export_owns) != 0, links = (e[i].node_and_flags & k_gt_export_links) != 0;
    if (o >= (uint32_t)G.n || (links && (e[i].c0 < 0 || e[i].c0 >= G.n || e[i].c1 < 0 || e[i].c1 >= G.n)) ||
        (owns && ((uint64_t)e[i].muts.off + e[i].muts.cnt > m0 || (uint64_t)e[i].miss.off + e[i].miss.cnt > m1 || (uint64_t)e[i].mfs.off + e[i].mfs.cnt > m2)))
      return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_apply_nodes: entry out of range");
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  st = gt_grow_keeping(h, G.mut_heap, G.used[0], (size_t)G.used[0] + m0 > G.mut_heap.n ? ((size_t)G.used[0] + m0) * 2 : G.mut_heap.n); if (st) return st;
  st = gt_grow_keeping(h, G.iv_heap, G.used[1], (size_t)G.used[1] + m1 > G.iv_heap.n ? ((size_t)G.used[1] + m1) * 2 : G.iv_heap.n); if (st) return st;
  st = gt_grow_keeping(h, G.fs_heap, G.used[2], (size_t)G.used[2] + m2 > G.fs_heap.n ? ((size_t)G.used[2] + m2) * 2 : G.fs_heap.n); if (st) return st;
  const uint8_t* r = buf + 32 + n * sizeof(GNodeExport);
  if (m0) HIP_TRY(hipMemcpy(G.mut_heap.p + G.used[0], r, m0 * sizeof(MutRec), hipMemcpyDefault));
  r += m0 * sizeof(MutRec);
  if (m1) HIP_TRY(hipMemcpy(G.iv_heap.p + G.used[1], r, m1 * sizeof(IvRec), hipMemcpyDefault));
  r += m1 * sizeof(IvRec);
  if (m2) HIP_TRY(hipMemcpy(G.fs_heap.p + G.used[2], r, m2 * sizeof(FsRec), hipMemcpyDefault));
  DevBuf<GNodeExport> d_in;
  HIP_TRY(d_in.alloc(n));
  if (n) HIP_TRY(hipMemcpy(d_in.p, buf + 32, n * sizeof(GNodeExport), hipMemcpyDefault));
  G.climb_current = false;
  hipLaunchKernelGGL(k_gt_apply, dim3((unsigned)((n + 255) / 256 + 1)), dim3(256), 0, h->stream, G.dev(), (const GNodeExport*)d_in.p, (int)n, G.used[0], G.used[1], G.used[2], new_root);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  G.used[0] += (uint32_t)m0; G.used[1] += (uint32_t)m1; G.used[2] += (uint32_t)m2;
  HIP_TRY(hipMemcpy(G.tops.p, G.used, sizeof(G.used), hipMemcpyHostToDevice));
  return EMAT_OK;
}

emat_status emat_tree_reassemble_end(emat_backend* h) {
  if (!h) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = gt_require(h, true); if (st) return st;
  if (!h->gt.parts_live) return fail(h, EMAT_ERR_STATE, "emat_tree_repartition first");
  return gt_reassemble_end(h);
}

/* debugging aid (not part of the boundary): how often the cut-state pools and the list heaps had to grow */
emat_status emat_debug_tree_counters(emat_backend* h, int32_t* out3) {
  if (!h || !out3) return EMAT_ERR_INVALID_ARGUMENT;
  out3[0] = h->gt.pool_regrows; out3[1] = h->gt.heap_regrows; out3[2] = h->gt.large_measures;
  return EMAT_OK;
}

}  // extern "C"
#endif  // EMAT_GTREE_HOST_HPP_
