// emat_build_host.hpp -- host driver of emat_tree_build_usher_like (kernel and finishing passes in emat_build.hpp).
// Included at the end of emat_backend.hip.
#ifndef EMAT_BUILD_HOST_HPP_
#define EMAT_BUILD_HOST_HPP_

namespace {

const char* build_status_text(int32_t s) {
  switch (s) {
    case 1: return "mutation pool full";
    case 2: return "site-delta buffer full";
    case 3: return "more tying regions than the list holds";
    case 4: return "inconsistent mutation chain (tip descriptors against the reference sequence)";
    case 5: return "a workgroup of the builder never reached a barrier";
    default: return "unknown";
  }
}

// the reference's input checks (phylo_tree.cpp:804-843), plus what the CSR form adds
std::string validate_tip_descs(const emat_tip_descs& td, const std::vector<uint8_t>& ref) {
  const int L = (int)ref.size(), n = td.num_tips;
  if (n < 2) return "at least two tips are needed";
  if (!td.t_min || !td.t_max || !td.delta_offset || !td.miss_offset) return "null array";
  if (td.delta_offset[0] != 0 || td.miss_offset[0] != 0) return "CSR offsets must start at 0";
  for (int i = 0; i < n; ++i) {
    if (!(td.t_min[i] <= td.t_max[i])) return "tip " + std::to_string(i) + ": t_min > t_max";
    if (td.delta_offset[i + 1] < td.delta_offset[i] || td.miss_offset[i + 1] < td.miss_offset[i]) return "CSR offsets must be non-decreasing";
    int prev_end = -1;
    for (int k = td.miss_offset[i]; k < td.miss_offset[i + 1]; ++k) {
      const int s = td.miss_start[k], e = td.miss_end[k];
      if (s < 0 || s >= L || e < 0 || e >= L + 1) return "tip " + std::to_string(i) + ": missing interval outside [0, " + std::to_string(L) + ")";
      if (s >= e || s <= prev_end) return "tip " + std::to_string(i) + ": missing intervals must be sorted, disjoint and non-adjacent";
      prev_end = e;
    }
    int prev_site = -1;
    for (int k = td.delta_offset[i]; k < td.delta_offset[i + 1]; ++k) {
      const int l = td.delta_site[k];
      if (l < 0 || l >= L) return "tip " + std::to_string(i) + ": site of a sequence delta outside [0, " + std::to_string(L) + ")";
      if (l <= prev_site) return "tip " + std::to_string(i) + ": sequence deltas must be in ascending site order, one per site";
      prev_site = l;
      if (td.delta_to[k] > 3 || td.delta_to[k] == ref[l]) return "tip " + std::to_string(i) + ": a sequence delta has equal 'from' and 'to' states (or a state outside 0..3)";
      for (int q = td.miss_offset[i]; q < td.miss_offset[i + 1]; ++q)
        if (l >= td.miss_start[q] && l < td.miss_end[q]) return "tip " + std::to_string(i) + ": site " + std::to_string(l) + " can't both be missing and carry a delta";
    }
  }
  return "";
}

emat_status build_usher_like(emat_backend* h, const emat_tip_descs& td, uint64_t seed) {
  auto set_error = [&](const std::string& s) { h->set_error(s); };
  const int n = td.num_tips, N = 2 * n - 1, L = h->L;
  const int nd_tot = td.delta_offset[n], nm_tot = td.miss_offset[n];
  int max_deltas = 0;
  for (int i = 0; i < n; ++i) max_deltas = std::max(max_deltas, td.delta_offset[i + 1] - td.delta_offset[i]);
  uint32_t pool_cap = (uint32_t)(4 * (size_t)nd_tot + 4096), sd_cap = (uint32_t)(2 * max_deltas + 4096), tie_cap = (uint32_t)N + pool_cap;
  DevBuf<uint8_t> d_ref, d_dto; DevBuf<int32_t> d_doff, d_dsite, d_moff, d_mstart, d_mend;
  HIP_TRY(d_ref.upload(h->ref.data(), (size_t)L)); HIP_TRY(d_doff.upload(td.delta_offset, (size_t)n + 1)); HIP_TRY(d_dsite.upload(td.delta_site, (size_t)nd_tot));
  HIP_TRY(d_dto.upload(td.delta_to, (size_t)nd_tot)); HIP_TRY(d_moff.upload(td.miss_offset, (size_t)n + 1)); HIP_TRY(d_mstart.upload(td.miss_start, (size_t)nm_tot)); HIP_TRY(d_mend.upload(td.miss_end, (size_t)nm_tot));
  for (int attempt = 0; attempt < 6; ++attempt) {
    // the stream starts over with every attempt: the tree is a function of (descriptors, seed) whatever the capacities were
    HostRng rng; rng.key = seed;
    auto uni_cc = [&](double lo, double hi) { return lo + (hi - lo) * ((double)(rng.next64() >> 11) * (1.0 / 9007199254740991.0)); };
    auto uni_oc = [&](double lo, double hi) { return lo + (hi - lo) * (((double)(rng.next64() >> 11) + 1.0) * 0x1.0p-53); };
    std::vector<int32_t> parent(N, EMAT_NO_NODE), c0(N, EMAT_NO_NODE), c1(N, EMAT_NO_NODE), sz(N, 1), ml_cnt(N, 0); std::vector<uint32_t> ml_off(N, 0u); std::vector<double> t(N, 0.0);
    std::vector<MutRec> pool(pool_cap);
    for (int i = 0; i < n; ++i) t[i] = uni_cc((double)td.t_min[i], (double)td.t_max[i]);       // phylo_tree.cpp:857-864
    uint32_t top = 0;
    {   // the first two tips hang off a root whose sequence is the reference sequence (:866-905)
      const int P = n, A = 0, B = 1;
      parent[P] = EMAT_NO_NODE; c0[P] = A; c1[P] = B; parent[A] = P; parent[B] = P; sz[P] = 3;
      const int nA = td.delta_offset[1] - td.delta_offset[0], nB = td.delta_offset[2] - td.delta_offset[1];
      const double tP = std::min(t[A] - (double)nA * 13.0, t[B] - (double)nB * 13.0) - 1.0;
      t[P] = tP;
      for (int tip : {A, B}) {
        ml_off[tip] = top;
        for (int k = td.delta_offset[tip]; k < td.delta_offset[tip + 1]; ++k) {
          MutRec r; r.t = uni_oc(tP, t[tip]); r.site = td.delta_site[k]; r.from = h->ref[td.delta_site[k]]; r.to = td.delta_to[k]; r.pad = 0;
          pool[top++] = r;
        }
        ml_cnt[tip] = (int32_t)(top - ml_off[tip]);
        std::stable_sort(pool.begin() + ml_off[tip], pool.begin() + top, [](const MutRec& a, const MutRec& b) { return a.t < b.t || (a.t == b.t && a.site < b.site); });
      }
    }
    DevBuf<int32_t> d_root, d_parent, d_c0, d_c1, d_sz, d_mlcnt, d_delta, d_vD0, d_vD1, d_vP0, d_vP1, d_a0, d_a1, d_inv, d_cnt, d_off, d_tnode, d_path, d_status;
    DevBuf<uint32_t> d_mloff, d_top; DevBuf<double> d_t, d_tmin, d_tmax; DevBuf<MutRec> d_pool; DevBuf<BDelta> d_sd; DevBuf<uint64_t> d_rng;
    const int32_t root0 = n; const int32_t status0[2] = {0, 0};
    const uint64_t rng0[4] = {rng.key, rng.counter, rng.spare, rng.has_spare ? 1ull : 0ull};
    HIP_TRY(d_root.upload(&root0, 1)); HIP_TRY(d_parent.upload(parent.data(), N)); HIP_TRY(d_c0.upload(c0.data(), N)); HIP_TRY(d_c1.upload(c1.data(), N)); HIP_TRY(d_sz.upload(sz.data(), N));
    HIP_TRY(d_mlcnt.upload(ml_cnt.data(), N)); HIP_TRY(d_mloff.upload(ml_off.data(), N)); HIP_TRY(d_t.upload(t.data(), N)); HIP_TRY(d_pool.upload(pool.data(), pool_cap)); HIP_TRY(d_top.upload(&top, 1));
    for (DevBuf<int32_t>* w : {&d_delta, &d_vD0, &d_vD1, &d_vP0, &d_vP1, &d_a0, &d_a1, &d_inv, &d_cnt, &d_off, &d_path}) HIP_TRY(w->alloc((size_t)N + 1));
    HIP_TRY(d_tnode.alloc(tie_cap)); HIP_TRY(d_tmin.alloc(tie_cap)); HIP_TRY(d_tmax.alloc(tie_cap)); HIP_TRY(d_sd.alloc(sd_cap));
    HIP_TRY(d_rng.upload(rng0, 4)); HIP_TRY(d_status.upload(status0, 2));
    // one workgroup per 2 048 nodes of the finished tree, at most one per four CUs (all of them must be resident: they meet at
    // barriers, and a meeting costs more the more workgroups attend: measured at 30 000 tips, 0.73 / 0.56 / 0.51 / 0.59 ms per tip
    // on 8 / 16 / 32 / 59 workgroups)
    int blocks = std::max(1, std::min(std::max(1, h->num_cus / 4), (N + 2 * k_build_threads - 1) / (2 * k_build_threads)));
    if (h->cfg_build_blocks > 0) blocks = std::max(1, std::min(h->cfg_build_blocks, h->num_cus > 0 ? h->num_cus : 1));
    DevBuf<int32_t> d_shared; HIP_TRY(d_shared.alloc(8 + 64 + 2 * (size_t)blocks));
    HIP_TRY(hipMemsetAsync(d_shared.p, 0, (8 + 64 + 2 * (size_t)blocks) * sizeof(int32_t), h->stream));
    // the difference arrays start out zero; the first three nodes' places in the visiting order (the root, then its second child, then its first)
    HIP_TRY(hipMemsetAsync(d_vD0.p, 0, ((size_t)N + 1) * sizeof(int32_t), h->stream)); HIP_TRY(hipMemsetAsync(d_vD1.p, 0, ((size_t)N + 1) * sizeof(int32_t), h->stream));
    { std::vector<int32_t> pre0((size_t)N + 1, 0); pre0[(size_t)n] = 0; pre0[1] = 1; pre0[0] = 2; HIP_TRY(hipStreamSynchronize(h->stream)); HIP_TRY(hipMemcpy(d_vP0.p, pre0.data(), pre0.size() * sizeof(int32_t), hipMemcpyHostToDevice)); }
    BuildDev b;
    b.n_tips = n; b.L = L; b.ref = d_ref.p; b.d_off = d_doff.p; b.d_site = d_dsite.p; b.d_to = d_dto.p; b.m_off = d_moff.p; b.m_start = d_mstart.p; b.m_end = d_mend.p;
    b.root = d_root.p; b.parent = d_parent.p; b.c0 = d_c0.p; b.c1 = d_c1.p; b.t = d_t.p; b.sz = d_sz.p; b.ml_off = d_mloff.p; b.ml_cnt = d_mlcnt.p; b.pool = d_pool.p; b.pool_cap = pool_cap; b.pool_top = d_top.p;
    b.delta = d_delta.p; b.vD[0] = d_vD0.p; b.vD[1] = d_vD1.p; b.vP[0] = d_vP0.p; b.vP[1] = d_vP1.p; b.anc[0] = d_a0.p; b.anc[1] = d_a1.p; b.inv = d_inv.p; b.cnt = d_cnt.p; b.off = d_off.p;
    b.tie_node = d_tnode.p; b.tie_tmin = d_tmin.p; b.tie_tmax = d_tmax.p; b.tie_cap = tie_cap; b.path = d_path.p; b.sd = d_sd.p; b.sd_cap = sd_cap; b.rng = d_rng.p; b.status = d_status.p;
    DevBuf<long long> d_prof; HIP_TRY(d_prof.alloc(8)); HIP_TRY(hipMemsetAsync(d_prof.p, 0, 8 * sizeof(long long), h->stream));
    b.prof = d_prof.p;
    { const int32_t none = EMAT_NO_NODE; HIP_TRY(hipMemcpyAsync(d_shared.p + 8, &none, sizeof none, hipMemcpyHostToDevice, h->stream)); HIP_TRY(hipStreamSynchronize(h->stream)); }
    b.grid_counter = (unsigned long long*)d_shared.p; b.gmin = d_shared.p + 2; b.stop_flag = d_shared.p + 3; b.gdesc = d_shared.p + 8; b.blk_sum = d_shared.p + 8 + 64; b.blk_sum2 = d_shared.p + 8 + 64 + blocks;
    if (n > 2) {
      // a cooperative launch: the runtime guarantees that every workgroup is resident at once (they meet at barriers of their own)
      int first_tip = 2, last_tip = n;
      void* args[3] = {(void*)&b, (void*)&first_tip, (void*)&last_tip};
      HIP_TRY(hipLaunchCooperativeKernel((const void*)k_build_usher_graft, dim3((unsigned)blocks), dim3(k_build_threads), args, 0, h->stream));
      HIP_TRY(hipStreamSynchronize(h->stream));
    }
    int32_t status[2];
    HIP_TRY(hipMemcpy(status, d_status.p, sizeof status, hipMemcpyDeviceToHost));
    if (verbose_reports() && n > 2) {
      long long pr[8]; HIP_TRY(hipMemcpy(pr, d_prof.p, sizeof pr, hipMemcpyDeviceToHost));
      const double k = 1e-5 / (double)(n - 2);   // ticks of 10 ns -> ms per tip
      fprintf(stderr, "[emat] build_usher_like: %d tips on %d workgroups, per tip: %.3f ms = parallel phases + barriers %.3f | tie sums %.3f (%.1f tying regions) | path + deltas %.3f (finding the path %.3f) | links + sizes %.3f | mutations %.3f\n",
              n, blocks, pr[0] * k, pr[1] * k, pr[2] * k, (double)pr[7] / (n - 2), pr[3] * k, pr[6] * k, pr[4] * k, pr[5] * k);
    }
    if (status[0] == 1) { pool_cap *= 2; tie_cap = (uint32_t)N + pool_cap; continue; }
    if (status[0] == 2) { sd_cap *= 4; continue; }
    if (status[0] == 3) { tie_cap *= 2; continue; }
    if (status[0] == 5) return fail(h, EMAT_ERR_INTERNAL, std::string("emat_tree_build_usher_like: ") + build_status_text(status[0]) + " at tip " + std::to_string(status[1]));
    if (status[0] != 0) return fail(h, EMAT_ERR_INVALID_ARGUMENT, std::string("emat_tree_build_usher_like: ") + build_status_text(status[0]) + " at tip " + std::to_string(status[1]));
    // back to the host for the passes that run once
    int32_t root = 0; uint64_t rng1[4];
    HIP_TRY(hipMemcpy(&root, d_root.p, 4, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(parent.data(), d_parent.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(c0.data(), d_c0.p, (size_t)N * 4, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(c1.data(), d_c1.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(t.data(), d_t.p, (size_t)N * 8, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(ml_off.data(), d_mloff.p, (size_t)N * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(ml_cnt.data(), d_mlcnt.p, (size_t)N * 4, hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(&top, d_top.p, 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pool.data(), d_pool.p, (size_t)top * sizeof(MutRec), hipMemcpyDeviceToHost)); HIP_TRY(hipMemcpy(rng1, d_rng.p, sizeof rng1, hipMemcpyDeviceToHost));
    rng.counter = rng1[1]; rng.spare = rng1[2]; rng.has_spare = rng1[3] != 0;
    std::vector<BHostNode> nodes(N);
    for (int v = 0; v < N; ++v) {
      BHostNode& nd = nodes[v];
      nd.parent = parent[v]; nd.c0 = c0[v]; nd.c1 = c1[v]; nd.t = t[v];
      if (v < n) { nd.t_min = td.t_min[v]; nd.t_max = td.t_max[v]; for (int k = td.miss_offset[v]; k < td.miss_offset[v + 1]; ++k) nd.miss.push_back({td.miss_start[k], td.miss_end[k]}); }
      for (int k = 0; k < ml_cnt[v]; ++k) { const MutRec& m = pool[ml_off[v] + k]; nd.muts.push_back(BHostMut{m.t, m.site, m.from, m.to}); }
    }
    try {
      b_fix_up_missations(nodes, root, h->ref);
      {   // pseudo_date (dates.cpp:63-82): leaves first, one draw per inner node in post-order (first child's subtree, second child's, the node)
        std::vector<std::pair<int, int>> st{{root, 0}};
        while (!st.empty()) {
          auto& [v, k] = st.back();
          if (nodes[v].c0 == EMAT_NO_NODE) { st.pop_back(); continue; }
          if (k < 2) { const int c = k == 0 ? nodes[v].c0 : nodes[v].c1; ++k; st.push_back({c, 0}); continue; }
          const BHostNode& l = nodes[nodes[v].c0]; const BHostNode& r = nodes[nodes[v].c1];
          const double est_l = l.t - (double)l.muts.size() * 13.0, est_r = r.t - (double)r.muts.size() * 13.0;
          nodes[v].t = std::min(est_l, est_r) - (0.5 + (1.5 - 0.5) * ((double)(rng.next64() >> 11) * 0x1.0p-53));
          st.pop_back();
        }
      }
      for (int v = 0; v < N; ++v) if (v != root) b_randomize_branch(nodes[v], nodes[nodes[v].parent].t, rng);   // randomize_mutation_times (phylo_tree.cpp:567-575)
    } catch (const std::exception& ex) { return fail(h, EMAT_ERR_INVALID_ARGUMENT, std::string("emat_tree_build_usher_like: ") + ex.what()); }
    FlatTree& f = h->built.tree;
    f = FlatTree(); f.resize_nodes(N); f.root = root;
    for (int v = 0; v < N; ++v) {
      const BHostNode& nd = nodes[v];
      f.parent[v] = nd.parent; f.child0[v] = nd.c0; f.child1[v] = nd.c1; f.t[v] = nd.t; f.t_min[v] = nd.t_min; f.t_max[v] = nd.t_max;
      for (auto& m : nd.muts) { f.mut_site.push_back(m.site); f.mut_from.push_back(m.from); f.mut_to.push_back(m.to); f.mut_t.push_back(m.t); }
      for (auto& iv : nd.miss) { f.miss_start.push_back(iv.first); f.miss_end.push_back(iv.second); }
      for (auto& fs : nd.mfs) { f.mfs_site.push_back(fs.first); f.mfs_state.push_back(fs.second); }
      f.mut_offset[v + 1] = (int32_t)f.mut_site.size(); f.miss_offset[v + 1] = (int32_t)f.miss_start.size(); f.mfs_offset[v + 1] = (int32_t)f.mfs_site.size();
    }
    h->built.ref = h->ref;
    h->built.valid = true;
    return EMAT_OK;
  }
  return fail(h, EMAT_ERR_CAPACITY, "emat_tree_build_usher_like: the work buffers kept overflowing");
}

}  // namespace

extern "C" {

emat_status emat_tree_build_usher_like(emat_backend* h, const emat_tip_descs* tips, uint64_t seed) {
  if (!h || !tips) return EMAT_ERR_INVALID_ARGUMENT;
  if (!bind_device(h)) return fail(h, EMAT_ERR_HIP, "hipSetDevice failed");
  if (!h->have_ref) return fail(h, EMAT_ERR_STATE, "emat_set_ref_sequence must come first: the descriptors are deltas against it");
  h->built.valid = false;
  const std::string bad = validate_tip_descs(*tips, h->ref);
  if (!bad.empty()) return fail(h, EMAT_ERR_INVALID_ARGUMENT, "emat_tree_build_usher_like: " + bad);
  if (h->host_only) return fail(h, EMAT_ERR_NO_DEVICE, "host-only handle (device = -1): the engine has no CPU fallback");
  // the builder's workgroups must all be resident: nothing of this handle may still be running (a pass, its side launches)
  { emat_status st = emat_synchronize(h); if (st) return st; }
  return build_usher_like(h, *tips, seed);
}
emat_status emat_tree_built_sizes(emat_backend* h, int32_t* num_nodes, int32_t* num_muts, int32_t* num_intervals, int32_t* num_from_states) {
  if (!h) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->built.valid) return fail(h, EMAT_ERR_STATE, "no tree has been built (emat_tree_build_usher_like)");
  const FlatTree& f = h->built.tree;
  if (num_nodes) *num_nodes = f.num_nodes(); if (num_muts) *num_muts = f.num_muts(); if (num_intervals) *num_intervals = f.num_intervals(); if (num_from_states) *num_from_states = f.num_from_states();
  return EMAT_OK;
}
emat_status emat_tree_built_get(emat_backend* h, emat_flat_tree* out) {
  if (!h || !out) return EMAT_ERR_INVALID_ARGUMENT;
  if (!h->built.valid) return fail(h, EMAT_ERR_STATE, "no tree has been built (emat_tree_build_usher_like)");
  FlatTree& f = h->built.tree;
  const int n = f.num_nodes();
  if (out->num_nodes < n || out->cap_muts < f.num_muts() || out->cap_intervals < f.num_intervals() || out->cap_from_states < f.num_from_states()) return fail(h, EMAT_ERR_BUFFER_TOO_SMALL, "emat_tree_built_get: arrays too small");
  out->num_nodes = n; out->root = f.root;
  std::copy(f.parent.begin(), f.parent.end(), out->parent); std::copy(f.child0.begin(), f.child0.end(), out->child0); std::copy(f.child1.begin(), f.child1.end(), out->child1);
  std::copy(f.t.begin(), f.t.end(), out->t); std::copy(f.t_min.begin(), f.t_min.end(), out->t_min); std::copy(f.t_max.begin(), f.t_max.end(), out->t_max);
  std::copy(f.mut_offset.begin(), f.mut_offset.end(), out->mut_offset); std::copy(f.mut_site.begin(), f.mut_site.end(), out->mut_site); std::copy(f.mut_from.begin(), f.mut_from.end(), out->mut_from);
  std::copy(f.mut_to.begin(), f.mut_to.end(), out->mut_to); std::copy(f.mut_t.begin(), f.mut_t.end(), out->mut_t);
  std::copy(f.miss_offset.begin(), f.miss_offset.end(), out->miss_offset); std::copy(f.miss_start.begin(), f.miss_start.end(), out->miss_start); std::copy(f.miss_end.begin(), f.miss_end.end(), out->miss_end);
  std::copy(f.mfs_offset.begin(), f.mfs_offset.end(), out->mfs_offset); std::copy(f.mfs_site.begin(), f.mfs_site.end(), out->mfs_site); std::copy(f.mfs_state.begin(), f.mfs_state.end(), out->mfs_state);
  return EMAT_OK;
}

}  // extern "C"
#endif  // EMAT_BUILD_HOST_HPP_
