// emat_multi.cpp -- ONE process, several GPUs: the run driver of include/emat_host.h over n backends, with the exchanges of a
// sharded cycle done here, in C++, over RCCL (xGMI) -- or through host memory where RCCL cannot be used (two backends on one
// device: the single-GPU test box).
//
// The reference's only parallel seam is in-process: Run::run_local_moves hands one task per Subrun to a thread pool and waits
// (core/run.cpp:682-693).  This file is that seam for a C++ `Run` that owns several MI355X: every GPU holds the whole tree in its
// HBM (a few tens of MB) and a contiguous block of the partition's parts; per cycle the shards
//   cut the same partition (no exchange: same seed, same topology),
//   run their blocks' moves side by side (no exchange: parts are independent between repartition and reassemble),
//   gather their own parts into their own copy of the tree, and ALL-GATHER what their parts own (times, lists, child links)
//   -- emat_tree_export_nodes writes into the device buffer RCCL sends, emat_tree_apply_nodes reads what arrived --
//   and ALL-REDUCE the two log-posterior totals (run.cpp:340-348).
// It uses nothing but the public C-ABI of include/emat_backend.h and include/emat_host.h, so it is also the worked example of a
// single-process adaptor.  RCCL is resolved at run time (dlopen of librccl.so: no link-time dependency, and no
// clash with the copy a host application such as PyTorch may have loaded already).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enumerators only: every function is looked up with dlsym

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/emat_backend.h"
#include "../../include/emat_host.h"
#include "host_parallel.hpp"   // verbose_reports

namespace {

std::string& rccl_library_name() { static std::string s; return s; }

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool load(std::string& err) {
    // emat_multi_set_rccl_library names the ONE library to try; otherwise the usual names
    const std::string& only = rccl_library_name();
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    if (!only.empty()) lib = dlopen(only.c_str(), RTLD_NOW | RTLD_LOCAL);
    else for (const char* n : names) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) { const char* e = dlerror(); err = std::string("librccl.so could not be loaded: ") + (e ? e : "?"); return false; }   // dlerror() clears what it returns: once
    auto sym = [&](const char* s) -> void* { void* p = dlsym(lib, s); if (!p) err = std::string("librccl.so lacks ") + s; return p; };
    CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
    GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
    AllReduce = (decltype(AllReduce))sym("ncclAllReduce"); AllGather = (decltype(AllGather))sym("ncclAllGather");
    GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
    return CommInitAll && CommDestroy && GroupStart && GroupEnd && AllReduce && AllGather && GetErrorString;
  }
};

struct Shard {
  int device = 0;
  emat_backend* backend = nullptr;
  emat_run* run = nullptr;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;       // the collectives of this shard (the engine launches on a stream of its own and hands over finished buffers)
  uint8_t* send = nullptr; size_t send_cap = 0;
  uint8_t* recv = nullptr; size_t recv_cap = 0;
  double* totals = nullptr;           // [4]: two in, two out
  std::vector<uint8_t> host_export;   // host exchange
  uint64_t export_bytes = 0;
};

}  // namespace

struct emat_multi {
  std::vector<Shard> shards;
  int32_t L = 0;
  int64_t nodes = 0;
  bool use_rccl = false;
  bool parts_out = false;
  bool following = false;   // shards 1.. take shard 0's partition draws (emat_run_follow_draws)
  Rccl rccl;
  std::string last_error;
  std::string exchange_note;

  emat_status fail(emat_status st, const std::string& m) { last_error = m; return st; }
  // One persistent worker thread per shard (round 6; until then every phase of a cycle started and joined n threads: six fan-outs per
  // cycle).  A phase hands every worker the same job with its shard's index and waits for all of them; the host work of a phase is per
  // shard, kernels are launched asynchronously, so the GPUs run side by side whatever the host does next.  Shard 0's job runs on the
  // calling thread.
  struct Workers {
    std::vector<std::thread> threads;
    std::mutex mu; std::condition_variable go, done;
    const std::function<int(int)>* job = nullptr;
    uint64_t generation = 0; int pending = 0; bool quit = false;
    std::vector<int> rc;
    void start(int n) {
      rc.assign((size_t)n, EMAT_OK);
      for (int i = 1; i < n; ++i) threads.emplace_back([this, i] {
        uint64_t seen = 0;
        for (;;) {
          const std::function<int(int)>* j;
          { std::unique_lock<std::mutex> lk(mu); go.wait(lk, [&] { return quit || generation != seen; }); if (quit) return; seen = generation; j = job; }
          const int r = (*j)(i);
          { std::lock_guard<std::mutex> lk(mu); rc[(size_t)i] = r; if (--pending == 0) done.notify_one(); }
        }
      });
    }
    void run(const std::function<int(int)>& f) {
      const int n = (int)rc.size();
      if (n > 1) { std::lock_guard<std::mutex> lk(mu); job = &f; pending = n - 1; ++generation; }
      if (n > 1) go.notify_all();
      rc[0] = f(0);
      if (n > 1) { std::unique_lock<std::mutex> lk(mu); done.wait(lk, [&] { return pending == 0; }); }
    }
    void stop() {
      { std::lock_guard<std::mutex> lk(mu); quit = true; }
      go.notify_all();
      for (auto& t : threads) t.join();
      threads.clear();
    }
  } workers;
  template <class F> emat_status on_every_shard(F&& f) {
    const int n = (int)shards.size();
    const std::function<int(int)> job = [&](int i) { return (int)f(i); };
    workers.run(job);
    const std::vector<int>& rc = workers.rc;
    for (int i = 0; i < n; ++i) if (rc[(size_t)i] != EMAT_OK) {
      const char* e = emat_run_last_error(shards[i].run);
      const char* b = emat_last_error(shards[i].backend);
      return fail((emat_status)rc[(size_t)i], "shard " + std::to_string(i) + " (device " + std::to_string(shards[i].device) + "): " + (e && *e ? e : (b ? b : "")));
    }
    return EMAT_OK;
  }
  emat_status hip_fail(hipError_t e, const char* what) { return fail(EMAT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); }
  emat_status nccl_fail(ncclResult_t r, const char* what) { return fail(EMAT_ERR_HIP, std::string(what) + ": " + rccl.GetErrorString(r)); }
};

#define M_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) return m->hip_fail(_e, #call); } while (0)
#define M_NCCL(call) do { ncclResult_t _r = (call); if (_r != ncclSuccess) return m->nccl_fail(_r, #call); } while (0)

namespace {

emat_status grow(emat_multi* m, Shard& s, uint8_t*& p, size_t& cap, size_t want) {
  if (want <= cap) return EMAT_OK;
  M_HIP(hipSetDevice(s.device));
  if (p) M_HIP(hipFree(p));
  p = nullptr; cap = 0;
  const size_t bytes = want + want / 4 + 4096;
  M_HIP(hipMalloc((void**)&p, bytes));
  cap = bytes;
  return EMAT_OK;
}

}  // namespace

extern "C" {

emat_status emat_run_create_multi(const int32_t* devices, int32_t n, const emat_config* cfg, const emat_flat_tree* tree, const uint8_t* ref_sequence,
                                  int32_t num_sites, uint64_t seed, int32_t exchange, emat_multi** out) {
  if (!devices || n < 1 || n > 64 || !cfg || !tree || !ref_sequence || !out || num_sites <= 0 || exchange < 0 || exchange > 2) return EMAT_ERR_INVALID_ARGUMENT;
  auto m = std::make_unique<emat_multi>();
  m->L = num_sites; m->nodes = tree->num_nodes;
  m->shards.resize((size_t)n);
  bool distinct = true;
  for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) if (devices[i] == devices[j]) distinct = false;
  auto cleanup = [&](emat_status st) { emat_multi_destroy(m.release()); return st; };
  for (int i = 0; i < n; ++i) {
    Shard& s = m->shards[(size_t)i];
    s.device = devices[i];
    emat_config c = *cfg; c.device = devices[i]; c.num_sites = num_sites;
    emat_status st = emat_backend_create(&c, &s.backend); if (st) return cleanup(st);
    st = emat_run_create(s.backend, tree, ref_sequence, num_sites, seed, &s.run); if (st) return cleanup(st);   // the same seed everywhere: the same partitions
    st = emat_run_set_shard(s.run, i, n); if (st) return cleanup(st);
    st = emat_run_set_device_tree(s.run, 1); if (st) return cleanup(st);
  }
  m->workers.start(n);
  // the exchange: RCCL when asked for (1) or possible (2: every shard on a device of its own and librccl.so loads), else host memory
  if (exchange == 1 && !distinct) return cleanup(EMAT_ERR_INVALID_ARGUMENT);   // RCCL refuses two ranks on one device
  if (exchange != 0 && distinct) {
    std::string err;
    if (m->rccl.load(err)) {
      std::vector<ncclComm_t> comms((size_t)n);
      std::vector<int> devs(devices, devices + n);
      ncclResult_t r = m->rccl.CommInitAll(comms.data(), n, devs.data());
      if (r == ncclSuccess) { for (int i = 0; i < n; ++i) m->shards[(size_t)i].comm = comms[(size_t)i]; m->use_rccl = true; m->exchange_note = "RCCL (ncclCommInitAll over " + std::to_string(n) + " device(s))"; }
      else err = std::string("ncclCommInitAll: ") + m->rccl.GetErrorString(r);
    }
    if (!m->use_rccl) { if (exchange == 1) { fprintf(stderr, "[emat_multi] %s\n", err.c_str()); return cleanup(EMAT_ERR_HIP); } m->exchange_note = "host memory (" + err + ")"; }
  } else m->exchange_note = distinct ? "host memory (asked for)" : "host memory (two shards share a device: RCCL takes one rank per device)";
  for (auto& s : m->shards) {
    if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&s.totals, 4 * sizeof(double)) != hipSuccess) return cleanup(EMAT_ERR_HIP);
  }
  if (emat::verbose_reports()) fprintf(stderr, "[emat_multi] %d shard(s), exchange: %s\n", n, m->exchange_note.c_str());
  *out = m.release();
  return EMAT_OK;
}

emat_status emat_multi_destroy(emat_multi* m) {
  if (!m) return EMAT_OK;
  m->workers.stop();
  for (auto& s : m->shards) {
    (void)hipSetDevice(s.device);
    if (s.comm) m->rccl.CommDestroy(s.comm);
    if (s.send) (void)hipFree(s.send);
    if (s.recv) (void)hipFree(s.recv);
    if (s.totals) (void)hipFree(s.totals);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    if (s.run) emat_run_destroy(s.run);
    if (s.backend) emat_backend_destroy(s.backend);
  }
  delete m;
  return EMAT_OK;
}
const char* emat_multi_last_error(const emat_multi* m) { return m ? m->last_error.c_str() : "null run"; }
const char* emat_multi_exchange(const emat_multi* m) { return m ? m->exchange_note.c_str() : ""; }
int32_t emat_multi_num_shards(const emat_multi* m) { return m ? (int32_t)m->shards.size() : 0; }
emat_backend* emat_multi_backend(emat_multi* m, int32_t i) { return (m && i >= 0 && i < (int)m->shards.size()) ? m->shards[(size_t)i].backend : nullptr; }
emat_run* emat_multi_shard(emat_multi* m, int32_t i) { return (m && i >= 0 && i < (int)m->shards.size()) ? m->shards[(size_t)i].run : nullptr; }

#define M_EVERY(expr) do { if (!m) return EMAT_ERR_INVALID_ARGUMENT; for (auto& s : m->shards) { emat_status st = (expr); if (st) return m->fail(st, emat_run_last_error(s.run)); } return EMAT_OK; } while (0)
emat_status emat_multi_set_num_parts(emat_multi* m, int32_t num_parts) { M_EVERY(emat_run_set_num_parts(s.run, num_parts)); }
emat_status emat_multi_set_max_part_nodes(emat_multi* m, int32_t max_nodes) { M_EVERY(emat_run_set_max_part_nodes(s.run, max_nodes)); }
/* emat_set_option on every shard's backend (before the first repartition) */
emat_status emat_multi_set_option(emat_multi* m, const char* key, const char* value) { M_EVERY(emat_set_option(s.backend, key, value)); }
/* Which RCCL to load (dlopen name or path) instead of trying the usual names; process-wide, before emat_run_create_multi.  NULL or "" = the usual names. */
emat_status emat_multi_set_rccl_library(const char* name) { rccl_library_name() = name ? name : ""; return EMAT_OK; }
/* test hook: can RCCL be loaded with the present setting?  EMAT_OK, or EMAT_ERR_HIP with the reason in `err` (NUL-terminated, cut to err_cap). */
emat_status emat_multi_debug_rccl_load(char* err, int32_t err_cap) {
  Rccl r; std::string e;
  const bool ok = r.load(e);
  if (r.lib) dlclose(r.lib);
  if (err && err_cap > 0) { const size_t n = std::min(e.size(), (size_t)err_cap - 1); memcpy(err, e.data(), n); err[n] = 0; }
  return ok ? EMAT_OK : EMAT_ERR_HIP;
}
emat_status emat_multi_set_hky(emat_multi* m, double mu, double kappa, const double pi[4], const double* nu_l) { M_EVERY(emat_run_set_hky(s.run, mu, kappa, pi, nu_l)); }
emat_status emat_multi_set_pop_model(emat_multi* m, const emat_pop_model* pm) { M_EVERY(emat_run_set_pop_model(s.run, pm)); }
emat_status emat_multi_set_coalescent_t_step(emat_multi* m, double t_step) { M_EVERY(emat_run_set_coalescent_t_step(s.run, t_step)); }
emat_status emat_multi_set_flags(emat_multi* m, int32_t only_displacing_inner_nodes, int32_t topology_moves_enabled) { M_EVERY(emat_run_set_flags(s.run, only_displacing_inner_nodes, topology_moves_enabled)); }
emat_status emat_multi_set_paranoid(emat_multi* m, int32_t on) { M_EVERY(emat_run_set_paranoid(s.run, on)); }

/* Run::repartition on every shard: the same stencil pick, the same cut, every shard builds the slabs of its own block of parts. */
emat_status emat_multi_repartition(emat_multi* m) {
  if (!m) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st;
  // ONE draw per process (round 6): shard 0 picks and refines the cycle's stencil -- on the host pool, with the whole machine to itself -- and
  // the others take its cut nodes (emat_run_follow_draws); every shard then cuts its own copy of the tree and builds its own block's slabs,
  // side by side.  (Until then every shard drew and refined the same stencil on a pool of its own.)
  if (m->shards.size() > 1) {
    if (!m->following) {   // (set up here rather than at creation: the partition settings are the caller's until the first cut)
      for (size_t i = 1; i < m->shards.size(); ++i) { st = emat_run_follow_draws(m->shards[i].run, m->shards[0].run); if (st) return m->fail(st, emat_run_last_error(m->shards[i].run)); }
      m->following = true;
    }
    st = emat_run_draw_partition(m->shards[0].run); if (st) return m->fail(st, emat_run_last_error(m->shards[0].run));
  }
  st = m->on_every_shard([&](int i) { return (int)emat_run_repartition(m->shards[(size_t)i].run); });
  if (st == EMAT_OK) m->parts_out = true;
  return st;
}
/* Run::run_local_moves (run.cpp:682-693) over all parts of the run: `count` / parts moves on every part, the remainder one move
 * each on the first parts.  Returns when every GPU has its kernels in flight; whatever reads the parts next waits for them. */
emat_status emat_multi_run_moves(emat_multi* m, int64_t count) {
  if (!m || count < 0) return EMAT_ERR_INVALID_ARGUMENT;
  if (!m->parts_out) return m->fail(EMAT_ERR_STATE, "emat_multi_repartition first");
  return m->on_every_shard([&](int i) { return (int)emat_run_moves_sharded(m->shards[(size_t)i].run, count); });
}
emat_status emat_multi_check_derived(emat_multi* m, double tol_scale) {
  if (!m) return EMAT_ERR_INVALID_ARGUMENT;
  return m->on_every_shard([&](int i) { return (int)emat_check_derived(m->shards[(size_t)i].backend, tol_scale, nullptr, nullptr); });
}

/* Run::reassemble (run.cpp:195-256) + normalize_root: every shard gathers its own parts into its own copy of the tree, the shards
 * exchange what their parts own, and every copy of the tree holds the same nodes again. */
emat_status emat_multi_reassemble(emat_multi* m) {
  if (!m) return EMAT_ERR_INVALID_ARGUMENT;
  if (!m->parts_out) return m->fail(EMAT_ERR_STATE, "emat_multi_repartition first");
  const int n = (int)m->shards.size();
  emat_status st;
  if (n == 1 && !m->use_rccl) {   // nothing to exchange
    st = m->on_every_shard([&](int i) { return (int)emat_run_reassemble(m->shards[(size_t)i].run); });
    if (st == EMAT_OK) m->parts_out = false;
    return st;
  }
  // (1) how the root sequence changed: known to the shard that holds the root part
  std::vector<int32_t> site((size_t)m->L); std::vector<uint8_t> from((size_t)m->L), to((size_t)m->L);
  int32_t nd = -1; std::atomic<int> owners{0};
  // (every shard is asked -- the call also waits for the shard's pass -- side by side; only the owner of the root part has room for an answer)
  int owner_shard = -1;
  for (int i = 0; i < n; ++i) { int32_t lo = 0, hi = 0, lr = -1; if (emat_run_shard_range(m->shards[(size_t)i].run, &lo, &hi, &lr) == EMAT_OK && lr >= 0) owner_shard = i; }
  std::vector<int32_t> counts((size_t)n, -1);
  st = m->on_every_shard([&](int i) {
    int32_t k = -1;
    const bool mine = (i == owner_shard);
    emat_status s1 = emat_tree_get_root_deltas(m->shards[(size_t)i].backend, &k, mine ? site.data() : nullptr, mine ? from.data() : nullptr, mine ? to.data() : nullptr, mine ? m->L : 0);
    if (s1 == EMAT_ERR_BUFFER_TOO_SMALL && !mine) s1 = EMAT_ERR_INTERNAL;   // (a shard the drivers do not know as the owner reports root changes)
    counts[(size_t)i] = k;
    if (k >= 0) ++owners;
    return (int)s1;
  });
  if (st) return st;
  if (owners.load() != 1 || owner_shard < 0 || counts[(size_t)owner_shard] < 0) return m->fail(EMAT_ERR_INTERNAL, "exactly one shard must hold the root part");
  nd = counts[(size_t)owner_shard];
  // (2) every shard: its own parts into its own copy of the tree; (3) what its parts own, as one buffer
  st = m->on_every_shard([&](int i) {
    Shard& s = m->shards[(size_t)i];
    emat_status s1 = emat_tree_gather_local(s.backend, nd, site.data(), from.data(), to.data()); if (s1) return (int)s1;
    return (int)emat_tree_export_nodes(s.backend, nullptr, 0, &s.export_bytes);
  });
  if (st) return st;
  uint64_t most = 0;
  for (auto& s : m->shards) most = std::max(most, s.export_bytes);
  most = (most + 255u) & ~(uint64_t)255u;
  if (m->use_rccl) {
    for (auto& s : m->shards) { st = grow(m, s, s.send, s.send_cap, most); if (st) return st; st = grow(m, s, s.recv, s.recv_cap, most * (uint64_t)n); if (st) return st; }
    st = m->on_every_shard([&](int i) { Shard& s = m->shards[(size_t)i]; uint64_t need = 0; return (int)emat_tree_export_nodes(s.backend, s.send, s.send_cap, &need); });   // kernels write into the buffer RCCL sends; returns with them finished
    if (st) return st;
    for (auto& s : m->shards) if (s.export_bytes < most) { M_HIP(hipSetDevice(s.device)); M_HIP(hipMemsetAsync(s.send + s.export_bytes, 0, most - s.export_bytes, s.stream)); }   // the padding up to the common stride is sent too: defined bytes
    M_NCCL(m->rccl.GroupStart());
    for (auto& s : m->shards) { M_HIP(hipSetDevice(s.device)); M_NCCL(m->rccl.AllGather(s.send, s.recv, most, ncclUint8, s.comm, s.stream)); }
    M_NCCL(m->rccl.GroupEnd());
    for (auto& s : m->shards) { M_HIP(hipSetDevice(s.device)); M_HIP(hipStreamSynchronize(s.stream)); }   // delivered before the apply kernels read it
    st = m->on_every_shard([&](int i) {
      Shard& s = m->shards[(size_t)i];
      for (int r = 0; r < n; ++r) if (r != i) { emat_status s1 = emat_tree_apply_nodes(s.backend, s.recv + (uint64_t)r * most, m->shards[(size_t)r].export_bytes); if (s1) return (int)s1; }
      return (int)EMAT_OK;
    });
  } else {
    st = m->on_every_shard([&](int i) { Shard& s = m->shards[(size_t)i]; s.host_export.resize(s.export_bytes); uint64_t need = 0; return (int)emat_tree_export_nodes(s.backend, s.host_export.data(), s.host_export.size(), &need); });
    if (st) return st;
    st = m->on_every_shard([&](int i) {
      Shard& s = m->shards[(size_t)i];
      for (int r = 0; r < n; ++r) if (r != i) { emat_status s1 = emat_tree_apply_nodes(s.backend, m->shards[(size_t)r].host_export.data(), m->shards[(size_t)r].export_bytes); if (s1) return (int)s1; }
      return (int)EMAT_OK;
    });
  }
  if (st) return st;
  // (4) mirrors refreshed; the drivers learn the new reference sequence
  st = m->on_every_shard([&](int i) {
    Shard& s = m->shards[(size_t)i];
    emat_status s1 = emat_tree_reassemble_end(s.backend); if (s1) return (int)s1;
    return (int)emat_run_note_device_reassembled(s.run, nd, site.data(), to.data());
  });
  if (st == EMAT_OK) m->parts_out = false;
  return st;
}

/* log G and the augmented coalescent prior summed over all parts of the run (run.cpp:340-348): one all-reduce of two doubles. */
emat_status emat_multi_get_totals(emat_multi* m, double* log_G, double* log_augmented_coalescent_prior) {
  if (!m) return EMAT_ERR_INVALID_ARGUMENT;
  const int n = (int)m->shards.size();
  std::vector<double> part((size_t)n * 2);
  emat_status st = m->on_every_shard([&](int i) { return (int)emat_get_totals(m->shards[(size_t)i].backend, &part[(size_t)i * 2], &part[(size_t)i * 2 + 1]); });
  if (st) return st;
  double g = 0.0, a = 0.0;
  if (m->use_rccl) {
    for (int i = 0; i < n; ++i) { Shard& s = m->shards[(size_t)i]; M_HIP(hipSetDevice(s.device)); M_HIP(hipMemcpyAsync(s.totals, &part[(size_t)i * 2], 2 * sizeof(double), hipMemcpyHostToDevice, s.stream)); }
    M_NCCL(m->rccl.GroupStart());
    for (auto& s : m->shards) { M_HIP(hipSetDevice(s.device)); M_NCCL(m->rccl.AllReduce(s.totals, s.totals + 2, 2, ncclFloat64, ncclSum, s.comm, s.stream)); }
    M_NCCL(m->rccl.GroupEnd());
    double got[2];
    Shard& s0 = m->shards[0];
    M_HIP(hipSetDevice(s0.device)); M_HIP(hipMemcpyAsync(got, s0.totals + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, s0.stream)); M_HIP(hipStreamSynchronize(s0.stream));
    for (size_t i = 1; i < m->shards.size(); ++i) { M_HIP(hipSetDevice(m->shards[i].device)); M_HIP(hipStreamSynchronize(m->shards[i].stream)); }
    g = got[0]; a = got[1];
  } else for (int i = 0; i < n; ++i) { g += part[(size_t)i * 2]; a += part[(size_t)i * 2 + 1]; }   // shard order: reproducible
  if (log_G) *log_G = g;
  if (log_augmented_coalescent_prior) *log_augmented_coalescent_prior = a;
  return EMAT_OK;
}

/* Run::do_mcmc_steps (run.cpp:622-657) without its global moves: repartition -> moves -> reassemble, `local_moves_per_cycle` moves per
 * cycle (<= 0: the reference's 50 x nodes). */
emat_status emat_multi_do_mcmc_steps(emat_multi* m, int64_t steps, int64_t local_moves_per_cycle) {
  if (!m || steps < 0) return EMAT_ERR_INVALID_ARGUMENT;
  if (local_moves_per_cycle <= 0) local_moves_per_cycle = 50 * m->nodes;
  int64_t done = 0;
  while (done < steps) {
    emat_status st = emat_multi_repartition(m); if (st) return st;
    const int64_t k = std::min(local_moves_per_cycle, steps - done);
    st = emat_multi_run_moves(m, k); if (st) return st;
    st = emat_multi_reassemble(m); if (st) return st;
    done += k;
  }
  return EMAT_OK;
}

/* The whole tree and the reference sequence as of the last reassemble (every shard holds the same; shard 0 is asked). */
emat_status emat_multi_tree_sizes(emat_multi* m, int32_t* nn, int32_t* nm, int32_t* ni, int32_t* nf) {
  if (!m) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = emat_run_tree_sizes(m->shards[0].run, nn, nm, ni, nf);
  return st ? m->fail(st, emat_run_last_error(m->shards[0].run)) : EMAT_OK;
}
emat_status emat_multi_tree_get(emat_multi* m, int32_t shard, emat_flat_tree* out, uint8_t* ref_sequence) {
  if (!m || shard < 0 || shard >= (int)m->shards.size()) return EMAT_ERR_INVALID_ARGUMENT;
  emat_status st = emat_run_tree_get(m->shards[(size_t)shard].run, out, ref_sequence);
  return st ? m->fail(st, emat_run_last_error(m->shards[(size_t)shard].run)) : EMAT_OK;
}

}  // extern "C"
