// Host-side helper: independent per-part work (subtree copies, slab encode / decode) spread over the host threads.
#ifndef EMAT_HOST_PARALLEL_HPP_
#define EMAT_HOST_PARALLEL_HPP_
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>

namespace emat {

inline int host_threads() {
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  if (const char* e = getenv("EMAT_HOST_THREADS")) return std::max(1, atoi(e));   // tuning knob
  return (int)std::min(hw, 16u);   // the per-part work is allocation-bound: measured flat beyond ~16 threads
}

// Calls f(i) for i in [0, n) in chunks of `grain` dealt dynamically to the threads; rethrows the first exception.
template <class F> void parallel_for(int n, F&& f, int grain = 32) {
  const int T = std::min(host_threads(), std::max(1, n / std::max(1, grain)));
  if (T <= 1) { for (int i = 0; i < n; ++i) f(i); return; }
  std::atomic<int> next{0};
  std::exception_ptr err; std::mutex err_mu;
  auto worker = [&] {
    try {
      for (;;) { const int i0 = next.fetch_add(grain); if (i0 >= n) break; for (int i = i0; i < std::min(n, i0 + grain); ++i) f(i); }
    } catch (...) { std::lock_guard<std::mutex> g(err_mu); if (!err) err = std::current_exception(); }
  };
  std::vector<std::thread> th; th.reserve(T - 1);
  for (int t = 1; t < T; ++t) th.emplace_back(worker);
  worker();
  for (auto& t : th) t.join();
  if (err) std::rethrow_exception(err);
}

}  // namespace emat
#endif  // EMAT_HOST_PARALLEL_HPP_
