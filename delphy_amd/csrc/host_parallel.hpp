// Host-side helper: independent per-part work (subtree copies, slab encode / decode) spread over the host threads.
#ifndef EMAT_HOST_PARALLEL_HPP_
#define EMAT_HOST_PARALLEL_HPP_
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <exception>
#include <functional>
#include <mutex>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

namespace emat {

inline int& host_threads_override() { static int n = 0; return n; }   // emat_set_host_threads: before the first parallel loop of the process
inline int host_threads() {
  unsigned hw = std::thread::hardware_concurrency();
  if (hw == 0) hw = 1;
  if (host_threads_override() > 0) return host_threads_override();   // tuning knob
  return (int)std::min(hw, 16u);   // the per-part work is allocation-bound: measured flat beyond ~16 threads
}

// A small pool of persistent workers: a host cycle issues a few dozen short parallel loops, and starting threads for
// each of them costs more than some of the loops.  One loop runs at a time (callers are serialised by `submit_mu`);
// the calling thread takes part in the work.
class HostPool {
 public:
  static HostPool& instance() { static HostPool p; return p; }
  // Calls f(i) for i in [0, n) in chunks of `grain` dealt dynamically to the threads; rethrows the first exception.
  // (`more_threads` > 0: a loop bound by cache misses rather than by allocation may use up to that many threads, if the machine has them)
  void run(int n, int grain, const std::function<void(int)>& f, int more_threads = 0) {
    int cap = host_threads();
    if (more_threads > cap && host_threads_override() <= 0) cap = std::min(more_threads, (int)std::max(1u, std::thread::hardware_concurrency()));
    const int want = std::min(cap, std::max(1, n / std::max(1, grain)));
    if (want <= 1 || in_worker()) { for (int i = 0; i < n; ++i) f(i); return; }
    std::lock_guard<std::mutex> submit(submit_mu_);
    ensure_workers(want - 1);
    {
      std::lock_guard<std::mutex> g(mu_);
      job_ = &f; n_ = n; grain_ = std::max(1, grain); next_.store(0); err_ = nullptr;
      helpers_wanted_ = want - 1; running_ = 0; ++epoch_;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> g(mu_);
    helpers_wanted_ = 0;                       // late wakers must not join a finished job
    done_cv_.wait(g, [&] { return running_ == 0; });
    job_ = nullptr;
    if (err_) std::rethrow_exception(err_);
  }

 private:
  HostPool() = default;
  ~HostPool() {
    { std::lock_guard<std::mutex> g(mu_); stop_ = true; }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  static bool& in_worker() { static thread_local bool w = false; return w; }
  void ensure_workers(int k) {
    while ((int)workers_.size() < k) workers_.emplace_back([this] { in_worker() = true; loop(); });
  }
  void work() {
    try {
      for (;;) { const int i0 = next_.fetch_add(grain_); if (i0 >= n_) break; for (int i = i0; i < std::min(n_, i0 + grain_); ++i) (*job_)(i); }
    } catch (...) { std::lock_guard<std::mutex> g(mu_); if (!err_) err_ = std::current_exception(); next_.store(n_); }
  }
  void loop() {
    uint64_t seen = 0;
    std::unique_lock<std::mutex> g(mu_);
    for (;;) {
      cv_.wait(g, [&] { return stop_ || (epoch_ != seen && helpers_wanted_ > 0); });
      if (stop_) return;
      seen = epoch_; --helpers_wanted_; ++running_;
      g.unlock();
      work();
      g.lock();
      if (--running_ == 0) done_cv_.notify_all();
    }
  }
  std::mutex submit_mu_, mu_;
  std::condition_variable cv_, done_cv_;
  std::vector<std::thread> workers_;
  const std::function<void(int)>* job_ = nullptr;
  int n_ = 0, grain_ = 1, helpers_wanted_ = 0, running_ = 0;
  std::atomic<int> next_{0};
  uint64_t epoch_ = 0;
  bool stop_ = false;
  std::exception_ptr err_;
};

template <class F> void parallel_for(int n, F&& f, int grain = 32, int more_threads = 0) {
  if (n <= 0) return;
  const std::function<void(int)> fn = [&f](int i) { f(i); };
  HostPool::instance().run(n, grain, fn, more_threads);
}

// The one environment variable the library reads.  EMAT_VERBOSE=<anything> turns the progress reports on stderr on; the value
// "spans" turns on the host-span accounting below INSTEAD (the reports do extra device reads and sorts, which the spans would book).
inline bool verbose_reports() { const char* e = getenv("EMAT_VERBOSE"); return e != nullptr && std::strcmp(e, "spans") != 0; }
inline bool verbose_spans() { const char* e = getenv("EMAT_VERBOSE"); return e != nullptr && std::strcmp(e, "spans") == 0; }

// Where the host's share of a cycle goes (EMAT_VERBOSE=spans): named spans, nested freely, summed per name over the life of the
// process and printed when a backend handle is destroyed.  Off, a span costs one load and a branch.
class HostSpans {
 public:
  static HostSpans& instance() { static HostSpans s; return s; }
  const bool on;
  void add(const char* name, double ms) { std::lock_guard<std::mutex> g(mu_); auto& e = acc_[name]; e.first += ms; e.second += 1; auto& v = all_[name]; v.push_back((float)ms); }
  void report() {
    std::lock_guard<std::mutex> g(mu_);
    if (!on || acc_.empty()) return;
    fprintf(stderr, "[emat] host spans (EMAT_VERBOSE=spans): total ms | calls | mean us | median us\n");
    for (const auto& kv : acc_) {
      std::vector<float>& v = all_[kv.first]; std::sort(v.begin(), v.end());
      fprintf(stderr, "[emat]   %-58s %10.1f %8lld %10.1f %10.1f\n", kv.first.c_str(), kv.second.first, (long long)kv.second.second, 1e3 * kv.second.first / (double)kv.second.second, 1e3 * v[v.size() / 2]);
    }
    acc_.clear(); all_.clear();
  }
 private:
  HostSpans() : on(verbose_spans()) {}
  std::mutex mu_;
  std::map<std::string, std::pair<double, long long>> acc_;
  std::map<std::string, std::vector<float>> all_;
};
struct HostSpan {
  const char* name; std::chrono::steady_clock::time_point t0;
  explicit HostSpan(const char* n) : name(HostSpans::instance().on ? n : nullptr) { if (name) t0 = std::chrono::steady_clock::now(); }
  ~HostSpan() { if (name) HostSpans::instance().add(name, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
};
struct HostLaps {   // consecutive stretches of one function: mark("name") books the time since the previous mark
  std::chrono::steady_clock::time_point last; const bool on;
  HostLaps() : on(HostSpans::instance().on) { if (on) last = std::chrono::steady_clock::now(); }
  void mark(const char* name) { if (!on) return; const auto t = std::chrono::steady_clock::now(); HostSpans::instance().add(name, std::chrono::duration<double, std::milli>(t - last).count()); last = t; }
};
#define EMAT_SPAN_CAT2(a, b) a##b
#define EMAT_SPAN_CAT(a, b) EMAT_SPAN_CAT2(a, b)
#define EMAT_SPAN(name) ::emat::HostSpan EMAT_SPAN_CAT(emat_span_, __LINE__)(name)

}  // namespace emat
#endif  // EMAT_HOST_PARALLEL_HPP_
