// A software grid barrier among the workgroups of ONE XCD of an MI355X, chosen by the hardware's XCC_ID register (not by blockIdx % 8):
// the launch has 8 x the workgroups, every workgroup reads XCC_ID, those on XCD 0 take a rank from a counter and take part, the
// others leave.  Fences: (1) agent scope (L2 write-back + invalidate), (3) s_waitcnt + buffer_inv sc0 (this CU's L1 only: the
// XCD's L2 is the point of coherence), (5) nothing but s_waitcnt, loads with sc1 (through L1).  Counts stale reads of a neighbour's slot.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ inline int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 15); }
template <int kFence> __global__ void __launch_bounds__(1024) k_bar(unsigned long long* counter, int* data, int* ranks, int rounds, long long* ticks, int payload) {
  __shared__ int s_rank, s_blocks;
  if (threadIdx.x == 0) {
    const bool in = kFence >= 6 || xcc_id() == 0;
    s_rank = in ? atomicAdd(&ranks[0], 1) : -1;
    atomicAdd(&ranks[1], 1);
    while (__hip_atomic_load(&ranks[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)gridDim.x) __builtin_amdgcn_s_sleep(1);
    s_blocks = __hip_atomic_load(&ranks[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  const int blk = s_rank; const unsigned long long blocks = (unsigned long long)s_blocks;
  if (blk < 0) return;
  unsigned long long meetings = 0;
  long long t0 = clock64();
  int* mine = data + (size_t)blk * payload; 
  for (int r = 0; r < rounds; ++r) {
    for (int i = threadIdx.x; i < payload; i += blockDim.x) mine[i] = r * 7 + i;     // everyone writes a block of `payload` words ...
    __syncthreads();
    ++meetings;
    if (kFence == 1) __threadfence();
    else if (kFence >= 6) { }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      if (kFence == 6 || kFence == 7) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // one release per workgroup (what the whole workgroup stored is in this CU's write path / this XCD's L2)
      atomicAdd(counter, 1ull);
      const unsigned long long target = meetings * blocks;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (kFence == 1) __threadfence();
    else if (kFence == 3) asm volatile("buffer_inv sc0" ::: "memory");
    else if (kFence == 6) { if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent"); __syncthreads(); }   // one acquire per workgroup: the L1 is the CU's
    else if (kFence == 7) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int* theirs = data + (size_t)((blk + 1) % blocks) * payload;                    // ... and after the barrier reads the neighbour's
    int bad = 0;
    for (int i = threadIdx.x; i < payload; i += blockDim.x) {
      const int v = kFence == 5 ? __hip_atomic_load(&theirs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : theirs[i];
      if (v != r * 7 + i) ++bad;
    }
    if (bad) atomicAdd((unsigned long long*)&ticks[1], (unsigned long long)bad);
    __syncthreads();
    // a second meeting so that nobody overwrites what a neighbour is still reading
    ++meetings;
    if (threadIdx.x == 0) {
      atomicAdd(counter, 1ull);
      const unsigned long long target = meetings * blocks;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  }
  if (blk == 0 && threadIdx.x == 0) { ticks[0] = clock64() - t0; ticks[2] = (long long)blocks; }
}
int main() {
  unsigned long long* counter; int* data; long long* ticks; int* ranks;
  const int payload = 4096;
  hipMalloc(&counter, 8); hipMalloc(&data, (size_t)256 * payload * 4); hipMalloc(&ticks, 32); hipMalloc(&ranks, 8);
  const int rounds = 2000;
  for (int grid : {64, 128, 256}) for (int mode : {1, 3, 5, 6, 7}) {
    hipMemset(counter, 0, 8); hipMemset(ticks, 0, 32); hipMemset(data, 0xff, (size_t)256 * payload * 4); hipMemset(ranks, 0, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int rr = rounds, pl = payload;
    void* args[] = {&counter, &data, &ranks, &rr, &ticks, &pl};
    hipEventRecord(e0, 0);
    hipError_t err = hipLaunchCooperativeKernel(mode == 1 ? (const void*)k_bar<1> : mode == 3 ? (const void*)k_bar<3> : mode == 5 ? (const void*)k_bar<5> : mode == 6 ? (const void*)k_bar<6> : (const void*)k_bar<7>, dim3(grid), dim3(1024), args, 0, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long t[4]; hipMemcpy(t, ticks, 32, hipMemcpyDeviceToHost);
    printf("grid %3d (%s): %2lld workgroups taking part, %s: %.2f us per round of two barriers + %d words each way, stale words %lld\n", grid, hipGetErrorString(err), t[2],
           mode == 1 ? "agent-scope fences          " : mode == 3 ? "s_waitcnt + buffer_inv sc0  " : mode == 5 ? "s_waitcnt, loads through L2 " : mode == 6 ? "ALL XCDs: 1 release + 1 wave acquires" : "ALL XCDs: 1 release + all acquire    ", 1e3 * ms / rounds, payload, t[1]);
  }
  return 0;
}
