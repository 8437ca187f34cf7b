// Cost of a software grid barrier on MI355X (8 XCDs, one L2 each): monotonic counter + spin, as emat_build.hpp's BGrid::sync,
// (a) with the workgroups spread over the XCDs as the dispatcher places them (workgroup i -> XCD i % 8),
// (b) with every participating workgroup on ONE XCD (8 x the grid, only the workgroups with blockIdx % 8 == 0 take part),
// each with the agent-scope fences (L2 write-back / invalidate across XCDs) and, for (b), with workgroup-scope fences + loads that go to L2.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int kFence> __global__ void k_bar(unsigned long long* counter, int* data, int stride, int rounds, long long* ticks) {
  if (blockIdx.x % stride != 0) return;
  const unsigned long long blocks = gridDim.x / stride; const int blk = blockIdx.x / stride;
  unsigned long long meetings = 0;
  long long t0 = clock64();
  for (int r = 0; r < rounds; ++r) {
    // some cross-workgroup traffic: everyone writes a slot, after the barrier reads the neighbour's
    if (threadIdx.x == 0) { if (kFence == 2) __hip_atomic_store(&data[blk], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else data[blk] = r; }
    __syncthreads();
    ++meetings;
    if (threadIdx.x == 0) {
      if (kFence == 1) __threadfence();
      else if (kFence == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      else if (kFence == 3 || kFence == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // L1 is write-through: the stores are in L2 once they are acknowledged
      atomicAdd(counter, 1ull);
      const unsigned long long target = meetings * blocks;
      while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (kFence == 1) __threadfence();
    else if (kFence == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    else if (kFence == 3) asm volatile("buffer_inv sc0" ::: "memory");   // drop this CU's L1: plain loads then see what the XCD's L2 holds
    else if (kFence == 4) asm volatile("buffer_inv sc1" ::: "memory");   // the acquire half of an agent-scope fence alone (no L2 write-back on the other side)
    int v = kFence == 2 ? __hip_atomic_load(&data[(blk + 1) % blocks], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : data[(blk + 1) % blocks];
    if (threadIdx.x == 0 && v != r) atomicAdd((unsigned long long*)&ticks[1], 1ull);   // stale read
  }
  if (blk == 0 && threadIdx.x == 0) ticks[0] = clock64() - t0;
}
int main() {
  unsigned long long* counter; int* data; long long* ticks;
  hipMalloc(&counter, 8); hipMalloc(&data, 4096 * 4); hipMalloc(&ticks, 16);
  const int rounds = 2000;
  for (int blocks : {8, 24, 30}) for (int mode = 0; mode < 5; ++mode) {
    const int stride = mode == 0 ? 1 : 8;
    hipMemset(counter, 0, 8); hipMemset(ticks, 0, 16); hipMemset(data, 0xff, 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    if (mode == 4) hipLaunchKernelGGL(k_bar<4>, dim3(blocks * stride), dim3(1024), 0, 0, counter, data, stride, rounds, ticks);
    else if (mode == 3) hipLaunchKernelGGL(k_bar<3>, dim3(blocks * stride), dim3(1024), 0, 0, counter, data, stride, rounds, ticks);
    else if (mode == 2) hipLaunchKernelGGL(k_bar<2>, dim3(blocks * stride), dim3(1024), 0, 0, counter, data, stride, rounds, ticks);
    else hipLaunchKernelGGL(k_bar<1>, dim3(blocks * stride), dim3(1024), 0, 0, counter, data, stride, rounds, ticks);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    long long t[2]; hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost);
    printf("%2d workgroups of 1024, %s: %.2f us per barrier, stale reads %lld\n", blocks,
           mode == 0 ? "spread over the XCDs, agent-scope fences     " : mode == 1 ? "all on one XCD, agent-scope fences           " : mode == 2 ? "all on one XCD, workgroup fences + L2 loads  " : mode == 3 ? "all on one XCD, waitcnt + buffer_inv sc0     " : "all on one XCD, waitcnt + buffer_inv sc1     ", 1e3 * ms / rounds, t[1]);
  }
  return 0;
}
