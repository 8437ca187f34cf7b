// What a call to a NON-LEAF device function costs on gfx950 when one lane of the wave is active, against all 64: a non-leaf
// function parks its return address in a VGPR lane (v_writelane) and must preserve that VGPR's INACTIVE lanes
// (s_xor_saveexec + scratch_store in the prologue, scratch_load + s_waitcnt vmcnt(0) before s_setpc in the epilogue).
// Also: an LDS store of the same address from 1 lane against 64 lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __attribute__((noinline)) double leaf(double x) { return x * 1.0000001 + 0.5; }
__device__ __attribute__((noinline)) double nonleaf(double x) { return leaf(x) + 1.0; }
__device__ __attribute__((noinline)) double nonleaf2(double x) { return nonleaf(x) * 0.999; }
__global__ void k_wwm(double* out, long long* ticks, int n_iter, int all_lanes) {
  __shared__ double cell[64];
  const int lane = threadIdx.x;
  cell[lane] = 0.0;
  __syncthreads();
  double x = 1.0;
  long long t[6] = {0, 0, 0, 0, 0, 0};
  if (all_lanes || lane == 0) {
    long long t0 = clock64();
    for (int i = 0; i < n_iter; ++i) x = leaf(x);
    long long t1 = clock64();
    for (int i = 0; i < n_iter; ++i) x = nonleaf(x);
    long long t2 = clock64();
    for (int i = 0; i < n_iter; ++i) x = nonleaf2(x);
    long long t3 = clock64();
    for (int i = 0; i < n_iter; ++i) { cell[3] = x; x = cell[3] + 1.0; }   // LDS store + dependent load, same address on every active lane
    long long t4 = clock64();
    t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3;
  }
  if (lane == 0) { out[blockIdx.x] = x; if (blockIdx.x == 0) for (int i = 0; i < 4; ++i) ticks[i] = t[i]; }
}
int main() {
  double* out; long long* ticks;
  hipMalloc(&out, 1 << 20); hipMalloc(&ticks, 64);
  const int n = 4096;
  for (int blocks : {1, 4096}) for (int all = 0; all < 2; ++all) {
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_wwm, dim3(blocks), dim3(64), 0, 0, out, ticks, n, all); hipDeviceSynchronize(); }
    long long t[4]; hipMemcpy(t, ticks, 32, hipMemcpyDeviceToHost);
    printf("%4d waves, %s: cycles per call: leaf %.1f | non-leaf (1 level) %.1f | non-leaf (2 levels) %.1f | LDS store+load %.1f\n", blocks, all ? "all lanes" : "lane 0   ",
           (double)t[0] / n, (double)t[1] / n, (double)t[2] / n, (double)t[3] / n);
  }
  return 0;
}
