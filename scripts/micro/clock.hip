// Effective shader clock under load: ratio of s_memtime (core clock) to the constant 100 MHz wall clock, measured by every
// wave of a full-chip launch of dependent f64 work (one active lane per wave, like k_run_moves).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(double* out, double* mhz, int iters) {
  if (threadIdx.x != 0) return;
  double x = 1.0 + blockIdx.x * 1e-9;
  long long c0 = clock64(); unsigned long long w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;
  long long c1 = clock64(); unsigned long long w1 = wall_clock64();
  out[blockIdx.x] = x;
  mhz[blockIdx.x] = (double)(c1 - c0) / ((double)(w1 - w0) / 100.0);
}
int main() {
  for (int blocks : {256, 4096, 16384}) {
    double *out, *mhz; hipMalloc(&out, blocks * 8); hipMalloc(&mhz, blocks * 8);
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, mhz, 2000000); hipDeviceSynchronize(); }
    std::vector<double> h(blocks); hipMemcpy(h.data(), mhz, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0, lo = 1e9, hi = 0; for (double v : h) { s += v; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    printf("%5d single-lane waves: core clock %.0f MHz (min %.0f, max %.0f)\n", blocks, s / blocks, lo, hi);
    hipFree(out); hipFree(mhz);
  }
  return 0;
}
