// Latency of the f64 transcendentals and of plain dependent f64 arithmetic on gfx950 with ONE active lane (what a chain pays):
// a dependent chain of n calls each, in clock64 cycles per call, inline and behind a noinline call as the engine has them.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __attribute__((noinline)) double n_log(double x) { return ::log(x); }
__device__ __attribute__((noinline)) double n_exp(double x) { return ::exp(x); }
__device__ __attribute__((noinline)) double n_log1p(double x) { return ::log1p(x); }
__device__ __attribute__((noinline)) double n_expm1(double x) { return ::expm1(x); }
__global__ void k(double* out, long long* ticks, int n, double seed) {
  if (threadIdx.x != 0) return;
  double x = seed; long long t[12]; int k = 0;
  t[k++] = clock64(); for (int i = 0; i < n; ++i) x = ::log(x + 3.0);
  t[k++] = clock64(); for (int i = 0; i < n; ++i) x = ::exp(x * 0.25);
  t[k++] = clock64(); for (int i = 0; i < n; ++i) x = ::log1p(x * 0.5);
  t[k++] = clock64(); for (int i = 0; i < n; ++i) x = ::expm1(x * 0.25) + 0.3;
  t[k++] = clock64(); for (int i = 0; i < n; ++i) x = n_log(x + 3.0);
  t[k++] = clock64(); for (int i = 0; i < n; ++i) x = n_exp(x * 0.25);
  t[k++] = clock64(); for (int i = 0; i < n; ++i) x = x * 1.0000001 + 0.5;          // one dependent fma
  t[k++] = clock64(); for (int i = 0; i < n; ++i) x = 1.0 / (x + 1.5);              // a division
  t[k++] = clock64(); for (int i = 0; i < n; ++i) x = sqrt(x + 2.0);
  t[k++] = clock64();
  out[0] = x;
  for (int i = 0; i + 1 < k; ++i) ticks[i] = t[i + 1] - t[i];
}
int main() {
  double* out; long long* ticks; hipMalloc(&out, 64); hipMalloc(&ticks, 128);
  const int n = 2000;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, ticks, n, 1.2345); hipDeviceSynchronize(); }
  long long t[9]; hipMemcpy(t, ticks, sizeof t, hipMemcpyDeviceToHost);
  const char* names[9] = {"log", "exp", "log1p", "expm1", "log (call)", "exp (call)", "fma", "div", "sqrt"};
  printf("cycles per dependent call, one lane, idle GPU:");
  for (int i = 0; i < 9; ++i) printf(" %s %.0f |", names[i], (double)t[i] / n);
  printf("\n");
  return 0;
}
