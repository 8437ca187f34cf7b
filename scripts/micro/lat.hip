// Latency of a dependent load chain on gfx950, one active lane: ds_read vs flat load hitting LDS vs global (L2-resident) vs private scratch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_lat(int* out, long long* ticks, int* gchain, int n_iter) {
  __shared__ int chain[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) chain[i] = (i * 17 + 5) & 1023;
  __syncthreads();
  if (threadIdx.x != 0) return;
  int j = 0;
  long long t0 = clock64();
  for (int i = 0; i < n_iter; ++i) j = chain[j];                       // ds_read
  long long t1 = clock64();
  int* volatile gen_v = chain; int* gen = gen_v;                        // generic pointer the compiler cannot see through
  int j2 = j & 1023;
  for (int i = 0; i < n_iter; ++i) j2 = gen[j2];                        // flat_load -> LDS
  long long t2 = clock64();
  int j3 = j2 & 1023;
  for (int i = 0; i < n_iter; ++i) j3 = gchain[j3];                     // global_load (L1/L2 hits)
  long long t3 = clock64();
  int* volatile gg_v = gchain; int* gg = gg_v;
  int j4 = j3 & 1023;
  for (int i = 0; i < n_iter; ++i) j4 = gg[j4];                         // flat_load -> global
  long long t4 = clock64();
  out[blockIdx.x] = j4;
  ticks[0] = t1 - t0; ticks[1] = t2 - t1; ticks[2] = t3 - t2; ticks[3] = t4 - t3;
}
int main() {
  int *out, *gchain; long long* ticks;
  hipMalloc(&out, 4096); hipMalloc(&ticks, 64); hipMalloc(&gchain, 4096);
  std::vector<int> h(1024); for (int i = 0; i < 1024; ++i) h[i] = (i * 17 + 5) & 1023;
  hipMemcpy(gchain, h.data(), 4096, hipMemcpyHostToDevice);
  const int n = 4096;
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k_lat, dim3(1), dim3(64), 0, 0, out, ticks, gchain, n); hipDeviceSynchronize(); }
  long long t[4]; hipMemcpy(t, ticks, 32, hipMemcpyDeviceToHost);
  printf("cycles per dependent load (one lane, idle GPU): ds_read %.1f | flat->LDS %.1f | global_load %.1f | flat->global %.1f\n", (double)t[0] / n, (double)t[1] / n, (double)t[2] / n, (double)t[3] / n);
  return 0;
}
