// Which counters see a FLAT store whose generic address falls in LDS, and which see one that falls in global memory:
// run under `rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_FLAT TCP_TCC_WRITE_REQ_sum SQ_INSTS_LDS` and compare the two kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_flat_to_lds(double* g, int n) {
  __shared__ double cell[256];
  double* volatile pv = cell; double* p = pv;   // generic pointer the compiler cannot see through
  if (threadIdx.x == 0) for (int i = 0; i < n; ++i) p[i & 255] = (double)i;
  __syncthreads();
  if (threadIdx.x == 0) g[0] = cell[7];
}
__global__ void k_flat_to_global(double* g, int n) {
  double* volatile pv = g + 256; double* p = pv;
  if (threadIdx.x == 0) for (int i = 0; i < n; ++i) p[i & 255] = (double)i;
}
int main() {
  double* g; if (hipMalloc(&g, 1 << 16) != hipSuccess) return 1;
  hipLaunchKernelGGL(k_flat_to_lds, dim3(1), dim3(64), 0, 0, g, 100000);
  hipLaunchKernelGGL(k_flat_to_global, dim3(1), dim3(64), 0, 0, g, 100000);
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  printf("done: 100000 single-lane flat stores in each kernel\n");
  return 0;
}
