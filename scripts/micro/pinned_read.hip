// How fast does the CPU read page-locked host memory?  hipHostMalloc (default / non-coherent / write-combined flags) against
// malloc + hipHostRegister and plain malloc: random byte reads and a sequential copy-out of 1.6 MB, after the GPU wrote the buffer.
//   hipcc -O2 -o pinned_read pinned_read.hip && ./pinned_read
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void fill(int* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = i * 7; }
static void probe(const char* name, int* host, int* dev, size_t n, hipStream_t s) {
  hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dev, (int)n);
  hipMemcpyAsync(host, dev, n * sizeof(int), hipMemcpyDeviceToHost, s);
  hipStreamSynchronize(s);
  unsigned long long sum = 0; unsigned x = 12345;
  double t0 = now();
  for (int k = 0; k < 200000; ++k) { x = x * 1664525u + 1013904223u; sum += (unsigned)host[x % n]; }
  double t1 = now();
  std::vector<int> out(n);
  double t2 = now();
  std::memcpy(out.data(), host, n * sizeof(int));
  double t3 = now();
  for (size_t i = 0; i < n; i += 16) sum += (unsigned)out[i];
  printf("%-44s random read %.1f ns each | sequential copy-out of %.1f MB %.0f us (%.2f GB/s) | %llu\n", name, (t1 - t0) * 1e3 / 200000, n * 4 / 1e6, t3 - t2, n * 4 / (t3 - t2) / 1e3, sum);
}
int main() {
  const size_t n = 400000;
  hipStream_t s; hipStreamCreate(&s);
  int* dev; hipMalloc((void**)&dev, n * sizeof(int));
  int* a; hipHostMalloc((void**)&a, n * sizeof(int), hipHostMallocDefault); probe("hipHostMalloc(default)", a, dev, n, s);
  int* b; hipHostMalloc((void**)&b, n * sizeof(int), hipHostMallocNonCoherent); probe("hipHostMalloc(NonCoherent)", b, dev, n, s);
  int* c; if (hipHostMalloc((void**)&c, n * sizeof(int), hipHostMallocCoherent) == hipSuccess) probe("hipHostMalloc(Coherent)", c, dev, n, s);
  int* d = (int*)aligned_alloc(4096, (n * sizeof(int) + 4095) & ~(size_t)4095); memset(d, 0, n * sizeof(int));
  if (hipHostRegister(d, n * sizeof(int), hipHostRegisterDefault) == hipSuccess) probe("malloc + hipHostRegister", d, dev, n, s); else printf("hipHostRegister failed\n");
  int* e = (int*)malloc(n * sizeof(int)); memset(e, 0, n * sizeof(int)); probe("malloc (pageable)", e, dev, n, s);
  return 0;
}
