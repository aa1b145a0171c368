// microbenchmark: agent-scope int64 atomic adds, scattered over an array, from a sub-grid of workgroups
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(long long* a, unsigned n_mask, int per_thread, int stride, int first) {
  if ((int)blockIdx.x % stride != first) return;
  unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < per_thread; i++) {
    x = x * 1664525u + 1013904223u;
    __hip_atomic_fetch_add(&a[(x >> 8) & n_mask], (long long)(i + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
int main() {
  long long* a; hipMalloc(&a, (size_t)(1 << 22) * 8); hipMemset(a, 0, (size_t)(1 << 22) * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int cfg[][4] = {{256, 1, 0, 19}, {256, 4, 0, 19}, {256, 8, 0, 19}, {256, 1, 0, 12}, {256, 4, 0, 12}, {1024, 1, 0, 19}, {1024, 4, 0, 19}};
  for (auto& c : cfg) {
    const int blocks = c[0], stride = c[1], first = c[2]; const unsigned mask = (1u << c[3]) - 1;
    const int per = 256;
    for (int rep = 0; rep < 3; rep++) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(1024), 0, 0, a, mask, per, stride, first);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double n = (double)(blocks / stride) * 1024 * per;
      if (rep == 2) printf("blocks %4d stride %d addresses 2^%d : %.1f M atomics in %.3f ms = %.1f G/s\n", blocks, stride, c[3], n / 1e6, ms, n / ms / 1e6);
    }
  }
  return 0;
}
