// micro-benchmark of the grid barrier used by k4_grid.hip (tools only, not part of liblcr)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Ctl { unsigned arrive, gen; };
template <int MODE>
__global__ void __launch_bounds__(1024) bar(Ctl* c, int n, int* data, int m) {
  unsigned gen = 0;
  long long acc = 0;
  for (int it = 0; it < n; it++) {
    if (m) for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) acc += data[i];
    __syncthreads();
    if (threadIdx.x == 0) {
      if (MODE == 0) __threadfence();
      if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (atomicAdd(&c->arrive, 1u) == gridDim.x - 1) {
        __hip_atomic_store(&c->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&c->gen, gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        while (__hip_atomic_load(&c->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
      }
      if (MODE == 0) __threadfence();
      if (MODE == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    gen++;
    __syncthreads();
  }
  if (acc == 12345) data[0] = 1;
}
int main() {
  Ctl* c; hipMalloc(&c, sizeof(Ctl));
  int* d; const int M = 8 << 20; hipMalloc(&d, M * 4); hipMemset(d, 0, M * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int blocks : {256, 512}) for (int threads : {256, 1024}) for (int mode = 0; mode < 3; mode++) for (int m : {0, 1 << 20, 8 << 20}) {
    hipMemset(c, 0, sizeof(Ctl));
    const int n = 2000;
    hipEventRecord(a);
    if (mode == 0) bar<0><<<blocks, threads>>>(c, n, d, m);
    if (mode == 1) bar<1><<<blocks, threads>>>(c, n, d, m);
    if (mode == 2) bar<2><<<blocks, threads>>>(c, n, d, m);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("blocks %d threads %d mode %d (0 threadfence, 1 none, 2 rel/acq fences) read %d MB/iter: %.2f us per barrier\n", blocks, threads, mode, m * 4 >> 20, ms * 1000 / n);
  }
  return 0;
}
