// micro-benchmark / visibility check of the grid barriers used by k4_grid.hip (tools only, not part of liblcr)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Ctl { unsigned arrive, gen, pad[14]; unsigned garr[32 * 16]; };
// MODE 0: __threadfence both sides, flat arrival; 1: no fences, flat; 3: no fences, two-level arrival
template <int MODE>
__device__ __forceinline__ void barrier(Ctl* c, unsigned& gen) {
  __syncthreads();
  if (threadIdx.x == 0) {
    if (MODE == 0) __threadfence();
    bool last;
    if (MODE == 3) {
      const unsigned grp = blockIdx.x >> 4, ngrp = (gridDim.x + 15) >> 4;
      const unsigned members = min(16u, gridDim.x - grp * 16);
      last = false;
      if (atomicAdd(&c->garr[grp * 16], 1u) == members - 1) {
        __hip_atomic_store(&c->garr[grp * 16], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = atomicAdd(&c->arrive, 1u) == ngrp - 1;
      }
    } else last = atomicAdd(&c->arrive, 1u) == gridDim.x - 1;
    if (last) {
      __hip_atomic_store(&c->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&c->gen, gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (__hip_atomic_load(&c->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
    }
    if (MODE == 0) __threadfence();
  }
  gen++;
  __syncthreads();
}
// every thread publishes a value with a device-coherent store, crosses the barrier, and reads another workgroup's
// value with a device-coherent load (MODE != 0) or plain accesses (MODE 0, fenced barrier)
template <int MODE>
__global__ void __launch_bounds__(1024) vis(Ctl* c, int n, unsigned* data, unsigned* bad, const int* big, int m) {
  unsigned gen = 0;
  const unsigned nt = gridDim.x * blockDim.x, tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned errs = 0;
  long long acc = 0;
  for (int it = 1; it <= n; it++) {
    if (MODE == 0) data[tid] = it; else __hip_atomic_store(&data[tid], (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (m) for (int i = tid; i < m; i += nt) acc += big[i];   // immutable data: stays in L2 across light barriers
    barrier<MODE>(c, gen);
    const unsigned src = (tid + 1024u * 37u + 17u) % nt;
    const unsigned v = MODE == 0 ? data[src] : __hip_atomic_load(&data[src], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != (unsigned)it) errs++;
    barrier<MODE>(c, gen);
  }
  if (errs) atomicAdd(bad, errs);
  if (acc == 12345) data[0] = 1;
}
int main() {
  Ctl* c; hipMalloc(&c, sizeof(Ctl));
  unsigned* d; hipMalloc(&d, 1 << 22); unsigned* bad; hipMalloc(&bad, 4);
  int* big; const int M = 8 << 20; hipMalloc(&big, M * 4); hipMemset(big, 0, M * 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int blocks : {256}) for (int mode : {0, 1, 3}) for (int m : {0, 8 << 20}) {
    hipMemset(c, 0, sizeof(Ctl)); hipMemset(bad, 0, 4); hipMemset(d, 0, 1 << 22);
    const int n = 1000;
    hipEventRecord(a);
    if (mode == 0) vis<0><<<blocks, 1024>>>(c, n, d, bad, big, m);
    if (mode == 1) vis<1><<<blocks, 1024>>>(c, n, d, bad, big, m);
    if (mode == 3) vis<3><<<blocks, 1024>>>(c, n, d, bad, big, m);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned hb = 0; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
    printf("blocks %d mode %d (0 fenced+plain, 1 light+coherent, 3 light two-level+coherent) read %d MB/iter: %.2f us per barrier, %u stale reads\n",
           blocks, mode, m * 4 >> 20, ms * 1000 / (2 * n), hb);
  }
  return 0;
}
