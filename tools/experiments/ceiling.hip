// tools/experiments/ceiling.hip -- speed-of-light kernels of the pileup stage's per-op-record decomposition.
// NOT product code: they compute nothing the pileup needs; each one keeps exactly the part of K0 / K1 that ANY design built on
// "one record per CIGAR op, tallied per tile in LDS" has to execute, and drops the bookkeeping around it:
//   ceil_scan      K0's floor: 16-byte loads of four ops per thread, decode, thread-local sums, two DPP wave scans + wave totals
//                  (reference / query advance), one dword stored per thread.  No heads, no geometry, no counting, no emission.
//   ceil_compare   K1's floor: a workgroup per tile with records; 16-byte piece loads (mode 0: perfectly coalesced, piece p = bytes
//                  [16 p, 16 p + 16); mode 1: byte-unaligned pairs of pieces, the way segments lie in the base array), the tile's
//                  reference in LDS read unaligned, XOR + SWAR mismatch word + valid-range table, popcount; NO piece -> record
//                  look-up, NO mismatch loop; then K1's own epilogue (six block scans, 13 coalesced plane stores).
//   ceil_records   the same + phase 1's floor: 8-byte records read coalesced, decoded, two (four with a transcript strand) LDS
//                  range atomics each.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/experiments/ceiling.hip -o gpurun_in/libceil.so ; driver: tools/ceiling.py
#include "../../longcallr_amd/csrc/lcr_dev.h"

#define TSTRIDE (LCR_TILE + 1)
#define REF_PAD 16
#define NPL 15

__global__ void __launch_bounds__(256) k_scan(const uint32_t* __restrict__ cg, uint32_t n_ops, uint32_t* __restrict__ out) {
  __shared__ int ws_ref[4], ws_q[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t j = blockIdx.x * 1024u + 4u * tid;
  uint32_t w[4] = {4u, 4u, 4u, 4u};
  if (j + 4 <= n_ops) { const uint4 v = *reinterpret_cast<const uint4*>(cg + j); w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w; }
  else for (int k = 0; k < 4; k++) if (j + k < n_ops) w[k] = cg[j + k];
  int xr[4], xq[4], tr = 0, tq = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int op = w[k] & 15, len = (int)(w[k] >> 4);
    const bool m = op == 0 || op == 7 || op == 8;
    xr[k] = tr; xq[k] = tq;
    tr += (m || op == 2 || op == 3) ? len : 0;
    tq += (m || op == 1) ? len : 0;
  }
  const int ir = wave_incl_scan(tr), iq = wave_incl_scan(tq);
  if (lane == 63) { ws_ref[wv] = ir; ws_q[wv] = iq; }
  __syncthreads();
  int br = ir - tr, bq = iq - tq;
  for (int i = 0; i < wv; i++) { br += ws_ref[i]; bq += ws_q[i]; }
  out[blockIdx.x * 256 + tid] = (uint32_t)((xr[0] + br) ^ (xq[1] + bq) ^ (xr[2] + br) ^ (xq[3] + bq));
}

template <int RECS>
__global__ void __launch_bounds__(256) k_compare(const uint8_t* __restrict__ bases, int64_t n_bases, int64_t n_pieces, int mode,
                                                  const unsigned long long* __restrict__ recs, int64_t n_recs,
                                                  const uint8_t* __restrict__ ref, int64_t n_cols, uint32_t* __restrict__ planes) {
  __shared__ uint32_t pl[NPL * TSTRIDE];
  __shared__ __attribute__((aligned(16))) uint8_t refl[REF_PAD + LCR_TILE + 32];
  __shared__ uint32_t vlt[17];
  const int tid = threadIdx.x;
  const int64_t gcol0 = ((int64_t)blockIdx.x * LCR_TILE) % (n_cols - LCR_TILE);
  if (tid < 17) {
    uint32_t m = 0;
    for (int j = 0; j < 4; j++) for (int kk = 0; kk < 4; kk++) if (4 * j + kk < tid) m |= 1u << (8 * kk + j);
    vlt[tid] = m;
  }
  for (int i = tid; i < NPL * TSTRIDE; i += 256) pl[i] = 0;
  for (int i = tid; i < REF_PAD + LCR_TILE + 32; i += 256) {
    const int col = i - REF_PAD;
    const uint8_t R = (col >= 0 && col < LCR_TILE) ? ref[gcol0 + col] : 0;
    refl[i] = (R == 'A' || R == 'C' || R == 'G' || R == 'T') ? R : 0xFF;
  }
  __syncthreads();
  const uint32_t* rl32 = reinterpret_cast<const uint32_t*>(refl);
  if (RECS) {   // phase 1's floor: this workgroup's share of the records, two per thread and round, range atomics
    const int64_t per = (n_recs + gridDim.x - 1) / gridDim.x, r0 = per * blockIdx.x, r1 = min(r0 + per, n_recs);
    for (int64_t r = r0 + 2 * tid; r < r1; r += 512) {
      const uint4 v = *reinterpret_cast<const uint4*>(recs + r);   // (two records; the pool is 16-byte aligned)
      const unsigned long long rc[2] = {((unsigned long long)v.y << 32) | v.x, ((unsigned long long)v.w << 32) | v.z};
#pragma unroll
      for (int x = 0; x < 2; x++) {
        const uint32_t hi = (uint32_t)(rc[x] >> 32);
        const int col0 = (int)((hi >> 8) & 255u), len = (int)min((hi >> 18) & 1023u, (uint32_t)(LCR_TILE - 1 - col0)) + 1;
        const int strand = (hi >> 28) & 1, ts = (hi >> 29) & 3;
        uint32_t* dp = pl + strand * TSTRIDE;
        atomicAdd(&dp[col0], 1u); atomicAdd(&dp[col0 + len], 0xFFFFFFFFu);
        if (ts) { uint32_t* tp = pl + (ts == 2 ? 3 : 2) * TSTRIDE; atomicAdd(&tp[col0], 1u); atomicAdd(&tp[col0 + len], 0xFFFFFFFFu); }
      }
    }
  }
  // phase 2's floor: this workgroup's share of the pieces, four in flight per thread
  const int64_t per = (n_pieces + gridDim.x - 1) / gridDim.x, p0 = per * blockIdx.x, p1 = min(p0 + per, n_pieces);
  uint32_t acc = 0;
  auto addr = [&](int64_t p) -> int64_t {
    if (mode == 0) return 16 * p;
    const int64_t s = p >> 1;   // a "segment" of two pieces at an odd byte address
    return min(s * 32 + ((s * 7) & 15) + 16 * (p & 1), n_bases - 16);
  };
  for (int64_t p = p0 + tid; p < p1; p += 4 * 256) {
    uint4 v[4]; int colA[4]; bool ok[4];
#pragma unroll
    for (int x = 0; x < 4; x++) {
      const int64_t q = p + x * 256;
      ok[x] = q < p1;
      colA[x] = (int)((q * 13) & 255);
      v[x] = ok[x] ? *reinterpret_cast<const uint4*>(bases + addr(q)) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int x = 0; x < 4; x++) {
      if (!ok[x]) continue;
      const int ci = colA[x] + REF_PAD, di = ci >> 2;
      const uint32_t sh = (uint32_t)(ci & 3);
      const uint32_t r0 = rl32[di], r1 = rl32[di + 1], r2 = rl32[di + 2], r3 = rl32[di + 3], r4 = rl32[di + 4];
      const uint32_t x0 = v[x].x ^ __builtin_amdgcn_alignbyte(r1, r0, sh), x1 = v[x].y ^ __builtin_amdgcn_alignbyte(r2, r1, sh);
      const uint32_t x2 = v[x].z ^ __builtin_amdgcn_alignbyte(r3, r2, sh), x3 = v[x].w ^ __builtin_amdgcn_alignbyte(r4, r3, sh);
      auto nzf = [](uint32_t y) -> uint32_t { return (y | ((y & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u; };
      uint32_t mm = (nzf(x0) >> 7) | (nzf(x1) >> 6) | (nzf(x2) >> 5) | (nzf(x3) >> 4);
      mm &= vlt[16 - (colA[x] & 1)];
      acc += (uint32_t)__builtin_popcount(mm);
    }
  }
  atomicAdd(&pl[7 * TSTRIDE + tid], acc);
  __syncthreads();
  {   // K1's epilogue: six difference arrays scanned together, 13 planes stored
    __shared__ int wsum5[6][4];
    const int lane = tid & 63, w = tid >> 6;
    int v[6], incl[6];
#pragma unroll
    for (int p = 0; p < 6; p++) {
      v[p] = (int)pl[p * TSTRIDE + tid];
      incl[p] = wave_incl_scan(v[p]);
      if (lane == 63) wsum5[p][w] = incl[p];
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 6; p++) {
      int add = incl[p];
      for (int i = 0; i < w; i++) add += wsum5[p][i];
      pl[p * TSTRIDE + tid] = (uint32_t)add;
    }
    __syncthreads();
  }
  const int col = tid;
  const uint8_t R = refl[REF_PAD + col];
  const int ri = R == 'A' ? 0 : R == 'C' ? 1 : R == 'G' ? 2 : R == 'T' ? 3 : -1;
  uint32_t f[4], rv[4], sf = 0, sr = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { const int hk = k < 2 ? k : 5 - k; f[k] = pl[(7 + hk) * TSTRIDE + col]; rv[k] = pl[(11 + hk) * TSTRIDE + col]; sf += f[k]; sr += rv[k]; }
  if (ri >= 0) {
    const uint32_t mf = pl[0 * TSTRIDE + col] - sf, mr = pl[1 * TSTRIDE + col] - sr;
#pragma unroll
    for (int k = 0; k < 4; k++) if (k == ri) { f[k] = mf; rv[k] = mr; }
  }
  const int64_t o = gcol0 + col;
#pragma unroll
  for (int k = 0; k < 4; k++) { planes[(int64_t)(LCR_PL_A + k) * n_cols + o] = f[k] + rv[k]; planes[(int64_t)(LCR_PL_FWD_A + k) * n_cols + o] = f[k]; }
  planes[(int64_t)LCR_PL_N * n_cols + o] = pl[5 * TSTRIDE + col];
  planes[(int64_t)LCR_PL_D * n_cols + o] = pl[4 * TSTRIDE + col];
  planes[(int64_t)LCR_PL_NI * n_cols + o] = pl[6 * TSTRIDE + col];
  planes[(int64_t)LCR_PL_TS_FWD * n_cols + o] = pl[2 * TSTRIDE + col];
  planes[(int64_t)LCR_PL_TS_REV * n_cols + o] = pl[3 * TSTRIDE + col];
}

template <class F>
static float timed(int reps, F launch) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  launch(); launch();
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a, nullptr);
  for (int i = 0; i < reps; i++) launch();
  (void)hipEventRecord(b, nullptr);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a); (void)hipEventDestroy(b);
  return hipGetLastError() == hipSuccess ? ms / reps : -1.f;
}

extern "C" {
float ceil_scan(const uint32_t* cigar, uint32_t n_ops, uint32_t* out, int reps) {
  const unsigned nb = (n_ops + 1023u) / 1024u;
  return timed(reps, [&] { hipLaunchKernelGGL(k_scan, dim3(nb), dim3(256), 0, nullptr, cigar, n_ops, out); });
}
float ceil_compare(const uint8_t* bases, int64_t n_bases, int64_t n_pieces, int mode, const unsigned long long* recs, int64_t n_recs,
                   const uint8_t* ref, int64_t n_cols, int n_wg, uint32_t* planes, int reps) {
  if (n_recs > 0)
    return timed(reps, [&] { hipLaunchKernelGGL(k_compare<1>, dim3(n_wg), dim3(256), 0, nullptr, bases, n_bases, n_pieces, mode, recs, n_recs, ref, n_cols, planes); });
  return timed(reps, [&] { hipLaunchKernelGGL(k_compare<0>, dim3(n_wg), dim3(256), 0, nullptr, bases, n_bases, n_pieces, mode, recs, n_recs, ref, n_cols, planes); });
}
}
