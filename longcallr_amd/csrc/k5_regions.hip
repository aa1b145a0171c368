// k5_regions.hip — region discovery on gfx950.
//
// Replaces find_isolated_regions_with_depth (reference src/util.rs:236-332, truncation off): the
// per-contig depth vector (+1 per reference position of every read span, introns and deletions
// included, util.rs:281-285) as a difference array + prefix scan, and the split into coverage islands
// as an ordered compaction of the positions where depth switches between 0 and > 0.
#include "lcr_dev.h"

// the part of the contig any read covers: out[0] = min start, out[1] = max end over the valid spans (out preset to INT_MAX, 0).
// The dense passes below run over that window only: a file with 1 700 reads on 13 kb of a 64 Mb contig (demo.bam) no longer
// clears, scans and compacts 64 M positions.
__global__ void __launch_bounds__(LCR_BLOCK)
k5_span_window(const int32_t* __restrict__ ref_start, const int32_t* __restrict__ ref_end, int32_t n, int64_t contig_len, int32_t* __restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  int lo = INT_MAX, hi = 0;
  if (r < n) {
    const int64_t s = ref_start[r];
    const int64_t e = min((int64_t)ref_end[r], contig_len);
    if (s >= 0 && s < e) { lo = (int)s; hi = (int)e; }
  }
  for (int d = 32; d >= 1; d >>= 1) { lo = min(lo, __shfl_xor(lo, d, 64)); hi = max(hi, __shfl_xor(hi, d, 64)); }
  if ((threadIdx.x & 63) == 0 && hi > 0) { atomicMin(&out[0], lo); atomicMax(&out[1], hi); }
}
void launch_k5_span_window(const int32_t* ref_start, const int32_t* ref_end, int32_t n, int64_t contig_len, int32_t* out, hipStream_t s) {
  if (n > 0) hipLaunchKernelGGL(k5_span_window, dim3((n + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, ref_start, ref_end, n, contig_len, out);
}

// diff is indexed from `lo` (the window's first position)
__global__ void __launch_bounds__(LCR_BLOCK)
k5_span_diff(const int32_t* __restrict__ ref_start, const int32_t* __restrict__ ref_end, int32_t n, int64_t contig_len, int64_t lo,
             uint32_t* __restrict__ diff) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int64_t s = ref_start[r];
  int64_t e = ref_end[r];
  if (e > contig_len) e = contig_len;
  if (s < 0 || s >= e) return;
  atomicAdd(&diff[s - lo], 1u);
  atomicAdd(&diff[e - lo], 0xFFFFFFFFu);
}

// depth[i] = ex[i + 1] (ex = exclusive scan of diff).  One block per 1024 positions counts the island
// starts / ends it contains; after the scan of the counts the same walk writes them in order.
template <bool WRITE>
__global__ void __launch_bounds__(LCR_BLOCK)
k5_bounds(const int32_t* __restrict__ ex, int64_t contig_len, int32_t* __restrict__ blk_cnt, const int32_t* __restrict__ blk_off,
          int32_t* __restrict__ starts, int32_t* __restrict__ ends) {
  __shared__ int wsum_s[LCR_BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int64_t base = (int64_t)blockIdx.x * 1024 + (int64_t)tid * 4;
  int fs[4], fe[4], cs = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int64_t i = base + k;
    fs[k] = fe[k] = 0;
    if (i < contig_len) {
      const int d = ex[i + 1];
      const int dp = i > 0 ? ex[i] : 0;                       // depth[i-1]
      const int dn = i + 1 < contig_len ? ex[i + 2] : 0;      // depth[i+1]
      fs[k] = d > 0 && dp == 0;
      fe[k] = d > 0 && dn == 0;
    }
    cs += fs[k];
  }
  // block prefix of the start counts
  int is = cs;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int ts = __shfl_up(is, d, 64);
    if (lane >= d) is += ts;
  }
  if (lane == 63) wsum_s[w] = is;
  __syncthreads();
  int as = 0;
  for (int k = 0; k < w; k++) as += wsum_s[k];
  if (!WRITE) {
    if (tid == LCR_BLOCK - 1) blk_cnt[blockIdx.x] = is + as;   // island starts in this block
    return;
  }
  // islands are numbered by their start; an end at position i closes island number (#starts at or before i) - 1
  int rs = blk_off[blockIdx.x] + as + is - cs;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (fs[k]) starts[rs] = (int32_t)(base + k);
    rs += fs[k];
    if (fe[k]) ends[rs - 1] = (int32_t)(base + k);
  }
}

// max depth of every island: one block per island
__global__ void __launch_bounds__(LCR_BLOCK)
k5_island_max(const int32_t* __restrict__ ex, const int32_t* __restrict__ starts, const int32_t* __restrict__ ends,
              uint32_t* __restrict__ maxcov) {
  __shared__ int red[LCR_BLOCK / 64];
  const int isl = blockIdx.x;
  const int s = starts[isl], e = ends[isl];
  int m = 0;
  for (int i = s + (int)threadIdx.x; i <= e; i += blockDim.x) m = max(m, ex[i + 1]);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = max(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) { for (int k = 1; k < LCR_BLOCK / 64; k++) m = max(m, red[k]); maxcov[isl] = (uint32_t)m; }
}

void launch_k5_span_diff(const int32_t* ref_start, const int32_t* ref_end, int32_t n, int64_t contig_len, int64_t lo, uint32_t* diff, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k5_span_diff, dim3((n + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, ref_start, ref_end, n, contig_len, lo, diff);
}
void launch_k5_bounds(bool write, const int32_t* ex, int64_t contig_len, int32_t n_blocks, int32_t* blk_cnt, const int32_t* blk_off,
                      int32_t* starts, int32_t* ends, hipStream_t s) {
  if (n_blocks == 0) return;
  if (write) hipLaunchKernelGGL(k5_bounds<true>, dim3(n_blocks), dim3(LCR_BLOCK), 0, s, ex, contig_len, blk_cnt, blk_off, starts, ends);
  else hipLaunchKernelGGL(k5_bounds<false>, dim3(n_blocks), dim3(LCR_BLOCK), 0, s, ex, contig_len, blk_cnt, blk_off, starts, ends);
}
void launch_k5_island_max(const int32_t* ex, const int32_t* starts, const int32_t* ends, int32_t n_islands, uint32_t* maxcov, hipStream_t s) {
  if (n_islands == 0) return;
  hipLaunchKernelGGL(k5_island_max, dim3(n_islands), dim3(LCR_BLOCK), 0, s, ex, starts, ends, maxcov);
}
