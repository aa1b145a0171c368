// k1_pileup.hip — K0 (read spans) and K1 (pileup tally) for gfx950.
//
// K1 replaces Profile::fill_data_into_freq_vec (reference src/util.rs:621-949).
//
// Design: column-tile gather.  One workgroup owns LCR_TILE consecutive pileup
// columns of one region and keeps every counter of those columns in LDS; each wave64 takes reads
// that overlap the tile, scans their CIGAR 64 ops at a time (wave prefix sums give every op's
// reference/query start) and
//   * turns D / N runs and whole M blocks into +1/-1 *difference-array* updates (2 LDS atomics per
//     op instead of one per base; prefix-scanned once at the end of the tile),
//   * streams the aligned read bases against the tile's reference bytes held in LDS: a base that
//     equals the reference byte and is not near a read end needs no further work (its count is
//     "depth - mismatches"); only mismatching, masked (poly-A / homopolymer / ONT end-trim) and
//     non-ACGT bases — a few % of the stream — touch per-column counters.
// All arithmetic is u32 adds => results are bit-exact regardless of order.  HBM traffic is the
// read bases once (each base belongs to exactly one tile), the CIGAR words of overlapping reads,
// and one coalesced write of the 13 count planes.
#include "lcr_dev.h"

// ---------------------------------------------------------------------------------------------
// K0: per read reference span + per region max span; validates CIGAR ops.
__global__ void __launch_bounds__(LCR_BLOCK) k0_spans(BatchView b) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= b.n_reads) return;
  const uint32_t* cg = b.cigar + b.cig_off[r];
  const uint32_t n = b.n_cig[r];
  int32_t span = 0;
  bool bad = false;
  for (uint32_t i = 0; i < n; i++) {
    uint32_t op = cg[i] & 15u, len = cg[i] >> 4;
    // M,=,X,D,N consume reference; I,S,H do not; anything else is the reference's panic branch
    if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += (int32_t)len;
    else if (!(op == 1 || op == 4 || op == 5)) bad = true;
  }
  b.ref_end[r] = b.pos[r] + span;
  if (bad) atomicExch(b.error_flag, 1);
  int g = region_of_read(b.read_begin, b.n_regions, r);
  atomicMax(&b.region_max_span[g], span);
}

void launch_k0_spans(const BatchView& b, hipStream_t s) {
  if (b.n_reads == 0) return;
  hipLaunchKernelGGL(k0_spans, dim3((b.n_reads + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, b);
}

// ---------------------------------------------------------------------------------------------
// LDS planes of one tile (u32 each, LCR_TILE + 1 entries so that "end" markers at tile_len fit)
enum {
  P_DIFF_DEPTH_F = 0,  // difference array: kept aligned bases of forward reads
  P_DIFF_DEPTH_R,      //                   ... of reverse reads
  P_DIFF_TS0,          // difference array: transcript_strands[0]
  P_DIFF_TS1,          //                   transcript_strands[1]
  P_DIFF_N,            // intron runs
  P_DIFF_D,            // deletion runs
  P_NI,                // insertions (plain counter)
  P_MM_F,              // 4 planes: mismatching base counts A,C,G,T of forward reads
  P_MM_R = P_MM_F + 4, // 4 planes: ... of reverse reads
  P_NPL = P_MM_R + 4
};
#define TSTRIDE (LCR_TILE + 1)

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

// inclusive block scan (256 threads) of one int per thread; returns inclusive value
__device__ __forceinline__ int block_incl_scan(int v, int* wsum /* 4 ints of LDS */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int s = wave_incl_scan(v, lane);
  if (lane == 63) wsum[w] = s;
  __syncthreads();
  int add = 0;
  for (int i = 0; i < w; i++) add += wsum[i];
  __syncthreads();
  return s + add;
}

__global__ void __launch_bounds__(LCR_BLOCK)
k1_pileup(BatchView b, DevParams prm, const int32_t* __restrict__ tile_region, const int32_t* __restrict__ tile_col0,
          int64_t n_cols, uint32_t* __restrict__ planes) {
  __shared__ uint32_t pl[P_NPL * TSTRIDE];
  __shared__ uint8_t refl[LCR_TILE];
  __shared__ int wsum[4];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = tile_region[blockIdx.x];
  const int tc0 = tile_col0[blockIdx.x];               // first column of the tile inside the region
  const int vec = b.len[g];
  const int tlen = min(LCR_TILE, vec - tc0);
  const int64_t start0 = b.start0[g];
  const int64_t gcol0 = b.col_off[g] + tc0;            // global column of tile column 0

  for (int i = tid; i < P_NPL * TSTRIDE; i += LCR_BLOCK) pl[i] = 0;
  for (int i = tid; i < LCR_TILE; i += LCR_BLOCK) {
    uint8_t R = i < tlen ? b.ref[gcol0 + i] : 0;
    // only upper-case ACGT can equal a read base (htslib decodes to upper case); anything else is
    // stored as 0xFF so that every base at such a column takes the explicit-count path
    refl[i] = (R == 'A' || R == 'C' || R == 'G' || R == 'T') ? R : 0xFF;
  }
  __syncthreads();

  // reads of region g that can overlap tile columns [tc0, tc0+tlen): pos < t1 and ref_end > t0
  const int rb = b.read_begin[g], re = b.read_begin[g + 1];
  const int64_t t0 = start0 + tc0, t1 = t0 + tlen;     // absolute reference interval of the tile
  const int64_t lo_pos = t0 - (int64_t)b.region_max_span[g];
  int r_lo, r_hi;
  {
    int lo = rb, hi = re;  // first read with pos >= lo_pos
    while (lo < hi) { int mid = (lo + hi) >> 1; if ((int64_t)b.pos[mid] >= lo_pos) hi = mid; else lo = mid + 1; }
    r_lo = lo;
    lo = rb; hi = re;      // first read with pos > t1 (a read starting at t1 with a leading insertion
                           // still adds `ni` to the tile's last column)
    while (lo < hi) { int mid = (lo + hi) >> 1; if ((int64_t)b.pos[mid] > t1) hi = mid; else lo = mid + 1; }
    r_hi = lo;
  }

  const int D = prm.dist_to_end, L = prm.polya_len;

  for (int r = r_lo + wave; r < r_hi; r += LCR_BLOCK / 64) {
    const int rend = b.ref_end[r];
    if ((int64_t)rend < t0) continue;
    const int rpos = b.pos[r];
    const uint32_t ncig = b.n_cig[r];
    const uint32_t* __restrict__ cg = b.cigar + b.cig_off[r];
    const uint8_t* __restrict__ seq = b.bases + b.seq_off[r];
    const int seq_len = b.seq_len[r];
    const int lead = b.lead[r];
    const int reb = seq_len - b.trail[r];
    const uint8_t fl = b.flags[r];
    const int strand = fl & 1;
    const int ts = (fl >> 1) & 3;
    // transcript_strands index (util.rs:803-819): (+,+)->0 (+,-)->1 (-,+)->1 (-,-)->0, none -> -1
    const int tsidx = ts == 0 ? -1 : ((strand == 0) == (ts == 1) ? 0 : 1);
    uint32_t* depth_pl = pl + (strand ? P_DIFF_DEPTH_R : P_DIFF_DEPTH_F) * TSTRIDE;
    uint32_t* ts_pl = pl + (tsidx == 1 ? P_DIFF_TS1 : P_DIFF_TS0) * TSTRIDE;
    uint32_t* mm_pl = pl + (strand ? P_MM_R : P_MM_F) * TSTRIDE;

    int ref_cur = rpos - (int)(start0 + tc0);  // tile-relative column of the next reference base
    int q_cur = lead > 0 ? lead : 0;           // util.rs:686-690
    for (uint32_t c0 = 0; c0 < ncig; c0 += 64) {
      if (ref_cur > tlen) break;               // nothing further right contributes (an insertion that
                                               // starts exactly at column tlen still counts on tlen-1)
      uint32_t word = (c0 + lane < ncig) ? cg[c0 + lane] : 0u;
      const int op = word & 15, len = (int)(word >> 4);
      const bool is_m = (op == 0 || op == 7 || op == 8) && len > 0;
      const bool is_dn = (op == 2 || op == 3) && len > 0;
      const int dr = (is_m || is_dn) ? len : 0;
      const int dq = (is_m || op == 1) ? len : 0;
      const int ir = wave_incl_scan(dr, lane), iq = wave_incl_scan(dq, lane);
      const int rs = ref_cur + ir - dr;        // tile-relative column where this op starts
      const int qs = q_cur + iq - dq;          // read offset where this op starts
      ref_cur += __shfl(ir, 63, 64);
      q_cur += __shfl(iq, 63, 64);

      // clip the op's column range to the tile
      int a = max(rs, 0), e = min(rs + len, tlen);
      if (is_dn && e > a) {  // util.rs:905-917 (D) / 930-942 (N): +1 per reference position
        uint32_t* dp = pl + (op == 3 ? P_DIFF_N : P_DIFF_D) * TSTRIDE;
        atomicAdd(&dp[a], 1u);
        atomicAdd(&dp[e], 0xFFFFFFFFu);
      }
      if (op == 1 && len > 0) {  // util.rs:918-929: counted on the previous column, 1 <= p < vec
        const int p = rs + tc0;  // pos_in_freq_vec
        if (p >= 1 && p < vec && rs - 1 >= 0 && rs - 1 < tlen) atomicAdd(&pl[P_NI * TSTRIDE + rs - 1], 1u);
      }
      const bool m_hit = is_m && e > a;
      if (m_hit) {  // whole block as a range update; per-base corrections follow below
        atomicAdd(&depth_pl[a], 1u);
        atomicAdd(&depth_pl[e], 0xFFFFFFFFu);
        if (tsidx >= 0) { atomicAdd(&ts_pl[a], 1u); atomicAdd(&ts_pl[e], 0xFFFFFFFFu); }
      }
      // stream the bases of every M block that intersects the tile
      unsigned long long mmask = __ballot(m_hit);
      while (mmask) {
        const int j = __ffsll((long long)mmask) - 1;
        mmask &= mmask - 1;
        const int ja = __shfl(a, j, 64), je = __shfl(e, j, 64);
        const int jq = __shfl(qs, j, 64) - __shfl(rs, j, 64);  // read offset = column + jq
        for (int col = ja + lane; col < je; col += 64) {
          const int c = col + jq;
          const uint8_t base = seq[c];
          const uint8_t R = refl[col];
          const bool zone = in_end_zone(c, lead, reb, D);
          if (base == R && !zone) continue;  // fast path: plain reference match
          bool masked = false;
          if (zone) masked = prm.ont ? true : polya_masked(seq, seq_len, c, L, b.ref[gcol0 + col]);
          if (masked) {  // contributes nothing (util.rs:801): undo the range update at this column
            atomicAdd(&depth_pl[col], 0xFFFFFFFFu);
            atomicAdd(&depth_pl[col + 1], 1u);
            if (tsidx >= 0) { atomicAdd(&ts_pl[col], 0xFFFFFFFFu); atomicAdd(&ts_pl[col + 1], 1u); }
          } else if (base != R) {
            const int bi = base_code(base);
            if (bi >= 0) atomicAdd(&mm_pl[bi * TSTRIDE + col], 1u);
            else {  // "Invalid nucleotide base" (util.rs:890-892): no allele count, ts still counted
              atomicAdd(&depth_pl[col], 0xFFFFFFFFu);
              atomicAdd(&depth_pl[col + 1], 1u);
            }
          }
        }
      }
    }
  }
  __syncthreads();

  // prefix-scan the six difference arrays, 4 consecutive columns per thread
  for (int p = P_DIFF_DEPTH_F; p <= P_DIFF_D; p++) {
    uint32_t* d = pl + p * TSTRIDE;
    const int i0 = tid * 4;
    int v0 = (int)d[i0], v1 = v0 + (int)d[i0 + 1], v2 = v1 + (int)d[i0 + 2], v3 = v2 + (int)d[i0 + 3];
    int incl = block_incl_scan(v3, wsum);
    int excl = incl - v3;
    d[i0] = (uint32_t)(excl + v0); d[i0 + 1] = (uint32_t)(excl + v1);
    d[i0 + 2] = (uint32_t)(excl + v2); d[i0 + 3] = (uint32_t)(excl + v3);
    __syncthreads();
  }

  // assemble the ABI planes and write them out (coalesced, one column per thread per pass)
  for (int col = tid; col < tlen; col += LCR_BLOCK) {
    const uint8_t R = refl[col];
    const int ri = R == 'A' ? 0 : R == 'C' ? 1 : R == 'G' ? 2 : R == 'T' ? 3 : -1;
    uint32_t f[4], rv[4];
    uint32_t sf = 0, sr = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      f[k] = pl[(P_MM_F + k) * TSTRIDE + col]; rv[k] = pl[(P_MM_R + k) * TSTRIDE + col];
      sf += f[k]; sr += rv[k];
    }
    if (ri >= 0) {
      const uint32_t mf = pl[P_DIFF_DEPTH_F * TSTRIDE + col] - sf, mr = pl[P_DIFF_DEPTH_R * TSTRIDE + col] - sr;
#pragma unroll
      for (int k = 0; k < 4; k++) if (k == ri) { f[k] = mf; rv[k] = mr; }
    }
    const int64_t o = gcol0 + col;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      planes[(int64_t)(LCR_PL_A + k) * n_cols + o] = f[k] + rv[k];
      planes[(int64_t)(LCR_PL_FWD_A + k) * n_cols + o] = f[k];
    }
    planes[(int64_t)LCR_PL_N * n_cols + o] = pl[P_DIFF_N * TSTRIDE + col];
    planes[(int64_t)LCR_PL_D * n_cols + o] = pl[P_DIFF_D * TSTRIDE + col];
    planes[(int64_t)LCR_PL_NI * n_cols + o] = pl[P_NI * TSTRIDE + col];
    planes[(int64_t)LCR_PL_TS_FWD * n_cols + o] = pl[P_DIFF_TS0 * TSTRIDE + col];
    planes[(int64_t)LCR_PL_TS_REV * n_cols + o] = pl[P_DIFF_TS1 * TSTRIDE + col];
  }
}

void launch_k1_pileup(const BatchView& b, const DevParams& p, const int32_t* tile_region, const int32_t* tile_col0,
                      int32_t n_tiles, int64_t n_cols, uint32_t* planes, hipStream_t s) {
  if (n_tiles == 0) return;
  hipLaunchKernelGGL(k1_pileup, dim3(n_tiles), dim3(LCR_BLOCK), 0, s, b, p, tile_region, tile_col0, n_cols, planes);
}
