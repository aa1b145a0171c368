// k1_pileup.hip — K0 (CIGAR decode + binning), K1 (pileup tally), K1z (poly-A mask fix) for gfx950.
//
// Together they replace Profile::fill_data_into_freq_vec (reference src/util.rs:621-949).
//
// Design.  The reference walks every read base by base.  Here:
//   K0  one wave64 per read scans the CIGAR 64 ops at a time (DPP prefix sums give every op's
//       reference / query start) and emits, per pileup tile (LCR_TILE columns of one region), compact
//       8-byte records:  M-segment (tile column, length, byte offset of its first read base, strand,
//       transcript-strand class), D-run (column, length), I-point (column).  ONT end trimming
//       (util.rs:745-751) is applied here by clipping M blocks to the untrimmed read interval.
//       Intron (N) runs never reach a tile: they become +1/-1 pairs in a global difference array that
//       is prefix-scanned once (introns dominate RNA-seq coverage).  Records are counting-sorted by
//       tile (count pass -> scan -> fill pass; slots are allocated per run of equal tiles with one
//       atomic per run).
//   K1  one workgroup owns one tile and keeps all its counters in LDS.  It never sees a CIGAR:
//       threads take one record each (coalesced 8-byte loads); whole M-segments and D-runs are
//       difference-array range updates (2 LDS atomics per record, prefix-scanned at the end); then
//       the segments' read bases are cut into 16-byte aligned pieces, one piece per thread, loaded
//       with one 16-byte global load each (512 independent loads in flight per workgroup) and XORed
//       against the tile's reference bytes in LDS: a piece that equals the reference is finished
//       (counts are "depth - mismatches"); only mismatching / non-ACGT bytes touch per-column counters.
//   K1z (HiFi presets only) re-visits the <= 2*dist_to_end read offsets next to each read end,
//       evaluates the poly-A / homopolymer window rule (util.rs:754-789) and subtracts the rare masked
//       bases from the finished planes with global atomics.
// All arithmetic is u32 adds => bit-exact regardless of order.  Base qualities are NOT read here: they
// are only needed at the < 1 % of columns that survive the count filters (k2_hist).
#include <algorithm>
#include <climits>

#include "lcr_dev.h"
#include "k2_eval.h"

// Ablation switches of the profiling experiments (tools/k1time.py): compile-time only (-DLCR_K1_ABLATE=n builds a
// measurement library that computes WRONG planes); the product build has no switch that skips work.
#ifdef LCR_K1_ABLATE
#define K1_ABL LCR_K1_ABLATE
#else
#define K1_ABL 0
#endif
#ifndef K1_THREADS
#define K1_THREADS LCR_TILE   // one column per thread in the tile epilogue
#endif
#define K1_WAVES (K1_THREADS / 64)
#define K1_CPT (LCR_TILE / K1_THREADS)  // columns per thread in the tile epilogue
#define K1_RPB 2                        // records per thread and batch
#define K1_PMAP (8 * K1_THREADS)         // pieces per batch with a direct piece -> record map in LDS
#ifndef K1_NT_STORES
#define K1_NT_STORES 0                  // k1_empty_tiles: non-temporal plane stores (measurement builds: -DK1_NT_STORES=1; measured slower, HISTORY.md Appendix C)
#endif
#ifndef K1_PIF
#define K1_PIF 4                        // 16-byte pieces in flight per thread (a batch of 1024 records has ~2500 pieces)
#endif

// (record layout: lcr_dev.h)

// ---------------------------------------------------------------------------------------------
// read -> region map (one block per region writes its read range)
__global__ void __launch_bounds__(LCR_BLOCK) k0_read_region(const int32_t* __restrict__ read_begin, int32_t* __restrict__ out) {
  const int g = blockIdx.x;
  for (int r = read_begin[g] + threadIdx.x; r < read_begin[g + 1]; r += blockDim.x) out[r] = g;
}
// tile table (one block per region): tile t of region g covers columns [(t - first) * LCR_TILE, ...)
__global__ void __launch_bounds__(LCR_BLOCK) k0_tiles(const int32_t* __restrict__ first_tile, int32_t* __restrict__ tile_region,
                                                       int32_t* __restrict__ tile_col0) {
  const int g = blockIdx.x, t0 = first_tile[g], t1 = first_tile[g + 1];
  for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) { tile_region[t] = g; tile_col0[t] = (t - t0) * LCR_TILE; }
}
// region table setup, one workgroup: first tile of every region (prefix sum of ceil(len / LCR_TILE)) and, for a
// device-resident batch, the four small region arrays written straight into pinned host memory (one kernel
// instead of four copies; the host validates them after its wait)
__global__ void __launch_bounds__(1024) k0_region_setup(const int64_t* __restrict__ start0, const int32_t* __restrict__ len,
                                                         const int64_t* __restrict__ col_off, const int32_t* __restrict__ read_begin,
                                                         int32_t ng, int32_t* __restrict__ first_tile, int64_t* h_start0,
                                                         int32_t* h_len, int64_t* h_col_off, int32_t* h_read_begin) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (h_start0) {
    for (int g = tid; g < ng; g += 1024) { h_start0[g] = start0[g]; h_len[g] = len[g]; }
    for (int g = tid; g <= ng; g += 1024) { h_col_off[g] = col_off[g]; h_read_begin[g] = read_begin[g]; }
  }
  if (tid == 0) { base_s = 0; first_tile[0] = 0; }
  __syncthreads();
  for (int g0 = 0; g0 < ng; g0 += 1024) {
    const int g = g0 + tid;
    const int n = g < ng ? (max(len[g], 0) + LCR_TILE - 1) / LCR_TILE : 0;
    const int incl = wave_incl_scan(n);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = base_s;
    for (int w = 0; w < wave; w++) before += wsum[w];
    if (g < ng) first_tile[g + 1] = before + incl;
    __syncthreads();
    if (tid == 1023) base_s = before + incl;
    __syncthreads();
  }
}
void launch_k0_region_setup(const int64_t* start0, const int32_t* len, const int64_t* col_off, const int32_t* read_begin, int32_t ng,
                            int32_t* first_tile, int64_t* h_start0, int32_t* h_len, int64_t* h_col_off, int32_t* h_read_begin,
                            hipStream_t s) {
  hipLaunchKernelGGL(k0_region_setup, dim3(1), dim3(1024), 0, s, start0, len, col_off, read_begin, ng, first_tile, h_start0, h_len,
                     h_col_off, h_read_begin);
}
void launch_k0_tiles(const int32_t* first_tile, int32_t n_regions, int32_t* tile_region, int32_t* tile_col0, hipStream_t s) {
  if (n_regions > 0) hipLaunchKernelGGL(k0_tiles, dim3(n_regions), dim3(LCR_BLOCK), 0, s, first_tile, tile_region, tile_col0);
}
void launch_k0_read_region(const BatchView& b, int32_t* read_region, hipStream_t s) {
  if (b.n_regions == 0) return;
  hipLaunchKernelGGL(k0_read_region, dim3(b.n_regions), dim3(LCR_BLOCK), 0, s, b.read_begin, read_region);
}
// both tables in one launch (a block per region fills its tiles, then its reads)
__global__ void __launch_bounds__(LCR_BLOCK) k0_tiles_read_region(const int32_t* __restrict__ first_tile, int32_t* __restrict__ tile_region,
                                                                   int32_t* __restrict__ tile_col0, const int32_t* __restrict__ read_begin,
                                                                   int32_t* __restrict__ read_region) {
  const int g = blockIdx.x, t0 = first_tile[g], t1 = first_tile[g + 1];
  for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) { tile_region[t] = g; tile_col0[t] = (t - t0) * LCR_TILE; }
  for (int r = read_begin[g] + threadIdx.x; r < read_begin[g + 1]; r += blockDim.x) read_region[r] = g;
}
void launch_k0_tiles_read_region(const BatchView& b, const int32_t* first_tile, int32_t* tile_region, int32_t* tile_col0, int32_t* read_region, hipStream_t s) {
  if (b.n_regions == 0) return;
  hipLaunchKernelGGL(k0_tiles_read_region, dim3(b.n_regions), dim3(LCR_BLOCK), 0, s, first_tile, tile_region, tile_col0, b.read_begin, read_region);
}

// ---------------------------------------------------------------------------------------------
// per-read header pack (once per batch): everything K0 needs about a read in one 64-byte line
__global__ void __launch_bounds__(LCR_BLOCK) k0_pack(BatchView b, ReadBin* __restrict__ out, int32_t* order_flag) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= b.n_reads) return;
  const int g = region_of_read(b, r);
  // precondition of k3_rows / lcr_fragments (binary searches on pos): a region's reads are sorted by position
  if (r > b.read_begin[g] && b.pos[r] < b.pos[r - 1]) *order_flag = 1;
  ReadBin h;
  h.rel_pos = (int32_t)((int64_t)b.pos[r] - b.start0[g]);
  h.vec = b.len[g]; h.ftile = b.region_first_tile[g]; h.n_cig = (int32_t)b.n_cig[r];
  h.gbase = b.col_off[g] + g;
  h.seq_off = b.seq_off[r]; h.cig_off = b.cig_off[r];
  h.lead = b.lead[r]; h.reb = b.seq_len[r] - b.trail[r];
  h.flags = b.flags[r]; h.pad_ = 0; h.pad2_ = 0;
  out[r] = h;
}
void launch_k0_pack(const BatchView& b, ReadBin* out, int32_t* order_flag, hipStream_t s) {
  if (b.n_reads == 0) return;
  hipLaunchKernelGGL(k0_pack, dim3((b.n_reads + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, b, out, order_flag);
}

// ---------------------------------------------------------------------------------------------
// The bind kernels of a batch, fused (round 6; VERDICT r05: five launches -- k0_region_setup, k0_cig_check, k0_tiles_read_region,
// k0_pack, k0_block_reads -- were 44 us per step of a device-resident batch, outside the stage's timers).  Two launches remain:
//   k0_bind_a  (before the host's one wait) block 0: the region table -- first tile of every region, the four small region arrays
//              into pinned host memory --, blocks >= 1: the layout check of the flat op space (k0_cig_check's three verdicts);
//   k0_bind_b  (behind it) a thread per read: region of the read (binary search in read_begin: a region's reads are a range),
//              read -> region table, the 64-byte header K0 reads (ReadBin), the order check of k3_rows' precondition, the op blocks'
//              first reads; the threads behind the reads fill the tile tables the same way (binary search in first_tile).
__global__ void __launch_bounds__(1024) k0_bind_a(const int64_t* __restrict__ start0, const int32_t* __restrict__ len,
                                                   const int64_t* __restrict__ col_off, const int32_t* __restrict__ read_begin, int32_t ng,
                                                   int32_t* __restrict__ first_tile, int64_t* h_start0, int32_t* h_len, int64_t* h_col_off,
                                                   int32_t* h_read_begin, const uint64_t* __restrict__ cig_off, const uint32_t* __restrict__ n_cig,
                                                   int32_t nr, int64_t n_cigar, int32_t* out) {
  const int tid = threadIdx.x;
  if (blockIdx.x > 0) {   // (k0_cig_check) out (pinned, zeroed): [1] ops not back to back, [2] ops beyond n_cigar, uint64 at byte 16 / 24: first op, end
    const int r = (int)(blockIdx.x - 1) * 1024 + tid;
    if (r + 1 < nr && cig_off[r + 1] != cig_off[r] + n_cig[r]) out[1] = 1;
    if (r < nr && (cig_off[r] > (uint64_t)n_cigar || (uint64_t)n_cig[r] > (uint64_t)n_cigar - cig_off[r])) out[2] = 1;
    if (r == 0) { uint64_t* g = reinterpret_cast<uint64_t*>(out + 4); g[0] = cig_off[0]; g[1] = cig_off[nr - 1] + n_cig[nr - 1]; }
    return;
  }
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int lane = tid & 63, wave = tid >> 6;
  if (h_start0) {
    for (int g = tid; g < ng; g += 1024) { h_start0[g] = start0[g]; h_len[g] = len[g]; }
    for (int g = tid; g <= ng; g += 1024) { h_col_off[g] = col_off[g]; h_read_begin[g] = read_begin[g]; }
  }
  if (tid == 0) { base_s = 0; first_tile[0] = 0; }
  __syncthreads();
  for (int g0 = 0; g0 < ng; g0 += 1024) {
    const int g = g0 + tid;
    const int n = g < ng ? (max(len[g], 0) + LCR_TILE - 1) / LCR_TILE : 0;
    const int incl = wave_incl_scan(n);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = base_s;
    for (int w = 0; w < wave; w++) before += wsum[w];
    if (g < ng) first_tile[g + 1] = before + incl;
    __syncthreads();
    if (tid == 1023) base_s = before + incl;
    __syncthreads();
  }
}
void launch_k0_bind_a(const int64_t* start0, const int32_t* len, const int64_t* col_off, const int32_t* read_begin, int32_t ng,
                      int32_t* first_tile, int64_t* h_start0, int32_t* h_len, int64_t* h_col_off, int32_t* h_read_begin,
                      const uint64_t* cig_off, const uint32_t* n_cig, int32_t nr, int64_t n_cigar, int32_t* out, hipStream_t s) {
  hipLaunchKernelGGL(k0_bind_a, dim3(1 + (nr + 1023) / 1024), dim3(1024), 0, s, start0, len, col_off, read_begin, ng, first_tile, h_start0, h_len,
                     h_col_off, h_read_begin, cig_off, n_cig, nr, n_cigar, out);
}

// last index g in [0, n) with a[g] <= x (a ascending, a[0] <= x): the region of a read / of a tile; empty regions are stepped over
__device__ __forceinline__ int last_le(const int32_t* __restrict__ a, int n, int x) {
  int lo = 0, hi = n;   // a[lo] <= x < a[hi] (a[n] = total > x)
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (a[mid] <= x) lo = mid; else hi = mid; }
  return lo;
}
__global__ void __launch_bounds__(LCR_BLOCK) k0_bind_b(BatchView b, ReadBin* __restrict__ rbin, int32_t* order_flag, int32_t* __restrict__ read_region,
                                                        int32_t n_tiles, int32_t* __restrict__ tile_region, int32_t* __restrict__ tile_col0,
                                                        uint64_t cig0, int32_t opb, int32_t n_blocks, int32_t* __restrict__ blk_first_read) {
  const int i = blockIdx.x * LCR_BLOCK + threadIdx.x;
  if (i >= b.n_reads) {
    const int t = i - b.n_reads;
    if (t < n_tiles) { const int g = last_le(b.region_first_tile, b.n_regions, t); tile_region[t] = g; tile_col0[t] = (t - b.region_first_tile[g]) * LCR_TILE; }
    return;
  }
  const int r = i;
  const int g = last_le(b.read_begin, b.n_regions, r);
  read_region[r] = g;
  const int32_t pos = b.pos[r];
  // precondition of k3_rows / lcr_fragments (binary searches on pos): a region's reads are sorted by position
  if (r > b.read_begin[g] && pos < b.pos[r - 1]) *order_flag = 1;
  ReadBin h;
  h.rel_pos = (int32_t)((int64_t)pos - b.start0[g]);
  h.vec = b.len[g]; h.ftile = b.region_first_tile[g]; h.n_cig = (int32_t)b.n_cig[r];
  h.gbase = b.col_off[g] + g;
  h.seq_off = b.seq_off[r]; h.cig_off = b.cig_off[r];
  h.lead = b.lead[r]; h.reb = b.seq_len[r] - b.trail[r];
  h.flags = b.flags[r]; h.pad_ = 0; h.pad2_ = 0;
  rbin[r] = h;
  // first read of every op block (k0_ops.hip): a read writes the entries of the block borders its ops span
  if (r == 0) blk_first_read[n_blocks] = b.n_reads - 1;
  const uint64_t cb = h.cig_off - cig0, ce = cb + (uint32_t)h.n_cig;
  for (uint64_t k = (cb + opb - 1) / opb; k * opb < ce; k++) blk_first_read[k] = r;
}
void launch_k0_bind_b(const BatchView& b, ReadBin* rbin, int32_t* order_flag, int32_t* read_region, int32_t n_tiles, int32_t* tile_region,
                      int32_t* tile_col0, uint64_t cig0, int32_t opb, int32_t n_blocks, int32_t* blk_first_read, hipStream_t s) {
  const int64_t n = (int64_t)b.n_reads + n_tiles;
  if (n == 0) return;
  hipLaunchKernelGGL(k0_bind_b, dim3((unsigned)((n + LCR_BLOCK - 1) / LCR_BLOCK)), dim3(LCR_BLOCK), 0, s, b, rbin, order_flag, read_region, n_tiles,
                     tile_region, tile_col0, cig0, opb, n_blocks, blk_first_read);
}

// ---------------------------------------------------------------------------------------------
// LDS planes of one tile (u32 each, LCR_TILE + 1 entries so that "end" markers at tile_len fit)
enum {
  P_DIFF_DEPTH_F = 0,  // difference array: kept aligned bases of forward reads
  P_DIFF_DEPTH_R,      //                   ... of reverse reads
  P_DIFF_TS0,          // difference array: transcript_strands[0]
  P_DIFF_TS1,          //                   transcript_strands[1]
  P_DIFF_D,            // deletion runs
  P_DIFF_N,            // intron runs that start or end in this tile (the tiles an intron covers entirely: tile_nbase)
  P_NI,                // insertions (plain counter)
  P_MM_F,              // 4 planes: mismatching base counts A,C,T,G (bits 1-2 of the ASCII code) of forward reads
  P_MM_R = P_MM_F + 4, // 4 planes: ... of reverse reads
  P_NPL = P_MM_R + 4
};
#define TSTRIDE (LCR_TILE + 1)
#define REF_PAD 16  // reference bytes are stored at refl[REF_PAD + column] so that piece starts may be "negative"

// inclusive scan of one int per thread over the whole block; the caller must pass a barrier before wsum is used again
__device__ __forceinline__ int block_incl_scan(int v, int* wsum /* K1_WAVES ints of LDS */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int s = wave_incl_scan(v);
  if (lane == 63) wsum[w] = s;
  __syncthreads();
  int add = 0;
  for (int i = 0; i < w; i++) add += wsum[i];
  return s + add;
}

__global__ void __launch_bounds__(K1_THREADS)
k1_pileup(BatchView b, DevParams prm, const int32_t* __restrict__ tile_region, const int32_t* __restrict__ tile_col0,
          int64_t n_cols, const int32_t* __restrict__ tile_fill, const int32_t* __restrict__ ent_off, const uint2* __restrict__ ents,
          const unsigned long long* __restrict__ recs,
          const int32_t* __restrict__ tile_nbase, uint32_t* __restrict__ planes, const int32_t* __restrict__ order,
          const int32_t* __restrict__ n_full, BinomTable bt, uint8_t* __restrict__ flt_flags, int32_t* __restrict__ flt_count) {
  // (tiles with records first, fullest first, the record-free ones behind them: k1_tiles_b.  Dealing the record-free
  // tiles -- pure streams of plane stores, 75 % of the stage's HBM writes -- between the full ones so that the stores run
  // under the tally was measured: 0.74 instead of 0.43 ms, XCD-aware or not; the full tiles' dependent loads then queue
  // behind a saturated write stream)
  // Record-free tiles -- uncovered, or inside introns only: most tiles of a spliced data set -- are written by
  // k1_empty_tiles; `order` lists them behind the tiles with records, whose number the tile pass left in *n_full.
  if ((int)blockIdx.x >= *n_full) return;
  const int tile = order[blockIdx.x];
  __shared__ uint32_t pl[P_NPL * TSTRIDE];
  __shared__ __attribute__((aligned(16))) uint8_t refl[REF_PAD + LCR_TILE + 32];
  __shared__ unsigned long long rec_s[K1_RPB * K1_THREADS];
  __shared__ int pstart[K1_RPB * K1_THREADS + 1];
  __shared__ uint16_t pown[K1_PMAP];
  __shared__ int wsum[K1_WAVES];

  const int tid = threadIdx.x;
  const int g = tile_region[tile];
  const int tc0 = tile_col0[tile];               // first column of the tile inside the region
  const int vec = b.len[g];
  const int tlen = min(LCR_TILE, vec - tc0);
  const int64_t gcol0 = b.col_off[g] + tc0;            // global column of tile column 0
  if (*b.error_flag != 0) return;   // K0 failed (bad CIGAR or record pool overflow): the stage is rejected or repeated
  // valid-byte masks of a 16-byte piece in the layout of the mismatch word below (bit 8k + j <-> byte 4j + k):
  // vge[lo] = bytes >= lo, vlt[hi] = bytes < hi
  __shared__ uint32_t vge[17], vlt[17];
  if (tid >= 64 && tid < 64 + 34) {
    const int v = (tid - 64) % 17;
    uint32_t m = 0;
    for (int j = 0; j < 4; j++)
      for (int kk = 0; kk < 4; kk++)
        if (tid - 64 < 17 ? 4 * j + kk >= v : 4 * j + kk < v) m |= 1u << (8 * kk + j);
    if (tid - 64 < 17) vge[v] = m; else vlt[v] = m;
  }
  for (int i = tid; i < P_NPL * TSTRIDE; i += K1_THREADS) pl[i] = 0;
  for (int i = tid; i < REF_PAD + LCR_TILE + 32; i += K1_THREADS) {
    const int col = i - REF_PAD;
    uint8_t R = (col >= 0 && col < tlen) ? b.ref[gcol0 + col] : 0;
    // only upper-case ACGT can equal a read base (htslib decodes to upper case); anything else is
    // stored as 0xFF so that every base at such a column takes the explicit-count path
    refl[i] = (R == 'A' || R == 'C' || R == 'G' || R == 'T') ? R : 0xFF;
  }
  __syncthreads();
  const uint32_t* rl32 = reinterpret_cast<const uint32_t*>(refl);

  // record batches of K1_RPB * K1_THREADS slots: fewer block-wide barriers per record.  The tile's records lie in groups of
  // <= 16 (k0_desc_bin: one 8-byte entry per group = pool offset, count), so slot j is record (j & 15) of entry j >> 4 -- no
  // search; the slots of a group beyond its count are idle.  The next batch is requested while this one is tallied.
  const int e0 = ent_off[tile];
  const int n_slots = (ent_off[tile + 1] - e0) * 16;
  static_assert(K1_RPB == 2, "a thread's two slots lie in one entry");
  auto load_recs = [&](int j, unsigned long long* out) -> int {   // slots j, j + 1 (j even); returns which of them hold a record
    out[0] = 0ull; out[1] = 0ull;
    int has = 0;
    if (j < n_slots) {
      const uint2 en = ents[e0 + (j >> 4)];
      const unsigned int k = (unsigned int)j & 15u;
      if (k < en.y) { out[0] = recs[en.x + k]; has |= 1; }
      if (k + 1 < en.y) { out[1] = recs[en.x + k + 1]; has |= 2; }
    }
    return has;
  };
  unsigned long long rec_nx[K1_RPB];
  int has_nx = load_recs(tid * K1_RPB, rec_nx);
  for (int rbase = 0; rbase < n_slots && K1_ABL != 3; rbase += K1_RPB * K1_THREADS) {
    unsigned long long rec_cur[K1_RPB];
#pragma unroll
    for (int x = 0; x < K1_RPB; x++) rec_cur[x] = rec_nx[x];
    const int has_cur = has_nx;
    has_nx = load_recs(rbase + K1_RPB * K1_THREADS + tid * K1_RPB, rec_nx);
    // ---- phase 1: K1_RPB records per thread (thread t owns batch slots K1_RPB*t .. K1_RPB*t + K1_RPB-1)
    int npc[K1_RPB], nsum = 0;
#pragma unroll
    for (int x = 0; x < K1_RPB; x++) {
      const int slot = tid * K1_RPB + x;
      const unsigned long long rec = rec_cur[x];
      const bool hasrec = (has_cur >> x) & 1;
      const unsigned long long off = rec & REC_OFF_MASK;
      const int col0 = (int)((rec >> 40) & 1023u), len = (int)((rec >> 50) & 1023u) + 1;
      int npieces = 0;
      if (hasrec) {
        if (off == REC_KIND_D) {          // util.rs:905-917: +1 per deleted reference position
          atomicAdd(&pl[P_DIFF_D * TSTRIDE + col0], 1u);
          atomicAdd(&pl[P_DIFF_D * TSTRIDE + col0 + len], 0xFFFFFFFFu);
        } else if (off == REC_KIND_I) {   // util.rs:918-929
          atomicAdd(&pl[P_NI * TSTRIDE + col0], 1u);
        } else if (off == REC_KIND_N) {   // util.rs:930-942: +1 per intron position (the part of the run inside this tile)
          atomicAdd(&pl[P_DIFF_N * TSTRIDE + col0], 1u);
          atomicAdd(&pl[P_DIFF_N * TSTRIDE + col0 + len], 0xFFFFFFFFu);
        } else {                          // M-segment: range update now, per-base corrections in phase 2
          const int strand = (int)((rec >> 60) & 1u), tscls = (int)((rec >> 61) & 3u);
          uint32_t* dp = pl + (strand ? P_DIFF_DEPTH_R : P_DIFF_DEPTH_F) * TSTRIDE;
          atomicAdd(&dp[col0], 1u);
          atomicAdd(&dp[col0 + len], 0xFFFFFFFFu);
          if (tscls) {
            uint32_t* tp = pl + (tscls == 2 ? P_DIFF_TS1 : P_DIFF_TS0) * TSTRIDE;
            atomicAdd(&tp[col0], 1u);
            atomicAdd(&tp[col0 + len], 0xFFFFFFFFu);
          }
          npieces = (len + 15) >> 4;   // 16-byte pieces from the segment's first base on (dword-unaligned 16-byte loads are fine on gfx950)
        }
      }
      rec_s[slot] = rec;
      npc[x] = K1_ABL == 1 ? 0 : npieces;
      nsum += npc[x];
    }
    int run = block_incl_scan(nsum, wsum) - nsum;   // exclusive prefix of this thread's first record
    int P = 0;                                      // pieces of the batch (every thread sums the wave totals)
#pragma unroll
    for (int i = 0; i < K1_WAVES; i++) P += wsum[i];
    // piece -> record map.  A thread fills the map entries of its own records straight from its registers: one
    // barrier covers pstart[] and pown[].
    // pown[k] = record of piece k * pstride: every piece when the batch has at most K1_PMAP of them (ONT: short
    // segments, stride 1), every pstride-th one otherwise (HiFi: segments of up to 33 pieces, 25 000 pieces per batch) --
    // a piece then starts at its map entry and steps forward over at most pstride records instead of a ten-step
    // binary search over pstart[]
    const int pstride = (P + K1_PMAP - 1) / K1_PMAP;   // (>= 1 unless the batch has no piece)
    if (tid == 0) pstart[0] = 0;
#pragma unroll
    for (int x = 0; x < K1_RPB; x++) {
      const int first = run;
      run += npc[x];
      pstart[tid * K1_RPB + x + 1] = run;
      if (pstride == 1) for (int q = first; q < run; q++) pown[q] = (uint16_t)(tid * K1_RPB + x);
      else if (pstride > 1) for (int k = (first + pstride - 1) / pstride; k * pstride < run; k++) pown[k] = (uint16_t)(tid * K1_RPB + x);
    }
    __syncthreads();
    // ---- phase 2: one 16-byte aligned piece of read bases per thread, four pieces in flight per thread
    struct Piece { uint4 v; int colA, k_lo, k_hi, strand; bool ok; };
    auto fetch = [&](int p) -> Piece {
      Piece q;
      q.ok = p < P;
      if (!q.ok) { q.v = make_uint4(0, 0, 0, 0); q.colA = 0; q.k_lo = q.k_hi = 0; q.strand = 0; return q; }
      int lo;  // last record with pstart <= p
      if (pstride == 1) lo = pown[p];
      else { lo = pown[p / pstride]; while (pstart[lo + 1] <= p) lo++; }
      const unsigned long long rc = rec_s[lo];
      const uint32_t rhi = (uint32_t)(rc >> 32);
      const long long soff = (long long)(rc & REC_OFF_MASK);
      const int scol = (int)((rhi >> 8) & 1023u), slen = (int)((rhi >> 18) & 1023u) + 1;
      q.strand = (int)((rhi >> 28) & 1u);
      const int d16 = 16 * (p - pstart[lo]);                              // piece d16 / 16 of the segment
      const long long A = soff + d16;                                     // byte address of the piece
      q.k_lo = 0; q.k_hi = min(16, slen - d16);                           // valid bytes
      q.colA = scol + d16;                                                // column of byte 0
      if (A + 16 <= b.n_bases) q.v = *reinterpret_cast<const uint4*>(b.bases + A);
      else {  // last partial 16 bytes of the whole base array
        uint32_t t[4] = {0, 0, 0, 0};
        for (int x = 0; x < 16; x++)
          if (A + x < b.n_bases) t[x >> 2] |= (uint32_t)b.bases[A + x] << (8 * (x & 3));
        q.v = make_uint4(t[0], t[1], t[2], t[3]);
      }
      return q;
    };
    auto tally = [&](const Piece& q) {
      if (!q.ok) return;
      // 16 reference bytes starting at column colA (unaligned in LDS): 5 dwords + byte alignment
      const int ci = q.colA + REF_PAD, di = ci >> 2;
      const uint32_t sh = (uint32_t)(ci & 3);
      const uint32_t r0 = rl32[di], r1 = rl32[di + 1], r2 = rl32[di + 2], r3 = rl32[di + 3], r4 = rl32[di + 4];
      const uint32_t x0 = q.v.x ^ __builtin_amdgcn_alignbyte(r1, r0, sh), x1 = q.v.y ^ __builtin_amdgcn_alignbyte(r2, r1, sh);
      const uint32_t x2 = q.v.z ^ __builtin_amdgcn_alignbyte(r3, r2, sh), x3 = q.v.w ^ __builtin_amdgcn_alignbyte(r4, r3, sh);
      // mismatching bytes among the valid ones as one word: bit 8k + j <-> byte k of dword j.  Per dword a carry-free
      // SWAR test (bit 7 of every non-zero byte), then the four flag words are interleaved by shifts; the valid range
      // comes from two table words in the same layout.
      auto nzf = [](uint32_t x) -> uint32_t { return (x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u; };
      uint32_t mm = (nzf(x0) >> 7) | (nzf(x1) >> 6) | (nzf(x2) >> 5) | (nzf(x3) >> 4);
      mm &= vlt[q.k_hi];
      if (K1_ABL == 5) mm = 0;
      // per mismatch (a few % of the bases; the lanes of a wave run as many trips as the worst piece has mismatches, so the
      // body is kept short and branch-free): the byte comes out of the piece's registers with one v_perm_b32
      uint32_t* dp = pl + (q.strand ? P_DIFF_DEPTH_R : P_DIFF_DEPTH_F) * TSTRIDE + q.colA;
      uint32_t* mp = pl + (q.strand ? P_MM_R : P_MM_F) * TSTRIDE + q.colA;
      while (mm) {
        const uint32_t pb = (uint32_t)__builtin_ctz(mm), jd = pb & 3u, kb = pb >> 3;   // dword jd, byte kb
        mm &= mm - 1;
        const uint32_t lo = (jd & 2u) ? q.v.z : q.v.x, hi = (jd & 2u) ? q.v.w : q.v.y;
        const uint32_t base = __builtin_amdgcn_perm(hi, lo, 0x0c0c0c00u | ((jd & 1u) << 2) | kb);   // byte kb of (jd & 1 ? hi : lo)
        const uint32_t dcol = 4u * jd + kb;
        // A,C,G,T (either case) -> plane h = bits 1-2 of the byte (A 0, C 1, T 2, G 3: the LDS mismatch planes are kept in THAT
        // order); anything else is "Invalid nucleotide base" (util.rs:890-892): no allele count (depth - 1), transcript strand
        // still counted.  Valid iff the byte, upper-cased, is the letter of its own class (one v_perm_b32 from "ACTG").
        const uint32_t h = (base >> 1) & 3u;
        const bool acgt = (base & 0xDFu) == __builtin_amdgcn_perm(0u, 0x47544341u /* 'A','C','T','G' */, 0x0c0c0c00u | h);
        if (acgt) atomicAdd(mp + h * TSTRIDE + dcol, 1u);
        else { atomicAdd(dp + dcol, 0xFFFFFFFFu); atomicAdd(dp + dcol + 1, 1u); }
      }
    };
    for (int p = tid; p < P; p += K1_PIF * K1_THREADS) {
      Piece q[K1_PIF];
#pragma unroll
      for (int x = 0; x < K1_PIF; x++) q[x] = fetch(p + x * K1_THREADS);
#pragma unroll
      for (int x = 0; x < K1_PIF; x++) tally(q[x]);
    }
    __syncthreads();
  }
  __syncthreads();

  // prefix-scan the six difference arrays together (one pair of barriers instead of six), K1_CPT consecutive
  // columns per thread
  {
    constexpr int NP = P_DIFF_N - P_DIFF_DEPTH_F + 1;
    static_assert(NP == 6, "six difference arrays");
    __shared__ int wsum5[NP][K1_WAVES];
    const int lane = tid & 63, w = tid >> 6;
    int v[NP][K1_CPT], incl[NP];
#pragma unroll
    for (int p = 0; p < NP; p++) {
      const uint32_t* d = pl + (P_DIFF_DEPTH_F + p) * TSTRIDE;
      int run = 0;
#pragma unroll
      for (int x = 0; x < K1_CPT; x++) { run += (int)d[tid * K1_CPT + x]; v[p][x] = run; }
      incl[p] = wave_incl_scan(run);
      if (lane == 63) wsum5[p][w] = incl[p];
      incl[p] -= run;   // exclusive inside the wave
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; p++) {
      int add = incl[p];
      for (int i = 0; i < w; i++) add += wsum5[p][i];
      uint32_t* d = pl + (P_DIFF_DEPTH_F + p) * TSTRIDE;
#pragma unroll
      for (int x = 0; x < K1_CPT; x++) d[tid * K1_CPT + x] = (uint32_t)(add + v[p][x]);
    }
    __syncthreads();
  }

  // assemble the ABI planes and write them out (coalesced: consecutive threads, consecutive columns)
  const uint32_t nbase = (uint32_t)tile_nbase[tile];
  static_assert(K1_CPT == 1, "the fused filter counts one column per thread");
  int n_pass = 0;
  for (int col = tid; col < tlen; col += K1_THREADS) {
    const uint8_t R = refl[REF_PAD + col];
    const int ri = R == 'A' ? 0 : R == 'C' ? 1 : R == 'G' ? 2 : R == 'T' ? 3 : -1;
    uint32_t f[4], rv[4];
    uint32_t sf = 0, sr = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {   // ABI order A,C,G,T <- LDS order A,C,T,G (the mismatch loop's class h)
      const int hk = k < 2 ? k : 5 - k;
      f[k] = pl[(P_MM_F + hk) * TSTRIDE + col]; rv[k] = pl[(P_MM_R + hk) * TSTRIDE + col];
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
    // intron plane: introns that cover the whole tile + the runs that start or end inside it
    planes[(int64_t)LCR_PL_N * n_cols + o] = nbase + pl[P_DIFF_N * TSTRIDE + col];
    planes[(int64_t)LCR_PL_D * n_cols + o] = pl[P_DIFF_D * TSTRIDE + col];
    planes[(int64_t)LCR_PL_NI * n_cols + o] = pl[P_NI * TSTRIDE + col];
    planes[(int64_t)LCR_PL_TS_FWD * n_cols + o] = pl[P_DIFF_TS0 * TSTRIDE + col];
    planes[(int64_t)LCR_PL_TS_REV * n_cols + o] = pl[P_DIFF_TS1 * TSTRIDE + col];
    // pass 1 of the candidate filters (candidate.rs:90-234, k2_eval.h) on the counts this thread holds: the presets without a poly-A pass
    // behind this kernel (ONT) get k2_filter's flags and per-tile counts here -- one pass over the planes less (round 6)
    if (flt_flags) {
      uint32_t cnt[4];
#pragma unroll
      for (int k = 0; k < 4; k++) cnt[k] = f[k] + rv[k];
      const ColEval ev = eval_counts(cnt, R, prm, bt, [&]() { return pl[P_DIFF_D * TSTRIDE + col]; }, [&]() { return nbase + pl[P_DIFF_N * TSTRIDE + col]; },
                                     [&](int k) { return f[k]; });
      flt_flags[o] = ev.pass ? 1 : 0;
      n_pass += ev.pass ? 1 : 0;
    }
  }
  if (flt_flags) {
    const int total = __syncthreads_count(n_pass);   // (one column per thread: n_pass is 0 or 1)
    if (tid == 0) flt_count[tile] = total;
  }
}

// Record-free tiles (uncovered, or inside introns only): every plane is 0 except the intron plane, which is the number of
// introns that cover the whole tile.  Three quarters of the tiles of a spliced data set and 75 % of the stage's HBM writes
// are like this: a pure store stream, written by small workgroups without LDS (inside k1_pileup's 512-thread, 52 KB
// workgroups -- three per CU -- the same stores ran at 3.7 TB/s; hipMemset reaches 6.8 on this part).
// One workgroup of 128 threads per tile: 16-byte stores, 4 consecutive columns per thread and plane.
// one record-free tile (the bt-th behind the tiles with records in `order`) by the `nthr` threads lt = 0 .. nthr - 1 (nthr = 128)
__device__ __forceinline__ void empty_tile_body(const BatchView& b, const int32_t* __restrict__ tile_region, const int32_t* __restrict__ tile_col0, int64_t n_cols,
                                                const int32_t* __restrict__ tile_nbase, uint32_t* __restrict__ planes, const int32_t* __restrict__ order, int nf, int bt,
                                                int lt, int zeroed, int32_t* __restrict__ flt_count) {
  const int tile = order[nf + bt];
  if (flt_count && lt == 0) flt_count[tile] = 0;   // (a record-free tile has no survivor of the count filters: k2_filter's verdict for it)
  const int g = tile_region[tile], tc0 = tile_col0[tile];
  const int tlen = min(LCR_TILE, b.len[g] - tc0);
  const int64_t gcol0 = b.col_off[g] + tc0;
  const uint32_t nb = (uint32_t)tile_nbase[tile];
  if (zeroed) {   // the planes were zeroed while K0 ran (lcr_pileup): only the intron plane of a tile inside introns is left to write
    if (nb != 0) for (int col = lt; col < tlen; col += 128) planes[(int64_t)LCR_PL_N * n_cols + gcol0 + col] = nb;
    return;
  }
  for (int col = lt * 4; col < tlen; col += 128 * 4) {
    const int64_t o = gcol0 + col;
    if (col + 4 <= tlen) {
#pragma unroll
      for (int k = 0; k < LCR_NPLANES; k++) {
        // (K1_NT_STORES: non-temporal stores -- 350 MB of zeros that nobody reads before the next batch would then not push the read bases k2_hist
        // asks for next out of the Infinity Cache; measured: k2_hist 0.24 -> 0.20-0.23 ms, but this kernel 53 -> ~110 us: slower in sum)
        typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
        const u4_t v = k == LCR_PL_N ? (u4_t){nb, nb, nb, nb} : (u4_t){0u, 0u, 0u, 0u};
        if (K1_NT_STORES) __builtin_nontemporal_store(v, reinterpret_cast<u4_t*>(planes + (int64_t)k * n_cols + o));
        else *reinterpret_cast<u4_t*>(planes + (int64_t)k * n_cols + o) = v;
      }
    } else {
      for (int c2 = col; c2 < tlen; c2++) {
#pragma unroll
        for (int k = 0; k < LCR_NPLANES; k++) planes[(int64_t)k * n_cols + gcol0 + c2] = k == LCR_PL_N ? nb : 0u;
      }
    }
  }
}
__global__ void __launch_bounds__(128) k1_empty_tiles(BatchView b, const int32_t* __restrict__ tile_region, const int32_t* __restrict__ tile_col0,
                                                       int64_t n_cols, const int32_t* __restrict__ tile_nbase, uint32_t* __restrict__ planes,
                                                       const int32_t* __restrict__ order, const int32_t* __restrict__ n_full, int zeroed, int n_tiles, int bg_tiles,
                                                       int32_t* __restrict__ flt_count) {
  const int nf = *n_full;
  if (*b.error_flag != 0) return;
  // bg_tiles > 0: a few workgroups walk all record-free tiles (a throttled store stream beside the tally, launch_k1_pileup)
  for (int bt = blockIdx.x; bt + nf < n_tiles; bt += bg_tiles > 0 ? (int)gridDim.x : n_tiles)
    empty_tile_body(b, tile_region, tile_col0, n_cols, tile_nbase, planes, order, nf, bt, (int)threadIdx.x, zeroed, flt_count);
}

// Workgroups are started in grid order and a tile's time goes with its records (none: a few us; 15 000: ~150 us), so K1
// takes its tiles through a permutation that puts the fullest first: a counting sort of the tile indices by
// floor(log2(records)), descending (the order inside a class is whatever the cursors give -- every tile writes only its
// own columns, so the planes do not depend on it).  The same two passes produce tile_nbase[t] = introns that cover tile t
// entirely (inclusive scan of K0's tile-level difference array: an intron adds +1 at the tile after its first and -1 at its
// last, both inside its region, so every region's entries sum to zero and no segment handling is needed), ent_off = the
// tiles' offsets in the entry list (exclusive scan of their entry counts), and K0's accounting totals.
// Two multi-block kernels (a single workgroup spent 60 us on load latency): pass A = class histogram + block sums,
// pass B = offsets + scatter.  Class 32 = record-free tiles (three quarters of them), ranked by ballots.
#define TS_TILES 1024   // tiles per workgroup (256 threads x 4)
struct TileScanTmp {    // scratch in HBM, cleared with K0's counters
  int cls_cnt[40];      // tiles per class
  int cls_cur[40];      // scatter cursors
  int n_full, pad_[7];  // tiles with records (pass B, for K1's two kernels)
};
__device__ __forceinline__ int tile_class(int fill) { return fill > 0 ? __clz(fill) : 32; }   // more records, lower class

__global__ void __launch_bounds__(256) k1_tiles_a(const int32_t* __restrict__ tile_fill, const int32_t* __restrict__ tile_ndiff,
                                                   const int32_t* __restrict__ tile_nent, int32_t n_tiles, TileScanTmp* __restrict__ tmp,
                                                   int2* __restrict__ blk_sum, const unsigned int* __restrict__ acct, int32_t n_acct,
                                                   unsigned int* __restrict__ ctl, unsigned int* __restrict__ host_ctl) {
  __shared__ int hist[33], ws[4][2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (blockIdx.x == 0 && w == 0) {   // K0's accounting slots -> control block, fetched by the host behind this kernel
    int it = 0, rc = 0, pt = 0, dt = 0;   // items, records; fullest pool / descriptor shard
    for (int i = lane; i < n_acct; i += 64) { it += (int)acct[32 * i]; rc += (int)acct[32 * i + 1]; pt = max(pt, (int)acct[32 * i + 2]); dt = max(dt, (int)acct[32 * i + 3]); }
    it = wave_incl_scan(it); rc = wave_incl_scan(rc);
    for (int o = 32; o > 0; o >>= 1) { pt = max(pt, __shfl_xor(pt, o, 64)); dt = max(dt, __shfl_xor(dt, o, 64)); }
    if (lane == 63) {
      ctl[0] = (unsigned int)pt; ctl[1] = (unsigned int)it; ctl[2] = (unsigned int)rc; ctl[4] = (unsigned int)dt;
      // the same five words straight into the host's pinned block (ctl[3] = K0's verdict): no copy in the queue in front of k1_tiles_b
      if (host_ctl) { host_ctl[0] = (unsigned int)pt; host_ctl[1] = (unsigned int)it; host_ctl[2] = (unsigned int)rc; host_ctl[3] = ctl[3]; host_ctl[4] = (unsigned int)dt; }
    }
  }
  if (tid < 33) hist[tid] = 0;
  __syncthreads();
  const int t0 = blockIdx.x * TS_TILES + tid * 4;
  int sd = 0, sc = 0, n_empty = 0;
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const int t = t0 + x;
    if (t < n_tiles) {
      sd += tile_ndiff[t]; sc += tile_nent[t];
      const int c = tile_class(tile_fill[t]);
      if (c == 32) n_empty++; else atomicAdd(&hist[c], 1);
    }
  }
  sd = wave_incl_scan(sd); sc = wave_incl_scan(sc); n_empty = wave_incl_scan(n_empty);
  if (lane == 63) { ws[w][0] = sd; ws[w][1] = sc; atomicAdd(&hist[32], n_empty); }
  __syncthreads();
  if (tid == 0) blk_sum[blockIdx.x] = make_int2(ws[0][0] + ws[1][0] + ws[2][0] + ws[3][0], ws[0][1] + ws[1][1] + ws[2][1] + ws[3][1]);
  if (tid < 33 && hist[tid]) atomicAdd(&tmp->cls_cnt[tid], hist[tid]);
}

__global__ void __launch_bounds__(256) k1_tiles_b(const int32_t* __restrict__ tile_fill, const int32_t* __restrict__ tile_ndiff,
                                                   const int32_t* __restrict__ tile_nent, int32_t n_tiles, TileScanTmp* __restrict__ tmp,
                                                   const int2* __restrict__ blk_sum, int32_t* __restrict__ tile_nbase, int32_t* __restrict__ ent_off,
                                                   int32_t* __restrict__ order) {
  __shared__ int hist[33], base[33], ws[4][2], pre[2];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid < 33) hist[tid] = 0;
  // sums of the blocks in front of this one
  int pd = 0, pc = 0;
  for (int i = tid; i < (int)blockIdx.x; i += 256) { const int2 v = blk_sum[i]; pd += v.x; pc += v.y; }
  pd = wave_incl_scan(pd); pc = wave_incl_scan(pc);
  if (lane == 63) { ws[w][0] = pd; ws[w][1] = pc; }
  __syncthreads();
  if (tid == 0) { pre[0] = ws[0][0] + ws[1][0] + ws[2][0] + ws[3][0]; pre[1] = ws[0][1] + ws[1][1] + ws[2][1] + ws[3][1]; }
  __syncthreads();
  const int t0 = blockIdx.x * TS_TILES + tid * 4;
  int vd[4], vc[4], cls[4], sd = 0, sc = 0;
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const int t = t0 + x;
    vd[x] = t < n_tiles ? tile_ndiff[t] : 0; vc[x] = t < n_tiles ? tile_nent[t] : 0;
    cls[x] = t < n_tiles ? tile_class(tile_fill[t]) : -1;
    sd += vd[x]; sc += vc[x];
    if (cls[x] >= 0 && cls[x] < 32) atomicAdd(&hist[cls[x]], 1);
  }
  const int id = wave_incl_scan(sd), ic = wave_incl_scan(sc);
  __syncthreads();   // (pre[] was read by everyone? no: ws is rewritten below, pre is not) -- ws reuse
  if (lane == 63) { ws[w][0] = id; ws[w][1] = ic; }
  // record-free tiles of this block, ranked by ballots: per-wave counts first
  int my_e = 0;
#pragma unroll
  for (int x = 0; x < 4; x++) my_e += cls[x] == 32 ? 1 : 0;
  const int ie = wave_incl_scan(my_e);
  __shared__ int we[4];
  if (lane == 63) we[w] = ie;
  __syncthreads();
  if (tid == 0) hist[32] = we[0] + we[1] + we[2] + we[3];
  __syncthreads();
  // this block's span in every class: class bases (exclusive scan of the class counts, class 0 first) + a cursor draw
  if (tid < 33 && hist[tid]) {
    int b0 = 0;
    for (int k = 0; k < tid; k++) b0 += tmp->cls_cnt[k];
    base[tid] = b0 + atomicAdd(&tmp->cls_cur[tid], hist[tid]);
  }
  int run_d = pre[0] + id - sd, run_c = pre[1] + ic - sc;
  for (int i = 0; i < w; i++) { run_d += ws[i][0]; run_c += ws[i][1]; }
  int e_before = ie - my_e;
  for (int i = 0; i < w; i++) e_before += we[i];
  if (tid < 33) hist[tid] = 0;   // (reused as the block's cursors; the bases are in base[])
  __syncthreads();
#pragma unroll
  for (int x = 0; x < 4; x++) {
    const int t = t0 + x;
    if (t < n_tiles) {
      run_d += vd[x]; tile_nbase[t] = run_d;
      ent_off[t] = run_c; run_c += vc[x];
      if (cls[x] == 32) order[base[32] + e_before++] = t;
      else order[base[cls[x]] + atomicAdd(&hist[cls[x]], 1)] = t;
      if (t == n_tiles - 1) { ent_off[n_tiles] = run_c; tmp->n_full = n_tiles - tmp->cls_cnt[32]; }
    }
  }
}

void launch_k1_pileup(const BatchView& b, const DevParams& p, const int32_t* tile_region, const int32_t* tile_col0,
                      int32_t n_tiles, int64_t n_cols, const int32_t* tile_fill, const int32_t* ent_off, const void* ents,
                      const unsigned long long* recs, const int32_t* tile_nbase, uint32_t* planes, const int32_t* order /* launch_k1_tiles_b */,
                      const int32_t* tiles_tmp, int zeroed, hipStream_t s, hipStream_t bg, hipEvent_t ev0, hipEvent_t ev1, int bg_wgs,
                      uint8_t* flt_flags, int32_t* flt_count) {
  static const BinomTable bt = make_binom_table();
  if (n_tiles == 0) return;
  if (bg_wgs == -4) {   // the tally alone: the record-free tiles go with the poly-A pass (launch_k1_zonefix_tiles, HiFi presets)
    hipLaunchKernelGGL(k1_pileup, dim3(n_tiles), dim3(K1_THREADS), 0, s, b, p, tile_region, tile_col0, n_cols, tile_fill, ent_off,
                       (const uint2*)ents, recs, tile_nbase, planes, order, tiles_tmp + 80 /* TileScanTmp::n_full */, bt, flt_flags, flt_count);
    return;
  }
  if (bg && bg_wgs == -3) {
    // HiFi presets (round 6): the record-free tiles' stores on a second queue BESIDE the poly-A pass that the caller queues on `s` behind the
    // tally (k1_zonefix_ends: instruction-bound, a few global atomics, on tiles with records only -- the store stream touches the other
    // tiles); the caller makes `s` wait for ev1 behind the poly-A pass
    hipLaunchKernelGGL(k1_pileup, dim3(n_tiles), dim3(K1_THREADS), 0, s, b, p, tile_region, tile_col0, n_cols, tile_fill, ent_off,
                       (const uint2*)ents, recs, tile_nbase, planes, order, tiles_tmp + 80 /* TileScanTmp::n_full */, bt, flt_flags, flt_count);
    hipEventRecord(ev0, s); hipStreamWaitEvent(bg, ev0, 0);
    hipLaunchKernelGGL(k1_empty_tiles, dim3(n_tiles), dim3(128), 0, bg, b, tile_region, tile_col0, n_cols, tile_nbase, planes, order, tiles_tmp + 80, zeroed, n_tiles, 0, flt_count);
    hipEventRecord(ev1, bg);
    return;
  }
  if (bg_wgs == -2) {
    // lcr_debug_set("bg_tiles", -2): the record-free tiles' stores FIRST, the tally behind them on the same queue -- the tally then is the last
    // kernel of the stage to touch memory, and what it read (the read bases) is what the Infinity Cache holds when k2_hist asks for the
    // survivors' bases (measured with the stores early on a second queue: k2_hist 0.24 -> 0.20 ms)
    hipLaunchKernelGGL(k1_empty_tiles, dim3(n_tiles), dim3(128), 0, s, b, tile_region, tile_col0, n_cols, tile_nbase, planes, order, tiles_tmp + 80, zeroed, n_tiles, 0, flt_count);
    hipLaunchKernelGGL(k1_pileup, dim3(n_tiles), dim3(K1_THREADS), 0, s, b, p, tile_region, tile_col0, n_cols, tile_fill, ent_off,
                       (const uint2*)ents, recs, tile_nbase, planes, order, tiles_tmp + 80 /* TileScanTmp::n_full */, bt, flt_flags, flt_count);
    return;
  }
  if (bg && bg_wgs < 0) {
    // (measurement switch, lcr_debug_set("bg_tiles", -1)) the record-free tiles' stores at full width on a second queue, started BEFORE the
    // caller queues k0_desc_bin + the tally on `s`: ev0 was recorded behind k1_tiles_b by the caller (launch_k1_empty_early)
    hipLaunchKernelGGL(k1_pileup, dim3(n_tiles), dim3(K1_THREADS), 0, s, b, p, tile_region, tile_col0, n_cols, tile_fill, ent_off,
                       (const uint2*)ents, recs, tile_nbase, planes, order, tiles_tmp + 80 /* TileScanTmp::n_full */, bt, flt_flags, flt_count);
    hipStreamWaitEvent(s, ev1, 0);
    return;
  }
  if (bg && bg_wgs > 0) {   // the record-free tiles' stores as a throttled stream on a second queue, beside the tally
    hipEventRecord(ev0, s); hipStreamWaitEvent(bg, ev0, 0);
    hipLaunchKernelGGL(k1_empty_tiles, dim3(bg_wgs), dim3(128), 0, bg, b, tile_region, tile_col0, n_cols, tile_nbase, planes, order, tiles_tmp + 80, zeroed, n_tiles, 1, flt_count);
    hipEventRecord(ev1, bg);
  }
  hipLaunchKernelGGL(k1_pileup, dim3(n_tiles), dim3(K1_THREADS), 0, s, b, p, tile_region, tile_col0, n_cols, tile_fill, ent_off,
                     (const uint2*)ents, recs, tile_nbase, planes, order, tiles_tmp + 80 /* TileScanTmp::n_full */, bt, flt_flags, flt_count);
  if (bg && bg_wgs > 0) hipStreamWaitEvent(s, ev1, 0);
  else hipLaunchKernelGGL(k1_empty_tiles, dim3(n_tiles), dim3(128), 0, s, b, tile_region, tile_col0, n_cols, tile_nbase, planes, order, tiles_tmp + 80, zeroed, n_tiles, 0, flt_count);
}
// the record-free tiles on queue `bg`, behind the tile passes of `s` (ev0) -- beside k0_desc_bin and the start of the tally; ev1 = done
void launch_k1_empty_early(const BatchView& b, const int32_t* tile_region, const int32_t* tile_col0, int32_t n_tiles, int64_t n_cols, const int32_t* tile_nbase,
                           uint32_t* planes, const int32_t* order, const int32_t* tiles_tmp, hipStream_t s, hipStream_t bg, hipEvent_t ev0, hipEvent_t ev1, int32_t* flt_count) {
  if (n_tiles == 0) return;
  hipEventRecord(ev0, s); hipStreamWaitEvent(bg, ev0, 0);
  hipLaunchKernelGGL(k1_empty_tiles, dim3(n_tiles), dim3(128), 0, bg, b, tile_region, tile_col0, n_cols, tile_nbase, planes, order, tiles_tmp + 80, 0, n_tiles, 0, flt_count);
  hipEventRecord(ev1, bg);
}
// the tile-order / intron-base / accounting pass alone (the host fetches K0's control block behind it, before K1 is queued)
// the tile passes alone (the host fetches K0's control block behind pass A, before the rest is queued)
size_t launch_k1_tiles_tmp_words(int32_t n_tiles) { return 88 + 2 * (size_t)((n_tiles + TS_TILES - 1) / TS_TILES) + 8; }
void launch_k1_tiles_a(int32_t n_tiles, const int32_t* tile_fill, const int32_t* tile_ndiff, const int32_t* tile_nent, int32_t* tmp /* zeroed */,
                       const unsigned int* acct, int32_t n_acct, unsigned int* ctl, unsigned int* host_ctl, hipStream_t s) {
  const int nb = (n_tiles + TS_TILES - 1) / TS_TILES;
  hipLaunchKernelGGL(k1_tiles_a, dim3(nb), dim3(256), 0, s, tile_fill, tile_ndiff, tile_nent, n_tiles, (TileScanTmp*)tmp, (int2*)(tmp + 88), acct, n_acct, ctl, host_ctl);
}
void launch_k1_tiles_b(int32_t n_tiles, const int32_t* tile_fill, const int32_t* tile_ndiff, const int32_t* tile_nent, int32_t* tmp,
                       int32_t* tile_nbase, int32_t* ent_off, int32_t* order, hipStream_t s) {
  const int nb = (n_tiles + TS_TILES - 1) / TS_TILES;
  hipLaunchKernelGGL(k1_tiles_b, dim3(nb), dim3(256), 0, s, tile_fill, tile_ndiff, tile_nent, n_tiles, (TileScanTmp*)tmp, (const int2*)(tmp + 88),
                     tile_nbase, ent_off, order);
}

// ---------------------------------------------------------------------------------------------
// K1z (HiFi presets): poly-A / homopolymer mask (util.rs:754-789).  One thread per (read, slot): slot
// s < D -> read offset c = lead + s; slot D + s -> c = reb - D + 1 + s (s < D-1): exactly the offsets
// with c - lead < D or reb - c < D.  The base at c is masked iff a window of L identical bases X in
// {A,C,G,T}, X != the column's reference byte, starts in [c-L, c+1] inside the read.  Masked bases were
// counted by K1 like any other base; they are rare, so they are subtracted with global atomics.
__global__ void __launch_bounds__(LCR_BLOCK)
k1_zonefix(BatchView b, int D, int L, int64_t n_cols, uint32_t* __restrict__ planes) {
  // a thread per (read, slot), 2 D slots per read (any D: the index runs over n_reads x 2 D)
  const int per = 2 * D;
  const long long idx = (long long)blockIdx.x * LCR_BLOCK + threadIdx.x;
  const long long rl = idx / per;
  const int s = (int)(idx - rl * per);
  if (rl >= b.n_reads || s >= per - 1) return;
  const int r = (int)rl;
  const int seq_len = b.seq_len[r], lead = b.lead[r], reb = seq_len - b.trail[r];
  const uint8_t* __restrict__ seq = b.bases + b.seq_off[r];
  const int c = s < D ? lead + s : reb - D + 1 + (s - D);
  if (c < lead || c >= reb) return;                    // not an aligned read offset
  if (s >= D && c - lead < D) return;                  // both zones overlap: offset already covered by slot c-lead
  uint32_t m = 0;  // bit X: a homopolymer window of X starts in [c-L, c+1]
  const long long gabs = (long long)b.seq_off[r] + c - L;   // absolute byte offset of read offset c-L
  if (L <= 6 && c - L >= 0 && c + L <= seq_len - 1 && (gabs & ~3ll) + 16 <= b.n_bases) {
    // fast path: the 2L+1 <= 13 window bytes lie inside 4 aligned dwords; one round trip, no loop of byte loads
    const uint32_t* wp = reinterpret_cast<const uint32_t*>(b.bases + (gabs & ~3ll));
    const uint32_t w0 = wp[0], w1 = wp[1], w2 = wp[2], w3 = wp[3], sh = (uint32_t)(gabs & 3ll);
    const uint32_t d[4] = {__builtin_amdgcn_alignbyte(w1, w0, sh), __builtin_amdgcn_alignbyte(w2, w1, sh),
                           __builtin_amdgcn_alignbyte(w3, w2, sh), w3 >> (8 * sh)};
    int run = 1;
    uint32_t prev = d[0] & 0xffu;
#pragma unroll
    for (int i = 1; i < 13; i++) {
      if (i <= 2 * L) {
        const uint32_t cur = (d[i >> 2] >> (8 * (i & 3))) & 0xffu;
        run = (cur == prev) ? run + 1 : 1;
        prev = cur;
        if (run >= L) m |= cur == 'A' ? 1u : cur == 'C' ? 2u : cur == 'G' ? 4u : cur == 'T' ? 8u : 0u;
      }
    }
  } else {
    const int lo = max(c - L, 0), hi = min(c + L, seq_len - 1);
    if (hi - lo + 1 < L) return;
    int run = 1;
    uint8_t prev = seq[lo];
    for (int i = lo + 1; i <= hi; i++) {
      const uint8_t cur = seq[i];
      run = (cur == prev) ? run + 1 : 1;
      prev = cur;
      if (run >= L) m |= cur == 'A' ? 1u : cur == 'C' ? 2u : cur == 'G' ? 4u : cur == 'T' ? 8u : 0u;
    }
  }
  if (m == 0) return;
  // column of read offset c: walk the CIGAR (HiFi reads have a handful of ops)
  const int g = region_of_read(b, r);
  const int vec = b.len[g];
  const uint32_t* __restrict__ cg = b.cigar + b.cig_off[r];
  const uint32_t ncig = b.n_cig[r];
  int col = -1;
  bool found = false;
  if (s < D) {   // leading zone: the op is among the first few
    int p = (int)((int64_t)b.pos[r] - b.start0[g]), q = lead > 0 ? lead : 0;
    for (uint32_t i = 0; i < ncig && !found; i++) {
      const int op = cg[i] & 15, len = (int)(cg[i] >> 4);
      if (op == 0 || op == 7 || op == 8) {
        if (c < q + len) { col = p + (c - q); found = true; }
        p += len; q += len;
      } else if (op == 1) {
        if (c < q + len) found = true;  // inside an insertion: no column
        q += len;
      } else if (op == 2 || op == 3) p += len;
    }
  } else {       // trailing zone: walk back from the read's end (K0 recorded its reference end), the op is among the last few
    int p = b.read_rend[r], q = reb;   // one past the last reference column / aligned read offset
    for (int i = (int)ncig - 1; i >= 0 && !found; i--) {
      const int op = cg[i] & 15, len = (int)(cg[i] >> 4);
      if (op == 0 || op == 7 || op == 8) {
        if (c >= q - len) { col = p - (q - c); found = true; }
        p -= len; q -= len;
      } else if (op == 1) {
        if (c >= q - len) found = true;  // inside an insertion: no column
        q -= len;
      } else if (op == 2 || op == 3) p -= len;
    }
  }
  if (col < 0 || col >= vec) return;
  const int64_t o = b.col_off[g] + col;
  const uint8_t R = b.ref[o];
  const uint32_t rbit = R == 'A' ? 1u : R == 'C' ? 2u : R == 'G' ? 4u : R == 'T' ? 8u : 0u;
  if ((m & ~rbit) == 0) return;
  // masked: the base contributes nothing (util.rs:801)
  const int fl = b.flags[r];
  const int strand = fl & 1, ts = (fl >> 1) & 3;
  const int bi = base_code(seq[c]);
  if (bi >= 0) {
    atomicSub(&planes[(int64_t)(LCR_PL_A + bi) * n_cols + o], 1u);
    if (strand == 0) atomicSub(&planes[(int64_t)(LCR_PL_FWD_A + bi) * n_cols + o], 1u);
  }
  if (ts != 0) atomicSub(&planes[(int64_t)(((strand == 0) == (ts == 1)) ? LCR_PL_TS_FWD : LCR_PL_TS_REV) * n_cols + o], 1u);
}

void launch_k1_zonefix_slots(const BatchView& b, int D, int L, int64_t n_cols, uint32_t* planes, hipStream_t s);
// K1z, one thread per read END (dist_to_end <= 63): the per-offset kernel above loads and scans an overlapping
// 2L+1 byte window for each of the 2D offsets of a read; here a thread loads the <= D + 2L bytes its zone's
// windows can touch once (seven 16-byte loads into registers), finds the homopolymer windows four bytes per step
// (adjacent-byte equality flags -> a 128-bit mask -> L-1 consecutive flags = a window start) and, per run of
// window starts [t, t'] of X in {A,C,G,T}, marks the offsets c in [t-1, t'+L] it masks (the same rule: window
// start in [c-L, c+1]) in a 64-bit mask per base; most read ends have none and stop there.  The marked offsets are
// then walked with a CIGAR cursor (forwards in the leading zone, backwards from the read's reference end in the
// trailing zone).
__device__ __forceinline__ void zonefix_ends_body(const BatchView& b, int id, int D, int L, int64_t n_cols, uint32_t* __restrict__ planes) {
  const int r = id >> 1, end = id & 1;
  if (r >= b.n_reads) return;
  const int seq_len = b.seq_len[r], lead = b.lead[r], reb = seq_len - b.trail[r];
  const int zlo = end == 0 ? lead : max(reb - D + 1, lead + D);
  const int zhi = end == 0 ? min(lead + D, reb) - 1 : reb - 1;
  if (zlo > zhi) return;
  const int lo = max(zlo - L, 0), hi = min(zhi + L, seq_len - 1);
  const int n = hi - lo + 1;                                  // <= 63 + 2 * 16 = 95 bytes
  if (n < L) return;
  const uint8_t* __restrict__ seq = b.bases + b.seq_off[r];
  const long long g0 = (long long)b.seq_off[r] + lo, ga = g0 & ~15ll;
  const int sh = (int)(g0 - ga);
  // the <= 112 bytes [ga, ga + 112) in registers: seven 16-byte loads in flight at once, byte i of the window is read byte lo + i - sh
  uint32_t d[29];
#pragma unroll
  for (int o = 0; o < 7; o++) {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (o * 16 < sh + n) {
      if (ga + o * 16 + 16 <= b.n_bases) v = *reinterpret_cast<const uint4*>(b.bases + ga + o * 16);
      else { uint32_t t[4] = {0, 0, 0, 0}; for (int x = 0; x < 16; x++) if (ga + o * 16 + x < b.n_bases) t[x >> 2] |= (uint32_t)b.bases[ga + o * 16 + x] << (8 * (x & 3)); v = make_uint4(t[0], t[1], t[2], t[3]); }
    }
    d[4 * o] = v.x; d[4 * o + 1] = v.y; d[4 * o + 2] = v.z; d[4 * o + 3] = v.w;
  }
  d[28] = 0;
  // eq bit i: byte i == byte i + 1 (four bytes per step: xor with the stream shifted by one byte, exact zero-byte flags, gathered to a nibble)
  uint32_t ew[4] = {0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < 28; k++) {
    const uint32_t x = d[k] ^ __builtin_amdgcn_alignbyte(d[k + 1], d[k], 1);
    const uint32_t z = ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x | 0x7f7f7f7fu);      // 0x80 in every zero byte of x
    ew[k >> 3] |= (((z >> 7) * 0x10204080u) >> 28) << (4 * (k & 7));
  }
  unsigned long long e0 = (unsigned long long)ew[0] | ((unsigned long long)ew[1] << 32), e1 = (unsigned long long)ew[2] | ((unsigned long long)ew[3] << 32);
  {   // only pairs inside the window: bits [sh, sh + n - 2]
    const int a0 = sh, a1 = sh + n - 2;   // a1 <= 15 + 93 = 108
    const unsigned long long below0 = a0 >= 64 ? ~0ull : ((1ull << a0) - 1ull), below1 = a0 >= 64 ? ((1ull << (a0 - 64)) - 1ull) : 0ull;
    const unsigned long long upto0 = a1 >= 63 ? ~0ull : ((2ull << a1) - 1ull), upto1 = a1 >= 64 ? ((2ull << (a1 - 64)) - 1ull) : 0ull;
    e0 &= upto0 & ~below0; e1 &= upto1 & ~below1;
  }
  // w bit i: a window of L identical bytes starts at byte i  (eq[i] & eq[i+1] & ... & eq[i+L-2])
  unsigned long long w0 = e0, w1 = e1;
  for (int j = 1; j <= L - 2; j++) { w0 &= (e0 >> j) | (e1 << (64 - j)); w1 &= e1 >> j; }
  unsigned long long m[4] = {0ull, 0ull, 0ull, 0ull};   // per base X: offsets c - zlo masked by a window of X
  while ((w0 | w1) != 0ull) {
    // one homopolymer run: its window starts are the consecutive set bits [t0, t0 + cnt)
    const int t0 = w0 != 0ull ? __builtin_ctzll(w0) : 64 + __builtin_ctzll(w1);
    unsigned long long s0, s1;   // the mask shifted down to t0
    if (t0 >= 64) { s0 = w1 >> (t0 - 64); s1 = 0ull; } else if (t0 == 0) { s0 = w0; s1 = w1; } else { s0 = (w0 >> t0) | (w1 << (64 - t0)); s1 = w1 >> t0; }
    const int cnt = ~s0 != 0ull ? __builtin_ctzll(~s0) : 64 + __builtin_ctzll(~s1);
    {   // clear bits [t0, t0 + cnt)
      const int c1 = t0 + cnt;   // exclusive, <= 128
      const unsigned long long lo0 = t0 >= 64 ? ~0ull : ((1ull << t0) - 1ull), lo1 = t0 >= 64 ? ((1ull << (t0 - 64)) - 1ull) : 0ull;
      const unsigned long long hi0 = c1 >= 64 ? ~0ull : ((1ull << c1) - 1ull), hi1 = c1 >= 128 ? ~0ull : c1 > 64 ? ((1ull << (c1 - 64)) - 1ull) : 0ull;
      w0 &= ~(hi0 & ~lo0); w1 &= ~(hi1 & ~lo1);
    }
    const int tr = lo + (t0 - sh);                          // read offset of the first window start
    const uint32_t cur = seq[tr];
    const int x = cur == 'A' ? 0 : cur == 'C' ? 1 : cur == 'G' ? 2 : cur == 'T' ? 3 : -1;
    if (x >= 0) {   // window start t masks c in [t-1, t+L]; starts tr .. tr+cnt-1
      const int c0 = max(tr - 1, zlo) - zlo, c1 = min(tr + cnt - 1 + L, zhi) - zlo;
      if (c0 <= c1) m[x] |= (c1 >= 63 ? ~0ull : ((2ull << c1) - 1ull)) & ~((1ull << c0) - 1ull);
    }
  }
  unsigned long long any = m[0] | m[1] | m[2] | m[3];
  if (any == 0ull) return;
#if defined(ZF_ABL) && ZF_ABL == 1
  if (any != 0x1234567ull) return;
#endif
  const int g = region_of_read(b, r);
  const int vec = b.len[g];
  const uint32_t* __restrict__ cg = b.cigar + b.cig_off[r];
  const int ncig = (int)b.n_cig[r];
  const int fl = b.flags[r];
  const int strand = fl & 1, ts = (fl >> 1) & 3;
  const int64_t cbase = b.col_off[g];
  auto fix = [&](int c, int col) {   // base at read offset c sits on column col: undo its counts if a window of X != ref masks it
    if (col < 0 || col >= vec) return;
    const int64_t o = cbase + col;
#if defined(ZF_ABL) && ZF_ABL == 2
    const uint8_t R = 'N';
#else
    const uint8_t R = b.ref[o];
#endif
    const int k = c - zlo;
    const uint32_t mm = (uint32_t)((m[0] >> k) & 1ull) | ((uint32_t)((m[1] >> k) & 1ull) << 1) | ((uint32_t)((m[2] >> k) & 1ull) << 2) |
                        ((uint32_t)((m[3] >> k) & 1ull) << 3);
    const uint32_t rbit = R == 'A' ? 1u : R == 'C' ? 2u : R == 'G' ? 4u : R == 'T' ? 8u : 0u;
    if ((mm & ~rbit) == 0) return;
    const int bi = base_code(seq[c]);
    if (bi >= 0) {
      atomicSub(&planes[(int64_t)(LCR_PL_A + bi) * n_cols + o], 1u);
      if (strand == 0) atomicSub(&planes[(int64_t)(LCR_PL_FWD_A + bi) * n_cols + o], 1u);
    }
    if (ts != 0) atomicSub(&planes[(int64_t)(((strand == 0) == (ts == 1)) ? LCR_PL_TS_FWD : LCR_PL_TS_REV) * n_cols + o], 1u);
  };
  // one pass over the MARKED offsets only (ascending in the leading zone, descending in the trailing one) with a CIGAR cursor
  // that moves monotonically: the lanes of a wave diverge here, so the trip count is the number of marked bases, not the zone size
  if (end == 0) {
    int i = 0, p = (int)((int64_t)b.pos[r] - b.start0[g]), q = lead > 0 ? lead : 0;
    uint32_t cv = ncig > 0 ? cg[0] : 0u;
    while (any != 0ull && i < ncig) {
      const int c = zlo + __builtin_ctzll(any);
      const int op = (int)(cv & 15u), len = (int)(cv >> 4);
      const bool isM = op == 0 || op == 7 || op == 8;
      const int ql = (isM || op == 1) ? len : 0;
      if (c < q + ql) {                   // the op holding read offset c (an insertion has no column)
        any &= any - 1ull;
        if (isM) fix(c, p + (c - q));
      } else {
        q += ql; p += (isM || op == 2 || op == 3) ? len : 0;
        i++; cv = i < ncig ? cg[i] : 0u;
      }
    }
  } else {          // backwards from the read's reference end (recorded by K0)
    int i = ncig - 1, p = b.read_rend[r], q = reb;   // one past the last column / aligned read offset of op i
    uint32_t cv = ncig > 0 ? cg[ncig - 1] : 0u;
    while (any != 0ull && i >= 0) {
      const int c = zlo + 63 - __builtin_clzll(any);
      const int op = (int)(cv & 15u), len = (int)(cv >> 4);
      const bool isM = op == 0 || op == 7 || op == 8;
      const int ql = (isM || op == 1) ? len : 0;
      if (c >= q - ql) {
        any &= ~(1ull << (c - zlo));
        if (isM) fix(c, p - (q - c));
      } else {
        q -= ql; p -= (isM || op == 2 || op == 3) ? len : 0;
        i--; cv = i >= 0 ? cg[i] : 0u;
      }
    }
  }
}

__global__ void __launch_bounds__(LCR_BLOCK)
k1_zonefix_ends(BatchView b, const ReadBin* __restrict__ rbin, int D, int L, int64_t n_cols, uint32_t* __restrict__ planes) {
  zonefix_ends_body(b, (int)(blockIdx.x * LCR_BLOCK + threadIdx.x), D, L, n_cols, planes);
}
// (measurement switch zonefix_fused; measured SLOWER: the tally group 1.21 -> 1.36 ms on the C4 share -- the pass's atomics and dependent loads queue behind the store stream)
// Round 6 (HiFi presets): the poly-A pass and the record-free tiles' stores in ONE launch.  The pass is instruction-bound (0.33 ms on the C4
// share), the stores are a pure HBM write stream (0.19-0.22 ms) on OTHER tiles (a masked base lies in a tile with records), and as two
// kernels on one queue they ran one after the other (on two queues the second stream lands on a hardware queue of the phase stage's: slower).
// Of every (rt + 1) consecutive workgroups the first takes 256 read ends, the others two record-free tiles each, until the read ends run out.
__global__ void __launch_bounds__(LCR_BLOCK)
k1_zonefix_ends_tiles(BatchView b, int D, int L, int64_t n_cols, uint32_t* __restrict__ planes, const int32_t* __restrict__ tile_region,
                      const int32_t* __restrict__ tile_col0, const int32_t* __restrict__ tile_nbase, const int32_t* __restrict__ order,
                      const int32_t* __restrict__ n_full, int n_tiles, int32_t* __restrict__ flt_count, unsigned int n_zf, unsigned int rt) {
  const unsigned int q = blockIdx.x;
  unsigned int unit;   // pair of record-free tiles, or ~0u: a workgroup of the poly-A pass
  unsigned int zf = 0;
  if (q < n_zf * (rt + 1u)) {
    const unsigned int a = q / (rt + 1u), r = q - a * (rt + 1u);
    if (r == 0) { unit = ~0u; zf = a; } else unit = a * rt + (r - 1u);
  } else unit = n_zf * rt + (q - n_zf * (rt + 1u));
  if (unit == ~0u) { zonefix_ends_body(b, (int)(zf * LCR_BLOCK + threadIdx.x), D, L, n_cols, planes); return; }
  const int nf = *n_full;
  if (*b.error_flag != 0) return;
  const int bt = 2 * (int)unit + (int)(threadIdx.x >> 7);
  if (bt + nf < n_tiles) empty_tile_body(b, tile_region, tile_col0, n_cols, tile_nbase, planes, order, nf, bt, (int)(threadIdx.x & 127), 0, flt_count);
}
// the tally alone + (behind it, one launch) the poly-A pass with the record-free tiles
bool launch_k1_zonefix_tiles(const BatchView& b, int D, int L, int64_t n_cols, uint32_t* planes, const int32_t* tile_region, const int32_t* tile_col0,
                             int32_t n_tiles, const int32_t* tile_nbase, const int32_t* order, const int32_t* tiles_tmp, int32_t* flt_count, hipStream_t s) {
  if (!(b.n_reads > 0 && D > 0 && D <= 63 && L >= 2 && L <= 16 && n_tiles > 0)) return false;
  const unsigned int n_zf = (unsigned int)((2ll * b.n_reads + LCR_BLOCK - 1) / LCR_BLOCK), n_units = (unsigned int)((n_tiles + 1) / 2);
  const unsigned int rt = n_units / n_zf;
  hipLaunchKernelGGL(k1_zonefix_ends_tiles, dim3(n_zf + n_units), dim3(LCR_BLOCK), 0, s, b, D, L, n_cols, planes, tile_region, tile_col0, tile_nbase, order,
                     tiles_tmp + 80, n_tiles, flt_count, n_zf, rt);
  return true;
}
void launch_k1_zonefix(const BatchView& b, const ReadBin* rbin, int D, int L, int64_t n_cols, uint32_t* planes, hipStream_t s) {
  if (b.n_reads > 0 && D > 0 && D <= 63 && L >= 2 && L <= 16) {   // (L = 1: every base is a window, the per-offset kernel)
    const int n = 2 * b.n_reads;
    hipLaunchKernelGGL(k1_zonefix_ends, dim3((unsigned)((n + LCR_BLOCK - 1) / LCR_BLOCK)), dim3(LCR_BLOCK), 0, s, b, rbin, D, L, n_cols, planes);
    return;
  }
  launch_k1_zonefix_slots(b, D, L, n_cols, planes, s);
}
void launch_k1_zonefix_slots(const BatchView& b, int D, int L, int64_t n_cols, uint32_t* planes, hipStream_t s) {
  if (b.n_reads == 0 || D <= 0) return;
  const long long n = (long long)b.n_reads * 2 * D;
  hipLaunchKernelGGL(k1_zonefix, dim3((unsigned)((n + LCR_BLOCK - 1) / LCR_BLOCK)), dim3(LCR_BLOCK), 0, s, b, D, L, n_cols, planes);
}
