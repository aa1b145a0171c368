// k1_pileup.hip — K0 (CIGAR binning) and K1 (pileup tally) for gfx950.
//
// K1 replaces Profile::fill_data_into_freq_vec (reference src/util.rs:621-949).
//
// Design: column-tile gather fed by a work list.
//   K0  one wave64 per read scans the CIGAR 64 ops at a time (wave prefix sums give every op's
//       reference / query start) and
//         * turns every intron (N) run into a +1/-1 pair in a global difference array (prefix-
//           scanned once per batch): introns dominate coverage (mean intron depth 1190 vs allele
//           depth 163 on demo.bam) but need no per-tile work at all,
//         * emits one work item (read, first op of the 64-op chunk, column / read offset there)
//           for every pileup tile that the chunk touches with an M / D / I op; items are counting-
//           sorted by tile (count pass -> scan -> fill pass).
//   K1  one workgroup (16 wave64) owns LCR_TILE consecutive columns of one region and keeps every
//       counter of those columns in LDS.  Each wave pulls 64 items at a time (coalesced), prefetches
//       their read headers lane-parallel, then for each item re-scans the 64-op chunk and
//         * turns D runs and whole M blocks into +1/-1 difference-array updates in LDS (2 atomics
//           per op instead of one per base; prefix-scanned at the end of the tile),
//         * streams the aligned read bases against the tile's reference bytes held in LDS: a base
//           equal to the reference byte and not near a read end needs no further work (its count is
//           "depth - mismatches"); only mismatching, masked (poly-A / homopolymer / ONT end-trim)
//           and non-ACGT bases — a few % of the stream — touch per-column counters.
// All arithmetic is u32 adds => results are bit-exact regardless of order.  HBM traffic: the read
// bases once (each base belongs to exactly one tile), CIGAR words (K0 twice + once per item), the
// work items, and one coalesced write of the 13 count planes.  Base qualities are NOT read here:
// they are only needed at the < 1 % of columns that survive the count filters (k2_hist).
#include <climits>

#include "lcr_dev.h"

#define K1_THREADS 512
#define K1_CPT (LCR_TILE / K1_THREADS)  // columns per thread in the tile epilogue
#define K1_STAGE 1024  // bytes of read bases staged per wave and pass (LDS budget: 2 workgroups per CU)
#define K1_WAVES (K1_THREADS / 64)

// wave64 inclusive add-scan with DPP row shifts / row broadcasts (6 VALU ops, no LDS round trips).
// update_dpp(old = 0, ..., bound_ctrl = false): lanes without a source keep 0, the identity.
__device__ __forceinline__ int wave_incl_scan(int v, int /*lane*/) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return v;
}
__device__ __forceinline__ int wave_incl_max(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v = max(v, t);
  }
  return v;
}

// ---------------------------------------------------------------------------------------------
// read -> region map (one block per region writes its read range)
__global__ void __launch_bounds__(LCR_BLOCK) k0_read_region(const int32_t* __restrict__ read_begin, int32_t* __restrict__ out) {
  const int g = blockIdx.x;
  for (int r = read_begin[g] + threadIdx.x; r < read_begin[g + 1]; r += blockDim.x) out[r] = g;
}
void launch_k0_read_region(const BatchView& b, int32_t* read_region, hipStream_t s) {
  if (b.n_regions == 0) return;
  hipLaunchKernelGGL(k0_read_region, dim3(b.n_regions), dim3(LCR_BLOCK), 0, s, b.read_begin, read_region);
}

// ---------------------------------------------------------------------------------------------
// K0: one wave per read.  pass 0: validate ops, intron difference array, items per tile.
//                          pass 1: write the items (slot = atomic counter per tile).
__global__ void __launch_bounds__(LCR_BLOCK)
k0_bin(BatchView b, int pass, int32_t* __restrict__ tile_count, const int32_t* __restrict__ tile_off,
       int32_t* __restrict__ tile_fill, WorkItem* __restrict__ items, uint32_t* __restrict__ ndiff) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * (LCR_BLOCK / 64) + (threadIdx.x >> 6);
  if (r >= b.n_reads) return;
  const int g = region_of_read(b, r);
  const int vec = b.len[g];
  const int64_t gbase = b.col_off[g] + g;  // one spare slot per region so that end markers never leak
  const int ftile = b.region_first_tile[g];
  const uint32_t ncig = b.n_cig[r];
  const uint32_t* __restrict__ cg = b.cigar + b.cig_off[r];
  const int lead = b.lead[r];
  int ref_cur = (int)((int64_t)b.pos[r] - b.start0[g]);
  int q_cur = lead > 0 ? lead : 0;
  for (uint32_t c0 = 0; c0 < ncig; c0 += 64) {
    if (ref_cur > vec && pass == 1) break;  // nothing at or right of column vec contributes
    const bool act = c0 + lane < ncig;
    const uint32_t word = act ? cg[c0 + lane] : 0u;
    const int op = word & 15, len = (int)(word >> 4);
    const bool is_m = act && (op == 0 || op == 7 || op == 8);
    const bool is_d = act && op == 2, is_n = act && op == 3, is_i = act && op == 1;
    if (pass == 0 && act && !(is_m || is_d || is_n || is_i || op == 4 || op == 5)) atomicExch(b.error_flag, 1);
    const int dr = (is_m || is_d || is_n) ? len : 0;
    const int dq = (is_m || is_i) ? len : 0;
    const int ir = wave_incl_scan(dr, lane), iq = wave_incl_scan(dq, lane);
    const int rs = ref_cur + ir - dr;
    const int a = max(rs, 0), e = min(rs + len, vec);
    if (pass == 0 && is_n && e > a) {  // util.rs:930-942
      atomicAdd(&ndiff[gbase + a], 1u);
      atomicAdd(&ndiff[gbase + e], 0xFFFFFFFFu);
    }
    int tlo = 0, thi = -1;
    if ((is_m || is_d) && len > 0 && e > a) { tlo = a / LCR_TILE; thi = (e - 1) / LCR_TILE; }
    else if (is_i && len > 0 && rs >= 1 && rs < vec) { tlo = thi = (rs - 1) / LCR_TILE; }  // util.rs:918-929
    int prevmax = wave_incl_max(thi, lane);
    prevmax = __shfl_up(prevmax, 1, 64);
    if (lane == 0) prevmax = -1;
    const int first = max(tlo, prevmax + 1);
    for (int t = first; t <= thi; t++) {  // tiles this lane is the first in the chunk to touch
      if (pass == 0) atomicAdd(&tile_count[ftile + t], 1);
      else {
        const int slot = atomicAdd(&tile_fill[ftile + t], 1);
        WorkItem it;
        it.read = (uint32_t)r; it.c0 = c0; it.ref_cur = ref_cur; it.q_cur = q_cur;
        items[tile_off[ftile + t] + slot] = it;
      }
    }
    ref_cur += __shfl(ir, 63, 64);
    q_cur += __shfl(iq, 63, 64);
  }
  // aligned read offsets must lie in [lead, seq_len - trail): the end-zone tests of K1 rely on it
  // (true for every valid BAM record: l_seq = sum of M/I/S/=/X lengths)
  if (pass == 0 && lane == 0 && ncig > 0 && q_cur != b.seq_len[r] - b.trail[r]) atomicExch(b.error_flag, 2);
}

void launch_k0_bin(const BatchView& b, int pass, int32_t* tile_count, const int32_t* tile_off, int32_t* tile_fill,
                   WorkItem* items, uint32_t* ndiff, hipStream_t s) {
  if (b.n_reads == 0) return;
  const int per = LCR_BLOCK / 64;
  hipLaunchKernelGGL(k0_bin, dim3((b.n_reads + per - 1) / per), dim3(LCR_BLOCK), 0, s, b, pass, tile_count, tile_off,
                     tile_fill, items, ndiff);
}

// ---------------------------------------------------------------------------------------------
// K1a (HiFi only): for every read offset c within dist_to_end of a read end, the set of bases X for
// which a window of L identical X starts in [c-L, c+1] (util.rs:754-789).  The base at c is masked
// iff that set contains a base other than the column's reference base.  One thread per (read, slot):
// slot s < D -> c = lead + s;  slot D + s -> c = reb - D + 1 + s  (s < D-1).
__global__ void __launch_bounds__(LCR_BLOCK) k1_hpmask(BatchView b, int D, int L, uint8_t* __restrict__ hp) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int per = 2 * D;
  const int r = (int)(gid / per), s = (int)(gid % per);
  if (r >= b.n_reads) return;
  const int seq_len = b.seq_len[r], lead = b.lead[r], reb = seq_len - b.trail[r];
  const int c = s < D ? lead + s : reb - D + 1 + (s - D);
  uint8_t m = 0;
  if (c >= 0 && c < seq_len && !(s == per - 1)) {
    const uint8_t* __restrict__ seq = b.bases + b.seq_off[r];
    int lo = max(c - L, 0), hi = min(c + L, seq_len - 1);
    if (hi - lo + 1 >= L) {
      int run = 1;
      uint8_t prev = seq[lo];
      for (int i = lo + 1; i <= hi; i++) {
        const uint8_t cur = seq[i];
        run = (cur == prev) ? run + 1 : 1;
        prev = cur;
        if (run >= L) m |= cur == 'A' ? 1 : cur == 'C' ? 2 : cur == 'G' ? 4 : cur == 'T' ? 8 : 0;
      }
    }
  }
  hp[gid] = m;
}
void launch_k1_hpmask(const BatchView& b, int D, int L, uint8_t* hp, hipStream_t s) {
  const long long n = (long long)b.n_reads * 2 * D;
  if (n == 0) return;
  hipLaunchKernelGGL(k1_hpmask, dim3((unsigned)((n + LCR_BLOCK - 1) / LCR_BLOCK)), dim3(LCR_BLOCK), 0, s, b, D, L, hp);
}

// ---------------------------------------------------------------------------------------------
// LDS planes of one tile (u32 each, LCR_TILE + 1 entries so that "end" markers at tile_len fit)
enum {
  P_DIFF_DEPTH_F = 0,  // difference array: kept aligned bases of forward reads
  P_DIFF_DEPTH_R,      //                   ... of reverse reads
  P_DIFF_TS0,          // difference array: transcript_strands[0]
  P_DIFF_TS1,          //                   transcript_strands[1]
  P_DIFF_D,            // deletion runs
  P_NI,                // insertions (plain counter)
  P_MM_F,              // 4 planes: mismatching base counts A,C,G,T of forward reads
  P_MM_R = P_MM_F + 4, // 4 planes: ... of reverse reads
  P_NPL = P_MM_R + 4
};
#define TSTRIDE (LCR_TILE + 1)

// inclusive scan of one int per thread over the whole block
__device__ __forceinline__ int block_incl_scan(int v, int* wsum /* K1_WAVES ints of LDS */) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int s = wave_incl_scan(v, lane);
  if (lane == 63) wsum[w] = s;
  __syncthreads();
  int add = 0;
  for (int i = 0; i < w; i++) add += wsum[i];
  __syncthreads();
  return s + add;
}

__global__ void __launch_bounds__(K1_THREADS)
k1_pileup(BatchView b, DevParams prm, const int32_t* __restrict__ tile_region, const int32_t* __restrict__ tile_col0,
          int64_t n_cols, const int32_t* __restrict__ tile_off, const WorkItem* __restrict__ items,
          const int32_t* __restrict__ nscan, const uint8_t* __restrict__ hp, uint32_t* __restrict__ planes) {
  __shared__ uint32_t pl[P_NPL * TSTRIDE];
  __shared__ __attribute__((aligned(16))) uint8_t refl[LCR_TILE];
  __shared__ int wsum[K1_WAVES];
  __shared__ uint4 stage_all[K1_WAVES][K1_STAGE / 16];  // per-wave staging buffer of read bases
  __shared__ __attribute__((aligned(16))) int lookup_all[K1_WAVES][256];  // per-wave histogram of op start bytes

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = tile_region[blockIdx.x];
  const int tc0 = tile_col0[blockIdx.x];               // first column of the tile inside the region
  const int vec = b.len[g];
  const int tlen = min(LCR_TILE, vec - tc0);
  const int64_t gcol0 = b.col_off[g] + tc0;            // global column of tile column 0

  const int i0 = tile_off[blockIdx.x], i1 = tile_off[blockIdx.x + 1];
  if (i0 == i1 && prm.dbg != 4) {
    // no M / D / I op touches this tile (pure intron or uncovered): every plane is 0 except the
    // intron plane, which comes from the global scan.  Most tiles of a spliced data set are like this.
    for (int col = tid; col < tlen; col += K1_THREADS) {
      const int64_t o = gcol0 + col;
#pragma unroll
      for (int k = 0; k < LCR_NPLANES; k++) planes[(int64_t)k * n_cols + o] = 0u;
      planes[(int64_t)LCR_PL_N * n_cols + o] = (uint32_t)nscan[o + g + 1];
    }
    return;
  }
  for (int i = tid; i < P_NPL * TSTRIDE; i += K1_THREADS) pl[i] = 0;
  for (int i = tid; i < LCR_TILE; i += K1_THREADS) {
    uint8_t R = i < tlen ? b.ref[gcol0 + i] : 0;
    // only upper-case ACGT can equal a read base (htslib decodes to upper case); anything else is
    // stored as 0xFF so that every base at such a column takes the explicit-count path
    refl[i] = (R == 'A' || R == 'C' || R == 'G' || R == 'T') ? R : 0xFF;
  }
  __syncthreads();

  const int D = prm.dist_to_end;

  // items are dealt round-robin to the waves (item i -> wave i % K1_WAVES) so that a tile with a few
  // hundred items keeps every wave busy; each wave fetches 64 of its items at a time, lane-parallel.
  // The per-item work is software-pipelined: the CIGAR words of item k+2 and the read bases of item
  // k+1 are in flight while item k is processed (each item otherwise pays two dependent HBM round trips).
  struct ItemPrep {
    int op, len, rs, qs, a, e;   // this lane's op of the 64-op chunk
    bool is_m;
    int lead, reb, strand, tsidx, q_lo, q_hi, jlo, jhi, n16;
    uint32_t rd_idx;
    unsigned long long mmask0;
    long long seq_abs, w0;
    uint4 pre;                   // lane's 16-byte piece of the first staging window
  };
  uint8_t* const stage_b = reinterpret_cast<uint8_t*>(stage_all[wave]);
  auto load16 = [&](long long off) -> uint4 {
    if (off + 16 <= b.n_bases) return *reinterpret_cast<const uint4*>(b.bases + off);
    uint32_t t[4] = {0, 0, 0, 0};  // last partial 16 bytes of the whole base array
    for (int x = 0; x < 16; x++)
      if (off + x < b.n_bases) t[x >> 2] |= (uint32_t)b.bases[off + x] << (8 * (x & 3));
    return make_uint4(t[0], t[1], t[2], t[3]);
  };
  for (int ibase = i0; ibase < i1 && prm.dbg != 3; ibase += K1_WAVES * 64) {
    const int avail = min(i1 - ibase, K1_WAVES * 64) - wave;                 // items from ibase+wave on
    const int n_here = avail > 0 ? (avail + K1_WAVES - 1) / K1_WAVES : 0;    // ... taking every K1_WAVES-th
    // lane-parallel fetch of up to 64 items and of their read headers
    WorkItem my; my.read = 0; my.c0 = 0; my.ref_cur = 0; my.q_cur = 0;
    if (lane < n_here) my = items[ibase + wave + K1_WAVES * lane];
    const uint32_t rr = my.read;
    const uint32_t h_ncig = b.n_cig[rr];
    const unsigned long long h_cig = b.cig_off[rr], h_seq = b.seq_off[rr];
    const int h_len = b.seq_len[rr], h_lead = b.lead[rr], h_trail = b.trail[rr];
    const int h_fl = b.flags[rr];

    auto load_word = [&](int k) -> uint32_t {
      const uint32_t c0 = __shfl(my.c0, k, 64);
      const uint32_t ncig = __shfl(h_ncig, k, 64);
      const uint32_t* __restrict__ cg = b.cigar + __shfl(h_cig, k, 64);
      return (c0 + lane < ncig) ? cg[c0 + lane] : 0u;
    };
    auto prep = [&](int k, uint32_t word, ItemPrep& st) {
      const int seq_len = __shfl(h_len, k, 64);
      st.lead = __shfl(h_lead, k, 64);
      st.reb = seq_len - __shfl(h_trail, k, 64);
      st.rd_idx = __shfl(rr, k, 64);
      const int fl = __shfl(h_fl, k, 64);
      const int ref_cur = __shfl(my.ref_cur, k, 64) - tc0;  // tile-relative column at op c0
      const int q_cur = __shfl(my.q_cur, k, 64);
      st.strand = fl & 1;
      const int ts = (fl >> 1) & 3;
      // transcript_strands index (util.rs:803-819): (+,+)->0 (+,-)->1 (-,+)->1 (-,-)->0, none -> -1
      st.tsidx = ts == 0 ? -1 : ((st.strand == 0) == (ts == 1) ? 0 : 1);
      st.op = word & 15; st.len = (int)(word >> 4);
      st.is_m = (st.op == 0 || st.op == 7 || st.op == 8) && st.len > 0;
      const bool is_dn = (st.op == 2 || st.op == 3) && st.len > 0;
      const int dr = (st.is_m || is_dn) ? st.len : 0;
      const int dq = (st.is_m || st.op == 1) ? st.len : 0;
      const int ir = wave_incl_scan(dr, lane), iq = wave_incl_scan(dq, lane);
      st.rs = ref_cur + ir - dr;        // tile-relative column where this op starts
      st.qs = q_cur + iq - dq;          // read offset where this op starts
      // clip the op's column range to the tile; ONT: also to the read offsets that survive the end
      // trim (util.rs:745-751), i.e. lead + D <= c <= reb - D, so trimmed bases are never touched
      st.a = max(st.rs, 0); st.e = min(st.rs + st.len, tlen);
      if (prm.ont && st.is_m) {
        st.a = max(st.a, st.rs + (st.lead + D - st.qs));
        st.e = min(st.e, st.rs + (st.reb - D + 1 - st.qs));
      }
      st.mmask0 = prm.dbg == 1 ? 0ull : __ballot(st.is_m && st.e > st.a);
      st.n16 = 0;
      if (st.mmask0 != 0ull) {
        st.jlo = __ffsll((long long)st.mmask0) - 1; st.jhi = 63 - __clzll((long long)st.mmask0);
        st.q_lo = __shfl(st.qs + (st.a - st.rs), st.jlo, 64);   // read range [q_lo, q_hi) compared in this tile
        st.q_hi = __shfl(st.qs + (st.e - st.rs), st.jhi, 64);
        st.seq_abs = (long long)__shfl(h_seq, k, 64);
        st.w0 = (st.seq_abs + st.q_lo) & ~15ll;
        const long long w1 = min(st.w0 + (long long)K1_STAGE, st.seq_abs + (long long)st.q_hi);
        st.n16 = (int)((w1 - st.w0 + 15) >> 4);               // <= 64: one piece per lane
        if (lane < st.n16) st.pre = load16(st.w0 + 16ll * lane);
      }
    };

    ItemPrep cur, nxt;
    uint32_t word_nn = 0;
    if (n_here > 0 && prm.dbg != 2) { prep(0, load_word(0), nxt); if (n_here > 1) word_nn = load_word(1); }
    for (int k = 0; k < n_here && prm.dbg != 2; k++) {
      cur = nxt;
      if (k + 1 < n_here) prep(k + 1, word_nn, nxt);
      if (k + 2 < n_here) word_nn = load_word(k + 2);
      const ItemPrep& st = cur;
      uint32_t* depth_pl = pl + (st.strand ? P_DIFF_DEPTH_R : P_DIFF_DEPTH_F) * TSTRIDE;
      uint32_t* ts_pl = pl + (st.tsidx == 1 ? P_DIFF_TS1 : P_DIFF_TS0) * TSTRIDE;
      uint32_t* mm_pl = pl + (st.strand ? P_MM_R : P_MM_F) * TSTRIDE;
      const int tsidx = st.tsidx, lead = st.lead, reb = st.reb;
      const int rs = st.rs, qs = st.qs, a = st.a, e = st.e, len = st.len, op = st.op;
      if (op == 2 && len > 0 && e > a) {  // util.rs:905-917: +1 per deleted reference position
        atomicAdd(&pl[P_DIFF_D * TSTRIDE + a], 1u);
        atomicAdd(&pl[P_DIFF_D * TSTRIDE + e], 0xFFFFFFFFu);
      }
      if (op == 1 && len > 0) {  // util.rs:918-929: counted on the previous column, 1 <= p < vec
        const int p = rs + tc0;  // pos_in_freq_vec
        if (p >= 1 && p < vec && rs - 1 >= 0 && rs - 1 < tlen) atomicAdd(&pl[P_NI * TSTRIDE + rs - 1], 1u);
      }
      if (st.is_m && e > a) {  // whole block as a range update; per-base corrections follow below
        atomicAdd(&depth_pl[a], 1u);
        atomicAdd(&depth_pl[e], 0xFFFFFFFFu);
        if (tsidx >= 0) { atomicAdd(&ts_pl[a], 1u); atomicAdd(&ts_pl[e], 0xFFFFFFFFu); }
      }
      if (st.mmask0 == 0ull) continue;
      // Compare the read bases of the chunk inside this tile with the reference.  The read range
      // [q_lo, q_hi) is contiguous in the read: it is staged in LDS by 16-byte coalesced loads
      // (first window prefetched above), then walked 4 bytes per lane.
      const int q_lo = st.q_lo, q_hi = st.q_hi;
      const long long seq_abs = st.seq_abs;
      const int jqm = st.is_m ? (rs - qs) : INT_MIN;       // column = read offset + jqm for M ops
      int* hist = lookup_all[wave];                          // 256 ints: ops starting at each staged byte
      const uint32_t* stage32 = reinterpret_cast<const uint32_t*>(stage_b);
      for (long long w0 = st.w0; w0 < seq_abs + q_hi; w0 += K1_STAGE) {
        const long long w1 = min(w0 + (long long)K1_STAGE, seq_abs + (long long)q_hi);
        const int n16 = (int)((w1 - w0 + 15) >> 4);
        if (w0 == st.w0) { if (lane < n16) stage_all[wave][lane] = st.pre; }
        else { if (lane < n16) stage_all[wave][lane] = load16(w0 + 16ll * lane); }
        __builtin_amdgcn_wave_barrier();
        const int sbase = (int)(seq_abs - w0);               // stage index of read offset c is c + sbase
        const int n_dw = (int)((w1 - w0 + 3) >> 2);
        const int kb = qs + sbase;                           // staged byte index where this op starts
        for (int d0 = 0; d0 < n_dw; d0 += 64) {              // uniform trip count: shuffles stay convergent
          const int d = d0 + lane;
          const int cb = 4 * d - sbase;                      // read offset of byte 0 of this dword
          const uint32_t basew = d < n_dw ? stage32[d] : 0u;
          // "last op with qs <= c" for all 256 staged bytes of this pass: histogram of op start bytes
          // (one LDS atomic per op) + prefix sum (4 local adds + one DPP scan)
          reinterpret_cast<int4*>(hist)[lane] = make_int4(0, 0, 0, 0);
          __builtin_amdgcn_wave_barrier();
          const int kk = kb - 4 * d0;
          const int nbefore = __popcll(__ballot(kk <= 0));   // ops that start at or before byte 0 of the pass
          if (kk > 0 && kk < 256) atomicAdd(&hist[kk], 1);
          __builtin_amdgcn_wave_barrier();
          const int4 h4 = reinterpret_cast<const int4*>(hist)[lane];
          const int s0 = h4.x, s1 = s0 + h4.y, s2 = s1 + h4.z, s3 = s2 + h4.w;
          const int excl = wave_incl_scan(s3, lane) - s3 + nbefore - 1;
          const int lo[4] = {excl + s0, excl + s1, excl + s2, excl + s3};
          // per byte: column offset of its op (INT_MIN: not an M op), reference byte, verdict
          uint32_t slow = 0;  // bit i: byte i needs individual attention (mismatch or near a read end)
          int dj[4];
#pragma unroll
          for (int i = 0; i < 4; i++) {
            dj[i] = __shfl(jqm, max(lo[i], 0), 64);          // executed by all lanes
            const int c = cb + i;
            const bool live = d < n_dw && c >= q_lo && c < q_hi && dj[i] != INT_MIN;
            const int col = live ? c + dj[i] : 0;
            const uint32_t R = refl[col];
            const uint32_t base = (basew >> (8 * i)) & 0xffu;
            const bool zone = !prm.ont && (c - lead < D || reb - c < D);  // lead <= c < reb (checked by K0)
            if (live && (base != R || zone)) slow |= 1u << i;
          }
          if (prm.dbg == 5) slow = 0;
          while (__ballot(slow != 0) != 0ull) {              // uniform loop
            if (slow == 0) continue;
            const int i = __ffs(slow) - 1;
            slow &= slow - 1;
            const int c = cb + i;
            const int col = c + (i == 0 ? dj[0] : i == 1 ? dj[1] : i == 2 ? dj[2] : dj[3]);
            const uint32_t base = (basew >> (8 * i)) & 0xffu;
            const uint32_t R = refl[col];
            const bool zone = !prm.ont && (c - lead < D || reb - c < D);
            bool masked = false;
            if (zone) {  // util.rs:754-789 via the precomputed window masks
              const uint32_t hm = hp[(long long)st.rd_idx * (2 * D) + (c - lead < D ? c - lead : D + (c - (reb - D + 1)))];
              const uint8_t Rraw = b.ref[gcol0 + col];
              const uint32_t rbit = Rraw == 'A' ? 1u : Rraw == 'C' ? 2u : Rraw == 'G' ? 4u : Rraw == 'T' ? 8u : 0u;
              masked = (hm & ~rbit) != 0;
            }
            // branch-free classification: 0..3 = mismatching A,C,G,T; otherwise undo depth (masked or non-ACGT)
            const uint32_t h = (base >> 1) & 3u;
            const uint32_t bi = h ^ (h >> 1);                // A,C,G,T -> 0,1,2,3 (either case)
            const bool acgt = ((base & 0xC0u) == 0x40u) && ((0x0010008Au >> (base & 31u)) & 1u);
            if (!masked && acgt) {
              if (base != R) atomicAdd(&mm_pl[bi * TSTRIDE + col], 1u);
            } else {  // masked: contributes nothing (util.rs:801); non-ACGT: no allele count (util.rs:890-892)
              atomicAdd(&depth_pl[col], 0xFFFFFFFFu);
              atomicAdd(&depth_pl[col + 1], 1u);
              if (masked && tsidx >= 0) { atomicAdd(&ts_pl[col], 0xFFFFFFFFu); atomicAdd(&ts_pl[col + 1], 1u); }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __syncthreads();

  // prefix-scan the five difference arrays, K1_CPT consecutive columns per thread
  for (int p = P_DIFF_DEPTH_F; p <= P_DIFF_D; p++) {
    uint32_t* d = pl + p * TSTRIDE;
    int v[K1_CPT], run = 0;
#pragma unroll
    for (int x = 0; x < K1_CPT; x++) { run += (int)d[tid * K1_CPT + x]; v[x] = run; }
    const int excl = block_incl_scan(run, wsum) - run;
#pragma unroll
    for (int x = 0; x < K1_CPT; x++) d[tid * K1_CPT + x] = (uint32_t)(excl + v[x]);
    __syncthreads();
  }

  // assemble the ABI planes and write them out (coalesced: consecutive threads, consecutive columns)
  for (int col = tid; col < tlen; col += K1_THREADS) {
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
    // intron plane: exclusive scan of the global difference array (one spare slot per region)
    planes[(int64_t)LCR_PL_N * n_cols + o] = (uint32_t)nscan[o + g + 1];
    planes[(int64_t)LCR_PL_D * n_cols + o] = pl[P_DIFF_D * TSTRIDE + col];
    planes[(int64_t)LCR_PL_NI * n_cols + o] = pl[P_NI * TSTRIDE + col];
    planes[(int64_t)LCR_PL_TS_FWD * n_cols + o] = pl[P_DIFF_TS0 * TSTRIDE + col];
    planes[(int64_t)LCR_PL_TS_REV * n_cols + o] = pl[P_DIFF_TS1 * TSTRIDE + col];
  }
}

void launch_k1_pileup(const BatchView& b, const DevParams& p, const int32_t* tile_region, const int32_t* tile_col0,
                      int32_t n_tiles, int64_t n_cols, const int32_t* tile_off, const WorkItem* items,
                      const int32_t* nscan, const uint8_t* hp, uint32_t* planes, hipStream_t s) {
  if (n_tiles == 0) return;
  hipLaunchKernelGGL(k1_pileup, dim3(n_tiles), dim3(K1_THREADS), 0, s, b, p, tile_region, tile_col0, n_cols, tile_off,
                     items, nscan, hp, planes);
}
