// k2_candidates.hip — K2: candidate selection + genotype likelihood (gfx950).
//
// Replaces SNPFrag::get_candidate_snps (reference src/candidate.rs:54-463).  Every filter of the
// reference loop is a plain `continue`, so their order does not change the candidate set; we run
// them in two passes:
//   pass 1 (k2_filter, one thread per column, count planes only): depth window, two major alleles,
//           reference validity, low-fraction / low-count, deletion, intron-fraction, strand bias
//           (candidate.rs:90-234 except :174-194).  > 99 % of columns stop here.
//   pass 2 (k2_hist + k2_gt, survivors only): per-allele quality histogram hist[4][31] rebuilt from
//           the reads (same trim / poly-A mask as K1), base-quality filter (candidate.rs:174-194),
//           log10-likelihoods as an exact integer histogram x 31-entry f64 LUT (order-free form of
//           candidate.rs:267-282), posterior / QUAL / GQ (candidate.rs:287-335), classification
//           (candidate.rs:337-460).
// The dense-cluster sweeps (candidate.rs:465-526) run on the device too: k2_dense, a workgroup per region, a thread per start index
// over the region's compacted candidates (below).
#include "lcr_dev.h"
#include "k2_eval.h"

// ---- generic 3-phase exclusive scan over int32 ------------------------------------------------
#define SCAN_ITEMS 4
#define SCAN_TILE (LCR_BLOCK * SCAN_ITEMS)

__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}
__device__ __forceinline__ long long wave_incl_scan_ll(long long v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    long long t = __shfl_up(v, d, 64);
    if (lane >= d) v += t;
  }
  return v;
}

template <typename OutT>
__global__ void __launch_bounds__(LCR_BLOCK) scan_phase1(const int32_t* __restrict__ in, OutT* __restrict__ out, int32_t n,
                                                         long long* __restrict__ block_sum) {
  __shared__ long long wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)tid * SCAN_ITEMS;
  long long v[SCAN_ITEMS], s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { v[k] = (base + k < n) ? (long long)in[base + k] : 0; s += v[k]; }
  long long incl = wave_incl_scan_ll(s, lane);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  long long add = 0;
  for (int i = 0; i < w; i++) add += wsum[i];
  long long run = incl - s + add;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) { if (base + k < n) out[base + k] = (OutT)run; run += v[k]; }
  if (tid == LCR_BLOCK - 1) block_sum[blockIdx.x] = incl + add;
}
__global__ void __launch_bounds__(1024) scan_phase2(long long* __restrict__ block_sum, int32_t n_blocks, long long* total) {
  __shared__ long long wsum[16];
  __shared__ long long carry_s;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < n_blocks; base += 1024) {
    long long v = (base + tid < n_blocks) ? block_sum[base + tid] : 0;
    long long incl = wave_incl_scan_ll(v, lane);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    long long add = carry_s;
    for (int i = 0; i < w; i++) add += wsum[i];
    if (base + tid < n_blocks) block_sum[base + tid] = incl - v + add;
    __syncthreads();
    if (tid == 1023) carry_s = incl + add;
    __syncthreads();
  }
  if (tid == 0 && total) *total = carry_s;
}
template <typename OutT>
__global__ void __launch_bounds__(LCR_BLOCK) scan_phase3(OutT* __restrict__ out, int32_t n, const long long* __restrict__ block_sum,
                                                         int write_last) {
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;
  const long long add = block_sum[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) out[base + k] += (OutT)add;
  (void)write_last;
}
// Phases 2 + 3 + the total in one kernel for scans of up to SCAN_FUSED_BLOCKS blocks (every scan of the stage calls): a block adds up
// the sums of the blocks before it itself (a few hundred coalesced words) instead of waiting for a one-block kernel to scan them --
// five tiny kernels per scan-and-gather become three (each is ~5 us of queue whatever it does).
#define SCAN_FUSED_BLOCKS 2048
template <typename OutT>
__global__ void __launch_bounds__(LCR_BLOCK) scan_phase3x(OutT* __restrict__ out, int32_t n, const long long* __restrict__ block_sum, int32_t n_blocks,
                                                          int32_t* __restrict__ total32, int64_t* __restrict__ total64) {
  __shared__ long long wsum[LCR_BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  long long s = 0;
  for (int j = tid; j < (int)blockIdx.x; j += LCR_BLOCK) s += block_sum[j];
  for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
  if (lane == 0) wsum[w] = s;
  __syncthreads();
  long long add = 0;
#pragma unroll
  for (int i = 0; i < LCR_BLOCK / 64; i++) add += wsum[i];
  const long long base = (long long)blockIdx.x * SCAN_TILE + (long long)tid * SCAN_ITEMS;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; k++) if (base + k < n) out[base + k] += (OutT)add;
  if ((int)blockIdx.x == n_blocks - 1 && tid == 0) {
    const long long tot = add + block_sum[n_blocks - 1];
    if (total32) *total32 = (int32_t)tot;
    if (total64) *total64 = (int64_t)tot;
  }
}
__global__ void write_total_i32(const long long* total, int32_t* dst) { *dst = (int32_t)*total; }
// out[i] = idx[i] < n_src ? src[idx[i]] : *total   (exclusive-scan values at selected positions, e.g. per region)
__global__ void gather_i32(const int32_t* __restrict__ src, const int32_t* __restrict__ idx, int32_t n, int32_t n_src,
                           const int32_t* __restrict__ total, int32_t* __restrict__ out, int32_t* __restrict__ host_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const int32_t v = idx[i] < n_src ? src[idx[i]] : *total; out[i] = v; if (host_out) host_out[i] = v; }
}
// host_out: the same values into pinned host memory (as the device sees it), or nullptr -- the host then needs no copy behind the kernel
void launch_gather_i32(const int32_t* src, const int32_t* idx, int32_t n, int32_t n_src, const int32_t* total, int32_t* out, hipStream_t s, int32_t* host_out) {
  if (n > 0) hipLaunchKernelGGL(gather_i32, dim3((n + 255) / 256), dim3(256), 0, s, src, idx, n, n_src, total, out, host_out);
}
__global__ void write_total_i64(const long long* total, int64_t* dst) { *dst = (int64_t)*total; }

// `tmp` is a caller-owned (per ctx) scratch buffer for the block sums: a ctx is single-threaded, two ctxs never share it
void launch_scan_i32(DevBuf& tmp, const int32_t* in, int32_t* out_excl, int32_t n, int32_t* total, hipStream_t s) {
  const int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  (void)tmp.reserve(((size_t)nb + 2) * sizeof(long long));
  long long* bs = tmp.as<long long>();
  if (n > 0 && nb <= SCAN_FUSED_BLOCKS) {
    hipLaunchKernelGGL(scan_phase1<int32_t>, dim3(nb), dim3(LCR_BLOCK), 0, s, in, out_excl, n, bs);
    hipLaunchKernelGGL(scan_phase3x<int32_t>, dim3(nb), dim3(LCR_BLOCK), 0, s, out_excl, n, bs, nb, total, (int64_t*)nullptr);
  } else if (n > 0) {
    hipLaunchKernelGGL(scan_phase1<int32_t>, dim3(nb), dim3(LCR_BLOCK), 0, s, in, out_excl, n, bs);
    hipLaunchKernelGGL(scan_phase2, dim3(1), dim3(1024), 0, s, bs, nb, bs + nb);
    hipLaunchKernelGGL(scan_phase3<int32_t>, dim3(nb), dim3(LCR_BLOCK), 0, s, out_excl, n, bs, 0);
    if (total) hipLaunchKernelGGL(write_total_i32, dim3(1), dim3(1), 0, s, bs + nb, total);
  } else if (total) {
    (void)hipMemsetAsync(total, 0, sizeof(int32_t), s);
  }
}
// out_excl has n+1 entries; the last receives the total
void launch_scan_i32_to_i64(DevBuf& tmp, const int32_t* in, int64_t* out_excl, int32_t n, hipStream_t s) {
  const int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  (void)tmp.reserve(((size_t)nb + 2) * sizeof(long long));
  long long* bs = tmp.as<long long>();
  if (n > 0 && nb <= SCAN_FUSED_BLOCKS) {
    hipLaunchKernelGGL(scan_phase1<int64_t>, dim3(nb), dim3(LCR_BLOCK), 0, s, in, out_excl, n, bs);
    hipLaunchKernelGGL(scan_phase3x<int64_t>, dim3(nb), dim3(LCR_BLOCK), 0, s, out_excl, n, bs, nb, (int32_t*)nullptr, out_excl + n);
  } else if (n > 0) {
    hipLaunchKernelGGL(scan_phase1<int64_t>, dim3(nb), dim3(LCR_BLOCK), 0, s, in, out_excl, n, bs);
    hipLaunchKernelGGL(scan_phase2, dim3(1), dim3(1024), 0, s, bs, nb, bs + nb);
    hipLaunchKernelGGL(scan_phase3<int64_t>, dim3(nb), dim3(LCR_BLOCK), 0, s, out_excl, n, bs, 0);
    hipLaunchKernelGGL(write_total_i64, dim3(1), dim3(1), 0, s, bs + nb, out_excl + n);
  } else {
    (void)hipMemsetAsync(out_excl, 0, sizeof(int64_t), s);
  }
}

// ---- pass 1 ------------------------------------------------------------------------------------
// (the filters on one column's counts: k2_eval.h, shared with k1_pileup's epilogue)
__global__ void k2_sor_threshold(float* out) { *out = strand_odds_ratio(5, 5, 9, 1); }  // candidate.rs:49-51

__device__ __forceinline__ ColEval eval_column(const uint32_t* __restrict__ planes, int64_t n_cols, int64_t o, uint8_t ref_base,
                                               const DevParams& prm, const BinomTable& bt) {
  uint32_t cnt[4];
#pragma unroll
  for (int k = 0; k < 4; k++) { cnt[k] = planes[(int64_t)(LCR_PL_A + k) * n_cols + o]; }
  return eval_counts(cnt, ref_base, prm, bt,
                     [&]() { return planes[(int64_t)LCR_PL_D * n_cols + o]; }, [&]() { return planes[(int64_t)LCR_PL_N * n_cols + o]; },
                     [&](int k) { return planes[(int64_t)(LCR_PL_FWD_A + k) * n_cols + o]; });
}

__global__ void __launch_bounds__(LCR_BLOCK)
k2_filter(BatchView b, DevParams prm, BinomTable bt, const int32_t* __restrict__ tile_region, const int32_t* __restrict__ tile_col0,
          int64_t n_cols, const uint32_t* __restrict__ planes, const int32_t* __restrict__ tile_fill, uint8_t* __restrict__ flags,
          int32_t* __restrict__ tile_count) {
  __shared__ int cnt_s;
  // a tile without M / D / I records (K0's fill counter) has depth 0 everywhere: no survivor, its planes and flags
  // are never read (k2_compact skips tiles whose count is 0) -- most tiles of spliced reads are like this
  if (tile_fill[blockIdx.x] == 0) { if (threadIdx.x == 0) tile_count[blockIdx.x] = 0; return; }
  const int g = tile_region[blockIdx.x], tc0 = tile_col0[blockIdx.x];
  const int tlen = min(LCR_TILE, b.len[g] - tc0);
  const int64_t gcol0 = b.col_off[g] + tc0;
  if (threadIdx.x == 0) cnt_s = 0;
  __syncthreads();
  int mine = 0;
  for (int col = threadIdx.x; col < tlen; col += LCR_BLOCK) {
    ColEval ev = eval_column(planes, n_cols, gcol0 + col, b.ref[gcol0 + col], prm, bt);
    flags[gcol0 + col] = ev.pass ? 1 : 0;
    mine += ev.pass ? 1 : 0;
  }
  if (mine) atomicAdd(&cnt_s, mine);
  __syncthreads();
  if (threadIdx.x == 0) tile_count[blockIdx.x] = cnt_s;
}

void launch_k2_filter(const BatchView& b, const DevParams& p, const int32_t* tile_region, const int32_t* tile_col0,
                      int32_t n_tiles, int64_t n_cols, const uint32_t* planes, const int32_t* tile_fill, uint8_t* flags,
                      int32_t* tile_count, hipStream_t s) {
  static const BinomTable bt = make_binom_table();
  if (n_tiles == 0) return;
  hipLaunchKernelGGL(k2_filter, dim3(n_tiles), dim3(LCR_BLOCK), 0, s, b, p, bt, tile_region, tile_col0, n_cols, planes,
                     tile_fill, flags, tile_count);
}

// ordered compaction of pass-1 survivors (tile order = (region, column) order)
__global__ void __launch_bounds__(LCR_BLOCK)
k2_compact(BatchView b, DevParams prm, BinomTable bt, const int32_t* __restrict__ tile_region,
           const int32_t* __restrict__ tile_col0, int64_t n_cols, const uint32_t* __restrict__ planes,
           const uint8_t* __restrict__ flags, const int32_t* __restrict__ tile_count, const int32_t* __restrict__ tile_off,
           Survivor* __restrict__ out, int32_t out_cap /* survivors `out` holds: a launch queued before the host knows their number drops the rest */) {
  __shared__ int wsum[4];
  if (tile_count[blockIdx.x] == 0) return;   // (its flags were not even written if the tile holds no records)
  const int g = tile_region[blockIdx.x], tc0 = tile_col0[blockIdx.x];
  const int tlen = min(LCR_TILE, b.len[g] - tc0);
  const int64_t gcol0 = b.col_off[g] + tc0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int base = tile_off[blockIdx.x];
  // thread t owns columns [4t, 4t+4) so that ranks follow column order
  int f[4], c = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) { int col = tid * 4 + k; f[k] = (col < tlen) ? flags[gcol0 + col] : 0; c += f[k]; }
  int incl = wave_incl_scan_i(c, lane);
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int add = 0;
  for (int i = 0; i < w; i++) add += wsum[i];
  int rank = base + incl - c + add;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (!f[k]) continue;
    const int col = tid * 4 + k;
    ColEval ev = eval_column(planes, n_cols, gcol0 + col, b.ref[gcol0 + col], prm, bt);
    Survivor sv;
    sv.gcol = gcol0 + col; sv.region = g; sv.col = tc0 + col;
    sv.ref_base = ev.ref_base; sv.allele1 = ev.allele1; sv.allele2 = ev.allele2; sv.n_alt = ev.n_alt;
    sv.cnt1 = ev.cnt1; sv.cnt2 = ev.cnt2; sv.depth = ev.depth; sv.af1 = ev.af1; sv.af2 = ev.af2;
    sv.ts_fwd = planes[(int64_t)LCR_PL_TS_FWD * n_cols + gcol0 + col];
    sv.ts_rev = planes[(int64_t)LCR_PL_TS_REV * n_cols + gcol0 + col];
    if (rank < out_cap) out[rank] = sv;
    rank++;
  }
}


// ---- pass 2a: per-survivor quality histograms --------------------------------------------------
// Sixteen lanes per read (row16_walk_sites, lcr_dev.h): the region's survivors (sorted by column) that
// fall inside the read's reference span are located against the CIGAR spread over the lanes, and
// (allele, clamped q) of every *kept* base (same trim / poly-A mask as K1) is added to
// hist[s][allele][q].  Reads that cover no survivor never load their CIGAR.
// The walk also leaves what K3 needs (fragment.rs:93-194 walks the same reads against the candidates, a subset of these
// survivors): per read the first LCR_HITS (survivor, base, raw quality) hits in survivor order -- every aligned base that faces a
// survivor, masked or not (the fragment walk takes every aligned base) -- and the hit count; reads with more hits than the list
// holds go to a list of their own (k3 walks those again).
// Round 6: the chain of dependent loads per read is four deep instead of seven -- read header + reference end | survivor offsets of the
// first / last TILE of the read's span (tile_off: k2_compact's offsets; no region look-up, no scan from the region's first survivor) beside
// the read's first 64 CIGAR ops, requested before anyone knows whether the read covers a survivor | the survivors' columns | base + quality.
__global__ void __launch_bounds__(LCR_BLOCK)
k2_hist(BatchView b, DevParams prm, const ReadBin* __restrict__ rbin, const Survivor* __restrict__ sv,
        const int32_t* __restrict__ tile_off, int32_t n_tiles, int32_t n_sv, uint32_t* __restrict__ hist, int32_t* __restrict__ hit_cnt, uint2* __restrict__ hit_list,
        int32_t* __restrict__ ovf_cnt, int32_t* __restrict__ ovf_list) {
  const int r = (blockIdx.x * LCR_BLOCK + threadIdx.x) >> 4;
  const bool live = r < b.n_reads;
  const int rr = live ? r : 0;
  const ReadBin h = rbin[rr];
  const int rend = b.read_rend[rr];
  const uint8_t* __restrict__ seq = b.bases + h.seq_off;
  const uint8_t* __restrict__ qual = b.quals + h.seq_off;
  const int l16 = threadIdx.x & 15, rbase = threadIdx.x & 48;
  // survivors of the tiles the read's span [max(rel_pos, 0), min(rend, vec)) touches: a contiguous index range (survivors are ordered
  // by tile, then by column)
  const int key = h.rel_pos > 0 ? h.rel_pos : 0, rend_c = min(rend, h.vec);
  const bool span = live && rend_c > key;
  const int t0 = h.ftile + (int)((unsigned int)key / (unsigned int)LCR_TILE), t1 = h.ftile + (int)((unsigned int)(max(rend_c, 1) - 1) / (unsigned int)LCR_TILE) + 1;
  uint32_t pre[4];
  {
    const uint32_t* __restrict__ cg = b.cigar + h.cig_off;
#pragma unroll
    for (int k = 0; k < 4; k++) pre[k] = (span && (uint32_t)(16 * k + l16) < (uint32_t)h.n_cig) ? cg[16 * k + l16] : 0u;
  }
  const int s_lo = span ? tile_off[t0] : 0, s_hi = span ? (t1 < n_tiles ? tile_off[t1] : n_sv) : 0;
  int nh = 0;   // hits of this read so far (row-uniform)
  uint2* const hl = hit_cnt ? hit_list + (size_t)rr * LCR_HITS : nullptr;   // (hit_cnt == nullptr: no hit lists asked for)
  int cur, s_end;
  auto site_col = [&](int i) { return sv[i].col; };
  row16_find_sites(span && s_lo < s_hi, h, rend, s_lo, s_hi, site_col, &cur, &s_end);
  row16_walk_range(b, h, cur, s_end, pre, site_col,
    [&](int cur_s, int c, bool hit) {   // lane <-> survivor
      uint8_t base = 0, rq = 0;
      if (hit) { base = seq[c]; rq = qual[c]; }
      if (hl) {
        const unsigned int hm = (unsigned int)(__ballot(hit) >> rbase) & 0xffffu;
        const int at = nh + __popc(hm & ((1u << l16) - 1u));
        if (hit && at < LCR_HITS) hl[at] = make_uint2((uint32_t)cur_s, (uint32_t)base | ((uint32_t)rq << 8));
        nh += __popc(hm);
      }
      if (!hit) return;
      const uint8_t bq = rq < 30 ? rq : 30;  // MAX_BASE_QUALITY (util.rs:711-715)
      bool masked = false;
      if (in_end_zone(c, h.lead, h.reb, prm.dist_to_end))
        masked = prm.ont ? true : polya_masked(seq, b.seq_len[rr], c, prm.polya_len, sv[cur_s].ref_base);
      const int bi = base_code(base);
      if (!masked && bi >= 0) atomicAdd(&hist[((int64_t)cur_s * 4 + bi) * 31 + bq], 1u);
    });
  if (hl && live && l16 == 0) {
    hit_cnt[r] = nh;
    if (nh > LCR_HITS) ovf_list[atomicAdd(ovf_cnt, 1)] = r;
  }
}

void launch_k2_hist(const BatchView& b, const DevParams& p, const ReadBin* rbin, const Survivor* sv,
                    const int32_t* tile_off, int32_t n_tiles, int32_t n_sv, uint32_t* hist, int32_t* hit_cnt, void* hit_list, int32_t* ovf_cnt, int32_t* ovf_list, hipStream_t s) {
  if (b.n_reads == 0) return;
  const int per = LCR_BLOCK / 16;
  hipLaunchKernelGGL(k2_hist, dim3((b.n_reads + per - 1) / per), dim3(LCR_BLOCK), 0, s, b, p, rbin, sv, tile_off, n_tiles, n_sv, hist, hit_cnt, (uint2*)hit_list,
                     ovf_cnt, ovf_list);
}

// ---- pass 2a, tile form.  When the survivors are DENSE (C5: every covered column of a 500x ONT-dRNA island passes the count
// filters: 2.2e5 survivors, 1.1e8 (read, survivor) observations, 47 ms of the walk above) the histograms are a second pileup, and
// K0's per-tile records are still there.  HT_SPLIT workgroups per tile, each with its share of the tile's records, keep
// hist[survivor][allele][q] as u16 pairs in LDS (HT_SLOTS survivors x 124 counters per pass over their records -- every survivor of a tile in one pass; a survivor's depth
// is <= max_depth <= 65 535) and add their non-zero counters to the zeroed histograms at the end.  Eight lanes per record, four
// consecutive bases per lane and load (round 5; before: sixteen lanes, a byte each).  What this replaced, measured on C5: a workgroup per tile
// with a thread per record 26 ms, with sixteen lanes per record and 64 slots 30 ms -- the island's bases sit in 845 of its 3 843
// tiles (6.4e5 bases each, up to four passes), so a workgroup per tile left most CUs idle behind a few long dependent chains.
// ONT presets only: their end trim is already cut out of the records, the HiFi poly-A mask is not (those presets keep the walk).
#ifndef HT_SLOTS
#define HT_SLOTS 256
#endif
#ifndef HT_SPLIT
#define HT_SPLIT 32   // (C5, round 4, 256 threads and 128 slots: 8 / 32 / 64 workgroups per tile 10.6 / 8.9 / 9.3 ms; 64 slots per pass: 10.0.  Round 5: 256 slots with
                      // 256 threads 12.7 ms (two workgroups per CU), with 512 threads 8.3 (split 16: 8.4; 1 024 threads, split 8: 10.8); the tally without branches 5.9; a dword load of four bases per lane 4.6; eight lanes per record instead of sixteen 3.8 (four: 3.7))
#endif
#ifndef HT_BLOCK
#define HT_BLOCK 512
#endif
__global__ void __launch_bounds__(HT_BLOCK)
k2_hist_tiles(BatchView b, const int32_t* __restrict__ tile_col0, const int32_t* __restrict__ tile_count, const int32_t* __restrict__ tile_off,
              const Survivor* __restrict__ sv, const int32_t* __restrict__ ent_off, const uint2* __restrict__ ents,
              const unsigned long long* __restrict__ recs, uint32_t* __restrict__ hist) {
  const int tile = blockIdx.x / HT_SPLIT, part = blockIdx.x % HT_SPLIT, cnt = tile_count[tile];
  if (cnt == 0) return;
  __shared__ uint32_t hp[HT_SLOTS * 62];           // counter (slot, allele, q) = half (slot * 124 + allele * 31 + q) & 1 of word >> 1
  __shared__ uint16_t slot_of[LCR_TILE];           // tile column -> survivor slot of this pass, 0xFFFF: none
  constexpr int HT_LANES = 8;   // lanes per record, four bases each (median M run of an ONT read: 26 bases)
  const int tid = threadIdx.x, grp = tid / HT_LANES, ln = tid % HT_LANES;
  const int s0 = tile_off[tile], tc0 = tile_col0[tile];
  const int e0 = ent_off[tile];
  const int n_ent = ent_off[tile + 1] - e0;
  const int j_lo = 16 * (int)((int64_t)n_ent * part / HT_SPLIT), j_hi = 16 * (int)((int64_t)n_ent * (part + 1) / HT_SPLIT);   // this part's record slots
  if (j_lo >= j_hi) return;
  for (int p0 = 0; p0 < cnt; p0 += HT_SLOTS) {
    const int pc = min(HT_SLOTS, cnt - p0);
    for (int i = tid; i < LCR_TILE; i += HT_BLOCK) slot_of[i] = 0xFFFFu;
    for (int i = tid; i < pc * 62; i += HT_BLOCK) hp[i] = 0u;
    __syncthreads();
    for (int i = tid; i < pc; i += HT_BLOCK) slot_of[sv[s0 + p0 + i].col - tc0] = (uint16_t)i;
    __syncthreads();
    const int c_lo = sv[s0 + p0].col - tc0, c_hi = sv[s0 + p0 + pc - 1].col - tc0;   // the pass's column range (survivors are in column order)
    // a 16-lane group per record; the records two iterations ahead are requested while this one is tallied
    auto fetch = [&](int j) -> unsigned long long {
      if (j >= j_hi) return ~0ull;
      const uint2 en = ents[e0 + (j >> 4)];
      return ((unsigned int)j & 15u) < en.y ? recs[en.x + ((unsigned int)j & 15u)] : ~0ull;
    };
    unsigned long long nx = fetch(j_lo + grp), nx2 = fetch(j_lo + grp + HT_BLOCK / HT_LANES);
    for (int j = j_lo + grp; j < j_hi; j += HT_BLOCK / HT_LANES) {
      const unsigned long long rec = nx;
      nx = nx2;
      nx2 = fetch(j + 2 * (HT_BLOCK / HT_LANES));
      const unsigned long long off = rec & REC_OFF_MASK;
      const bool live = off < REC_KIND_N;          // (else: idle slot, D / I / N record: no base -- an empty range, no branch)
      const int col0 = (int)((rec >> 40) & 1023u), len = (int)((rec >> 50) & 1023u) + 1;
      const int d_lo = max(0, c_lo - col0), d_hi = live ? min(len - 1, c_hi - col0) : -1;
      // (no branch before the atomic: a switch on the base and early returns cost twice as many scalar as vector instructions here)
      auto tally = [&](int dd, bool in, uint32_t base, uint32_t qual) {
        const uint32_t slot = slot_of[min(col0 + max(dd, 0), LCR_TILE - 1)];
        const uint32_t u = (base & 0xDFu) - 65u;                      // 'A' 'C' 'G' 'T' (either case, util.rs:822-889) -> 0, 2, 6, 19
        const bool acgt = u < 20u && ((0x80045u >> u) & 1u);
        const uint32_t k = ((base & 0xDFu) >> 1) & 3u, bi = k ^ (k >> 1);   // -> 0, 1, 2, 3
        const uint32_t q = min(qual, 30u);   // MAX_BASE_QUALITY (util.rs:711-715)
        const uint32_t idx = slot * 124u + bi * 31u + q;
        if (in && acgt && slot != 0xFFFFu) atomicAdd(&hp[idx >> 1], 1u << (16 * (idx & 1u)));
      };
      // a lane takes FOUR consecutive bases with one (unaligned) dword load each of the bases and the qualities -- two memory instructions
      // per four bases instead of eight; the last three bytes of the batch's arrays are read by byte
      for (int d0 = d_lo + 4 * ln; d0 <= d_hi; d0 += 4 * HT_LANES) {
        const unsigned long long at = off + (unsigned long long)d0;
        uint32_t bw, qw;
        if (at + 4 <= (unsigned long long)b.n_bases) {
          bw = *reinterpret_cast<const uint32_t*>(b.bases + at); qw = *reinterpret_cast<const uint32_t*>(b.quals + at);
        } else {
          bw = 0; qw = 0;
          for (int t = 0; t < 4; t++) if (at + t < (unsigned long long)b.n_bases) { bw |= (uint32_t)b.bases[at + t] << (8 * t); qw |= (uint32_t)b.quals[at + t] << (8 * t); }
        }
#pragma unroll
        for (int t = 0; t < 4; t++) tally(d0 + t, d0 + t <= d_hi, (bw >> (8 * t)) & 0xFFu, (qw >> (8 * t)) & 0xFFu);
      }
    }
    __syncthreads();
    uint32_t* out = hist + (int64_t)(s0 + p0) * 124;
    for (int i = tid; i < pc * 124; i += HT_BLOCK) {
      const uint32_t v = (hp[i >> 1] >> (16 * (i & 1))) & 0xffffu;
      if (v) atomicAdd(&out[i], v);
    }
    __syncthreads();
  }
}

void launch_k2_hist_tiles(const BatchView& b, const int32_t* tile_col0, int32_t n_tiles, const int32_t* tile_count,
                          const int32_t* tile_off, const Survivor* sv, const int32_t* ent_off, const void* ents, const unsigned long long* recs,
                          uint32_t* hist /* zeroed */, hipStream_t s) {
  if (n_tiles == 0) return;
  hipLaunchKernelGGL(k2_hist_tiles, dim3((unsigned)n_tiles * HT_SPLIT), dim3(HT_BLOCK), 0, s, b, tile_col0, tile_count, tile_off, sv, ent_off,
                     (const uint2*)ents, recs, hist);
}

// ---- pass 2b: genotype likelihood + classification --------------------------------------------
struct GtConst {
  double le[31], l1e[31];   // log10(e_q), log10(1-e_q) with e_q = 0.1^(q/10)  (candidate.rs:268)
  double log10_2;
  double log_prior[3];      // log10 of (theta/2, theta, 1-1.5 theta)
};
static GtConst make_gt_const() {
  GtConst c;
  for (int q = 0; q <= 30; q++) {
    double e = pow(0.1, (double)q / 10.0);
    c.le[q] = log10(e);
    c.l1e[q] = log10(1.0 - e);
  }
  c.log10_2 = log10(2.0);
  const double theta = 0.001;
  c.log_prior[0] = log10(theta / 2.0); c.log_prior[1] = log10(theta); c.log_prior[2] = log10(1.0 - 1.5 * theta);
  return c;
}

__global__ void __launch_bounds__(LCR_BLOCK)
k2_gt(DevParams prm, GtConst gc, const Survivor* __restrict__ sv, int32_t n_sv, const uint32_t* __restrict__ hist,
      const int64_t* __restrict__ start0, lcr_candidate* __restrict__ out, int32_t* __restrict__ keep) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n_sv) return;
  const Survivor v = sv[s];
  const uint32_t* h = hist + (int64_t)s * 124;
  keep[s] = 0;
  const int ri = base_code(v.ref_base);
  // base-quality filter (candidate.rs:174-194): first non-reference major allele needs >= 2 quals >= min_baseq
  {
    int ai = -1; uint32_t ac = 0;
    if (v.allele1 != v.ref_base) { ai = base_code(v.allele1); ac = v.cnt1; }
    else if (v.allele2 != v.ref_base) { ai = base_code(v.allele2); ac = v.cnt2; }
    if (ai >= 0) {
      uint32_t pass = 0;
      for (uint32_t q = prm.min_baseq; q <= 30; q++) pass += h[ai * 31 + q];
      if (ac > 0 && pass < 2) return;
    }
  }
  // log10 likelihoods: integer histogram x LUT, q ascending; zero counts are skipped so that
  // q = 0 (log10(1-1) = -inf) never produces 0 * -inf
  double l0 = 0.0, l2 = 0.0;
  uint32_t num_reads = 0;
  for (int q = 0; q <= 30; q++) {
    const uint32_t hm = h[ri * 31 + q];
    uint32_t hx = 0;
#pragma unroll
    for (int bq = 0; bq < 4; bq++) if (bq != ri) hx += h[bq * 31 + q];
    num_reads += hm + hx;
    if (hm) { l0 += (double)hm * gc.le[q]; l2 += (double)hm * gc.l1e[q]; }
    if (hx) { l0 += (double)hx * gc.l1e[q]; l2 += (double)hx * gc.le[q]; }
  }
  double loglik[3] = {l0, 0.0, l2};
  loglik[1] -= (double)num_reads * gc.log10_2;
  // posterior / QUAL / GQ (candidate.rs:287-335)
  double logprob[3] = {loglik[0] + gc.log_prior[0], loglik[1] + gc.log_prior[1], loglik[2] + gc.log_prior[2]};
  const double mlp = fmax(fmax(logprob[0], logprob[1]), logprob[2]);
  double vp[3] = {pow(10.0, logprob[0] - mlp), pow(10.0, logprob[1] - mlp), pow(10.0, logprob[2] - mlp)};
  const double svp = vp[0] + vp[1] + vp[2];
  vp[2] = vp[2] / svp;
  const double variant_quality = -10.0 * log10(fmax(10e-301, vp[2]));
  const double ml = fmax(fmax(loglik[0], loglik[1]), loglik[2]);
  double gl[3] = {pow(10.0, loglik[0] - ml), pow(10.0, loglik[1] - ml), pow(10.0, loglik[2] - ml)};
  const double sgl = gl[0] + gl[1] + gl[2];
  double gp[3] = {gl[0] / sgl, gl[1] / sgl, gl[2] / sgl};
  double ph[3] = {-10.0 * log10(gp[0]), -10.0 * log10(gp[1]), -10.0 * log10(gp[2])};
  for (int i = 1; i < 3; i++) {  // insertion sort, is_less = (a < b)  (Rust sort_by on 3 elements)
    double x = ph[i];
    int j = i;
    while (j > 0 && x < ph[j - 1]) { ph[j] = ph[j - 1]; j--; }
    ph[j] = x;
  }
  const double genotype_quality = ph[1] - ph[0];

  lcr_candidate c;
  c.pos = start0[v.region] + v.col;  // 0-based reference position (candidate.rs:74,339)
  c.region = v.region;
  c.ref_base = v.ref_base; c.allele1 = v.allele1; c.allele2 = v.allele2; c.n_alt = v.n_alt;
  c.cnt1 = v.cnt1; c.cnt2 = v.cnt2; c.depth = v.depth; c.af1 = v.af1; c.af2 = v.af2;
  c.haplotype = 0; c.phase_set = 0; c.phase_score = 0.0;
  for (int k = 0; k < 3; k++) { c.loglik[k] = loglik[k]; c.gt_prob[k] = gp[k]; }
  c.qual = variant_quality; c.gq = genotype_quality;
  if (gp[0] > gp[1] && gp[0] > gp[2]) { c.variant_type = 2; c.genotype = -1; }
  else if (gp[1] > gp[0] && gp[1] > gp[2]) { c.variant_type = 1; c.genotype = 0; }
  else { c.variant_type = 0; c.genotype = 1; }
  c.flags = 0;
  if (variant_quality < (double)prm.min_qual) return;  // candidate.rs:374
  // classification (candidate.rs:379-460)
  uint8_t ref_allele_base, alt0; float altf0, altf1 = 0.0f;
  if (v.allele1 == v.ref_base) { ref_allele_base = v.allele1; alt0 = v.allele2; altf0 = v.af2; }
  else if (v.allele2 == v.ref_base) { ref_allele_base = v.allele2; alt0 = v.allele1; altf0 = v.af1; }
  else { ref_allele_base = v.ref_base; alt0 = v.allele1; altf0 = v.af1; altf1 = v.af2; }
  const int fwd_t = (int)v.ts_fwd, rev_t = (int)v.ts_rev;
  bool kept = false;
  if (ref_allele_base == 'A' && alt0 == 'G' && (fwd_t > rev_t * 2 || (fwd_t == 0 && rev_t == 0)) && c.variant_type != 2) {
    c.flags = LCR_F_RNA_EDIT; kept = true;
  } else if (ref_allele_base == 'T' && alt0 == 'C' && (rev_t > fwd_t * 2 || (fwd_t == 0 && rev_t == 0)) && c.variant_type != 2) {
    c.flags = LCR_F_RNA_EDIT; kept = true;
  } else if (v.n_alt == 1 && altf0 < prm.min_af) {
    c.flags = LCR_F_CAND_SOMATIC; kept = true;
  } else if (c.variant_type == 2) {
    if (v.n_alt == 2 && altf0 >= prm.min_af && altf1 >= prm.min_af) { c.variant_type = 3; c.genotype = -1; }
    c.flags = LCR_F_HOM | LCR_F_FOR_PHASING; kept = true;
  } else if (c.variant_type == 1) {
    if (v.n_alt == 2) { c.variant_type = 3; c.genotype = -1; c.flags = LCR_F_HOM | LCR_F_FOR_PHASING; }
    else c.flags = LCR_F_HET | LCR_F_FOR_PHASING;
    kept = true;
  }
  if (!kept) return;  // variant_type 0 (candidate.rs:457-460)
  out[s] = c;
  keep[s] = 1;
}

void launch_k2_gt(const DevParams& p, const Survivor* sv, int32_t n_sv, const uint32_t* hist, const int64_t* start0,
                  lcr_candidate* out, int32_t* keep, hipStream_t s) {
  static const GtConst g_gtc = make_gt_const();  // host libm values: the device only adds/multiplies them
  if (n_sv == 0) return;
  hipLaunchKernelGGL(k2_gt, dim3((n_sv + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, p, g_gtc, sv, n_sv, hist, start0, out, keep);
}

void launch_k2_compact(const BatchView& b, const DevParams& p, const int32_t* tile_region, const int32_t* tile_col0,
                       int32_t n_tiles, int64_t n_cols, const uint32_t* planes, const uint8_t* flags,
                       const int32_t* tile_count, const int32_t* tile_off, Survivor* out, int32_t out_cap, hipStream_t s) {
  static const BinomTable bt = make_binom_table();
  if (n_tiles == 0) return;
  hipLaunchKernelGGL(k2_compact, dim3(n_tiles), dim3(LCR_BLOCK), 0, s, b, p, bt, tile_region, tile_col0, n_cols,
                     planes, flags, tile_count, tile_off, out, out_cap);
}

float lcr_device_sor_threshold(hipStream_t s) {
  float* d = nullptr;
  float h = 0.f;
  if (hipMalloc(&d, sizeof(float)) != hipSuccess) return 0.f;
  hipLaunchKernelGGL(k2_sor_threshold, dim3(1), dim3(1), 0, s, d);
  (void)hipMemcpyAsync(&h, d, sizeof(float), hipMemcpyDeviceToHost, s);
  (void)hipStreamSynchronize(s);
  (void)hipFree(d);
  return h;
}

// ---- pass 3: ordered compaction of the kept candidates and the dense-cluster sweep ---------------
// kept survivor s -> candidate slot pos[s] (exclusive scan of keep); a region's candidates stay contiguous
__global__ void __launch_bounds__(LCR_BLOCK)
k2_scatter(const lcr_candidate* __restrict__ tmp, const int32_t* __restrict__ keep, const int32_t* __restrict__ pos, int32_t n_sv,
           lcr_candidate* __restrict__ out, const int32_t* __restrict__ sv_region_off, int32_t n_regions, int32_t* __restrict__ cand_off) {
  // 128-byte records: eight 16-byte words per record, one word per thread (coalesced both ways)
  const int64_t id = (int64_t)blockIdx.x * LCR_BLOCK + threadIdx.x;
  // (the candidates' region offsets on the side: cand_off[g] = pos[sv_region_off[g]] -- what a gather kernel of its own did)
  if (id <= n_regions) cand_off[id] = sv_region_off[id] < n_sv ? pos[sv_region_off[id]] : pos[n_sv];
  const int s = (int)(id >> 3), w = (int)(id & 7);
  if (s >= n_sv || !keep[s]) return;
  reinterpret_cast<uint4*>(out + pos[s])[w] = reinterpret_cast<const uint4*>(tmp + s)[w];
}
// candidate.rs:465-526, one thread per region over its candidates [lo, hi) in position order: windows of
// >= min_dense_cnt het/hom candidates within dense_win bp, and of >= 3 within 5 bp (`tk in i..j` excludes j),
// are marked dense and taken out of phasing.  idx = scratch list of the region's het/hom candidates.
__global__ void __launch_bounds__(256)
k2_dense(lcr_candidate* __restrict__ cand, const int32_t* __restrict__ cand_off, int32_t n_regions, int32_t* __restrict__ idx,
         uint32_t dense_win, uint32_t min_dense_cnt, lcr_candidate* __restrict__ h_cand, int32_t* __restrict__ h_off) {
  // The two dense-cluster sweeps of candidate.rs:465-526, one workgroup per region.  The reference's loops are sequential, but
  // what they do is order-free: every start index i marks ONE interval [i, e_i) of the het / hom list (flags |= DENSE, &= ~FOR_PHASING:
  // idempotent), so a thread per start index finds its interval with a forward scan and marks it with atomics.  (One thread per
  // region took 7.4 ms on C5's 4 687 candidates.)
  __shared__ int wcnt[4], base_s;
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int lo = cand_off[g], hi = cand_off[g + 1];
  int32_t* concat = idx + lo;
  // ordered compaction of the het / hom candidates
  if (tid == 0) base_s = 0;
  __syncthreads();
  for (int i0 = lo; i0 < hi; i0 += 256) {
    const int i = i0 + tid;
    const bool keep = i < hi && (cand[i].flags & (LCR_F_HOM | LCR_F_HET)) != 0;
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wcnt[w] = __popcll(m);
    __syncthreads();
    int before = base_s;
    for (int k = 0; k < w; k++) before += wcnt[k];
    if (keep) concat[before + __popcll(m & ((1ull << lane) - 1ull))] = i;
    __syncthreads();
    if (tid == 0) base_s += wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
    __syncthreads();
  }
  const int n = base_s;
  auto mark = [&](int i, int j) {   // `tk in i..j`: the last index is not marked (candidate.rs:478, :510)
    for (int tk = i; tk < j; tk++) { atomicOr(&cand[concat[tk]].flags, (uint32_t)LCR_F_DENSE); atomicAnd(&cand[concat[tk]].flags, ~(uint32_t)LCR_F_FOR_PHASING); }
  };
  // sweep(i, W, cnt): the reference walks j = i, i + 1, ...: at the first j with pos[j] - pos[i] > W it marks [i, j) if j - i >= cnt and
  // stops; if it reaches j = n - 1 without that, it marks [i, n - 1) if n - i >= cnt
  auto sweep = [&](int i, int64_t W, uint32_t cnt) {
    const int64_t start_pos = cand[concat[i]].pos;
    int j = i;
    while (j < n && cand[concat[j]].pos - start_pos <= W) j++;
    if (j < n) { if ((uint32_t)(j - i) >= cnt) mark(i, j); }
    else if ((uint32_t)(n - i) >= cnt) mark(i, n - 1);
  };
  for (int i = tid; i < n; i += 256) {
    sweep(i, (int64_t)dense_win, min_dense_cnt);   // `diff > dense_win_size`
    sweep(i, 4, 3u);                               // `diff >= 5`, three or more
  }
  // The region's records are final: they and the region's offset leave for pinned host memory here (h_cand != nullptr).  The host cannot
  // size a copy before it has the count, and a copy of the records' CAPACITY (every survivor kept: 3 MB on C3) on a second queue held up
  // the fragment stage's first kernel for as long as it ran -- a kernel's end-of-kernel release waits for the device's writes to host
  // memory in flight (30 us per step).
  if (!h_cand) return;
  __syncthreads();
  const uint4* src = reinterpret_cast<const uint4*>(cand + lo);
  uint4* dst = reinterpret_cast<uint4*>(h_cand + lo);
  const int n16 = (hi - lo) * (int)(sizeof(lcr_candidate) / 16);
  for (int i = tid; i < n16; i += 256) dst[i] = src[i];
  if (tid == 0) { h_off[g] = lo; if (g == n_regions - 1) h_off[n_regions] = hi; }
}
void launch_k2_finish(DevBuf& scan_tmp, const lcr_candidate* tmp, const int32_t* keep, int32_t n_sv, const int32_t* sv_region_off,
                      int32_t n_regions, int32_t* pos /* n_sv + 1 */, int32_t* idx /* n_sv */, lcr_candidate* out,
                      int32_t* cand_off /* n_regions + 1 */, uint32_t dense_win, uint32_t min_dense_cnt, hipStream_t s,
                      lcr_candidate* h_cand /* pinned (device pointer): the records and ... */, int32_t* h_off /* ... their offsets, or nullptr */) {
  static_assert(sizeof(lcr_candidate) % 16 == 0, "k2_dense exports 16-byte words");
  // pos = exclusive scan of keep, pos[n_sv] = number of candidates; cand_off[g] = pos[sv_region_off[g]]
  launch_scan_i32(scan_tmp, keep, pos, n_sv, pos + n_sv, s);
  if (n_sv > 0) {   // (cand_off[g] = pos[sv_region_off[g]] rides on the scatter kernel; without survivors a gather of zeros)
    const int64_t nthreads = std::max<int64_t>((int64_t)n_sv * 8, (int64_t)n_regions + 1);
    hipLaunchKernelGGL(k2_scatter, dim3((unsigned)((nthreads + LCR_BLOCK - 1) / LCR_BLOCK)), dim3(LCR_BLOCK), 0, s, tmp, keep, pos, n_sv, out,
                       sv_region_off, n_regions, cand_off);
  } else launch_gather_i32(pos, sv_region_off, n_regions + 1, n_sv, pos + n_sv, cand_off, s);
  if (n_regions > 0) hipLaunchKernelGGL(k2_dense, dim3(n_regions), dim3(256), 0, s, out, cand_off, n_regions, idx, dense_win, min_dense_cnt, h_cand, h_off);
}
