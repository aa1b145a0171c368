// k3_fragments.hip — K3: read x SNP fragment matrix in CSR (gfx950).
//
// Replaces SNPFrag::get_fragments (reference src/fragment.rs:10-309): per read one CIGAR walk against
// the region's candidates (sorted by position), sixteen lanes per read.  Two passes (count -> scan -> fill)
// give a canonical CSR: row = k-th read of the region that starts at or before the last candidate
// (fragment.rs:51-54, empty rows included), columns ascending.  Entry = (candidate index, u8 value
// q5 | p-bit | base code).  Entries with p = 0 or at dense candidates are dropped (fragment.rs:148).
#include "lcr_dev.h"

// rows per region: reads with pos <= last candidate pos (reads are sorted by pos)
__global__ void k3_rows(BatchView b, const lcr_candidate* __restrict__ cand, const int32_t* __restrict__ cand_region_off,
                        int32_t* __restrict__ region_rows, int32_t* __restrict__ host_out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= b.n_regions) return;
  const int c0 = cand_region_off[g], c1 = cand_region_off[g + 1];
  if (c1 <= c0) { region_rows[g] = 0; if (host_out) host_out[g] = 0; return; }  // fragment.rs:24-26
  const int64_t last = cand[c1 - 1].pos;
  int lo = b.read_begin[g], hi = b.read_begin[g + 1];
  const int rb = lo;
  while (lo < hi) { int mid = (lo + hi) >> 1; if ((int64_t)b.pos[mid] > last) hi = mid; else lo = mid + 1; }
  region_rows[g] = lo - rb;
  if (host_out) host_out[g] = lo - rb;
}
void launch_k3_rows(const BatchView& b, const lcr_candidate* cand, const int32_t* cand_region_off, int32_t* region_rows,
                    hipStream_t s, int32_t* host_out) {
  if (b.n_regions == 0) return;
  hipLaunchKernelGGL(k3_rows, dim3((b.n_regions + 255) / 256), dim3(256), 0, s, b, cand, cand_region_off, region_rows, host_out);
}

// first row of every region: exclusive prefix sum of region_rows (one workgroup; a batch has at most a few
// thousand regions), so that lcr_fragments starts without an upload
__global__ void __launch_bounds__(1024) k3_row_offsets(const int32_t* __restrict__ region_rows, int32_t ng, int32_t* __restrict__ row_region_off) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) { base_s = 0; row_region_off[0] = 0; }
  __syncthreads();
  for (int g0 = 0; g0 < ng; g0 += 1024) {
    const int g = g0 + tid;
    const int incl = wave_incl_scan(g < ng ? region_rows[g] : 0);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = base_s;
    for (int w = 0; w < wave; w++) before += wsum[w];
    if (g < ng) row_region_off[g + 1] = before + incl;
    __syncthreads();
    if (tid == 1023) base_s = before + incl;
    __syncthreads();
  }
}
void launch_k3_row_offsets(const int32_t* region_rows, int32_t ng, int32_t* row_region_off, hipStream_t s) {
  hipLaunchKernelGGL(k3_row_offsets, dim3(1), dim3(1024), 0, s, region_rows, ng, row_region_off);
}
// k3_rows + k3_row_offsets in one launch (one workgroup: a thread per region finds its rows, then the prefix sums)
__global__ void __launch_bounds__(1024) k3_rows_offsets(BatchView b, const lcr_candidate* __restrict__ cand, const int32_t* __restrict__ cand_region_off,
                                                        int32_t* __restrict__ region_rows, int32_t* __restrict__ row_region_off, int32_t* __restrict__ host_rows) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ng = b.n_regions;
  if (tid == 0) { base_s = 0; row_region_off[0] = 0; }
  __syncthreads();
  for (int g0 = 0; g0 < ng; g0 += 1024) {
    const int g = g0 + tid;
    int rows = 0;
    if (g < ng) {
      const int c0 = cand_region_off[g], c1 = cand_region_off[g + 1];
      if (c1 > c0) {   // fragment.rs:24-26, 51-54: reads with pos <= the last candidate's (reads are sorted by pos)
        const int64_t last = cand[c1 - 1].pos;
        int lo = b.read_begin[g], hi = b.read_begin[g + 1];
        const int rb = lo;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if ((int64_t)b.pos[mid] > last) hi = mid; else lo = mid + 1; }
        rows = lo - rb;
      }
      region_rows[g] = rows;
      if (host_rows) host_rows[g] = rows;
    }
    const int incl = wave_incl_scan(rows);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int before = base_s;
    for (int w = 0; w < wave; w++) before += wsum[w];
    if (g < ng) row_region_off[g + 1] = before + incl;
    __syncthreads();
    if (tid == 1023) base_s = before + incl;
    __syncthreads();
  }
}
void launch_k3_rows_offsets(const BatchView& b, const lcr_candidate* cand, const int32_t* cand_region_off, int32_t* region_rows, int32_t* row_region_off,
                            hipStream_t s, int32_t* host_rows) {
  hipLaunchKernelGGL(k3_rows_offsets, dim3(1), dim3(1024), 0, s, b, cand, cand_region_off, region_rows, row_region_off, host_rows);
}

// first entry of every region: row_ptr at the regions' first rows ([ng] = all entries)
__global__ void k3_region_entries(const int64_t* __restrict__ row_ptr, const int32_t* __restrict__ row_region_off, int32_t ng,
                                  int64_t* __restrict__ region_e_off, int64_t* __restrict__ host_out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g <= ng) { const int64_t v = row_ptr[row_region_off[g]]; region_e_off[g] = v; if (host_out) host_out[g] = v; }
}
void launch_k3_region_entries(const int64_t* row_ptr, const int32_t* row_region_off, int32_t ng, int64_t* region_e_off, hipStream_t s, int64_t* host_out) {
  hipLaunchKernelGGL(k3_region_entries, dim3((ng + 256) / 256), dim3(256), 0, s, row_ptr, row_region_off, ng, region_e_off, host_out);
}

// Sixteen lanes per row (row16_walk_sites, lcr_dev.h): the region's candidates inside the read's reference
// span are located against the CIGAR spread over the lanes; a row without such candidates never loads
// its CIGAR.  No trimming here: the reference's fragment walk takes every aligned base.
// The count pass keeps the first K3_INLINE entries of every row in a provisional per-row slot (most rows have
// fewer: a read crosses a handful of candidates), so the second pass is a copy for those rows (k3_place) and
// only longer rows are walked again (FILL: rows with more than K3_INLINE entries).
#define K3_INLINE 16
static_assert(K3_INLINE == LCR_HITS, "a row's provisional slot holds what k2_hist's hit list holds");
// one read (sixteen lanes) of the walk; `live` rows only do work
template <bool FILL>
__device__ __forceinline__ void k3_walk_read(const BatchView& b, const ReadBin* __restrict__ rbin, const lcr_candidate* __restrict__ cand,
        const int32_t* __restrict__ cand_region_off, const int32_t* __restrict__ row_region_off, int r_, bool live_in,
        int32_t* __restrict__ row_cnt, uint32_t* __restrict__ row_links, const int64_t* __restrict__ row_ptr,
        int32_t* __restrict__ col, uint8_t* __restrict__ val) {
  // rows are the first region_rows[g] reads of each region: index by read (one load for the region)
  const int r = (live_in && r_ < b.n_reads) ? r_ : 0;
  const int g = region_of_read(b, r);
  const int k = r - b.read_begin[g];
  const int row0 = row_region_off[g];
  bool live = live_in && r_ < b.n_reads && k < row_region_off[g + 1] - row0;
  const int row = live ? row0 + k : 0;
  if (FILL) {   // shorter rows were placed from their provisional slots: most waves have nothing left to do
    live = live && row_cnt[row] > K3_INLINE;
    if (!__any(live)) return;
  }
  const int l16 = threadIdx.x & 15;
  const int c_lo = cand_region_off[g], c_hi = cand_region_off[g + 1];
  const ReadBin h = rbin[r];
  const int64_t start0 = b.start0[g];
  const uint8_t* __restrict__ seq = b.bases + h.seq_off;
  const uint8_t* __restrict__ qual = b.quals + h.seq_off;
  int cnt = 0;
  uint32_t links = 0;
  // FILL: final position of the row's entries; count pass: the row's provisional slot
  int64_t w = FILL ? row_ptr[row] : (int64_t)row * K3_INLINE;
  const int rbase = threadIdx.x & 48;
  row16_walk_sites(b, live, h, b.read_rend[r], c_lo, c_hi,
    [&](int i) { return (int)(cand[i].pos - start0); },
    [&](int idx, int qq, bool hit) {   // lane <-> candidate; entries keep the candidates' order
      int p = 0; uint8_t base = 0; bool fphase = false;
      if (hit) {
        const lcr_candidate& c = cand[idx];
        base = seq[qq];
        if (base == c.ref_base) p = 1;                                                // fragment.rs:134-135
        else if (base == c.allele1 || base == c.allele2) p = -1;                      // fragment.rs:136-140
        if (c.flags & LCR_F_DENSE) p = 0;                                             // fragment.rs:148-152
        fphase = (c.flags & LCR_F_FOR_PHASING) != 0;                                  // fragment.rs:144-146,242-250
      }
      const unsigned int em = (unsigned int)(__ballot(p != 0) >> rbase) & 0xffffu;
      const unsigned int ph = (unsigned int)(__ballot(p != 0 && fphase) >> rbase) & 0xffffu;
      if (p != 0) {
        const int at = cnt + __popc(em & ((1u << l16) - 1u));
        if (FILL || at < K3_INLINE) {
          const uint8_t bq = qual[qq] < 30 ? qual[qq] : 30;                            // fragment.rs:127-131
          col[w + at] = idx;
          val[w + at] = (uint8_t)(bq | (p == 1 ? 32 : 0) | (base_code(base) << 6));
        }
      }
      cnt += __popc(em); links += __popc(ph);
    });
  if (!FILL && live && l16 == 0) { row_cnt[row] = cnt; row_links[row] = links; }
}
// every read of the batch (batches without hit lists: the dense-survivor path of lcr_candidates)
template <bool FILL>
__global__ void __launch_bounds__(LCR_BLOCK)
k3_walk(BatchView b, const ReadBin* __restrict__ rbin, const lcr_candidate* __restrict__ cand,
        const int32_t* __restrict__ cand_region_off, const int32_t* __restrict__ row_region_off, int32_t n_rows,
        int32_t* __restrict__ row_cnt, uint32_t* __restrict__ row_links, const int64_t* __restrict__ row_ptr,
        int32_t* __restrict__ col, uint8_t* __restrict__ val) {
  k3_walk_read<FILL>(b, rbin, cand, cand_region_off, row_region_off, (int)((blockIdx.x * LCR_BLOCK + threadIdx.x) >> 4), true, row_cnt, row_links, row_ptr, col, val);
}
// the reads of a list (those with more hits than k2_hist's list holds): a fixed grid walks it
template <bool FILL>
__global__ void __launch_bounds__(LCR_BLOCK)
k3_walk_list(BatchView b, const ReadBin* __restrict__ rbin, const lcr_candidate* __restrict__ cand,
             const int32_t* __restrict__ cand_region_off, const int32_t* __restrict__ row_region_off, const int32_t* __restrict__ n_list,
             const int32_t* __restrict__ list, int32_t* __restrict__ row_cnt, uint32_t* __restrict__ row_links, const int64_t* __restrict__ row_ptr,
             int32_t* __restrict__ col, uint8_t* __restrict__ val) {
  const int n = *n_list, per = LCR_BLOCK / 16;
  for (int i0 = (int)blockIdx.x * per; i0 < n; i0 += (int)gridDim.x * per) {   // (uniform over the workgroup; the waves' rows differ)
    const int i = i0 + (int)(threadIdx.x >> 4);
    const int w0 = i0 + (int)((threadIdx.x >> 6) << 2);   // first list index of this wave
    if (w0 >= n) continue;                                 // (wave-uniform)
    k3_walk_read<FILL>(b, rbin, cand, cand_region_off, row_region_off, i < n ? list[i] : 0, i < n, row_cnt, row_links, row_ptr, col, val);
  }
}

// Count pass from k2_hist's hit lists: a row's entries are its hits at survivors that became candidates, in the same order --
// no CIGAR is read.  Sixteen lanes per read, lane <-> hit; reads with more hits than the list holds are left to k3_walk_list.
__global__ void __launch_bounds__(LCR_BLOCK)
k3_hits(BatchView b, const lcr_candidate* __restrict__ cand, const int32_t* __restrict__ row_region_off, const int32_t* __restrict__ hit_cnt,
        const uint2* __restrict__ hit_list, const int32_t* __restrict__ keep, const int32_t* __restrict__ pos,
        int32_t* __restrict__ row_cnt, uint32_t* __restrict__ row_links, int32_t* __restrict__ col, uint8_t* __restrict__ val) {
  const int r_ = (blockIdx.x * LCR_BLOCK + threadIdx.x) >> 4;
  const int r = r_ < b.n_reads ? r_ : 0;
  const int l16 = threadIdx.x & 15, rbase = threadIdx.x & 48;
  // two short dependent chains side by side (the kernel is a handful of loads per row: their latency is all it costs):
  // read -> region -> row, and read -> hit -> survivor's candidate -> its alleles
  const int nh = hit_cnt[r];
  const uint2 hv = hit_list[(size_t)r * LCR_HITS + l16];   // (slots beyond nh hold stale values: used only below nh)
  const int g = region_of_read(b, r);
  const bool has = r_ < b.n_reads && nh <= LCR_HITS && l16 < nh;
  const int kp = has ? keep[hv.x] : 0;
  int idx = has ? pos[hv.x] : 0;
  const int k = r - b.read_begin[g];
  const int row0 = row_region_off[g];
  const bool live = r_ < b.n_reads && k < row_region_off[g + 1] - row0 && nh <= LCR_HITS;
  const int row = live ? row0 + k : 0;
  int p = 0; uint8_t base = 0, rq = 0; bool fphase = false;
  if (has && kp) {
    const lcr_candidate& c = cand[idx];
    base = (uint8_t)(hv.y & 0xffu); rq = (uint8_t)((hv.y >> 8) & 0xffu);
    if (base == c.ref_base) p = 1;                                                // fragment.rs:134-135
    else if (base == c.allele1 || base == c.allele2) p = -1;                      // fragment.rs:136-140
    if (c.flags & LCR_F_DENSE) p = 0;                                             // fragment.rs:148-152
    fphase = (c.flags & LCR_F_FOR_PHASING) != 0;                                  // fragment.rs:144-146,242-250
  }
  if (!live) p = 0;   // (a read behind the region's last candidate has no row: fragment.rs:51-54)
  const unsigned int em = (unsigned int)(__ballot(p != 0) >> rbase) & 0xffffu;
  const unsigned int ph = (unsigned int)(__ballot(p != 0 && fphase) >> rbase) & 0xffffu;
  if (p != 0) {
    const int64_t at = (int64_t)row * K3_INLINE + __popc(em & ((1u << l16) - 1u));
    const uint8_t bq = rq < 30 ? rq : 30;                                            // fragment.rs:127-131
    col[at] = idx;
    val[at] = (uint8_t)(bq | (p == 1 ? 32 : 0) | (base_code(base) << 6));
  }
  if (live && l16 == 0) { row_cnt[row] = __popc(em); row_links[row] = __popc(ph); }
}

// rows with at most K3_INLINE entries: provisional slot -> final CSR position (one thread per row)
__global__ void __launch_bounds__(LCR_BLOCK)
k3_place(int32_t n_rows, const int32_t* __restrict__ row_cnt, const int64_t* __restrict__ row_ptr,
         const int32_t* __restrict__ tmp_col, const uint8_t* __restrict__ tmp_val, int32_t* __restrict__ col, uint8_t* __restrict__ val) {
  const int row = blockIdx.x * LCR_BLOCK + threadIdx.x;
  if (row >= n_rows) return;
  const int n = row_cnt[row];
  if (n > K3_INLINE) return;
  const int64_t at = row_ptr[row], from = (int64_t)row * K3_INLINE;
  for (int e = 0; e < n; e++) { col[at + e] = tmp_col[from + e]; val[at + e] = tmp_val[from + e]; }
}

void launch_k3_count(const BatchView& b, const ReadBin* rbin, const lcr_candidate* cand, const int32_t* cand_region_off,
                     const int32_t* row_region_off, int32_t n_rows, int32_t* row_cnt, uint32_t* row_links, int32_t* tmp_col,
                     uint8_t* tmp_val, const K3Hits& hits, hipStream_t s) {
  if (n_rows == 0) return;
  const int per = LCR_BLOCK / 16;
  if (hits.hit_cnt) {
    hipLaunchKernelGGL(k3_hits, dim3((b.n_reads + per - 1) / per), dim3(LCR_BLOCK), 0, s, b, cand, row_region_off, hits.hit_cnt, (const uint2*)hits.hit_list,
                       hits.keep, hits.pos, row_cnt, row_links, tmp_col, tmp_val);
    hipLaunchKernelGGL(k3_walk_list<false>, dim3(2048), dim3(LCR_BLOCK), 0, s, b, rbin, cand, cand_region_off, row_region_off, hits.ovf_cnt, hits.ovf_list,
                       row_cnt, row_links, (const int64_t*)nullptr, tmp_col, tmp_val);
    return;
  }
  hipLaunchKernelGGL(k3_walk<false>, dim3((b.n_reads + per - 1) / per), dim3(LCR_BLOCK), 0, s, b, rbin, cand,
                     cand_region_off, row_region_off, n_rows, row_cnt, row_links, (const int64_t*)nullptr, tmp_col, tmp_val);
}
void launch_k3_fill(const BatchView& b, const ReadBin* rbin, const lcr_candidate* cand, const int32_t* cand_region_off,
                    const int32_t* row_region_off, int32_t n_rows, int32_t* row_cnt, const int64_t* row_ptr, const int32_t* tmp_col,
                    const uint8_t* tmp_val, int32_t* col, uint8_t* val, const K3Hits& hits, hipStream_t s) {
  if (n_rows == 0) return;
  const int per = LCR_BLOCK / 16;
  hipLaunchKernelGGL(k3_place, dim3((n_rows + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, n_rows, row_cnt, row_ptr, tmp_col, tmp_val, col, val);
  if (hits.hit_cnt)   // (a row of more than K3_INLINE entries has more than LCR_HITS hits: it is on the list)
    hipLaunchKernelGGL(k3_walk_list<true>, dim3(1024), dim3(LCR_BLOCK), 0, s, b, rbin, cand, cand_region_off, row_region_off, hits.ovf_cnt, hits.ovf_list,
                       row_cnt, (uint32_t*)nullptr, row_ptr, col, val);
  else
    hipLaunchKernelGGL(k3_walk<true>, dim3((b.n_reads + per - 1) / per), dim3(LCR_BLOCK), 0, s, b, rbin, cand,
                       cand_region_off, row_region_off, n_rows, row_cnt, (uint32_t*)nullptr, row_ptr, col, val);
}
int launch_k3_inline() { return K3_INLINE; }
