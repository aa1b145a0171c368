// k3_fragments.hip — K3: read x SNP fragment matrix in CSR (gfx950).
//
// Replaces SNPFrag::get_fragments (reference src/fragment.rs:10-309): per read one CIGAR walk with
// a cursor over the region's candidates (sorted by position).  Two passes (count -> scan -> fill)
// give a canonical CSR: row = k-th read of the region that starts at or before the last candidate
// (fragment.rs:51-54, empty rows included), columns ascending.  Entry = (candidate index, u8 value
// q5 | p-bit | base code).  Entries with p = 0 or at dense candidates are dropped (fragment.rs:148).
#include "lcr_dev.h"

// rows per region: reads with pos <= last candidate pos (reads are sorted by pos)
__global__ void k3_rows(BatchView b, const lcr_candidate* __restrict__ cand, const int32_t* __restrict__ cand_region_off,
                        int32_t* __restrict__ region_rows) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= b.n_regions) return;
  const int c0 = cand_region_off[g], c1 = cand_region_off[g + 1];
  if (c1 <= c0) { region_rows[g] = 0; return; }  // fragment.rs:24-26
  const int64_t last = cand[c1 - 1].pos;
  int lo = b.read_begin[g], hi = b.read_begin[g + 1];
  const int rb = lo;
  while (lo < hi) { int mid = (lo + hi) >> 1; if ((int64_t)b.pos[mid] > last) hi = mid; else lo = mid + 1; }
  region_rows[g] = lo - rb;
}
void launch_k3_rows(const BatchView& b, const lcr_candidate* cand, const int32_t* cand_region_off, int32_t* region_rows,
                    hipStream_t s) {
  if (b.n_regions == 0) return;
  hipLaunchKernelGGL(k3_rows, dim3((b.n_regions + 255) / 256), dim3(256), 0, s, b, cand, cand_region_off, region_rows);
}

template <bool FILL>
__global__ void __launch_bounds__(LCR_BLOCK)
k3_walk(BatchView b, const lcr_candidate* __restrict__ cand, const int32_t* __restrict__ cand_region_off,
        const int32_t* __restrict__ row_region_off, int32_t n_rows, int32_t* __restrict__ row_cnt,
        uint32_t* __restrict__ row_links, const int64_t* __restrict__ row_ptr, int32_t* __restrict__ col,
        uint8_t* __restrict__ val) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  int g;
  {  // last region with row_region_off[g] <= row
    int lo = 0, hi = b.n_regions;
    while (hi - lo > 1) { int mid = (lo + hi) >> 1; if (row_region_off[mid] <= row) lo = mid; else hi = mid; }
    g = lo;
  }
  const int r = b.read_begin[g] + (row - row_region_off[g]);
  const int c_lo = cand_region_off[g], c_hi = cand_region_off[g + 1];
  const int64_t pos = b.pos[r];
  int idx;
  {  // fragment.rs:63-80: first candidate with pos >= read pos
    int lo = c_lo, hi = c_hi;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cand[mid].pos >= pos) hi = mid; else lo = mid + 1; }
    idx = lo;
  }
  const uint32_t ncig = b.n_cig[r];
  const uint32_t* __restrict__ cg = b.cigar + b.cig_off[r];
  const uint8_t* __restrict__ seq = b.bases + b.seq_off[r];
  const uint8_t* __restrict__ qual = b.quals + b.seq_off[r];
  int64_t pr = pos;
  int64_t q = b.lead[r];
  int cnt = 0;
  uint32_t links = 0;
  int64_t w = FILL ? row_ptr[row] : 0;
  for (uint32_t i = 0; i < ncig && idx < c_hi; i++) {
    const int op = cg[i] & 15;
    const int64_t len = (int64_t)(cg[i] >> 4);
    if (op == 0 || op == 7 || op == 8) {
      while (idx < c_hi && cand[idx].pos < pr + len) {
        const lcr_candidate& c = cand[idx];
        const int64_t qq = q + (c.pos - pr);
        const uint8_t base = seq[qq];
        int p = 0;
        if (base == c.ref_base) p = 1;                                                // fragment.rs:134-135
        else if (base == c.allele1 || base == c.allele2) p = -1;                      // fragment.rs:136-140
        if (p != 0 && !(c.flags & LCR_F_DENSE)) {                                     // fragment.rs:148-152
          if (FILL) {
            const uint8_t bq = qual[qq] < 30 ? qual[qq] : 30;                          // fragment.rs:127-131
            col[w] = idx;
            val[w] = (uint8_t)(bq | (p == 1 ? 32 : 0) | (base_code(base) << 6));
            w++;
          }
          cnt++;
          if (c.flags & LCR_F_FOR_PHASING) links++;                                   // fragment.rs:144-146,242-250
        }
        idx++;
      }
      pr += len; q += len;
    } else if (op == 1) {
      q += len;
    } else if (op == 2 || op == 3) {
      while (idx < c_hi && cand[idx].pos < pr + len) idx++;                           // fragment.rs:166-189
      pr += len;
    }
  }
  if (!FILL) { row_cnt[row] = cnt; row_links[row] = links; }
}

void launch_k3_count(const BatchView& b, const lcr_candidate* cand, const int32_t* cand_region_off,
                     const int32_t* row_region_off, int32_t n_rows, int32_t* row_cnt, uint32_t* row_links,
                     hipStream_t s) {
  if (n_rows == 0) return;
  hipLaunchKernelGGL(k3_walk<false>, dim3((n_rows + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, b, cand,
                     cand_region_off, row_region_off, n_rows, row_cnt, row_links, (const int64_t*)nullptr,
                     (int32_t*)nullptr, (uint8_t*)nullptr);
}
void launch_k3_fill(const BatchView& b, const lcr_candidate* cand, const int32_t* cand_region_off,
                    const int32_t* row_region_off, int32_t n_rows, const int64_t* row_ptr, int32_t* col, uint8_t* val,
                    hipStream_t s) {
  if (n_rows == 0) return;
  hipLaunchKernelGGL(k3_walk<true>, dim3((n_rows + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, b, cand,
                     cand_region_off, row_region_off, n_rows, (int32_t*)nullptr, (uint32_t*)nullptr, row_ptr, col, val);
}
