// lcr_dev.h — shared host/device declarations of liblcr (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <climits>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/lcr.h"

#ifndef LCR_TILE
#define LCR_TILE 256       // pileup columns per workgroup tile (512: K1 0.385 instead of 0.334 ms on C3, K0 0.23 = 0.24)
#endif
#define LCR_BLOCK 256      // threads per workgroup (4 wave64)
#define LCR_WAVE 64

// Device view of a bound batch (all pointers in HBM).
struct BatchView {
  int32_t n_reads, n_regions;
  int64_t n_bases;           // bytes in bases / quals
  const int32_t* pos;
  const int32_t* seq_len;
  const int32_t* lead;
  const int32_t* trail;
  const uint8_t* flags;
  const uint64_t* seq_off;
  const uint64_t* cig_off;
  const uint32_t* n_cig;
  const uint8_t* bases;
  const uint8_t* quals;
  const uint32_t* cigar;
  const int64_t* start0;
  const int32_t* len;
  const int64_t* col_off;
  const int32_t* read_begin;
  const uint8_t* ref;
  // derived by K0 (lcr_load_batch)
  const int32_t* region_first_tile;  // n_regions+1: first pileup tile of each region
  const int32_t* read_region;        // n_reads: region index of each read
  int32_t* error_flag;               // != 0 -> unknown CIGAR op seen
  int32_t* read_rend;                // n_reads: region-relative column one past the read's last reference base (K0 pass 0)
};

// Per-read header packed once per batch (k0_pack) so that K0 needs a single 64-byte load per read.
struct ReadBin {
  int32_t rel_pos;     // pos - start0[region]
  int32_t vec;         // region length in columns
  int32_t ftile;       // first tile of the region
  int32_t n_cig;
  int64_t gbase;       // col_off[region] + region (slot base in the intron difference array)
  uint64_t seq_off, cig_off;
  int32_t lead, reb;   // leading soft clip, seq_len - trailing soft clip
  int32_t flags, pad_;
  int64_t pad2_;
};
static_assert(sizeof(ReadBin) == 64, "ReadBin must be 64 bytes");

struct DevParams {
  int32_t ont;
  int32_t dist_to_end, polya_len;
  uint32_t min_baseq, min_depth, max_depth, min_qual, low_cnt_cut, min_linkers;
  int32_t use_strand_bias;
  float min_af, min_af_intron, low_frac_cut;
  float sor_threshold;
  int32_t dbg;  // (unused; the K1 ablation switches are compile-time, see k1_pileup.hip)
};

// pass-1 survivor of the candidate filters (one per column that reaches the likelihood block)
// K0's per-tile records (k0_ops.hip writes them, k1_pileup.hip and k2_hist_tiles read them), 64 bit:
// [0,40) byte offset of the first read base | [40,50) tile column | [50,60) length-1 | [60] reverse strand |
// [61,63) transcript-strand class (0 none, 1 -> [0], 2 -> [1]); D / I / N records carry a marker in the offset field
#define REC_OFF_MASK 0xFFFFFFFFFFull
#define REC_KIND_D 0xFFFFFFFFFFull  // offset field of a D-run record
#define REC_KIND_I 0xFFFFFFFFFEull  // offset field of an I-point record
#define REC_KIND_N 0xFFFFFFFFFDull  // offset field of an N-run record (the part of an intron inside its first / last tile)

struct Survivor {
  int64_t gcol;      // global column index (col_off[region] + column)
  int32_t region;
  int32_t col;       // column inside region
  uint8_t ref_base, allele1, allele2, n_alt;
  uint32_t cnt1, cnt2, depth;
  uint32_t ts_fwd, ts_rev;
  float af1, af2;
};

#define HIPCHK(ctx, expr)                                                              \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                  \
      return LCR_E_DEVICE;                                                             \
    }                                                                                  \
  } while (0)

// Process-wide cache of the blocks contexts give back (lcr_api.hip).  A context that is destroyed hands its device and page-locked
// buffers to the cache, the next context on that device takes them from there: a worker that recreates its context per task does not
// churn hipMalloc / hipFree (GBs per context), and -- profiles/r06_stall.txt -- a context created right behind a large upload no longer
// loses 65-85 ms in its first steps.  Bounded (LCR_CACHE_DEV_BYTES / LCR_CACHE_HOST_BYTES per device, beyond that blocks are freed);
// lcr_release_cached_memory() returns everything to the runtime.
void* lcr_cache_take(int host, size_t want, size_t* cap);   // a cached block with want <= cap <= 2 * want + 1 MiB of the current device, or nullptr
bool lcr_cache_put(int host, void* p, size_t cap);          // false: the cache is full, the caller frees the block

// growable device buffer
// Measurement aid (lcr_debug_set("host_trace", 1)): wall-clock marks of the calling thread inside the stage calls, printed to stderr by
// lcr_host_trace_flush (end of lcr_phase) as microseconds since the first mark -- where does the host wait, when does it queue what.
extern int g_lcr_host_trace;
void lcr_host_trace_mark(const char* tag);
void lcr_host_trace_flush();
#define HT(tag) do { if (g_lcr_host_trace) lcr_host_trace_mark(tag); } while (0)

// Byte fill on a stream by a kernel of the library's own (the same contract as hipMemsetAsync for device memory).  The runtime's fill is a blit
// kernel too, but behind an event record it starts 35 - 57 us late on this platform (profiles/r06_timeline.txt); lcr_debug_set("own_fill", 0)
// goes back to hipMemsetAsync.
extern int g_lcr_own_fill;
hipError_t lcr_fill_async(void* p, int byte, size_t bytes, hipStream_t s);
hipError_t lcr_fill_multi_async(int n /* <= 4 */, void* const* ptrs, const int* bytes_val, const size_t* sizes, hipStream_t s);

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    // (hipFree waits for the device before it releases a block: so does handing one to the cache -- a kernel in flight may still read it)
    if (p) { (void)hipDeviceSynchronize(); if (!lcr_cache_put(0, p, cap)) (void)hipFree(p); }
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if ((p = lcr_cache_take(0, want, &cap)) != nullptr) return hipSuccess;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want; else p = nullptr;
    return e;
  }
  // (the owner has drained its queues: lcr_ctx_destroy)
  void release() { if (p && !lcr_cache_put(0, p, cap)) (void)hipFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};
struct HostBuf {  // pinned
  void* p = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) { (void)hipDeviceSynchronize(); if (!lcr_cache_put(1, p, cap)) (void)hipHostFree(p); }
    p = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    if ((p = lcr_cache_take(1, want, &cap)) != nullptr) return hipSuccess;
    hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
    if (e == hipSuccess) cap = want; else p = nullptr;
    return e;
  }
  void release() { if (p && !lcr_cache_put(1, p, cap)) (void)hipHostFree(p); p = nullptr; cap = 0; }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- phasing LUT (fixed point, scale 2^40; DESIGN.md "Decision arithmetic") ----
struct PhaseLutDev {
  int64_t fe[31], f1e[31];           // log10(eps), log10(1-eps), eps = 10^(-q/10) (q=0 -> q=1)
  int64_t f_homref, f_homvar, f_het0, f_log2;
};

// ---- kernel launchers (defined in the .hip files) ----
void launch_k0_region_setup(const int64_t* start0, const int32_t* len, const int64_t* col_off, const int32_t* read_begin, int32_t ng,
                            int32_t* first_tile, int64_t* h_start0, int32_t* h_len, int64_t* h_col_off, int32_t* h_read_begin,
                            hipStream_t s);
void launch_k0_tiles(const int32_t* first_tile, int32_t n_regions, int32_t* tile_region, int32_t* tile_col0, hipStream_t s);
void launch_k0_read_region(const BatchView& b, int32_t* read_region, hipStream_t s);
void launch_k0_tiles_read_region(const BatchView& b, const int32_t* first_tile, int32_t* tile_region, int32_t* tile_col0, int32_t* read_region, hipStream_t s);   // launch_k0_tiles + launch_k0_read_region in one kernel
void launch_k0_pack(const BatchView& b, ReadBin* out, int32_t* order_flag /* pinned host memory, device address */, hipStream_t s);
void launch_k0_bind_a(const int64_t* start0, const int32_t* len, const int64_t* col_off, const int32_t* read_begin, int32_t ng,
                      int32_t* first_tile, int64_t* h_start0, int32_t* h_len, int64_t* h_col_off, int32_t* h_read_begin,
                      const uint64_t* cig_off, const uint32_t* n_cig, int32_t nr, int64_t n_cigar, int32_t* out, hipStream_t s);
void launch_k0_bind_b(const BatchView& b, ReadBin* rbin, int32_t* order_flag, int32_t* read_region, int32_t n_tiles, int32_t* tile_region,
                      int32_t* tile_col0, uint64_t cig0, int32_t opb, int32_t n_blocks, int32_t* blk_first_read, hipStream_t s);
// K0 (k0_ops.hip): op-parallel CIGAR decode + per-tile record binning; load-time helpers
void launch_k0_cig_check(const uint64_t* cig_off, const uint32_t* n_cig, int32_t nr, int64_t n_cigar, int32_t* out, hipStream_t s);
void launch_k0_cig_compact(const uint32_t* cigar, const uint64_t* cig_off, const uint32_t* n_cig, const int32_t* new_off, int32_t nr,
                           uint32_t* out, uint64_t* out_off, hipStream_t s);
int launch_k0_opb();   // ops per K0 workgroup
void launch_k0_block_reads(const ReadBin* rbin, int32_t nr, uint64_t cig0, int32_t opb, int32_t n_blocks, int32_t* blk_first_read, hipStream_t s);
void launch_k0_ops(const BatchView& b, const ReadBin* rb, const int32_t* blk_first_read, uint64_t cig0, uint32_t n_ops, int ont, int D,
                   int32_t n_tiles, int32_t* tile_fill, int32_t* tile_nchunks, int32_t* tile_ndiff, void* ctl, unsigned int* acct /* launch_k0_acct_words() zeroed words */,
                   unsigned int pool_sub /* slots per shard */, unsigned long long* recs, unsigned int desc_sub, uint32_t* desc_tile, void* desc_val /* uint2 */,
                   void* read_scan /* n_reads x int2 scratch */, hipStream_t s);
int launch_k0_acct_words();
int launch_k0_acct_slots();
void launch_k0_desc_bin(const void* ctl, const unsigned int* acct, unsigned int desc_sub, const uint32_t* desc_tile, const void* desc_val,
                        const int32_t* chunk_off, int32_t* cursor /* n_tiles zeroed ints */, void* sorted /* uint2, all shards */, int32_t n_blocks_hint,
                        hipStream_t s);
size_t launch_k1_tiles_tmp_words(int32_t n_tiles);
void launch_k1_tiles_a(int32_t n_tiles, const int32_t* tile_fill, const int32_t* tile_ndiff, const int32_t* tile_nent, int32_t* tmp /* zeroed */,
                       const unsigned int* acct, int32_t n_acct, unsigned int* ctl, unsigned int* host_ctl /* pinned host block as the device sees it, or nullptr */, hipStream_t s);
void launch_k1_tiles_b(int32_t n_tiles, const int32_t* tile_fill, const int32_t* tile_ndiff, const int32_t* tile_nent, int32_t* tmp,
                       int32_t* tile_nbase, int32_t* ent_off, int32_t* order, hipStream_t s);
void launch_k1_pileup(const BatchView& b, const DevParams& p, const int32_t* tile_region, const int32_t* tile_col0,
                      int32_t n_tiles, int64_t n_cols, const int32_t* tile_fill, const int32_t* chunk_off, const void* chunks,
                      const unsigned long long* recs, const int32_t* tile_nbase, uint32_t* planes, const int32_t* order,
                      const int32_t* tiles_tmp /* scratch of launch_k1_tiles_a / _b */, int zeroed /* the planes were zeroed beforehand */, hipStream_t s,
                      hipStream_t bg = nullptr, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, int bg_wgs = 0 /* > 0: the record-free tiles on queue bg, this many workgroups */,
                      uint8_t* flt_flags = nullptr /* n_cols: != nullptr = pass 1 of the candidate filters in the tally's epilogue (k2_filter's flags) */, int32_t* flt_count = nullptr /* n_tiles: survivors per tile */);
void launch_k1_empty_early(const BatchView& b, const int32_t* tile_region, const int32_t* tile_col0, int32_t n_tiles, int64_t n_cols, const int32_t* tile_nbase,
                           uint32_t* planes, const int32_t* order, const int32_t* tiles_tmp, hipStream_t s, hipStream_t bg, hipEvent_t ev0, hipEvent_t ev1, int32_t* flt_count);
bool launch_k1_zonefix_tiles(const BatchView& b, int D, int L, int64_t n_cols, uint32_t* planes, const int32_t* tile_region, const int32_t* tile_col0,
                             int32_t n_tiles, const int32_t* tile_nbase, const int32_t* order, const int32_t* tiles_tmp, int32_t* flt_count, hipStream_t s);   // poly-A pass + record-free tiles in one launch; false: not applicable (launch them separately)
void launch_k1_zonefix(const BatchView& b, const ReadBin* rbin, int D, int L, int64_t n_cols, uint32_t* planes, hipStream_t s);
void launch_k2_filter(const BatchView& b, const DevParams& p, const int32_t* tile_region, const int32_t* tile_col0,
                      int32_t n_tiles, int64_t n_cols, const uint32_t* planes, const int32_t* tile_fill, uint8_t* flags,
                      int32_t* tile_count, hipStream_t s);   // tile_fill: K0's record counters of the last lcr_pileup
void launch_gather_i32(const int32_t* src, const int32_t* idx, int32_t n, int32_t n_src, const int32_t* total, int32_t* out, hipStream_t s,
                       int32_t* host_out = nullptr /* pinned host memory as the device sees it: the same values, no copy needed */);
void launch_scan_i32(DevBuf& tmp, const int32_t* in, int32_t* out_excl, int32_t n, int32_t* total, hipStream_t s);
void launch_k2_compact(const BatchView& b, const DevParams& p, const int32_t* tile_region, const int32_t* tile_col0,
                       int32_t n_tiles, int64_t n_cols, const uint32_t* planes, const uint8_t* flags,
                       const int32_t* tile_count, const int32_t* tile_off, Survivor* out, int32_t out_cap, hipStream_t s);
float lcr_device_sor_threshold(hipStream_t s);
#define LCR_HITS 16   // (read, survivor) hits k2_hist keeps per read for K3 (= K3's inline entries per row)
void launch_k2_hist(const BatchView& b, const DevParams& p, const ReadBin* rbin, const Survivor* sv, const int32_t* tile_off /* survivors in front of every tile (k2_compact's offsets) */, int32_t n_tiles, int32_t n_sv,
                    uint32_t* hist /* n_sv * 4 * 31 */, int32_t* hit_cnt /* n_reads, or nullptr: no hit lists */, void* hit_list /* n_reads x LCR_HITS x uint2 */,
                    int32_t* ovf_cnt /* zeroed */, int32_t* ovf_list /* n_reads */, hipStream_t s);
// the same histograms from K0's per-tile records instead of a walk over the reads (ONT presets, batches whose survivors are dense)
void launch_k2_hist_tiles(const BatchView& b, const int32_t* tile_col0, int32_t n_tiles, const int32_t* tile_count,
                          const int32_t* tile_off, const Survivor* sv, const int32_t* ent_off, const void* ents, const unsigned long long* recs,
                          uint32_t* hist, hipStream_t s);
void launch_k2_gt(const DevParams& p, const Survivor* sv, int32_t n_sv, const uint32_t* hist, const int64_t* start0,
                  lcr_candidate* out, int32_t* keep, hipStream_t s);
void launch_k2_finish(DevBuf& scan_tmp, const lcr_candidate* tmp, const int32_t* keep, int32_t n_sv, const int32_t* sv_region_off,
                      int32_t n_regions, int32_t* pos, int32_t* idx, lcr_candidate* out, int32_t* cand_off, uint32_t dense_win,
                      uint32_t min_dense_cnt, hipStream_t s, lcr_candidate* h_cand = nullptr, int32_t* h_off = nullptr);
void launch_k3_row_offsets(const int32_t* region_rows, int32_t ng, int32_t* row_region_off, hipStream_t s);
void launch_k3_region_entries(const int64_t* row_ptr, const int32_t* row_region_off, int32_t ng, int64_t* region_e_off, hipStream_t s, int64_t* host_out = nullptr);
struct K3Hits {   // what k2_hist left for K3 (hit_cnt == nullptr: nothing -- K3 walks every read's CIGAR itself)
  const int32_t* hit_cnt; const void* hit_list; const int32_t* ovf_cnt; const int32_t* ovf_list;
  const int32_t* keep; const int32_t* pos;   // survivor s is candidate pos[s] if keep[s]
};
void launch_k3_count(const BatchView& b, const ReadBin* rbin, const lcr_candidate* cand, const int32_t* cand_region_off,
                     const int32_t* row_region_off, int32_t n_rows, int32_t* row_cnt, uint32_t* row_links, int32_t* tmp_col,
                     uint8_t* tmp_val, const K3Hits& hits, hipStream_t s);   // tmp_*: launch_k3_inline() provisional entries per row
void launch_k3_fill(const BatchView& b, const ReadBin* rbin, const lcr_candidate* cand, const int32_t* cand_region_off,
                    const int32_t* row_region_off, int32_t n_rows, int32_t* row_cnt, const int64_t* row_ptr, const int32_t* tmp_col,
                    const uint8_t* tmp_val, int32_t* col, uint8_t* val, const K3Hits& hits, hipStream_t s);
int launch_k3_inline();
void launch_k3_rows(const BatchView& b, const lcr_candidate* cand, const int32_t* cand_region_off, int32_t* region_rows,
                    hipStream_t s, int32_t* host_out = nullptr /* pinned host memory as the device sees it */);
void launch_k3_rows_offsets(const BatchView& b, const lcr_candidate* cand, const int32_t* cand_region_off, int32_t* region_rows, int32_t* row_region_off,
                            hipStream_t s, int32_t* host_rows = nullptr);   // launch_k3_rows + launch_k3_row_offsets in one kernel
void launch_scan_i32_to_i64(DevBuf& tmp, const int32_t* in, int64_t* out_excl, int32_t n, hipStream_t s);

void launch_k5_span_window(const int32_t* ref_start, const int32_t* ref_end, int32_t n, int64_t contig_len, int32_t* out /* {INT_MAX, 0} */, hipStream_t s);
void launch_k5_span_diff(const int32_t* ref_start, const int32_t* ref_end, int32_t n, int64_t contig_len, int64_t lo /* first position of diff */, uint32_t* diff, hipStream_t s);
void launch_k5_bounds(bool write, const int32_t* ex, int64_t contig_len, int32_t n_blocks, int32_t* blk_cnt, const int32_t* blk_off,
                      int32_t* starts, int32_t* ends, hipStream_t s);
void launch_k5_island_max(const int32_t* ex, const int32_t* starts, const int32_t* ends, int32_t n_islands, uint32_t* maxcov, hipStream_t s);

// device helpers shared by kernels -------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int base_code(uint8_t b) {
  // A,C,G,T (either case, as util.rs:822-889 matches 'A'|'a' ...) -> 0..3, else -1
  switch (b) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}
__device__ __forceinline__ int iabs_(int x) { return x < 0 ? -x : x; }

// util.rs:745-751: |curr_pos - leading_softclips| < D or |curr_pos - (seq_len - trailing)| < D
__device__ __forceinline__ bool in_end_zone(int c, int lead, int reb, int D) {
  return iabs_(c - lead) < D || iabs_(c - reb) < D;
}

// util.rs:754-789 restated as a single scan: the base at read position c facing reference byte R is
// masked iff a window of L identical bases X in {A,C,G,T}, X != R, starts at some t in [c-L, c+1]
// and lies inside the read.  Window [t, t+L-1] is all-X iff the L-1 adjacent equalities hold.
__device__ __forceinline__ bool polya_masked(const uint8_t* __restrict__ seq, int seq_len, int c, int L, uint8_t R) {
  int lo = c - L;            // first byte that can belong to a window
  int hi = c + L;            // last byte that can belong to a window (window t=c+1 ends at c+L)
  if (lo < 0) lo = 0;
  if (hi > seq_len - 1) hi = seq_len - 1;
  if (hi - lo + 1 < L) return false;
  int run = 1;               // length of the run of equal bytes ending at i
  uint8_t prev = seq[lo];
  bool masked = false;
  for (int i = lo + 1; i <= hi; i++) {
    uint8_t cur = seq[i];
    run = (cur == prev) ? run + 1 : 1;
    prev = cur;
    if (run >= L) {
      // window [i-L+1, i] is homopolymer of `cur`; its start t = i-L+1 must be in [c-L, c+1]
      int t = i - L + 1;
      if (t >= c - L && t <= c + 1 && cur != R && (cur == 'A' || cur == 'C' || cur == 'G' || cur == 'T')) masked = true;
    }
  }
  return masked;
}


// wave64 inclusive add-scan with DPP row shifts / row broadcasts (6 VALU ops, no LDS round trips).
// update_dpp(old = 0, ..., bound_ctrl = false): lanes without a source keep 0, the identity.
__device__ __forceinline__ int wave_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1,3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2,3
  return v;
}

// inclusive add-scan inside each 16-lane DPP row (four reads share a wave in the site walkers)
__device__ __forceinline__ int row16_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  return v;
}

// Sixteen lanes (one DPP row) locate, for one read, the bases that face a sorted list of sites
// (region-relative columns site_col(i), i in [s_lo, s_hi), ascending).  This is the cursor walk of
// util.rs:700-944 / fragment.rs:60-240 turned inside out so that no load depends on a previous site:
//   1. the sites inside the read's reference span [max(rel_pos,0), rend) are found with one probe of
//      sixteen site columns per step (lanes <-> sites);
//   2. the CIGAR is read 64 ops at a time (four words per lane, one round trip), reference / read start
//      offsets of the ops come from two row scans per 16 ops (lanes <-> ops), and every covered site
//      records the read offset of its base in "its" lane (site j of the batch -> lane j);
//   3. sink(i, c, hit) is then called ONCE per batch of <= 16 sites with all sixteen lanes active:
//      lane j gets site i = first + j and c = read offset of the base an M/=/X op puts on that site
//      (hit = false: no such base).  The sink does its loads / atomics for up to 16 sites in parallel
//      and may use row ballots to keep the sites' order.
// A read that covers no site returns before touching its CIGAR.  Four reads share a wave64.
// `live` = this row has a read; rows without one pass live = false (their lanes still call).
// step 1 of the walk: first site with column >= key = max(rel_pos, 0) (cur) and first site with column >= rend (s_end) among the
// ascending sites [s_lo, s_hi); live = false or no site inside the span: cur >= s_end
template <class ColFn>
__device__ __forceinline__ void row16_find_sites(bool live, const ReadBin& h, int rend, int s_lo, int s_hi, ColFn site_col, int* cur_out, int* s_end_out) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, rbase = lane & 48;
  int cur = s_hi, s_end = s_hi;
  if (live) {
    const int key = h.rel_pos > 0 ? h.rel_pos : 0;
    bool found_cur = false;
    for (int base = s_lo; base < s_hi; base += 16) {
      const int i = base + l16;
      const int colv = i < s_hi ? site_col(i) : INT_MAX;
      const unsigned int ge_key = (unsigned int)(__ballot(colv >= key) >> rbase) & 0xffffu;
      const unsigned int ge_end = (unsigned int)(__ballot(colv >= rend) >> rbase) & 0xffffu;
      if (!found_cur && ge_key) { cur = base + __ffs((int)ge_key) - 1; found_cur = true; }
      if (ge_end) { s_end = base + __ffs((int)ge_end) - 1; break; }
    }
    if (!found_cur) cur = s_end;
  }
  *cur_out = cur; *s_end_out = s_end;
}

// steps 2 + 3: the sites [cur, s_end) of the read against its CIGAR.  pre != nullptr: the read's first 64 ops (four words per lane, op
// 16 k + l16 in pre[k], 0 beyond the CIGAR) were requested by the caller beside its own loads (k2_hist: one round trip less in the chain)
template <class ColFn, class Sink>
__device__ __forceinline__ void row16_walk_range(const BatchView& b, const ReadBin& h, int cur, int s_end, const uint32_t* pre, ColFn site_col, Sink sink) {
  const int lane = threadIdx.x & 63, l16 = lane & 15, rbase = lane & 48;
  if (cur >= s_end) return;   // (row-uniform)
  const uint32_t ncig = (uint32_t)h.n_cig;
  const uint32_t* __restrict__ cg = b.cigar + h.cig_off;
  for (int first = cur; first < s_end; first += 16) {   // batches of 16 covered sites (one batch almost always)
    const int n = min(16, s_end - first);
    const int my_cc = l16 < n ? site_col(first + l16) : INT_MAX;
    const int last_cc = __shfl(my_cc, rbase + n - 1, 64);
    int myc = -1;
    int ref_cur = h.rel_pos;
    int q_cur = h.lead > 0 ? h.lead : 0;
    int j = 0;                                           // next site of the batch (row-uniform)
    int cc = __shfl(my_cc, rbase, 64);
    for (uint32_t c0 = 0; c0 < ncig && j < n; c0 += 64) {
      uint32_t w4[4];
      if (pre && c0 == 0) {
#pragma unroll
        for (int k = 0; k < 4; k++) w4[k] = pre[k];
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) w4[k] = c0 + 16 * k + l16 < ncig ? cg[c0 + 16 * k + l16] : 0u;
      }
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t w = w4[k];
        const int op = w & 15, len = (int)(w >> 4);
        const bool is_m = op == 0 || op == 7 || op == 8;     // (a padding word is a 0-length M: covers nothing)
        const int dr = (is_m || op == 2 || op == 3) ? len : 0;
        const int dq = (is_m || op == 1) ? len : 0;
        const int ir = row16_incl_scan(dr), iq = row16_incl_scan(dq);
        const int rs = ref_cur + ir - dr, qs = q_cur + iq - dq;
        const int chunk_end = ref_cur + __shfl(ir, rbase + 15, 64);
        while (j < n && cc < chunk_end) {
          const unsigned int m = (unsigned int)(__ballot(is_m && cc >= rs && cc < rs + len) >> rbase) & 0xffffu;
          if (m) {
            const int L = rbase + __ffs((int)m) - 1;
            const int c = __shfl(qs, L, 64) + (cc - __shfl(rs, L, 64));
            if (l16 == j) myc = c;
          }
          j++;
          cc = __shfl(my_cc, rbase + (j < n ? j : 0), 64);
        }
        ref_cur = chunk_end;
        q_cur += __shfl(iq, rbase + 15, 64);
      }
      if (ref_cur > last_cc) break;
    }
    sink(first + l16, myc, myc >= 0);
  }
}

template <class ColFn, class Sink>
__device__ __forceinline__ void row16_walk_sites(const BatchView& b, bool live, const ReadBin& h, int rend, int s_lo, int s_hi,
                                                 ColFn site_col, Sink sink) {
  int cur, s_end;
  row16_find_sites(live, h, rend, s_lo, s_hi, site_col, &cur, &s_end);
  row16_walk_range(b, h, cur, s_end, nullptr, site_col, sink);
}

// region index of read r: precomputed by k0_read_region (a binary search over read_begin would cost
// ~log2(n_regions) dependent global loads per read in every read-parallel kernel)
#define region_of_read(b_, r_) ((b_).read_region[(r_)])
#endif
