// k0_ops.hip — K0 of the pileup stage for gfx950: op-parallel CIGAR decode + per-tile record binning.
//
// Replaces the CIGAR walk of Profile::fill_data_into_freq_vec (reference src/util.rs:692-947): what the reference does
// per read, base by base, is cut here into 8-byte records per pileup tile (LCR_TILE columns of one region) that K1
// (k1_pileup.hip) tallies without ever seeing a CIGAR:
//   M-segment  (tile column, length, byte offset of its first read base, strand, transcript-strand class)
//   D-run      (tile column, length)                    util.rs:905-917
//   I-point    (tile column)                            util.rs:918-929
//   N-run      (tile column, length)                    util.rs:930-942; only in the tiles where an intron starts or
//              ends -- the tiles it covers entirely get +1 in a tile-level difference array (tile_ndiff) instead.
// ONT end trimming (util.rs:745-751) is applied here by clipping M blocks to the untrimmed read interval.
//
// Design (DESIGN.md §4 "K0").  The unit of work is the CIGAR *op*, not the read: the flat CIGAR array of the batch is
// cut into blocks of K0_OPB consecutive ops (a workgroup each), whatever reads they belong to.
//   1. heads   the reads that begin inside the block are found from a per-block "first read" table (k0_pack) -- their
//              headers go to LDS, their first op is marked;
//   2. scan    every thread takes K0_OPT ops (lane <-> consecutive ops): reference / query advance per op, wave scans
//              (DPP) + chunk totals give the exclusive prefix of both over the block, a max-scan of the head marks
//              gives every op its read; a read's start (pos, leading clip) minus the prefix at its first op turns the
//              block-wide prefix into that read's reference column / read offset (the first read of a block may have
//              begun in an earlier block: wave 0 sums the ops it has there);
//   3. count   records per tile of the block in LDS (a window of K0_WIN tiles from the block's first read on; ops are
//              position-ordered inside a read and reads inside a region, so one add per RUN of equal tiles in a wave);
//   4. reserve ONE pool allocation per BLOCK (a read-parallel K0 needed one per (read, tile) and a level table to find
//              the slots again): the block's records lie back to back in the pool, grouped by tile (prefix sum of the LDS
//              counters); every (block, tile) group is announced by a 12-byte chunk descriptor (tile; pool offset, count);
//   5. emit    slots drawn from the LDS cursors, records written.
// k0_desc_bin then sorts the descriptors by tile (a counting sort on ~1 % as many items as there are records) and K1
// walks its tile's chunks.  The dependent global chain is per block (first-read table -> headers -> one allocation), not
// per read, and every lane decodes an op whatever the reads' lengths.  All arithmetic is integer: the records, hence K1's counts,
// do not depend on the order in which blocks run.
#include <algorithm>
#include <climits>

#include "lcr_dev.h"

#ifndef K0_THREADS
#define K0_THREADS 256
#endif
#ifndef K0_OPT
#define K0_OPT 4      // ops per thread: a block takes K0_OPT * K0_THREADS consecutive ops of the flat CIGAR array
#endif
#ifndef K0_ABL
#define K0_ABL 0      // measurement builds only (tools/build_variant.sh -DK0_ABL=n): leave the kernel after phase n -- WRONG planes
#endif
#define K0_NW (K0_THREADS / 64)
#define K0_WIN K0_THREADS   // tiles of a block's LDS window (one counter per thread)
#define K0_ACC 64     // shards of the record pool / descriptor array, each with its own 128-byte line of counters: [0] items, [1] records,
                      // [2] pool top, [3] descriptor top (10^5 blocks allocating from ONE word queue up at its memory channel)
#ifndef K0_HCAP
#define K0_HCAP 160
#endif
// K0_HCAP: read headers of a block kept in LDS (more reads in a block: the rest come from global memory)

// (record layout: lcr_dev.h)

// wave64 inclusive max-scan (same DPP pattern as wave_incl_scan; identity 0: the scanned marks are >= 0)
__device__ __forceinline__ int wave_incl_max(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false));
  return v;
}

// ---------------------------------------------------------------------------------------------
// load-time helpers (lcr_load_batch)

// the op-parallel kernel indexes ONE flat op space: every read's ops must follow the previous read's.
// out (pinned host memory, zeroed): int32 [1] = 1 if they do not; uint64 at byte 16: first op, byte 24: end of the last read's ops
__global__ void __launch_bounds__(LCR_BLOCK) k0_cig_check(const uint64_t* __restrict__ cig_off, const uint32_t* __restrict__ n_cig,
                                                           int32_t nr, int64_t n_cigar, int32_t* out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r + 1 < nr && cig_off[r + 1] != cig_off[r] + n_cig[r]) out[1] = 1;
  // out[2]: a read's ops reach beyond the caller's array (cig_off + n_cig > n_cigar)
  if (r < nr && (cig_off[r] > (uint64_t)n_cigar || (uint64_t)n_cig[r] > (uint64_t)n_cigar - cig_off[r])) out[2] = 1;
  if (r == 0) {
    uint64_t* g = reinterpret_cast<uint64_t*>(out + 4);
    g[0] = cig_off[0]; g[1] = cig_off[nr - 1] + n_cig[nr - 1];
  }
}
void launch_k0_cig_check(const uint64_t* cig_off, const uint32_t* n_cig, int32_t nr, int64_t n_cigar, int32_t* out, hipStream_t s) {
  if (nr > 0) hipLaunchKernelGGL(k0_cig_check, dim3((nr + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, cig_off, n_cig, nr, n_cigar, out);
}
// CIGARs that do not lie back to back (the ABI allows any cig_off) are copied into a contiguous array once per batch:
// new_off = exclusive scan of n_cig (launch_scan_i32), one wave per read copies
__global__ void __launch_bounds__(LCR_BLOCK) k0_cig_compact(const uint32_t* __restrict__ cigar, const uint64_t* __restrict__ cig_off,
                                                             const uint32_t* __restrict__ n_cig, const int32_t* __restrict__ new_off,
                                                             int32_t nr, uint32_t* __restrict__ out, uint64_t* __restrict__ out_off) {
  const int r = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (r >= nr) return;
  const uint32_t* src = cigar + cig_off[r];
  uint32_t* dst = out + new_off[r];
  for (uint32_t i = lane; i < n_cig[r]; i += 64) dst[i] = src[i];
  if (lane == 0) out_off[r] = (uint64_t)new_off[r];
}
void launch_k0_cig_compact(const uint32_t* cigar, const uint64_t* cig_off, const uint32_t* n_cig, const int32_t* new_off, int32_t nr,
                           uint32_t* out, uint64_t* out_off, hipStream_t s) {
  if (nr > 0) hipLaunchKernelGGL(k0_cig_compact, dim3((unsigned)(((int64_t)nr * 64 + LCR_BLOCK - 1) / LCR_BLOCK)), dim3(LCR_BLOCK), 0, s,
                                 cigar, cig_off, n_cig, new_off, nr, out, out_off);
}

// first read of every op block: the read that owns op blk * K0_OPB (thread per read: a read writes the entries of the
// block borders its ops span -- none or one for almost every read)
__global__ void __launch_bounds__(LCR_BLOCK) k0_block_reads(const ReadBin* __restrict__ rbin, int32_t nr, uint64_t cig0, int32_t opb,
                                                             int32_t n_blocks, int32_t* __restrict__ blk_first_read) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nr) return;
  if (r == 0) blk_first_read[n_blocks] = nr - 1;   // (sentinel: the last block's reads end with the batch)
  const uint64_t cb = rbin[r].cig_off - cig0, ce = cb + (uint32_t)rbin[r].n_cig;
  for (uint64_t k = (cb + opb - 1) / opb; k * opb < ce; k++) blk_first_read[k] = r;
}
void launch_k0_block_reads(const ReadBin* rbin, int32_t nr, uint64_t cig0, int32_t opb, int32_t n_blocks, int32_t* blk_first_read, hipStream_t s) {
  if (nr > 0) hipLaunchKernelGGL(k0_block_reads, dim3((nr + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, rbin, nr, cig0, opb, n_blocks, blk_first_read);
}

// ---------------------------------------------------------------------------------------------
struct K0Hdr {           // what an op needs to know about its read (48 bytes, LDS)
  int32_t refbase;       // + an op's block-wide exclusive prefix of the reference advance = the region-relative column where the op starts
  int32_t qbase;         // the same for the read offset where it starts
  int32_t qlo, qhi;      // ONT end trim (util.rs:745-751) as read offsets: aligned bases in [qlo, qhi) are kept (HiFi: all)
  int32_t vec, ftile;    // region length in columns, first tile of the region
  int32_t cend;          // block-relative index one past the read's last op
  uint32_t hi;           // bits [32, 64) of the read's M records: strand, transcript-strand class
  uint64_t seq_off;
  int32_t reb, rel_pos;  // seq_len - trailing soft clip (CIGAR vs l_seq check); pos - region start
};

struct K0Ctl {           // control block in HBM behind the tile counters (cleared with them, fetched with one copy)
  unsigned int pool_top, n_items, n_recs;
  int32_t error;
  unsigned int desc_top, pad_[3];
};

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Thread t owns the OPT consecutive ops [j0 + OPT t, j0 + OPT (t + 1)) of its block (one 16-byte load for OPT = 4): the
// prefix sums need ONE wave scan per quantity (thread-local sums first), and "the previous op" is a register, or the
// neighbour lane's last op.
template <int OPT>
__global__ void __launch_bounds__(K0_THREADS)
k0_ops(BatchView b, const ReadBin* __restrict__ rbin, const int32_t* __restrict__ blk_first_read, uint64_t cig0, uint32_t n_ops,
       int ont, int D, int32_t n_tiles, int32_t* __restrict__ tile_fill, int32_t* __restrict__ tile_nchunks, int32_t* __restrict__ tile_ndiff,
       K0Ctl* __restrict__ ctl, unsigned int* __restrict__ acct, unsigned int pool_sub, unsigned long long* __restrict__ recs,
       unsigned int desc_sub, uint32_t* __restrict__ desc_tile, uint2* __restrict__ desc_val, int2* __restrict__ read_scan) {
  constexpr int OPB = OPT * K0_THREADS, WOPS = OPT * 64;
  __shared__ __attribute__((aligned(8))) uint16_t ridh[OPB];   // head marks: index in the block's read list of the read whose first op sits here
  __shared__ K0Hdr hdr[K0_HCAP];
  __shared__ int ws_ref[K0_NW], ws_q[K0_NW], ws_rid[K0_NW];    // wave totals of the three scans
  __shared__ int cnt[K0_WIN], cur[K0_WIN];
  __shared__ unsigned int s_base[3];
  __shared__ int runbase[OPB];                                 // first slot of the run that starts at this op
  __shared__ int s_carry[2], s_acc[2];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const uint32_t j0 = blockIdx.x * (uint32_t)OPB;
  if (j0 >= n_ops) return;
  const uint32_t jend = min(j0 + (uint32_t)OPB, n_ops);
  const bool last_block = jend == n_ops;
  const uint32_t* __restrict__ cg = b.cigar + cig0;

  // ---- the thread's ops
  uint32_t w[OPT];
  {
    const uint32_t j = j0 + OPT * tid;
    if (OPT == 4 && j + 4 <= jend) {
      const uint4 v = *reinterpret_cast<const uint4*>(cg + j);   // (dword-aligned 16-byte load)
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
      for (int k = 0; k < OPT; k++) w[k] = j + k < jend ? cg[j + k] : 0x4u;   // (padding: 0S)
    }
  }
  const int r_lo = blk_first_read[blockIdx.x];
  // (the next block's first read is the last read that can begin in this one)
  const int n_blk_reads = blk_first_read[blockIdx.x + 1] - r_lo + 1;
#pragma unroll
  for (int k = 0; k < OPT; k++) ridh[k * K0_THREADS + tid] = 0;
  cnt[tid] = 0;
  if (tid == 0) { s_acc[0] = 0; s_acc[1] = 0; }
  __syncthreads();

  // ---- 1. heads: headers of the block's reads -> LDS, first ops marked; reads without any op are settled here
  auto load_hdr = [&](int r, uint64_t* cb_out, int* ncig_out) {
    const ReadBin h = rbin[r];
    K0Hdr x;
    x.refbase = h.rel_pos; x.qbase = h.lead > 0 ? h.lead : 0;
    x.qlo = ont ? h.lead + D : -(1 << 30); x.qhi = ont ? h.reb - D + 1 : (1 << 30);
    x.vec = h.vec; x.ftile = h.ftile; x.seq_off = h.seq_off; x.reb = h.reb; x.rel_pos = h.rel_pos;
    const int strand = h.flags & 1, ts = (h.flags >> 1) & 3;
    const uint32_t tscls = ts == 0 ? 0u : ((strand == 0) == (ts == 1) ? 1u : 2u);
    x.hi = ((uint32_t)strand << 28) | (tscls << 29);
    const uint64_t cb = h.cig_off - cig0;
    x.cend = (int32_t)min<int64_t>((int64_t)cb + h.n_cig - (int64_t)j0, INT_MAX);
    *cb_out = cb; *ncig_out = h.n_cig;
    return x;
  };
  auto settle_empty = [&](int r, const K0Hdr& x) {   // a read without a CIGAR op: reference end = start, l_seq must be all soft clip
    b.read_rend[r] = x.rel_pos;
    if (x.qbase != x.reb) atomicExch(&ctl->error, 2);
  };
  for (int rr = tid; rr < n_blk_reads; rr += K0_THREADS) {
    uint64_t cb; int ncig;
    const K0Hdr x = load_hdr(r_lo + rr, &cb, &ncig);
    // (a read without ops whose offset is the block's end sits in front of the read that owns that op: it is this block's --
    // the next block starts at the owner)
    if (rr == 0 || cb < (uint64_t)jend || (cb == (uint64_t)jend && (last_block || ncig == 0))) {
      if (rr < K0_HCAP) hdr[rr] = x;
      if (rr > 0) { if (ncig > 0) ridh[cb - j0] = (uint16_t)rr; else settle_empty(r_lo + rr, x); }
    }
  }
  if (blockIdx.x == 0)   // reads without ops in front of the first read that has one
    for (int r = tid; r < r_lo; r += K0_THREADS) { uint64_t cb; int ncig; const K0Hdr x = load_hdr(r, &cb, &ncig); settle_empty(r, x); }
  // ops the first read has in earlier blocks: their reference / query advance is this block's carry-in
  if (wv == 0) {
    const uint64_t cb0 = rbin[r_lo].cig_off - cig0;
    int cr = 0, cq = 0;
    for (uint64_t j = cb0 + lane; j < (uint64_t)j0; j += 64) {
      const uint32_t x = cg[j];
      const int op = x & 15, len = (int)(x >> 4);
      const bool m = op == 0 || op == 7 || op == 8;
      cr += (m || op == 2 || op == 3) ? len : 0;
      cq += (m || op == 1) ? len : 0;
    }
    cr = wave_incl_scan(cr); cq = wave_incl_scan(cq);
    if (lane == 63) { s_carry[0] = cr; s_carry[1] = cq; }
  }
  __syncthreads();

  if (K0_ABL == 1) return;
  // ---- 2. scan: exclusive prefixes of the reference / query advance over the block, read of every op
  int dr[OPT], dq[OPT], xr[OPT], xq[OPT], rid[OPT], mark[OPT];
  {
    bool bad_op = false;
    int tr = 0, tq = 0, tm = 0;
    const uint2 mk = *reinterpret_cast<const uint2*>(&ridh[OPT * tid]);   // (OPT == 4: four marks in one read)
#pragma unroll
    for (int k = 0; k < OPT; k++) {
      const int op = w[k] & 15, len = (int)(w[k] >> 4);
      const bool m = op == 0 || op == 7 || op == 8;
      if (!(m || (op >= 1 && op <= 5))) bad_op = true;   // P, B and the undefined codes (util.rs:944 panics)
      dr[k] = (m || op == 2 || op == 3) ? len : 0;
      dq[k] = (m || op == 1) ? len : 0;
      mark[k] = OPT == 4 ? (int)(((k < 2 ? mk.x : mk.y) >> (16 * (k & 1))) & 0xffffu) : (int)ridh[OPT * tid + k];
      xr[k] = tr; xq[k] = tq;          // thread-local exclusive
      tr += dr[k]; tq += dq[k];
      tm = max(tm, mark[k]); rid[k] = tm;
    }
    if (bad_op) atomicExch(&ctl->error, 1);
    const int ir = wave_incl_scan(tr), iq = wave_incl_scan(tq), im = wave_incl_max(tm);
    if (lane == 63) { ws_ref[wv] = ir; ws_q[wv] = iq; ws_rid[wv] = im; }
    __syncthreads();
    int br = ir - tr, bq = iq - tq, bm = __shfl_up(im, 1, 64);
    if (lane == 0) bm = 0;
    for (int i = 0; i < wv; i++) { br += ws_ref[i]; bq += ws_q[i]; bm = max(bm, ws_rid[i]); }
#pragma unroll
    for (int k = 0; k < OPT; k++) {
      xr[k] += br; xq[k] += bq; rid[k] = max(rid[k], bm);
      if (mark[k] != 0) {   // this op is its read's first: from here on the read's ops subtract the prefix reached here
        if (mark[k] < K0_HCAP) { hdr[mark[k]].refbase -= xr[k]; hdr[mark[k]].qbase -= xq[k]; }
        else read_scan[r_lo + mark[k]] = make_int2(xr[k], xq[k]);
      }
    }
    if (tid == 0) { hdr[0].refbase += s_carry[0]; hdr[0].qbase += s_carry[1]; }
  }
  __syncthreads();

  if (K0_ABL == 2) { if (xr[0] + xq[1] + rid[2] == 0x7fffffff) ctl->error = 9; return; }
  // ---- per op: the columns [a, e) it contributes records for (util.rs:692-947), its tiles
  // window of tiles kept in LDS: from the first tile the block's first read can touch (reads are position-sorted inside
  // a region, regions own consecutive tiles)
  int a[OPT], e[OPT], tb[OPT], kind[OPT];   // columns, region's first tile, kind: 0 none, 1 M, 2 D, 3 I, 4 N
  unsigned int rlo[OPT], rhi[OPT];          // record bits that do not depend on the tile: M: offset of the base on column 0 (low word, high byte), strand, ts
  unsigned int my_items = 0;
  const int win0 = hdr[0].ftile + (int)((unsigned int)max(hdr[0].rel_pos - 1, 0) / (unsigned int)LCR_TILE);   // first tile of the LDS window
#pragma unroll
  for (int k = 0; k < OPT; k++) {
    const int p = OPT * tid + k;
    const bool act = j0 + p < jend;
    K0Hdr H;
    if (rid[k] < K0_HCAP) H = hdr[rid[k]];
    else {
      uint64_t cb; int ncig;
      H = load_hdr(r_lo + rid[k], &cb, &ncig);
      const int2 sb = read_scan[r_lo + rid[k]];
      H.refbase -= sb.x; H.qbase -= sb.y;
    }
    const int op = w[k] & 15, len = (int)(w[k] >> 4);
    const int rs = H.refbase + xr[k];   // region-relative column where this op starts
    const int qs = H.qbase + xq[k];     // read offset where this op starts
    const bool m = act && (op == 0 || op == 7 || op == 8), d = act && op == 2, ins = act && op == 1, isn = act && op == 3;
    int aa = max(rs, 0), ee = min(rs + len, H.vec);
    if (m) { aa = max(aa, rs + (H.qlo - qs)); ee = min(ee, rs + (H.qhi - qs)); }   // ONT end trim (util.rs:745-751)
    int kd = ((m || d || isn) && len > 0 && ee > aa) ? (m ? 1 : d ? 2 : 4) : 0;
    if (ins && len > 0 && rs >= 1 && rs < H.vec) { kd = 3; aa = rs - 1; ee = rs; }
    kind[k] = kd; a[k] = aa; e[k] = ee; tb[k] = H.ftile;
    if (kd == 1) {   // byte offset of the base that would sit on column 0 of the region; the record adds its first column
      const unsigned long long o0 = H.seq_off + (unsigned long long)(long long)(qs - rs);
      rlo[k] = (unsigned int)o0; rhi[k] = ((unsigned int)(o0 >> 32) & 0xffu) | H.hi;
    } else {
      rlo[k] = kd == 2 ? (unsigned int)REC_KIND_D : kd == 3 ? (unsigned int)REC_KIND_I : (unsigned int)REC_KIND_N;
      rhi[k] = 0xffu;
    }
    if (kd) my_items++;
    if (act && p == H.cend - 1) {   // the read's last op: reference end for K2 / K3, CIGAR vs l_seq / soft clips
      b.read_rend[r_lo + rid[k]] = rs + dr[k];
      if (qs + dq[k] != H.reb) atomicExch(&ctl->error, 2);
    }
    if (kd == 4) {   // tiles an intron covers entirely: +1 from the tile after its first to the tile before its last
      const int ta = H.ftile + (int)((unsigned int)aa / (unsigned int)LCR_TILE), te = H.ftile + (int)((unsigned int)(ee - 1) / (unsigned int)LCR_TILE);
      if (te > ta + 1) { atomicAdd(&tile_ndiff[ta + 1], 1); atomicAdd(&tile_ndiff[te], -1); }
    }
  }

  if (K0_ABL == 3) { int x = 0; for (int k = 0; k < OPT; k++) x += a[k] + e[k] + tb[k] + kind[k] + (int)rlo[k] + (int)rhi[k]; if (x == 0x7fffffff) ctl->error = 9; return; }
  // record of op k in tile t (global tile index): columns [max(a, c0), min(e, c0 + LCR_TILE)) with c0 = (t - tb) * LCR_TILE
  auto make_rec = [&](int k, int t) -> unsigned long long {
    const int c0 = (t - tb[k]) * LCR_TILE;
    const int c_lo = max(a[k], c0), c_hi = min(e[k], c0 + LCR_TILE);
    unsigned int lo = rlo[k], hi = rhi[k] | ((unsigned int)(c_lo - c0) << 8) | ((unsigned int)(c_hi - c_lo - 1) << 18);
    if (kind[k] == 1) {   // 40-bit add of the column
      const unsigned int s = lo + (unsigned int)c_lo;
      hi = (hi & ~0xffu) | (((hi & 0xffu) + (s < lo ? 1u : 0u)) & 0xffu);
      lo = s;
    }
    return ((unsigned long long)hi << 32) | lo;
  };
  // (kind != 0: 0 <= a < e, so the divisions are unsigned shifts)
  auto first_tile = [&](int k) { return tb[k] + (int)((unsigned int)a[k] / (unsigned int)LCR_TILE); };
  auto last_tile = [&](int k) { return tb[k] + (int)((unsigned int)(e[k] - 1) / (unsigned int)LCR_TILE); };
  // an intron only leaves records in its first and last tile; the other kinds in every tile they span
  auto next_tile = [&](int k, int t, int te) { return kind[k] == 4 ? te : t + 1; };

  // ---- 3. runs of ops whose first tile is the same (ops are position-ordered inside a read and reads inside a region) and
  // count: one LDS add per run, by its last op; the other tiles of an op that crosses a tile border are added one by one
  int key[OPT], rank[OPT], hpos[OPT];
  bool is_last[OPT];
  {
#pragma unroll
    for (int k = 0; k < OPT; k++) {
      const int tp = kind[k] ? first_tile(k) - win0 : -1;
      key[k] = (tp >= 0 && tp < K0_WIN) ? tp : -1;
    }
    int pk = __shfl_up(key[OPT - 1], 1, 64);
    if (lane == 0) pk = -2;   // (a run never crosses a wave)
    bool head[OPT];
    int hp = 0;               // 1 + position (inside the wave's ops) of the last run head so far, 0: none in this thread yet
#pragma unroll
    for (int k = 0; k < OPT; k++) {
      head[k] = key[k] != (k == 0 ? pk : key[k - 1]);
      if (head[k]) hp = OPT * lane + k + 1;
      hpos[k] = hp;
    }
    const int ih = wave_incl_max(hp);
    int bh = __shfl_up(ih, 1, 64);
    if (lane == 0) bh = 0;
    const int nh = __shfl_down(head[0] ? 1 : 0, 1, 64);
#pragma unroll
    for (int k = 0; k < OPT; k++) {
      hpos[k] = max(hpos[k], bh) - 1;
      rank[k] = OPT * lane + k - hpos[k];
      is_last[k] = k + 1 < OPT ? head[k + 1] : (lane == 63 || nh != 0);
      if (is_last[k] && key[k] >= 0) atomicAdd(&cnt[key[k]], rank[k] + 1);
      if (kind[k]) {
        const int te = last_tile(k);
        for (int t = first_tile(k); t < te;) {
          t = next_tile(k, t, te);
          const int wi = t - win0;
          if (wi >= 0 && wi < K0_WIN) atomicAdd(&cnt[wi], 1);
        }
      }
    }
  }
  __syncthreads();

  if (K0_ABL == 4) { int x = 0; for (int k = 0; k < OPT; k++) x += rank[k] + hpos[k] + (is_last[k] ? 1 : 0); if (x == 0x7fffffff) ctl->error = 9; return; }
  // ---- 4. reserve: ONE allocation per block.  Exclusive prefix of the window's counters = where a tile's group starts
  // inside the block's span of the pool; the groups are announced by chunk descriptors.
  unsigned int my_recs = 0;
  const unsigned int shard = blockIdx.x % K0_ACC;
  {
    static_assert(K0_WIN == K0_THREADS, "one window counter per thread");
    const int c = cnt[tid];
    const int c16 = (c + 15) & ~15;   // a group takes whole 16-slot units of the pool (K1 finds slot j in entry j >> 4)
    const int ic = wave_incl_scan(c16);
    const unsigned long long nzb = __ballot(c > 0);
    if (lane == 63) ws_ref[wv] = ic;
    if (lane == 0) ws_q[wv] = __popcll(nzb);
    __syncthreads();
    int before = 0, dbefore = 0, total = 0, dtotal = 0;
    for (int i = 0; i < K0_NW; i++) { if (i < wv) { before += ws_ref[i]; dbefore += ws_q[i]; } total += ws_ref[i]; dtotal += ws_q[i]; }
    if (tid == 0 && total > 0) {   // the block's shard: pool [shard * pool_sub, (shard + 1) * pool_sub), descriptors alike
      unsigned int* sh = acct + 32 * shard;
      const unsigned int at = atomicAdd(&sh[2], (unsigned int)total);
      const unsigned int dt = atomicAdd(&sh[3], (unsigned int)dtotal);
      const bool fits = at + (unsigned int)total <= pool_sub && dt + (unsigned int)dtotal <= desc_sub;
      if (!fits) atomicExch(&ctl->error, 3);
      s_base[0] = shard * pool_sub + at; s_base[1] = shard * desc_sub + dt; s_base[2] = fits ? 1u : 0u;
    }
    cur[tid] = before + ic - c16;
    my_recs += (unsigned int)c;
    __syncthreads();
    if (c > 0 && s_base[2]) {
      const int tile = win0 + tid;
      const unsigned int di = s_base[1] + (unsigned int)dbefore + (unsigned int)__popcll(nzb & ((1ull << lane) - 1ull));
      desc_tile[di] = (uint32_t)tile; desc_val[di] = make_uint2(s_base[0] + (unsigned int)(before + ic - c16), (unsigned int)c);
      atomicAdd(&tile_fill[tile], c);
      atomicAdd(&tile_nchunks[tile], c16 >> 4);
    }
  }
  const unsigned int blk_at = s_base[0];
  const bool blk_fits = s_base[2] != 0;   // (a shard that ran full: the stage is repeated with larger pools, nothing is written)

  // ---- 5. emit: the last op of a run draws the run's slots and leaves their base where the run began
  auto put = [&](int slot, unsigned long long rec) { if (blk_fits) recs[blk_at + (unsigned int)slot] = rec; };
  int* rbw = runbase + wv * WOPS;
#pragma unroll
  for (int k = 0; k < OPT; k++)
    if (is_last[k] && key[k] >= 0) rbw[hpos[k]] = atomicAdd(&cur[key[k]], rank[k] + 1);
  wave_lds_sync();
  bool overflow = false;   // this lane has records outside the LDS window
#pragma unroll
  for (int k = 0; k < OPT; k++) {
    if (key[k] >= 0) put(rbw[hpos[k]] + rank[k], make_rec(k, first_tile(k)));
    else if (kind[k]) overflow = true;
    if (kind[k]) {
      const int te = last_tile(k);
      for (int t = first_tile(k); t < te;) {
        t = next_tile(k, t, te);
        const int wi = t - win0;
        if (wi >= 0 && wi < K0_WIN) put(atomicAdd(&cur[wi], 1), make_rec(k, t));
        else overflow = true;
      }
    }
  }
  // records outside the window (a read that spans more than K0_WIN tiles; unsorted reads): a chunk of one record each
  if (overflow) {
#pragma unroll
    for (int k = 0; k < OPT; k++) {
      if (!kind[k]) continue;
      const int te = last_tile(k);
      for (int t = first_tile(k);; t = next_tile(k, t, te)) {
        if (t - win0 < 0 || t - win0 >= K0_WIN) {
          unsigned int* sh = acct + 32 * shard;
          const unsigned int at = atomicAdd(&sh[2], 1u), di = atomicAdd(&sh[3], 1u);
          if (at >= pool_sub || di >= desc_sub) atomicExch(&ctl->error, 3);
          else {
            recs[shard * pool_sub + at] = make_rec(k, t);
            desc_tile[shard * desc_sub + di] = (uint32_t)t; desc_val[shard * desc_sub + di] = make_uint2(shard * pool_sub + at, 1u);
            atomicAdd(&tile_fill[t], 1);
            atomicAdd(&tile_nchunks[t], 1);
          }
          my_recs++;
        }
        if (t >= te) break;
      }
    }
  }
  // byte accounting and pool sizing: a block's totals go to one of K0_ACC slots (10^5 blocks adding to ONE word would
  // queue up at its L2 channel); k1_tile_order sums them into the control block
  my_items = (unsigned int)wave_incl_scan((int)my_items);
  my_recs = (unsigned int)wave_incl_scan((int)my_recs);
  if (lane == 63) { atomicAdd(&s_acc[0], (int)my_items); atomicAdd(&s_acc[1], (int)my_recs); }
  __syncthreads();
  if (tid == 0) {
    unsigned int* acc = acct + 32 * shard;
    if (s_acc[0]) atomicAdd(&acc[0], (unsigned int)s_acc[0]);
    if (s_acc[1]) atomicAdd(&acc[1], (unsigned int)s_acc[1]);
  }
}

// a batch without a single CIGAR op (no op block runs): every read is settled here
__global__ void __launch_bounds__(LCR_BLOCK) k0_empty_reads(BatchView b, const ReadBin* __restrict__ rbin, K0Ctl* __restrict__ ctl) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= b.n_reads) return;
  b.read_rend[r] = rbin[r].rel_pos;
  if ((rbin[r].lead > 0 ? rbin[r].lead : 0) != rbin[r].reb) atomicExch(&ctl->error, 2);
}

int launch_k0_opb() { return K0_OPT * K0_THREADS; }
int launch_k0_acct_words() { return 32 * K0_ACC; }
int launch_k0_acct_slots() { return K0_ACC; }

// chunk descriptors -> per-tile entry lists: a group of c records becomes ceil(c / 16) entries (pool offset, count <= 16) of its
// tile's list sorted[chunk_off[t] .. chunk_off[t + 1]) (chunk_off = exclusive scan of the tiles' entry counts, k1_tile_order);
// the order inside a tile is whatever the cursors give -- K1 only adds
__global__ void __launch_bounds__(LCR_BLOCK) k0_desc_bin(const K0Ctl* __restrict__ ctl, const unsigned int* __restrict__ acct, unsigned int desc_sub,
                                                          const uint32_t* __restrict__ desc_tile, const uint2* __restrict__ desc_val,
                                                          const int32_t* __restrict__ chunk_off, int32_t* __restrict__ cursor, uint2* __restrict__ sorted) {
  if (ctl->error != 0) return;
  const unsigned int shard = blockIdx.x % K0_ACC, per = gridDim.x / K0_ACC;   // (the grid is a multiple of K0_ACC blocks)
  const unsigned int n = min(acct[32 * shard + 3], desc_sub);
  for (unsigned int i = (blockIdx.x / K0_ACC) * LCR_BLOCK + threadIdx.x; i < n; i += per * LCR_BLOCK) {
    const uint32_t t = desc_tile[shard * desc_sub + i];
    const uint2 d = desc_val[shard * desc_sub + i];
    const int ne = (int)((d.y + 15u) >> 4);
    uint2* dst = sorted + chunk_off[t] + atomicAdd(&cursor[t], ne);
    for (int e = 0; e < ne; e++) dst[e] = make_uint2(d.x + 16u * (unsigned int)e, min(16u, d.y - 16u * (unsigned int)e));
  }
}
void launch_k0_desc_bin(const void* ctl, const unsigned int* acct, unsigned int desc_sub, const uint32_t* desc_tile, const void* desc_val,
                        const int32_t* chunk_off, int32_t* cursor, void* sorted, int32_t n_blocks_hint, hipStream_t s) {
  const int per = std::max(1, std::min((n_blocks_hint + K0_ACC - 1) / K0_ACC, 32));
  hipLaunchKernelGGL(k0_desc_bin, dim3(per * K0_ACC), dim3(LCR_BLOCK), 0, s, (const K0Ctl*)ctl, acct, desc_sub, desc_tile, (const uint2*)desc_val,
                     chunk_off, cursor, (uint2*)sorted);
}

void launch_k0_ops(const BatchView& b, const ReadBin* rb, const int32_t* blk_first_read, uint64_t cig0, uint32_t n_ops, int ont, int D,
                   int32_t n_tiles, int32_t* tile_fill, int32_t* tile_nchunks, int32_t* tile_ndiff, void* ctl, unsigned int* acct,
                   unsigned int pool_sub, unsigned long long* recs, unsigned int desc_sub, uint32_t* desc_tile, void* desc_val, void* read_scan,
                   hipStream_t s) {
  if (b.n_reads == 0) return;
  if (n_ops == 0) { hipLaunchKernelGGL(k0_empty_reads, dim3((b.n_reads + LCR_BLOCK - 1) / LCR_BLOCK), dim3(LCR_BLOCK), 0, s, b, rb, (K0Ctl*)ctl); return; }
  const int opb = launch_k0_opb();
  hipLaunchKernelGGL(k0_ops<K0_OPT>, dim3((n_ops + opb - 1) / opb), dim3(K0_THREADS), 0, s, b, rb, blk_first_read, cig0, n_ops, ont, D, n_tiles,
                     tile_fill, tile_nchunks, tile_ndiff, (K0Ctl*)ctl, acct, pool_sub, recs, desc_sub, desc_tile, (uint2*)desc_val, (int2*)read_scan);
}
