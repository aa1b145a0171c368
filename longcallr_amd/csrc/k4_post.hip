// k4_post.hip — K4, post-phase steps with one workgroup per region (reference src/thread.rs:168-201, snpfrags.rs:191-733);
// the steps themselves are k4_post.h's post_run.  Host control: k4_phase.hip; the all-CU form for one large region is in k4_grid.hip.
#include <climits>
#include "k4_dev.h"
#include "k4_grid.h"
#include "k4_kernels.h"
#include "k4_post.h"

namespace {

// ---------------------------------------------------------------------------------------------
// k4_post: the post-phase sequence of thread.rs:168-201, one workgroup per region with the region's fragment rows
// staged in LDS and a row-ordered column index (stable counting sort by one wave per row part); the steps
// themselves are k4_post.h's post_run, shared with the all-CUs-on-one-region form (k4_gpost, k4_grid.hip).
// ---------------------------------------------------------------------------------------------
template <int NT>
__global__ void __launch_bounds__(NT) k4_post(PostIn in, const int32_t* __restrict__ slots, int32_t n_slots, PostLut lut) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  constexpr int NW = NT / 64;
  __shared__ int sm[2][16];
  __shared__ long long red[NW];
  __shared__ int wg_bc;
  __shared__ double stage[NW * 4 * POST_SSTR];
  if ((int)blockIdx.x >= n_slots) return;
  const int g = slots[blockIdx.x], tid = threadIdx.x, lane = tid & 63;
  const int r0 = in.row_region_off[g], nrow = in.row_region_off[g + 1] - r0;
  const int c0 = in.cand_off[g], S = in.cand_off[g + 1] - c0;
  if (S == 0) return;
  const int64_t e_base = in.row_ptr[r0];
  const int E = (int)(in.row_ptr[r0 + nrow] - e_base);
  const PostLayout L = post_layout(nrow, E, S);
  PostView<uint16_t> v;
  v.g = g; v.S = S; v.nrow = nrow; v.E = E; v.F = in.reg[g].R; v.r0 = r0; v.c0 = c0;
  v.le = (double*)lds; v.l1e = v.le + 32;
  v.sps = (double*)(lds + L.sps); v.rpa = (double*)(lds + L.rpa); v.rpb = (double*)(lds + L.rpb);
  v.sflags = (uint32_t*)(lds + L.sflags); v.soflags = (uint32_t*)(lds + L.soflags);
  v.parent = (int32_t*)(lds + L.parent);
  v.rptr = (uint16_t*)(lds + L.rptr); v.ecol = (uint16_t*)(lds + L.ecol);
  v.erow = (uint16_t*)(lds + L.erow); v.cent = (uint16_t*)(lds + L.cent);
  v.ccptr = (uint16_t*)(lds + L.ccptr);
  v.ev = lds + L.eval;
  v.tag = (int8_t*)(lds + L.tag); v.asg = lds + L.asg; v.fp = lds + L.fp; v.lok = lds + L.lok;
  v.dirty = lds + L.dirty;
  v.shap = (int8_t*)(lds + L.shap); v.sgt = (int8_t*)(lds + L.sgt); v.svt = (int8_t*)(lds + L.svt);
  v.rcode = lds + L.rcode;
  v.cand = in.cand + c0;
  v.stage = stage;
  uint16_t* rptr = v.rptr; uint16_t* ecol = v.ecol; uint16_t* erow = v.erow; uint16_t* cent = v.cent; uint16_t* ccptr = v.ccptr;

  int n_mark = 0;
  auto mark = [&]() { if (in.dbg_clk && tid == 0) in.dbg_clk[(size_t)g * 16 + n_mark] = (long long)wall_clock64(); n_mark++; };
  mark();
  // ---- stage: LUT, SNP state, rows, entries, row-ordered column index
  if (tid < 31) { v.le[tid] = lut.le[tid]; v.l1e[tid] = lut.l1e[tid]; }
  for (int i = tid; i < S; i += NT) {
    v.sflags[i] = v.soflags[i] = v.cand[i].flags;
    v.shap[i] = in.st_delta[c0 + i]; v.sgt[i] = in.st_eta[c0 + i]; v.svt[i] = (int8_t)v.cand[i].variant_type;
    v.sps[i] = v.cand[i].phase_score;
    v.parent[i] = 0;
  }
  for (int r = tid; r < nrow; r += NT) {
    const int isp = in.links[r0 + r] >= in.min_linkers ? 1 : 0;
    rptr[r] = (uint16_t)(in.row_ptr[r0 + r] - e_base);
    v.lok[r] = (uint8_t)isp; v.fp[r] = (uint8_t)isp; v.asg[r] = 0; v.tag[r] = 0;
  }
  if (tid == 0) rptr[nrow] = (uint16_t)E;
  __syncthreads();
  for (int k = tid; k < v.F; k += NT) v.tag[in.prow_src[r0 + k]] = in.st_sigma[r0 + k];   // the optimiser's haplotags
  mark();
  // row-ordered column index: wave q fills the entries of the q-th part of the rows (stable inside a
  // part: 64 entries at a time in (row, column) order, equal columns keep their order), the parts'
  // slots inside a column follow each other
  int32_t* qcnt = (int32_t*)(lds + L.qcnt);
  const int rq = (nrow + NW - 1) / NW;   // rows per part (one part per wave)
  for (int i = tid; i < NW * S; i += NT) qcnt[i] = 0;
  __syncthreads();
  for (int r = tid; r < nrow; r += NT)
    for (int e = rptr[r]; e < rptr[r + 1]; e++) {
      const int ci = in.col[e_base + e] - c0;
      ecol[e] = (uint16_t)ci; erow[e] = (uint16_t)r; v.ev[e] = in.val[e_base + e];
      atomicAdd(&qcnt[(r / rq) * S + ci], 1);
    }
  __syncthreads();
  {
    int carry = 0;
    for (int base = 0; base < S; base += NT) {
      const int i = base + tid;
      int x = 0;
      if (i < S) for (int q = 0; q < NW; q++) x += qcnt[q * S + i];
      int ex, d0, tot, d1;
      block_scan2n<NW, 16>(x, 0, ex, d0, tot, d1, sm);
      if (i < S) {
        int at = carry + ex;
        ccptr[i] = (uint16_t)at;
        for (int q = 0; q < NW; q++) { const int n = qcnt[q * S + i]; qcnt[q * S + i] = at; at += n; }
      }
      carry += tot;
    }
    if (tid == 0) ccptr[S] = (uint16_t)carry;
  }
  __syncthreads();
  {
    const int q = tid >> 6;
    int32_t* cur = qcnt + q * S;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int e_lo = rptr[min(q * rq, nrow)], e_hi = rptr[min((q + 1) * rq, nrow)];
    for (int base = e_lo; base < e_hi; base += 64) {
      const int e = base + lane;
      const bool valid = e < e_hi;
      const int c = valid ? (int)ecol[e] : -1;
      unsigned long long rem = __ballot(valid);
      while (rem) {
        const int cc = __shfl(c, __ffsll((long long)rem) - 1, 64);
        const unsigned long long m = __ballot(c == cc);
        const int at = cur[cc];
        if (c == cc) cent[at + __popcll(m & below)] = (uint16_t)e;
        wave_lds_sync();
        if (lane == 0) cur[cc] = at + __popcll(m);
        rem &= ~m;
      }
      wave_lds_sync();
    }
  }
  __syncthreads();
  mark();
  WgScope sc{red, &wg_bc};
  post_run(sc, in, lut, v, mark);
}
}  // namespace

hipError_t launch_k4_post(int threads, unsigned n_blocks, size_t dyn_lds, hipStream_t s, const PostIn& in, const int32_t* slots, int32_t n_slots,
                          const PostLut& lut) {
  // 33 KB of static stage buffers + up to 64 KB of region image
  hipError_t e = hipSuccess;
  if (threads == CHAIN_THREADS) {
    if ((e = k4_set_dyn_lds_once(reinterpret_cast<const void*>(&k4_post<CHAIN_THREADS>), 96 * 1024, 4)) != hipSuccess) return e;
    hipLaunchKernelGGL(k4_post<CHAIN_THREADS>, dim3(n_blocks), dim3(CHAIN_THREADS), dyn_lds, s, in, slots, n_slots, lut);
  } else {   // CHAIN_THREADS / 2 == 2 * LCR_BLOCK
    if ((e = k4_set_dyn_lds_once(reinterpret_cast<const void*>(&k4_post<CHAIN_THREADS / 2>), 96 * 1024, 5)) != hipSuccess) return e;
    hipLaunchKernelGGL(k4_post<CHAIN_THREADS / 2>, dim3(n_blocks), dim3(CHAIN_THREADS / 2), dyn_lds, s, in, slots, n_slots, lut);
  }
  return hipSuccess;
}
