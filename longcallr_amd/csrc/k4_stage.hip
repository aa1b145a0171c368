// k4_stage.hip — K4, staging: the phase matrices of every region from K3's fragment CSR (reference src/fragment.rs:144-146,253-255;
// the per-SNP constants of cross_optimize, phase.rs:810-976).  Host control: k4_phase.hip; the all-CU form for one large region is in k4_grid.hip.
#include <climits>
#include "k4_dev.h"
#include "k4_kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------
// k4_stage: phase matrices on the device, one workgroup per region, straight from K3's fragment CSR.
// For every region: the rows with >= min_linkers linked SNPs (fragment.rs:253-255) restricted to the
// phase sites (for_phasing candidates, fragment.rs:144-146) as CSR + CSC mirror, the per-SNP constants
// of cross_optimize and the region descriptor.  Slices sit at offsets derived from K3's own offsets
// (rows: r0 + g, SNPs: c0 + g, entries: row_ptr[r0]) so no cross-region scan is needed.  The CSC fill
// order inside a column is whatever the atomics give: every consumer only sums over a column.
// ---------------------------------------------------------------------------------------------

__global__ void __launch_bounds__(STAGE_THREADS) k4_stage(StageIn in, StageOut out, PhaseLutDev lut) {
  constexpr int NW = STAGE_THREADS / 64;
  __shared__ int sm[2][16];
  __shared__ int s_max[3];   // [2]: largest distance between two for_phasing entries of one fragment row (LD band width)
  __shared__ long long s_ft[NW];
  __shared__ long long s_fe[32], s_f1e[32];
  const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = in.row_region_off[g], nrow = in.row_region_off[g + 1] - r0;
  const int c0 = in.cand_off[g], S = in.cand_off[g + 1] - c0;
  const int64_t e_base = in.row_ptr[r0];
  RegionDev rd{};
  rd.S = S; rd.rp_off = r0 + g; rd.cp_off = c0 + g; rd.e_off = e_base; rd.sig_off = r0; rd.snp_off = c0;
  rd.seed = region_seed(in.seed, in.start0[g]);
  if (S == 0) {
    if (tid == 0) { out.reg[g] = rd; out.stat[g] = StageStat{0, 0, 0, 0, 0, 0}; out.prow_ptr[rd.rp_off] = 0; out.ccol_ptr[rd.cp_off] = 0; }
    return;
  }
  if (tid < 32) { s_fe[tid] = tid < 31 ? lut.fe[tid] : 0; s_f1e[tid] = tid < 31 ? lut.f1e[tid] : 0; }
  if (tid < 3) s_max[tid] = 0;
  // The region's slice of the fragment matrix is brought into LDS with coalesced loads when it fits (any
  // realistic region does): the per-row entry loops below are chains of dependent loads, a microsecond per link
  // from HBM, and there are four of them per row.  Larger regions run the same code on global memory.
  __shared__ uint16_t s_col[STG_E];
  __shared__ uint8_t s_val[STG_E];
  __shared__ uint16_t s_rp[STG_R + 1];
  __shared__ uint8_t s_isp[STG_R];
  __shared__ uint8_t s_fp[STG_S];
  __shared__ int s_cur[STG_S];
  const int64_t E_all = in.row_ptr[r0 + nrow] - e_base;
  if (S > 0 && E_all >= in.grid_min) return;   // k4_stage_grid (k4_grid.hip) stages this region with all CUs
  const bool staged = nrow <= STG_R && E_all <= STG_E && S <= STG_S;
  for (int i = tid; i < S; i += STAGE_THREADS) {
    const lcr_candidate& c = in.cand[c0 + i];
    const uint8_t fp = (c.flags & LCR_F_FOR_PHASING) ? 1 : 0;
    out.snp_fp[c0 + i] = fp;
    out.snp_vt[c0 + i] = (int8_t)c.variant_type;
    out.snp_cons[c0 + i] = 0;
    if (staged) { s_fp[i] = fp; s_cur[i] = 0; } else out.cursor[c0 + i] = 0;
  }
  if (staged) {
    for (int e = tid; e < (int)E_all; e += STAGE_THREADS) { s_col[e] = (uint16_t)(in.col[e_base + e] - c0); s_val[e] = in.val[e_base + e]; }
    for (int r = tid; r <= nrow; r += STAGE_THREADS) s_rp[r] = (uint16_t)(in.row_ptr[r0 + r] - e_base);
    for (int r = tid; r < nrow; r += STAGE_THREADS) s_isp[r] = in.links[r0 + r] >= in.min_linkers ? 1 : 0;
  }
  __syncthreads();
  int32_t* prp = out.prow_ptr + rd.rp_off;
  int32_t* pcp = out.ccol_ptr + rd.cp_off;
  int R = 0, E = 0;
  auto build = [&](auto staged_tag) {
    constexpr bool ST = decltype(staged_tag)::value;
    auto isp_of = [&](int r) -> int { if constexpr (ST) return s_isp[r]; else return in.links[r0 + r] >= in.min_linkers ? 1 : 0; };
    auto rp_of = [&](int r) -> int { if constexpr (ST) return s_rp[r]; else return (int)(in.row_ptr[r0 + r] - e_base); };   // region relative
    auto col_of = [&](int e) -> int { if constexpr (ST) return s_col[e]; else return in.col[e_base + e] - c0; };              // region relative
    auto val_of = [&](int e) -> uint8_t { if constexpr (ST) return s_val[e]; else return in.val[e_base + e]; };
    auto fp_of = [&](int i) -> bool { if constexpr (ST) return s_fp[i] != 0; else return out.snp_fp[c0 + i] != 0; };
    auto bump = [&](int i) -> int { if constexpr (ST) return atomicAdd(&s_cur[i], 1); else return atomicAdd(&out.cursor[c0 + i], 1); };
    // ---- pass 1: phasing rows and their phase-site entries (CSR), column counts
    for (int base = 0; base < nrow; base += STAGE_THREADS) {
      const int r = base + tid;
      int isp = 0, cnt = 0, eb = 0, ee = 0;
      if (r < nrow) {
        isp = isp_of(r);
        eb = rp_of(r); ee = rp_of(r + 1);
        int first = -1, last = -1;   // every fragment row counts for the LD pair table (fragment.rs:208-240)
        for (int e = eb; e < ee; e++) { const int ci = col_of(e); if (fp_of(ci)) { cnt++; if (first < 0) first = ci; last = ci; } }
        if (last > first) atomicMax(&s_max[2], last - first);
        if (!isp) cnt = 0;
      }
      int k, eo, tk, te;
      block_scan2n<NW, 16>(isp, cnt, k, eo, tk, te, sm);
      if (isp) {
        k += R; eo += E;
        prp[k] = eo;
        out.prow_src[r0 + k] = r;
        for (int e = eb; e < ee; e++) {
          const int ci = col_of(e);
          if (!fp_of(ci)) continue;
          out.pcol[e_base + eo] = ci; out.pval[e_base + eo] = val_of(e) & 63;
          bump(ci);
          eo++;
        }
      }
      R += tk; E += te;
    }
    if (tid == 0) prp[R] = E;
    __syncthreads();
    // ---- column offsets
    {
      int carry = 0;
      for (int base = 0; base < S; base += STAGE_THREADS) {
        const int i = base + tid;
        int v = 0;
        if (i < S) { if constexpr (ST) v = s_cur[i]; else v = out.cursor[c0 + i]; }
        int ex, dummy, tot, tdummy;
        block_scan2n<NW, 16>(v, 0, ex, dummy, tot, tdummy, sm);
        if (i < S) { pcp[i] = carry + ex; if constexpr (ST) s_cur[i] = carry + ex; else out.cursor[c0 + i] = carry + ex; }
        carry += tot;
      }
      if (tid == 0) pcp[S] = carry;
    }
    __syncthreads();
    // ---- pass 2: CSC mirror (phasing-row index, value)
    {
      int Rk = 0;
      for (int base = 0; base < nrow; base += STAGE_THREADS) {
        const int r = base + tid;
        const int isp = r < nrow ? isp_of(r) : 0;
        int k, dummy, tk, tdummy;
        block_scan2n<NW, 16>(isp, 0, k, dummy, tk, tdummy, sm);
        if (isp) {
          k += Rk;
          const int ee = rp_of(r + 1);
          for (int e = rp_of(r); e < ee; e++) {
            const int ci = col_of(e);
            if (!fp_of(ci)) continue;
            const int pos = bump(ci);
            out.crow[e_base + pos] = k; out.cval[e_base + pos] = val_of(e) & 63;
          }
        }
        Rk += tk;
      }
    }
    __syncthreads();
  };
  if (staged) build(std::true_type{}); else build(std::false_type{});
  // ---- per-SNP constants: F = sum fe, W = sum w, Cref = sum (p==+1 ? f1e : fe), Cvar = sum (p==-1 ? f1e : fe)
  long long ft = 0;
  for (int i = wave; i < S; i += NW) {
    long long F = 0, W = 0, Cr = 0, Cv = 0;
    for (int e = pcp[i] + lane; e < pcp[i + 1]; e += 64) {
      const uint8_t v = out.cval[e_base + e];
      const long long fe = s_fe[v & 31], f1 = s_f1e[v & 31];
      F += fe; W += f1 - fe;
      Cr += (v & 32) ? f1 : fe; Cv += (v & 32) ? fe : f1;
    }
    F = wave_sum_ll_dpp(F); W = wave_sum_ll_dpp(W); Cr = wave_sum_ll_dpp(Cr); Cv = wave_sum_ll_dpp(Cv);
    if (lane == 0) { long long* sc = out.snp_const + 4ll * (c0 + i); sc[0] = F; sc[1] = W; sc[2] = Cr; sc[3] = Cv; }
    ft += F;
  }
  if (lane == 0) s_ft[wave] = ft;
  // ---- per-lane share of k4_enum_reg's row partition (enumeration regions only)
  if (S <= (int)in.max_enum_snps && tid < 64) {
    const uint32_t c = enum_chunk((uint32_t)E);
    auto lower = [&](uint32_t target) { int lo = 0, hi = R; while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint32_t)prp[mid] < target) lo = mid + 1; else hi = mid; } return lo; };
    const int f0 = lower((uint32_t)tid * c), f1 = lower((uint32_t)(tid + 1) * c);
    atomicMax(&s_max[0], prp[f1] - prp[f0]);
    atomicMax(&s_max[1], f1 - f0);
  }
  __syncthreads();
  if (tid == 0) {
    long long ftot = 0;
    for (int w = 0; w < NW; w++) ftot += s_ft[w];
    rd.R = R; rd.f_total = ftot;
    out.reg[g] = rd;
    out.stat[g] = StageStat{R, E, max(s_max[0], (int)enum_chunk((uint32_t)E)), s_max[1], (int)E_all, s_max[2]};
  }
}

}  // namespace

void launch_k4_stage(int32_t n_regions, hipStream_t s, const StageIn& in, const StageOut& out, const PhaseLutDev& lut) {
  if (n_regions > 0) hipLaunchKernelGGL(k4_stage, dim3((unsigned)n_regions), dim3(STAGE_THREADS), 0, s, in, out, lut);
}
