// k4_dev.h — device code shared by the K4 translation units (k4_phase.hip, k4_grid.hip): counter-based RNG,
// region / matrix descriptors, the one-workgroup cross_optimize (phase.rs:810-976), wave and workgroup scans.
// Plain kernel-argument structs are global types; functions live in an anonymous namespace (one copy per translation unit, no RDC).
#pragma once
#include <climits>
#include "lcr_phase_host.h"   // (pulls in k4_types.h)

namespace {

const double FX_SCALE = 1099511627776.0;  // 2^40

__host__ __device__ inline uint64_t mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ inline double u01(uint64_t seed, uint64_t ctr) {
  uint64_t z = mix64(seed + (ctr + 1) * 0x9E3779B97F4A7C15ULL);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
__host__ __device__ inline uint64_t region_seed(uint64_t seed, int64_t start0) { return mix64(seed + 0xD1B54A32D192ED03ULL * (uint64_t)(start0 + 1)); }

__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// A sigma decision whose fixed-point sums tie exactly (A == B): the reference-order f64 scores of the row decide
// (cal_sigma_delta_eta_log, phase.rs:77-96: the three running sums over the row's entries in list order, q = 1 - log_q1 /
// (log_q2 + log_q3); flip iff q < qn, phase.rs:845-858).  le / l1e = the host's libm values of log10(eps_q), log10(1 - eps_q).
// *het: the row has an entry at a het site (a row without one scores the same for both signs, term by term: no census entry).
// (get(i, &d, &eta): delta / eta of SNP i in the state the row is scored against -- byte arrays, or the bit masks of the batched rounds)
template <class Get>
__device__ __forceinline__ bool tie_row_flips_g(const int32_t* rp, const int32_t* pc, const uint8_t* pv, int row, Get get,
                                                int sigma, const double* le, const double* l1e, bool* het) {
  double lp = 0.0, lm = 0.0;   // log_q2 (sigma = +1), log_q3 (sigma = -1)
  bool h = false;
  if (rp[row + 1] - rp[row] <= 2) {   // two entries: a tie means a + b against b + a -- the same double; nothing to decide
    for (int e = rp[row]; e < rp[row + 1]; e++) { int d, eta; get(pc[e], d, eta); h |= eta == 0; }
    *het = h;
    return false;
  }
  for (int e = rp[row]; e < rp[row + 1]; e++) {
    const int i = pc[e];
    const uint8_t v = pv[e];
    int d, eta;
    get(i, d, eta);
    const int p = (v & 32) ? 1 : -1, q = v & 31;
    const int xp = eta == 0 ? d : eta, xm = eta == 0 ? -d : eta;   // x of aki for sigma = +1 / -1 (phase.rs:32-49)
    h |= eta == 0;
    lp += p == xp ? l1e[q] : le[q];
    lm += p == xm ? l1e[q] : le[q];
  }
  *het = h;
  if (!h || lp == lm) return false;
  const double l1 = sigma == 1 ? lp : lm, l1n = sigma == 1 ? lm : lp, den = lp + lm;
  const double q = 1.0 - l1 / den, qn = 1.0 - l1n / den;
  return q < qn;
}
__device__ __forceinline__ bool tie_row_flips(const int32_t* rp, const int32_t* pc, const uint8_t* pv, int row, const int8_t* dl, const int8_t* et,
                                              int sigma, const double* le, const double* l1e, bool* het) {
  return tie_row_flips_g(rp, pc, pv, row, [&](int i, int& d, int& eta) { d = dl[i]; eta = et[i]; }, sigma, le, l1e, het);
}
// census + decision of one tied row (all sigma-step forms): true = flip
template <class Get>
__device__ __forceinline__ bool tie_row_decide_g(const PhaseDev& P, const int32_t* rp, const int32_t* pc, const uint8_t* pv, int row, Get get,
                                                 int sigma, const double* le, const double* l1e) {
  bool het;
  if (P.tie_arith < 2) {
    het = false;
    for (int e = rp[row]; e < rp[row + 1]; e++) { int d, eta; get(pc[e], d, eta); het |= eta == 0; }
    if (het) TIE_COUNT(P.tie_ctr, TIE_SIGMA_UNRES, 1ull);
    return false;
  }
  const bool f = tie_row_flips_g(rp, pc, pv, row, get, sigma, le, l1e, &het);
  if (het) { TIE_COUNT(P.tie_ctr, TIE_SIGMA_F64, 1ull); if (f) TIE_COUNT(P.tie_ctr, TIE_SIGMA_FLIPS, 1ull); }
  return f;
}
__device__ __forceinline__ bool tie_row_decide(const PhaseDev& P, const int32_t* rp, const int32_t* pc, const uint8_t* pv, int row, const int8_t* dl,
                                               const int8_t* et, int sigma, const double* le, const double* l1e) {
  return tie_row_decide_g(P, rp, pc, pv, row, [&](int i, int& d, int& eta) { d = dl[i]; eta = et[i]; }, sigma, le, l1e);
}

// one cross_optimize (phase.rs:810-976); returns the exact objective (phase.rs:257-276) to all threads.
// Every emission term is fe[q] + hit * w[q] with w[q] = f1e[q] - fe[q] > 0 and hit = [p == x]
// (aki, phase.rs:32-49), so per row / column only the data dependent sum of w over the hits is
// accumulated; the sigma/delta independent parts are per-SNP constants (PhaseDev::snp_const).
// a region's phase matrix: global memory, or a copy the calling kernel staged in LDS
struct MatView { const int32_t* rp; const int32_t* pc; const uint8_t* pv; const int32_t* cp; const int32_t* cr; const uint8_t* cv;
                 const uint8_t* fp; const uint8_t* cons; };
__device__ __forceinline__ MatView global_view(const PhaseDev& P, const RegionDev& rd) {
  return MatView{P.prow_ptr + rd.rp_off, P.pcol + rd.e_off, P.pval + rd.e_off, P.ccol_ptr + rd.cp_off, P.crow + rd.e_off,
                 P.cval + rd.e_off, P.snp_fp + rd.snp_off, P.snp_cons + rd.snp_off};
}
__host__ __device__ inline uint32_t matview_bytes(uint32_t R, uint32_t S, uint32_t E) {
  return 4 * (R + 1) + 4 * (S + 1) + 8 * E + 2 * ((E + 3) & ~3u) + 2 * ((S + 3) & ~3u);
}
// copy the region's matrix into LDS at `dst` (4-byte aligned); all threads of the workgroup call
__device__ __forceinline__ MatView stage_view(const PhaseDev& P, const RegionDev& rd, uint8_t* dst, uint32_t E) {
  const MatView g = global_view(P, rd);
  int32_t* rp = (int32_t*)dst; int32_t* cp = rp + rd.R + 1; int32_t* pc = cp + rd.S + 1; int32_t* cr = pc + E;
  uint8_t* pv = (uint8_t*)(cr + E); uint8_t* cv = pv + ((E + 3) & ~3u); uint8_t* fp = cv + ((E + 3) & ~3u); uint8_t* cons = fp + ((rd.S + 3) & ~3u);
  for (int i = threadIdx.x; i <= rd.R; i += blockDim.x) rp[i] = g.rp[i];
  for (int i = threadIdx.x; i <= rd.S; i += blockDim.x) cp[i] = g.cp[i];
  for (int i = threadIdx.x; i < (int)E; i += blockDim.x) { pc[i] = g.pc[i]; cr[i] = g.cr[i]; pv[i] = g.pv[i]; cv[i] = g.cv[i]; }
  for (int i = threadIdx.x; i < rd.S; i += blockDim.x) { fp[i] = g.fp[i]; cons[i] = g.cons[i]; }
  __syncthreads();
  return MatView{rp, pc, pv, cp, cr, cv, fp, cons};
}

#define LCR_DPP_LL(v, ctrl, rmask) \
  (((long long)__builtin_amdgcn_update_dpp(0, (int)((v) >> 32), ctrl, rmask, 0xf, false) << 32) | \
   (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), ctrl, rmask, 0xf, false))
// wave64 sum of int64 through DPP row shifts / broadcasts; every lane gets the total
__device__ __forceinline__ long long wave_sum_ll_dpp(long long v) {
  v += LCR_DPP_LL(v, 0x111, 0xf);
  v += LCR_DPP_LL(v, 0x112, 0xf);
  v += LCR_DPP_LL(v, 0x114, 0xf);
  v += LCR_DPP_LL(v, 0x118, 0xf);
  v += LCR_DPP_LL(v, 0x142, 0xa);
  v += LCR_DPP_LL(v, 0x143, 0xc);
  const int lo = __builtin_amdgcn_readlane((int)v, 63), hi = __builtin_amdgcn_readlane((int)(v >> 32), 63);
  return ((long long)hi << 32) | (unsigned)lo;
}
constexpr int CROSS_MACC = 2048;   // SNPs with a per-column accumulator in LDS (entry-balanced delta step)
// With accumulators in LDS (macc: one word per SNP, racc: one per row) all three sweeps -- sigma step, delta step,
// objective -- are balanced over the CSC entries: thread t owns the entries [t*c, (t+1)*c), loads them four at a time
// (row and value byte, then the row's sigma: independent LDS reads) and adds its terms into the accumulators (integer,
// order-free).  A thread per row / a wave per SNP instead waits for the longest row or column in every iteration.
// What bounds a sweep now is the LDS itself: ~7 bank-cycles per entry (row 4 B, value 1 B, sigma 1 B, w 8 B, atomic
// 8 B; measured 2 900 cycles for 4 242 entries, the same with half the waves doing twice the batches, PMC-free check
// with s_memtime), so fewer, wider accesses per entry are what would make it faster, not more parallelism.
__device__ __forceinline__ long long cross_optimize(const PhaseDev& P, const RegionDev& rd, const MatView& mv, int8_t* sg, int8_t* dl, int8_t* et,
                                    bool keep_conserved, bool with_genotype, long long* red, const long long* wl,
                                    unsigned long long* macc = nullptr /* macc_cap zeros in LDS, or nullptr */, int macc_cap = CROSS_MACC,
                                    unsigned long long* racc = nullptr /* racc_cap words of LDS (any content), or nullptr */, int racc_cap = 0,
                                    const long long* snp_const_lds = nullptr /* the region's 4 S per-SNP constants in LDS, or nullptr */,
                                    int* iters_out = nullptr, long long* prof = nullptr /* thread 0: six step timers (LCR_PHASE_PROF) */,
                                    int* flags = nullptr /* three ints of LDS: one barrier per iteration for both "anything changed" bits */,
                                    double* qrow = nullptr /* 2 R doubles of scratch: the COMPLETE tie contract (classes 2 / 4 too) in the plain form below; S <= 32 */,
                                    unsigned long long* tie_local = nullptr /* two words of LDS: the class-2 ties and class-4 steps this call does NOT resolve are
                                                                              counted there instead of the census (the chain kernel decides what they mean) */,
                                    double* qsnp = nullptr /* 2 S doubles */, int8_t* chs = nullptr /* 2 S bytes: with qrow, the complete contract for any S */) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int32_t* rp = mv.rp;
  const int32_t* pc = mv.pc;
  const uint8_t* pv = mv.pv;
  const int32_t* cp = mv.cp;
  const int32_t* cr = mv.cr;
  const uint8_t* cv = mv.cv;
  const uint8_t* fp = mv.fp;
  const uint8_t* cons = mv.cons;
  const long long* sc = snp_const_lds ? snp_const_lds : P.snp_const + 4ll * rd.snp_off;   // (read in every delta step)
  const double* const le64 = P.lut64->le; const double* const l1e64 = P.lut64->l1e;   // (the f64 tie path: rare, from global memory)
  bool hg_inc = true, h_inc = true;
  int iters = 0;
  auto count_unres = [&](int local, int which) { if (tie_local) atomicAdd(&tie_local[local], 1ull); else TIE_COUNT(P.tie_ctr, which, 1ull); };
  long long tk0 = prof ? (long long)wall_clock64() : 0;
  auto tick = [&](int k) { if (prof) { const long long t = (long long)wall_clock64(); prof[k] += t - tk0; tk0 = t; } };
  const bool cols_balanced = macc && rd.S <= macc_cap;
  const bool rows_balanced = cols_balanced && racc && flags && rd.R <= racc_cap;   // (the three-barrier form below)
  int ce0 = 0, ce1 = 0, col_first = 0;   // this thread's CSC entries and the column of the first one
  if (cols_balanced) {
    const int E = cp[rd.S];
    const int c = (E + (int)blockDim.x - 1) / (int)blockDim.x;
    ce0 = min(E, tid * c); ce1 = min(E, ce0 + c);
    if (ce0 < ce1) { int lo = 0, hi = rd.S; while (lo < hi) { const int mid = (lo + hi) >> 1; if (cp[mid + 1] <= ce0) lo = mid + 1; else hi = mid; } col_first = lo; }
  }
  if (rows_balanced) {
    for (int row = tid; row < rd.R; row += blockDim.x) racc[row] = 0;
    __syncthreads();
  }
  tick(0);
  // fn(column, eta, delta, row, sigma of the row, value byte) for every entry of this thread, in column order
  // (pending = true: the sigma step's sums are still in racc[], a row's new sigma is its old one times their sign)
  auto sweep = [&](bool pending, auto fn) {
    if (ce0 >= ce1) return;
    constexpr int K = 4;   // entries per batch of loads
    int i = col_first, cend = cp[i + 1], d = dl[i], h = et[i];
    for (int e = ce0; e < ce1; e += K) {
      int r8[K], v8[K], s8[K];
#pragma unroll
      for (int k = 0; k < K; k++) { const int ee = min(e + k, ce1 - 1); r8[k] = cr[ee]; v8[k] = cv[ee]; }
#pragma unroll
      for (int k = 0; k < K; k++) s8[k] = sg[r8[k]];
      if (pending) {
#pragma unroll
        for (int k = 0; k < K; k++) if (reinterpret_cast<const int*>(racc)[2 * r8[k] + 1] < 0) s8[k] = -s8[k];   // (sign = high dword)
      }
#pragma unroll
      for (int k = 0; k < K; k++)
        if (e + k < ce1) {
          if (e + k >= cend) { do { i++; cend = cp[i + 1]; } while (e + k >= cend); d = dl[i]; h = et[i]; }
          fn(i, h, d, r8[k], s8[k], v8[k]);
        }
    }
  };
  // per-SNP decision of the delta / eta step (phase.rs:872-959): the best of (d,0) (-d,0) (d,+1) (d,-1); bit 0: it improves, bit 1: it changes
  // the state without improving (a tie change).  Round 6: a tie at the maximum (class 2: the first maximum is kept where the reference's f64
  // scores might pick the other) is COUNTED here too -- the chain kernels do not resolve classes 2 / 4, but lcr_get_tie_census now says so.
  auto decide_snp = [&](int i, long long M, int ncol) -> int {
    const int d = dl[i], h = et[i];
    const long long het = P.lut.f_het0 - (long long)ncol * P.lut.f_log2;  // phase.rs:136-144
    const long long F = sc[4 * i], W = sc[4 * i + 1];
    long long N[4] = {F + M + het, F + W - M + het, sc[4 * i + 2] + P.lut.f_homref, sc[4 * i + 3] + P.lut.f_homvar};
    int ch;
    bool tie = false;
    if (with_genotype) { ch = 0; for (int t = 1; t < 4; t++) if (N[t] > N[ch]) ch = t; for (int t = 0; t < 4; t++) tie |= t != ch && N[t] == N[ch]; }   // phase.rs:908-921
    else if (h == 0) { ch = N[1] > N[0] ? 1 : 0; tie = N[1] == N[0]; }                        // phase.rs:923-930
    else { ch = N[3] > N[2] ? 3 : 2; tie = N[3] == N[2]; }                                    // phase.rs:931-938
    if (tie) count_unres(0, TIE_DELTA_UNRES);
    const int cur = h == 0 ? 0 : (h == 1 ? 2 : 3);
    dl[i] = (int8_t)(ch == 1 ? -d : d);
    et[i] = (int8_t)(ch <= 1 ? 0 : (ch == 2 ? 1 : -1));
    return N[ch] > N[cur] ? 1 : (ch != cur ? 2 : 0);
  };
  if (rows_balanced) {
    // Three barriers per iteration: sigma sweep | delta sweep (sigma taken from the pending sums) | row and SNP
    // decisions + both "anything changed" bits through one LDS word (three words in rotation: the word of call k + 2
    // is cleared after the barrier of call k, when its last readers -- call k - 1 -- are done).
    // The objective comes out of the last iteration: when nothing changed in it, the delta sweep's column sums M_i
    // are the het SNPs' hit sums of the final state and a hom SNP's are constants (Cref - F, Cvar - F), so the SNP
    // decisions add them up (red[0..2] in rotation with the flag words) and the closing sweep is only needed after
    // the iteration cap.
    if (tid < 3) { flags[tid] = 0; red[tid] = 0; }
    __syncthreads();
    int fp_at = 0;
    bool settled = false;
    long long settled_sum = 0;
    while (hg_inc | h_inc) {
      sweep(false, [&](int, int h, int d, int row, int s, int v) {
        if (h == 0) { const long long w = wl[v & 31]; atomicAdd(&racc[row], (unsigned long long)((((v & 32) ? 1 : -1) == s * d) ? w : -w)); }
      });
      __syncthreads();
      tick(1);
      // rows whose sums tie exactly: the f64 scores decide (before the delta sweep takes the rows' new sigma from the signs)
      {
        for (int row = tid; row < rd.R; row += blockDim.x)
          if (racc[row] == 0ull && rp[row + 1] > rp[row] && tie_row_decide(P, rp, pc, pv, row, dl, et, sg[row], le64, l1e64)) racc[row] = 0x8000000000000000ull;   // (the marker of a tie flip: negative, and no sum of table values)
      }
      __syncthreads();
      {
        long long M = 0;
        int mi = -1;
        sweep(true, [&](int i, int, int d, int, int s, int v) {
          if (i != mi) { if (M) atomicAdd(&macc[mi], (unsigned long long)M); M = 0; mi = i; }
          if (((v & 32) ? 1 : -1) == s * d) M += wl[v & 31];
        });
        if (M) atomicAdd(&macc[mi], (unsigned long long)M);
      }
      __syncthreads();
      tick(3);
      int chg = 0;
      for (int row = tid; row < rd.R; row += blockDim.x) {
        const long long diff = (long long)racc[row];
        racc[row] = 0;
        if (diff < 0) { sg[row] = (int8_t)(-sg[row]); chg |= diff != LLONG_MIN ? 1 : 4; }   // (LLONG_MIN: the marker of a tie flip -- not an improvement; bit 2: a tie change)
      }
      long long tsum = 0;
      for (int i = tid; i < rd.S; i += blockDim.x) {
        const long long M2 = (long long)macc[i];
        macc[i] = 0;
        const int h = et[i];
        const long long term = h == 0 ? M2 : (h == 1 ? sc[4 * i + 2] - sc[4 * i] : sc[4 * i + 3] - sc[4 * i]);
        tsum += term;
        if (!fp[i] || (keep_conserved && cons[i]) || cp[i + 1] == cp[i]) continue;
        { const int dv = decide_snp(i, M2, cp[i + 1] - cp[i]); chg |= (dv & 1) ? 2 : ((dv & 2) ? 8 : 0); }
      }
      if (wave * 64 < rd.S) {   // (wave-uniform: the waves that own SNPs)
        tsum = wave_sum_ll_dpp(tsum);
        if (lane == 0 && tsum) atomicAdd(reinterpret_cast<unsigned long long*>(&red[fp_at]), (unsigned long long)tsum);
      }
      const int wchg = (__ballot(chg & 1) ? 1 : 0) | (__ballot(chg & 2) ? 2 : 0) | (__ballot(chg & 4) ? 4 : 0) | (__ballot(chg & 8) ? 8 : 0);
      if (lane == 0 && wchg) atomicOr(&flags[fp_at], wchg);
      __syncthreads();
      const int r = flags[fp_at];
      // class 4 (a step whose only changes were tie changes: "no improvement" here, the reference's sums of f64 scores might say otherwise): counted
      if (tid == 0) { if ((r & 4) && !(r & 1)) count_unres(1, TIE_STEP_UNRES); if ((r & 8) && !(r & 2)) count_unres(1, TIE_STEP_UNRES); }
      settled = (r & 3) == 0; settled_sum = red[fp_at];
      fp_at = fp_at == 2 ? 0 : fp_at + 1;
      if (tid == 0) { flags[fp_at == 2 ? 0 : fp_at + 1] = 0; red[fp_at == 2 ? 0 : fp_at + 1] = 0; }
      tick(4);
      if (!(r & 1)) h_inc = false; else { h_inc = true; hg_inc = true; }     // after the sigma step
      if (!(r & 2)) hg_inc = false; else { hg_inc = true; h_inc = true; }    // after the delta / eta step
      if (++iters > 20) break;  // phase.rs:967-972
    }
    if (settled) {
      if (iters_out) *iters_out = iters;
      __syncthreads();   // (red[] is the next call's)
      tick(5);
      return rd.f_total + settled_sum;
    }
    __syncthreads();
  } else
  while (hg_inc | h_inc) {
    // ---- sigma step (phase.rs:824-862): A - B = sum over het sites of (+w if p == sigma*delta else -w);
    //      flip every row with A < B (sites with eta != 0 contribute equally to both)
    // Complete tie contract (qrow != nullptr, P.tie_arith >= 3): beside the sigma ties, (class 2) a delta / eta choice with two equal
    // maxima takes the first maximum of the reference's f64 scores, and (class 4) a step whose only changes were tie changes is an
    // improvement iff the reference's sums of scores say so (check_new_haplotag / check_new_haplotype_genotype, phase.rs:278-355).
    // Without qrow those events are counted as unresolved ("first maximum", "no improvement").
    const bool full = qrow != nullptr && P.tie_arith >= 3 && (rd.S <= 32 || (qsnp && chs));
    __shared__ double s_qn32[32], s_qo32[32];
    __shared__ int8_t s_ch32[32], s_cur32[32];
    double* const s_qn = qsnp ? qsnp : s_qn32; double* const s_qo = qsnp ? qsnp + rd.S : s_qo32;   // (the chain kernel's regions: global scratch)
    int8_t* const s_ch = chs ? chs : s_ch32; int8_t* const s_cur = chs ? chs + rd.S : s_cur32;
    const int n_ch = chs ? rd.S : 32;
    __shared__ int s_verdict;
    int any = 0, anyt = 0;
    for (int row = tid; row < rd.R; row += blockDim.x) {
      const int s = sg[row];
      long long diff = 0;
      for (int e = rp[row]; e < rp[row + 1]; e++) {
        const int i = pc[e];
        const uint8_t v = pv[e];
        if (et[i] == 0) { const long long w = wl[v & 31]; diff += (((v & 32) ? 1 : -1) == s * dl[i]) ? w : -w; }
      }
      if (diff < 0) { sg[row] = (int8_t)(-s); any = 1; }
      else if (diff == 0 && rp[row + 1] > rp[row] && tie_row_decide(P, rp, pc, pv, row, dl, et, s, le64, l1e64)) { sg[row] = (int8_t)(full ? -2 * s : -s); anyt = 1; }   // (Jacobi: a row reads its own sigma only; +-2: flipped by a tie, until the step's verdict below)
    }
    any = __syncthreads_or(any);
    anyt = __syncthreads_or(anyt);
    if (anyt && !any) {   // a step of tie flips only
      if (!full) { if (tid == 0) count_unres(1, TIE_STEP_UNRES); }
      else {
        if (tid == 0) TIE_COUNT(P.tie_ctr, TIE_STEP_F64, 1ull);
        for (int row = tid; row < rd.R; row += blockDim.x) {   // every row's score under the new and the old sigma
          double qn = 0.0, qo = 0.0;
          if (rp[row + 1] > rp[row]) {
            double lp = 0.0, lm = 0.0;   // log_q2 (sigma = +1), log_q3 (sigma = -1), entry order (phase.rs:82-90)
            for (int e = rp[row]; e < rp[row + 1]; e++) {
              const int i = pc[e];
              const uint8_t v = pv[e];
              const int p = (v & 32) ? 1 : -1, q = v & 31, eta = et[i], d = dl[i];
              lp += p == (eta == 0 ? d : eta) ? l1e64[q] : le64[q];
              lm += p == (eta == 0 ? -d : eta) ? l1e64[q] : le64[q];
            }
            const int sv = sg[row], sn = sv < 0 ? -1 : 1, so = (sv == 2 || sv == -2) ? -sn : sn;
            const double den = lp + lm;
            qn = 1.0 - (sn == 1 ? lp : lm) / den; qo = 1.0 - (so == 1 ? lp : lm) / den;
          }
          qrow[2 * row] = qn; qrow[2 * row + 1] = qo;   // (a row without an entry is not in the reference's sums: + 0.0 changes nothing)
        }
        __syncthreads();
        if (wave == 0) {   // the two sums, rows in order
          double logp = 0.0, pre = 0.0;
          for (int base = 0; base < rd.R; base += 64) {
            const int row = base + lane;
            const double a = row < rd.R ? qrow[2 * row] : 0.0, b2 = row < rd.R ? qrow[2 * row + 1] : 0.0;
            const int alo = (int)__double2loint(a), ahi = (int)__double2hiint(a), blo = (int)__double2loint(b2), bhi = (int)__double2hiint(b2);
            const int n = min(64, rd.R - base);
            for (int k = 0; k < n; k++) {
              logp += __hiloint2double(__shfl(ahi, k, 64), __shfl(alo, k, 64));
              pre += __hiloint2double(__shfl(bhi, k, 64), __shfl(blo, k, 64));
            }
          }
          if (lane == 0) s_verdict = logp > pre ? 1 : 0;
        }
        __syncthreads();
        any = s_verdict;
      }
    }
    if (anyt && full) {   // back to +-1
      for (int row = tid; row < rd.R; row += blockDim.x) { const int sv = sg[row]; if (sv == 2) sg[row] = 1; else if (sv == -2) sg[row] = -1; }
      __syncthreads();
    }
    tick(2);
    if (!any) h_inc = false; else { h_inc = true; hg_inc = true; }
    // ---- delta/eta step (phase.rs:872-959): per SNP the best of (d,0) (-d,0) (d,+1) (d,-1)
    any = 0; anyt = 0;
    if (full) { for (int i = tid; i < n_ch; i += blockDim.x) s_ch[i] = -1; __syncthreads(); }
    // the four f64 scores of SNP i for its delta d (phase.rs:128-176): four running sums over its column in row order + priors
    auto col_scores = [&](int i, int d, double* q4) {
      double Sd = 0.0, Sn = 0.0, Shr = 0.0, Shv = 0.0;
      uint32_t cov = 0;
      for (int e = cp[i]; e < cp[i + 1]; e++) {
        const uint8_t v = cv[e];
        const int p = (v & 32) ? 1 : -1, q = v & 31, x = sg[cr[e]] * d;
        Sd += p == x ? l1e64[q] : le64[q]; Sn += p == -x ? l1e64[q] : le64[q];
        Shr += p == 1 ? l1e64[q] : le64[q]; Shv += p == -1 ? l1e64[q] : le64[q];
        cov++;
      }
      const double p_het = P.lut64->log_theta - (double)cov * P.lut64->log2;
      const double hv = Shv + P.lut64->p_homvar, hr = Shr + P.lut64->p_homref, hd = Sd + p_het, hn = Sn + p_het;
      const double den_d = hv + hd + hr + hn, den_n = hv + hn + hr + hd;
      q4[0] = 1.0 - hd / den_d; q4[1] = 1.0 - hn / den_n; q4[2] = 1.0 - hr / den_d; q4[3] = 1.0 - hv / den_d;
    };
    auto decide = [&](int i, long long M, int ncol) {
      const int d = dl[i], h = et[i];
      const long long het = P.lut.f_het0 - (long long)ncol * P.lut.f_log2;  // phase.rs:136-144
      const long long F = sc[4 * i], W = sc[4 * i + 1];
      long long N[4] = {F + M + het, F + W - M + het, sc[4 * i + 2] + P.lut.f_homref, sc[4 * i + 3] + P.lut.f_homvar};
      int ch;
      bool tie = false;
      if (with_genotype) { ch = 0; for (int t = 1; t < 4; t++) if (N[t] > N[ch]) ch = t; for (int t = 0; t < 4; t++) tie |= t != ch && N[t] == N[ch]; }   // phase.rs:908-921
      else if (h == 0) { ch = N[1] > N[0] ? 1 : 0; tie = N[1] == N[0]; }                            // phase.rs:923-930
      else { ch = N[3] > N[2] ? 3 : 2; tie = N[3] == N[2]; }                                        // phase.rs:931-938
      if (tie) {
        if (!full) count_unres(0, TIE_DELTA_UNRES);
        else {   // the first maximum of the f64 scores (among the candidates the mode looks at)
          TIE_COUNT(P.tie_ctr, TIE_STEP_F64, 1ull);
          double q4[4];
          col_scores(i, d, q4);
          int cf;
          if (with_genotype) { const double mx = fmax(q4[0], fmax(q4[1], fmax(q4[2], q4[3]))); cf = q4[0] == mx ? 0 : q4[1] == mx ? 1 : q4[2] == mx ? 2 : 3; }
          else if (h == 0) { const double mx = fmax(q4[0], q4[1]); cf = q4[0] == mx ? 0 : 1; }
          else { const double mx = fmax(q4[2], q4[3]); cf = q4[2] == mx ? 2 : 3; }
          if (N[cf] == N[ch]) ch = cf;
        }
      }
      const int cur = h == 0 ? 0 : (h == 1 ? 2 : 3);
      if (N[ch] > N[cur]) any = 1; else if (ch != cur) anyt = 1;
      if (full) { s_ch[i] = (int8_t)ch; s_cur[i] = (int8_t)cur; }
      dl[i] = (int8_t)(ch == 1 ? -d : d);
      et[i] = (int8_t)(ch <= 1 ? 0 : (ch == 2 ? 1 : -1));
    };
    if (cols_balanced) {
      // per-column sum of w over the hits; a thread's run of entries of one column is summed in a register first
      long long M = 0;
      int mi = -1;
      sweep(false, [&](int i, int, int d, int, int s, int v) {
        if (i != mi) { if (M) atomicAdd(&macc[mi], (unsigned long long)M); M = 0; mi = i; }
        if (((v & 32) ? 1 : -1) == s * d) M += wl[v & 31];
      });
      if (M) atomicAdd(&macc[mi], (unsigned long long)M);
      __syncthreads();
      tick(3);
      for (int i = tid; i < rd.S; i += blockDim.x) {
        const long long M2 = (long long)macc[i];
        macc[i] = 0;
        if (!fp[i] || (keep_conserved && cons[i]) || cp[i + 1] == cp[i]) continue;
        decide(i, M2, cp[i + 1] - cp[i]);
      }
    } else {
      for (int i = wave; i < rd.S; i += nw) {
        if (!fp[i]) continue;
        if (keep_conserved && cons[i]) continue;
        const int c0 = cp[i], c1 = cp[i + 1];
        if (c1 == c0) continue;
        const int d = dl[i];
        long long M = 0;  // sum of w over the entries with p == sigma * d
        for (int e = c0 + lane; e < c1; e += 64) {
          const uint8_t v = cv[e];
          if (((v & 32) ? 1 : -1) == sg[cr[e]] * d) M += wl[v & 31];
        }
        M = wave_sum_ll(M);
        if (lane == 0) decide(i, M, c1 - c0);
      }
    }
    any = __syncthreads_or(any);
    anyt = __syncthreads_or(anyt);
    if (anyt && !any) {   // a step of tie changes only: the sums of the SNPs' scores (check_new_haplotype_genotype, phase.rs:316-355), SNPs in order
      if (!full) { if (tid == 0) count_unres(1, TIE_STEP_UNRES); }
      else {
        if (tid == 0) TIE_COUNT(P.tie_ctr, TIE_STEP_F64, 1ull);
        for (int i = tid; i < rd.S; i += blockDim.x) {
          double qn = 0.0, qo = 0.0;
          const int ch = s_ch[i];
          if (ch >= 0) {
            double q4[4];
            col_scores(i, ch == 1 ? -dl[i] : dl[i], q4);   // (the scores are those of the delta the step started from)
            qn = q4[ch]; qo = q4[s_cur[i]];
          }
          s_qn[i] = qn; s_qo[i] = qo;
        }
        __syncthreads();
        if (tid == 0) {
          double logp = 0.0, pre = 0.0;
          for (int i = 0; i < rd.S; i++) { logp += s_qn[i]; pre += s_qo[i]; }
          s_verdict = logp > pre ? 1 : 0;
        }
        __syncthreads();
        any = s_verdict;
      }
    }
    tick(4);
    if (!any) hg_inc = false; else { hg_inc = true; h_inc = true; }
    if (++iters > 20) break;  // phase.rs:967-972
  }
  if (iters_out) *iters_out = iters;
  // ---- objective (phase.rs:257-276) = f_total + sum of w over the hits
  long long acc = 0;
  if (cols_balanced) {
    sweep(false, [&](int, int h, int d, int, int s, int v) {
      const int x = h == 0 ? s * d : h;
      if (((v & 32) ? 1 : -1) == x) acc += wl[v & 31];
    });
  } else
  for (int row = tid; row < rd.R; row += blockDim.x) {
    const int s = sg[row];
    for (int e = rp[row]; e < rp[row + 1]; e++) {
      const int i = pc[e];
      const uint8_t v = pv[e];
      const int x = et[i] == 0 ? s * dl[i] : et[i];
      if (((v & 32) ? 1 : -1) == x) acc += wl[v & 31];
    }
  }
  acc = wave_sum_ll_dpp(acc);
  __syncthreads();   // (red[] may still be read from the previous call)
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  long long total = rd.f_total;
  for (int w = 0; w < nw; w++) total += red[w];
  tick(5);
  return total;
}

// w[q] = f1e[q] - fe[q] into LDS (dynamic indexing of a kernel-argument table would go through memory)
__device__ __forceinline__ void load_w(const PhaseDev& P, long long* wl) {
  if (threadIdx.x < 32) wl[threadIdx.x] = threadIdx.x < 31 ? P.lut.f1e[threadIdx.x] - P.lut.fe[threadIdx.x] : 0;
  __syncthreads();
}

__device__ __forceinline__ int8_t init_genotype(int8_t vt) { return vt == 0 ? 1 : (vt == 1 ? 0 : -1); }  // phase.rs:682-691

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exclusive scan of two ints over a workgroup of NW waves; returns the totals through ta / tb
template <int NW, int SMW>
__device__ __forceinline__ void block_scan2n(int a, int b, int& ea, int& eb, int& ta, int& tb, int (*sm)[SMW]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int ia = a, ib = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int ua = __shfl_up(ia, d, 64), ub = __shfl_up(ib, d, 64);
    if (lane >= d) { ia += ua; ib += ub; }
  }
  __syncthreads();
  if (lane == 63) { sm[0][wave] = ia; sm[1][wave] = ib; }
  __syncthreads();
  int oa = 0, ob = 0; ta = 0; tb = 0;
  for (int w = 0; w < NW; w++) { if (w < wave) { oa += sm[0][w]; ob += sm[1][w]; } ta += sm[0][w]; tb += sm[1][w]; }
  ea = oa + ia - a; eb = ob + ib - b;
}
// the same for a workgroup whose number of waves is only known at run time (<= 16)
__device__ __forceinline__ void block_scan2_rt(int a, int b, int& ea, int& eb, int& ta, int& tb, int (*sm)[16]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int ia = a, ib = b;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int ua = __shfl_up(ia, d, 64), ub = __shfl_up(ib, d, 64);
    if (lane >= d) { ia += ua; ib += ub; }
  }
  __syncthreads();
  if (lane == 63) { sm[0][wave] = ia; sm[1][wave] = ib; }
  __syncthreads();
  int oa = 0, ob = 0; ta = 0; tb = 0;
  for (int w = 0; w < nw; w++) { if (w < wave) { oa += sm[0][w]; ob += sm[1][w]; } ta += sm[0][w]; tb += sm[1][w]; }
  ea = oa + ia - a; eb = ob + ib - b;
}
__device__ __forceinline__ void block_scan2(int a, int b, int& ea, int& eb, int& ta, int& tb, int (*sm)[8]) {
  block_scan2n<4, 8>(a, b, ea, eb, ta, tb, sm);
}


// ------------------------------------------------------------------------------------------------------------
// scopes: the thread numbering and the barrier a step is written against.  WgScope = one workgroup (barrier =
// __syncthreads); GridScope = all workgroups of a persistent launch (barrier = grid-wide, on an atomic counter).
// All workgroups of a grid launch are resident (the host sizes the grid by the CU count), so spinning on the
// generation counter cannot starve an unscheduled workgroup.  __threadfence() is an agent-scope fence: it writes this
// XCD's L2 back before the arrival and invalidates it after the release, which is what makes the other XCDs' plain
// stores visible (MI355X has one L2 per XCD).
// ------------------------------------------------------------------------------------------------------------
struct WgScope {
  long long* red;   // LDS, one per wave
  int* bc;          // LDS broadcast slot
  uint8_t* scratch = nullptr;   // LDS the caller does not need before its first cross_optimize (16-byte aligned)
  uint32_t scratch_bytes = 0;
  __device__ uint8_t* lds_scratch(uint32_t* bytes) const { *bytes = scratch_bytes; return scratch; }
  __device__ int tid() const { return threadIdx.x; }
  __device__ int nt() const { return blockDim.x; }
  __device__ int wave() const { return threadIdx.x >> 6; }
  __device__ int nwaves() const { return blockDim.x >> 6; }
  __device__ int blk() const { return 0; }
  __device__ int nblk() const { return 1; }
  __device__ void sync() { __syncthreads(); }
  __device__ int sync_or(int v) { return __syncthreads_or(v); }
  __device__ int bcast(int v) {   // value of thread 0 to everybody
    __syncthreads();
    if (threadIdx.x == 0) *bc = v;
    __syncthreads();
    const int r = *bc;
    __syncthreads();
    return r;
  }
  __device__ int ld(const int32_t* p) const { return *p; }
  __device__ long long sync_sum(long long v) {
    v = wave_sum_ll(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    long long t = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += red[w];
    __syncthreads();
    return t;
  }
};

struct GridScope {
  GridCtl* c;
  long long* red;      // LDS, one per wave
  unsigned long long* bc;   // LDS broadcast slots (two)
  unsigned gen;        // barriers passed so far (uniform over the grid)
  // a SUB-GRID: workgroups {b : b % stride == first} of the launch as a scope of their own (own GridCtl).  The speculative
  // rounds (k4_grid.hip) run one half-round per sub-grid: stride 8 puts a sub-grid on ONE XCD (workgroups go to the XCDs
  // round-robin), so its barriers stay inside one L2's neighbourhood.  stride 0 = the whole launch.
  int stride = 0, first = 0;
  __device__ uint8_t* lds_scratch(uint32_t* bytes) const { *bytes = 0; return nullptr; }
  __device__ int blk() const { return stride ? (int)blockIdx.x / stride : (int)blockIdx.x; }
  __device__ int nblk() const { return stride ? (int)gridDim.x / stride : (int)gridDim.x; }
  __device__ int tid() const { return blk() * blockDim.x + threadIdx.x; }
  __device__ int nt() const { return nblk() * blockDim.x; }
  __device__ int wave() const { return tid() >> 6; }
  __device__ int nwaves() const { return nt() >> 6; }
  // FENCED = false: no L2 write-back / invalidate.  For phases whose cross-workgroup data is read and written with
  // device-coherent (agent-scope relaxed atomic, `sc1`) accesses only: immutable data then stays in the L2s across the
  // barrier (tools/grid_barrier_bench.hip: 4.7 us instead of 15-20 us per barrier, no stale reads).
  // One barrier = one returning atomic on the arrival word + polling the generation word, and an OR over the grid
  // rides along for free: a workgroup adds 1 (+ 0x10000 if its flag is set) to `arrive`, the last arriver publishes
  // "any flag set" in bit 31 of the new generation, which is the word everybody polls.  (The OR used to be its own
  // atomic before and its own load after the barrier: two more device round trips of ~2 us each, per barrier.)
  template <bool FENCED = true>
  __device__ unsigned arrive_wait_(unsigned vflag = 0) {   // thread 0 of the workgroup; returns the OR of the flags
    const unsigned g = gen & 0x7FFFFFFFu;
    if (FENCED) __threadfence();
    const unsigned old = atomicAdd(&c->arrive, 1u + (vflag ? 0x10000u : 0u));
    unsigned any;
    if ((old & 0xFFFFu) == (unsigned)nblk() - 1u) {
      any = ((old >> 16) != 0u || vflag) ? 1u : 0u;
      // last arriver: the slots of parity (g+1) were read before their readers arrived here and are written again
      // only after this barrier opens
      __hip_atomic_store(&c->acc[(g + 1) & 1], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&c->arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (FENCED) __threadfence();
      __hip_atomic_store(&c->gen, ((g + 1) & 0x7FFFFFFFu) | (any << 31), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      unsigned x;
      while (((x = __hip_atomic_load(&c->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 0x7FFFFFFFu) == g) __builtin_amdgcn_s_sleep(1);
      any = x >> 31;
    }
    if (FENCED) __threadfence();
    return any;
  }
  __device__ void sync_light() {
    __syncthreads();
    if (threadIdx.x == 0) arrive_wait_<false>();
    gen++;
    __syncthreads();
  }
  // OR of `v` and sum of `x` over the grid in one light barrier (sum == nullptr: the OR alone, no reduction traffic)
  __device__ int sync_or_sum_light(int v, long long x, long long* sum) {
    if (sum) x = wave_sum_ll(x);
    v = __syncthreads_or(v);
    if (sum) {
      if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      if (sum) {
        long long t = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += red[w];
        // (returning: this workgroup's share is in the sum before it is counted as arrived)
        if (t) (void)__hip_atomic_fetch_add(&c->acc[gen & 1], (unsigned long long)t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      bc[1] = arrive_wait_<false>(v ? 1u : 0u);
      if (sum) bc[0] = __hip_atomic_load(&c->acc[gen & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gen++;
    __syncthreads();
    if (sum) *sum = (long long)bc[0];
    const int r = (int)bc[1];
    __syncthreads();
    return r;
  }
  __device__ int sync_or_light(int v) { return sync_or_sum_light(v, 0, nullptr); }
  __device__ void sync() {
    __syncthreads();
    if (threadIdx.x == 0) arrive_wait_();
    gen++;
    __syncthreads();
  }
  __device__ int sync_or(int v) {
    v = __syncthreads_or(v);
    if (threadIdx.x == 0) *bc = arrive_wait_(v ? 1u : 0u);
    gen++;
    __syncthreads();
    const int r = (int)*bc;
    __syncthreads();
    return r;
  }
  __device__ int bcast(int v) {   // value of thread 0 of workgroup 0 to everybody
    __syncthreads();
    if (threadIdx.x == 0) {
      if (blockIdx.x == 0) __hip_atomic_store(&c->slot, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      arrive_wait_();
      *bc = (unsigned long long)(unsigned)__hip_atomic_load(&c->slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gen++;
    __syncthreads();
    const int r = (int)(unsigned)*bc;
    __syncthreads();
    return r;
  }
  __device__ int ld(const int32_t* p) const { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
  __device__ long long sync_sum(long long v) {
    v = wave_sum_ll(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
      long long t = 0;
      for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += red[w];
      if (t) atomicAdd(&c->acc[gen & 1], (unsigned long long)t);
      arrive_wait_();
      *bc = __hip_atomic_load(&c->acc[gen & 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    gen++;
    __syncthreads();
    const long long r = (long long)*bc;
    __syncthreads();
    return r;
  }
};

// a wave's own stores, made by one lane, before loads of the same addresses by its other lanes
__device__ __forceinline__ void wave_mem_sync() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); }

}  // namespace
