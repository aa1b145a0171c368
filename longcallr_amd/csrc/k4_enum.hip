// k4_enum.hip — K4, enumeration branch: all 2^S restarts of cross_optimize for regions with S <= max_enum_snps
// (reference src/phase.rs:1097-1122 over cross_optimize :810-976).  Host control: k4_phase.hip (PhaseHost::run).
#include <climits>
#include "k4_dev.h"
#include "k4_kernels.h"

namespace {

// ---------------------------------------------------------------------------------------------
// Enumeration restarts, register-resident form.  A region's phase matrix is a few KB (rows x <= 31
// SNPs) while its 2^S restarts each sweep it ~7 times: one workgroup stages the matrix in LDS once,
// every wave64 copies "its lane's share" of the entries into VGPRs, and then runs complete restarts
// with the matrix in registers, delta / eta in wave-uniform bit masks and sigma as a bit vector in LDS.
// Only wave-level synchronisation inside a restart.  Same decisions as cross_optimize() above.
//   sigma step : lane <-> a run of whole rows in CSR order (~E/64 entries); two VGPRs per entry hold the
//                23-bit + signed 24-bit limbs of w[q] with the metadata in the bits v_mad_i32_i24 ignores
//   delta step : lane <-> a contiguous chunk of the CSC entries; per-SNP sums M[i] by LDS atomics
//                (integer, order-free), then lane i takes SNP i's four-way decision
//   objective  : sum over SNPs of the chosen branch's data term, which the last delta step already
//                holds (sigma does not change after it) -- no extra pass over the matrix.
// ---------------------------------------------------------------------------------------------
struct EnumTile { int32_t slot; uint32_t e0, ne; };   // restarts e0 .. e0+ne-1 of one region
__device__ __forceinline__ EnumTile enum_tile_of(const PhaseDev& P, const EnumSpan* __restrict__ spans, int n_spans, uint32_t per, bool winner) {
  EnumTile t;
  const uint32_t bid = blockIdx.x;
  if (winner) { t.slot = spans[bid].slot; t.e0 = 0; t.ne = 1; return t; }
  const int lane = threadIdx.x & 63;
  // level 1: 64 evenly spaced spans; level 2: the spans of the hit segment (n_spans <= 4096), else a plain search
  int lo = 0, hi = n_spans;   // answer in [lo, hi): last span with tile0 <= bid
  if (n_spans <= 4096) {
    const int step = (n_spans + 63) / 64;
    const int i1 = lane * step;
    const unsigned long long m1 = __ballot(i1 < n_spans && spans[min(i1, n_spans - 1)].tile0 <= bid);
    const int seg = __popcll(m1) - 1;          // spans[0].tile0 == 0 <= bid: at least one bit
    lo = seg * step; hi = min(n_spans, lo + step);
    const int i2 = lo + lane;
    const unsigned long long m2 = __ballot(i2 < hi && spans[min(i2, n_spans - 1)].tile0 <= bid);
    lo = lo + __popcll(m2) - 1;
  } else {
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (spans[mid].tile0 <= bid) lo = mid; else hi = mid; }
  }
  t.slot = spans[lo].slot;
  const uint32_t n = 1u << P.reg[t.slot].S;
  t.e0 = (bid - spans[lo].tile0) * per;
  t.ne = min(per, n - t.e0);
  return t;
}

// ---------------------------------------------------------------------------------------------
// `prob > largest_prob` over the restarts of one region (phase.rs:1113-1119), run by the tile that completes the region.
// The fixed-point objectives decide; among the restarts of MAXIMAL objective the reference keeps the first one unless a
// later one's f64 sum (cal_overall_probability, phase.rs:257-276: every phase entry's log10 term, fragment by fragment,
// in one running sum) is greater by rounding noise.  Every restart left its final state in st_words, so:
//   1. the restarts of maximal objective, in ascending order, ENUM_TCAP - 1 at a time beside the best so far;
//   2. a signature per configuration (a hash of the match bits of all entries in row order): equal signatures = the same
//      sequence of terms = the same f64 sum -- the usual case (restarts that reach one optimum, or its mirror image);
//   3. only if signatures differ: a lane per configuration adds its terms in the reference's order (f64, LUT of the
//      host's libm values), then thread 0 walks the list: strictly greater replaces;
//   4. the winner's state goes to the region's result slots (no re-run of the winning restart).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void enum_resolve(const PhaseDev& P, const RegionDev& rd, int slot, uint8_t* lds, const EnumLayout& L, uint32_t E,
                                             const long long* __restrict__ o, const unsigned long long* __restrict__ st, uint32_t n_jobs) {
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int R = rd.R, S = rd.S;
  const uint32_t nk = (uint32_t)(R + 63) / 64, sw = enum_state_words((uint32_t)R);
  const uint2* csr = (const uint2*)(lds + L.csr);
  const uint16_t* rp = (const uint16_t*)(lds + L.rp);
  const uint8_t* qrow = lds + L.qrow;
  const double* lut = (const double*)(lds + L.lut);
  uint32_t* t_e = (uint32_t*)(lds + L.res);                                   // [ENUM_TCAP] restart
  unsigned long long* t_sig = (unsigned long long*)(lds + L.res + 4 * ENUM_TCAP + 4 * ENUM_TCAP);   // [ENUM_TCAP] signature (8-byte aligned)
  double* t_sum = (double*)(lds + L.res + 16 * ENUM_TCAP);                    // [ENUM_TCAP] f64 objective
  __shared__ long long s_best[ENUM_WAVES];
  __shared__ uint32_t s_n, s_cursor, s_differ, s_win;
  __shared__ double s_winsum;
  __shared__ int s_have_sum;
  auto ld_obj = [&](uint32_t e) { return __hip_atomic_load(&o[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto ld_st = [&](uint32_t e, uint32_t w) { return __hip_atomic_load(&st[(size_t)e * sw + w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // ---- the maximal objective
  long long best = LLONG_MIN;
  for (uint32_t e = tid; e < n_jobs; e += nt) { const long long v = ld_obj(e); if (v > best) best = v; }
  for (int d = 32; d >= 1; d >>= 1) { const long long ob = __shfl_xor(best, d, 64); if (ob > best) best = ob; }
  if (lane == 0) s_best[wave] = best;
  if (tid == 0) { s_cursor = 0; s_win = 0xffffffffu; s_have_sum = 0; s_winsum = 0.0; }
  __syncthreads();
  for (int w = 0; w < ENUM_WAVES; w++) if (s_best[w] > best) best = s_best[w];
  // ---- chunks of the restarts of maximal objective, in ascending order; slot 0 of a chunk = the best so far
  for (;;) {
    const uint32_t cur = s_cursor;
    if (cur >= n_jobs) break;
    __syncthreads();
    // wave 0 collects the next <= TCAP - 1 (first chunk: TCAP) restarts with o[e] == best from `cur` on, 64 candidates per step
    if (wave == 0) {
      const bool first = s_win == 0xffffffffu;
      uint32_t n = first ? 0u : 1u, e0 = cur;
      if (!first && lane == 0) t_e[0] = s_win;
      while (e0 < n_jobs && n < ENUM_TCAP) {
        const uint32_t e = e0 + lane;
        const bool hit = e < n_jobs && ld_obj(e) == best;
        const unsigned long long m = __ballot(hit);
        const uint32_t rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        if (hit && n + rank < ENUM_TCAP) t_e[n + rank] = e;
        const uint32_t c = (uint32_t)__popcll(m);
        if (n + c > ENUM_TCAP) {   // the chunk is full inside this step: resume behind the last restart taken
          const uint32_t take = ENUM_TCAP - n;
          // position of the take-th set bit
          uint32_t last = 0;
          { const unsigned long long sel = __ballot(hit && rank == take - 1); last = (uint32_t)__ffsll((long long)sel) - 1u; }
          e0 = e0 + last + 1; n = ENUM_TCAP;
          break;
        }
        n += c; e0 += 64;
      }
      if (lane == 0) { s_n = n; s_cursor = e0 < n_jobs ? e0 : n_jobs; }
    }
    __syncthreads();
    const uint32_t n = s_n;
    // ---- signatures: a wave per configuration, a lane per row
    unsigned long long* sgw = (unsigned long long*)(lds + L.state + wave * L.stride);   // (the waves' sigma words: free now)
    for (uint32_t j = wave; j < n; j += ENUM_WAVES) {
      const uint32_t e = t_e[j];
      for (uint32_t k = lane; k < nk; k += 64) sgw[k] = ld_st(e, k);
      const unsigned long long m0 = ld_st(e, nk);
      const uint32_t dneg = (uint32_t)m0, eta0 = (uint32_t)(m0 >> 32), etap = (uint32_t)ld_st(e, nk + 1);
      wave_lds_sync();
      unsigned long long h = 0;
      for (int k = lane; k < R; k += 64) {
        const uint32_t sneg = (uint32_t)(sgw[k >> 6] >> (k & 63)) & 1u;
        uint32_t word = 0;
        int bit = 0;
        for (int x = rp[k]; x < rp[k + 1]; x++, bit++) {
          const uint32_t m = csr[x].x >> 24, i = m & 31u, pbit = (m >> 5) & 1u;
          const uint32_t match = ((eta0 >> i) & 1u) ? (pbit ^ sneg ^ ((dneg >> i) & 1u)) : (((etap >> i) & 1u) ? pbit : pbit ^ 1u);
          word |= match << bit;
        }
        h += mix64(((unsigned long long)(uint32_t)k << 32 | word) + 0x9E3779B97F4A7C15ULL);
      }
      h = (unsigned long long)wave_sum_ll((long long)h);
      if (lane == 0) t_sig[j] = h;
      wave_lds_sync();
    }
    __syncthreads();
    if (tid == 0) { uint32_t d = 0; for (uint32_t j = 1; j < n; j++) d |= t_sig[j] != t_sig[0] ? 1u : 0u; s_differ = d; }
    __syncthreads();
    if (s_differ) {
      // ---- f64 objectives (phase.rs:257-276): a lane per configuration, every lane walks the whole matrix in row order
      if (tid == 0) atomicAdd(&P.tie_ctr[TIE_BEST_F64], 1ull);
      for (uint32_t j = tid; j < n; j += nt) {
        const uint32_t e = t_e[j];
        if (j == 0 && s_have_sum) { t_sum[0] = s_winsum; continue; }
        const unsigned long long m0 = ld_st(e, nk);
        const uint32_t dneg = (uint32_t)m0, eta0 = (uint32_t)(m0 >> 32), etap = (uint32_t)ld_st(e, nk + 1);
        double acc = 0.0;
        unsigned long long wsg = 0;
        for (int k = 0; k < R; k++) {
          if ((k & 63) == 0) wsg = ld_st(e, (uint32_t)k >> 6);
          const uint32_t sneg = (uint32_t)(wsg >> (k & 63)) & 1u;
          for (int x = rp[k]; x < rp[k + 1]; x++) {
            const uint32_t m = csr[x].x >> 24, i = m & 31u, pbit = (m >> 5) & 1u;
            const uint32_t match = ((eta0 >> i) & 1u) ? (pbit ^ sneg ^ ((dneg >> i) & 1u)) : (((etap >> i) & 1u) ? pbit : pbit ^ 1u);
            acc += lut[(match ? 32u : 0u) + qrow[x]];
          }
        }
        t_sum[j] = acc;
      }
      __syncthreads();
    }
    if (tid == 0) {   // phase.rs:1117: strictly greater replaces (equal signatures: equal sums, the earlier one stays)
      uint32_t bj = 0;
      if (s_differ) for (uint32_t j = 1; j < n; j++) if (t_sum[j] > t_sum[bj]) bj = j;
      s_win = t_e[bj];
      if (s_differ) { s_winsum = t_sum[bj]; s_have_sum = 1; }
      // (equal signatures throughout: the best so far keeps its sum, known or not -- a later chunk that differs computes it)
    }
    __syncthreads();
  }
  // ---- the winner's state -> the region's result slots
  const uint32_t we = s_win;
  const unsigned long long m0 = ld_st(we, nk);
  const uint32_t dneg = (uint32_t)m0, eta0 = (uint32_t)(m0 >> 32), etap = (uint32_t)ld_st(we, nk + 1);
  for (int i = tid; i < S; i += nt) {
    P.st_delta[rd.snp_off + i] = (int8_t)(((dneg >> i) & 1u) ? -1 : 1);
    P.st_eta[rd.snp_off + i] = (int8_t)(((eta0 >> i) & 1u) ? 0 : (((etap >> i) & 1u) ? 1 : -1));
  }
  for (int row = tid; row < R; row += nt) P.st_sigma[rd.sig_off + row] = (int8_t)(((ld_st(we, (uint32_t)row >> 6) >> (row & 63)) & 1ull) ? -1 : 1);
  if (tid == 0) P.st_obj[slot] = best;
}

// tiles of restarts of regions whose per-lane share is <= CK entries (host decides).  Every restart stores its objective and
// its final state; the tile that completes its region decides the winner (enum_resolve).
template <int CK>
__global__ void __launch_bounds__(64 * ENUM_WAVES, 3)   // (three waves per SIMD: <= 168 VGPRs)
k4_enum_reg(PhaseDev P, const EnumSpan* __restrict__ spans, int32_t n_spans, uint32_t per, const int64_t* __restrict__ job_base,
            long long* __restrict__ job_obj, const int64_t* __restrict__ st_base, unsigned long long* __restrict__ st_words,
            uint32_t* __restrict__ tiles_done) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const EnumTile t = enum_tile_of(P, spans, n_spans, per, false);
  const RegionDev rd = P.reg[t.slot];
  const int R = rd.R, S = rd.S;
  const uint32_t E = (uint32_t)P.prow_ptr[rd.rp_off + R];
  const EnumLayout L = enum_layout(R, E);
  uint2* wl2 = (uint2*)lds;
  double* lut = (double*)(lds + L.lut);
  uint2* csr = (uint2*)(lds + L.csr);
  uint32_t* csc = (uint32_t*)(lds + L.csc);
  uint16_t* rp = (uint16_t*)(lds + L.rp); uint16_t* first_row = (uint16_t*)(lds + L.first_row);
  uint8_t* qrow = lds + L.qrow;
  const int tid = threadIdx.x, nt = blockDim.x;
  const uint32_t c = enum_chunk(E);
  // ---- stage the region (once per workgroup)
  if (tid < 32) {
    const long long w = tid < 31 ? P.lut.f1e[tid] - P.lut.fe[tid] : 0;
    wl2[tid] = make_uint2((uint32_t)w & 0x7fffffu, (uint32_t)(w >> 23) & 0xffffffu);   // w = hi * 2^23 + lo, hi signed (w < 0 for q <= 3)
  }
  if (tid < 64) lut[tid] = (tid & 31) < 31 ? (tid < 32 ? P.lut64->le[tid] : P.lut64->l1e[tid - 32]) : 0.0;
  const int32_t* g_rp = P.prow_ptr + rd.rp_off;
  for (int r = tid; r <= R; r += nt) rp[r] = (uint16_t)g_rp[r];
  __shared__ int32_t cps[33];
  if (tid <= S && tid < 33) cps[tid] = P.ccol_ptr[rd.cp_off + tid];
  __syncthreads();
  for (int l = tid; l <= 64; l += nt) {   // first row whose start offset is >= l * c
    const uint32_t target = (uint32_t)l * c;
    int lo = 0, hi = R;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (rp[mid] < target) lo = mid + 1; else hi = mid; }
    first_row[l] = (uint16_t)lo;
  }
  for (int e = tid; e < (int)E; e += nt) {
    const uint32_t cv = P.cval[rd.e_off + e];
    int col = 0;
    for (int i = 0; i < S; i++) col += (int)((uint32_t)e >= (uint32_t)cps[i + 1]);
    csc[e] = (uint32_t)P.crow[rd.e_off + e] | ((uint32_t)col << 16) | ((cv & 32u) << 16) | ((cv & 31u) << 22) | 0x80000000u;
  }
  __syncthreads();
  for (int r = tid; r < R; r += nt) {
    const int e0 = rp[r], e1 = rp[r + 1];
    if (e0 == e1) continue;
    const uint32_t owner = (uint32_t)e0 / c;
    const uint32_t roff = (uint32_t)r - first_row[owner];
    for (int e = e0; e < e1; e++) {
      const uint32_t v = P.pval[rd.e_off + e];
      const uint32_t meta = (uint32_t)P.pcol[rd.e_off + e] | (v & 32u) | (e + 1 == e1 ? 64u : 0u) | 128u;
      const uint2 w = wl2[v & 31u];
      csr[e] = make_uint2(w.x | (meta << 24), w.y | (roff << 24));
      qrow[e] = (uint8_t)(v & 31u);
    }
  }
  __syncthreads();
  // ---- per wave: my share of the matrix into registers
  const int lane = tid & 63, wave = tid >> 6;
  unsigned long long* sgb = (unsigned long long*)(lds + L.state + wave * L.stride);   // bit = 1: sigma == -1
  unsigned long long* Macc = sgb + (R + 63) / 64 + 1;
  const int r_a = first_row[lane];
  // CK > 0: the lane's entries live in VGPRs; CK == 0: any share size, entries are re-read from LDS
  constexpr int NREG = CK > 0 ? CK : 1;
  uint32_t re0[NREG], re1[NREG], ce[NREG];
  const int s0 = rp[r_a], s1 = rp[first_row[lane + 1]];
  const int c0 = min((int)E, lane * (int)c), c1 = min((int)E, (lane + 1) * (int)c);
  if (CK > 0) {
#pragma unroll
    for (int x = 0; x < NREG; x++) {
      const uint2 v = s0 + x < s1 ? csr[s0 + x] : make_uint2(0, 0);
      re0[x] = v.x; re1[x] = v.y;
      ce[x] = c0 + x < c1 ? csc[c0 + x] : 0;
    }
  }
  // wave-uniform trip counts of the two unrolled entry loops
  int n_sig, n_del;
  {
    int n = rp[first_row[lane + 1]] - rp[r_a];
    for (int d = 32; d >= 1; d >>= 1) n = max(n, __shfl_xor(n, d, 64));
    n_sig = __builtin_amdgcn_readfirstlane(n);
    n_del = (int)min(c, E);
  }
  const uint32_t smask = S >= 32 ? 0xffffffffu : ((1u << S) - 1u);
  // lane i < S owns SNP i
  long long cF = 0, cW = 0, cRef = 0, cVar = 0, het = 0;
  bool live = false; int eta_init = 0;
  if (lane < S) {
    const long long* sc = P.snp_const + 4ll * (rd.snp_off + lane);
    cF = sc[0]; cW = sc[1]; cRef = sc[2] + P.lut.f_homref; cVar = sc[3] + P.lut.f_homvar;
    const int n = P.ccol_ptr[rd.cp_off + lane + 1] - P.ccol_ptr[rd.cp_off + lane];
    het = P.lut.f_het0 - (long long)n * P.lut.f_log2;                     // phase.rs:136-144
    live = P.snp_fp[rd.snp_off + lane] != 0 && n > 0;
    eta_init = init_genotype(P.snp_vt[rd.snp_off + lane]);
  }
  const uint32_t e0_init = (uint32_t)__ballot(lane < S && eta_init == 0), ep_init = (uint32_t)__ballot(lane < S && eta_init == 1);
  const int nk = (R + 63) / 64;
  const int wsh = r_a & 63;
  const uint32_t ne = t.ne;
  unsigned long long* const st_reg = st_words + st_base[t.slot];
  const uint32_t stw = enum_state_words((uint32_t)R);
  uint32_t n_tie_f64 = 0, n_tie_flip = 0, n_dtie = 0, n_step = 0;   // census of this wave's restarts (lane 0 adds them up at the end)
  // the f64 scores of the rows in `tm` (fixed-point ties at rows with a het entry): q < qn of phase.rs:77-96, 845-858 -> flip
  auto tie_rows_f64 = [&](unsigned long long tm, const unsigned long long win, const uint32_t dneg, const uint32_t eta0, const uint32_t etap) -> unsigned long long {
    unsigned long long ft = 0;
    while (tm) {
      const int roff = __ffsll((long long)tm) - 1;
      tm &= tm - 1;
      const int row = r_a + roff;
      const uint32_t sneg = (uint32_t)(win >> roff) & 1u;
      double lp = 0.0, lm = 0.0;   // the running sums log_q2 (sigma = +1) and log_q3 (sigma = -1), entry order
      for (int x = rp[row]; x < rp[row + 1]; x++) {
        const uint32_t m = csr[x].x >> 24, i = m & 31u, pbit = (m >> 5) & 1u;
        const uint32_t isHet = (eta0 >> i) & 1u;
        const uint32_t mp = isHet ? (pbit ^ ((dneg >> i) & 1u)) : (((etap >> i) & 1u) ? pbit : pbit ^ 1u);
        const uint32_t mm = isHet ? mp ^ 1u : mp;
        const uint32_t q = qrow[x];
        lp += lut[(mp ? 32u : 0u) + q];
        lm += lut[(mm ? 32u : 0u) + q];
      }
      const double l1 = sneg ? lm : lp, l1n = sneg ? lp : lm;
      const double den = lp + lm;
      const double q = 1.0 - l1 / den, qn = 1.0 - l1n / den;
      if (q < qn) ft |= 1ull << roff;
    }
    return ft;
  };
  // one restart by this wave: its objective goes to job_obj[], its final state to st_words
  auto run_restart = [&](const uint32_t e_in) {
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)e_in);   // (wave-uniform: keep it in SGPRs)
    uint32_t dneg = e & smask;            // bit i: delta_i == -1 (doubling order of phase.rs:1099-1106)
    uint32_t eta0 = e0_init, etap = ep_init;   // eta_i == 0 / eta_i == +1
    // init_assignment (phase.rs:673-680): u01() < 0.5  <=>  top bit of the draw clear  -> sigma = -1
    const uint64_t ctr0 = (uint64_t)S + (uint64_t)R + (uint64_t)e * (uint64_t)R;
    for (int k = 0; k <= nk; k++) {
      const int row = lane + 64 * k;
      const bool neg = row < R && (mix64(rd.seed + (ctr0 + row + 1) * 0x9E3779B97F4A7C15ULL) >> 63) == 0;
      const unsigned long long b = __ballot(neg);
      if (lane == 0) sgb[k] = b;
    }
    if (lane < 32) Macc[lane] = 0;
    wave_lds_sync();
    bool hg_inc = true, h_inc = true;
    int iters = 0;
    long long obj_i = 0;
    while (hg_inc | h_inc) {
      // ---- sigma step (phase.rs:824-862)
      {
        const unsigned long long w0 = sgb[r_a >> 6], w1 = sgb[(r_a >> 6) + 1];
        const unsigned long long win = wsh ? (w0 >> wsh) | (w1 << (64 - wsh)) : w0;
        int alo = 0, ahi = 0;
        uint32_t uacc = 0;
        unsigned long long fm = 0, tm = 0;
        auto sig_one = [&](uint32_t v0, uint32_t v1) {
          const uint32_t m = v0 >> 24, i = m & 31u, roff = v1 >> 24;
          const uint32_t sneg = (uint32_t)(win >> roff);
          const uint32_t use = (m >> 7) & (eta0 >> i) & 1u;                 // het sites only
          const uint32_t hit = ((m >> 5) ^ sneg ^ (dneg >> i)) & use;       // p == sigma * delta
          const uint32_t mis = hit ^ use;
          alo += __mul24((int)hit, (int)v0) - __mul24((int)mis, (int)v0);   // A - B of phase.rs:824-862
          ahi += __mul24((int)hit, (int)v1) - __mul24((int)mis, (int)v1);
          uacc |= use;
          const bool end = (m >> 6) & 1u;
          // sign of ahi * 2^23 + alo: fold alo's carry into ahi, the remainder is in [0, 2^23)
          const int top = ahi + (alo >> 23);
          if (end && top < 0) fm |= 1ull << roff;
          // A == B at a row with a het entry: the f64 scores decide (a row without one scores the same for both signs, term by term)
          if (end && uacc && top == 0 && (alo & 0x7fffff) == 0) tm |= 1ull << roff;
          alo = end ? 0 : alo; ahi = end ? 0 : ahi; uacc = end ? 0u : uacc;
        };
        if (CK > 0) {
#pragma unroll
          for (int x = 0; x < NREG; x++) {
            if (x >= n_sig) break;
            // opaque to the optimiser: otherwise every field extraction is hoisted out of the restart loop
            // into its own VGPR (x CK entries) and the kernel drops to one wave per SIMD
            asm volatile("" : "+v"(re0[x]), "+v"(re1[x]));
            sig_one(re0[x], re1[x]);
          }
        } else {
          for (int x0 = 0; x0 < n_sig; x0 += 4) {
            uint2 v[4];
#pragma unroll
            for (int u = 0; u < 4; u++) v[u] = s0 + x0 + u < s1 ? csr[s0 + x0 + u] : make_uint2(0, 0);
#pragma unroll
            for (int u = 0; u < 4; u++) sig_one(v[u].x, v[u].y);
          }
        }
        const bool any = __ballot(fm != 0) != 0;   // a strict improvement (A < B somewhere)
        if (__ballot(tm != 0)) {
          n_tie_f64 += (uint32_t)__popcll(tm);
          const unsigned long long ft = tie_rows_f64(tm, win, dneg, eta0, etap);
          n_tie_flip += (uint32_t)__popcll(ft);
          fm |= ft;
          if (!any && __ballot(ft != 0)) n_step++;   // only tie flips: "no improvement" (check_new_haplotag's sums are not formed)
        }
        if (fm) {
          atomicXor(&sgb[r_a >> 6], fm << wsh);
          if (wsh && (fm >> (64 - wsh))) atomicXor(&sgb[(r_a >> 6) + 1], fm >> (64 - wsh));
        }
        wave_lds_sync();
        if (!any) h_inc = false; else { h_inc = true; hg_inc = true; }
      }
      // ---- delta / eta step (phase.rs:872-959): a lane's chunk is CSC-ordered (SNP index non-decreasing)
      {
        constexpr int HB = 8;   // look-ups of one batch in flight, then its run-length flush
        int cur = -1; int alo = 0, ahi = 0;
        auto del_batch = [&](const uint32_t* v8) {
          uint32_t sw[HB]; uint2 wq[HB];
#pragma unroll
          for (int x = 0; x < HB; x++) {
            const uint32_t row = v8[x] & 0xffffu;
            sw[x] = ((const uint32_t*)sgb)[row >> 5];
            wq[x] = wl2[(v8[x] >> 22) & 31u];
          }
#pragma unroll
          for (int x = 0; x < HB; x++) {
            const uint32_t v = v8[x];
            const int i = (v >> 16) & 31;
            const uint32_t hit = ((v >> 21) ^ (sw[x] >> (v & 31u)) ^ (dneg >> i)) & (v >> 31);
            if ((v >> 31) && i != cur) {
              if (alo | ahi) atomicAdd(&Macc[cur], (unsigned long long)(((long long)ahi << 23) + alo));
              cur = i; alo = 0; ahi = 0;
            }
            alo += __mul24((int)hit, (int)wq[x].x);
            ahi += __mul24((int)hit, (int)wq[x].y);   // sign-extends the 24-bit hi limb
          }
        };
        if (CK > 0) {
#pragma unroll
          for (int h = 0; h < NREG; h += HB) {
            if (h >= n_del) break;
            uint32_t v8[HB];
#pragma unroll
            for (int x = 0; x < HB; x++) { asm volatile("" : "+v"(ce[h + x < NREG ? h + x : 0])); v8[x] = ce[h + x < NREG ? h + x : 0]; }
            del_batch(v8);
          }
        } else {
          for (int h = 0; h < n_del; h += HB) {
            uint32_t v8[HB];
#pragma unroll
            for (int x = 0; x < HB; x++) v8[x] = c0 + h + x < c1 ? csc[c0 + h + x] : 0;
            del_batch(v8);
          }
        }
        if (alo | ahi) atomicAdd(&Macc[cur], (unsigned long long)(((long long)ahi << 23) + alo));
      }
      wave_lds_sync();
      bool changed = false, dtie = false;
      int d_new = (dneg >> lane) & 1u, h_new = ((eta0 >> lane) & 1u) ? 0 : (((etap >> lane) & 1u) ? 1 : -1);
      if (live) {
        const long long M = (long long)Macc[lane];
        Macc[lane] = 0;
        const long long N0 = cF + M + het, N1 = cF + cW - M + het;
        int ch = 0; long long nb = N0;                       // first maximum (phase.rs:908-921)
        if (N1 > nb) { ch = 1; nb = N1; }
        if (cRef > nb) { ch = 2; nb = cRef; }
        if (cVar > nb) { ch = 3; nb = cVar; }
        dtie = (int)(N0 == nb) + (int)(N1 == nb) + (int)(cRef == nb) + (int)(cVar == nb) > 1;   // a tie at the maximum: the first one is kept
        const long long ncur = h_new == 0 ? N0 : (h_new == 1 ? cRef : cVar);
        changed = nb > ncur;
        if (ch == 1) d_new ^= 1;
        h_new = ch <= 1 ? 0 : (ch == 2 ? 1 : -1);
        obj_i = ch <= 1 ? nb - het : (ch == 2 ? cRef - P.lut.f_homref : cVar - P.lut.f_homvar);
      }
      const uint32_t dneg_n = (uint32_t)__ballot(lane < S && d_new);
      const uint32_t eta0_n = (uint32_t)__ballot(lane < S && h_new == 0);
      const uint32_t etap_n = (uint32_t)__ballot(lane < S && h_new == 1);
      const bool any2 = __ballot(changed) != 0;
      n_dtie += (uint32_t)__popcll(__ballot(dtie));
      if (!any2 && (dneg_n != dneg || eta0_n != eta0 || etap_n != etap)) n_step++;   // only tie changes in this step
      dneg = dneg_n; eta0 = eta0_n; etap = etap_n;
      wave_lds_sync();
      if (!any2) hg_inc = false; else { hg_inc = true; h_inc = true; }
      if (++iters > 20) break;  // phase.rs:967-972
    }
    // objective (phase.rs:257-276) = sum over phase entries of fe + hit * w = sum_i (F_i + hits_i) over live SNPs
    const long long total = wave_sum_ll_dpp(obj_i);
    // (device-coherent stores: read by the region's last tile, possibly on another XCD)
    unsigned long long* stp = st_reg + (size_t)e * stw;
    for (int k = lane; k < nk; k += 64) __hip_atomic_store(&stp[k], sgb[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lane == 0) {
      __hip_atomic_store(&stp[nk], (unsigned long long)dneg | ((unsigned long long)eta0 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&stp[nk + 1], (unsigned long long)etap, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(&job_obj[job_base[t.slot] + e], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    wave_lds_sync();
  };
  for (uint32_t kk = wave; kk < ne; kk += ENUM_WAVES) run_restart(t.e0 + kk);
  if (lane == 0) {
    if (n_tie_f64) atomicAdd(&P.tie_ctr[TIE_SIGMA_F64], (unsigned long long)n_tie_f64);
    if (n_tie_flip) atomicAdd(&P.tie_ctr[TIE_SIGMA_FLIPS], (unsigned long long)n_tie_flip);
    if (n_dtie) atomicAdd(&P.tie_ctr[TIE_DELTA_UNRES], (unsigned long long)n_dtie);
    if (n_step) atomicAdd(&P.tie_ctr[TIE_STEP_UNRES], (unsigned long long)n_step);
  }
  // The tile that completes its region decides the winner (`prob > largest_prob`, phase.rs:1113-1119) from the objectives and
  // states all tiles stored device-coherently; every wave's stores are acknowledged before the barrier that lets thread 0
  // count the tile.  The matrix is still staged here.
  __shared__ uint32_t s_last;
  __syncthreads();
  const uint32_t n_jobs = 1u << S;
  if (tid == 0) s_last = atomicAdd(&tiles_done[t.slot], 1u) == (n_jobs + per - 1) / per - 1 ? 1u : 0u;
  __syncthreads();
  if (!s_last) return;
  enum_resolve(P, rd, t.slot, lds, L, E, job_obj + job_base[t.slot], st_reg, n_jobs);
}

// the same tiles for regions whose matrix does not fit the LDS budget: one restart at a time per workgroup
__global__ void __launch_bounds__(LCR_BLOCK)
k4_enum_big(PhaseDev P, const EnumSpan* __restrict__ spans, int32_t n_spans, uint32_t per, const int64_t* __restrict__ job_base,
            long long* __restrict__ job_obj, const uint32_t* __restrict__ win_e) {
  __shared__ long long red[LCR_BLOCK / 64];
  __shared__ long long wl[32];
  const EnumTile t = enum_tile_of(P, spans, n_spans, per, win_e != nullptr);
  const RegionDev rd = P.reg[t.slot];
  load_w(P, wl);
  int8_t* base = P.scratch + (size_t)blockIdx.x * P.scratch_stride;
  int8_t* sg = base; int8_t* dl = base + rd.R; int8_t* et = dl + rd.S;
  const int8_t* vt = P.snp_vt + rd.snp_off;
  const uint32_t ne = win_e ? 1u : t.ne;
  for (uint32_t k = 0; k < ne; k++) {
    const uint32_t e = win_e ? win_e[t.slot] : t.e0 + k;
    for (int i = threadIdx.x; i < rd.S; i += blockDim.x) { dl[i] = ((e >> i) & 1u) ? -1 : 1; et[i] = init_genotype(vt[i]); }
    const uint64_t ctr0 = (uint64_t)rd.S + (uint64_t)rd.R + (uint64_t)e * (uint64_t)rd.R;
    for (int row = threadIdx.x; row < rd.R; row += blockDim.x) sg[row] = u01(rd.seed, ctr0 + row) < 0.5 ? -1 : 1;
    __syncthreads();
    const long long obj = cross_optimize(P, rd, global_view(P, rd), sg, dl, et, false, true, red, wl);
    if (win_e) {
      for (int i = threadIdx.x; i < rd.S; i += blockDim.x) { P.st_delta[rd.snp_off + i] = dl[i]; P.st_eta[rd.snp_off + i] = et[i]; }
      for (int row = threadIdx.x; row < rd.R; row += blockDim.x) P.st_sigma[rd.sig_off + row] = sg[row];
      if (threadIdx.x == 0) P.st_obj[t.slot] = obj;
    } else if (threadIdx.x == 0) job_obj[job_base[t.slot] + e] = obj;
    __syncthreads();
  }
}

// winner of each enumeration region of the global-memory fallback class: first maximum over e (`prob > largest_prob`,
// phase.rs:1113-1119); two restarts of maximal objective are not compared by their f64 sums here (counted: TIE_BEST_UNRES)
__global__ void __launch_bounds__(64) k4_enum_pick(const EnumSpan* __restrict__ spans, int32_t n, const RegionDev* __restrict__ reg,
                                                    const int64_t* __restrict__ job_base, const long long* __restrict__ job_obj,
                                                    uint32_t* __restrict__ win_e, unsigned long long* __restrict__ tie_ctr) {
  const int k = blockIdx.x;
  if (k >= n) return;
  const int slot = spans[k].slot;
  const uint32_t nj = 1u << reg[slot].S;
  const long long* o = job_obj + job_base[slot];
  long long best = LLONG_MIN; uint32_t be = 0xffffffffu;
  for (uint32_t e = threadIdx.x; e < nj; e += 64) { const long long v = o[e]; if (v > best) { best = v; be = e; } }
  for (int d = 32; d >= 1; d >>= 1) {
    const long long ob = __shfl_xor(best, d, 64); const uint32_t oe = __shfl_xor(be, d, 64);
    if (ob > best || (ob == best && oe < be)) { best = ob; be = oe; }
  }
  int at_max = 0;
  for (uint32_t e = threadIdx.x; e < nj; e += 64) at_max += o[e] == best ? 1 : 0;
  for (int d = 32; d >= 1; d >>= 1) at_max += __shfl_xor(at_max, d, 64);
  if (threadIdx.x == 0) { win_e[slot] = be; if (at_max > 1) atomicAdd(&tie_ctr[TIE_BEST_UNRES], 1ull); }
}
}  // namespace

void launch_k4_enum_reg(int ck, unsigned n_blocks, size_t dyn_lds, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, int32_t n_spans,
                        uint32_t per, const int64_t* job_base, long long* job_obj, const int64_t* st_base, unsigned long long* st_words, uint32_t* done) {
  const dim3 blk(64 * ENUM_WAVES);
  if (ck == 32) hipLaunchKernelGGL(k4_enum_reg<32>, dim3(n_blocks), blk, dyn_lds, s, P, spans, n_spans, per, job_base, job_obj, st_base, st_words, done);
  else hipLaunchKernelGGL(k4_enum_reg<0>, dim3(n_blocks), blk, dyn_lds, s, P, spans, n_spans, per, job_base, job_obj, st_base, st_words, done);
}
void launch_k4_enum_big(unsigned n_blocks, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, int32_t n_spans, uint32_t per,
                        const int64_t* job_base, long long* job_obj, const uint32_t* win_e) {
  hipLaunchKernelGGL(k4_enum_big, dim3(n_blocks), dim3(LCR_BLOCK), 0, s, P, spans, n_spans, per, job_base, job_obj, win_e);
}
void launch_k4_enum_pick(int32_t n, hipStream_t s, const EnumSpan* spans, const RegionDev* reg, const int64_t* job_base, const long long* job_obj,
                         uint32_t* win_e, unsigned long long* tie_ctr) {
  hipLaunchKernelGGL(k4_enum_pick, dim3((unsigned)n), dim3(64), 0, s, spans, n, reg, job_base, job_obj, win_e, tie_ctr);
}
