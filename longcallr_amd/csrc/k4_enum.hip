// k4_enum.hip — K4, enumeration branch: all 2^S restarts of cross_optimize for regions with S <= max_enum_snps
// (reference src/phase.rs:1097-1122 over cross_optimize :810-976).  Host control: k4_phase.hip (PhaseHost::run).
#include <climits>
#include <type_traits>
#include "k4_dev.h"
#include "k4_kernels.h"

#ifndef ENUM_OCC
#define ENUM_OCC 3
#endif
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
// k4_enum_resolve: `prob > largest_prob` over the restarts of one region (phase.rs:1113-1119) -- a workgroup per region, launched
// per kernel class behind the class's restarts on its queue.
// The fixed-point objectives decide; among the restarts of MAXIMAL objective the reference keeps the first one unless a
// later one's f64 sum (cal_overall_probability, phase.rs:257-276: every phase entry's log10 term, fragment by fragment,
// in ONE running sum) is greater by rounding noise.  Every restart left its final state and a SIGNATURE in st_words -- a
// hash of the match bits of all entries (k4_enum_reg): equal signatures = the same sequence of terms = the same f64 sum.
//   0. all restarts of maximal objective carry one signature (restarts that reached one optimum with the same sigma
//      everywhere, or its mirror image): the first of them wins, nothing is summed;
//   1. else the first one is the REFERENCE configuration: its terms (f64, table of the host's libm values) are formed in
//      parallel and ONE lane adds them in the reference's order, keeping the running sum behind every entry, pse[0 .. E);
//   2. the EVENTS of that chain: entries where the sum crosses a power of two, or where s + t lies exactly half way between two
//      doubles (TwoSum's error term) -- the only places where two running sums a few ulps apart that take the same terms do
//      not keep their distance (inside one binade fl(s + t) rounds t on its own);
//   3. the others, a lane each, ENUM_TCAP at a time in ascending order.  A configuration of the reference's class -- same
//      eta, delta mirrored on whole GROUPS of het sites (components of "share a row") -- differs from it in the sigma of a few
//      rows only (rows whose two orientations score alike: their sigma is whatever init_assignment drew).  Such a row is
//      re-added alone from the reference's running sum; once a row arrives elsewhere the distance dl is carried from event
//      to event.  Another class, a delta that differs inside a group (macroscopic row differences), or a distance beyond
//      ~1000 ulps: the lane adds all of its terms (full_sum);
//   4. strictly-greater-replaces over the list in order (phase.rs:1117) as a reduction (largest sum, earliest of equals);
//      the winner's state goes to the region's result slots (no re-run of the winning restart).
// ---------------------------------------------------------------------------------------------
// measurement build (-DENUM_PROF): the slowest workgroup's time (10 ns units) up to a point of k4_enum_resolve -> tie census slot i
#ifdef ENUM_PROF
#define ENUM_PT(i) do { if (threadIdx.x == 0) atomicMax(&P.tie_ctr[i], (unsigned long long)((long long)wall_clock64() - prof_t0)); } while (0)
#else
#define ENUM_PT(i) do { } while (0)
#endif
__global__ void __launch_bounds__(64 * ENUM_WAVES)
k4_enum_resolve(PhaseDev P, const EnumSpan* __restrict__ spans, const int64_t* __restrict__ job_base, const long long* __restrict__ job_obj,
                const int64_t* __restrict__ st_base, const unsigned long long* __restrict__ st_words) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const long long prof_t0 = (long long)wall_clock64(); (void)prof_t0;
  const int slot = spans[blockIdx.x].slot;
  const RegionDev rd = P.reg[slot];
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int R = rd.R, S = rd.S;
  const uint32_t E = (uint32_t)P.prow_ptr[rd.rp_off + R];
  const uint32_t n_jobs = 1u << S;
  const long long* const o = job_obj + job_base[slot];
  const unsigned long long* const st = st_words + st_base[slot];
  const uint32_t nk = (uint32_t)(R + 63) / 64, sw = enum_state_words((uint32_t)R);
  const ResolveLayout L = resolve_layout((uint32_t)R, E, (uint32_t)S);
  uint16_t* rp = (uint16_t*)(lds + L.rp);
  uint16_t* ent16 = (uint16_t*)(lds + L.ent16);
  double* lut = (double*)(lds + L.lut);
  double* pse = (double*)(lds + L.pse);                                                 // [E] running sum of the reference configuration behind every entry
  unsigned long long* sg_ref = (unsigned long long*)(lds + L.sg_ref);                  // [nk] its sigma words
  unsigned long long* hetw = (unsigned long long*)(lds + L.hetw);                      // [nk] rows with an entry at one of its het sites
  unsigned long long* repmask = (unsigned long long*)(lds + L.repmask);                // [S][nk] rows whose first het site (of the reference's eta) is SNP i
  uint32_t* rowsnps = (uint32_t*)(lds + L.rowsnps);                                    // [R] the SNPs of a row's entries as a mask
  __shared__ uint32_t s_adj[32], s_grp[32];                                             // het sites a site shares a row with; the same closed under "shares a row" (its group)
  uint32_t* t_e = (uint32_t*)(lds + L.res);                                             // [ENUM_TCAP] restart
  double* t_sum = (double*)(lds + L.res + 8 * ENUM_TCAP);                               // [ENUM_TCAP] f64 objective
  constexpr uint32_t EVCAP = 96;
  double* ev_t = (double*)(lds + L.res + 16 * ENUM_TCAP);                              // [EVCAP] events of the reference's chain: its term there ...
  uint16_t* ev_x = (uint16_t*)(lds + L.res + 16 * ENUM_TCAP + 8 * EVCAP);              // ... and the entry
  __shared__ uint32_t s_nev;
  __shared__ long long s_best[ENUM_WAVES];
  __shared__ uint32_t s_wcnt[ENUM_WAVES], s_dif[ENUM_WAVES], s_ntied;
  const uint32_t TLCAP = resolve_tlcap((uint32_t)S);
  uint16_t* tl = (uint16_t*)(lds + L.res + 16 * ENUM_TCAP + 96 * 10 + 16);   // [TLCAP] the restarts of maximal objective
  __shared__ uint32_t s_win, s_wj[ENUM_WAVES], s_first;   // s_first: the first restart of maximal objective (32 bits: tl[] holds 16)
  __shared__ double s_winsum, s_wsum[ENUM_WAVES];
  __shared__ int s_flat;
  // (plain cached loads: the restarts' kernel has completed -- k4_enum_resolve is a launch of its own behind it)
  auto ld_obj = [&](uint32_t e) { return o[e]; };
  auto ld_st = [&](uint32_t e, uint32_t w) { return st[(size_t)e * sw + w]; };
  // ---- stage the region: row pointers, the row-order entries (SNP | allele | q | last-of-row), the table of libm values
  if (tid < 64) lut[tid] = (tid & 31) < 31 ? (tid < 32 ? P.lut64->le[tid] : P.lut64->l1e[tid - 32]) : 0.0;
  { const int32_t* g_rp = P.prow_ptr + rd.rp_off; for (int r = tid; r <= R; r += nt) rp[r] = (uint16_t)g_rp[r]; }
  for (int e = tid; e < (int)E; e += nt) {   // (a thread per entry; the last-of-row marks follow from the row pointers)
    const uint32_t v = P.pval[rd.e_off + e];
    ent16[e] = (uint16_t)(((uint32_t)P.pcol[rd.e_off + e] & 31u) | (v & 32u) | ((v & 31u) << 6));
  }
  __syncthreads();
  for (int r = tid; r < R; r += nt) if (rp[r + 1] > rp[r]) ent16[rp[r + 1] - 1] |= 0x800u;
  ENUM_PT(0);
  // ---- the maximal objective and the restarts that have it, in ascending order (tl[0 .. n_tied)); do all of them carry the
  // first one's signature?  (the objectives of the first 1 024 restarts stay in registers between the two sweeps)
  constexpr int OV = 4;
  long long ov[OV];
  long long best = LLONG_MIN;
#pragma unroll
  for (int k = 0; k < OV; k++) { const uint32_t e = (uint32_t)tid + (uint32_t)k * 256u; ov[k] = e < n_jobs ? ld_obj(e) : LLONG_MIN; }
#pragma unroll
  for (int k = 0; k < OV; k++) if (ov[k] > best) best = ov[k];
  for (uint32_t e = tid + OV * 256u; e < n_jobs; e += nt) { const long long v = ld_obj(e); if (v > best) best = v; }
  for (int d = 32; d >= 1; d >>= 1) { const long long ob = __shfl_xor(best, d, 64); if (ob > best) best = ob; }
  if (lane == 0) s_best[wave] = best;
  if (tid == 0) { s_ntied = 0; s_first = 0xffffffffu; }
  __syncthreads();
  for (int w = 0; w < ENUM_WAVES; w++) if (s_best[w] > best) best = s_best[w];
  for (uint32_t p0 = 0; p0 < n_jobs; p0 += 256u) {   // a pass of 256 restarts, thread = restart: ranks by ballot, wave offsets through LDS
    const uint32_t e = p0 + (uint32_t)tid;
    const long long v = p0 < OV * 256u ? ov[p0 >> 8] : (e < n_jobs ? ld_obj(e) : LLONG_MIN);
    const bool hit = e < n_jobs && v == best;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wcnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = s_ntied, tot = 0;
    for (int w = 0; w < ENUM_WAVES; w++) { if (w < wave) off += s_wcnt[w]; tot += s_wcnt[w]; }
    const uint32_t at = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (hit && at < TLCAP) tl[at] = (uint16_t)e;
    if (hit && at == 0) s_first = e;   // (the list is in ascending order: slot 0 is the first maximum; regions of more than 2^16 restarts keep all 32 bits here)
    __syncthreads();
    if (tid == 0) s_ntied += tot;
    __syncthreads();
  }
  const uint32_t n_tied_all = s_ntied, n_tied = min(n_tied_all, TLCAP);
  const uint32_t first = s_first;
  const unsigned long long sig0 = ld_st(first, nk + 2);
  uint32_t dif = 0;
  for (uint32_t j = tid; j < n_tied; j += nt) if (ld_st(tl[j], nk + 2) != sig0) dif = 1;
  dif = __ballot(dif != 0) ? 1u : 0u;
  if (lane == 0) s_dif[wave] = dif;
  if (n_tied_all > TLCAP || n_jobs > 65536u) {   // (more configurations of maximal objective than the list holds: first maximum wins, counted)
    if (tid == 0) TIE_COUNT(P.tie_ctr, TIE_BEST_UNRES, 1ull);
    dif = 0;
    if (lane == 0) s_dif[wave] = 0;
  }
  if (tid == 0) { s_win = first; s_winsum = 0.0; s_flat = 1; }
  __syncthreads();
  for (int w = 0; w < ENUM_WAVES; w++) dif |= s_dif[w];
  ENUM_PT(1);
#ifdef ENUM_ABL_NOSUM
  dif = 0;
#endif
  if (dif && P.tie_arith < 1) { if (tid == 0) TIE_COUNT(P.tie_ctr, TIE_BEST_UNRES, 1ull); dif = 0; }
#ifdef ENUM_PROF
  long long prof_t5 = 0;
  __shared__ uint32_t s_tloc[ENUM_WAVES], s_tfull[ENUM_WAVES];
  if (tid < ENUM_WAVES) { s_tloc[tid] = 0; s_tfull[tid] = 0; }
#endif
  if (dif) {
    if (tid == 0) TIE_COUNT(P.tie_ctr, TIE_BEST_F64, 1ull);
    // ---- the reference configuration
    const unsigned long long rm0 = ld_st(first, nk);
    const uint32_t r_dneg = (uint32_t)rm0, r_eta0 = (uint32_t)(rm0 >> 32), r_etap = (uint32_t)ld_st(first, nk + 1);
    { int empty = 0; for (int k = tid; k < R; k += nt) empty |= rp[k] == rp[k + 1] ? 1 : 0; if (empty) s_flat = 0; }   // (rows without an entry: min_linkers = 0)
    for (uint32_t x = E + tid; x < ((E + 7u) & ~7u); x += nt) ent16[x] = 0;   // padding of the last 16-byte read
    for (uint32_t k = tid; k < nk; k += nt) sg_ref[k] = ld_st(first, k);
    // rows by their first het site, and the groups of het sites that rows tie together: a configuration whose delta is mirrored
    // on whole groups (independent groups of SNPs settle independently) differs from the reference exactly in the rows whose
    // sigma is not mirrored along with their group
    const bool rows_ok = true;
    if (tid < 32) s_adj[tid] = 0;
    __syncthreads();
    for (uint32_t k0 = wave * 64; k0 < (uint32_t)R; k0 += 64 * ENUM_WAVES) {   // a lane per row
      const uint32_t k = k0 + lane;
      uint32_t ms = 0;
      if (k < (uint32_t)R) for (int x = rp[k]; x < rp[k + 1]; x++) ms |= 1u << (ent16[x] & 31u);
      if (k < (uint32_t)R) rowsnps[k] = ms;
      const uint32_t hs = ms & r_eta0;
      const unsigned long long m = __ballot(hs != 0);
      if (lane == 0) hetw[k0 >> 6] = m;
      const int rep = hs ? __ffs((int)hs) - 1 : -1;
      if (hs & (hs - 1)) for (uint32_t rest = hs; rest; rest &= rest - 1) atomicOr(&s_adj[__ffs((int)rest) - 1], hs);   // (a row with two or more het sites ties them together)
      if (rows_ok) for (int i = 0; i < S; i++) { const unsigned long long mi = __ballot(rep == i); if (lane == 0) repmask[(uint32_t)i * nk + (k0 >> 6)] = mi; }
    }
    __syncthreads();
    if (wave == 0) {   // groups = connected components over "shares a row" (<= 32 sites): lane i holds site i's set, five doubling rounds
      uint32_t g = lane < 32 && ((r_eta0 >> lane) & 1u) ? (1u << lane) | s_adj[lane & 31] : 0u;
      for (int round = 0; round < 5; round++)
        for (int j2 = 0; j2 < 32; j2++) { const uint32_t v = (uint32_t)__shfl((int)g, j2, 64); if ((g >> j2) & 1u) g |= v; }
      if (lane < 32) s_grp[lane] = g;
    }
    __syncthreads();
    ENUM_PT(2);
    const bool prefix_ok = true;
    // a configuration's terms added in the reference's order; ps_out != nullptr: the running sum before every row is kept
    auto full_sum = [&](const uint32_t e, const uint32_t dneg, const uint32_t eta0, const uint32_t etap) -> double {
      // match = [p == x] as three bit operations per entry (x depends on sigma only at het sites):
      //   het site: p == sigma * delta <=> pbit ^ sneg ^ dneg_i = 1;  hom site: p == eta <=> pbit ^ [eta_i == -1] = 1
      //   => match = pbit ^ c_i ^ (sneg & het_i),  c = (dneg & eta0) | (~etap & ~eta0)
      const uint32_t cm = (dneg & eta0) | (~etap & ~eta0);
      double acc = 0.0;
      if (s_flat) {
        // the entries as one flat run (every row has an entry: the last-entry flags count the rows), eight per 16-byte read;
        // the entry words are the same for every lane: decoded on the scalar unit.  The next read and this batch's eight
        // table values are in flight while the previous adds run.
        unsigned long long wsg = ld_st(e, 0), wnext = 0;
        int k = 0;
        uint32_t sneg = 0u - ((uint32_t)wsg & 1u);   // all ones: sigma == -1
        const uint32_t E8 = (E + 7u) & ~7u;
        uint4 nxt = E8 ? *(const uint4*)ent16 : make_uint4(0, 0, 0, 0);
        for (uint32_t x = 0; x < E8; x += 8) {
          const uint4 raw = nxt;
          if (x + 8 < E8) nxt = *(const uint4*)(ent16 + x + 8);
          const uint32_t r[4] = {(uint32_t)__builtin_amdgcn_readfirstlane((int)raw.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.y),
                                 (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.z), (uint32_t)__builtin_amdgcn_readfirstlane((int)raw.w)};
          double tv[8];
#pragma unroll
          for (int u = 0; u < 8; u++) {
            const uint32_t v = (r[u >> 1] >> ((u & 1) * 16)) & 0xffffu;
            const uint32_t i = v & 31u, pbit = (v >> 5) & 1u, q = (v >> 6) & 31u;   // (scalar)
            const uint32_t match = (pbit ^ (cm >> i) ^ ((sneg & eta0) >> i)) & 1u;
            tv[u] = lut[(match << 5) + q];
            if (v & 0x800u) {   // row end (uniform): the next row's sigma
              k++;
              if ((k & 63) == 32 && (uint32_t)(k >> 6) + 1 < nk) wnext = ld_st(e, (uint32_t)(k >> 6) + 1);
              if ((k & 63) == 0) wsg = wnext;
              sneg = 0u - ((uint32_t)(wsg >> (k & 63)) & 1u);
            }
          }
#pragma unroll
          for (int u = 0; u < 8; u++) if (x + u < E) acc += tv[u];
        }
      } else {
        unsigned long long wsg = 0;
        for (int k = 0; k < R; k++) {
          if ((k & 63) == 0) wsg = ld_st(e, (uint32_t)k >> 6);
          const uint32_t sneg = 0u - ((uint32_t)(wsg >> (k & 63)) & 1u);
          for (int x = rp[k]; x < rp[k + 1]; x++) {
            const uint32_t m = ent16[x], i = m & 31u, pbit = (m >> 5) & 1u;
            const uint32_t match = (pbit ^ (cm >> i) ^ ((sneg & eta0) >> i)) & 1u;
            acc += lut[(match << 5) + ((m >> 6) & 31u)];
          }
        }
      }
      return acc;
    };
    // the reference configuration's terms, a thread per entry (row by row: a lane per row) ...
    {
      const uint32_t cm = (r_dneg & r_eta0) | (~r_etap & ~r_eta0);
      for (int k = tid; k < R; k += nt) {
        const uint32_t sneg = 0u - ((uint32_t)(sg_ref[k >> 6] >> (k & 63)) & 1u);
        for (int x = rp[k]; x < rp[k + 1]; x++) {
          const uint32_t m = ent16[x], i = m & 31u, pbit = (m >> 5) & 1u;
          const uint32_t match = (pbit ^ (cm >> i) ^ ((sneg & r_eta0) >> i)) & 1u;
          pse[x] = lut[(match << 5) + ((m >> 6) & 31u)];
        }
      }
      for (uint32_t x = E + tid; x < ((E + 7u) & ~7u); x += nt) pse[x] = 0.0;   // (padding of the last batch)
    }
    __syncthreads();
    ENUM_PT(3);
    // ... and their running sum by ONE lane: the adds are the reference's chain (phase.rs:257-276), eight terms per round trip
    // (the next batch's terms are on their way while this one's adds run)
    if (tid == 0) {
      double acc = 0.0;
      const uint32_t E8 = (E + 7u) & ~7u;
      double t[8], n8[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = E8 ? pse[u] : 0.0;
      for (uint32_t x = 0; x < E8; x += 8) {
        const uint32_t xn = x + 8 < E8 ? x + 8 : x;
#pragma unroll
        for (int u = 0; u < 8; u++) n8[u] = pse[xn + u];
#pragma unroll
        for (int u = 0; u < 8; u++) { acc += t[u]; t[u] = acc; }
#pragma unroll
        for (int u = 0; u < 8; u++) pse[x + u] = t[u];
#pragma unroll
        for (int u = 0; u < 8; u++) t[u] = n8[u];
      }
    }
    __syncthreads();
    ENUM_PT(4);
    // EVENTS of the reference's chain: an addition s + t that stays inside s's binade rounds t on its own (s is a multiple of the
    // binade's ulp), so two running sums a few ulps apart that take the same terms KEEP their distance -- except where the sum
    // crosses a power of two (the ulp doubles) or s + t lies exactly half way between two doubles (ties-to-even looks at s).
    // Those entries (a dozen or two), with the reference's term, in ascending order; a thread per row finds them.
    if (tid == 0) s_nev = 0;
    __syncthreads();
    {
      const uint32_t cm = (r_dneg & r_eta0) | (~r_etap & ~r_eta0);
      for (int k = tid; k < R; k += nt) {
        const uint32_t sneg = 0u - ((uint32_t)(sg_ref[k >> 6] >> (k & 63)) & 1u);
        for (int x = rp[k]; x < rp[k + 1]; x++) {
          const uint32_t m = ent16[x], i = m & 31u, pbit = (m >> 5) & 1u;
          const uint32_t match = (pbit ^ (cm >> i) ^ ((sneg & r_eta0) >> i)) & 1u;
          const double t = lut[(match << 5) + ((m >> 6) & 31u)];
          const double a = x ? pse[x - 1] : 0.0, c = pse[x];
          // crossing zone: the exponent of |a| less 4096 ulps differs from that of |c| plus 4096 ulps
          const unsigned long long ba = (unsigned long long)__double_as_longlong(fabs(a)), bc = (unsigned long long)__double_as_longlong(fabs(c));
          const bool cross = ((ba > 4096ull ? ba - 4096ull : 0ull) >> 52) != ((bc + 4096ull) >> 52);
          // exact tie: the error of fl(a + t) (TwoSum) is half an ulp of the result
          const double bb = c - a, err = (a - (c - bb)) + (t - bb);
          const double half_ulp = __longlong_as_double((long long)((((bc >> 52) & 0x7ffull) - 53ull) << 52));
          const bool tie = ((bc >> 52) & 0x7ffull) > 53ull && fabs(err) == half_ulp;
          if (cross || tie) { const uint32_t at = atomicAdd(&s_nev, 1u); if (at < EVCAP) { ev_x[at] = (uint16_t)x; ev_t[at] = t; } }
        }
      }
    }
    __syncthreads();
    {   // ascending order: every event finds its rank (an entry has at most one event)
      const uint32_t ne = min(s_nev, EVCAP);
      uint16_t xx = 0; double tt = 0.0; uint32_t rank = 0;
      if ((uint32_t)tid < ne) { xx = ev_x[tid]; tt = ev_t[tid]; for (uint32_t j = 0; j < ne; j++) rank += ev_x[j] < xx ? 1u : 0u; }
      __syncthreads();
      if ((uint32_t)tid < ne) { ev_x[rank] = xx; ev_t[rank] = tt; }
    }
    __syncthreads();
    const bool events_ok = s_nev <= EVCAP;
#ifdef ENUM_VERIFY
    if (tid == 0) { atomicMax(&P.tie_ctr[3], (unsigned long long)s_nev); if (!events_ok) atomicAdd(&P.tie_ctr[5], 1ull); atomicAdd(&P.tie_ctr[1], (unsigned long long)s_nev); }
#endif
    const uint32_t n_ev = min(s_nev, EVCAP);
    ENUM_PT(5);
#ifdef ENUM_PROF
    prof_t5 = (long long)wall_clock64();
#endif
    // ---- chunks of the restarts of maximal objective, in ascending order
    for (uint32_t cur = 0; cur < n_tied; cur += ENUM_TCAP) {
      __syncthreads();
      const uint32_t n = min(ENUM_TCAP, n_tied - cur);
      if ((uint32_t)tid < n) t_e[tid] = tl[cur + tid];
      __syncthreads();
#ifdef ENUM_PROF
      const long long pf0 = (long long)wall_clock64();
#endif
      // ---- a lane per configuration: rows that differ from the reference's, each re-added from the reference's running sum
      bool fallback = false;
      uint32_t my_e = 0, dneg = 0, eta0 = 0, etap = 0;
      if ((uint32_t)tid < n) {
        my_e = t_e[tid];
        const unsigned long long m0 = ld_st(my_e, nk);
        dneg = (uint32_t)m0; eta0 = (uint32_t)(m0 >> 32); etap = (uint32_t)ld_st(my_e, nk + 1);
        const uint32_t dd = (dneg ^ r_dneg) & r_eta0;
        // the same eta: an entry's term differs from the reference's iff it is a het entry with (delta_i differs) != (sigma_k differs)
        const bool same_class = prefix_ok && events_ok && eta0 == r_eta0 && etap == r_etap;
        const uint32_t cm = (dneg & eta0) | (~etap & ~eta0);
        // The walk over the rows whose terms differ from the reference's, in ascending order.  In step with the reference
        // (dl == 0) a row is re-added alone from the reference's running sum; once a row arrives elsewhere (dl != 0: this
        // configuration's sum = the reference's + dl) the distance is carried from event to event of the reference's chain,
        // each of them and each further row of its own added explicitly from (reference's sum before it) + dl.
        double dl = 0.0;
        uint32_t evp = 0;
        auto row_check = [&](const int k, const uint32_t sneg) {
          const int xa = rp[k], xb = rp[k + 1];
          if (dl != 0.0)
            for (; evp < n_ev && (int)ev_x[evp] < xa; evp++) {   // the reference's events on the way (its own terms there)
              const int x = ev_x[evp];
              const double s1 = ((x ? pse[x - 1] : 0.0) + dl) + ev_t[evp];
              dl = s1 - pse[x];
            }
          double s2 = (xa ? pse[xa - 1] : 0.0) + dl;
          for (int x = xa; x < xb; x += 8) {   // eight entries per round trip: their words, their table values, then the adds in order
            uint32_t m8[8]; double t8[8];
#pragma unroll
            for (int u = 0; u < 8; u++) m8[u] = ent16[min(x + u, xb - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const uint32_t i = m8[u] & 31u, pbit = (m8[u] >> 5) & 1u;
              const uint32_t match = (pbit ^ (cm >> i) ^ ((sneg & eta0) >> i)) & 1u;
              t8[u] = lut[(match << 5) + ((m8[u] >> 6) & 31u)];
            }
#pragma unroll
            for (int u = 0; u < 8; u++) if (x + u < xb) s2 += t8[u];
          }
          dl = s2 - pse[xb - 1];
          while (evp < n_ev && (int)ev_x[evp] < xb) evp++;   // (events inside this row were added with it)
          if (fabs(dl) > fabs(s2) * 0x1p-42) fallback = true;   // (more than ~1000 ulps apart: outside what the crossing zones cover)
        };
        auto finish = [&]() -> double {
          if (dl != 0.0)
            for (; evp < n_ev; evp++) {
              const int x = ev_x[evp];
              const double s1 = ((x ? pse[x - 1] : 0.0) + dl) + ev_t[evp];
              dl = s1 - pse[x];
            }
          return (E ? pse[E - 1] : 0.0) + dl;
        };
        // delta mirrored on whole groups of het sites only (else some row mixes mirrored and unmirrored sites: add everything)
        bool whole = true;
        for (uint32_t rest = dd; rest && whole;) { const int i = __ffs((int)rest) - 1; const uint32_t gm = s_grp[i]; whole = (dd & gm) == gm; rest &= ~gm; }
        if (!same_class) { fallback = true;
#ifdef ENUM_VERIFY
          atomicAdd(&P.tie_ctr[4], 1ull);
#endif
        }
        else if (!whole) fallback = true;   // delta differs inside a group (a site whose two orientations score alike): its rows' sums differ by more than rounding
        else {
          const bool all = dd == r_eta0 && dd != 0u;   // the mirror image of the whole configuration: sigma and delta negated together
          for (uint32_t w0 = 0; w0 < nk && !fallback; w0 += 8) {
            unsigned long long d4[8];
#pragma unroll
            for (int u = 0; u < 8; u++) d4[u] = w0 + u < nk ? ld_st(my_e, w0 + u) : 0ull;
#pragma unroll
            for (int u = 0; u < 8; u++) {
              if (w0 + u >= nk) break;
              unsigned long long flip = all ? ~0ull : 0ull;   // rows whose group is mirrored
              if (!all) for (uint32_t rest = dd; rest; rest &= rest - 1) flip |= repmask[(uint32_t)(__ffs((int)rest) - 1) * nk + w0 + u];
              unsigned long long d = (d4[u] ^ sg_ref[w0 + u] ^ flip) & hetw[w0 + u];
              while (d && !fallback) {
                const int b = __ffsll((long long)d) - 1;
                d &= d - 1;
                row_check((int)(w0 + u) * 64 + b, 0u - ((uint32_t)(d4[u] >> b) & 1u));
              }
            }
          }
          if (!fallback) t_sum[tid] = finish();
        }
      }
#ifdef ENUM_PROF
      const long long pf1 = (long long)wall_clock64();
      if (lane == 0) atomicAdd(&s_tloc[wave], (uint32_t)(pf1 - pf0));
#endif
      if (__ballot(fallback)) { if (fallback) t_sum[tid] = full_sum(my_e, dneg, eta0, etap); }
#ifdef ENUM_PROF
      if (lane == 0) atomicAdd(&s_tfull[wave], (uint32_t)((long long)wall_clock64() - pf1));
#endif
#ifdef ENUM_VERIFY   // (measurement build: every sum that left the reference's chain, once more by adding all terms)
      { const bool dv = (uint32_t)tid < n && !fallback && t_sum[tid] != (E ? pse[E - 1] : 0.0);
        if (__ballot(dv)) { if (dv) { atomicAdd(&P.tie_ctr[7], 1ull); if (full_sum(my_e, dneg, eta0, etap) != t_sum[tid]) atomicAdd(&P.tie_ctr[2], 1ull); } }
        if (fallback) atomicAdd(&P.tie_ctr[6], 1ull); }
#endif
      __syncthreads();
      {   // phase.rs:1117: strictly greater replaces -- the largest sum, the earliest of equals (a reduction per wave, then over the waves)
        double bs = (uint32_t)tid < n ? t_sum[tid] : -__builtin_huge_val();
        uint32_t bj = (uint32_t)tid;
        for (int d = 32; d >= 1; d >>= 1) {
          const double os = __shfl_xor(bs, d, 64); const uint32_t oj = (uint32_t)__shfl_xor((int)bj, d, 64);
          if (os > bs || (os == bs && oj < bj)) { bs = os; bj = oj; }
        }
        if (lane == 0) { s_wsum[wave] = bs; s_wj[wave] = bj; }
        __syncthreads();
        if (tid == 0) {
          double cb = s_wsum[0]; uint32_t cj = s_wj[0];
          for (int w = 1; w < ENUM_WAVES; w++) if (s_wsum[w] > cb) { cb = s_wsum[w]; cj = s_wj[w]; }
          if (cur == 0 || cb > s_winsum) { s_win = t_e[cj]; s_winsum = cb; }
        }
      }
      __syncthreads();
    }
  }
  if (dif) ENUM_PT(6);
#ifdef ENUM_PROF
  __shared__ unsigned long long s_prof;
  if (tid == 0) s_prof = 0;
  __syncthreads();
  if (tid == 0 && dif) s_prof = ((unsigned long long)(((long long)wall_clock64() - prof_t5) & 0xfffff) << 24) | ((unsigned long long)min(max(max(s_tloc[0], s_tloc[1]), max(s_tloc[2], s_tloc[3])) >> 4, 4095u) << 12) | min(max(max(s_tfull[0], s_tfull[1]), max(s_tfull[2], s_tfull[3])) >> 4, 4095u);
#endif
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
#ifdef ENUM_PROF
  __syncthreads();
  if (tid == 0) atomicMax(&P.tie_ctr[7], ((unsigned long long)(((long long)wall_clock64() - prof_t0) & 0xfffff) << 44) | s_prof);
#endif
}

// wave64 inclusive add-scan, DPP row shifts / broadcasts (lanes without a source add 0)
__device__ __forceinline__ int enum_incl_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// the f64 scores of ONE row whose fixed-point sums tie (a row with a het entry and more than two entries): q < qn of
// phase.rs:77-96, 845-858 -> flip.  sneg: the row's sigma is -1.
__device__ __forceinline__ bool enum_tie_row_flips(const int row, const uint32_t sneg, const uint32_t dneg, const uint32_t eta0, const uint32_t etap,
                                                   const uint16_t* rp, const uint16_t* ent16, const double* lut) {
  double lp = 0.0, lm = 0.0;   // the running sums log_q2 (sigma = +1) and log_q3 (sigma = -1), entry order
  const int x1 = rp[row + 1];
  for (int x = rp[row]; x < x1; x += 4) {   // four entries per round trip: their words, then their table values, then the adds in order
    uint32_t v[4]; double tp[4], tm4[4];
#pragma unroll
    for (int u = 0; u < 4; u++) v[u] = ent16[min(x + u, x1 - 1)];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const uint32_t i = v[u] & 31u, pbit = (v[u] >> 5) & 1u, q = (v[u] >> 6) & 31u;
      const uint32_t isHet = (eta0 >> i) & 1u;
      const uint32_t mp = isHet ? (pbit ^ ((dneg >> i) & 1u)) : (((etap >> i) & 1u) ? pbit : pbit ^ 1u);
      const uint32_t mm = isHet ? mp ^ 1u : mp;
      tp[u] = lut[(mp ? 32u : 0u) + q]; tm4[u] = lut[(mm ? 32u : 0u) + q];
    }
#pragma unroll
    for (int u = 0; u < 4; u++) if (x + u < x1) { lp += tp[u]; lm += tm4[u]; }
  }
  if (lp == lm) return false;   // q == qn: nothing to decide (the usual case: the het entries' terms pair up in place)
  const double l1 = sneg ? lm : lp, l1n = sneg ? lp : lm;
  const double den = lp + lm;
  const double q = 1.0 - l1 / den, qn = 1.0 - l1n / den;
  return q < qn;
}

// tiles of restarts of regions whose per-lane share is <= CK entries (host decides).  Every restart stores its objective and --
// unless a better one is known already -- its final state and signature; k4_enum_resolve decides the winner afterwards.
template <int CK>
__global__ void __launch_bounds__(64 * ENUM_WAVES, ENUM_OCC)   // (three waves per SIMD: <= 168 VGPRs)
k4_enum_reg(PhaseDev P, const EnumSpan* __restrict__ spans, int32_t n_spans, uint32_t per, const int64_t* __restrict__ job_base,
            long long* __restrict__ job_obj, const int64_t* __restrict__ st_base, unsigned long long* __restrict__ st_words,
            long long* __restrict__ region_best, uint32_t* __restrict__ redo /* [0]: count, [4 ..]: (slot, restart) pairs; nullptr: none */,
            uint32_t redo_cap) {
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
  uint16_t* ent16 = (uint16_t*)(lds + L.ent16);
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
      ent16[e] = (uint16_t)((meta & 63u) | ((v & 31u) << 6) | (e + 1 == e1 ? 0x800u : 0u));
    }
  }
  __syncthreads();
  // ---- per wave: my share of the matrix into registers
  const int lane = tid & 63, wave = tid >> 6;
  unsigned long long* sgb = (unsigned long long*)(lds + L.state + wave * L.stride);   // bit = 1: sigma == -1
  unsigned long long* Macc = sgb + (R + 63) / 64 + 1;
  uint32_t* tq = (uint32_t*)(Macc + 32);                 // queue of the wave's tied rows (sigma step)
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
  uint32_t n_tie_f64 = 0, n_tie_flip = 0, n_dtie = 0, n_step = 0, n_tie_unres = 0;   // census of this wave's restarts (lane 0 adds them up at the end)
  // rows of a lane as bits of a mask: 32 bits in the register-resident form (the host sends a region there only if no lane owns
  // more than 32 rows), 64 in the streaming form
#ifdef ENUM_MASK64
  using mask_t = unsigned long long;
#else
  using mask_t = typename std::conditional<(CK > 0), uint32_t, unsigned long long>::type;
#endif
  // my rows of more than two entries (the rows whose tie needs the f64 scores, see the sigma step)
  mask_t bigm = 0;
  for (int row = r_a; row < (int)first_row[lane + 1]; row++) if (rp[row + 1] - rp[row] > 2) bigm |= (mask_t)1 << (row - r_a);
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
    const uint32_t ev0 = n_dtie + n_step, n_step0 = n_step, n_dtie0 = n_dtie;   // (wave-uniform) this restart's ties at a delta / eta maximum + tie-only steps
    while (hg_inc | h_inc) {
      // ---- sigma step (phase.rs:824-862)
      {
        const unsigned long long w0 = sgb[r_a >> 6], w1 = sgb[(r_a >> 6) + 1];
        const mask_t win = (mask_t)(wsh ? (w0 >> wsh) | (w1 << (64 - wsh)) : w0);
        int alo = 0, ahi = 0;
        uint32_t uacc = 0;
        mask_t fm = 0, tm = 0;
        auto sig_one = [&](uint32_t v0, uint32_t v1) {
          const uint32_t m = v0 >> 24, i = m & 31u, roff = v1 >> 24;
          const uint32_t sneg = (uint32_t)(win >> roff);
          const uint32_t use = (m >> 7) & (eta0 >> i) & 1u;                 // het sites only
          const uint32_t hit = ((m >> 5) ^ sneg ^ (dneg >> i)) & use;       // p == sigma * delta
          const uint32_t mis = hit ^ use;
          alo += __mul24((int)hit, (int)v0) - __mul24((int)mis, (int)v0);   // A - B of phase.rs:824-862
          ahi += __mul24((int)hit, (int)v1) - __mul24((int)mis, (int)v1);
          uacc |= use;
          const uint32_t em = (uint32_t)((int)(v0 << 1) >> 31);   // all ones at the last entry of a row
          // sign of ahi * 2^23 + alo: fold alo's carry into ahi, the remainder is in [0, 2^23)
          const int top = ahi + (alo >> 23);
          const mask_t bit = (mask_t)(long long)(int)em & ((mask_t)1 << roff);   // (sign-extended: the streaming form's masks have 64 bits)
          fm |= bit & (mask_t)(long long)(top >> 31);
          // A == B at a row with a het entry: the f64 scores decide (a row without one scores the same for both signs, term by term)
          const uint32_t zz = (uint32_t)top | ((uint32_t)alo << 9) | (uacc ^ 1u);
          tm |= zz == 0u ? bit : (mask_t)0;
          alo &= (int)~em; ahi &= (int)~em; uacc &= ~em;   // (bit operations: the compiler's selects cost a compare more)
        };
        if (CK > 0) {
#pragma unroll
          for (int x = 0; x < NREG; x++) {
            if ((x & 3) == 0 && x >= n_sig) break;   // (a lane's slots beyond its share are zero words: no het entry, no row end)
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
#ifdef ENUM_ABL_NOTIE
        tm = 0;
#endif
        if (P.tie_arith < 2) { if (tm) n_tie_unres += (uint32_t)__popcll((unsigned long long)tm); }
        else if (__ballot(tm != 0)) {
          n_tie_f64 += (uint32_t)__popcll((unsigned long long)tm);
          // The tied rows of the whole wave go through a queue and are scored a lane each, whoever owns them (a lane that
          // scored its own rows one after the other kept the wave waiting for the lane with the most: HiFi data ties in ~60
          // rows per step).  A row of two entries ties only with both at het sites, one matching and one not, at one quality:
          // log_q2 = a + b, log_q3 = b + a -- the same double (the first addition, to 0.0, is exact): not queued.
          // Queue slots come from a scan of the lanes' counts (no atomic round trips).
          mask_t own = 0;   // (rows the queue had no room for)
          const mask_t tb = tm & bigm;
          const int cnt = __popcll((unsigned long long)tb), incl = enum_incl_scan(cnt);
          const uint32_t n_all = (uint32_t)__builtin_amdgcn_readlane(incl, 63);
          uint32_t slot = (uint32_t)(incl - cnt);
          for (mask_t t2 = tb; t2; t2 &= t2 - 1, slot++) {
            const int roff = __ffsll((long long)(unsigned long long)t2) - 1;
            if (slot < ENUM_TQ) tq[slot] = (uint32_t)(r_a + roff) | ((uint32_t)(win >> roff) & 1u) << 16; else own |= (mask_t)1 << roff;
          }
          if (n_all) wave_lds_sync();
          const uint32_t nq = min(n_all, ENUM_TQ);
          uint32_t nfl = 0;
          for (uint32_t j = lane; j < nq; j += 64) {
            const uint32_t ent = tq[j], row = ent & 0xffffu;
            if (enum_tie_row_flips((int)row, ent >> 16, dneg, eta0, etap, rp, ent16, lut)) { atomicXor(&sgb[row >> 6], 1ull << (row & 63u)); nfl++; }
          }
          for (; own; own &= own - 1) {
            const int roff = __ffsll((long long)(unsigned long long)own) - 1;
            if (enum_tie_row_flips(r_a + roff, (uint32_t)(win >> roff) & 1u, dneg, eta0, etap, rp, ent16, lut)) { fm |= (mask_t)1 << roff; nfl++; }
          }
          n_tie_flip += nfl;
          if (!any && __ballot(nfl != 0)) n_step++;   // only tie flips: "no improvement" (check_new_haplotag's sums are not formed)
        }
        if (fm) {
          const unsigned long long fm64 = (unsigned long long)fm;
          atomicXor(&sgb[r_a >> 6], fm64 << wsh);
          if (wsh && (fm64 >> (64 - wsh))) atomicXor(&sgb[(r_a >> 6) + 1], fm64 >> (64 - wsh));
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
    // A restart that met such a tie took "first maximum" / "no improvement" there.  With the complete tie contract (tie_arith >= 3) it
    // goes to the repair list: k4_enum_redo runs it once more through the one-workgroup cross_optimize, which decides those ties by
    // the reference's f64 scores, and overwrites what this wave leaves (rare: C4 meets 147 such steps in 45 000 x 1 000 restarts).
    bool redone = false;
    if (redo && n_dtie + n_step != ev0 && P.tie_arith >= 3) {
      uint32_t at = 0;
      if (lane == 0) at = atomicAdd(&redo[0], 1u);
      at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
      if (at < redo_cap) {
        if (lane == 0) { redo[4 + 2 * at] = (uint32_t)t.slot; redo[5 + 2 * at] = e; }
        const uint32_t dd = n_dtie - n_dtie0, ds = n_step - n_step0;   // (not unresolved: the repair pass decides them)
        n_dtie -= dd; n_step -= ds;
        redone = true;
      }
    }
    // objective (phase.rs:257-276) = sum over phase entries of fe + hit * w = sum_i (F_i + hits_i) over live SNPs
    const long long total = wave_sum_ll_dpp(obj_i);
    // a restart below the best objective seen so far in this region can never win: neither signature nor state is kept
    // (region_best: monotone, device-coherent; a stale smaller value only costs a store that is not needed)
    const long long seen = __hip_atomic_load(&region_best[t.slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool keep = __builtin_amdgcn_readfirstlane((int)(total >= seen)) != 0;
    // (a restart on the repair list does not raise the bar: its objective may still change, and the others keep their state
    // against the best of the restarts that stand)
    if (total > seen && lane == 0 && !redone) (void)__hip_atomic_fetch_max(&region_best[t.slot], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // signature of the final configuration: a hash of the match bits [p == x] of all entries -- what the f64 form of the
    // objective (a running sum of LUT[match][q] over the entries in row order) depends on beside the matrix (enum_resolve)
    unsigned long long sig = 0;
    if (keep) {
      const unsigned long long w0 = sgb[r_a >> 6], w1 = sgb[(r_a >> 6) + 1];
      const mask_t win = (mask_t)(wsh ? (w0 >> wsh) | (w1 << (64 - wsh)) : w0);
      unsigned long long hs = 0;
      uint32_t word = 0; int nb = 0, widx = 0;
      auto sig_bit = [&](uint32_t v0, uint32_t v1) {
        const uint32_t m = v0 >> 24, i = m & 31u, roff = v1 >> 24, pbit = (m >> 5) & 1u;
        const uint32_t sneg = (uint32_t)(win >> roff) & 1u;
        const uint32_t match = ((eta0 >> i) & 1u) ? (pbit ^ sneg ^ ((dneg >> i) & 1u)) : (((etap >> i) & 1u) ? pbit : pbit ^ 1u);
        word |= (match & (m >> 7)) << nb;
        if (++nb == 32) { hs = mix64(hs ^ (word + (unsigned long long)(lane * 64 + widx + 1) * 0x9E3779B97F4A7C15ULL)); word = 0; nb = 0; widx++; }
      };
#ifdef ENUM_ABL_NOSIG
      if (false) {
#else
      if (CK > 0) {
#endif
#pragma unroll
        for (int x = 0; x < NREG; x++) {
          if (x >= n_sig) break;
          asm volatile("" : "+v"(re0[x]), "+v"(re1[x]));
          sig_bit(re0[x], re1[x]);
        }
#ifdef ENUM_ABL_NOSIG
      } else if (false) {
#else
      } else {
#endif
        for (int x0 = 0; x0 < n_sig; x0 += 4) {
          uint2 v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) v[u] = s0 + x0 + u < s1 ? csr[s0 + x0 + u] : make_uint2(0, 0);
#pragma unroll
          for (int u = 0; u < 4; u++) sig_bit(v[u].x, v[u].y);
        }
      }
      if (nb) hs = mix64(hs ^ (word + (unsigned long long)(lane * 64 + widx + 1) * 0x9E3779B97F4A7C15ULL));
      sig = (unsigned long long)wave_sum_ll((long long)hs);
    }
    // (device-coherent stores: read by the region's last tile, possibly on another XCD)
    unsigned long long* stp = st_reg + (size_t)e * stw;
    // (plain stores: their reader is the next launch, k4_enum_resolve)
    if (keep) for (int k = lane; k < nk; k += 64) stp[k] = sgb[k];
    if (lane == 0) {
      if (keep) { stp[nk] = (unsigned long long)dneg | ((unsigned long long)eta0 << 32); stp[nk + 1] = (unsigned long long)etap; stp[nk + 2] = sig; }
      job_obj[job_base[t.slot] + e] = total;
    }
    wave_lds_sync();
  };
  for (uint32_t kk = wave; kk < ne; kk += ENUM_WAVES) run_restart(t.e0 + kk);
  n_tie_f64 = (uint32_t)wave_sum_ll((long long)n_tie_f64); n_tie_flip = (uint32_t)wave_sum_ll((long long)n_tie_flip); n_tie_unres = (uint32_t)wave_sum_ll((long long)n_tie_unres);   // (per-lane counts of the lanes' own rows)
  if (lane == 0) {
    if (n_tie_f64) TIE_COUNT(P.tie_ctr, TIE_SIGMA_F64, (unsigned long long)n_tie_f64);
    if (n_tie_flip) TIE_COUNT(P.tie_ctr, TIE_SIGMA_FLIPS, (unsigned long long)n_tie_flip);
    if (n_tie_unres) TIE_COUNT(P.tie_ctr, TIE_SIGMA_UNRES, (unsigned long long)n_tie_unres);
    if (n_dtie) TIE_COUNT(P.tie_ctr, TIE_DELTA_UNRES, (unsigned long long)n_dtie);
    if (n_step) TIE_COUNT(P.tie_ctr, TIE_STEP_UNRES, (unsigned long long)n_step);
  }
}

// ---------------------------------------------------------------------------------------------
// k4_enum_bits: the same restarts, EIGHT per wave as bit states (round 5; what k4_grid_batch.h does for the chain's speculative
// half-rounds).  The 2^S restarts of a region differ in their start delta and sigma only, so a wave carries restarts e .. e + 7 as eight
// bits per row (sg8: bit s = sigma of restart s is -1) and per SNP (het / delta < 0 / eta == +1): an entry of the matrix is read and
// decoded once, and its contribution to a row or column sum costs three instructions per restart -- the masks are "spread" (state s at
// bit 2 s), het | mis << 1 is the 2-bit signed factor of every state, one v_bfe_i32 and two v_mad_i32_i24 on the limbs of w.
//   sigma step : lane <-> a run of whole rows in CSR order as in k4_enum_reg, the entries streamed from LDS; at a row's last entry the
//                lane takes the eight decisions (sign of the sum; an exact tie with a het entry goes to the wave's queue, where a lane
//                per (row, state) forms the f64 scores -- enum_tie_row_flips, unchanged)
//   delta step : lane <-> a chunk of the CSC entries, M[state][SNP] by LDS atomics; then a lane per (state, SNP), eight SNPs per pass,
//                takes the four-way decision; the ballots ARE the restarts' new masks
//   The eight restarts run in lock step; one that has settled (or made 21 iterations) is frozen.  Everything a restart leaves --
//   objective, final state, signature, repair-list entry, census -- is what k4_enum_reg leaves (tests compare the two kernels).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t enum_spread8(uint32_t x) { x = (x | (x << 4)) & 0x0F0Fu; x = (x | (x << 2)) & 0x3333u; x = (x | (x << 1)) & 0x5555u; return x; }
__device__ __forceinline__ int enum_mad24(int a, int b, int c) { int d; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }

#ifndef ENUM_BITS_OCC
#define ENUM_BITS_OCC 2
#endif
__global__ void __launch_bounds__(64 * ENUM_WAVES, ENUM_BITS_OCC)   // (waves per SIMD the register allocation aims at; -DENUM_BITS_OCC: measurement builds)
k4_enum_bits(PhaseDev P, const EnumSpan* __restrict__ spans, int32_t n_spans, uint32_t per, const int64_t* __restrict__ job_base,
             long long* __restrict__ job_obj, const int64_t* __restrict__ st_base, unsigned long long* __restrict__ st_words,
             long long* __restrict__ region_best, uint32_t* __restrict__ redo, uint32_t redo_cap) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const EnumTile t = enum_tile_of(P, spans, n_spans, per, false);
  const RegionDev rd = P.reg[t.slot];
  const int R = rd.R, S = rd.S;
  const uint32_t E = (uint32_t)P.prow_ptr[rd.rp_off + R];
  const EnumLayout L = enum_layout(R, E, true, (uint32_t)S);
  const int Sp = (int)enum_bits_sp((uint32_t)S);
  uint2* wl2 = (uint2*)lds;
  double* lut = (double*)(lds + L.lut);
  uint2* csr = (uint2*)(lds + L.csr);
  uint32_t* csc = (uint32_t*)(lds + L.csc);
  uint16_t* rp = (uint16_t*)(lds + L.rp); uint16_t* first_row = (uint16_t*)(lds + L.first_row);
  uint16_t* ent16 = (uint16_t*)(lds + L.ent16);
  const int tid = threadIdx.x, nt = blockDim.x;
  const uint32_t c = enum_chunk(E);
  __shared__ long long s_cF[32], s_cW[32], s_cRef[32], s_cVar[32], s_het[32];
  __shared__ uint8_t s_live[32];
  // ---- stage the region (once per workgroup; as k4_enum_reg)
  if (tid < 32) {
    const long long w = tid < 31 ? P.lut.f1e[tid] - P.lut.fe[tid] : 0;
    wl2[tid] = make_uint2((uint32_t)w & 0x7fffffu, (uint32_t)(w >> 23) & 0xffffffu);
  }
  if (tid < 64) lut[tid] = (tid & 31) < 31 ? (tid < 32 ? P.lut64->le[tid] : P.lut64->l1e[tid - 32]) : 0.0;
  const int32_t* g_rp = P.prow_ptr + rd.rp_off;
  for (int r = tid; r <= R; r += nt) rp[r] = (uint16_t)g_rp[r];
  __shared__ int32_t cps[33];
  if (tid <= S && tid < 33) cps[tid] = P.ccol_ptr[rd.cp_off + tid];
  int eta_init = 0;
  if (tid < 32) {
    long long cF = 0, cW = 0, cRef = 0, cVar = 0, het = 0; bool live = false;
    if (tid < S) {
      const long long* sc = P.snp_const + 4ll * (rd.snp_off + tid);
      cF = sc[0]; cW = sc[1]; cRef = sc[2] + P.lut.f_homref; cVar = sc[3] + P.lut.f_homvar;
      const int n = P.ccol_ptr[rd.cp_off + tid + 1] - P.ccol_ptr[rd.cp_off + tid];
      het = P.lut.f_het0 - (long long)n * P.lut.f_log2;                     // phase.rs:136-144
      live = P.snp_fp[rd.snp_off + tid] != 0 && n > 0;
      eta_init = init_genotype(P.snp_vt[rd.snp_off + tid]);
    }
    s_cF[tid] = cF; s_cW[tid] = cW; s_cRef[tid] = cRef; s_cVar[tid] = cVar; s_het[tid] = het; s_live[tid] = live ? 1 : 0;
  }
  __shared__ uint32_t s_e0i, s_epi;
  if (tid < 64) {   // (the first wave: lanes < 32 hold the SNPs' start genotypes)
    const uint32_t a = (uint32_t)__ballot(tid < S && tid < 32 && eta_init == 0), b = (uint32_t)__ballot(tid < S && tid < 32 && eta_init == 1);
    if (tid == 0) { s_e0i = a; s_epi = b; }
  }
  __syncthreads();
  for (int e = tid; e < (int)E; e += nt) {
    const uint32_t cv = P.cval[rd.e_off + e];
    int col = 0;
    for (int i = 0; i < S; i++) col += (int)((uint32_t)e >= (uint32_t)cps[i + 1]);
    csc[e] = (uint32_t)P.crow[rd.e_off + e] | ((uint32_t)col << 16) | ((cv & 32u) << 16) | ((cv & 31u) << 22) | 0x80000000u;
  }
  // Round 6: the rows of ONE entry (57 % of the phasing rows of the ONT-cDNA / MAS-Seq batches: a read over one het site) are taken out of
  // the sigma step's main loop: their decision is the sign of one term -- bit operations, no sums, no tie -- in a pass of their own, a
  // lane per row.  The rows of MORE than one entry are sorted by their length (descending, row order within a length: the same order in every
  // workgroup of a region -- the signatures are compared across them) and dealt to the lanes 64 at a time, a LANE PER ROW in lock step: the
  // eight-sums row end, as long as three entries' work, runs once per 64 rows (with a lane per run of rows some lane met a row end at nearly
  // every step, so the whole wave ran it at nearly every step).
  //   perm[0 .. nM)        the rows of more than one entry, (length descending, row ascending); group g = perm[64 g .. 64 g + 64)
  //   gof[g]               first csr slot of group g, in units of 64 slots; its entries: csr[64 gof[g] + 64 k + j] = entry k of row perm[64 g + j]
  //                        (zero words behind a row's last entry: the group is as long as its first row); gof[nG] = end of the groups
  //   perm[nM .. nM + nS)  the rows of one entry, in row order: csr[eM + x], eM = 64 gof[nG]
  uint16_t* const perm = (uint16_t*)(lds + L.pos);
  uint16_t* const gof = perm + ((R + 3) & ~3);
  uint16_t* const kidx = (uint16_t*)(lds + L.state);   // (staging only: the waves' state area is set up behind the barrier below)
  __shared__ int s_cnt[3];
  __shared__ int s_lstart[32];   // first position in perm of the rows of length L
  __syncthreads();
  if (tid < 64) {
    const int ln = tid;
    const unsigned long long lt = (1ull << ln) - 1ull;
    int nS = 0;
    int tot[32];   // rows of length L so far (wave-uniform)
#pragma unroll
    for (int Lk = 0; Lk < 32; Lk++) tot[Lk] = 0;
    for (int base = 0; base < R; base += 64) {
      const int r = base + ln;
      const int n = r < R ? min(31, (int)rp[r + 1] - (int)rp[r]) : 0;   // (a row has at most S <= 31 entries)
      const unsigned long long bs = __ballot(n == 1);
      if (n == 1) kidx[r] = (uint16_t)(nS + __popcll(bs & lt));
      nS += __popcll(bs);
      if (__ballot(n >= 2)) {
#pragma unroll
        for (int Lk = 2; Lk < 32; Lk++) {
          if (Lk <= S) {   // (a row has at most S entries; no break: the loop has to unroll for tot[] to stay in registers)
            const unsigned long long bm = __ballot(n == Lk);
            if (n == Lk) kidx[r] = (uint16_t)(tot[Lk] + __popcll(bm & lt));   // rank among the rows of its length, in row order
            tot[Lk] += __popcll(bm);
          }
        }
      }
    }
    int mine = 0;
#pragma unroll
    for (int Lk = 2; Lk < 32; Lk++) if (ln == Lk) mine = tot[Lk];
    // descending lengths: start[L] = rows longer than L
    int longer = 0;
    for (int Lk = 31; Lk >= 2; Lk--) { const int tl = __shfl(mine, Lk, 64); if (ln == Lk) s_lstart[ln] = longer; longer += tl; }
    if (ln == 0) { s_cnt[0] = longer; s_cnt[2] = nS; }
  }
  __syncthreads();
  const int nM = s_cnt[0], nS = s_cnt[2], nG = (nM + 63) >> 6;
  for (int r = tid; r < R; r += nt) {
    const int n = min(31, (int)rp[r + 1] - (int)rp[r]);
    if (n >= 2) perm[s_lstart[n] + kidx[r]] = (uint16_t)r;
  }
  __syncthreads();
  if (tid < 64) {   // group offsets: a group is as long as its first (longest) row
    int carry = 0;
    for (int g0 = 0; g0 <= nG; g0 += 64) {
      const int g = g0 + tid;
      const int len = g < nG ? min(31, (int)rp[perm[64 * g] + 1] - (int)rp[perm[64 * g]]) : 0;
      const int inc = wave_incl_scan(len);
      if (g <= nG) gof[g] = (uint16_t)(carry + inc - len);
      carry += __builtin_amdgcn_readlane(inc, 63);
    }
  }
  __syncthreads();
  const int eM = 64 * (int)gof[nG];
  for (int x = tid; x < eM; x += nt) csr[x] = make_uint2(0u, 0u);
  __syncthreads();
  for (int r = tid; r < R; r += nt) {
    const int e0 = rp[r], e1 = rp[r + 1];
    if (e0 == e1) continue;
    int at, step;
    if (e1 - e0 >= 2) {
      const int k = s_lstart[min(31, e1 - e0)] + kidx[r];
      at = 64 * (int)gof[k >> 6] + (k & 63); step = 64;
    } else { perm[nM + kidx[r]] = (uint16_t)r; at = eM + kidx[r]; step = 1; }
    for (int e = e0; e < e1; e++) {
      const uint32_t v = P.pval[rd.e_off + e];
      const uint32_t meta = (uint32_t)P.pcol[rd.e_off + e] | (v & 32u) | (e + 1 == e1 ? 64u : 0u) | 128u;
      const uint2 w = wl2[v & 31u];
      csr[at + (e - e0) * step] = make_uint2(w.x | (meta << 24), w.y);
      ent16[e] = (uint16_t)((meta & 63u) | ((v & 31u) << 6) | (e + 1 == e1 ? 0x800u : 0u));
    }
  }
  __syncthreads();
  // ---- per wave
  const int lane = tid & 63, wave = tid >> 6;
  uint8_t* const wst = lds + L.state + wave * L.stride;
  uint8_t* const sg8 = wst;                                                     // [R]: bit s = sigma of restart s is -1
  unsigned long long* const Macc = (unsigned long long*)(wst + ((R + 15) & ~7u)); // [8][Sp]
  uint2* const mt = (uint2*)(Macc + 8 * Sp);                                    // [32]: .x = het16 | dneg16 << 16 (spread), .y = dneg8 | het8 << 8 | etap8 << 16
  uint32_t* const ms = (uint32_t*)(mt + 32);                                    // [3][8]: dneg, eta0, etap of every restart
  uint32_t* const tq = ms + 24;                                            // queue of tied rows: row | tie8 << 16 | sneg8 << 24
  uint32_t* const tq_n = tq + ENUM_TQ;
  const int c0 = min((int)E, lane * (int)c), c1 = min((int)E, (lane + 1) * (int)c);
  const int n_del = (int)min(c, E);
  const uint32_t smask = S >= 32 ? 0xffffffffu : ((1u << S) - 1u);
  const uint32_t e0_init = s_e0i, ep_init = s_epi;
  const int nk = (R + 63) / 64;
  unsigned long long* const st_reg = st_words + st_base[t.slot];
  const uint32_t stw = enum_state_words((uint32_t)R);
  uint32_t n_tie_f64 = 0, n_tie_flip = 0, n_tie_unres = 0;   // per lane
  uint32_t n_dtie = 0, n_step = 0;                          // wave-uniform
  const int n_pass = (S + 7) >> 3;
  for (uint32_t g0 = 8u * wave; g0 < t.ne; g0 += 8u * ENUM_WAVES) {
    const uint32_t e_base = t.e0 + g0;                      // (a multiple of 8: restart s of the group is e_base + s)
    const int nst = (int)min(8u, t.ne - g0);
    const uint32_t valid8 = (1u << nst) - 1u;
    // ---- start states
    if (lane < 8) { ms[lane] = (e_base + (uint32_t)lane) & smask; ms[8 + lane] = e0_init; ms[16 + lane] = ep_init; }
    {
      // init_assignment (phase.rs:673-680): top bit of the draw clear -> sigma = -1.  The draw of (restart e_base + s, row) is mix64(seed + (S + R +
      // (e_base + s) R + row + 1) G): the argument advances by R G per restart and by 64 G per 64 rows (adds instead of 64-bit multiplies), and only the top
      // bit of the second multiply of mix64 is formed (three 32-bit multiplies instead of the full product: the last xor-shift cannot reach bit 63).
      constexpr uint64_t G = 0x9E3779B97F4A7C15ULL;
      const uint64_t RG = (uint64_t)R * G;
      uint64_t a_row = rd.seed + ((uint64_t)S + (uint64_t)R + (uint64_t)e_base * (uint64_t)R + (uint64_t)lane + 1ull) * G;
      for (int k = 0; k < nk; k++) {
        const int row = lane + 64 * k;
        uint32_t b = 0;
        uint64_t a = a_row;
#pragma unroll
        for (int s = 0; s < 8; s++) {
          uint64_t z = (a ^ (a >> 30)) * 0xBF58476D1CE4E5B9ULL;
          z ^= z >> 27;
          const uint32_t zl = (uint32_t)z, zh = (uint32_t)(z >> 32);
          const uint32_t top = __umulhi(zl, 0x133111EBu) + zl * 0x94D049BBu + zh * 0x133111EBu;   // bits 32..63 of z * 0x94D049BB133111EB
          b |= (uint32_t)((top >> 31) == 0) << s;
          a += RG;
        }
        if (row < R) sg8[row] = (uint8_t)b;
        a_row += 64ull * G;
      }
    }
    for (int k = lane; k < 8 * Sp; k += 64) Macc[k] = 0;
    if (lane == 0) tq_n[0] = 0;
    wave_lds_sync();
    uint32_t act = valid8, hinc = valid8, hginc = valid8;
    int iters = 0;
    uint32_t ev_d[8], ev_s[8];   // this group's delta / eta ties and tie-only steps per restart (wave-uniform)
#pragma unroll
    for (int s = 0; s < 8; s++) { ev_d[s] = 0; ev_s[s] = 0; }
    long long objl[4] = {0, 0, 0, 0};   // lane (state, SNP of pass p): the chosen branch's data term of the restart's last iteration
    while (act) {
      // ---- the SNPs' masks of this iteration from the restarts' masks
      if (lane < 32) {
        uint32_t dn = 0, h8 = 0, ep = 0;
#pragma unroll
        for (int s = 0; s < 8; s++) { dn |= ((ms[s] >> lane) & 1u) << s; h8 |= ((ms[8 + s] >> lane) & 1u) << s; ep |= ((ms[16 + s] >> lane) & 1u) << s; }
        mt[lane] = make_uint2(enum_spread8(h8) | (enum_spread8(dn) << 16), dn | (h8 << 8) | (ep << 16));
      }
      wave_lds_sync();
      // ---- sigma step (phase.rs:824-862)
      uint32_t any8 = 0, tflip8 = 0;
      {
        int alo[8], ahi[8];
#pragma unroll
        for (int s = 0; s < 8; s++) { alo[s] = 0; ahi[s] = 0; }
        for (int g = 0; g < nG; g++) {   // 64 rows of (nearly) one length, a lane per row
          const int gb = __builtin_amdgcn_readfirstlane((int)gof[g]), Lg = __builtin_amdgcn_readfirstlane((int)gof[g + 1]) - gb;
          const int kq = 64 * g + lane;
          const int row = kq < nM ? (int)perm[kq] : 0;
          const uint32_t sgb = kq < nM ? (uint32_t)sg8[row] : 0u;
          const uint32_t sg16 = enum_spread8(sgb);
          const uint2* const pe = csr + 64 * gb + lane;
          uint32_t uacc = 0;
          for (int x0 = 0; x0 < Lg; x0 += 2) {   // (two entries per step: most groups are two to four long, and Lg is the wave's)
            const bool two = x0 + 1 < Lg;
            uint2 v[2]; uint32_t mmv[2];
            v[0] = pe[64 * x0]; v[1] = two ? pe[64 * (x0 + 1)] : make_uint2(0, 0);
            mmv[0] = mt[(v[0].x >> 24) & 31u].x; mmv[1] = mt[(v[1].x >> 24) & 31u].x;
#pragma unroll
            for (int u = 0; u < 2; u++) {
              if (u == 1 && !two) break;
              const uint32_t v0 = v[u].x, v1 = v[u].y;
              const uint32_t m = v0 >> 24;
              const uint32_t mm = mmv[u];
              const uint32_t use16 = (m & 128u) ? (mm & 0xFFFFu) : 0u;                          // het sites only (a zero word: behind the row's last entry)
              const uint32_t hit16 = (((m & 32u) ? 0x5555u : 0u) ^ sg16 ^ (mm >> 16)) & use16;   // p == sigma * delta
              const uint32_t code = use16 | ((hit16 ^ use16) << 1);                              // 01: +w (A), 11: -w (B)
              uacc |= use16;
#pragma unroll
              for (int s = 0; s < 8; s++) {
                const int sgn = __builtin_amdgcn_sbfe((int)code, 2 * s, 2);
                alo[s] = enum_mad24(sgn, (int)v0, alo[s]); ahi[s] = enum_mad24(sgn, (int)v1, ahi[s]);
              }
            }
          }
          {   // the rows' eight decisions
            uint32_t fl = 0, tie = 0;
#pragma unroll
            for (int s = 0; s < 8; s++) {
              const int top = ahi[s] + (alo[s] >> 23);   // sign of ahi * 2^23 + alo
              fl |= (uint32_t)(top < 0) << s;
              tie |= (uint32_t)(top == 0 && (alo[s] & 0x7fffff) == 0 && ((uacc >> (2 * s)) & 1u)) << s;
              alo[s] = 0; ahi[s] = 0;
            }
            fl &= act; tie &= act;
            any8 |= fl;
            if (fl) sg8[row] = (uint8_t)(sgb ^ fl);
            if (tie) {
              // A == B at a row with a het entry: the f64 scores decide (a row without one scores the same for both signs, term by term)
              if (P.tie_arith < 2) n_tie_unres += (uint32_t)__popc(tie);
              else {
                n_tie_f64 += (uint32_t)__popc(tie);
                if (rp[row + 1] - rp[row] > 2) {   // (two entries: log_q2 = a + b, log_q3 = b + a -- the same double)
                  const uint32_t at = atomicAdd(tq_n, 1u);
                  if (at < ENUM_TQ) tq[at] = (uint32_t)row | (tie << 16) | (sgb << 24);
                  else {
                    for (uint32_t tt = tie; tt; tt &= tt - 1u) {
                      const int s = __ffs((int)tt) - 1;
                      if (enum_tie_row_flips(row, (sgb >> s) & 1u, ms[s], ms[8 + s], ms[16 + s], rp, ent16, lut)) {
                        atomicXor((uint32_t*)(sg8 + (row & ~3)), 1u << (8 * (row & 3) + s));
                        n_tie_flip++; tflip8 |= 1u << s;
                      }
                    }
                  }
                }
              }
            }
          }
        }
        // ---- the rows of one entry, a lane per row: the sum is +-w, the row flips where it is negative (w < 0 for q <= 3) -- no tie (w != 0)
        for (int x0 = lane; x0 < nS; x0 += 4 * 64) {
          uint2 v[4]; int rw[4];
#pragma unroll
          for (int u = 0; u < 4; u++) { const int x = x0 + 64 * u; v[u] = x < nS ? csr[eM + x] : make_uint2(0, 0); rw[u] = x < nS ? (int)perm[nM + x] : 0; }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const uint32_t v0 = v[u].x, v1 = v[u].y;
            const uint32_t m = v0 >> 24, i = m & 31u;
            if (!(m & 128u)) continue;
            const int row = rw[u];
            const uint32_t sb = sg8[row], s16 = enum_spread8(sb);
            const uint32_t mm = mt[i].x;
            const uint32_t use16 = mm & 0xFFFFu;                                              // het sites only
            const uint32_t hit16 = (((m & 32u) ? 0x5555u : 0u) ^ s16 ^ (mm >> 16)) & use16;   // p == sigma * delta: the term is + w
            const bool wneg = ((v1 >> 23) & 1u) != 0;                                         // (sign of the signed 24-bit limb = sign of w)
            uint32_t f = wneg ? hit16 : (use16 & ~hit16);                                     // states whose sum is negative, at bit 2 s
            f &= 0x5555u; f = (f | (f >> 1)) & 0x3333u; f = (f | (f >> 2)) & 0x0F0Fu; f = (f | (f >> 4)) & 0xFFu;
            f &= act;
            any8 |= f;
            if (f) sg8[row] = (uint8_t)(sb ^ f);
          }
        }
      }
      wave_lds_sync();
      {   // the queue: a lane per (row, state)
        const uint32_t nq = min(tq_n[0], ENUM_TQ);
        for (uint32_t b0 = 0; b0 < nq; b0 += 8) {
          const uint32_t j = b0 + (uint32_t)(lane >> 3);
          const int s = lane & 7;
          if (j < nq) {
            const uint32_t ent = tq[j], row = ent & 0xffffu;
            if ((ent >> (16 + s)) & 1u) {
              if (enum_tie_row_flips((int)row, (ent >> (24 + s)) & 1u, ms[s], ms[8 + s], ms[16 + s], rp, ent16, lut)) {
                atomicXor((uint32_t*)(sg8 + (row & ~3u)), 1u << (8 * (row & 3u) + s));
                n_tie_flip++; tflip8 |= 1u << s;
              }
            }
          }
        }
        if (lane == 0) tq_n[0] = 0;
      }
      {   // OR over the wave
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { any8 |= (uint32_t)__shfl_xor((int)any8, d, 64); tflip8 |= (uint32_t)__shfl_xor((int)tflip8, d, 64); }
      }
#pragma unroll
      for (int s = 0; s < 8; s++) if (((tflip8 & ~any8) >> s) & 1u) { n_step++; ev_s[s]++; }   // only tie flips: "no improvement"
      // (!any: h_inc = false; else both true)
      hinc = (hinc & ~act) | (any8 & act); hginc |= any8 & act;
      wave_lds_sync();
      // ---- delta / eta step (phase.rs:872-959): a lane's chunk is CSC-ordered (SNP index non-decreasing)
      {
        int cur = -1;
        uint32_t alo[8]; int ahi[8];
#pragma unroll
        for (int s = 0; s < 8; s++) { alo[s] = 0; ahi[s] = 0; }
        auto flush = [&]() {
#pragma unroll
          for (int s = 0; s < 8; s++) {
            if (alo[s] | (uint32_t)ahi[s]) atomicAdd(&Macc[s * Sp + cur], (unsigned long long)((long long)alo[s] + (long long)ahi[s] * (1ll << 23)));
            alo[s] = 0; ahi[s] = 0;
          }
        };
        for (int h = 0; h < n_del; h += 4) {
          uint32_t v4[4], sw[4], dn[4]; uint2 wq[4];
#pragma unroll
          for (int x = 0; x < 4; x++) v4[x] = c0 + h + x < c1 ? csc[c0 + h + x] : 0u;
#pragma unroll
          for (int x = 0; x < 4; x++) { sw[x] = sg8[v4[x] & 0xffffu]; wq[x] = wl2[(v4[x] >> 22) & 31u]; dn[x] = mt[(v4[x] >> 16) & 31u].y & 0xFFu; }
#pragma unroll
          for (int x = 0; x < 4; x++) {
            const uint32_t v = v4[x];
            const int i = (v >> 16) & 31;
            if ((v >> 31) && i != cur) { if (cur >= 0) flush(); cur = i; }
            const uint32_t hit = (v >> 31) ? ((((v >> 21) & 1u) ? 0xFFu : 0u) ^ sw[x] ^ dn[x]) & 0xFFu : 0u;
            const int whi = ((int)(wq[x].y << 8)) >> 8;   // the signed 24-bit limb
#pragma unroll
            for (int s = 0; s < 8; s++) {
              const uint32_t b = (hit >> s) & 1u;
              alo[s] += __umul24(b, wq[x].x); ahi[s] += __mul24((int)b, whi);
            }
          }
        }
        if (cur >= 0) flush();
      }
      wave_lds_sync();
      // a lane per (state, SNP), eight SNPs per pass: the four-way decision (first maximum, phase.rs:908-921)
      uint32_t any2 = 0, chg8 = 0;   // per state: an improvement / a mask changed
      {
        const int s = lane >> 3;
        const uint32_t od = ms[s], oe0 = ms[8 + s], oep = ms[16 + s];
        uint32_t nd = od, ne0 = oe0, nep = oep;
        for (int p = 0; p < n_pass; p++) {
          const int i = 8 * p + (lane & 7);
          const bool in = i < S;
          const bool live = in && s_live[i & 31] != 0;
          int d_new = (int)((od >> i) & 1u), h_new = ((oe0 >> i) & 1u) ? 0 : (((oep >> i) & 1u) ? 1 : -1);
          bool changed = false, dtie = false;
          if (live && ((act >> s) & 1u)) {
            const long long M = (long long)Macc[s * Sp + i];
            const long long cF = s_cF[i], cW = s_cW[i], cRef = s_cRef[i], cVar = s_cVar[i], het = s_het[i];
            const long long N0 = cF + M + het, N1 = cF + cW - M + het;
            int ch = 0; long long nb = N0;
            if (N1 > nb) { ch = 1; nb = N1; }
            if (cRef > nb) { ch = 2; nb = cRef; }
            if (cVar > nb) { ch = 3; nb = cVar; }
            dtie = (int)(N0 == nb) + (int)(N1 == nb) + (int)(cRef == nb) + (int)(cVar == nb) > 1;   // a tie at the maximum: the first one is kept
            const long long ncur = h_new == 0 ? N0 : (h_new == 1 ? cRef : cVar);
            changed = nb > ncur;
            if (ch == 1) d_new ^= 1;
            h_new = ch <= 1 ? 0 : (ch == 2 ? 1 : -1);
            objl[p] = ch <= 1 ? nb - het : (ch == 2 ? cRef - P.lut.f_homref : cVar - P.lut.f_homvar);
          }
          if (in) Macc[s * Sp + i] = 0;
          const unsigned long long bd = __ballot(in && d_new), b0 = __ballot(in && h_new == 0), bp = __ballot(in && h_new == 1);
          const unsigned long long bc = __ballot(changed), bt = __ballot(dtie);
          const uint32_t sh = 8u * (uint32_t)s, keep = ~(0xFFu << (8 * p));
          nd = (nd & keep) | ((uint32_t)((bd >> sh) & 0xFFull) << (8 * p));
          ne0 = (ne0 & keep) | ((uint32_t)((b0 >> sh) & 0xFFull) << (8 * p));
          nep = (nep & keep) | ((uint32_t)((bp >> sh) & 0xFFull) << (8 * p));
#pragma unroll
          for (int q = 0; q < 8; q++) {
            if ((bc >> (8 * q)) & 0xFFull) any2 |= 1u << q;
            const uint32_t nt8 = (uint32_t)__popcll((bt >> (8 * q)) & 0xFFull);
            n_dtie += nt8; ev_d[q] += nt8;
          }
        }
        nd &= smask; ne0 &= smask; nep &= smask;
        const bool frozen = !((act >> s) & 1u);
        if (frozen) { nd = od; ne0 = oe0; nep = oep; }
        const unsigned long long bm = __ballot(nd != od || ne0 != oe0 || nep != oep);
#pragma unroll
        for (int q = 0; q < 8; q++) if ((bm >> (8 * q)) & 0xFFull) chg8 |= 1u << q;
        wave_lds_sync();
        if ((lane & 7) == 0) { ms[s] = nd; ms[8 + s] = ne0; ms[16 + s] = nep; }
      }
      any2 &= act;
#pragma unroll
      for (int q = 0; q < 8; q++) if (((chg8 & ~any2 & act) >> q) & 1u) { n_step++; ev_s[q]++; }   // only tie changes in this step
      // (!any2: hg_inc = false; else both true)
      hginc = (hginc & ~act) | (any2 & act); hinc |= any2 & act;
      wave_lds_sync();
      iters++;
      act &= hinc | hginc;
      if (iters > 20) act = 0;  // phase.rs:967-972
    }
    // ---- what the restarts leave
    long long tot = objl[0] + objl[1] + objl[2] + objl[3];
    tot += __shfl_xor(tot, 1, 64); tot += __shfl_xor(tot, 2, 64); tot += __shfl_xor(tot, 4, 64);   // lanes 8 s .. 8 s + 7: the objective of restart s
    // the SNPs' masks of the FINAL states
    if (lane < 32) {
      uint32_t dn = 0, h8 = 0, ep = 0;
#pragma unroll
      for (int s = 0; s < 8; s++) { dn |= ((ms[s] >> lane) & 1u) << s; h8 |= ((ms[8 + s] >> lane) & 1u) << s; ep |= ((ms[16 + s] >> lane) & 1u) << s; }
      mt[lane] = make_uint2(0u, dn | (h8 << 8) | (ep << 16));
    }
    wave_lds_sync();
    // signatures of the final configurations (as k4_enum_reg: a hash of the match bits [p == x] of the lane's entries, 32 at a time;
    // compared only among the restarts of one region)
    unsigned long long hs[8];
    {
      uint32_t word[8];
#pragma unroll
      for (int s = 0; s < 8; s++) { hs[s] = 0; word[s] = 0; }
      int nb = 0, widx = 0;
      // (the lane's entries: entry k of its row of every group -- slot 64 x + lane of the groups' part of csr, row perm[64 g + lane], g the group
      // of slot row x --, then the rows of one entry: lane <-> row x = lane + 64 j)
      const int n_grp = (int)gof[nG], n_one = (nS + 63) / 64;
      int gq = 0;
      for (int x = 0; x < n_grp + n_one; x++) {
        const bool one = x >= n_grp;
        const int xs = lane + 64 * (x - n_grp);
        if (!one) while ((int)gof[gq + 1] <= x) gq++;   // (wave-uniform)
        const uint2 v = one ? (xs < nS ? csr[eM + xs] : make_uint2(0, 0)) : csr[64 * x + lane];
        const uint32_t m = v.x >> 24, i = m & 31u;
        const uint32_t y = mt[i].y, dn8 = y & 0xFFu, h8 = (y >> 8) & 0xFFu, ep8 = (y >> 16) & 0xFFu;
        const int srow = (m & 128u) ? (int)perm[one ? nM + xs : 64 * gq + lane] : 0;
        const uint32_t p8 = (m & 32u) ? 0xFFu : 0u, sn8 = sg8[srow];
        const uint32_t match = (m & 128u) ? ((h8 & (p8 ^ sn8 ^ dn8)) | (~h8 & ~(p8 ^ ep8))) & 0xFFu : 0u;
#pragma unroll
        for (int s = 0; s < 8; s++) word[s] |= ((match >> s) & 1u) << nb;
        if (++nb == 32) {
#pragma unroll
          for (int s = 0; s < 8; s++) { hs[s] = mix64(hs[s] ^ (word[s] + (unsigned long long)(lane * 64 + widx + 1) * 0x9E3779B97F4A7C15ULL)); word[s] = 0; }
          nb = 0; widx++;
        }
      }
      if (nb) {
#pragma unroll
        for (int s = 0; s < 8; s++) hs[s] = mix64(hs[s] ^ (word[s] + (unsigned long long)(lane * 64 + widx + 1) * 0x9E3779B97F4A7C15ULL));
      }
    }
#pragma unroll
    for (int s = 0; s < 8; s++) {
      if (s >= nst) break;
      const uint32_t e = e_base + (uint32_t)s;
      const long long total = ((long long)__builtin_amdgcn_readlane((int)(tot >> 32), 8 * s) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)tot, 8 * s);
      // a restart that met a tie at a delta / eta maximum or a tie-only step goes to the repair list (as k4_enum_reg)
      bool redone = false;
      if (redo && ev_d[s] + ev_s[s] != 0 && P.tie_arith >= 3) {
        uint32_t at = 0;
        if (lane == 0) at = atomicAdd(&redo[0], 1u);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        if (at < redo_cap) {
          if (lane == 0) { redo[4 + 2 * at] = (uint32_t)t.slot; redo[5 + 2 * at] = e; }
          n_dtie -= ev_d[s]; n_step -= ev_s[s];   // (not unresolved: the repair pass decides them)
          redone = true;
        }
      }
      const long long seen = __hip_atomic_load(&region_best[t.slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool keep = __builtin_amdgcn_readfirstlane((int)(total >= seen)) != 0;
      if (total > seen && lane == 0 && !redone) (void)__hip_atomic_fetch_max(&region_best[t.slot], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned long long* stp = st_reg + (size_t)e * stw;
      if (keep) {
        const unsigned long long sig = (unsigned long long)wave_sum_ll((long long)hs[s]);
        for (int k = 0; k < nk; k++) {
          const int row = 64 * k + lane;
          const unsigned long long w = __ballot(row < R && ((sg8[min(row, max(R - 1, 0))] >> s) & 1u));
          if (lane == 0) stp[k] = w;
        }
        if (lane == 0) { stp[nk] = (unsigned long long)ms[s] | ((unsigned long long)ms[8 + s] << 32); stp[nk + 1] = (unsigned long long)ms[16 + s]; stp[nk + 2] = sig; }
      }
      if (lane == 0) job_obj[job_base[t.slot] + e] = total;
    }
    wave_lds_sync();
  }
  n_tie_f64 = (uint32_t)wave_sum_ll((long long)n_tie_f64); n_tie_flip = (uint32_t)wave_sum_ll((long long)n_tie_flip); n_tie_unres = (uint32_t)wave_sum_ll((long long)n_tie_unres);
  if (lane == 0) {
    if (n_tie_f64) TIE_COUNT(P.tie_ctr, TIE_SIGMA_F64, (unsigned long long)n_tie_f64);
    if (n_tie_flip) TIE_COUNT(P.tie_ctr, TIE_SIGMA_FLIPS, (unsigned long long)n_tie_flip);
    if (n_tie_unres) TIE_COUNT(P.tie_ctr, TIE_SIGMA_UNRES, (unsigned long long)n_tie_unres);
    if (n_dtie) TIE_COUNT(P.tie_ctr, TIE_DELTA_UNRES, (unsigned long long)n_dtie);
    if (n_step) TIE_COUNT(P.tie_ctr, TIE_STEP_UNRES, (unsigned long long)n_step);
  }
}

// the same tiles for regions whose matrix does not fit the LDS budget: one restart at a time per workgroup.  Every restart leaves
// its objective and -- st_words != nullptr and the region has a span there (st_base >= 0) -- its final state in the layout of the
// other classes (sigma bits | delta < 0, eta == 0 masks | eta == +1 mask | a signature of all of it) for the resolve kernels.
// qrow (2 R doubles per workgroup): cross_optimize's scratch for the complete tie contract (k4_dev.h).
// (mv / macc: the matrix the sweeps read -- global memory, or the repair pass's copy in LDS -- and, with it, <= 32 zeroed LDS words for the
// entry-balanced delta sweep of the plain form; same integers, same decisions)
__device__ __forceinline__ void enum_big_restart(const PhaseDev& P, const RegionDev& rd, int slot, uint32_t e, bool winner, int8_t* base, double* qrow,
                                                 long long* red, const long long* wl, unsigned long long* s_sig, const int64_t* __restrict__ job_base,
                                                 long long* __restrict__ job_obj, const int64_t* __restrict__ st_base, unsigned long long* __restrict__ st_words,
                                                 const MatView& mv, unsigned long long* macc = nullptr) {
  int8_t* sg = base; int8_t* dl = base + rd.R; int8_t* et = dl + rd.S;
  const int8_t* vt = P.snp_vt + rd.snp_off;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t nk = (uint32_t)(rd.R + 63) / 64, sw = enum_state_words((uint32_t)rd.R);
  const bool keep = !winner && st_words && st_base[slot] >= 0;
  for (int i = threadIdx.x; i < rd.S; i += blockDim.x) { dl[i] = ((e >> i) & 1u) ? -1 : 1; et[i] = init_genotype(vt[i]); }
  const uint64_t ctr0 = (uint64_t)rd.S + (uint64_t)rd.R + (uint64_t)e * (uint64_t)rd.R;
  for (int row = threadIdx.x; row < rd.R; row += blockDim.x) sg[row] = u01(rd.seed, ctr0 + row) < 0.5 ? -1 : 1;
  __syncthreads();
  const long long obj = cross_optimize(P, rd, mv, sg, dl, et, false, true, red, wl, macc, 32, nullptr, 0, nullptr, nullptr, nullptr, nullptr, qrow);
  if (winner) {
    for (int i = threadIdx.x; i < rd.S; i += blockDim.x) { P.st_delta[rd.snp_off + i] = dl[i]; P.st_eta[rd.snp_off + i] = et[i]; }
    for (int row = threadIdx.x; row < rd.R; row += blockDim.x) P.st_sigma[rd.sig_off + row] = sg[row];
    if (threadIdx.x == 0) P.st_obj[slot] = obj;
  } else {
    if (threadIdx.x == 0) job_obj[job_base[slot] + e] = obj;
    if (keep) {
      unsigned long long* dst = st_words + st_base[slot] + (size_t)e * sw;
      unsigned long long h = 0;
      for (uint32_t w0 = (uint32_t)wave; w0 < nk; w0 += LCR_BLOCK / 64) {   // a wave per 64 rows
        const uint32_t row = 64u * w0 + (uint32_t)lane;
        const unsigned long long word = __ballot(row < (uint32_t)rd.R && sg[row] < 0);
        if (lane == 0) dst[w0] = word;
        h ^= mix64(word + 0x9E3779B97F4A7C15ull * (w0 + 1));
      }
      if (lane == 0) s_sig[wave] = h;
      __syncthreads();
      if (wave == 0) {   // (S <= 31: a restart index has 32 bits)
        const bool in = lane < rd.S;
        const unsigned long long dneg = __ballot(in && dl[in ? lane : 0] < 0), eta0 = __ballot(in && et[in ? lane : 0] == 0), etap = __ballot(in && et[in ? lane : 0] == 1);
        if (lane == 0) {
          const unsigned long long w0 = (dneg & 0xffffffffull) | (eta0 << 32), w1 = etap & 0xffffffffull;
          unsigned long long sig = mix64(w0 + 1) ^ mix64(w1 + 0x5851F42D4C957F2Dull);
          for (int w = 0; w < LCR_BLOCK / 64; w++) sig ^= s_sig[w];
          dst[nk] = w0; dst[nk + 1] = w1; dst[nk + 2] = sig;
        }
      }
    }
  }
  __syncthreads();
}
__global__ void __launch_bounds__(LCR_BLOCK)
k4_enum_big(PhaseDev P, const EnumSpan* __restrict__ spans, int32_t n_spans, uint32_t per, const int64_t* __restrict__ job_base,
            long long* __restrict__ job_obj, const uint32_t* __restrict__ win_e, const int64_t* __restrict__ st_base,
            unsigned long long* __restrict__ st_words, double* __restrict__ qrow, int64_t qrow_stride) {
  __shared__ long long red[LCR_BLOCK / 64];
  __shared__ long long wl[32];
  __shared__ unsigned long long s_sig[LCR_BLOCK / 64];
  const EnumTile t = enum_tile_of(P, spans, n_spans, per, win_e != nullptr);
  const RegionDev rd = P.reg[t.slot];
  load_w(P, wl);
  int8_t* base = P.scratch + (size_t)blockIdx.x * P.scratch_stride;
  double* qr = qrow ? qrow + (size_t)blockIdx.x * qrow_stride : nullptr;
  const uint32_t ne = win_e ? 1u : t.ne;
  for (uint32_t k = 0; k < ne; k++)
    enum_big_restart(P, rd, t.slot, win_e ? win_e[t.slot] : t.e0 + k, win_e != nullptr, base, qr, red, wl, s_sig, job_base, job_obj, st_base, st_words, global_view(P, rd));
}
// the repair pass of the LDS classes: the restarts on a list (k4_enum_reg: those that met a tie of class 2 / 4) once more, through the
// one-workgroup cross_optimize with the complete tie contract; objective and final state overwrite what the fast kernel left (the
// state's signature is the global-memory class's: never equal to a fast kernel's, so the resolve kernel forms its f64 sum itself)
__global__ void __launch_bounds__(LCR_BLOCK)
k4_enum_redo(PhaseDev P, const uint32_t* __restrict__ redo, uint32_t redo_cap, int8_t* __restrict__ scratch, int32_t scratch_stride, double* __restrict__ qrow,
             int64_t qrow_stride, const int64_t* __restrict__ job_base, long long* __restrict__ job_obj, const int64_t* __restrict__ st_base,
             unsigned long long* __restrict__ st_words, uint32_t lds_bytes) {
  __shared__ long long red[LCR_BLOCK / 64];
  __shared__ long long wl[32];
  __shared__ unsigned long long s_sig[LCR_BLOCK / 64];
  __shared__ unsigned long long macc[32];
  extern __shared__ __attribute__((aligned(16))) uint8_t redo_dyn[];   // lds_bytes: working state + the region's matrix, when they fit
  const uint32_t n = min(redo[0], redo_cap);
  if (blockIdx.x >= n) return;
  load_w(P, wl);
  if (threadIdx.x < 32) macc[threadIdx.x] = 0;
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
    const int slot = (int)redo[4 + 2 * i];
    const RegionDev rd = P.reg[slot];
    const uint32_t E = (uint32_t)P.prow_ptr[rd.rp_off + rd.R];
    const uint32_t st_b = ((uint32_t)(rd.R + 2 * rd.S) + 63u) & ~63u;
    __syncthreads();   // (the previous restart's readers of the LDS image are done)
    // Round 6: on the C4 share this pass was the longest kernel of the stage's tail (0.81 ms: the one-workgroup cross_optimize sweeping the
    // matrix in global memory, a wave per SNP), and the tail ended after the next batch's pileup: with the matrix and the state in LDS and
    // the entry-balanced delta sweep a restart is several times shorter.  Two call sites: the sweeps' pointers keep ONE provenance each.
    if (st_b + matview_bytes((uint32_t)rd.R, (uint32_t)rd.S, E) <= lds_bytes)
      enum_big_restart(P, rd, slot, redo[5 + 2 * i], false, reinterpret_cast<int8_t*>(redo_dyn), qrow + (size_t)blockIdx.x * qrow_stride,
                       red, wl, s_sig, job_base, job_obj, st_base, st_words, stage_view(P, rd, redo_dyn + st_b, E), macc);
    else
      enum_big_restart(P, rd, slot, redo[5 + 2 * i], false, scratch + (size_t)blockIdx.x * scratch_stride, qrow + (size_t)blockIdx.x * qrow_stride,
                       red, wl, s_sig, job_base, job_obj, st_base, st_words, global_view(P, rd));
  }
}

// Winner of each enumeration region of the global-memory class: `prob > largest_prob` (phase.rs:1113-1119) over the restarts in
// order.  Decided on the exact objectives; among the restarts of MAXIMAL objective whose final configurations differ (signatures),
// on the reference-order f64 sum of cal_overall_probability (phase.rs:257-276) of every distinct configuration: the first restart
// with the largest sum wins (strictly-greater-replaces).  A workgroup per region; the f64 sum of a configuration: a thread per row
// writes the rows' terms in entry order, wave 0 adds them up one by one (64 terms per load, v_readlane per add).  The winner's
// index goes to win_e: k4_enum_big runs that restart once more and leaves its state in the region's result slots.
// Not resolved (counted, first maximum kept): states not kept (st_base < 0: beyond the memory budget), more than TLB_CAP maxima.
constexpr uint32_t TLB_CAP = 1024;
__global__ void __launch_bounds__(LCR_BLOCK)
k4_enum_resolve_big(PhaseDev P, const EnumSpan* __restrict__ spans, const int64_t* __restrict__ job_base, const long long* __restrict__ job_obj,
                    const int64_t* __restrict__ st_base, const unsigned long long* __restrict__ st_words, uint32_t* __restrict__ win_e,
                    double* __restrict__ terms, int64_t terms_stride) {
  __shared__ uint32_t tl[TLB_CAP];
  __shared__ unsigned long long sig[TLB_CAP];
  __shared__ double sum[TLB_CAP];
  __shared__ uint16_t rep[TLB_CAP];
  __shared__ double lut[64];
  __shared__ long long s_best[LCR_BLOCK / 64];
  __shared__ uint32_t s_wcnt[LCR_BLOCK / 64], s_ntied, s_first, s_dif;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int slot = spans[blockIdx.x].slot;
  const RegionDev rd = P.reg[slot];
  const int R = rd.R, S = rd.S;
  const uint32_t n_jobs = 1u << S, nk = (uint32_t)(R + 63) / 64, sw = enum_state_words((uint32_t)R);
  const long long* o = job_obj + job_base[slot];
  const bool have_st = st_words && st_base[slot] >= 0;
  const unsigned long long* st = st_words + (have_st ? st_base[slot] : 0);
  if (tid < 64) lut[tid] = (tid & 31) < 31 ? (tid < 32 ? P.lut64->le[tid] : P.lut64->l1e[tid - 32]) : 0.0;
  long long best = LLONG_MIN;
  for (uint32_t e = tid; e < n_jobs; e += LCR_BLOCK) { const long long v = o[e]; if (v > best) best = v; }
  for (int d = 32; d >= 1; d >>= 1) { const long long ob = __shfl_xor(best, d, 64); if (ob > best) best = ob; }
  if (lane == 0) s_best[wave] = best;
  if (tid == 0) { s_ntied = 0; s_first = 0xffffffffu; s_dif = 0; }
  __syncthreads();
  for (int w = 0; w < LCR_BLOCK / 64; w++) if (s_best[w] > best) best = s_best[w];
  for (uint32_t p0 = 0; p0 < n_jobs; p0 += LCR_BLOCK) {   // the maxima in ascending order (ranks by ballot, wave offsets through LDS)
    const uint32_t e = p0 + (uint32_t)tid;
    const bool hit = e < n_jobs && o[e] == best;
    const unsigned long long m = __ballot(hit);
    if (lane == 0) s_wcnt[wave] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t off = s_ntied, tot = 0;
    for (int w = 0; w < LCR_BLOCK / 64; w++) { if (w < wave) off += s_wcnt[w]; tot += s_wcnt[w]; }
    const uint32_t at = off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (hit && at < TLB_CAP) tl[at] = e;
    if (hit && at == 0) s_first = e;
    __syncthreads();
    if (tid == 0) s_ntied += tot;
    __syncthreads();
  }
  const uint32_t n_all = s_ntied, n_tied = min(n_all, TLB_CAP), first = s_first;
  uint32_t win = first;
  if (n_all > 1) {
    if (!have_st || n_all > TLB_CAP || P.tie_arith < 1) { if (tid == 0) TIE_COUNT(P.tie_ctr, TIE_BEST_UNRES, 1ull); }
    else {
      for (uint32_t j = tid; j < n_tied; j += LCR_BLOCK) sig[j] = st[(size_t)tl[j] * sw + nk + 2];
      __syncthreads();
      for (uint32_t j = tid; j < n_tied; j += LCR_BLOCK) {   // the earliest restart with the same configuration
        uint32_t r = j;
        for (uint32_t i = 0; i < j; i++) if (sig[i] == sig[j]) { r = i; break; }
        rep[j] = (uint16_t)r;
        if (r != 0) s_dif = 1;
      }
      __syncthreads();
      if (s_dif) {
        if (tid == 0) TIE_COUNT(P.tie_ctr, TIE_BEST_F64, 1ull);
        const MatView mv = global_view(P, rd);
        const uint32_t E = (uint32_t)mv.rp[R];
        double* tb = terms + (size_t)blockIdx.x * terms_stride;
        for (uint32_t j = 0; j < n_tied; j++) {
          if (rep[j] != j) continue;   // (uniform: LDS)
          const unsigned long long* cf = st + (size_t)tl[j] * sw;
          const unsigned long long rm0 = cf[nk];
          const uint32_t dneg = (uint32_t)rm0, eta0 = (uint32_t)(rm0 >> 32), etap = (uint32_t)cf[nk + 1];
          // match = [p == x]: het site: p == sigma * delta; hom site: p == eta (k4_enum_resolve's full_sum)
          const uint32_t cm = (dneg & eta0) | (~etap & ~eta0);
          for (int row = tid; row < R; row += LCR_BLOCK) {
            const uint32_t sneg = 0u - (uint32_t)((cf[row >> 6] >> (row & 63)) & 1ull);
            for (int x = mv.rp[row]; x < mv.rp[row + 1]; x++) {
              const uint32_t i = (uint32_t)mv.pc[x], v = mv.pv[x];
              const uint32_t pbit = (v >> 5) & 1u, q = v & 31u;
              const uint32_t match = (pbit ^ (cm >> i) ^ ((sneg & eta0) >> i)) & 1u;
              tb[x] = lut[(match << 5) + q];
            }
          }
          __threadfence_block();
          __syncthreads();
          if (wave == 0) {
            double acc = 0.0;
            for (uint32_t b0 = 0; b0 < E; b0 += 64) {
              const double tv = b0 + (uint32_t)lane < E ? tb[b0 + lane] : 0.0;
              const int lo = (int)__double2loint(tv), hi = (int)__double2hiint(tv);
              const uint32_t n = min(64u, E - b0);
#pragma unroll 8
              for (uint32_t kk = 0; kk < 64; kk++)
                if (kk < n) acc += __hiloint2double(__builtin_amdgcn_readlane(hi, (int)kk), __builtin_amdgcn_readlane(lo, (int)kk));
            }
            if (lane == 0) sum[j] = acc;
          }
          __syncthreads();
        }
        if (tid == 0) {   // strictly-greater-replaces over the maxima in order (phase.rs:1117)
          double bs = sum[0];
          for (uint32_t j = 1; j < n_tied; j++) { const double sj = sum[rep[j]]; if (sj > bs) { bs = sj; win = tl[j]; } }
          s_first = win;
        }
        __syncthreads();
        win = s_first;
      }
    }
  }
  if (tid == 0) win_e[slot] = win;
}
}  // namespace

void launch_k4_enum_reg(int ck, unsigned n_blocks, size_t dyn_lds, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, int32_t n_spans,
                        uint32_t per, const int64_t* job_base, long long* job_obj, const int64_t* st_base, unsigned long long* st_words,
                        long long* region_best, uint32_t* redo, uint32_t redo_cap) {
  const dim3 blk(64 * ENUM_WAVES);
  if (ck < 0) hipLaunchKernelGGL(k4_enum_bits, dim3(n_blocks), blk, dyn_lds, s, P, spans, n_spans, per, job_base, job_obj, st_base, st_words, region_best, redo, redo_cap);
  else if (ck == 32) hipLaunchKernelGGL(k4_enum_reg<32>, dim3(n_blocks), blk, dyn_lds, s, P, spans, n_spans, per, job_base, job_obj, st_base, st_words, region_best, redo, redo_cap);
  else hipLaunchKernelGGL(k4_enum_reg<0>, dim3(n_blocks), blk, dyn_lds, s, P, spans, n_spans, per, job_base, job_obj, st_base, st_words, region_best, redo, redo_cap);
}
hipError_t k4_set_dyn_lds_once(const void* fn, int bytes, int slot);   // (k4_grid.hip: once per device)
void launch_k4_enum_redo(unsigned n_blocks, hipStream_t s, const PhaseDev& P, const uint32_t* redo, uint32_t redo_cap, int8_t* scratch, int32_t scratch_stride,
                         double* qrow, int64_t qrow_stride, const int64_t* job_base, long long* job_obj, const int64_t* st_base, unsigned long long* st_words,
                         uint32_t lds_bytes) {
  lds_bytes = std::min<uint32_t>(lds_bytes, 128 * 1024);
  if (lds_bytes > 32 * 1024 && k4_set_dyn_lds_once(reinterpret_cast<const void*>(&k4_enum_redo), 128 * 1024, 6) != hipSuccess) { (void)hipGetLastError(); lds_bytes = 32 * 1024; }
  hipLaunchKernelGGL(k4_enum_redo, dim3(n_blocks), dim3(LCR_BLOCK), lds_bytes, s, P, redo, redo_cap, scratch, scratch_stride, qrow, qrow_stride, job_base, job_obj, st_base, st_words, lds_bytes);
}
void launch_k4_enum_resolve(unsigned n_regions, size_t dyn_lds, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, const int64_t* job_base,
                            const long long* job_obj, const int64_t* st_base, const unsigned long long* st_words) {
  hipLaunchKernelGGL(k4_enum_resolve, dim3(n_regions), dim3(64 * ENUM_WAVES), dyn_lds, s, P, spans, job_base, job_obj, st_base, st_words);
}
void launch_k4_enum_big(unsigned n_blocks, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, int32_t n_spans, uint32_t per,
                        const int64_t* job_base, long long* job_obj, const uint32_t* win_e, const int64_t* st_base, unsigned long long* st_words,
                        double* qrow, int64_t qrow_stride) {
  hipLaunchKernelGGL(k4_enum_big, dim3(n_blocks), dim3(LCR_BLOCK), 0, s, P, spans, n_spans, per, job_base, job_obj, win_e, st_base, st_words, qrow, qrow_stride);
}
void launch_k4_enum_resolve_big(int32_t n, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, const int64_t* job_base, const long long* job_obj,
                                const int64_t* st_base, const unsigned long long* st_words, uint32_t* win_e, double* terms, int64_t terms_stride) {
  hipLaunchKernelGGL(k4_enum_resolve_big, dim3((unsigned)n), dim3(LCR_BLOCK), 0, s, P, spans, job_base, job_obj, st_base, st_words, win_e, terms, terms_stride);
}
