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

// tiles of restarts of regions whose per-lane share is <= CK entries (host decides); win_e != nullptr:
// re-run restart win_e[slot] of each tile's region and store its state.
template <int CK>
__global__ void __launch_bounds__(64 * ENUM_WAVES, 3)   // (three waves per SIMD: <= 168 VGPRs)
k4_enum_reg(PhaseDev P, const EnumSpan* __restrict__ spans, int32_t n_spans, uint32_t per, const int64_t* __restrict__ job_base,
            long long* __restrict__ job_obj, const uint32_t* __restrict__ win_e, uint32_t* __restrict__ tiles_done) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  const EnumTile t = enum_tile_of(P, spans, n_spans, per, win_e != nullptr);
  const RegionDev rd = P.reg[t.slot];
  const int R = rd.R, S = rd.S;
  const uint32_t E = (uint32_t)P.prow_ptr[rd.rp_off + R];
  const EnumLayout L = enum_layout(R, E);
  uint2* wl2 = (uint2*)lds;
  uint2* csr = (uint2*)(lds + L.csr);
  uint32_t* csc = (uint32_t*)(lds + L.csc);
  uint16_t* rp = (uint16_t*)(lds + L.rp); uint16_t* first_row = (uint16_t*)(lds + L.first_row);
  const int tid = threadIdx.x, nt = blockDim.x;
  const uint32_t c = enum_chunk(E);
  // ---- stage the region (once per workgroup)
  if (tid < 32) {
    const long long w = tid < 31 ? P.lut.f1e[tid] - P.lut.fe[tid] : 0;
    wl2[tid] = make_uint2((uint32_t)w & 0x7fffffu, (uint32_t)(w >> 23) & 0xffffffu);   // w = hi * 2^23 + lo, hi signed (w < 0 for q <= 3)
  }
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
  const uint32_t ne = win_e ? 1u : t.ne;
  // one restart by this wave: its objective goes to job_obj[], or -- the winner's re-run -- its state to the region's slot
  auto run_restart = [&](const uint32_t e_in, const bool mat_in) {
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)e_in);   // (wave-uniform: keep them in SGPRs)
    const bool materialise = __builtin_amdgcn_readfirstlane((int)mat_in) != 0;
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
        unsigned long long fm = 0;
        auto sig_one = [&](uint32_t v0, uint32_t v1) {
          const uint32_t m = v0 >> 24, i = m & 31u, roff = v1 >> 24;
          const uint32_t sneg = (uint32_t)(win >> roff);
          const uint32_t use = (m >> 7) & (eta0 >> i) & 1u;                 // het sites only
          const uint32_t hit = ((m >> 5) ^ sneg ^ (dneg >> i)) & use;       // p == sigma * delta
          const uint32_t mis = hit ^ use;
          alo += __mul24((int)hit, (int)v0) - __mul24((int)mis, (int)v0);   // A - B of phase.rs:824-862
          ahi += __mul24((int)hit, (int)v1) - __mul24((int)mis, (int)v1);
          const bool end = (m >> 6) & 1u;
          // sign of ahi * 2^23 + alo: fold alo's carry into ahi, the remainder is in [0, 2^23)
          if (end && ahi + (alo >> 23) < 0) fm |= 1ull << roff;
          alo = end ? 0 : alo; ahi = end ? 0 : ahi;
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
        const bool any = __ballot(fm != 0) != 0;
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
      bool changed = false;
      int d_new = (dneg >> lane) & 1u, h_new = ((eta0 >> lane) & 1u) ? 0 : (((etap >> lane) & 1u) ? 1 : -1);
      if (live) {
        const long long M = (long long)Macc[lane];
        Macc[lane] = 0;
        const long long N0 = cF + M + het, N1 = cF + cW - M + het;
        int ch = 0; long long nb = N0;                       // first maximum (phase.rs:908-921)
        if (N1 > nb) { ch = 1; nb = N1; }
        if (cRef > nb) { ch = 2; nb = cRef; }
        if (cVar > nb) { ch = 3; nb = cVar; }
        const long long ncur = h_new == 0 ? N0 : (h_new == 1 ? cRef : cVar);
        changed = nb > ncur;
        if (ch == 1) d_new ^= 1;
        h_new = ch <= 1 ? 0 : (ch == 2 ? 1 : -1);
        obj_i = ch <= 1 ? nb - het : (ch == 2 ? cRef - P.lut.f_homref : cVar - P.lut.f_homvar);
      }
      dneg = (uint32_t)__ballot(lane < S && d_new);
      eta0 = (uint32_t)__ballot(lane < S && h_new == 0);
      etap = (uint32_t)__ballot(lane < S && h_new == 1);
      const bool any2 = __ballot(changed) != 0;
      wave_lds_sync();
      if (!any2) hg_inc = false; else { hg_inc = true; h_inc = true; }
      if (++iters > 20) break;  // phase.rs:967-972
    }
    // objective (phase.rs:257-276) = sum over phase entries of fe + hit * w = sum_i (F_i + hits_i) over live SNPs
    const long long total = wave_sum_ll_dpp(obj_i);
    if (materialise) {
      if (lane < S) {
        P.st_delta[rd.snp_off + lane] = (int8_t)(((dneg >> lane) & 1u) ? -1 : 1);
        P.st_eta[rd.snp_off + lane] = (int8_t)(((eta0 >> lane) & 1u) ? 0 : (((etap >> lane) & 1u) ? 1 : -1));
      }
      for (int k = 0; k < nk; k++) {
        const int row = lane + 64 * k;
        if (row < R) P.st_sigma[rd.sig_off + row] = (int8_t)(((sgb[k] >> lane) & 1ull) ? -1 : 1);
      }
      if (lane == 0) P.st_obj[t.slot] = total;
    } else if (lane == 0) __hip_atomic_store(&job_obj[job_base[t.slot] + e], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (device-coherent: read by the region's last tile)
    wave_lds_sync();
  };
  // Pass 0: this tile's restarts.  Then the tile that completes its region picks the winner (first maximum, `prob >
  // largest_prob`, phase.rs:1113-1119) and -- pass 1, wave 0 -- runs that restart once more to leave its state: the matrix
  // is still staged here, and neither a pick kernel nor a second launch sits between the enumeration and the post-phase
  // kernel.  The objectives were stored device-coherently and every wave's stores are acknowledged before the barrier
  // that lets thread 0 count the tile.  (One call site for both passes: a second inlined copy costs the 32-entry
  // instantiation its third wave per SIMD.)
  __shared__ uint32_t s_last, s_win;
  __shared__ long long s_best[ENUM_WAVES];
  __shared__ uint32_t s_be[ENUM_WAVES];
  for (int pass = 0; pass < 2; pass++) {
    const uint32_t n_run = pass == 0 ? ne : (wave == 0 ? 1u : 0u);
    for (uint32_t kk = pass == 0 ? wave : 0u; kk < n_run; kk += ENUM_WAVES)
      run_restart(pass == 1 ? s_win : (win_e ? win_e[t.slot] : t.e0 + kk), pass == 1 || win_e != nullptr);
    if (pass == 1 || win_e || !tiles_done) break;
    __syncthreads();
    const uint32_t n_jobs = 1u << S;
    if (tid == 0) s_last = atomicAdd(&tiles_done[t.slot], 1u) == (n_jobs + per - 1) / per - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) break;
    const long long* o = job_obj + job_base[t.slot];
    long long best = LLONG_MIN; uint32_t be = 0xffffffffu;
    for (uint32_t e = tid; e < n_jobs; e += nt) {
      const long long v = __hip_atomic_load(&o[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (v > best) { best = v; be = e; }   // (ascending e per thread: the first maximum of its share)
    }
    for (int d = 32; d >= 1; d >>= 1) {
      const long long ob = __shfl_xor(best, d, 64); const uint32_t oe = __shfl_xor(be, d, 64);
      if (ob > best || (ob == best && oe < be)) { best = ob; be = oe; }
    }
    if (lane == 0) { s_best[wave] = best; s_be[wave] = be; }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < ENUM_WAVES; w++) if (s_best[w] > best || (s_best[w] == best && s_be[w] < be)) { best = s_best[w]; be = s_be[w]; }
      s_win = be;
    }
    __syncthreads();
  }
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

// winner of each enumeration region: first maximum over e (`prob > largest_prob`, phase.rs:1113-1119)
__global__ void __launch_bounds__(64) k4_enum_pick(const int32_t* __restrict__ slots, int32_t n, const RegionDev* __restrict__ reg,
                                                    const int64_t* __restrict__ job_base, const long long* __restrict__ job_obj,
                                                    uint32_t* __restrict__ win_e) {
  const int k = blockIdx.x;
  if (k >= n) return;
  const int slot = slots[k];
  const uint32_t nj = 1u << reg[slot].S;
  const long long* o = job_obj + job_base[slot];
  long long best = LLONG_MIN; uint32_t be = 0xffffffffu;
  for (uint32_t e = threadIdx.x; e < nj; e += 64) { const long long v = o[e]; if (v > best) { best = v; be = e; } }
  for (int d = 32; d >= 1; d >>= 1) {
    const long long ob = __shfl_xor(best, d, 64); const uint32_t oe = __shfl_xor(be, d, 64);
    if (ob > best || (ob == best && oe < be)) { best = ob; be = oe; }
  }
  if (threadIdx.x == 0) win_e[slot] = be;
}
}  // namespace

void launch_k4_enum_reg(int ck, unsigned n_blocks, size_t dyn_lds, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, int32_t n_spans,
                        uint32_t per, const int64_t* job_base, long long* job_obj, const uint32_t* win_e, uint32_t* done) {
  const dim3 blk(64 * ENUM_WAVES);
  if (ck == 32) hipLaunchKernelGGL(k4_enum_reg<32>, dim3(n_blocks), blk, dyn_lds, s, P, spans, n_spans, per, job_base, job_obj, win_e, done);
  else hipLaunchKernelGGL(k4_enum_reg<0>, dim3(n_blocks), blk, dyn_lds, s, P, spans, n_spans, per, job_base, job_obj, win_e, done);
}
void launch_k4_enum_big(unsigned n_blocks, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, int32_t n_spans, uint32_t per,
                        const int64_t* job_base, long long* job_obj, const uint32_t* win_e) {
  hipLaunchKernelGGL(k4_enum_big, dim3(n_blocks), dim3(LCR_BLOCK), 0, s, P, spans, n_spans, per, job_base, job_obj, win_e);
}
void launch_k4_enum_pick(int32_t n, hipStream_t s, const int32_t* slots, const RegionDev* reg, const int64_t* job_base, const long long* job_obj,
                         uint32_t* win_e) {
  hipLaunchKernelGGL(k4_enum_pick, dim3((unsigned)n), dim3(64), 0, s, slots, n, reg, job_base, job_obj, win_e);
}
