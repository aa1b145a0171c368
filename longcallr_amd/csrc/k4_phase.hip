// k4_phase.hip — K4: haplotype phasing optimiser on gfx950 + its host control.
//
// Replaces SNPFrag::phase (reference src/phase.rs:1087-1296) with its kernels cross_optimize
// (phase.rs:810-976) and the probability functions phase.rs:32-49,77-96,128-176,257-276, plus the
// post-phase sequence of src/thread.rs:168-201 (snpfrags.rs:191-733) as a host epilogue.
//
// Device side.  One workgroup runs one complete cross_optimize (alternating sigma / delta-eta
// Jacobi steps until neither improves, <= 21 iterations) on one region's phase matrix:
//   * sigma step: one thread per read row over the CSR slice,
//   * delta/eta step: one wave64 per SNP column over the CSC mirror, wave-reduced,
//   * objective: block reduction.
// Decision arithmetic is exact: every emission term log10(eps_q) / log10(1-eps_q) comes from a
// 31-entry table in fixed point (scale 2^40, int64), so sums are order-free and every comparison
// the reference makes on f64 ratio scores (q < qn, argmax q1..q4, prob > largest_prob) becomes an
// integer comparison of the log sums (the ratios 1 - A/D share a negative denominator D).  See
// DESIGN.md "Decision arithmetic" for why this equals the reference except on rounding-noise ties.
//   * S <= max_enum_snps: all 2^S enumeration restarts (phase.rs:1097-1122) run as independent
//     workgroups in one launch; the winner (first maximum, as `prob > largest_prob`) is re-run to
//     materialise its state.
//   * S  > max_enum_snps: the sequential chain (phase.rs:1123-1233) runs inside one workgroup per
//     region: launch A = first cross_optimize; host does the LD-block flip pass (sum-of-ratios f64
//     decision, phase.rs:1298-1394); launch B = all perturbation rounds with best-state tracking.
// rand::thread_rng() is replaced by a counter-based generator evaluated at the draw index the
// reference's call order implies, so restarts can run in parallel.
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <atomic>
#include <functional>
#include <thread>

#include "lcr_phase_host.h"

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
inline uint64_t region_seed(uint64_t seed, int64_t start0) { return mix64(seed + 0xD1B54A32D192ED03ULL * (uint64_t)(start0 + 1)); }

struct RegionDev {
  int32_t R, S;          // phasing rows, candidates
  int32_t rp_off;        // prow_ptr offset (R+1 entries)
  int32_t cp_off;        // ccol_ptr offset (S+1 entries)
  int64_t e_off;         // offset of this region's entries in pcol/pval and crow/cval
  int32_t sig_off;       // offset into per-row state arrays
  int32_t snp_off;       // offset into per-SNP arrays
  uint64_t seed;
  long long f_total;     // sum of fe[q] over all phase entries (the sigma/delta independent part of the objective)
};

struct PhaseDev {
  const RegionDev* reg;
  const int32_t* prow_ptr; const int32_t* pcol; const uint8_t* pval;
  const int32_t* ccol_ptr; const int32_t* crow; const uint8_t* cval;
  const uint8_t* snp_fp; const int8_t* snp_vt; const uint8_t* snp_cons;
  const long long* snp_const;  // per SNP: F = sum fe, W = sum w, Cref = sum (p==+1 ? f1e : fe), Cvar = sum (p==-1 ? f1e : fe)
  int8_t* st_sigma; int8_t* st_delta; int8_t* st_eta; long long* st_obj;  // per region best / result state
  int8_t* scratch; int32_t scratch_stride;                                // per block working state
  int32_t lds_state;                                                      // 1: working state lives in dynamic LDS
  PhaseLutDev lut;
};

__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
  return v;
}

// one cross_optimize (phase.rs:810-976); returns the exact objective (phase.rs:257-276) to all threads.
// Every emission term is fe[q] + hit * w[q] with w[q] = f1e[q] - fe[q] > 0 and hit = [p == x]
// (aki, phase.rs:32-49), so per row / column only the data dependent sum of w over the hits is
// accumulated; the sigma/delta independent parts are per-SNP constants (PhaseDev::snp_const).
__device__ long long cross_optimize(const PhaseDev& P, const RegionDev& rd, int8_t* sg, int8_t* dl, int8_t* et,
                                    bool keep_conserved, bool with_genotype, long long* red, const long long* wl) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const int32_t* rp = P.prow_ptr + rd.rp_off;
  const int32_t* pc = P.pcol + rd.e_off;
  const uint8_t* pv = P.pval + rd.e_off;
  const int32_t* cp = P.ccol_ptr + rd.cp_off;
  const int32_t* cr = P.crow + rd.e_off;
  const uint8_t* cv = P.cval + rd.e_off;
  const uint8_t* fp = P.snp_fp + rd.snp_off;
  const uint8_t* cons = P.snp_cons + rd.snp_off;
  const long long* sc = P.snp_const + 4ll * rd.snp_off;
  bool hg_inc = true, h_inc = true;
  int iters = 0;
  while (hg_inc | h_inc) {
    // ---- sigma step (phase.rs:824-862): A - B = sum over het sites of (+w if p == sigma*delta else -w);
    //      flip every row with A < B (sites with eta != 0 contribute equally to both)
    int any = 0;
    for (int row = tid; row < rd.R; row += blockDim.x) {
      const int s = sg[row];
      long long diff = 0;
      for (int e = rp[row]; e < rp[row + 1]; e++) {
        const int i = pc[e];
        const uint8_t v = pv[e];
        if (et[i] == 0) { const long long w = wl[v & 31]; diff += (((v & 32) ? 1 : -1) == s * dl[i]) ? w : -w; }
      }
      if (diff < 0) { sg[row] = (int8_t)(-s); any = 1; }
    }
    any = __syncthreads_or(any);
    if (!any) h_inc = false; else { h_inc = true; hg_inc = true; }
    // ---- delta/eta step (phase.rs:872-959): per SNP the best of (d,0) (-d,0) (d,+1) (d,-1)
    any = 0;
    for (int i = wave; i < rd.S; i += nw) {
      if (!fp[i]) continue;
      if (keep_conserved && cons[i]) continue;
      const int c0 = cp[i], c1 = cp[i + 1];
      if (c1 == c0) continue;
      const int d = dl[i], h = et[i];
      long long M = 0;  // sum of w over the entries with p == sigma * d
      for (int e = c0 + lane; e < c1; e += 64) {
        const uint8_t v = cv[e];
        if (((v & 32) ? 1 : -1) == sg[cr[e]] * d) M += wl[v & 31];
      }
      M = wave_sum_ll(M);
      if (lane == 0) {
        const long long het = P.lut.f_het0 - (long long)(c1 - c0) * P.lut.f_log2;  // phase.rs:136-144
        const long long F = sc[4 * i], W = sc[4 * i + 1];
        long long N[4] = {F + M + het, F + W - M + het, sc[4 * i + 2] + P.lut.f_homref, sc[4 * i + 3] + P.lut.f_homvar};
        int ch;
        if (with_genotype) { ch = 0; for (int t = 1; t < 4; t++) if (N[t] > N[ch]) ch = t; }   // phase.rs:908-921
        else if (h == 0) ch = N[1] > N[0] ? 1 : 0;                                                // phase.rs:923-930
        else ch = N[3] > N[2] ? 3 : 2;                                                            // phase.rs:931-938
        const int cur = h == 0 ? 0 : (h == 1 ? 2 : 3);
        if (N[ch] > N[cur]) any = 1;
        dl[i] = (int8_t)(ch == 1 ? -d : d);
        et[i] = (int8_t)(ch <= 1 ? 0 : (ch == 2 ? 1 : -1));
      }
    }
    any = __syncthreads_or(any);
    if (!any) hg_inc = false; else { hg_inc = true; h_inc = true; }
    if (++iters > 20) break;  // phase.rs:967-972
  }
  // ---- objective (phase.rs:257-276) = f_total + sum of w over the hits
  long long acc = 0;
  for (int row = tid; row < rd.R; row += blockDim.x) {
    const int s = sg[row];
    for (int e = rp[row]; e < rp[row + 1]; e++) {
      const int i = pc[e];
      const uint8_t v = pv[e];
      const int x = et[i] == 0 ? s * dl[i] : et[i];
      if (((v & 32) ? 1 : -1) == x) acc += wl[v & 31];
    }
  }
  acc = wave_sum_ll(acc);
  __syncthreads();
  if (lane == 0) red[wave] = acc;
  __syncthreads();
  long long total = rd.f_total;
  for (int w = 0; w < nw; w++) total += red[w];
  __syncthreads();
  return total;
}

// w[q] = f1e[q] - fe[q] into LDS (dynamic indexing of a kernel-argument table would go through memory)
__device__ __forceinline__ void load_w(const PhaseDev& P, long long* wl) {
  if (threadIdx.x < 32) wl[threadIdx.x] = threadIdx.x < 31 ? P.lut.f1e[threadIdx.x] - P.lut.fe[threadIdx.x] : 0;
  __syncthreads();
}

__device__ __forceinline__ int8_t init_genotype(int8_t vt) { return vt == 0 ? 1 : (vt == 1 ? 0 : -1); }  // phase.rs:682-691

// enumeration restarts (phase.rs:1097-1122).  job -> (region slot, enumeration index e)
__global__ void __launch_bounds__(LCR_BLOCK)
k4_enum(PhaseDev P, const int32_t* __restrict__ job_slot, const uint32_t* __restrict__ job_e, int32_t n_jobs,
        long long* __restrict__ job_obj, int materialize) {
  __shared__ long long red[LCR_BLOCK / 64];
  __shared__ long long wl[32];
  load_w(P, wl);
  extern __shared__ __attribute__((aligned(16))) int8_t dyn_state[];  // working sigma|delta|eta when it fits in LDS
  int8_t* base = P.lds_state ? dyn_state : P.scratch + (size_t)blockIdx.x * P.scratch_stride;
  for (int job = blockIdx.x; job < n_jobs; job += gridDim.x) {
    const RegionDev rd = P.reg[job_slot[job]];
    const uint32_t e = job_e[job];
    int8_t* sg = base; int8_t* dl = base + rd.R; int8_t* et = dl + rd.S;
    const int8_t* vt = P.snp_vt + rd.snp_off;
    // hap[e][i] = -1 iff bit i of e (the doubling order of phase.rs:1099-1106)
    for (int i = threadIdx.x; i < rd.S; i += blockDim.x) { dl[i] = ((e >> i) & 1u) ? -1 : 1; et[i] = init_genotype(vt[i]); }
    // init_assignment (phase.rs:673-680): draws continue after thread.rs:162-163's S + F draws
    const uint64_t ctr0 = (uint64_t)rd.S + (uint64_t)rd.R + (uint64_t)e * (uint64_t)rd.R;
    for (int row = threadIdx.x; row < rd.R; row += blockDim.x) sg[row] = u01(rd.seed, ctr0 + row) < 0.5 ? -1 : 1;
    __syncthreads();
    const long long obj = cross_optimize(P, rd, sg, dl, et, false, true, red, wl);
    if (threadIdx.x == 0) job_obj[job] = obj;
    if (materialize) {
      for (int i = threadIdx.x; i < rd.S; i += blockDim.x) { P.st_delta[rd.snp_off + i] = dl[i]; P.st_eta[rd.snp_off + i] = et[i]; }
      for (int row = threadIdx.x; row < rd.R; row += blockDim.x) P.st_sigma[rd.sig_off + row] = sg[row];
      if (threadIdx.x == 0) P.st_obj[job_slot[job]] = obj;
    }
    __syncthreads();
  }
}

// chain, part A (phase.rs:1124-1132): delta from init_haplotypes_LD2 (host), random sigma, keep_conserved
__global__ void __launch_bounds__(LCR_BLOCK) k4_chain_a(PhaseDev P, const int32_t* __restrict__ slots, int32_t n) {
  __shared__ long long red[LCR_BLOCK / 64];
  __shared__ long long wl[32];
  if ((int)blockIdx.x >= n) return;
  load_w(P, wl);
  const int slot = slots[blockIdx.x];
  const RegionDev rd = P.reg[slot];
  int8_t* sg = P.st_sigma + rd.sig_off; int8_t* dl = P.st_delta + rd.snp_off; int8_t* et = P.st_eta + rd.snp_off;
  const int8_t* vt = P.snp_vt + rd.snp_off;
  for (int i = threadIdx.x; i < rd.S; i += blockDim.x) et[i] = init_genotype(vt[i]);
  const uint64_t ctr0 = 2 * (uint64_t)rd.S + (uint64_t)rd.R;  // after S+F (thread.rs) and S (init_haplotypes_LD2)
  for (int row = threadIdx.x; row < rd.R; row += blockDim.x) sg[row] = u01(rd.seed, ctr0 + row) < 0.5 ? -1 : 1;
  __syncthreads();
  const long long obj = cross_optimize(P, rd, sg, dl, et, true, false, red, wl);
  if (threadIdx.x == 0) P.st_obj[slot] = obj;
}

// chain, part B (phase.rs:1197-1233): perturbation rounds with best-state tracking
__global__ void __launch_bounds__(LCR_BLOCK) k4_chain_b(PhaseDev P, const int32_t* __restrict__ slots, int32_t n) {
  __shared__ long long red[LCR_BLOCK / 64];
  __shared__ long long wl[32];
  if ((int)blockIdx.x >= n) return;
  load_w(P, wl);
  const int slot = slots[blockIdx.x];
  const RegionDev rd = P.reg[slot];
  int8_t* bsg = P.st_sigma + rd.sig_off; int8_t* bdl = P.st_delta + rd.snp_off; int8_t* bet = P.st_eta + rd.snp_off;
  extern __shared__ __attribute__((aligned(16))) int8_t dyn_state[];
  int8_t* sg = P.lds_state ? dyn_state : P.scratch + (size_t)blockIdx.x * P.scratch_stride;
  int8_t* dl = sg + rd.R; int8_t* et = dl + rd.S;
  long long best = P.st_obj[slot];
  auto load_best = [&]() {
    for (int i = threadIdx.x; i < rd.S; i += blockDim.x) { dl[i] = bdl[i]; et[i] = bet[i]; }
    for (int row = threadIdx.x; row < rd.R; row += blockDim.x) sg[row] = bsg[row];
    __syncthreads();
  };
  auto save_if_better = [&](long long obj) {
    if (obj > best) {  // uniform: every thread holds the same obj / best
      best = obj;
      for (int i = threadIdx.x; i < rd.S; i += blockDim.x) { bdl[i] = dl[i]; bet[i] = et[i]; }
      for (int row = threadIdx.x; row < rd.R; row += blockDim.x) bsg[row] = sg[row];
    }
    __syncthreads();
  };
  load_best();
  const uint64_t SF = (uint64_t)rd.S + (uint64_t)rd.R;
  for (int tidx = 0; tidx <= rd.S / 4; tidx++) {
    const uint64_t ctr_t = 2 * SF + (uint64_t)tidx * SF;
    const bool flip = (tidx & 1) == 1;
    for (int i = threadIdx.x; i < rd.S; i += blockDim.x) {  // phase.rs:1199-1208
      const double rg = u01(rd.seed, ctr_t + i);
      if (rg < 0.1) dl[i] = flip ? 1 : -1;
      else if (rg >= 0.9) dl[i] = flip ? -1 : 1;
    }
    __syncthreads();
    long long obj = cross_optimize(P, rd, sg, dl, et, false, false, red, wl);
    save_if_better(obj);
    load_best();
    for (int row = threadIdx.x; row < rd.R; row += blockDim.x)  // phase.rs:1217-1224
      if (u01(rd.seed, ctr_t + rd.S + row) < 0.1) sg[row] = (int8_t)(-sg[row]);
    __syncthreads();
    obj = cross_optimize(P, rd, sg, dl, et, false, false, red, wl);
    save_if_better(obj);
    load_best();
  }
  if (threadIdx.x == 0) P.st_obj[slot] = best;
}

// ================================= host side ====================================================

// petgraph 0.6.4 GraphMap<usize,_,Undirected> semantics needed by the reference: node order =
// insertion order, adjacency in edge-insertion order, kosaraju_scc = DfsPostOrder pass over nodes in
// insertion order followed by a LIFO Dfs in reverse finish order (candidate.rs:733, snpfrags.rs:704).
struct PGraph {
  std::vector<int> order;
  std::map<int, std::vector<int>> adj;
  std::set<std::pair<int, int>> edges;
  static std::pair<int, int> key(int a, int b) { return a <= b ? std::make_pair(a, b) : std::make_pair(b, a); }
  bool has_node(int a) const { return adj.count(a) != 0; }
  void add_node(int a) { if (!adj.count(a)) { adj[a]; order.push_back(a); } }
  bool has_edge(int a, int b) const { return edges.count(key(a, b)) != 0; }
  void add_edge(int a, int b) {
    if (!edges.insert(key(a, b)).second) return;
    add_node(a); adj[a].push_back(b);
    if (a != b) { add_node(b); adj[b].push_back(a); }
  }
  std::vector<std::vector<int>> components() const {
    std::set<int> seen, done;
    std::vector<int> fin, st;
    for (int r : order) {
      if (seen.count(r)) continue;
      st.assign(1, r);
      while (!st.empty()) {
        const int x = st.back();
        if (seen.insert(x).second) { for (int y : adj.at(x)) if (!seen.count(y)) st.push_back(y); }
        else { st.pop_back(); if (done.insert(x).second) fin.push_back(x); }
      }
    }
    std::vector<std::vector<int>> out;
    seen.clear();
    for (auto it = fin.rbegin(); it != fin.rend(); ++it) {
      if (seen.count(*it)) continue;
      st.assign(1, *it);
      std::vector<int> comp;
      while (!st.empty()) {
        const int x = st.back(); st.pop_back();
        if (!seen.insert(x).second) continue;
        for (int y : adj.at(x)) if (!seen.count(y)) st.push_back(y);
        comp.push_back(x);
      }
      out.push_back(comp);
    }
    return out;
  }
};

struct HostLut {
  double le[31], l1e[31];  // log10(eps), log10(1-eps), eps = 10^(-q/10) (fragment.rs:132); q=0 treated as q=1
  double p_homref, p_homvar, log_theta, log2;
  PhaseLutDev dev;
  HostLut() {
    for (int q = 0; q <= 30; q++) {
      const int qq = q == 0 ? 1 : q;
      const double eps = std::pow(10.0, -(double)qq / 10.0);
      le[q] = std::log10(eps); l1e[q] = std::log10(1.0 - eps);
      dev.fe[q] = std::llround(le[q] * FX_SCALE); dev.f1e[q] = std::llround(l1e[q] * FX_SCALE);
    }
    p_homref = std::log10(1.0 - 1.5 * 0.001); p_homvar = std::log10(0.5 * 0.001);
    log_theta = std::log10(0.001); log2 = std::log10(2.0);
    dev.f_homref = std::llround(p_homref * FX_SCALE); dev.f_homvar = std::llround(p_homvar * FX_SCALE);
    dev.f_het0 = std::llround(log_theta * FX_SCALE); dev.f_log2 = std::llround(log2 * FX_SCALE);
  }
};
const HostLut& hlut() { static HostLut l; return l; }

struct Obs { int sigma; uint8_t v; };  // one (read haplotag, entry value) observation of a SNP column

inline double lg(int sigma, int delta, int eta, uint8_t v) {  // log10(aki(...)), phase.rs:32-49
  const int p = (v & 32) ? 1 : -1, x = eta == 0 ? sigma * delta : eta;
  return p == x ? hlut().l1e[v & 31] : hlut().le[v & 31];
}
// phase.rs:128-176.  The five log sums of one call are running sums over the same observation order;
// the sum for eta != 0 does not depend on delta, and log_q1 repeats one of the other four, so the four
// distinct sums are computed once (identical addition sequences => identical doubles) and each of
// the reference's calls is assembled from them.
struct ColScores {
  double het_d = 0, het_nd = 0, homref = 0, homvar = 0;  // sum log10 aki for (delta,0), (-delta,0), (.,+1), (.,-1)
  double p_het = 0;
  ColScores(int delta_i, const std::vector<Obs>& o) {
    p_het = o.empty() ? hlut().log_theta : hlut().log_theta - (double)(uint32_t)o.size() * hlut().log2;
    for (const Obs& x : o) {
      het_d += lg(x.sigma, delta_i, 0, x.v); het_nd += lg(x.sigma, -delta_i, 0, x.v);
      homref += lg(x.sigma, delta_i, 1, x.v); homvar += lg(x.sigma, delta_i, -1, x.v);
    }
  }
  // cal_delta_eta_sigma_log(sign * delta_i, eta_i, ...), sign = +1 / -1
  double score(int sign, int eta_i) const {
    const double hd = sign > 0 ? het_d : het_nd, hn = sign > 0 ? het_nd : het_d;
    double q1 = eta_i == 0 ? hd : (eta_i == 1 ? homref : homvar);
    q1 += eta_i == 0 ? p_het : (eta_i == 1 ? hlut().p_homref : hlut().p_homvar);
    const double q2 = homvar + hlut().p_homvar, q3 = hd + p_het, q4 = homref + hlut().p_homref, q5 = hn + p_het;
    return 1.0 - q1 / (q2 + q3 + q4 + q5);
  }
};
double delta_eta_sigma_log(int delta_i, int eta_i, const std::vector<Obs>& o) { return ColScores(delta_i, o).score(1, eta_i); }
// phase.rs:238-255
double phase_score_log(int delta_i, int eta_i, const std::vector<Obs>& o) {
  double q1 = 0, q2 = 0, q3 = 0;
  for (const Obs& x : o) q1 += lg(x.sigma, delta_i, eta_i, x.v);
  for (const Obs& x : o) { q2 += lg(x.sigma, 1, eta_i, x.v); q3 += lg(x.sigma, -1, eta_i, x.v); }
  return 1.0 - q1 / (q2 + q3);
}

// host view of one region: full fragment rows (all entries) + mutable phasing state
struct RegionHost {
  int g = 0, S = 0, nrow = 0;
  int c0 = 0;                      // first candidate (global index)
  int r0 = 0;                      // first row (global index)
  const int64_t* row_ptr = nullptr;  // global CSR (host copy)
  const int32_t* col = nullptr;
  const uint8_t* val = nullptr;
  const uint32_t* links = nullptr;
  lcr_candidate* cand = nullptr;   // cand[0..S)
  std::vector<uint8_t> phase_site; // per entry of this region (index e - row_ptr[r0])
  std::vector<int8_t> tag;         // haplotag per row
  std::vector<uint8_t> asg, fp;    // assignment, for_phasing per row
  std::vector<std::vector<int>> cover;  // per SNP: rows (local) in push order (fragment.rs:293-306)
  std::vector<int> fp_rows;        // local rows with for_phasing at K3 time (the phase matrix rows)
  uint64_t seed = 0, ctr = 0;
  uint32_t min_linkers = 1;
  int64_t e0 = 0;
  double rnd() { return u01(seed, ctr++); }
  int64_t eb(int r) const { return row_ptr[r0 + r]; }
  int64_t ee(int r) const { return row_ptr[r0 + r + 1]; }
  int lc(int64_t e) const { return col[e] - c0; }
  bool fphase(int i) const { return (cand[i].flags & LCR_F_FOR_PHASING) != 0; }

  // snpfrags.rs:548-625
  void assign_reads_haplotype(double cutoff) {
    for (int r = 0; r < nrow; r++) {
      if (!fp[r]) continue;
      const int sigma_k = tag[r];
      double q1 = 0, q2 = 0, q3 = 0, n1 = 0;
      int n = 0;
      for (int64_t e = eb(r); e < ee(r); e++) {
        const int i = lc(e);
        if (!phase_site[e - e0] && fphase(i)) phase_site[e - e0] = 1;
        if (!fphase(i) || cand[i].haplotype == 0 || cand[i].genotype != 0) continue;
        // cal_sigma_delta_eta_log (phase.rs:77-96) for sigma_k and -sigma_k share log_q2/log_q3
        q1 += lg(sigma_k, cand[i].haplotype, 0, val[e]);
        n1 += lg(-sigma_k, cand[i].haplotype, 0, val[e]);
        n++;
      }
      if (sigma_k == 0 || n == 0) { asg[r] = 0; tag[r] = 0; continue; }
      for (int64_t e = eb(r); e < ee(r); e++) {
        const int i = lc(e);
        if (!fphase(i) || cand[i].haplotype == 0 || cand[i].genotype != 0) continue;
        q2 += lg(1, cand[i].haplotype, 0, val[e]); q3 += lg(-1, cand[i].haplotype, 0, val[e]);
      }
      const double q = 1.0 - q1 / (q2 + q3), qn = 1.0 - n1 / (q2 + q3);
      if (std::fabs(q - qn) >= cutoff) {
        if (q >= qn) asg[r] = sigma_k == 1 ? 1 : 2;
        else if (sigma_k == 1) { asg[r] = 2; tag[r] = -1; }
        else { asg[r] = 1; tag[r] = 1; }
      } else { asg[r] = 0; tag[r] = 0; }
    }
  }

  void gather(int ti, bool need_assigned, bool het_skip_unassigned, std::vector<Obs>& o, int& hap1, int& hap2) const {
    o.clear(); hap1 = hap2 = 0;
    for (int r : cover[ti]) {
      if (!fp[r] || links[r0 + r] < min_linkers) continue;
      if (need_assigned && asg[r] == 0) continue;
      if (het_skip_unassigned && asg[r] == 0) continue;
      for (int64_t e = eb(r); e < ee(r); e++)
        if (lc(e) == ti) {
          if (asg[r] == 1) hap1++; else if (asg[r] == 2) hap2++;
          o.push_back({tag[r], val[e]});
        }
    }
  }

  // snpfrags.rs:378-546
  void assign_snp_haplotype_genotype() {
    std::vector<Obs> o;
    for (int ti = 0; ti < S; ti++) {
      lcr_candidate& snp = cand[ti];
      if (!fphase(ti)) { snp.flags |= LCR_F_NON_SELECTED; continue; }
      if (cover[ti].empty()) { snp.flags |= LCR_F_SINGLE; continue; }
      const int delta_i = snp.haplotype;
      int hap1, hap2;
      gather(ti, false, snp.variant_type == 1, o, hap1, hap2);
      if (o.empty()) { snp.flags |= LCR_F_NON_SELECTED; continue; }
      const ColScores cs(delta_i, o);
      const double q1 = cs.score(1, 0), q2 = cs.score(-1, 0), q3 = cs.score(1, 1), q4 = cs.score(1, -1);
      const double mx = std::fmax(q1, std::fmax(q2, std::fmax(q3, q4)));
      if (q1 == mx) { snp.haplotype = delta_i; snp.genotype = 0; snp.variant_type = 1; }
      else if (q2 == mx) { snp.haplotype = -delta_i; snp.genotype = 0; snp.variant_type = 1; }
      else if (q3 == mx) { snp.haplotype = delta_i; snp.genotype = 1; snp.variant_type = 0; }
      else if (q4 == mx) { snp.haplotype = delta_i; snp.genotype = -1; if (snp.variant_type != 2 && snp.variant_type != 3) snp.variant_type = 2; }
      else continue;  // NaN scores: the reference panics here
      if (snp.genotype != 0) { snp.flags |= LCR_F_NON_SELECTED; continue; }
      if (hap1 >= 1 && hap2 >= 1) snp.phase_score = -10.0 * std::log10(1.0 - phase_score_log(snp.haplotype, snp.genotype, o));
      else snp.phase_score = 0.19940219;
    }
  }

  // snpfrags.rs:191-376 (eval_rna_edit_var_phase / eval_low_frac_var_phase)
  void eval_rescue(uint32_t list_flag, float min_phase_score, bool low_frac) {
    std::vector<int> list;  // edit_snps / somatic_snps are fixed at candidate time (snpfrags.rs:20-26)
    for (int i = 0; i < S; i++) if (orig_flags[i] & list_flag) list.push_back(i);
    std::vector<Obs> o;
    for (int ti : list) {
      lcr_candidate& snp = cand[ti];
      if (cover[ti].empty()) { snp.flags |= LCR_F_SINGLE; continue; }
      if (snp.variant_type != 1) { snp.flags |= LCR_F_NON_SELECTED; continue; }
      int hap1, hap2;
      gather(ti, true, false, o, hap1, hap2);
      if (o.empty() || hap1 < 2 || hap2 < 2) { snp.flags |= LCR_F_SINGLE; continue; }
      const double ps1 = -10.0 * std::log10(1.0 - phase_score_log(1, 0, o));
      const double ps2 = -10.0 * std::log10(1.0 - phase_score_log(-1, 0, o));
      snp.flags &= ~(uint32_t)LCR_F_SINGLE;
      if (std::fmax(ps1, ps2) >= (double)min_phase_score) {
        snp.flags &= ~(uint32_t)(LCR_F_NON_SELECTED | LCR_F_RNA_EDIT);
        if (low_frac) snp.flags &= ~(uint32_t)LCR_F_CAND_SOMATIC;
        snp.flags |= LCR_F_FOR_PHASING;
        for (int r : cover[ti]) {
          fp[r] = 1;
          if (tag[r] == 0 || asg[r] == 0) tag[r] = rnd() < 0.5 ? -1 : 1;
        }
        snp.haplotype = ps1 >= ps2 ? 1 : -1;
        snp.genotype = 0; snp.variant_type = 1; snp.phase_score = std::fmax(ps1, ps2);
      } else {
        snp.flags |= LCR_F_NON_SELECTED;
        if (low_frac) { snp.flags |= LCR_F_CAND_SOMATIC; snp.flags &= ~(uint32_t)LCR_F_FOR_PHASING; }
        else snp.flags |= LCR_F_RNA_EDIT;
      }
    }
  }
  std::vector<uint32_t> orig_flags;

  // snpfrags.rs:628-733
  // snpfrags.rs:628-733.  The reference builds a petgraph GraphMap whose nodes are the PASS het SNPs
  // (added in index order), adds an edge per read and allele-consistent SNP pair, and walks
  // kosaraju_scc: components come out in descending order of their first-inserted (= smallest-index)
  // node, a component's phase set is pos+1 of that node, and a read takes the phase set of the first
  // component in that order that owns one of its edges.  Union-find gives exactly that.
  void assign_phase_set(float min_phase_score, uint32_t* row_ps /* global rows */) {
    std::vector<int> parent(S, -1);  // -1: not a node
    for (int i = 0; i < S; i++) {
      const lcr_candidate& s = cand[i];
      if (s.genotype != 0 || s.variant_type != 1) continue;
      if (s.flags & (LCR_F_DENSE | LCR_F_RNA_EDIT)) continue;
      if (s.phase_score < (double)min_phase_score) continue;
      parent[i] = i;
    }
    auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    auto unite = [&](int x, int y) { x = find(x); y = find(y); if (x != y) { if (x < y) parent[y] = x; else parent[x] = y; } };  // root = min index
    int ns[64], np[64];
    auto row_nodes = [&](int r) {
      int n = 0;
      for (int64_t e = eb(r); e < ee(r) && n < 64; e++)
        if (parent[lc(e)] >= 0) { ns[n] = lc(e); np[n] = (val[e] & 32) ? 1 : -1; n++; }
      return n;
    };
    for (int r = 0; r < nrow; r++) {
      if (!fp[r] || asg[r] == 0) continue;
      const int n = row_nodes(r);
      for (int x = 0; x < n; x++)
        for (int y = x + 1; y < n; y++)
          if (cand[ns[x]].haplotype * cand[ns[y]].haplotype == np[x] * np[y]) unite(ns[x], ns[y]);
    }
    for (int i = 0; i < S; i++) if (parent[i] >= 0) cand[i].phase_set = (uint32_t)(cand[find(i)].pos + 1);
    for (int r = 0; r < nrow; r++) {
      if (!fp[r] || asg[r] == 0) continue;
      const int n = row_nodes(r);
      int best = -1;  // largest component root among the components that own an edge of this read
      if (n == 1) best = find(ns[0]);  // self loop (snpfrags.rs:659-665)
      for (int x = 0; x < n; x++)
        for (int y = x + 1; y < n; y++)
          if (cand[ns[x]].haplotype * cand[ns[y]].haplotype == np[x] * np[y]) best = std::max(best, find(ns[x]));
      if (best >= 0) row_ps[r0 + r] = (uint32_t)(cand[best].pos + 1);
    }
  }

  // exact objective of the current state over the phase matrix (flat CSR of the phasing rows)
  long long objective_fx(const std::vector<int32_t>& prow_ptr, const std::vector<int32_t>& pcol,
                         const std::vector<uint8_t>& pval) const {
    long long s = 0;
    const PhaseLutDev& L = hlut().dev;
    for (size_t k = 0; k < fp_rows.size(); k++) {
      const int sg = tag[fp_rows[k]];
      for (int e = prow_ptr[k]; e < prow_ptr[k + 1]; e++) {
        const int p = (pval[e] & 32) ? 1 : -1;
        const int x = cand[pcol[e]].genotype == 0 ? sg * cand[pcol[e]].haplotype : cand[pcol[e]].genotype;
        s += p == x ? L.f1e[pval[e] & 31] : L.fe[pval[e] & 31];
      }
    }
    return s;
  }
};

}  // namespace

int PhaseHost::run(const PhaseInputs& in, const lcr_params& prm, hipStream_t stream, std::string* err) {
#define PCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_); return LCR_E_DEVICE; } } while (0)
  const int ng = in.n_regions, nrow = in.n_rows;
  const int64_t nnz = in.nnz;
  const bool prof = getenv("LCR_PHASE_PROF") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) { if (!prof) return; auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[phase] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count()); t_last = t; };
  std::vector<lcr_candidate>& cand = *in.cand;
  haplotag.assign(nrow, 0); assignment.assign(nrow, 0); phase_set.assign(nrow, 0); objective.assign(ng, 0.0);
  // host copy of the fragment matrix (needed by the LD-block pass and the post-phase epilogue)
  // (pinned staging buffers: the pageable path of hipMemcpyAsync costs an extra host copy)
  PCHK(h_pin[0].reserve((size_t)(nrow + 1) * 8)); PCHK(h_pin[1].reserve(std::max<size_t>(nnz, 1) * 4));
  PCHK(h_pin[2].reserve(std::max<size_t>(nnz, 1))); PCHK(h_pin[3].reserve(std::max<size_t>(nrow, 1) * 4));
  int64_t* const row_ptr_p = h_pin[0].as<int64_t>();
  int32_t* const col_p = h_pin[1].as<int32_t>();
  uint8_t* const val_p = h_pin[2].as<uint8_t>();
  uint32_t* const links_p = h_pin[3].as<uint32_t>();
  PCHK(hipMemcpyAsync(row_ptr_p, in.d_row_ptr, (size_t)(nrow + 1) * 8, hipMemcpyDeviceToHost, stream));
  if (nnz) {
    PCHK(hipMemcpyAsync(col_p, in.d_col, (size_t)nnz * 4, hipMemcpyDeviceToHost, stream));
    PCHK(hipMemcpyAsync(val_p, in.d_val, (size_t)nnz, hipMemcpyDeviceToHost, stream));
  }
  if (nrow) PCHK(hipMemcpyAsync(links_p, in.d_row_links, (size_t)nrow * 4, hipMemcpyDeviceToHost, stream));
  PCHK(hipStreamSynchronize(stream));
  struct Arr64 { int64_t* p; int64_t& operator[](size_t i) const { return p[i]; } int64_t* data() const { return p; } } row_ptr{row_ptr_p};
  struct Arr32 { int32_t* p; int32_t& operator[](size_t i) const { return p[i]; } int32_t* data() const { return p; } } col{col_p};
  struct Arr8 { uint8_t* p; uint8_t& operator[](size_t i) const { return p[i]; } uint8_t* data() const { return p; } } val{val_p};
  struct ArrU { uint32_t* p; uint32_t& operator[](size_t i) const { return p[i]; } uint32_t* data() const { return p; } } links{links_p};
  lap("d2h fragment matrix");

  std::vector<RegionHost> R(ng);
  std::vector<std::vector<std::vector<int>>> ld_blocks(ng);
  struct RegionBuild {  // per-region pieces, built in parallel, concatenated below
    std::vector<int32_t> prow_ptr, pcol, ccol_ptr, crow;
    std::vector<uint8_t> pval, cval, fp, cons;
    std::vector<int8_t> vt, delta0;
    std::vector<long long> snp_const;  // 4 per SNP: F, W, Cref, Cvar
    long long f_total = 0;
  };
  std::vector<RegionBuild> RB(ng);
  (void)hlut();

  auto prep = [&](int g) {
    RegionHost& rh = R[g];
    RegionBuild& rb = RB[g];
    rh.g = g; rh.c0 = in.cand_region_off[g]; rh.S = in.cand_region_off[g + 1] - rh.c0;
    rh.r0 = in.row_region_off[g]; rh.nrow = in.row_region_off[g + 1] - rh.r0;
    rh.row_ptr = row_ptr.data(); rh.col = col.data(); rh.val = val.data(); rh.links = links.data();
    rh.cand = cand.data() + rh.c0;
    rh.seed = region_seed(prm.seed, in.region_start0[g]);
    rh.min_linkers = prm.min_linkers;
    rh.e0 = nrow ? row_ptr[rh.r0] : 0;
    if (rh.S == 0) return;
    const int64_t e1 = row_ptr[rh.r0 + rh.nrow];
    rh.phase_site.assign((size_t)(e1 - rh.e0), 0);
    rh.tag.assign(rh.nrow, 0); rh.asg.assign(rh.nrow, 0); rh.fp.assign(rh.nrow, 0);
    rh.cover.assign(rh.S, {});
    rh.orig_flags.resize(rh.S);
    for (int i = 0; i < rh.S; i++) rh.orig_flags[i] = rh.cand[i].flags;
    // phase matrix of the region: flat CSR over the phasing rows (entries at phase sites only) + CSC mirror
    std::vector<int32_t> ccnt(rh.S + 1, 0);
    rb.prow_ptr.push_back(0);
    for (int r = 0; r < rh.nrow; r++) {
      for (int64_t e = rh.eb(r); e < rh.ee(r); e++) {
        const int i = rh.lc(e);
        rh.cover[i].push_back(r);
        if (rh.fphase(i)) rh.phase_site[e - rh.e0] = 1;  // fragment.rs:144-146 snapshot
      }
      if (links[rh.r0 + r] >= prm.min_linkers) {          // fragment.rs:253-255
        rh.fp[r] = 1;
        rh.fp_rows.push_back(r);
        for (int64_t e = rh.eb(r); e < rh.ee(r); e++)
          if (rh.phase_site[e - rh.e0]) { rb.pcol.push_back(rh.lc(e)); rb.pval.push_back((uint8_t)(val[e] & 63)); ccnt[rh.lc(e) + 1]++; }
        rb.prow_ptr.push_back((int32_t)rb.pcol.size());
      }
    }
    const int32_t acc = (int32_t)rb.pcol.size();
    const size_t n_prow = rh.fp_rows.size();
    for (int i = 0; i < rh.S; i++) ccnt[i + 1] += ccnt[i];
    rb.ccol_ptr.assign(ccnt.begin(), ccnt.end());
    rb.crow.resize(acc); rb.cval.resize(acc);
    std::vector<int32_t> fill(ccnt.begin(), ccnt.end() - 1);
    for (size_t k = 0; k < n_prow; k++)
      for (int e = rb.prow_ptr[k]; e < rb.prow_ptr[k + 1]; e++) {
        const int i = rb.pcol[e];
        rb.crow[fill[i]] = (int32_t)k; rb.cval[fill[i]] = rb.pval[e]; fill[i]++;
      }
    rb.fp.resize(rh.S); rb.vt.resize(rh.S); rb.cons.assign(rh.S, 0); rb.delta0.assign(rh.S, 1);
    rb.snp_const.assign(4 * (size_t)rh.S, 0);
    for (int e = 0; e < acc; e++) {
      const PhaseLutDev& LD = hlut().dev;
      const int q = rb.pval[e] & 31, i = rb.pcol[e];
      const bool pref = (rb.pval[e] & 32) != 0;
      rb.snp_const[4 * i] += LD.fe[q]; rb.snp_const[4 * i + 1] += LD.f1e[q] - LD.fe[q];
      rb.snp_const[4 * i + 2] += pref ? LD.f1e[q] : LD.fe[q]; rb.snp_const[4 * i + 3] += pref ? LD.fe[q] : LD.f1e[q];
      rb.f_total += LD.fe[q];
    }
    for (int i = 0; i < rh.S; i++) { rb.fp[i] = rh.fphase(i) ? 1 : 0; rb.vt[i] = (int8_t)rh.cand[i].variant_type; }
    // thread.rs:162-163: init_haplotypes + init_assignment consume S + F draws; both are overwritten
    if ((uint32_t)rh.S <= prm.max_enum_snps) return;
    // ---- divide_snps_into_blocks (candidate.rs:615-747) + init_haplotypes_LD2 (phase.rs:609-671), host
    const int F = (int)rh.fp_rows.size();
    std::map<std::pair<int, int>, std::array<int, 4>> pairs;  // (i<j) -> counts [ref/alt i][ref/alt j]
    auto one_ref = [&](int i) {  // exactly one of the two major alleles is the reference (candidate.rs:637-660)
      const lcr_candidate& c = rh.cand[i];
      return (c.allele1 == c.ref_base) != (c.allele2 == c.ref_base);
    };
    for (size_t k = 0; k < n_prow; k++)
      for (int x = rb.prow_ptr[k]; x < rb.prow_ptr[k + 1]; x++)
        for (int y = x + 1; y < rb.prow_ptr[k + 1]; y++) {
          int i = rb.pcol[x], j = rb.pcol[y];
          int pi = (rb.pval[x] & 32) ? 0 : 1, pj = (rb.pval[y] & 32) ? 0 : 1;
          if (i > j) { std::swap(i, j); std::swap(pi, pj); }
          pairs[{i, j}][pi * 2 + pj]++;
        }
    std::map<std::pair<int, int>, int> ld_weight;  // perfect-LD pairs (score == 0) -> weight
    std::vector<std::pair<int, int>> pass;
    for (auto& kv : pairs) {
      const int i = kv.first.first, j = kv.first.second;
      if (!one_ref(i) || !one_ref(j)) continue;
      const lcr_candidate &si = rh.cand[i], &sj = rh.cand[j];
      if (si.af1 == 0.0f || si.af2 == 0.0f || sj.af1 == 0.0f || sj.af2 == 0.0f) continue;
      const auto& c = kv.second;  // snp.rs:158-188
      const int cis = c[0] + c[3], trans = c[1] + c[2];
      const int c1 = std::min(cis, trans), c2 = std::max(cis, trans);
      const int weight = cis > trans ? c2 : -c2;
      if (c2 > 0 && c1 == 0) { pass.push_back({i, j}); ld_weight[{i, j}] = weight; }
    }
    PGraph lg;
    for (auto& pq : pass) lg.add_edge(pq.first, pq.second);  // std::map order == (i asc, j asc) loop order
    // edges with |weight| < ld_weight_threshold are removed (candidate.rs:703-711); petgraph swap_removes
    for (auto& kv : ld_weight)
      if ((uint32_t)std::abs(kv.second) < prm.ld_weight_threshold) {
        lg.edges.erase(PGraph::key(kv.first.first, kv.first.second));
        auto rm = [&](int x, int y) { auto& v = lg.adj[x]; auto f = std::find(v.begin(), v.end(), y); if (f != v.end()) { *f = v.back(); v.pop_back(); } };
        rm(kv.first.first, kv.first.second); rm(kv.first.second, kv.first.first);
      }
    ld_blocks[g] = lg.components();
    // init_haplotypes_LD2: S random draws (ctr S+F ..), then BFS propagation inside each block
    const uint64_t c_ld = (uint64_t)rh.S + (uint64_t)F;
    int8_t* d0 = rb.delta0.data();
    uint8_t* cons = rb.cons.data();
    for (int i = 0; i < rh.S; i++) d0[i] = u01(rh.seed, c_ld + i) < 0.5 ? 1 : -1;
    const int thr = (int)prm.ld_weight_threshold;
    for (auto& block : ld_blocks[g]) {
      if (block.size() < 2) continue;
      std::set<int> disc; std::vector<int> queue, visited;
      size_t qh = 0;
      disc.insert(block[0]); queue.push_back(block[0]);
      d0[block[0]] = 1;
      visited.push_back(block[0]);
      while (qh < queue.size()) {  // petgraph Bfs: pop front, push unseen neighbours
        const int nx = queue[qh++];
        for (int y : lg.adj.at(nx)) if (disc.insert(y).second) queue.push_back(y);
        for (int vis : visited) {
          if (vis == nx) continue;
          auto f = ld_weight.find({std::min(vis, nx), std::max(vis, nx)});
          if (f == ld_weight.end()) continue;  // pair absent, not valid, or not perfect LD
          if (f->second >= thr) { d0[nx] = d0[vis]; break; }
          if (f->second <= -thr) { d0[nx] = (int8_t)(-d0[vis]); break; }
        }
        visited.push_back(nx);
      }
      for (int i : block) cons[i] = 1;
    }
  };
  if (!pool) {
    int nthreads = (int)std::thread::hardware_concurrency();
    if (const char* e = getenv("LCR_HOST_THREADS")) nthreads = atoi(e);
    nthreads = std::max(1, std::min(nthreads, 48));
    pool = new HostPool(nthreads > 1 ? nthreads : 0);
  }
  auto for_regions = [&](const std::function<void(int)>& fn) { pool->parallel_for(ng, fn); };
  for_regions(prep);

  // ---- concatenate the per-region slices (serial, memcpy-sized)
  std::vector<RegionDev> rdev;
  std::vector<int> slot_of(ng, -1);
  std::vector<int32_t> h_prow_ptr, h_pcol, h_ccol_ptr, h_crow;
  std::vector<uint8_t> h_pval, h_cval, h_fp, h_cons;
  std::vector<int8_t> h_vt, h_delta0;
  std::vector<long long> h_snp_const;
  int32_t sig_total = 0, snp_total = 0, max_state = 0;
  std::vector<int32_t> enum_slots, chain_slots;
  auto app = [](auto& dst, const auto& src) { dst.insert(dst.end(), src.begin(), src.end()); };
  for (int g = 0; g < ng; g++) {
    RegionHost& rh = R[g];
    if (rh.S == 0) continue;
    RegionBuild& rb = RB[g];
    RegionDev rd{};
    rd.R = (int32_t)rh.fp_rows.size(); rd.S = rh.S;
    rd.rp_off = (int32_t)h_prow_ptr.size(); rd.cp_off = (int32_t)h_ccol_ptr.size(); rd.e_off = (int64_t)h_pcol.size();
    rd.sig_off = sig_total; rd.snp_off = snp_total; rd.seed = rh.seed; rd.f_total = rb.f_total;
    sig_total += rd.R; snp_total += rd.S;
    max_state = std::max(max_state, rd.R + 2 * rd.S);
    app(h_prow_ptr, rb.prow_ptr); app(h_pcol, rb.pcol); app(h_pval, rb.pval);
    app(h_ccol_ptr, rb.ccol_ptr); app(h_crow, rb.crow); app(h_cval, rb.cval);
    app(h_fp, rb.fp); app(h_vt, rb.vt); app(h_cons, rb.cons); app(h_delta0, rb.delta0); app(h_snp_const, rb.snp_const);
    slot_of[g] = (int)rdev.size();
    if ((uint32_t)rh.S <= prm.max_enum_snps) enum_slots.push_back(slot_of[g]); else chain_slots.push_back(slot_of[g]);
    rdev.push_back(rd);
  }
  lap("host region prep + LD");
  const HostLut& L = hlut();
  if (!rdev.empty()) {
    // ---- upload the phase matrices
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
      hipError_t e = b.reserve(std::max<size_t>(bytes, 16));
      if (e != hipSuccess) return e;
      return bytes ? hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, stream) : hipSuccess;
    };
    DevBuf &b_reg = d_state[0], &b_prp = d_state[1], &b_pc = d_state[2], &b_pv = d_state[3], &b_cp = d_state[4],
           &b_cr = d_state[5], &b_cv = d_state[6], &b_snp = d_state[7], &b_st = d_state[8], &b_scr = d_state[9],
           &b_job = d_state[10], &b_obj = d_state[11], &b_sc = d_state[12];
    PCHK(up(b_reg, rdev.data(), rdev.size() * sizeof(RegionDev)));
    PCHK(up(b_prp, h_prow_ptr.data(), h_prow_ptr.size() * 4));
    PCHK(up(b_pc, h_pcol.data(), h_pcol.size() * 4));
    PCHK(up(b_pv, h_pval.data(), h_pval.size()));
    PCHK(up(b_cp, h_ccol_ptr.data(), h_ccol_ptr.size() * 4));
    PCHK(up(b_cr, h_crow.data(), h_crow.size() * 4));
    PCHK(up(b_cv, h_cval.data(), h_cval.size()));
    // per-SNP arrays: fp | vt | cons, each snp_total bytes
    std::vector<uint8_t> snp_pack((size_t)snp_total * 3 + 16);
    memcpy(snp_pack.data(), h_fp.data(), snp_total);
    memcpy(snp_pack.data() + snp_total, h_vt.data(), snp_total);
    memcpy(snp_pack.data() + 2 * (size_t)snp_total, h_cons.data(), snp_total);
    PCHK(up(b_snp, snp_pack.data(), snp_pack.size()));
    PCHK(up(b_sc, h_snp_const.data(), h_snp_const.size() * sizeof(long long)));
    // state: sigma[sig_total] | delta[snp_total] | eta[snp_total] | obj[n_slots] (8-byte aligned)
    const size_t st_sig = 0, st_del = ((size_t)sig_total + 15) & ~(size_t)15, st_eta = st_del + (((size_t)snp_total + 15) & ~(size_t)15);
    const size_t st_obj = st_eta + (((size_t)snp_total + 15) & ~(size_t)15);
    PCHK(b_st.reserve(st_obj + rdev.size() * 8 + 16));
    PCHK(hipMemsetAsync(b_st.p, 0, st_obj + rdev.size() * 8, stream));
    PCHK(hipMemcpyAsync(b_st.as<int8_t>() + st_del, h_delta0.data(), snp_total, hipMemcpyHostToDevice, stream));
    const int32_t stride = (max_state + 63) & ~63;
    const int n_blocks_max = 2048;
    PCHK(b_scr.reserve((size_t)stride * n_blocks_max + 64));

    PhaseDev P{};
    P.reg = b_reg.as<RegionDev>();
    P.prow_ptr = b_prp.as<int32_t>(); P.pcol = b_pc.as<int32_t>(); P.pval = b_pv.as<uint8_t>();
    P.ccol_ptr = b_cp.as<int32_t>(); P.crow = b_cr.as<int32_t>(); P.cval = b_cv.as<uint8_t>();
    P.snp_const = b_sc.as<long long>();
    P.snp_fp = b_snp.as<uint8_t>(); P.snp_vt = b_snp.as<int8_t>() + snp_total; P.snp_cons = b_snp.as<uint8_t>() + 2 * (size_t)snp_total;
    P.st_sigma = b_st.as<int8_t>() + st_sig; P.st_delta = b_st.as<int8_t>() + st_del; P.st_eta = b_st.as<int8_t>() + st_eta;
    P.st_obj = (long long*)(b_st.as<int8_t>() + st_obj);
    P.scratch = b_scr.as<int8_t>(); P.scratch_stride = stride;
    const size_t dyn_bytes = stride <= 48 * 1024 ? (size_t)stride : 0;  // working state in LDS when it fits
    P.lds_state = dyn_bytes ? 1 : 0;
    P.lut = L.dev;

    PCHK(hipStreamSynchronize(stream));
    lap("upload phase matrices");
    // ---- enumeration regions: all restarts in one launch, then re-run the winners
    if (!enum_slots.empty()) {
      std::vector<int32_t> job_slot; std::vector<uint32_t> job_e; std::vector<size_t> first_job;
      for (int s : enum_slots) {
        first_job.push_back(job_slot.size());
        const uint32_t n = 1u << rdev[s].S;
        for (uint32_t e = 0; e < n; e++) { job_slot.push_back(s); job_e.push_back(e); }
      }
      first_job.push_back(job_slot.size());
      const int nj = (int)job_slot.size();
      PCHK(b_job.reserve((size_t)nj * 8 + 64));
      PCHK(b_obj.reserve((size_t)nj * 8 + 64));
      int32_t* d_js = b_job.as<int32_t>(); uint32_t* d_je = (uint32_t*)(b_job.as<int32_t>() + nj);
      PCHK(hipMemcpyAsync(d_js, job_slot.data(), (size_t)nj * 4, hipMemcpyHostToDevice, stream));
      PCHK(hipMemcpyAsync(d_je, job_e.data(), (size_t)nj * 4, hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(k4_enum, dim3(std::min(nj, n_blocks_max)), dim3(LCR_BLOCK), dyn_bytes, stream, P, d_js, d_je, nj,
                         b_obj.as<long long>(), 0);
      std::vector<long long> obj(nj);
      PCHK(hipMemcpyAsync(obj.data(), b_obj.p, (size_t)nj * 8, hipMemcpyDeviceToHost, stream));
      PCHK(hipStreamSynchronize(stream));
      PCHK(hipGetLastError());
      std::vector<int32_t> win_slot; std::vector<uint32_t> win_e;
      for (size_t k = 0; k < enum_slots.size(); k++) {
        size_t best = first_job[k];
        for (size_t j = first_job[k] + 1; j < first_job[k + 1]; j++) if (obj[j] > obj[best]) best = j;  // first maximum
        win_slot.push_back(enum_slots[k]); win_e.push_back(job_e[best]);
      }
      const int nw = (int)win_slot.size();
      PCHK(hipMemcpyAsync(d_js, win_slot.data(), (size_t)nw * 4, hipMemcpyHostToDevice, stream));
      PCHK(hipMemcpyAsync(d_js + nw, win_e.data(), (size_t)nw * 4, hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(k4_enum, dim3(std::min(nw, n_blocks_max)), dim3(LCR_BLOCK), dyn_bytes, stream, P, d_js, (uint32_t*)(d_js + nw), nw,
                         b_obj.as<long long>(), 1);
      PCHK(hipGetLastError());
      PCHK(hipStreamSynchronize(stream));  // win_slot / win_e are pageable host vectors
    }
    lap("enum kernels");
    // ---- chain regions
    const size_t st_bytes = st_obj + rdev.size() * 8;
    PCHK(h_pin[4].reserve(st_bytes + 16));
    struct StHost { int8_t* p; size_t n; int8_t* data() const { return p; } size_t size() const { return n; } } st_host{h_pin[4].as<int8_t>(), st_bytes};
    auto pull_state = [&]() -> hipError_t {
      hipError_t e = hipMemcpyAsync(st_host.data(), b_st.p, st_host.size(), hipMemcpyDeviceToHost, stream);
      if (e != hipSuccess) return e;
      return hipStreamSynchronize(stream);
    };
    if (!chain_slots.empty()) {
      const int nc = (int)chain_slots.size();
      if ((size_t)stride * nc + 64 > b_scr.cap) PCHK(b_scr.reserve((size_t)stride * nc + 64));
      P.scratch = b_scr.as<int8_t>();
      DevBuf& b_slots = b_job;
      PCHK(b_slots.reserve((size_t)nc * 4 + 64));
      PCHK(hipMemcpyAsync(b_slots.p, chain_slots.data(), (size_t)nc * 4, hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(k4_chain_a, dim3(nc), dim3(LCR_BLOCK), 0, stream, P, b_slots.as<int32_t>(), nc);
      PCHK(hipGetLastError());
      PCHK(pull_state());
      // LD-block flip pass on the host (phase.rs:1298-1394): a sum-of-ratios f64 decision per block
      auto block_pass = [&](int g) {
        if (slot_of[g] < 0 || (uint32_t)R[g].S <= prm.max_enum_snps) return;
        RegionHost& rh = R[g];
        const RegionDev& rd = rdev[slot_of[g]];
        int8_t* sg = st_host.data() + st_sig + rd.sig_off; int8_t* dl = st_host.data() + st_del + rd.snp_off;
        int8_t* et = st_host.data() + st_eta + rd.snp_off;
        long long* ob = (long long*)(st_host.data() + st_obj) + slot_of[g];
        for (size_t k = 0; k < rh.fp_rows.size(); k++) rh.tag[rh.fp_rows[k]] = sg[k];
        for (int i = 0; i < rh.S; i++) { rh.cand[i].haplotype = dl[i]; rh.cand[i].genotype = et[i]; }
        std::map<int, int> new_hap, new_tag;
        std::vector<Obs> o, oflip;
        for (auto& block : ld_blocks[g]) {
          std::set<int> bset(block.begin(), block.end());
          std::map<int, int> flipmap;
          double q = 0.0, qf = 0.0;
          for (int idx : block) {
            o.clear(); oflip.clear();
            for (int r : rh.cover[idx]) {
              if (!rh.fp[r] || rh.tag[r] == 0) continue;
              bool flip_read = true;  // only entries *before* idx in the row can veto (phase.rs:1331-1349)
              for (int64_t e = rh.eb(r); e < rh.ee(r); e++) {
                if (!bset.count(rh.lc(e))) flip_read = false;
                if (rh.lc(e) == idx) {
                  if (!rh.phase_site[e - rh.e0]) continue;
                  const int s = rh.tag[r], sf = flip_read ? -s : s;
                  o.push_back({s, val[e]}); oflip.push_back({sf, val[e]});
                  flipmap[r] = sf;
                }
              }
            }
            q += delta_eta_sigma_log(rh.cand[idx].haplotype, rh.cand[idx].genotype, o);      // phase.rs:178-236
            qf += delta_eta_sigma_log(-rh.cand[idx].haplotype, rh.cand[idx].genotype, oflip);
          }
          const bool do_flip = q < qf;
          for (int idx : block) new_hap[idx] = do_flip ? -rh.cand[idx].haplotype : rh.cand[idx].haplotype;
          for (int r = 0; r < rh.nrow; r++) {  // every block rewrites the whole map (phase.rs:1364-1378)
            auto f = flipmap.find(r);
            new_tag[r] = (do_flip && f != flipmap.end()) ? f->second : rh.tag[r];
          }
        }
        std::vector<int8_t> old_tag(rh.tag), old_hap(rh.S);
        for (int i = 0; i < rh.S; i++) old_hap[i] = (int8_t)rh.cand[i].haplotype;
        for (auto& kv : new_hap) rh.cand[kv.first].haplotype = kv.second;
        for (auto& kv : new_tag) rh.tag[kv.first] = (int8_t)kv.second;
        const long long obj2 = rh.objective_fx(RB[g].prow_ptr, RB[g].pcol, RB[g].pval);
        if (obj2 > *ob) {  // `prob > largest_prob` (phase.rs:1140-1144): keep the flipped state
          *ob = obj2;
          for (size_t k = 0; k < rh.fp_rows.size(); k++) sg[k] = rh.tag[rh.fp_rows[k]];
          for (int i = 0; i < rh.S; i++) dl[i] = (int8_t)rh.cand[i].haplotype;
        } else {             // load_best_configuration: back to launch A's state
          rh.tag = old_tag;
          for (int i = 0; i < rh.S; i++) rh.cand[i].haplotype = old_hap[i];
        }
      };
      for_regions(block_pass);
      PCHK(hipMemcpyAsync(b_st.p, st_host.data(), st_host.size(), hipMemcpyHostToDevice, stream));
      hipLaunchKernelGGL(k4_chain_b, dim3(nc), dim3(LCR_BLOCK), dyn_bytes, stream, P, b_slots.as<int32_t>(), nc);
      PCHK(hipGetLastError());
    }
    PCHK(pull_state());
    PCHK(hipGetLastError());
    lap("chain kernels + block pass");
    // ---- scatter device results into the host region views
    for (int g = 0; g < ng; g++) {
      if (slot_of[g] < 0) continue;
      RegionHost& rh = R[g];
      const RegionDev& rd = rdev[slot_of[g]];
      const int8_t* sg = st_host.data() + st_sig + rd.sig_off; const int8_t* dl = st_host.data() + st_del + rd.snp_off;
      const int8_t* et = st_host.data() + st_eta + rd.snp_off;
      const long long ob = *((const long long*)(st_host.data() + st_obj) + slot_of[g]);
      std::fill(rh.tag.begin(), rh.tag.end(), 0);
      for (size_t k = 0; k < rh.fp_rows.size(); k++) rh.tag[rh.fp_rows[k]] = sg[k];
      for (int i = 0; i < rh.S; i++) { rh.cand[i].haplotype = dl[i]; rh.cand[i].genotype = et[i]; }
      objective[g] = (double)ob / FX_SCALE;
      const uint64_t S = rh.S, F = rd.R;
      rh.ctr = (uint32_t)rh.S <= prm.max_enum_snps ? S + F + ((uint64_t)1 << S) * F : 2 * (S + F) + (S / 4 + 1) * (S + F);
    }
  }

  lap("scatter");
  // ---- post-phase epilogue, thread.rs:168-201.  Regions are independent (the reference runs them as
  // rayon tasks, thread.rs:77): a small host thread pool walks them; results do not depend on the
  // thread count (per-region RNG stream, disjoint output rows).
  auto epilogue = [&](int g) {
    RegionHost& rh = R[g];
    if (rh.S == 0) return;
    rh.assign_reads_haplotype(prm.read_assign_cutoff);
    rh.assign_snp_haplotype_genotype();
    rh.assign_reads_haplotype(prm.read_assign_cutoff);
    rh.assign_snp_haplotype_genotype();
    const float relaxed = prm.min_phase_score - 3.0f;
    rh.eval_rescue(LCR_F_RNA_EDIT, relaxed, false);
    rh.eval_rescue(LCR_F_CAND_SOMATIC, relaxed, true);
    rh.assign_reads_haplotype(prm.read_assign_cutoff);
    rh.assign_snp_haplotype_genotype();
    rh.assign_phase_set(prm.min_phase_score, phase_set.data());
    for (int r = 0; r < rh.nrow; r++) { haplotag[rh.r0 + r] = rh.tag[r]; assignment[rh.r0 + r] = rh.asg[r]; }
  };
  for_regions(epilogue);
  lap("post-phase epilogue");
  return LCR_OK;
#undef PCHK
}
