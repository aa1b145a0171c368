// k4_kernels.h — what the host control of the phase stage (k4_phase.hip) shares with its kernel units
// (k4_enum.hip, k4_stage.hip, k4_post.hip): launch geometry, LDS layouts and the launchers.
#pragma once
#include "k4_types.h"

// ---- enumeration restarts (k4_enum.hip) ----
// The grid of an enumeration kernel is the concatenation of its regions' tiles; the host uploads one span per region
// (a few hundred) instead of one record per tile (tens of thousands), the workgroup finds its span with two rounds
// of a 64-way search (spans are sorted by tile0).  Winner re-runs have one workgroup per span.
struct EnumSpan { int32_t slot; uint32_t tile0; };
constexpr int ENUM_WAVES = 4;
constexpr uint32_t ENUM_TILE_JOBS = 16;
constexpr uint32_t ENUM_LDS_BYTES = 48 * 1024;

// LDS image: wl2[32] {lo23, hi24 (signed)} | lut64[64] (log10 eps_q, log10 (1 - eps_q): the f64 tie path) | csr[E] {lo | meta << 24,
//            hi | row_in_lane << 24} | csc[E] | rp[R+1] u16 | first_row[65] u16 | qrow[E] u8 (q of the entries in row order) |
//            per wave: sigma bits (u64 words, +1 pad) and M[32] | scratch of the region's last tile (enum_resolve)
//   csr meta : bits 0-4 SNP, 5 allele (1: p == +1), 6 last entry of its row, 7 valid
//   csc      : bits 0-15 row, 16-20 SNP, 21 allele, 22-26 q, 31 valid
constexpr uint32_t ENUM_TCAP = 128;   // configurations of maximal objective compared at a time (enum_resolve)
struct EnumLayout { uint32_t lut, csr, csc, rp, first_row, qrow, state, stride, res, total; };
__host__ __device__ inline EnumLayout enum_layout(uint32_t R, uint32_t E) {
  EnumLayout L;
  uint32_t o = 256;
  L.lut = o; o += 512;
  L.csr = o; o += 8 * E;
  L.csc = o; o += 4 * E;
  L.rp = o; o += 2 * (R + 1);
  L.first_row = o; o += 2 * 65;
  L.qrow = o; o += E;
  o = (o + 15) & ~15u;
  L.state = o;
  L.stride = 8 * ((R + 63) / 64 + 1) + 8 * 32;
  o += ENUM_WAVES * L.stride;
  L.res = o; o += ENUM_TCAP * 24;      // per compared configuration: restart, signature, f64 objective
  L.total = o;
  return L;
}
// words of one restart's saved final state: sigma bits | delta, eta == 0 masks | eta == +1 mask
__host__ __device__ inline uint32_t enum_state_words(uint32_t R) { return (R + 63) / 64 + 2; }
// lane l owns the rows whose first entry index lies in [l*c, (l+1)*c), c = ceil(E / 64)
__host__ __device__ inline uint32_t enum_chunk(uint32_t E) { return E ? (E + 63) / 64 : 1; }

// CK = 32 | 0: k4_enum_reg<CK> (the per-lane share of the region's entries held in registers; 0 = streamed from LDS)
// st_base[slot]: first word of the region's 2^S saved states in st_words
void launch_k4_enum_reg(int ck, unsigned n_blocks, size_t dyn_lds, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, int32_t n_spans,
                        uint32_t per, const int64_t* job_base, long long* job_obj, const int64_t* st_base, unsigned long long* st_words, uint32_t* done);
void launch_k4_enum_big(unsigned n_blocks, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, int32_t n_spans, uint32_t per,
                        const int64_t* job_base, long long* job_obj, const uint32_t* win_e);
void launch_k4_enum_pick(int32_t n, hipStream_t s, const EnumSpan* spans, const RegionDev* reg, const int64_t* job_base, const long long* job_obj,
                         uint32_t* win_e, unsigned long long* tie_ctr);

// ---- phase matrices (k4_stage.hip) ----
constexpr int STAGE_THREADS = 256;    // (1024 threads per region were measured: more barrier cost than latency saved)
constexpr int STG_E = 8192, STG_R = 4096, STG_S = 512;   // k4_stage: a region's slice of the fragment matrix that is staged in LDS
void launch_k4_stage(int32_t n_regions, hipStream_t s, const StageIn& in, const StageOut& out, const PhaseLutDev& lut);

// ---- post-phase steps, one workgroup per region (k4_post.hip) ----
constexpr int CHAIN_THREADS = 1024;   // k4_post of the chain regions: 16 waves
constexpr int POST_MAX_ROWS = 8192, POST_MAX_ENTRIES = 8192, POST_MAX_SNPS = 512;
struct PostLayout { uint32_t sps, rpa, rpb, sflags, soflags, parent, qcnt, rptr, ecol, erow, cent, ccptr, eval, tag, asg, fp, lok, dirty, shap, sgt, svt, rcode, total; };
__host__ __device__ inline PostLayout post_layout(uint32_t nrow, uint32_t E, uint32_t S) {
  PostLayout L;
  uint32_t o = 64 * 8;                       // le[32] | l1e[32]
  L.sps = o; o += 8 * S;                     // phase_score
  L.rpa = o; o += 8 * S; L.rpb = o; o += 8 * S;   // rescue: the two candidate phase scores
  L.sflags = o; o += 4 * S; L.soflags = o; o += 4 * S; L.parent = o; o += 4 * S;
  L.qcnt = o; o += 4 * 16 * S;                // per (row part, SNP): entry count, then fill cursor (<= 16 waves)
  L.rptr = o; o += 2 * (nrow + 2);
  L.ecol = o; o += 2 * E; L.erow = o; o += 2 * E; L.cent = o; o += 2 * E;
  L.ccptr = o; o += 2 * (S + 2);
  L.eval = o; o += E;
  L.tag = o; o += nrow; L.asg = o; o += nrow; L.fp = o; o += nrow; L.lok = o; o += nrow;
  L.dirty = o; o += nrow;                    // rescue: rows whose fp / tag changed in the current round
  L.shap = o; o += S; L.sgt = o; o += S; L.svt = o; o += S; L.rcode = o; o += S;
  L.total = (o + 15) & ~15u;
  return L;
}
// threads = CHAIN_THREADS | CHAIN_THREADS / 2 (= 2 * LCR_BLOCK)
static_assert(CHAIN_THREADS / 2 == 2 * LCR_BLOCK, "k4_post has two instantiations");
hipError_t launch_k4_post(int threads, unsigned n_blocks, size_t dyn_lds, hipStream_t s, const PostIn& in, const int32_t* slots, int32_t n_slots,
                          const PostLut& lut);
