// k4_kernels.h — what the host control of the phase stage (k4_phase.hip) shares with its kernel units
// (k4_enum.hip, k4_stage.hip, k4_post.hip): launch geometry, LDS layouts and the launchers.
#pragma once
#include "k4_types.h"

// ---- enumeration restarts (k4_enum.hip) ----
// The grid of an enumeration kernel is the concatenation of its regions' tiles; the host uploads one span per region
// (a few hundred) instead of one record per tile (tens of thousands), the workgroup finds its span with two rounds
// of a 64-way search (spans are sorted by tile0).  k4_enum_resolve and the winner re-runs of the global-memory class have one
// workgroup per span.
struct EnumSpan { int32_t slot; uint32_t tile0; };
constexpr int ENUM_WAVES = 4;
constexpr uint32_t ENUM_TILE_JOBS = 16;
constexpr uint32_t ENUM_LDS_BYTES = 48 * 1024;   // image of a region of the register / streaming classes (three workgroups per CU)
constexpr uint32_t ENUM_LDS_MAX = 63 * 1024;     // ... of the large-image streaming class (a launch of its own; 64 KB less the kernels' static arrays)

// LDS image: wl2[32] {lo23, hi24 (signed)} | lut64[64] (log10 eps_q, log10 (1 - eps_q): the f64 tie path) | csr[E] {lo | meta << 24,
//            hi | row_in_lane << 24} | csc[E] | rp[R+1] u16 | first_row[65] u16 | ent16[E] (row-order entries of the f64 tie paths) |
//            per wave: sigma bits (u64 words, +1 pad), M[32] and the queue of tied rows
//   csr meta : bits 0-4 SNP, 5 allele (1: p == +1), 6 last entry of its row, 7 valid
//   csc      : bits 0-15 row, 16-20 SNP, 21 allele, 22-26 q, 31 valid
//   ent16    : bits 0-4 SNP, 5 allele, 6-10 q, 11 last entry of its row
constexpr uint32_t ENUM_TQ = 126;     // tied rows a wave queues per sigma step (+ 2 words: the count)
constexpr uint32_t ENUM_TCAP = 256;   // configurations of maximal objective compared at a time (enum_resolve): a lane each
// k4_enum_bits (eight restarts per wave as bit states): per wave sigma of the eight restarts as a byte per row, M[state][32], the SNPs' masks
// [32], the per-restart masks [3][8] and the queue of tied rows
#ifndef ENUM_BITS_PER_V
#define ENUM_BITS_PER_V 64
#endif
constexpr uint32_t ENUM_BITS_PER = ENUM_BITS_PER_V;   // restarts per tile of k4_enum_bits (ENUM_WAVES waves x 8 x 2; -DENUM_BITS_PER_V: measurement builds)
__host__ __device__ inline uint32_t enum_bits_sp(uint32_t S) { return (S + 7) & ~7u; }   // SNP slots of M[state][]
__host__ __device__ inline uint32_t enum_bits_stride(uint32_t R, uint32_t S) { return ((R + 15) & ~7u) + 8 * 8 * enum_bits_sp(S) + 8 * 32 + 4 * 24 + 4 * (ENUM_TQ + 2); }
struct EnumLayout { uint32_t lut, csr, csc, rp, first_row, ent16, pos, state, stride, total; };
__host__ __device__ inline EnumLayout enum_layout(uint32_t R, uint32_t E, bool bits = false, uint32_t S = 32) {
  EnumLayout L;
  uint32_t o = 256;
  L.lut = o; o += 512;
  L.csr = o; o += 8 * (E + (bits ? 64 * (S < 31 ? S : 31) : 0));   // (k4_enum_bits: the rows of more than one entry in groups of 64 as long as their longest: <= 64 x the longest row of padding)
  L.csc = o; o += 4 * E;
  L.rp = o; o += 2 * (R + 1);
  L.first_row = o; o += 2 * 65 + (bits ? 2 * 64 : 0);   // (k4_enum_bits: + the entries of every lane's run that belong to rows of more than one entry)
  o = (o + 7) & ~7u;
  L.ent16 = o; o += 2 * ((E + 3) & ~3u);
  L.pos = o; if (bits) o += 2 * ((R + 3) & ~3u) + 2 * ((R + 4) & ~3u);   // (k4_enum_bits: the rows in the order [more than one entry | one entry] + the entry prefix of the former)
  o = (o + 15) & ~15u;
  L.state = o;
  L.stride = bits ? enum_bits_stride(R, S) : 8 * ((R + 63) / 64 + 1) + 8 * 32 + 4 * (ENUM_TQ + 2);
  o += ENUM_WAVES * L.stride;
  L.total = o;
  return L;
}
// words of one restart's saved final state: sigma bits | delta, eta == 0 masks | eta == +1 mask | signature of its term sequence
__host__ __device__ inline uint32_t enum_state_words(uint32_t R) { return (R + 63) / 64 + 3; }
// LDS image of k4_enum_resolve (one workgroup per region): table | rp u16 | ent16 | pse f64[E + 8] | sigma words of the reference
// configuration | rows with a het entry | rows by first het site [S][nk] | per compared configuration: restart, f64 objective |
// events of the reference's chain
// entries of k4_enum_resolve's list of the restarts of maximal objective: every restart of a region of up to 12 SNPs (round 6: the list held 2 048, and a
// sweep with max_enum_snps = 12 met a region with more maxima than that -- first maximum kept, counted); 4 096 beyond (counted when exceeded)
__host__ __device__ inline uint32_t resolve_tlcap(uint32_t S) { return S >= 12 ? 4096u : (S < 6 ? 64u : (1u << S)); }
struct ResolveLayout { uint32_t lut, rp, ent16, pse, sg_ref, hetw, repmask, rowsnps, res, total; };
__host__ __device__ inline ResolveLayout resolve_layout(uint32_t R, uint32_t E, uint32_t S) {
  ResolveLayout L;
  const uint32_t nk = (R + 63) / 64;
  uint32_t o = 0;
  L.lut = o; o += 512;
  L.rp = o; o += 2 * (R + 2);
  o = (o + 15) & ~15u;
  L.ent16 = o; o += 2 * ((E + 7) & ~7u);
  o = (o + 15) & ~15u;
  L.pse = o; o += 8 * ((E + 8) & ~7u);
  L.sg_ref = o; o += 8 * (nk + 1);
  L.hetw = o; o += 8 * (nk + 1);
  L.repmask = o; o += 8 * nk * (S ? S : 1);
  L.rowsnps = o; o += 4 * (R + 1);        // the SNPs of a row's entries as a mask
  o = (o + 7) & ~7u;
  L.res = o; o += ENUM_TCAP * 16 + 96 * 10 + 16 + 2 * resolve_tlcap(S);   // + the list of the restarts of maximal objective
  L.total = (o + 15) & ~15u;
  return L;
}
// lane l owns the rows whose first entry index lies in [l*c, (l+1)*c), c = ceil(E / 64)
__host__ __device__ inline uint32_t enum_chunk(uint32_t E) { return E ? (E + 63) / 64 : 1; }

// CK = 32 | 0: k4_enum_reg<CK> (the per-lane share of the region's entries held in registers; 0 = streamed from LDS); -1: k4_enum_bits
// st_base[slot]: first word of the region's 2^S saved states in st_words
void launch_k4_enum_reg(int ck, unsigned n_blocks, size_t dyn_lds, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, int32_t n_spans,
                        uint32_t per, const int64_t* job_base, long long* job_obj, const int64_t* st_base, unsigned long long* st_words,
                        long long* region_best, uint32_t* redo /* repair list: [0] count (zeroed), [4 ..] (slot, restart) pairs; nullptr: none */, uint32_t redo_cap);
void launch_k4_enum_redo(unsigned n_blocks, hipStream_t s, const PhaseDev& P, const uint32_t* redo, uint32_t redo_cap, int8_t* scratch /* n_blocks x stride */,
                         int32_t scratch_stride, double* qrow /* n_blocks x qrow_stride */, int64_t qrow_stride, const int64_t* job_base, long long* job_obj,
                         const int64_t* st_base, unsigned long long* st_words, uint32_t lds_bytes /* dynamic LDS: state + matrix of a restart's region, where they fit */);
void launch_k4_enum_resolve(unsigned n_regions, size_t dyn_lds, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, const int64_t* job_base,
                            const long long* job_obj, const int64_t* st_base, const unsigned long long* st_words);
void launch_k4_enum_big(unsigned n_blocks, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, int32_t n_spans, uint32_t per,
                        const int64_t* job_base, long long* job_obj, const uint32_t* win_e /* nullptr: all restarts; else the winners once more */,
                        const int64_t* st_base /* per region: first word of its saved states, < 0: not kept */, unsigned long long* st_words,
                        double* qrow /* n_blocks x qrow_stride doubles (2 per row of the largest region): the complete tie contract, or nullptr */, int64_t qrow_stride);
void launch_k4_enum_resolve_big(int32_t n, hipStream_t s, const PhaseDev& P, const EnumSpan* spans, const int64_t* job_base, const long long* job_obj,
                                const int64_t* st_base, const unsigned long long* st_words, uint32_t* win_e, double* terms /* n x terms_stride */,
                                int64_t terms_stride);

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
