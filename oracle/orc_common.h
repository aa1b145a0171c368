/*
 * oracle/orc_common.h — TEST INFRASTRUCTURE ONLY.
 *
 * Shared declarations of the CPU oracle: a structure-faithful restatement, written from the Rust
 * text, of longcallR v1.12.0's per-region hot path (pileup -> candidates -> fragments -> phasing).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
 * library (longcallr_amd/csrc) never includes, links or calls anything under oracle/.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference has no tests, golden vectors or expected
 * outputs, cargo/rustc are absent (no oracle/_ref can be built), and demo/chr20.fa is missing.
 * The oracle is pinned by (i) hand-derived known-answer tests (tests/test_oracle_kat.py) and
 * (ii) an independent NumPy restatement (oracle/oracle_np.py) written separately from the same
 * Rust source, compared on demo.bam and synthetic inputs.
 */
#ifndef ORC_COMMON_H
#define ORC_COMMON_H
#include <stdint.h>
#include "../include/lcr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Injected counter-based RNG replacing rand::thread_rng() (phase.rs:444,611,674,1198;
 * snpfrags.rs:256,349).  u01(seed, ctr) is a pure function so that a GPU can evaluate draw #ctr
 * without replaying the stream; the oracle simply draws ctr = 0,1,2,... in the reference's call
 * order. */
static inline uint64_t orc_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
static inline double orc_u01(uint64_t seed, uint64_t ctr) {
  uint64_t z = orc_mix64(seed + (ctr + 1) * 0x9E3779B97F4A7C15ULL);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
static inline uint64_t orc_region_seed(uint64_t seed, int64_t start0) {
  return orc_mix64(seed + 0xD1B54A32D192ED03ULL * (uint64_t)(start0 + 1));
}

/* decision arithmetic of the optimiser (see DESIGN.md "Decision arithmetic") */
#define ORC_MODE_F64 0   /* reference-order f64 running sums (faithful to phase.rs)            */
#define ORC_MODE_EXACT 1 /* exact fixed-point sums, scale 2^40 (liblcr's contract of rounds 1-3) */
/* Modes 0 / 1 evaluate BOTH arithmetics at every decision and count the decisions on which they disagree
 * (orc_get_stats [2]: rounding-noise ties).  The *_ONLY modes take the same decisions as 0 / 1 without the other
 * arithmetic: ORC_MODE_F64_ONLY is the reference's work, nothing more (bench.py's cpu_baseline); ORC_MODE_EXACT_ONLY
 * drops the libm calls of the f64 scores (full-size parity runs). */
#define ORC_MODE_F64_ONLY 2
#define ORC_MODE_EXACT_ONLY 3
/* ORC_MODE_TIE (round 4, liblcr's contract now): every decision by the exact fixed-point sums, and a decision whose
 * fixed-point sums TIE exactly is taken by the reference-order f64 ratio scores of that row / column / configuration --
 * the arithmetic of phase.rs on exactly the decisions where arithmetic matters (the rest are sign tests of sums that
 * differ macroscopically).  Four classes of ties, selectable with orc_set_tie_mask for the residual tables:
 *   1 sigma flips (phase.rs:845-858)          2 the delta / eta choice (phase.rs:905-940)
 *   4 "did the step improve" after an iteration whose only changes were tie changes (check_new_*, phase.rs:278-355)
 *   8 `prob > largest_prob` between configurations of equal fixed-point objective (phase.rs:1117,1129 ...)        */
#define ORC_MODE_TIE 4
static inline int orc_mode_fx(int mode) { return mode == 1 || mode == 3; }
static inline int orc_mode_both(int mode) { return mode < 2; }
static inline int orc_mode_tie(int mode) { return mode == 4; }

typedef struct orc_region orc_region;

orc_region* orc_region_create(const lcr_reads* reads, int32_t read_begin, int32_t read_end,
                              int64_t start0, int32_t len, const uint8_t* ref_window,
                              const lcr_params* params);
void orc_region_destroy(orc_region*);

void orc_pileup(orc_region*);     /* util.rs:621-949                                            */
void orc_candidates(orc_region*); /* candidate.rs:54-528                                        */
void orc_fragments(orc_region*);  /* fragment.rs:10-309                                         */
void orc_phase(orc_region*, int mode); /* thread.rs:162-166 + phase.rs:1087-1296               */
void orc_post_phase(orc_region*); /* thread.rs:168-201 (snpfrags.rs:191-733)                    */
/* Indexed gathers (round 4): n_threads >= 1 replaces the reference's per-fragment linear searches
 * (`for fe in list if fe.snp_idx == i`, phase.rs:890-898, snpfrags.rs:403-414 ...) by a position index built with the
 * fragments -- the SAME entries in the SAME order, libm log10 of the 62 possible emission values from a table of the
 * same libm values -- and runs the Jacobi steps of cross_optimize (independent per row / per SNP by construction,
 * phase.rs:859-862,943-946) on n_threads threads; every f64 sum keeps its order.  0 (default) = the structure-faithful
 * gathers (what bench.py's cpu_baseline times).  Equality of the two forms is a test (tests/test_oracle_batch.py). */
void orc_set_fast(orc_region*, int n_threads);
void orc_set_tie_mask(orc_region*, int mask);   /* ORC_MODE_TIE: which tie classes the f64 scores resolve (default 15); mask | chain_mask << 8 | 1 << 16
                                                 * gives the chain branch (S > max_enum_snps) a mask of its own */
/* tie census of the last orc_phase (any mode that evaluates the fixed-point sums): [0] sigma decisions with A == B,
 * [1] delta/eta decisions with a tie at the maximum, [2] steps whose only changes were tie changes, [3] best-configuration
 * compares at equal fixed-point objective, [4..7] the same four counted only where the f64 scores then decide differently
 * from "a tie changes nothing", [8] sigma ties of rows with an entry at a het site, [9] sigma ties whose two f64 scores differ */
void orc_get_tie_census(const orc_region*, int64_t* out10);

/* getters (flat copies, same formats as include/lcr.h) */
void orc_get_planes(const orc_region*, uint32_t* out /* LCR_NPLANES*len */);
/* per-column clamped quality list of one allele (BaseQual, util.rs:71-77), read order */
int32_t orc_get_baseq(const orc_region*, int32_t col, int allele /*0..3*/, uint8_t* out, int32_t cap);
int32_t orc_n_cand(const orc_region*);
void orc_get_cands(const orc_region*, lcr_candidate* out);
/* hist-based evaluation of the genotype-likelihood block for candidate i (the order-free form the
 * GPU uses): loglik[3], gt_prob[3], qual, gq */
void orc_cand_gt_hist(const orc_region*, int32_t col, double* out8);
int32_t orc_n_rows(const orc_region*);
int64_t orc_nnz(const orc_region*);
void orc_get_fragmat(const orc_region*, int64_t* row_ptr, int32_t* row_read, int32_t* col,
                     uint8_t* val, uint8_t* row_for_phasing, uint32_t* row_links);
/* LD blocks (candidate.rs:615-747): returns number of blocks; block b's members are
 * members[off[b]..off[b+1]) in petgraph DFS order */
int32_t orc_get_ld_blocks(const orc_region*, int32_t* off, int32_t* members, int32_t cap);
void orc_get_phase(const orc_region*, int8_t* haplotag, uint8_t* assignment, uint32_t* phase_set,
                   double* objective);
/* counters: [0] cross_optimize calls, [1] total iterations, [2] f64-vs-exact decision
 * disagreements seen (rounding-noise ties), [3] monotonicity assert violations */
void orc_get_stats(const orc_region*, int64_t* out4);
/* first call (before orc_phase): start logging which half-rounds of the perturbation loop (phase.rs:1198-1233) raise the best
 * objective; later calls return the flags (what a speculative execution of several half-rounds at once can count on) */
int32_t orc_round_log(orc_region*, uint8_t* out, int32_t cap);
/* VCF body text of this region (vcf.rs:27-306 + thread.rs:266-303); returns length */
int64_t orc_vcf_text(orc_region*, const char* chrom, char* buf, int64_t cap);

/* ---- a whole batch on a native thread pool: the analogue of the reference's rayon par_iter over regions
 * (thread.rs:77); no Python per region.  upto: 0 = pileup, 1 = + candidates, 2 = + fragments, 3 = + phase and
 * post-phase.  Results are kept per region and handed out concatenated in batch order, in the formats of
 * include/lcr.h (what the lcr_get_* calls of the HIP path return for the same batch).  keep_planes = 0 drops the
 * pileup columns of a region as soon as its candidates are known (memory of the full-size configs). */
typedef struct orc_batch orc_batch;
orc_batch* orc_run_batch(const lcr_reads* reads, const lcr_regions* regions, const lcr_params* params, int mode,
                         int n_threads, int upto, int keep_planes);
/* the same with the indexed gathers on fast_threads threads per region (orc_set_fast) and a tie mask (orc_set_tie_mask) */
orc_batch* orc_run_batch_opts(const lcr_reads* reads, const lcr_regions* regions, const lcr_params* params, int mode,
                              int n_threads, int upto, int keep_planes, int fast_threads, int tie_mask);
void orc_batch_tie_census(const orc_batch*, int64_t* out /* n_regions x 10, as orc_get_tie_census */);
void orc_batch_destroy(orc_batch*);
double orc_batch_seconds(const orc_batch*);        /* wall time of the pool                                   */
int32_t orc_batch_threads(const orc_batch*);
/* cand_off / row_off: n_regions + 1 int32; nnz_off: n_regions + 1 int64 (entries of the fragment matrix) */
void orc_batch_offsets(const orc_batch*, int32_t* cand_off, int32_t* row_off, int64_t* nnz_off);
void orc_batch_planes(const orc_batch*, uint32_t* out /* LCR_NPLANES x n_cols, plane-major over the batch */);
void orc_batch_cands(const orc_batch*, lcr_candidate* out);   /* region field = region index */
/* fragment matrix as it stood after get_fragments (the rescue steps of post-phase change for_phasing):
 * row_ptr batch-wide (n_rows + 1), col = candidate index inside the region */
void orc_batch_fragmat(const orc_batch*, int64_t* row_ptr, int32_t* row_read, int32_t* col, uint8_t* val,
                       uint8_t* row_for_phasing, uint32_t* row_links);
void orc_batch_phase(const orc_batch*, int8_t* haplotag, uint8_t* assignment, uint32_t* phase_set, double* objective /* n_regions */);
void orc_batch_stats(const orc_batch*, int64_t* out /* n_regions x 4, as orc_get_stats */);
/* VCF text of all regions back to back; off[g] .. off[g + 1] = region g's records; returns the total length */
int64_t orc_batch_vcf(orc_batch*, const char* chrom, char* buf, int64_t cap, int64_t* off);
orc_region* orc_batch_region(orc_batch*, int32_t g);   /* for the per-region getters (LD blocks ...) */

/* scalar functions exposed for known-answer tests */
float orc_strand_odds_ratio(int ref_fw, int ref_rv, int alt_fw, int alt_rv); /* candidate.rs:24-35 */
double orc_binomial_two_tailed(uint64_t successes, uint64_t trials);       /* candidate.rs:37-47 */
void orc_two_major_alleles(const uint32_t cnt[4], uint8_t ref_base, uint8_t* a1, uint32_t* c1,
                           uint8_t* a2, uint32_t* c2);                       /* util.rs:162-176   */
double orc_aki(int sigma, int delta, int eta, int p, double err);            /* phase.rs:32-49    */
double orc_cal_sigma_delta_eta_log(int sigma_k, int n, const int* delta, const int* eta,
                                   const int* ps, const double* probs);      /* phase.rs:77-96    */
double orc_cal_delta_eta_sigma_log(int delta_i, int eta_i, int n, const int* sigma, const int* ps,
                                   const double* probs);                     /* phase.rs:128-176  */
double orc_cal_phase_score_log(int delta_i, int eta_i, int n, const int* sigma, const int* ps,
                               const double* probs);                         /* phase.rs:238-255  */

#ifdef __cplusplus
}
#endif
#endif
