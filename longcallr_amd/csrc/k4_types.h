// k4_types.h — plain kernel-argument structs of the K4 kernels (shared by k4_phase.hip, k4_grid.hip and the host driver).
#pragma once
#include "lcr_dev.h"

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
  int32_t lds_mat;                                                        // bytes of dynamic LDS behind the state for a matrix copy
  PhaseLutDev lut;
  // Decision arithmetic (round 4, DESIGN.md "Decision arithmetic"): every decision by the exact fixed-point sums; a decision whose
  // sums TIE exactly is taken by the reference-order f64 ratio scores of that row / configuration (phase.rs:77-96, 257-276) --
  // lut64 = the host's libm values of log10(eps_q), log10(1 - eps_q); tie_ctr = the census of the ties met (TIE_* below)
  const struct PostLut* lut64;
  unsigned long long* tie_ctr;
  int32_t tie_arith;   // 0: fixed point only (ties change nothing), 1: + configurations of equal objective by their f64 sums, 2: + sigma ties by the f64 scores,
                       // 3: + (enumeration kernels) delta / eta ties at the maximum and the verdict of tie-only steps
  int32_t pad_;
};
// census of exact fixed-point ties of one lcr_phase call (lcr_get_tie_census): the RESOLVED classes follow the reference's f64
// arithmetic; an UNRESOLVED count other than zero means a decision fell to "a tie changes nothing" where the reference's f64
// rounding noise might have decided otherwise
enum { TIE_SIGMA_F64 = 0,      // sigma decisions with A == B at a row with an entry at a het site: decided by the f64 scores
       TIE_SIGMA_FLIPS = 1,    // ... of which flipped (q < qn)
       TIE_DELTA_UNRES = 2,    // delta / eta choices with a tie at the maximum (phase.rs:905-940): first maximum kept (chain regions: only with chain_ties = 0)
       TIE_STEP_UNRES = 3,     // steps whose only changes were tie changes (check_new_*, phase.rs:278-355): taken as "no improvement"
       TIE_BEST_F64 = 4,       // regions whose configurations of maximal objective differ: `prob > largest_prob` by the f64 sums (k4_chain_wg: per compare)
       TIE_BEST_UNRES = 5,     // ... left to "first maximum wins" (fallback kernels)
       TIE_SIGMA_UNRES = 6,    // sigma ties in kernels without the f64 path
       TIE_STEP_F64 = 7,       // delta / eta ties at the maximum + tie-only steps decided by the f64 scores (enumeration kernels, round 5; k4_chain_wg<COMPLETE>, round 6)
       TIE_NCTR = 8 };
// (-DENUM_PROF, a measurement build: the census slots carry k4_enum_resolve's times instead)
#ifdef ENUM_PROF
#define TIE_COUNT(ctr, which, n) do { } while (0)
#else
#define TIE_COUNT(ctr, which, n) atomicAdd(&(ctr)[which], (n))
#endif

// per-region sizes k4_stage reports to the host
struct StageStat { int32_t R, E, max_n, max_rows, E_all, W; };   // max_*: per-lane share of k4_enum_reg's row partition; E_all: all entries;
                                                                  // W: largest distance (in SNP indices) between two for_phasing entries of one fragment row
struct PostLut { double le[31], l1e[31]; double p_homref, p_homvar, log_theta, log2; };


// ---- chain regions on the device (k4_grid.hip): LD blocks, LD-seeded start, block-flip pass, perturbation rounds
struct ChainDesc {
  int32_t slot;        // region index
  int32_t W;           // band width of the pair table (>= 1)
  int64_t tbl_off;     // pair table of the region: S x W cells of {cis, trans} u32 counters, offset in cells
  int64_t adj_off;     // adjacency lists (<= 2 S W entries), offset in entries
  int64_t part_off;    // per (row part, SNP) counters of the ordered column index, offset in int32
  int32_t n_parts;
  int32_t fast_lds;    // grid scope: bytes of dynamic LDS for the device-coherent perturbation rounds (0: generic path)
  int32_t batch_lds;   // grid scope: bytes of dynamic LDS for the batched rounds (k4_grid_batch.h; 0: not for this region)
  int32_t term_off;    // workgroup scope: the region's slice of ChainDev::tie_terms, in entries
};
struct GridCtl { unsigned arrive, gen, flag[2]; unsigned long long acc[2]; int slot; unsigned pad_[7]; };   // grid barrier + reductions
struct ChainDev {
  PhaseDev P;                     // phase matrices; st_* = best / result state of every region
  const ChainDesc* desc;
  // K3's fragment matrix and the candidates (pair counts and the flip veto look at every entry of a row)
  const int64_t* row_ptr; const int32_t* col; const uint8_t* val;
  const lcr_candidate* cand; const int32_t* cand_off; const int32_t* row_region_off;
  const int32_t* prow_src;        // phasing row -> fragment row (region relative), at sig_off + k
  uint32_t* ld_tbl; int32_t* ld_adj; int32_t* part_cnt;
  // per SNP (at snp_off + i, *_ptr arrays at cp_off + i)
  int32_t* adj_ptr; int32_t* blk_ptr; int32_t* blk_of; int32_t* blk_pos; int32_t* blk_nodes;
  int32_t* stack; int32_t* queue; uint8_t* seen; uint8_t* ld_ok; int8_t* new_hap;
  double* qs; double* qfs;
  int32_t* blk_info;              // per region: number of blocks, flip verdict of the last block
  // per phasing row (at sig_off + k) / per phase entry (at e_off + e)
  int32_t* flipcol; int32_t* erow; int32_t* cent;
  // working state of the grid path (global memory; the one-workgroup path keeps it in LDS)
  int8_t* w_sigma; int8_t* w_delta; int8_t* w_eta; unsigned long long* macc;
  unsigned long long* sig_words;  // grid scope: sigma as bit vectors, 2 x ceil(R / 64) words per region at 2 * (sig_off / 64 + region)
  GridCtl* ctl;
  // speculative rounds at grid scope (k4_grid.hip chain_rounds_fast): spec_lanes sub-grids, each with a working state of its own
  // (sigma words: spec_ng per lane; delta | eta bytes: 2 x spec_s8 per lane), its result and its barrier words
  unsigned long long* spec_sig; int8_t* spec_de; long long* spec_res; GridCtl* spec_ctl;
  int32_t spec_lanes, spec_ng, spec_s8;
  // the region's phase entries as one dword each -- value byte << 24 | SNP (row order) / value byte << 24 | row (column order) --
  // so that the rounds read four entries per 16-byte load; pk_cap entries each (+ 8 of padding), filled when the rounds begin
  uint32_t* pk_csr; uint32_t* pk_csc; int64_t pk_cap;
  // batched rounds (k4_grid_batch.h): groups of four entries of one row (bt_cap4 groups), {first group, rounds} and best sigma bits of every block
  // of 64 sorted rows, the sort's permutation and its inverse, sigma bits of the eight states per row, delta / eta bit masks per SNP,
  // barrier payload and counters (K4_GRID_BATCH_CTL_BYTES, zeroed by the launcher)
  uint32_t* bt_pk4; int32_t* bt_up4; uint32_t* bt_bs32; int32_t* bt_perm; int32_t* bt_inv; uint8_t* bt_sig8; uint32_t* bt_m32; void* bt_ctl; int64_t bt_cap4; int32_t spec_batch;
  // ties of classes 2 / 4 / 8 at workgroup scope (k4_chain_wg): tie_flag != nullptr = resolve them (nullptr: count them as unresolved; the
  // array itself is spare), and the scratch of the complete contract -- two f64 scores per phasing row (at 2 (sig_off + k)) and per SNP (at 2 (snp_off + i)), two choice bytes per SNP
  int32_t* tie_flag; double* tie_qrow; double* tie_qsnp; int8_t* tie_ch;
  double* tie_terms;              // class 8: the f64 terms of two configurations, entry by entry (2 per phase entry, at 2 (term_off + e))
  long long* dbg;                 // LCR_PHASE_PROF: 100 MHz timestamps of the chain steps of a grid launch, 16 per launch (else nullptr)
  double le[31], l1e[31], p_homref, p_homvar, log_theta, log2;   // libm values of the block-flip sums (host table)
};


// ---- staging of the phase matrices (k4_stage, k4_phase.hip; k4_stage_grid, k4_grid.hip)
struct StageIn {
  const int64_t* row_ptr; const int32_t* col; const uint8_t* val; const uint32_t* links;
  const lcr_candidate* cand; const int32_t* cand_off; const int32_t* row_region_off; const int64_t* start0;
  uint32_t min_linkers, max_enum_snps; uint64_t seed;
  int64_t grid_min;    // regions with at least this many fragment entries are staged by k4_stage_grid
};
struct StageOut {
  RegionDev* reg; StageStat* stat;
  int32_t* prow_ptr; int32_t* pcol; uint8_t* pval; int32_t* ccol_ptr; int32_t* crow; uint8_t* cval;
  uint8_t* snp_fp; int8_t* snp_vt; uint8_t* snp_cons; long long* snp_const; int32_t* cursor;
  int32_t* prow_src;   // per phasing row (at r0 + k): its fragment row, region relative
};
// HBM image of ONE region for k4_gpost (post-phase steps with all CUs on the region)
struct PostScratch {
  double *sps, *rpa, *rpb; uint32_t *sflags, *soflags; int32_t *parent, *ccptr;
  int32_t *rptr, *ecol, *erow, *cent; uint8_t* ev;
  int8_t* tag; uint8_t *asg, *fp, *lok, *dirty; int8_t *shap, *sgt, *svt; uint8_t* rcode;
  int32_t* pcnt; int32_t n_parts, pad_;
  int32_t *fdirt, *minf, *ndraw, *gwords;   // rescue lists: first member to change a row, a member's row minimum, its draws; {round end}
  GridCtl* ctl;
};
// arguments of the post-phase kernels (k4_post: one workgroup per region; k4_gpost: all CUs on one region)
struct PostIn {
  const int64_t* row_ptr; const int32_t* col; const uint8_t* val; const uint32_t* links;
  lcr_candidate* cand; const int32_t* cand_off; const int32_t* row_region_off; const int64_t* start0;
  const int8_t* st_sigma; const int8_t* st_delta; const int8_t* st_eta;
  int8_t* haplotag; uint8_t* assignment; uint32_t* phase_set;   // per-row results: pinned host memory, written by the kernel
  uint32_t* d_rec;   // the same as 12-byte records in HBM (lcr_read_record: row, haplotag | assignment << 8, phase set) for consumers on the device (multi-GPU gather)
  const long long* st_obj; long long* h_obj; lcr_candidate* h_cand;   // objective / candidate mirror in pinned host memory
  uint32_t min_linkers, max_enum_snps; uint64_t seed; double cutoff; float min_phase_score;
  long long* dbg_clk;   // LCR_PHASE_PROF: 100 MHz timestamps of every workgroup's steps, 16 per region (nullptr otherwise)
  const RegionDev* reg; const int32_t* prow_src;   // phasing rows of the region (k4_stage): count, and their fragment rows
};
