/*
 * lcr.h — C ABI of liblcr: the MI355X-native replacement for longcallR's per-region
 * pileup -> candidate/genotype -> read x SNP fragment matrix -> haplotype phasing hot path.
 *
 * The reference (huangnengCSU/longcallR v1.12.0, Rust) has no FFI seam; the seam is cut at the
 * five call sites of the per-region closure src/thread.rs:77-222.  Each entry point below names
 * the reference method it replaces.  BAM/FASTA decode, the CLI and the VCF/BAM writers stay on
 * the host: the ABI takes *decoded, already filtered* reads (filters of src/util.rs:652-668 /
 * src/fragment.rs:32-49 are the caller's job), grouped by region.
 *
 * Batched form: the reference runs one rayon task per region (src/thread.rs:77).  A GPU launch
 * per region would be launch-bound, so every entry point takes a *batch* of regions; results are
 * per region and independent of batch composition.
 *
 * Conventions: plain pointers + sizes, no C++/torch types.  All functions return 0 (LCR_OK) or a
 * negative LCR_E_* code and never abort/throw across the boundary (the reference panics instead).
 * A ctx is single-threaded; create one per host worker thread (mirrors one rayon worker).  Output
 * pointers handed back by lcr_get_* point into ctx-owned pinned host buffers that stay valid until
 * the next call of the same getter on that ctx or lcr_ctx_destroy.
 */
#ifndef LCR_H
#define LCR_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LCR_OK 0
#define LCR_E_ARG (-1)     /* bad argument / inconsistent sizes                                   */
#define LCR_E_CIGAR (-2)   /* unknown CIGAR op (reference: panic, util.rs:944, fragment.rs:192)    */
#define LCR_E_DEVICE (-3)  /* HIP runtime error (message via lcr_last_error)                       */
#define LCR_E_STATE (-4)   /* call order violated (e.g. lcr_candidates before lcr_pileup)          */
#define LCR_E_NOMEM (-5)
#define LCR_W_HW_QUEUES 1  /* lcr_ctx_set_async_phase(on): accepted, but the process has fewer than 8 hardware queues (below) */

#define LCR_PLATFORM_HIFI 0 /* main.rs:35-38 Platform::Hifi */
#define LCR_PLATFORM_ONT 1  /* Platform::Ont  */

#define LCR_MEM_HOST 0   /* pointers in lcr_reads / lcr_regions are host memory (copied H2D)       */
#define LCR_MEM_DEVICE 1 /* pointers are device (HBM) memory, used in place                         */

typedef struct lcr_ctx lcr_ctx;

/* ---- inputs ------------------------------------------------------------------------------- */

/* Decoded reads, SoA, borrowed for the duration of the call.  Reads are grouped by region
 * (lcr_regions.read_begin) and inside a region keep BAM order (sorted by pos).  A read that a
 * caller fetches for two regions appears twice.
 * flags: bit0 = reverse strand; bits1-2 = `ts` aux tag: 0 absent, 1 '+', 2 '-' (util.rs:674-680). */
typedef struct {
  int32_t mem;              /* LCR_MEM_HOST or LCR_MEM_DEVICE                                       */
  int32_t n_reads;
  int64_t n_bases;          /* total bytes in bases/quals                                          */
  int64_t n_cigar;          /* total u32 in cigar                                                  */
  const int32_t* pos;       /* 0-based leftmost reference position (record.pos())                  */
  const int32_t* seq_len;   /* l_seq                                                               */
  const int32_t* lead_clip; /* cigar.leading_softclips()  (looks past a hard clip)                 */
  const int32_t* trail_clip;/* cigar.trailing_softclips()                                          */
  const uint8_t* flags;
  const uint64_t* seq_off;  /* offset of read's first base in bases/quals                          */
  const uint64_t* cig_off;  /* offset of read's first op in cigar                                  */
  const uint32_t* n_cig;
  const uint8_t* bases;     /* decoded upper-case ASCII, as htslib yields (=ACMGRSVTWYHKDBN)       */
  const uint8_t* quals;     /* raw phred                                                           */
  const uint32_t* cigar;    /* BAM encoding len<<4|op, ops MIDNSHP=X (P and B are LCR_E_CIGAR)     */
} lcr_reads;

/* A batch of regions.  Region i covers 0-based reference columns [start0[i], start0[i]+len[i])
 * (= util.rs:639-642: vec_size = end-start, first column = start-1) and owns reads
 * [read_begin[i], read_begin[i+1]).  ref holds the reference bytes of every window back to back
 * (case preserved, util.rs:646-648); window i starts at ref + col_off[i]. */
typedef struct {
  int32_t mem;
  int32_t n_regions;
  const int64_t* start0;    /* n_regions                                                            */
  const int32_t* len;       /* n_regions                                                            */
  const int64_t* col_off;   /* n_regions+1 prefix sums of len                                       */
  const int32_t* read_begin;/* n_regions+1                                                          */
  const uint8_t* ref;       /* col_off[n_regions] bytes                                             */
} lcr_regions;

/* Thresholds; the code values of main.rs:272-396 are the spec (see lcr_params_preset). */
typedef struct {
  int32_t platform;         /* LCR_PLATFORM_*                                                      */
  uint32_t min_baseq;       /* main.rs min_baseq (10)                                              */
  uint32_t dist_to_end;     /* distance_to_read_end                                                */
  uint32_t polya_len;       /* polya_tail_length (<= 16)                                           */
  uint32_t min_depth, max_depth;
  uint32_t min_qual;        /* min_variant_qual                                                    */
  uint32_t dense_win, min_dense_cnt;
  uint32_t low_cnt_cut;     /* low_allele_cnt_cutoff                                               */
  uint32_t min_linkers;
  uint32_t max_enum_snps;
  uint32_t ld_weight_threshold; /* thread.rs:166 passes 1                                           */
  int32_t use_strand_bias;
  float min_af;             /* min_allele_freq                                                     */
  float min_af_intron;      /* min_allele_freq_include_intron                                      */
  float low_frac_cut;       /* low_allele_frac_cutoff                                              */
  float min_phase_score;
  double read_assign_cutoff;/* min_read_assignment_diff                                            */
  uint64_t seed;            /* replaces rand::thread_rng() (phase.rs:444,611,674,1198)             */
} lcr_params;

/* preset: 0 hifi-isoseq, 1 hifi-masseq, 2 ont-cdna, 3 ont-drna (main.rs:272-396). */
int lcr_params_preset(int preset, lcr_params* out);

/* ---- outputs ------------------------------------------------------------------------------ */

/* Pileup columns (replaces Vec<BaseFreq>, util.rs:100-127) as u32 planes of n_cols each.
 * plane[k] = planes + k*n_cols.  Reverse-strand counts are cnt - fwd.  The never-read fields of
 * BaseFreq (forward_cnt, backward_cnt, distance_to_end, i) are not produced. */
enum {
  LCR_PL_A = 0, LCR_PL_C, LCR_PL_G, LCR_PL_T, /* a,c,g,t                                            */
  LCR_PL_N,                                    /* n  (intron)                                        */
  LCR_PL_D,                                    /* d  (deletion)                                      */
  LCR_PL_NI,                                   /* ni (insertion after this column)                   */
  LCR_PL_FWD_A, LCR_PL_FWD_C, LCR_PL_FWD_G, LCR_PL_FWD_T, /* base_strands.x[0]                      */
  LCR_PL_TS_FWD, LCR_PL_TS_REV,               /* transcript_strands[0], [1]                         */
  LCR_NPLANES
};
typedef struct {
  int64_t n_cols;
  const uint32_t* planes;   /* LCR_NPLANES * n_cols                                                 */
} lcr_columns;

/* One candidate site (replaces CandidateSNP, snp.rs:39-90).  Array sorted by (region, pos). */
enum {
  LCR_F_RNA_EDIT = 1, LCR_F_DENSE = 2, LCR_F_HET = 4, LCR_F_FOR_PHASING = 8, LCR_F_HOM = 16,
  LCR_F_SINGLE = 32, LCR_F_NON_SELECTED = 64, LCR_F_CAND_SOMATIC = 128
};
typedef struct {
  int64_t pos;              /* 0-based                                                              */
  int32_t region;
  uint8_t ref_base, allele1, allele2, n_alt; /* ASCII; n_alt = alternate_alleles.num               */
  uint32_t cnt1, cnt2, depth;
  float af1, af2;           /* allele_freqs (f32, candidate.rs:97-98)                               */
  int32_t variant_type;     /* 0 homref 1 het 2 homvar 3 triallelic                                 */
  int32_t genotype;         /* eta: -1 homvar, 0 het, 1 homref                                      */
  int32_t haplotype;        /* delta: +1/-1, 0 unassigned                                           */
  uint32_t flags;           /* LCR_F_*                                                              */
  uint32_t phase_set;
  double loglik[3];         /* log10 L: [0] homvar [1] het [2] homref (candidate.rs:267-282)        */
  double gt_prob[3];        /* genotype_probability                                                 */
  double qual;              /* variant_quality                                                      */
  double gq;                /* genotype_quality                                                     */
  double phase_score;
} lcr_candidate;
typedef struct {
  int32_t n_cand;
  int32_t n_regions;
  const lcr_candidate* cand;
  const int32_t* region_off; /* n_regions+1: candidates of region i are [off[i], off[i+1])          */
} lcr_candidate_list;

/* Read x SNP fragment matrix (replaces Vec<Fragment>, snp.rs:197-239) in CSR.  Row k = k-th
 * fragment pushed by get_fragments (fragment.rs:293-307), i.e. every read of the region that starts
 * at or before the region's last candidate, empty rows included.
 * val: bits0-4 q (clamped to 30), bit5 = 1 if p=+1 (ref) / 0 if p=-1 (alt), bits6-7 base code ACGT,
 * col: index into lcr_candidate_list.cand (global, batch-wide). */
typedef struct {
  int32_t n_rows;
  int64_t nnz;
  int32_t n_regions;
  const int32_t* row_region_off; /* n_regions+1                                                     */
  const int64_t* row_ptr;        /* n_rows+1                                                        */
  const int32_t* row_read;       /* n_rows: index into lcr_reads                                    */
  const int32_t* col;            /* nnz                                                             */
  const uint8_t* val;            /* nnz                                                             */
  const uint8_t* row_for_phasing;/* n_rows                                                          */
  const uint32_t* row_links;     /* n_rows: num_hete_links                                          */
} lcr_fragmat;

/* Phasing result: candidates updated in place (haplotype, genotype, variant_type, phase_score,
 * phase_set, flags) plus per row sigma / assignment / phase set. */
typedef struct {
  int32_t n_rows;
  int32_t n_regions;
  const int8_t* haplotag;    /* sigma in {-1,0,1}                                                   */
  const uint8_t* assignment; /* 0 unassigned, 1 hap1, 2 hap2 (snpfrags.rs:548-625)                   */
  const uint32_t* phase_set; /* read -> PS (snpfrags.rs:628-733), 0 = none                           */
  const double* objective;   /* n_regions: best cal_overall_probability (phase.rs:257-276)           */
} lcr_phase_result;

/* ---- entry points ---------------------------------------------------------------------------- */

int lcr_ctx_create(int device, lcr_ctx** out);
void lcr_ctx_destroy(lcr_ctx*);
const char* lcr_last_error(const lcr_ctx*);
/* Launch on an existing hipStream_t (e.g. torch's current stream); NULL = ctx-owned stream. */
int lcr_ctx_set_stream(lcr_ctx*, void* hip_stream);
int lcr_ctx_sync(lcr_ctx*);

/* Bind a batch (reads + regions).  LCR_MEM_HOST inputs are copied to HBM here (the call returns when the copies are done) --
 * page-locked arrays (lcr_host_alloc / lcr_host_register) straight by DMA, pageable ones through page-locked staging buffers of
 * the context, 8 MB at a time (four lanes on threads of their own for uploads of 64 MB and more: ~47 GB/s), not through the runtime's
 * pin-the-caller's-pages path, which raised device memory faults on ROCm 7 / MI355X.  LCR_MEM_DEVICE inputs are used in place and must outlive
 * the calls below. */
int lcr_load_batch(lcr_ctx*, const lcr_reads*, const lcr_regions*);
/* Asynchronous input path for a caller whose reads are decoded on the host (the reference's loop, thread.rs:77-143, decodes the
 * next region while the current one is processed).  A ctx has two staging slots in HBM and an upload stream of its own:
 *   lcr_load_batch_async(ctx, reads, regions, slot)  enqueues the H2D copies of a LCR_MEM_HOST batch into slot 0 / 1 and returns;
 *       the host arrays must stay valid -- and should be page-locked (lcr_host_alloc / lcr_host_register), otherwise the
 *       runtime stages the copy and the call blocks for most of it -- until lcr_bind_batch(slot) has returned.
 *   lcr_bind_batch(ctx, slot)  makes that batch the current one (waits for its upload, then as lcr_load_batch of a
 *       device-resident batch); the stage calls follow.  Uploading into the slot that is bound invalidates the bound batch.
 * So batch i + 1 crosses PCIe while batch i's kernels run; results are those of lcr_load_batch on the same arrays. */
int lcr_load_batch_async(lcr_ctx*, const lcr_reads*, const lcr_regions*, int32_t slot);
int lcr_bind_batch(lcr_ctx*, int32_t slot);
/* page-locked host memory for the arrays of lcr_reads / lcr_regions (hipHostMalloc), or page-lock memory the caller owns */
int lcr_host_alloc(size_t bytes, void** out);
void lcr_host_free(void* p);
int lcr_host_register(void* p, size_t bytes);
int lcr_host_unregister(void* p);

/* The four stage calls below queue their kernels on the context's stream and return as soon as the host has what it
 * needs to go on (error verdicts, sizes); the last kernels of a stage may still be running.  Later stage calls queue
 * behind them; the lcr_get_* calls and lcr_ctx_sync wait.  An error a stage can raise is always raised by that call. */
/* replaces Profile::fill_data_into_freq_vec (util.rs:621-949); thread.rs:93-103.
 * Returns LCR_E_CIGAR for an unknown CIGAR op or a CIGAR inconsistent with l_seq / soft clips. */
int lcr_pileup(lcr_ctx*, const lcr_params*);
int lcr_get_columns(lcr_ctx*, lcr_columns* out);

/* replaces SNPFrag::get_candidate_snps (candidate.rs:54-528); thread.rs:118-133 */
int lcr_candidates(lcr_ctx*, const lcr_params*);
int lcr_get_candidates(lcr_ctx*, lcr_candidate_list* out);
/* The same records in device memory (HBM), current after lcr_candidates and after lcr_phase: for consumers that stay
 * on the device, e.g. the multi-GPU gather of result records (thread.rs:204-221 collects them per region). The pointer
 * is valid until the next lcr_candidates / lcr_load_batch on this ctx. */
int lcr_get_candidates_device(lcr_ctx*, const lcr_candidate** dev_cand, int32_t* n_cand);

/* replaces SNPFrag::get_fragments (fragment.rs:10-309); thread.rs:136-143 */
int lcr_fragments(lcr_ctx*, const lcr_params*);
int lcr_get_fragmat(lcr_ctx*, lcr_fragmat* out);

/* replaces init_haplotypes/init_assignment + SNPFrag::phase (phase.rs:1087-1296) and the
 * post-phase sequence thread.rs:162-201 (assign_reads_haplotype, assign_snp_haplotype_genotype,
 * eval_rna_edit_var_phase, eval_low_frac_var_phase, assign_phase_set). */
int lcr_phase(lcr_ctx*, const lcr_params*);
int lcr_get_phase_result(lcr_ctx*, lcr_phase_result* out);
/* The per-row results once more as 12-byte records in device memory (HBM), current after lcr_phase: the second record type
 * of the multi-GPU gather (the reference collects read -> HP / PS per region, thread.rs:204-221).  row = fragment row of
 * lcr_get_fragmat (batch-wide).  The pointer is valid until the next lcr_phase / lcr_load_batch on this ctx. */
typedef struct {
  int32_t row;
  int8_t haplotag;           /* sigma */
  uint8_t assignment;        /* 0 unassigned, 1 hap1, 2 hap2 */
  uint16_t pad_;
  uint32_t phase_set;        /* 0 = none */
} lcr_read_record;
int lcr_get_read_records_device(lcr_ctx*, const lcr_read_record** dev_rec, int32_t* n_rows);

/* Everything the last lcr_phase produced, in one call -- the getter of a PIPELINED caller (thread.rs:204-221 collects per region what
 * the closure of thread.rs:93-201 returns).  The getters above answer for the batch that is bound: once the next batch has been bound
 * (lcr_load_batch of a device batch, lcr_bind_batch) they return LCR_E_STATE.  This one stays valid across lcr_load_batch / lcr_bind_batch
 * and lcr_pileup of the next batch -- none of them touches the buffers named here -- until the next lcr_candidates; with the
 * asynchronous phase stage it is the call that waits for the stage in flight, so the order
 *   lcr_phase(N); lcr_load_batch(N + 1); lcr_pileup(N + 1); lcr_collect_phase(results of N); lcr_candidates(N + 1); ...
 * delivers every batch's results while batch N + 1's pileup runs beside batch N's resolve / post-phase tails.  Host arrays as in
 * lcr_candidate_list / lcr_phase_result (candidates updated by the stage); dev_cand / dev_read_rec: the same records in HBM. */
typedef struct {
  int32_t n_regions, n_rows, n_cand, pad_;
  const lcr_candidate* cand;            /* n_cand records, region by region                         */
  const int32_t* cand_region_off;       /* n_regions + 1                                            */
  const int32_t* row_region_off;        /* n_regions + 1: fragment rows of every region              */
  const int8_t* haplotag;               /* n_rows, as lcr_phase_result                               */
  const uint8_t* assignment;
  const uint32_t* phase_set;
  const double* objective;              /* n_regions                                                */
  const lcr_candidate* dev_cand;        /* HBM: n_cand records                                      */
  const lcr_read_record* dev_read_rec;  /* HBM: n_rows records                                      */
} lcr_phase_collected;
int lcr_collect_phase(lcr_ctx*, lcr_phase_collected* out);

/* LD blocks of one region of the last lcr_phase (SNPFrag.ld_blocks, snpfrags.rs:29; built by divide_snps_into_blocks,
 * candidate.rs:615-747): block b = snp_idx[block_off[b] .. block_off[b + 1]) (candidate indices inside the region), in
 * the reference's block and node order.  Only regions with more than max_enum_snps candidates build blocks (the
 * enumeration branch, phase.rs:1097-1122, never reads them): others report n_blocks = 0. */
int lcr_get_ld_blocks(lcr_ctx*, int32_t region, int32_t* n_blocks, const int32_t** block_off, const int32_t** snp_idx);

/* Decision arithmetic of the optimiser and its census (round 4).  The reference decides sigma flips, the delta / eta choice and
 * `prob > largest_prob` on f64 ratio scores / sums accumulated in list order (phase.rs:77-96, 128-176, 257-276, 845-858, 905-940,
 * 1117); every term is one of 62 constants, so liblcr takes each decision on the EXACT fixed-point sums (order-free, hence
 * parallel and reproducible) and, where those sums tie exactly -- the only decisions on which summation order can matter --,
 * on the reference-order f64 scores of that row / configuration.  lcr_get_tie_census reports the ties of the last lcr_phase:
 *   out[0] sigma decisions with A == B at rows with an entry at a het site, decided by the f64 scores of phase.rs:77-96
 *   out[1] ... of which flipped (q < qn)
 *   out[2] delta / eta choices with a tie at the maximum where the first maximum was kept (UNRESOLVED; round 6: reported by chain regions of
 *          workgroup scope only when lcr_debug_set("chain_ties", 0) keeps them from running again under the complete contract -- the
 *          all-CU chain kernels, regions of >= 2^17 phase entries, neither resolve nor count them)
 *   out[3] steps whose only changes were tie changes, taken as "no improvement" (check_new_*, phase.rs:278-355; UNRESOLVED: as out[2])
 *   out[4] enumeration branch: regions whose configurations of maximal objective differ and were compared by their f64 sums
 *          (phase.rs:257-276); chain regions of workgroup scope (round 6): compares of two configurations of equal objective whose
 *          match bits differ, decided by their f64 sums
 *   out[5] compares that fell to "first maximum wins" (UNRESOLVED: more maxima than the list holds, states beyond the memory budget;
 *          chain regions with "chain_ties" = 0)
 *   out[6] sigma ties met by kernels without the f64 path (UNRESOLVED)
 *   out[7] (round 5: enumeration branch; round 6: chain regions of workgroup scope too) delta / eta ties at the maximum decided by the f64
 *          scores of phase.rs:128-176 + tie-only steps decided by the reference's sums of scores (check_new_haplotag /
 *          check_new_haplotype_genotype)
 * All UNRESOLVED counts zero = every decision of the call was the reference arithmetic's decision. */
int lcr_get_tie_census(lcr_ctx*, uint64_t out[8]);

/* Region discovery: replaces find_isolated_regions_with_depth (util.rs:236-332, truncation
 * off) for one contig.  ref_start / ref_end are record.reference_start() / reference_end() of the reads that
 * pass the filters of util.rs:264-279 (mem = LCR_MEM_HOST or LCR_MEM_DEVICE).  Region i covers 0-based columns
 * [start0[i], start0[i] + len[i]) — the reference's 1-based [start, end) = [start0+1, start0+len+1).  As in the
 * reference, cursors and max_coverage are reset only when a region is emitted (`region_end > region_start`,
 * util.rs:297-310): a single-column island stays pending and starts the region that ends with the next island. */
typedef struct {
  int32_t n_regions;
  const int64_t* start0;
  const int32_t* len;
  const uint32_t* max_cov;
} lcr_region_list;
int lcr_discover_regions(lcr_ctx*, int32_t mem, int32_t n_reads, const int32_t* ref_start, const int32_t* ref_end,
                         int64_t contig_len, lcr_region_list* out);

/* ---- BGZF / BAM decode -> lcr_reads on the host -------------------------------------------
 * Replaces the rust-htslib IndexedReader calls of util.rs:636-691 and fragment.rs:19-59 (and the read pass of
 * region discovery, util.rs:256-287): the file is inflated once (all BGZF blocks in parallel on n_threads host
 * threads, <= 0: all hardware threads), every record is indexed once, and the batches for lcr_load_batch are cut
 * out of that index, so the pileup and the fragment stage share one decode (the reference inflates every
 * region's blocks twice).  The compressed file is mapped; only ONE contig's inflated bytes and record index are
 * resident at a time (loaded on demand, replaced when another contig is asked for).  The file must be coordinate-sorted (the reference needs its .bai too).  No GPU is
 * involved; a handle is used by one thread at a time, output pointers stay valid until the next call on the
 * handle or lcr_bam_close. */
typedef struct lcr_bam lcr_bam;
typedef struct {            /* util.rs:652-668 / fragment.rs:32-49                                        */
  uint8_t min_mapq;         /* mapq < min_mapq is dropped                                                 */
  int32_t min_read_length;  /* l_seq < min_read_length is dropped                                         */
  float divergence;         /* a `de` tag of type f with value >= divergence drops the read               */
} lcr_read_filter;          /* unmapped / secondary / supplementary records are always dropped            */
/* *out is set even when the call fails (unless out of memory): lcr_bam_last_error explains, lcr_bam_close frees */
int lcr_bam_open(const char* path, int32_t n_threads, lcr_bam** out);
/* The same with the residency policy spelled out: a file whose INFLATED size is at most keep_bytes is inflated once, at open, and
 * stays inflated (lcr_bam_spans / lcr_bam_batch / lcr_bam_write_phased then never inflate a block again); a larger one keeps
 * one contig at a time resident and inflates a contig's blocks when it is first used.  lcr_bam_open = keep_bytes of 4 GiB;
 * 0 = never keep (the memory bound of one contig). */
int lcr_bam_open_keep(const char* path, int32_t n_threads, int64_t keep_bytes, lcr_bam** out);
void lcr_bam_close(lcr_bam*);
const char* lcr_bam_last_error(const lcr_bam*);
int lcr_bam_refs(lcr_bam*, int32_t* n_ref, const char* const** names, const int64_t** lengths);
int lcr_bam_n_records(lcr_bam*, int64_t* n);
/* bytes of inflated stream + record index held right now (the whole stream of a file within keep_bytes, else one contig at a time) and their peak since lcr_bam_open */
int lcr_bam_resident(lcr_bam*, int64_t* now, int64_t* peak);
/* record.reference_start() / reference_end() of the reads of contig ref_id that pass the filter, in file order:
 * the input of lcr_discover_regions (util.rs:264-285) */
int lcr_bam_spans(lcr_bam*, int32_t ref_id, const lcr_read_filter*, int32_t* n, const int32_t** ref_start,
                  const int32_t** ref_end);
/* The passing reads of contig ref_id grouped by region with htslib's fetch rule on the reference's numbers
 * (util.rs:637: [start0 + 1, start0 + len + 1) as 0-based half-open; a read that overlaps two regions is listed
 * in both): fills *reads (mem = LCR_MEM_HOST) and *read_begin (n_regions + 1) for lcr_load_batch.  name_off /
 * names (optional): read names, NUL-terminated, name i at names + name_off[i] (the qname maps of thread.rs). */
int lcr_bam_batch(lcr_bam*, int32_t ref_id, const lcr_read_filter*, int32_t n_regions, const int64_t* start0,
                  const int32_t* len, lcr_reads* reads, const int32_t** read_begin, const uint64_t** name_off,
                  const char** names);

/* ---- phased BAM, replaces thread.rs:307-361 ------------------------------------------------
 * Writes to out_path the header of the opened file and, region by region in the order given (region i = columns
 * [start0[i], start0[i] + len[i]) of contig region_ref[i]), the records the reference's loop keeps: fetched by the
 * region (util.rs:637 rule), not unmapped / secondary / supplementary, and lying inside the region
 * (reference_start + 1 >= start && reference_end + 1 <= end, thread.rs:340-345).  A record whose name is among the
 * n_tagged names gets `HP:i:<hp>` (int32) appended when hp is 1 or 2 and `PS:I:<ps>` (uint32) when ps != 0, unless
 * it already carries that tag; of several entries with one name the first counts (hp < 0: no assignment entry,
 * thread.rs:308-325).  BGZF blocks of 0xff00 bytes are deflated on n_threads threads (<= 0: as lcr_bam_open) at
 * `level` (-1 = zlib default, as the reference's writer).  Parity is defined on the inflated stream. */
int lcr_bam_write_phased(lcr_bam*, const char* out_path, int32_t n_regions, const int32_t* region_ref, const int64_t* start0,
                         const int32_t* len, int64_t n_tagged, const uint64_t* name_off, const char* names, const int32_t* hp,
                         const uint32_t* ps, int32_t level, int32_t n_threads);

/* Decoded reads -> BAM, the inverse of lcr_bam_batch (synthetic data sets, round-trip tests; the reference has no such
 * writer: its inputs come from minimap2): one contig, the reads of `rd` (LCR_MEM_HOST, sorted by position) as records
 * "r<index>", mapq 60, flag 0 / 16, `ts:A:+` / `-` from lcr_reads.flags; BGZF at `level`, deflated on n_threads threads. */
int lcr_bam_write_reads(const char* out_path, const char* contig, int64_t contig_len, const lcr_reads* rd, int32_t level, int32_t n_threads);

/* Asynchronous phase stage (round 5; off by default).  on = 1: lcr_phase returns as soon as its kernels are queued -- on queues of the
 * stage's own, behind the fragment stage's kernels -- and its results are collected by whoever asks for them first: lcr_collect_phase
 * (valid after the next batch has been bound: the pipelined order is spelled out there), every getter of the bound batch
 * (lcr_get_candidates*, lcr_get_phase_result, lcr_get_read_records_device, lcr_get_tie_census, lcr_get_ld_blocks), lcr_ctx_sync, the
 * next lcr_candidates / lcr_phase.  The caller's loop (thread.rs:93-201 per region; here per batch) can then bind the next
 * device-resident batch and queue its pileup at once: lcr_pileup's kernels are held back until the dense part of the stage in flight
 * -- the enumeration restarts -- is done, and run beside its resolve / post-phase tails, which leave most CUs idle.
 * CONTRACT CHANGE while it is on: the input arrays of a device-resident batch must stay unchanged until that batch's results have
 * been collected (a getter or lcr_ctx_sync) -- with the synchronous stage they are free when lcr_phase returns.  A host batch
 * (lcr_load_batch with LCR_MEM_HOST, lcr_load_batch_async) waits for the stage in flight by itself.  Persistent all-CU launches,
 * the host epilogue and phase_prof make lcr_phase wait as before.  The stage then uses four queues: the process should run with
 * GPU_MAX_HW_QUEUES >= 8 in its environment (ROCm's default of 4 maps two of them onto one hardware queue: correct, but measured
 * without gain).  The HIP runtime reads that variable when it starts, so the library cannot set it: lcr_ctx_set_async_phase(ctx, 1)
 * returns LCR_W_HW_QUEUES (> 0: the mode IS on) when the variable is unset or below 8, and lcr_last_error says what to export. */
int lcr_ctx_set_async_phase(lcr_ctx*, int on);

/* Regions whose phase matrix is far beyond one CU are phased by persistent all-CU kernels; two such launches on one GPU --
 * of two contexts or two processes -- must not overlap, so they are serialised per device by a process-local mutex and an
 * flock on <dir>/grid_<pci bus id>.lock.  dir defaults to /tmp/liblcr-locks (created 1777 by whoever comes first); it has to be this user's
 * or root's, and sticky if others may write into it -- otherwise lcr_phase fails with LCR_E_DEVICE and says so (round 6: it used to fall back
 * to a per-user directory silently, which would have left two users' launches unserialised).  A directory named here is created 0700 and
 * has to pass the same test; "/tmp/liblcr-<uid>" is the per-user lock for a machine whose GPUs are not shared.  Processes that share a GPU must see the same directory
 * (containers without a common /tmp: name one on a shared mount).  The lock is waited for at most ten minutes; a lock that cannot
 * be taken makes lcr_phase fail with LCR_E_DEVICE -- it is never skipped. */
int lcr_ctx_set_lock_dir(lcr_ctx*, const char* dir);

/* Debug / test switches (the library reads no environment variable): key = "phase_prof", "post_host", "grid_min_entries",
 * "grid_generic", "grid_spec_lanes", "post_half", "enum_force_big", "enum_force_stream", "host_threads", "tie_arith", "timing_mask",
 * "k3_hits" (0: the fragment stage walks the CIGARs itself), "async_phase" (1: lcr_phase returns with its kernels in flight),
 * "enum_bits" (0: the enumeration restarts one per wave, the kernels of rounds 2-4; default 1: eight per wave as bit states),
 * "grid_spec_batch" (0: the all-CU chain kernel's speculative half-rounds side by side on sub-grids; default 1: as eight bits of one state)
 * (bit k: only the kernel groups LCR_K_* k are timed when timing is enabled; 0 = all) (see PhaseDebug in
 * csrc/lcr_phase_host.h), "hist_tiles" (quality histograms from K0's records: 0 = when the survivors are dense, 1 = whenever the
 * preset allows, -1 = never); round 6: "chain_ties" (0: chain regions of workgroup scope keep the tie contract of round 5: sigma ties
 * only), "redo_lds" (bytes of dynamic LDS of the enumeration branch's repair pass; 0: its matrices in global memory), "fuse_filter",
 * "bg_tiles", "zonefix_overlap", "zonefix_fused" (measurement switches of the pileup stage), "spec_compact" (0: lcr_candidates waits for the
 * survivors' number before it queues their compaction), "phase_prio", "no_gate" (measurement switches of the asynchronous stage),
 * "own_fill" (0: hipMemsetAsync instead of the
 * library's fill kernels), "fill_selftest" (checks those kernels against the host; LCR_E_DEVICE on a difference), "host_trace" (1: a
 * "[host]" line of wall-clock marks per lcr_phase on stderr; process-wide like own_fill).  Unknown key: LCR_E_ARG.  The defaults are the
 * product behaviour. */
int lcr_debug_set(lcr_ctx*, const char* key, int64_t value);

/* Timing: HIP-event time (ms) of the last launch of each kernel on the ctx's stream.  (LCR_K_PHASE brackets what lcr_phase puts on the
 * ctx's stream: with the asynchronous phase stage that is the hand-over to the stage's queues only, not the stage.) */
enum { LCR_K_SPANS = 0 /* K0: CIGAR decode + binning */, LCR_K_PILEUP, LCR_K_CAND_FILTER, LCR_K_CAND_HIST, LCR_K_CAND_GT,
       LCR_K_FRAG_COUNT, LCR_K_FRAG_FILL, LCR_K_PHASE,
       LCR_K_BIND /* lcr_load_batch: read headers, read / tile -> region tables, op blocks' first reads (part of the pileup stage) */,
       LCR_K_BIND_TABLE /* lcr_load_batch of a device batch: region table + CIGAR layout check, before its one host wait */, LCR_NKERNELS };
int lcr_enable_timing(lcr_ctx*, int on);
int lcr_kernel_ms(lcr_ctx*, int kernel, float* ms);
/* Bytes the pileup tally kernel (K1) of the last lcr_pileup has to move: read bases once (B) + 8-byte
 * records + (1 + 4 + 4*LCR_NPLANES)*L; and the implementation-independent figure of the whole pileup
 * stage (K0 + K1): B + 4C + 37R + (4*LCR_NPLANES + 1)*L.  See DESIGN.md. */
int lcr_pileup_bytes(lcr_ctx*, int64_t* bytes);
int lcr_pileup_stage_bytes(lcr_ctx*, int64_t* bytes);

/* Memory given back by contexts (lcr_ctx_destroy, buffers that grow) is kept in a process-wide cache per device -- at most 24 GiB of HBM
 * and 2 GiB of page-locked host memory each -- and handed to the next context: a worker that recreates its context per task does not
 * churn hipMalloc / hipFree.  lcr_release_cached_memory returns all of it to the runtime (a process that shares the GPU with another
 * allocator calls it between phases); lcr_set_cache_limits changes the two bounds (bytes per device; 0 = keep nothing). */
int lcr_release_cached_memory(void);
int lcr_set_cache_limits(int64_t device_bytes, int64_t host_bytes);

const char* lcr_version(void);

#ifdef __cplusplus
}
#endif
#endif
