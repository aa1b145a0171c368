"""ctypes mirror of include/lcr.h (the C ABI of liblcr).

Only plain-data structures live here; no compute.  Both the product loader (``_lib.py``) and the
test-only oracle loader (``oracle/orc.py``) build their argument structs from these definitions.
"""
import ctypes as C

import numpy as np

LCR_PLATFORM_HIFI, LCR_PLATFORM_ONT = 0, 1
LCR_MEM_HOST, LCR_MEM_DEVICE = 0, 1

(PL_A, PL_C, PL_G, PL_T, PL_N, PL_D, PL_NI, PL_FWD_A, PL_FWD_C, PL_FWD_G, PL_FWD_T, PL_TS_FWD,
 PL_TS_REV, NPLANES) = range(14)
PLANE_NAMES = ["a", "c", "g", "t", "n", "d", "ni", "fwd_a", "fwd_c", "fwd_g", "fwd_t", "ts_fwd", "ts_rev"]

F_RNA_EDIT, F_DENSE, F_HET, F_FOR_PHASING, F_HOM, F_SINGLE, F_NON_SELECTED, F_CAND_SOMATIC = (
    1, 2, 4, 8, 16, 32, 64, 128)

(K_SPANS, K_PILEUP, K_CAND_FILTER, K_CAND_HIST, K_CAND_GT, K_FRAG_COUNT, K_FRAG_FILL, K_PHASE, K_BIND, K_BIND_TABLE,
 NKERNELS) = range(11)


class LcrReads(C.Structure):
    _fields_ = [
        ("mem", C.c_int32), ("n_reads", C.c_int32), ("n_bases", C.c_int64), ("n_cigar", C.c_int64),
        ("pos", C.c_void_p), ("seq_len", C.c_void_p), ("lead_clip", C.c_void_p),
        ("trail_clip", C.c_void_p), ("flags", C.c_void_p), ("seq_off", C.c_void_p),
        ("cig_off", C.c_void_p), ("n_cig", C.c_void_p), ("bases", C.c_void_p),
        ("quals", C.c_void_p), ("cigar", C.c_void_p),
    ]


class LcrReadFilter(C.Structure):   # include/lcr.h lcr_read_filter (util.rs:652-668)
    _fields_ = [("min_mapq", C.c_uint8), ("min_read_length", C.c_int32), ("divergence", C.c_float)]


class LcrRegions(C.Structure):
    _fields_ = [
        ("mem", C.c_int32), ("n_regions", C.c_int32), ("start0", C.c_void_p), ("len", C.c_void_p),
        ("col_off", C.c_void_p), ("read_begin", C.c_void_p), ("ref", C.c_void_p),
    ]


class LcrParams(C.Structure):
    _fields_ = [
        ("platform", C.c_int32), ("min_baseq", C.c_uint32), ("dist_to_end", C.c_uint32),
        ("polya_len", C.c_uint32), ("min_depth", C.c_uint32), ("max_depth", C.c_uint32),
        ("min_qual", C.c_uint32), ("dense_win", C.c_uint32), ("min_dense_cnt", C.c_uint32),
        ("low_cnt_cut", C.c_uint32), ("min_linkers", C.c_uint32), ("max_enum_snps", C.c_uint32),
        ("ld_weight_threshold", C.c_uint32), ("use_strand_bias", C.c_int32),
        ("min_af", C.c_float), ("min_af_intron", C.c_float), ("low_frac_cut", C.c_float),
        ("min_phase_score", C.c_float), ("read_assign_cutoff", C.c_double), ("seed", C.c_uint64),
    ]


class LcrColumns(C.Structure):
    _fields_ = [("n_cols", C.c_int64), ("planes", C.c_void_p)]


CAND_DTYPE = np.dtype([
    ("pos", "<i8"), ("region", "<i4"), ("ref_base", "u1"), ("allele1", "u1"), ("allele2", "u1"),
    ("n_alt", "u1"), ("cnt1", "<u4"), ("cnt2", "<u4"), ("depth", "<u4"), ("af1", "<f4"),
    ("af2", "<f4"), ("variant_type", "<i4"), ("genotype", "<i4"), ("haplotype", "<i4"),
    ("flags", "<u4"), ("phase_set", "<u4"), ("loglik", "<f8", 3),
    ("gt_prob", "<f8", 3), ("qual", "<f8"), ("gq", "<f8"), ("phase_score", "<f8"),
], align=True)
assert CAND_DTYPE.itemsize == 128, CAND_DTYPE.itemsize


class LcrCandidateList(C.Structure):
    _fields_ = [("n_cand", C.c_int32), ("n_regions", C.c_int32), ("cand", C.c_void_p),
                ("region_off", C.c_void_p)]


class LcrFragmat(C.Structure):
    _fields_ = [
        ("n_rows", C.c_int32), ("nnz", C.c_int64), ("n_regions", C.c_int32),
        ("row_region_off", C.c_void_p), ("row_ptr", C.c_void_p), ("row_read", C.c_void_p),
        ("col", C.c_void_p), ("val", C.c_void_p), ("row_for_phasing", C.c_void_p),
        ("row_links", C.c_void_p),
    ]


class LcrRegionList(C.Structure):
    _fields_ = [("n_regions", C.c_int32), ("start0", C.c_void_p), ("len", C.c_void_p), ("max_cov", C.c_void_p)]


class LcrPhaseResult(C.Structure):
    _fields_ = [("n_rows", C.c_int32), ("n_regions", C.c_int32), ("haplotag", C.c_void_p),
                ("assignment", C.c_void_p), ("phase_set", C.c_void_p), ("objective", C.c_void_p)]


class LcrPhaseCollected(C.Structure):   # include/lcr.h: lcr_phase_collected
    _fields_ = [("n_regions", C.c_int32), ("n_rows", C.c_int32), ("n_cand", C.c_int32), ("pad_", C.c_int32),
                ("cand", C.c_void_p), ("cand_region_off", C.c_void_p), ("row_region_off", C.c_void_p),
                ("haplotag", C.c_void_p), ("assignment", C.c_void_p), ("phase_set", C.c_void_p), ("objective", C.c_void_p),
                ("dev_cand", C.c_void_p), ("dev_read_rec", C.c_void_p)]


# presets: the code values of main.rs:272-396 (not the help text)
PRESETS = {
    "hifi-isoseq": dict(platform=LCR_PLATFORM_HIFI, min_depth=6, min_phase_score=11.0, min_af=0.15,
                        dist_to_end=40, use_strand_bias=1),
    "hifi-masseq": dict(platform=LCR_PLATFORM_HIFI, min_depth=6, min_phase_score=11.0, min_af=0.15,
                        dist_to_end=40, use_strand_bias=0),
    "ont-cdna": dict(platform=LCR_PLATFORM_ONT, min_depth=10, min_phase_score=13.0, min_af=0.20,
                     dist_to_end=20, use_strand_bias=1),
    "ont-drna": dict(platform=LCR_PLATFORM_ONT, min_depth=10, min_phase_score=13.0, min_af=0.20,
                     dist_to_end=20, use_strand_bias=0),
}
PRESET_IDS = {"hifi-isoseq": 0, "hifi-masseq": 1, "ont-cdna": 2, "ont-drna": 3}
# host-side read filters of the presets (main.rs: min_mapq, min_read_length, divergence)
READ_FILTER = dict(min_mapq=20, min_read_length=500, divergence=0.5)


def make_params(preset="hifi-masseq", seed=2025, **over):
    p = LcrParams()
    base = dict(min_baseq=10, polya_len=5, max_depth=50000, min_qual=2, dense_win=100,
                min_dense_cnt=5, low_cnt_cut=10, min_linkers=1, max_enum_snps=10,
                ld_weight_threshold=1, min_af_intron=0.0, low_frac_cut=0.05,
                read_assign_cutoff=0.0, seed=seed)
    base.update(PRESETS[preset])
    base.update(over)
    for k, v in base.items():
        setattr(p, k, v)
    return p


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


# lcr_read_record (include/lcr.h): per-row results in HBM for the multi-GPU gather
READ_REC_DTYPE = np.dtype([("row", "<i4"), ("haplotag", "i1"), ("assignment", "u1"), ("pad_", "<u2"), ("phase_set", "<u4")])
assert READ_REC_DTYPE.itemsize == 12


class ReadBatch:
    """Host SoA of decoded reads + the regions that own them (numpy, C-contiguous)."""

    FIELDS = ["pos", "seq_len", "lead_clip", "trail_clip", "flags", "seq_off", "cig_off", "n_cig",
              "bases", "quals", "cigar"]
    DTYPES = dict(pos=np.int32, seq_len=np.int32, lead_clip=np.int32, trail_clip=np.int32,
                  flags=np.uint8, seq_off=np.uint64, cig_off=np.uint64, n_cig=np.uint32,
                  bases=np.uint8, quals=np.uint8, cigar=np.uint32)

    def __init__(self, **kw):
        for f in self.FIELDS:
            setattr(self, f, np.ascontiguousarray(kw[f], dtype=self.DTYPES[f]))
        self.start0 = np.ascontiguousarray(kw["start0"], dtype=np.int64)
        self.len = np.ascontiguousarray(kw["len"], dtype=np.int32)
        self.read_begin = np.ascontiguousarray(kw["read_begin"], dtype=np.int32)
        self.ref = np.ascontiguousarray(kw["ref"], dtype=np.uint8)
        self.col_off = np.zeros(len(self.len) + 1, dtype=np.int64)
        np.cumsum(self.len, out=self.col_off[1:])
        self.names = kw.get("names")
        assert self.ref.size == self.col_off[-1]
        assert self.read_begin[-1] == self.pos.size

    @property
    def n_reads(self):
        return int(self.pos.size)

    @property
    def n_regions(self):
        return int(self.len.size)

    def c_reads(self):
        r = LcrReads()
        r.mem = LCR_MEM_HOST
        r.n_reads = self.n_reads
        r.n_bases = int(self.bases.size)
        r.n_cigar = int(self.cigar.size)
        for f in self.FIELDS:
            setattr(r, f, _ptr(getattr(self, f)))
        return r

    def c_regions(self):
        g = LcrRegions()
        g.mem = LCR_MEM_HOST
        g.n_regions = self.n_regions
        g.start0, g.len, g.col_off = _ptr(self.start0), _ptr(self.len), _ptr(self.col_off)
        g.read_begin, g.ref = _ptr(self.read_begin), _ptr(self.ref)
        return g

    def pileup_algorithmic_bytes(self):
        """2B + 4C + 32R + (4*NPLANES + 1)*L  (DESIGN.md, K1)."""
        return (2 * int(self.bases.size) + 4 * int(self.cigar.size) + 32 * self.n_reads
                + (4 * NPLANES + 1) * int(self.col_off[-1]))
