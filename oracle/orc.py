"""TEST INFRASTRUCTURE ONLY — ctypes loader for the CPU oracle (oracle/lcr_oracle.cpp).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED BY THE REFERENCE (see orc_common.h): no reference tests/goldens exist and the Rust
toolchain is absent, so the oracle is pinned by hand-derived KATs + an independent NumPy
restatement (oracle/oracle_np.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from longcallr_amd import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblcr_oracle.so")
MODE_F64, MODE_EXACT, MODE_F64_ONLY, MODE_EXACT_ONLY, MODE_TIE = 0, 1, 2, 3, 4
# the tie classes liblcr resolves with the reference's f64 arithmetic (include/lcr.h, lcr_get_tie_census): ALL FOUR in the
# enumeration branch -- sigma ties (1), delta / eta ties at the maximum (2), the verdict of tie-only steps (4), `prob > largest_prob`
# at equal objective (8) --, sigma ties in the chain branch
# (round 6: all four classes in chain regions of workgroup scope too -- k4_chain_wg runs a region that met a tie of classes 2 / 4 / 8 again
# under the complete contract; a chain region that gets all CUs (k4_chain_grid: >= 2^17 phase entries, or the tests' grid_min_entries = 0)
# resolves sigma ties only)
TIE_MASK_LIBLCR = 15 | (15 << 8) | (1 << 16)
TIE_MASK_LIBLCR_GRID = 15 | (1 << 8) | (1 << 16)


def tie_mask(enum_mask, chain_mask):
    return enum_mask | (chain_mask << 8) | (1 << 16)


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("lcr_oracle.cpp", "orc_common.h")] + [
        os.path.join(_HERE, "..", "include", "lcr.h")]
    if force or not os.path.exists(_SO) or any(
            os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_SO) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(build())
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        l.orc_region_create.restype = vp
        l.orc_region_create.argtypes = [C.POINTER(_abi.LcrReads), i32, i32, i64, i32, vp,
                                        C.POINTER(_abi.LcrParams)]
        for f in ("orc_region_destroy", "orc_pileup", "orc_candidates", "orc_fragments",
                  "orc_post_phase"):
            getattr(l, f).argtypes = [vp]
            getattr(l, f).restype = None
        l.orc_phase.argtypes = [vp, C.c_int]
        l.orc_phase.restype = None
        l.orc_get_planes.argtypes = [vp, vp]
        l.orc_get_baseq.argtypes = [vp, i32, C.c_int, vp, i32]
        l.orc_get_baseq.restype = i32
        l.orc_n_cand.argtypes = [vp]
        l.orc_n_cand.restype = i32
        l.orc_get_cands.argtypes = [vp, vp]
        l.orc_cand_gt_hist.argtypes = [vp, i32, vp]
        l.orc_n_rows.argtypes = [vp]
        l.orc_n_rows.restype = i32
        l.orc_nnz.argtypes = [vp]
        l.orc_nnz.restype = i64
        l.orc_get_fragmat.argtypes = [vp] * 7
        l.orc_get_ld_blocks.argtypes = [vp, vp, vp, i32]
        l.orc_get_ld_blocks.restype = i32
        l.orc_get_phase.argtypes = [vp] * 5
        l.orc_get_stats.argtypes = [vp, vp]
        l.orc_vcf_text.argtypes = [vp, C.c_char_p, C.c_char_p, i64]
        l.orc_vcf_text.restype = i64
        l.orc_run_batch.restype = vp
        l.orc_run_batch.argtypes = [C.POINTER(_abi.LcrReads), C.POINTER(_abi.LcrRegions), C.POINTER(_abi.LcrParams),
                                    C.c_int, C.c_int, C.c_int, C.c_int]
        l.orc_run_batch_opts.restype = vp
        l.orc_run_batch_opts.argtypes = [C.POINTER(_abi.LcrReads), C.POINTER(_abi.LcrRegions), C.POINTER(_abi.LcrParams),
                                         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        l.orc_batch_tie_census.argtypes = [vp, vp]
        l.orc_set_fast.argtypes = [vp, C.c_int]
        l.orc_set_fast.restype = None
        l.orc_set_tie_mask.argtypes = [vp, C.c_int]
        l.orc_set_tie_mask.restype = None
        l.orc_get_tie_census.argtypes = [vp, vp]
        l.orc_batch_destroy.argtypes = [vp]
        l.orc_batch_destroy.restype = None
        l.orc_batch_seconds.argtypes = [vp]
        l.orc_batch_seconds.restype = dbl
        l.orc_batch_threads.argtypes = [vp]
        l.orc_batch_threads.restype = i32
        l.orc_batch_offsets.argtypes = [vp] * 4
        l.orc_batch_planes.argtypes = [vp, vp]
        l.orc_batch_cands.argtypes = [vp, vp]
        l.orc_batch_fragmat.argtypes = [vp] * 7
        l.orc_batch_phase.argtypes = [vp] * 5
        l.orc_batch_stats.argtypes = [vp, vp]
        l.orc_batch_vcf.argtypes = [vp, C.c_char_p, vp, i64, vp]
        l.orc_batch_vcf.restype = i64
        l.orc_batch_region.argtypes = [vp, i32]
        l.orc_batch_region.restype = vp
        l.orc_strand_odds_ratio.argtypes = [C.c_int] * 4
        l.orc_strand_odds_ratio.restype = C.c_float
        l.orc_binomial_two_tailed.argtypes = [C.c_uint64, C.c_uint64]
        l.orc_binomial_two_tailed.restype = dbl
        l.orc_two_major_alleles.argtypes = [vp, C.c_uint8, vp, vp, vp, vp]
        l.orc_aki.argtypes = [C.c_int] * 4 + [dbl]
        l.orc_aki.restype = dbl
        l.orc_cal_sigma_delta_eta_log.argtypes = [C.c_int, C.c_int, vp, vp, vp, vp]
        l.orc_cal_sigma_delta_eta_log.restype = dbl
        l.orc_cal_delta_eta_sigma_log.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp]
        l.orc_cal_delta_eta_sigma_log.restype = dbl
        l.orc_cal_phase_score_log.argtypes = [C.c_int, C.c_int, C.c_int, vp, vp, vp]
        l.orc_cal_phase_score_log.restype = dbl
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Region:
    """One region of a ReadBatch run through the oracle, stage by stage."""

    def __init__(self, batch, region_idx, params):
        self.batch, self.ri, self.params = batch, region_idx, params
        self._reads = batch.c_reads()
        o = int(batch.col_off[region_idx])
        self.len = int(batch.len[region_idx])
        self._ref = np.ascontiguousarray(batch.ref[o:o + self.len])
        self.h = lib().orc_region_create(
            C.byref(self._reads), int(batch.read_begin[region_idx]),
            int(batch.read_begin[region_idx + 1]), int(batch.start0[region_idx]), self.len,
            _p(self._ref), C.byref(params))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_region_destroy(self.h)
            self.h = None

    def set_fast(self, threads=1):
        """indexed gathers (same entries, same order) on `threads` threads; 0 = the reference's linear searches"""
        lib().orc_set_fast(self.h, threads)
        return self

    def set_tie_mask(self, mask):
        lib().orc_set_tie_mask(self.h, mask)
        return self

    def tie_census(self):
        out = np.zeros(10, np.int64)
        lib().orc_get_tie_census(self.h, _p(out))
        return out

    def pileup(self):
        lib().orc_pileup(self.h)
        return self

    def candidates(self):
        lib().orc_candidates(self.h)
        return self

    def fragments(self):
        lib().orc_fragments(self.h)
        return self

    def phase(self, mode=MODE_EXACT):
        lib().orc_phase(self.h, mode)
        return self

    def post_phase(self):
        lib().orc_post_phase(self.h)
        return self

    def run_all(self, mode=MODE_EXACT):
        return self.pileup().candidates().fragments().phase(mode).post_phase()

    def planes(self):
        out = np.zeros((_abi.NPLANES, self.len), dtype=np.uint32)
        lib().orc_get_planes(self.h, _p(out))
        return out

    def baseq(self, col, allele):
        buf = np.zeros(1 << 17, dtype=np.uint8)
        n = lib().orc_get_baseq(self.h, col, allele, _p(buf), buf.size)
        return buf[:n].copy()

    def cands(self):
        n = lib().orc_n_cand(self.h)
        out = np.zeros(n, dtype=_abi.CAND_DTYPE)
        if n:
            lib().orc_get_cands(self.h, _p(out))
        out["region"] = self.ri
        return out

    def cand_gt_hist(self, col):
        out = np.zeros(8, dtype=np.float64)
        lib().orc_cand_gt_hist(self.h, col, _p(out))
        return out

    def fragmat(self):
        n, nnz = lib().orc_n_rows(self.h), lib().orc_nnz(self.h)
        d = dict(row_ptr=np.zeros(n + 1, np.int64), row_read=np.zeros(n, np.int32),
                 col=np.zeros(nnz, np.int32), val=np.zeros(nnz, np.uint8),
                 row_for_phasing=np.zeros(n, np.uint8), row_links=np.zeros(n, np.uint32))
        lib().orc_get_fragmat(self.h, *[_p(d[k]) for k in
                                        ("row_ptr", "row_read", "col", "val", "row_for_phasing", "row_links")])
        return d

    def ld_blocks(self):
        n = lib().orc_n_cand(self.h)
        off, mem = np.zeros(n + 2, np.int32), np.zeros(n + 1, np.int32)
        nb = lib().orc_get_ld_blocks(self.h, _p(off), _p(mem), mem.size)
        return [mem[off[b]:off[b + 1]].tolist() for b in range(nb)]

    def phase_result(self):
        n = lib().orc_n_rows(self.h)
        tag, asg, ps = np.zeros(n, np.int8), np.zeros(n, np.uint8), np.zeros(n, np.uint32)
        obj = np.zeros(1, np.float64)
        lib().orc_get_phase(self.h, _p(tag), _p(asg), _p(ps), _p(obj))
        return dict(haplotag=tag, assignment=asg, phase_set=ps, objective=float(obj[0]))

    def stats(self):
        out = np.zeros(4, np.int64)
        lib().orc_get_stats(self.h, _p(out))
        return dict(zip(("cross_optimize_calls", "iterations", "noise_ties", "assert_violations"),
                        out.tolist()))

    def vcf_text(self, chrom="chr20"):
        cap = 1 << 20
        buf = C.create_string_buffer(cap)
        n = lib().orc_vcf_text(self.h, chrom.encode(), buf, cap)
        return buf.raw[:n].decode()


UPTO = {"pileup": 0, "cands": 1, "frag": 2, "post": 3}


class Batch:
    """A whole ReadBatch through the oracle on a native thread pool (orc_run_batch): the analogue of the reference's
    rayon par_iter over regions (thread.rs:77).  Results come back concatenated in batch order, in the formats the
    lcr_get_* calls of the HIP path use, so a full-size comparison is a handful of array compares."""

    def __init__(self, batch, params, mode=MODE_EXACT_ONLY, threads=0, upto="post", keep_planes=True, fast=0, tie_mask=15):
        self.batch, self.params, self.ng = batch, params, batch.n_regions
        self._reads, self._regions = batch.c_reads(), batch.c_regions()
        self.upto = UPTO[upto]
        self.h = lib().orc_run_batch_opts(C.byref(self._reads), C.byref(self._regions), C.byref(params), mode, threads,
                                          self.upto, 1 if keep_planes else 0, fast, tie_mask)
        self.seconds, self.threads = lib().orc_batch_seconds(self.h), lib().orc_batch_threads(self.h)
        self.cand_off, self.row_off = np.zeros(self.ng + 1, np.int32), np.zeros(self.ng + 1, np.int32)
        self.nnz_off = np.zeros(self.ng + 1, np.int64)
        lib().orc_batch_offsets(self.h, _p(self.cand_off), _p(self.row_off), _p(self.nnz_off))

    def close(self):
        if getattr(self, "h", None):
            lib().orc_batch_destroy(self.h)
            self.h = None

    __del__ = close

    def planes(self):
        out = np.zeros((_abi.NPLANES, int(self.batch.col_off[-1])), dtype=np.uint32)
        lib().orc_batch_planes(self.h, _p(out))
        return out

    def cands(self):
        out = np.zeros(int(self.cand_off[-1]), dtype=_abi.CAND_DTYPE)
        if out.size:
            lib().orc_batch_cands(self.h, _p(out))
        return out

    def fragmat(self):
        n, nnz = int(self.row_off[-1]), int(self.nnz_off[-1])
        d = dict(row_ptr=np.zeros(n + 1, np.int64), row_read=np.zeros(n, np.int32), col=np.zeros(nnz, np.int32),
                 val=np.zeros(nnz, np.uint8), row_for_phasing=np.zeros(n, np.uint8), row_links=np.zeros(n, np.uint32))
        lib().orc_batch_fragmat(self.h, *[_p(d[k]) for k in ("row_ptr", "row_read", "col", "val", "row_for_phasing", "row_links")])
        d["row_region_off"] = self.row_off
        return d

    def phase_result(self):
        n = int(self.row_off[-1])
        d = dict(haplotag=np.zeros(n, np.int8), assignment=np.zeros(n, np.uint8), phase_set=np.zeros(n, np.uint32),
                 objective=np.zeros(self.ng, np.float64))
        lib().orc_batch_phase(self.h, *[_p(d[k]) for k in ("haplotag", "assignment", "phase_set", "objective")])
        return d

    def stats(self):
        """per region: cross_optimize calls, iterations, noise ties (modes 0 / 1 only), assert violations"""
        out = np.zeros((self.ng, 4), np.int64)
        lib().orc_batch_stats(self.h, _p(out))
        return out

    def tie_census(self):
        """per region, ORC_MODE_TIE / F64 / EXACT: ties among sigma decisions, delta/eta decisions, tie-only steps, best-pick
        compares, then the same four where the f64 scores decide differently from `a tie changes nothing`"""
        out = np.zeros((self.ng, 10), np.int64)
        lib().orc_batch_tie_census(self.h, _p(out))
        return out

    def vcf_texts(self, chrom="chrS"):
        off = np.zeros(self.ng + 1, np.int64)
        cap = 1 << 16
        while True:
            buf = C.create_string_buffer(cap)
            n = lib().orc_batch_vcf(self.h, chrom.encode(), buf, cap, _p(off))
            if n < cap:
                break
            cap = int(n) + 16
        raw = buf.raw
        return [raw[off[g]:off[g + 1]].decode() for g in range(self.ng)]

    def ld_blocks(self, g):
        r = lib().orc_batch_region(self.h, g)
        n = int(self.cand_off[g + 1] - self.cand_off[g])
        off, mem = np.zeros(n + 2, np.int32), np.zeros(n + 1, np.int32)
        nb = lib().orc_get_ld_blocks(r, _p(off), _p(mem), mem.size)
        return [mem[off[b]:off[b + 1]].tolist() for b in range(nb)]
