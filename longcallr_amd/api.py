"""Host-side mirror of the reference's per-region call sequence (src/thread.rs:93-201) on top of
the C ABI.  Names follow the reference methods they replace:

    Engine.fill_data_into_freq_vec  -> Profile::fill_data_into_freq_vec   (util.rs:621)
    Engine.get_candidate_snps       -> SNPFrag::get_candidate_snps        (candidate.rs:54)
    Engine.get_fragments            -> SNPFrag::get_fragments             (fragment.rs:10)
    Engine.phase                    -> SNPFrag::phase + post-phase steps  (phase.rs:1087, snpfrags.rs)

All compute happens in liblcr.so (HIP); this module only marshals numpy / device pointers.
"""
import ctypes as C

import numpy as np

from . import _abi, _lib
from ._lib import LcrError


def _view(ptr, dtype, n):
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (np.dtype(dtype).itemsize * n)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n).copy()


def _view_nocopy(ptr, dtype, n):
    """the context's own buffer as an array (no copy): valid until the call that rewrites it"""
    if n == 0 or not ptr:
        return np.zeros(0, dtype=dtype)
    buf = (C.c_char * (np.dtype(dtype).itemsize * n)).from_address(ptr)
    return np.frombuffer(buf, dtype=dtype, count=n)


_DEBUG_ENV = {"LCR_PHASE_PROF": "phase_prof", "LCR_POST_HOST": "post_host", "LCR_GRID_MIN_ENTRIES": "grid_min_entries",
              "LCR_GRID_GENERIC": "grid_generic", "LCR_GRID_SPEC_LANES": "grid_spec_lanes", "LCR_GRID_SPEC_BATCH": "grid_spec_batch", "LCR_ENUM_BITS": "enum_bits", "LCR_POST_HALF": "post_half", "LCR_ENUM_FORCE_BIG": "enum_force_big",
              "LCR_ENUM_FORCE_STREAM": "enum_force_stream", "LCR_HOST_THREADS": "host_threads", "LCR_HIST_TILES": "hist_tiles", "LCR_TIE_ARITH": "tie_arith", "LCR_CHAIN_TIES": "chain_ties", "LCR_PLANE_PREFILL": "plane_prefill", "LCR_BG_TILES": "bg_tiles", "LCR_K3_HITS": "k3_hits", "LCR_FUSE_FILTER": "fuse_filter", "LCR_ZF_OVERLAP": "zonefix_overlap", "LCR_ZF_FUSED": "zonefix_fused", "LCR_ASYNC_PHASE": "async_phase", "LCR_HOST_TRACE": "host_trace", "LCR_OWN_FILL": "own_fill", "LCR_REDO_LDS": "redo_lds", "LCR_PHASE_PRIO": "phase_prio", "LCR_SPEC_COMPACT": "spec_compact", "LCR_NO_GATE": "no_gate"}


class Engine:
    def __init__(self, device=0, params=None, timing=False):
        self.lib = _lib.load()
        h = C.c_void_p()
        rc = self.lib.lcr_ctx_create(device, C.byref(h))
        if rc != 0:
            raise LcrError("lcr_ctx_create(device=%d) failed with %d: no usable HIP device; "
                           "liblcr has no CPU fallback" % (device, rc))
        self.h = h
        self.params = params if params is not None else _abi.make_params()
        self._keep = None
        if timing:
            self.lib.lcr_enable_timing(self.h, 1)
            if timing is not True:   # an iterable of _abi.K_* : only these kernel groups get their two event records per call
                self.debug_set("timing_mask", sum(1 << int(k) for k in timing))
        # developer / test hooks: the library reads no environment variable; this mirror hands LCR_* switches on (tests, tools)
        import os
        for env, key in _DEBUG_ENV.items():
            v = os.environ.get(env)
            if v is not None:
                self.debug_set(key, int(v) if v.lstrip("-").isdigit() else 1)
        if os.environ.get("LCR_LOCK_DIR"):
            self._chk(self.lib.lcr_ctx_set_lock_dir(self.h, os.fsencode(os.environ["LCR_LOCK_DIR"])), "lcr_ctx_set_lock_dir")

    def debug_set(self, key, value):
        self._chk(self.lib.lcr_debug_set(self.h, key.encode(), int(value)), "lcr_debug_set")
        return self

    def close(self):
        if getattr(self, "h", None):
            self.lib.lcr_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc != 0:
            raise LcrError("%s failed (%d): %s" % (what, rc, self.lib.lcr_last_error(self.h).decode()))

    def set_stream(self, hip_stream_ptr):
        self._chk(self.lib.lcr_ctx_set_stream(self.h, C.c_void_p(hip_stream_ptr)), "lcr_ctx_set_stream")

    def sync(self):
        self._chk(self.lib.lcr_ctx_sync(self.h), "lcr_ctx_sync")

    def set_async_phase(self, on=True):
        """lcr_ctx_set_async_phase: phase() returns with its kernels in flight; getters / sync() collect the results (include/lcr.h)"""
        rc = self.lib.lcr_ctx_set_async_phase(self.h, 1 if on else 0)
        self.async_warning = self.lib.lcr_last_error(self.h).decode() if rc > 0 else None   # (LCR_W_HW_QUEUES: on, but GPU_MAX_HW_QUEUES < 8)
        if rc < 0:
            self._chk(rc, "lcr_ctx_set_async_phase")
        return self

    # ---- batch binding -------------------------------------------------------------------------
    def load_batch(self, batch):
        """batch: _abi.ReadBatch (host numpy) or a (LcrReads, LcrRegions, keepalive) device triple."""
        if isinstance(batch, _abi.ReadBatch):
            reads, regions = batch.c_reads(), batch.c_regions()
            self._keep = (batch, reads, regions)
        else:
            reads, regions, keep = batch
            self._keep = (keep, reads, regions)
        self._chk(self.lib.lcr_load_batch(self.h, C.byref(reads), C.byref(regions)), "lcr_load_batch")
        return self

    def load_batch_async(self, batch, slot):
        """Enqueue the upload of a host batch (_abi.ReadBatch) into staging slot 0 / 1 and return; bind_batch(slot) makes it the
        current batch.  The batch's arrays should be page-locked (host_register) for the copy to run beside the kernels."""
        reads, regions = batch.c_reads(), batch.c_regions()
        if not hasattr(self, "_keep_slot"):
            self._keep_slot = {}
        self._chk(self.lib.lcr_load_batch_async(self.h, C.byref(reads), C.byref(regions), int(slot)), "lcr_load_batch_async")
        self._keep_slot[int(slot)] = (batch, reads, regions)
        return self

    def bind_batch(self, slot):
        self._chk(self.lib.lcr_bind_batch(self.h, int(slot)), "lcr_bind_batch")
        return self

    def discover_regions(self, ref_start, ref_end, contig_len):
        """find_isolated_regions_with_depth (util.rs:236-332) for one contig -> [(start0, len, max_cov)]."""
        rs = np.ascontiguousarray(ref_start, dtype=np.int32)
        re_ = np.ascontiguousarray(ref_end, dtype=np.int32)
        o = _abi.LcrRegionList()
        self._chk(self.lib.lcr_discover_regions(self.h, _abi.LCR_MEM_HOST, int(rs.size), rs.ctypes.data, re_.ctypes.data,
                                                int(contig_len), C.byref(o)), "lcr_discover_regions")
        return list(zip(_view(o.start0, np.int64, o.n_regions).tolist(), _view(o.len, np.int32, o.n_regions).tolist(),
                        _view(o.max_cov, np.uint32, o.n_regions).tolist()))

    # ---- stages ----------------------------------------------------------------------------------
    def fill_data_into_freq_vec(self):
        self._chk(self.lib.lcr_pileup(self.h, C.byref(self.params)), "lcr_pileup")
        return self

    def get_candidate_snps(self):
        self._chk(self.lib.lcr_candidates(self.h, C.byref(self.params)), "lcr_candidates")
        return self

    def get_fragments(self):
        self._chk(self.lib.lcr_fragments(self.h, C.byref(self.params)), "lcr_fragments")
        return self

    def phase(self):
        self._chk(self.lib.lcr_phase(self.h, C.byref(self.params)), "lcr_phase")
        return self

    def run_all(self):
        return self.fill_data_into_freq_vec().get_candidate_snps().get_fragments().phase()

    # ---- results ---------------------------------------------------------------------------------
    def columns(self):
        o = _abi.LcrColumns()
        self._chk(self.lib.lcr_get_columns(self.h, C.byref(o)), "lcr_get_columns")
        return _view(o.planes, np.uint32, _abi.NPLANES * o.n_cols).reshape(_abi.NPLANES, o.n_cols)

    def candidates(self):
        o = _abi.LcrCandidateList()
        self._chk(self.lib.lcr_get_candidates(self.h, C.byref(o)), "lcr_get_candidates")
        return (_view(o.cand, _abi.CAND_DTYPE, o.n_cand), _view(o.region_off, np.int32, o.n_regions + 1))

    def candidates_device(self):
        """(device pointer, count) of the candidate records in HBM (current after get_candidate_snps / phase)."""
        ptr, n = C.c_void_p(), C.c_int32()
        self._chk(self.lib.lcr_get_candidates_device(self.h, C.byref(ptr), C.byref(n)), "lcr_get_candidates_device")
        return int(ptr.value or 0), int(n.value)

    def read_records_device(self):
        """(device pointer, count) of the per-row results as 12-byte records in HBM (current after phase)."""
        ptr, n = C.c_void_p(), C.c_int32()
        self._chk(self.lib.lcr_get_read_records_device(self.h, C.byref(ptr), C.byref(n)), "lcr_get_read_records_device")
        return int(ptr.value or 0), int(n.value)

    def fragmat(self):
        o = _abi.LcrFragmat()
        self._chk(self.lib.lcr_get_fragmat(self.h, C.byref(o)), "lcr_get_fragmat")
        return dict(
            row_region_off=_view(o.row_region_off, np.int32, o.n_regions + 1),
            row_ptr=_view(o.row_ptr, np.int64, o.n_rows + 1), row_read=_view(o.row_read, np.int32, o.n_rows),
            col=_view(o.col, np.int32, o.nnz), val=_view(o.val, np.uint8, o.nnz),
            row_for_phasing=_view(o.row_for_phasing, np.uint8, o.n_rows),
            row_links=_view(o.row_links, np.uint32, o.n_rows))

    def phase_result(self):
        o = _abi.LcrPhaseResult()
        self._chk(self.lib.lcr_get_phase_result(self.h, C.byref(o)), "lcr_get_phase_result")
        return dict(haplotag=_view(o.haplotag, np.int8, o.n_rows), assignment=_view(o.assignment, np.uint8, o.n_rows),
                    phase_set=_view(o.phase_set, np.uint32, o.n_rows),
                    objective=_view(o.objective, np.float64, o.n_regions))

    def collect_phase(self, copy=False):
        """lcr_collect_phase: everything the last phase() produced -- valid after the NEXT batch has been bound and its pileup queued (until
        the next get_candidate_snps()): the getter of a pipelined caller under set_async_phase.  copy=False: the arrays ARE the context's
        buffers (read them before the next get_candidate_snps())."""
        o = _abi.LcrPhaseCollected()
        self._chk(self.lib.lcr_collect_phase(self.h, C.byref(o)), "lcr_collect_phase")
        v = _view if copy else _view_nocopy
        return dict(cand=v(o.cand, _abi.CAND_DTYPE, o.n_cand), cand_region_off=v(o.cand_region_off, np.int32, o.n_regions + 1),
                    row_region_off=v(o.row_region_off, np.int32, o.n_regions + 1),
                    haplotag=v(o.haplotag, np.int8, o.n_rows), assignment=v(o.assignment, np.uint8, o.n_rows),
                    phase_set=v(o.phase_set, np.uint32, o.n_rows), objective=v(o.objective, np.float64, o.n_regions),
                    dev_cand=(int(o.dev_cand or 0), int(o.n_cand)), dev_read_rec=(int(o.dev_read_rec or 0), int(o.n_rows)))

    def ld_blocks(self, region):
        """SNPFrag.ld_blocks of one region after phase(): list of lists of candidate indices (reference order)."""
        n, off, idx = C.c_int32(), C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        self._chk(self.lib.lcr_get_ld_blocks(self.h, int(region), C.byref(n), C.byref(off), C.byref(idx)), "lcr_get_ld_blocks")
        return [[idx[k] for k in range(off[b], off[b + 1])] for b in range(n.value)]

    TIE_FIELDS = ("sigma_f64", "sigma_flips", "delta_unresolved", "step_unresolved", "best_f64", "best_unresolved", "sigma_unresolved", "delta_step_f64")

    def tie_census(self):
        """exact fixed-point ties of the last phase() and how they were decided (include/lcr.h: lcr_get_tie_census)"""
        out = (C.c_uint64 * 8)()
        self._chk(self.lib.lcr_get_tie_census(self.h, out), "lcr_get_tie_census")
        return dict(zip(self.TIE_FIELDS, [int(x) for x in out]))

    def kernel_ms(self, k):
        ms = C.c_float()
        self._chk(self.lib.lcr_kernel_ms(self.h, k, C.byref(ms)), "lcr_kernel_ms")
        return float(ms.value)

    def pileup_bytes(self):
        b = C.c_int64()
        self._chk(self.lib.lcr_pileup_bytes(self.h, C.byref(b)), "lcr_pileup_bytes")
        return int(b.value)

    def pileup_stage_bytes(self):
        b = C.c_int64()
        self._chk(self.lib.lcr_pileup_stage_bytes(self.h, C.byref(b)), "lcr_pileup_stage_bytes")
        return int(b.value)


def host_register(*arrays):
    """Page-lock the memory of numpy arrays (lcr_host_register) so that lcr_load_batch_async copies them without staging;
    returns the list to hand to host_unregister."""
    lib = _lib.load()
    done = []
    for a in arrays:
        if a is None or a.nbytes == 0:
            continue
        if lib.lcr_host_register(C.c_void_p(a.ctypes.data), a.nbytes) != 0:
            host_unregister(done)
            raise LcrError("lcr_host_register failed")
        done.append(a)
    return done


def host_unregister(arrays):
    lib = _lib.load()
    for a in arrays:
        lib.lcr_host_unregister(C.c_void_p(a.ctypes.data))
