"""ctypes loader of liblcr.so (the HIP product library).  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("LCR_LIB") or os.path.join(_HERE, "liblcr.so")   # LCR_LIB: developer hook (A/B timing of two builds)

# every symbol include/lcr.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "lcr_params_preset", "lcr_ctx_create", "lcr_ctx_destroy", "lcr_last_error", "lcr_ctx_set_stream",
    "lcr_ctx_sync", "lcr_ctx_set_lock_dir", "lcr_ctx_set_async_phase", "lcr_debug_set", "lcr_load_batch", "lcr_load_batch_async", "lcr_bind_batch",
    "lcr_host_alloc", "lcr_host_free", "lcr_host_register", "lcr_host_unregister", "lcr_pileup", "lcr_get_columns", "lcr_candidates",
    "lcr_get_candidates", "lcr_get_candidates_device", "lcr_fragments", "lcr_get_fragmat", "lcr_phase", "lcr_get_phase_result", "lcr_get_read_records_device", "lcr_collect_phase", "lcr_get_ld_blocks", "lcr_get_tie_census",
    "lcr_enable_timing", "lcr_kernel_ms", "lcr_pileup_bytes", "lcr_pileup_stage_bytes", "lcr_discover_regions", "lcr_version", "lcr_release_cached_memory", "lcr_set_cache_limits",
    "lcr_bam_open", "lcr_bam_open_keep", "lcr_bam_close", "lcr_bam_last_error", "lcr_bam_refs", "lcr_bam_n_records", "lcr_bam_resident", "lcr_bam_spans", "lcr_bam_batch", "lcr_bam_write_phased", "lcr_bam_write_reads",
]

_lib = None


class LcrError(RuntimeError):
    pass


def load():
    """Load liblcr.so.  Raises if the HIP extension has not been built (python -m longcallr_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise LcrError("liblcr.so is missing: build it with `python -m longcallr_amd.build` "
                       "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7; importing torch first
    # makes liblcr.so (NEEDED libamdhip64.so.7) bind to that already-loaded runtime instead of a second
    # copy from /opt/rocm, which could not see the GPU ("No HIP GPUs are available").
    try:
        import torch  # noqa: F401
    except Exception:  # torch is plumbing only; liblcr works without it
        pass
    l = C.CDLL(SO_PATH)
    vp, i32 = C.c_void_p, C.c_int
    l.lcr_version.restype = C.c_char_p
    l.lcr_last_error.restype = C.c_char_p
    l.lcr_last_error.argtypes = [vp]
    l.lcr_params_preset.argtypes = [i32, C.POINTER(_abi.LcrParams)]
    l.lcr_ctx_create.argtypes = [i32, C.POINTER(vp)]
    l.lcr_ctx_destroy.argtypes = [vp]
    l.lcr_ctx_destroy.restype = None
    l.lcr_ctx_set_stream.argtypes = [vp, vp]
    l.lcr_ctx_sync.argtypes = [vp]
    l.lcr_ctx_set_lock_dir.argtypes = [vp, C.c_char_p]
    l.lcr_ctx_set_async_phase.argtypes = [vp, C.c_int32]
    l.lcr_debug_set.argtypes = [vp, C.c_char_p, C.c_int64]
    l.lcr_load_batch.argtypes = [vp, C.POINTER(_abi.LcrReads), C.POINTER(_abi.LcrRegions)]
    l.lcr_load_batch_async.argtypes = [vp, C.POINTER(_abi.LcrReads), C.POINTER(_abi.LcrRegions), C.c_int32]
    l.lcr_bind_batch.argtypes = [vp, C.c_int32]
    l.lcr_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
    l.lcr_host_free.argtypes = [vp]
    l.lcr_host_free.restype = None
    l.lcr_host_register.argtypes = [vp, C.c_size_t]
    l.lcr_host_unregister.argtypes = [vp]
    for f in ("lcr_pileup", "lcr_candidates", "lcr_fragments", "lcr_phase"):
        getattr(l, f).argtypes = [vp, C.POINTER(_abi.LcrParams)]
    l.lcr_get_columns.argtypes = [vp, C.POINTER(_abi.LcrColumns)]
    l.lcr_get_candidates.argtypes = [vp, C.POINTER(_abi.LcrCandidateList)]
    l.lcr_get_candidates_device.argtypes = [vp, C.POINTER(vp), C.POINTER(i32)]
    l.lcr_get_read_records_device.argtypes = [vp, C.POINTER(vp), C.POINTER(i32)]
    l.lcr_get_fragmat.argtypes = [vp, C.POINTER(_abi.LcrFragmat)]
    l.lcr_get_phase_result.argtypes = [vp, C.POINTER(_abi.LcrPhaseResult)]
    l.lcr_collect_phase.argtypes = [vp, C.POINTER(_abi.LcrPhaseCollected)]
    l.lcr_release_cached_memory.argtypes = []
    l.lcr_set_cache_limits.argtypes = [C.c_int64, C.c_int64]
    l.lcr_get_tie_census.argtypes = [vp, C.POINTER(C.c_uint64)]
    l.lcr_get_ld_blocks.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32))]
    l.lcr_discover_regions.argtypes = [vp, C.c_int32, C.c_int32, vp, vp, C.c_int64, C.POINTER(_abi.LcrRegionList)]
    l.lcr_enable_timing.argtypes = [vp, i32]
    l.lcr_kernel_ms.argtypes = [vp, i32, C.POINTER(C.c_float)]
    l.lcr_pileup_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
    l.lcr_pileup_stage_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
    flt = C.POINTER(_abi.LcrReadFilter)
    l.lcr_bam_open.argtypes = [C.c_char_p, C.c_int32, C.POINTER(vp)]
    l.lcr_bam_open_keep.argtypes = [C.c_char_p, C.c_int32, C.c_int64, C.POINTER(vp)]
    l.lcr_bam_close.argtypes = [vp]
    l.lcr_bam_close.restype = None
    l.lcr_bam_last_error.argtypes = [vp]
    l.lcr_bam_last_error.restype = C.c_char_p
    l.lcr_bam_refs.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_int64))]
    l.lcr_bam_n_records.argtypes = [vp, C.POINTER(C.c_int64)]
    l.lcr_bam_resident.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    l.lcr_bam_spans.argtypes = [vp, C.c_int32, flt, C.POINTER(C.c_int32), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32))]
    l.lcr_bam_batch.argtypes = [vp, C.c_int32, flt, C.c_int32, vp, vp, C.POINTER(_abi.LcrReads), C.POINTER(C.POINTER(C.c_int32)),
                                C.POINTER(C.POINTER(C.c_uint64)), C.POINTER(C.c_char_p)]
    l.lcr_bam_write_phased.argtypes = [vp, C.c_char_p, C.c_int32, vp, vp, vp, C.c_int64, vp, C.c_char_p, vp, vp, C.c_int32, C.c_int32]
    l.lcr_bam_write_reads.argtypes = [C.c_char_p, C.c_char_p, C.c_int64, C.POINTER(_abi.LcrReads), C.c_int32, C.c_int32]
    _lib = l
    return l
