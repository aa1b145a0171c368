"""tools/stall_hunt3.py -- what triggers the 65-85 ms stalls of bench.py's side stages (profiles/r06_stall.txt)?

tools/stall_hunt2.py reproduced them (a fresh upload + a fresh context per pass: 4-10 of 24-32 passes; ONE resident batch: 0 of 200)
and tools/experiments/ioslow.c showed the main thread spinning in user space inside a HIP call while the HSA runtime's async-event
thread sleeps in AMDKFD_IOC_WAIT_EVENTS; the stall ends when that thread wakes.  This tool varies one condition at a time over the
same four batches (generator seeds 2..5), `reps` x 4 x 2 passes of 4 + 15 steps per variant:
  base          bench.py's seeds stage as it was (pin_memory().to(), timing events around the pileup stage, empty_cache)
  no_timing     no HIP events with timestamps (lcr_enable_timing off)
  no_empty      torch.cuda.empty_cache() left out
  keep_pinned   the page-locked staging tensors of the upload are kept alive until the pass is over
  settle        torch.cuda.synchronize() + 0.25 s of sleep between the upload and the first step
  sync_phase    lcr_ctx_set_async_phase off
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import ctypes as C

import numpy as np
import torch

import bench
from longcallr_amd import _abi, api, synth


def to_device_keep(batch, dev, keep_pinned):
    t, pinned = {}, []
    for f in batch.FIELDS + ["start0", "len", "col_off", "read_begin", "ref"]:
        a = getattr(batch, f)
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        elif a.dtype == np.uint32:
            a = a.view(np.int32)
        p = torch.from_numpy(np.ascontiguousarray(a)).pin_memory()
        if keep_pinned:
            pinned.append(p)
        t[f] = p.to(dev)
    reads, regions = batch.c_reads(), batch.c_regions()
    reads.mem = regions.mem = _abi.LCR_MEM_DEVICE
    for f in batch.FIELDS:
        setattr(reads, f, C.c_void_p(t[f].data_ptr()))
    for f in ["start0", "len", "col_off", "read_begin", "ref"]:
        setattr(regions, f, C.c_void_p(t[f].data_ptr()))
    return (reads, regions, t), pinned


def one_pass(variant, params, batch, steps=15, warm=4):
    dev = torch.device("cuda", 0)
    dv, pinned = to_device_keep(batch, dev, variant == "keep_pinned")
    if variant == "settle":
        torch.cuda.synchronize(); time.sleep(0.25)
    timing = False if variant == "no_timing" else (_abi.K_SPANS, _abi.K_PILEUP)
    E = api.Engine(0, params, timing=timing)
    E.set_async_phase(variant != "sync_phase")
    bench.run_steps_simple(E, dv, warm)
    torch.cuda.synchronize()
    tot = []
    for _ in range(steps):
        t0 = time.perf_counter()
        E.load_batch(dv)
        E.fill_data_into_freq_vec().get_candidate_snps().get_fragments().phase()
        if timing:
            E.kernel_ms(_abi.K_SPANS) + E.kernel_ms(_abi.K_PILEUP)
        tot.append((time.perf_counter() - t0) * 1e3)
    E.sync()
    E.close()
    del dv, pinned
    if variant != "no_empty":
        torch.cuda.empty_cache()
    return float(np.median(tot)), float(max(tot)), int(np.argmax(tot))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["base", "no_timing", "no_empty", "keep_pinned", "settle", "sync_phase", "base"]
    params = _abi.make_params(synth.preset_for("ont-cdna"))
    batches = [bench.build_workload("c3", seed=s) for s in (2, 3, 4, 5)]
    for v in variants:
        res = []
        for rep in range(reps):
            for b in batches:
                for k in range(2):
                    res.append(one_pass(v, params, b))
        stalls = [(round(m, 1), i) for (_, m, i) in res if m > 20]
        print(json.dumps(dict(variant=v, GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES"), passes=len(res), median_step_ms=round(float(np.median([r[0] for r in res])), 3),
                              passes_with_a_stall=len(stalls), stalls_ms_at_step=stalls)), flush=True)


if __name__ == "__main__":
    main()
