"""tools/stall_hunt4.py -- which object's lifetime triggers the 65-85 ms stalls (profiles/r06_stall.txt)?

The earlier tools of the round: a fresh upload + a fresh context per pass stalls in 4-10 of 24-32 passes, 200 fresh-context passes over
ONE resident batch never do; under the stall the calling thread spins in user space inside a HIP call while the HSA runtime's
async-event thread sleeps in AMDKFD_IOC_WAIT_EVENTS.  This one separates the upload from the context, with the slow ABI call named:
  base        bench.py's side stage: per pass a fresh upload (pin_memory().to()), a fresh context, 4 + 15 steps, close, del, empty_cache
  keep_ctx    ONE context for all passes (what a long-lived worker holds, thread.rs:77-143), a fresh upload per pass, the old one freed
  same_dv     one upload per seed kept across the passes, a fresh context per pass
  no_empty    base without torch.cuda.empty_cache()
  leak_dv     base, but the uploads are kept alive until the variant is over (nothing is freed between passes)
  presync     base + hipDeviceSynchronize + 50 ms of sleep after the upload and after the context's warm-up
  ctx_first   base, but the context is created BEFORE the upload (its buffers are still allocated by the first warm-up step, after it)
  create_sleep  base + 120 ms of sleep between lcr_ctx_create and the warm-up steps
  warm_sleep  base + 120 ms of sleep after the warm-up steps
  idle150     ONE context, ONE resident upload (nothing is created, copied or freed between passes), 150 ms of sleep before every pass:
              is an idle GPU enough?
  copy_idle   as idle150, but instead of sleeping the host copies 1 GB to the device (torch, pinned) before every pass
  hwq4        base with GPU_MAX_HW_QUEUES=4 (set by the caller in the environment: tools/r06_stall6.sh)
"""
import ctypes as C
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch

import bench
from longcallr_amd import _abi, api, synth

CALLS = ("load", "pileup", "cand", "frag", "phase", "kms")


def steps_timed(E, dv, steps):
    rows = []
    for _ in range(steps):
        ts = [time.perf_counter()]
        E.load_batch(dv); ts.append(time.perf_counter())
        E.fill_data_into_freq_vec(); ts.append(time.perf_counter())
        E.get_candidate_snps(); ts.append(time.perf_counter())
        E.get_fragments(); ts.append(time.perf_counter())
        E.phase(); ts.append(time.perf_counter())
        E.kernel_ms(_abi.K_SPANS) + E.kernel_ms(_abi.K_PILEUP); ts.append(time.perf_counter())
        rows.append([(ts[i + 1] - ts[i]) * 1e3 for i in range(6)])
    E.sync()
    return rows


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    variants = sys.argv[2].split(",") if len(sys.argv) > 2 else ["base", "keep_ctx", "same_dv", "no_empty", "leak_dv", "presync", "base"]
    params = _abi.make_params(synth.preset_for("ont-cdna"))
    dev = torch.device("cuda", 0)
    batches = [bench.build_workload("c3", seed=s) for s in (2, 3, 4, 5)]
    for v in variants:
        res, leaked = [], []
        E_keep = None
        for rep in range(reps):
            for b in batches:
                dv_keep = bench.to_device(b, torch, dev) if v in ("same_dv", "idle150", "copy_idle") else None
                for k in range(2):
                    E_pre = None
                    if v == "ctx_first":
                        E_pre = api.Engine(0, params, timing=(_abi.K_SPANS, _abi.K_PILEUP)); E_pre.set_async_phase(True)
                    dv = dv_keep if dv_keep is not None else bench.to_device(b, torch, dev)
                    if v == "presync":
                        torch.cuda.synchronize(); time.sleep(0.05)
                    if v == "idle150":
                        torch.cuda.synchronize(); time.sleep(0.15)
                    if v == "copy_idle":
                        junk = bench.to_device(b, torch, dev); torch.cuda.synchronize(); del junk
                    if v in ("keep_ctx", "idle150", "copy_idle"):
                        if E_keep is None:
                            E_keep = api.Engine(0, params, timing=(_abi.K_SPANS, _abi.K_PILEUP)); E_keep.set_async_phase(True)
                        E = E_keep
                    elif E_pre is not None:
                        E = E_pre
                    else:
                        E = api.Engine(0, params, timing=(_abi.K_SPANS, _abi.K_PILEUP)); E.set_async_phase(True)
                    if v == "create_sleep":
                        time.sleep(0.12)
                    bench.run_steps_simple(E, dv, 4)
                    torch.cuda.synchronize()
                    if v == "presync":
                        time.sleep(0.05)
                    if v == "warm_sleep":
                        time.sleep(0.12)
                    rows = steps_timed(E, dv, 15)
                    tot = [sum(r) for r in rows]
                    i = int(np.argmax(tot))
                    res.append((float(np.median(tot)), tot[i], i, CALLS[int(np.argmax(rows[i]))], round(max(rows[i]), 1)))
                    if v not in ("keep_ctx", "idle150", "copy_idle"):
                        E.close()
                    if v == "leak_dv":
                        leaked.append(dv)
                    del dv
                    if v not in ("no_empty", "leak_dv"):
                        torch.cuda.empty_cache()
                del dv_keep
        if E_keep is not None:
            E_keep.close()
        del leaked
        torch.cuda.empty_cache()
        stalls = [(round(m, 1), i, c, ms) for (_, m, i, c, ms) in res if m > 20]
        print(json.dumps(dict(variant=v, passes=len(res), median_step_ms=round(float(np.median([r[0] for r in res])), 3),
                              max_over_median=round(max(r[1] for r in res) / float(np.median([r[0] for r in res])), 1),
                              passes_with_a_stall=len(stalls), stalls_ms_step_call=stalls)), flush=True)


if __name__ == "__main__":
    main()
