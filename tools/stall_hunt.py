"""tools/stall_hunt.py -- where do the 60-85 ms stalls of fresh contexts come from (VERDICT r05 "weak" item 3)?

Runs P passes of W + K steps of the C3 step, every pass in a FRESH context (what bench.py's side stages do), and records per step
the wall time of every ABI call; Python's collector is watched through gc.callbacks.  Variants: --gc off (collector disabled),
--no-empty-cache (torch.cuda.empty_cache() between the passes left out), --reuse (one context for all passes).
Output: one JSON line per variant with max / median of the step times, the slow steps with their per-call breakdown and the collector's
pauses that fall inside them."""
import argparse
import gc
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


def run_variant(name, dv, params, passes, warm, steps, use_gc, empty_cache, reuse, async_phase):
    gc_events = []
    t_gc = [0.0]

    def cb(phase, info):
        if phase == "start":
            t_gc[0] = time.perf_counter()
        else:
            gc_events.append((t_gc[0], time.perf_counter() - t_gc[0], info["generation"]))
    gc.callbacks.append(cb)
    if not use_gc:
        gc.disable()
    rows = []   # (pass, step, t_start, total, per-call...)
    E = None
    t_create = []
    for p in range(passes):
        if E is None or not reuse:
            t0 = time.perf_counter()
            E = api.Engine(0, params, timing=(_abi.K_SPANS, _abi.K_PILEUP))
            E.set_async_phase(async_phase)
            t_create.append(time.perf_counter() - t0)
        for s in range(warm + steps):
            ts = [time.perf_counter()]
            E.load_batch(dv); ts.append(time.perf_counter())
            E.fill_data_into_freq_vec(); ts.append(time.perf_counter())
            E.get_candidate_snps(); ts.append(time.perf_counter())
            E.get_fragments(); ts.append(time.perf_counter())
            E.phase(); ts.append(time.perf_counter())
            E.kernel_ms(_abi.K_SPANS) + E.kernel_ms(_abi.K_PILEUP); ts.append(time.perf_counter())
            rows.append((p, s, ts[0], ts[-1] - ts[0]) + tuple(ts[i + 1] - ts[i] for i in range(6)))
        E.sync()
        if not reuse:
            E.close()
            if empty_cache:
                torch.cuda.empty_cache()
    if reuse and E is not None:
        E.close()
    gc.callbacks.remove(cb)
    gc.enable()
    a = np.array([r[3] for r in rows]) * 1e3
    timed = np.array([r[3] for r in rows if r[1] >= warm]) * 1e3
    med = float(np.median(timed))
    slow = []
    for r in rows:
        if r[3] * 1e3 > 3 * med and r[1] >= 1:   # (step 0 of a fresh context allocates: expected)
            inside = [(round(d * 1e3, 2), g) for (t, d, g) in gc_events if r[2] <= t <= r[2] + r[3]]
            slow.append(dict(pass_=r[0], step=r[1], ms=round(r[3] * 1e3, 2), calls={c: round(r[4 + i] * 1e3, 2) for i, c in enumerate(CALLS)}, gc=inside))
    per_pass = [float(np.sum([r[3] for r in rows if r[0] == p and r[1] >= warm]) * 1e3 / steps) for p in range(passes)]
    out = dict(variant=name, passes=passes, warm=warm, steps=steps, step_ms_p50=med, step_ms_p99=float(np.percentile(timed, 99)), step_ms_max=float(timed.max()),
               pass_ms_per_step_max_over_median=float(max(per_pass) / np.median(per_pass)), first_step_ms_median=float(np.median([r[3] for r in rows if r[1] == 0]) * 1e3),
               ctx_create_ms_median=float(np.median(t_create) * 1e3), gc_events=[(round(d * 1e3, 2), g) for (_, d, g) in gc_events if d > 1e-3][:40],
               n_gc=len(gc_events), slow_steps=slow[:60], all_steps_max_ms=float(a.max()))
    print(json.dumps(out), flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--passes", type=int, default=40)
    ap.add_argument("--warm", type=int, default=5)
    ap.add_argument("--steps", type=int, default=15)
    ap.add_argument("--variants", default="bench,gc_off,no_empty_cache,reuse,sync_phase")
    ap.add_argument("--seed", type=int, default=2)
    a = ap.parse_args()
    batch = bench.build_workload("c3", seed=a.seed)
    params = _abi.make_params(synth.preset_for("ont-cdna"))
    dv = bench.to_device(batch, torch, torch.device("cuda", 0))
    torch.cuda.synchronize()
    for v in a.variants.split(","):
        run_variant(v, dv, params, a.passes, a.warm, a.steps, use_gc=v != "gc_off", empty_cache=v != "no_empty_cache", reuse=v == "reuse",
                    async_phase=v != "sync_phase")


if __name__ == "__main__":
    main()
