"""tools/stall_hunt2.py -- the hunt, second attempt: 200 fresh-context passes over ONE resident batch showed no stall
(tools/stall_hunt.py, profiles/r06_stall.txt), so this one repeats bench.py's seeds stage literally -- a new batch per seed, uploaded
through pin_memory().to(), a fresh context, 4 warm-up + 15 timed steps, close, empty_cache -- with the wall time of every ABI call,
and brackets the timed window with two HIP calls nothing else makes (hipRuntimeGetVersion / hipDriverGetVersion) so that a
rocprofv3 --hip-trace of the run can be cut to the windows."""
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

hip = C.CDLL("libamdhip64.so")
CALLS = ("load", "pileup", "cand", "frag", "phase", "kms")


def one_pass(dev, params, batch, steps, warm, tag):
    dv = bench.to_device(batch, torch, torch.device("cuda", dev))
    E = api.Engine(dev, params, timing=(_abi.K_SPANS, _abi.K_PILEUP))
    E.set_async_phase(True)
    bench.run_steps_simple(E, dv, warm)
    torch.cuda.synchronize()
    v = C.c_int()
    hip.hipRuntimeGetVersion(C.byref(v))
    rows, t_abs = [], []
    t0 = time.perf_counter()
    for _ in range(steps):
        ts = [time.perf_counter()]
        t_abs.append(ts[0])
        E.load_batch(dv); ts.append(time.perf_counter())
        E.fill_data_into_freq_vec(); ts.append(time.perf_counter())
        E.get_candidate_snps(); ts.append(time.perf_counter())
        E.get_fragments(); ts.append(time.perf_counter())
        E.phase(); ts.append(time.perf_counter())
        E.kernel_ms(_abi.K_SPANS) + E.kernel_ms(_abi.K_PILEUP); ts.append(time.perf_counter())
        rows.append([round((ts[i + 1] - ts[i]) * 1e3, 3) for i in range(6)])
    E.sync()
    dt = (time.perf_counter() - t0) / steps
    hip.hipDriverGetVersion(C.byref(v))
    tot = [sum(r) for r in rows]
    slow = [dict(step=i, t_abs=round(t_abs[i], 6), ms=round(t, 2), calls=dict(zip(CALLS, rows[i]))) for i, t in enumerate(tot) if t > 3 * float(np.median(tot))]
    print(json.dumps(dict(tag=tag, ms_per_step=round(dt * 1e3, 3), median_step=round(float(np.median(tot)), 3), max_step=round(max(tot), 3), slow=slow)), flush=True)
    E.close()
    del dv
    torch.cuda.empty_cache()
    return dt


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    params = _abi.make_params(synth.preset_for("ont-cdna"))
    res = []
    for rep in range(reps):
        for s in (2, 3, 4, 5):
            b = bench.build_workload("c3", seed=s)
            for k in range(2):
                res.append(one_pass(0, params, b, 15, 4, "rep%d seed%d pass%d" % (rep, s, k)))
    r = np.array(res) * 1e3
    print(json.dumps(dict(passes=len(res), ms_per_step_median=float(np.median(r)), ms_per_step_max=float(r.max()), max_over_median=float(r.max() / np.median(r)),
                          passes_over_1p5x=int((r > 1.5 * np.median(r)).sum()))))


if __name__ == "__main__":
    main()
