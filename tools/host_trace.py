"""Host-side durations of the ABI calls of the pipelined step (bench.py pipelined_step) on the headline workload: where does the calling thread wait?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np
import torch
import bench
from longcallr_amd import _abi, api, synth
WL = os.environ.get("HT_WORKLOAD", "c3")
params = _abi.make_params(synth.preset_for("ont-cdna" if WL == "c3" else "masseq"))
dev = torch.device("cuda", 0)
b = bench.build_workload(WL, seed=1)
dv = bench.to_device(b, torch, dev)
timing = tuple(getattr(_abi, k) for k in bench.PILE_TIMERS) if os.environ.get("HT_TIMERS", "1") == "1" else ()
E = api.Engine(0, params, timing=timing); E.set_async_phase(True)
bench.run_steps_simple(E, dv, 20)
names = ("load_batch", "pileup", "collect", "candidates", "fragments", "phase")
rows = []
sink = [0, 0, 0]
E.load_batch(dv); E.fill_data_into_freq_vec(); E.get_candidate_snps().get_fragments().phase()
for _ in range(60):
    ts = [time.perf_counter()]
    E.load_batch(dv); ts.append(time.perf_counter())
    E.fill_data_into_freq_vec(); ts.append(time.perf_counter())
    bench.consume(E.collect_phase(), sink); ts.append(time.perf_counter())
    E.get_candidate_snps(); ts.append(time.perf_counter())
    E.get_fragments(); ts.append(time.perf_counter())
    E.phase(); ts.append(time.perf_counter())
    rows.append(np.diff(ts) * 1e6)
E.collect_phase(); E.sync()
rows = np.array(rows)
print("median us per call:", {n: round(float(np.median(rows[:, i])), 1) for i, n in enumerate(names)}, "step", round(float(np.median(rows.sum(1))), 1))
