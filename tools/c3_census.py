"""Tie census of the headline batch (C3) with the chain regions' second run on and off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bench
from longcallr_amd import _abi, api, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
params = _abi.make_params(synth.preset_for("ont-cdna" if wl == "c3" else "masseq")) if hasattr(synth, "preset_for") else None
b = bench.build_workload(wl, seed=1)
dv = bench.to_device(b, torch, torch.device("cuda", 0))
for ct in (0, 1):
    E = api.Engine(0, params)
    E.debug_set("chain_ties", ct)
    E.load_batch(dv).run_all()
    print(wl, "chain_ties", ct, E.tie_census(), flush=True)
    E.close()
