#!/bin/bash
# round 6: phase ablations of k0_ops (K0_ABL = 1..4: leave after heads / scan / geometry / count) and k1_pileup (LCR_K1_ABLATE 1 / 3 / 5) on C3
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06; mkdir -p $O
python - > $O/ablation_c3.txt 2>$O/ablation_c3.err <<'PY'
import os, sys, subprocess
sys.path.insert(0, os.getcwd())
code = r'''
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from longcallr_amd import _abi, api, synth
import bench
b = bench.build_workload("c3")
p = _abi.make_params(synth.preset_for(bench.WORKLOADS["c3"][0]), seed=2025)
dv = bench.to_device(b, torch, torch.device("cuda", 0))
E = api.Engine(0, p, timing=True)
ts = []
for i in range(30):
    try:
        E.load_batch(dv); E.fill_data_into_freq_vec(); E.sync()
    except Exception as e:
        pass
    ts.append((E.kernel_ms(_abi.K_SPANS), E.kernel_ms(_abi.K_PILEUP)))
ts = np.array(ts[10:])
print(os.environ.get("LCR_LIB", "product"), "k0 %.4f ms  k1 group %.4f ms" % (ts[:, 0].min(), ts[:, 1].min()), flush=True)
'''
for lib in [None] + ["gpurun_in/liblcr_k0abl%d.so" % n for n in (1, 2, 3, 4)] + ["gpurun_in/liblcr_k1abl%d.so" % n for n in (1, 3, 5)]:
    env = dict(os.environ)
    if lib: env["LCR_LIB"] = os.path.abspath(lib)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ("FAILED " + str(lib) + " " + r.stderr[-400:]), flush=True)
PY
cat $O/ablation_c3.txt; tail -3 $O/ablation_c3.err
