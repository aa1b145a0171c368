import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
from longcallr_amd import _abi, api, synth
import bench
prof = sys.argv[1] if len(sys.argv) > 1 else "ont-cdna"
batch = synth.make_genes(prof, n_genes=400, gene_len=25000, depth=40, seed=1)   # (C3: 400 distinct genes, bench.py's workload)
p = _abi.make_params(synth.preset_for(prof))
dev = torch.device("cuda", 0)
dv = bench.to_device(batch, torch, dev)
E = api.Engine(0, p)
for _ in range(20): E.load_batch(dv).run_all()
acc = [0.0]*5; n = 30
for _ in range(n):
    t = [time.perf_counter()]
    E.load_batch(dv); E.sync(); t.append(time.perf_counter())
    E.fill_data_into_freq_vec(); E.sync(); t.append(time.perf_counter())
    E.get_candidate_snps(); E.sync(); t.append(time.perf_counter())
    E.get_fragments(); E.sync(); t.append(time.perf_counter())
    E.phase(); E.sync(); t.append(time.perf_counter())
    for k in range(5): acc[k] += t[k+1]-t[k]
print("ms: load %.3f pileup %.3f cand %.3f frag %.3f phase %.3f" % tuple(a/n*1e3 for a in acc))
os.environ["LCR_PHASE_PROF"] = "1"
for _ in range(3): E.load_batch(dv).run_all()
