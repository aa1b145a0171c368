import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from longcallr_amd import _abi, api, synth
import bench
prof = sys.argv[1] if len(sys.argv) > 1 else "ont-cdna"
batch = synth.make_genes(prof, n_genes=400, gene_len=25000, depth=40, seed=1)   # (C3: 400 distinct genes, bench.py's workload)
p = _abi.make_params(synth.preset_for(prof))
dev = torch.device("cuda", 0)
reads, regions, keep = bench.to_device(batch, torch, dev)
torch.cuda.synchronize()
E = api.Engine(0, p, timing=True)
E.load_batch((reads, regions, keep))
print("bases", batch.bases.size, "cigar", batch.cigar.size, "reads", batch.n_reads)
# (ablations: rebuild liblcr with HIPCC flags -DLCR_K1_ABLATE=1|3|5, see k1_pileup.hip; the product build has no switch)
for dbg in ["product build"]:
    ts = []
    for _ in range(4):
        E.fill_data_into_freq_vec(); ts.append(E.kernel_ms(_abi.K_PILEUP))
    print("dbg", dbg, "k0 ms", E.kernel_ms(_abi.K_SPANS), "k1 ms", min(ts), "GB/s", E.pileup_bytes() / min(ts) / 1e6, "stage GB/s", E.pileup_stage_bytes() / (min(ts) + E.kernel_ms(_abi.K_SPANS)) / 1e6)
