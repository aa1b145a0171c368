"""Repeat the full hot path on one batch and check that every output is bit-identical every time (the slot
allocation order of K0, LDS atomics and the two-queue phase stage must not leak into results)."""
import os, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from longcallr_amd import _abi, api, synth
import bench
prof = sys.argv[1] if len(sys.argv) > 1 else "ont-cdna"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
batch = synth.make_genes(prof, n_genes=200, gene_len=25000, depth=40, seed=1000)
p = _abi.make_params(synth.preset_for(prof), seed=7)
dv = bench.to_device(batch, torch, torch.device("cuda", 0))
E = api.Engine(0, p)
ref = None
for it in range(n):
    E.load_batch(dv).run_all()
    c, off = E.candidates(); fm = E.fragmat(); pr = E.phase_result()
    h = hashlib.sha256()
    for a in (E.columns(), c, off, fm["row_ptr"], fm["col"], fm["val"], fm["row_links"], pr["haplotag"], pr["assignment"], pr["phase_set"], pr["objective"]):
        h.update(np.ascontiguousarray(a).tobytes())
    d = h.hexdigest()
    if ref is None: ref = d
    assert d == ref, "run %d differs" % it
print(prof, "%d identical runs" % n, ref[:16])
