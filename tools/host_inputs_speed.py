"""PCIe-inclusive step: the C3 batch handed over as host buffers (LCR_MEM_HOST) every step (never bench.py's `value`)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from longcallr_amd import _abi, api, synth
import bench
batch = synth.make_genes("ont-cdna", n_genes=400, gene_len=25000, depth=40, seed=1)   # (C3: 400 distinct genes, bench.py's workload)
p = _abi.make_params("ont-cdna")
E = api.Engine(0, p)
for _ in range(5): E.load_batch(batch).run_all()
n = 20
t0 = time.perf_counter()
for _ in range(n): E.load_batch(batch).run_all()
E.sync(); dt = (time.perf_counter() - t0) / n
nbytes = sum(getattr(batch, f).nbytes for f in _abi.ReadBatch.FIELDS) + batch.ref.nbytes
print("host-resident inputs: %.2f ms/step, %.2f GB uploaded per step = %.1f GB/s effective, %.2e sites/s"
      % (dt * 1e3, nbytes / 1e9, nbytes / dt / 1e9, int(batch.col_off[-1]) / dt))
