"""configs[1]: demo.bam (one 13 kb region, 1 697 reads) — GPU hot path vs the CPU oracle port, end to end per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
from longcallr_amd import _abi, api
from oracle import orc
orc.build()
b = helpers.demo_batch()
p = _abi.make_params("hifi-masseq")
E = api.Engine(0, p)
for _ in range(150):   # steady state: the first ~50 passes of a fresh process run at lower clocks
    E.load_batch(b).run_all()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    E.load_batch(b).run_all()
E.sync()
gpu = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
m = 5
for _ in range(m):
    orc.Region(b, 0, p).run_all(orc.MODE_F64)
cpu = (time.perf_counter() - t0) / m
L = int(b.len[0])
print("demo.bam: %d columns, %d reads | GPU %.3f ms/pass = %.3g sites/s (host-resident inputs, H2D included) | CPU port 1 thread %.1f ms/pass = %.3g sites/s | ratio %.0fx"
      % (L, b.n_reads, gpu * 1e3, L / gpu, cpu * 1e3, L / cpu, cpu / gpu))
