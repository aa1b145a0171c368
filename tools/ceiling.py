"""tools/ceiling.py -- runs tools/experiments/ceiling.hip on C3's arrays and prints the ceiling of the per-op-record pileup
decomposition beside the product kernels' times (output kept as profiles/r06_pileup_ceiling.txt).
  hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/experiments/ceiling.hip -o gpurun_in/libceil.so   (here; the .so travels)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
from longcallr_amd import _abi, api, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl = sys.argv[1] if len(sys.argv) > 1 else "c3"
lib = C.CDLL(os.path.join(ROOT, "gpurun_in", "libceil.so"))
lib.ceil_scan.restype = C.c_float
lib.ceil_scan.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
lib.ceil_compare.restype = C.c_float
lib.ceil_compare.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int]

batch = bench.build_workload(wl, seed=1)
prof = "ont-cdna" if wl == "c3" else "masseq"
p = _abi.make_params(synth.preset_for(prof))
dev = torch.device("cuda", 0)
dv = bench.to_device(batch, torch, dev)
t = dv[2]
E = api.Engine(0, p, timing=True)
E.load_batch(dv)
k0, k1 = [], []
for _ in range(6):
    E.load_batch(dv); E.fill_data_into_freq_vec()
    k0.append(E.kernel_ms(_abi.K_SPANS)); k1.append(E.kernel_ms(_abi.K_PILEUP))
planes = E.columns()
L = planes.shape[1]
T = 256
nt = L // T
act = (planes[[0, 1, 2, 3, 5, 6]].sum(axis=0) > 0)[:nt * T].reshape(nt, T).any(axis=1)
nP = planes[4][:nt * T].reshape(nt, T)
act |= (nP.max(axis=1) != nP.min(axis=1))
n_full = int(act.sum())
B, Cops, R = int(batch.bases.size), int(batch.cigar.size), int(batch.n_reads)
stage_bytes = E.pileup_stage_bytes()
print("workload %s: B = %d aligned-read bases, C = %d CIGAR ops, R = %d reads, L = %d columns, %d of %d tiles of %d columns hold records" % (wl, B, Cops, R, L, n_full, nt, T))
print("strict algorithmic bytes of the stage (B + 4C + 37R + 53L) = %d" % stage_bytes)
print("product kernels (HIP events, min of 6): k0_ops group %.4f ms, tally group (k1_tiles_a/b + k0_desc_bin + k1_pileup + k1_empty_tiles) %.4f ms, sum %.4f ms = %.1f %% of 8 TB/s"
      % (min(k0), min(k1), min(k0) + min(k1), stage_bytes / ((min(k0) + min(k1)) * 1e-3) / 8e12 * 100))
del planes
out = torch.empty(((Cops + 1023) // 1024) * 256, dtype=torch.int32, device=dev)
pl = torch.empty(13 * L, dtype=torch.int32, device=dev)
reps = 20
ms_scan = lib.ceil_scan(t["cigar"].data_ptr(), Cops, out.data_ptr(), reps)
print("ceil_scan   : %.4f ms  (4C = %d B read at %.2f TB/s)" % (ms_scan, 4 * Cops, 4 * Cops / ms_scan / 1e9))
# records: as many as there are M / D / I / N ops, in K0's layout, random geometry (mean M length as in the batch)
ops = batch.cigar & 15
n_rec = int(np.isin(ops, [0, 1, 2, 3, 7, 8]).sum())
m_len = (batch.cigar >> 4)[np.isin(ops, [0, 7, 8])]
n_m = int(m_len.size)
pieces_real = int(((m_len + 15) // 16).sum())   # (ignores tile crossings: + a few %)
rng = np.random.default_rng(1)
hi = (rng.integers(0, 256, n_rec, dtype=np.uint64) << 8) | (np.minimum(rng.geometric(1.0 / max(float(m_len.mean()), 1.0), n_rec), 255).astype(np.uint64) << 18) | \
     (rng.integers(0, 2, n_rec, dtype=np.uint64) << 28) | (rng.integers(0, 3, n_rec, dtype=np.uint64) << 29)
recs = torch.from_numpy(((hi << 32) | rng.integers(0, 1 << 32, n_rec, dtype=np.uint64)).view(np.int64)).pin_memory().to(dev)
print("records: %d (M %d, mean length %.1f, median %d); 16-byte pieces of the M segments: %d (ideal B / 16 = %d)" % (n_rec, n_m, m_len.mean(), int(np.median(m_len)), pieces_real, B // 16))
res = {}
for name, npieces, mode, nrec in (("compare, ideal pieces, coalesced", B // 16, 0, 0), ("compare, ideal pieces, unaligned", B // 16, 1, 0),
                                  ("compare, real piece count, unaligned", pieces_real, 1, 0), ("compare + records, real piece count, unaligned", pieces_real, 1, n_rec)):
    ms = lib.ceil_compare(t["bases"].data_ptr(), B, npieces, mode, recs.data_ptr(), nrec, t["ref"].data_ptr(), L, n_full, pl.data_ptr(), reps)
    res[name] = ms
    print("ceil_compare: %.4f ms  %s  (%d pieces%s, %d workgroups)" % (ms, name, npieces, ", %d records" % nrec if nrec else "", n_full))
empty = 0.053 if wl == "c3" else None
e_cols = (nt - n_full) * T
print("record-free tiles: %d columns x 52 B = %d B of stores (k1_empty_tiles: 0.053 ms on C3)" % (e_cols, e_cols * 52))
for name in res:
    tot = ms_scan + res[name] + (empty or 0.0)
    print("ceiling = scan %.4f + [%s] %.4f + record-free stores %.3f = %.4f ms  => %.1f %% of 8 TB/s on the strict bytes" % (ms_scan, name, res[name], empty or 0.0, tot, stage_bytes / (tot * 1e-3) / 8e12 * 100))
