import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as t
from oracle import orc
from longcallr_amd import _abi, api, synth
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 377
prof = ("ont-cdna", "masseq", "ont-drna")[seed % 3]
mx = (2, 5, 12, 14, 10)[seed % 5]
b = synth.make_batch(prof, n_genes=4, gene_len=(9000, 16000)[seed & 1], depth=(30, 45)[(seed >> 1) & 1], seed=100 + seed)
p = _abi.make_params(synth.preset_for(prof), seed=seed, max_enum_snps=mx)
print("lib", os.environ.get("LCR_LIB"), "enum_bits", os.environ.get("LCR_ENUM_BITS"), "fuse", os.environ.get("LCR_FUSE_FILTER"))
regs = t.oracle_all(orc, b, p)
E = api.Engine(0, p); E.load_batch(b).run_all()
c, off = E.candidates()
print("census", E.tie_census())
for g, R in enumerate(regs):
    oc = R.cands(); hc = c[off[g]:off[g + 1]]
    if len(oc) != len(hc):
        print("region", g, "candidate counts differ", len(oc), len(hc)); continue
    for f in t.INT_FIELDS:
        if f != "region" and not np.array_equal(oc[f], hc[f]):
            d = np.flatnonzero(oc[f] != hc[f])
            print("region", g, "S", len(oc), f, "differs at", d[:8], "oracle", oc[f][d[:8]], "hip", hc[f][d[:8]], "oracle census", R.tie_census())
    A = orc.Region(b, g, p).set_fast(1).run_all(orc.MODE_F64)
    ac = A.cands()
    for f in ("variant_type", "genotype", "haplotype"):
        if not np.array_equal(ac[f], oc[f]): print("   (F64 vs TIE oracle differ in", f, "region", g, ")")
        if not np.array_equal(ac[f], hc[f]): print("   (F64 oracle vs HIP differ in", f, "region", g, ")")
E.close()
