"""Sweep of the chain branch of the phase stage (S > max_enum_snps: LD blocks, block flip, perturbation rounds, post-phase)
against the oracle: ONT-dRNA / ONT-cDNA gene batches of seeds a .. b, each through the one-workgroup kernels and -- every
fourth seed -- with all CUs on every region (LCR_GRID_MIN_ENTRIES=0).  usage: fuzz_chain.py a b"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as t
from oracle import orc
from longcallr_amd import _abi, api, synth
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = n_chain = n_reg = n_unres = 0
for seed in range(a, b):
    prof = ("ont-drna", "ont-cdna")[seed & 1]
    depth = (35, 60, 90)[seed % 3]
    batch = synth.make_batch(prof, n_genes=3, gene_len=(12000, 20000)[(seed >> 1) & 1], depth=depth, seed=seed)
    p = _abi.make_params(prof, seed=seed)
    for grid in ((False, True) if seed % 4 == 0 else (False,)):
        if grid:
            os.environ["LCR_GRID_MIN_ENTRIES"] = "0"
        t.ORACLE_TIE_MASK[0] = orc.TIE_MASK_LIBLCR_GRID if grid else None   # (all CUs on a region: sigma ties only)
        try:
            c = t.full_check(api.Engine, orc, batch, p)
            if not grid:
                S = np.bincount(c["region"], minlength=batch.n_regions)
                n_chain += int((S > p.max_enum_snps).sum()); n_reg += batch.n_regions
        except AssertionError as e:
            # forced all-CU staging sends the ENUMERATION regions of the batch through the global-memory kernel, which keeps the
            # fixed-point contract (ties counted as unresolved): such a batch must agree under that contract on both sides
            ok = False
            if grid:
                os.environ["LCR_TIE_ARITH"] = "0"; saved = dict(t.ORACLE_TIE_MASK); t.ORACLE_TIE_MASK[0] = 0
                try:
                    t.full_check(api.Engine, orc, batch, p); ok = True; n_unres += 1
                except AssertionError:
                    pass
                finally:
                    os.environ.pop("LCR_TIE_ARITH", None); t.ORACLE_TIE_MASK.clear(); t.ORACLE_TIE_MASK.update(saved)
            if not ok:
                bad += 1
                print("MISMATCH seed", seed, prof, "grid" if grid else "wg", str(e)[:200], flush=True)
        finally:
            os.environ.pop("LCR_GRID_MIN_ENTRIES", None)
            t.ORACLE_TIE_MASK[0] = None
print("seeds %d..%d: %d regions, %d on the chain branch, %d mismatches (%d forced-grid batches agreed under the fixed-point contract only: ties of their global-memory enumeration regions)" % (a, b, n_reg, n_chain, bad, n_unres))
