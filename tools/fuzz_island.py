"""Coverage islands of other shapes through the all-CU kernels (k4_stage_grid, k4_chain_grid with the batched rounds, k4_gpost, k2_hist_tiles)
against the oracle: usage fuzz_island.py a b"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as t
from oracle import orc
from longcallr_amd import _abi, api, synth
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    n_loci, depth = (2, 3, 5, 6)[seed % 4], (90, 150, 260)[seed % 3]
    b = synth.make_island("ont-drna-c5", n_loci=n_loci, locus_len=(12000, 25000)[seed & 1], depth=depth, seed=40 + seed)
    try:
        c = t.full_check(api.Engine, orc, b, _abi.make_params("ont-drna", seed=seed))
        print("seed", seed, "loci", n_loci, "depth", depth, "candidates", c.size, flush=True)
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, n_loci, depth, str(e)[:300], flush=True)
print("island sweep: %d mismatches" % bad)
