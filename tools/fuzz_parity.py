"""Wider sweep of tests/test_gpu_parity.py::test_random_cigar_structures: seeds a .. b, all presets."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as t
from oracle import orc
from longcallr_amd import _abi, api
a, b = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(a, b):
    batch = t._random_batch(seed)
    for preset in ("hifi-masseq", "hifi-isoseq", "ont-cdna", "ont-drna"):
        try:
            t.full_check(api.Engine, orc, batch, _abi.make_params(preset, seed=seed, min_depth=3))
        except AssertionError as e:
            bad += 1
            print("MISMATCH seed", seed, preset, str(e)[:200], flush=True)
print("seeds %d..%d x 4 presets: %d mismatches" % (a, b, bad))
