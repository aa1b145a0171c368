"""k4_enum_bits against the oracle with other enumeration thresholds (S up to 14: two decision passes, 2^14 restarts; S <= 2: groups of
fewer than eight restarts) and against k4_enum_reg (results + census)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import test_gpu_parity as t
from oracle import orc
from longcallr_amd import _abi, api, synth
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    prof = ("ont-cdna", "masseq", "ont-drna")[seed % 3]
    mx = (2, 5, 12, 14, 10)[seed % 5]
    b = synth.make_batch(prof, n_genes=4, gene_len=(9000, 16000)[seed & 1], depth=(30, 45)[(seed >> 1) & 1], seed=100 + seed)
    p = _abi.make_params(synth.preset_for(prof), seed=seed, max_enum_snps=mx)
    try:
        t.full_check(api.Engine, orc, b, p)
        got = {}
        for v in ("1", "0"):
            os.environ["LCR_ENUM_BITS"] = v
            E = api.Engine(0, p); E.load_batch(b).run_all(); got[v] = (t._result_bytes(E), dict(E.tie_census())); E.close()
        os.environ.pop("LCR_ENUM_BITS")
        assert got["1"] == got["0"], "kernels differ: %s %s" % (got["1"][1], got["0"][1])
    except AssertionError as e:
        bad += 1; print("MISMATCH seed", seed, prof, mx, str(e)[:300], flush=True)
print("enum sweep: %d mismatches" % bad)
