"""BASELINE configs[4] (C5) at full size: the HIP path against the oracle with indexed gathers + threaded Jacobi steps
(orc_set_fast) -- planes, candidates, fragment matrix, sigma / delta / eta, objective, phase sets, VCF text of the one
1 Mb region (phase.rs:1123-1233: 2 345 cross_optimize calls).  usage: c5_oracle.py [threads] [mode]"""
import sys, time, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from longcallr_amd import _abi, api, synth, vcf
from oracle import orc

threads = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = int(sys.argv[2]) if len(sys.argv) > 2 else orc.MODE_TIE
t = time.time()
b = synth.make_island("ont-drna-c5", n_loci=40, locus_len=25000, depth=500, seed=5)
p = _abi.make_params("ont-drna", seed=5)
print("built %.1f s" % (time.time() - t), flush=True)
E = api.Engine(0, p)
t = time.time(); E.load_batch(b).run_all(); E.phase_result(); print("HIP %.2f s (first call)" % (time.time() - t), flush=True)
c, off = E.candidates(); pr = E.phase_result(); fm = E.fragmat()
print("HIP census", E.tie_census(), flush=True)
R = orc.Region(b, 0, p).set_fast(threads).set_tie_mask(orc.TIE_MASK_LIBLCR)
t = time.time(); R.pileup().candidates().fragments(); t16 = time.time() - t
t = time.time(); R.phase(mode); tp = time.time() - t
t = time.time(); R.post_phase(); tpp = time.time() - t
print("oracle (%d threads): P1-P6 %.1f s, phase %.1f s, post-phase %.1f s" % (threads, t16, tp, tpp), R.stats(), R.tie_census().tolist(), flush=True)
op, oc = R.phase_result(), R.cands()
ok = {f: bool(np.array_equal(pr[f], op[f])) for f in ("haplotag", "assignment", "phase_set")}
ok["objective"] = bool(pr["objective"][0] == op["objective"])
for f in ("pos", "variant_type", "genotype", "haplotype", "flags", "phase_set"):
    ok["cand." + f] = bool(np.array_equal(c[f], oc[f]))
ok["phase_score<=1e-4"] = bool(np.all(np.abs(c["phase_score"] - oc["phase_score"]) <= 1e-4))
ok["vcf"] = vcf.format_records(c, "chrS", p.min_phase_score) == R.vcf_text("chrS")
ok["ld_blocks"] = E.ld_blocks(0) == R.ld_blocks()
print(ok, "objective", pr["objective"][0], op["objective"], flush=True)
print("ALL EQUAL" if all(ok.values()) else "DIFFERENT")
