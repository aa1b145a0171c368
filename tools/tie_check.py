"""HIP path vs the oracle's decision modes, region by region (GPU box): which regions equal ORC_MODE_TIE (mask given), ORC_MODE_F64,
ORC_MODE_EXACT; the tie census of both sides.  usage: tie_check.py c3|c4|small [mask]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from longcallr_amd import _abi, api, synth, vcf
from oracle import orc
import bench

which = sys.argv[1] if len(sys.argv) > 1 else "small"
mask = int(sys.argv[2]) if len(sys.argv) > 2 else 15
if which == "small":
    b, preset = synth.make_batch("ont-cdna", n_genes=24, gene_len=12000, depth=40, seed=5), "ont-cdna"
elif which == "drna":
    b, preset = synth.make_batch("ont-drna", n_genes=24, gene_len=20000, depth=45, seed=31), "ont-drna"
elif which == "island":
    b, preset = synth.make_island("ont-drna-c5", n_loci=4, locus_len=25000, depth=200, seed=3), "ont-drna"
else:
    b, preset = bench.build_workload(which), ("ont-cdna" if which == "c3" else "hifi-masseq")
p = _abi.make_params(preset, seed=2025)
E = api.Engine(0, p)
E.load_batch(b).run_all()
t = time.time(); E.load_batch(b).run_all(); E.phase_result(); print("HIP step %.2f ms" % ((time.time() - t) * 1e3))
c, off = E.candidates(); fm = E.fragmat(); pr = E.phase_result()
print("HIP tie census", E.tie_census())
S = np.diff(off)
modes = {"TIE(%d)" % mask: dict(mode=orc.MODE_TIE, tie_mask=mask), "F64": dict(mode=orc.MODE_F64), "EXACT": dict(mode=orc.MODE_EXACT)}
for name, kw in modes.items():
    O = orc.Batch(b, p, keep_planes=False, fast=1, **kw)
    oc, op, ot = O.cands(), O.phase_result(), O.vcf_texts()
    bad = []
    for g in range(b.n_regions):
        r0, r1 = fm["row_region_off"][g], fm["row_region_off"][g + 1]
        same = all(np.array_equal(pr[f][r0:r1], op[f][r0:r1]) for f in ("haplotag", "assignment", "phase_set"))
        gc, rc = c[off[g]:off[g + 1]], oc[off[g]:off[g + 1]]
        same = same and all(np.array_equal(gc[f], rc[f]) for f in ("pos", "variant_type", "genotype", "haplotype", "flags", "phase_set"))
        same = same and vcf.format_records(gc, "chrS", p.min_phase_score) == ot[g]
        if not same:
            bad.append(g)
    cen = O.tie_census()
    en = S <= p.max_enum_snps
    print("%-8s regions differing from HIP: %d %s (enum %d, chain %d) | oracle census enum regions: sigma ties w/ het %d flips %d delta ties %d tie-only steps %d best-pick f64 %d | chain regions: sigma ties w/ het %d flips %d equal-objective compares %d (f64 greater %d)"
          % (name, len(bad), bad[:12], sum(1 for g in bad if en[g]), sum(1 for g in bad if not en[g]), cen[en, 8].sum(), cen[en, 4].sum(), cen[en, 1].sum(), cen[en, 2].sum(), cen[en, 7].sum(), cen[~en, 8].sum(), cen[~en, 4].sum(), cen[~en, 3].sum(), cen[~en, 7].sum()))
    O.close()
