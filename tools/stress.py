import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from longcallr_amd import _abi, api, synth
t0=time.time()
b = synth.make_batch("ont-drna", n_genes=1, gene_len=int(sys.argv[1]), depth=float(sys.argv[2]), seed=77)
print("batch: reads", b.n_reads, "bases", b.bases.size, "regions", b.n_regions, "gen s", round(time.time()-t0,1), flush=True)
p = _abi.make_params("ont-drna", seed=3)
E = api.Engine(0, p, timing=True)
for it in range(2):
    t=time.perf_counter(); E.load_batch(b); E.sync(); t1=time.perf_counter()
    E.fill_data_into_freq_vec(); E.sync(); t2=time.perf_counter()
    E.get_candidate_snps(); E.sync(); t3=time.perf_counter()
    E.get_fragments(); E.sync(); t4=time.perf_counter()
    E.phase(); E.sync(); t5=time.perf_counter()
    c, off = E.candidates(); fm = E.fragmat(); pr = E.phase_result()
    print("iter", it, "ms: load %.1f pileup %.1f cand %.1f frag %.1f phase %.1f | cands %d rows %d nnz %d het-phased %d" % (
        (t1-t)*1e3,(t2-t1)*1e3,(t3-t2)*1e3,(t4-t3)*1e3,(t5-t4)*1e3, len(c), len(fm["row_read"]), len(fm["col"]), int((c["phase_set"]>0).sum())), flush=True)
