"""K0 / K1 development check (gpurun): planes against the oracle pool on demo.bam + small synthetic batches of every
preset, then HIP-event times of the pileup stage on C3 (and the C4 share with --c4)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from longcallr_amd import _abi, api, synth
from oracle import orc
import helpers, bench
orc.build()
ok = True
cases = [("demo", helpers.demo_batch(), _abi.make_params("hifi-masseq"))]
for prof, seed in (("ont-cdna", 11), ("masseq", 13), ("ont-drna", 14)):
    cases.append((prof, synth.make_batch(prof, n_genes=5, gene_len=9000, depth=35, seed=seed), _abi.make_params(synth.preset_for(prof), seed=seed)))
cases.append(("c5-small", synth.make_island("ont-drna-c5", n_loci=4, locus_len=25000, depth=200, seed=3), _abi.make_params("ont-drna", seed=11)))
if "--time-only" in sys.argv:
    cases = []
for name, b, p in cases:
    O = orc.Batch(b, p, upto="pileup")
    E = api.Engine(0, p)
    E.load_batch(b).fill_data_into_freq_vec()
    pl, ref = E.columns(), O.planes()
    bad = [n for k, n in enumerate(_abi.PLANE_NAMES) if not np.array_equal(pl[k], ref[k])]
    print(name, "planes", "OK" if not bad else "DIFFER: %s" % bad, flush=True)
    if bad:
        ok = False
        k = _abi.PLANE_NAMES.index(bad[0])
        d = np.flatnonzero(pl[k] != ref[k])
        print("   first diffs at", d[:10], "gpu", pl[k][d[:10]], "ref", ref[k][d[:10]], "n", d.size, "sum gpu/ref", int(pl[k].sum()), int(ref[k].sum()))
    E.close()
import torch
for wl in (["c3"] + (["c4"] if "--c4" in sys.argv else [])):
    if os.environ.get("K0_SUB"):   # experiment: substitution rate of the generator (in-process, a quarter of the genes)
        synth.PROFILES[bench.WORKLOADS[wl][0]]["sub"] = float(os.environ["K0_SUB"])
        b = bench.build_shard(wl, 1, 0, genes=bench.WORKLOADS[wl][1] // 4, workers=1)[0]
    else:
        b = bench.build_workload(wl)
    p = _abi.make_params(synth.preset_for(bench.WORKLOADS[wl][0]), seed=2025)
    dv = bench.to_device(b, torch, torch.device("cuda", 0))
    E = api.Engine(0, p, timing=True)
    ts = []
    for i in range(30):
        E.load_batch(dv); E.fill_data_into_freq_vec(); E.sync()
        ts.append((E.kernel_ms(_abi.K_SPANS), E.kernel_ms(_abi.K_PILEUP)))
    ts = np.array(ts[10:])
    sb = E.pileup_stage_bytes()
    print(wl, "k0 %.4f ms  k1 %.4f ms  stage %.4f ms  frac %.4f" % (ts[:, 0].mean(), ts[:, 1].mean(), ts.sum(axis=1).mean(), sb / (ts.sum(axis=1).mean() * 1e-3) / 8e12), flush=True)
    E.close()
print("ALL OK" if ok else "FAILED")
