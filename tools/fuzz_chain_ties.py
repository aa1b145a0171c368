"""Sweep of chain regions BUILT TO MEET TIES of classes 2 / 4 (a delta / eta choice with two equal maxima, a step of tie changes only) against
the oracle with those classes resolved (ORC_MODE_TIE, orc.TIE_MASK_LIBLCR: chain mask 15): error-free reads of two haplotypes over n het sites,
equal base qualities, and at a few sites the allele flipped in a subset of the reads (a fixed half of each haplotype, or a random subset), so
that column sums cancel exactly.  Every seed: full_check (planes, candidates, fragments, phase result, post-phase) + the census -- nothing
unresolved, and the decided ties are the oracle's count.  usage: fuzz_chain_ties.py a b"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers
import test_gpu_parity as t
from oracle import orc
from longcallr_amd import _abi, api
a, b = int(sys.argv[1]), int(sys.argv[2])
alt_of = {ord("A"): ord("C"), ord("C"): ord("A"), ord("G"): ord("T"), ord("T"): ord("G")}
bad = n_chain = n_met2 = n_met4 = ties2 = ties4 = 0
for seed in range(a, b):
    rng = np.random.default_rng(seed)
    n_snps = int(rng.integers(11, 49)); n_reads = int(rng.choice([8, 12, 16, 24, 40])); n_flip = int(rng.integers(1, 7))
    batch, sites = helpers.two_haplotype_batch(n_snps=n_snps, n_reads=n_reads, seed=seed)
    bases = batch.bases.copy()
    for j in rng.choice(n_snps, size=min(n_flip, n_snps), replace=False):
        x = sites[0][j] - 5000
        half = rng.random() < 0.5
        for k in range(batch.n_reads):
            if ((k // 2) % 2 == 0) if half else (rng.random() < 0.5):
                o = int(batch.seq_off[k]) + x
                bases[o] = alt_of[int(bases[o])]
    b2 = _abi.ReadBatch(**{f: getattr(batch, f) for f in batch.FIELDS if f != "bases"}, bases=bases, start0=batch.start0, len=batch.len,
                        read_begin=batch.read_begin, ref=batch.ref)
    p = _abi.make_params(("hifi-masseq", "ont-cdna")[seed & 1], seed=seed)
    try:
        c = t.full_check(api.Engine, orc, b2, p)
        if len(c) <= p.max_enum_snps:
            continue
        n_chain += 1
        oc = t.oracle_all(orc, b2, p)[0].tie_census()
        E = api.Engine(0, p)
        E.load_batch(b2).run_all()
        hc = E.tie_census()
        E.close()
        assert hc["delta_unresolved"] == 0 and hc["step_unresolved"] == 0 and hc["delta_step_f64"] == int(oc[1]) + int(oc[2]), (hc, oc.tolist())
        n_met2 += int(oc[1] > 0); n_met4 += int(oc[2] > 0); ties2 += int(oc[1]); ties4 += int(oc[2])
    except AssertionError as e:
        bad += 1
        print("MISMATCH seed", seed, n_snps, n_reads, n_flip, str(e)[:300], flush=True)
print("seeds %d..%d: %d chain regions, %d met class-2 ties (%d ties), %d met tie-only steps (%d steps), %d mismatches" % (a, b, n_chain, n_met2, ties2, n_met4, ties4, bad))
