#!/usr/bin/env python
"""BASELINE configs[4] (C5): ONE coverage island, 1 Mb x 500x ONT-dRNA, ~5 000 candidate sites — the single-region stress
of the phase stage (all CUs on one region, k4_grid.hip).  Prints sizes, per-call wall times and, with --repeat, checks
that repeated runs are bit-identical."""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loci", type=int, default=40)
    ap.add_argument("--locus-len", type=int, default=25000)
    ap.add_argument("--depth", type=float, default=500.0)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--repeat", type=int, default=2)
    a = ap.parse_args()
    from longcallr_amd import _abi, api, synth
    t0 = time.perf_counter()
    b = synth.make_island("ont-drna-c5", n_loci=a.loci, locus_len=a.locus_len, depth=a.depth, seed=a.seed)
    gen_s = time.perf_counter() - t0
    params = _abi.make_params("ont-drna", seed=a.seed)
    E = api.Engine(0, params)
    out = dict(workload="C5: %d loci x %d bp as one island, %.0fx" % (a.loci, a.locus_len, a.depth), generate_s=gen_s,
               columns=int(b.len[0]), reads=b.n_reads, aligned_bases=int(b.bases.size), runs=[])
    digest = None
    for it in range(a.repeat):
        ms = {}
        def timed(name, fn):
            ts = time.perf_counter(); fn(); E.sync(); ms[name] = (time.perf_counter() - ts) * 1e3
        timed("lcr_load_batch", lambda: E.load_batch(b))
        timed("lcr_pileup", E.fill_data_into_freq_vec)
        timed("lcr_candidates", E.get_candidate_snps)
        timed("lcr_fragments", E.get_fragments)
        timed("lcr_phase", E.phase)
        c, off = E.candidates()
        fm, pr = E.fragmat(), E.phase_result()
        h = hashlib.sha1(c.tobytes() + pr["haplotag"].tobytes() + pr["assignment"].tobytes() + pr["phase_set"].tobytes()).hexdigest()
        if digest is None:
            digest = h
        assert h == digest, "repeated runs differ"
        out["runs"].append(ms)
    blocks = E.ld_blocks(0)
    fp = (c["flags"] & _abi.F_FOR_PHASING) != 0
    out.update(candidates=int(c.size), for_phasing=int(fp.sum()), dense=int(((c["flags"] & _abi.F_DENSE) != 0).sum()),
               rows=int(fm["row_read"].size), nnz=int(fm["col"].size), phasing_rows=int(fm["row_for_phasing"].sum()),
               ld_blocks=len(blocks), largest_block=max([len(x) for x in blocks] + [0]),
               assigned_reads=int((pr["assignment"] != 0).sum()), phased_het=int(((c["phase_set"] != 0)).sum()),
               objective=float(pr["objective"][0]), digest=digest,
               cross_optimize_calls=1 + 2 * (int(c.size) // 4 + 1) if c.size > params.max_enum_snps else None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
