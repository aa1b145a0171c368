"""Generates tests/golden/demo_pseudo_ref.fa from tests/golden/demo.bam.

demo.bam is the reference repo's demo data file (demo/demo.bam, HG002 Revio MAS-Seq,
chr20:16,729,961-16,743,217).  demo/chr20.fa is NOT in the reference checkout and the BAM has no
MD/cs tags, so the true reference bases are unrecoverable: the fixture reference is the
per-column majority base (ties A<C<G<T) over all aligned bases of the reads that pass the
hifi-masseq read filter, 'N' where uncovered.  Consequence: demo-based tests are
self-consistency parity (GPU vs oracle on identical inputs), not parity with the authors' output.

Run from the repo root:  python tests/golden/make_demo_fixture.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from longcallr_amd import bamio  # noqa: E402

here = os.path.dirname(os.path.abspath(__file__))
refs, recs = bamio.read_bam(os.path.join(here, "demo.bam"))
keep = [r for r in recs if bamio.passes_filter(r)]
rid = keep[0]["ref_id"]
(start0, length, maxcov), = bamio.discover_regions(keep, rid, refs[rid][1])
cnt = np.zeros((4, length), dtype=np.int64)
code = np.full(256, -1, dtype=np.int64)
for i, b in enumerate(b"ACGT"):
    code[b] = i
for r in keep:
    rp, qp = r["pos"] - start0, r["lead"]
    for c in r["cigar"]:
        op, ln = int(c & 15), int(c >> 4)
        if op in (0, 7, 8):
            lo, hi = max(rp, 0), min(rp + ln, length)
            if hi > lo:
                b = code[r["seq"][qp + (lo - rp): qp + (hi - rp)]]
                ok = b >= 0
                np.add.at(cnt, (b[ok], np.arange(lo, hi)[ok]), 1)
            rp += ln
            qp += ln
        elif op == 1:
            qp += ln
        elif op in (2, 3):
            rp += ln
ref = np.frombuffer(b"ACGT", dtype=np.uint8)[np.argmax(cnt, axis=0)].copy()
ref[cnt.sum(axis=0) == 0] = ord("N")
with open(os.path.join(here, "demo_pseudo_ref.fa"), "w") as f:
    f.write(">%s:%d-%d pseudo-reference (majority base of demo.bam pileup; NOT GRCh38)\n"
            % (refs[rid][0], start0 + 1, start0 + length))
    s = ref.tobytes().decode()
    for i in range(0, len(s), 80):
        f.write(s[i:i + 80] + "\n")
print(refs[rid][0], start0, length, maxcov, {c: int((ref == ord(c)).sum()) for c in "ACGTN"})
