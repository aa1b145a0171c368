#!/usr/bin/env python
"""BAM + reference FASTA (+ .fai) -> phased VCF (and phased BAM) on one MI355X: longcallr_amd.pipeline.run.

  python tools/run_pipeline.py -b reads.bam -f ref.fa -o out.vcf [--out-bam phased.bam] [-p hifi-masseq] [-c chr20,chr21]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longcallr_amd import pipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-b", "--bam", required=True)
    ap.add_argument("-f", "--ref", required=True, help="FASTA with a .fai next to it")
    ap.add_argument("-o", "--out-vcf", required=True)
    ap.add_argument("--out-bam")
    ap.add_argument("-p", "--preset", default="hifi-masseq", choices=["hifi-isoseq", "hifi-masseq", "ont-cdna", "ont-drna"])
    ap.add_argument("-c", "--contigs", help="comma-separated subset")
    ap.add_argument("-t", "--threads", type=int, default=0, help="host threads of the BAM decoder / writer (0 = all)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--seed", type=int, default=2025)
    a = ap.parse_args()
    st = pipeline.run(a.bam, a.ref, a.out_vcf, a.out_bam, preset=a.preset, contigs=a.contigs.split(",") if a.contigs else None,
                      device=a.device, threads=a.threads, seed=a.seed)
    print(json.dumps(st))


if __name__ == "__main__":
    main()
