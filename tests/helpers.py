"""Shared test helpers: demo fixture loader and small hand-built read batches."""
import functools
import os

import numpy as np

from longcallr_amd import _abi, bamio

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_pseudo_ref():
    lines = open(os.path.join(GOLDEN, "demo_pseudo_ref.fa")).read().split("\n")
    return np.frombuffer("".join(lines[1:]).encode(), dtype=np.uint8).copy()


@functools.lru_cache(maxsize=1)
def demo_batch():
    """demo.bam (hifi-masseq read filter) as a one-region ReadBatch over the pseudo-reference."""
    refs, recs = bamio.read_bam(os.path.join(GOLDEN, "demo.bam"))
    keep = [r for r in recs if bamio.passes_filter(r, **_abi.READ_FILTER)]
    rid = keep[0]["ref_id"]
    (start0, length, _), = bamio.discover_regions(keep, rid, refs[rid][1])
    ref = load_pseudo_ref()
    assert ref.size == length
    return bamio.build_batch(keep, [(start0, length)], [ref])


CIG = {c: i for i, c in enumerate("MIDNSHP=X")}


def mk_batch(reads, regions):  # noqa: C901
    """reads: list of dict(pos, seq(str), qual(list|int), cigar(str like '5M2I3M'), rev=0, ts=0,
    region=idx).  regions: list of (start0, ref_str).  Reads must be listed grouped by region."""
    import re
    cols = {f: [] for f in ["pos", "seq_len", "lead_clip", "trail_clip", "flags", "n_cig"]}
    bases, quals, cigs, seq_off, cig_off = [], [], [], [], []
    so = co = 0
    read_begin = [0] * (len(regions) + 1)
    for r in reads:
        ops = [(int(n), CIG[c]) for n, c in re.findall(r"(\d+)([MIDNSHP=X])", r["cigar"])]
        seq = r["seq"].encode()
        q = r.get("qual", 30)
        q = [q] * len(seq) if isinstance(q, int) else list(q)
        assert len(q) == len(seq)
        lead = ops[0][0] if ops[0][1] == 4 else (ops[1][0] if ops[0][1] == 5 and len(ops) > 1 and ops[1][1] == 4 else 0)
        trail = ops[-1][0] if ops[-1][1] == 4 else (ops[-2][0] if ops[-1][1] == 5 and len(ops) > 1 and ops[-2][1] == 4 else 0)
        cols["pos"].append(r["pos"]); cols["seq_len"].append(len(seq))
        cols["lead_clip"].append(lead); cols["trail_clip"].append(trail)
        cols["flags"].append((1 if r.get("rev") else 0) | (r.get("ts", 0) << 1))
        cols["n_cig"].append(len(ops))
        seq_off.append(so); cig_off.append(co)
        bases.append(np.frombuffer(seq, dtype=np.uint8)); quals.append(np.array(q, dtype=np.uint8))
        cigs.append(np.array([(n << 4) | o for n, o in ops], dtype=np.uint32))
        so += len(seq); co += len(ops)
        read_begin[r.get("region", 0) + 1] += 1
    read_begin = np.cumsum(read_begin)
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return _abi.ReadBatch(
        seq_off=np.array(seq_off, dtype=np.uint64), cig_off=np.array(cig_off, dtype=np.uint64),
        bases=cat(bases, np.uint8), quals=cat(quals, np.uint8), cigar=cat(cigs, np.uint32),
        start0=[s for s, _ in regions], len=[len(x) for _, x in regions], read_begin=read_begin,
        ref=np.frombuffer("".join(x for _, x in regions).encode(), dtype=np.uint8), **cols)


def two_haplotype_batch(n_snps=11, groups=1, n_reads=60, seed=0, edit_sites=(), edit_frac=0.6, qual=30):
    """Hand-checkable phasing instance: error-free reads of two haplotypes over `groups` stretches of 2 kb that no read
    connects, n_snps het sites per stretch (alt on haplotype A only), every read covers its whole stretch.  edit_sites:
    offsets (inside stretch 0) of A>G sites whose G sits on `edit_frac` of haplotype A's reads -- with ts '+' reads
    that is the RNA-editing class of candidate.rs:379-407."""
    rng = np.random.default_rng(seed)
    span, gap = 2000, 3000
    L = groups * span + (groups - 1) * gap
    ref = rng.choice(list("ACGT"), size=L)
    alt_of = {"A": "C", "C": "A", "G": "T", "T": "G"}
    sites = []
    for gi in range(groups):
        o = gi * (span + gap)
        sites.append([o + 100 + (span - 200) * k // (n_snps - 1) for k in range(n_snps)])
    for x in edit_sites:
        ref[x] = "A"
    ref = "".join(ref)
    reads = []
    for gi in range(groups):
        o = gi * (span + gap)
        for k in range(n_reads):
            hap_a = k % 2 == 0
            s = list(ref[o:o + span])
            for x in sites[gi]:
                if hap_a:
                    s[x - o] = alt_of[ref[x]]
            if gi == 0:
                for x in edit_sites:
                    if hap_a and rng.random() < edit_frac:
                        s[x] = "G"
            reads.append(dict(pos=5000 + o, seq="".join(s), qual=qual, cigar="%dM" % span, rev=k // 2 % 2, ts=1 + (k // 2 % 2), region=0))
    reads.sort(key=lambda r: r["pos"])
    return mk_batch(reads, [(5000, ref)]), [[5000 + x for x in g] for g in sites]
