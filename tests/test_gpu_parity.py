"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the CPU oracle on the
same inputs.  Bar: bit-exact for every integer / byte / index output (count planes, candidate set,
alleles, GT class, flags, DP, fragment matrix, sigma/delta/eta, assignments, phase sets, int-cast
QUAL/GQ); |rel| <= 1e-9 for f64 likelihoods (order-free histogram sum vs the reference's running
sum) and <= 1e-4 absolute for phase objective / PQ (fixed-point objective, BASELINE north_star).
Decision arithmetic of the optimiser (round 4): the oracle runs in ORC_MODE_TIE with the tie classes liblcr
resolves (orc.TIE_MASK_LIBLCR) -- exact fixed-point sums, exact ties by the reference-order f64 scores -- and once
more in ORC_MODE_F64 (reference-order f64 everywhere): wherever the census shows no tie of an unresolved class the
three must agree (check_f64_mode)."""
import numpy as np
import pytest

import helpers
from longcallr_amd import _abi, api, synth, vcf

pytestmark = pytest.mark.gpu

INT_FIELDS = ["pos", "region", "ref_base", "allele1", "allele2", "n_alt", "cnt1", "cnt2", "depth",
              "variant_type", "genotype", "haplotype", "flags", "phase_set"]


def as_i32(x):
    x = np.asarray(x, dtype=np.float64)
    return np.where(np.isnan(x), 0, np.clip(np.nan_to_num(x, posinf=2147483647.0, neginf=-2147483648.0),
                                            -2147483648.0, 2147483647.0)).astype(np.int64)


def close(a, b, rel):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    same_inf = np.isinf(a) & np.isinf(b) & (np.sign(a) == np.sign(b))
    with np.errstate(invalid="ignore"):
        return bool(np.all(same_inf | (np.abs(a - b) <= rel * np.maximum(1.0, np.abs(b)))))


def oracle_all(orc, batch, params, upto="post"):
    regs = []
    for g in range(batch.n_regions):
        R = orc.Region(batch, g, params).set_fast(1).pileup()
        if upto != "pileup":
            R.candidates()
        if upto in ("frag", "post"):
            R.fragments()
            R.fm_snapshot = R.fragmat()  # for_phasing of rows changes in the post-phase rescue steps
        if upto == "post":
            R.set_tie_mask(orc.TIE_MASK_LIBLCR if ORACLE_TIE_MASK[0] is None else ORACLE_TIE_MASK[0]).phase(orc.MODE_TIE).post_phase()
        regs.append(R)
    return regs


F64_CHECKED = {"regions": 0, "no_unresolved_tie": 0}
ORACLE_TIE_MASK = {0: None}     # None: orc.TIE_MASK_LIBLCR (a test of a fallback kernel or of the tie_arith switch sets its own)


def unresolved_ties(census_tie, census_f64, is_chain):
    """an oracle region's census (orc_get_tie_census) of the classes liblcr leaves to `a tie changes nothing` -- in the CHAIN branch
    only (the enumeration branch resolves all four classes since round 5): delta / eta ties at the maximum [1], steps whose only
    changes were tie changes [2] (counted by ORC_MODE_TIE), a later configuration of equal objective whose f64 sum is greater [7]"""
    return int(census_f64[1]) + int(census_tie[2]) + int(census_f64[7]) if is_chain else 0


def check_f64_mode(orc, batch, params, regs, chrom):
    """ORC_MODE_F64 -- the reference's f64 ratio scores / sums in the reference's order at EVERY decision -- reaches the
    phasing of ORC_MODE_TIE (what liblcr computes) in every region whose census shows no tie of a class liblcr does not
    resolve; how many regions that is gets recorded (all of them on the BASELINE configs)."""
    if ORACLE_TIE_MASK[0] not in (None, orc.TIE_MASK_LIBLCR):
        return   # (a test of a fallback kernel / of the tie_arith switch: the claim is about the classes liblcr resolves by default)
    for g, R in enumerate(regs):
        A = orc.Region(batch, g, params).set_fast(1).run_all(orc.MODE_F64)
        F64_CHECKED["regions"] += 1
        is_chain = len(R.cands()) > params.max_enum_snps
        if unresolved_ties(R.tie_census(), A.tie_census(), is_chain):
            continue
        F64_CHECKED["no_unresolved_tie"] += 1
        pa, pb = A.phase_result(), R.phase_result()
        for f in ("haplotag", "assignment", "phase_set"):
            assert np.array_equal(pa[f], pb[f]), "F64 vs TIE %s region %d" % (f, g)
        assert abs(pa["objective"] - pb["objective"]) < 1e-6
        ca, cb = A.cands(), R.cands()
        for f in INT_FIELDS:
            assert np.array_equal(ca[f], cb[f]), "F64 vs TIE cand.%s region %d" % (f, g)
        assert A.vcf_text(chrom) == R.vcf_text(chrom)


def check_pileup(E, regs, batch):
    pl = E.columns()
    for g, R in enumerate(regs):
        o, n = int(batch.col_off[g]), int(batch.len[g])
        ref = R.planes()
        for k, name in enumerate(_abi.PLANE_NAMES):
            assert np.array_equal(pl[k, o:o + n], ref[k]), "plane %s region %d" % (name, g)


def check_cands(E, regs, phased):
    c, off = E.candidates()
    for g, R in enumerate(regs):
        rc = R.cands()
        gc = c[off[g]:off[g + 1]]
        assert len(gc) == len(rc), "candidate count region %d: %d vs %d" % (g, len(gc), len(rc))
        for f in INT_FIELDS:
            assert np.array_equal(gc[f], rc[f]), "cand.%s region %d" % (f, g)
        assert np.array_equal(gc["af1"], rc["af1"]) and np.array_equal(gc["af2"], rc["af2"])
        # oracle loglik is the reference-order running sum; GPU is histogram x LUT
        assert close(gc["loglik"], rc["loglik"], 1e-9) and close(gc["gt_prob"], rc["gt_prob"], 1e-9)
        assert close(gc["qual"], rc["qual"], 1e-9) and close(gc["gq"], rc["gq"], 1e-9)
        assert np.array_equal(as_i32(gc["qual"]), as_i32(rc["qual"])), "QUAL as i32"
        assert np.array_equal(as_i32(gc["gq"]), as_i32(rc["gq"])), "GQ as i32"
        if phased:
            assert np.all(np.abs(gc["phase_score"] - rc["phase_score"]) <= 1e-4)
    return c, off


def check_fragmat(E, regs):
    fm = E.fragmat()
    for g, R in enumerate(regs):
        rf = R.fm_snapshot
        r0, r1 = fm["row_region_off"][g], fm["row_region_off"][g + 1]
        assert r1 - r0 == len(rf["row_read"]), "rows region %d" % g
        e0, e1 = fm["row_ptr"][r0], fm["row_ptr"][r1]
        assert np.array_equal(fm["row_ptr"][r0:r1 + 1] - e0, rf["row_ptr"])
        base = E.candidates()[1][g]
        assert np.array_equal(fm["col"][e0:e1] - base, rf["col"]) and np.array_equal(fm["val"][e0:e1], rf["val"])
        assert np.array_equal(fm["row_links"][r0:r1], rf["row_links"])
        assert np.array_equal(fm["row_for_phasing"][r0:r1], rf["row_for_phasing"])
        assert np.array_equal(fm["row_read"][r0:r1], rf["row_read"])
    return fm


def check_phase(E, regs, fm):
    pr = E.phase_result()
    for g, R in enumerate(regs):
        rp = R.phase_result()
        r0, r1 = fm["row_region_off"][g], fm["row_region_off"][g + 1]
        assert np.array_equal(pr["haplotag"][r0:r1], rp["haplotag"]), "sigma region %d" % g
        assert np.array_equal(pr["assignment"][r0:r1], rp["assignment"]), "assignment region %d" % g
        assert np.array_equal(pr["phase_set"][r0:r1], rp["phase_set"]), "read PS region %d" % g
        assert abs(pr["objective"][g] - rp["objective"]) <= 1e-4, "objective region %d" % g
        assert pr["objective"][g] == rp["objective"], "fixed-point objective must match exactly"


def full_check(engine_cls, orc, batch, params, chrom="chrS"):
    regs = oracle_all(orc, batch, params)
    check_f64_mode(orc, batch, params, regs, chrom)
    E = engine_cls(0, params)
    E.load_batch(batch).fill_data_into_freq_vec()
    check_pileup(E, regs, batch)
    E.get_candidate_snps().get_fragments()
    fm = check_fragmat(E, regs)
    E.phase()
    c, off = check_cands(E, regs, phased=True)
    check_phase(E, regs, fm)
    for g, R in enumerate(regs):
        assert vcf.format_records(c[off[g]:off[g + 1]], chrom, params.min_phase_score) == R.vcf_text(chrom)
        if off[g + 1] - off[g] > params.max_enum_snps:   # chain region: LD blocks in the reference's block / node order
            assert E.ld_blocks(g) == R.ld_blocks(), "LD blocks region %d" % g
    E.close()
    return c


def test_demo_bam_full_pipeline(engine_cls, orc):
    """configs[0]/[1]: demo.bam, hifi-masseq preset, pseudo-reference (self-consistency parity)."""
    c = full_check(engine_cls, orc, helpers.demo_batch(), _abi.make_params("hifi-masseq"), "chr20")
    assert len(c) == 19


@pytest.mark.parametrize("profile,seed", [("ont-cdna", 11), ("ont-cdna", 12), ("masseq", 13), ("ont-drna", 14)])
def test_synthetic_multi_region(engine_cls, orc, profile, seed):
    b = synth.make_batch(profile, n_genes=5, gene_len=9000, depth=35, seed=seed)
    full_check(engine_cls, orc, b, _abi.make_params(synth.preset_for(profile), seed=seed))


def test_chain_path_many_snps(engine_cls, orc):
    """S > max_enum_snps: LD blocks, block-flip pass and perturbation rounds (phase.rs:1123-1233)."""
    b = synth.make_batch("ont-drna", n_genes=2, gene_len=40000, depth=50, seed=21)
    c = full_check(engine_cls, orc, b, _abi.make_params("ont-drna", seed=5))
    assert max(np.bincount(c["region"])) > 10


@pytest.mark.parametrize("grid_min", ["0", "100000000"])
def test_chain_scopes_agree_with_the_oracle(engine_cls, orc, monkeypatch, grid_min):
    """The chain regions' kernel in both scopes -- all CUs on one region behind grid barriers (LCR_GRID_MIN_ENTRIES=0)
    and one workgroup per region -- gives the oracle's LD blocks, sigma / delta / eta, objective and VCF text."""
    monkeypatch.setenv("LCR_GRID_MIN_ENTRIES", grid_min)
    b = synth.make_batch("ont-drna", n_genes=3, gene_len=30000, depth=60, seed=23)
    c = full_check(engine_cls, orc, b, _abi.make_params("ont-drna", seed=7))
    assert max(np.bincount(c["region"])) > 10
    b = synth.make_batch("ont-cdna", n_genes=2, gene_len=40000, depth=80, seed=24)
    full_check(engine_cls, orc, b, _abi.make_params("ont-cdna", seed=8))


def test_min_linkers_above_one(engine_cls, orc):
    """min_linkers = 2: reads with a single phase site are not phasing rows, but their entries still count in the
    allele-pair table the LD blocks come from (fragment.rs:208-240 runs before the for_phasing test)."""
    b = synth.make_batch("ont-drna", n_genes=2, gene_len=30000, depth=50, seed=25)
    full_check(engine_cls, orc, b, _abi.make_params("ont-drna", seed=3, min_linkers=2))
    full_check(engine_cls, orc, b, _abi.make_params("ont-drna", seed=3, min_linkers=3))


def test_hand_derived_phasing_instances(engine_cls, orc):
    """The known-answer instances of tests/test_oracle_np_phase.py (both oracle restatements agree on them and with the
    hand-derived answers) through the GPU: 11-SNP chain with one complete LD block, two phase sets, RNA-edit rescue."""
    b, sites = helpers.two_haplotype_batch(n_snps=11, n_reads=40)
    c = full_check(engine_cls, orc, b, _abi.make_params("hifi-masseq", seed=9))
    assert c["pos"].tolist() == sites[0] and set(c["phase_set"].tolist()) == {sites[0][0] + 1}
    b, sites = helpers.two_haplotype_batch(n_snps=6, groups=2, n_reads=40)
    c = full_check(engine_cls, orc, b, _abi.make_params("hifi-masseq", seed=4))
    assert c["phase_set"].tolist() == [sites[0][0] + 1] * 6 + [sites[1][0] + 1] * 6
    for mps in (8.0, 60.0):
        b, sites = helpers.two_haplotype_batch(n_snps=5, n_reads=60, edit_sites=(777,), edit_frac=0.95, seed=2)
        b.flags[:] = 0 | (1 << 1)
        c = full_check(engine_cls, orc, b, _abi.make_params("hifi-masseq", seed=4, min_phase_score=mps))
        e = c[c["pos"] == 5000 + 777]
        assert len(e) == 1 and bool(e["flags"][0] & _abi.F_FOR_PHASING) == (mps == 8.0)
    assert F64_CHECKED["no_unresolved_tie"] >= 4


def test_k0_cigar_lengths(engine_cls, orc):
    """HiFi reads (~9 ops), ONT-like reads (~56 ops) and demo.bam (up to 118 ops) through the op-parallel K0."""
    full_check(engine_cls, orc, helpers.demo_batch(), _abi.make_params("hifi-masseq"), "chr20")
    b = synth.make_batch("ont-cdna", n_genes=3, gene_len=9000, depth=35, seed=12)
    full_check(engine_cls, orc, b, _abi.make_params("ont-cdna", seed=12))
    b = synth.make_batch("masseq", n_genes=3, gene_len=9000, depth=35, seed=12)
    full_check(engine_cls, orc, b, _abi.make_params("hifi-masseq", seed=12))


def _with_reads(b, **over):
    kw = {f: getattr(b, f) for f in _abi.ReadBatch.FIELDS + ["start0", "len", "read_begin", "ref"]}
    kw.update(over)
    return _abi.ReadBatch(**kw)


def test_k0_op_space_layouts(engine_cls, orc):
    """The op-parallel K0 (k0_ops.hip) cuts ONE flat op space into blocks of 1 024 ops.  (i) CIGARs that do not lie back to
    back (the ABI allows any cig_off: here reversed order with gaps) are copied into a contiguous array at load time;
    (ii) host and device batches take different paths to the op-space geometry -- same planes, candidates, phasing."""
    b = synth.make_batch("ont-cdna", n_genes=3, gene_len=9000, depth=30, seed=41)
    p = _abi.make_params("ont-cdna", seed=41)
    n = b.n_cig.astype(np.int64)
    new_off = np.zeros(b.n_reads, dtype=np.int64)
    cur = 7
    for r in range(b.n_reads - 1, -1, -1):   # reads laid out back to front, 3 unused words between them
        new_off[r] = cur
        cur += int(n[r]) + 3
    cig = np.full(cur + 5, 0xFFFFFFF6, dtype=np.uint32)   # (filler: an undefined op code -- must never be decoded)
    for r in range(b.n_reads):
        cig[new_off[r]:new_off[r] + n[r]] = b.cigar[int(b.cig_off[r]):int(b.cig_off[r]) + int(n[r])]
    full_check(engine_cls, orc, _with_reads(b, cig_off=new_off.astype(np.uint64), cigar=cig), p)


def test_k0_block_borders_and_empty_reads(engine_cls, orc):
    """Reads whose ops straddle K0's op blocks (one read of ~2 600 ops spans three blocks: the carry-in of its earlier
    blocks), reads WITHOUT any op (first, in the middle, last; reference end = start, l_seq must be 0), a block with more
    reads than its LDS header table holds (hundreds of one-op reads), and a batch without a single op."""
    rng = np.random.default_rng(5)
    L = 6000
    ref = "".join(rng.choice(list("ACGT"), size=L))
    reads = []
    # one long read: 1M 1I 1M 1D ... over ~1 300 columns -> ~2 600 ops
    ops, seq, col = [], [], 100
    for i in range(650):
        ops += ["2M", "1I", "1M", "1D"]
        seq += [ref[col:col + 2], "A", ref[col + 2]]
        col += 4
    reads.append(dict(pos=100, seq="".join(seq), qual=20, cigar="".join(ops), ts=1))
    # 700 reads of one op each (a K0 block then holds > 160 reads), then ordinary reads
    for i in range(700):
        s = 200 + i
        reads.append(dict(pos=s, seq=ref[s:s + 60], qual=25, cigar="60M", rev=i % 2, ts=1 + i % 2))
    for i in range(60):
        s = 1000 + 20 * i
        sq = list(ref[s:s + 300])
        if i % 2:
            sq[150] = "A" if ref[s + 150] != "A" else "C"
        reads.append(dict(pos=s, seq="".join(sq), qual=30, cigar="100M1D200M" if i % 3 else "300M", rev=i % 2, ts=1))
    reads.sort(key=lambda r: r["pos"])
    b = helpers.mk_batch(reads, [(0, ref)])
    p = _abi.make_params("ont-cdna", seed=3, min_depth=2)
    full_check(engine_cls, orc, b, p)
    # reads without ops: index 0, one in the middle, the last one
    nr = b.n_reads
    ins = [0, nr // 2, nr]
    def insert(a, vals):
        return np.insert(a, ins, vals)
    pos = insert(b.pos, [b.pos[0], b.pos[nr // 2], b.pos[-1]])
    b2 = _with_reads(b, pos=pos, seq_len=insert(b.seq_len, 0), lead_clip=insert(b.lead_clip, 0), trail_clip=insert(b.trail_clip, 0),
                     flags=insert(b.flags, 0), seq_off=insert(b.seq_off, [0, b.seq_off[nr // 2], b.bases.size]),
                     cig_off=insert(b.cig_off, [0, b.cig_off[nr // 2], b.cigar.size]), n_cig=insert(b.n_cig, 0),
                     read_begin=[0, nr + 3])
    c2 = full_check(engine_cls, orc, b2, p)
    E = engine_cls(0, p)
    E.load_batch(b).run_all()
    assert E.candidates()[0].tobytes() == c2.tobytes()   # the empty reads change nothing
    # a batch of empty reads only: every stage runs, nothing is found; an empty read with l_seq != 0 is a CIGAR error
    b3 = _abi.ReadBatch(pos=[10, 20], seq_len=[0, 0], lead_clip=[0, 0], trail_clip=[0, 0], flags=[0, 0], seq_off=[0, 0], cig_off=[0, 0],
                        n_cig=[0, 0], bases=np.zeros(0, np.uint8), quals=np.zeros(0, np.uint8), cigar=np.zeros(0, np.uint32),
                        start0=[0], len=[L], read_begin=[0, 2], ref=np.frombuffer(ref.encode(), np.uint8))
    E.load_batch(b3).run_all()
    assert E.candidates()[0].size == 0 and not E.columns().any()
    from longcallr_amd._lib import LcrError
    b4 = _with_reads(b3, seq_len=[0, 5], bases=np.full(5, 65, np.uint8), quals=np.full(5, 30, np.uint8))
    with pytest.raises(LcrError, match="CIGAR inconsistent"):
        E.load_batch(b4).run_all()
    E.close()


def test_k0_empty_read_on_a_block_border(engine_cls, orc):
    """A read without CIGAR ops whose cig_off is exactly the end of a K0 block (1 024 ops) belongs to that block, not to the next
    one (which starts at the read that owns the op): its reference end is set and its l_seq is checked."""
    rng = np.random.default_rng(23)
    L = 2400
    ref = "".join(rng.choice(list("ACGT"), size=L))
    reads = []
    for i in range(1034):
        s = 100 + i
        sq = list(ref[s:s + 60])
        if i % 2 and s <= 700 < s + 60:
            sq[700 - s] = "A" if ref[700] != "A" else "C"
        reads.append(dict(pos=s, seq="".join(sq), qual=25, cigar="60M", rev=(i // 2) % 2, ts=1 + (i // 2) % 2))
    b = helpers.mk_batch(reads, [(0, ref)])
    assert int(b.cig_off[1024]) == 1024
    ins = [1024]
    mk = lambda seq_len: _with_reads(b, pos=np.insert(b.pos, ins, b.pos[1024]), seq_len=np.insert(b.seq_len, ins, seq_len), lead_clip=np.insert(b.lead_clip, ins, 0),
                                     trail_clip=np.insert(b.trail_clip, ins, 0), flags=np.insert(b.flags, ins, 0), seq_off=np.insert(b.seq_off, ins, b.seq_off[1024]),
                                     cig_off=np.insert(b.cig_off, ins, 1024), n_cig=np.insert(b.n_cig, ins, 0), read_begin=[0, b.n_reads + 1])
    p = _abi.make_params("ont-cdna", seed=3, min_depth=2)
    c = full_check(engine_cls, orc, mk(0), p)
    assert c.size >= 1
    from longcallr_amd._lib import LcrError
    E = engine_cls(0, p)
    with pytest.raises(LcrError, match="CIGAR inconsistent"):
        E.load_batch(mk(5)).run_all()          # five bases and no op to carry them
    E.close()


def test_k0_reads_beyond_the_tile_window(engine_cls, orc):
    """A K0 block keeps the record counters of 256 tiles (65 536 columns from its first read on) in LDS; records beyond --
    reads with introns of 100 kb and more -- take one pool allocation each.  Same planes / candidates / phasing."""
    rng = np.random.default_rng(9)
    L = 260000
    ref = "".join(rng.choice(list("ACGT"), size=L))
    reads = []
    for i in range(40):
        s = 500 + 3 * i
        a = list(ref[s:s + 400]); c = list(ref[s + 400 + 200000:s + 800 + 200000])
        if i % 2:
            a[800 - s] = "G" if ref[800] != "G" else "T"                    # het sites at fixed columns: 800 and 200 950
            c[200950 - (s + 400 + 200000)] = "G" if ref[200950] != "G" else "T"
        reads.append(dict(pos=s, seq="".join(a + c), qual=28, cigar="400M200000N400M", rev=(i // 2) % 2, ts=1 + (i // 2) % 2))
    b = helpers.mk_batch(reads, [(0, ref)])
    c = full_check(engine_cls, orc, b, _abi.make_params("ont-cdna", seed=2, min_depth=2))
    assert c.size >= 2


def test_k0_thousands_of_records_beyond_the_tile_window(engine_cls, orc):
    """3 000 reads whose 200 short ops all lie behind an intron of 100 kb: every record behind the intron is outside its
    block's LDS window of 256 tiles and takes a pool slot, a descriptor and an ENTRY of its own -- 6*10^5 entries where
    pool / 16 is 10^5 (the entry list was sized by the pool alone once: ADVICE round 3).  Same planes / candidates / phasing."""
    rng = np.random.default_rng(19)
    L = 102000
    ref = "".join(rng.choice(list("ACGT"), size=L))
    alt = lambda x: "G" if ref[x] != "G" else "T"
    tail = "3M1D" * 100                                        # 200 ops, 300 read bases over 400 reference positions
    reads = []
    for i in range(3000):
        s = 500 + i // 10
        a = list(ref[s:s + 100])
        t0 = s + 100 + 100000
        c = []
        for k in range(100):
            c += list(ref[t0 + 4 * k:t0 + 4 * k + 3])
        hap = i % 2
        if hap:
            if 0 <= 650 - s < 100:
                a[650 - s] = alt(650)                          # het sites at fixed columns: 650 (before the intron) ...
            for x in (100960, 101002):                         # ... and two behind it, on M positions of every read
                k, o = divmod(x - t0, 4)
                if 0 <= k < 100 and o < 3:
                    c[3 * k + o] = alt(x)
        reads.append(dict(pos=s, seq="".join(a + c), qual=27, cigar="100M100000N" + tail, rev=(i // 2) % 2, ts=1 + (i // 2) % 2))
    b = helpers.mk_batch(reads, [(0, ref)])
    c = full_check(engine_cls, orc, b, _abi.make_params("ont-cdna", seed=4, min_depth=2))
    assert c.size >= 2


@pytest.mark.parametrize("hook", ["LCR_ENUM_BITS=0", "LCR_ENUM_BITS=0+LCR_ENUM_FORCE_STREAM", "LCR_ENUM_BITS=0+LCR_ENUM_FORCE_STREAM=2", "LCR_ENUM_FORCE_STREAM",
                                  "LCR_ENUM_FORCE_BIG", "LCR_POST_HOST", "LCR_POST_HALF", "LCR_K3_HITS=0"])
def test_fallback_device_paths(engine_cls, orc, monkeypatch, hook):
    """The size-dependent fallbacks of the phase stage give the same results as the default kernels: the enumeration restarts one
    per wave (LCR_ENUM_BITS=0: the kernels of rounds 2-4, register-resident /
    with LDS-streamed entries (=2: in the launch of the regions with a large LDS image) / from global
    memory, post-phase epilogue on the host, the eight-wave epilogue of the chain regions (taken when a batch has more chain
    regions than the device has CUs), the fragment matrix's count pass walking the CIGARs itself instead of taking the hits
    the candidate stage's walk left (LCR_K3_HITS=0: the path of batches whose histograms came from the tiles)."""
    for h in hook.split("+"):
        name, _, value = h.partition("=")
        monkeypatch.setenv(name, value or "1")
    hook = hook.split("+")[-1].partition("=")[0]
    b = synth.make_batch("ont-drna", n_genes=3, gene_len=20000, depth=45, seed=14)
    full_check(engine_cls, orc, b, _abi.make_params("ont-drna", seed=14))
    full_check(engine_cls, orc, helpers.demo_batch(), _abi.make_params("hifi-masseq"), "chr20")
    if hook == "LCR_ENUM_FORCE_BIG":
        # a batch in which `prob > largest_prob` between two restarts of equal objective IS decided by the f64 sums (oracle
        # census: region 6, one compare): k4_enum_resolve_big has to find the same winner
        b = synth.make_batch("masseq", n_genes=12, gene_len=16000, depth=40, seed=2)
        p = _abi.make_params("hifi-masseq", seed=2025)
        regs = oracle_all(orc, b, p)
        assert sum(int(R.tie_census()[7]) for R in regs) >= 1
        full_check(engine_cls, orc, b, p)


def test_enumeration_kernels_agree(engine_cls, monkeypatch):
    """k4_enum_bits (eight restarts per wave as bit states, the default) and k4_enum_reg (one restart per wave) leave the same
    objectives and states: identical results AND an identical tie census (every f64-scored row, every flip, every repair-list
    entry is met by both), on batches with sigma ties and with tie-only steps."""
    for b, p in ((synth.make_batch("ont-cdna", n_genes=10, gene_len=16000, depth=40, seed=7), _abi.make_params("ont-cdna", seed=2025)),
                 (synth.make_batch("masseq", n_genes=12, gene_len=16000, depth=40, seed=2), _abi.make_params("hifi-masseq", seed=2025))):
        got = {}
        for v in ("1", "0"):
            monkeypatch.setenv("LCR_ENUM_BITS", v)
            E = engine_cls(0, p)
            E.load_batch(b).run_all()
            got[v] = (_result_bytes(E), dict(E.tie_census()))
            E.close()
        assert got["1"][0] == got["0"][0]
        assert got["1"][1] == got["0"][1], (got["1"][1], got["0"][1])
        assert got["1"][1]["sigma_f64"] > 0
    monkeypatch.delenv("LCR_ENUM_BITS")


@pytest.mark.parametrize("tie_arith,enum_mask,chain_mask", [("0", 0, 0), ("1", 8, 0), ("2", 9, 1), ("3", 15, 15)])
def test_tie_arithmetic_switch(engine_cls, orc, monkeypatch, tie_arith, enum_mask, chain_mask):
    """lcr_debug_set("tie_arith"): 0 = every decision on the fixed-point sums alone (ORC_MODE_TIE with no class resolved = the
    contract of rounds 1-3, ORC_MODE_EXACT), 1 = configurations of equal objective by their f64 sums, 2 (default) = also the
    sigma ties by the f64 scores, 3 (default) = also the delta / eta ties at the maximum and the verdict of tie-only steps in the
    enumeration branch (repair pass); each equals the oracle with the same classes, and the census says what was met."""
    monkeypatch.setenv("LCR_TIE_ARITH", tie_arith)
    monkeypatch.setitem(ORACLE_TIE_MASK, 0, orc.tie_mask(enum_mask, chain_mask))
    for b, p in ((synth.make_batch("ont-drna", n_genes=4, gene_len=20000, depth=45, seed=31), _abi.make_params("ont-drna", seed=31)),
                 (synth.make_batch("ont-cdna", n_genes=6, gene_len=12000, depth=40, seed=5), _abi.make_params("ont-cdna", seed=2025))):
        full_check(engine_cls, orc, b, p)
        E = engine_cls(0, p)
        E.load_batch(b).run_all()
        hc = E.tie_census()
        E.close()
        if tie_arith in ("2", "3"):
            assert hc["sigma_f64"] > 0 and hc["sigma_unresolved"] == 0
        else:
            assert hc["sigma_f64"] == 0 and hc["sigma_unresolved"] > 0
        if tie_arith == "0":
            assert hc["best_f64"] == 0
            X = [orc.Region(b, g, p).set_fast(1).run_all(orc.MODE_EXACT) for g in range(b.n_regions)]
            T = [orc.Region(b, g, p).set_fast(1).set_tie_mask(orc.tie_mask(0, 0)).run_all(orc.MODE_TIE) for g in range(b.n_regions)]
            assert all(x.vcf_text("c") == t.vcf_text("c") and np.array_equal(x.phase_result()["haplotag"], t.phase_result()["haplotag"]) for x, t in zip(X, T))
    # levels 2 and 3 differ only where a step's changes are all tie changes: the batch of test_tie_only_steps_take_the_repair_pass meets
    # 16 of them -- level 3 must send them through the repair pass, level 2 must count them as unresolved (the switch used to
    # clamp 3 to 2 and the cases above could not tell)
    if tie_arith in ("2", "3"):
        b = synth.make_batch("masseq", n_genes=12, gene_len=16000, depth=40, seed=1)
        p = _abi.make_params("hifi-masseq", seed=2025)
        full_check(engine_cls, orc, b, p)
        E = engine_cls(0, p)
        E.load_batch(b).run_all()
        hc = E.tie_census()
        E.close()
        if tie_arith == "3":
            assert hc["delta_step_f64"] >= 10 and hc["step_unresolved"] == 0 and hc["delta_unresolved"] == 0, hc
        else:
            assert hc["delta_step_f64"] == 0 and hc["step_unresolved"] >= 10, hc


def test_fragment_rows_beyond_the_hit_lists(engine_cls, orc):
    """Reads that face more survivors than k2_hist's per-read hit list holds (LCR_HITS = 16): 40 het sites under every read plus
    error columns -- those rows take the list walk (k3_walk_list, count and fill) beside rows that fit; both K3 paths and the
    oracle agree."""
    b, sites = helpers.two_haplotype_batch(n_snps=40, n_reads=36, seed=6)
    rng = np.random.default_rng(8)
    short = helpers.two_haplotype_batch(n_snps=40, n_reads=36, seed=6)[0]   # (same reference and sites)
    del short
    p = _abi.make_params("hifi-masseq", seed=5)
    c = full_check(engine_cls, orc, b, p)
    assert len(c) == 40
    E = engine_cls(0, p)
    E.load_batch(b).run_all()
    fm1 = {k: v.copy() for k, v in E.fragmat().items()}
    E.debug_set("k3_hits", 0)
    E.load_batch(b).run_all()
    fm0 = E.fragmat()
    for k in fm1:
        assert np.array_equal(fm1[k], fm0[k]), k
    assert np.diff(fm1["row_ptr"]).max() == 40
    E.close()


def test_filter_pass_in_the_tally_epilogue(engine_cls, orc):
    """Round 6: on the ONT presets (no poly-A pass behind the tally) k1_pileup's epilogue takes pass 1 of the candidate filters
    (candidate.rs:90-234, k2_eval.h) on the counts it holds, and lcr_candidates uses those flags when it is called with the same filter
    parameters.  Both ways -- and a call with other parameters than the pileup's, which must take k2_filter's pass -- give the oracle's
    candidates; strand-bias preset (ont-cdna) and the one without (ont-drna)."""
    for prof, preset, seed in (("ont-cdna", "ont-cdna", 21), ("ont-drna", "ont-drna", 22)):
        b = synth.make_batch(prof, n_genes=4, gene_len=11000, depth=40, seed=seed)
        p = _abi.make_params(preset, seed=seed)
        full_check(engine_cls, orc, b, p)                       # (default: fused)
        E1, E0 = engine_cls(0, p), engine_cls(0, p)
        E0.debug_set("fuse_filter", 0)
        E1.load_batch(b).run_all(); E0.load_batch(b).run_all()
        assert _result_bytes(E1) == _result_bytes(E0)
        # other filter parameters at lcr_candidates than at lcr_pileup: the flags of the epilogue do not apply
        p2 = _abi.make_params(preset, seed=seed, min_depth=int(p.min_depth) + 6)
        E1.load_batch(b).fill_data_into_freq_vec()
        E1.params = p2
        E1.get_candidate_snps().get_fragments().phase()
        E0.params = p2
        E0.load_batch(b).run_all()
        assert _result_bytes(E1) == _result_bytes(E0)
        c2 = E1.candidates()[0]
        E1.params = p
        E1.load_batch(b).run_all()
        assert E1.candidates()[0].size >= c2.size
        E1.close(); E0.close()


def test_survivor_compaction_queued_before_its_count_is_known(engine_cls, orc):
    """Round 6: lcr_candidates queues the survivors' compaction before the host knows their number, into buffers sized by the context's
    previous batch (+ a quarter).  A context that has seen a small batch and then gets one with many times the survivors (the kernel drops
    what does not fit and is run again), and the other way round, must produce what a fresh context does -- and the oracle."""
    small = synth.make_batch("ont-cdna", n_genes=2, gene_len=6000, depth=25, seed=3)
    big = synth.make_batch("ont-cdna", n_genes=6, gene_len=14000, depth=60, seed=4)
    p = _abi.make_params("ont-cdna")
    E = engine_cls(0, p)
    got = {}
    for name, b in (("small", small), ("big", big), ("small2", small), ("big2", big)):
        E.load_batch(b).run_all()
        c, off = E.candidates()
        got[name] = (c.tobytes(), off.tobytes(), E.phase_result()["haplotag"].tobytes())
    E.close()
    assert got["small"] == got["small2"] and got["big"] == got["big2"]
    for name, b in (("small", small), ("big", big)):
        F = engine_cls(0, p)
        F.debug_set("spec_compact", 0)
        F.load_batch(b).run_all()
        c, off = F.candidates()
        assert got[name] == (c.tobytes(), off.tobytes(), F.phase_result()["haplotag"].tobytes()), name
        F.close()
    full_check(engine_cls, orc, big, p)


def test_fill_kernels_and_the_runtime_fallback(engine_cls, orc, monkeypatch):
    """Round 6: the stage calls fill their buffers with kernels of the library's own (lcr_fill_async, several ranges per launch: every launch
    costs ~5 us of queue, and the runtime's fill started late behind an event record).  Every alignment of both ends against the host
    (lcr_debug_set("fill_selftest")), and the whole path once with lcr_debug_set("own_fill", 0) = hipMemsetAsync."""
    E = engine_cls(0, _abi.make_params("ont-cdna"))
    E.debug_set("fill_selftest", 1)
    E.close()
    monkeypatch.setenv("LCR_OWN_FILL", "0")
    b = synth.make_batch("ont-cdna", n_genes=3, gene_len=8000, depth=30, seed=7)
    full_check(engine_cls, orc, b, _abi.make_params("ont-cdna"))


def test_chain_ties_of_classes_2_and_4(engine_cls, orc, monkeypatch):
    """Round 6: chain regions of workgroup scope that meet a tie of class 2 (a delta / eta choice with two equal maxima), class 4 (a step
    whose only changes were tie changes) or class 8 (a later configuration of equal objective that differs from the best one) are run again
    by k4_chain_wg's COMPLETE instantiation, which decides them by the reference-order f64 scores / sums.  On chain regions built to meet such ties -- twelve het sites, three of them with the allele flipped in half the reads of
    each haplotype, equal qualities: 30-54 delta ties and up to two tie-only steps per region -- the HIP results are the oracle's with
    every class resolved (ORC_MODE_TIE, chain mask 15), none is reported unresolved and the decided ones are the oracle's census, count for
    count.  With lcr_debug_set("chain_ties", 0) the first run's result stands: the oracle's under chain mask 1, and delta_unresolved /
    step_unresolved are ITS census, count for count."""
    alt_of = {ord("A"): ord("C"), ord("C"): ord("A"), ord("G"): ord("T"), ord("T"): ord("G")}
    met = [0, 0]
    for seed in (16, 21, 22, 29, 38):
        b, sites = helpers.two_haplotype_batch(n_snps=12, n_reads=12, seed=seed)
        rng = np.random.default_rng(seed)
        bases = b.bases.copy()
        for j in rng.choice(12, size=3, replace=False):
            x = sites[0][j] - 5000
            for k in range(b.n_reads):
                if (k // 2) % 2 == 0:
                    o = int(b.seq_off[k]) + x
                    bases[o] = alt_of[int(bases[o])]
        b2 = _abi.ReadBatch(**{f: getattr(b, f) for f in b.FIELDS if f != "bases"}, bases=bases, start0=b.start0, len=b.len, read_begin=b.read_begin, ref=b.ref)
        p = _abi.make_params("hifi-masseq", seed=seed)
        for resolve in (1, 0):
            monkeypatch.setenv("LCR_CHAIN_TIES", str(resolve))
            monkeypatch.setitem(ORACLE_TIE_MASK, 0, orc.TIE_MASK_LIBLCR if resolve else orc.TIE_MASK_LIBLCR_GRID)
            c = full_check(engine_cls, orc, b2, p)
            assert len(c) == 12                         # one chain region (S > max_enum_snps = 10)
            regs = oracle_all(orc, b2, p)
            oc = regs[0].tie_census()
            E = engine_cls(0, p)
            E.load_batch(b2).run_all()
            hc = E.tie_census()
            E.close()
            if resolve:
                assert hc["delta_unresolved"] == 0 and hc["step_unresolved"] == 0 and hc["delta_step_f64"] == int(oc[1]) + int(oc[2]), (seed, hc, oc.tolist())
            else:
                assert hc["delta_unresolved"] == int(oc[1]) and hc["step_unresolved"] == int(oc[2]), (seed, hc, oc.tolist())
                met[0] += int(oc[1]); met[1] += int(oc[2])
    assert met[0] > 100 and met[1] >= 1, met


def test_tie_only_steps_take_the_repair_pass(engine_cls, orc):
    """A batch whose enumeration restarts meet steps with tie changes only (oracle census: 16 of them in one region): the fast
    kernels put those restarts on the repair list, k4_enum_redo decides the steps by the reference's sums of f64 scores, and
    the census reports them as decided -- none unresolved.  The same through the global-memory class."""
    b = synth.make_batch("masseq", n_genes=12, gene_len=16000, depth=40, seed=1)
    p = _abi.make_params("hifi-masseq", seed=2025)
    regs = oracle_all(orc, b, p)
    n_steps = sum(int(R.tie_census()[2]) for R in regs)
    assert n_steps >= 10
    full_check(engine_cls, orc, b, p)
    for big in (0, 1):
        E = engine_cls(0, p)
        E.debug_set("enum_force_big", big)
        E.load_batch(b).run_all()
        hc = E.tie_census()
        E.close()
        assert hc["delta_step_f64"] >= n_steps and hc["step_unresolved"] == 0 and hc["delta_unresolved"] == 0 and hc["best_unresolved"] == 0, hc


def test_strand_bias_and_isoseq_preset(engine_cls, orc):
    b = synth.make_batch("ont-cdna", n_genes=3, gene_len=8000, depth=50, seed=31)
    full_check(engine_cls, orc, b, _abi.make_params("hifi-isoseq", seed=1))


def test_batch_composition_independence(engine_cls, orc):
    """A region's result must not depend on which other regions share the launch."""
    b = synth.make_batch("ont-cdna", n_genes=3, gene_len=7000, depth=30, seed=41)
    p = _abi.make_params("ont-cdna")
    E = engine_cls(0, p)
    E.load_batch(b).run_all()
    c_all, off = E.candidates()
    # region 1 alone
    rb, re_ = int(b.read_begin[1]), int(b.read_begin[2])
    so, eo = int(b.seq_off[rb]), int(b.seq_off[re_ - 1] + b.seq_len[re_ - 1])
    co, ce = int(b.cig_off[rb]), int(b.cig_off[re_ - 1] + b.n_cig[re_ - 1])
    one = _abi.ReadBatch(pos=b.pos[rb:re_], seq_len=b.seq_len[rb:re_], lead_clip=b.lead_clip[rb:re_],
                         trail_clip=b.trail_clip[rb:re_], flags=b.flags[rb:re_], seq_off=b.seq_off[rb:re_] - so,
                         cig_off=b.cig_off[rb:re_] - co, n_cig=b.n_cig[rb:re_], bases=b.bases[so:eo],
                         quals=b.quals[so:eo], cigar=b.cigar[co:ce], start0=b.start0[1:2], len=b.len[1:2],
                         read_begin=[0, re_ - rb], ref=b.ref[int(b.col_off[1]):int(b.col_off[2])])
    E2 = engine_cls(0, p)
    E2.load_batch(one).run_all()
    c_one, _ = E2.candidates()
    sub = c_all[off[1]:off[2]].copy()
    sub["region"] = 0
    assert sub.tobytes() == c_one.tobytes()


def test_edge_cases(engine_cls, orc):
    """Ragged / degenerate inputs: lower-case and N reference, IUPAC and N read bases, q = 0 and q > 30,
    leading insertion at a tile boundary, deletions / introns clipped by the window, reads starting
    left of the window, hard+soft clips, a region without reads, a 1-column region."""
    L = 2100
    rng = np.random.default_rng(7)
    ref = "".join(rng.choice(list("ACGT"), size=L))
    ref = ref[:500] + ref[500:520].lower() + "NNNN" + ref[524:]
    reads = []
    def seq_of(pos0, n):
        return ref[pos0:pos0 + n].upper().replace("N", "A")
    for k in range(40):
        p0 = 1000 + int(rng.integers(0, 40))
        s = list(seq_of(p0, 600))
        s[100] = "R"; s[101] = "N"
        if k % 2:
            s[300] = "T" if s[300] != "T" else "G"
        q = rng.integers(0, 45, size=600).tolist()
        reads.append(dict(pos=1000 + p0, seq="".join(s), qual=q, cigar="600M", rev=k % 2, ts=1 + k % 2))
    # read starting left of the window, with a deletion and an intron crossing the window start
    reads.insert(0, dict(pos=900, seq=seq_of(0, 50) + seq_of(130, 500), qual=20, cigar="50M30D20N500M", ts=1))
    # leading soft clip + insertion right at tile boundary column 1024 (pos 2024), hard clip in front
    reads.append(dict(pos=2024, seq="A" * 10 + "CCC" + seq_of(1024, 520), qual=33, cigar="5H10S3I520M", ts=2))
    reads.append(dict(pos=2024, seq="GG" + seq_of(1024, 520), qual=9, cigar="2I520M7H"))
    reads.sort(key=lambda r: r["pos"])
    for r in reads:
        r["region"] = 0
    lone = dict(pos=9000, seq="ACGT" * 130, qual=30, cigar="520M", region=2)
    b = helpers.mk_batch(reads + [lone], [(1000, ref), (5000, "ACGTACGTAC"), (9000, "ACGT" * 130), (9900, "A")])
    p = _abi.make_params("hifi-masseq", min_depth=3)
    full_check(engine_cls, orc, b, p)
    p = _abi.make_params("ont-cdna", min_depth=3)
    full_check(engine_cls, orc, b, p)


def _low_fraction_batch(seed, frac=0.3):
    """80 reads over 3 kb, four ordinary het SNPs (alt on haplotype A) and two sites whose alt allele sits on
    a fraction of haplotype A's reads only (allele frequency 5-20 %: the low-fraction "somatic" list,
    candidate.rs:410-417)."""
    L = 3000
    rng = np.random.default_rng(seed)
    ref = "".join(rng.choice(list("ACGT"), size=L))
    alt_of = {"A": "C", "C": "A", "G": "T", "T": "G"}   # never A>G / T>C: those are the RNA-edit class
    reads = []
    for k in range(80):
        hap_a = k % 2 == 0
        p0, n = int(rng.integers(0, 600)), int(rng.integers(1800, 2300))
        s = list(ref[p0:p0 + n])
        for x in (600, 1100, 1700, 2300):
            if hap_a and p0 <= x < p0 + n:
                s[x - p0] = alt_of[ref[x]]
        for x in (900, 2000):
            if hap_a and rng.random() < frac and p0 <= x < p0 + n:
                s[x - p0] = alt_of[ref[x]]
        reads.append(dict(pos=1000 + p0, seq="".join(s), qual=30, cigar="%dM" % len(s), rev=int(rng.integers(0, 2)), region=0))
    reads.sort(key=lambda r: r["pos"])
    return helpers.mk_batch(reads, [(1000, ref)])


@pytest.mark.parametrize("min_phase_score,rescued", [(13.0, False), (4.0, True)])
def test_low_fraction_rescue(engine_cls, orc, min_phase_score, rescued):
    """eval_low_frac_var_phase (snpfrags.rs:283-376): both outcomes of the rescue of a low-fraction site, incl. the
    random haplotags its unassigned reads receive (counter-based draws in the reference's call order)."""
    b = _low_fraction_batch(seed=1)
    c = full_check(engine_cls, orc, b, _abi.make_params("ont-cdna", seed=3, min_phase_score=min_phase_score))
    som = c[(c["pos"] == 1000 + 900) | (c["pos"] == 1000 + 2000)]
    assert len(som) >= 1
    if rescued:
        assert np.all((som["flags"] & _abi.F_FOR_PHASING) != 0) and np.all(som["phase_score"] > 1.0)
    else:
        assert np.all((som["flags"] & _abi.F_CAND_SOMATIC) != 0) and np.all((som["flags"] & _abi.F_FOR_PHASING) == 0)


def test_long_deletions_grow_the_record_pool(engine_cls, orc):
    """K0 sizes its record pool from ops + reads + bases / tile; deletion runs that cross dozens of tiles
    exceed that estimate: the stage must notice the overflow, repeat with a larger pool and still agree
    with the oracle (planes, and the rest of the pipeline on the same batch)."""
    L = 60 * 1024
    rng = np.random.default_rng(3)
    ref = "".join(rng.choice(list("ACGT"), size=L))
    reads = []
    for k in range(300):
        p0 = int(rng.integers(0, 200))
        dlen = 50 * 1024 + int(rng.integers(0, 2000))
        tail = p0 + 300 + dlen
        s = list(ref[p0:p0 + 300] + ref[tail:tail + 300])
        if k % 2:
            s[150] = "T" if s[150] != "T" else "G"
        reads.append(dict(pos=5000 + p0, seq="".join(s), qual=30, cigar="300M%dD300M" % dlen, rev=k % 2, ts=1 + k % 2, region=0))
    reads.sort(key=lambda r: r["pos"])
    b = helpers.mk_batch(reads, [(5000, ref)])
    full_check(engine_cls, orc, b, _abi.make_params("hifi-masseq", min_depth=3))


def test_dense_ops_span_several_pool_levels(engine_cls, orc):
    """One read can put hundreds of records into a single tile (alternating 1-base ops): its slot
    reservation then covers several pool levels at once, all of which it has to allocate, and reads whose
    CIGARs are longer than the 64 ops K0 keeps in registers (here 300 - 550 ops) take the reload path."""
    L = 4096
    rng = np.random.default_rng(11)
    ref = "".join(rng.choice(list("ACGT"), size=L))
    reads = []
    for k in range(40):
        p0 = 100 + int(rng.integers(0, 600))
        n = 150 + int(rng.integers(0, 120))
        seq, cig, rp = [], [], p0
        for i in range(n):
            if i % 2 == 0:
                seq.append(ref[rp]); rp += 1; cig.append("1M"); seq.append("ACGT"[(i // 2) % 4]); cig.append("1I")
            else:
                seq.append(ref[rp]); rp += 1; cig.append("1M"); rp += 1; cig.append("1D")
        seq.append(ref[rp:rp + 40]); cig.append("40M")
        reads.append(dict(pos=1000 + p0, seq="".join(seq), qual=25, cigar="".join(cig), rev=k % 2, ts=k % 3, region=0))
    reads.sort(key=lambda r: r["pos"])
    b = helpers.mk_batch(reads, [(1000, ref)])
    full_check(engine_cls, orc, b, _abi.make_params("ont-cdna", min_depth=3))


def test_contexts_on_concurrent_host_threads(engine_cls):
    """A context per worker thread (the reference runs its regions on a rayon pool): two contexts driven by
    two host threads at the same time give, batch by batch, what one context gives alone."""
    import threading
    batches = [synth.make_batch("ont-cdna", n_genes=6, gene_len=9000, depth=35, seed=71),
               synth.make_batch("ont-drna", n_genes=4, gene_len=20000, depth=40, seed=72)]
    params = [_abi.make_params("ont-cdna", seed=3), _abi.make_params("ont-drna", seed=4)]

    def snapshot(E):
        c, off = E.candidates()
        pr = E.phase_result()
        return (E.columns().copy(), c.copy(), off.copy(), E.fragmat()["val"].copy(), pr["haplotag"].copy(),
                pr["phase_set"].copy(), pr["objective"].copy())

    want = []
    for b, p in zip(batches, params):
        E = engine_cls(0, p)
        E.load_batch(b).run_all()
        want.append(snapshot(E))
        E.close()
    got, errs = [[], []], []

    def worker(j):
        try:
            E = engine_cls(0, params[j])
            for _ in range(6):
                E.load_batch(batches[j]).run_all()
                got[j].append(snapshot(E))
            E.close()
        except BaseException as e:   # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(j,)) for j in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    for j in range(2):
        for snap in got[j]:
            for a, b in zip(snap, want[j]):
                assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes()


def test_deep_region_beyond_the_lds_images(engine_cls, orc):
    """A region with more rows / entries than the LDS images of k4_stage (4096 rows, 8192 entries) and k4_post
    hold: the phase stage then works from global memory and finishes with the host epilogue."""
    b = synth.make_batch("ont-cdna", n_genes=1, gene_len=9000, depth=600, seed=91)
    full_check(engine_cls, orc, b, _abi.make_params("ont-cdna", seed=9))


def _deep_low_s_batch(n_snps, n_reads, seed, err=0.08):
    """one gene, depth n_reads, n_snps het sites: two haplotypes + allele errors at the het sites (ref <-> alt, so the sites stay
    biallelic) and mixed base qualities -- the restarts end in several optima and many of them tie in objective"""
    b, sites = helpers.two_haplotype_batch(n_snps=n_snps, n_reads=n_reads, seed=seed)
    rng = np.random.default_rng(seed + 100)
    alt_of = {ord("A"): ord("C"), ord("C"): ord("A"), ord("G"): ord("T"), ord("T"): ord("G")}
    b.quals[:] = rng.choice(np.array([7, 12, 18, 25, 35], dtype=np.uint8), size=b.quals.size)
    for k in range(b.n_reads):
        for x in sites[0]:
            if rng.random() < err:
                i = int(b.seq_off[k]) + (x - int(b.pos[k]))
                r = int(b.ref[x - 5000])
                b.bases[i] = alt_of[r] if b.bases[i] == r else r
    return b


@pytest.mark.parametrize("preset,n_snps,n_reads", [("hifi-masseq", 5, 3200), ("ont-cdna", 8, 3000), ("hifi-masseq", 3, 4000)])
def test_deep_region_with_few_sites(engine_cls, orc, preset, n_snps, n_reads):
    """What real RNA-seq meets first: ONE highly expressed gene, depth >= 3 000, 3-8 het sites.  Its phase
    matrix is beyond the enumeration kernels' LDS images, so the restarts run from global memory (k4_enum_big) -- compared under
    the FULL oracle mask (sigma ties and `prob > largest_prob` at equal objective by the reference-order f64 sums) and with
    ORC_MODE_F64, and the census reports no unresolved tie."""
    b = _deep_low_s_batch(n_snps, n_reads, seed=n_snps)
    p = _abi.make_params(preset, seed=3, max_depth=100000)
    c = full_check(engine_cls, orc, b, p)
    assert len(c) == n_snps
    E = engine_cls(0, p)
    E.load_batch(b).run_all()
    hc = E.tie_census()
    E.close()
    assert hc["sigma_unresolved"] == 0 and hc["best_unresolved"] == 0, hc


def test_bam_file_to_vcf_through_the_native_decoder(engine_cls, orc):
    """The caller's side of the path, end to end: liblcr's BAM decoder (N1) -> GPU region discovery (N3) -> batch ->
    pileup / candidates / fragments / phase, against the oracle on the batch the Python reader builds."""
    import os
    from longcallr_amd import bamio
    nb = bamio.NativeBam(os.path.join(helpers.GOLDEN, "demo.bam"), 4)
    rid = [n for n, _ in nb.refs].index("chr20")
    p = _abi.make_params("hifi-masseq")
    E = engine_cls(0, p)
    rs, re_ = nb.spans(rid, **_abi.READ_FILTER)
    regions = E.discover_regions(rs, re_, nb.refs[rid][1])
    assert regions == [(16729960, 13256, 1649)]
    E.close()
    b = nb.batch(rid, [(s, l) for s, l, _ in regions], [helpers.load_pseudo_ref()], **_abi.READ_FILTER)
    ref_b = helpers.demo_batch()
    for f in _abi.ReadBatch.FIELDS + ["start0", "len", "read_begin", "ref"]:
        assert np.array_equal(getattr(b, f), getattr(ref_b, f)), f
    c = full_check(engine_cls, orc, b, p, "chr20")
    assert len(c) == 19


def test_phased_bam_from_the_gpu_results(engine_cls, tmp_path):
    """N4 end to end: demo.bam -> pipeline -> lcr_bam_write_phased; every written record carries exactly the HP / PS
    the phase stage gave its read (thread.rs:204-221, 346-357)."""
    import os
    import struct
    from longcallr_amd import bamio
    src = os.path.join(helpers.GOLDEN, "demo.bam")
    nb = bamio.NativeBam(src, 4)
    b = helpers.demo_batch()
    E = engine_cls(0, _abi.make_params("hifi-masseq"))
    E.load_batch(b).run_all()
    fm, pr = E.fragmat(), E.phase_result()
    names = [b.names[r] for r in fm["row_read"]]
    asg, ps = pr["assignment"].astype(np.int32), pr["phase_set"]
    hp = np.where((fm["row_for_phasing"] != 0) | (asg != 0), asg, -1)
    assert (asg != 0).sum() > 500 and (ps != 0).sum() > 500
    rid = [n for n, _ in nb.refs].index("chr20")
    out = str(tmp_path / "phased.bam")
    nb.write_phased(out, [(rid, int(b.start0[0]), int(b.len[0]))], names, hp, ps)
    _, recs = bamio.read_bam(out, keep_raw=True)
    want_hp = {n: int(h) for n, h in zip(names, hp) if h > 0}
    want_ps = {n: int(p) for n, p in zip(names, ps) if p != 0}
    seen = 0
    for r in recs:
        tail = r["raw"][-14:]
        got_ps = struct.unpack("<I", tail[-4:])[0] if tail[-7:-4] == b"PSI" else None
        rest = tail[:-7] if got_ps is not None else tail
        got_hp = struct.unpack("<i", rest[-4:])[0] if rest[-7:-4] == b"HPi" else None
        assert got_hp == want_hp.get(r["name"]) and got_ps == want_ps.get(r["name"]), r["name"]
        seen += got_hp is not None
    assert seen == len(want_hp) and len(recs) >= len(names)
    E.close()


def test_driver_bam_and_fasta_to_vcf_and_phased_bam(engine_cls, orc, tmp_path):
    """longcallr_amd.pipeline.run: the loop of thread.rs:26-361 around the hot path, files in, files out."""
    import os
    from longcallr_amd import bamio, pipeline
    src = os.path.join(helpers.GOLDEN, "demo.bam")
    refs, _ = bamio.read_bam(src)
    b = helpers.demo_batch()
    start0, length = int(b.start0[0]), int(b.len[0])
    fa = str(tmp_path / "pseudo.fa")
    with open(fa, "wb") as f, open(fa + ".fai", "w") as fi:   # chr20 = N everywhere but the demo window (pseudo-reference)
        for name, ln in refs:
            if name not in ("chr19", "chr20"):
                continue
            seq = np.full(ln, ord("N"), np.uint8)
            if name == "chr20":
                seq[start0:start0 + length] = helpers.load_pseudo_ref()
            f.write(b">" + name.encode() + b" pseudo\n" + seq.tobytes() + b"\n")
            fi.write("%s\t%d\t0\t%d\t%d\n" % (name, ln, ln, ln + 1))
    out_vcf, out_bam = str(tmp_path / "out.vcf"), str(tmp_path / "out.bam")
    st = pipeline.run(src, fa, out_vcf, out_bam, preset="hifi-masseq", threads=4)
    assert st["contigs"] == 1 and st["regions"] == 1 and st["reads"] == b.n_reads and st["candidates"] == 19
    text = open(out_vcf).read()
    head, body = text[:text.index("#CHROM")], text[text.index("#CHROM"):].split("\n", 1)[1]
    assert head.startswith("##fileformat=VCFv4.3\n##contig=<ID=chr19,length=58617616>\n##contig=<ID=chr20,length=64444167>\n##FILTER=<ID=PASS")
    p = _abi.make_params("hifi-masseq")
    R = oracle_all(orc, b, p)[0]
    assert body == R.vcf_text("chr20") and st["vcf_records"] == body.count("\n")
    _, recs = bamio.read_bam(out_bam)
    assert len(recs) >= b.n_reads and all(r["ref_id"] == [n for n, _ in refs].index("chr20") for r in recs)


def _random_batch(seed):
    """Reads with random CIGAR structure (M / = / X runs, insertions, deletions, introns, soft and hard clips,
    ops of length 1 next to ops of hundreds of bases), random strands / ts tags / qualities (0 .. 50), two haplotypes
    with het SNPs, in 1 - 3 regions whose lengths are not multiples of the tile size."""
    rng = np.random.default_rng(seed)
    regions, reads = [], []
    start = 1000
    for g in range(int(rng.integers(1, 4))):
        L = int(rng.integers(700, 2600))
        ref = rng.choice(list("ACGT"), size=L)
        if rng.random() < 0.5:
            ref[int(rng.integers(0, L))] = "N"
        snps = sorted(rng.choice(np.arange(50, L - 50), size=int(rng.integers(2, 14)), replace=False).tolist())
        alt = {p_: rng.choice([c for c in "ACGT" if c != ref[p_]]) for p_ in snps}
        n_reads = int(rng.integers(25, 70))
        rr = []
        for k in range(n_reads):
            hap = k % 2
            pos = int(rng.integers(0, L // 2))
            rp, seq, cig = pos, [], []
            if rng.random() < 0.3:
                cig.append("%dH" % rng.integers(1, 9))
            if rng.random() < 0.5:
                n = int(rng.integers(1, 25)); cig.append("%dS" % n); seq.extend(rng.choice(list("ACGT"), size=n))
            budget = int(rng.integers(150, 900))
            first = True
            while rp < L - 5 and budget > 0:
                u = rng.random()
                if first or u < 0.62:
                    n = int(min(L - rp, budget, rng.choice([1, 2, 7, 40, 150, 400])))
                    for c in range(rp, rp + n):
                        bse = ref[c] if ref[c] != "N" else "A"
                        if c in alt and hap == 1:
                            bse = alt[c]
                        if rng.random() < 0.02:
                            bse = rng.choice(list("ACGTN"))
                        seq.append(bse)
                    cig.append("%d%s" % (n, rng.choice(["M", "M", "M", "=", "X"]))); rp += n; budget -= n
                elif u < 0.74:
                    n = int(rng.choice([1, 1, 2, 5])); cig.append("%dI" % n); seq.extend(rng.choice(list("ACGT"), size=n))
                elif u < 0.86:
                    n = int(min(L - rp - 1, rng.choice([1, 1, 3, 30]))); 
                    if n > 0:
                        cig.append("%dD" % n); rp += n
                else:
                    n = int(min(L - rp - 1, rng.choice([20, 120, 700])))
                    if n > 0:
                        cig.append("%dN" % n); rp += n
                first = False
            if not cig or cig[-1][-1] in "DNI":   # end on an aligned block
                n = int(min(L - rp, 10))
                if n <= 0:
                    continue
                seq.extend((ref[c] if ref[c] != "N" else "A") for c in range(rp, rp + n)); cig.append("%dM" % n)
            if rng.random() < 0.4:
                n = int(rng.integers(1, 20)); cig.append("%dS" % n); seq.extend(rng.choice(list("ACGT"), size=n))
            if rng.random() < 0.2:
                cig.append("%dH" % rng.integers(1, 5))
            rr.append(dict(pos=start + pos, seq="".join(seq), qual=rng.integers(0, 51, size=len(seq)).tolist(), cigar="".join(cig),
                           rev=int(rng.integers(0, 2)), ts=int(rng.integers(0, 3)), region=g))
        rr.sort(key=lambda r: r["pos"])
        reads.extend(rr)
        regions.append((start, "".join(ref)))
        start += L + int(rng.integers(500, 5000))
    return helpers.mk_batch(reads, regions)


@pytest.mark.parametrize("seed", list(range(100, 112)))
def test_random_cigar_structures(engine_cls, orc, seed):
    b = _random_batch(seed)
    preset = ["hifi-masseq", "hifi-isoseq", "ont-cdna", "ont-drna"][seed % 4]
    full_check(engine_cls, orc, b, _abi.make_params(preset, seed=seed, min_depth=3))


def test_driver_on_a_synthetic_two_contig_bam(engine_cls, tmp_path):
    """pipeline.run (native decode, GPU region discovery, native batching) against the same steps done by the
    Python restatements (bamio.read_bam / discover_regions / build_batch) on a BAM written from synthetic
    spliced reads on two contigs: identical VCF text, region for region."""
    import struct
    import zlib
    from longcallr_amd import bamio, pipeline, vcf as vcfmod
    b = synth.make_batch("ont-cdna", n_genes=6, gene_len=8000, depth=30, seed=77)
    contig_of = [0, 0, 0, 1, 1, 1]
    names = ["ctgA", "ctgB"]
    clen = [int(max(b.start0[g] + b.len[g] for g in range(6) if contig_of[g] == c)) + 777 for c in range(2)]
    recs = []
    for g in range(6):
        for r in range(b.read_begin[g], b.read_begin[g + 1]):
            so, co, n, nc = int(b.seq_off[r]), int(b.cig_off[r]), int(b.seq_len[r]), int(b.n_cig[r])
            seq = b.bases[so:so + n]
            packed = np.zeros((n + 1) // 2, np.uint8)
            code = np.array([("=ACMGRSVTWYHKDBN").index(chr(c)) for c in seq], np.uint8)
            packed[:len(code[0::2])] |= code[0::2] << 4
            packed[:len(code[1::2])] |= code[1::2]
            fl = int(b.flags[r])
            aux = (b"tsA+" if (fl >> 1) == 1 else b"tsA-" if (fl >> 1) == 2 else b"") + b"def" + struct.pack("<f", 0.01)
            name = ("r%d_%d" % (g, r)).encode() + b"\0"
            body = (struct.pack("<iiBBHHHiiii", contig_of[g], int(b.pos[r]), len(name), 60, 4680, nc, 16 if fl & 1 else 0, n, -1, -1, 0)
                    + name + b.cigar[co:co + nc].astype("<u4").tobytes() + packed.tobytes() + b.quals[so:so + n].tobytes() + aux)
            recs.append((contig_of[g], int(b.pos[r]), len(recs), struct.pack("<i", len(body)) + body))
    recs.sort(key=lambda t: t[:3])
    text = b"@HD\tVN:1.6\tSO:coordinate\n"
    raw = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 2)
    for nm, ln in zip(names, clen):
        raw += struct.pack("<i", len(nm) + 1) + nm.encode() + b"\0" + struct.pack("<i", ln)
    raw += b"".join(t[3] for t in recs)
    out = []
    for off in list(range(0, len(raw), 60000)) + [None]:
        chunk = b"" if off is None else raw[off:off + 60000]
        co_ = zlib.compressobj(1, zlib.DEFLATED, -15)
        cd = co_.compress(chunk) + co_.flush()
        out.append(b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, 12 + 6 + len(cd) + 8 - 1)
                   + cd + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    bam = str(tmp_path / "syn.bam")
    open(bam, "wb").write(b"".join(out))
    fa = str(tmp_path / "syn.fa")
    ctg_seq = [np.full(clen[c], ord("N"), np.uint8) for c in range(2)]
    for g in range(6):
        o = int(b.col_off[g])
        ctg_seq[contig_of[g]][int(b.start0[g]):int(b.start0[g]) + int(b.len[g])] = b.ref[o:o + int(b.len[g])]
    with open(fa, "wb") as f, open(fa + ".fai", "w") as fi:
        for c in range(2):
            f.write(b">" + names[c].encode() + b"\n" + ctg_seq[c].tobytes() + b"\n")
            fi.write("%s\t%d\t0\t%d\t%d\n" % (names[c], clen[c], clen[c], clen[c] + 1))
    out_vcf = str(tmp_path / "syn.vcf")
    st = pipeline.run(bam, fa, out_vcf, preset="ont-cdna", threads=4, seed=2025)
    assert st["contigs"] == 2 and st["reads"] == b.n_reads
    body = open(out_vcf).read().split("#CHROM")[1].split("\n", 1)[1]
    # the same with the Python restatements
    refs, precs = bamio.read_bam(bam)
    p = _abi.make_params("ont-cdna", seed=2025)
    want = []
    n_regions = 0
    for c in range(2):
        keep = [r for r in precs if r["ref_id"] == c and bamio.passes_filter(r, **_abi.READ_FILTER)]
        regions = bamio.discover_regions(keep, c, clen[c])
        n_regions += len(regions)
        wins = [ctg_seq[c][s_:s_ + l_] for s_, l_, _ in regions]
        pb = bamio.build_batch(keep, [(s_, l_) for s_, l_, _ in regions], wins)
        E = engine_cls(0, p)
        E.load_batch(pb).run_all()
        want.append(vcfmod.format_records(E.candidates()[0], names[c], p.min_phase_score))
        E.close()
    assert st["regions"] == n_regions == 6
    assert body == "".join(want) and body.count("\n") >= 8
    # sharded and chunked: two contexts on the device list [0, 0] (one host thread each), chunks of a single region --
    # the same VCF and the same phased BAM, byte for byte on the inflated stream
    out_bam1, out_bam2, out_vcf2 = str(tmp_path / "p1.bam"), str(tmp_path / "p2.bam"), str(tmp_path / "syn2.vcf")
    pipeline.run(bam, fa, out_vcf, out_bam1, preset="ont-cdna", threads=4, seed=2025)
    st2 = pipeline.run(bam, fa, out_vcf2, out_bam2, preset="ont-cdna", threads=4, seed=2025, devices=[0, 0], chunk_cost=1.0)
    assert st2["chunks"] == 6 and st2["reads"] == st["reads"] and st2["candidates"] == st["candidates"]
    assert open(out_vcf2).read() == open(out_vcf).read()
    assert bamio.bgzf_decompress(out_bam1) == bamio.bgzf_decompress(out_bam2)
    # the engines' asynchronous phase stage (pipeline.run's default: a chunk's upload + pileup beside the previous chunk's tails, results
    # through lcr_collect_phase) against the synchronous stage, one context over six single-region chunks: the same files
    out_bam3, out_vcf3 = str(tmp_path / "p3.bam"), str(tmp_path / "syn3.vcf")
    st3 = pipeline.run(bam, fa, out_vcf3, out_bam3, preset="ont-cdna", threads=4, seed=2025, chunk_cost=1.0, async_phase=False)
    out_bam4, out_vcf4 = str(tmp_path / "p4.bam"), str(tmp_path / "syn4.vcf")
    st4 = pipeline.run(bam, fa, out_vcf4, out_bam4, preset="ont-cdna", threads=4, seed=2025, chunk_cost=1.0, async_phase=True)
    assert st3["chunks"] == st4["chunks"] == 6 and st3 == st4
    assert open(out_vcf3).read() == open(out_vcf4).read() == open(out_vcf).read()
    assert bamio.bgzf_decompress(out_bam3) == bamio.bgzf_decompress(out_bam4) == bamio.bgzf_decompress(out_bam1)


@pytest.mark.parametrize("max_enum_snps", [0, 3, 12])
def test_enumeration_threshold_variants(engine_cls, orc, max_enum_snps):
    """max_enum_snps moves regions between the 2^S enumeration (12: a region with 11 SNPs = 2048 restarts) and the
    chain path (0 / 3: every region, phase.rs:1097-1123)."""
    b = synth.make_batch("ont-drna", n_genes=2, gene_len=12000, depth=35, seed=5)
    full_check(engine_cls, orc, b, _abi.make_params("ont-drna", seed=9, max_enum_snps=max_enum_snps))


def test_enumeration_of_more_than_65536_restarts(engine_cls, orc):
    """max_enum_snps = 17 on a 17-SNP region: 131 072 restarts -- the winner's index needs more than the 16 bits of
    k4_enum_resolve's list of maximal restarts.  Allele errors at the het sites make the restarts end in
    different optima, so the first maximum is not restart 0 by construction."""
    b, sites = helpers.two_haplotype_batch(n_snps=17, n_reads=16, seed=3)
    rng = np.random.default_rng(5)
    alt_of = {ord("A"): ord("C"), ord("C"): ord("A"), ord("G"): ord("T"), ord("T"): ord("G")}
    for k in range(b.n_reads):
        for x in sites[0]:
            if rng.random() < 0.2:   # toggle ref <-> alt: the sites stay biallelic
                i = int(b.seq_off[k]) + (x - int(b.pos[k]))
                r = int(b.ref[x - 5000])
                b.bases[i] = alt_of[r] if b.bases[i] == r else r
    c = full_check(engine_cls, orc, b, _abi.make_params("hifi-masseq", seed=9, max_enum_snps=17))
    assert len(c) == 17


def test_empty_batch_and_errors(engine_cls):
    from longcallr_amd.api import LcrError
    p = _abi.make_params()
    E = engine_cls(0, p)
    empty = helpers.mk_batch([], [])
    E.load_batch(empty).run_all()
    assert E.columns().shape == (_abi.NPLANES, 0) and len(E.candidates()[0]) == 0
    with pytest.raises(LcrError):  # call order
        engine_cls(0, p).fill_data_into_freq_vec()
    bad = helpers.mk_batch([dict(pos=10, seq="ACGT" * 5, cigar="10M2P10M")], [(0, "A" * 64)])
    with pytest.raises(LcrError, match="CIGAR"):  # unknown op: the reference panics (util.rs:944)
        engine_cls(0, p).load_batch(bad).fill_data_into_freq_vec()
    bad2 = helpers.mk_batch([dict(pos=10, seq="ACGT" * 5, cigar="10M")], [(0, "A" * 64)])  # l_seq 20 != 10
    with pytest.raises(LcrError, match="inconsistent"):
        engine_cls(0, p).load_batch(bad2).fill_data_into_freq_vec()
    # preconditions a foreign caller can violate are LCR_E_ARG, not silently wrong results: a region's reads sorted by pos
    unsorted = helpers.mk_batch([dict(pos=30, seq="ACGT" * 5, cigar="20M"), dict(pos=10, seq="ACGT" * 5, cigar="20M")], [(0, "A" * 64)])
    with pytest.raises(LcrError, match="sorted"):
        engine_cls(0, p).load_batch(unsorted).fill_data_into_freq_vec()
    with pytest.raises(LcrError, match="ld_weight_threshold"):
        engine_cls(0, _abi.make_params(ld_weight_threshold=2)).load_batch(helpers.demo_batch()).run_all()
    # CIGARs anywhere in the caller's array are fine (test_k0_op_space_layouts) -- beyond its end they are an argument error,
    # from host arrays and from device arrays alike
    two = helpers.mk_batch([dict(pos=10, seq="ACGT" * 5, cigar="20M"), dict(pos=12, seq="ACGT" * 5, cigar="10M1D10M")], [(0, "A" * 64)])
    oob = _with_reads(two, cig_off=np.array([0, 2], np.uint64))     # the second read's three ops end at 5 > n_cigar = 4
    with pytest.raises(LcrError, match="beyond n_cigar"):
        engine_cls(0, p).load_batch(oob)
    import torch
    import bench
    with pytest.raises(LcrError, match="beyond n_cigar"):
        engine_cls(0, p).load_batch(bench.to_device(oob, torch, torch.device("cuda", 0)))


def test_device_resident_inputs_and_idempotence(engine_cls):
    """LCR_MEM_DEVICE: torch tensors hand their HBM pointers to the ABI; two runs are identical."""
    import torch
    import ctypes as C
    b = synth.make_batch("masseq", n_genes=4, gene_len=6000, depth=25, seed=51)
    p = _abi.make_params("hifi-masseq")
    E = engine_cls(0, p)
    E.load_batch(b).run_all()
    ref_planes, ref_c = E.columns(), E.candidates()[0]
    dev = {f: torch.from_numpy(getattr(b, f).view(np.int64) if getattr(b, f).dtype == np.uint64 else
                               (getattr(b, f).view(np.int32) if getattr(b, f).dtype == np.uint32 else getattr(b, f))
                               ).pin_memory().cuda() for f in b.FIELDS + ["start0", "len", "col_off", "read_begin", "ref"]}
    torch.cuda.synchronize()
    reads, regions = b.c_reads(), b.c_regions()
    reads.mem = regions.mem = _abi.LCR_MEM_DEVICE
    for f in b.FIELDS:
        setattr(reads, f, C.c_void_p(dev[f].data_ptr()))
    for f in ["start0", "len", "col_off", "read_begin", "ref"]:
        setattr(regions, f, C.c_void_p(dev[f].data_ptr()))
    for _ in range(2):
        E2 = engine_cls(0, p)
        E2.load_batch((reads, regions, dev)).run_all()
        assert np.array_equal(E2.columns(), ref_planes)
        assert E2.candidates()[0].tobytes() == ref_c.tobytes()


@pytest.mark.gpu
def test_quality_histograms_from_the_tile_records(engine_cls, orc):
    """k2_hist_tiles (dense survivors: the histograms are taken from K0's per-tile records, u16 counters in LDS) against the walk
    over the reads and the oracle: forced on sparse ONT batches (lcr_debug_set hist_tiles), taken unasked by a deep island."""
    for prof, preset, seed in (("ont-cdna", "ont-cdna", 21), ("ont-drna", "ont-drna", 22)):
        b = synth.make_batch(prof, n_genes=4, gene_len=8000, depth=45, seed=seed)
        p = _abi.make_params(preset, seed=5)
        E = engine_cls(0, p)
        E.debug_set("hist_tiles", -1)
        E.load_batch(b).run_all()
        want = (E.candidates()[0].tobytes(), E.phase_result()["haplotag"].tobytes())
        E.debug_set("hist_tiles", 1)
        E.load_batch(b).run_all()
        assert (E.candidates()[0].tobytes(), E.phase_result()["haplotag"].tobytes()) == want, prof
        E.close()
    deep = synth.make_island("ont-drna-c5", n_loci=2, locus_len=6000, depth=400, seed=8)
    full_check(engine_cls, orc, deep, _abi.make_params("ont-drna", seed=5))    # (survivors on most covered columns: the tile form by itself)
    # HiFi presets keep the walk (the poly-A mask is not in the records), whatever the switch says
    b = synth.make_batch("masseq", n_genes=3, gene_len=7000, depth=35, seed=23)
    p = _abi.make_params("hifi-masseq", seed=5)
    E = engine_cls(0, p)
    E.load_batch(b).run_all(); want = E.candidates()[0].tobytes()
    E.debug_set("hist_tiles", 1)
    E.load_batch(b).run_all()
    assert E.candidates()[0].tobytes() == want
    E.close()


@pytest.mark.gpu
def test_async_input_path_double_buffered(engine_cls):
    """lcr_load_batch_async / lcr_bind_batch: batch i + 1 is uploaded into the other staging slot while batch i's stages run; the
    results are byte-identical to lcr_load_batch of the same host arrays, whatever the slot and the order; page-locked arrays."""
    from longcallr_amd import api
    p = _abi.make_params("ont-cdna", seed=9)
    batches = [synth.make_batch("ont-cdna", n_genes=3, gene_len=7000, depth=30, seed=60 + k) for k in range(3)]
    batches.append(synth.make_batch("ont-cdna", n_genes=1, gene_len=3000, depth=12, seed=70))

    def results(E):
        pr, fm = E.phase_result(), E.fragmat()
        return (E.columns().tobytes(), E.candidates()[0].tobytes(), fm["col"].tobytes(), fm["val"].tobytes(),
                pr["haplotag"].tobytes(), pr["phase_set"].tobytes())
    want = []
    E = engine_cls(0, p)
    for b in batches:
        E.load_batch(b).run_all()
        want.append(results(E))
    E.close()
    E = engine_cls(0, p)
    with pytest.raises(Exception, match="before lcr_load_batch_async"):
        E.bind_batch(1)
    with pytest.raises(Exception, match="slot must be 0 or 1"):
        E.load_batch_async(batches[0], 2)
    pinned = api.host_register(*[getattr(b, f) for b in batches for f in ("bases", "quals", "cigar")])
    try:
        E.load_batch_async(batches[0], 0)
        for k in range(len(batches)):
            E.bind_batch(k & 1)
            if k + 1 < len(batches):
                E.load_batch_async(batches[k + 1], (k + 1) & 1)     # crosses PCIe while this batch's kernels run
            E.run_all()
            assert results(E) == want[k], "batch %d through the asynchronous input path differs" % k
        # uploading into the BOUND slot invalidates the bound batch (the call waits for its kernels first)
        E.load_batch_async(batches[1], (len(batches) - 1) & 1)
        with pytest.raises(Exception, match="before lcr_load_batch"):
            E.fill_data_into_freq_vec()
        E.bind_batch((len(batches) - 1) & 1).run_all()
        assert results(E) == want[1]
        # the synchronous path still works on the same ctx afterwards
        E.load_batch(batches[2]).run_all()
        assert results(E) == want[2]
    finally:
        E.close()
        api.host_unregister(pinned)


def _pileup_properties(E, b):
    """plane sums equal the number of kept aligned bases / intron / deletion positions computed with plain numpy
    run-length arithmetic; fwd <= cnt"""
    pl = E.columns()
    ops, lens = b.cigar & 15, (b.cigar >> 4).astype(np.int64)
    assert int(pl[_abi.PL_N].sum()) == int(lens[ops == 3].sum())      # windows cover every read fully
    assert int(pl[_abi.PL_D].sum()) == int(lens[ops == 2].sum())
    assert int(pl[_abi.PL_NI].sum()) == int((ops == 1).sum())
    m_total = int(lens[np.isin(ops, [0, 7, 8])].sum())
    kept = int(pl[:4].sum())
    assert kept <= m_total and kept > 0.9 * m_total                     # ONT end-trim removes <= 2*20 per read
    assert np.all(pl[_abi.PL_FWD_A:_abi.PL_FWD_T + 1] <= pl[:4])
    assert np.all(pl[_abi.PL_TS_FWD] + pl[_abi.PL_TS_REV] == pl[:4].sum(axis=0))  # every read has ts, no non-ACGT
    return pl


def _result_bytes(E):
    pr = E.phase_result()
    return (E.candidates()[0].tobytes(), pr["haplotag"].tobytes(), pr["assignment"].tobytes(), pr["phase_set"].tobytes(),
            pr["objective"].tobytes())


def _phase_properties(E, max_enum_snps=10):
    """structural properties of the phase stage's outputs that hold at any size"""
    c, off = E.candidates()
    fm, pr = E.fragmat(), E.phase_result()
    assert set(np.unique(pr["haplotag"])) <= {-1, 0, 1} and set(np.unique(pr["assignment"])) <= {0, 1, 2}
    assert np.all((pr["assignment"] == 0) == (pr["haplotag"] == 0))           # snpfrags.rs:548-625: unassigned <=> tag 0
    assert np.all(np.where(pr["assignment"] == 1, pr["haplotag"] == 1, True)) and np.all(np.where(pr["assignment"] == 2, pr["haplotag"] == -1, True))
    assert np.all(np.isfinite(pr["objective"])) and np.all(pr["objective"] <= 0)
    for g in range(len(off) - 1):
        cg = c[off[g]:off[g + 1]]
        pos1 = set((cg["pos"] + 1).tolist())
        ps = cg["phase_set"][cg["phase_set"] != 0]
        assert set(ps.tolist()) <= pos1                                           # a phase set is named after one of its SNPs
        r0, r1 = fm["row_region_off"][g], fm["row_region_off"][g + 1]
        rps = pr["phase_set"][r0:r1]
        assert set(rps[rps != 0].tolist()) <= set(ps.tolist())
        assert np.all(pr["assignment"][r0:r1][rps != 0] != 0)                     # only assigned reads carry a phase set
        if len(cg) > max_enum_snps:                                               # LD blocks: disjoint, >= 2 SNPs, smallest first, descending
            blocks = E.ld_blocks(g)
            flat = [i for bl in blocks for i in bl]
            assert len(flat) == len(set(flat)) and all(len(bl) >= 2 and bl[0] == min(bl) for bl in blocks)
            assert [bl[0] for bl in blocks] == sorted([bl[0] for bl in blocks], reverse=True)
            assert all((cg["flags"][i] & _abi.F_FOR_PHASING) != 0 for i in flat)
    return c, off, fm, pr


FULL_SIZE_STATS = {}   # written to gpurun_out/full_size_parity.json (numbers quoted in DESIGN.md §2)


def _dump_full_size_stats():
    import json, os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        json.dump(FULL_SIZE_STATS, open(os.path.join(d, "full_size_parity.json"), "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def batch_check(E, O, b, params, upto="post", chrom="chrS"):
    """HIP results of a whole batch (engine E, stages already run up to `upto`) against the oracle pool's (orc.Batch O):
    whole-array compares of everything full_check compares per region."""
    pl = E.columns()
    assert np.array_equal(pl, O.planes()), "count planes"
    if upto == "pileup":
        return
    c, off = E.candidates()
    oc = O.cands()
    assert np.array_equal(off, O.cand_off), "candidates per region"
    for f in INT_FIELDS:
        assert np.array_equal(c[f], oc[f]), "cand.%s" % f
    assert np.array_equal(c["af1"], oc["af1"]) and np.array_equal(c["af2"], oc["af2"])
    assert close(c["loglik"], oc["loglik"], 1e-9) and close(c["gt_prob"], oc["gt_prob"], 1e-9)
    assert close(c["qual"], oc["qual"], 1e-9) and close(c["gq"], oc["gq"], 1e-9)
    assert np.array_equal(as_i32(c["qual"]), as_i32(oc["qual"])) and np.array_equal(as_i32(c["gq"]), as_i32(oc["gq"]))
    fm, of = E.fragmat(), O.fragmat()
    assert np.array_equal(fm["row_region_off"], of["row_region_off"]) and np.array_equal(fm["row_ptr"], of["row_ptr"])
    row_region = np.repeat(np.arange(b.n_regions), np.diff(fm["row_region_off"]))
    col_base = np.repeat(off[:-1][row_region], np.diff(fm["row_ptr"]))      # the ABI's col is batch-wide
    assert np.array_equal(fm["col"] - col_base, of["col"]) and np.array_equal(fm["val"], of["val"])
    for f in ("row_links", "row_for_phasing", "row_read"):
        assert np.array_equal(fm[f], of[f]), f
    if upto != "post":
        return
    assert np.all(np.abs(c["phase_score"] - oc["phase_score"]) <= 1e-4)
    pr, op = E.phase_result(), O.phase_result()
    for f in ("haplotag", "assignment", "phase_set"):
        assert np.array_equal(pr[f], op[f]), f
    assert np.array_equal(pr["objective"], op["objective"]), "fixed-point objective must match exactly"
    texts = O.vcf_texts(chrom)
    for g in range(b.n_regions):
        assert vcf.format_records(c[off[g]:off[g + 1]], chrom, params.min_phase_score) == texts[g], "VCF text region %d" % g
    chain = [g for g in range(b.n_regions) if off[g + 1] - off[g] > params.max_enum_snps]
    for g in chain[:64]:
        assert E.ld_blocks(g) == O.ld_blocks(g), "LD blocks region %d" % g
    return c, off, len(chain)


def tie_statistics(orc, E, b, params, O, name, chrom="chrS"):
    """DESIGN.md "Decision arithmetic": the batch once more through ORC_MODE_F64 (the reference's f64 ratio scores / sums in the
    reference's order at every decision).  O = the ORC_MODE_TIE run the HIP results were just compared with; the HIP results must
    equal the F64 run as well in every region without a tie of an unresolved class -- the census of both sides says which
    classes occurred -- and the numbers quoted in DESIGN.md are recorded: regions, ties per class, regions that differ."""
    A = orc.Batch(b, params, mode=orc.MODE_F64, keep_planes=False, fast=1)
    X = orc.Batch(b, params, mode=orc.MODE_EXACT_ONLY, keep_planes=False, fast=1)      # (the contract of rounds 1-3: what the tie handling changed)
    cf, ct = A.tie_census(), O.tie_census()
    S = np.diff(O.cand_off)
    chain = S > params.max_enum_snps
    unres = np.where(chain, cf[:, 1] + ct[:, 2] + cf[:, 7], 0)   # (the enumeration branch resolves all four classes since round 5)
    c, off = E.candidates()
    pr = E.phase_result()
    ta, tx = A.vcf_texts(chrom), X.vcf_texts(chrom)
    pa = A.phase_result()
    differ_f64 = differ_exact = 0
    for g in range(b.n_regions):
        r0, r1 = O.row_off[g], O.row_off[g + 1]
        hip_vcf = vcf.format_records(c[off[g]:off[g + 1]], chrom, params.min_phase_score)
        same = hip_vcf == ta[g] and all(np.array_equal(pr[f][r0:r1], pa[f][r0:r1]) for f in ("haplotag", "assignment", "phase_set"))
        if unres[g] == 0:
            assert same, "HIP differs from ORC_MODE_F64 in region %d although no tie of an unresolved class occurred" % g
        differ_f64 += not same
        differ_exact += hip_vcf != tx[g]
    hc = E.tie_census()
    en = ~chain
    # the enumeration kernels' census is the oracle's (the chain kernels at grid scope also count speculative half-rounds)
    assert hc["sigma_f64"] >= int(cf[en, 8].sum()) and hc["sigma_flips"] >= int(cf[en, 4].sum())
    # nothing fell to "a tie changes nothing" in the enumeration branch, and its tie-only steps / delta ties went through the repair pass
    assert hc["sigma_unresolved"] == 0 and hc["best_unresolved"] == 0 and hc["delta_step_f64"] >= int((cf[en, 1] + ct[en, 2]).sum())
    if not chain.any() or int((cf[chain, 1] + ct[chain, 2]).sum()) == 0:
        assert hc["delta_unresolved"] == 0 and hc["step_unresolved"] == 0
    FULL_SIZE_STATS[name] = dict(regions=int(b.n_regions), chain_regions=int(chain.sum()),
                                 regions_where_hip_differs_from_f64=int(differ_f64), regions_where_hip_differs_from_round3_fixed_point=int(differ_exact),
                                 regions_with_a_tie_of_an_unresolved_class=int((unres > 0).sum()),
                                 oracle_sigma_ties_at_rows_with_a_het_entry=int(cf[:, 8].sum()), oracle_sigma_ties_flipped_by_f64=int(cf[:, 4].sum()),
                                 oracle_delta_eta_ties=int(cf[:, 1].sum()), oracle_tie_only_steps=int(ct[:, 2].sum()),
                                 oracle_equal_objective_compares=int(cf[:, 3].sum()), oracle_equal_objective_f64_greater=int(cf[:, 7].sum()),
                                 hip_census=hc, vcf_records=int(sum(len(t.splitlines()) for t in ta)),
                                 oracle_f64_seconds=A.seconds, oracle_tie_seconds=O.seconds, oracle_threads=int(A.threads))
    _dump_full_size_stats()
    A.close(); X.close()
    return FULL_SIZE_STATS[name]


def test_c3_full_size_against_the_oracle(engine_cls, orc):
    """BASELINE configs[2] at its full size (C3 as bench.py builds it: 400 distinct genes x 25 kb, 40x ONT-cDNA,
    4.6 10^8 aligned bases): every region through the oracle on a native thread pool (thread.rs:77) and the HIP path
    compared with it at full_check level -- planes, candidates, fragment matrix, sigma / delta / eta, objective, phase
    sets, VCF text, LD blocks -- plus determinism across two runs and the tie statistics of the F64 arithmetic."""
    import bench
    b = bench.build_workload("c3")
    assert b.n_regions == 400 and b.bases.size > 4.0e8
    p = _abi.make_params("ont-cdna", seed=2025)
    O = orc.Batch(b, p, mode=orc.MODE_TIE, fast=1, tie_mask=orc.TIE_MASK_LIBLCR)
    E = engine_cls(0, p)
    E.load_batch(b).fill_data_into_freq_vec()
    pl = _pileup_properties(E, b)
    E.get_candidate_snps().get_fragments().phase()
    _phase_properties(E)
    c, off, n_chain = batch_check(E, O, b, p)
    r1 = _result_bytes(E)
    E.load_batch(b).run_all()
    assert np.array_equal(E.columns(), pl) and _result_bytes(E) == r1
    st = tie_statistics(orc, E, b, p, O, "c3")
    assert st["regions_where_hip_differs_from_f64"] == 0       # BASELINE configs[2]: the reference's arithmetic in all 400 regions
    E.close()
    st["candidates"] = int(c.size)
    _dump_full_size_stats()
    O.close()


def test_c4_share_against_the_oracle(engine_cls, orc):
    """BASELINE configs[3]: one GPU's share of C4 as bench.py builds it (1 000 MAS-Seq regions x 25 kb at 60x,
    1.6 10^9 aligned bases, hifi-masseq preset: poly-A mask, no end trim) against the oracle pool, full_check level."""
    import bench
    b = bench.build_workload("c4")
    assert b.n_regions == 1000 and b.bases.size > 1.5e9
    p = _abi.make_params("hifi-masseq", seed=2025)
    O = orc.Batch(b, p, mode=orc.MODE_TIE, keep_planes=True, fast=1, tie_mask=orc.TIE_MASK_LIBLCR)
    E = engine_cls(0, p)
    E.load_batch(b).run_all()
    _phase_properties(E)
    c, off, n_chain = batch_check(E, O, b, p)
    st = tie_statistics(orc, E, b, p, O, "c4_share")
    assert st["regions_where_hip_differs_from_f64"] == 0       # one GPU's share of BASELINE configs[3]: all 1 000 regions
    E.close()
    st["candidates"] = int(c.size)
    _dump_full_size_stats()
    O.close()


def test_c5_island_against_the_oracle(engine_cls, orc):
    """BASELINE configs[4] at a size the oracle finishes (4 loci = 100 kb x 200x as ONE region, ~480 candidates, chain
    path): the region is staged, phased (LD blocks, block flip, 2 x 120 perturbation calls) and post-processed by the
    all-CUs kernels (k4_stage_grid, k4_chain_grid with the device-coherent rounds, k4_gpost) -- bit-exact like every
    other region."""
    b = synth.make_island("ont-drna-c5", n_loci=4, locus_len=25000, depth=200, seed=3)
    assert b.n_regions == 1
    c = full_check(engine_cls, orc, b, _abi.make_params("ont-drna", seed=11))
    assert c.size > 400


def test_c5_scopes_and_paths_agree(engine_cls, monkeypatch):
    """One island (200 kb x 150x) through the forms of the chain kernel -- all CUs with eight speculative half-rounds per pass
    over the matrix (default), with the half-rounds side by side on sub-grids (8, 1, 4, 16 lanes), all CUs with fenced
    barriers only (LCR_GRID_GENERIC), one workgroup -- and through the host epilogue: identical bytes."""
    b = synth.make_island("ont-drna-c5", n_loci=8, locus_len=25000, depth=150, seed=4)
    p = _abi.make_params("ont-drna", seed=12)

    def run():
        E = engine_cls(0, p)
        E.load_batch(b).run_all()
        _phase_properties(E)
        r = _result_bytes(E) + (repr(E.ld_blocks(0)),)
        E.close()
        return r
    ref = run()      # (default: eight speculative half-rounds per pass over the matrix, k4_grid_batch.h)
    monkeypatch.setenv("LCR_GRID_SPEC_BATCH", "0")   # the half-rounds side by side, each on an eighth of the workgroups -- one XCD
    assert run() == ref, "speculative rounds side by side"
    for lanes in ("1", "4", "16"):   # one half-round after the other; other batch widths: same commits, same bytes
        monkeypatch.setenv("LCR_GRID_SPEC_LANES", lanes)
        assert run() == ref, "speculative rounds with %s lanes" % lanes
    monkeypatch.delenv("LCR_GRID_SPEC_LANES")
    monkeypatch.delenv("LCR_GRID_SPEC_BATCH")
    monkeypatch.setenv("LCR_GRID_GENERIC", "1")
    assert run() == ref
    monkeypatch.delenv("LCR_GRID_GENERIC")
    monkeypatch.setenv("LCR_GRID_MIN_ENTRIES", "1000000000")
    assert run() == ref
    monkeypatch.delenv("LCR_GRID_MIN_ENTRIES")
    monkeypatch.setenv("LCR_POST_HOST", "1")
    got = run()
    # (the host epilogue's phase_score goes through the host libm's log10: compare everything but the last bits of it)
    assert got[1:] == ref[1:]
    ca, cb = np.frombuffer(got[0], dtype=_abi.CAND_DTYPE), np.frombuffer(ref[0], dtype=_abi.CAND_DTYPE)
    for f in INT_FIELDS:
        assert np.array_equal(ca[f], cb[f]), f
    assert np.all(np.abs(ca["phase_score"] - cb["phase_score"]) <= 1e-9)


def test_c5_full_size(engine_cls, orc):
    """BASELINE configs[4] at full size: ONE region of ~1 Mb at ~500x ONT-dRNA (3.3 10^5 reads, ~4 700 candidate sites,
    8 10^6 matrix entries, 2 345 cross_optimize calls).  Pileup planes, candidates and the fragment matrix (P1-P6) against a
    live oracle run; the phase stage (LD blocks, LD-seeded start, block flip, 1 172 perturbation rounds: phase.rs:1123-1233) and
    the post-phase steps against the oracle's results for the same input kept in tests/golden/c5_full_size_oracle.npz (the oracle
    with indexed gathers and threaded Jacobi steps needs 3.5 minutes for it; make_c5_golden.py; LCR_C5_ORACLE=1 runs it live):
    sigma of every row, assignments, phase sets, every candidate field, the VCF text, the LD blocks, the objective exactly."""
    import hashlib, importlib.util, os
    spec = importlib.util.spec_from_file_location("make_c5_golden", os.path.join(helpers.GOLDEN, "make_c5_golden.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    b, p = mk.build()
    assert b.n_regions == 1 and b.len[0] > 900000 and b.bases.size > 5.0e8
    G = np.load(os.path.join(helpers.GOLDEN, "c5_full_size_oracle.npz"))
    assert mk.input_digest(b) == G["input_sha256"].tobytes().decode(), "synth.make_island no longer builds the input the fixture was made from"
    E = engine_cls(0, p)
    E.load_batch(b).fill_data_into_freq_vec()
    pl = _pileup_properties(E, b)
    covered = (pl[:4].sum(axis=0) + pl[_abi.PL_N] + pl[_abi.PL_D]) > 0
    assert covered[100:-100].all()                                            # one coverage island (read ends are trimmed)
    E.get_candidate_snps().get_fragments()
    # P1-P6 of the one region against the oracle (one thread: the reference's unit of parallelism is the region): planes,
    # candidate records before phasing, the whole fragment matrix
    O = orc.Batch(b, p, mode=orc.MODE_EXACT_ONLY, upto="frag")
    batch_check(E, O, b, p, upto="frag")
    FULL_SIZE_STATS["c5_p1_p6"] = dict(oracle_seconds=O.seconds, oracle_threads=int(O.threads), candidates=int(O.cand_off[-1]), fragment_nnz=int(O.nnz_off[-1]))
    O.close()
    E.phase()
    c, off, fm, pr = _phase_properties(E)
    assert 4000 < c.size < 6500 and fm["col"].size > 5e6
    # ---- the phase stage and the post-phase steps against the oracle's results
    if os.environ.get("LCR_C5_ORACLE"):
        R, secs = mk.run_oracle(b, p, int(os.environ.get("LCR_C5_ORACLE_THREADS", "64")))
        op, oc, otext, oblocks = R.phase_result(), R.cands(), R.vcf_text("chrS"), R.ld_blocks()
        oobj = op["objective"]
    else:
        op = {f: G[f] for f in ("haplotag", "assignment", "phase_set")}
        oc = G["cands"].view(_abi.CAND_DTYPE)
        otext, oobj = G["vcf"].tobytes().decode(), float(G["objective"])
        oblocks = [G["ld_snps"][G["ld_off"][k]:G["ld_off"][k + 1]].tolist() for k in range(G["ld_off"].size - 1)]
    for f in ("haplotag", "assignment", "phase_set"):
        assert np.array_equal(pr[f], op[f]), f
    assert pr["objective"][0] == oobj, "fixed-point objective must match exactly"
    for f in INT_FIELDS:
        if f != "region":
            assert np.array_equal(c[f], oc[f]), "cand.%s" % f
    assert np.all(np.abs(c["phase_score"] - oc["phase_score"]) <= 1e-4)
    assert vcf.format_records(c, "chrS", p.min_phase_score) == otext
    assert E.ld_blocks(0) == oblocks
    # ---- and against ORC_MODE_F64 at full size: the reference's f64 scores / sums in the reference's order at EVERY decision
    # of the island's 2 345 cross_optimize calls -- 416 s of the oracle on 200 threads, kept as digests (tests/golden/c5_full_size_oracle_f64.json,
    # make_c5_golden.py --mode f64).  The HIP path's outputs hash to the same values; the f64 objective is the fixed-point one within 1e-4.
    import json
    F = json.load(open(os.path.join(helpers.GOLDEN, "c5_full_size_oracle_f64.json")))
    assert F["input_sha256"] == G["input_sha256"].tobytes().decode() and F["mode"] == "ORC_MODE_F64"
    hd = mk.digests(pr, c, vcf.format_records(c, "chrS", p.min_phase_score), E.ld_blocks(0))
    assert hd == F["digests"], {k: hd[k] == F["digests"][k] for k in hd}
    assert abs(pr["objective"][0] - F["objective_f64"]) < 1e-4 and abs(float(np.sum(c["phase_score"])) - F["phase_score_sum"]) < 1e-4 * c.size
    assert F["tie_census"][1] == 0 and F["tie_census"][2] == 0 and F["tie_census"][7] == 0   # the oracle met no delta / eta tie, no tie-only step, no later equal-objective configuration with the greater f64 sum
    hc = E.tie_census()
    assert hc["sigma_unresolved"] == 0 and hc["delta_unresolved"] == 0 and hc["sigma_f64"] >= int(G["tie_census"][8])
    FULL_SIZE_STATS["c5_phase"] = dict(oracle_seconds=float(G["oracle_seconds"]), oracle_threads=int(G["oracle_threads"]), cross_optimize_calls=int(G["stats"][0]),
                                       iterations=int(G["stats"][1]), oracle_sigma_ties_at_rows_with_a_het_entry=int(G["tie_census"][8]),
                                       oracle_sigma_ties_flipped_by_f64=int(G["tie_census"][4]), oracle_equal_objective_compares=int(G["tie_census"][3]),
                                       oracle_equal_objective_f64_greater=int(G["tie_census"][7]), hip_census=hc, live_oracle=bool(os.environ.get("LCR_C5_ORACLE")))
    _dump_full_size_stats()
    fp = (c["flags"] & _abi.F_FOR_PHASING) != 0
    assert fp.sum() > 2000 and (pr["assignment"] != 0).mean() > 0.95            # the reads carry real haplotype signal
    r1 = _result_bytes(E)
    E.load_batch(b).run_all()
    assert np.array_equal(E.columns(), pl) and _result_bytes(E) == r1


@pytest.mark.parametrize("grid_min", [None, "0"])
def test_two_ranks_share_one_gpu(grid_min):
    """SURVEY §8(e) on the one GPU of the test box: two processes (torch.distributed.run, gloo), each on its
    shard.assign_regions share of ONE list of distinct MAS-Seq genes (bench.py's N > 1 construction; ONT-dRNA genes in
    the second case), candidate and read records gathered to rank 0 == the single-process run.  With
    LCR_GRID_MIN_ENTRIES=0 both processes launch persistent all-CU kernels at the same time: the device-wide lock
    (k4_phase.hip GridLock) keeps them from waiting for each other's workgroups."""
    import os
    import subprocess
    import sys
    env = dict(os.environ)
    if grid_min is not None:
        env["LCR_GRID_MIN_ENTRIES"] = grid_min
        env["LCR_TEST_SHARD_PROFILE"] = "ont-drna"   # chain regions: every one of them becomes a persistent launch
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_shard_worker.py")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(29000 + os.getpid() % 2000), worker]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SHARD-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def _torchrun(args, env=None, timeout=600, nproc=1):
    import os
    import subprocess
    import sys
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(29000 + (os.getpid() * 7 + len(args[0])) % 2000)] + list(args)
    return subprocess.run(cmd, env=env or dict(os.environ), capture_output=True, text=True, timeout=timeout)


def test_nccl_group_gathers_device_records():
    """The first RCCL run made boring: a world-size-1 `nccl` process group on the test GPU drives
    shard.RecordGather with device-pointer records for three overlapped batches == the host getters, byte for byte
    (tests/dist_nccl_worker.py; thread.rs:204-221)."""
    import os
    r = _torchrun([os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_nccl_worker.py")])
    assert r.returncode == 0 and "NCCL-GATHER-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_under_the_launcher_with_one_rank():
    """`bench.py --gpus 1` under torch.distributed.run, as the driver launches N > 1: the nccl group, the device-pointer
    gathers of every step, the world-size / backend asserts and the per-rank block of the JSON line all execute."""
    import json
    import os
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    r = _torchrun([bench, "--gpus", "1", "--steps", "3", "--warmup", "1", "--prewarm", "2", "--quick", "--genes", "24", "--gene-len", "12000"])
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    pr = line["config"]["per_rank"]
    assert pr["backend"] == "nccl" and pr["world_size"] == 1 and len(pr["ms_per_step"]) == 1
    assert pr["gather_bytes_per_step"][0] > 0 and line["config"]["gathered_records_last_batch"]["candidates"] > 0
    assert line["n_gpus"] == 1 and line["value"] > 0


def test_bench_dry_run_at_eight_ranks():
    """The first real 8-GPU line must not fail on shard construction or capacity negotiation (thread.rs:77,204-221):
    `bench.py --gpus 8 --dist-backend gloo --quick` -- EIGHT ranks sharing the one GPU, one list of 8 x 16 distinct MAS-Seq genes through
    shard.assign_regions (LPT on len x max_coverage), both RecordGathers every step, the per-rank block with eight entries -- and the records
    gathered on rank 0 == what ONE rank gathers over the whole list (same genes: the list depends on world x genes only)."""
    import json
    import os
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    common = ["--dist-backend", "gloo", "--quick", "--workload", "c4", "--steps", "2", "--warmup", "1", "--prewarm", "1", "--gene-len", "12000"]
    r8 = _torchrun([bench, "--gpus", "8", "--genes", "16"] + common, nproc=8, timeout=900)
    assert r8.returncode == 0, r8.stdout[-2000:] + r8.stderr[-4000:]
    l8 = json.loads([x for x in r8.stdout.splitlines() if x.startswith("{")][-1])
    pr = l8["config"]["per_rank"]
    assert l8["n_gpus"] == 8 and pr["world_size"] == 8 and pr["backend"] == "gloo"
    for k in ("ms_per_step", "columns", "aligned_bases", "reads", "regions", "lpt_cost", "gather_wait_ms_per_step", "gather_bytes_per_step"):
        assert len(pr[k]) == 8, k
    assert sum(pr["regions"]) == 128 and min(pr["regions"]) >= 1 and min(pr["gather_bytes_per_step"]) > 0
    assert pr["lpt_cost_imbalance_max_over_mean"] < 1.5      # LPT over 128 genes on 8 ranks
    g8 = l8["config"]["gathered_records_last_batch"]
    assert g8["candidates"] == l8["config"]["candidates"] > 0 and g8["reads"] > 0
    r1 = _torchrun([bench, "--gpus", "1", "--genes", "128"] + common, nproc=1, timeout=900)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-4000:]
    l1 = json.loads([x for x in r1.stdout.splitlines() if x.startswith("{")][-1])
    g1 = l1["config"]["gathered_records_last_batch"]
    assert g1 == g8, (g1, g8)
    for k in ("columns", "aligned_bases", "reads", "candidates", "fragment_nnz", "phasing_reads"):
        assert l1["config"][k] == l8["config"][k], k


def test_asynchronous_phase_stage(engine_cls):
    """lcr_ctx_set_async_phase(ctx, 1): lcr_phase returns with its kernels in flight on the stage's own queues; the next batch is
    bound and its pileup queued behind the stage's restarts (beside its tails), every getter / the next lcr_candidates collects the
    results first.  Three batches
    back to back (no getter in between), then getters in every order == the synchronous engine, byte for byte."""
    p = _abi.make_params("ont-cdna", seed=77)
    bs = [synth.make_batch("ont-cdna", n_genes=n, gene_len=10000, depth=35, seed=60 + k) for k, n in enumerate((5, 3, 6))]
    import torch
    dev = torch.device("cuda", 0)
    import bench as _bench
    dv = [_bench.to_device(b, torch, dev) for b in bs]   # device-resident inputs: the next bind does not wait for the phase stage
    Es, Ea = engine_cls(0, p), engine_cls(0, p)
    Ea.set_async_phase(True)
    want = []
    for d in dv:
        Es.load_batch(d).run_all()
        want.append((_result_bytes(Es), Es.columns().copy()))
    for rep in range(2):
        for k, d in enumerate(dv):      # back to back: batch k + 1's load + pileup are queued while batch k's phase stage runs
            Ea.load_batch(d).run_all()
            if rep == 1:
                if k == 0:
                    pr = Ea.phase_result(); c = Ea.candidates()[0]
                elif k == 1:
                    c = Ea.candidates()[0]; pr = Ea.phase_result()
                assert _result_bytes(Ea) == want[k][0], "async batch %d" % k
                assert np.array_equal(Ea.columns(), want[k][1])
        assert _result_bytes(Ea) == want[-1][0]
    # host batches: lcr_load_batch waits for the stage in flight (it rewrites the staging buffers)
    for k, b in enumerate(bs):
        Ea.load_batch(b).run_all()
    assert _result_bytes(Ea) == want[-1][0]
    Es.close(); Ea.close()


def test_pipelined_collect_of_the_asynchronous_stage(engine_cls):
    """with the asynchronous stage the per-batch getters answer LCR_E_STATE once the next batch is bound -- lcr_collect_phase is
    the getter that outlives the binding.  Order of a pipelined caller: phase(k); load(k + 1); pileup(k + 1); collect (-> k); candidates(k + 1) ...
    Every batch's collected candidates, per-row results, objectives and both HBM record arrays == the synchronous engine's, byte for byte;
    the per-batch getters refuse after the binding; collect refuses after the next lcr_candidates."""
    import torch
    import bench as _bench
    p = _abi.make_params("ont-cdna", seed=77)
    bs = [synth.make_batch("ont-cdna", n_genes=n, gene_len=10000, depth=35, seed=160 + k) for k, n in enumerate((4, 6, 3, 5))]
    dev = torch.device("cuda", 0)
    dv = [_bench.to_device(b, torch, dev) for b in bs]
    Es = engine_cls(0, p)
    want = []
    for d in dv:
        Es.load_batch(d).run_all()
        c, off = Es.candidates(); pr = Es.phase_result(); fm = Es.fragmat()
        pc, nc = Es.candidates_device(); prr, nr = Es.read_records_device()
        want.append(dict(cand=c.tobytes(), off=off.tobytes(), rro=fm["row_region_off"].tobytes(), tag=pr["haplotag"].tobytes(), asg=pr["assignment"].tobytes(),
                         ps=pr["phase_set"].tobytes(), obj=pr["objective"].tobytes(), n=(nc, nr)))
    Es.close()

    def dev_bytes(ptr_n, itemsize):
        ptr, n = ptr_n
        out = torch.empty(n * itemsize, dtype=torch.uint8, device=dev)
        if n:
            import ctypes as C
            hip = C.CDLL("libamdhip64.so")
            assert hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(ptr), C.c_size_t(n * itemsize), 3) == 0
        return out.cpu().numpy().tobytes()

    import os
    for async_on in (True, False):
        Ea = engine_cls(0, p)
        Ea.set_async_phase(async_on)
        if async_on:   # LCR_W_HW_QUEUES: the mode is on either way, the library says when the process has too few hardware queues for it to pay
            assert Ea.async_warning is None or "GPU_MAX_HW_QUEUES" in Ea.async_warning
            keep_env = os.environ.get("GPU_MAX_HW_QUEUES")
            os.environ["GPU_MAX_HW_QUEUES"] = "4"
            assert "GPU_MAX_HW_QUEUES" in (Ea.set_async_phase(True).async_warning or "")
            os.environ["GPU_MAX_HW_QUEUES"] = "8"
            assert Ea.set_async_phase(True).async_warning is None
            if keep_env is None:
                del os.environ["GPU_MAX_HW_QUEUES"]
            else:
                os.environ["GPU_MAX_HW_QUEUES"] = keep_env
        Ea.load_batch(dv[0]).run_all()
        for k in range(1, len(dv) + 1):
            if k < len(dv):
                Ea.load_batch(dv[k]).fill_data_into_freq_vec()
                with pytest.raises(api.LcrError):      # the bound batch has no phase results: the per-batch getters refuse
                    Ea.phase_result()
            r = Ea.collect_phase(copy=True)            # batch k - 1
            w = want[k - 1]
            assert r["cand"].tobytes() == w["cand"] and r["cand_region_off"].tobytes() == w["off"] and r["row_region_off"].tobytes() == w["rro"], k
            assert r["haplotag"].tobytes() == w["tag"] and r["assignment"].tobytes() == w["asg"] and r["phase_set"].tobytes() == w["ps"] and r["objective"].tobytes() == w["obj"], k
            assert (r["dev_cand"][1], r["dev_read_rec"][1]) == w["n"]
            assert dev_bytes(r["dev_cand"], _abi.CAND_DTYPE.itemsize) == w["cand"]
            rec = np.frombuffer(dev_bytes(r["dev_read_rec"], 12), dtype=_abi.READ_REC_DTYPE)
            assert rec["haplotag"].tobytes() == w["tag"] and rec["assignment"].tobytes() == w["asg"] and rec["phase_set"].tobytes() == w["ps"]
            if k < len(dv):
                Ea.get_candidate_snps()
                with pytest.raises(api.LcrError):      # its buffers are being rewritten
                    Ea.collect_phase()
                Ea.get_fragments().phase()
        Ea.close()


def test_region_discovery_gpu(engine_cls):
    """lcr_discover_regions vs the loop-for-loop restatement of util.rs:236-332."""
    import os
    from longcallr_amd import bamio
    from oracle import oracle_np
    E = engine_cls(0, _abi.make_params())
    spans = [(5, 6), (10, 20), (12, 18), (30, 31), (40, 41), (50, 60), (70, 71), (1023, 1025), (2047, 2049), (3000, 3001)]
    assert E.discover_regions([s for s, _ in spans], [e for _, e in spans], 4096) == oracle_np.discover_regions(spans, 4096)
    assert E.discover_regions([], [], 1000) == [] and E.discover_regions([0], [10], 10) == [(0, 10, 1)]
    rng = np.random.default_rng(3)
    st = np.sort(rng.integers(0, 200000, size=3000)); ln = rng.integers(1, 400, size=3000)
    spans = list(zip(st.tolist(), (st + ln).tolist()))
    assert E.discover_regions(st, st + ln, 200100) == oracle_np.discover_regions(spans, 200100)
    # demo.bam (whole chr20 depth vector: 64 M positions)
    refs, recs = bamio.read_bam(os.path.join(helpers.GOLDEN, "demo.bam"))
    keep = [r for r in recs if bamio.passes_filter(r)]
    rs = np.array([r["pos"] for r in keep]); re_ = rs + np.array([max(r["ref_len"], 1) for r in keep])
    assert E.discover_regions(rs, re_, 64444167) == bamio.discover_regions(keep, keep[0]["ref_id"], 64444167) == [(16729960, 13256, 1649)]


def test_one_context_over_batches_of_different_sizes(engine_cls):
    """A context keeps its buffers (and their capacities) across batches: one that sees a small batch, a ten times
    larger one and the small one again gives the bytes a fresh context gives for each."""
    small = synth.make_batch("ont-cdna", n_genes=1, gene_len=6000, depth=25, seed=71)
    large = synth.make_batch("ont-drna", n_genes=6, gene_len=20000, depth=60, seed=72)
    p = _abi.make_params("ont-cdna", seed=5)

    def fresh(b):
        E = engine_cls(0, p)
        E.load_batch(b).run_all()
        r = _result_bytes(E) + (E.fragmat()["col"].tobytes(),)
        E.close()
        return r
    want = {id(small): fresh(small), id(large): fresh(large)}
    E = engine_cls(0, p)
    for b in (small, large, small, large, large, small):
        E.load_batch(b).run_all()
        assert _result_bytes(E) + (E.fragmat()["col"].tobytes(),) == want[id(b)]
    E.close()


@pytest.mark.parametrize("dist_to_end,polya_len", [(40, 5), (63, 16), (100, 7), (10, 3), (1, 1), (300, 5), (40, 24), (2000, 31)])
def test_poly_a_mask_zone_sizes(engine_cls, orc, dist_to_end, polya_len):
    """The HiFi presets' poly-A / homopolymer mask (util.rs:754-789) for zones of several widths and window lengths: up to
    63 offsets a thread per read end corrects K1's counts (k1_zonefix_ends), wider zones and windows beyond 16 bases take a
    thread per offset (k1_zonefix: any width, the reference has no limit); MAS-Seq reads with aligned and soft-clipped poly-A tails, planes bit-exact against the oracle."""
    b = synth.make_batch("masseq", n_genes=2, gene_len=8000, depth=40, seed=53)
    p = _abi.make_params("hifi-masseq", seed=3, dist_to_end=dist_to_end, polya_len=polya_len)
    regs = oracle_all(orc, b, p)
    E = engine_cls(0, p)
    E.load_batch(b).fill_data_into_freq_vec()
    check_pileup(E, regs, b)
    E.close()
    if dist_to_end >= 300:      # (the same width as an ONT end trim, util.rs:745-751: K0 clips the M blocks)
        p2 = _abi.make_params("ont-cdna", seed=3, dist_to_end=dist_to_end, polya_len=polya_len)
        E2 = engine_cls(0, p2)
        E2.load_batch(b).fill_data_into_freq_vec()
        check_pileup(E2, oracle_all(orc, b, p2, upto="pileup"), b)
        E2.close()


def test_bench_line_contract():
    """`python bench.py` prints ONE JSON line with the fields the driver reads: metric / value / unit / n_gpus / steps /
    warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype / data / config.workload, a `roofline` object
    (bound, achieved, peak, unit, frac = achieved / peak, traffic) and -- without --no-cpu-baseline -- `cpu_baseline`."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--prewarm", "1", "--no-c5",
                          "--cpu-budget", "2"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "stages"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert 0.0 < r["frac"] < 1.0 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert abs(d["value"] - d["config"]["columns"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6


def test_bench_several_batches_in_flight():
    """`bench.py --inflight 2`: two contexts on two host threads share the GPU; the line keeps its contract and says so."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "2", "--prewarm", "2", "--no-c5",
                          "--no-cpu-baseline", "--no-traffic", "--inflight", "2"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["config"]["batches_in_flight_per_gpu"] == 2 and d["steps"] == 4 and d["value"] > 0 and "cpu_baseline" not in d
