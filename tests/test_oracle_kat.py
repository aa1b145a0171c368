"""Hand-derived known-answer tests pinning the CPU oracle (the reference has no tests of its own;
every expected value below is computed by hand / closed form from the cited Rust lines)."""
import math

import numpy as np
import pytest

import helpers
from longcallr_amd import _abi


def test_sor_threshold(orc):
    # candidate.rs:24-35,49-51: ln(12/60 + 60/12) + ln(6/6) - ln(2/10) = ln 5.2 + ln 5
    v = orc.lib().orc_strand_odds_ratio(5, 5, 9, 1)
    assert abs(v - (math.log(5.2) + math.log(5.0))) < 2e-6
    assert orc.lib().orc_strand_odds_ratio(5, 5, 1, 9) == v  # symmetric in the alt strands
    assert orc.lib().orc_strand_odds_ratio(10, 10, 5, 5) == pytest.approx(math.log(2.0), abs=1e-6)


def test_binomial_two_tailed(orc):
    f = orc.lib().orc_binomial_two_tailed
    assert f(0, 5) == pytest.approx(2 * 0.5 ** 5)            # 2*cdf(0)
    assert f(5, 5) == pytest.approx(2 * 0.5 ** 5)            # 2*(1-cdf(n-1))
    assert f(2, 4) == pytest.approx(2 * min(11 / 16, 1 - 5 / 16))
    assert f(1, 10) < 0.05 < f(2, 10)                        # 2*11/1024 = 0.0215 ; 2*56/1024 = 0.109
    assert f(9, 30) < 0.05 < f(10, 30)                       # the decision boundary at n = 30


def test_two_major_alleles_ties(orc):
    def tm(cnt, ref):
        c = np.array(cnt, dtype=np.uint32)
        a1, a2 = np.zeros(1, np.uint8), np.zeros(1, np.uint8)
        c1, c2 = np.zeros(1, np.uint32), np.zeros(1, np.uint32)
        orc.lib().orc_two_major_alleles(c.ctypes.data, ord(ref), a1.ctypes.data, c1.ctypes.data, a2.ctypes.data, c2.ctypes.data)
        return chr(a1[0]), int(c1[0]), chr(a2[0]), int(c2[0])
    assert tm([5, 3, 3, 0], "G") == ("A", 5, "G", 3)   # tie for 2nd: reference allele preferred (util.rs:166-167)
    assert tm([5, 3, 0, 3], "T") == ("A", 5, "T", 3)   # ... also when it sorts 4th (util.rs:168-169)
    assert tm([5, 3, 3, 0], "T") == ("A", 5, "C", 3)   # tie, ref not among them: stable order A<C<G<T
    assert tm([2, 2, 2, 2], "G") == ("A", 2, "G", 2)
    assert tm([0, 7, 0, 9], "C") == ("T", 9, "C", 7)


def test_probability_functions(orc):
    L = orc.lib()
    assert L.orc_aki(1, 1, 0, 1, 0.01) == 0.99 and L.orc_aki(1, -1, 0, 1, 0.01) == 0.01
    assert L.orc_aki(-1, 1, 1, 1, 0.2) == 0.8          # eta = +1: x = eta regardless of sigma*delta
    arr = lambda x, t: np.array(x, dtype=t)
    delta, eta, ps, pr = arr([1, -1], np.int32), arr([0, 0], np.int32), arr([1, 1], np.int32), arr([0.001, 0.1], np.float64)
    # read (sigma=1): site0 matches (1-1e-3), site1 mismatches (0.1); flipped: 1e-3 and 0.9
    a = math.log10(0.999) + math.log10(0.1); b = math.log10(0.001) + math.log10(0.9)
    got = L.orc_cal_sigma_delta_eta_log(1, 2, delta.ctypes.data, eta.ctypes.data, ps.ctypes.data, pr.ctypes.data)
    assert got == pytest.approx(1 - a / (a + b), rel=1e-14)
    sigma = arr([1, 1, -1], np.int32); ps3 = arr([1, 1, -1], np.int32); pr3 = arr([0.001] * 3, np.float64)
    # SNP column, delta=1, eta=0: all three observations consistent -> 3*log10(.999)
    l_ok, l_bad = 3 * math.log10(0.999), 3 * math.log10(0.001)
    het = math.log10(0.001) - 3 * math.log10(2)
    hr, hv = math.log10(0.9985), math.log10(0.0005)
    l_ref = 2 * math.log10(0.999) + math.log10(0.001)   # eta=+1: p must be +1
    l_var = 2 * math.log10(0.001) + math.log10(0.999)
    want = 1 - (l_ok + het) / ((l_var + hv) + (l_ok + het) + (l_ref + hr) + (l_bad + het))
    got = L.orc_cal_delta_eta_sigma_log(1, 0, 3, sigma.ctypes.data, ps3.ctypes.data, pr3.ctypes.data)
    assert got == pytest.approx(want, rel=1e-13)
    got = L.orc_cal_phase_score_log(1, 0, 3, sigma.ctypes.data, ps3.ctypes.data, pr3.ctypes.data)
    assert got == pytest.approx(1 - l_ok / (l_ok + l_bad), rel=1e-13)


REF = "ACGTTGCAAGGCTTACGATCGGATCCTAGCATGCAAGTCGATCGTAGCTAGCTAGGATCGATCGATTACGGCTAGCTAGGCTAAGCTTAGC" * 8


def _col(planes, i):
    return {n: int(planes[k][i]) for k, n in enumerate(_abi.PLANE_NAMES)}


def test_pileup_cigar_semantics(orc):
    """util.rs:692-947: window clipping, D / N runs, insertion on the previous column, soft/hard clips."""
    ref = REF[:200]
    rd = [dict(pos=95, seq=REF[95:105] + "TT" + REF[108:128], qual=30, cigar="2H10M2I3D20M4S"[:0] + "10M2I3D20M", ts=1),
          dict(pos=100, seq="GGG" + REF[100:110] + REF[160:170], qual=30, cigar="3I10M50N10M", rev=1, ts=1),
          dict(pos=99, seq="AAAA" + REF[99:105], qual=30, cigar="4S6M")]
    b = helpers.mk_batch(rd, [(100, ref[100:200])])
    p = _abi.make_params("hifi-masseq", dist_to_end=0)       # no end zone: isolate the CIGAR walk
    pl = orc.Region(b, 0, p).pileup().planes()
    tot = lambda i: sum(int(pl[k][i]) for k in range(4))
    assert [tot(i) for i in range(6)] == [3, 3, 3, 3, 3, 1]   # read0 cols 0-4 (95..104), read1 0-9, read2 0-4 (pos 99: 1 col clipped)
    assert _col(pl, 4)["ni"] == 1                              # read0's 2I after ref 104 -> column 4 (util.rs:918-929)
    assert [int(pl[_abi.PL_D][i]) for i in (4, 5, 6, 7, 8)] == [0, 1, 1, 1, 0]
    assert int(pl[_abi.PL_N][10]) == 1 and int(pl[_abi.PL_N][59]) == 1 and int(pl[_abi.PL_N][60]) == 0
    assert int(pl[_abi.PL_NI].sum()) == 1                      # read1's leading 3I sits at pos_in_freq_vec 0 < 1: not counted
    assert tot(60) == 1 and tot(69) == 1 and tot(70) == 0
    # strands: read1 is reverse -> counted in cnt but not fwd; ts: (+,+)->ts_fwd, (-,+)->ts_rev, no tag -> none
    c0 = _col(pl, 0)
    assert c0["ts_fwd"] == 1 and c0["ts_rev"] == 1
    assert sum(c0["fwd_" + x] for x in "acgt") == 2


def test_polya_mask_truth_table(orc):
    """util.rs:754-789: windows [c-L, c+1] of L identical bases != column ref base mask the base."""
    L = 5
    body = REF[1:61]  # starts with C: the homopolymer run is exactly the 5 leading A
    ref = "G" * 5 + body  # first 5 aligned bases are AAAAA over ref GGGGG
    rd = [dict(pos=0, seq="AAAAA" + body, qual=30, cigar="65M")] * 6
    b = helpers.mk_batch(rd, [(0, ref)])
    pl = orc.Region(b, 0, _abi.make_params("hifi-masseq", dist_to_end=40)).pileup().planes()
    depth = pl[:4].sum(axis=0)
    # bases 0..4 lie in window t=0; base 5 is masked by window t = c-L = 0 (does not contain it) if ref[5] != 'A'
    assert list(depth[:5]) == [0] * 5
    assert depth[5] == (0 if ref[5] != "A" else 6)
    assert depth[6] == 6                                     # window starts must be >= c-L = 1: AAAA+body[0] is not a run
    # same read over a reference that *is* A under the run: those five are kept (ref_base != 'A' fails);
    # column 5 (ref C) is still masked by the neighbouring window t = c-L
    b2 = helpers.mk_batch(rd, [(0, "A" * 5 + body)])
    pl2 = orc.Region(b2, 0, _abi.make_params("hifi-masseq", dist_to_end=40)).pileup().planes()
    assert list(pl2[:4].sum(axis=0)[:7]) == [6] * 5 + [0, 6]
    # ONT: everything within D of either end is trimmed, no window scan (util.rs:745-751)
    pl3 = orc.Region(b, 0, _abi.make_params("ont-cdna", dist_to_end=20)).pileup().planes()
    d3 = pl3[:4].sum(axis=0)
    assert list(d3[:20]) == [0] * 20 and d3[20] == 6 and d3[44] == 6 and list(d3[46:65]) == [0] * 19
    # mixed A/T window does not trigger (A and T are counted separately, util.rs:765-786)
    rd4 = [dict(pos=0, seq="AATAA" + body, qual=30, cigar="65M")] * 6
    pl4 = orc.Region(helpers.mk_batch(rd4, [(0, ref)]), 0, _abi.make_params("hifi-masseq")).pileup().planes()
    assert pl4[:4].sum(axis=0)[0] == 6


def test_genotype_likelihood_closed_form(orc):
    """candidate.rs:236-335 on a 10-read column: 6 ref (q30) + 4 alt (q20)."""
    ref = REF[:120]
    alt = "T" if ref[60] != "T" else "G"
    rd = [dict(pos=0, seq=ref, qual=30, cigar="120M")] * 6 + \
         [dict(pos=0, seq=ref[:60] + alt + ref[61:], qual=20, cigar="120M")] * 4
    b = helpers.mk_batch(rd, [(0, ref)])
    p = _abi.make_params("hifi-masseq", dist_to_end=0)
    R = orc.Region(b, 0, p).pileup().candidates()
    c = R.cands()
    assert len(c) == 1 and c["pos"][0] == 60 and c["depth"][0] == 10
    e30, e20 = 0.1 ** 3.0, 0.1 ** 2.0
    l0 = 6 * math.log10(e30) + 4 * math.log10(1 - e20)
    l2 = 6 * math.log10(1 - e30) + 4 * math.log10(e20)
    l1 = -10 * math.log10(2)
    assert c["loglik"][0] == pytest.approx([l0, l1, l2], rel=1e-12)
    post = np.array([l0 + math.log10(0.0005), l1 + math.log10(0.001), l2 + math.log10(0.9985)])
    pr = 10 ** (post - post.max()); pr /= pr.sum()
    assert c["qual"][0] == pytest.approx(-10 * math.log10(pr[2]), rel=1e-9)
    g = 10 ** (np.array([l0, l1, l2]) - max(l0, l1, l2)); g /= g.sum()
    ph = sorted(-10 * np.log10(g))
    assert c["gq"][0] == pytest.approx(ph[1] - ph[0], rel=1e-9)
    assert c["variant_type"][0] == 1 and c["genotype"][0] == 0 and c["flags"][0] & _abi.F_HET
    assert c["af1"][0] == np.float32(6) / np.float32(10) and c["af2"][0] == np.float32(4) / np.float32(10)
    # hist-based (order-free) evaluation agrees with the running sum
    h = R.cand_gt_hist(60)
    assert h[:3] == pytest.approx([l0, l1, l2], rel=1e-12)


def test_dense_window_off_by_one(orc):
    """candidate.rs:471-497: `tk in i..j` excludes the last index of a dense window."""
    ref = REF[:400]
    sites = [50, 60, 70, 80, 90, 300]
    alt = lambda ch: "T" if ch != "T" else "G"
    mut = "".join(alt(ch) if i in sites else ch for i, ch in enumerate(ref))
    rd = [dict(pos=0, seq=ref, qual=30, cigar="400M")] * 6 + [dict(pos=0, seq=mut, qual=30, cigar="400M")] * 6
    b = helpers.mk_batch(rd, [(0, ref)])
    c = orc.Region(b, 0, _abi.make_params("hifi-masseq", dist_to_end=0)).pileup().candidates().cands()
    assert list(c["pos"]) == sites
    dense = [bool(f & _abi.F_DENSE) for f in c["flags"]]
    # i=0: diff to 300 exceeds 100 at j=5 with j-i = 5 >= min_dense_cnt -> marks indices 0..4 ; index 5 stays
    assert dense == [True] * 5 + [False]
    assert all(not (f & _abi.F_FOR_PHASING) for f in c["flags"][:5]) and c["flags"][5] & _abi.F_FOR_PHASING


def test_fragment_matrix_and_two_snp_phasing(orc):
    """fragment.rs:93-307 + phase.rs enumeration on a hand-built 2-SNP region: two clean haplotypes."""
    ref = REF[:300]
    alt = lambda ch: "T" if ch != "T" else "G"
    s1, s2 = 100, 180
    h1 = ref
    h2 = ref[:s1] + alt(ref[s1]) + ref[s1 + 1:s2] + alt(ref[s2]) + ref[s2 + 1:]
    rd = [dict(pos=0, seq=h1, qual=30, cigar="300M")] * 8 + [dict(pos=0, seq=h2, qual=30, cigar="300M")] * 8
    rd.append(dict(pos=0, seq=h2[:150] + h2[190:], qual=30, cigar="150M40N110M"))  # intron over SNP 2
    b = helpers.mk_batch(rd, [(0, ref)])
    p = _abi.make_params("hifi-masseq", dist_to_end=0)
    R = orc.Region(b, 0, p).pileup().candidates().fragments()
    fm = R.fragmat()
    assert list(np.diff(fm["row_ptr"])) == [2] * 16 + [1]
    v = fm["val"]
    assert all((x & 31) == 30 for x in v) and [int(x >> 5) & 1 for x in v[:4]] == [1, 1, 1, 1] and (v[-1] >> 5) & 1 == 0
    R.phase(orc.MODE_EXACT).post_phase()
    c, pr = R.cands(), R.phase_result()
    assert list(c["variant_type"]) == [1, 1] and c["haplotype"][0] == c["haplotype"][1]  # cis: same delta
    tags = pr["haplotag"]
    assert len(set(tags[:8])) == 1 and len(set(tags[8:])) == 1 and tags[0] == -tags[8]
    assert set(pr["assignment"]) == {1, 2} and c["phase_set"][0] == c["phase_set"][1] == s1 + 1
    txt = R.vcf_text("chrT").splitlines()
    assert len(txt) == 2 and txt[0].split("\t")[6] == "PASS" and txt[0].split("\t")[9].split(":")[0] in ("0|1", "1|0")
    # both decision modes agree on this instance
    R2 = orc.Region(b, 0, p).run_all(orc.MODE_F64)
    assert np.array_equal(R2.phase_result()["haplotag"], tags)
