"""The phasing half of the oracle, pinned by a second restatement: oracle/oracle_np_phase.py (plain Python objects that
mirror the Rust structs, written from the Rust text) against oracle/lcr_oracle.cpp (ORC_MODE_F64, the reference's
arithmetic) -- fragment matrix, LD blocks in the reference's block / node order, sigma / delta / eta after
SNPFrag::phase (both branches), the post-phase steps, phase sets -- plus hand-derived known answers."""
import numpy as np
import pytest

import helpers
from longcallr_amd import _abi, synth
from oracle import oracle_np_phase as onp2

F = _abi


def compare_region(orc, batch, g, prm, expect_chain=None):
    R = orc.Region(batch, g, prm).pileup().candidates().fragments()
    c0 = R.cands()
    sf, read_ps = onp2.run_region(batch, g, prm, c0)
    # ---- P6: fragment matrix
    fm = R.fragmat()
    assert len(sf.fragmat_snapshot) == len(fm["row_read"])
    for k, (read, ents, links, fp) in enumerate(sf.fragmat_snapshot):
        assert read == fm["row_read"][k]
        e0, e1 = fm["row_ptr"][k], fm["row_ptr"][k + 1]
        assert [e[0] for e in ents] == fm["col"][e0:e1].tolist()
        want = [(e[2] & 31) | (32 if e[3] == 1 else 0) for e in ents]
        assert [int(v) & 63 for v in fm["val"][e0:e1]] == want
        assert links == fm["row_links"][k] and int(fp) == fm["row_for_phasing"][k]
    if c0.size == 0:
        return sf, R
    R.phase(orc.MODE_F64)
    chain = c0.size > prm.max_enum_snps
    if expect_chain is not None:
        assert chain == expect_chain
    # ---- P7: LD blocks (the reference builds them in both branches)
    assert sf.ld_blocks == R.ld_blocks()
    R.post_phase()
    st = R.stats()
    assert st["assert_violations"] == 0
    # ---- P12 / P13: number of cross_optimize calls and the objective; P14-P17: everything the VCF / BAM writers read
    assert sf.n_cross == st["cross_optimize_calls"]
    pr = R.phase_result()
    assert sf.objective == pytest.approx(pr["objective"], abs=1e-9)
    assert [f.haplotag for f in sf.fragments] == pr["haplotag"].tolist()
    assert [f.assignment for f in sf.fragments] == pr["assignment"].tolist()
    assert [read_ps.get(k, 0) for k in range(len(sf.fragments))] == pr["phase_set"].tolist()
    c1 = R.cands()
    for s, c in zip(sf.candidate_snps, c1):
        assert (s.haplotype, s.genotype, s.variant_type, s.phase_set) == (c["haplotype"], c["genotype"], c["variant_type"], c["phase_set"])
        fl = int(c["flags"])
        assert (s.rna_editing, s.for_phasing, s.single, s.non_selected, s.cand_somatic) == (
            bool(fl & F.F_RNA_EDIT), bool(fl & F.F_FOR_PHASING), bool(fl & F.F_SINGLE), bool(fl & F.F_NON_SELECTED), bool(fl & F.F_CAND_SOMATIC))
        assert s.phase_score == pytest.approx(c["phase_score"], abs=1e-9)
    return sf, R


def test_demo_bam_chain(orc):
    sf, R = compare_region(orc, helpers.demo_batch(), 0, _abi.make_params("hifi-masseq"), expect_chain=True)
    assert len(sf.candidate_snps) == 19 and sf.n_cross == 1 + 2 * (19 // 4 + 1)


@pytest.mark.parametrize("profile,preset,seed", [("ont-cdna", "ont-cdna", 5), ("masseq", "hifi-masseq", 6), ("ont-drna", "ont-drna", 7),
                                                 ("ont-cdna", "hifi-isoseq", 8)])
def test_synthetic_regions(orc, profile, preset, seed):
    b = synth.make_batch(profile, n_genes=3, gene_len=7000, depth=25, seed=seed)
    prm = _abi.make_params(preset, seed=seed, max_enum_snps=6)     # (2^S restarts in pure Python: keep S small)
    n = 0
    for g in range(b.n_regions):
        sf, _ = compare_region(orc, b, g, prm)
        n += len(sf.candidate_snps)
    assert n > 0


def test_chain_region_with_blocks(orc):
    """S > max_enum_snps: LD-seeded start, block flip, perturbation rounds; several LD blocks"""
    b = synth.make_batch("ont-drna", n_genes=1, gene_len=30000, depth=30, seed=21)
    sf, R = compare_region(orc, b, 0, _abi.make_params("ont-drna", seed=5), expect_chain=True)
    assert len(sf.ld_blocks) >= 1 and max(len(x) for x in sf.ld_blocks) >= 2


def test_min_linkers_two(orc):
    b = synth.make_batch("ont-drna", n_genes=1, gene_len=20000, depth=25, seed=22)
    compare_region(orc, b, 0, _abi.make_params("ont-drna", seed=6, min_linkers=2))


def test_low_fraction_rescue(orc):
    """eval_low_frac_var_phase rescues a low-fraction site that phases with its neighbours (draws for the rescued reads)"""
    from test_gpu_parity import _low_fraction_batch
    b = _low_fraction_batch(seed=1)
    sf, R = compare_region(orc, b, 0, _abi.make_params("ont-cdna", seed=3, min_phase_score=4.0))
    assert sf.somatic_snps and all(sf.candidate_snps[i].for_phasing and not sf.candidate_snps[i].cand_somatic for i in sf.somatic_snps)
    sf, R = compare_region(orc, b, 0, _abi.make_params("ont-cdna", seed=3, min_phase_score=13.0))
    assert sf.somatic_snps and all(sf.candidate_snps[i].cand_somatic and not sf.candidate_snps[i].for_phasing for i in sf.somatic_snps)


# ---- hand-derived known answers, checked on BOTH restatements -------------------------------------------------------
def test_kat_eleven_snp_chain(orc):
    """S = 11 > max_enum_snps: the chain branch on an instance whose optimum is known by hand.  Every pair of SNPs is in
    perfect LD (no read conflicts), so the LD graph is complete: one block, and petgraph's Dfs (largest unvisited
    neighbour first) lists it as 0, 10, 9, ..., 1.  The optimum explains every observation: the objective is
    nnz * log10(1 - 10^-3), haplotype A's reads carry one haplotag and B's the other, delta has one sign, every SNP is
    het with phase set = first SNP's position + 1."""
    import math
    b, sites = helpers.two_haplotype_batch(n_snps=11, n_reads=40)
    prm = _abi.make_params("hifi-masseq", seed=9)
    sf, R = compare_region(orc, b, 0, prm, expect_chain=True)
    assert [s.pos for s in sf.candidate_snps] == sites[0]
    assert sf.ld_blocks == [[0] + list(range(10, 0, -1))]
    assert sf.n_cross == 1 + 2 * (11 // 4 + 1)
    nnz = sum(len(f.list) for f in sf.fragments)
    assert nnz == 40 * 11 and sf.objective == pytest.approx(nnz * math.log10(1.0 - 1e-3), abs=1e-9)
    tags = [f.haplotag for f in sf.fragments]
    assert set(tags[0::2]) | set(tags[1::2]) == {-1, 1} and len(set(tags[0::2])) == 1 and tags[0] == -tags[1]   # reads alternate A, B
    assert len({s.haplotype for s in sf.candidate_snps}) == 1 and all(s.genotype == 0 and s.variant_type == 1 for s in sf.candidate_snps)
    assert all(s.phase_set == sites[0][0] + 1 for s in sf.candidate_snps)
    assert all(f.assignment in (1, 2) for f in sf.fragments)
    # sigma * delta = p on every entry: a haplotype-A read (alt alleles, p = -1) carries -delta
    d = sf.candidate_snps[0].haplotype
    assert all(f.haplotag * d == f.list[0].p for f in sf.fragments)


def test_kat_two_phase_sets(orc):
    """Two stretches no read connects: two components of the phase-set graph (snpfrags.rs:628-733), each named after its
    first SNP (pos + 1), the reads of a stretch carry that stretch's phase set; no LD pair joins the stretches."""
    b, sites = helpers.two_haplotype_batch(n_snps=6, groups=2, n_reads=40)
    sf, R = compare_region(orc, b, 0, _abi.make_params("hifi-masseq", seed=4), expect_chain=True)
    assert [s.pos for s in sf.candidate_snps] == sites[0] + sites[1]
    assert [s.phase_set for s in sf.candidate_snps] == [sites[0][0] + 1] * 6 + [sites[1][0] + 1] * 6
    assert sorted(map(sorted, sf.ld_blocks)) == [[0, 1, 2, 3, 4, 5], [6, 7, 8, 9, 10, 11]]
    assert sf.ld_blocks[0][0] == 6 and sf.ld_blocks[1][0] == 0      # kosaraju_scc: descending first node
    ps = R.phase_result()["phase_set"]
    assert set(ps[:40].tolist()) == {sites[0][0] + 1} and set(ps[40:].tolist()) == {sites[1][0] + 1}


@pytest.mark.parametrize("min_phase_score,rescued", [(8.0, True), (60.0, False)])
def test_kat_rna_edit_rescue(orc, min_phase_score, rescued):
    """An A>G site on '+' transcripts is set aside as RNA editing (candidate.rs:379-407) and re-enters the phasing in
    eval_rna_edit_var_phase (snpfrags.rs:191-281) when its phase score reaches min_phase_score - 3: G only ever
    appears on haplotype A, so the score is high; with an unreachable threshold it stays an editing site."""
    b, sites = helpers.two_haplotype_batch(n_snps=5, n_reads=60, edit_sites=(777,), edit_frac=0.95, seed=2)
    for r in range(b.n_reads):                      # all reads on the forward strand with ts '+'
        b.flags[r] = 0 | (1 << 1)
    sf, R = compare_region(orc, b, 0, _abi.make_params("hifi-masseq", seed=4, min_phase_score=min_phase_score))
    e = [i for i, s in enumerate(sf.candidate_snps) if s.pos == 5000 + 777]
    assert e == sf.edit_snps and len(e) == 1
    s = sf.candidate_snps[e[0]]
    if rescued:
        assert s.for_phasing and not s.rna_editing and s.genotype == 0 and s.phase_score >= min_phase_score - 3
        assert s.phase_set == sites[0][0] + 1
    else:
        assert s.rna_editing and s.non_selected and s.phase_set == 0
