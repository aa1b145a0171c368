"""The two independent restatements (C++ oracle, vectorised NumPy) must agree: this is what pins
the oracle in the absence of reference tests / goldens / a buildable reference binary."""
import math

import numpy as np
import pytest

import helpers
from longcallr_amd import _abi, synth
from oracle import oracle_np


def _compare_region(orc, batch, g, prm):
    R = orc.Region(batch, g, prm).pileup().candidates()
    pl = R.planes()
    pu = oracle_np.pileup(batch, g, prm)
    for k in range(4):
        assert np.array_equal(pl[_abi.PL_A + k], pu["cnt"][k])
        assert np.array_equal(pl[_abi.PL_FWD_A + k], pu["fwd"][k])
    assert np.array_equal(pl[_abi.PL_N], pu["n"]) and np.array_equal(pl[_abi.PL_D], pu["d"])
    assert np.array_equal(pl[_abi.PL_NI], pu["ni"])
    assert np.array_equal(pl[_abi.PL_TS_FWD], pu["ts"][0]) and np.array_equal(pl[_abi.PL_TS_REV], pu["ts"][1])
    # per-allele quality lists of a few deep columns == the NumPy histograms
    for col in np.argsort(-pu["cnt"].sum(axis=0))[:5]:
        for a in range(4):
            q = R.baseq(int(col), a)
            assert np.array_equal(np.bincount(q, minlength=31), pu["hist"][a, :, col])
    c = R.cands()
    want = oracle_np.candidates(pu, int(batch.start0[g]), prm)
    assert [int(x) for x in c["pos"]] == [w["pos"] for w in want]
    for got, w in zip(c, want):
        assert (chr(got["ref_base"]), chr(got["allele1"]), chr(got["allele2"])) == (w["ref"], w["a1"], w["a2"])
        assert int(got["depth"]) == w["depth"] and int(got["genotype"]) == w["gt"]
        assert int(got["variant_type"]) == w["vt"]
        kind = ("edit" if got["flags"] & _abi.F_RNA_EDIT else "somatic" if got["flags"] & _abi.F_CAND_SOMATIC
                else "hom" if got["flags"] & _abi.F_HOM else "het")
        assert kind == w["kind"]
        assert np.allclose(got["loglik"], w["loglik"], rtol=1e-10, atol=1e-9)
        assert got["qual"] == pytest.approx(w["qual"], rel=1e-9) and (
            got["gq"] == pytest.approx(w["gq"], rel=1e-9) or (math.isinf(got["gq"]) and math.isinf(w["gq"])))
        assert got["af1"] == np.float32(w["af1"]) and got["af2"] == np.float32(w["af2"])
    return len(c)


def test_demo_bam(orc):
    assert _compare_region(orc, helpers.demo_batch(), 0, _abi.make_params("hifi-masseq")) == 19


@pytest.mark.parametrize("profile,preset", [("ont-cdna", "ont-cdna"), ("ont-cdna", "hifi-isoseq"), ("masseq", "hifi-masseq"),
                                            ("ont-drna", "ont-drna")])
def test_synthetic(orc, profile, preset):
    b = synth.make_batch(profile, n_genes=2, gene_len=6000, depth=30, seed=5)
    n = sum(_compare_region(orc, b, g, _abi.make_params(preset)) for g in range(b.n_regions))
    assert n > 0


def test_probability_functions_agree(orc):
    rng = np.random.default_rng(3)
    L = orc.lib()
    for _ in range(50):
        n = int(rng.integers(1, 12))
        sg = rng.choice([-1, 1], n).astype(np.int32); dl = rng.choice([-1, 1], n).astype(np.int32)
        et = rng.choice([-1, 0, 0, 1], n).astype(np.int32); ps = rng.choice([-1, 1], n).astype(np.int32)
        pr = (10.0 ** (-rng.integers(1, 31, n) / 10.0)).astype(np.float64)
        a = L.orc_cal_sigma_delta_eta_log(1, n, dl.ctypes.data, et.ctypes.data, ps.ctypes.data, pr.ctypes.data)
        assert a == oracle_np.cal_sigma_delta_eta_log(1, dl.tolist(), et.tolist(), ps.tolist(), pr.tolist())
        for d, e in ((1, 0), (-1, 0), (1, 1), (1, -1)):
            a = L.orc_cal_delta_eta_sigma_log(d, e, n, sg.ctypes.data, ps.ctypes.data, pr.ctypes.data)
            assert a == oracle_np.cal_delta_eta_sigma_log(d, e, sg.tolist(), ps.tolist(), pr.tolist())
        a = L.orc_cal_phase_score_log(-1, 0, n, sg.ctypes.data, ps.ctypes.data, pr.ctypes.data)
        assert a == oracle_np.cal_phase_score_log(-1, 0, sg.tolist(), ps.tolist(), pr.tolist())


def test_decision_modes_agree_on_demo(orc):
    """ORC_MODE_F64 (reference-order f64) and ORC_MODE_EXACT (fixed point, the GPU contract) reach the
    same phasing on demo.bam; the objective differs only by fixed-point rounding (<< 1e-4)."""
    b, p = helpers.demo_batch(), _abi.make_params("hifi-masseq")
    A = orc.Region(b, 0, p).run_all(orc.MODE_F64)
    B = orc.Region(b, 0, p).run_all(orc.MODE_EXACT)
    assert A.stats()["noise_ties"] == 0 and A.stats()["assert_violations"] == 0
    pa, pb = A.phase_result(), B.phase_result()
    assert np.array_equal(pa["haplotag"], pb["haplotag"]) and np.array_equal(pa["assignment"], pb["assignment"])
    assert abs(pa["objective"] - pb["objective"]) < 1e-8
    assert A.vcf_text() == B.vcf_text()
