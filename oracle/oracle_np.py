"""TEST INFRASTRUCTURE ONLY — independent NumPy restatement of the pileup / candidate / likelihood
rows and of the five probability functions (P8-P11).

Written separately from oracle/lcr_oracle.cpp, from the same Rust text, in a different (vectorised)
formulation, so that the two restatements pin each other (the reference has no tests, goldens or
buildable binary: PARITY UNPINNED BY THE REFERENCE).  Only tests may import this module.
"""
import math

import numpy as np

MAXQ = 30  # main.rs:20


def _polya_mask(seq, L):
    """hp[t] = base code X (1..4 for A,C,G,T) if seq[t:t+L] is a homopolymer of X, else 0
    (util.rs:765-786: A/T windows -> poly_a_flag, C/G windows -> homopolymer_flag; both skip)."""
    n = seq.size
    hp = np.zeros(max(n, 0), dtype=np.int8)
    if n < L:
        return hp
    win = np.lib.stride_tricks.sliding_window_view(seq, L)
    same = np.all(win == win[:, :1], axis=1)
    code = np.zeros(256, dtype=np.int8)
    for i, b in enumerate(b"ACGT"):
        code[b] = i + 1
    hp[:n - L + 1] = np.where(same, code[win[:, 0]], 0)
    return hp


def pileup(batch, g, prm):
    """util.rs:621-949 for region g.  Returns dict(cnt[4,L], fwd[4,L], n, d, ni, ts[2,L]) and the
    per-allele quality histograms hist[4,31,L] (order-free summary of BaseQual)."""
    L = int(batch.len[g]); start0 = int(batch.start0[g])
    ref = batch.ref[int(batch.col_off[g]):int(batch.col_off[g]) + L]
    cnt = np.zeros((4, L), np.int64); fwd = np.zeros((4, L), np.int64); ts = np.zeros((2, L), np.int64)
    n = np.zeros(L + 1, np.int64); d = np.zeros(L + 1, np.int64); ni = np.zeros(L, np.int64)
    hist = np.zeros((4, MAXQ + 1, L), np.int64)
    code = np.full(256, -1, np.int64)
    for i, b in enumerate(b"ACGT"):
        code[b] = i; code[b + 32] = i
    D, PL, ont = int(prm.dist_to_end), int(prm.polya_len), prm.platform == 1
    for r in range(int(batch.read_begin[g]), int(batch.read_begin[g + 1])):
        so = int(batch.seq_off[r]); sl = int(batch.seq_len[r])
        seq = batch.bases[so:so + sl]; qual = batch.quals[so:so + sl]
        lead, trail = int(batch.lead_clip[r]), int(batch.trail_clip[r])
        strand = int(batch.flags[r]) & 1; tsv = (int(batch.flags[r]) >> 1) & 3
        tsi = -1 if tsv == 0 else (0 if (strand == 0) == (tsv == 1) else 1)
        hp = None
        p = int(batch.pos[r]) - start0
        q = lead if lead > 0 else 0
        co = int(batch.cig_off[r])
        for w in batch.cigar[co:co + int(batch.n_cig[r])]:
            op, ln = int(w) & 15, int(w) >> 4
            if op in (4, 5):
                continue
            if op in (0, 7, 8):
                lo, hi = max(p, 0), min(p + ln, L)
                if hi > lo:
                    cols = np.arange(lo, hi)
                    c = q + (cols - p)
                    zone = (np.abs(c - lead) < D) | (np.abs(c - (sl - trail)) < D)
                    keep = np.ones(cols.size, bool)
                    if ont:
                        keep &= ~zone
                    elif zone.any():
                        if hp is None:
                            hp = _polya_mask(seq, PL)
                        refcode = code[ref[cols]] + 1  # 1..4 or 0; lower-case never equals an upper-case X
                        refcode = np.where((ref[cols] >= 97), 0, refcode)
                        masked = np.zeros(cols.size, bool)
                        for t in range(-PL, 2):  # window starts c-L .. c+1 (util.rs:758)
                            tt = c + t
                            ok = (tt >= 0) & (tt + PL - 1 < sl)
                            x = np.where(ok, hp[np.clip(tt, 0, sl - 1)], 0)
                            masked |= (x > 0) & (x != refcode)
                        keep &= ~(zone & masked)
                    cols, c = cols[keep], c[keep]
                    if tsi >= 0:
                        np.add.at(ts[tsi], cols, 1)
                    b = code[seq[c]]
                    okb = b >= 0
                    np.add.at(cnt, (b[okb], cols[okb]), 1)
                    if strand == 0:
                        np.add.at(fwd, (b[okb], cols[okb]), 1)
                    np.add.at(hist, (b[okb], np.minimum(qual[c[okb]], MAXQ), cols[okb]), 1)
                p += ln; q += ln
            elif op == 1:
                if 1 <= p < L:
                    ni[p - 1] += 1
                q += ln
            elif op in (2, 3):
                lo, hi = max(p, 0), min(p + ln, L)
                if hi > lo:
                    tgt = d if op == 2 else n
                    tgt[lo] += 1; tgt[hi] -= 1
                p += ln
            else:
                raise ValueError("unknown cigar op")
    return dict(cnt=cnt, fwd=fwd, ts=ts, n=np.cumsum(n)[:L], d=np.cumsum(d)[:L], ni=ni, hist=hist, ref=ref)


def strand_odds_ratio(ref_fw, ref_rv, alt_fw, alt_rv):
    f = np.float32
    x00, x01, x10, x11 = f(ref_fw + 1), f(ref_rv + 1), f(alt_fw + 1), f(alt_rv + 1)
    sym = f(f(x00 * x11) / f(x01 * x10)) + f(f(x01 * x10) / f(x00 * x11))
    rr = f(min(x00, x01) / max(x00, x01)); ar = f(min(x10, x11) / max(x10, x11))
    return f(f(np.log(f(sym))) + f(np.log(rr))) - f(np.log(ar))


def binomial_two_tailed(k, n):
    cdf = lambda x: 1.0 if x >= n else sum(math.comb(n, i) for i in range(x + 1)) / 2.0 ** n
    if k == 0:
        return 2.0 * cdf(0)
    if k == n:
        return 2.0 * (1.0 - cdf(n - 1))
    return 2.0 * min(cdf(k), 1.0 - cdf(k - 1))


def two_major(cnt4, ref_base):
    x = sorted(zip("ACGT", cnt4), key=lambda t: -t[1])  # Python sort is stable, like Rust sort_by
    if x[0][0] != ref_base and x[1][0] != ref_base:
        if x[2][1] == x[1][1] and x[2][0] == ref_base:
            return x[0], x[2]
        if x[3][1] == x[1][1] and x[3][0] == ref_base:
            return x[0], x[3]
    return x[0], x[1]


def candidates(pu, start0, prm):
    """candidate.rs:75-463 (before the dense sweep).  Returns list of dicts in position order."""
    out = []
    L = pu["cnt"].shape[1]
    thr = strand_odds_ratio(5, 5, 9, 1)
    f32 = np.float32
    depth = pu["cnt"].sum(axis=0)
    for col in np.flatnonzero((depth >= prm.min_depth) & (depth <= prm.max_depth)):
        R = chr(pu["ref"][col]); tot = int(depth[col])
        c4 = [int(v) for v in pu["cnt"][:, col]]
        (a1, c1), (a2, c2) = two_major(c4, R)
        af1, af2 = f32(c1) / f32(tot), f32(c2) / f32(tot)
        if a1 == R:
            nalt, refb, alts = 1, a1, [(a2, c2, af2)]
        elif a2 == R:
            nalt, refb, alts = 1, a2, [(a1, c1, af1)]
        else:
            nalt, refb, alts = 2, R, [(a1, c1, af1), (a2, c2, af2)]
        if refb not in "ACGTacgt":
            continue
        if nalt == 1:
            if tot < 200 and alts[0][2] < f32(prm.low_frac_cut):
                continue
            if tot >= 200 and alts[0][1] < prm.low_cnt_cut:
                continue
        dd, nn = int(pu["d"][col]), int(pu["n"][col])
        if dd >= alts[0][1]:
            continue
        if f32(c1 + c2) / f32(tot + dd + nn) < f32(prm.min_af_intron):
            continue
        idx = "ACGT".index
        probe = a1 if a1 != R else (a2 if a2 != R else None)
        if probe is not None:
            pc = c1 if probe == a1 else c2
            if pc > 0 and int(pu["hist"][idx(probe), prm.min_baseq:, col].sum()) < 2:
                continue
        if prm.use_strand_bias:
            st = lambda b: (int(pu["fwd"][idx(b.upper()), col]), int(pu["cnt"][idx(b.upper()), col] - pu["fwd"][idx(b.upper()), col]))
            rf, rr = st(refb)
            sor = max(strand_odds_ratio(rf, rr, *st(a[0])) for a in alts)
            if sor > thr:
                continue
            if nalt == 1:
                afw, arv = st(alts[0][0])
                if afw + arv <= 30 and binomial_two_tailed(afw, afw + arv) < 0.05:
                    continue
                if afw * arv == 0:
                    continue
        if R not in "ACGT":
            continue
        ri = idx(R)
        hm = pu["hist"][ri, :, col]
        hx = pu["hist"][:, :, col].sum(axis=0) - hm
        q = np.arange(MAXQ + 1, dtype=np.float64)
        with np.errstate(divide="ignore"):
            e = np.power(0.1, q / 10.0)
            le, l1e = np.log10(e), np.log10(1.0 - e)
        def term(h, l):
            with np.errstate(invalid="ignore"):
                return float(np.sum(np.where(h > 0, h * l, 0.0)))
        l0 = term(hm, le) + term(hx, l1e)
        l2 = term(hm, l1e) + term(hx, le)
        loglik = np.array([l0, -tot * math.log10(2.0), l2])
        theta = 0.001
        lp = loglik + np.log10([theta / 2, theta, 1 - 1.5 * theta])
        with np.errstate(over="ignore", invalid="ignore", divide="ignore"):
            vp = np.power(10.0, lp - np.max(lp)); vp = vp / vp.sum()
            qual = -10.0 * math.log10(max(10e-301, vp[2])) if not math.isnan(vp[2]) else 3000.0
            gl = np.power(10.0, loglik - np.max(loglik)); gp = gl / gl.sum()
            ph = np.sort(-10.0 * np.log10(gp))
        gq = ph[1] - ph[0]
        if gp[0] > gp[1] and gp[0] > gp[2]:
            vt, gt = 2, -1
        elif gp[1] > gp[0] and gp[1] > gp[2]:
            vt, gt = 1, 0
        else:
            vt, gt = 0, 1
        if qual < prm.min_qual:
            continue
        tf, tr = int(pu["ts"][0, col]), int(pu["ts"][1, col])
        kind = None
        if refb == "A" and alts[0][0] == "G" and (tf > tr * 2 or (tf == 0 and tr == 0)) and vt != 2:
            kind = "edit"
        elif refb == "T" and alts[0][0] == "C" and (tr > tf * 2 or (tf == 0 and tr == 0)) and vt != 2:
            kind = "edit"
        elif nalt == 1 and alts[0][2] < f32(prm.min_af):
            kind = "somatic"
        elif vt == 2:
            if nalt == 2 and alts[0][2] >= f32(prm.min_af) and alts[1][2] >= f32(prm.min_af):
                vt, gt = 3, -1
            kind = "hom"
        elif vt == 1:
            if nalt == 2:
                vt, gt, kind = 3, -1, "hom"
            else:
                kind = "het"
        if kind is None:
            continue
        out.append(dict(pos=start0 + int(col), ref=R, a1=a1, a2=a2, depth=tot, vt=vt, gt=gt, kind=kind,
                        loglik=loglik, qual=qual, gq=gq, af1=float(af1), af2=float(af2)))
    return out


# ---- probability functions (phase.rs:32-49,77-96,128-176,238-255) on plain lists -----------------
def aki(sigma, delta, eta, p, err):
    x = sigma * delta if eta == 0 else eta
    return 1.0 - err if p == x else err


def cal_sigma_delta_eta_log(sigma_k, delta, eta, ps, probs):
    l = lambda s: sum(math.log10(aki(s, d, e, p, pr)) for d, e, p, pr in zip(delta, eta, ps, probs))
    return 1.0 - l(sigma_k) / (l(1) + l(-1))


def cal_delta_eta_sigma_log(delta_i, eta_i, sigma, ps, probs):
    l = lambda d, e: sum(math.log10(aki(s, d, e, p, pr)) for s, p, pr in zip(sigma, ps, probs))
    hr, hv = math.log10(1 - 1.5 * 0.001), math.log10(0.5 * 0.001)
    het = math.log10(0.001) - (len(sigma) * math.log10(2.0) if sigma else 0.0)
    prior = {0: het, 1: hr, -1: hv}
    num = l(delta_i, eta_i) + prior[eta_i]
    den = (l(delta_i, -1) + hv) + (l(delta_i, 0) + het) + (l(delta_i, 1) + hr) + (l(-delta_i, 0) + het)
    return 1.0 - num / den


def cal_phase_score_log(delta_i, eta_i, sigma, ps, probs):
    l = lambda d: sum(math.log10(aki(s, d, eta_i, p, pr)) for s, p, pr in zip(sigma, ps, probs))
    return 1.0 - l(delta_i) / (l(1) + l(-1))


def discover_regions(spans, ref_len):
    """find_isolated_regions_with_depth (util.rs:236-332, truncation off), restated loop for loop:
    spans = [(reference_start, reference_end)] of the filtered reads of one contig.
    Returns [(start0, len, max_cov)] with start0 = Region.start - 1, len = Region.end - Region.start.

    Quirk kept (util.rs:297-310): the cursors are reset only when a region is emitted, and a region is
    emitted only if region_end > region_start, so a single-column island is not dropped: it stays
    pending and becomes the start of a region that runs through the next island (gap included)."""
    depth = [0] * ref_len
    for s, e in spans:
        for i in range(max(s, 0), min(e, ref_len)):
            depth[i] += 1
    out = []
    region_start = region_end = -1
    max_coverage = 0
    for i in range(ref_len):
        if depth[i] > max_coverage:
            max_coverage = depth[i]
        if depth[i] == 0:
            if region_end > region_start:
                out.append((region_start, region_end - region_start + 1, max_coverage))
                region_start = region_end = -1
                max_coverage = 0
        else:
            if region_start == -1:
                region_start = region_end = i
            else:
                region_end = i
    if region_end > region_start:
        out.append((region_start, region_end - region_start + 1, max_coverage))
    return out


def phased_bam_records(records, regions, haplotag_queue, phaseset_queue):
    """The phased-BAM loop of thread.rs:307-361 restated on plain tuples -- which records are written, in which order, with which tags.
      records          [(ref_id, reference_start, reference_end, flag, qname, has_HP_tag, has_PS_tag)] in file order (one
                       coordinate-sorted file; reference_end as htslib reports it: start + reference length, at least start + 1)
      regions          [(ref_id, start, end)] = Region.start / Region.end as the reference holds them (1-based start, exclusive end)
      haplotag_queue   [(qname, assignment)], phaseset_queue [(qname, phase_set)] in queue order
    Returns [(record index, HP or None, PS or None)] in output order.
      thread.rs:308-325  the FIRST entry of a name wins in both maps (later ones are skipped)
      thread.rs:331-334  fetch((chr, start, end)): htslib takes the numbers as a 0-based half-open interval -- records that OVERLAP it
      thread.rs:336-338  unmapped / secondary / supplementary records are skipped (duplicates, QC-fail, low MAPQ are NOT)
      thread.rs:339-345  reference_start + 1 < start or reference_end + 1 > end: skipped ("reads beyond the region boundary")
      thread.rs:347-352  HP:i only for a non-zero assignment; :353-356 PS for every name in the map; rust-htslib's push_aux
                         refuses a tag the record already carries (the Result is dropped: the record keeps its own tag)
      every region is walked in list order: a record inside two regions is written twice (the TODO of thread.rs:332)."""
    hp, ps = {}, {}
    for name, a in haplotag_queue:
        if name not in hp:
            hp[name] = a
    for name, p in phaseset_queue:
        if name not in ps:
            ps[name] = p
    out = []
    for ref_id, start, end in regions:
        for idx, (rid, rs, re_, flag, name, has_hp, has_ps) in enumerate(records):
            if rid != ref_id or not (rs < end and re_ > start):
                continue
            if flag & 0x4 or flag & 0x100 or flag & 0x800:
                continue
            if rs + 1 < start or re_ + 1 > end:
                continue
            h = hp.get(name)
            p = ps.get(name)
            out.append((idx, h if (h is not None and h != 0 and not has_hp) else None, p if (p is not None and not has_ps) else None))
    return out
