"""TEST INFRASTRUCTURE ONLY — second, independent restatement of the PHASING half of the hot path:
fragment matrix, LD blocks (P7), cross_optimize (P12), the restart policy of SNPFrag::phase
(P13, both branches, with cross_optimize_by_block) and the post-phase steps (P14-P17).

Written from the Rust text of /root/reference/src (fragment.rs:10-309, candidate.rs:615-747, snp.rs:158-188,
phase.rs:609-701,810-1394, snpfrags.rs:191-733, thread.rs:160-201) as plain Python objects that mirror the Rust
structs -- not from oracle/lcr_oracle.cpp, whose phasing half this module pins (tests/test_oracle_np_phase.py
compares the two on demo.bam, the synthetic profiles and chain regions).  The reference has no tests, goldens or
buildable binary here (PARITY UNPINNED BY THE REFERENCE); two independent readings that agree are the strongest pin
available.  Only tests may import this module.

Conventions shared with the C++ oracle (they are substitutions, not restatements, and are documented in DESIGN.md):
  * rand::thread_rng() is replaced by the counter-based u01(region_seed, ctr), ctr = 0, 1, 2 ... in the reference's call
    order (orc_common.h); petgraph 0.6.4 (GraphMap / kosaraju_scc / Bfs) is restated from its published algorithm;
  * arithmetic is the reference's: f64 running sums in the reference's loop order (the C++ oracle's ORC_MODE_F64);
    where the reference sums over a HashMap (check_new_haplotag / check_new_haplotype_genotype) the keys are taken in
    ascending order.
"""
import math

from . import oracle_np as onp

M64 = (1 << 64) - 1


def _mix64(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def u01(seed, ctr):
    z = _mix64((seed + (ctr + 1) * 0x9E3779B97F4A7C15) & M64)
    return (z >> 11) * (1.0 / 9007199254740992.0)


def region_seed(seed, start0):
    return _mix64((seed + 0xD1B54A32D192ED03 * (start0 + 1)) & M64)


class Snp:
    """CandidateSNP (snp.rs:39-90), the fields the phasing half reads or writes"""

    def __init__(self, pos, reference, alleles, allele_freqs, variant_type, genotype, rna_editing, dense, for_phasing,
                 hom_var, cand_somatic, phase_score):
        self.pos, self.reference, self.alleles, self.allele_freqs = pos, reference, alleles, allele_freqs
        self.variant_type, self.genotype, self.haplotype = variant_type, genotype, 0
        self.rna_editing, self.dense, self.for_phasing, self.hom_var = rna_editing, dense, for_phasing, hom_var
        self.cand_somatic = cand_somatic
        self.single = self.non_selected = False
        self.phase_score, self.phase_set = phase_score, 0
        self.snp_cover_fragments = []


class FragElem:
    def __init__(self, snp_idx, pos, base, baseq, strand, p, phase_site):
        self.snp_idx, self.pos, self.base, self.baseq, self.strand, self.p, self.phase_site = snp_idx, pos, base, baseq, strand, p, phase_site
        self.prob = math.pow(10.0, -float(baseq) / 10.0)   # fragment.rs:132


class Fragment:
    def __init__(self, idx, read):
        self.fragment_idx, self.read = idx, read
        self.list = []
        self.haplotag = self.assignment = 0
        self.for_phasing = False
        self.num_hete_links = 0


class SNPFrag:
    def __init__(self, snps, min_linkers, seed, start0):
        self.candidate_snps = snps
        self.fragments = []
        self.allele_pairs = {}      # (i, j) -> {"ld_pairs": {(b1, b2): n}, "valid", "score", "weight"}
        self.ld_blocks = []
        self.min_linkers = min_linkers
        # edit_snps / somatic_snps are fixed at candidate time (snpfrags.rs:20-26, candidate.rs:379-417)
        self.edit_snps = [i for i, s in enumerate(snps) if s.rna_editing]
        self.somatic_snps = [i for i, s in enumerate(snps) if s.cand_somatic]
        self.seed, self.ctr = region_seed(seed, start0), 0

    def rnd(self):
        r = u01(self.seed, self.ctr)
        self.ctr += 1
        return r

    # ------------------------------------------------------------------ fragment.rs:10-309
    def get_fragments(self, batch, g):
        snps = self.candidate_snps
        if not snps:
            return
        start0 = int(batch.start0[g])
        for r in range(int(batch.read_begin[g]), int(batch.read_begin[g + 1])):
            pos = int(batch.pos[r])                       # the reads of a batch passed the filters of fragment.rs:32-49
            if pos > snps[-1].pos:
                continue
            so = int(batch.seq_off[r])
            seq = batch.bases[so:so + int(batch.seq_len[r])]
            qual = batch.quals[so:so + int(batch.seq_len[r])]
            strand = int(batch.flags[r]) & 1
            pos_on_ref, pos_on_query = pos, int(batch.lead_clip[r])
            idx = 0
            if pos > snps[0].pos:
                while idx < len(snps) and snps[idx].pos < pos:
                    idx += 1
            snp_pos = snps[idx].pos
            frag = Fragment(len(self.fragments), r)
            co = int(batch.cig_off[r])
            for w in batch.cigar[co:co + int(batch.n_cig[r])]:
                op, ln = int(w) & 15, int(w) >> 4
                if op in (4, 5):
                    continue
                if op in (0, 7, 8):
                    for _ in range(ln):
                        if pos_on_ref == snp_pos:
                            s = snps[idx]
                            base = chr(seq[pos_on_query])
                            q = int(qual[pos_on_query])
                            if base == s.reference:
                                p = 1
                            elif base in (s.alleles[0], s.alleles[1]) and base != s.reference:
                                p = -1
                            else:
                                p = 0
                            fe = FragElem(idx, pos_on_ref, base, q if q < 30 else 30, strand, p, s.for_phasing)
                            if not s.dense and p != 0:
                                frag.list.append(fe)
                            idx += 1
                            if idx < len(snps):
                                snp_pos = snps[idx].pos
                        pos_on_query += 1
                        pos_on_ref += 1
                elif op == 1:
                    pos_on_query += ln
                elif op in (2, 3):
                    for _ in range(ln):
                        if pos_on_ref == snp_pos:
                            idx += 1
                            if idx < len(snps):
                                snp_pos = snps[idx].pos
                        pos_on_ref += 1
                else:
                    raise ValueError("unknown cigar operation")
            lst = frag.list
            for i in range(len(lst)):                     # fragment.rs:208-240
                for j in range(i + 1, len(lst)):
                    a, b = (lst[i], lst[j]) if lst[i].snp_idx < lst[j].snp_idx else (lst[j], lst[i])
                    ent = self.allele_pairs.setdefault((a.snp_idx, b.snp_idx), dict(ld_pairs={}, valid=False, score=0.0, weight=0))
                    ent["ld_pairs"][(a.base, b.base)] = ent["ld_pairs"].get((a.base, b.base), 0) + 1
            frag.num_hete_links = sum(1 for fe in lst if fe.phase_site)
            frag.for_phasing = frag.num_hete_links >= self.min_linkers
            for fe in lst:
                snps[fe.snp_idx].snp_cover_fragments.append(frag.fragment_idx)
            self.fragments.append(frag)
        del start0

    # ------------------------------------------------------------------ candidate.rs:615-747, snp.rs:158-188
    def divide_snps_into_blocks(self, ld_weight_threshold):
        snps = self.candidate_snps
        ld_idxes = [i for i, s in enumerate(snps) if s.for_phasing]

        def ref_alt(s):
            if s.alleles[0] == s.reference and s.alleles[1] != s.reference:
                return s.alleles[0], s.allele_freqs[0], s.alleles[1], s.allele_freqs[1]
            if s.alleles[0] != s.reference and s.alleles[1] == s.reference:
                return s.alleles[1], s.allele_freqs[1], s.alleles[0], s.allele_freqs[0]
            return None
        pass_ld_pair = []
        for a in range(len(ld_idxes)):
            for b in range(a + 1, len(ld_idxes)):
                i1, i2 = ld_idxes[a], ld_idxes[b]
                r1, r2 = ref_alt(snps[i1]), ref_alt(snps[i2])
                if r1 is None or r2 is None or (i1, i2) not in self.allele_pairs:
                    continue
                if r1[1] == 0.0 or r1[3] == 0.0 or r2[1] == 0.0 or r2[3] == 0.0:
                    continue
                ent = self.allele_pairs[(i1, i2)]
                cnt = [ent["ld_pairs"].get(k, 0) for k in ((r1[0], r2[0]), (r1[0], r2[2]), (r1[2], r2[0]), (r1[2], r2[2]))]
                cis, trans = cnt[0] + cnt[3], cnt[1] + cnt[2]
                c1, c2 = min(cis, trans), max(cis, trans)
                score = float("nan") if c2 == 0 else float(c1) / float(c2)      # (f32 in the reference: only == 0.0 is tested)
                if cis > trans:
                    ent["score"], ent["weight"] = score, c2
                else:
                    ent["score"], ent["weight"] = -1.0 * score, -c2
                ent["valid"] = True
                if ent["score"] == 0.0:
                    pass_ld_pair.append((i1, i2))
        graph = Graph()
        for n1, n2 in pass_ld_pair:
            graph.add_edge(n1, n2, self.allele_pairs[(n1, n2)]["weight"])
        for n1, n2 in [(x, y) for x, y, w in graph.all_edges() if abs(w) < ld_weight_threshold]:
            graph.remove_edge(n1, n2)
        self.ld_blocks = graph.kosaraju_scc()
        return graph

    # ------------------------------------------------------------------ phase.rs:609-701
    def init_haplotypes(self):        # phase.rs:443-448
        for s in self.candidate_snps:
            s.haplotype = 1 if self.rnd() < 0.5 else -1

    def init_assignment(self):        # phase.rs:673-680
        for f in self.fragments:
            if f.for_phasing:
                f.haplotag = -1 if self.rnd() < 0.5 else 1

    def init_genotype(self):          # phase.rs:682-691
        for s in self.candidate_snps:
            s.genotype = {0: 1, 1: 0, 2: -1, 3: -1}.get(s.variant_type, s.genotype)

    def init_haplotypes_ld2(self, graph, thr):
        snps = self.candidate_snps
        for s in snps:
            s.haplotype = 1 if self.rnd() < 0.5 else -1
        conserved = set()
        for block in self.ld_blocks:
            if len(block) < 2:
                continue
            snps[block[0]].haplotype = 1
            visited = [block[0]]
            order = graph.bfs(block[0])
            for nx in order:                 # (Bfs yields the start node first; `visited == nx` is skipped below)
                for v in visited:
                    if v == nx:
                        continue
                    key = (v, nx) if v < nx else (nx, v)
                    ent = self.allele_pairs.get(key)
                    if ent is None or not ent["valid"] or ent["score"] != 0.0:
                        continue
                    w = ent["weight"]
                    if w >= thr:
                        snps[nx].haplotype = snps[v].haplotype
                        break
                    elif w <= -thr:
                        snps[nx].haplotype = -snps[v].haplotype
                        break
                visited.append(nx)
            conserved.update(block)
        return conserved

    # ------------------------------------------------------------------ phase.rs:257-355, 810-976
    def row_view(self, k):
        f = self.fragments[k]
        d, e, ps, pr = [], [], [], []
        for fe in f.list:
            if not fe.phase_site:
                continue
            ps.append(fe.p); pr.append(fe.prob)
            d.append(self.candidate_snps[fe.snp_idx].haplotype); e.append(self.candidate_snps[fe.snp_idx].genotype)
        return d, e, ps, pr

    def col_view(self, i):
        sg, ps, pr = [], [], []
        for k in self.candidate_snps[i].snp_cover_fragments:
            f = self.fragments[k]
            if not f.for_phasing or f.haplotag == 0:
                continue
            for fe in f.list:
                if fe.snp_idx == i and fe.phase_site:
                    ps.append(fe.p); pr.append(fe.prob); sg.append(f.haplotag)
        return sg, ps, pr

    def cal_overall_probability(self):
        logp = 0.0
        for f in self.fragments:
            if not f.for_phasing or f.haplotag == 0:
                continue
            for fe in f.list:
                if fe.phase_site:
                    s = self.candidate_snps[fe.snp_idx]
                    logp += math.log10(onp.aki(f.haplotag, s.haplotype, s.genotype, fe.p, fe.prob))
        return logp

    def cross_optimize(self, conserved, keep_conserved, with_genotype):
        self.n_cross += 1
        hg_inc = h_inc = True
        num_iters = 0
        snps = self.candidate_snps
        while hg_inc or h_inc:
            tmp, logp, pre = {}, 0.0, 0.0
            for k, f in enumerate(self.fragments):
                if not f.for_phasing or f.haplotag == 0:
                    continue
                d, e, ps, pr = self.row_view(k)
                if not d:
                    continue
                q = onp.cal_sigma_delta_eta_log(f.haplotag, d, e, ps, pr)
                qn = onp.cal_sigma_delta_eta_log(-f.haplotag, d, e, ps, pr)
                tmp[k] = -f.haplotag if q < qn else f.haplotag
                logp += qn if q < qn else q          # check_new_haplotag, phase.rs:278-314 (keys ascending)
                pre += q
            check = 1 if logp > pre else 0           # (== 0; a decrease would panic in the reference)
            for k, h in tmp.items():
                self.fragments[k].haplotag = h
            if check == 0:
                h_inc = False
            else:
                h_inc = hg_inc = True
            tmp, logp, pre = {}, 0.0, 0.0
            for i, s in enumerate(snps):
                if not s.for_phasing or (keep_conserved and i in conserved):
                    continue
                sg, ps, pr = self.col_view(i)
                if not sg:
                    continue
                q1 = onp.cal_delta_eta_sigma_log(s.haplotype, 0, sg, ps, pr)
                q2 = onp.cal_delta_eta_sigma_log(-s.haplotype, 0, sg, ps, pr)
                q3 = onp.cal_delta_eta_sigma_log(s.haplotype, 1, sg, ps, pr)
                q4 = onp.cal_delta_eta_sigma_log(s.haplotype, -1, sg, ps, pr)
                cur = {0: q1, 1: q3, -1: q4}[s.genotype]
                if with_genotype:
                    mx = max(q1, max(q2, max(q3, q4)))
                    pick = ((s.haplotype, 0), q1) if q1 == mx else ((-s.haplotype, 0), q2) if q2 == mx else \
                        ((s.haplotype, 1), q3) if q3 == mx else ((s.haplotype, -1), q4)
                elif s.genotype == 0:
                    mx = max(q1, q2)
                    pick = ((s.haplotype, 0), q1) if q1 == mx else ((-s.haplotype, 0), q2)
                else:
                    mx = max(q3, q4)
                    pick = ((s.haplotype, 1), q3) if q3 == mx else ((s.haplotype, -1), q4)
                tmp[i] = pick[0]
                logp += pick[1]                       # check_new_haplotype_genotype, phase.rs:316-355
                pre += cur
            check = 1 if logp > pre else 0
            for i, (h, gt) in tmp.items():
                snps[i].haplotype, snps[i].genotype = h, gt
            if check == 0:
                hg_inc = False
            else:
                hg_inc = h_inc = True
            num_iters += 1
            if num_iters > 20:
                break
        return self.cal_overall_probability()

    # ------------------------------------------------------------------ phase.rs:1298-1394
    def cross_optimize_by_block(self):
        snps = self.candidate_snps
        tmp_hap, tmp_tag = {}, {}
        for block in self.ld_blocks:
            bset = set(block)
            db, dbf, eb, sb, sbf, psb, prb = [], [], [], [], [], [], []
            flip_map = {}
            for idx in block:
                db.append(snps[idx].haplotype); dbf.append(-snps[idx].haplotype); eb.append(snps[idx].genotype)
                sg, sgf, ps, pr = [], [], [], []
                for k in snps[idx].snp_cover_fragments:
                    f = self.fragments[k]
                    if not f.for_phasing or f.haplotag == 0:
                        continue
                    flip_read = True
                    for fe in f.list:
                        if fe.snp_idx not in bset:
                            flip_read = False
                        if fe.snp_idx == idx:
                            if not fe.phase_site:
                                continue
                            ps.append(fe.p); pr.append(fe.prob)
                            t = -f.haplotag if flip_read else f.haplotag
                            sgf.append(t); flip_map[k] = t
                            sg.append(f.haplotag)
                sb.append(sg); sbf.append(sgf); psb.append(ps); prb.append(pr)
            q = sum_block(db, eb, sb, psb, prb)
            qf = sum_block(dbf, eb, sbf, psb, prb)
            if q < qf:
                for i, idx in enumerate(block):
                    tmp_hap[idx] = dbf[i]
                for k, f in enumerate(self.fragments):
                    tmp_tag[k] = flip_map.get(k, f.haplotag)
            else:
                for i, idx in enumerate(block):
                    tmp_hap[idx] = db[i]
                for k, f in enumerate(self.fragments):
                    tmp_tag[k] = f.haplotag
        for i, h in tmp_hap.items():
            snps[i].haplotype = h
        for k, h in tmp_tag.items():
            self.fragments[k].haplotag = h
        return self.cal_overall_probability()

    # ------------------------------------------------------------------ phase.rs:1064-1296
    def save(self):
        return ([s.haplotype for s in self.candidate_snps], [s.genotype for s in self.candidate_snps],
                [f.haplotag for f in self.fragments])

    def load(self, st):
        for s, h, gt in zip(self.candidate_snps, st[0], st[1]):
            s.haplotype, s.genotype = h, gt
        for f, h in zip(self.fragments, st[2]):
            f.haplotag = h

    def phase(self, ld_weight_threshold, max_enum_snps):
        largest, best = float("-inf"), None
        snps = self.candidate_snps
        graph = self.divide_snps_into_blocks(ld_weight_threshold)
        self.n_cross = 0

        def attempt(prob):
            nonlocal largest, best
            if prob > largest:
                largest, best = prob, self.save()
        if len(snps) <= max_enum_snps:
            haps = [[1] * len(snps)]
            for ti in range(len(snps)):
                for tj in range(len(haps)):
                    t = list(haps[tj]); t[ti] = -t[ti]
                    haps.append(t)
            for hap in haps:
                for s, h in zip(snps, hap):
                    s.haplotype = h
                self.init_assignment()
                self.init_genotype()
                attempt(self.cross_optimize(set(), False, True))
            self.load(best)
        else:
            conserved = self.init_haplotypes_ld2(graph, ld_weight_threshold)
            self.init_genotype()
            self.init_assignment()
            attempt(self.cross_optimize(conserved, True, False))
            self.load(best)
            attempt(self.cross_optimize_by_block())
            self.load(best)
            for tidx in range(len(snps) // 4 + 1):
                flip = tidx % 2 == 1
                for s in snps:
                    rg = self.rnd()
                    if rg < 0.1:
                        s.haplotype = 1 if flip else -1
                    elif rg >= 0.9:
                        s.haplotype = -1 if flip else 1
                attempt(self.cross_optimize(conserved, False, False))
                self.load(best)
                for f in self.fragments:
                    if not f.for_phasing or f.haplotag == 0:
                        continue
                    if self.rnd() < 0.1:
                        f.haplotag *= -1
                attempt(self.cross_optimize(conserved, False, False))
                self.load(best)
            self.load(best)
        self.objective = largest

    # ------------------------------------------------------------------ snpfrags.rs:191-376
    def _eval_rescue(self, lst, min_phase_score, low_frac):
        snps = self.candidate_snps
        for ti in lst:
            s = snps[ti]
            if not s.snp_cover_fragments:
                s.single = True
                continue
            if s.variant_type != 1:
                s.non_selected = True
                continue
            sg, ps, pr = [], [], []
            h1 = h2 = 0
            for k in s.snp_cover_fragments:
                f = self.fragments[k]
                if not f.for_phasing or f.assignment == 0 or f.num_hete_links < self.min_linkers:
                    continue
                for fe in f.list:
                    if fe.snp_idx == ti:
                        if f.assignment == 1:
                            h1 += 1
                        elif f.assignment == 2:
                            h2 += 1
                        ps.append(fe.p); pr.append(fe.prob); sg.append(f.haplotag)
            if not sg or h1 < 2 or h2 < 2:
                s.single = True
                continue
            p1 = -10.0 * math.log10(1.0 - onp.cal_phase_score_log(1, 0, sg, ps, pr))
            p2 = -10.0 * math.log10(1.0 - onp.cal_phase_score_log(-1, 0, sg, ps, pr))
            s.single = False
            if max(p1, p2) >= float(min_phase_score):
                s.non_selected = False
                if low_frac:
                    s.cand_somatic = False
                s.rna_editing = False
                s.for_phasing = True
                for k in s.snp_cover_fragments:
                    f = self.fragments[k]
                    f.for_phasing = True
                    if f.haplotag == 0 or f.assignment == 0:
                        f.haplotag = -1 if self.rnd() < 0.5 else 1
                s.haplotype = 1 if p1 >= p2 else -1
                s.genotype, s.variant_type, s.phase_score = 0, 1, max(p1, p2)
            else:
                s.non_selected = True
                if low_frac:
                    s.cand_somatic = True
                    s.for_phasing = False
                else:
                    s.rna_editing = True

    def eval_rna_edit_var_phase(self, min_phase_score):
        self._eval_rescue(self.edit_snps, min_phase_score, False)

    def eval_low_frac_var_phase(self, min_phase_score):
        self._eval_rescue(self.somatic_snps, min_phase_score, True)

    # ------------------------------------------------------------------ snpfrags.rs:378-546
    def assign_snp_haplotype_genotype(self):
        for ti, s in enumerate(self.candidate_snps):
            if not s.for_phasing:
                s.non_selected = True
                continue
            if not s.snp_cover_fragments:
                s.single = True
                continue
            d = s.haplotype
            sg, ps, pr = [], [], []
            h1 = h2 = 0
            for k in s.snp_cover_fragments:
                f = self.fragments[k]
                if not f.for_phasing or f.num_hete_links < self.min_linkers:
                    continue
                if s.variant_type == 1 and f.assignment == 0:
                    continue
                for fe in f.list:
                    if fe.snp_idx == ti:
                        if f.assignment == 1:
                            h1 += 1
                        elif f.assignment == 2:
                            h2 += 1
                        ps.append(fe.p); pr.append(fe.prob); sg.append(f.haplotag)
            if not sg:
                s.non_selected = True
                continue
            q1 = onp.cal_delta_eta_sigma_log(d, 0, sg, ps, pr)
            q2 = onp.cal_delta_eta_sigma_log(-d, 0, sg, ps, pr)
            q3 = onp.cal_delta_eta_sigma_log(d, 1, sg, ps, pr)
            q4 = onp.cal_delta_eta_sigma_log(d, -1, sg, ps, pr)
            mx = max(q1, max(q2, max(q3, q4)))
            if q1 == mx:
                s.haplotype, s.genotype, s.variant_type = d, 0, 1
            elif q2 == mx:
                s.haplotype, s.genotype, s.variant_type = -d, 0, 1
            elif q3 == mx:
                s.haplotype, s.genotype, s.variant_type = d, 1, 0
            elif q4 == mx:
                s.haplotype, s.genotype = d, -1
                if s.variant_type not in (2, 3):
                    s.variant_type = 2
            else:
                raise ArithmeticError("genotype optimization failed")
            if s.genotype != 0:
                s.non_selected = True
                continue
            if sg and h1 >= 1 and h2 >= 1:
                s.phase_score = -10.0 * math.log10(1.0 - onp.cal_phase_score_log(s.haplotype, s.genotype, sg, ps, pr))
            else:
                s.phase_score = 0.19940219

    # ------------------------------------------------------------------ snpfrags.rs:548-625
    def assign_reads_haplotype(self, cutoff):
        snps = self.candidate_snps
        for f in self.fragments:
            if not f.for_phasing:
                continue
            d, e, ps, pr = [], [], [], []
            for fe in f.list:
                s = snps[fe.snp_idx]
                if not fe.phase_site and s.for_phasing:
                    fe.phase_site = True
                if not s.for_phasing or s.haplotype == 0 or s.genotype != 0:
                    continue
                ps.append(fe.p); pr.append(fe.prob); d.append(s.haplotype); e.append(s.genotype)
            if f.haplotag == 0 or not d:
                f.assignment = f.haplotag = 0
                continue
            q = onp.cal_sigma_delta_eta_log(f.haplotag, d, e, ps, pr)
            qn = onp.cal_sigma_delta_eta_log(-f.haplotag, d, e, ps, pr)
            if abs(q - qn) >= cutoff:
                if q >= qn:
                    f.assignment = 1 if f.haplotag == 1 else 2
                elif f.haplotag == 1:
                    f.assignment, f.haplotag = 2, -1
                else:
                    f.assignment, f.haplotag = 1, 1
            else:
                f.assignment = f.haplotag = 0

    # ------------------------------------------------------------------ snpfrags.rs:628-733
    def assign_phase_set(self, min_phase_score):
        snps = self.candidate_snps
        graph = Graph()
        for i, s in enumerate(snps):
            if s.genotype != 0 or s.variant_type != 1 or s.dense or s.rna_editing:
                continue
            if s.phase_score < float(min_phase_score):
                continue
            graph.add_node(i)
        for k, f in enumerate(self.fragments):
            if not f.for_phasing or f.assignment == 0:
                continue
            nodes = [fe.snp_idx for fe in f.list if graph.contains_node(fe.snp_idx)]
            if len(nodes) == 1:
                graph.append_edge(nodes[0], nodes[0], k)
            if len(nodes) >= 2:
                for j0 in range(len(nodes)):
                    for j1 in range(len(nodes)):
                        if j0 == j1:
                            continue
                        ap = [0, 0]
                        for fe in f.list:
                            if fe.snp_idx == nodes[j0]:
                                ap[0] = fe.p
                            elif fe.snp_idx == nodes[j1]:
                                ap[1] = fe.p
                        if snps[nodes[j0]].haplotype * snps[nodes[j1]].haplotype != ap[0] * ap[1]:
                            continue
                        graph.append_edge(nodes[j0], nodes[j1], k)
        read_ps = {}
        for comp in graph.kosaraju_scc():
            phase_id = 0
            for node in comp:
                if phase_id == 0:
                    phase_id = snps[node].pos + 1
                snps[node].phase_set = phase_id
                for _, _, frags in graph.edges(node):
                    for k in frags:
                        read_ps.setdefault(k, phase_id)
        return read_ps


def sum_block(delta, eta, sigma, ps, probs):          # cal_block_delta_eta_sigma_log, phase.rs:178-236
    tot = 0.0
    for i in range(len(delta)):
        tot += onp.cal_delta_eta_sigma_log(delta[i], eta[i], sigma[i], ps[i], probs[i])
    return tot


class Graph:
    """petgraph 0.6.4 GraphMap<usize, W, Undirected>, the operations the reference uses: nodes in insertion order,
    adjacency lists in edge-insertion order (IndexMap), remove_edge = swap_remove on both lists; kosaraju_scc = a
    DfsPostOrder over the nodes in insertion order, then Dfs from the nodes in reverse finishing order; Bfs marks a
    node discovered when it is pushed."""

    def __init__(self):
        self.nodes, self.adj, self.w = [], {}, {}

    def add_node(self, a):
        if a not in self.adj:
            self.adj[a] = []
            self.nodes.append(a)

    def contains_node(self, a):
        return a in self.adj

    @staticmethod
    def _key(a, b):
        return (a, b) if a <= b else (b, a)

    def contains_edge(self, a, b):
        return self._key(a, b) in self.w

    def add_edge(self, a, b, w):
        if self.contains_edge(a, b):
            self.w[self._key(a, b)] = w
            return
        self.add_node(a); self.adj[a].append(b)
        if a != b:
            self.add_node(b); self.adj[b].append(a)
        self.w[self._key(a, b)] = w

    def append_edge(self, a, b, k):     # assign_phase_set: the weight is the list of fragments on the edge
        if self.contains_edge(a, b):
            self.w[self._key(a, b)].append(k)
        else:
            self.add_edge(a, b, [k])

    def remove_edge(self, a, b):
        def rm(x, y):
            lst = self.adj[x]
            i = lst.index(y)
            lst[i] = lst[-1]
            lst.pop()
        rm(a, b)
        if a != b:
            rm(b, a)
        del self.w[self._key(a, b)]

    def all_edges(self):
        return [(a, b, w) for (a, b), w in self.w.items()]

    def edges(self, a):
        return [(a, b, self.w[self._key(a, b)]) for b in self.adj[a]]

    def kosaraju_scc(self):
        finished, seen, done = [], set(), set()
        for r in self.nodes:                       # DfsPostOrder (petgraph visit/traversal.rs)
            if r in seen:
                continue
            stack = [r]
            while stack:
                x = stack[-1]
                if x not in seen:
                    seen.add(x)
                    for y in self.adj[x]:
                        if y not in seen:
                            stack.append(y)
                else:
                    stack.pop()
                    if x not in done:
                        done.add(x)
                        finished.append(x)
        out, seen = [], set()
        for r in reversed(finished):               # Dfs: pop, skip visited, push unvisited neighbours
            if r in seen:
                continue
            comp, stack = [], [r]
            while stack:
                x = stack.pop()
                if x in seen:
                    continue
                seen.add(x)
                for y in self.adj[x]:
                    if y not in seen:
                        stack.append(y)
                comp.append(x)
            out.append(comp)
        return out

    def bfs(self, start):
        disc, queue, order = {start}, [start], []
        while queue:
            x = queue.pop(0)
            for y in self.adj[x]:
                if y not in disc:
                    disc.add(y)
                    queue.append(y)
            order.append(x)
        return order


def run_region(batch, g, prm, cands):
    """thread.rs:136-201 for region g.  cands: the candidate records of the region as of get_candidate_snps (the
    candidate half is pinned separately, tests/test_oracle_np.py), e.g. orc.Region(...).candidates().cands().
    Returns the SNPFrag after the post-phase steps and the read -> phase set map."""
    F = dict(edit=1, dense=2, het=4, fp=8, hom=16, single=32, nonsel=64, som=128)
    snps = [Snp(int(c["pos"]), chr(c["ref_base"]), (chr(c["allele1"]), chr(c["allele2"])), (float(c["af1"]), float(c["af2"])),
                int(c["variant_type"]), int(c["genotype"]), bool(c["flags"] & F["edit"]), bool(c["flags"] & F["dense"]),
                bool(c["flags"] & F["fp"]), bool(c["flags"] & F["hom"]), bool(c["flags"] & F["som"]), float(c["phase_score"]))
            for c in cands]
    sf = SNPFrag(snps, int(prm.min_linkers), int(prm.seed), int(batch.start0[g]))
    sf.get_fragments(batch, g)
    sf.fragmat_snapshot = [(f.read, [(fe.snp_idx, fe.base, fe.baseq, fe.p) for fe in f.list], f.num_hete_links, f.for_phasing)
                           for f in sf.fragments]
    if not snps:
        return sf, {}
    sf.init_haplotypes()        # thread.rs:162-163 (both are overwritten by phase(), but they consume draws)
    sf.init_assignment()
    sf.phase(1, int(prm.max_enum_snps))
    cut = float(prm.read_assign_cutoff)
    sf.assign_reads_haplotype(cut); sf.assign_snp_haplotype_genotype()
    sf.assign_reads_haplotype(cut); sf.assign_snp_haplotype_genotype()
    relaxed = float(prm.min_phase_score) - 3.0       # (f32 arithmetic in the reference: exact for the presets' values)
    sf.eval_rna_edit_var_phase(relaxed)
    sf.eval_low_frac_var_phase(relaxed)
    sf.assign_reads_haplotype(cut); sf.assign_snp_haplotype_genotype()
    read_ps = sf.assign_phase_set(float(prm.min_phase_score))
    return sf, read_ps
