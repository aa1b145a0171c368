"""VCF body text of the phased candidates — the parity diff surface.

Host formatting only; mirrors SNPFrag::output_phased_vcf (reference src/vcf.rs:27-306) and the
record writer of src/thread.rs:266-303 (records without an ALT allele are silently skipped).
"""
from . import _abi


def _as_i32(x):  # Rust `f64 as i32`: saturating, NaN -> 0
    if x != x:
        return 0
    if x >= 2147483647.0:
        return 2147483647
    if x <= -2147483648.0:
        return -2147483648
    return int(x)


def format_records(cands, chrom, min_phase_score):
    out = []
    for s in cands:
        ref, a1, a2 = chr(s["ref_base"]), chr(s["allele1"]), chr(s["allele2"])
        fl, vt, gtp = int(s["flags"]), int(s["variant_type"]), int(s["genotype"])
        alt, af = [], [0.0, 0.0]

        def one_alt():
            if a1 != ref:
                alt[:] = [a1]; af[0] = float(s["af1"])
            elif a2 != ref:
                alt[:] = [a2]; af[0] = float(s["af2"])

        def two_alt():
            alt[:] = [a1, a2]; af[0] = float(s["af1"]); af[1] = float(s["af2"])

        def by_genotype():
            nonlocal gt, filt
            if gtp in (-1, 1):
                one_alt()
                gt, filt = ("1/1", "PASS") if gtp == -1 else ("0/0", "HomRef")
            elif gtp == 0:
                two_alt()
                gt, filt = "1/2", "Multiallelic"

        gt, filt = "0/0", ""
        gq, dp, qual = _as_i32(float(s["gq"])), int(s["depth"]), _as_i32(float(s["qual"]))
        if fl & _abi.F_DENSE:  # vcf.rs:31-78
            if vt in (1, 2):
                one_alt()
            elif vt == 3:
                two_alt()
            if vt not in (1, 2, 3):
                continue
            gt = {1: "0/1", 2: "1/1", 3: "1/2"}[vt]
            filt, info, fmt = "dn", "RDS=dense_snp", "GT:GQ:DP:AF"
            sample = ("%s:%d:%d:%.2f,%.2f" % (gt, gq, dp, af[0], af[1]) if vt == 3
                      else "%s:%d:%d:%.2f" % (gt, gq, dp, af[0]))
        elif fl & _abi.F_NON_SELECTED:  # vcf.rs:80-174
            info, fmt = "RDS=noselect", "GT:GQ:DP:AF"
            if fl & _abi.F_RNA_EDIT:
                if vt not in (1, 2):
                    continue
                one_alt()
                filt = "RnaEdit"
                gt = "0/1" if vt == 1 else "1/1"
                sample = "%s:%d:%d:%.2f" % (gt, gq, dp, af[0])
            else:
                if vt in (0, 1, 2):
                    one_alt()
                    gt, filt = {0: ("0/0", "HomRef"), 1: ("0/1", "LowQual"), 2: ("1/1", "PASS")}[vt]
                else:
                    by_genotype()
                sample = ("%s:%d:%d:%.2f" % (gt, gq, dp, af[0]) if gt in ("0/0", "0/1", "1/1")
                          else "%s:%d:%d:%.2f,%.2f" % (gt, gq, dp, af[0], af[1]))
        else:  # vcf.rs:175-303
            info, fmt = "RDS=select", "GT:GQ:PS:DP:AF:PQ"
            if float(s["phase_score"]) >= float(min_phase_score):
                if vt == 1:
                    one_alt()
                    gt, filt = ("0|1" if int(s["haplotype"]) == 1 else "1|0"), "PASS"
            else:
                if vt in (0, 1, 2):
                    one_alt()
                    gt, filt = {0: ("0/0", "HomRef"), 1: ("0/1", "LowQual"), 2: ("1/1", "PASS")}[vt]
                else:
                    by_genotype()
            ps = str(int(s["phase_set"])) if int(s["phase_set"]) != 0 else "."
            pq = float(s["phase_score"])
            sample = ("%s:%d:%s:%d:%.2f:%.2f" % (gt, gq, ps, dp, af[0], pq) if gt in ("0/0", "0/1", "1/1", "0|1", "1|0")
                      else "%s:%d:%s:%d:%.2f,%.2f:%.2f" % (gt, gq, ps, dp, af[0], af[1], pq))
        if len(alt) not in (1, 2):
            continue
        out.append("%s\t%d\t.\t%s\t%s\t%d\t%s\t%s\t%s\t%s\n" % (
            chrom, int(s["pos"]) + 1, ref, ",".join(alt), qual, filt, info, fmt, sample))
    return "".join(out)
