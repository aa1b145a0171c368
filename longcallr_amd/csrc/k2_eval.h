// k2_eval.h — pass 1 of the candidate filters on ONE column's counts (candidate.rs:90-234 except :174-194): shared by k2_filter
// (k2_candidates.hip: a thread per column, counts read from the planes) and by k1_pileup's epilogue (k1_pileup.hip: the ONT presets
// take the filter verdict of a column straight from the tile's LDS planes -- round 6 -- so that k2_filter's pass over the planes goes).
#pragma once
#include "lcr_dev.h"

struct BinomTable { uint32_t reject[31]; };  // bit k of reject[n]: binomial_two_tailed(k, n, .5) < 0.05

static inline BinomTable make_binom_table() {
  // exact sums of C(n,k)/2^n restate statrs 0.16 Binomial(0.5, n).cdf for the n <= 30 the reference
  // allows (candidate.rs:37-47, 223-229)
  BinomTable t;
  for (int n = 0; n <= 30; n++) {
    t.reject[n] = 0;
    double cdf[32];
    double c = 1.0, s = 0.0, p2 = 1.0;
    for (int i = 0; i < n; i++) p2 *= 2.0;
    for (int k = 0; k <= n; k++) { s += c; cdf[k] = (k >= n) ? 1.0 : s / p2; c = c * (double)(n - k) / (double)(k + 1); }
    for (int k = 0; k <= n; k++) {
      double p;
      if (k == 0) p = 2.0 * cdf[0];
      else if (k == n) p = 2.0 * (1.0 - (n - 1 >= n ? 1.0 : cdf[n - 1]));
      else { double lo = cdf[k], hi = 1.0 - cdf[k - 1]; p = 2.0 * (lo < hi ? lo : hi); }
      if (p < 0.05) t.reject[n] |= (1u << k);
    }
  }
  return t;
}

#ifdef __HIPCC__
// candidate.rs:24-35 in f32 (no contraction: built with -ffp-contract=off)
__device__ __forceinline__ float strand_odds_ratio(int ref_fw, int ref_rv, int alt_fw, int alt_rv) {
  float x00 = (float)(ref_fw + 1), x01 = (float)(ref_rv + 1), x10 = (float)(alt_fw + 1), x11 = (float)(alt_rv + 1);
  float sym = (x00 * x11) / (x01 * x10) + (x01 * x10) / (x00 * x11);
  float ref_ratio = fminf(x00, x01) / fmaxf(x00, x01);
  float alt_ratio = fminf(x10, x11) / fmaxf(x10, x11);
  return logf(sym) + logf(ref_ratio) - logf(alt_ratio);
}

struct ColEval {
  bool pass;
  uint8_t ref_base, allele1, allele2, n_alt;
  uint32_t cnt1, cnt2, depth;
  float af1, af2;
};

// BaseFreq::get_two_major_alleles (util.rs:162-176): stable sort by count, descending
__device__ __forceinline__ void two_major(const uint32_t cnt[4], uint8_t ref_base, uint8_t* a1, uint32_t* c1, uint8_t* a2,
                                          uint32_t* c2) {
  const uint8_t ch[4] = {'A', 'C', 'G', 'T'};
  int order[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int rank = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) rank += (cnt[j] > cnt[i]) || (cnt[j] == cnt[i] && j < i);
    order[rank] = i;
  }
  int second = 1;
  if (ch[order[0]] != ref_base && ch[order[1]] != ref_base) {
    if (cnt[order[2]] == cnt[order[1]] && ch[order[2]] == ref_base) second = 2;
    else if (cnt[order[3]] == cnt[order[1]] && ch[order[3]] == ref_base) second = 3;
  }
  *a1 = ch[order[0]]; *c1 = cnt[order[0]]; *a2 = ch[order[second]]; *c2 = cnt[order[second]];
}

// cnt[4]: A, C, G, T of the column; get_d / get_n / get_fwd(k): deletions, intron bases, forward-strand count of allele k -- asked for
// only behind the early exits (k2_filter reads them from HBM then; K1 has them in registers)
template <class GetD, class GetN, class GetFwd>
__device__ __forceinline__ ColEval eval_counts(const uint32_t cnt[4], uint8_t ref_base, const DevParams& prm, const BinomTable& bt,
                                               GetD get_d, GetN get_n, GetFwd get_fwd) {
  ColEval ev;
  ev.pass = false;
  const uint32_t total = cnt[0] + cnt[1] + cnt[2] + cnt[3];
  ev.depth = total;
  if (total < prm.min_depth || total > prm.max_depth) return ev;  // candidate.rs:90-94
  // a reference byte other than upper-case ACGT never reaches a candidate (candidate.rs:132,243-265)
  if (!(ref_base == 'A' || ref_base == 'C' || ref_base == 'G' || ref_base == 'T')) return ev;
  two_major(cnt, ref_base, &ev.allele1, &ev.cnt1, &ev.allele2, &ev.cnt2);
  ev.af1 = (float)ev.cnt1 / (float)total;
  ev.af2 = (float)ev.cnt2 / (float)total;
  ev.ref_base = ref_base;
  uint8_t alt0; uint32_t altc0; float altf0;
  if (ev.allele1 == ref_base) { ev.n_alt = 1; alt0 = ev.allele2; altc0 = ev.cnt2; altf0 = ev.af2; }
  else if (ev.allele2 == ref_base) { ev.n_alt = 1; alt0 = ev.allele1; altc0 = ev.cnt1; altf0 = ev.af1; }
  else { ev.n_alt = 2; alt0 = ev.allele1; altc0 = ev.cnt1; altf0 = ev.af1; }
  if (ev.n_alt == 1) {  // candidate.rs:142-155
    if (total < 200 && altf0 < prm.low_frac_cut) return ev;
    if (total >= 200 && altc0 < prm.low_cnt_cut) return ev;
  }
  const uint32_t d = get_d(), n = get_n();
  if (d >= altc0) return ev;  // candidate.rs:165
  if ((float)(ev.cnt1 + ev.cnt2) / (float)(total + d + n) < prm.min_af_intron) return ev;  // candidate.rs:170
  if (prm.use_strand_bias) {  // candidate.rs:199-234
    uint32_t fwd[4];
#pragma unroll
    for (int k = 0; k < 4; k++) fwd[k] = get_fwd(k);
    const int ri = base_code(ref_base), a0 = base_code(alt0);
    const int ref_fw = (int)fwd[ri], ref_rv = (int)(cnt[ri] - fwd[ri]);
    const int alt_fw = (int)fwd[a0], alt_rv = (int)(cnt[a0] - fwd[a0]);
    float sor = strand_odds_ratio(ref_fw, ref_rv, alt_fw, alt_rv);
    if (ev.n_alt == 2) {
      const int a1i = base_code(ev.allele2);
      sor = fmaxf(sor, strand_odds_ratio(ref_fw, ref_rv, (int)fwd[a1i], (int)(cnt[a1i] - fwd[a1i])));
    }
    if (sor > prm.sor_threshold) return ev;
    if (ev.n_alt == 1) {
      if (alt_fw + alt_rv <= 30 && ((bt.reject[alt_fw + alt_rv] >> alt_fw) & 1u)) return ev;
      if (alt_fw * alt_rv == 0) return ev;
    }
  }
  ev.pass = true;
  return ev;
}
#endif
