// k4_post.h — the post-phase sequence of thread.rs:168-201 on the device, written once for two scopes:
//   assign_reads_haplotype + assign_het_var_haplotype (x2), eval_rna_edit_var_phase, eval_low_frac_var_phase,
//   assign_reads_haplotype + assign_het_var_haplotype, assign_phase_set   (snpfrags.rs:191-733).
// WgScope  (k4_post, k4_phase.hip): one workgroup per region, the region's fragment rows in LDS (16-bit indices);
// GridScope (k4_gpost, k4_grid.hip): all workgroups on one region, the same arrays in HBM (32-bit indices) — for
// regions beyond the LDS image (config C5).  These are f64 sum-of-ratio decisions: every log10(eps) / log10(1-eps)
// term comes from the table of libm values the host path uses (kernel argument), sums run in the reference's
// observation order (a read's entries in column order, a SNP's reads in row order) and -ffp-contract=off keeps
// a*b+c unfused, so the decisions are the host path's bit for bit; only phase_score's final log10 is the device libm.
#pragma once
#include <climits>
#include <type_traits>
#include "k4_dev.h"

namespace {

// the region image the steps work on: LDS (IDX = uint16_t) or HBM (IDX = int32_t)
template <class IDX>
struct PostView {
  int g, S, nrow, E, F;          // region, candidates, fragment rows, entries, phasing rows
  int r0, c0;
  double *le, *l1e;              // LDS in both scopes
  double *sps, *rpa, *rpb;
  uint32_t *sflags, *soflags; int32_t* parent;
  IDX *rptr, *ecol, *erow, *cent, *ccptr;
  uint8_t* ev;
  int8_t* tag; uint8_t *asg, *fp, *lok, *dirty;
  int8_t *shap, *sgt, *svt; uint8_t* rcode;
  lcr_candidate* cand;
  double* stage;                 // LDS: per wave 4 * 65 doubles
  int32_t *fdirt = nullptr, *minf = nullptr, *ndraw = nullptr, *gwords = nullptr;   // grid scope: the rescue lists' parallel commit
};

constexpr int POST_SSTR = 65;   // stage row stride in doubles (lanes a = 0..3 read different banks)

// everything after the staging of the region image; `mark` records step timestamps (profiling)
template <class SC, class IDX, class Mark>
__device__ void post_run(SC& sc, const PostIn& in, const PostLut& lut, PostView<IDX>& v, Mark mark) {
  const int lane = threadIdx.x & 63;
  const int S = v.S, nrow = v.nrow;
  double* le = v.le; double* l1e = v.l1e;
  double* sps = v.sps; double* rpa = v.rpa; double* rpb = v.rpb;
  uint32_t* sflags = v.sflags; uint32_t* soflags = v.soflags; int32_t* parent = v.parent;
  IDX* rptr = v.rptr; IDX* ecol = v.ecol; IDX* erow = v.erow; IDX* cent = v.cent; IDX* ccptr = v.ccptr;
  uint8_t* ev = v.ev; int8_t* tag = v.tag; uint8_t* asg = v.asg; uint8_t* fp = v.fp; uint8_t* lok = v.lok; uint8_t* dirty = v.dirty;
  int8_t* shap = v.shap; int8_t* sgt = v.sgt; int8_t* svt = v.svt; uint8_t* rcode = v.rcode;
  lcr_candidate* cand = v.cand;

  auto lg = [&](int sigma, int delta, int eta, uint8_t x) -> double {   // log10(aki(...)), phase.rs:32-49
    const int pp = (x & 32) ? 1 : -1, xx = eta == 0 ? sigma * delta : eta;
    return pp == xx ? l1e[x & 31] : le[x & 31];
  };
  // Ordered sums over the observations of SNP column ti, one wave per column: 64 column entries at a time
  // are loaded and filtered by the lanes (lane <-> entry) and each lane stages its NACC terms (or +0.0, the
  // exact identity here: the sums start at +0.0 and every term is a finite log) in LDS; then lane a adds the
  // 64 staged terms of accumulator a in entry order.  The additions are the host's, in the host's order;
  // only the loads and the filter run in parallel.  Results are returned wave-uniform.
  double* stg = v.stage + (threadIdx.x >> 6) * (4 * POST_SSTR);
  auto col_sums = [&](int ti, bool skip_unassigned, auto term, double* acc, int nacc, int& hap1, int& hap2, int& nobs) {
    hap1 = hap2 = nobs = 0;
    double mine = 0.0;   // lane a < nacc: running sum of accumulator a
    const int kb = ccptr[ti], ke = ccptr[ti + 1];
    for (int k0 = kb; k0 < ke; k0 += 64) {
      const int k = k0 + lane;
      bool keep = false; int r = 0, e = 0;
      if (k < ke) { e = cent[k]; r = erow[e]; keep = fp[r] && lok[r] && !(skip_unassigned && asg[r] == 0); }
      double t[4] = {0.0, 0.0, 0.0, 0.0};
      if (keep) term((int)tag[r], ev[e], t);
      for (int a = 0; a < nacc; a++) stg[a * POST_SSTR + lane] = t[a];
      const unsigned long long km = __ballot(keep);
      hap1 += __popcll(__ballot(keep && asg[r] == 1)); hap2 += __popcll(__ballot(keep && asg[r] == 2));
      nobs += __popcll(km);
      wave_lds_sync();
      const int nk = min(64, ke - k0);
      if (lane < nacc) {
        const double* src = stg + lane * POST_SSTR;
        int j = 0;
        for (; j + 8 <= nk; j += 8) {
          double x[8];
#pragma unroll
          for (int u = 0; u < 8; u++) x[u] = src[j + u];
#pragma unroll
          for (int u = 0; u < 8; u++) mine += x[u];
        }
        for (; j < nk; j++) mine += src[j];
      }
      wave_lds_sync();
    }
    for (int a = 0; a < nacc; a++)
      acc[a] = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(mine), a), __builtin_amdgcn_readlane(__double2loint(mine), a));
  };
  // phase.rs:238-255 over the kept observations of column ti (wave-uniform result)
  auto psl = [&](int ti, int delta_i, int eta_i, bool skip_unassigned) -> double {
    double q[3]; int h1, h2, nb;
    col_sums(ti, skip_unassigned, [&](int sg, uint8_t x, double* t) {
      t[0] = lg(sg, delta_i, eta_i, x); t[1] = lg(sg, 1, eta_i, x); t[2] = lg(sg, -1, eta_i, x);
    }, q, 3, h1, h2, nb);
    return 1.0 - q[0] / (q[1] + q[2]);
  };
  // snpfrags.rs:548-625
  auto reads_hap = [&]() {
    for (int r = sc.tid(); r < nrow; r += sc.nt()) {
      if (!fp[r]) continue;
      const int sigma_k = tag[r];
      double q1 = 0, q2 = 0, q3 = 0, n1 = 0;
      int n = 0;
      for (int e = rptr[r]; e < (int)rptr[r + 1]; e++) {
        const int i = ecol[e];
        if (!(sflags[i] & LCR_F_FOR_PHASING) || shap[i] == 0 || sgt[i] != 0) continue;
        q1 += lg(sigma_k, shap[i], 0, ev[e]);
        n1 += lg(-sigma_k, shap[i], 0, ev[e]);
        n++;
      }
      if (sigma_k == 0 || n == 0) { asg[r] = 0; tag[r] = 0; continue; }
      for (int e = rptr[r]; e < (int)rptr[r + 1]; e++) {
        const int i = ecol[e];
        if (!(sflags[i] & LCR_F_FOR_PHASING) || shap[i] == 0 || sgt[i] != 0) continue;
        q2 += lg(1, shap[i], 0, ev[e]); q3 += lg(-1, shap[i], 0, ev[e]);
      }
      const double q = 1.0 - q1 / (q2 + q3), qn = 1.0 - n1 / (q2 + q3);
      if (fabs(q - qn) >= in.cutoff) {
        if (q >= qn) asg[r] = sigma_k == 1 ? 1 : 2;
        else if (sigma_k == 1) { asg[r] = 2; tag[r] = -1; }
        else { asg[r] = 1; tag[r] = 1; }
      } else { asg[r] = 0; tag[r] = 0; }
    }
    sc.sync();
  };
  // snpfrags.rs:378-546, one wave per SNP (all lanes hold the same values; lane 0 writes).  Returns (to every thread)
  // whether a SNP's haplotype / genotype / variant type really changed (see the pass sequence at the end).
  auto snp_hap = [&]() -> bool {
    int chg = 0;
    for (int ti = sc.wave(); ti < S; ti += sc.nwaves()) {
      if (!(sflags[ti] & LCR_F_FOR_PHASING)) { if (lane == 0) sflags[ti] |= LCR_F_NON_SELECTED; continue; }
      if (ccptr[ti] == ccptr[ti + 1]) { if (lane == 0) sflags[ti] |= LCR_F_SINGLE; continue; }
      const int delta_i = shap[ti];
      const bool het_skip = svt[ti] == 1;
      int hap1, hap2, nobs;
      double sum[4];   // het_d, het_nd, homref, homvar
      col_sums(ti, het_skip, [&](int sg, uint8_t x, double* t) {
        t[0] = lg(sg, delta_i, 0, x); t[1] = lg(sg, -delta_i, 0, x); t[2] = lg(sg, delta_i, 1, x); t[3] = lg(sg, delta_i, -1, x);
      }, sum, 4, hap1, hap2, nobs);
      if (nobs == 0) { if (lane == 0) sflags[ti] |= LCR_F_NON_SELECTED; continue; }
      const double het_d = sum[0], het_nd = sum[1], homref = sum[2], homvar = sum[3];
      const double p_het = lut.log_theta - (double)(uint32_t)nobs * lut.log2;
      auto score = [&](int sign, int eta_i) -> double {   // cal_delta_eta_sigma_log, phase.rs:128-176
        const double hd = sign > 0 ? het_d : het_nd, hn = sign > 0 ? het_nd : het_d;
        double q1 = eta_i == 0 ? hd : (eta_i == 1 ? homref : homvar);
        q1 += eta_i == 0 ? p_het : (eta_i == 1 ? lut.p_homref : lut.p_homvar);
        const double q2 = homvar + lut.p_homvar, q3 = hd + p_het, q4 = homref + lut.p_homref, q5 = hn + p_het;
        return 1.0 - q1 / (q2 + q3 + q4 + q5);
      };
      const double q1 = score(1, 0), q2 = score(-1, 0), q3 = score(1, 1), q4 = score(1, -1);
      const double mx = fmax(q1, fmax(q2, fmax(q3, q4)));
      int nh = delta_i, ng_ = 0, nv = svt[ti];
      if (q1 == mx) { nh = delta_i; ng_ = 0; nv = 1; }
      else if (q2 == mx) { nh = -delta_i; ng_ = 0; nv = 1; }
      else if (q3 == mx) { nh = delta_i; ng_ = 1; nv = 0; }
      else if (q4 == mx) { nh = delta_i; ng_ = -1; if (nv != 2 && nv != 3) nv = 2; }
      else continue;  // NaN scores: the reference panics here
      double ps = sps[ti];
      uint32_t fl = sflags[ti];
      if (ng_ != 0) fl |= LCR_F_NON_SELECTED;
      else if (hap1 >= 1 && hap2 >= 1) {
        // phase_score_log(nh, 0): q2 / q3 = sums of lg(sigma, +1 / -1, 0, v), q1 = the one of nh -- the very
        // addition sequences of het_d / het_nd above (same observations, same order) when delta_i = +-1
        if (delta_i == 1 || delta_i == -1) {
          const double s2 = delta_i == 1 ? het_d : het_nd, s3 = delta_i == 1 ? het_nd : het_d;
          ps = -10.0 * log10(1.0 - (1.0 - (nh == 1 ? s2 : s3) / (s2 + s3)));
        } else ps = -10.0 * log10(1.0 - psl(ti, nh, ng_, het_skip));
      }
      else ps = 0.19940219;
      if (lane == 0) {
        if (shap[ti] != nh || sgt[ti] != ng_ || svt[ti] != nv) chg = 1;   // (fl only ORs bits neither half reads)
        shap[ti] = (int8_t)nh; sgt[ti] = (int8_t)ng_; svt[ti] = (int8_t)nv; sflags[ti] = fl; sps[ti] = ps;
      }
    }
    return sc.sync_or(chg) != 0;
  };
  // snpfrags.rs:191-376.  The list is walked in index order and a successful rescue changes fp / tag of
  // its reads (and draws random numbers), which later list members see.  All pending members are
  // evaluated in parallel (a wave each) against the current state; the first wave of the scope then commits them in
  // order, marking the rows a success really changes (fp 0 -> 1, tag drawn): a later member whose column holds
  // no such row was evaluated on the state the reference would show it and is committed in the same
  // round, the first member that does see a changed row starts the next round.
  unsigned long long ctr = 0;   // first wave: draws so far (thread.rs call order, see PhaseHost::run)
  {
    const unsigned long long Su = (unsigned long long)S, Fu = (unsigned long long)v.F;
    ctr = (uint32_t)S <= in.max_enum_snps ? Su + Fu + (1ull << S) * Fu : 2 * (Su + Fu) + (Su / 4 + 1) * (Su + Fu);
  }
  auto rescue = [&](uint32_t list_flag, float min_ps, bool low_frac, uint64_t rseed) -> bool {   // true: some SNP state changed
    int start = 0, chg = 0;
    if constexpr (std::is_same<SC, GridScope>::value) {
      // GRID SCOPE: the same rounds with the commit in parallel as well (one wave walking the members one after the other, three
      // dependent trips to memory per 512 entries of a column, was 12.5 of k4_gpost's 20 ms on C5).  Whether a row is changed by a
      // success depends on the row alone (touch = draw || !fp), so: (C) every success marks the rows it would change with its index
      // (atomicMin: the FIRST member to change the row) and counts its draws; (D) a member is stale iff one of its rows carries a
      // smaller index than its own -- the first stale member ends the round, exactly the serial rule; (F) the members before it commit
      // side by side: two committed successes share only rows neither changes (a shared row that one of them changes makes the later one
      // stale), and a success's first draw is the draws of the successes before it.  A pending member is evaluated again only if one
      // of its rows was changed since (its row minimum below the round's end): the others' scores are those the serial walk would see.
      int32_t* const fdirt = v.fdirt; int32_t* const minf = v.minf; int32_t* const ndraw = v.ndraw; int32_t* const gw = v.gwords;
      for (int ti = sc.tid(); ti < S; ti += sc.nt()) minf[ti] = -1;   // (not evaluated yet)
      for (int r = sc.tid(); r < nrow; r += sc.nt()) fdirt[r] = INT_MAX;
      // the round's end lives in gw[round & 1]: the word of the NEXT round is reset while stragglers may still be reading this round's
      // (a workgroup leaves the barrier below microseconds after another; one word reset in place could be read as S by a late one)
      if (sc.tid() == 0) { gw[0] = S; gw[1] = S; }
      int rnd = 0;
      sc.sync();
      for (;; rnd ^= 1) {
        for (int ti = start + sc.wave(); ti < S; ti += sc.nwaves()) {   // (B) + (C): the same wave has member ti in every loop
          if (!(soflags[ti] & list_flag)) { if (lane == 0) rcode[ti] = 0; continue; }
          uint8_t code = rcode[ti];
          if (minf[ti] < start) {   // (else: no row of its column changed since it was evaluated)
          code = 0;
          if (ccptr[ti] == ccptr[ti + 1]) code = 1;
          else if (svt[ti] != 1) code = 2;
          else {
            double q[2]; int hap1, hap2, nobs;
            col_sums(ti, true, [&](int sg, uint8_t x, double* t) { t[0] = lg(sg, 1, 0, x); t[1] = lg(sg, -1, 0, x); },
                     q, 2, hap1, hap2, nobs);
            if (nobs == 0 || hap1 < 2 || hap2 < 2) code = 3;
            else {
              const double pa = -10.0 * log10(1.0 - (1.0 - q[0] / (q[0] + q[1])));
              const double pb = -10.0 * log10(1.0 - (1.0 - q[1] / (q[0] + q[1])));
              if (lane == 0) { rpa[ti] = pa; rpb[ti] = pb; }
              code = fmax(pa, pb) >= (double)min_ps ? 4 : 5;
            }
          }
          if (lane == 0) rcode[ti] = code;
          }
          if (code != 4) continue;
          int nd = 0;
          const int kb = (int)ccptr[ti], ke = (int)ccptr[ti + 1];
          for (int k0 = kb; k0 < ke; k0 += 64 * 8) {
            int rr[8]; int tg[8], ag[8], fv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int k = k0 + 64 * u + lane; rr[u] = k < ke ? (int)erow[cent[k]] : -1; }
#pragma unroll
            for (int u = 0; u < 8; u++) { const int r = max(rr[u], 0); tg[u] = tag[r]; ag[u] = asg[r]; fv[u] = fp[r]; }
#pragma unroll
            for (int u = 0; u < 8; u++) {
              const bool in_col = rr[u] >= 0;
              const bool draw = in_col && (tg[u] == 0 || ag[u] == 0);
              if (in_col && (draw || !fv[u])) atomicMin(&fdirt[rr[u]], ti);
              nd += __popcll(__ballot(draw));
            }
          }
          if (lane == 0) ndraw[ti] = nd;
        }
        sc.sync();
        for (int ti = start + sc.wave(); ti < S; ti += sc.nwaves()) {   // (D)
          if (rcode[ti] < 3) continue;   // (codes 1 / 2 do not look at the rows)
          int m = INT_MAX;
          const int kb = (int)ccptr[ti], ke = (int)ccptr[ti + 1];
          for (int k0 = kb; k0 < ke; k0 += 64 * 8) {
            int rr[8];
#pragma unroll
            for (int u = 0; u < 8; u++) { const int k = k0 + 64 * u + lane; rr[u] = k < ke ? (int)erow[cent[k]] : -1; }
#pragma unroll
            for (int u = 0; u < 8; u++) if (rr[u] >= 0) m = min(m, fdirt[rr[u]]);
          }
#pragma unroll
          for (int d = 32; d >= 1; d >>= 1) m = min(m, __shfl_xor(m, d, 64));
          if (lane == 0) { minf[ti] = m; if (m < ti) atomicMin(&gw[rnd], ti); }
        }
        sc.sync();
        const int end = gw[rnd];
        {   // draws of the successes of [start, t): every wave sums what it needs (a few hundred members at most)
          auto draws_before = [&](int t) -> unsigned long long {
            unsigned long long n = 0;
            for (int x = start + lane; x < t; x += 64) if (rcode[x] == 4) n += (unsigned long long)ndraw[x];
            return (unsigned long long)wave_sum_ll_dpp((long long)n);
          };
          for (int ti = start + sc.wave(); ti < end; ti += sc.nwaves()) {   // (F)
            const uint8_t code = rcode[ti];
            if (code == 0) continue;
            if (code == 4) {
              unsigned long long c = ctr + draws_before(ti);
              const int kb4 = (int)ccptr[ti], ke4 = (int)ccptr[ti + 1];
              for (int k0 = kb4; k0 < ke4; k0 += 64 * 8) {
                int rr[8]; int tg[8], ag[8];
#pragma unroll
                for (int u = 0; u < 8; u++) { const int k = k0 + 64 * u + lane; rr[u] = k < ke4 ? (int)erow[cent[k]] : -1; }
#pragma unroll
                for (int u = 0; u < 8; u++) { const int r = max(rr[u], 0); tg[u] = tag[r]; ag[u] = asg[r]; }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                  if (k0 + 64 * u >= ke4) break;
                  const bool in_col = rr[u] >= 0;
                  const int r = max(rr[u], 0);
                  const bool draw = in_col && (tg[u] == 0 || ag[u] == 0);
                  const unsigned long long dm = __ballot(draw);
                  if (in_col) fp[r] = 1;
                  if (draw) tag[r] = u01(rseed, c + (unsigned long long)__popcll(dm & ((1ull << lane) - 1ull))) < 0.5 ? -1 : 1;
                  c += (unsigned long long)__popcll(dm);
                }
              }
            }
            if (lane == 0) {
              const uint32_t fl_old = sflags[ti];
              if (code == 1 || code == 3) sflags[ti] |= LCR_F_SINGLE;
              else if (code == 2) sflags[ti] |= LCR_F_NON_SELECTED;
              else if (code == 5) {
                sflags[ti] &= ~(uint32_t)LCR_F_SINGLE;
                sflags[ti] |= LCR_F_NON_SELECTED;
                if (low_frac) { sflags[ti] |= LCR_F_CAND_SOMATIC; sflags[ti] &= ~(uint32_t)LCR_F_FOR_PHASING; }
                else sflags[ti] |= LCR_F_RNA_EDIT;
              } else {   // rescued
                sflags[ti] &= ~(uint32_t)(LCR_F_SINGLE | LCR_F_NON_SELECTED | LCR_F_RNA_EDIT);
                if (low_frac) sflags[ti] &= ~(uint32_t)LCR_F_CAND_SOMATIC;
                sflags[ti] |= LCR_F_FOR_PHASING;
                shap[ti] = rpa[ti] >= rpb[ti] ? 1 : -1;
                sgt[ti] = 0; svt[ti] = 1; sps[ti] = fmax(rpa[ti], rpb[ti]);
                chg = 1;
              }
              if ((sflags[ti] ^ fl_old) & LCR_F_FOR_PHASING) chg = 1;
            }
          }
          ctr += draws_before(end);
        }
        if (in.dbg_clk && sc.tid() == 0) in.dbg_clk[(size_t)v.g * 16 + (low_frac ? 12 : 11)] += 1;   // LCR_PHASE_PROF: rounds of the list
        if (end < S) {   // the next round's marks start from a clean slate; its end word is the OTHER one (last read a whole round ago)
          for (int r = sc.tid(); r < nrow; r += sc.nt()) fdirt[r] = INT_MAX;
          if (sc.tid() == 0) gw[rnd ^ 1] = S;
        }
        sc.sync();
        start = end;
        if (start >= S) break;
      }
      return sc.sync_or(chg) != 0;
    }
    for (;;) {
      for (int r = sc.tid(); r < nrow; r += sc.nt()) dirty[r] = 0;
      for (int ti = start + sc.wave(); ti < S; ti += sc.nwaves()) {
        uint8_t code = 0;
        if (soflags[ti] & list_flag) {
          if (ccptr[ti] == ccptr[ti + 1]) code = 1;
          else if (svt[ti] != 1) code = 2;
          else {
            double q[2]; int hap1, hap2, nobs;   // gather(need_assigned); phase_score_log(+-1, 0) share q2 / q3
            col_sums(ti, true, [&](int sg, uint8_t x, double* t) { t[0] = lg(sg, 1, 0, x); t[1] = lg(sg, -1, 0, x); },
                     q, 2, hap1, hap2, nobs);
            if (nobs == 0 || hap1 < 2 || hap2 < 2) code = 3;
            else {
              const double pa = -10.0 * log10(1.0 - (1.0 - q[0] / (q[0] + q[1])));
              const double pb = -10.0 * log10(1.0 - (1.0 - q[1] / (q[0] + q[1])));
              if (lane == 0) { rpa[ti] = pa; rpb[ti] = pb; }
              code = fmax(pa, pb) >= (double)min_ps ? 4 : 5;
            }
          }
        }
        if (lane == 0) rcode[ti] = code;
      }
      sc.sync();
      int ti = start;
      if (sc.wave() == 0) {
        bool changed = false;   // (wave-uniform) a success of this round changed some row
        for (; ti < S; ti++) {
          const uint8_t code = rcode[ti];
          if (code == 0) continue;
          if (code >= 3 && changed) {   // codes 1 / 2 do not look at the rows
            // (eight independent cent -> erow -> dirty chains per lane in flight: one at a time, the three dependent loads per
            // 64 entries of a 1 770-entry column made this check -- by ONE wave, member after member -- 17 of k4_gpost's 25 ms on C5)
            bool stale = false;
            const int kb = (int)ccptr[ti], ke = (int)ccptr[ti + 1];
            for (int k0 = kb; k0 < ke; k0 += 64 * 8) {
              int rr[8];
#pragma unroll
              for (int u = 0; u < 8; u++) { const int k = k0 + 64 * u + lane; rr[u] = k < ke ? (int)erow[cent[k]] : -1; }
#pragma unroll
              for (int u = 0; u < 8; u++) if (rr[u] >= 0 && dirty[rr[u]]) stale = true;
            }
            if (__any(stale)) break;    // evaluated on an outdated state: next round starts here
          }
          if (code == 4) {
            // rows in column order: the draws keep their order.  The rows of a column are distinct, so the states of 8 x 64 of them
            // are fetched together (cent -> erow -> tag / asg / fp one 64-entry slice at a time was three dependent trips to memory
            // per slice, by the one committing wave: most of k4_gpost's 25 ms on C5) and then walked slice by slice.
            const int kb4 = (int)ccptr[ti], ke4 = (int)ccptr[ti + 1];
            for (int k0 = kb4; k0 < ke4; k0 += 64 * 8) {
              int rr[8]; int tg[8], ag[8], fv[8];
#pragma unroll
              for (int u = 0; u < 8; u++) { const int k = k0 + 64 * u + lane; rr[u] = k < ke4 ? (int)erow[cent[k]] : -1; }
#pragma unroll
              for (int u = 0; u < 8; u++) { const int r = max(rr[u], 0); tg[u] = tag[r]; ag[u] = asg[r]; fv[u] = fp[r]; }
#pragma unroll
              for (int u = 0; u < 8; u++) {
                if (k0 + 64 * u >= ke4) break;
                const bool in_col = rr[u] >= 0;
                const int r = max(rr[u], 0);
                const bool draw = in_col && (tg[u] == 0 || ag[u] == 0);
                const unsigned long long dm = __ballot(draw);
                const bool touch = in_col && (draw || !fv[u]);
                if (touch) dirty[r] = 1;
                if (__any(touch)) changed = true;
                if (in_col) fp[r] = 1;
                if (draw) tag[r] = u01(rseed, ctr + (unsigned long long)__popcll(dm & ((1ull << lane) - 1ull))) < 0.5 ? -1 : 1;
                ctr += (unsigned long long)__popcll(dm);
              }
            }
          }
          if (lane == 0) {
            const uint32_t fl_old = sflags[ti];
            if (code == 1 || code == 3) sflags[ti] |= LCR_F_SINGLE;
            else if (code == 2) sflags[ti] |= LCR_F_NON_SELECTED;
            else if (code == 5) {
              sflags[ti] &= ~(uint32_t)LCR_F_SINGLE;
              sflags[ti] |= LCR_F_NON_SELECTED;
              if (low_frac) { sflags[ti] |= LCR_F_CAND_SOMATIC; sflags[ti] &= ~(uint32_t)LCR_F_FOR_PHASING; }
              else sflags[ti] |= LCR_F_RNA_EDIT;
            } else {   // rescued
              sflags[ti] &= ~(uint32_t)(LCR_F_SINGLE | LCR_F_NON_SELECTED | LCR_F_RNA_EDIT);
              if (low_frac) sflags[ti] &= ~(uint32_t)LCR_F_CAND_SOMATIC;
              sflags[ti] |= LCR_F_FOR_PHASING;
              shap[ti] = rpa[ti] >= rpb[ti] ? 1 : -1;
              sgt[ti] = 0; svt[ti] = 1; sps[ti] = fmax(rpa[ti], rpb[ti]);
              chg = 1;
            }
            if ((sflags[ti] ^ fl_old) & LCR_F_FOR_PHASING) chg = 1;
          }
          wave_mem_sync();
        }
      }
      start = sc.bcast(ti);   // (a barrier: the committed state is visible to everybody)
      if (in.dbg_clk && sc.tid() == 0) in.dbg_clk[(size_t)v.g * 16 + (low_frac ? 12 : 11)] += 1;   // LCR_PHASE_PROF: rounds of the list
      if (start >= S) break;
    }
    return sc.sync_or(chg) != 0;
  };

  // snpfrags.rs:628-733: connected components of the PASS het SNPs (edges = allele-consistent SNP pairs of
  // a read); component label = smallest SNP index (see RegionHost::assign_phase_set), by min-label
  // propagation over the reads + pointer jumping until no edge joins two labels
  auto phase_set = [&]() {
    for (int i = sc.tid(); i < S; i += sc.nt()) {
      const bool node = sgt[i] == 0 && svt[i] == 1 && !(sflags[i] & (LCR_F_DENSE | LCR_F_RNA_EDIT)) &&
                        !(sps[i] < (double)in.min_phase_score);
      parent[i] = node ? i : -1;
    }
    sc.sync();
    // The reference adds an edge for every pair (x < y) among the first 64 PASS-het entries of a row whose alleles
    // agree with the haplotypes: hap[x] * hap[y] == (alleles differ ? -1 : 1), i.e. hap[x] * allele[x] == hap[y] *
    // allele[y].  So a row's entries fall into two classes by the sign of hap * allele and every class is a clique:
    // the components (all that is used of the graph) are those of "class members joined to their class", O(n) per
    // row instead of O(n^2).  for_classes visits the members of both classes and returns the number of valid entries.
    auto for_classes = [&](int r, auto fn) -> int {
      int n = 0;
      for (int e = rptr[r]; e < (int)rptr[r + 1] && n < 64; e++) {
        const int x = ecol[e];
        if (parent[x] < 0) continue;
        const int c = shap[x] * ((ev[e] & 32) ? -1 : 1);
        if (c) fn(x, c > 0 ? 0 : 1);
        n++;
      }
      return n;
    };
    for (;;) {
      int any = 0;
      for (int r = sc.tid(); r < nrow; r += sc.nt()) {
        if (!fp[r] || asg[r] == 0) continue;
        int lo[2] = {INT_MAX, INT_MAX}, hi[2] = {-1, -1};
        for_classes(r, [&](int x, int k) { const int l = sc.ld(&parent[x]); lo[k] = min(lo[k], l); hi[k] = max(hi[k], l); });
        if ((hi[0] >= 0 && lo[0] != hi[0]) || (hi[1] >= 0 && lo[1] != hi[1])) {
          any = 1;
          for_classes(r, [&](int x, int k) {
            const int l = sc.ld(&parent[x]);
            if (l != lo[k]) { atomicMin(&parent[x], lo[k]); atomicMin(&parent[l], lo[k]); }
          });
        }
      }
      any = sc.sync_or(any);
      for (int i = sc.tid(); i < S; i += sc.nt())
        if (parent[i] >= 0) { int l = sc.ld(&parent[i]); for (;;) { const int p = sc.ld(&parent[l]); if (p == l) break; l = p; } atomicMin(&parent[i], l); }
      sc.sync();
      if (!any) break;
    }
    for (int i = sc.tid(); i < S; i += sc.nt()) if (parent[i] >= 0) cand[i].phase_set = (uint32_t)(cand[parent[i]].pos + 1);
    for (int r = sc.tid(); r < nrow; r += sc.nt()) {
      uint32_t ps = 0;
      if (fp[r] && asg[r] != 0) {
        // largest component root among the components that own an edge of this read = among its classes of two or more
        int root[2] = {-1, -1}, cnt[2] = {0, 0}, first = -1;
        const int n = for_classes(r, [&](int x, int k) { root[k] = parent[x]; cnt[k]++; });
        int best = max(cnt[0] >= 2 ? root[0] : -1, cnt[1] >= 2 ? root[1] : -1);
        if (n == 1) {               // self loop (snpfrags.rs:659-665)
          for (int e = rptr[r]; e < (int)rptr[r + 1]; e++) if (parent[ecol[e]] >= 0) { first = ecol[e]; break; }
          best = parent[first];
        }
        if (best >= 0) ps = (uint32_t)(cand[best].pos + 1);
      }
      in.phase_set[v.r0 + r] = ps;
      in.d_rec[3 * (size_t)(v.r0 + r) + 2] = ps;
    }
    sc.sync();
  };

  const uint64_t rseed = region_seed(in.seed, in.start0[v.g]);
  // thread.rs:168-201 runs (assign reads, assign SNPs) twice, the two rescue lists, and the pair once more.  The
  // pair reads the SNPs' haplotype / genotype / variant type / FOR_PHASING bit and the rows' fp / tag and is
  // idempotent on them: reads_hap applied to its own output under the same SNP state changes nothing (a flipped
  // row's q / qn swap, bit for bit), snp_hap then recomputes the same values and ORs the same flag bits.  So a
  // pair is a no-op -- and skipped -- when neither the previous snp_hap nor the rescues changed one of those
  // fields; the other flag bits are only ever written.
  reads_hap(); mark();
  bool redo = snp_hap(); mark();
  if (redo) { reads_hap(); redo = snp_hap(); }
  mark();
  const float relaxed = in.min_phase_score - 3.0f;
  const bool c1 = rescue(LCR_F_RNA_EDIT, relaxed, false, rseed);
  const bool c2 = rescue(LCR_F_CAND_SOMATIC, relaxed, true, rseed);
  mark();
  if (redo || c1 || c2) { reads_hap(); snp_hap(); }
  mark();
  phase_set();
  mark();
  // results straight into pinned host memory (device-visible): the host only waits for the kernels, no copies follow
  for (int i = sc.tid(); i < S; i += sc.nt()) {
    cand[i].haplotype = shap[i]; cand[i].genotype = sgt[i]; cand[i].variant_type = svt[i];
    cand[i].flags = sflags[i]; cand[i].phase_score = sps[i];
    in.h_cand[v.c0 + i] = cand[i];   // (phase_set was written by this thread above)
  }
  if (sc.tid() == 0) in.h_obj[v.g] = in.st_obj[v.g];
  for (int r = sc.tid(); r < nrow; r += sc.nt()) {
    in.haplotag[v.r0 + r] = tag[r]; in.assignment[v.r0 + r] = asg[r];
    in.d_rec[3 * (size_t)(v.r0 + r)] = (uint32_t)(v.r0 + r);
    in.d_rec[3 * (size_t)(v.r0 + r) + 1] = (uint32_t)(uint8_t)tag[r] | ((uint32_t)asg[r] << 8);
  }
  mark();
}

}  // namespace
