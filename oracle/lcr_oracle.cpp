/*
 * oracle/lcr_oracle.cpp — TEST INFRASTRUCTURE ONLY (see orc_common.h).
 *
 * Structure-faithful CPU restatement of longcallR v1.12.0's per-region hot path: same AoS
 * BaseFreq-with-Vec-per-allele layout, libm pow/log10 per observation, same loop order and
 * tie-breaks.  std::map replaces HashMap where the reference's iteration order is unspecified;
 * an injected counter-based RNG replaces rand::thread_rng() (orc_common.h).  PARITY UNPINNED BY
 * THE REFERENCE (no reference tests / goldens exist, Rust toolchain absent): pinned by hand-derived
 * KATs and by an independent NumPy restatement.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference/src).
 */
#include "orc_common.h"

#include <algorithm>
#include <array>
#include <atomic>
#include <cassert>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <set>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {

const uint8_t MAX_BASE_QUALITY = 30; /* main.rs:20 */

/* ---------- data model (util.rs:71-127, snp.rs:39-245, snpfrags.rs:14-53) ---------- */
struct BaseFreq {
  uint32_t a = 0, c = 0, g = 0, t = 0, n = 0, d = 0, ni = 0;
  char ref_base = 0;
  std::vector<uint8_t> baseq[4]; /* a,c,g,t in push (read) order */
  int base_strands[4][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
  int transcript_strands[2] = {0, 0};
  uint32_t cnt(int i) const { return i == 0 ? a : i == 1 ? c : i == 2 ? g : t; }
};

struct Cand {
  int64_t pos = 0;
  char alleles[2] = {0, 0};
  float allele_freqs[2] = {0, 0};
  uint32_t allele_cnt[2] = {0, 0};
  char reference = 0;
  uint32_t n_alt = 0;
  uint32_t depth = 0;
  int variant_type = 0;
  double variant_quality = 0;
  double loglik[3] = {0, 0, 0};
  double genotype_probability[3] = {0, 0, 0};
  double genotype_quality = 0;
  int genotype = 0;
  int haplotype = 0;
  double phase_score = 0;
  std::vector<int> cover; /* snp_cover_fragments */
  std::vector<int> cover_pos; /* oracle-only index: cover[j]'s entry of this SNP is frags[cover[j]].list[cover_pos[j]] */
  bool rna_editing = false, dense = false, het_var = false, for_phasing = false, hom_var = false,
       single = false, non_selected = false, cand_somatic = false;
  uint32_t phase_set = 0;
};

struct FragElem {
  int snp_idx;
  int64_t pos;
  char base;
  uint8_t baseq;
  int strand;
  int p;
  double prob;
  bool phase_site;
};

struct Fragment {
  int fragment_idx = 0;
  int read = 0;
  std::vector<FragElem> list;
  int haplotag = 0;
  int assignment = 0;
  double assignment_score = 0;
  uint32_t num_hete_links = 0;
  bool for_phasing = false;
};

struct LDPair {
  std::map<std::pair<uint8_t, uint8_t>, uint32_t> ld_pairs;
  bool valid = false;
  float score = 0;
  int weight = 0;
};

/* petgraph 0.6.4 GraphMap<usize,_,Undirected> emulation: node order = insertion order,
 * adjacency in edge-insertion order (graphmap.rs add_edge). */
struct GraphMap {
  std::vector<int> nodes;            /* insertion order */
  std::map<int, std::vector<int>> adj;
  std::map<std::pair<int, int>, int> w;
  static std::pair<int, int> key(int a, int b) { return a <= b ? std::make_pair(a, b) : std::make_pair(b, a); }
  bool contains_node(int a) const { return adj.count(a) != 0; }
  void add_node(int a) {
    if (!adj.count(a)) { adj[a] = {}; nodes.push_back(a); }
  }
  bool contains_edge(int a, int b) const { return w.count(key(a, b)) != 0; }
  void add_edge(int a, int b, int weight) {
    auto k = key(a, b);
    if (w.count(k)) { w[k] = weight; return; }
    w[k] = weight;
    add_node(a);
    adj[a].push_back(b);
    if (a != b) { add_node(b); adj[b].push_back(a); }
  }
  /* petgraph::algo::kosaraju_scc on an undirected GraphMap (algo/mod.rs): first pass DfsPostOrder
   * over nodes in insertion order, second pass Dfs (LIFO stack) in reverse finish order. */
  std::vector<std::vector<int>> kosaraju_scc() const {
    std::set<int> discovered, finished;
    std::vector<int> finish_order, stack;
    for (int i : nodes) {
      if (discovered.count(i)) continue;
      stack.clear();
      stack.push_back(i);
      while (!stack.empty()) {
        int nx = stack.back();
        if (discovered.insert(nx).second) {
          for (int succ : adj.at(nx))
            if (!discovered.count(succ)) stack.push_back(succ);
        } else {
          stack.pop_back();
          if (finished.insert(nx).second) finish_order.push_back(nx);
        }
      }
    }
    std::vector<std::vector<int>> sccs;
    discovered.clear();
    for (auto it = finish_order.rbegin(); it != finish_order.rend(); ++it) {
      int i = *it;
      if (discovered.count(i)) continue;
      stack.clear();
      stack.push_back(i);
      std::vector<int> scc;
      while (!stack.empty()) {
        int node = stack.back();
        stack.pop_back();
        if (discovered.insert(node).second) {
          for (int succ : adj.at(node))
            if (!discovered.count(succ)) stack.push_back(succ);
          scc.push_back(node);
        }
      }
      sccs.push_back(scc);
    }
    return sccs;
  }
};

inline int base_index(char b) {
  switch (b) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
  }
}

/* Rust `x as i32` for f64: saturating, NaN -> 0 */
inline int32_t as_i32(double x) {
  if (std::isnan(x)) return 0;
  if (x >= 2147483647.0) return 2147483647;
  if (x <= -2147483648.0) return (-2147483647 - 1);
  return (int32_t)x;
}

/* ---------- probability kernels (phase.rs:32-49,77-96,128-176,238-255) ---------- */
inline double aki(int sigma, int delta, int eta, int base_allele, double error_rate) {
  int x = (eta == 0) ? sigma * delta : eta;
  return (base_allele == x) ? 1.0 - error_rate : error_rate;
}

double cal_sigma_delta_eta_log(int sigma_k, const std::vector<int>& delta, const std::vector<int>& eta,
                               const std::vector<int>& ps, const std::vector<double>& probs) {
  double log_q1 = 0.0, log_q2 = 0.0, log_q3 = 0.0;
  for (size_t i = 0; i < delta.size(); i++) log_q1 += std::log10(aki(sigma_k, delta[i], eta[i], ps[i], probs[i]));
  for (size_t i = 0; i < delta.size(); i++) {
    log_q2 += std::log10(aki(1, delta[i], eta[i], ps[i], probs[i]));
    log_q3 += std::log10(aki(-1, delta[i], eta[i], ps[i], probs[i]));
  }
  return 1.0 - log_q1 / (log_q2 + log_q3);
}

const double THETA = 0.001;
inline double prior_homref_log() { return std::log10(1.0 - 1.5 * THETA); }
inline double prior_homvar_log() { return std::log10(0.5 * THETA); }
inline double prior_hetvar_log(size_t cov) {
  if (cov == 0) return std::log10(0.001);
  return std::log10(0.001) - (double)(uint32_t)cov * std::log10(2.0);
}

double cal_delta_eta_sigma_log(int delta_i, int eta_i, const std::vector<int>& sigma, const std::vector<int>& ps,
                               const std::vector<double>& probs) {
  double log_q1 = 0.0, log_q2 = 0.0, log_q3 = 0.0, log_q4 = 0.0, log_q5 = 0.0;
  double p_homref = prior_homref_log(), p_homvar = prior_homvar_log(), p_het = prior_hetvar_log(sigma.size());
  for (size_t k = 0; k < sigma.size(); k++) log_q1 += std::log10(aki(sigma[k], delta_i, eta_i, ps[k], probs[k]));
  if (eta_i == 0) log_q1 += p_het;
  else if (eta_i == 1) log_q1 += p_homref;
  else log_q1 += p_homvar;
  for (size_t k = 0; k < sigma.size(); k++) {
    log_q2 += std::log10(aki(sigma[k], delta_i, -1, ps[k], probs[k]));
    log_q3 += std::log10(aki(sigma[k], delta_i, 0, ps[k], probs[k]));
    log_q4 += std::log10(aki(sigma[k], delta_i, 1, ps[k], probs[k]));
    log_q5 += std::log10(aki(sigma[k], delta_i * (-1), 0, ps[k], probs[k]));
  }
  log_q2 += p_homvar;
  log_q3 += p_het;
  log_q4 += p_homref;
  log_q5 += p_het;
  return 1.0 - log_q1 / (log_q2 + log_q3 + log_q4 + log_q5);
}

/* phase.rs:178-236 */
double cal_block_delta_eta_sigma_log(const std::vector<int>& bdelta, const std::vector<int>& beta,
                                     const std::vector<std::vector<int>>& bsigma,
                                     const std::vector<std::vector<int>>& bps,
                                     const std::vector<std::vector<double>>& bprobs) {
  double sum_prob = 0.0;
  for (size_t i = 0; i < bdelta.size(); i++) sum_prob += cal_delta_eta_sigma_log(bdelta[i], beta[i], bsigma[i], bps[i], bprobs[i]);
  return sum_prob;
}

double cal_phase_score_log(int delta_i, int eta_i, const std::vector<int>& sigma, const std::vector<int>& ps,
                           const std::vector<double>& probs) {
  double log_q1 = 0.0, log_q2 = 0.0, log_q3 = 0.0;
  for (size_t k = 0; k < sigma.size(); k++) log_q1 += std::log10(aki(sigma[k], delta_i, eta_i, ps[k], probs[k]));
  for (size_t k = 0; k < sigma.size(); k++) {
    log_q2 += std::log10(aki(sigma[k], 1, eta_i, ps[k], probs[k]));
    log_q3 += std::log10(aki(sigma[k], -1, eta_i, ps[k], probs[k]));
  }
  return 1.0 - log_q1 / (log_q2 + log_q3);
}

/* candidate.rs:24-35 (f32 arithmetic) */
float cal_strand_odds_ratio(int ref_fw, int ref_rv, int alt_fw, int alt_rv) {
  float x00 = (float)(ref_fw + 1), x01 = (float)(ref_rv + 1), x10 = (float)(alt_fw + 1), x11 = (float)(alt_rv + 1);
  float symmetrical_ratio = (x00 * x11) / (x01 * x10) + (x01 * x10) / (x00 * x11);
  float ref_ratio = std::fmin(x00, x01) / std::fmax(x00, x01);
  float alt_ratio = std::fmin(x10, x11) / std::fmax(x10, x11);
  return std::log(symmetrical_ratio) + std::log(ref_ratio) - std::log(alt_ratio);
}

/* candidate.rs:37-47; statrs 0.16 Binomial(p=0.5, n).cdf is restated as the exact sum
 * sum_{i<=k} C(n,i)/2^n (exact in f64 for the n<=30 the caller allows). */
double binom_cdf_half(uint64_t n, uint64_t k) {
  if (k >= n) return 1.0;
  double c = 1.0, s = 0.0;
  for (uint64_t i = 0; i <= k; i++) {
    s += c;
    c = c * (double)(n - i) / (double)(i + 1);
  }
  return s / std::pow(2.0, (double)n);
}
double binomial_two_tailed(uint64_t successes, uint64_t trials) {
  if (successes == 0) return 2.0 * binom_cdf_half(trials, 0);
  if (successes == trials) return 2.0 * (1.0 - binom_cdf_half(trials, trials - 1));
  return 2.0 * std::fmin(binom_cdf_half(trials, successes), 1.0 - binom_cdf_half(trials, successes - 1));
}

/* util.rs:162-176; Rust sort_by is stable */
void get_two_major_alleles(const uint32_t cnt[4], char ref_base, char* a1, uint32_t* c1, char* a2, uint32_t* c2) {
  std::pair<char, uint32_t> x[4] = {{'A', cnt[0]}, {'C', cnt[1]}, {'G', cnt[2]}, {'T', cnt[3]}};
  std::stable_sort(x, x + 4, [](const std::pair<char, uint32_t>& p, const std::pair<char, uint32_t>& q) { return p.second > q.second; });
  int second = 1;
  if (x[0].first != ref_base && x[1].first != ref_base) {
    if (x[2].second == x[1].second && x[2].first == ref_base) second = 2;
    else if (x[3].second == x[1].second && x[3].first == ref_base) second = 3;
  }
  *a1 = x[0].first; *c1 = x[0].second; *a2 = x[second].first; *c2 = x[second].second;
}

/* fixed-point LUT of the phasing emission terms (ORC_MODE_EXACT), scale 2^40.
 * eps(q) = 10^(-q/10) (fragment.rs:132); q = 0 (eps = 1, where the reference NaN-panics at
 * phase.rs:307) is treated as q = 1 — documented deviation from a crash. */
const double FX_SCALE = 1099511627776.0; /* 2^40 */
struct PhaseLut {
  double le[31], l1e[31];
  int64_t fe[31], f1e[31];
  int64_t f_homref, f_homvar, f_het0, f_log2;
  PhaseLut() {
    for (int q = 0; q <= 30; q++) {
      int qq = q == 0 ? 1 : q;
      double eps = std::pow(10.0, -(double)qq / 10.0);
      le[q] = std::log10(eps);
      l1e[q] = std::log10(1.0 - eps);
      fe[q] = std::llround(le[q] * FX_SCALE);
      f1e[q] = std::llround(l1e[q] * FX_SCALE);
    }
    f_homref = std::llround(prior_homref_log() * FX_SCALE);
    f_homvar = std::llround(prior_homvar_log() * FX_SCALE);
    f_het0 = std::llround(std::log10(0.001) * FX_SCALE);
    f_log2 = std::llround(std::log10(2.0) * FX_SCALE);
  }
};
const PhaseLut& plut() { static PhaseLut l; return l; }
inline double phase_prob(uint8_t q) { return std::pow(10.0, -(double)(q == 0 ? 1 : q) / 10.0); }
inline int64_t fx_aki(int sigma, int delta, int eta, int p, uint8_t q) {
  int x = (eta == 0) ? sigma * delta : eta;
  return (p == x) ? plut().f1e[q] : plut().fe[q];
}

/* log10(aki(...)) from the table: aki returns eps or 1 - eps with eps = phase_prob(q), so its log10 is one of 62 libm
 * values -- the SAME bits the reference-order code gets from its libm call per observation (indexed form, orc_set_fast). */
inline double lut_laki(int sigma, int delta, int eta, int p, uint8_t q) {
  const int x = (eta == 0) ? sigma * delta : eta;
  return (p == x) ? plut().l1e[q] : plut().le[q];
}

/* a small persistent pool for the indexed form's Jacobi steps: run(n, fn) calls fn(i) for i in [0, n) on all threads */
class ParPool {
 public:
  explicit ParPool(int n_threads) : nt_(std::max(1, n_threads)) {
    for (int t = 1; t < nt_; t++) th_.emplace_back([this]() { loop(); });
  }
  ~ParPool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto& t : th_) t.join();
  }
  int threads() const { return nt_; }
  void run(int64_t n, int64_t chunk, const std::function<void(int64_t)>& fn) {
    if (nt_ == 1 || n <= chunk) { for (int64_t i = 0; i < n; i++) fn(i); return; }
    { std::lock_guard<std::mutex> l(m_); fn_ = &fn; n_ = n; chunk_ = chunk; next_.store(0); left_ = nt_ - 1; gen_++; }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [this]() { return left_ == 0; });
  }
 private:
  void work() {
    for (;;) {
      const int64_t b = next_.fetch_add(chunk_);
      if (b >= n_) return;
      const int64_t e = std::min(n_, b + chunk_);
      for (int64_t i = b; i < e; i++) (*fn_)(i);
    }
  }
  void loop() {
    uint64_t seen = 0;
    for (;;) {
      { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&]() { return gen_ != seen; }); seen = gen_; if (stop_) return; }
      work();
      { std::lock_guard<std::mutex> l(m_); if (--left_ == 0) done_.notify_one(); }
    }
  }
  int nt_;
  std::vector<std::thread> th_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int64_t)>* fn_ = nullptr;
  int64_t n_ = 0, chunk_ = 1;
  std::atomic<int64_t> next_{0};
  int left_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

}  // namespace

/* ================================================================================== */
struct orc_region {
  const lcr_reads* reads;
  int32_t rb, re;
  int64_t start0;
  int32_t len;
  const uint8_t* ref;
  lcr_params prm;

  std::vector<BaseFreq> freq;
  std::vector<Cand> cands;
  std::vector<int> homo_snps, edit_snps, het_snps, somatic_snps;
  std::vector<Fragment> frags;
  std::map<std::pair<int, int>, LDPair> allele_pairs;
  std::vector<std::vector<int>> ld_blocks;
  uint64_t seed, ctr = 0;
  double best_objective = 0;
  int64_t stats[4] = {0, 0, 0, 0};
  std::map<int, uint32_t> read_phase_set; /* fragment idx -> PS */
  std::vector<uint8_t>* round_log = nullptr; /* orc_round_log: 1 per half-round of phase.rs:1198-1233 that raised largest_prob */
  int fast_threads = 0;   /* orc_set_fast: 0 = the reference's linear searches, >= 1 = indexed gathers on that many threads */
  int tie_mask = 15;      /* ORC_MODE_TIE: tie classes resolved by the f64 scores */
  int64_t census[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  ParPool* pool = nullptr;
  ~orc_region() { delete pool; delete round_log; }

  double rnd() { return orc_u01(seed, ctr++); }

  /* ---------- P1: Profile::fill_data_into_freq_vec (util.rs:621-949) ---------- */
  void pileup() {
    const int vec_size = len;
    freq.assign(vec_size, BaseFreq());
    for (int i = 0; i < vec_size; i++) freq[i].ref_base = (char)ref[i]; /* util.rs:646-648 */
    const int64_t polya_tail_length = prm.polya_len;
    const bool ont = prm.platform == LCR_PLATFORM_ONT;
    for (int r = rb; r < re; r++) {
      const uint8_t* seq = reads->bases + reads->seq_off[r];
      const uint8_t* base_qual = reads->quals + reads->seq_off[r];
      const int64_t seq_len = reads->seq_len[r];
      const int strand = (reads->flags[r] & 1) ? 1 : 0;
      const int ts = (reads->flags[r] >> 1) & 3; /* 0 none, 1 '+', 2 '-' */
      const int64_t leading_softclips = reads->lead_clip[r];
      const int64_t trailing_softclips = reads->trail_clip[r];
      int32_t pos_in_freq_vec = (int32_t)((int64_t)reads->pos[r] - start0);
      int64_t pos_in_read = leading_softclips > 0 ? leading_softclips : 0;
      const uint32_t* cig = reads->cigar + reads->cig_off[r];
      const uint32_t ncig = reads->n_cig[r];
      bool stop = false;
      for (uint32_t cg_idx = 0; cg_idx < ncig && !stop; cg_idx++) {
        const uint32_t cg_len = cig[cg_idx] >> 4;
        const uint32_t op = cig[cg_idx] & 15;
        switch (op) {
          case 4: case 5: /* S, H (util.rs:695-697) */
            break;
          case 0: case 7: case 8: { /* M, =, X (util.rs:698-904) */
            for (uint32_t cgi = 0; cgi < cg_len; cgi++) {
              if (pos_in_freq_vec < 0) { pos_in_freq_vec++; pos_in_read++; continue; }
              if (pos_in_freq_vec >= vec_size) break;
              const char base = (char)seq[pos_in_read];
              const uint8_t baseq = base_qual[pos_in_read] < MAX_BASE_QUALITY ? base_qual[pos_in_read] : MAX_BASE_QUALITY;
              BaseFreq& bf = freq[pos_in_freq_vec];
              const char ref_base = bf.ref_base;
              bool poly_a_flag = false, homopolymer_flag = false, trim_flag = false;
              const int64_t dist = prm.dist_to_end;
              const int64_t curr_pos = pos_in_read;
              const int64_t read_end_boundary = seq_len - trailing_softclips;
              if (ont) { /* util.rs:745-751 */
                if (std::llabs(curr_pos - leading_softclips) < dist || std::llabs(curr_pos - read_end_boundary) < dist) trim_flag = true;
              }
              if (!trim_flag) { /* util.rs:754-789 */
                if (std::llabs(curr_pos - leading_softclips) < dist || std::llabs(curr_pos - read_end_boundary) < dist) {
                  for (int64_t tmpi = curr_pos - polya_tail_length; tmpi <= curr_pos + 1; tmpi++) {
                    if (tmpi < 0 || tmpi + polya_tail_length - 1 >= seq_len) continue;
                    int64_t poly_counts[4] = {0, 0, 0, 0}; /* A, T, C, G */
                    for (int64_t tmpj = 0; tmpj < polya_tail_length; tmpj++) {
                      const uint8_t b = seq[tmpi + tmpj];
                      if (b == 'A' && ref_base != 'A') poly_counts[0]++;
                      else if (b == 'T' && ref_base != 'T') poly_counts[1]++;
                      else if (b == 'C' && ref_base != 'C') poly_counts[2]++;
                      else if (b == 'G' && ref_base != 'G') poly_counts[3]++;
                    }
                    if (poly_counts[0] >= polya_tail_length || poly_counts[1] >= polya_tail_length) poly_a_flag = true;
                    if (poly_counts[2] >= polya_tail_length || poly_counts[3] >= polya_tail_length) homopolymer_flag = true;
                  }
                }
              }
              if (!trim_flag && !poly_a_flag && !homopolymer_flag) {
                if (strand == 0) { /* util.rs:803-819 */
                  if (ts == 1) bf.transcript_strands[0]++;
                  else if (ts == 2) bf.transcript_strands[1]++;
                } else {
                  if (ts == 1) bf.transcript_strands[1]++;
                  else if (ts == 2) bf.transcript_strands[0]++;
                }
                const int bi = base_index(base);
                if (bi >= 0) { /* util.rs:821-889 */
                  if (bi == 0) bf.a++; else if (bi == 1) bf.c++; else if (bi == 2) bf.g++; else bf.t++;
                  bf.baseq[bi].push_back(baseq);
                  bf.base_strands[bi][strand]++;
                } /* else: "Invalid nucleotide base" (util.rs:890-892), nothing tallied */
              }
              pos_in_freq_vec++;
              pos_in_read++;
            }
            break;
          }
          case 2: { /* D (util.rs:905-917) */
            for (uint32_t k = 0; k < cg_len; k++) {
              if (pos_in_freq_vec < 0) { pos_in_freq_vec++; continue; }
              if (pos_in_freq_vec >= vec_size) break;
              freq[pos_in_freq_vec].d++;
              pos_in_freq_vec++;
            }
            break;
          }
          case 1: { /* I (util.rs:918-929) */
            if (pos_in_freq_vec < 1) { pos_in_read += cg_len; break; }
            if (pos_in_freq_vec >= vec_size) { stop = true; break; }
            freq[pos_in_freq_vec - 1].ni++;
            pos_in_read += cg_len;
            break;
          }
          case 3: { /* N (util.rs:930-942) */
            for (uint32_t k = 0; k < cg_len; k++) {
              if (pos_in_freq_vec < 0) { pos_in_freq_vec++; continue; }
              if (pos_in_freq_vec >= vec_size) break;
              freq[pos_in_freq_vec].n++;
              pos_in_freq_vec++;
            }
            break;
          }
          default: /* util.rs:943-945 panics; lcr_load_batch rejects such input */
            stop = true;
            break;
        }
      }
    }
  }

  /* genotype-likelihood block, candidate.rs:236-335, given the per-allele quality lists */
  static bool gt_block(const BaseFreq& bf, double loglik[3], double gprob[3], double* vq, double* gq) {
    const double theta = 0.001;
    const double background_prob[3] = {theta / 2.0, theta, 1.0 - 1.5 * theta};
    int ri = -1;
    if (bf.ref_base == 'A') ri = 0; else if (bf.ref_base == 'C') ri = 1; else if (bf.ref_base == 'G') ri = 2; else if (bf.ref_base == 'T') ri = 3;
    else return false; /* 'N' or unknown ref base: continue (candidate.rs:254-265) */
    loglik[0] = loglik[1] = loglik[2] = 0.0;
    for (uint8_t bq : bf.baseq[ri]) {
      double error_rate = std::pow(0.1, (double)bq / 10.0);
      loglik[0] += std::log10(error_rate);
      loglik[2] += std::log10(1.0 - error_rate);
    }
    for (int o = 0; o < 4; o++) {
      if (o == ri) continue;
      for (uint8_t bq : bf.baseq[o]) {
        double error_rate = std::pow(0.1, (double)bq / 10.0);
        loglik[0] += std::log10(1.0 - error_rate);
        loglik[2] += std::log10(error_rate);
      }
    }
    uint32_t num_reads = bf.a + bf.c + bf.g + bf.t;
    loglik[1] -= (double)num_reads * std::log10(2.0);
    gt_tail(loglik, gprob, vq, gq);
    (void)background_prob;
    return true;
  }

  /* candidate.rs:287-335: posterior, QUAL, GQ from loglik[3] */
  static void gt_tail(const double loglik[3], double gprob[3], double* vq, double* gq) {
    const double theta = 0.001;
    const double background_prob[3] = {theta / 2.0, theta, 1.0 - 1.5 * theta};
    double logprob[3] = {loglik[0], loglik[1], loglik[2]};
    logprob[0] += std::log10(background_prob[0]);
    logprob[1] += std::log10(background_prob[1]);
    logprob[2] += std::log10(background_prob[2]);
    double max_logprob = std::fmax(std::fmax(logprob[0], logprob[1]), logprob[2]);
    logprob[0] -= max_logprob; logprob[1] -= max_logprob; logprob[2] -= max_logprob;
    double variant_prob[3] = {std::pow(10.0, logprob[0]), std::pow(10.0, logprob[1]), std::pow(10.0, logprob[2])};
    double sum_variant_prob = variant_prob[0] + variant_prob[1] + variant_prob[2];
    variant_prob[0] /= sum_variant_prob; variant_prob[1] /= sum_variant_prob; variant_prob[2] /= sum_variant_prob;
    *vq = -10.0 * std::log10(std::fmax(10e-301, variant_prob[2]));
    double l[3] = {loglik[0], loglik[1], loglik[2]};
    double mx = std::fmax(std::fmax(l[0], l[1]), l[2]);
    l[0] = std::pow(10.0, l[0] - mx); l[1] = std::pow(10.0, l[1] - mx); l[2] = std::pow(10.0, l[2] - mx);
    double s = l[0] + l[1] + l[2];
    gprob[0] = l[0] / s; gprob[1] = l[1] / s; gprob[2] = l[2] / s;
    double ph[3] = {-10.0 * std::log10(gprob[0]), -10.0 * std::log10(gprob[1]), -10.0 * std::log10(gprob[2])};
    /* Rust sort_by(cmp_f64) on 3 elements = insertion sort with is_less = (a < b) */
    for (int i = 1; i < 3; i++) {
      double v = ph[i];
      int j = i;
      while (j > 0 && v < ph[j - 1]) { ph[j] = ph[j - 1]; j--; }
      ph[j] = v;
    }
    *gq = ph[1] - ph[0];
  }

  /* ---------- P2-P5: SNPFrag::get_candidate_snps (candidate.rs:54-528) ---------- */
  void candidates() {
    cands.clear(); homo_snps.clear(); edit_snps.clear(); het_snps.clear(); somatic_snps.clear();
    const float SOR_THRESHOLD = cal_strand_odds_ratio(5, 5, 9, 1); /* candidate.rs:49-51 */
    int64_t position = start0;
    for (size_t bfidx = 0; bfidx < freq.size(); bfidx++, position++) {
      const BaseFreq& bf = freq[bfidx];
      const uint32_t total_allele_count = bf.a + bf.c + bf.g + bf.t;
      if (total_allele_count < prm.min_depth || total_allele_count > prm.max_depth) continue;
      char allele1, allele2; uint32_t allele1_cnt, allele2_cnt;
      uint32_t cnt4[4] = {bf.a, bf.c, bf.g, bf.t};
      get_two_major_alleles(cnt4, bf.ref_base, &allele1, &allele1_cnt, &allele2, &allele2_cnt);
      const float allele1_freq = (float)allele1_cnt / (float)total_allele_count;
      const float allele2_freq = (float)allele2_cnt / (float)total_allele_count;
      char ref_allele_base; uint32_t alt_num; char alt_base[2] = {0, 0}; float alt_freq[2] = {0, 0}; uint32_t alt_cnt[2] = {0, 0};
      if (allele1 == bf.ref_base) {
        ref_allele_base = allele1; alt_num = 1; alt_base[0] = allele2; alt_freq[0] = allele2_freq; alt_cnt[0] = allele2_cnt;
      } else if (allele2 == bf.ref_base) {
        ref_allele_base = allele2; alt_num = 1; alt_base[0] = allele1; alt_freq[0] = allele1_freq; alt_cnt[0] = allele1_cnt;
      } else {
        ref_allele_base = bf.ref_base; alt_num = 2;
        alt_base[0] = allele1; alt_freq[0] = allele1_freq; alt_cnt[0] = allele1_cnt;
        alt_base[1] = allele2; alt_freq[1] = allele2_freq; alt_cnt[1] = allele2_cnt;
      }
      if (base_index(ref_allele_base) < 0) continue; /* VALID_ALLELES, candidate.rs:132 */
      if (alt_num == 1) { /* candidate.rs:142-155 */
        if (total_allele_count < 200 && alt_freq[0] < prm.low_frac_cut) continue;
        if (total_allele_count >= 200 && alt_cnt[0] < prm.low_cnt_cut) continue;
      }
      if (bf.d >= alt_cnt[0]) continue; /* candidate.rs:165 */
      if ((float)(allele1_cnt + allele2_cnt) / (float)(bf.a + bf.c + bf.g + bf.t + bf.d + bf.n) < prm.min_af_intron) continue;
      /* base-quality filter, candidate.rs:174-194 */
      if (allele1 != bf.ref_base) {
        const auto& quals = bf.baseq[base_index(allele1)];
        size_t pass = 0; for (uint8_t bq : quals) if (bq >= prm.min_baseq) pass++;
        if (allele1_cnt > 0 && pass < 2) continue;
      } else if (allele2 != bf.ref_base) {
        const auto& quals = bf.baseq[base_index(allele2)];
        size_t pass = 0; for (uint8_t bq : quals) if (bq >= prm.min_baseq) pass++;
        if (allele2_cnt > 0 && pass < 2) continue;
      }
      if (prm.use_strand_bias) { /* candidate.rs:199-234 */
        const int* rs = bf.base_strands[base_index(ref_allele_base)];
        float sor;
        if (alt_num == 1) {
          const int* as = bf.base_strands[base_index(alt_base[0])];
          sor = cal_strand_odds_ratio(rs[0], rs[1], as[0], as[1]);
        } else {
          const int* a1s = bf.base_strands[base_index(alt_base[0])];
          const int* a2s = bf.base_strands[base_index(alt_base[1])];
          float sor1 = cal_strand_odds_ratio(rs[0], rs[1], a1s[0], a1s[1]);
          float sor2 = cal_strand_odds_ratio(rs[0], rs[1], a2s[0], a2s[1]);
          sor = std::fmax(sor1, sor2);
        }
        if (sor > SOR_THRESHOLD) continue;
        if (alt_num == 1) {
          const int* as = bf.base_strands[base_index(alt_base[0])];
          if (as[0] + as[1] <= 30) {
            double p = binomial_two_tailed((uint64_t)as[0], (uint64_t)(as[0] + as[1]));
            if (p < 0.05) continue;
          }
          if (as[0] * as[1] == 0) continue;
        }
      }
      double loglik[3], gprob[3], variant_quality, genotype_quality;
      if (!gt_block(bf, loglik, gprob, &variant_quality, &genotype_quality)) continue;
      Cand cs;
      cs.pos = position;
      cs.alleles[0] = allele1; cs.alleles[1] = allele2;
      cs.allele_cnt[0] = allele1_cnt; cs.allele_cnt[1] = allele2_cnt;
      cs.allele_freqs[0] = allele1_freq; cs.allele_freqs[1] = allele2_freq;
      cs.reference = bf.ref_base;
      cs.n_alt = alt_num;
      cs.depth = total_allele_count;
      cs.variant_quality = variant_quality;
      for (int i = 0; i < 3; i++) { cs.genotype_probability[i] = gprob[i]; cs.loglik[i] = loglik[i]; }
      cs.genotype_quality = genotype_quality;
      if (gprob[0] > gprob[1] && gprob[0] > gprob[2]) { cs.variant_type = 2; cs.genotype = -1; }
      else if (gprob[1] > gprob[0] && gprob[1] > gprob[2]) { cs.variant_type = 1; cs.genotype = 0; }
      else { cs.variant_type = 0; cs.genotype = 1; }
      if (variant_quality < (double)prm.min_qual) continue; /* candidate.rs:374 */
      const int fwd_t = bf.transcript_strands[0], rev_t = bf.transcript_strands[1];
      if (ref_allele_base == 'A' && alt_base[0] == 'G' && (fwd_t > rev_t * 2 || (fwd_t == 0 && rev_t == 0)) && cs.variant_type != 2) {
        cs.rna_editing = true; cs.for_phasing = false;
        cands.push_back(cs); edit_snps.push_back((int)cands.size() - 1); continue;
      }
      if (ref_allele_base == 'T' && alt_base[0] == 'C' && (rev_t > fwd_t * 2 || (fwd_t == 0 && rev_t == 0)) && cs.variant_type != 2) {
        cs.rna_editing = true; cs.for_phasing = false;
        cands.push_back(cs); edit_snps.push_back((int)cands.size() - 1); continue;
      }
      if (alt_num == 1 && alt_freq[0] < prm.min_af) { /* candidate.rs:410-417 */
        cs.cand_somatic = true; cs.for_phasing = false;
        cands.push_back(cs); somatic_snps.push_back((int)cands.size() - 1); continue;
      }
      if (cs.variant_type == 2) {
        if (alt_num == 2 && alt_freq[0] >= prm.min_af && alt_freq[1] >= prm.min_af) { cs.variant_type = 3; cs.genotype = -1; }
        cs.hom_var = true; cs.for_phasing = true;
        cands.push_back(cs); homo_snps.push_back((int)cands.size() - 1); continue;
      }
      if (cs.variant_type == 1) {
        if (alt_num == 2) {
          cs.variant_type = 3; cs.genotype = -1; cs.hom_var = true; cs.for_phasing = true;
          cands.push_back(cs); homo_snps.push_back((int)cands.size() - 1); continue;
        }
        cs.het_var = true; cs.for_phasing = true;
        cands.push_back(cs); het_snps.push_back((int)cands.size() - 1); continue;
      }
      /* variant_type == 0: dropped (candidate.rs:457-460) */
    }
    /* dense filters, candidate.rs:465-526 */
    std::vector<int> concat(homo_snps);
    concat.insert(concat.end(), het_snps.begin(), het_snps.end());
    std::sort(concat.begin(), concat.end());
    const size_t n = concat.size();
    for (size_t i = 0; i < n; i++) {
      int64_t start_pos = cands[concat[i]].pos;
      for (size_t j = i; j < n; j++) {
        int64_t diff = cands[concat[j]].pos - start_pos;
        if (diff > (int64_t)prm.dense_win) {
          if ((uint32_t)(j - i) >= prm.min_dense_cnt)
            for (size_t tk = i; tk < j; tk++) { cands[concat[tk]].dense = true; cands[concat[tk]].for_phasing = false; }
          break;
        }
        if (j == n - 1 && (uint32_t)(j - i + 1) >= prm.min_dense_cnt)
          for (size_t tk = i; tk < j; tk++) { cands[concat[tk]].dense = true; cands[concat[tk]].for_phasing = false; }
      }
    }
    for (size_t i = 0; i < n; i++) {
      int64_t start_pos = cands[concat[i]].pos;
      for (size_t j = i; j < n; j++) {
        int64_t diff = cands[concat[j]].pos - start_pos;
        if (diff >= 5) {
          if ((uint32_t)(j - i) >= 3)
            for (size_t tk = i; tk < j; tk++) { cands[concat[tk]].dense = true; cands[concat[tk]].for_phasing = false; }
          break;
        }
        if (j == n - 1 && (uint32_t)(j - i + 1) >= 3)
          for (size_t tk = i; tk < j; tk++) { cands[concat[tk]].dense = true; cands[concat[tk]].for_phasing = false; }
      }
    }
    auto is_dense = [&](int i) { return cands[i].dense; };
    homo_snps.erase(std::remove_if(homo_snps.begin(), homo_snps.end(), is_dense), homo_snps.end());
    het_snps.erase(std::remove_if(het_snps.begin(), het_snps.end(), is_dense), het_snps.end());
  }

  /* ---------- P6: SNPFrag::get_fragments (fragment.rs:10-309) ---------- */
  void fragments() {
    frags.clear(); allele_pairs.clear();
    for (auto& c : cands) { c.cover.clear(); c.cover_pos.clear(); }
    if (cands.empty()) return;
    const int ncand = (int)cands.size();
    for (int r = rb; r < re; r++) {
      const int64_t pos = reads->pos[r];
      if (pos > cands.back().pos) continue; /* fragment.rs:51-54 */
      const uint8_t* seq = reads->bases + reads->seq_off[r];
      const uint8_t* qual = reads->quals + reads->seq_off[r];
      const int strand = (reads->flags[r] & 1) ? 1 : 0;
      int64_t pos_on_ref = pos;
      int64_t pos_on_query = reads->lead_clip[r];
      int idx = 0;
      int64_t snp_pos = -1;
      char alleles[2];
      if (pos <= cands.front().pos) {
        snp_pos = cands[idx].pos; alleles[0] = cands[idx].alleles[0]; alleles[1] = cands[idx].alleles[1];
      } else {
        while (idx < ncand) { if (cands[idx].pos >= pos) break; idx++; }
        assert(idx < ncand);
        snp_pos = cands[idx].pos; alleles[0] = cands[idx].alleles[0]; alleles[1] = cands[idx].alleles[1];
      }
      Fragment fragment;
      fragment.read = r;
      fragment.fragment_idx = (int)frags.size();
      const uint32_t* cig = reads->cigar + reads->cig_off[r];
      auto advance = [&]() {
        idx++;
        if (idx < ncand) { snp_pos = cands[idx].pos; alleles[0] = cands[idx].alleles[0]; alleles[1] = cands[idx].alleles[1]; }
      };
      for (uint32_t ci = 0; ci < reads->n_cig[r]; ci++) {
        const uint32_t cg_len = cig[ci] >> 4, op = cig[ci] & 15;
        if (op == 4 || op == 5) continue;
        if (op == 0 || op == 7 || op == 8) {
          for (uint32_t k = 0; k < cg_len; k++) {
            if (pos_on_ref == snp_pos) {
              FragElem fe;
              fe.snp_idx = idx;
              fe.pos = pos_on_ref;
              fe.base = (char)seq[pos_on_query];
              fe.baseq = qual[pos_on_query] < 30 ? qual[pos_on_query] : 30;
              fe.strand = strand;
              fe.prob = phase_prob(fe.baseq); /* fragment.rs:132 (q=0 -> q=1, see PhaseLut) */
              if (fe.base == cands[idx].reference) fe.p = 1;
              else if ((fe.base == alleles[0] || fe.base == alleles[1]) && fe.base != cands[idx].reference) fe.p = -1;
              else fe.p = 0;
              fe.phase_site = cands[idx].for_phasing;
              if (!cands[idx].dense && fe.p != 0) fragment.list.push_back(fe);
              advance();
            }
            pos_on_query++;
            pos_on_ref++;
          }
        } else if (op == 1) {
          pos_on_query += cg_len;
        } else if (op == 2 || op == 3) {
          for (uint32_t k = 0; k < cg_len; k++) {
            if (pos_on_ref == snp_pos) advance();
            pos_on_ref++;
          }
        }
      }
      /* pairwise allele co-occurrence, fragment.rs:208-240 */
      for (size_t i = 0; i < fragment.list.size(); i++)
        for (size_t j = i + 1; j < fragment.list.size(); j++) {
          const FragElem &ei = fragment.list[i], &ej = fragment.list[j];
          int s_idx, e_idx; uint8_t s_b, e_b;
          if (ei.snp_idx < ej.snp_idx) { s_idx = ei.snp_idx; e_idx = ej.snp_idx; s_b = ei.base; e_b = ej.base; }
          else { s_idx = ej.snp_idx; e_idx = ei.snp_idx; s_b = ej.base; e_b = ei.base; }
          allele_pairs[{s_idx, e_idx}].ld_pairs[{s_b, e_b}] += 1;
        }
      uint32_t hete_links = 0;
      for (auto& fe : fragment.list) if (fe.phase_site) hete_links++;
      fragment.num_hete_links = hete_links;
      fragment.for_phasing = hete_links >= prm.min_linkers; /* fragment.rs:253-255 */
      /* adjacent-SNP `edges` (fragment.rs:256-292) are written but never read on the live path */
      for (size_t e = 0; e < fragment.list.size(); e++) { Cand& c = cands[fragment.list[e].snp_idx]; c.cover.push_back(fragment.fragment_idx); c.cover_pos.push_back((int)e); }
      frags.push_back(fragment);
    }
  }

  /* ---------- P7: divide_snps_into_blocks (candidate.rs:615-747, snp.rs:158-195) ---------- */
  GraphMap divide_snps_into_blocks() {
    std::vector<int> ld_idxes;
    for (int i = 0; i < (int)cands.size(); i++) if (cands[i].for_phasing) ld_idxes.push_back(i);
    std::vector<std::pair<int, int>> pass_ld_pair;
    for (size_t i = 0; i < ld_idxes.size(); i++)
      for (size_t j = i + 1; j < ld_idxes.size(); j++) {
        const int idx1 = ld_idxes[i], idx2 = ld_idxes[j];
        const Cand &s1 = cands[idx1], &s2 = cands[idx2];
        uint8_t r1, a1, r2, a2; float r1f, a1f, r2f, a2f;
        if (s1.alleles[0] == s1.reference && s1.alleles[1] != s1.reference) { r1 = s1.alleles[0]; r1f = s1.allele_freqs[0]; a1 = s1.alleles[1]; a1f = s1.allele_freqs[1]; }
        else if (s1.alleles[0] != s1.reference && s1.alleles[1] == s1.reference) { r1 = s1.alleles[1]; r1f = s1.allele_freqs[1]; a1 = s1.alleles[0]; a1f = s1.allele_freqs[0]; }
        else continue;
        if (s2.alleles[0] == s2.reference && s2.alleles[1] != s2.reference) { r2 = s2.alleles[0]; r2f = s2.allele_freqs[0]; a2 = s2.alleles[1]; a2f = s2.allele_freqs[1]; }
        else if (s2.alleles[0] != s2.reference && s2.alleles[1] == s2.reference) { r2 = s2.alleles[1]; r2f = s2.allele_freqs[1]; a2 = s2.alleles[0]; a2f = s2.allele_freqs[0]; }
        else continue;
        auto it = allele_pairs.find({idx1, idx2});
        if (it == allele_pairs.end()) continue;
        if (r1f == 0.0f || a1f == 0.0f || r2f == 0.0f || a2f == 0.0f) continue;
        LDPair& lp = it->second;
        int count[4] = {0, 0, 0, 0}; /* snp.rs:158-188 calculate_ld */
        auto get = [&](uint8_t x, uint8_t y) { auto f = lp.ld_pairs.find({x, y}); return f == lp.ld_pairs.end() ? 0 : (int)f->second; };
        count[0] = get(r1, r2); count[1] = get(r1, a2); count[2] = get(a1, r2); count[3] = get(a1, a2);
        int c1 = std::min(count[0] + count[3], count[1] + count[2]);
        int c2 = std::max(count[0] + count[3], count[1] + count[2]);
        float score = (float)c1 / (float)c2;
        int weight;
        if ((count[0] + count[3]) > (count[1] + count[2])) { weight = c2; }
        else { score = -1.0f * score; weight = -1 * c2; }
        lp.score = score; lp.weight = weight; lp.valid = true;
        if (score == 0.0f) pass_ld_pair.push_back({idx1, idx2});
      }
    GraphMap g;
    for (auto& pr : pass_ld_pair) {
      int wgt = allele_pairs[{pr.first, pr.second}].weight;
      if (g.contains_edge(pr.first, pr.second)) g.w[GraphMap::key(pr.first, pr.second)] += wgt;
      else g.add_edge(pr.first, pr.second, wgt);
    }
    /* remove edges with |weight| < threshold (candidate.rs:703-711); with threshold 1 none exist
     * (a passing pair has c2 > 0), restated for completeness: petgraph remove_edge swap_removes
     * adjacency entries. */
    std::vector<std::pair<int, int>> low;
    for (auto& e : g.w) if ((uint32_t)std::abs(e.second) < prm.ld_weight_threshold) low.push_back(e.first);
    for (auto& e : low) {
      g.w.erase(e);
      auto rm = [&](int a, int b) { auto& v = g.adj[a]; auto f = std::find(v.begin(), v.end(), b); if (f != v.end()) { *f = v.back(); v.pop_back(); } };
      rm(e.first, e.second); if (e.first != e.second) rm(e.second, e.first);
    }
    ld_blocks = g.kosaraju_scc();
    return g;
  }

  /* ---------- gather helpers ---------- */
  struct RowView { std::vector<int> delta, eta, ps; std::vector<double> probs; std::vector<uint8_t> q; };
  void row_gather(int k, RowView& v) const {
    v.delta.clear(); v.eta.clear(); v.ps.clear(); v.probs.clear(); v.q.clear();
    for (const FragElem& fe : frags[k].list) {
      if (!fe.phase_site) continue;
      v.ps.push_back(fe.p); v.probs.push_back(fe.prob); v.q.push_back(fe.baseq);
      v.delta.push_back(cands[fe.snp_idx].haplotype); v.eta.push_back(cands[fe.snp_idx].genotype);
    }
  }
  struct ColView { std::vector<int> sigma, ps; std::vector<double> probs; std::vector<uint8_t> q; };
  void col_gather(int i, ColView& v) const { /* phase.rs:883-899 incl. its per-fragment linear search */
    v.sigma.clear(); v.ps.clear(); v.probs.clear(); v.q.clear();
    for (int k : cands[i].cover) {
      if (!frags[k].for_phasing || frags[k].haplotag == 0) continue;
      for (const FragElem& fe : frags[k].list) {
        if (fe.snp_idx == i) {
          if (!fe.phase_site) continue;
          v.ps.push_back(fe.p); v.probs.push_back(fe.prob); v.q.push_back(fe.baseq); v.sigma.push_back(frags[k].haplotag);
        }
      }
    }
  }

  /* ---------- indexed form (orc_set_fast): the phase entries of phase() as CSR (rows, list order) + CSC (SNPs, cover
   * order).  for_phasing of rows / SNPs and phase_site of entries do not change inside phase(), haplotags do. ---------- */
  struct PhaseIdx {
    std::vector<int64_t> rptr, cptr;
    std::vector<int32_t> rsnp, crow;
    std::vector<int8_t> rp, cp;
    std::vector<uint8_t> rq, cq;
  } pidx;
  void build_phase_index() {
    const size_t nf = frags.size(), nc = cands.size();
    pidx = PhaseIdx();
    pidx.rptr.assign(nf + 1, 0); pidx.cptr.assign(nc + 1, 0);
    for (size_t k = 0; k < nf; k++) {
      if (frags[k].for_phasing)
        for (const FragElem& fe : frags[k].list)
          if (fe.phase_site) { pidx.rsnp.push_back(fe.snp_idx); pidx.rp.push_back((int8_t)fe.p); pidx.rq.push_back(fe.baseq); }
      pidx.rptr[k + 1] = (int64_t)pidx.rsnp.size();
    }
    for (size_t i = 0; i < nc; i++) {
      if (cands[i].for_phasing)
        for (size_t j = 0; j < cands[i].cover.size(); j++) {
          const int k = cands[i].cover[j];
          if (!frags[k].for_phasing) continue;
          const FragElem& fe = frags[k].list[cands[i].cover_pos[j]];   /* == the entry the linear search finds */
          if (!fe.phase_site) continue;
          pidx.crow.push_back(k); pidx.cp.push_back((int8_t)fe.p); pidx.cq.push_back(fe.baseq);
        }
      pidx.cptr[i + 1] = (int64_t)pidx.crow.size();
    }
  }
  void par_for(int64_t n, int64_t chunk, const std::function<void(int64_t)>& fn) {
    if (fast_threads > 1) { if (!pool || pool->threads() != fast_threads) { delete pool; pool = new ParPool(fast_threads); } pool->run(n, chunk, fn); }
    else for (int64_t i = 0; i < n; i++) fn(i);
  }

  /* phase.rs:257-276 (F64) / exact fixed-point sum, compared as int64 and reported / 2^40 (EXACT, TIE) */
  mutable int64_t last_obj_fx = 0;
  mutable double last_obj_f64 = 0.0;
  double overall_f64() const {
    double logp = 0.0;
    if (fast_threads) {   /* same entries, same order; log10 from the table of the same libm values */
      for (size_t k = 0; k < frags.size(); k++) {
        if (!frags[k].for_phasing || frags[k].haplotag == 0) continue;
        for (int64_t e = pidx.rptr[k]; e < pidx.rptr[k + 1]; e++) {
          const Cand& c = cands[pidx.rsnp[e]];
          logp += lut_laki(frags[k].haplotag, c.haplotype, c.genotype, pidx.rp[e], pidx.rq[e]);
        }
      }
      return logp;
    }
    for (size_t k = 0; k < frags.size(); k++) {
      if (!frags[k].for_phasing || frags[k].haplotag == 0) continue;
      for (const FragElem& fe : frags[k].list) {
        if (!fe.phase_site) continue;
        logp += std::log10(aki(frags[k].haplotag, cands[fe.snp_idx].haplotype, cands[fe.snp_idx].genotype, fe.p, fe.prob));
      }
    }
    return logp;
  }
  int64_t overall_fx() {
    if (fast_threads) {
      const int64_t nf = (int64_t)frags.size(), CH = 4096, nch = (nf + CH - 1) / CH;
      std::vector<int64_t> part((size_t)nch, 0);
      par_for(nch, 1, [&](int64_t ci) {
        int64_t sum = 0;
        for (int64_t k = ci * CH; k < std::min(nf, (ci + 1) * CH); k++) {
          if (!frags[k].for_phasing || frags[k].haplotag == 0) continue;
          for (int64_t e = pidx.rptr[k]; e < pidx.rptr[k + 1]; e++) {
            const Cand& c = cands[pidx.rsnp[e]];
            sum += fx_aki(frags[k].haplotag, c.haplotype, c.genotype, pidx.rp[e], pidx.rq[e]);
          }
        }
        part[(size_t)ci] = sum;
      });
      int64_t sum = 0;
      for (int64_t v : part) sum += v;   /* integers: any order */
      return sum;
    }
    int64_t sum = 0;
    for (size_t k = 0; k < frags.size(); k++) {
      if (!frags[k].for_phasing || frags[k].haplotag == 0) continue;
      for (const FragElem& fe : frags[k].list) {
        if (!fe.phase_site) continue;
        const Cand& c = cands[fe.snp_idx];
        sum += fx_aki(frags[k].haplotag, c.haplotype, c.genotype, fe.p, fe.baseq);
      }
    }
    return sum;
  }
  double cal_overall_probability(int mode) {
    if (orc_mode_tie(mode)) {   /* the fixed-point sum decides; the f64 sum is kept for ties between configurations (better()) */
      last_obj_fx = overall_fx();
      last_obj_f64 = overall_f64();
      return (double)last_obj_fx / FX_SCALE;
    }
    if (!orc_mode_fx(mode)) {
      if (orc_mode_both(mode)) last_obj_fx = overall_fx();   /* for the tie counter of better() */
      return last_obj_f64 = overall_f64();
    }
    last_obj_fx = overall_fx();
    return (double)last_obj_fx / FX_SCALE;
  }

  /* one row's / one SNP's scores in both arithmetics */
  struct RowDec { bool has = false; int64_t A = 0, B = 0; double q = 0.0, qn = 0.0; };
  struct ColDec { bool has = false; int64_t N[4] = {0, 0, 0, 0}; double q[4] = {0.0, 0.0, 0.0, 0.0}; };
  /* q = cal_sigma_delta_eta_log(sigma_k, ...), qn = the same for -sigma_k (phase.rs:77-96); A / B their fixed-point log sums */
  void row_scores(int k, bool use_fx, bool use_f64, RowView& rv, RowDec& d) const {
    d = RowDec();
    const int sigma_k = frags[k].haplotag;
    if (fast_threads) {
      const int64_t e0 = pidx.rptr[k], e1 = pidx.rptr[k + 1];
      if (e0 == e1) return;
      d.has = true;
      if (use_f64) {   /* the three running sums of phase.rs:82-90, entry order; log_q1 of sigma_k = +1 IS log_q2 */
        double lp = 0.0, lm = 0.0;
        for (int64_t e = e0; e < e1; e++) {
          const Cand& c = cands[pidx.rsnp[e]];
          lp += lut_laki(1, c.haplotype, c.genotype, pidx.rp[e], pidx.rq[e]);
          lm += lut_laki(-1, c.haplotype, c.genotype, pidx.rp[e], pidx.rq[e]);
        }
        const double l1 = sigma_k == 1 ? lp : lm, l1n = sigma_k == 1 ? lm : lp;
        d.q = 1.0 - l1 / (lp + lm); d.qn = 1.0 - l1n / (lp + lm);
      }
      if (use_fx)
        for (int64_t e = e0; e < e1; e++) {
          const Cand& c = cands[pidx.rsnp[e]];
          d.A += fx_aki(sigma_k, c.haplotype, c.genotype, pidx.rp[e], pidx.rq[e]);
          d.B += fx_aki(-sigma_k, c.haplotype, c.genotype, pidx.rp[e], pidx.rq[e]);
        }
      return;
    }
    row_gather(k, rv);
    if (rv.delta.empty()) return;
    d.has = true;
    if (use_f64) {
      d.q = cal_sigma_delta_eta_log(sigma_k, rv.delta, rv.eta, rv.ps, rv.probs);
      d.qn = cal_sigma_delta_eta_log(-sigma_k, rv.delta, rv.eta, rv.ps, rv.probs);
    }
    if (use_fx)
      for (size_t e = 0; e < rv.delta.size(); e++) { d.A += fx_aki(sigma_k, rv.delta[e], rv.eta[e], rv.ps[e], rv.q[e]); d.B += fx_aki(-sigma_k, rv.delta[e], rv.eta[e], rv.ps[e], rv.q[e]); }
  }
  /* q[0..3] = cal_delta_eta_sigma_log of (d,0) (-d,0) (d,1) (d,-1) (phase.rs:128-176); N[0..3] their fixed-point numerators */
  void col_scores(int i, bool use_fx, bool use_f64, ColView& cv, ColDec& d) const {
    d = ColDec();
    const int delta_i = cands[i].haplotype;
    size_t cov = 0;
    if (fast_threads) {
      /* the five running sums of one call are taken over the same entries in the same order whatever (delta, eta) is asked
       * for, so the four calls share four sums: Sd = sum log aki(s, delta_i, 0), Sn = (.., -delta_i, 0), Shr = (.., 1),
       * Shv = (.., -1); each call then adds its priors and forms its own denominator in its own order */
      double Sd = 0.0, Sn = 0.0, Shr = 0.0, Shv = 0.0;
      for (int64_t e = pidx.cptr[i]; e < pidx.cptr[i + 1]; e++) {
        const int sg = frags[pidx.crow[e]].haplotag;
        if (sg == 0) continue;
        cov++;
        const int pp = pidx.cp[e]; const uint8_t qq = pidx.cq[e];
        if (use_f64) {
          Sd += lut_laki(sg, delta_i, 0, pp, qq); Sn += lut_laki(sg, -delta_i, 0, pp, qq);
          Shr += lut_laki(sg, delta_i, 1, pp, qq); Shv += lut_laki(sg, delta_i, -1, pp, qq);
        }
        if (use_fx) {
          d.N[0] += fx_aki(sg, delta_i, 0, pp, qq); d.N[1] += fx_aki(sg, -delta_i, 0, pp, qq);
          d.N[2] += fx_aki(sg, delta_i, 1, pp, qq); d.N[3] += fx_aki(sg, delta_i, -1, pp, qq);
        }
      }
      if (cov == 0) return;
      d.has = true;
      if (use_f64) {
        const double p_homref = prior_homref_log(), p_homvar = prior_homvar_log(), p_het = prior_hetvar_log(cov);
        /* call (dd, ee): log_q1 = S(dd, ee) + prior(ee); log_q2 = Shv + homvar; log_q3 = S(dd, 0) + het; log_q4 = Shr + homref;
         * log_q5 = S(-dd, 0) + het; 1 - log_q1 / (((log_q2 + log_q3) + log_q4) + log_q5) */
        const double hv = Shv + p_homvar, hr = Shr + p_homref, hd = Sd + p_het, hn = Sn + p_het;
        const double den_d = hv + hd + hr + hn;   /* delta_i asked: log_q3 = hd, log_q5 = hn */
        const double den_n = hv + hn + hr + hd;   /* -delta_i asked */
        d.q[0] = 1.0 - hd / den_d; d.q[1] = 1.0 - hn / den_n; d.q[2] = 1.0 - hr / den_d; d.q[3] = 1.0 - hv / den_d;
      }
    } else {
      col_gather(i, cv);
      if (cv.sigma.empty()) return;
      d.has = true; cov = cv.sigma.size();
      if (use_f64) {
        d.q[0] = cal_delta_eta_sigma_log(delta_i, 0, cv.sigma, cv.ps, cv.probs);
        d.q[1] = cal_delta_eta_sigma_log(-delta_i, 0, cv.sigma, cv.ps, cv.probs);
        d.q[2] = cal_delta_eta_sigma_log(delta_i, 1, cv.sigma, cv.ps, cv.probs);
        d.q[3] = cal_delta_eta_sigma_log(delta_i, -1, cv.sigma, cv.ps, cv.probs);
      }
      for (size_t e = 0; use_fx && e < cv.sigma.size(); e++) {
        d.N[0] += fx_aki(cv.sigma[e], delta_i, 0, cv.ps[e], cv.q[e]);
        d.N[1] += fx_aki(cv.sigma[e], -delta_i, 0, cv.ps[e], cv.q[e]);
        d.N[2] += fx_aki(cv.sigma[e], delta_i, 1, cv.ps[e], cv.q[e]);
        d.N[3] += fx_aki(cv.sigma[e], delta_i, -1, cv.ps[e], cv.q[e]);
      }
    }
    if (use_fx) {
      const int64_t het = plut().f_het0 - (int64_t)cov * plut().f_log2;
      d.N[0] += het; d.N[1] += het; d.N[2] += plut().f_homref; d.N[3] += plut().f_homvar;
    }
  }
  /* the reference's choice among the four scores: first maximum (phase.rs:905-940); -1 = NaN scores */
  static int choose_f64(const double q[4], bool with_genotype, int eta_i) {
    if (with_genotype) {
      const double max_q = std::fmax(q[0], std::fmax(q[1], std::fmax(q[2], q[3])));
      return q[0] == max_q ? 0 : q[1] == max_q ? 1 : q[2] == max_q ? 2 : q[3] == max_q ? 3 : -1;
    }
    if (eta_i == 0) { const double max_q = std::fmax(q[0], q[1]); return q[0] == max_q ? 0 : q[1] == max_q ? 1 : -1; }
    const double max_q = std::fmax(q[2], q[3]);
    return q[2] == max_q ? 2 : q[3] == max_q ? 3 : -1;
  }
  static int choose_fx(const int64_t N[4], bool with_genotype, int eta_i, bool* tie) {
    int ch;
    if (with_genotype) { ch = 0; for (int t = 1; t < 4; t++) if (N[t] > N[ch]) ch = t; *tie = false; for (int t = 0; t < 4; t++) if (t != ch && N[t] == N[ch]) *tie = true; }
    else if (eta_i == 0) { ch = N[1] > N[0] ? 1 : 0; *tie = N[1] == N[0]; }
    else { ch = N[3] > N[2] ? 3 : 2; *tie = N[3] == N[2]; }
    return ch;
  }

  /* ---------- P12: cross_optimize (phase.rs:810-976) ---------- */
  double cross_optimize(int mode, const std::set<int>& conserved, bool keep_conserved, bool with_genotype) {
    stats[0]++;
    /* by_fx: decisions by the exact fixed-point sums (else by the reference's f64 ratio scores); both: the other
     * arithmetic is evaluated too and disagreements are counted (stats[2]); the *_ONLY modes skip it; tie: fixed point,
     * exact ties by the f64 scores (ORC_MODE_TIE) */
    const bool by_fx = orc_mode_fx(mode), both = orc_mode_both(mode), tie = orc_mode_tie(mode);
    const bool use_fx = by_fx || both || tie, use_f64 = !by_fx || both;   /* (TIE evaluates the f64 scores of every row too: the oracle is not the place to save them) */
    bool hg_inc = true, h_inc = true;
    int num_iters = 0;
    RowView rv; ColView cv;
    const int nf = (int)frags.size(), nc = (int)cands.size();
    std::vector<RowDec> rdec; std::vector<ColDec> cdec;
    if (fast_threads) { rdec.resize(nf); cdec.resize(nc); }
    while (hg_inc | h_inc) {
      stats[1]++;
      /* sigma step, phase.rs:824-862 */
      std::map<int, int> tmp_haplotag;
      double logp = 0.0, pre_logp = 0.0; bool any_strict = false, any_tie_change = false;
      /* one row's decision from its scores: new haplotag, the two terms of check_new_haplotag's sums, counters */
      struct RowOut { bool has = false, strict = false, tiechg = false; int newtag = 0; double q_old = 0.0, q_new = 0.0; int c[5] = {0, 0, 0, 0, 0}; };
      auto row_decide = [&](int k, const RowDec& d) -> RowOut {
        RowOut o;
        if (!d.has) return o;
        o.has = true;
        const int sigma_k = frags[k].haplotag;
        const bool flip_f64 = d.q < d.qn, flip_fx = d.A < d.B;
        if (both && flip_f64 != flip_fx) o.c[4] = 1;   /* stats[2] */
        bool flip = by_fx ? flip_fx : flip_f64;
        if (use_fx && d.A == d.B) {
          o.c[0] = 1; if (flip_f64) o.c[1] = 1;         /* census[0], [4] */
          bool het = false;   /* a row without an entry at a het site scores the same for both signs, term by term */
          for (const FragElem& fe : frags[k].list) if (fe.phase_site && cands[fe.snp_idx].genotype == 0) het = true;
          if (het) o.c[2] = 1;                           /* census[8] */
          if (d.q != d.qn) o.c[3] = 1;                   /* census[9] */
        }
        if (tie) {
          flip = flip_fx;
          if (d.A == d.B && (tie_mask & 1)) flip = flip_f64;
          if (d.A < d.B) o.strict = true; else if (flip) o.tiechg = true;
        } else if (flip) o.strict = true;
        o.newtag = flip ? -sigma_k : sigma_k;
        /* check_new_haplotag, phase.rs:278-314 (sums in key order instead of HashMap order) */
        o.q_new = flip ? d.qn : d.q; o.q_old = d.q;
        return o;
      };
      std::vector<RowOut> rout;
      if (fast_threads) {   /* Jacobi: every row is scored against the old state (phase.rs:859-862) -- independent */
        rout.resize(nf);
        par_for(nf, 512, [&](int64_t k) {
          rdec[k] = RowDec(); rout[k] = RowOut();
          if (!frags[k].for_phasing || frags[k].haplotag == 0) return;
          RowView dummy; row_scores((int)k, use_fx, use_f64, dummy, rdec[k]);
          rout[k] = row_decide((int)k, rdec[k]);
        });
        const int64_t CH = 4096, nch = (nf + CH - 1) / CH;
        std::vector<std::array<int64_t, 7>> part((size_t)nch);
        par_for(nch, 1, [&](int64_t ci) {
          std::array<int64_t, 7> a = {0, 0, 0, 0, 0, 0, 0};
          for (int64_t k = ci * CH; k < std::min<int64_t>(nf, (ci + 1) * CH); k++) {
            const RowOut& o = rout[k];
            if (!o.has) continue;
            for (int x = 0; x < 5; x++) a[x] += o.c[x];
            a[5] |= o.strict ? 1 : 0; a[6] |= o.tiechg ? 1 : 0;
          }
          part[(size_t)ci] = a;
        });
        for (const auto& a : part) { census[0] += a[0]; census[4] += a[1]; census[8] += a[2]; census[9] += a[3]; stats[2] += a[4]; any_strict |= a[5] != 0; any_tie_change |= a[6] != 0; }
        /* the sums of check_new_haplotag, in key order -- wherever their value is looked at */
        if (!tie ? !by_fx : (!any_strict && any_tie_change))
          for (int k = 0; k < nf; k++) if (rout[k].has) { logp += rout[k].q_new; pre_logp += rout[k].q_old; }
      } else
      for (int k = 0; k < nf; k++) {
        if (!frags[k].for_phasing || frags[k].haplotag == 0) continue;
        RowDec d;
        row_scores(k, use_fx, use_f64, rv, d);
        const RowOut o = row_decide(k, d);
        if (!o.has) continue;
        census[0] += o.c[0]; census[4] += o.c[1]; census[8] += o.c[2]; census[9] += o.c[3]; stats[2] += o.c[4];
        any_strict |= o.strict; any_tie_change |= o.tiechg;
        tmp_haplotag[k] = o.newtag;
        logp += o.q_new; pre_logp += o.q_old;
      }
      int check_val;
      if (tie) {
        check_val = any_strict ? 1 : 0;
        if (!any_strict && any_tie_change) { census[2]++; if (logp > pre_logp) census[6]++; if (tie_mask & 4) check_val = logp > pre_logp ? 1 : 0; }
      } else if (!by_fx) { check_val = logp > pre_logp ? 1 : (logp == pre_logp ? 0 : -1); if (check_val < 0) { stats[3]++; check_val = 0; } }
      else check_val = any_strict ? 1 : 0;
      if (fast_threads) par_for(nf, 4096, [&](int64_t k) { if (rout[k].has) frags[k].haplotag = rout[k].newtag; });
      for (auto& kv : tmp_haplotag) frags[kv.first].haplotag = kv.second;
      if (check_val == 0) h_inc = false; else { h_inc = true; hg_inc = true; }
      /* delta/eta step, phase.rs:872-959 */
      std::map<int, std::pair<int, int>> tmp_hg;
      logp = 0.0; pre_logp = 0.0; any_strict = false; any_tie_change = false;
      if (fast_threads)
        par_for(nc, 1, [&](int64_t i) {
          cdec[i] = ColDec();
          if (!cands[i].for_phasing) return;
          if (keep_conserved && conserved.count((int)i)) return;
          ColView dummy; col_scores((int)i, use_fx, use_f64, dummy, cdec[i]);
        });
      for (int i = 0; i < nc; i++) {
        if (!cands[i].for_phasing) continue;
        if (keep_conserved && conserved.count(i)) continue;
        const int delta_i = cands[i].haplotype, eta_i = cands[i].genotype;
        ColDec d;
        if (fast_threads) d = cdec[i]; else col_scores(i, use_fx, use_f64, cv, d);
        if (!d.has) continue;
        int ch_f64 = -1, ch_fx = -1; /* 0:(d,0) 1:(-d,0) 2:(d,1) 3:(d,-1) */
        bool fx_tie = false;
        if (use_f64) ch_f64 = choose_f64(d.q, with_genotype, eta_i);
        if (use_fx) ch_fx = choose_fx(d.N, with_genotype, eta_i, &fx_tie);
        if (both && ch_f64 != ch_fx) stats[2]++;
        if (use_fx && fx_tie) { census[1]++; if (ch_f64 != ch_fx) census[5]++; }
        int ch = by_fx ? ch_fx : ch_f64;
        if (tie) { ch = ch_fx; if (fx_tie && (tie_mask & 2)) ch = ch_f64; }
        if (ch < 0) continue; /* NaN scores: the reference inserts nothing (or panics) */
        const std::pair<int, int> pick[4] = {{delta_i, 0}, {-delta_i, 0}, {delta_i, 1}, {delta_i, -1}};
        tmp_hg[i] = pick[ch];
        /* check_new_haplotype_genotype, phase.rs:316-355 */
        const int cur = eta_i == 0 ? 0 : (eta_i == 1 ? 2 : 3);
        logp += d.q[ch]; pre_logp += d.q[cur];
        if (d.N[ch] > d.N[cur]) any_strict = true; else if (ch != cur) any_tie_change = true;
      }
      if (tie) {
        check_val = any_strict ? 1 : 0;
        if (!any_strict && any_tie_change) { census[2]++; if (logp > pre_logp) census[6]++; if (tie_mask & 4) check_val = logp > pre_logp ? 1 : 0; }
      } else if (!by_fx) { check_val = logp > pre_logp ? 1 : (logp == pre_logp ? 0 : -1); if (check_val < 0) { stats[3]++; check_val = 0; } }
      else check_val = any_strict ? 1 : 0;
      for (auto& kv : tmp_hg) { cands[kv.first].haplotype = kv.second.first; cands[kv.first].genotype = kv.second.second; }
      if (check_val == 0) hg_inc = false; else { hg_inc = true; h_inc = true; }
      num_iters++;
      if (num_iters > 20) break; /* phase.rs:968-972 */
    }
    if (getenv("ORC_DEBUG_ITERS")) fprintf(stderr, "[orc] cross_optimize iters %d\n", num_iters);
    return cal_overall_probability(mode);
  }

  /* ---------- cross_optimize_by_block (phase.rs:1298-1394), f64 in both modes ---------- */
  double cross_optimize_by_block(int mode) {
    std::map<int, int> tmp_haplotype, tmp_haplotag;
    for (const auto& block : ld_blocks) {
      std::vector<int> delta_block, delta_block_flip, eta_block;
      std::vector<std::vector<int>> sigma_block, sigma_block_flip, ps_block;
      std::vector<std::vector<double>> probs_block;
      std::map<int, int> sigma_flip_map;
      std::set<int> block_set(block.begin(), block.end());
      std::vector<char> in_blk;   /* indexed form: the same membership test as a marker array */
      if (fast_threads) { in_blk.assign(cands.size(), 0); for (int x : block) in_blk[x] = 1; }
      for (int idx : block) {
        delta_block.push_back(cands[idx].haplotype);
        delta_block_flip.push_back(-cands[idx].haplotype);
        eta_block.push_back(cands[idx].genotype);
        std::vector<int> sigma, sigma_flip, ps; std::vector<double> probs;
        for (size_t cj = 0; cj < cands[idx].cover.size(); cj++) {
          const int k = cands[idx].cover[cj];
          if (!frags[k].for_phasing || frags[k].haplotag == 0) continue;
          bool flip_read = true;
          /* indexed form: the walk stops at the SNP's own entry (list[cover_pos]) -- nothing behind it is looked at by the
           * reference's loop either: flip_read is consumed at the match and no second entry matches */
          const size_t e_end = fast_threads ? (size_t)cands[idx].cover_pos[cj] + 1 : frags[k].list.size();
          for (size_t e = 0; e < e_end; e++) {
            const FragElem& fe = frags[k].list[e];
            if (fast_threads ? !in_blk[fe.snp_idx] : !block_set.count(fe.snp_idx)) flip_read = false;
            if (fe.snp_idx == idx) {
              if (!fe.phase_site) continue;
              ps.push_back(fe.p); probs.push_back(fe.prob);
              if (flip_read) { sigma_flip.push_back(-frags[k].haplotag); sigma_flip_map[k] = -frags[k].haplotag; }
              else { sigma_flip.push_back(frags[k].haplotag); sigma_flip_map[k] = frags[k].haplotag; }
              sigma.push_back(frags[k].haplotag);
            }
          }
        }
        sigma_block.push_back(sigma); sigma_block_flip.push_back(sigma_flip); ps_block.push_back(ps); probs_block.push_back(probs);
      }
      const double q = cal_block_delta_eta_sigma_log(delta_block, eta_block, sigma_block, ps_block, probs_block);
      const double q_flip = cal_block_delta_eta_sigma_log(delta_block_flip, eta_block, sigma_block_flip, ps_block, probs_block);
      if (q < q_flip) {
        for (size_t i = 0; i < block.size(); i++) tmp_haplotype[block[i]] = delta_block_flip[i];
        for (int k = 0; k < (int)frags.size(); k++) {
          auto f = sigma_flip_map.find(k);
          tmp_haplotag[k] = f != sigma_flip_map.end() ? f->second : frags[k].haplotag;
        }
      } else {
        for (size_t i = 0; i < block.size(); i++) tmp_haplotype[block[i]] = delta_block[i];
        for (int k = 0; k < (int)frags.size(); k++) tmp_haplotag[k] = frags[k].haplotag;
      }
    }
    for (auto& kv : tmp_haplotype) cands[kv.first].haplotype = kv.second;
    for (auto& kv : tmp_haplotag) frags[kv.first].haplotag = kv.second;
    return cal_overall_probability(mode);
  }

  struct Best { std::vector<int> hap, gen, tag; };
  void save_best(Best& b) const {
    b.hap.resize(cands.size()); b.gen.resize(cands.size()); b.tag.resize(frags.size());
    for (size_t i = 0; i < cands.size(); i++) { b.hap[i] = cands[i].haplotype; b.gen[i] = cands[i].genotype; }
    for (size_t k = 0; k < frags.size(); k++) b.tag[k] = frags[k].haplotag;
  }
  void load_best(const Best& b) {
    for (size_t i = 0; i < cands.size(); i++) { cands[i].haplotype = b.hap[i]; cands[i].genotype = b.gen[i]; }
    for (size_t k = 0; k < frags.size(); k++) frags[k].haplotag = b.tag[k];
  }
  void init_assignment() { /* phase.rs:673-680 */
    for (auto& f : frags) if (f.for_phasing) f.haplotag = rnd() < 0.5 ? -1 : 1;
  }
  void init_genotype() { /* phase.rs:682-691 */
    for (auto& c : cands) {
      if (c.variant_type == 0) c.genotype = 1; else if (c.variant_type == 1) c.genotype = 0;
      else if (c.variant_type == 2 || c.variant_type == 3) c.genotype = -1;
    }
  }
  /* phase.rs:609-671 with petgraph Bfs (visit/traversal.rs) */
  std::set<int> init_haplotypes_LD2(const GraphMap& g) {
    for (auto& c : cands) c.haplotype = rnd() < 0.5 ? 1 : -1;
    std::set<int> conserved;
    const int thr = (int)prm.ld_weight_threshold;
    for (const auto& block : ld_blocks) {
      if (block.size() < 2) continue;
      std::set<int> discovered; std::vector<int> queue; size_t qh = 0;
      discovered.insert(block[0]); queue.push_back(block[0]);
      std::vector<int> visited_nodes;
      cands[block[0]].haplotype = 1;
      visited_nodes.push_back(block[0]);
      while (qh < queue.size()) {
        const int nx = queue[qh++];
        for (int succ : g.adj.at(nx)) if (discovered.insert(succ).second) queue.push_back(succ);
        for (int visited_idx : visited_nodes) {
          int from_idx, to_idx;
          if (visited_idx < nx) { from_idx = visited_idx; to_idx = nx; }
          else if (visited_idx > nx) { from_idx = nx; to_idx = visited_idx; }
          else continue;
          auto it = allele_pairs.find({from_idx, to_idx});
          if (it == allele_pairs.end()) continue;
          if (!it->second.valid) continue;
          if (it->second.score != 0.0f) continue;
          const int weight = it->second.weight;
          if (weight >= thr) { cands[nx].haplotype = cands[visited_idx].haplotype; break; }
          else if (weight <= -thr) { cands[nx].haplotype = -cands[visited_idx].haplotype; break; }
        }
        visited_nodes.push_back(nx);
      }
      for (int idx : block) conserved.insert(idx);
    }
    return conserved;
  }

  /* ---------- thread.rs:162-166 + P13 SNPFrag::phase (phase.rs:1087-1296) ---------- */
  void phase(int mode) {
    ctr = 0;
    for (int i = 0; i < 10; i++) census[i] = 0;
    const int tie_mask_all = tie_mask;   /* bit 16: bits 8-15 are the mask of the chain branch (S > max_enum_snps) */
    if (tie_mask_all & 0x10000) tie_mask = cands.size() <= prm.max_enum_snps ? (tie_mask_all & 255) : ((tie_mask_all >> 8) & 255);
    if (fast_threads) build_phase_index();
    for (auto& c : cands) c.haplotype = rnd() < 0.5 ? 1 : -1; /* init_haplotypes, phase.rs:443-448 */
    init_assignment();
    double largest_prob = -std::numeric_limits<double>::infinity();
    int64_t largest_fx = std::numeric_limits<int64_t>::min();
    double largest_f64 = -std::numeric_limits<double>::infinity();   /* ORC_MODE_TIE: the f64 objective of the best configuration */
    auto better = [&](double prob) {  /* `prob > largest_prob` (phase.rs:1117,1129,...); EXACT compares the int64 sums */
      const bool b_f64 = prob > largest_prob, b_fx = last_obj_fx > largest_fx;
      if (orc_mode_both(mode) && b_f64 != b_fx) stats[2]++;   /* two restarts of equal objective (an uninformative SNP flipped): the f64 sums differ by rounding noise */
      bool b = orc_mode_fx(mode) ? b_fx : b_f64;
      if (orc_mode_tie(mode)) {   /* equal fixed-point objectives: the reference's f64 sums of the two configurations decide */
        b = b_fx;
        if (last_obj_fx == largest_fx) { census[3]++; if (last_obj_f64 > largest_f64) census[7]++; if (tie_mask & 8) b = last_obj_f64 > largest_f64; }
      } else if (orc_mode_both(mode) && last_obj_fx == largest_fx) { census[3]++; if (b_f64) census[7]++; }
      if (b) { largest_prob = prob; largest_fx = last_obj_fx; largest_f64 = last_obj_f64; }
      return b;
    };
    Best best;
    std::set<int> conserved;
    GraphMap ld_graph = divide_snps_into_blocks();
    const size_t S = cands.size();
    if (S <= prm.max_enum_snps) {
      std::vector<std::vector<int>> haplotype_enum(1, std::vector<int>(S, 1));
      for (size_t ti = 0; ti < S; ti++) {
        const size_t n0 = haplotype_enum.size();
        for (size_t tj = 0; tj < n0; tj++) { std::vector<int> t = haplotype_enum[tj]; t[ti] = -t[ti]; haplotype_enum.push_back(t); }
      }
      for (const auto& hap : haplotype_enum) {
        for (size_t i = 0; i < S; i++) cands[i].haplotype = hap[i];
        init_assignment();
        init_genotype();
        const double prob = cross_optimize(mode, conserved, false, true);
        if (better(prob)) save_best(best);
      }
      load_best(best);
    } else {
      conserved = init_haplotypes_LD2(ld_graph);
      init_genotype();
      init_assignment();
      double prob = cross_optimize(mode, conserved, true, false);
      if (better(prob)) save_best(best);
      load_best(best);
      prob = cross_optimize_by_block(mode);
      if (better(prob)) save_best(best);
      load_best(best);
      for (size_t tidx = 0; tidx <= S / 4; tidx++) {
        const bool flip = tidx % 2 == 1;
        for (auto& c : cands) {
          const double rg = rnd();
          if (rg < 0.1) c.haplotype = flip ? 1 : -1;
          else if (rg >= 0.9) c.haplotype = flip ? -1 : 1;
        }
        prob = cross_optimize(mode, conserved, false, false);
        { const bool b_ = better(prob); if (b_) save_best(best); if (round_log) round_log->push_back(b_ ? 1 : 0); }
        load_best(best);
        for (auto& f : frags) {
          if (!f.for_phasing || f.haplotag == 0) continue;
          if (rnd() < 0.1) f.haplotag *= -1;
        }
        prob = cross_optimize(mode, conserved, false, false);
        { const bool b_ = better(prob); if (b_) save_best(best); if (round_log) round_log->push_back(b_ ? 1 : 0); }
        load_best(best);
      }
      load_best(best);
    }
    best_objective = largest_prob;
    tie_mask = tie_mask_all;
  }

  /* ---------- P14: assign_reads_haplotype (snpfrags.rs:548-625) ---------- */
  void assign_reads_haplotype(double cutoff) {
    for (auto& f : frags) {
      if (!f.for_phasing) continue;
      const int sigma_k = f.haplotag;
      std::vector<int> delta, eta, ps; std::vector<double> probs;
      for (auto& fe : f.list) {
        if (!fe.phase_site && cands[fe.snp_idx].for_phasing) fe.phase_site = true;
        if (!cands[fe.snp_idx].for_phasing) continue;
        if (cands[fe.snp_idx].haplotype == 0) continue;
        if (cands[fe.snp_idx].genotype != 0) continue;
        ps.push_back(fe.p); probs.push_back(fe.prob);
        delta.push_back(cands[fe.snp_idx].haplotype); eta.push_back(cands[fe.snp_idx].genotype);
      }
      if (sigma_k == 0) { f.assignment = 0; f.haplotag = 0; f.assignment_score = 0.0; continue; }
      if (delta.empty()) { f.assignment = 0; f.haplotag = 0; f.assignment_score = 0.0; continue; }
      const double q = cal_sigma_delta_eta_log(sigma_k, delta, eta, ps, probs);
      const double qn = cal_sigma_delta_eta_log(-sigma_k, delta, eta, ps, probs);
      if (std::fabs(q - qn) >= cutoff) {
        if (q >= qn) { f.assignment = sigma_k == 1 ? 1 : 2; f.assignment_score = q; }
        else if (sigma_k == 1) { f.assignment = 2; f.assignment_score = qn; f.haplotag = -1; }
        else { f.assignment = 1; f.assignment_score = qn; f.haplotag = 1; }
      } else { f.assignment = 0; f.haplotag = 0; f.assignment_score = 0.0; }
    }
  }

  /* ---------- P15: assign_snp_haplotype_genotype (snpfrags.rs:378-546) ---------- */
  void assign_snp_haplotype_genotype() {
    for (int ti = 0; ti < (int)cands.size(); ti++) {
      Cand& snp = cands[ti];
      if (!snp.for_phasing) { snp.non_selected = true; continue; }
      if (snp.cover.empty()) { snp.single = true; continue; }
      const int delta_i = snp.haplotype;
      std::vector<int> sigma, ps; std::vector<double> probs;
      int hap1 = 0, hap2 = 0;
      for (size_t cj = 0; cj < snp.cover.size(); cj++) {
        const int k = snp.cover[cj];
        if (!frags[k].for_phasing || frags[k].num_hete_links < prm.min_linkers) continue;
        if (snp.variant_type == 1 && frags[k].assignment == 0) continue;
        /* (indexed form: the one matching entry directly instead of the search for it) */
        const size_t e0 = fast_threads ? (size_t)snp.cover_pos[cj] : 0, e1 = fast_threads ? e0 + 1 : frags[k].list.size();
        for (size_t e = e0; e < e1; e++) {
          const FragElem& fe = frags[k].list[e];
          if (fe.snp_idx == ti) {
            if (frags[k].assignment == 1) hap1++; else if (frags[k].assignment == 2) hap2++;
            ps.push_back(fe.p); probs.push_back(fe.prob); sigma.push_back(frags[k].haplotag);
          }
        }
      }
      if (sigma.empty()) { snp.non_selected = true; continue; }
      const double q1 = cal_delta_eta_sigma_log(delta_i, 0, sigma, ps, probs);
      const double q2 = cal_delta_eta_sigma_log(-delta_i, 0, sigma, ps, probs);
      const double q3 = cal_delta_eta_sigma_log(delta_i, 1, sigma, ps, probs);
      const double q4 = cal_delta_eta_sigma_log(delta_i, -1, sigma, ps, probs);
      const double max_q = std::fmax(q1, std::fmax(q2, std::fmax(q3, q4)));
      if (q1 == max_q) { snp.haplotype = delta_i; snp.genotype = 0; snp.variant_type = 1; }
      else if (q2 == max_q) { snp.haplotype = -delta_i; snp.genotype = 0; snp.variant_type = 1; }
      else if (q3 == max_q) { snp.haplotype = delta_i; snp.genotype = 1; snp.variant_type = 0; }
      else if (q4 == max_q) { snp.haplotype = delta_i; snp.genotype = -1; if (snp.variant_type != 2 && snp.variant_type != 3) snp.variant_type = 2; }
      else { stats[3]++; continue; } /* reference panics (NaN) */
      if (snp.genotype != 0) { snp.non_selected = true; continue; }
      if (!sigma.empty() && hap1 >= 1 && hap2 >= 1)
        snp.phase_score = -10.0 * std::log10(1.0 - cal_phase_score_log(snp.haplotype, snp.genotype, sigma, ps, probs));
      else
        snp.phase_score = 0.19940219;
    }
  }

  /* ---------- P16: eval_rna_edit_var_phase / eval_low_frac_var_phase (snpfrags.rs:191-376) ---------- */
  void eval_rescue(const std::vector<int>& list, float min_phase_score, bool low_frac) {
    for (int ti : list) {
      Cand& snp = cands[ti];
      if (snp.cover.empty()) { snp.single = true; continue; }
      if (snp.variant_type != 1) { snp.non_selected = true; continue; }
      std::vector<int> sigma, ps; std::vector<double> probs;
      int hap1 = 0, hap2 = 0;
      for (size_t cj = 0; cj < snp.cover.size(); cj++) {
        const int k = snp.cover[cj];
        if (!frags[k].for_phasing || frags[k].assignment == 0 || frags[k].num_hete_links < prm.min_linkers) continue;
        const size_t e0 = fast_threads ? (size_t)snp.cover_pos[cj] : 0, e1 = fast_threads ? e0 + 1 : frags[k].list.size();
        for (size_t e = e0; e < e1; e++) {
          const FragElem& fe = frags[k].list[e];
          if (fe.snp_idx == ti) {
            if (frags[k].assignment == 1) hap1++; else if (frags[k].assignment == 2) hap2++;
            ps.push_back(fe.p); probs.push_back(fe.prob); sigma.push_back(frags[k].haplotag);
          }
        }
      }
      if (sigma.empty() || hap1 < 2 || hap2 < 2) { snp.single = true; continue; }
      const double ps1 = -10.0 * std::log10(1.0 - cal_phase_score_log(1, 0, sigma, ps, probs));
      const double ps2 = -10.0 * std::log10(1.0 - cal_phase_score_log(-1, 0, sigma, ps, probs));
      snp.single = false;
      if (std::fmax(ps1, ps2) >= (double)min_phase_score) {
        snp.non_selected = false;
        if (low_frac) snp.cand_somatic = false;
        snp.rna_editing = false;
        snp.for_phasing = true;
        for (int k : snp.cover) {
          frags[k].for_phasing = true;
          if (frags[k].haplotag == 0 || frags[k].assignment == 0) frags[k].haplotag = rnd() < 0.5 ? -1 : 1;
        }
        snp.haplotype = ps1 >= ps2 ? 1 : -1;
        snp.genotype = 0;
        snp.variant_type = 1;
        snp.phase_score = std::fmax(ps1, ps2);
      } else {
        snp.non_selected = true;
        if (low_frac) { snp.cand_somatic = true; snp.for_phasing = false; }
        else snp.rna_editing = true;
      }
    }
  }

  /* ---------- P17: assign_phase_set (snpfrags.rs:628-733) ---------- */
  void assign_phase_set(float min_phase_score) {
    read_phase_set.clear();
    GraphMap graph;
    std::map<std::pair<int, int>, std::vector<int>> efrags;
    for (int i = 0; i < (int)cands.size(); i++) {
      const Cand& snp = cands[i];
      if (snp.genotype != 0 || snp.variant_type != 1) continue;
      if (snp.dense || snp.rna_editing) continue;
      if (snp.phase_score < (double)min_phase_score) continue;
      graph.add_node(i);
    }
    for (int k = 0; k < (int)frags.size(); k++) {
      const Fragment& frag = frags[k];
      if (!frag.for_phasing || frag.assignment == 0) continue;
      std::vector<int> node_snps;
      for (const FragElem& fe : frag.list) if (graph.contains_node(fe.snp_idx)) node_snps.push_back(fe.snp_idx);
      if (node_snps.size() == 1) {
        if (!graph.contains_edge(node_snps[0], node_snps[0])) graph.add_edge(node_snps[0], node_snps[0], 0);
        efrags[GraphMap::key(node_snps[0], node_snps[0])].push_back(k);
      }
      if (node_snps.size() >= 2) {
        for (size_t j0 = 0; j0 < node_snps.size(); j0++)
          for (size_t j1 = 0; j1 < node_snps.size(); j1++) {
            if (j0 == j1) continue;
            const int hp0 = cands[node_snps[j0]].haplotype, hp1 = cands[node_snps[j1]].haplotype;
            int ap0 = 0, ap1 = 0;
            for (const FragElem& fe : frag.list) {
              if (fe.snp_idx == node_snps[j0]) ap0 = fe.p;
              else if (fe.snp_idx == node_snps[j1]) ap1 = fe.p;
            }
            if (hp0 * hp1 != ap0 * ap1) continue;
            if (!graph.contains_edge(node_snps[j0], node_snps[j1])) graph.add_edge(node_snps[j0], node_snps[j1], 0);
            efrags[GraphMap::key(node_snps[j0], node_snps[j1])].push_back(k);
          }
      }
    }
    auto scc = graph.kosaraju_scc();
    for (const auto& comp : scc) {
      uint32_t phase_id = 0;
      for (int node : comp) {
        if (phase_id == 0) phase_id = (uint32_t)(cands[node].pos + 1);
        cands[node].phase_set = phase_id;
        for (int nb : graph.adj.at(node)) {
          for (int k : efrags[GraphMap::key(node, nb)]) {
            if (read_phase_set.count(k)) continue;
            read_phase_set[k] = phase_id;
          }
        }
      }
    }
  }

  void post_phase() { /* thread.rs:168-201 */
    assign_reads_haplotype(prm.read_assign_cutoff);
    assign_snp_haplotype_genotype();
    assign_reads_haplotype(prm.read_assign_cutoff);
    assign_snp_haplotype_genotype();
    const float relaxed = prm.min_phase_score - 3.0f;
    eval_rescue(edit_snps, relaxed, false);
    eval_rescue(somatic_snps, relaxed, true);
    assign_reads_haplotype(prm.read_assign_cutoff);
    assign_snp_haplotype_genotype();
    assign_phase_set(prm.min_phase_score);
  }

  /* ---------- P18: output_phased_vcf (vcf.rs:27-306) + writer (thread.rs:266-303) ---------- */
  std::string vcf_text(const char* chrom) const {
    std::string out;
    char line[512], gtbuf[256];
    const float mps = prm.min_phase_score;
    for (const Cand& snp : cands) {
      std::vector<char> alt; float af[2] = {0.0f, 0.0f};
      const char* filter = ""; const char* info = ""; const char* format = ""; const char* gt = "0/0";
      auto one_alt = [&]() {
        if (snp.alleles[0] != snp.reference) { alt = {snp.alleles[0]}; af[0] = snp.allele_freqs[0]; }
        else if (snp.alleles[1] != snp.reference) { alt = {snp.alleles[1]}; af[0] = snp.allele_freqs[1]; }
      };
      auto two_alt = [&]() { alt = {snp.alleles[0], snp.alleles[1]}; af[0] = snp.allele_freqs[0]; af[1] = snp.allele_freqs[1]; };
      if (snp.dense) {
        if (snp.variant_type == 1 || snp.variant_type == 2) one_alt();
        else if (snp.variant_type == 3) two_alt();
        filter = "dn"; info = "RDS=dense_snp";
        if (snp.variant_type == 1) gt = "0/1"; else if (snp.variant_type == 2) gt = "1/1"; else if (snp.variant_type == 3) gt = "1/2"; else continue;
        if (snp.variant_type == 3) snprintf(gtbuf, sizeof gtbuf, "%s:%d:%u:%.2f,%.2f", gt, as_i32(snp.genotype_quality), snp.depth, af[0], af[1]);
        else snprintf(gtbuf, sizeof gtbuf, "%s:%d:%u:%.2f", gt, as_i32(snp.genotype_quality), snp.depth, af[0]);
        format = "GT:GQ:DP:AF";
      } else if (snp.non_selected) {
        if (snp.rna_editing) {
          if (snp.variant_type == 1 || snp.variant_type == 2) one_alt(); else continue;
          filter = "RnaEdit"; info = "RDS=noselect";
          if (snp.variant_type == 1) gt = "0/1"; else if (snp.variant_type == 2) gt = "1/1";
          snprintf(gtbuf, sizeof gtbuf, "%s:%d:%u:%.2f", gt, as_i32(snp.genotype_quality), snp.depth, af[0]);
          format = "GT:GQ:DP:AF";
        } else {
          if (snp.variant_type == 0 || snp.variant_type == 1 || snp.variant_type == 2) {
            one_alt();
            if (snp.variant_type == 0) { gt = "0/0"; filter = "HomRef"; }
            else if (snp.variant_type == 1) { gt = "0/1"; filter = "LowQual"; }
            else { gt = "1/1"; filter = "PASS"; }
          } else {
            if (snp.genotype == -1 || snp.genotype == 1) {
              one_alt();
              if (snp.genotype == -1) { gt = "1/1"; filter = "PASS"; } else { gt = "0/0"; filter = "HomRef"; }
            } else if (snp.genotype == 0) { two_alt(); gt = "1/2"; filter = "Multiallelic"; }
          }
          info = "RDS=noselect";
          if (!strcmp(gt, "0/0") || !strcmp(gt, "0/1") || !strcmp(gt, "1/1")) snprintf(gtbuf, sizeof gtbuf, "%s:%d:%u:%.2f", gt, as_i32(snp.genotype_quality), snp.depth, af[0]);
          else snprintf(gtbuf, sizeof gtbuf, "%s:%d:%u:%.2f,%.2f", gt, as_i32(snp.genotype_quality), snp.depth, af[0], af[1]);
          format = "GT:GQ:DP:AF";
        }
      } else {
        if (snp.phase_score >= (double)mps) {
          if (snp.variant_type == 1) { one_alt(); gt = snp.haplotype == 1 ? "0|1" : "1|0"; filter = "PASS"; }
        } else {
          if (snp.variant_type == 0) { one_alt(); gt = "0/0"; filter = "HomRef"; }
          else if (snp.variant_type == 1) { one_alt(); gt = "0/1"; filter = "LowQual"; }
          else if (snp.variant_type == 2) { one_alt(); gt = "1/1"; filter = "PASS"; }
          else {
            if (snp.genotype == -1 || snp.genotype == 1) {
              one_alt();
              if (snp.genotype == -1) { gt = "1/1"; filter = "PASS"; } else { gt = "0/0"; filter = "HomRef"; }
            } else if (snp.genotype == 0) { two_alt(); gt = "1/2"; filter = "Multiallelic"; }
          }
        }
        info = "RDS=select";
        char psbuf[32];
        if (snp.phase_set != 0) snprintf(psbuf, sizeof psbuf, "%u", snp.phase_set); else snprintf(psbuf, sizeof psbuf, ".");
        if (!strcmp(gt, "0/0") || !strcmp(gt, "0/1") || !strcmp(gt, "1/1") || !strcmp(gt, "0|1") || !strcmp(gt, "1|0"))
          snprintf(gtbuf, sizeof gtbuf, "%s:%d:%s:%u:%.2f:%.2f", gt, as_i32(snp.genotype_quality), psbuf, snp.depth, af[0], snp.phase_score);
        else
          snprintf(gtbuf, sizeof gtbuf, "%s:%d:%s:%u:%.2f,%.2f:%.2f", gt, as_i32(snp.genotype_quality), psbuf, snp.depth, af[0], af[1], snp.phase_score);
        format = "GT:GQ:PS:DP:AF:PQ";
      }
      /* writer: records with 0 ALT alleles are silently skipped (thread.rs:266-303) */
      if (alt.size() == 1)
        snprintf(line, sizeof line, "%s\t%lld\t.\t%c\t%c\t%d\t%s\t%s\t%s\t%s\n", chrom, (long long)snp.pos + 1, snp.reference, alt[0], as_i32(snp.variant_quality), filter, info, format, gtbuf);
      else if (alt.size() == 2)
        snprintf(line, sizeof line, "%s\t%lld\t.\t%c\t%c,%c\t%d\t%s\t%s\t%s\t%s\n", chrom, (long long)snp.pos + 1, snp.reference, alt[0], alt[1], as_i32(snp.variant_quality), filter, info, format, gtbuf);
      else continue;
      out += line;
    }
    return out;
  }
};

/* ================================================================================== */
extern "C" {

orc_region* orc_region_create(const lcr_reads* reads, int32_t read_begin, int32_t read_end, int64_t start0,
                              int32_t len, const uint8_t* ref_window, const lcr_params* params) {
  orc_region* r = new orc_region();
  r->reads = reads; r->rb = read_begin; r->re = read_end; r->start0 = start0; r->len = len; r->ref = ref_window;
  r->prm = *params;
  r->seed = orc_region_seed(params->seed, start0);
  return r;
}
void orc_region_destroy(orc_region* r) { delete r; }
void orc_pileup(orc_region* r) { r->pileup(); }
void orc_candidates(orc_region* r) { r->candidates(); }
void orc_fragments(orc_region* r) { r->fragments(); }
void orc_phase(orc_region* r, int mode) { r->phase(mode); }
void orc_post_phase(orc_region* r) { r->post_phase(); }
void orc_set_fast(orc_region* r, int n_threads) { r->fast_threads = n_threads < 0 ? 0 : n_threads; }
void orc_set_tie_mask(orc_region* r, int mask) { r->tie_mask = mask; }
void orc_get_tie_census(const orc_region* r, int64_t* out10) { for (int i = 0; i < 10; i++) out10[i] = r->census[i]; }

void orc_get_planes(const orc_region* r, uint32_t* out) {
  const int64_t L = r->len;
  for (int64_t i = 0; i < L; i++) {
    const BaseFreq& bf = r->freq[i];
    out[LCR_PL_A * L + i] = bf.a; out[LCR_PL_C * L + i] = bf.c; out[LCR_PL_G * L + i] = bf.g; out[LCR_PL_T * L + i] = bf.t;
    out[LCR_PL_N * L + i] = bf.n; out[LCR_PL_D * L + i] = bf.d; out[LCR_PL_NI * L + i] = bf.ni;
    for (int b = 0; b < 4; b++) out[(LCR_PL_FWD_A + b) * L + i] = (uint32_t)bf.base_strands[b][0];
    out[LCR_PL_TS_FWD * L + i] = (uint32_t)bf.transcript_strands[0];
    out[LCR_PL_TS_REV * L + i] = (uint32_t)bf.transcript_strands[1];
  }
}
int32_t orc_get_baseq(const orc_region* r, int32_t col, int allele, uint8_t* out, int32_t cap) {
  const auto& v = r->freq[col].baseq[allele];
  int32_t n = (int32_t)v.size();
  for (int32_t i = 0; i < n && i < cap; i++) out[i] = v[i];
  return n;
}
int32_t orc_n_cand(const orc_region* r) { return (int32_t)r->cands.size(); }
void orc_get_cands(const orc_region* r, lcr_candidate* out) {
  for (size_t i = 0; i < r->cands.size(); i++) {
    const Cand& c = r->cands[i];
    lcr_candidate& o = out[i];
    memset(&o, 0, sizeof o);
    o.pos = c.pos; o.region = 0; o.ref_base = (uint8_t)c.reference; o.allele1 = (uint8_t)c.alleles[0]; o.allele2 = (uint8_t)c.alleles[1];
    o.n_alt = (uint8_t)c.n_alt; o.cnt1 = c.allele_cnt[0]; o.cnt2 = c.allele_cnt[1]; o.depth = c.depth;
    o.af1 = c.allele_freqs[0]; o.af2 = c.allele_freqs[1];
    o.variant_type = c.variant_type; o.genotype = c.genotype; o.haplotype = c.haplotype;
    o.flags = (c.rna_editing ? LCR_F_RNA_EDIT : 0) | (c.dense ? LCR_F_DENSE : 0) | (c.het_var ? LCR_F_HET : 0) |
              (c.for_phasing ? LCR_F_FOR_PHASING : 0) | (c.hom_var ? LCR_F_HOM : 0) | (c.single ? LCR_F_SINGLE : 0) |
              (c.non_selected ? LCR_F_NON_SELECTED : 0) | (c.cand_somatic ? LCR_F_CAND_SOMATIC : 0);
    o.phase_set = c.phase_set;
    for (int k = 0; k < 3; k++) { o.loglik[k] = c.loglik[k]; o.gt_prob[k] = c.genotype_probability[k]; }
    o.qual = c.variant_quality; o.gq = c.genotype_quality; o.phase_score = c.phase_score;
  }
}
void orc_cand_gt_hist(const orc_region* r, int32_t col, double* out8) {
  /* order-free evaluation of candidate.rs:267-282: integer histogram hist[match?][q] x 31-entry
   * f64 LUT, summed q ascending (match term, then mismatch term).  cnt==0 terms are skipped so
   * that q=0 (log10(1-1) = -inf) never yields 0 * -inf. */
  const BaseFreq& bf = r->freq[col];
  int ri = base_index(bf.ref_base);
  uint64_t hm[31] = {0}, hx[31] = {0};
  for (int b = 0; b < 4; b++) for (uint8_t q : bf.baseq[b]) { if (b == ri) hm[q]++; else hx[q]++; }
  double l0 = 0.0, l2 = 0.0;
  for (int q = 0; q <= 30; q++) {
    double e = std::pow(0.1, (double)q / 10.0);
    double le = std::log10(e), l1e = std::log10(1.0 - e);
    if (hm[q]) { l0 += (double)hm[q] * le; l2 += (double)hm[q] * l1e; }
    if (hx[q]) { l0 += (double)hx[q] * l1e; l2 += (double)hx[q] * le; }
  }
  double loglik[3] = {l0, 0.0, l2};
  loglik[1] -= (double)(bf.a + bf.c + bf.g + bf.t) * std::log10(2.0);
  double gprob[3], vq, gq;
  orc_region::gt_tail(loglik, gprob, &vq, &gq);
  out8[0] = loglik[0]; out8[1] = loglik[1]; out8[2] = loglik[2];
  out8[3] = gprob[0]; out8[4] = gprob[1]; out8[5] = gprob[2]; out8[6] = vq; out8[7] = gq;
}
int32_t orc_n_rows(const orc_region* r) { return (int32_t)r->frags.size(); }
int64_t orc_nnz(const orc_region* r) { int64_t n = 0; for (auto& f : r->frags) n += (int64_t)f.list.size(); return n; }
void orc_get_fragmat(const orc_region* r, int64_t* row_ptr, int32_t* row_read, int32_t* col, uint8_t* val,
                     uint8_t* row_for_phasing, uint32_t* row_links) {
  int64_t p = 0;
  for (size_t k = 0; k < r->frags.size(); k++) {
    const Fragment& f = r->frags[k];
    row_ptr[k] = p; row_read[k] = f.read; row_for_phasing[k] = f.for_phasing ? 1 : 0; row_links[k] = f.num_hete_links;
    for (const FragElem& fe : f.list) {
      col[p] = fe.snp_idx;
      val[p] = (uint8_t)(fe.baseq | (fe.p == 1 ? 32 : 0) | (base_index(fe.base) << 6));
      p++;
    }
  }
  row_ptr[r->frags.size()] = p;
}
int32_t orc_get_ld_blocks(const orc_region* r, int32_t* off, int32_t* members, int32_t cap) {
  int32_t p = 0;
  for (size_t b = 0; b < r->ld_blocks.size(); b++) {
    off[b] = p;
    for (int m : r->ld_blocks[b]) { if (p < cap) members[p] = m; p++; }
  }
  off[r->ld_blocks.size()] = p;
  return (int32_t)r->ld_blocks.size();
}
void orc_get_phase(const orc_region* r, int8_t* haplotag, uint8_t* assignment, uint32_t* phase_set, double* objective) {
  for (size_t k = 0; k < r->frags.size(); k++) {
    haplotag[k] = (int8_t)r->frags[k].haplotag; assignment[k] = (uint8_t)r->frags[k].assignment;
    auto f = r->read_phase_set.find((int)k);
    phase_set[k] = f == r->read_phase_set.end() ? 0 : f->second;
  }
  *objective = r->best_objective;
}
/* which half-rounds of the perturbation loop (phase.rs:1198-1233) improved the best objective: call before orc_phase; returns
 * the number of half-rounds logged so far, copies min(n, cap) flags */
int32_t orc_round_log(orc_region* r, uint8_t* out, int32_t cap) {
  if (!r->round_log) { r->round_log = new std::vector<uint8_t>(); return 0; }
  const int32_t n = (int32_t)r->round_log->size();
  for (int32_t i = 0; i < n && i < cap; i++) out[i] = (*r->round_log)[i];
  return n;
}
void orc_get_stats(const orc_region* r, int64_t* out4) { for (int i = 0; i < 4; i++) out4[i] = r->stats[i]; }
int64_t orc_vcf_text(orc_region* r, const char* chrom, char* buf, int64_t cap) {
  std::string s = r->vcf_text(chrom);
  if ((int64_t)s.size() < cap) memcpy(buf, s.c_str(), s.size() + 1);
  return (int64_t)s.size();
}

/* ---- batch runner: regions pulled from an atomic counter by n_threads native threads (thread.rs:77) ---- */
struct orc_batch {
  std::vector<orc_region*> regs;
  std::vector<std::vector<uint32_t>> planes;   /* per region: LCR_NPLANES x len */
  struct Snap { std::vector<int64_t> row_ptr; std::vector<int32_t> row_read, col; std::vector<uint8_t> val, fp; std::vector<uint32_t> links; };
  std::vector<Snap> fm;
  std::vector<int64_t> col_off;
  int upto = 0, threads = 1;
  double seconds = 0.0;
  ~orc_batch() { for (auto* r : regs) delete r; }
};

orc_batch* orc_run_batch_opts(const lcr_reads* reads, const lcr_regions* rg, const lcr_params* params, int mode, int n_threads,
                              int upto, int keep_planes, int fast_threads, int tie_mask);
orc_batch* orc_run_batch(const lcr_reads* reads, const lcr_regions* rg, const lcr_params* params, int mode, int n_threads,
                         int upto, int keep_planes) {
  return orc_run_batch_opts(reads, rg, params, mode, n_threads, upto, keep_planes, 0, 15);
}
orc_batch* orc_run_batch_opts(const lcr_reads* reads, const lcr_regions* rg, const lcr_params* params, int mode, int n_threads,
                              int upto, int keep_planes, int fast_threads, int tie_mask) {
  orc_batch* B = new orc_batch();
  const int ng = rg->n_regions;
  B->regs.assign(ng, nullptr); B->planes.resize(ng); B->fm.resize(ng); B->upto = upto;
  B->col_off.assign(rg->col_off, rg->col_off + ng + 1);
  if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
  n_threads = std::max(1, std::min(n_threads, std::max(ng, 1)));
  B->threads = n_threads;
  std::atomic<int> next(0);
  /* heaviest regions first (reads x length), like a work-stealing pool would end up balancing them */
  std::vector<int> order(ng);
  for (int g = 0; g < ng; g++) order[g] = g;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    return (int64_t)(rg->read_begin[a + 1] - rg->read_begin[a]) * rg->len[a] > (int64_t)(rg->read_begin[b + 1] - rg->read_begin[b]) * rg->len[b];
  });
  auto work = [&]() {
    for (;;) {
      const int k = next.fetch_add(1);
      if (k >= ng) return;
      const int g = order[k];
      orc_region* r = orc_region_create(reads, rg->read_begin[g], rg->read_begin[g + 1], rg->start0[g], rg->len[g], rg->ref + rg->col_off[g], params);
      B->regs[g] = r;
      r->fast_threads = fast_threads < 0 ? 0 : fast_threads; r->tie_mask = tie_mask;
      r->pileup();
      if (keep_planes) { B->planes[g].assign((size_t)LCR_NPLANES * (size_t)r->len, 0u); orc_get_planes(r, B->planes[g].data()); }
      if (upto >= 1) r->candidates();
      if (upto >= 1 || !keep_planes) { std::vector<BaseFreq>().swap(r->freq); }
      if (upto >= 2) {
        r->fragments();
        orc_batch::Snap& s = B->fm[g];
        const size_t n = r->frags.size(); const int64_t nnz = orc_nnz(r);
        s.row_ptr.resize(n + 1); s.row_read.resize(n); s.col.resize(nnz); s.val.resize(nnz); s.fp.resize(n); s.links.resize(n);
        orc_get_fragmat(r, s.row_ptr.data(), s.row_read.data(), s.col.data(), s.val.data(), s.fp.data(), s.links.data());
      }
      if (upto >= 3) { r->phase(mode); r->post_phase(); }
    }
  };
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; t++) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  B->seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return B;
}
void orc_batch_destroy(orc_batch* B) { delete B; }
double orc_batch_seconds(const orc_batch* B) { return B->seconds; }
int32_t orc_batch_threads(const orc_batch* B) { return B->threads; }
void orc_batch_offsets(const orc_batch* B, int32_t* cand_off, int32_t* row_off, int64_t* nnz_off) {
  cand_off[0] = 0; row_off[0] = 0; nnz_off[0] = 0;
  for (size_t g = 0; g < B->regs.size(); g++) {
    cand_off[g + 1] = cand_off[g] + (int32_t)B->regs[g]->cands.size();
    row_off[g + 1] = row_off[g] + (int32_t)B->fm[g].row_read.size();
    nnz_off[g + 1] = nnz_off[g] + (int64_t)B->fm[g].col.size();
  }
}
void orc_batch_planes(const orc_batch* B, uint32_t* out) {
  const int64_t n_cols = B->col_off.back();
  for (size_t g = 0; g < B->regs.size(); g++) {
    const int64_t L = B->regs[g]->len, o = B->col_off[g];
    if (B->planes[g].empty()) continue;
    for (int k = 0; k < LCR_NPLANES; k++) memcpy(out + (int64_t)k * n_cols + o, B->planes[g].data() + (int64_t)k * L, (size_t)L * 4);
  }
}
void orc_batch_cands(const orc_batch* B, lcr_candidate* out) {
  for (size_t g = 0; g < B->regs.size(); g++) {
    orc_get_cands(B->regs[g], out);
    for (size_t i = 0; i < B->regs[g]->cands.size(); i++) out[i].region = (int32_t)g;
    out += B->regs[g]->cands.size();
  }
}
void orc_batch_fragmat(const orc_batch* B, int64_t* row_ptr, int32_t* row_read, int32_t* col, uint8_t* val, uint8_t* fp, uint32_t* links) {
  int64_t e = 0; size_t r = 0;
  for (size_t g = 0; g < B->regs.size(); g++) {
    const orc_batch::Snap& s = B->fm[g];
    const size_t n = s.row_read.size();
    for (size_t k = 0; k < n; k++) { row_ptr[r + k] = e + s.row_ptr[k]; row_read[r + k] = s.row_read[k]; fp[r + k] = s.fp[k]; links[r + k] = s.links[k]; }
    if (!s.col.empty()) { memcpy(col + e, s.col.data(), s.col.size() * 4); memcpy(val + e, s.val.data(), s.val.size()); }
    e += (int64_t)s.col.size(); r += n;
  }
  row_ptr[r] = e;
}
void orc_batch_phase(const orc_batch* B, int8_t* haplotag, uint8_t* assignment, uint32_t* phase_set, double* objective) {
  size_t r = 0;
  for (size_t g = 0; g < B->regs.size(); g++) {
    orc_get_phase(B->regs[g], haplotag + r, assignment + r, phase_set + r, objective + g);
    r += B->regs[g]->frags.size();
  }
}
void orc_batch_tie_census(const orc_batch* B, int64_t* out) { for (size_t g = 0; g < B->regs.size(); g++) orc_get_tie_census(B->regs[g], out + 10 * g); }
void orc_batch_stats(const orc_batch* B, int64_t* out) { for (size_t g = 0; g < B->regs.size(); g++) orc_get_stats(B->regs[g], out + 4 * g); }
int64_t orc_batch_vcf(orc_batch* B, const char* chrom, char* buf, int64_t cap, int64_t* off) {
  int64_t n = 0;
  for (size_t g = 0; g < B->regs.size(); g++) {
    off[g] = n;
    const std::string s = B->regs[g]->vcf_text(chrom);
    if (n + (int64_t)s.size() < cap) memcpy(buf + n, s.data(), s.size());
    n += (int64_t)s.size();
  }
  off[B->regs.size()] = n;
  return n;
}
orc_region* orc_batch_region(orc_batch* B, int32_t g) { return B->regs[g]; }

float orc_strand_odds_ratio(int a, int b, int c, int d) { return cal_strand_odds_ratio(a, b, c, d); }
double orc_binomial_two_tailed(uint64_t s, uint64_t t) { return binomial_two_tailed(s, t); }
void orc_two_major_alleles(const uint32_t cnt[4], uint8_t ref_base, uint8_t* a1, uint32_t* c1, uint8_t* a2, uint32_t* c2) {
  char x1, x2;
  get_two_major_alleles(cnt, (char)ref_base, &x1, c1, &x2, c2);
  *a1 = (uint8_t)x1; *a2 = (uint8_t)x2;
}
double orc_aki(int sigma, int delta, int eta, int p, double err) { return aki(sigma, delta, eta, p, err); }
double orc_cal_sigma_delta_eta_log(int sigma_k, int n, const int* delta, const int* eta, const int* ps, const double* probs) {
  return cal_sigma_delta_eta_log(sigma_k, std::vector<int>(delta, delta + n), std::vector<int>(eta, eta + n), std::vector<int>(ps, ps + n), std::vector<double>(probs, probs + n));
}
double orc_cal_delta_eta_sigma_log(int delta_i, int eta_i, int n, const int* sigma, const int* ps, const double* probs) {
  return cal_delta_eta_sigma_log(delta_i, eta_i, std::vector<int>(sigma, sigma + n), std::vector<int>(ps, ps + n), std::vector<double>(probs, probs + n));
}
double orc_cal_phase_score_log(int delta_i, int eta_i, int n, const int* sigma, const int* ps, const double* probs) {
  return cal_phase_score_log(delta_i, eta_i, std::vector<int>(sigma, sigma + n), std::vector<int>(ps, ps + n), std::vector<double>(probs, probs + n));
}

} /* extern "C" */
