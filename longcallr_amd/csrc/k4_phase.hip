// k4_phase.hip — K4: haplotype phasing on gfx950, HOST CONTROL (buffer sizing, launches, the LD-block flip pass, host epilogue).
//
// Replaces SNPFrag::phase (reference src/phase.rs:1087-1296) with its kernel cross_optimize
// (phase.rs:810-976) and the probability functions phase.rs:32-49,77-96,128-176,257-276, plus the
// post-phase sequence of src/thread.rs:168-201 (snpfrags.rs:191-733).
//
// Kernel units (in launch order; one queue stages + enumerates + post-processes, a second queue carries the
// few chain regions, see PhaseHost::run); k4_kernels.h / k4_grid.h hold their launchers and LDS layouts:
//   k4_stage.hip  k4_stage      phase matrices (CSR + CSC, per-SNP constants) of every region from K3's fragment CSR
//   k4_enum.hip   k4_enum_reg   S <= max_enum_snps: all 2^S enumeration restarts (phase.rs:1097-1122), one wave64 per
//                               restart with the matrix in registers (<0>: streamed from LDS); every restart leaves its
//                               objective, final state and signature
//                 k4_enum_resolve  `prob > largest_prob` (phase.rs:1113-1119) per region: the maximal objective, and among
//                               the configurations that have it the reference's f64 sums (phase.rs:257-276) decide
//                 k4_enum_big + k4_enum_pick  fallback on global memory (matrix beyond the LDS budget): first maximum,
//                               winner re-run; its ties are counted as unresolved
//   k4_grid.hip   k4_chain_wg / k4_chain_grid   S > max_enum_snps: the sequential chain (phase.rs:1123-1233) with one
//                               workgroup per region, or all CUs on one large region (LD-block flip pass, perturbation rounds)
//   k4_post.hip   k4_post       post-phase assignment, rescue and phase sets (f64, reference observation order)
// cross_optimize alternates sigma / delta-eta Jacobi steps until neither improves (<= 21 iterations).
// Its decision arithmetic is exact: every emission term log10(eps_q) / log10(1-eps_q) comes from a
// 31-entry table in fixed point (scale 2^40, int64), so sums are order-free and every comparison
// the reference makes on f64 ratio scores (q < qn, argmax q1..q4, prob > largest_prob) becomes an
// integer comparison of the log sums (the ratios 1 - A/D share a negative denominator D).  Where such a comparison is an
// exact TIE the reference's outcome is the rounding noise of its reference-order f64 sums: those sums are formed there
// (sigma ties per row, `prob > largest_prob` per configuration; PhaseDebug::tie_arith, lcr_get_tie_census).  See
// DESIGN.md "Decision arithmetic".
// rand::thread_rng() is replaced by a counter-based generator evaluated at the draw index the
// reference's call order implies, so restarts can run in parallel.
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <climits>
#include <map>
#include <set>
#include <atomic>
#include <functional>
#include <thread>
#include <mutex>
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>
#include <sys/stat.h>
#include <cerrno>

#include "k4_dev.h"
#include "k4_grid.h"
#include "k4_post.h"
#include "k4_kernels.h"

namespace {

// ================================= host side ====================================================

struct HostLut {
  double le[31], l1e[31];  // log10(eps), log10(1-eps), eps = 10^(-q/10) (fragment.rs:132); q=0 treated as q=1
  double p_homref, p_homvar, log_theta, log2;
  PhaseLutDev dev;
  HostLut() {
    for (int q = 0; q <= 30; q++) {
      const int qq = q == 0 ? 1 : q;
      const double eps = std::pow(10.0, -(double)qq / 10.0);
      le[q] = std::log10(eps); l1e[q] = std::log10(1.0 - eps);
      dev.fe[q] = std::llround(le[q] * FX_SCALE); dev.f1e[q] = std::llround(l1e[q] * FX_SCALE);
    }
    p_homref = std::log10(1.0 - 1.5 * 0.001); p_homvar = std::log10(0.5 * 0.001);
    log_theta = std::log10(0.001); log2 = std::log10(2.0);
    dev.f_homref = std::llround(p_homref * FX_SCALE); dev.f_homvar = std::llround(p_homvar * FX_SCALE);
    dev.f_het0 = std::llround(log_theta * FX_SCALE); dev.f_log2 = std::llround(log2 * FX_SCALE);
  }
};
const HostLut& hlut() { static HostLut l; return l; }

struct Obs { int sigma; uint8_t v; };  // one (read haplotag, entry value) observation of a SNP column

inline double lg(int sigma, int delta, int eta, uint8_t v) {  // log10(aki(...)), phase.rs:32-49
  const int p = (v & 32) ? 1 : -1, x = eta == 0 ? sigma * delta : eta;
  return p == x ? hlut().l1e[v & 31] : hlut().le[v & 31];
}
// phase.rs:128-176.  The five log sums of one call are running sums over the same observation order;
// the sum for eta != 0 does not depend on delta, and log_q1 repeats one of the other four, so the four
// distinct sums are computed once (identical addition sequences => identical doubles) and each of
// the reference's calls is assembled from them.
struct ColScores {
  double het_d = 0, het_nd = 0, homref = 0, homvar = 0;  // sum log10 aki for (delta,0), (-delta,0), (.,+1), (.,-1)
  double p_het = 0;
  ColScores(int delta_i, const std::vector<Obs>& o) {
    p_het = o.empty() ? hlut().log_theta : hlut().log_theta - (double)(uint32_t)o.size() * hlut().log2;
    for (const Obs& x : o) {
      het_d += lg(x.sigma, delta_i, 0, x.v); het_nd += lg(x.sigma, -delta_i, 0, x.v);
      homref += lg(x.sigma, delta_i, 1, x.v); homvar += lg(x.sigma, delta_i, -1, x.v);
    }
  }
  // cal_delta_eta_sigma_log(sign * delta_i, eta_i, ...), sign = +1 / -1
  double score(int sign, int eta_i) const {
    const double hd = sign > 0 ? het_d : het_nd, hn = sign > 0 ? het_nd : het_d;
    double q1 = eta_i == 0 ? hd : (eta_i == 1 ? homref : homvar);
    q1 += eta_i == 0 ? p_het : (eta_i == 1 ? hlut().p_homref : hlut().p_homvar);
    const double q2 = homvar + hlut().p_homvar, q3 = hd + p_het, q4 = homref + hlut().p_homref, q5 = hn + p_het;
    return 1.0 - q1 / (q2 + q3 + q4 + q5);
  }
};
double delta_eta_sigma_log(int delta_i, int eta_i, const std::vector<Obs>& o) { return ColScores(delta_i, o).score(1, eta_i); }
// phase.rs:238-255
double phase_score_log(int delta_i, int eta_i, const std::vector<Obs>& o) {
  double q1 = 0, q2 = 0, q3 = 0;
  for (const Obs& x : o) q1 += lg(x.sigma, delta_i, eta_i, x.v);
  for (const Obs& x : o) { q2 += lg(x.sigma, 1, eta_i, x.v); q3 += lg(x.sigma, -1, eta_i, x.v); }
  return 1.0 - q1 / (q2 + q3);
}

// host view of one region: full fragment rows (all entries) + mutable phasing state
struct RegionHost {
  int g = 0, S = 0, nrow = 0;
  int c0 = 0;                      // first candidate (global index)
  int r0 = 0;                      // first row (global index)
  const int64_t* row_ptr = nullptr;  // global CSR (host copy)
  const int32_t* col = nullptr;
  const uint8_t* val = nullptr;
  const uint32_t* links = nullptr;
  lcr_candidate* cand = nullptr;   // cand[0..S)
  std::vector<uint8_t> phase_site; // per entry of this region (index e - row_ptr[r0])
  std::vector<int8_t> tag;         // haplotag per row
  std::vector<uint8_t> asg, fp;    // assignment, for_phasing per row
  std::vector<std::vector<int>> cover;  // per SNP: rows (local) in push order (fragment.rs:293-306)
  std::vector<int> fp_rows;        // local rows with for_phasing at K3 time (the phase matrix rows)
  uint64_t seed = 0, ctr = 0;
  uint32_t min_linkers = 1;
  int64_t e0 = 0;
  double rnd() { return u01(seed, ctr++); }
  int64_t eb(int r) const { return row_ptr[r0 + r]; }
  int64_t ee(int r) const { return row_ptr[r0 + r + 1]; }
  int lc(int64_t e) const { return col[e] - c0; }
  bool fphase(int i) const { return (cand[i].flags & LCR_F_FOR_PHASING) != 0; }

  // snpfrags.rs:548-625
  void assign_reads_haplotype(double cutoff) {
    for (int r = 0; r < nrow; r++) {
      if (!fp[r]) continue;
      const int sigma_k = tag[r];
      double q1 = 0, q2 = 0, q3 = 0, n1 = 0;
      int n = 0;
      for (int64_t e = eb(r); e < ee(r); e++) {
        const int i = lc(e);
        if (!phase_site[e - e0] && fphase(i)) phase_site[e - e0] = 1;
        if (!fphase(i) || cand[i].haplotype == 0 || cand[i].genotype != 0) continue;
        // cal_sigma_delta_eta_log (phase.rs:77-96) for sigma_k and -sigma_k share log_q2/log_q3
        q1 += lg(sigma_k, cand[i].haplotype, 0, val[e]);
        n1 += lg(-sigma_k, cand[i].haplotype, 0, val[e]);
        n++;
      }
      if (sigma_k == 0 || n == 0) { asg[r] = 0; tag[r] = 0; continue; }
      for (int64_t e = eb(r); e < ee(r); e++) {
        const int i = lc(e);
        if (!fphase(i) || cand[i].haplotype == 0 || cand[i].genotype != 0) continue;
        q2 += lg(1, cand[i].haplotype, 0, val[e]); q3 += lg(-1, cand[i].haplotype, 0, val[e]);
      }
      const double q = 1.0 - q1 / (q2 + q3), qn = 1.0 - n1 / (q2 + q3);
      if (std::fabs(q - qn) >= cutoff) {
        if (q >= qn) asg[r] = sigma_k == 1 ? 1 : 2;
        else if (sigma_k == 1) { asg[r] = 2; tag[r] = -1; }
        else { asg[r] = 1; tag[r] = 1; }
      } else { asg[r] = 0; tag[r] = 0; }
    }
  }

  void gather(int ti, bool need_assigned, bool het_skip_unassigned, std::vector<Obs>& o, int& hap1, int& hap2) const {
    o.clear(); hap1 = hap2 = 0;
    for (int r : cover[ti]) {
      if (!fp[r] || links[r0 + r] < min_linkers) continue;
      if (need_assigned && asg[r] == 0) continue;
      if (het_skip_unassigned && asg[r] == 0) continue;
      for (int64_t e = eb(r); e < ee(r); e++)
        if (lc(e) == ti) {
          if (asg[r] == 1) hap1++; else if (asg[r] == 2) hap2++;
          o.push_back({tag[r], val[e]});
        }
    }
  }

  // snpfrags.rs:378-546
  void assign_snp_haplotype_genotype() {
    std::vector<Obs> o;
    for (int ti = 0; ti < S; ti++) {
      lcr_candidate& snp = cand[ti];
      if (!fphase(ti)) { snp.flags |= LCR_F_NON_SELECTED; continue; }
      if (cover[ti].empty()) { snp.flags |= LCR_F_SINGLE; continue; }
      const int delta_i = snp.haplotype;
      int hap1, hap2;
      gather(ti, false, snp.variant_type == 1, o, hap1, hap2);
      if (o.empty()) { snp.flags |= LCR_F_NON_SELECTED; continue; }
      const ColScores cs(delta_i, o);
      const double q1 = cs.score(1, 0), q2 = cs.score(-1, 0), q3 = cs.score(1, 1), q4 = cs.score(1, -1);
      const double mx = std::fmax(q1, std::fmax(q2, std::fmax(q3, q4)));
      if (q1 == mx) { snp.haplotype = delta_i; snp.genotype = 0; snp.variant_type = 1; }
      else if (q2 == mx) { snp.haplotype = -delta_i; snp.genotype = 0; snp.variant_type = 1; }
      else if (q3 == mx) { snp.haplotype = delta_i; snp.genotype = 1; snp.variant_type = 0; }
      else if (q4 == mx) { snp.haplotype = delta_i; snp.genotype = -1; if (snp.variant_type != 2 && snp.variant_type != 3) snp.variant_type = 2; }
      else continue;  // NaN scores: the reference panics here
      if (snp.genotype != 0) { snp.flags |= LCR_F_NON_SELECTED; continue; }
      if (hap1 >= 1 && hap2 >= 1) snp.phase_score = -10.0 * std::log10(1.0 - phase_score_log(snp.haplotype, snp.genotype, o));
      else snp.phase_score = 0.19940219;
    }
  }

  // snpfrags.rs:191-376 (eval_rna_edit_var_phase / eval_low_frac_var_phase)
  void eval_rescue(uint32_t list_flag, float min_phase_score, bool low_frac) {
    std::vector<int> list;  // edit_snps / somatic_snps are fixed at candidate time (snpfrags.rs:20-26)
    for (int i = 0; i < S; i++) if (orig_flags[i] & list_flag) list.push_back(i);
    std::vector<Obs> o;
    for (int ti : list) {
      lcr_candidate& snp = cand[ti];
      if (cover[ti].empty()) { snp.flags |= LCR_F_SINGLE; continue; }
      if (snp.variant_type != 1) { snp.flags |= LCR_F_NON_SELECTED; continue; }
      int hap1, hap2;
      gather(ti, true, false, o, hap1, hap2);
      if (o.empty() || hap1 < 2 || hap2 < 2) { snp.flags |= LCR_F_SINGLE; continue; }
      const double ps1 = -10.0 * std::log10(1.0 - phase_score_log(1, 0, o));
      const double ps2 = -10.0 * std::log10(1.0 - phase_score_log(-1, 0, o));
      snp.flags &= ~(uint32_t)LCR_F_SINGLE;
      if (std::fmax(ps1, ps2) >= (double)min_phase_score) {
        snp.flags &= ~(uint32_t)(LCR_F_NON_SELECTED | LCR_F_RNA_EDIT);
        if (low_frac) snp.flags &= ~(uint32_t)LCR_F_CAND_SOMATIC;
        snp.flags |= LCR_F_FOR_PHASING;
        for (int r : cover[ti]) {
          fp[r] = 1;
          if (tag[r] == 0 || asg[r] == 0) tag[r] = rnd() < 0.5 ? -1 : 1;
        }
        snp.haplotype = ps1 >= ps2 ? 1 : -1;
        snp.genotype = 0; snp.variant_type = 1; snp.phase_score = std::fmax(ps1, ps2);
      } else {
        snp.flags |= LCR_F_NON_SELECTED;
        if (low_frac) { snp.flags |= LCR_F_CAND_SOMATIC; snp.flags &= ~(uint32_t)LCR_F_FOR_PHASING; }
        else snp.flags |= LCR_F_RNA_EDIT;
      }
    }
  }
  std::vector<uint32_t> orig_flags;

  // snpfrags.rs:628-733
  // snpfrags.rs:628-733.  The reference builds a petgraph GraphMap whose nodes are the PASS het SNPs
  // (added in index order), adds an edge per read and allele-consistent SNP pair, and walks
  // kosaraju_scc: components come out in descending order of their first-inserted (= smallest-index)
  // node, a component's phase set is pos+1 of that node, and a read takes the phase set of the first
  // component in that order that owns one of its edges.  Union-find gives exactly that.
  void assign_phase_set(float min_phase_score, uint32_t* row_ps /* global rows */) {
    std::vector<int> parent(S, -1);  // -1: not a node
    for (int i = 0; i < S; i++) {
      const lcr_candidate& s = cand[i];
      if (s.genotype != 0 || s.variant_type != 1) continue;
      if (s.flags & (LCR_F_DENSE | LCR_F_RNA_EDIT)) continue;
      if (s.phase_score < (double)min_phase_score) continue;
      parent[i] = i;
    }
    auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
    auto unite = [&](int x, int y) { x = find(x); y = find(y); if (x != y) { if (x < y) parent[y] = x; else parent[x] = y; } };  // root = min index
    int ns[64], np[64];
    auto row_nodes = [&](int r) {
      int n = 0;
      for (int64_t e = eb(r); e < ee(r) && n < 64; e++)
        if (parent[lc(e)] >= 0) { ns[n] = lc(e); np[n] = (val[e] & 32) ? 1 : -1; n++; }
      return n;
    };
    for (int r = 0; r < nrow; r++) {
      if (!fp[r] || asg[r] == 0) continue;
      const int n = row_nodes(r);
      for (int x = 0; x < n; x++)
        for (int y = x + 1; y < n; y++)
          if (cand[ns[x]].haplotype * cand[ns[y]].haplotype == np[x] * np[y]) unite(ns[x], ns[y]);
    }
    for (int i = 0; i < S; i++) if (parent[i] >= 0) cand[i].phase_set = (uint32_t)(cand[find(i)].pos + 1);
    for (int r = 0; r < nrow; r++) {
      if (!fp[r] || asg[r] == 0) continue;
      const int n = row_nodes(r);
      int best = -1;  // largest component root among the components that own an edge of this read
      if (n == 1) best = find(ns[0]);  // self loop (snpfrags.rs:659-665)
      for (int x = 0; x < n; x++)
        for (int y = x + 1; y < n; y++)
          if (cand[ns[x]].haplotype * cand[ns[y]].haplotype == np[x] * np[y]) best = std::max(best, find(ns[x]));
      if (best >= 0) row_ps[r0 + r] = (uint32_t)(cand[best].pos + 1);
    }
  }

};

struct PhaseWork {   // host epilogue structures, reused across calls
  std::vector<RegionHost> R;
};

}  // namespace

// One persistent (grid-barrier) kernel at a time per device, across host threads and processes: two such launches
// that each hold a part of the CUs would wait for each other's workgroups forever.
// Persistent all-CU launches of different processes (or contexts) on one GPU would wait for each other's workgroups forever:
// they are serialised per DEVICE -- a process-local mutex keyed by the PCI bus id, and across processes an flock on
// <lock_dir>/grid_<bus id>.lock (lock_dir: lcr_ctx_set_lock_dir, created 0700; default the machine-wide /tmp/liblcr-locks, refused unless it
// is a directory of this user; the file is opened O_NOFOLLOW | O_CLOEXEC).  A lock that cannot be taken is an error, never
// silently skipped.  The destructor drains the queues it was given before it lets go (error paths return early).
struct GridLock {
  int fd = -1;
  bool held = false;
  std::mutex* dev_mu = nullptr;
  hipStream_t q[3] = {nullptr, nullptr, nullptr};
  int n_q = 0;
  static std::mutex& device_mutex(const std::string& bus) {
    static std::mutex table_mu;
    static std::map<std::string, std::mutex*> table;
    std::lock_guard<std::mutex> g(table_mu);
    auto it = table.find(bus);
    if (it == table.end()) it = table.emplace(bus, new std::mutex()).first;
    return *it->second;
  }
  // returns an error text, or "" when the lock is held
  std::string acquire(const std::string& lock_dir, hipStream_t a, hipStream_t b, hipStream_t c) {
    if (held) return "";
    q[0] = a; q[1] = b; q[2] = c; n_q = 3;
    int dev = 0;
    char bus[64] = "gpu";
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetPCIBusId(bus, sizeof(bus), dev);
    for (char* ch = bus; *ch; ch++) if (*ch == ':' || *ch == '/') *ch = '_';
    std::string dir = lock_dir;
    bool shared = dir.empty();
    //  a directory is trusted if it is the caller's or root's, and -- when others can write into it -- sticky: the owner
    // of a directory can unlink and replace the lock file, which would let two processes' persistent launches meet on the GPU, and a
    // foreign owner could hold the lock for ever.  The machine-wide default /tmp/liblcr-locks is used when it passes that test (this
    // user created it, or an administrator did); otherwise the lock is per user: /tmp/liblcr-<uid>.
    auto trusted = [&](const std::string& d) {
      struct stat st;
      if (lstat(d.c_str(), &st) != 0 || !S_ISDIR(st.st_mode)) return false;
      if (st.st_uid != getuid() && st.st_uid != 0) return false;
      return !(st.st_mode & 0022) || (st.st_mode & S_ISVTX) || st.st_uid == getuid();
    };
    if (shared) {
      dir = "/tmp/liblcr-locks";
      if (mkdir(dir.c_str(), 01777) == 0) (void)chmod(dir.c_str(), 01777);   // (the umask)
      //  an untrusted machine-wide directory is an error, not a silent per-user lock: whoever created it would go on
      // trusting it while everybody else locked /tmp/liblcr-<uid>, and two users' persistent launches on one GPU would no longer be
      // serialised.  The per-user lock is there for whoever asks for it: lcr_ctx_set_lock_dir(ctx, "/tmp/liblcr-<uid>").
      if (!trusted(dir)) return "the machine-wide lock directory " + dir + " exists but is not root's or this user's (or is writable by others without the sticky bit): "
                                "remove it, have an administrator create it 1777, or name a directory all processes that share this GPU see with lcr_ctx_set_lock_dir";
    }
    if (!shared && mkdir(dir.c_str(), 0700) != 0 && errno != EEXIST) return "cannot create lock directory " + dir + ": " + strerror(errno);
    if (!trusted(dir)) return "lock directory " + dir + " is not a directory of this user or of root (sticky if others may write): lcr_ctx_set_lock_dir names another one";
    const std::string path = dir + "/grid_" + bus + ".lock";
    dev_mu = &device_mutex(bus);
    dev_mu->lock();
    auto fail = [&](const std::string& e) { if (fd >= 0) { close(fd); fd = -1; } dev_mu->unlock(); dev_mu = nullptr; return e; };
    const auto t_start = std::chrono::steady_clock::now();
    for (;;) {
      fd = open(path.c_str(), O_CREAT | O_RDWR | O_NOFOLLOW | O_CLOEXEC, shared ? 0666 : 0600);
      if (fd < 0) return fail("cannot open " + path + ": " + strerror(errno));
      if (shared) { struct stat fs; if (fstat(fd, &fs) == 0 && fs.st_uid == getuid()) (void)fchmod(fd, 0666); }   // (created under a umask)
      // bounded wait (a process that died holding the lock has released it; one that hangs must not hang everybody else for ever)
      int rc;
      while ((rc = flock(fd, LOCK_EX | LOCK_NB)) != 0 && (errno == EWOULDBLOCK || errno == EINTR)) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count() > 600.0)
          return fail("the device lock " + path + " has been held by another process for 10 minutes");
        usleep(200);
      }
      if (rc != 0) return fail("flock(" + path + "): " + strerror(errno));
      // the file we hold must still be the one the path names (it may have been unlinked and recreated between open and flock)
      struct stat fa, fb;
      if (fstat(fd, &fa) == 0 && stat(path.c_str(), &fb) == 0 && fa.st_ino == fb.st_ino && fa.st_dev == fb.st_dev) break;
      (void)flock(fd, LOCK_UN); close(fd); fd = -1;
    }
    held = true;
    return "";
  }
  ~GridLock() {
    if (!held) return;
    for (int i = 0; i < n_q; i++) (void)hipStreamSynchronize(q[i]);   // (a persistent kernel may still be running on an error path; nullptr = the null stream, drained too)
    if (fd >= 0) { (void)flock(fd, LOCK_UN); close(fd); }
    if (dev_mu) dev_mu->unlock();
  }
};

void PhaseHost::free_work() { delete static_cast<PhaseWork*>(work); work = nullptr; }

int PhaseHost::ld_blocks(const PhaseInputs& in, int region, std::vector<int32_t>* off, std::vector<int32_t>* snps, hipStream_t s, std::string* err) {
  off->assign(1, 0); snps->clear();
  for (const ChainDesc& d : chain_desc) {
    if (d.slot != region) continue;
    const int c0 = in.cand_region_off[region];
    int32_t info[2] = {0, 0};
    auto chk = [&](hipError_t e) { if (e != hipSuccess && err) *err = hipGetErrorString(e); return e == hipSuccess; };
    if (!chk(hipMemcpyAsync(info, chain_dev.blk_info + 2 * region, 8, hipMemcpyDeviceToHost, s)) || !chk(hipStreamSynchronize(s))) return LCR_E_DEVICE;
    const int nb = info[0];
    if (nb == 0) return LCR_OK;
    std::vector<int32_t> ptr(nb + 1);
    if (!chk(hipMemcpyAsync(ptr.data(), chain_dev.blk_ptr + c0 + region, (size_t)(nb + 1) * 4, hipMemcpyDeviceToHost, s)) || !chk(hipStreamSynchronize(s))) return LCR_E_DEVICE;
    std::vector<int32_t> nodes(ptr[nb]);
    if (!chk(hipMemcpyAsync(nodes.data(), chain_dev.blk_nodes + c0, (size_t)ptr[nb] * 4, hipMemcpyDeviceToHost, s)) || !chk(hipStreamSynchronize(s))) return LCR_E_DEVICE;
    for (int b = nb - 1; b >= 0; b--) {   // kosaraju_scc emits the blocks in descending order of their smallest node
      snps->insert(snps->end(), nodes.begin() + ptr[b], nodes.begin() + ptr[b + 1]);
      off->push_back((int32_t)snps->size());
    }
    return LCR_OK;
  }
  return LCR_OK;   // not a chain region: no blocks are built (phase.rs:1097-1122 never looks at them)
}

#define PCHK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { if (err) *err = std::string(#expr) + ": " + hipGetErrorString(e_); return LCR_E_DEVICE; } } while (0)
// collects what the last run() left in flight: waits for its queues, fetches the tie census, copies the candidates' updated
// records and the objectives out of the pinned blocks k4_post wrote (the per-row results are read in place)
int PhaseHost::settle(std::string* err) {
  if (!pending) return LCR_OK;
  pending = false;
  PCHK(hipStreamSynchronize(side));
  PCHK(hipMemcpyAsync(h_pin[11].p, d_tie.p, TIE_NCTR * 8, hipMemcpyDeviceToHost, q_first));   // (`side` is drained: every kernel that counts is done or ahead in the first queue)
  PCHK(hipStreamSynchronize(q_first));
  PCHK(hipGetLastError());
  memcpy(tie_census, h_pin[11].p, TIE_NCTR * 8);
  uint8_t* const h_res = h_pin[7].as<uint8_t>();
  const long long* const h_obj = (const long long*)(h_pin[9].as<uint8_t>() + pend.hc_obj);
  std::vector<lcr_candidate>& cand = *pend.cand;
  for (int g = 0; g < pend.ng; g++) {
    const int c0 = pend.cand_off[g], S = pend.cand_off[g + 1] - c0;
    if (S == 0 || pend.host_post[g]) continue;
    memcpy(cand.data() + c0, h_pin[9].as<lcr_candidate>() + c0, (size_t)S * sizeof(lcr_candidate));
    objective[g] = (double)h_obj[g] / FX_SCALE;
  }
  r_haplotag = (int8_t*)(h_res + pend.res_tag); r_assignment = h_res + pend.res_asg; r_phase_set = (uint32_t*)(h_res + pend.res_ps);
  read_rec_stale = pend.any_host_post;   // (rows of regions that took the host epilogue: records rebuilt on demand)
  return LCR_OK;
}

int PhaseHost::run(const PhaseInputs& in, const lcr_params& prm, hipStream_t user_stream, std::string* err) {
  { const int rc = settle(err); if (rc) return rc; }   // (a previous call nobody asked about: its pinned blocks are rewritten below)
  HT("phase");
  // Two queues: `stream` stages the phase matrices and runs the enumeration regions (S <= max_enum_snps) with their
  // post-phase kernel; `side` runs the chain regions (S > max_enum_snps): LD blocks, LD-seeded start, block-flip pass
  // and perturbation rounds in ONE kernel per region class (k4_grid.hip: a workgroup per region, or all CUs on one
  // large region), then their post-phase kernel.  The host only sizes buffers and launches.
  if (prm.ld_weight_threshold != 1) {
    if (err) *err = "ld_weight_threshold must be 1: SNPFrag::phase is only ever called with 1 (thread.rs:166)";
    return LCR_E_ARG;
  }
  const int ng = in.n_regions, nrow = in.n_rows;
  const int64_t nnz = in.nnz;
  const int ncand = in.cand_region_off[ng];
  const bool prof = dbg.prof != 0;
  auto t_last = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) { if (!prof) return; auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[phase] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - t_last).count()); t_last = t; };
  std::vector<lcr_candidate>& cand = *in.cand;
  objective.assign(ng, 0.0);
  // async_phase (opt-in): everything below is queued on the stage's OWN first queue, behind what the caller's stream holds now (the
  // fragment stage's kernels), and the call returns without waiting.  Default: the caller's stream, results collected before the
  // call returns -- measured on C3 (profiles/r05_async_phase.txt): the next batch's pileup then time-shares the CUs with the
  // restarts (K0 214 -> 385 us, k4_enum_reg<32> 647 -> 1142 us), the resolve / post-phase tails stay exposed because the next
  // lcr_candidates has to wait for this stage anyway, and a fourth stream of the process pushes `side` and `aux` onto one hardware
  // queue: 2.19 -> 2.12 ms per step, and the pileup kernels' own durations (the roofline's measurement) grow by half.
  const bool async_mode = dbg.async_phase != 0;
  auto new_queue = [&](hipStream_t* q) -> hipError_t {
    if (!dbg.phase_prio) return hipStreamCreateWithFlags(q, hipStreamNonBlocking);
    int least = 0, greatest = 0;
    const hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(q, hipStreamNonBlocking, greatest);
  };
  if (async_mode && !main_q) PCHK(new_queue(&main_q));
  if (!ev_user) { PCHK(hipEventCreateWithFlags(&ev_user, hipEventDisableTiming)); PCHK(hipEventCreateWithFlags(&ev_gate[0], hipEventDisableTiming)); PCHK(hipEventCreateWithFlags(&ev_gate[1], hipEventDisableTiming)); }
  gate_set[0] = gate_set[1] = false;
  hipStream_t const stream = async_mode ? main_q : user_stream;
  // (round 6) the staging kernel and the fills in front of it stay on the CALLER's queue -- right behind the fragment stage's kernels, no
  // queue-to-queue hand-over in front of them (16 us on this platform) --, the host waits for the sizes there, and the stage's own queue
  // picks up behind an event that is long complete when the first restarts are queued
  hipStream_t const sq = user_stream;
  q_first = stream;
  if (!side) {
    PCHK(new_queue(&side));
    PCHK(hipEventCreateWithFlags(&ev_in, hipEventDisableTiming));
    PCHK(hipEventCreateWithFlags(&ev_csr, hipEventDisableTiming));
    PCHK(new_queue(&aux));
    PCHK(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
    PCHK(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  }
  const HostLut& L = hlut();

  // ---- device buffers
  DevBuf &b_reg = d_state[0], &b_prp = d_state[1], &b_pc = d_state[2], &b_pv = d_state[3], &b_cp = d_state[4],
         &b_cr = d_state[5], &b_cv = d_state[6], &b_snp = d_state[7], &b_st = d_state[8], &b_scr = d_state[9],
         &b_job = d_state[10], &b_obj = d_state[11], &b_sc = d_state[12], &b_stat = d_state[13], &b_cur = d_state[14],
         &b_stc = d_state[15], &b_slots = d_state[16], &b_psrc = d_state[21], &b_desc = d_state[22], &b_tbl = d_state[23],
         &b_adj = d_state[24], &b_part = d_state[25], &b_snpi = d_state[26], &b_snpb = d_state[27], &b_q = d_state[28],
         &b_info = d_state[29], &b_rowi = d_state[30], &b_enti = d_state[31], &b_work = d_state[32], &b_macc = d_state[33],
         &b_ctl = d_state[34], &b_terms = d_state[17], &b_qrow = d_state[18], &b_redo = d_state[19], &b_btot = d_state[35], &b_ps = d_state[36], &b_pse = d_state[37], &b_psp = d_state[38];
  const size_t nnz1 = (size_t)std::max<int64_t>(nnz, 1), nc1 = (size_t)std::max(ncand, 1), nr1 = (size_t)std::max(nrow, 1);
  PCHK(b_reg.reserve((size_t)std::max(ng, 1) * sizeof(RegionDev)));
  PCHK(b_stat.reserve((size_t)std::max(ng, 1) * sizeof(StageStat)));
  PCHK(b_prp.reserve((nr1 + ng + 1) * 4)); PCHK(b_pc.reserve(nnz1 * 4)); PCHK(b_pv.reserve(nnz1));
  PCHK(b_cp.reserve((nc1 + ng + 1) * 4)); PCHK(b_cr.reserve(nnz1 * 4)); PCHK(b_cv.reserve(nnz1));
  PCHK(b_snp.reserve(nc1 * 3 + 16)); PCHK(b_sc.reserve(nc1 * 4 * sizeof(long long))); PCHK(b_cur.reserve(nc1 * 4));
  PCHK(b_psrc.reserve(nr1 * 4));
  // state: sigma[n_rows] | delta[n_cand] | eta[n_cand] | obj[n_regions] (8-byte aligned); one copy per queue
  const size_t st_sig = 0, st_del = (nr1 + 15) & ~(size_t)15, st_eta = st_del + ((nc1 + 15) & ~(size_t)15);
  const size_t st_obj = st_eta + ((nc1 + 15) & ~(size_t)15);
  const size_t st_bytes = st_obj + (size_t)std::max(ng, 1) * 8;
  PCHK(b_st.reserve(st_bytes + 16)); PCHK(b_stc.reserve(st_bytes + 16));
  PCHK(h_pin[5].reserve((size_t)std::max(ng, 1) * sizeof(StageStat)));
  // results of the device epilogue live in pinned host memory that k4_post writes itself (every row belongs to a
  // region with candidates, fragment.rs:24-26, so every row is written): phase set u32 | haplotag | assignment, and
  // candidate mirror | objectives
  const size_t res_ps = 0, res_tag = nr1 * 4, res_asg = nr1 * 5, res_bytes = nr1 * 6;
  const size_t hc_obj = (nc1 * sizeof(lcr_candidate) + 15) & ~(size_t)15;
  PCHK(h_pin[7].reserve(res_bytes)); PCHK(h_pin[9].reserve(hc_obj + (size_t)std::max(ng, 1) * 8));
  PCHK(d_read_rec.reserve(nr1 * 12));   // per-row results once more, as records in HBM (lcr_get_read_records_device)
  uint8_t* d_res = nullptr; uint8_t* d_hc = nullptr;   // device-side addresses of the two pinned blocks
  PCHK(hipHostGetDevicePointer((void**)&d_res, h_pin[7].p, 0));
  PCHK(hipHostGetDevicePointer((void**)&d_hc, h_pin[9].p, 0));

  PhaseDev P{};
  P.reg = b_reg.as<RegionDev>();
  P.prow_ptr = b_prp.as<int32_t>(); P.pcol = b_pc.as<int32_t>(); P.pval = b_pv.as<uint8_t>();
  P.ccol_ptr = b_cp.as<int32_t>(); P.crow = b_cr.as<int32_t>(); P.cval = b_cv.as<uint8_t>();
  P.snp_const = b_sc.as<long long>();
  P.snp_fp = b_snp.as<uint8_t>(); P.snp_vt = b_snp.as<int8_t>() + nc1; P.snp_cons = b_snp.as<uint8_t>() + 2 * nc1;
  P.st_sigma = b_st.as<int8_t>() + st_sig; P.st_delta = b_st.as<int8_t>() + st_del; P.st_eta = b_st.as<int8_t>() + st_eta;
  P.st_obj = (long long*)(b_st.as<int8_t>() + st_obj);
  P.lut = L.dev;
  PostLut plut;
  for (int q = 0; q < 31; q++) { plut.le[q] = L.le[q]; plut.l1e[q] = L.l1e[q]; }
  plut.p_homref = L.p_homref; plut.p_homvar = L.p_homvar; plut.log_theta = L.log_theta; plut.log2 = L.log2;
  PCHK(d_lut64.reserve(sizeof(PostLut))); PCHK(d_tie.reserve(TIE_NCTR * 8)); PCHK(h_pin[11].reserve(TIE_NCTR * 8));
  if (!lut64_ready) { PCHK(hipMemcpyAsync(d_lut64.p, &plut, sizeof(PostLut), hipMemcpyHostToDevice, sq)); PCHK(hipStreamSynchronize(sq)); lut64_ready = true; }
  // the stage's fills in ONE launch in front of the staging kernel (the host's wait for the sizes ends with that kernel, and every command
  // between it and the first restart is idle GPU time): the tie census, the result state, the enumeration branch's best objective seen per
  // region (0x8080...: far below any objective) and the repair lists' counts (the pairs behind them are written before they are read)
  constexpr uint32_t REDO_CAP = 8192, REDO_GRID = 512;   // the repair pass of the enumeration branch (below)
  PCHK(d_rbest_buf.reserve((size_t)std::max(ng, 1) * 8 + 64));
  if (dbg.tie_arith >= 3) PCHK(b_redo.reserve(2 * (16 + 8 * (size_t)REDO_CAP) + 64));
  { void* const fp[4] = {d_tie.p, b_st.p, d_rbest_buf.p, b_redo.p};
    const int fv[4] = {0, 0, 0x80, 0};
    const size_t fs[4] = {TIE_NCTR * 8, ng ? st_bytes : 0, (size_t)ng * 8, dbg.tie_arith >= 3 ? 2 * (16 + 8 * (size_t)REDO_CAP) : 0};
    PCHK(lcr_fill_multi_async(4, fp, fv, fs, sq)); }
  P.lut64 = d_lut64.as<PostLut>(); P.tie_ctr = d_tie.as<unsigned long long>(); P.tie_arith = dbg.tie_arith;

  // ---- queue `stream`: stage the phase matrices -- launched before the host sorts the regions into their kernel classes (the
  // device would idle for that long); the per-region sizes arrive in pinned host memory
  const int64_t grid_min = dbg.grid_min >= 0 ? dbg.grid_min : (1 << 17);   // chain regions with at least this many phase entries get all CUs (tests: 0 = every region)
  StageStat* const stat = h_pin[5].as<StageStat>();
  StageIn si{in.d_row_ptr, in.d_col, in.d_val, in.d_row_links, in.d_cand, in.d_cand_off, in.d_row_region_off, in.d_start0,
             prm.min_linkers, prm.max_enum_snps, prm.seed, std::max<int64_t>(grid_min, 1)};
  StageOut so{};
  if (ng) {
    StageStat* d_stat = nullptr;   // the per-region sizes go straight into pinned host memory (no copy behind the staging kernels)
    PCHK(hipHostGetDevicePointer((void**)&d_stat, stat, 0));
    so = StageOut{b_reg.as<RegionDev>(), d_stat, b_prp.as<int32_t>(), b_pc.as<int32_t>(), b_pv.as<uint8_t>(),
                  b_cp.as<int32_t>(), b_cr.as<int32_t>(), b_cv.as<uint8_t>(), b_snp.as<uint8_t>(), b_snp.as<int8_t>() + nc1,
                  b_snp.as<uint8_t>() + 2 * nc1, b_sc.as<long long>(), b_cur.as<int32_t>(), b_psrc.as<int32_t>()};
    launch_k4_stage((int32_t)ng, sq, si, so, L.dev);
    PCHK(hipGetLastError());
  HT("  ph:stage_q");
  }

  // enumeration (S <= max_enum_snps) and chain regions
  std::vector<int32_t> enum_slots, chain_slots;
  for (int g = 0; g < ng; g++) {
    const int S = in.cand_region_off[g + 1] - in.cand_region_off[g];
    if (S == 0) continue;
    if ((uint32_t)S <= prm.max_enum_snps) enum_slots.push_back(g); else chain_slots.push_back(g);
  }
  // post-phase epilogue: k4_post (a workgroup per region, the region in LDS) for every region that fits its LDS image;
  // the host epilogue (RegionHost) for the others and, as a cross-check, for all regions under LCR_POST_HOST=1.  The
  // regions' sizes are on the host already (lcr_fragments), so this is known before anything is queued.
  const bool force_host_post = dbg.post_host != 0;
  std::vector<uint8_t> host_post(ng, 0), grid_post(ng, 0), grid_stage(ng, 0);
  bool any_host_post = false;
  uint32_t post_lds = 0;
  std::vector<int32_t> gpost_slots, gstage_slots;
  size_t gp_S = 1, gp_rows = 1, gp_E = 1;   // largest region image of k4_gpost
  for (int g = 0; g < ng; g++) {
    const int S = in.cand_region_off[g + 1] - in.cand_region_off[g];
    if (S == 0) continue;
    const int nr_g = in.row_region_off[g + 1] - in.row_region_off[g];
    const int64_t E_all = in.region_e_off[g + 1] - in.region_e_off[g];
    if (E_all > INT_MAX - 64) { if (err) *err = "a region's fragment matrix has more than 2^31 entries"; return LCR_E_ARG; }
    if (E_all >= std::max<int64_t>(grid_min, 1)) { grid_stage[g] = 1; gstage_slots.push_back(g); }
    bool fits = !(nr_g > POST_MAX_ROWS || E_all > POST_MAX_ENTRIES || S > POST_MAX_SNPS);
    uint32_t need = 0;
    if (fits) { need = post_layout(nr_g, (uint32_t)E_all, S).total; if (need > 64 * 1024) fits = false; }
    if (force_host_post) { host_post[g] = 1; any_host_post = true; }
    else if (!fits || (grid_min == 0 && (uint32_t)S > prm.max_enum_snps)) {
      grid_post[g] = 1; gpost_slots.push_back(g);
      gp_S = std::max(gp_S, (size_t)S); gp_rows = std::max(gp_rows, (size_t)nr_g); gp_E = std::max(gp_E, (size_t)E_all);
    } else post_lds = std::max(post_lds, need);
  }
  GridLock grid_lock;   // held from the first persistent launch until this call returns (all queues are drained by then)
#define GRID_LOCK() do { const std::string e_ = grid_lock.acquire(lock_dir, stream, side, aux); if (!e_.empty()) { if (err) *err = "device lock of the persistent launches: " + e_; return LCR_E_DEVICE; } } while (0)

  // ---- queue `side`: the fragment matrix goes to the host (pinned) only when a region takes the host epilogue
  PCHK(hipEventRecord(ev_in, sq));
  PCHK(hipStreamWaitEvent(side, ev_in, 0));
  if (any_host_post) {
    PCHK(h_pin[0].reserve((nr1 + 1) * 8)); PCHK(h_pin[1].reserve(nnz1 * 4));
    PCHK(h_pin[2].reserve(nnz1)); PCHK(h_pin[3].reserve(nr1 * 4));
    PCHK(hipMemcpyAsync(h_pin[0].p, in.d_row_ptr, (size_t)(nrow + 1) * 8, hipMemcpyDeviceToHost, side));
    if (nnz) {
      PCHK(hipMemcpyAsync(h_pin[1].p, in.d_col, (size_t)nnz * 4, hipMemcpyDeviceToHost, side));
      PCHK(hipMemcpyAsync(h_pin[2].p, in.d_val, (size_t)nnz, hipMemcpyDeviceToHost, side));
    }
    if (nrow) PCHK(hipMemcpyAsync(h_pin[3].p, in.d_row_links, (size_t)nrow * 4, hipMemcpyDeviceToHost, side));
  }

  // ---- queue `stream`: stage the phase matrices, fetch the per-region sizes
  if (ng) {
    if (!gstage_slots.empty()) {   // large regions: all CUs on one region at a time (every persistent launch goes to `side`)
      GRID_LOCK();
      PCHK(b_ctl.reserve((4 + 16) * sizeof(GridCtl))); PCHK(b_btot.reserve((size_t)(2 * std::max(1, k4_grid_blocks()) + 1) * 4 + 64));
      for (int g : gstage_slots) PCHK(k4_stage_launch_grid(si, so, L.dev, g, b_ctl.as<GridCtl>(), b_btot.as<int32_t>(), side));
      PCHK(hipEventRecord(ev_join, side));
      PCHK(hipStreamWaitEvent(sq, ev_join, 0));
    }
    PCHK(hipEventRecord(ev_csr, sq));   // the chain kernels on `side` read the staged matrices
  }
  if (async_mode) { PCHK(hipEventRecord(ev_user, sq)); PCHK(hipStreamWaitEvent(stream, ev_user, 0)); }
  HT("ph:presync");
  if (!pool) {
    // one ctx per GPU: share the host's hardware threads between the GPUs of the node
    int ndev = 1;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) ndev = 1;
    int nthreads = (int)std::thread::hardware_concurrency() / ndev;
    if (dbg.host_threads > 0) nthreads = dbg.host_threads;
    nthreads = std::max(1, std::min(nthreads, dbg.host_threads > 0 ? 256 : 48));
    pool = new HostPool(nthreads > 1 ? nthreads : 0);
  }
  if (ng) PCHK(hipStreamSynchronize(sq));
  HT("ph:synced");
  lap("stage + sizes");

  PostIn pin{in.d_row_ptr, in.d_col, in.d_val, in.d_row_links, const_cast<lcr_candidate*>(in.d_cand), in.d_cand_off,
             in.d_row_region_off, in.d_start0, P.st_sigma, P.st_delta, P.st_eta, (int8_t*)(d_res + res_tag), d_res + res_asg,
             (uint32_t*)(d_res + res_ps), d_read_rec.as<uint32_t>(), P.st_obj, (long long*)(d_hc + hc_obj), (lcr_candidate*)d_hc, prm.min_linkers, prm.max_enum_snps, prm.seed, prm.read_assign_cutoff, prm.min_phase_score, nullptr, P.reg, b_psrc.as<int32_t>()};
  if (prof) { PCHK(d_state[20].reserve(((size_t)(ng + 1) * 16 + 2 * 1024) * 8)); PCHK(hipMemsetAsync(d_state[20].p, 0, ((size_t)(ng + 1) * 16 + 2 * 1024) * 8, stream)); pin.dbg_clk = d_state[20].as<long long>(); }

  // ---- chain regions on queue `side` (their own copy of the state arrays)
  PhaseDev Pc = P;
  Pc.st_sigma = b_stc.as<int8_t>() + st_sig; Pc.st_delta = b_stc.as<int8_t>() + st_del; Pc.st_eta = b_stc.as<int8_t>() + st_eta;
  Pc.st_obj = (long long*)(b_stc.as<int8_t>() + st_obj);
  auto launch_chain_regions = [&]() -> int {
  if (!chain_slots.empty()) {
    // a region whose phase matrix is far beyond one CU gets all of them (persistent launch with grid barriers); the
    // others run side by side, one workgroup each.  LCR_GRID_MIN_ENTRIES moves the boundary (tests: 0 = every region).
    std::vector<int32_t> wide, big;
    const bool grid_generic = dbg.grid_generic != 0;   // test hook
    bool w_fits_limbs = true;   // (the batched rounds split w into a 23-bit and a signed limb; w < 0 for q < 4)
    for (int q = 0; q < 31; q++) w_fits_limbs = w_fits_limbs && std::llabs(P.lut.f1e[q] - P.lut.fe[q]) < (1ll << 45);
    for (int g : chain_slots) ((int64_t)stat[g].E >= grid_min ? big : wide).push_back(g);
    // longest first: a batch with more chain regions than CUs runs them in two generations (one sixteen-wave workgroup
    // per CU), and the workgroups are started in launch order -- the second generation should be the short regions
    // (cost ~ entries x rounds, rounds ~ SNPs / 4)
    std::stable_sort(wide.begin(), wide.end(), [&](int a, int b) {
      const int64_t ca = (int64_t)stat[a].E * (in.cand_region_off[a + 1] - in.cand_region_off[a] + 8);
      const int64_t cb = (int64_t)stat[b].E * (in.cand_region_off[b + 1] - in.cand_region_off[b] + 8);
      return ca > cb;
    });
    const int n_small = (int)wide.size(), n_big = (int)big.size(), nc = n_small + n_big;
    const int grid_waves = std::max(1, k4_grid_blocks()) * 16;
    std::vector<ChainDesc> desc(nc);
    int64_t tbl_cells = 0, adj_n = 0, part_n = 0, term_n = 0;
    int32_t max_state = 0;
    for (int k = 0; k < nc; k++) {
      const int g = k < n_small ? wide[k] : big[k - n_small];
      const int S = in.cand_region_off[g + 1] - in.cand_region_off[g];
      ChainDesc& d = desc[k];
      d.slot = g; d.W = std::max(1, std::min(stat[g].W, S)); d.fast_lds = 0;
      d.batch_lds = 0; d.term_off = 0;
      if (k < n_small) { d.term_off = (int32_t)std::min<int64_t>(term_n, INT32_MAX); term_n += stat[g].E; }
      if (k >= n_small && !grid_generic) {
        const size_t need = k4_grid_fast_lds(stat[g].R, S); if (need <= (size_t)K4_GRID_FAST_LDS_MAX) d.fast_lds = (int32_t)need;
        const size_t need_b = k4_grid_batch_lds(S); if (dbg.spec_batch && w_fits_limbs && need_b <= (size_t)K4_GRID_FAST_LDS_MAX) d.batch_lds = (int32_t)need_b;
      }
      d.tbl_off = tbl_cells; tbl_cells += (int64_t)S * d.W;
      d.adj_off = adj_n; adj_n += 2 * (int64_t)S * d.W;
      d.n_parts = k < n_small ? 16 : (int32_t)std::max<int64_t>(16, std::min<int64_t>(grid_waves, (int64_t)(1 << 23) / S));
      d.part_off = part_n; part_n += (int64_t)d.n_parts * S;
      if (k < n_small) max_state = std::max(max_state, stat[g].R + 2 * S);
    }
    // post-phase: device epilogue for the chain regions that fit it
    std::vector<int32_t> post_slots;
    for (int k = 0; k < nc; k++) if (!host_post[desc[k].slot] && !grid_post[desc[k].slot]) post_slots.push_back(desc[k].slot);
    const size_t nps = post_slots.size();
    const size_t desc_bytes = ((size_t)nc * sizeof(ChainDesc) + 15) & ~(size_t)15;
    PCHK(h_pin[10].reserve(desc_bytes + nps * 4 + 64));
    PCHK(b_desc.reserve(desc_bytes + 64)); PCHK(b_slots.reserve(nps * 4 + 64));
    memcpy(h_pin[10].p, desc.data(), (size_t)nc * sizeof(ChainDesc));
    memcpy(h_pin[10].as<uint8_t>() + desc_bytes, post_slots.data(), nps * 4);
    PCHK(b_tbl.reserve((size_t)tbl_cells * 8 + 64)); PCHK(b_adj.reserve((size_t)adj_n * 4 + 64)); PCHK(b_part.reserve((size_t)part_n * 4 + 64));
    const size_t ni = nc1 + (size_t)ng + 1;   // per-SNP int32 arrays: adj_ptr, blk_ptr (ni each), blk_of, blk_pos, blk_nodes, queue (nc1), stack (2 nc1)
    PCHK(b_snpi.reserve((2 * ni + 6 * nc1) * 4 + 64)); PCHK(b_snpb.reserve(3 * nc1 + 64)); PCHK(b_q.reserve(2 * nc1 * 8 + 64));
    PCHK(b_info.reserve((size_t)std::max(ng, 1) * 8 + 64)); PCHK(b_rowi.reserve(nr1 * 4 + 64)); PCHK(b_enti.reserve(2 * nnz1 * 4 + 64));
    PCHK(b_work.reserve(st_bytes + 64)); PCHK(b_macc.reserve(nc1 * 8 + 64)); PCHK(d_state[39].reserve((2 * (nr1 / 64 + (size_t)ng + 2)) * 8 + 64)); PCHK(b_ctl.reserve((4 + 16) * sizeof(GridCtl)));
    ChainDev C{};
    {   // speculative half-rounds of the all-CU regions: lanes x (sigma words | delta, eta bytes | result) of the largest such region
      size_t ng_max = 1, s_max = 16;
      for (int k = n_small; k < nc; k++) { const int g = desc[k].slot; ng_max = std::max(ng_max, (size_t)(stat[g].R + 63) / 64 + 1); s_max = std::max(s_max, (size_t)(in.cand_region_off[g + 1] - in.cand_region_off[g])); }
      const size_t s8 = (s_max + 15) & ~(size_t)15, lanes = (size_t)std::max(1, dbg.spec_lanes);
      PCHK(d_spec_sig.reserve(lanes * ng_max * 8 + 64)); PCHK(d_spec_de.reserve(lanes * 2 * s8 + 64)); PCHK(d_spec_res.reserve(lanes * 8 + 64));
      C.spec_sig = d_spec_sig.as<unsigned long long>(); C.spec_de = d_spec_de.as<int8_t>(); C.spec_res = d_spec_res.as<long long>();
      C.spec_lanes = (int32_t)lanes; C.spec_ng = (int32_t)ng_max; C.spec_s8 = (int32_t)s8;
      size_t e_max = 0, g4_max = 0, r_max = 0, sb_max = 0;   // packed entries of the largest all-CU region with device-coherent rounds
      for (int k = n_small; k < nc; k++) if (desc[k].fast_lds || desc[k].batch_lds) e_max = std::max(e_max, (size_t)stat[desc[k].slot].E);
      for (int k = n_small; k < nc; k++) if (desc[k].batch_lds) {
        const int g = desc[k].slot;
        g4_max = std::max(g4_max, ((size_t)stat[g].E + 3 * (size_t)stat[g].R + 3) / 4); r_max = std::max(r_max, (size_t)stat[g].R);
        sb_max = std::max(sb_max, (size_t)(in.cand_region_off[g + 1] - in.cand_region_off[g]));
      }
      if (g4_max) {   // batched rounds: groups | unit pointers | sigma bytes | SNP masks | barrier payload
        auto al = [](size_t x) { return (x + 63) / 64 * 64; };
        const size_t nb = r_max / 64 + 2;   // blocks of 64 sorted rows; the groups: every row padded to its block's longest (<= 64 (S / 4 + 1) more)
        g4_max += 64 * (sb_max / 4 + 2) + 64;
        const size_t o_up = (g4_max + 4) * 16, o_bs = o_up + al(nb * 8), o_pm = o_bs + al(nb * 8), o_iv = o_pm + al(r_max * 4 + 256), o_sig = o_iv + al(r_max * 4 + 256),
                     o_m = o_sig + al(r_max + 64), o_ctl = o_m + al((sb_max + 2) * 4);
        PCHK(d_bt.reserve(o_ctl + K4_GRID_BATCH_CTL_BYTES + 64));
        uint8_t* bp = d_bt.as<uint8_t>();
        C.bt_pk4 = (uint32_t*)bp; C.bt_up4 = (int32_t*)(bp + o_up); C.bt_bs32 = (uint32_t*)(bp + o_bs); C.bt_perm = (int32_t*)(bp + o_pm); C.bt_inv = (int32_t*)(bp + o_iv);
        C.bt_sig8 = bp + o_sig; C.bt_m32 = (uint32_t*)(bp + o_m); C.bt_ctl = bp + o_ctl;
        C.bt_cap4 = (int64_t)g4_max; C.spec_batch = 1;
      }
      C.pk_cap = (int64_t)e_max;
      if (e_max) { PCHK(d_pk.reserve(2 * (e_max + 8) * 4 + 64)); C.pk_csr = d_pk.as<uint32_t>(); C.pk_csc = C.pk_csr + (e_max + 8); }
    }
    Pc.scratch = nullptr; Pc.scratch_stride = 0; Pc.lds_state = 0; Pc.lds_mat = 0;
    C.P = Pc;
    C.desc = b_desc.as<ChainDesc>();
    C.row_ptr = in.d_row_ptr; C.col = in.d_col; C.val = in.d_val;
    C.cand = in.d_cand; C.cand_off = in.d_cand_off; C.row_region_off = in.d_row_region_off;
    C.prow_src = b_psrc.as<int32_t>();
    C.ld_tbl = b_tbl.as<uint32_t>(); C.ld_adj = b_adj.as<int32_t>(); C.part_cnt = b_part.as<int32_t>();
    int32_t* si32 = b_snpi.as<int32_t>();
    C.adj_ptr = si32; C.blk_ptr = si32 + ni; C.blk_of = si32 + 2 * ni; C.blk_pos = C.blk_of + nc1; C.blk_nodes = C.blk_pos + nc1;
    C.queue = C.blk_nodes + nc1; C.stack = C.queue + nc1;
    C.seen = b_snpb.as<uint8_t>(); C.ld_ok = C.seen + nc1; C.new_hap = (int8_t*)(C.ld_ok + nc1);
    C.qs = b_q.as<double>(); C.qfs = C.qs + nc1;
    C.blk_info = b_info.as<int32_t>();
    C.flipcol = b_rowi.as<int32_t>(); C.erow = b_enti.as<int32_t>(); C.cent = C.erow + nnz1;
    C.w_sigma = b_work.as<int8_t>() + st_sig; C.w_delta = b_work.as<int8_t>() + st_del; C.w_eta = b_work.as<int8_t>() + st_eta;
    C.macc = b_macc.as<unsigned long long>(); C.ctl = b_ctl.as<GridCtl>(); C.spec_ctl = b_ctl.as<GridCtl>() + 4; C.sig_words = d_state[39].as<unsigned long long>();
    for (int q = 0; q < 31; q++) { C.le[q] = L.le[q]; C.l1e[q] = L.l1e[q]; }
    C.p_homref = L.p_homref; C.p_homvar = L.p_homvar; C.log_theta = L.log_theta; C.log2 = L.log2;
    if (dbg.chain_ties && dbg.tie_arith >= 3 && n_small && term_n < INT32_MAX) {
      PCHK(d_tie_terms.reserve((size_t)term_n * 16 + 64)); C.tie_terms = d_tie_terms.as<double>();
      PCHK(d_tie_flag.reserve((size_t)std::max(ng, 1) * 4 + 64)); PCHK(d_tie_q.reserve((2 * nr1 + 2 * nc1) * 8 + 64)); PCHK(d_tie_ch.reserve(2 * nc1 + 64));
      C.tie_flag = d_tie_flag.as<int32_t>(); C.tie_qrow = d_tie_q.as<double>(); C.tie_qsnp = C.tie_qrow + 2 * nr1; C.tie_ch = d_tie_ch.as<int8_t>();
    }
    if (prof) { C.dbg = d_state[20].as<long long>() + (size_t)ng * 16; PCHK(hipMemsetAsync(C.dbg, 0, (16 + 2 * 1024) * 8, side)); }   // 16 step timers + per-workgroup sigma / delta step times
    chain_dev = C; chain_desc = desc;   // (lcr_get_ld_blocks reads the blocks back)
    PCHK(hipStreamWaitEvent(side, ev_csr, 0));
    PCHK(hipMemcpyAsync(b_desc.p, h_pin[10].p, (size_t)nc * sizeof(ChainDesc), hipMemcpyHostToDevice, side));
    if (nps) PCHK(hipMemcpyAsync(b_slots.p, h_pin[10].as<uint8_t>() + desc_bytes, nps * 4, hipMemcpyHostToDevice, side));
    // the working state lives in dynamic LDS when it fits, with room behind it for the largest matrix of the class
    auto launch_class = [&](const std::vector<int32_t>& regs, int first, int32_t mstate, size_t lds_budget) -> hipError_t {
      if (regs.empty()) return hipSuccess;
      ChainDev Ck = C;
      const int32_t stride = (mstate + 63) & ~63;
      const size_t dyn_state = (size_t)stride <= lds_budget * 3 / 4 ? (size_t)stride : 0;
      Ck.P.scratch_stride = stride; Ck.P.lds_state = dyn_state ? 1 : 0;
      uint32_t want = 0;
      for (int g : regs) {
        const uint32_t m = matview_bytes(stat[g].R, in.cand_region_off[g + 1] - in.cand_region_off[g], stat[g].E);
        if (dyn_state && dyn_state + m + 64 <= lds_budget) want = std::max(want, m);
      }
      Ck.P.lds_mat = (int32_t)want;
      return k4_chain_launch_wg(Ck, first, (int)regs.size(), dyn_state + (size_t)want, side);
    };
    PCHK(launch_class(wide, 0, max_state, 64 * 1024));
    if (n_big) GRID_LOCK();
    for (int k = 0; k < n_big; k++) PCHK(k4_chain_launch_grid(C, n_small + k, (size_t)std::max(desc[n_small + k].fast_lds, desc[n_small + k].batch_lds), side));
    PostIn pinc = pin;
    pinc.st_sigma = Pc.st_sigma; pinc.st_delta = Pc.st_delta; pinc.st_eta = Pc.st_eta; pinc.st_obj = Pc.st_obj;
    if (nps) {
      // 33 KB of static stage buffers + up to 64 KB of region image
      // more chain regions than CUs (a sixteen-wave workgroup has a CU to itself): eight waves, two regions per CU --
      // one generation of workgroups instead of two
      if (((int)nps > std::max(1, k4_grid_blocks()) || dbg.post_half /* test hook */) && post_lds <= 56 * 1024) {
        PCHK(launch_k4_post(CHAIN_THREADS / 2, (unsigned)nps, post_lds, side, pinc, b_slots.as<int32_t>(), (int32_t)nps, plut));
      } else {
        PCHK(launch_k4_post(CHAIN_THREADS, (unsigned)nps, post_lds, side, pinc, b_slots.as<int32_t>(), (int32_t)nps, plut));
      }
    }
    PCHK(hipGetLastError());
  } else chain_desc.clear();
  return LCR_OK;
  };
  auto launch_enum_regions = [&]() -> int {
  // ---- enumeration regions: all restarts in one launch per class, then k4_enum_resolve per class (the winner's state is taken
  // from what its restart left; only the global-memory class re-runs its winner)
  if (!enum_slots.empty()) {
    // heaviest regions first (tiles are started in grid order: the kernel's tail should be the light ones; the post-phase
    // kernel's workgroups follow the same order)
    // (this block is on the critical path -- the device idles between k4_stage and the first restart: one sort of packed keys,
    // one layout computation per region, no allocation beyond the lists themselves: 48 -> ~10 us for C3's 366 regions)
    {
      std::vector<uint64_t>& key = enum_keys;   // descending E, ties in ascending region order (= the stable sort it replaces)
      key.resize(enum_slots.size());
      for (size_t k = 0; k < enum_slots.size(); k++) key[k] = ((uint64_t)(0xffffffffu - (uint32_t)stat[enum_slots[k]].E) << 32) | (uint32_t)enum_slots[k];
      std::sort(key.begin(), key.end());
      for (size_t k = 0; k < enum_slots.size(); k++) enum_slots[k] = (int32_t)(uint32_t)key[k];
    }
    HT("en:sorted");
    int32_t max_state = 0;
    for (int g : enum_slots) max_state = std::max(max_state, stat[g].R + 2 * (in.cand_region_off[g + 1] - in.cand_region_off[g]));
    const int32_t stride = (max_state + 63) & ~63;
    P.scratch_stride = stride;
    const bool force_big = dbg.enum_force_big != 0;        // test hooks: exercise the fallback kernels
    const bool force_stream = dbg.enum_force_stream != 0;
    // class 2: register-resident kernel (<= 32 entries per lane; smaller instantiations were measured: separate
    // launches each pay their own tail, one CK=32 launch with early exits is faster); class 3: same kernel
    // streaming its entries from LDS (any share size); class 4: global-memory kernel (matrix larger than the
    // LDS budget); classes 0 / 1 unused.  (A second register-resident instantiation with 40 entries per lane -- three
    // quarters of C4's restarts sit at 33 .. 36 -- needs 207 VGPRs, two waves per SIMD, and is slower than streaming
    // from LDS at six: 0.61 vs 0.51 ms per launch pair on C4, 0.23 vs 0.13 on C3.)
    constexpr int NCLS = 5;
    std::vector<EnumSpan> spans[NCLS];
    for (int k = 1; k < 4; k++) spans[k].reserve(enum_slots.size());
    size_t n_t[NCLS] = {0, 0, 0, 0, 0};   // tiles per class = grid of the class's kernel
#ifndef ENUM_PER3
#define ENUM_PER3 2u
#endif
    const bool ebits = dbg.enum_bits != 0;   // classes 1 - 3 by k4_enum_bits: eight restarts per wave
    const uint32_t per_of[NCLS] = {1u, ENUM_PER3 * ENUM_WAVES, ENUM_TILE_JOBS, ebits ? ENUM_BITS_PER : ENUM_PER3 * ENUM_WAVES, 1u};
    std::vector<int64_t>& job_base = enum_job_base; std::vector<int64_t>& st_base = enum_st_base;   // st_base: first word of the region's saved restart states (classes 1 - 3)
    job_base.assign(ng, 0); st_base.assign(ng, 0);
    int64_t nj = 0, st_words = 0, big_words = 0, big_emax = 0;
    uint32_t lds_need[NCLS] = {0, 0, 0, 0, 0}, res_lds[NCLS] = {0, 0, 0, 0, 0};   // (res_lds: k4_enum_resolve's image of the class's largest region)
    // enumeration regions with the device epilogue: those of the streaming class (the largest matrices, so the longest epilogues)
    // are post-processed on their own queue right behind their resolve kernel, beside the register class's resolve; the others
    // behind everything on `stream`
    std::vector<int32_t> post_slots, post_slots_b, post_slots_c;   // (c: the global-memory class, behind its winners' second launch)
    post_slots.reserve(enum_slots.size());
    for (int g : enum_slots) {
      const int S = in.cand_region_off[g + 1] - in.cand_region_off[g];
      const StageStat& st = stat[g];
      const EnumLayout EL_old = enum_layout(st.R, st.E), EL_bits = enum_layout(st.R, st.E, true, (uint32_t)std::min(S, 32));
      // k4_enum_bits where its image (a sigma byte per row and wave) fits; regions beyond it keep the streaming kernel (class 1, a launch of its own)
      const bool use_bits = ebits && S <= 31 && EL_bits.total <= ENUM_LDS_MAX;
      const EnumLayout EL = use_bits ? EL_bits : EL_old;
      const uint32_t RL = resolve_layout((uint32_t)st.R, (uint32_t)st.E, (uint32_t)S).total;
      int cls = 4;
      if (!force_big && st.R < 65536 && st.E < 65536 && S <= 31 && EL.total <= ENUM_LDS_MAX && RL <= ENUM_LDS_MAX && st.max_rows <= 64)
#ifdef ENUM_MASK64
        cls = force_stream ? 3 : (st.max_n <= 32 ? 2 : 3);
#else
        cls = force_stream ? 3 : (st.max_n <= 32 && st.max_rows <= 32 ? 2 : 3);
#endif   // (register-resident form: <= 32 entries and <= 32 rows per lane)   // (the 8 / 16 instantiations: one launch has one tail; a 40-entry one: below)
      // class 1: the streaming kernel for the few regions whose image needs more than ENUM_LDS_BYTES (up to the 64 KB a launch gets
      // without opting in): a launch of their own, so that their LDS does not set the occupancy of class 3's tiles
      // (C4 share: three such regions used to take the global-memory kernel BEHIND class 3 on its queue, 0.68 ms of the critical
      // path: step 5.49 -> 4.97 ms; a queue of their own was measured too: HIP maps a fourth stream onto one of the first three's
      // hardware queues, no difference)
      // with k4_enum_bits: class 3 = its regions (one launch), class 1 = the LDS-resident regions beyond its image (streaming kernel)
      if (ebits && cls < 4) cls = use_bits ? 3 : 1;
      if (!ebits && (cls == 2 || cls == 3) && (std::max(EL.total, RL) > ENUM_LDS_BYTES || dbg.enum_force_stream == 2 /* test hook */)) cls = 1;
      //  the saved restart states of the LDS classes are allocated up front: a raised max_enum_snps (2^20 restarts x
      // R / 64 + 3 words) goes to the global-memory class instead, whose states are kept only inside a budget
      if (cls < 4 && (int64_t)((uint64_t)1 << std::min(S, 62)) * enum_state_words((uint32_t)st.R) > ((int64_t)1 << 27)) cls = 4;
      if (cls < 4) { lds_need[cls] = std::max(lds_need[cls], EL.total); res_lds[cls] = std::max(res_lds[cls], RL); }
      job_base[g] = nj;
      const uint64_t n = 1ull << S;
      // saved restart states: every region of classes 1 - 3; of the global-memory class while they fit a budget of 2^26 words
      // (512 MB -- beyond it that region's equal-objective restarts fall to "first maximum", counted as unresolved)
      st_base[g] = -1;
      if (cls < 4) { st_base[g] = st_words; st_words += (int64_t)n * enum_state_words((uint32_t)st.R); }
      else if ((int64_t)n * enum_state_words((uint32_t)st.R) <= ((int64_t)1 << 26) - big_words) {
        st_base[g] = st_words; st_words += (int64_t)n * enum_state_words((uint32_t)st.R); big_words += (int64_t)n * enum_state_words((uint32_t)st.R);
        big_emax = std::max<int64_t>(big_emax, st.E);
      }
      if (n_t[cls] + (n + per_of[cls] - 1) / per_of[cls] > 0x7fffffffull) { if (err) *err = "too many enumeration restarts for one launch"; return LCR_E_ARG; }
      spans[cls].push_back({g, (uint32_t)n_t[cls]});
      n_t[cls] += (size_t)((n + per_of[cls] - 1) / per_of[cls]);
      nj += (int64_t)n;
      if (!host_post[g] && !grid_post[g]) (cls == 3 ? post_slots_b : (cls == 4 ? post_slots_c : post_slots)).push_back(g);
      if (prof && cls == 4) fprintf(stderr, "[phase]   global-memory enumeration region %d: R %d E %d S %d max_rows %d max_n %d, LDS image %u B\n", g, st.R, st.E, S, st.max_rows, st.max_n, EL.total);
    }
    if (prof) fprintf(stderr, "[phase]   enumeration classes: register %zu regions / %zu tiles, streaming %zu / %zu, streaming with a large image %zu / %zu, global %zu / %zu\n", spans[2].size(), n_t[2], spans[3].size(), n_t[3], spans[1].size(), n_t[1], spans[4].size(), n_t[4]);
    // one upload: spans of every class | job_base | slots | post slots ; then job objectives and winners
    size_t n_w[NCLS], s_off[NCLS], n_spans = 0;
    for (int k = 0; k < NCLS; k++) { n_w[k] = spans[k].size(); s_off[k] = n_spans; n_spans += n_w[k]; }
    const size_t ns = enum_slots.size(), nps_a = post_slots.size(), nps_b = post_slots_b.size(), nps_c = post_slots_c.size(), nps = nps_a + nps_b + nps_c;
    const size_t off_jb_al = (n_spans * sizeof(EnumSpan) + 7) & ~(size_t)7;
    const size_t off_sb = off_jb_al + (size_t)ng * 8;   // st_base
    const size_t up_bytes = off_sb + (size_t)ng * 8 + (ns + nps) * 4;
    HT("en:classes");
    PCHK(d_enum_st.reserve((size_t)std::max<int64_t>(st_words, 1) * 8));
    PCHK(b_job.reserve(up_bytes + 64));
    PCHK(b_obj.reserve((size_t)nj * 8 + (size_t)ng * 8 + (size_t)ng * 8 + 64));   // objectives | winners | tiles done | best objective seen per region
    PCHK(h_pin[8].reserve(up_bytes + 64));   // pinned: the upload is queued, not staged
    uint8_t* const up = h_pin[8].as<uint8_t>();
    for (int k = 0; k < NCLS; k++) memcpy(up + s_off[k] * sizeof(EnumSpan), spans[k].data(), n_w[k] * sizeof(EnumSpan));
    memcpy(up + off_jb_al, job_base.data(), (size_t)ng * 8);
    memcpy(up + off_sb, st_base.data(), (size_t)ng * 8);
    memcpy(up + off_sb + (size_t)ng * 8, enum_slots.data(), ns * 4);
    memcpy(up + off_sb + (size_t)ng * 8 + ns * 4, post_slots.data(), nps_a * 4);
    memcpy(up + off_sb + (size_t)ng * 8 + (ns + nps_a) * 4, post_slots_b.data(), nps_b * 4);
    memcpy(up + off_sb + (size_t)ng * 8 + (ns + nps_a + nps_b) * 4, post_slots_c.data(), nps_c * 4);
    PCHK(hipMemcpyAsync(b_job.p, up, up_bytes, hipMemcpyHostToDevice, stream));
    HT("en:up_q");
    const EnumSpan* d_sp = b_job.as<EnumSpan>();
    const int64_t* d_jb = (const int64_t*)(b_job.as<uint8_t>() + off_jb_al);
    const int64_t* d_sb = (const int64_t*)(b_job.as<uint8_t>() + off_sb);
    const int32_t* d_sl = (const int32_t*)(b_job.as<uint8_t>() + off_sb + (size_t)ng * 8);
    const int32_t* d_psl = d_sl + ns;
    long long* d_obj = b_obj.as<long long>();
    uint32_t* d_win = (uint32_t*)(d_obj + nj);
    const size_t n_big_blocks = std::max(n_t[4], n_w[4]);
    // the repair pass (tie classes 2 / 4 met by a fast kernel's restart): per launch queue a list of REDO_CAP restarts and REDO_GRID
    // workgroups' worth of state scratch + score scratch (2 doubles per row), behind the global-memory class's own
    const bool full_ties = dbg.tie_arith >= 3;
    int32_t max_rows_any = 1;
    for (int g : enum_slots) max_rows_any = std::max(max_rows_any, stat[g].R);
    const int64_t qstride = ((int64_t)2 * max_rows_any + 15) & ~(int64_t)15;
    const size_t n_scr_blocks = n_big_blocks + 2 * (size_t)REDO_GRID;
    PCHK(b_scr.reserve((size_t)stride * n_scr_blocks + 64));
    P.scratch = b_scr.as<int8_t>();
    if (full_ties) {
      PCHK(b_qrow.reserve((size_t)qstride * n_scr_blocks * 8 + 64));
    }
    uint32_t* const redo_a = full_ties ? b_redo.as<uint32_t>() : nullptr;                          // classes 1 / 2 (first queue)
    uint32_t* const redo_b = full_ties ? b_redo.as<uint32_t>() + 4 + 2 * REDO_CAP : nullptr;       // class 3 (`aux`)
    double* const qrow_all = full_ties ? b_qrow.as<double>() : nullptr;
    auto launch_redo = [&](uint32_t* redo, int which, hipStream_t q) {
      if (!redo) return;
      const size_t b0 = n_big_blocks + (size_t)which * REDO_GRID;
      launch_k4_enum_redo(REDO_GRID, q, P, redo, REDO_CAP, b_scr.as<int8_t>() + (size_t)stride * b0, stride, qrow_all + (size_t)qstride * b0, qstride,
                          d_jb, d_obj, d_sb, d_enum_st.as<unsigned long long>(),
                          // (the matrix in LDS where it fits -- not for the queue that, with k4_enum_bits, only carries the regions beyond ITS image: their
                          // matrices are the largest, and that launch runs beside the bit-state kernel, where 512 workgroups asking for 64 KB each wait for
                          // it to drain even when the list is empty: 5 us -> 0.98 ms on the C4 share)
                          (which == 0 && ebits) ? 0u : (uint32_t)std::max(0, dbg.redo_lds));
    };
    // the classes touch disjoint regions: class 2 on `stream`, classes 3 / 4 beside it on `aux` (their tails overlap)
    long long* const d_rbest = d_rbest_buf.as<long long>();   // (filled in front of the staging kernel)
    auto launch = [&](const size_t* cnt, const uint32_t* win) -> hipError_t {
      const bool fork = (cnt[2] || (ebits && cnt[1])) && (cnt[3] || cnt[4]);   // (with k4_enum_bits: the streaming kernel's few large regions beside its launch)
      hipStream_t s34 = fork ? aux : stream, s1 = stream;
      hipError_t e = hipSuccess;
      if (fork) { if ((e = hipEventRecord(ev_fork, stream)) != hipSuccess) return e; if ((e = hipStreamWaitEvent(aux, ev_fork, 0)) != hipSuccess) return e; }
      if (cnt[1]) {   // (first, ahead of class 2 on its queue: the largest matrices have the longest restarts)
        launch_k4_enum_reg(0, (unsigned)cnt[1], lds_need[1], s1, P, d_sp + s_off[1], (int32_t)n_w[1], per_of[1], d_jb, d_obj, d_sb, d_enum_st.as<unsigned long long>(), d_rbest, redo_a, REDO_CAP);
        if (!win && !cnt[2]) launch_redo(redo_a, 0, s1);   // (with a class 2 launch behind it on this queue: one repair pass for both, below)
        if (!win && !cnt[2]) launch_k4_enum_resolve((unsigned)n_w[1], res_lds[1], s1, P, d_sp + s_off[1], d_jb, d_obj, d_sb, d_enum_st.as<unsigned long long>());
      }
      // classes 2 / 3: all restarts (both kernels queued first), then per class `prob > largest_prob` over each region's restarts from
      // the objectives, signatures and states they left (phase.rs:1113-1119; ties between configurations of maximal objective by
      // their f64 sums) -- a workgroup per region -- and the post-phase kernel of the class's regions, each behind its own
      // restarts on its own queue (the streaming class has the largest matrices, so the longest epilogues: they run beside the
      // register class's resolve instead of behind it)
      unsigned long long* const d_st = d_enum_st.as<unsigned long long>();
      if (cnt[2]) launch_k4_enum_reg(32, (unsigned)cnt[2], lds_need[2], stream, P, d_sp + s_off[2], (int32_t)n_w[2], per_of[2], d_jb, d_obj, d_sb, d_st, d_rbest, redo_a, REDO_CAP);
      if (cnt[3]) launch_k4_enum_reg(ebits ? -1 : 0, (unsigned)cnt[3], lds_need[3], s34, P, d_sp + s_off[3], (int32_t)n_w[3], per_of[3], d_jb, d_obj, d_sb, d_st, d_rbest, redo_b, REDO_CAP);
      if (!win && async_mode) {   // (the dense part of the stage ends here on both queues: the next batch's pileup waits for these)
        if (cnt[1] || cnt[2]) { if ((e = hipEventRecord(ev_gate[0], stream)) != hipSuccess) return e; gate_set[0] = true; }
        if (fork && cnt[3]) { if ((e = hipEventRecord(ev_gate[1], s34)) != hipSuccess) return e; gate_set[1] = true; }
        else if (!fork && cnt[3]) { if ((e = hipEventRecord(ev_gate[0], stream)) != hipSuccess) return e; gate_set[0] = true; }
      }
      if (!win) {
        if (cnt[2]) launch_redo(redo_a, 0, stream);
        if (cnt[1] && cnt[2]) launch_k4_enum_resolve((unsigned)n_w[1], res_lds[1], s1, P, d_sp + s_off[1], d_jb, d_obj, d_sb, d_enum_st.as<unsigned long long>());
        if (cnt[2]) launch_k4_enum_resolve((unsigned)n_w[2], res_lds[2], stream, P, d_sp + s_off[2], d_jb, d_obj, d_sb, d_st);
        // eight waves per region: the slowest region (most rows) sets the kernel's length, and every row sweep of the
        // epilogue is a pass of <threads> rows (148 -> 103 us on C3)
        if (nps_a && (e = launch_k4_post(2 * LCR_BLOCK, (unsigned)nps_a, post_lds, stream, pin, d_psl, (int32_t)nps_a, plut)) != hipSuccess) return e;
        if (cnt[3]) { launch_redo(redo_b, 1, s34); launch_k4_enum_resolve((unsigned)n_w[3], res_lds[3], s34, P, d_sp + s_off[3], d_jb, d_obj, d_sb, d_st); }
        if (nps_b && (e = launch_k4_post(2 * LCR_BLOCK, (unsigned)nps_b, post_lds, s34, pin, d_psl + nps_a, (int32_t)nps_b, plut)) != hipSuccess) return e;
      }
      if (cnt[4]) launch_k4_enum_big((unsigned)cnt[4], s34, P, d_sp + s_off[4], (int32_t)n_w[4], per_of[4], d_jb, d_obj, win, d_sb, d_enum_st.as<unsigned long long>(), qrow_all, qstride);
      if (fork) { if ((e = hipEventRecord(ev_join, aux)) != hipSuccess) return e; if ((e = hipStreamWaitEvent(stream, ev_join, 0)) != hipSuccess) return e; }
      return e;
    };
    PCHK(launch(n_t, nullptr));
    HT("en:launched");
    if (n_w[4]) {   // the global-memory class: its own resolve kernel (equal objectives by the f64 sums) and the winners' second launch
      const size_t only4[NCLS] = {0, 0, 0, 0, n_w[4]};
      const int64_t tstride = (big_emax + 63) & ~(int64_t)63;
      PCHK(b_terms.reserve((size_t)std::max<int64_t>(tstride, 64) * n_w[4] * 8 + 64));
      launch_k4_enum_resolve_big((int32_t)n_w[4], stream, P, d_sp + s_off[4], d_jb, d_obj, d_sb, d_enum_st.as<unsigned long long>(), d_win, b_terms.as<double>(), tstride);
      PCHK(launch(only4, d_win));
    }
    if (nps_c) PCHK(launch_k4_post(2 * LCR_BLOCK, (unsigned)nps_c, post_lds, stream, pin, d_psl + nps_a + nps_b, (int32_t)nps_c, plut));
    PCHK(hipGetLastError());
  }
  return LCR_OK;
  };
  // Order of the two queues' launches: the enumeration tiles (thousands of four-wave workgroups) and the chain regions'
  // sixteen-wave workgroups (a CU's worth of LDS and wave slots each) compete for the same CUs, and whoever is launched
  // second only finds room as the first drains.  A few chain regions go first (C3: sixteen of them, 0.2 ms of work that
  // used to end with the enumeration 0.47 ms after its launch: phase stage 0.83 -> 0.72 ms); when they would fill the
  // device themselves (ONT-dRNA: 368) the few enumeration regions go first.
  // (round 4: the enumeration's host preparation now ends ~50 us earlier when it goes first, and its kernel + k4_enum_resolve + k4_post
  // are the critical path: C3 2.48 -> 2.44 ms, C4 5.59 -> 5.57 with the chain regions behind it in every case)
  const bool chain_first = false;
  if (chain_first) { const int rc = launch_chain_regions(); if (rc) return rc; }
  { const int rc = launch_enum_regions(); if (rc) return rc; }
  HT("ph:enum_q");
  if (!chain_first) { const int rc = launch_chain_regions(); if (rc) return rc; }
  if (prof && !enum_slots.empty()) {   // share sizes of the enumeration regions (entries per lane decide the kernel class)
    std::vector<int> mn; uint64_t jobs[3] = {0, 0, 0};
    for (int g : enum_slots) {
      mn.push_back(stat[g].max_n);
      const int S = in.cand_region_off[g + 1] - in.cand_region_off[g];
      jobs[stat[g].max_n <= 32 ? 0 : stat[g].max_n <= 48 ? 1 : 2] += 1ull << S;
    }
    std::sort(mn.begin(), mn.end());
    fprintf(stderr, "[phase]   enumeration regions: %zu, entries per lane min %d median %d p90 %d max %d; restarts with <= 32: %llu, 33-48: %llu, > 48: %llu\n",
            mn.size(), mn.front(), mn[mn.size() / 2], mn[mn.size() * 9 / 10], mn.back(), (unsigned long long)jobs[0], (unsigned long long)jobs[1], (unsigned long long)jobs[2]);
  }
  lap("enum launch");
  // ---- post-phase steps of the regions beyond k4_post's LDS image: all CUs on one region at a time, on `side`
  if (!gpost_slots.empty()) {
    GRID_LOCK();
    const int grid_waves = std::max(1, k4_grid_blocks()) * 16;
    PostScratch ps{};
    ps.n_parts = (int32_t)std::max<int64_t>(16, std::min<int64_t>(grid_waves, (int64_t)(1 << 23) / (int64_t)gp_S));
    const size_t S8 = (gp_S + 8) & ~(size_t)7, R8 = (gp_rows + 8) & ~(size_t)7, E8 = (gp_E + 8) & ~(size_t)7;
    PCHK(b_ps.reserve(S8 * (3 * 8 + 4 * 4 + 4 + 2 * 4) + R8 * (4 + 5 + 4) + 64 + 64));   // per SNP: 3 doubles, 4 + 2 int32, 4 bytes; per row: rptr, fdirt + 5 bytes
    PCHK(b_pse.reserve(E8 * (3 * 4 + 1) + 64));
    PCHK(b_psp.reserve((size_t)ps.n_parts * gp_S * 4 + 64));
    PCHK(b_ctl.reserve((4 + 16) * sizeof(GridCtl)));
    uint8_t* p = b_ps.as<uint8_t>();
    ps.sps = (double*)p; p += 8 * S8; ps.rpa = (double*)p; p += 8 * S8; ps.rpb = (double*)p; p += 8 * S8;
    ps.sflags = (uint32_t*)p; p += 4 * S8; ps.soflags = (uint32_t*)p; p += 4 * S8; ps.parent = (int32_t*)p; p += 4 * S8; ps.ccptr = (int32_t*)p; p += 4 * S8;
    ps.rptr = (int32_t*)p; p += 4 * R8;
    ps.shap = (int8_t*)p; p += S8; ps.sgt = (int8_t*)p; p += S8; ps.svt = (int8_t*)p; p += S8; ps.rcode = p; p += S8;
    ps.tag = (int8_t*)p; p += R8; ps.asg = p; p += R8; ps.fp = p; p += R8; ps.lok = p; p += R8; ps.dirty = p; p += R8;
    ps.fdirt = (int32_t*)p; p += 4 * R8; ps.minf = (int32_t*)p; p += 4 * S8; ps.ndraw = (int32_t*)p; p += 4 * S8; ps.gwords = (int32_t*)p; p += 64;
    uint8_t* q = b_pse.as<uint8_t>();
    ps.ecol = (int32_t*)q; q += 4 * E8; ps.erow = (int32_t*)q; q += 4 * E8; ps.cent = (int32_t*)q; q += 4 * E8; ps.ev = q;
    ps.pcnt = b_psp.as<int32_t>();
    ps.ctl = b_ctl.as<GridCtl>() + 1;
    PostIn pinc = pin;
    pinc.st_sigma = Pc.st_sigma; pinc.st_delta = Pc.st_delta; pinc.st_eta = Pc.st_eta; pinc.st_obj = Pc.st_obj;
    bool waited = false;
    for (int g : gpost_slots) {
      const bool chain = (uint32_t)(in.cand_region_off[g + 1] - in.cand_region_off[g]) > prm.max_enum_snps;
      if (!chain && !waited) {   // an enumeration region: its winner is materialised on `stream`
        PCHK(hipEventRecord(ev_fork, stream));
        PCHK(hipStreamWaitEvent(side, ev_fork, 0));
        waited = true;
      }
      PCHK(k4_post_launch_grid(chain ? pinc : pin, ps, g, plut, side));
    }
  }
  lap("chain launch");
  pend.ng = ng; pend.any_host_post = any_host_post;
  pend.cand_off.assign(in.cand_region_off, in.cand_region_off + ng + 1); pend.host_post = host_post;
  pend.res_ps = res_ps; pend.res_tag = res_tag; pend.res_asg = res_asg; pend.hc_obj = hc_obj; pend.cand = in.cand;
  pending = true;
  // Everything is queued.  Without a reason to wait the call returns here: settle() collects the results when somebody asks.
  HT("ph:ret");
  if (g_lcr_host_trace) lcr_host_trace_flush();
  if (async_mode && !prof && !any_host_post && !grid_lock.held) return LCR_OK;
  PCHK(hipStreamSynchronize(side));
  lap("chain kernels");
  if (prof && chain_dev.dbg && !chain_desc.empty()) {   // steps of the last grid-scope chain launch
    long long clk[16];
    PCHK(hipMemcpy(clk, chain_dev.dbg, sizeof(clk), hipMemcpyDeviceToHost));
    static const char* nm[] = {"ordered column index", "pair table", "LD graph", "components + seed", "cross_optimize A", "block flip", "perturbation rounds"};
    fprintf(stderr, "[phase]     chain steps of %s\n", chain_desc.back().fast_lds ? "the last all-CU launch" : "the last launch (one-workgroup form: its first region)");
    for (int k = 0; k < 7; k++) fprintf(stderr, "[phase]     grid chain: %-24s %9.1f us\n", nm[k], (double)(clk[k + 1] - clk[k]) / 100.0);
    if (!chain_desc.back().fast_lds) fprintf(stderr, "[phase]     one-workgroup chain: %lld cross_optimize calls, %lld iterations\n", clk[14], clk[15]);
    if (!chain_desc.back().fast_lds) {
      static const char* nm3[] = {"setup", "sigma sweep", "row decisions", "delta sweep", "SNP decisions", "objective"};
      for (int k = 0; k < 6; k++) fprintf(stderr, "[phase]     one-workgroup cross_optimize: %-16s %8.1f us in total\n", nm3[k], (double)clk[8 + k] / 100.0);
    }
    if (chain_desc.back().fast_lds) {   // per-workgroup work time of the two half steps (own work, without the barriers)
      std::vector<long long> wg(2 * 1024);
      PCHK(hipMemcpy(wg.data(), chain_dev.dbg + 16, wg.size() * 8, hipMemcpyDeviceToHost));
      const int nb = std::max(1, k4_grid_blocks());
      for (int h = 0; h < 2; h++) {
        std::vector<std::pair<long long, int>> t;
        for (int k = 0; k < nb && k < 1024; k++) t.push_back({wg[h * 1024 + k], k});
        std::sort(t.begin(), t.end());
        const double it = (double)std::max<long long>(clk[15], 1) * 100.0;
        fprintf(stderr, "[phase]     grid chain rounds: %s step per workgroup and iteration: min %.1f (wg %d)  median %.1f  p90 %.1f  max %.1f us (wg %d)\n",
                h ? "delta" : "sigma", t.front().first / it, t.front().second, t[t.size() / 2].first / it, t[t.size() * 9 / 10].first / it, t.back().first / it, t.back().second);
        if (dbg.prof > 1) { for (int k = 0; k < nb && k < 1024; k++) fprintf(stderr, "%s%.1f", k % 16 ? " " : "\n[phase]       ", wg[h * 1024 + k] / it); fprintf(stderr, "\n"); }
      }
    }
    fprintf(stderr, "[phase]     grid chain: %lld iterations in the rounds\n", clk[15]);
    if (chain_desc.back().batch_lds && clk[14] && chain_dev.bt_ctl) {   // rows by their number of groups (BatchCtl::hist at byte 128)
      unsigned hist[256];
      PCHK(hipMemcpy(hist, (const uint8_t*)chain_dev.bt_ctl + 128, sizeof(hist), hipMemcpyDeviceToHost));
      std::string line;
      for (int l = 0; l < 256; l++) if (hist[l]) line += " " + std::to_string(l) + ":" + std::to_string(hist[l]);
      fprintf(stderr, "[phase]     grid chain rounds (batched): rows by groups%s\n", line.c_str());
    }
    if (chain_desc.back().batch_lds && clk[14]) fprintf(stderr, "[phase]     grid chain rounds (batched): %lld groups of four het-site entries incl. padding (%d rows, %d entries)\n", clk[14], stat[chain_desc.back().slot].R, stat[chain_desc.back().slot].E);
    static const char* nm2[] = {"stage delta/eta", "sigma step (workgroup 0)", "barrier 1", "stage sigma / invalidate", "delta step (workgroup 0)", "barrier 2"};
    for (int k = 0; k < 6; k++) fprintf(stderr, "[phase]     grid chain rounds: %-26s %9.1f us per iteration\n", nm2[k], (double)clk[8 + k] / 100.0 / (double)std::max<long long>(clk[15], 1));
  }
  { const int rc = settle(err); if (rc) return rc; }
  lap("enumeration kernels");
  uint8_t* const h_res = h_pin[7].as<uint8_t>();   // (the host epilogue below writes its regions' rows into the same pinned block)
  int8_t* const h_tag = (int8_t*)(h_res + res_tag); uint8_t* const h_asg = h_res + res_asg; uint32_t* const h_ps = (uint32_t*)(h_res + res_ps);
  if (prof && pin.dbg_clk) {   // steps of k4_post: the slowest workgroup of each kind of region, and the median total
    std::vector<long long> clk((size_t)ng * 16);
    PCHK(hipMemcpy(clk.data(), pin.dbg_clk, clk.size() * 8, hipMemcpyDeviceToHost));
    static const char* nm[] = {"stage rows", "stage entries + column index", "reads_hap", "snp_hap", "reads_hap + snp_hap", "rescue x2",
                               "reads_hap + snp_hap", "phase_set", "write back"};
    for (int chain = 0; chain < 2; chain++) {
      std::vector<std::pair<long long, int>> tot;
      for (int g = 0; g < ng; g++) {
        const int S = in.cand_region_off[g + 1] - in.cand_region_off[g];
        if (S == 0 || host_post[g] || grid_post[g] || ((uint32_t)S > prm.max_enum_snps) != (chain == 1)) continue;
        tot.push_back({clk[(size_t)g * 16 + 9] - clk[(size_t)g * 16], g});
      }
      if (tot.empty()) continue;
      std::sort(tot.begin(), tot.end());
      const int g = tot.back().second;
      fprintf(stderr, "[phase]   k4_post %s regions: %zu workgroups, median %.1f us, slowest %.1f us (region %d: %d rows, %d entries, %d SNPs)\n",
              chain ? "chain" : "enumeration", tot.size(), (double)tot[tot.size() / 2].first / 100.0, (double)tot.back().first / 100.0, g,
              in.row_region_off[g + 1] - in.row_region_off[g], stat[g].E_all, in.cand_region_off[g + 1] - in.cand_region_off[g]);
      for (int k = 0; k < 9; k++) fprintf(stderr, "[phase]     %-30s %7.1f us\n", nm[k], (double)(clk[(size_t)g * 16 + k + 1] - clk[(size_t)g * 16 + k]) / 100.0);
    }
  }
  if (prof && pin.dbg_clk)   // the same steps for the regions that took k4_gpost (all CUs on the region)
    {
      std::vector<long long> clk((size_t)ng * 16);
      PCHK(hipMemcpy(clk.data(), pin.dbg_clk, clk.size() * 8, hipMemcpyDeviceToHost));
      static const char* nm[] = {"stage rows + entries", "column index", "reads_hap", "snp_hap", "reads_hap + snp_hap", "rescue x2",
                                 "reads_hap + snp_hap", "phase_set", "write back"};
      for (int g = 0; g < ng; g++) {
        if (!grid_post[g]) continue;
        fprintf(stderr, "[phase]   k4_gpost region %d: %.1f us\n", g, (double)(clk[(size_t)g * 16 + 9] - clk[(size_t)g * 16]) / 100.0);
        for (int k = 0; k < 9; k++) fprintf(stderr, "[phase]     %-30s %7.1f us\n", nm[k], (double)(clk[(size_t)g * 16 + k + 1] - clk[(size_t)g * 16 + k]) / 100.0);
        fprintf(stderr, "[phase]     rescue rounds: RNA-edit list %lld, low-fraction list %lld\n", clk[(size_t)g * 16 + 11], clk[(size_t)g * 16 + 12]);
      }
    }
  if (!any_host_post) { lap("results"); return LCR_OK; }

  // ---- host epilogue (thread.rs:168-201) for the regions that did not take k4_post.  Regions are independent (the
  // reference runs them as rayon tasks, thread.rs:77): a small host thread pool walks them; results do not depend
  // on the thread count (per-region RNG stream, disjoint output rows).
  PCHK(h_pin[4].reserve(st_bytes + 16)); PCHK(h_pin[6].reserve(st_bytes + 16));
  int8_t* const st1 = h_pin[4].as<int8_t>();   // enumeration results
  int8_t* const st2 = h_pin[6].as<int8_t>();   // chain results
  PCHK(hipMemcpyAsync(st1, b_st.p, st_bytes, hipMemcpyDeviceToHost, stream));
  PCHK(hipMemcpyAsync(st2, b_stc.p, st_bytes, hipMemcpyDeviceToHost, stream));
  PCHK(hipStreamSynchronize(stream));
  if (!work) work = new PhaseWork();
  PhaseWork& W = *static_cast<PhaseWork*>(work);
  if ((int)W.R.size() < ng) W.R.resize(ng);
  std::vector<RegionHost>& R = W.R;
  auto epilogue = [&](int g) {
    if (!host_post[g]) return;
    RegionHost& rh = R[g];
    rh.g = g; rh.c0 = in.cand_region_off[g]; rh.S = in.cand_region_off[g + 1] - rh.c0;
    rh.r0 = in.row_region_off[g]; rh.nrow = in.row_region_off[g + 1] - rh.r0;
    rh.row_ptr = h_pin[0].as<int64_t>(); rh.col = h_pin[1].as<int32_t>(); rh.val = h_pin[2].as<uint8_t>(); rh.links = h_pin[3].as<uint32_t>();
    rh.cand = cand.data() + rh.c0;
    rh.seed = region_seed(prm.seed, in.region_start0[g]);
    rh.min_linkers = prm.min_linkers;
    rh.e0 = rh.row_ptr[rh.r0];
    const bool chain = (uint32_t)rh.S > prm.max_enum_snps;
    const int64_t e1 = rh.row_ptr[rh.r0 + rh.nrow];
    rh.phase_site.assign((size_t)(e1 - rh.e0), 0);
    rh.tag.assign(rh.nrow, 0); rh.asg.assign(rh.nrow, 0); rh.fp.assign(rh.nrow, 0);
    if ((int)rh.cover.size() < rh.S) rh.cover.resize(rh.S);
    for (int i = 0; i < rh.S; i++) rh.cover[i].clear();
    rh.fp_rows.clear();
    rh.orig_flags.resize(rh.S);
    for (int i = 0; i < rh.S; i++) rh.orig_flags[i] = rh.cand[i].flags;
    for (int r = 0; r < rh.nrow; r++) {
      for (int64_t e = rh.eb(r); e < rh.ee(r); e++) {
        const int i = rh.lc(e);
        rh.cover[i].push_back(r);
        if (rh.fphase(i)) rh.phase_site[e - rh.e0] = 1;  // fragment.rs:144-146 snapshot
      }
      if (rh.links[rh.r0 + r] >= prm.min_linkers) { rh.fp[r] = 1; rh.fp_rows.push_back(r); }   // fragment.rs:253-255
    }
    // the optimiser's result
    const int8_t* st = chain ? st2 : st1;
    const int8_t* sg = st + st_sig + rh.r0; const int8_t* dl = st + st_del + rh.c0; const int8_t* et = st + st_eta + rh.c0;
    for (size_t k = 0; k < rh.fp_rows.size(); k++) rh.tag[rh.fp_rows[k]] = sg[k];
    for (int i = 0; i < rh.S; i++) { rh.cand[i].haplotype = dl[i]; rh.cand[i].genotype = et[i]; }
    objective[g] = (double)(*((const long long*)(st + st_obj) + g)) / FX_SCALE;
    // draws so far: thread.rs:162-163 (S + F, overwritten), then the optimiser's (see k4_post)
    const uint64_t S = rh.S, F = rh.fp_rows.size();
    rh.ctr = !chain ? S + F + ((uint64_t)1 << S) * F : 2 * (S + F) + (S / 4 + 1) * (S + F);
    rh.assign_reads_haplotype(prm.read_assign_cutoff);
    rh.assign_snp_haplotype_genotype();
    rh.assign_reads_haplotype(prm.read_assign_cutoff);
    rh.assign_snp_haplotype_genotype();
    const float relaxed = prm.min_phase_score - 3.0f;
    rh.eval_rescue(LCR_F_RNA_EDIT, relaxed, false);
    rh.eval_rescue(LCR_F_CAND_SOMATIC, relaxed, true);
    rh.assign_reads_haplotype(prm.read_assign_cutoff);
    rh.assign_snp_haplotype_genotype();
    for (int r = 0; r < rh.nrow; r++) h_ps[rh.r0 + r] = 0;
    rh.assign_phase_set(prm.min_phase_score, h_ps);
    for (int r = 0; r < rh.nrow; r++) { h_tag[rh.r0 + r] = rh.tag[r]; h_asg[rh.r0 + r] = rh.asg[r]; }
  };
  pool->parallel_for(ng, epilogue);
  lap("host post-phase epilogue");
  // keep the device copy of the candidates current (lcr_get_candidates_device)
  for (int g = 0; g < ng; g++) {
    const int c0 = in.cand_region_off[g], S = in.cand_region_off[g + 1] - c0;
    if (S && host_post[g]) PCHK(hipMemcpyAsync(const_cast<lcr_candidate*>(in.d_cand) + c0, cand.data() + c0, (size_t)S * sizeof(lcr_candidate), hipMemcpyHostToDevice, stream));
  }
  PCHK(hipStreamSynchronize(stream));
  return LCR_OK;
#undef PCHK
}
