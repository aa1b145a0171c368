// lcr_phase_host.h — host driver of the phasing stage (K4 kernels + sequential control).
#pragma once
#include <string>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <semaphore.h>
#include <thread>

#include "lcr_dev.h"
#include "k4_types.h"

struct PhaseInputs {
  int32_t n_regions = 0, n_rows = 0;
  int64_t nnz = 0;
  const int32_t* row_region_off = nullptr;   // host, n_regions+1
  const int32_t* cand_region_off = nullptr;  // host, n_regions+1
  const int64_t* region_start0 = nullptr;    // host
  const int64_t* region_e_off = nullptr;     // host, n_regions+1: first entry of every region in the fragment matrix
  const int64_t* d_row_ptr = nullptr;        // device CSR
  const int32_t* d_col = nullptr;
  const uint8_t* d_val = nullptr;
  const uint32_t* d_row_links = nullptr;
  std::vector<lcr_candidate>* cand = nullptr;  // host candidates, updated in place
  const lcr_candidate* d_cand = nullptr;     // device copy of the same candidates (as K3 saw them)
  const int32_t* d_cand_off = nullptr;       // device, n_regions+1
  const int32_t* d_row_region_off = nullptr; // device, n_regions+1
  const int64_t* d_start0 = nullptr;         // device, region start columns
};

// Persistent host worker pool: regions are independent units of host-side work (the reference runs
// them as rayon tasks, thread.rs:77); parallel_for hands out region indices through an atomic counter.
// Only as many workers as there are items are woken (one semaphore each: a broadcast on a condition variable
// made every worker take the mutex in turn, ~3 us apiece, whatever the number of items), the caller takes a
// share of the items itself, and the last worker to finish posts the completion semaphore.
class HostPool {
 public:
  explicit HostPool(int n) {
    sem_init(&done_, 0, 0);
    for (int i = 0; i < n; i++) {
      w_.emplace_back(new Worker());
      sem_init(&w_.back()->go, 0, 0);
    }
    for (int i = 0; i < n; i++) w_[i]->t = std::thread([this, i]() { loop(i); });
  }
  ~HostPool() {
    stop_.store(true);
    for (auto& w : w_) sem_post(&w->go);
    for (auto& w : w_) { w->t.join(); sem_destroy(&w->go); }
    sem_destroy(&done_);
  }
  int size() const { return (int)w_.size(); }
  void parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (w_.empty() || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
    const int k = std::min(n - 1, (int)w_.size());
    fn_ = &fn; n_ = n;
    next_.store(0, std::memory_order_relaxed);
    pending_.store(k, std::memory_order_release);
    for (int i = 0; i < k; i++) sem_post(&w_[i]->go);
    for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) fn(i);
    while (sem_wait(&done_) != 0) {}   // (EINTR)
    fn_ = nullptr;
  }
 private:
  struct Worker { std::thread t; sem_t go; };
  void loop(int w) {
    for (;;) {
      while (sem_wait(&w_[w]->go) != 0) {}
      if (stop_.load()) return;
      const std::function<void(int)>* fn = fn_;
      const int n = n_;
      for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) (*fn)(i);
      if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) sem_post(&done_);
    }
  }
  std::vector<std::unique_ptr<Worker>> w_;
  sem_t done_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0;
  std::atomic<int> next_{0}, pending_{0};
  std::atomic<bool> stop_{false};
};

// One persistent helper thread per context: runs a job beside the calling thread (creating a thread per call costs
// tens of microseconds at the head of the chain regions' critical path).
class HelperThread {
 public:
  HelperThread() {
    sem_init(&go_, 0, 0); sem_init(&done_, 0, 0);
    t_ = std::thread([this]() {
      for (;;) {
        while (sem_wait(&go_) != 0) {}
        if (stop_) return;
        job_();
        sem_post(&done_);
      }
    });
  }
  ~HelperThread() { stop_ = true; sem_post(&go_); t_.join(); sem_destroy(&go_); sem_destroy(&done_); }
  void start(std::function<void()> job) { job_ = std::move(job); busy_ = true; sem_post(&go_); }
  void join() { if (busy_) { while (sem_wait(&done_) != 0) {} busy_ = false; } }
 private:
  std::thread t_;
  sem_t go_, done_;
  std::function<void()> job_;
  bool busy_ = false;
  std::atomic<bool> stop_{false};
};

// Debug / test switches of the phase stage (lcr_debug_set, include/lcr.h): none of them is read from the environment inside
// the library; the defaults are the product behaviour.
struct PhaseDebug {
  int prof = 0;                 // "phase_prof": per-step timers to stderr (2: also the per-workgroup histogram)
  int post_host = 0;            // "post_host": every region through the host epilogue (cross-check of k4_post)
  long long grid_min = -1;      // "grid_min_entries": chain regions with at least this many phase entries get all CUs (-1: 2^17)
  int grid_generic = 0;         // "grid_generic": fenced grid barriers only
  int post_half = 0;            // "post_half": the eight-wave epilogue of the chain regions
  int enum_force_big = 0;       // "enum_force_big" / "enum_force_stream": the fallback enumeration kernels
  int enum_force_stream = 0;    // (2: the large-image launch of the streaming kernel)
  int enum_bits = 1;            // "enum_bits": the enumeration restarts of the LDS classes eight per wave as bit states (k4_enum_bits)
  int spec_batch = 1;           // "grid_spec_batch": eight speculative half-rounds per pass over the matrix (k4_grid_batch.h); 0: the side-by-side lanes below
  int spec_lanes = 8;           // "grid_spec_lanes": half-rounds of the perturbation loop run at once at grid scope (1: one after the other; C5 with packed entries: 454 / 370 / 348 / 366 ms with 2 / 4 / 8 / 16 -- eight lanes = one XCD each)
  int phase_prio = 0;           // "phase_prio" (measurement switch): 1 = the stage's own queues are created at the device's greatest priority
  int no_gate = 0;              // "no_gate" (measurement switch): 1 = the next batch's K0 does not wait for the restarts of the stage in flight
  int redo_lds = 64 * 1024;     // "redo_lds": bytes of dynamic LDS of the enumeration branch's repair pass (k4_enum_redo: state + matrix of a restart's region where they fit; 0: global memory)
  int chain_ties = 1;           // "chain_ties": chain regions of workgroup scope that meet a class-2 / class-4 tie run again under the complete tie contract (0: counted as unresolved)
  int tie_arith = 3;            // "tie_arith": which exact fixed-point ties the reference-order f64 arithmetic decides (PhaseDev::tie_arith; 3 = all that liblcr resolves)
  int host_threads = 0;         // "host_threads": size of the host pool of the host epilogue (0: hardware threads / devices, <= 48)
  int async_phase = 0;          // "async_phase": lcr_phase returns when its kernels are queued (on a queue of its own); settle() collects the results
};

struct PhaseHost {
  PhaseDebug dbg;
  std::string lock_dir;          // directory of the per-GPU lock file of persistent launches ("" = /tmp/liblcr-<uid>)
  std::vector<int8_t> haplotag;
  std::vector<uint8_t> assignment;
  std::vector<uint32_t> phase_set;
  std::vector<double> objective;
  const int8_t* r_haplotag = nullptr;      // results of the last run (host vectors above or pinned buffers)
  const uint8_t* r_assignment = nullptr;
  const uint32_t* r_phase_set = nullptr;
  DevBuf d_state[40];
  DevBuf d_tie_flag, d_tie_q, d_tie_ch, d_tie_terms;
  DevBuf d_rbest_buf;   // enumeration branch: best objective seen per region (filled before the staging kernel)   // k4_chain_wg: regions that met a class-2 / class-4 tie, scratch of their second run
  DevBuf d_spec_sig, d_spec_de, d_spec_res, d_pk, d_bt;   // working states / results of the speculative half-rounds (k4_grid.hip)
  DevBuf d_read_rec;             // per-row results as 12-byte records in HBM, written by k4_post
  DevBuf d_lut64, d_tie, d_enum_st;   // f64 table of the tie paths (PostLut), census counters, final states of the enumeration restarts
  bool lut64_ready = false;
  unsigned long long tie_census[TIE_NCTR] = {0, 0, 0, 0, 0, 0, 0, 0};   // of the last run (lcr_get_tie_census)
  bool read_rec_stale = false;   // some regions took the host epilogue: the records are rebuilt from the host arrays on demand
  HostBuf h_pin[12];   // pinned staging: row_ptr, col, val, links, enum state, region sizes, chain state, results, job tables, chain start
  hipStream_t side = nullptr;   // second queue: fragment matrix download + chain regions
  hipEvent_t ev_in = nullptr, ev_csr = nullptr, ev_fork = nullptr, ev_join = nullptr;
  hipStream_t aux = nullptr;   // enumeration classes 3 / 4 beside class 2
  // lcr_debug_set("async_phase", 1) (round 5, opt-in): the stage's FIRST queue is its own too, lcr_phase returns when everything is
  // queued, the caller's stream is free for the next batch's lcr_load_batch / lcr_pileup, and the results are collected by
  // settle(): every getter, lcr_ctx_sync, the next lcr_candidates / lcr_phase call it.  Persistent all-CU launches (device
  // lock), the host epilogue and phase_prof settle before run() returns.  Default: the caller's stream, settle() inside run().
  hipStream_t main_q = nullptr, q_first = nullptr;   // q_first: the queue the last run() used as its first
  // async_phase: the next batch's pileup is gated on these -- recorded behind the enumeration restarts on the stage's first queue and
  // on `aux`: the dense part of the stage.  What follows them (repair pass, resolve, post-phase: a few hundred workgroups) leaves
  // most CUs idle, and that is where the next pileup's kernels run -- beside the restarts they would only time-share the VALUs.
  hipEvent_t ev_gate[2] = {nullptr, nullptr};
  bool gate_set[2] = {false, false};
  // makes `s` wait for the dense part of a stage in flight (no-op otherwise)
  hipError_t gate_stream(hipStream_t s) {
    if (!pending || dbg.no_gate) return hipSuccess;
    for (int k = 0; k < 2; k++) if (gate_set[k]) { hipError_t e = hipStreamWaitEvent(s, ev_gate[k], 0); if (e != hipSuccess) return e; }
    return hipSuccess;
  }
  hipEvent_t ev_user = nullptr;
  bool pending = false;
  struct Pending {
    int ng = 0; bool any_host_post = false;
    std::vector<int32_t> cand_off; std::vector<uint8_t> host_post;
    size_t res_ps = 0, res_tag = 0, res_asg = 0, hc_obj = 0;
    std::vector<lcr_candidate>* cand = nullptr;
  } pend;
  int settle(std::string* err);
  HostPool* pool = nullptr;
  HelperThread* helper_thread = nullptr;
  void* work = nullptr;   // PhaseWork (k4_phase.hip): per-region host state reused across calls
  ChainDev chain_dev{};                // chain-region buffers of the last run (LD blocks are read back from them)
  std::vector<ChainDesc> chain_desc;
  std::vector<uint64_t> enum_keys;                   // scratch of the enumeration launch preparation, kept across calls
  std::vector<int64_t> enum_job_base, enum_st_base;
  // LD blocks of one region of the last run in the reference's order (candidate.rs:733-745): off[n_blocks + 1], SNP indices
  int ld_blocks(const PhaseInputs& in, int region, std::vector<int32_t>* off, std::vector<int32_t>* snps, hipStream_t s, std::string* err);
  void free_work();
  int run(const PhaseInputs& in, const lcr_params& p, hipStream_t s, std::string* err);
  void release() {
    for (auto& b : d_state) b.release();
    d_lut64.release(); d_tie.release(); d_enum_st.release(); lut64_ready = false;
    d_read_rec.release(); d_spec_sig.release(); d_spec_de.release(); d_spec_res.release(); d_pk.release(); d_bt.release();
    for (auto& b : h_pin) b.release();
    if (side) { (void)hipStreamDestroy(side); side = nullptr; }
    if (ev_in) { (void)hipEventDestroy(ev_in); ev_in = nullptr; }
    if (ev_csr) { (void)hipEventDestroy(ev_csr); ev_csr = nullptr; }
    if (ev_fork) { (void)hipEventDestroy(ev_fork); ev_fork = nullptr; }
    if (ev_join) { (void)hipEventDestroy(ev_join); ev_join = nullptr; }
    if (aux) { (void)hipStreamDestroy(aux); aux = nullptr; }
    if (main_q) { (void)hipStreamDestroy(main_q); main_q = nullptr; }
    if (ev_user) { (void)hipEventDestroy(ev_user); ev_user = nullptr; }
    for (int k = 0; k < 2; k++) { if (ev_gate[k]) (void)hipEventDestroy(ev_gate[k]); ev_gate[k] = nullptr; gate_set[k] = false; }
    pending = false;
    delete helper_thread; helper_thread = nullptr;
    delete pool; pool = nullptr;
    free_work();
  }
};
