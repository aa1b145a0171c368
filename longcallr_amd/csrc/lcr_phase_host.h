// lcr_phase_host.h — host driver of the phasing stage (K4 kernels + sequential control).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>

#include "lcr_dev.h"

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
class HostPool {
 public:
  explicit HostPool(int n) {
    for (int i = 0; i < n; i++) workers_.emplace_back([this]() { loop(); });
  }
  ~HostPool() {
    { std::lock_guard<std::mutex> l(m_); stop_ = true; gen_++; }
    cv_.notify_all();
    for (auto& w : workers_) w.join();
  }
  int size() const { return (int)workers_.size(); }
  void parallel_for(int n, const std::function<void(int)>& fn) {
    if (n <= 0) return;
    if (workers_.empty() || n == 1) { for (int i = 0; i < n; i++) fn(i); return; }
    { std::lock_guard<std::mutex> l(m_); fn_ = &fn; n_ = n; next_.store(0); pending_ = (int)workers_.size(); gen_++; }
    cv_.notify_all();
    std::unique_lock<std::mutex> l(m_);
    done_.wait(l, [this]() { return pending_ == 0; });
    fn_ = nullptr;
  }
 private:
  void loop() {
    unsigned long seen = 0;
    for (;;) {
      const std::function<void(int)>* fn;
      int n;
      { std::unique_lock<std::mutex> l(m_);
        cv_.wait(l, [&]() { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
        fn = fn_; n = n_; }
      for (int i = next_.fetch_add(1); i < n; i = next_.fetch_add(1)) (*fn)(i);
      { std::lock_guard<std::mutex> l(m_); if (--pending_ == 0) done_.notify_all(); }
    }
  }
  std::vector<std::thread> workers_;
  std::mutex m_;
  std::condition_variable cv_, done_;
  const std::function<void(int)>* fn_ = nullptr;
  int n_ = 0, pending_ = 0;
  unsigned long gen_ = 0;
  bool stop_ = false;
  std::atomic<int> next_{0};
};

struct PhaseHost {
  std::vector<int8_t> haplotag;
  std::vector<uint8_t> assignment;
  std::vector<uint32_t> phase_set;
  std::vector<double> objective;
  const int8_t* r_haplotag = nullptr;      // results of the last run (host vectors above or pinned buffers)
  const uint8_t* r_assignment = nullptr;
  const uint32_t* r_phase_set = nullptr;
  DevBuf d_state[21];
  HostBuf h_pin[11];   // pinned staging: row_ptr, col, val, links, enum state, region sizes, chain state, results, job tables, chain start
  hipStream_t side = nullptr;   // second queue: fragment matrix download + chain regions
  hipEvent_t ev_in = nullptr, ev_csr = nullptr, ev_fork = nullptr, ev_join = nullptr;
  hipStream_t aux = nullptr;   // enumeration classes 3 / 4 beside class 2
  HostPool* pool = nullptr;
  void* work = nullptr;   // PhaseWork (k4_phase.hip): per-region host state reused across calls
  void free_work();
  int run(const PhaseInputs& in, const lcr_params& p, hipStream_t s, std::string* err);
  void release() {
    for (auto& b : d_state) b.release();
    for (auto& b : h_pin) b.release();
    if (side) { (void)hipStreamDestroy(side); side = nullptr; }
    if (ev_in) { (void)hipEventDestroy(ev_in); ev_in = nullptr; }
    if (ev_csr) { (void)hipEventDestroy(ev_csr); ev_csr = nullptr; }
    if (ev_fork) { (void)hipEventDestroy(ev_fork); ev_fork = nullptr; }
    if (ev_join) { (void)hipEventDestroy(ev_join); ev_join = nullptr; }
    if (aux) { (void)hipStreamDestroy(aux); aux = nullptr; }
    delete pool; pool = nullptr;
    free_work();
  }
};
