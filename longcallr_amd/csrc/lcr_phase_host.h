// lcr_phase_host.h — host driver of the phasing stage (K4 kernels + sequential control).
#pragma once
#include "lcr_dev.h"

struct PhaseInputs {
  int32_t n_regions = 0, n_rows = 0;
  int64_t nnz = 0;
  const int32_t* row_region_off = nullptr;   // host, n_regions+1
  const int32_t* cand_region_off = nullptr;  // host, n_regions+1
  const int64_t* region_start0 = nullptr;    // host
  const int64_t* d_row_ptr = nullptr;        // device CSR
  const int32_t* d_col = nullptr;
  const uint8_t* d_val = nullptr;
  const uint32_t* d_row_links = nullptr;
  std::vector<lcr_candidate>* cand = nullptr;  // host candidates, updated in place
};

struct PhaseHost {
  std::vector<int8_t> haplotag;
  std::vector<uint8_t> assignment;
  std::vector<uint32_t> phase_set;
  std::vector<double> objective;
  DevBuf d_state[12];
  int run(const PhaseInputs& in, const lcr_params& p, hipStream_t s, std::string* err);
  void release() { for (auto& b : d_state) b.release(); }
};
