// k4_phase.hip — placeholder until the phasing kernels land (next commit).
#include "lcr_phase_host.h"
int PhaseHost::run(const PhaseInputs&, const lcr_params&, hipStream_t, std::string* err) {
  if (err) *err = "lcr_phase: not implemented yet";
  return LCR_E_STATE;
}
