// k4_grid.h — launchers of the chain-region kernels (k4_grid.hip) for the host control in k4_phase.hip.
#pragma once
#include "k4_dev.h"

// one workgroup per chain region: regions desc[first .. first + n)
hipError_t k4_chain_launch_wg(const ChainDev& C, int first, int n, size_t dyn_lds, hipStream_t s);
// all CUs on the single region desc[which] (persistent launch with grid barriers; C.ctl is reset here)
hipError_t k4_chain_launch_grid(const ChainDev& C, int which, hipStream_t s);
// workgroups of a grid launch (co-resident by construction); 0 = no device
int k4_grid_blocks();
