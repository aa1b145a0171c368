// k4_grid.h — launchers of the chain-region kernels (k4_grid.hip) for the host control in k4_phase.hip.
#pragma once
#include "k4_dev.h"

// one workgroup per chain region: regions desc[first .. first + n)
hipError_t k4_chain_launch_wg(const ChainDev& C, int first, int n, size_t dyn_lds, hipStream_t s);
// all CUs on the single region desc[which] (persistent launch with grid barriers; C.ctl is reset here)
hipError_t k4_chain_launch_grid(const ChainDev& C, int which, size_t dyn_lds, hipStream_t s);
// dynamic LDS the device-coherent perturbation rounds may use (sigma bit vector + delta / eta / het-delta bytes of the region)
constexpr int K4_GRID_FAST_LDS_MAX = 96 * 1024;
inline size_t k4_grid_fast_lds(int64_t R, int64_t S) { return (size_t)(8 * ((R + 63) / 64) + 3 * S + 64); }
// the batched rounds (k4_grid_batch.h): spread masks of every SNP + team slots + set-up histogram
constexpr int K4_GRID_BATCH_MAX_WG = 512;
constexpr int K4_GRID_BATCH_CTL_BYTES = 128 + 2048 + 128 + K4_GRID_BATCH_MAX_WG * 128;
inline size_t k4_grid_batch_lds(int64_t S) { return (size_t)(12 * (S + 2) + 24 * 1024); }
// workgroups of a grid launch (co-resident by construction); 0 = no device
int k4_grid_blocks();
// k4_stage for one large region with all CUs; blk_tot: 2 * k4_grid_blocks() + 1 int32 of scratch
hipError_t k4_stage_launch_grid(const StageIn& in, const StageOut& out, const PhaseLutDev& lut, int g, GridCtl* ctl, int32_t* blk_tot, hipStream_t s);
// post-phase steps for one large region with all CUs
hipError_t k4_post_launch_grid(const PostIn& post_in, const PostScratch& ps, int g, const PostLut& lut, hipStream_t s);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel slot 0..7, device)
hipError_t k4_set_dyn_lds_once(const void* fn, int bytes, int slot);
