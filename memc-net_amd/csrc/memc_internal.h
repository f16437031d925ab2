// memc_internal.h -- private to libmemc_hip.so and its measurement scripts; NOT part of the drop-in ABI
// (that is include/memc_warp.h).
#pragma once
#include "memc_warp.h"

#ifdef __cplusplus
extern "C" {
#endif

// A/B measurement hooks (tools/bench_ops.py).  variant < 0 restores automatic selection.
// The fi_fwd / projection hooks change which of several equivalent kernels a launcher picks (ablation arms,
// documented at each kernel, excepted).
void memc_debug_set_fi_fwd_variant(int variant);
void memc_debug_set_projection_variant(int variant);
void memc_debug_set_fi_bwd_variant(int variant);
void memc_debug_set_extra_lds(int bytes);        // bilinear forward only: pads its LDS request (fewer workgroups per CU)
void memc_debug_set_bl_cap(int which);           // 2x2-footprint kernels' LDS staging budget: < 0 each kernel's default, 0 = 48 KiB, 1 = 39 KiB, 2 = 31 KiB (bilinear forward only)
void memc_debug_set_walk(int stripe_width);     // < 0: each launcher's default; 0: strips; n: stripes n tile columns wide
int memc_debug_set_trace_buffer(void *device_u64_buffer);   // gridDim.x * 16 slots, written by fi_bwd variant 9      // > 0: ablation arms, results deliberately wrong

#ifdef __cplusplus
}
#endif
