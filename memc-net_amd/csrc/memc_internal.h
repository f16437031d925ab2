// memc_internal.h -- private to the library's sources and its measurement scripts; NOT part of the drop-in ABI
// (that is include/memc_warp.h).
//
// The memc_debug_* hooks below exist ONLY in the measurement build (make measure -> lib/libmemc_hip_measure.so,
// compiled with -DMEMC_MEASURE).  The product library libmemc_hip.so neither defines nor exports them, carries none of
// the ablation kernels they select (several of which return wrong results by construction: they time one phase of
// a kernel) and reads nothing from the environment.  tools/measure.py binds the measurement library; tests that
// force a particular kernel path use it too.
#pragma once
#include "memc_warp.h"

#ifdef MEMC_MEASURE
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

// A/B measurement hooks.  variant < 0 restores automatic selection.
void memc_debug_set_fi_fwd_variant(int variant);
void memc_debug_set_projection_variant(int variant);
void memc_debug_set_fi_bwd_variant(int variant);
void memc_debug_set_extra_lds(int bytes);        // bilinear forward only: pads its LDS request (fewer workgroups per CU)
void memc_debug_set_bl_cap(int which);           // 2x2-footprint kernels' LDS staging budget: < 0 each kernel's default, 0 = 48 KiB, 1 = 39 KiB, 2 = 31 KiB (bilinear forward only)
void memc_debug_set_walk(int stripe_width);      // < 0: each launcher's default; 0: strips; n: stripes n tile columns wide
int memc_debug_set_trace_buffer(void *device_u64_buffer);        // gridDim.x * 16 slots, written by fi_bwd variant 9
int memc_debug_set_trace_buffer_proj(void *device_u64_buffer);   // the same for the projection's trace arm

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
