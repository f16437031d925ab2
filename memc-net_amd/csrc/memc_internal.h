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

#if defined(__cplusplus) && defined(__HIPCC__)
#include <hip/hip_runtime.h>
namespace memc {
// fi_bwd_cn.hip: FilterInterpolation backward for C >= 4 (a ragged last chunk is padded), fs == 4 (tap-gradient kernel +
// owner-computes image gradient).  1: taken, 0: not taken (the caller falls back to the direct kernel; for this class of channel counts
// gradinput1 has then been cleared: it is STORED on every path), -1: launch error.  Strides as in the C ABI.
// force_direct: measurement arm -- clear and decline.
int fi_bwd_cn_launch(hipStream_t stream, int w, int h, int channel, int batch,
                     int s1b, int s1c, int s1h, int s2b, int s2c, int s2h, int s3b, int s3c, int s3h,
                     const float *input1, const float *input2, const float *input3, const float *gradoutput,
                     float *gradinput1, float *gradinput2, float *gradinput3, bool force_direct);
// the class of channel counts fi_bwd_cn.hip takes -- and for which gradinput1 is stored on every path
bool fi_bwd_cn_class(int channel, int filter_size);
#ifdef MEMC_MEASURE
extern bool g_bwd_cn_allow_c3;               // arm (bl_cap 5): the bilinear warp's RGB backward through the owner kernels
#endif
// fi_bwd_c3.hip: the same operator for RGB (C == 3), fs == 4, LDS-tiled.  1: taken, 0: geometry not 16-byte aligned
// (the caller takes the direct kernel), -1: launch error.  variant: measurement arm (-1 in the product).
int fi_bwd_c3_launch(hipStream_t stream, int w, int h, int batch,
                     int s1b, int s1c, int s1h, int s2b, int s2c, int s2h, int s3b, int s3c, int s3h,
                     const float *input1, const float *input2, const float *input3, const float *gradoutput,
                     float *gradinput1, float *gradinput2, float *gradinput3, int variant);
#ifdef MEMC_MEASURE
// arms/fi_bwd_c3_arms.hip (measurement build only): the kernels of rounds 1-2 and their ablation arms
int fi_bwd_c3_arm_launch(int variant, hipStream_t stream, int w, int h, int ntx, int nty, int batch,
                         int s1b, int s1c, int s1h, int s2b, int s2c, int s2h, int s3b, int s3c, int s3h,
                         const float *input1, const float *input2, const float *input3, const float *gradoutput,
                         float *gradinput1, float *gradinput2, float *gradinput3);
int fi_bwd_c3_arms_set_trace_buffer(unsigned long long *device_buffer);
#endif
// ... and the bilinear warp's backward (Interpolation / InterpolationCh) for the same class of channel counts
int bl_bwd_cn_launch(hipStream_t stream, int w, int h, int channel, int batch,
                     int s1b, int s1c, int s1h, int s2b, int s2c, int s2h,
                     const float *input1, const float *input2, const float *gradoutput,
                     float *gradinput1, float *gradinput2, bool force_direct);
}  // namespace memc
#endif

// Which kernel family a launcher chose: recorded per host thread, in both builds; memc_last_kernel_path()
// (include/memc_warp.h) hands it to the caller, so that a C-ABI user sees when a shape fell to the scalar / direct kernels.
#ifdef __cplusplus
namespace memc {
extern thread_local const char *t_last_path;
}
#define MEMC_PATH(name) (memc::t_last_path = (name))
#endif

#ifdef MEMC_MEASURE
#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)

// A/B measurement hooks.  variant < 0 restores automatic selection.
void memc_debug_set_fi_fwd_variant(int variant);
void memc_debug_set_fi_phase(int period_times_65536_plus_window);   /* arm 26: the chip-wide write window, ticks of 10 ns */
void memc_debug_set_projection_variant(int variant);
void memc_debug_set_projection_scratch_blocks(int n);   // (Depth)FlowProjection forward: cached scratch blocks a call may look at (1 .. 8)
void memc_debug_set_projection_stall_us(int us);        // ... idle this long between the owner kernel and the kernels behind it
void memc_debug_set_fi_bwd_variant(int variant);
void memc_debug_set_extra_lds(int bytes);        // bilinear forward only: pads its LDS request (fewer workgroups per CU)
void memc_debug_set_bl_cap(int which);           // 2x2-footprint kernels' LDS staging budget: < 0 each kernel's default, 0 = 48 KiB, 1 = 39 KiB, 2 = 31 KiB (bilinear forward only); RGB backward, packed planes: 3 / 4 = 64 x 16 tiles on 256 lanes in 39 / 48 KiB
void memc_debug_set_walk(int stripe_width);      // < 0: each launcher's default; 0: strips; n: stripes n tile columns wide
int memc_debug_set_trace_buffer(void *device_u64_buffer);        // gridDim.x * 16 slots, written by fi_bwd variant 9
int memc_debug_set_trace_buffer_proj(void *device_u64_buffer);   // the same for the projection's trace arm
void memc_debug_set_bl_bwd_direct(int on);       // bilinear backward: 1 = the direct kernel for any channel count
int memc_debug_set_trace_buffer_cn(void *device_u64_buffer);
const char *memc_debug_last_path(void);          // == memc_last_kernel_path() (kept for the round-2/3 tools)

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
