// interpolation.hip -- plain bilinear backward-warp (Interpolation / InterpolationCh), forward and
// backward, for gfx950.
//
// Replaces my_package/src/my_lib_kernel.cu:507-793 of the reference (kernels :507,:578, launchers :687,:740).
// InterpolationCh reuses the same kernels, exactly as the reference's glue does (my_lib_cuda.c:519,579: the
// Ch entry points call the InterpolationLayer_*_kernel launchers; the dedicated .cu copies :797-1083 are
// unused); the `channel == 3` restriction of Interpolation lives in the layer entry point.
// Semantics: SURVEY.md appendix A.5.
//
// One lane = one output site, one wave = 64 consecutive sites of an image row (coalesced 256-B flow reads
// and output writes); the four corner gathers of neighbouring lanes fall into the same few cache lines.
#include "memc_common.hpp"
#include "memc_internal.h"

namespace memc {

template <int CT>
__global__ __launch_bounds__(256) void bl_fwd(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h,
    const float *__restrict__ in1, const float *__restrict__ flow, float *__restrict__ out)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = ld_stream(flow_b);
    const float fy = ld_stream(flow_b + s2c);
    const BlSite s = bl_locate<true>(x, y, W, H, fx, fy);
    const float *in_b = in1 + b * s1b;
    float *out_p = out + b * s1b + (int64_t)y * s1h + x;
    const int nc = CT > 0 ? CT : C;
    constexpr int kUnroll = CT > 0 ? CT : 4;
    if (s.valid) {
        const int oTL = s.T * s1h + s.L, oTR = s.T * s1h + s.R;
        const int oBL = s.Bm * s1h + s.L, oBR = s.Bm * s1h + s.R;
        const float w00 = (1 - s.a) * (1 - s.b), w01 = s.a * (1 - s.b);
        const float w10 = (1 - s.a) * s.b, w11 = s.a * s.b;
#pragma unroll kUnroll
        for (int c = 0; c < nc; c++) {
            const float *p = in_b + c * s1c;
            st_stream(out_p + c * s1c, w00 * p[oTL] + w01 * p[oTR] + w10 * p[oBL] + w11 * p[oBR]);
        }
    } else {
        for (int c = 0; c < nc; c++) st_stream(out_p + c * s1c, 0.0f);   // my_lib_kernel.cu:566-570
    }
}

template <int CT>
__global__ __launch_bounds__(256) void bl_bwd(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ gout,
    float *__restrict__ gin1, float *__restrict__ gin2)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = ld_stream(flow_b);
    const float fy = ld_stream(flow_b + s2c);
    const BlSite s = bl_locate<true>(x, y, W, H, fx, fy);
    if (!s.valid) return;                                   // buffers keep the caller's zeros
    const float x2 = (float)x + fx, y2 = (float)y + fy;
    const int oTL = s.T * s1h + s.L, oTR = s.T * s1h + s.R;
    const int oBL = s.Bm * s1h + s.L, oBR = s.Bm * s1h + s.R;
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    const float *gout_p = gout + b * s1b + (int64_t)y * s1h + x;
    const float gam_x = (float)s.Bm - y2;                    // clamped corner, my_lib_kernel.cu:634
    const float gam_y = (float)s.R - x2;                     // :652
    float botx = 0.0f, boty = 0.0f;
    const int nc = CT > 0 ? CT : C;
    constexpr int kUnroll = CT > 0 ? CT : 4;
#pragma unroll kUnroll
    for (int c = 0; c < nc; c++) {
        const float *p = in_b + c * s1c;
        float *q = gin1_b + c * s1c;
        const float g = ld_stream(gout_p + c * s1c);
        const float vTL = p[oTL], vTR = p[oTR], vBL = p[oBL], vBR = p[oBR];
        atomic_add_f32(q + oTL, g * (1 - s.a) * (1 - s.b));
        atomic_add_f32(q + oTR, g * s.a * (1 - s.b));
        atomic_add_f32(q + oBL, g * (1 - s.a) * s.b);
        atomic_add_f32(q + oBR, g * s.a * s.b);
        float tmp = 0.0f;
        tmp += gam_x * (vTR - vTL);
        tmp += (1 - gam_x) * (vBR - vBL);
        botx += g * tmp;
        tmp = 0.0f;
        tmp += gam_y * (vBL - vTL);
        tmp += (1 - gam_y) * (vBR - vTR);
        boty += g * tmp;
    }
    float *g2 = gin2 + b * s2b + (int64_t)y * s2h + x;
    st_stream(g2, botx);                                    // assignment, my_lib_kernel.cu:649,669
    st_stream(g2 + s2c, boty);
}

static int launch_bl_fwd(hipStream_t stream, int w, int h, int channel, int batch, int s1b, int s1c, int s1h,
                         int s2b, int s2c, int s2h, const float *input1, const float *input2, float *output)
{
    if (w <= 0 || h <= 0 || channel <= 0 || batch <= 0) return 0;
    const int tiles_x = (w + kWave - 1) / kWave, tiles_y = (h + 3) / 4;
    const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
    if (channel == 3)
        hipLaunchKernelGGL(bl_fwd<3>, dim3(nwg), dim3(256), 0, stream, w, h, channel, tiles_x, tiles_y,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2, output);
    else
        hipLaunchKernelGGL(bl_fwd<0>, dim3(nwg), dim3(256), 0, stream, w, h, channel, tiles_x, tiles_y,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2, output);
    return launch_status();
}

static int launch_bl_bwd(hipStream_t stream, int w, int h, int channel, int batch, int s1b, int s1c, int s1h,
                         int s2b, int s2c, int s2h, const float *input1, const float *input2,
                         const float *gradoutput, float *gradinput1, float *gradinput2)
{
    if (w <= 0 || h <= 0 || channel <= 0 || batch <= 0) return 0;
    const int tiles_x = (w + kWave - 1) / kWave, tiles_y = (h + 3) / 4;
    const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
    if (channel == 3)
        hipLaunchKernelGGL(bl_bwd<3>, dim3(nwg), dim3(256), 0, stream, w, h, channel, tiles_x, tiles_y,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2,
                           gradoutput, gradinput1, gradinput2);
    else
        hipLaunchKernelGGL(bl_bwd<0>, dim3(nwg), dim3(256), 0, stream, w, h, channel, tiles_x, tiles_y,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2,
                           gradoutput, gradinput1, gradinput2);
    return launch_status();
}

}  // namespace memc

using namespace memc;

extern "C" int InterpolationLayer_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const float *input1, const float *input2, float *output)
{
    (void)nElement; (void)s1w; (void)s2w;
    return launch_bl_fwd((hipStream_t)stream, w, h, channel, batch, s1b, s1c, s1h, s2b, s2c, s2h, input1, input2, output);
}

extern "C" int InterpolationLayer_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const float *input1, const float *input2, const float *gradoutput, float *gradinput1, float *gradinput2)
{
    (void)nElement; (void)s1w; (void)s2w;
    return launch_bl_bwd((hipStream_t)stream, w, h, channel, batch, s1b, s1c, s1h, s2b, s2c, s2h, input1, input2,
                         gradoutput, gradinput1, gradinput2);
}

extern "C" int InterpolationChLayer_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const float *input1, const float *input2, float *output)
{
    (void)nElement; (void)s1w; (void)s2w;
    return launch_bl_fwd((hipStream_t)stream, w, h, channel, batch, s1b, s1c, s1h, s2b, s2c, s2h, input1, input2, output);
}

extern "C" int InterpolationChLayer_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const float *input1, const float *input2, const float *gradoutput, float *gradinput1, float *gradinput2)
{
    (void)nElement; (void)s1w; (void)s2w;
    return launch_bl_bwd((hipStream_t)stream, w, h, channel, batch, s1b, s1c, s1h, s2b, s2c, s2h, input1, input2,
                         gradoutput, gradinput1, gradinput2);
}
