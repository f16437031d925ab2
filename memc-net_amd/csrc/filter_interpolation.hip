// filter_interpolation.hip -- the fused adaptive warp (bilinear flow sample x per-pixel fs*fs filter),
// forward and backward, for gfx950.
//
// Replaces my_package/src/my_lib_kernel.cu:1087-1627 of the reference (kernels :1087,:1220, launchers
// :1520,:1571).  Semantics: SURVEY.md appendix A.1/A.2; float evaluation order follows the reference
// (quadrant sums accumulate row-major from 0, then the four-term blend).
//
// Design (HBM-bound gather stencil, ~1.5 flop/B, no MFMA):
//   * one lane = one output site; a wavefront = 64 consecutive sites of one image row, so every tap-plane,
//     flow and output access of a wave is one fully coalesced 256-B segment;
//   * the fs*fs taps and the two flow components of a site are read ONCE into registers and reused for
//     every channel (the reference re-reads all 16 taps per channel, my_lib_kernel.cu:1149-1150);
//   * taps / flow / output are single-use streams -> non-temporal, so L1/L2 keep the source image, whose
//     4x4 windows overlap between neighbouring lanes and rows;
//   * the workgroup->tile map is XCD-chunked (memc_common.hpp) so halo rows are shared inside one L2;
//   * 64-bit batch/channel offsets (4K x batch 8 x 64 channels exceeds 2^31 elements), 32-bit in-plane.
#include "memc_common.hpp"
#include "memc_internal.h"

namespace memc {

// --------------------------------------------------------------------------------------------------
// Forward, fs == 4, direct gather from global memory through L1/L2.
//   ROWS waves per workgroup, one image row each: tile = 64 x ROWS sites.
//   CT > 0: channel count known at compile time (fully unrolled); CT == 0: run-time channel loop.
// --------------------------------------------------------------------------------------------------
template <int CT, int ROWS>
__global__ __launch_bounds__(64 * ROWS) void fi_fwd_direct_fs4(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    float *__restrict__ out)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * ROWS + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = ld_stream(flow_b);
    const float fy = ld_stream(flow_b + s2c);
    const float *tap_p = filt + b * s3b + (int64_t)y * s3h + x;
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = ld_stream(tap_p + k * s3c);

    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    const float *in_b = in1 + b * s1b;
    float *out_p = out + b * s1b + (int64_t)y * s1h + x;
    const int nc = CT > 0 ? CT : C;
    constexpr int kUnroll = CT > 0 ? CT : 4;

    if (s.valid) {
        int ro[4], co[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ro[k] = clampi(s.iy - 1 + k, H - 1) * s1h;
            co[k] = clampi(s.ix - 1 + k, W - 1);
        }
        const float w00 = (1 - s.a) * (1 - s.b), w01 = s.a * (1 - s.b);
        const float w10 = (1 - s.a) * s.b, w11 = s.a * s.b;
#pragma unroll kUnroll
        for (int c = 0; c < nc; c++) {
            const float *p = in_b + c * s1c;
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) v[j * 4 + i] = p[ro[j] + co[i]];
            // quadrant sums: rows {0,1} top / {2,3} bottom, cols {0,1} left / {2,3} right
            float TL = 0.0f, TR = 0.0f, BL = 0.0f, BR = 0.0f;
            TL += v[0] * t[0];   TL += v[1] * t[1];   TL += v[4] * t[4];   TL += v[5] * t[5];
            TR += v[2] * t[2];   TR += v[3] * t[3];   TR += v[6] * t[6];   TR += v[7] * t[7];
            BL += v[8] * t[8];   BL += v[9] * t[9];   BL += v[12] * t[12]; BL += v[13] * t[13];
            BR += v[10] * t[10]; BR += v[11] * t[11]; BR += v[14] * t[14]; BR += v[15] * t[15];
            st_stream(out_p + c * s1c, w00 * TL + w01 * TR + w10 * BL + w11 * BR);
        }
    } else {
        // out-of-range site copies the input pixel (my_lib_kernel.cu:1209-1214)
        const float *p = in_b + (int64_t)y * s1h + x;
        for (int c = 0; c < nc; c++) st_stream(out_p + c * s1c, p[c * s1c]);
    }
}

// --------------------------------------------------------------------------------------------------
// Forward, any filter size (run-time loops, taps read from global per use).  Rare path: the networks
// only ever use fs == 4 (MEMC_Net_star.py:30 `filter_size = 4`).
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ float fi_quad_sum(const float *p, int s1h, int W, int H, const float *tap_p,
                                             int64_t s3c, int fs, int L, int T, int j0, int j1, int i0, int i1)
{
    float acc = 0.0f;
    for (int j = j0; j <= j1; j++) {
        const int jj = clampi(j, H - 1) * s1h;
        for (int i = i0; i <= i1; i++)
            acc += p[jj + clampi(i, W - 1)] * tap_p[((j - T) * fs + (i - L)) * s3c];
    }
    return acc;
}

__global__ __launch_bounds__(256) void fi_fwd_generic(
    int W, int H, int C, int fs, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    float *__restrict__ out)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = flow_b[0], fy = flow_b[s2c];
    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    const float *tap_p = filt + b * s3b + (int64_t)y * s3h + x;
    const float *in_b = in1 + b * s1b;
    float *out_p = out + b * s1b + (int64_t)y * s1h + x;
    if (s.valid) {
        const int L = s.ix + 1 - fs / 2, T = s.iy + 1 - fs / 2, R = L + fs, Bm = T + fs;
        for (int c = 0; c < C; c++) {
            const float *p = in_b + c * s1c;
            const float TL = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, T, s.iy, L, s.ix);
            const float TR = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, T, s.iy, s.ix + 1, R - 1);
            const float BL = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, s.iy + 1, Bm - 1, L, s.ix);
            const float BR = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, s.iy + 1, Bm - 1, s.ix + 1, R - 1);
            out_p[c * s1c] = (1 - s.a) * (1 - s.b) * TL + s.a * (1 - s.b) * TR +
                             (1 - s.a) * s.b * BL + s.a * s.b * BR;
        }
    } else {
        const float *p = in_b + (int64_t)y * s1h + x;
        for (int c = 0; c < C; c++) out_p[c * s1c] = p[c * s1c];
    }
}

// --------------------------------------------------------------------------------------------------
// Measurement arm only (bench_ops.py): a kernel with the REFERENCE's structure -- block (32,16), one
// thread per site, taps re-read from global for every channel, no streaming hints, blockIdx-ordered
// tiles -- to show what a straight port achieves on MI355X.  Never selected by the product path.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void fi_fwd_refshape(
    int W, int H, int C, int fs,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *in1, const float *flow, const float *filt, float *out)
{
    const int x = blockIdx.x * 32 + threadIdx.x;
    const int y = blockIdx.y * 16 + threadIdx.y;
    const int b = blockIdx.z;
    if (x >= W || y >= H) return;
    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = flow_b[0], fy = flow_b[s2c];
    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    const float *tap_p = filt + b * s3b + (int64_t)y * s3h + x;
    const float *in_b = in1 + b * s1b;
    float *out_p = out + b * s1b + (int64_t)y * s1h + x;
    if (s.valid) {
        const int L = s.ix + 1 - fs / 2, T = s.iy + 1 - fs / 2, R = L + fs, Bm = T + fs;
        for (int c = 0; c < C; c++) {
            const float *p = in_b + c * s1c;
            const float TL = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, T, s.iy, L, s.ix);
            const float TR = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, T, s.iy, s.ix + 1, R - 1);
            const float BL = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, s.iy + 1, Bm - 1, L, s.ix);
            const float BR = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, s.iy + 1, Bm - 1, s.ix + 1, R - 1);
            out_p[c * s1c] = (1 - s.a) * (1 - s.b) * TL + s.a * (1 - s.b) * TR +
                             (1 - s.a) * s.b * BL + s.a * s.b * BR;
        }
    } else {
        const float *p = in_b + (int64_t)y * s1h + x;
        for (int c = 0; c < C; c++) out_p[c * s1c] = p[c * s1c];
    }
}

// --------------------------------------------------------------------------------------------------
// Backward, fs == 4, direct.  Per valid site (my_lib_kernel.cu:1248-1515):
//   gradinput1 += scatter of g * wq * tap      (fp32 atomics; neighbouring lanes hit neighbouring cells)
//   gradinput3 += sum_c g * wq * in            (each site owns its 16 taps: accumulated in registers over
//                                               the channels, ONE read-modify-write per tap; the
//                                               reference issues 16*C atomics for this)
//   gradinput2  = flow gradients from the quadrant sums (assignment; the sums are computed once, the
//                 reference recomputes them twice more)
// Invalid sites write nothing: the buffers keep the caller's zeros.
// --------------------------------------------------------------------------------------------------
template <int CT, int ROWS>
__global__ __launch_bounds__(64 * ROWS) void fi_bwd_direct_fs4(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    const float *__restrict__ gout, float *__restrict__ gin1, float *__restrict__ gin2,
    float *__restrict__ gin3)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * ROWS + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = ld_stream(flow_b);
    const float fy = ld_stream(flow_b + s2c);
    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    if (!s.valid) return;

    const float *tap_p = filt + b * s3b + (int64_t)y * s3h + x;
    float t[16], gt[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        t[k] = ld_stream(tap_p + k * s3c);
        gt[k] = 0.0f;
    }
    int ro[4], co[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ro[k] = clampi(s.iy - 1 + k, H - 1) * s1h;
        co[k] = clampi(s.ix - 1 + k, W - 1);
    }
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    const float *gout_p = gout + b * s1b + (int64_t)y * s1h + x;
    const int nc = CT > 0 ? CT : C;
    constexpr int kUnroll = CT > 0 ? CT : 2;
    float botx = 0.0f, boty = 0.0f;
    const float gam_x = 1.0f - s.b, gam_y = 1.0f - s.a;
#pragma unroll kUnroll
    for (int c = 0; c < nc; c++) {
        const float *p = in_b + c * s1c;
        float *q = gin1_b + c * s1c;
        const float g = ld_stream(gout_p + c * s1c);
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) v[j * 4 + i] = p[ro[j] + co[i]];
        const float wq[4] = {g * (1 - s.a) * (1 - s.b), g * s.a * (1 - s.b),
                             g * (1 - s.a) * s.b, g * s.a * s.b};
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int k = j * 4 + i;
                const float wgt = wq[(j >> 1) * 2 + (i >> 1)];
                atomic_add_f32(q + ro[j] + co[i], wgt * t[k]);
                gt[k] += wgt * v[k];
            }
        float TL = 0.0f, TR = 0.0f, BL = 0.0f, BR = 0.0f;
        TL += v[0] * t[0];   TL += v[1] * t[1];   TL += v[4] * t[4];   TL += v[5] * t[5];
        TR += v[2] * t[2];   TR += v[3] * t[3];   TR += v[6] * t[6];   TR += v[7] * t[7];
        BL += v[8] * t[8];   BL += v[9] * t[9];   BL += v[12] * t[12]; BL += v[13] * t[13];
        BR += v[10] * t[10]; BR += v[11] * t[11]; BR += v[14] * t[14]; BR += v[15] * t[15];
        float tmp = 0.0f;
        tmp += gam_x * (TR - TL);
        tmp += (1.0f - gam_x) * (BR - BL);
        botx += g * tmp;
        tmp = 0.0f;
        tmp += gam_y * (BL - TL);
        tmp += (1.0f - gam_y) * (BR - TR);
        boty += g * tmp;
    }
    float *g3 = gin3 + b * s3b + (int64_t)y * s3h + x;
#pragma unroll
    for (int k = 0; k < 16; k++) g3[k * s3c] += gt[k];
    float *g2 = gin2 + b * s2b + (int64_t)y * s2h + x;
    st_stream(g2, botx);
    st_stream(g2 + s2c, boty);
}

// Backward, any filter size (rare path; run-time loops).
__global__ __launch_bounds__(256) void fi_bwd_generic(
    int W, int H, int C, int fs, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    const float *__restrict__ gout, float *__restrict__ gin1, float *__restrict__ gin2,
    float *__restrict__ gin3)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;
    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = flow_b[0], fy = flow_b[s2c];
    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    if (!s.valid) return;
    const int L = s.ix + 1 - fs / 2, T = s.iy + 1 - fs / 2, R = L + fs, Bm = T + fs;
    const float *tap_p = filt + b * s3b + (int64_t)y * s3h + x;
    float *g3 = gin3 + b * s3b + (int64_t)y * s3h + x;
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    const float *gout_p = gout + b * s1b + (int64_t)y * s1h + x;
    float botx = 0.0f, boty = 0.0f;
    const float gam_x = 1.0f - s.b, gam_y = 1.0f - s.a;
    for (int c = 0; c < C; c++) {
        const float *p = in_b + c * s1c;
        float *q = gin1_b + c * s1c;
        const float g = gout_p[c * s1c];
        for (int j = T; j < Bm; j++) {
            const int jj = clampi(j, H - 1) * s1h;
            for (int i = L; i < R; i++) {
                const int ii = clampi(i, W - 1);
                const float wgt = (j <= s.iy) ? ((i <= s.ix) ? g * (1 - s.a) * (1 - s.b) : g * s.a * (1 - s.b))
                                              : ((i <= s.ix) ? g * (1 - s.a) * s.b : g * s.a * s.b);
                const int64_t k = ((j - T) * fs + (i - L)) * s3c;
                atomic_add_f32(q + jj + ii, wgt * tap_p[k]);
                g3[k] += wgt * p[jj + ii];     // this site owns its taps: plain read-modify-write
            }
        }
        const float TL = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, T, s.iy, L, s.ix);
        const float TR = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, T, s.iy, s.ix + 1, R - 1);
        const float BL = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, s.iy + 1, Bm - 1, L, s.ix);
        const float BR = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, s.iy + 1, Bm - 1, s.ix + 1, R - 1);
        float tmp = 0.0f;
        tmp += gam_x * (TR - TL);
        tmp += (1.0f - gam_x) * (BR - BL);
        botx += g * tmp;
        tmp = 0.0f;
        tmp += gam_y * (BL - TL);
        tmp += (1.0f - gam_y) * (BR - TR);
        boty += g * tmp;
    }
    float *g2 = gin2 + b * s2b + (int64_t)y * s2h + x;
    g2[0] = botx;
    g2[s2c] = boty;
}

}  // namespace memc

using namespace memc;

// Variant selection for A/B measurement (memc_internal.h); -1 = automatic.
static int g_fi_fwd_variant = -1;
extern "C" void memc_debug_set_fi_fwd_variant(int v) { g_fi_fwd_variant = v; }

extern "C" int FilterInterpolationLayer_gpu_forward_kernel(
    memc_stream_t stream_, const int nElement, const int w, const int h, const int channel, const int batch,
    const int filter_size,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const int s3b, const int s3c, const int s3h, const int s3w,
    const float *input1, const float *input2, const float *input3, float *output)
{
    (void)nElement; (void)s1w; (void)s2w; (void)s3w;
    hipStream_t stream = (hipStream_t)stream_;
    if (w <= 0 || h <= 0 || channel <= 0 || batch <= 0) return 0;
    const int variant = g_fi_fwd_variant;

    if (variant == 0) {  // reference-structure measurement arm
        dim3 block(32, 16, 1), grid((w + 31) / 32, (h + 15) / 16, batch);
        hipLaunchKernelGGL(fi_fwd_refshape, grid, block, 0, stream, w, h, channel, filter_size,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, output);
        return launch_status();
    }
    const int tiles_x = (w + kWave - 1) / kWave;
    if (filter_size != 4) {
        const int tiles_y = (h + 3) / 4;
        const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
        hipLaunchKernelGGL(fi_fwd_generic, dim3(nwg), dim3(256), 0, stream, w, h, channel, filter_size,
                           tiles_x, tiles_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, output);
        return launch_status();
    }
#define MEMC_FI_FWD_LAUNCH(CT, ROWS)                                                                       \
    do {                                                                                                   \
        const int tiles_y = (h + (ROWS) - 1) / (ROWS);                                                     \
        const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;                                          \
        hipLaunchKernelGGL((fi_fwd_direct_fs4<CT, ROWS>), dim3(nwg), dim3(64 * (ROWS)), 0, stream, w, h,   \
                           channel, tiles_x, tiles_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b,       \
                           (int64_t)s2c, s2h, (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3,     \
                           output);                                                                        \
    } while (0)
    const int rows = (variant == 2) ? 8 : (variant == 3 ? 2 : 4);
    if (channel == 3) {
        if (rows == 8) MEMC_FI_FWD_LAUNCH(3, 8);
        else if (rows == 2) MEMC_FI_FWD_LAUNCH(3, 2);
        else MEMC_FI_FWD_LAUNCH(3, 4);
    } else {
        if (rows == 8) MEMC_FI_FWD_LAUNCH(0, 8);
        else if (rows == 2) MEMC_FI_FWD_LAUNCH(0, 2);
        else MEMC_FI_FWD_LAUNCH(0, 4);
    }
#undef MEMC_FI_FWD_LAUNCH
    return launch_status();
}

extern "C" int FilterInterpolationLayer_gpu_backward_kernel(
    memc_stream_t stream_, const int nElement, const int w, const int h, const int channel, const int batch,
    const int filter_size,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const int s3b, const int s3c, const int s3h, const int s3w,
    const float *input1, const float *input2, const float *input3,
    const float *gradoutput, float *gradinput1, float *gradinput2, float *gradinput3)
{
    (void)nElement; (void)s1w; (void)s2w; (void)s3w;
    hipStream_t stream = (hipStream_t)stream_;
    if (w <= 0 || h <= 0 || channel <= 0 || batch <= 0) return 0;
    const int tiles_x = (w + kWave - 1) / kWave;
    const int tiles_y = (h + 3) / 4;
    const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
    if (filter_size != 4) {
        hipLaunchKernelGGL(fi_bwd_generic, dim3(nwg), dim3(256), 0, stream, w, h, channel, filter_size,
                           tiles_x, tiles_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, gradoutput,
                           gradinput1, gradinput2, gradinput3);
    } else if (channel == 3) {
        hipLaunchKernelGGL((fi_bwd_direct_fs4<3, 4>), dim3(nwg), dim3(256), 0, stream, w, h, channel,
                           tiles_x, tiles_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, gradoutput,
                           gradinput1, gradinput2, gradinput3);
    } else {
        hipLaunchKernelGGL((fi_bwd_direct_fs4<0, 4>), dim3(nwg), dim3(256), 0, stream, w, h, channel,
                           tiles_x, tiles_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, gradoutput,
                           gradinput1, gradinput2, gradinput3);
    }
    return launch_status();
}
