// flow_prologue.hip -- EXTENSION (no reference counterpart): the prologue of FlowProjection in
// the networks, networks/MEMC_Net_star.py:172-176:
//
//     flow = F.interpolate(div_flow * flownets(pair) / 2.0, scale_factor=4, mode="bilinear")      (then FlowProject)
//
// as ONE kernel: the quarter-resolution flow is read (0.5 B per output site, L2-resident), scaled and bilinearly
// upsampled x4, and each output element is written once as part of a dwordx4 (8 B per site) -- instead of a scaling
// kernel, a division kernel and torch's upsampling kernel.  gfx950 only.  Built to SETTLE the question the survey
// raised (section 8f-2, "fuse the prologue"): it does not pay -- see the numbers at the kernel -- and is not used by
// the networks by default.
//
// Sampling follows torch.nn.functional.interpolate(mode="bilinear") exactly (ATen UpSample.h,
// area_pixel_compute_source_index): align_corners: src = dst * (in - 1) / (out - 1); otherwise src = max((dst + 0.5)
// / 4 - 0.5, 0); i0 = (int)src, i1 = i0 + (i0 < in - 1), l1 = src - i0, l0 = 1 - l1;
// out = l0y * (l0x * a + l1x * b) + l1y * (l0x * c + l1x * d), the scaled sample being (mul * v) / div.
//
// Why the upsampling is NOT folded into the owner kernel's scan (measured): that kernel re-reads its
// flow 4.75x through the L1 and is bound by workgroups per CU and VALU issue; replacing each 16-byte load by four
// bilinear evaluations (~50 VALU per quad) would add ~60 % to its instruction count to save the 8 B/site this kernel
// writes and the projection reads once from HBM.
#include "memc_common.hpp"
#include "memc_internal.h"
#include "memc_tile.hpp"

#include <math.h>

namespace memc {

__device__ __forceinline__ void up_index(int dst, int in, float scale, bool align, int &i0, int &i1, float &l0, float &l1)
{
    float src = align ? scale * (float)dst : fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.0f);
    i0 = min((int)src, in - 1);
    i1 = i0 + (i0 < in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
}

// One workgroup = one output row of one plane (its row indices and weights are wave-uniform); one lane = four
// consecutive output columns, i.e. ONE low-resolution column X: without align_corners the four outputs 4X .. 4X+3
// sample the columns X-1, X, X+1 (clamped), so a lane scales six samples, not sixteen.  INV: the divisor is a power
// of two (the networks' 2.0) -- multiply by its exact reciprocal.
// Measured (tools/bench_ops.py --only prologue, 32x2x180x320 -> 720x1280): 95 us; the torch expression it replaces
// (two tiny element-wise kernels on the low-resolution tensor + ATen's upsampling kernel): 84 us; a variant with
// one 4x4 output block per lane (nine samples for sixteen outputs): 122 us.  ATen's kernel is already write-bound:
// there is nothing to win here, the networks keep the torch expression (fused_upsample = False).
template <bool ALIGN, bool INV>
__global__ __launch_bounds__(320) void flow_upsample4(
    int w, int h, int channels, int64_t sib, int64_t sic, int sih, int64_t sob, int64_t soc, int soh,
    float mul, float div, float scale_y, float scale_x, const float *__restrict__ in, float *__restrict__ out)
{
    const int H = 4 * h;
    const int y = blockIdx.x % H, p = blockIdx.x / H, b = p / channels, c = p % channels;
    int y0, y1;
    float ly0, ly1;
    up_index(y, h, scale_y, ALIGN, y0, y1, ly0, ly1);
    const float *r0 = in + b * sib + c * sic + (int64_t)y0 * sih, *r1 = in + b * sib + c * sic + (int64_t)y1 * sih;
    float *orow = out + b * sob + c * soc + (int64_t)y * soh;
    const float inv = 1.0f / div;
    auto scaled = [&](float v) { return INV ? (mul * v) * inv : (mul * v) / div; };
    for (int X = threadIdx.x; X < w; X += blockDim.x) {
        f32x4 v;
        if (ALIGN) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int x0, x1;
                float lx0, lx1;
                up_index(4 * X + j, w, scale_x, true, x0, x1, lx0, lx1);
                const float a = scaled(r0[x0]), bq = scaled(r0[x1]), cq = scaled(r1[x0]), d = scaled(r1[x1]);
                v[j] = ly0 * (lx0 * a + lx1 * bq) + ly1 * (lx0 * cq + lx1 * d);
            }
        } else {
            const int xm = max(X - 1, 0), xp = min(X + 1, w - 1);
            const float t[3] = {scaled(r0[xm]), scaled(r0[X]), scaled(r0[xp])};
            const float u[3] = {scaled(r1[xm]), scaled(r1[X]), scaled(r1[xp])};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                int x0, x1;
                float lx0, lx1;
                up_index(4 * X + j, w, 0.25f, false, x0, x1, lx0, lx1);
                // x0 is X - 1 (clamped) for j < 2 and X for j >= 2; x1 = x0 + 1 (clamped)
                const int k0 = j < 2 ? 0 : 1;
                const float a = (j < 2 && X == 0) ? t[1] : t[k0], cq = (j < 2 && X == 0) ? u[1] : u[k0];
                const float bq = x1 == x0 ? a : ((j < 2 && X == 0) ? t[2] : t[k0 + 1]);
                const float d = x1 == x0 ? cq : ((j < 2 && X == 0) ? u[2] : u[k0 + 1]);
                v[j] = ly0 * (lx0 * a + lx1 * bq) + ly1 * (lx0 * cq + lx1 * d);
            }
        }
        st_stream4(orow + 4 * X, v);
    }
}

}  // namespace memc

using namespace memc;

// input [B, C, h, w] (element strides sib, sic, sih; unit w stride) -> output [B, C, 4h, 4w] (sob, soc, soh; 16-byte
// aligned rows), output = bilinear_x4((mul * input) / div).  Returns 0, or -1 on a bad geometry / launch error.
extern "C" int FlowUpsample4_gpu_forward_kernel(
    memc_stream_t stream_, const int w, const int h, const int channel, const int batch,
    const int sib, const int sic, const int sih, const int sob, const int soc, const int soh,
    const float mul, const float div, const int align_corners, const float *input, float *output)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (w <= 0 || h <= 0 || channel <= 0 || batch <= 0) return 0;
    if (!vec4_ok(4 * w, {sob, soc, soh}, {output})) return -1;
    const int W = 4 * w, H = 4 * h;
    const float sy = align_corners ? (H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.0f) : 0.25f;
    const float sx = align_corners ? (W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.0f) : 0.25f;
    int e = 0;
    const bool inv = div != 0.0f && frexpf(div, &e) == 0.5f;                 // a power of two: 1 / div is exact
    const unsigned grid = (unsigned)batch * channel * H;
    const unsigned threads = w >= 320 ? 320 : (unsigned)((w + 63) / 64 * 64);
#define MEMC_UP(ALIGN, INV)                                                                                   \
    hipLaunchKernelGGL((flow_upsample4<ALIGN, INV>), dim3(grid), dim3(threads), 0, stream, w, h, channel, (int64_t)sib, \
                       (int64_t)sic, sih, (int64_t)sob, (int64_t)soc, soh, mul, div, sy, sx, input, output)
    if (align_corners) { if (inv) MEMC_UP(true, true); else MEMC_UP(true, false); }
    else { if (inv) MEMC_UP(false, true); else MEMC_UP(false, false); }
#undef MEMC_UP
    return launch_status();
}
