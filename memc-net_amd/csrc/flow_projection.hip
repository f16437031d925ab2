// flow_projection.hip -- FlowProjection / DepthFlowProjection: forward splat of -flow to the intermediate
// frame, count-normalisation, hole filling; and their backward passes.  gfx950 only.
//
// Replaces my_package/src/my_lib_kernel.cu:1630-2516 of the reference (scatter :1630/:2053, averaging
// :1696/:2122, hole fill :1742/:2169, backward :1837/:2265; launchers :1905,:1994,:2365,:2458).
// Semantics: SURVEY.md appendix A.3/A.4, including the reference's observable quirks: duplicate adds when the
// clamped right/bottom neighbour coincides with the left/top one, and the hole fill's dead downward search.
//
// The depth-weighted operator is the same code with a per-site weight d (DEPTH == true); the plain
// operator is d == 1.
#include "memc_common.hpp"
#include "memc_internal.h"

namespace memc {

// --------------------------------------------------------------------------------------------------
// Pass 1: scatter.  One lane = one source site; 12 fp32 atomics per valid site (8 flow + 4 count) into
// the caller-zeroed output / count planes.  With smooth flow the 64 lanes of a wave target 64 (nearly)
// consecutive cells, so each wave-level atomic touches two or three cache lines.
// --------------------------------------------------------------------------------------------------
template <bool DEPTH>
__global__ __launch_bounds__(256) void proj_scatter(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_p = flow + b * s1b + (int64_t)y * s1h + x;
    const float fx = ld_stream(flow_p);
    const float fy = ld_stream(flow_p + s1c);
    const BlSite s = bl_locate<false>(x, y, W, H, fx, fy);
    if (!s.valid) return;
    float vx = -fx, vy = -fy, vc = 1.0f;
    if (DEPTH) {
        const float d = ld_stream(depth + b * sdb + (int64_t)y * sdh + x);
        vx = -d * fx;                                       // my_lib_kernel.cu:2102-2109
        vy = -d * fy;
        vc = d * 1.0f;                                      // :2111-2114
    }
    float *ox = out + b * s1b, *oy = ox + s1c, *cn = count + b * scb;
    const int oT = s.T * s1h, oB = s.Bm * s1h, cT = s.T * sch, cB = s.Bm * sch;
    atomic_add_f32(ox + oT + s.L, vx);  atomic_add_f32(ox + oT + s.R, vx);
    atomic_add_f32(ox + oB + s.L, vx);  atomic_add_f32(ox + oB + s.R, vx);
    atomic_add_f32(oy + oT + s.L, vy);  atomic_add_f32(oy + oT + s.R, vy);
    atomic_add_f32(oy + oB + s.L, vy);  atomic_add_f32(oy + oB + s.R, vy);
    atomic_add_f32(cn + cT + s.L, vc);  atomic_add_f32(cn + cT + s.R, vc);
    atomic_add_f32(cn + cB + s.L, vc);  atomic_add_f32(cn + cB + s.R, vc);
}

// --------------------------------------------------------------------------------------------------
// Pass 2: out /= count where count > 0 (my_lib_kernel.cu:1730-1735).
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proj_average(
    int W, int H, int tiles_x, int tiles_y, int64_t s1b, int64_t s1c, int s1h, int64_t scb, int sch,
    const float *__restrict__ count, float *__restrict__ out)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;
    const float t = count[b * scb + (int64_t)y * sch + x];
    if (t > 0.0f) {
        float *o = out + b * s1b + (int64_t)y * s1h + x;
        o[0] = o[0] / t;
        o[s1c] = o[s1c] / t;
    }
}

// --------------------------------------------------------------------------------------------------
// Pass 3 (fillhole != 0): every cell with count <= 0 takes the mean of the nearest non-empty cells to
// its left / right / above (my_lib_kernel.cu:1776-1832).  The reference's downward search never runs
// (`while (down_temp = 0.0f && ...)`, :1799): its term is 0 * out[own cell] and contributes nothing.
// Reads touch only cells with count != 0, writes only cells with count <= 0: race-free, deterministic.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proj_fillhole(
    int W, int H, int tiles_x, int tiles_y, int64_t s1b, int64_t s1c, int s1h, int64_t scb, int sch,
    const float *__restrict__ count, float *out)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;
    const float *cn = count + b * scb;
    if (!(cn[(int64_t)y * sch + x] <= 0.0f)) return;

    int lo = x;  float lt = 0.0f;
    while (lt == 0.0f && lo - 1 >= 0) { lo--; lt = cn[(int64_t)y * sch + lo]; }
    int ro = x;  float rt = 0.0f;
    while (rt == 0.0f && ro + 1 <= W - 1) { ro++; rt = cn[(int64_t)y * sch + ro]; }
    int uo = y;  float ut = 0.0f;
    while (ut == 0.0f && uo - 1 >= 0) { uo--; ut = cn[(int64_t)uo * sch + x]; }
    const float dt = 0.0f;                                  // dead downward search
    if (lt + rt + ut + dt <= 0.0f) return;
    const float fl = lt > 0.0f ? 1.0f : 0.0f, fr = rt > 0.0f ? 1.0f : 0.0f;
    const float fu = ut > 0.0f ? 1.0f : 0.0f, fd = 0.0f;
    float *o = out + b * s1b;
#pragma unroll
    for (int k = 0; k < 2; k++, o += s1c) {
        float *self = o + (int64_t)y * s1h + x;
        *self = (fl * o[(int64_t)y * s1h + lo] + fr * o[(int64_t)y * s1h + ro] +
                 fu * o[(int64_t)uo * s1h + x] + fd * *self) / (fl + fr + fu + fd);
    }
}

// --------------------------------------------------------------------------------------------------
// Backward (my_lib_kernel.cu:1866-1897 and :2296-2360): pure gather, no atomics.  gradinput buffers
// are read-modify-written so that a caller-provided non-zero initial value accumulates as in the
// reference (`+=`, four sequential terms per component).
// --------------------------------------------------------------------------------------------------
template <bool DEPTH>
__global__ __launch_bounds__(256) void proj_bwd(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth, const float *__restrict__ count,
    const float *__restrict__ fwd_out, const float *__restrict__ gout,
    float *__restrict__ gin1, float *__restrict__ gin2)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_p = flow + b * s1b + (int64_t)y * s1h + x;
    const float fx = ld_stream(flow_p);
    const float fy = ld_stream(flow_p + s1c);
    const BlSite s = bl_locate<false>(x, y, W, H, fx, fy);
    if (!s.valid) return;
    const float *cn = count + b * scb;
    const float c00 = cn[s.T * sch + s.L], c01 = cn[s.T * sch + s.R];
    const float c10 = cn[s.Bm * sch + s.L], c11 = cn[s.Bm * sch + s.R];
    const int o00 = s.T * s1h + s.L, o01 = s.T * s1h + s.R, o10 = s.Bm * s1h + s.L, o11 = s.Bm * s1h + s.R;
    float d = 1.0f;
    if (DEPTH) d = ld_stream(depth + b * sdb + (int64_t)y * sdh + x);
    float gd = 0.0f;
    if (DEPTH) gd = gin2[b * sdb + (int64_t)y * sdh + x];
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float *go = gout + b * s1b + k * s1c;
        float *gp = gin1 + b * s1b + k * s1c + (int64_t)y * s1h + x;
        const float g00 = go[o00], g01 = go[o01], g10 = go[o10], g11 = go[o11];
        float g = *gp;
        if (DEPTH) {
            g += -g00 * d / c00;  g += -g01 * d / c01;  g += -g10 * d / c10;  g += -g11 * d / c11;
            const float *fo = fwd_out + b * s1b + k * s1c;
            const float f = k ? fy : fx;
            gd += -g00 / c00 * (f - fo[o00]);
            gd += -g01 / c01 * (f - fo[o01]);
            gd += -g10 / c10 * (f - fo[o10]);
            gd += -g11 / c11 * (f - fo[o11]);
        } else {
            g += -g00 / c00;  g += -g01 / c01;  g += -g10 / c10;  g += -g11 / c11;
        }
        *gp = g;
    }
    if (DEPTH) gin2[b * sdb + (int64_t)y * sdh + x] = gd;
}

static int g_proj_variant = -1;

template <bool DEPTH>
static int launch_proj_fwd(hipStream_t stream, int w, int h, int batch, int fillhole,
                           int s1b, int s1c, int s1h, int sdb, int sdh, int scb, int sch,
                           const float *flow, const float *depth, float *count, float *out)
{
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    const int tiles_x = (w + kWave - 1) / kWave, tiles_y = (h + 3) / 4;
    const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
    hipLaunchKernelGGL(proj_scatter<DEPTH>, dim3(nwg), dim3(256), 0, stream, w, h, tiles_x, tiles_y,
                       (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)sdb, sdh, (int64_t)scb, sch, flow, depth, count, out);
    if (launch_status() != 0) return -1;
    hipLaunchKernelGGL(proj_average, dim3(nwg), dim3(256), 0, stream, w, h, tiles_x, tiles_y,
                       (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)scb, sch, count, out);
    if (launch_status() != 0) return -1;
    if (fillhole) {
        hipLaunchKernelGGL(proj_fillhole, dim3(nwg), dim3(256), 0, stream, w, h, tiles_x, tiles_y,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)scb, sch, count, out);
        if (launch_status() != 0) return -1;
    }
    return 0;
}

template <bool DEPTH>
static int launch_proj_bwd(hipStream_t stream, int w, int h, int batch,
                           int s1b, int s1c, int s1h, int sdb, int sdh, int scb, int sch,
                           const float *flow, const float *depth, const float *count, const float *fwd_out,
                           const float *gout, float *gin1, float *gin2)
{
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    const int tiles_x = (w + kWave - 1) / kWave, tiles_y = (h + 3) / 4;
    const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
    hipLaunchKernelGGL(proj_bwd<DEPTH>, dim3(nwg), dim3(256), 0, stream, w, h, tiles_x, tiles_y,
                       (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)sdb, sdh, (int64_t)scb, sch, flow, depth, count,
                       fwd_out, gout, gin1, gin2);
    return launch_status();
}

}  // namespace memc

using namespace memc;

extern "C" void memc_debug_set_projection_variant(int v) { g_proj_variant = v; }

extern "C" int FlowProjection_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int fillhole,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int scb, const int scc, const int sch, const int scw,
    const float *input1, float *count, float *output)
{
    (void)nElement; (void)channel; (void)s1w; (void)scc; (void)scw;
    return launch_proj_fwd<false>((hipStream_t)stream, w, h, batch, fillhole, s1b, s1c, s1h, 0, 0, scb, sch,
                                  input1, nullptr, count, output);
}

extern "C" int FlowProjection_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int scb, const int scc, const int sch, const int scw,
    const float *input1, const float *count, const float *gradoutput, float *gradinput1)
{
    (void)nElement; (void)channel; (void)s1w; (void)scc; (void)scw;
    return launch_proj_bwd<false>((hipStream_t)stream, w, h, batch, s1b, s1c, s1h, 0, 0, scb, sch, input1, nullptr,
                                  count, nullptr, gradoutput, gradinput1, nullptr);
}

extern "C" int DepthFlowProjection_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int fillhole,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const int scb, const int scc, const int sch, const int scw,
    const float *input1, const float *input2, float *count, float *output)
{
    (void)nElement; (void)channel; (void)s1w; (void)s2c; (void)s2w; (void)scc; (void)scw;
    return launch_proj_fwd<true>((hipStream_t)stream, w, h, batch, fillhole, s1b, s1c, s1h, s2b, s2h, scb, sch,
                                 input1, input2, count, output);
}

extern "C" int DepthFlowProjection_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const int scb, const int scc, const int sch, const int scw,
    const float *input1, const float *input2, const float *count, const float *output,
    const float *gradoutput, float *gradinput1, float *gradinput2)
{
    (void)nElement; (void)channel; (void)s1w; (void)s2c; (void)s2w; (void)scc; (void)scw;
    return launch_proj_bwd<true>((hipStream_t)stream, w, h, batch, s1b, s1c, s1h, s2b, s2h, scb, sch, input1, input2,
                                 count, output, gradoutput, gradinput1, gradinput2);
}
