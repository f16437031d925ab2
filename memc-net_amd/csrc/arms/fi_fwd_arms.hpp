// arms/fi_fwd_arms.hpp -- MEASUREMENT BUILD ONLY: the reference-shaped forward (block 32 x 16, taps re-read per channel,
// no LDS): what a straight port gives on this chip (2130 us, 16.6 % of the HBM peak -- DESIGN.md section 4).  Textually
// included by filter_interpolation.hip under MEMC_MEASURE; never part of libmemc_hip.so.
#ifndef MEMC_MEASURE
#error "measurement arms: build with -DMEMC_MEASURE (make measure)"
#endif
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

