// memc_fi.hpp -- site geometry shared by the FilterInterpolation kernels (filter_interpolation.hip, fi_bwd_cn.hip).
#pragma once

#include "memc_tile.hpp"

namespace memc {

struct FiSite4 {          // geometry of a lane's four sites
    int ix[4], iy[4];
    float a[4], b[4];
    unsigned valid;       // bit j
};

// sites of this lane whose (clamped) window lies inside the band
__device__ __forceinline__ unsigned fi_covered(const Region &r, const FiSite4 &g, int W, int H)
{
    unsigned m = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        if (((g.valid >> j) & 1) &&
            r.covers(max(g.ix[j] - 1, 0), min(g.ix[j] + 2, W - 1), max(g.iy[j] - 1, 0), min(g.iy[j] + 2, H - 1)))
            m |= 1u << j;
    return m;
}

// Scalar evaluation of ONE site for channels [0, nch) of `plane0`, everything read from global memory
// (flow, taps, image): the rare path for sites whose source window is not in the staged LDS region, and the
// body of the any-filter-size kernel.  Same arithmetic order as the fast path.
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

// One site of the backward, everything from global memory, image gradient with global atomics: the rare path for
// sites that no LDS band covers, and the body of the any-filter-size kernel (my_lib_kernel.cu:1248-1515).
__device__ __noinline__ inline void fi_bwd_site_scalar(int x, int y, int W, int H, int C, int fs,
                                                const float *in_b, float *gin1_b, int64_t s1c, int s1h,
                                                const float *flow_p, float *g2, int64_t s2c,
                                                const float *tap_p, float *g3, int64_t s3c, const float *gout_p)
{
    const float fx = flow_p[0], fy = flow_p[s2c];
    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    if (!s.valid) return;
    const int L = s.ix + 1 - fs / 2, T = s.iy + 1 - fs / 2, R = L + fs, Bm = T + fs;
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
                if (c == 0) g3[k] = wgt * p[jj + ii]; else g3[k] += wgt * p[jj + ii];
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
    g2[0] = botx;
    g2[s2c] = boty;
}

}  // namespace memc
