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


// gradinput3 and gradinput2 of ONE site straight from global memory (mixed quads of the tiled RGB backward: some of a
// lane's four sites belong to another band or are invalid).  Assigns both, like the tiled path; the image
// gradient of such a site still goes through the tile's LDS planes.
__device__ __noinline__ inline void fi_bwd_site_taps(int x, int y, int W, int H, const float *in_b, int64_t s1c, int s1h,
                                              const float *flow_p, float *g2, int64_t s2c, const float *tap_p,
                                              float *g3, int64_t s3c, const float *gout_p)
{
    const FiSite s = fi_locate(x, y, W, H, flow_p[0], flow_p[s2c]);
    if (!s.valid) return;
    const float g0 = gout_p[0], g1 = gout_p[s1c], gc2 = gout_p[2 * s1c];
    float gx = 0.0f, gy = 0.0f;
    for (int k = 0; k < 4; k++) {
        const float *row = in_b + (int64_t)clampi(s.iy - 1 + k, H - 1) * s1h;
        for (int m = 0; m < 4; m++) {
            const float *p = row + clampi(s.ix - 1 + m, W - 1);
            float sv = 0.0f;
            sv += g0 * p[0];  sv += g1 * p[s1c];  sv += gc2 * p[2 * s1c];
            const float wa = m < 2 ? (1 - s.a) : s.a, wb = k < 2 ? (1 - s.b) : s.b;
            g3[(k * 4 + m) * s3c] = (wa * wb) * sv;
            const float st = sv * tap_p[(k * 4 + m) * s3c];
            gx += (m < 2 ? -wb : wb) * st;
            gy += (k < 2 ? -wa : wa) * st;
        }
    }
    g2[0] = gx;
    g2[s2c] = gy;
}

// gradinput2 / gradinput3 are fully DEFINED by the backward kernels (the Python layer hands them over
// uninitialised -- their memsets were 72 B/site, a seventh of the call): a quad that contains an invalid site
// first stores zeros to its 16 + 2 float4; its valid sites are then stored site by site (fi_bwd_site_taps), by the
// same lane and therefore after these.  Quads of four valid sites are stored by phase 1 or by fi_bwd_site_taps.
__device__ __forceinline__ void fi_bwd_zero_invalid(bool inb, unsigned valid, float *gin2_b, int64_t s2c, unsigned o2,
                                                    float *gin3_b, int64_t s3c, unsigned o3)
{
    if (!inb || valid == 0xFu) return;         // rare (image borders, |flow| guard): ordinary 64-bit addressing
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    float *q3 = gin3_b + (o3 >> 2), *q2 = gin2_b + (o2 >> 2);
#pragma unroll 1
    for (int k = 0; k < 16; k++) *reinterpret_cast<f32x4u *>(q3 + k * s3c) = z;
    *reinterpret_cast<f32x4u *>(q2) = z;
    *reinterpret_cast<f32x4u *>(q2 + s2c) = z;
}

}  // namespace memc
