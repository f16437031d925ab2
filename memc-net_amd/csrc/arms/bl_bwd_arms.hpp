// arms/bl_bwd_arms.hpp -- MEASUREMENT BUILD ONLY: the RGB bilinear-warp backward of rounds 1-2 (one transposed fp64 LDS
// plane that the colours take in turn; memc_debug_set_bl_cap 0 / 1 select it with a 48 / 39 KiB staging budget), the A/B
// baseline of bl_bwd_c3_pk.  Textually included by interpolation.hip under MEMC_MEASURE; never part of libmemc_hip.so.
#ifndef MEMC_MEASURE
#error "measurement arms: build with -DMEMC_MEASURE (make measure)"
#endif
// Backward, tiled, RGB: image gradient splatted into LDS accumulators and flushed with coalesced atomics
// (memc_tile.hpp "LDS-privatised scatter"); the flow gradient needs the four corner values, gathered from a
// staged LDS image of the same box.
template <int CAP>
__global__ __launch_bounds__(256, 3) void bl_bwd_tiled_c3(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ gout,
    float *__restrict__ gin1, float *__restrict__ gin2, int sw)
{
    constexpr int LX = 16;
    using G = TileGeom<LX, CAP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // phase 1 uses the LDS as the staged image (48 KiB of pixel quads), phase 2 re-uses the same bytes as ONE
    // transposed fp64 accumulator plane (AccT, 32 KiB) that the colour channels take in turn: three workgroups
    // per CU, and the 16-lane groups of a ds_add_f64 hit adjacent slots (see memc_tile.hpp)
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    double *acc = reinterpret_cast<double *>(smem);
    int *bb = reinterpret_cast<int *>(smem + G::kCapPx * 16);

    const TileCoord tc = tile_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, sw);
    if (tc.tx >= tiles_x) return;
    const int b = tc.b, tile_x0 = tc.tx * G::kTW, tile_y0 = tc.ty * G::kTH;
    const int x = tile_x0 + 4 * (threadIdx.x % LX), y = tile_y0 + threadIdx.x / LX;
    const bool inb = x < W && y < H;
    const int xs = min(x, W - 4), ys = min(y, H - 1);
    const float *flow_p = flow + b * s2b + (int64_t)ys * s2h + xs;
    const f32x4 fx4 = ld_stream4(flow_p), fy4 = ld_stream4(flow_p + s2c);
    const float *gout_p = gout + b * s1b + (int64_t)ys * s1h + xs;
    f32x4 go[3];
#pragma unroll
    for (int c = 0; c < 3; c++) go[c] = ld_stream4(gout_p + c * s1c);

    BlSite st[4];
    int cmin = INT_MAX, cmax = -1, rmin = INT_MAX, rmax = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        st[j] = bl_locate<true>(x + j, y, W, H, fx4[j], fy4[j]);
        st[j].valid = st[j].valid && inb;
        if (st[j].valid) {
            cmin = min(cmin, st[j].L);  cmax = max(cmax, st[j].R);
            rmin = min(rmin, st[j].T);  rmax = max(rmax, st[j].Bm);
        }
    }
    const Region r = tile_region<LX, false, CAP>(cmin, cmax, rmin, rmax, tile_x0, tile_y0, bb);
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    tile_stage<LX, 3>(r, in_b, s1c, s1h, tile);
    __syncthreads();

    // ---- phase 1: flow gradient from the four corner values
    f32x4 gx4 = {0.f, 0.f, 0.f, 0.f}, gy4 = gx4;
    unsigned staged_mask = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!st[j].valid) continue;
        const BlSite &s = st[j];
        const float x2 = (float)(x + j) + fx4[j], y2 = (float)y + fy4[j];
        const float gam_x = (float)s.Bm - y2, gam_y = (float)s.R - x2;   // clamped corners, my_lib_kernel.cu:634,652
        const bool staged = r.covers(s.L, s.R, s.T, s.Bm);
        staged_mask |= (staged ? 1u : 0u) << j;
        f32x4 vTL, vTR, vBL, vBR;
        if (staged) {
            const int rT = (s.T - r.y0) * r.pitch, rB = (s.Bm - r.y0) * r.pitch;
            const int cL = swz_col(s.L - r.x0), cR = swz_col(s.R - r.x0);
            vTL = tile[rT + cL];  vTR = tile[rT + cR];  vBL = tile[rB + cL];  vBR = tile[rB + cR];
        } else {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float *p = in_b + c * s1c;
                vTL[c] = p[s.T * s1h + s.L];   vTR[c] = p[s.T * s1h + s.R];
                vBL[c] = p[s.Bm * s1h + s.L];  vBR[c] = p[s.Bm * s1h + s.R];
            }
        }
        float botx = 0.0f, boty = 0.0f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float gv = go[c][j];
            float tmp = 0.0f;
            tmp += gam_x * (vTR[c] - vTL[c]);
            tmp += (1 - gam_x) * (vBR[c] - vBL[c]);
            botx += gv * tmp;
            tmp = 0.0f;
            tmp += gam_y * (vBL[c] - vTL[c]);
            tmp += (1 - gam_y) * (vBR[c] - vTR[c]);
            boty += gv * tmp;
        }
        gx4[j] = botx;
        gy4[j] = boty;
    }
    // gradinput2 is ASSIGNED at valid sites (my_lib_kernel.cu:649,669); the reference leaves the other sites at the
    // caller's zeros, this kernel stores those zeros itself so the buffer needs no memset beforehand
    if (inb) {
        float *g2 = gin2 + b * s2b + (int64_t)y * s2h + x;
        st_stream4(g2, gx4);
        st_stream4(g2 + s2c, gy4);
    }
    __syncthreads();                           // the image has been read: the LDS becomes the accumulators

    // ---- phase 2: image gradient, 4 fp64 LDS adds per site and channel
    acct_zero<1>(acc);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (!st[j].valid) continue;
            const BlSite &s = st[j];
            const bool staged = (staged_mask >> j) & 1;
            const float gv = c == 0 ? go[0][j] : (c == 1 ? go[1][j] : go[2][j]);
            const float a00 = gv * (1 - s.a) * (1 - s.b), a01 = gv * s.a * (1 - s.b);
            const float a10 = gv * (1 - s.a) * s.b, a11 = gv * s.a * s.b;
            if (staged) {
                const int aT = (s.T - r.y0) * AccT::kPitch, aB = (s.Bm - r.y0) * AccT::kPitch;
                const int aL = acct_col(s.L - r.x0), aR = acct_col(s.R - r.x0);
                lds_add_f64(acc + aT + aL, (double)a00);  lds_add_f64(acc + aT + aR, (double)a01);
                lds_add_f64(acc + aB + aL, (double)a10);  lds_add_f64(acc + aB + aR, (double)a11);
            } else {
                float *q = gin1_b + c * s1c;
                atomic_add_f32(q + s.T * s1h + s.L, a00);   atomic_add_f32(q + s.T * s1h + s.R, a01);
                atomic_add_f32(q + s.Bm * s1h + s.L, a10);  atomic_add_f32(q + s.Bm * s1h + s.R, a11);
            }
        }
        __syncthreads();
        acct_flush_zero(r, acc, gin1_b + c * s1c, s1h);        // leaves the plane zeroed for the next channel
        __syncthreads();
    }
}

