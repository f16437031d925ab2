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
#include "memc_tile.hpp"
#include "memc_pk.hpp"

namespace memc {

template <int CT>
__global__ __launch_bounds__(256) void bl_fwd(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h,
    const float *__restrict__ in1, const float *__restrict__ flow, float *__restrict__ out, int x0)
{
    // x0: the first column this launch serves (0; W & ~3 when the tiled kernel took the whole quads of a ragged width)
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = x0 + tx * kWave + (threadIdx.x & (kWave - 1));
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
    float *__restrict__ gin1, float *__restrict__ gin2, int x0)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = x0 + tx * kWave + (threadIdx.x & (kWave - 1));   // (x0: see bl_fwd)
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = ld_stream(flow_b);
    const float fy = ld_stream(flow_b + s2c);
    const BlSite s = bl_locate<true>(x, y, W, H, fx, fy);
    float *g2 = gin2 + b * s2b + (int64_t)y * s2h + x;
    if (!s.valid) {                                         // the reference leaves the caller's zeros here
        st_stream(g2, 0.0f);
        st_stream(g2 + s2c, 0.0f);
        return;
    }
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
    st_stream(g2, botx);                                    // assignment, my_lib_kernel.cu:649,669
    st_stream(g2 + s2c, boty);
}

// ==================================================================================================
// Vectorised, LDS-tiled production kernels (W % 4 == 0, 16-B aligned geometry): same machinery as the
// adaptive warp (memc_tile.hpp) with a 2x2 footprint.  The scalar kernels above stay as the fallback.
// ==================================================================================================
struct BlSite4 {
    int oTL[4], oTR[4], oBL[4], oBR[4];   // LDS pixel indices of the four corners (0 when not staged)
    float w00[4], w01[4], w10[4], w11[4];
    unsigned valid, staged;
};

template <int NCH, bool RAG = false>
__device__ __forceinline__ void bl_fwd_chunk(const Region &r, const BlSite4 &g, const BlSite (&st)[4], bool inb,
                                             const float *__restrict__ plane0, float *__restrict__ out_p,
                                             int64_t s1c, int s1h, f32x4 *tile)
{
    tile_stage<16, NCH, 256, RAG>(r, plane0, s1c, s1h, tile);
    __syncthreads();
    if (!inb) return;
    f32x4 res[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        res[j] = g.w00[j] * tile[g.oTL[j]] + g.w01[j] * tile[g.oTR[j]] + g.w10[j] * tile[g.oBL[j]] +
                 g.w11[j] * tile[g.oBR[j]];
        const bool valid = (g.valid >> j) & 1, staged = (g.staged >> j) & 1;
        if (valid && !staged) {               // rare: corners outside the staged box -> global gathers
            const BlSite &s = st[j];
#pragma unroll
            for (int c = 0; c < NCH; c++) {
                const float *p = plane0 + c * s1c;
                res[j][c] = g.w00[j] * p[s.T * s1h + s.L] + g.w01[j] * p[s.T * s1h + s.R] +
                            g.w10[j] * p[s.Bm * s1h + s.L] + g.w11[j] * p[s.Bm * s1h + s.R];
            }
        }
        if (!valid) res[j] = f32x4{0.f, 0.f, 0.f, 0.f};     // out-of-range site -> 0 (my_lib_kernel.cu:566-570)
    }
#pragma unroll
    for (int c = 0; c < NCH; c++)
        st_stream4(out_p + c * s1c, f32x4{res[0][c], res[1][c], res[2][c], res[3][c]});
}

// (64 x 32 tiles on 512 lanes, which pay for the RGB backward, lose here: 234 against 201 us -- this kernel is bound by the
// number of independent tile chains per CU, see the launcher -- profiles/r03_bl_bwd_ab.txt.)
// RAG: a ragged width (W % 4 != 0, round 5) -- this kernel serves the whole quads, sites x < W & ~3, with the image's true
// width in every clamp, validity test and staged box (whose last quad is loaded ragged-safely: memc_tile.hpp); the one to
// three columns behind them go to the one-lane-per-site kernel (launcher).
template <int CT, int CAP, bool RAG = false>
__global__ __launch_bounds__(256, 2) void bl_fwd_tiled(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h,
    const float *__restrict__ in1, const float *__restrict__ flow, float *__restrict__ out, int sw)
{
    constexpr int LX = 16;
    using G = TileGeom<LX, CAP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    int *bb = reinterpret_cast<int *>(smem + G::kCapPx * 16);

    const TileCoord tc = tile_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, sw);
    if (tc.tx >= tiles_x) return;
    const int b = tc.b, tile_x0 = tc.tx * G::kTW, tile_y0 = tc.ty * G::kTH;
    const int x = tile_x0 + 4 * (threadIdx.x % LX), y = tile_y0 + threadIdx.x / LX;
    const int Ws = RAG ? W & ~3 : W;
    const bool inb = x < Ws && y < H;
    const int xs = min(x, Ws - 4), ys = min(y, H - 1);
    const float *flow_p = flow + b * s2b + (int64_t)ys * s2h + xs;
    const f32x4 fx4 = ld_stream4(flow_p), fy4 = ld_stream4(flow_p + s2c);

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
    Region r = tile_region<LX, true, CAP>(cmin, cmax, rmin, rmax, tile_x0, tile_y0, bb);
    r.wimg = RAG ? W : 0;
    BlSite4 g;
    g.valid = g.staged = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const BlSite &s = st[j];
        const bool staged = s.valid && r.covers(s.L, s.R, s.T, s.Bm);
        g.valid |= (s.valid ? 1u : 0u) << j;
        g.staged |= (staged ? 1u : 0u) << j;
        const int rT = staged ? (s.T - r.y0) * r.pitch : 0, rB = staged ? (s.Bm - r.y0) * r.pitch : 0;
        const int cL = staged ? swz_col(s.L - r.x0) : 0, cR = staged ? swz_col(s.R - r.x0) : 0;
        g.oTL[j] = rT + cL;  g.oTR[j] = rT + cR;  g.oBL[j] = rB + cL;  g.oBR[j] = rB + cR;
        g.w00[j] = (1 - s.a) * (1 - s.b);  g.w01[j] = s.a * (1 - s.b);
        g.w10[j] = (1 - s.a) * s.b;        g.w11[j] = s.a * s.b;
    }
    const float *in_b = in1 + b * s1b;
    float *out_p = out + b * s1b + (int64_t)y * s1h + x;
    if (CT == 3) {
        bl_fwd_chunk<3, RAG>(r, g, st, inb, in_b, out_p, s1c, s1h, tile);
    } else {
        int c0 = 0;
#pragma unroll 1
        for (; c0 + 4 <= C; c0 += 4) {
            if (c0 > 0) __syncthreads();
            // keep what only depends on the site inside the loop: hoisted, the per-plane addresses derived from it
            // (staging rows, the rare global corner gathers) outgrow the register file and spill
#pragma unroll
            for (int j = 0; j < 4; j++) {
                asm volatile("" : "+v"(g.oTL[j]), "+v"(g.oTR[j]), "+v"(g.oBL[j]), "+v"(g.oBR[j]));
                asm volatile("" : "+v"(st[j].L), "+v"(st[j].R), "+v"(st[j].T), "+v"(st[j].Bm));
            }
            bl_fwd_chunk<4, RAG>(r, g, st, inb, in_b + c0 * s1c, out_p + c0 * s1c, s1c, s1h, tile);
        }
        if (c0 < C) {
            if (c0 > 0) __syncthreads();
            const int nch = C - c0;
            if (nch == 3)      bl_fwd_chunk<3, RAG>(r, g, st, inb, in_b + c0 * s1c, out_p + c0 * s1c, s1c, s1h, tile);
            else if (nch == 2) bl_fwd_chunk<2, RAG>(r, g, st, inb, in_b + c0 * s1c, out_p + c0 * s1c, s1c, s1h, tile);
            else               bl_fwd_chunk<1, RAG>(r, g, st, inb, in_b + c0 * s1c, out_p + c0 * s1c, s1c, s1h, tile);
        }
    }
}

#ifdef MEMC_MEASURE
#include "arms/bl_bwd_arms.hpp"           // bl_bwd_tiled_c3 (rounds 1-2): measurement build only
#endif

// --------------------------------------------------------------------------------------------------
// Backward, RGB, round 3: packed fixed-point planes (memc_pk.hpp), image gradient FIRST -- the scheme of fi_bwd_c3.hip
// with a 2 x 2 footprint.  The bilinear weights are <= 1, so the tile's block exponent comes from its largest
// |gradoutput| alone.  48 KiB of LDS: the two planes, then -- the same bytes -- the staged image; pitch by the box's
// width (96 x 32, 80 x 38 or 64 x 48 cells; the fp64 plane of rounds 1-2 had a fixed pitch).
//   load (planes zeroed meanwhile) -> box -> request image rows -> adds -> flush -> rows to LDS -> flow gradient.
// Sites whose corners fall outside the (clipped) box scatter with global atomics and gather from global memory, as
// before; a tile with a non-finite gradoutput scatters everything with global atomics.
// --------------------------------------------------------------------------------------------------
// NT lanes take a tile of 64 x NT/16 sites; CAP is the staging budget in cells.  Product: 512 lanes, 64 x 32 tiles, 4096 cells
// (64 KiB, two workgroups of eight waves per CU).  Measurement arms: 256 lanes with 3072 (48 KiB, 3 per CU) or 2496 (39 KiB, 4).
template <int CAP, int NT = 256, bool RAG = false>          // RAG: a ragged width, see bl_fwd_tiled; both kernels ADD into gradinput1
__global__ __launch_bounds__(NT, NT == 256 && CAP == 3072 ? 3 : 4) void bl_bwd_c3_pk(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ gout,
    float *__restrict__ gin1, float *__restrict__ gin2, int sw)
{
    constexpr int LX = 16;
    using G = TileGeom<LX, CAP, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    unsigned long long *const accA = reinterpret_cast<unsigned long long *>(smem);   // the planes alias the image
    unsigned long long *const accB = accA + CAP;
    int *bb = reinterpret_cast<int *>(smem + G::kCapPx * 16);            // 4 ints per wave: boxes; then 4 per wave: bound statistics
    int *mx = bb + 4 * (NT / kWave);

    const TileCoord tc = tile_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, sw);
    if (tc.tx >= tiles_x) return;
    const int b = tc.b, tile_x0 = tc.tx * G::kTW, tile_y0 = tc.ty * G::kTH;
    const unsigned tid = tid_now();
    const int x = tile_x0 + 4 * (int)(tid % LX), y = tile_y0 + (int)(tid / LX);
    const int Ws = RAG ? W & ~3 : W;
    const bool inb = x < Ws && y < H;
    const int xs = min(x, Ws - 4), ys = min(y, H - 1);
    const float *flow_p = flow + b * s2b + (int64_t)ys * s2h + xs;
    const f32x4 fx4 = ld_stream4(flow_p), fy4 = ld_stream4(flow_p + s2c);
    const float *gout_p = gout + b * s1b + (int64_t)ys * s1h + xs;
    f32x4 go[3];
#pragma unroll
    for (int c = 0; c < 3; c++) go[c] = ld_stream4(gout_p + c * s1c);
    {                                          // both planes, while the loads are in flight
        f32x4 *pz = reinterpret_cast<f32x4 *>(smem);
#pragma unroll
        for (int i = 0; i < (CAP + NT - 1) / NT; i++)
            if (CAP % NT == 0 || (int)tid + i * NT < CAP) pz[tid + i * NT] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

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
    // per-site bounds of the packed planes (memc_pk.hpp): the bilinear weights are <= 1, so a site's bound is its largest
    // |gradoutput|; published per wave and handed over by the barrier inside tile_region
    int sbits[4];
    unsigned vmask = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int mg = 0;
#pragma unroll
        for (int c = 0; c < 3; c++) mg = max(mg, __float_as_int(go[c][j]) & 0x7FFFFFFF);
        sbits[j] = mg;                         // (>= 0x7F800000: Inf / NaN -- per-site global atomics)
        vmask |= (st[j].valid ? 1u : 0u) << j;
    }
    pk_tile_publish(mx, tid, sbits, sbits, vmask, 0x3F800000);
    Region r = tile_region<LX, true, CAP, NT>(cmin, cmax, rmin, rmax, tile_x0, tile_y0, bb);
    r.wimg = RAG ? W : 0;
    const PkTile ps = pk_tile_resolve<NT / kWave>(mx);
    // packed: through the planes; everything else with a non-zero bound (beyond the tile's block exponent, or not finite)
    // scatters with global atomics, exactly as the reference does
    const unsigned packed = pk_packed_sites(ps, sbits, vmask);
    const int mode = ps.any;                   // 0: no packed site has anything to add (workgroup-uniform)
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    const StageSlot sl = stage_slots<NT>(r);
    StageRegs<3> sr;
    tile_stage_load<3, RAG>(r, sl, in_b, s1c, s1h, sr);        // in flight during adds and flush

    // ---- image gradient: 4 corners x 3 colours per site = 8 packed LDS adds
    unsigned staged_mask = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!st[j].valid) continue;
        const BlSite &s = st[j];
        const bool staged = r.covers(s.L, s.R, s.T, s.Bm);
        staged_mask |= (staged ? 1u : 0u) << j;
        if (sbits[j] == 0) continue;           // a zero gradient adds nothing
        if (staged && ((packed >> j) & 1u)) {
            const int aT = (s.T - r.y0) * r.pitch, aB = (s.Bm - r.y0) * r.pitch;
            const int aL = pk_col(s.L - r.x0, r.pitch >> 2), aR = pk_col(s.R - r.x0, r.pitch >> 2);
            const float g0 = ps.sa * go[0][j], g1 = ps.sa * go[1][j], g2 = ps.sa * go[2][j];
            pk_add3(accA, accB, aT + aL, g0, g1, g2, ps.sb * ((1 - s.a) * (1 - s.b)));
            pk_add3(accA, accB, aT + aR, g0, g1, g2, ps.sb * (s.a * (1 - s.b)));
            pk_add3(accA, accB, aB + aL, g0, g1, g2, ps.sb * ((1 - s.a) * s.b));
            pk_add3(accA, accB, aB + aR, g0, g1, g2, ps.sb * (s.a * s.b));
        } else {
#pragma unroll 1
            for (int c = 0; c < 3; c++) {
                const float gv = c == 0 ? go[0][j] : (c == 1 ? go[1][j] : go[2][j]);
                float *q = gin1_b + c * s1c;
                atomic_add_f32(q + s.T * s1h + s.L, gv * (1 - s.a) * (1 - s.b));
                atomic_add_f32(q + s.T * s1h + s.R, gv * s.a * (1 - s.b));
                atomic_add_f32(q + s.Bm * s1h + s.L, gv * (1 - s.a) * s.b);
                atomic_add_f32(q + s.Bm * s1h + s.R, gv * s.a * s.b);
            }
        }
    }
    if (mode == 1) {                           // (workgroup-uniform)
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kStageIts; it++)                 // the staged rows have landed: take the wait here, not
#pragma unroll                                                 // behind the flush's conditional atomics (vmcnt is in order)
            for (int c = 0; c < 3; c++)
                asm volatile("" : "+v"(sr.v[it][c][0]), "+v"(sr.v[it][c][1]), "+v"(sr.v[it][c][2]), "+v"(sr.v[it][c][3]));
        pk_flush<NT>(r, accA, accB, ps.inv, gin1_b, s1c, s1h);
    }
    __syncthreads();                           // the planes have been read (or never used): the LDS becomes the image
    tile_stage_store<3, RAG>(r, sl, sr, tile);
    __syncthreads();

    // ---- flow gradient from the four corner values
    f32x4 gx4 = {0.f, 0.f, 0.f, 0.f}, gy4 = gx4;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!st[j].valid) continue;
        const BlSite &s = st[j];
        const float x2 = (float)(x + j) + fx4[j], y2 = (float)y + fy4[j];
        const float gam_x = (float)s.Bm - y2, gam_y = (float)s.R - x2;   // clamped corners, my_lib_kernel.cu:634,652
        f32x4 vTL, vTR, vBL, vBR;
        if ((staged_mask >> j) & 1) {
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
}

static int launch_bl_fwd(hipStream_t stream, int w, int h, int channel, int batch, int s1b, int s1c, int s1h,
                         int s2b, int s2c, int s2h, const float *input1, const float *input2, float *output)
{
    if (w <= 0 || h <= 0 || channel <= 0 || batch <= 0) return 0;
    // A width that is not a multiple of four (round 5): the tiled kernel takes the whole quads (sites x < ws), the one-lane-
    // per-site kernel the one to three columns behind them.
    const int ws = w & ~3;
    const int tail_y = (h + 3) / 4;
    if (ws >= 4) {
        using G = TileGeom<16>;
        const int ntx = (ws + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;
        const int sw = g_tile_walk_sw >= 0 ? g_tile_walk_sw : kDefaultStripe;
        MEMC_PATH(channel == 3 ? "bl_fwd:tiled_c3" : "bl_fwd:tiled_chunks");
#define MEMC_BL_FWD_R(CT, CAP, RAG)                                                                             \
            hipLaunchKernelGGL((bl_fwd_tiled<CT, CAP, RAG>), dim3(walk_grid(ntx, nty, batch, sw)), dim3(256),       \
                               (tile_lds_bytes<16, CAP>() + g_extra_lds), stream, w, h, channel, ntx, nty, (int64_t)s1b, \
                               (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2, output, sw)
#define MEMC_BL_FWD(CT, CAP)                                                                                    \
            do {                                                                                                    \
                if (ws < w) MEMC_BL_FWD_R(CT, CAP, true);                                                           \
                else MEMC_BL_FWD_R(CT, CAP, false);                                                                 \
            } while (0)
        if (channel == 3) {
            // 39 KiB instead of 48: 4 workgroups per CU.  The kernel is bound by the latency of a tile's serial chain
            // (1 / 2 / 3 per CU: 483 / 280 / 215 us), its 2x2 footprint rarely needs the rows given up: 218 -> 190 us
#ifdef MEMC_MEASURE
            if (g_cap_sel == 0) MEMC_BL_FWD(3, 3072);
            else if (g_cap_sel == 2) MEMC_BL_FWD(3, 1984);     // 5 per CU: 198-202 us
            else
#endif
            MEMC_BL_FWD(3, 2496);
        } else {
            MEMC_BL_FWD(0, 3072);
        }
#undef MEMC_BL_FWD
#undef MEMC_BL_FWD_R
        if (ws < w) {                          // the ragged row's last columns
            if (channel == 3)
                hipLaunchKernelGGL(bl_fwd<3>, dim3((unsigned)tail_y * batch), dim3(256), 0, stream, w, h, channel, 1, tail_y,
                                   (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2, output, ws);
            else
                hipLaunchKernelGGL(bl_fwd<0>, dim3((unsigned)tail_y * batch), dim3(256), 0, stream, w, h, channel, 1, tail_y,
                                   (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2, output, ws);
        }
        return launch_status();
    }
    const int tiles_x = (w + kWave - 1) / kWave, tiles_y = (h + 3) / 4;
    const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
    MEMC_PATH("bl_fwd:direct");
    if (channel == 3)
        hipLaunchKernelGGL(bl_fwd<3>, dim3(nwg), dim3(256), 0, stream, w, h, channel, tiles_x, tiles_y,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2, output, 0);
    else
        hipLaunchKernelGGL(bl_fwd<0>, dim3(nwg), dim3(256), 0, stream, w, h, channel, tiles_x, tiles_y,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2, output, 0);
    return launch_status();
}

#ifdef MEMC_MEASURE
static int g_bl_bwd_direct = 0;                // A/B: the direct (global atomics) kernel for any channel count
#endif

static int launch_bl_bwd(hipStream_t stream, int w, int h, int channel, int batch, int s1b, int s1c, int s1h,
                         int s2b, int s2c, int s2h, const float *input1, const float *input2,
                         const float *gradoutput, float *gradinput1, float *gradinput2)
{
    if (w <= 0 || h <= 0 || channel <= 0 || batch <= 0) return 0;
    const int ws = w & ~3;                                 // (a ragged width: see launch_bl_fwd)
    const int tail_y = (h + 3) / 4;
#ifdef MEMC_MEASURE
    const bool arm_needs_quads = (g_cap_sel == 0 || g_cap_sel == 1) && ws < w;    // (the rounds 1-2 kernel: whole widths only)
#else
    constexpr bool arm_needs_quads = false;
#endif
    if (channel == 3 && g_cap_sel != 5 && plane_fits_u32(w, h, {s1h}) && ws >= 4 && !arm_needs_quads) {
        using G = TileGeom<16>;
        const int ntx = (ws + G::kTW - 1) / G::kTW;
        [[maybe_unused]] const int nty = (h + G::kTH - 1) / G::kTH;
        const int sw = g_tile_walk_sw >= 0 ? g_tile_walk_sw : kDefaultStripe;
        MEMC_PATH("bl_bwd:tiled_c3");
#ifdef MEMC_MEASURE
#define MEMC_BL_BWD(CAP)                                                                                        \
        hipLaunchKernelGGL(bl_bwd_tiled_c3<CAP>, dim3(walk_grid(ntx, nty, batch, sw)), dim3(256),                  \
                           (tile_lds_bytes<16, CAP>()), stream, w, h, ntx, nty, (int64_t)s1b, (int64_t)s1c, s1h,  \
                           (int64_t)s2b, (int64_t)s2c, s2h, input1, input2, gradoutput, gradinput1, gradinput2, sw)
#endif
        // Product: the packed-plane kernel.  The fp64-plane kernel of rounds 1-2 stays as a measurement arm (bl_cap 0: its
        // 48 KiB budget; 1: 39 KiB, which LOSES, 522 -> 601 us: the fixed-pitch plane gets 26 rows instead of 32)
#ifdef MEMC_MEASURE
        if (g_cap_sel == 1) MEMC_BL_BWD(2496);
        else if (g_cap_sel == 0) MEMC_BL_BWD(3072);
        else
#endif
#define MEMC_BL_BWD_PK_R(CAP, NT, RAG)                                                                          \
        do {                                                                                                       \
            constexpr size_t lds = CAP * 16 + 32 * (NT / kWave);    /* image or planes; 32 bytes per wave: box, bounds */ \
            allow_big_lds(bl_bwd_c3_pk<CAP, NT, RAG>, lds);    /* per launch: the attribute belongs to the CURRENT device */ \
            const int ntyk = (h + NT / 16 - 1) / (NT / 16);                                                        \
            hipLaunchKernelGGL((bl_bwd_c3_pk<CAP, NT, RAG>), dim3(walk_grid(ntx, ntyk, batch, sw)), dim3(NT), lds, \
                               stream, w, h, ntx, ntyk, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, \
                               s2h, input1, input2, gradoutput, gradinput1, gradinput2, sw);                       \
        } while (0)
#define MEMC_BL_BWD_PK(CAP, NT)                                                                                 \
        do {                                                                                                       \
            if (ws < w) MEMC_BL_BWD_PK_R(CAP, NT, true);                                                           \
            else MEMC_BL_BWD_PK_R(CAP, NT, false);                                                                 \
        } while (0)
        // 64 x 32 tiles on 512 lanes, 64 KiB, two workgroups per CU.  The flush's global atomics bound this kernel, and a
        // bigger tile's box holds fewer cells per site: against 64 x 16 tiles on 256 lanes (bl_cap 4; 3: the same in
        // 39 KiB) 526 -> 484 us smooth, 759 -> 640 i.i.d., 478 -> 459 video; 64 x 64 tiles on 1024 lanes, one workgroup per CU
        // and nothing to overlap its phases with, lose: 548 / 885 / 512 (profiles/r03_bl_bwd_ab.txt).
#ifdef MEMC_MEASURE
        if (g_cap_sel == 3) MEMC_BL_BWD_PK(2496, 256);
        else if (g_cap_sel == 4) MEMC_BL_BWD_PK(3072, 256);
        else
#endif
        MEMC_BL_BWD_PK(4096, 512);
#undef MEMC_BL_BWD_PK
#undef MEMC_BL_BWD_PK_R
#undef MEMC_BL_BWD
        if (ws < w)                            // the ragged row's last columns: both kernels ADD into gradinput1
            hipLaunchKernelGGL(bl_bwd<3>, dim3((unsigned)tail_y * batch), dim3(256), 0, stream, w, h, channel, 1, tail_y,
                               (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2,
                               gradoutput, gradinput1, gradinput2, ws);
        return launch_status();
    }
#ifdef MEMC_MEASURE
    g_bwd_cn_allow_c3 = g_cap_sel == 5;                    // arm: RGB through the owner kernels (one chunk, a padded plane)
    if (channel != 3 || g_cap_sel == 5) {
#else
    if (channel != 3) {                                    // many channels: fi_bwd_cn.hip (owner-computes)
#endif
#ifdef MEMC_MEASURE
        const bool direct_only = g_bl_bwd_direct != 0;
#else
        constexpr bool direct_only = false;
#endif
        const int taken = bl_bwd_cn_launch(stream, w, h, channel, batch, s1b, s1c, s1h, s2b, s2c, s2h, input1, input2,
                                           gradoutput, gradinput1, gradinput2, direct_only);
        if (taken != 0) {
            MEMC_PATH("bl_bwd:owner");
            return taken > 0 ? 0 : -1;
        }
    }
    const int tiles_x = (w + kWave - 1) / kWave, tiles_y = (h + 3) / 4;
    const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
    MEMC_PATH("bl_bwd:direct");
    if (channel == 3)
        hipLaunchKernelGGL(bl_bwd<3>, dim3(nwg), dim3(256), 0, stream, w, h, channel, tiles_x, tiles_y,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2,
                           gradoutput, gradinput1, gradinput2, 0);
    else
        hipLaunchKernelGGL(bl_bwd<0>, dim3(nwg), dim3(256), 0, stream, w, h, channel, tiles_x, tiles_y,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1, input2,
                           gradoutput, gradinput1, gradinput2, 0);
    return launch_status();
}

}  // namespace memc

using namespace memc;

#ifdef MEMC_MEASURE
int memc::g_tile_walk_sw = -1;
int memc::g_extra_lds = 0;
int memc::g_cap_sel = -1;
extern "C" void memc_debug_set_bl_cap(int which) { memc::g_cap_sel = which; }
extern "C" void memc_debug_set_bl_bwd_direct(int on) { memc::g_bl_bwd_direct = on; }
extern "C" void memc_debug_set_extra_lds(int bytes) { memc::g_extra_lds = bytes > 0 ? bytes : 0; }
extern "C" void memc_debug_set_walk(int stripe_width) { memc::g_tile_walk_sw = stripe_width; }
#endif

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
