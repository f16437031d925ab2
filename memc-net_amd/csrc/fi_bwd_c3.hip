// fi_bwd_c3.hip -- FilterInterpolation backward, RGB (C == 3), fs == 4: the LDS-tiled kernels.
//
// Replaces my_package/src/my_lib_kernel.cu:1220-1518 (kernel) / :1571-1627 (launcher) of the reference for the only
// channel count its networks back-propagate through (networks/MEMC_Net_star.py:266-277; the context warps are
// detached, :285).  Semantics: SURVEY.md appendix A.2.
#include "memc_common.hpp"
#include "memc_internal.h"
#include "memc_tile.hpp"
#include "memc_fi.hpp"

namespace memc {

// --------------------------------------------------------------------------------------------------
// Backward, fs == 4, RGB, LDS-tiled and vectorised.  Same tile / box machinery as the forward kernel:
//   * streams (flow, 16 tap planes, 3 gradoutput planes) as dwordx4;
//   * the image box is staged into LDS pixel quads (needed for the tap and flow gradients);
//   * the image gradient -- 48 scattered adds per site -- goes, one colour channel at a time, into transposed fp64
//     LDS accumulator planes (ds_add_f64: twenty times the rate of ds_add_f32 on this chip; AccT in
//     memc_tile.hpp) and is flushed once per cell, rounded to fp32, with row-coalesced global atomics;
//   * gradinput3 (each site owns its taps) is stored once per site as dwordx4 (the caller zero-fills it);
//     gradinput2 is assigned.
// Sites whose window is not staged are redone by fi_bwd_site_scalar with global atomics.
// --------------------------------------------------------------------------------------------------
// Per-workgroup phase timestamps (shader clock) for tools/trace_kernel.py; written by the ABL == 9 arm only.
__device__ unsigned long long *g_trace_buf = nullptr;
constexpr int kTraceSlots = 16;
template <bool ON>
__device__ __forceinline__ void trace_mark(int slot)
{
    if (ON && threadIdx.x == 0) g_trace_buf[(size_t)blockIdx.x * kTraceSlots + slot] = __builtin_readcyclecounter();
}

// ABL != 0 are MEASUREMENT arms (tools/bench_ops.py --bwd-variants; their results are wrong by construction):
//   1 no fp64 LDS adds (zero + flush kept; zero cells are not flushed)   2 no phase 2 at all
//   3 phase 1 without its LDS reads                                      5 flush with plain stores
//   4 accumulate but never flush (plane re-zeroed instead)
//   9 production + phase timestamps
// gradinput3 and gradinput2 of ONE site straight from global memory (mixed quads of the tiled backward: some of a
// lane's four sites belong to another band or are invalid).  Assigns both, like the tiled path; the image
// gradient of such a site still goes through the tiled phase 2.
__device__ __noinline__ void fi_bwd_site_taps(int x, int y, int W, int H, const float *in_b, int64_t s1c, int s1h,
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

struct FiBwdIn {
    f32x4 fx, fy, go[3], tp[16];
};

// Phase 1 of one band: tap and flow gradients of the sites in `fast` from the staged image.
// With s = sum_c g_c * in_c(tap cell) (3 FMAs per tap), and q the tap's quadrant:
//     gradinput3[tap] = wq * s,   gradinput2.x = sum_taps cx[q] * s * tap,   gradinput2.y likewise,
// where wq = {(1-a)(1-b), a(1-b), (1-a)b, ab}, cx = {-(1-b), (1-b), -b, b}, cy = {-(1-a), -a, (1-a), a}.
// (The reference sums per channel first -- same value up to fp32 re-association, ~1e-7 relative.)
// Tap rows are the outer loop so that only one row of tap gradients (4 float4) is live at a time.
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
    for (int k = 0; k < 16; k++) *reinterpret_cast<f32x4 *>(q3 + k * s3c) = z;
    *reinterpret_cast<f32x4 *>(q2) = z;
    *reinterpret_cast<f32x4 *>(q2 + s2c) = z;
}

template <int ABL>
__device__ __forceinline__ void fi_bwd_phase1(const Region &r, unsigned fast, FiSite4 &g, f32x4 (&tp)[16],
                                              const f32x4 (&go)[3], const f32x4 *tile, int W, int H,
                                              float *gin2_b, int64_t s2c, unsigned o2, float *gin3_b, int64_t s3c,
                                              unsigned o3)
{
    // keep tap splats / weights inside the caller's band loop (hoisted, they spill)
#pragma unroll
    for (int k = 0; k < 16; k++)
        asm volatile("" : "+v"(tp[k][0]), "+v"(tp[k][1]), "+v"(tp[k][2]), "+v"(tp[k][3]));
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("" : "+v"(g.ix[j]), "+v"(g.iy[j]), "+v"(g.a[j]), "+v"(g.b[j]));
    // Only quads that this band owns completely (the common case) take this path -- ONE exec-masked region
    // without inner control flow, every store unconditional (the buffers are zero-filled by the caller:
    // 0 + g == g); mixed quads are redone per site by fi_bwd_site_taps.  Any load or data-dependent merge inside
    // the nest makes the compiler split it and spill the partial sums.
    if (fast != 0xFu) return;
    f32x4 gx4 = {0.f, 0.f, 0.f, 0.f}, gy4 = gx4;
    // Loop order (tap row, tap column, site): one float4 of tap gradients is live at a time and four image reads
    // are in flight; cell addresses are recomputed per use (the asm keeps them from being CSE'd into a table) --
    // the kernel lives or dies by fitting 168 registers without a spill.
#pragma unroll
    for (int k = 0; k < 4; k++) {
        int ro[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            asm volatile("" : "+v"(g.ix[j]));
            ro[j] = (clampi(g.iy[j] - 1 + k, H - 1) - r.y0) * r.pitch;
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
            f32x4 gt;                          // gt[j]: gradient of tap (k, m) of site j
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const float a = g.a[j], bt = g.b[j];
                const int co = swz_col(clampi(g.ix[j] - 1 + m, W - 1) - r.x0);
                const f32x4 pix = ABL == 3 ? f32x4{a, bt, a, bt} : tile[ro[j] + co];
                float sv = 0.0f;
                sv += go[0][j] * pix[0];  sv += go[1][j] * pix[1];  sv += go[2][j] * pix[2];
                const float wa = m < 2 ? (1 - a) : a, wb = k < 2 ? (1 - bt) : bt;
                gt[j] = (wa * wb) * sv;
                const float st = sv * tp[k * 4 + m][j];
                gx4[j] += (m < 2 ? -wb : wb) * st;
                gy4[j] += (k < 2 ? -wa : wa) * st;
            }
            st_stream4_u(gin3_b + (k * 4 + m) * s3c, o3, gt);
        }
    }
    st_stream4_u(gin2_b, o2, gx4);             // gradinput2 is ASSIGNED
    st_stream4_u(gin2_b + s2c, o2, gy4);
}

// The 16 ds_add_f64 of channel c of the sites in `fast` into the transposed plane `acc` (AccT, memc_tile.hpp).
template <int ABL>
__device__ __forceinline__ void fi_bwd_adds(const Region &r, unsigned fast, FiSite4 &g, const f32x4 (&tp)[16],
                                            const f32x4 (&go)[3], int c, double *acc, int W, int H)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (ABL == 1 || !((fast >> j) & 1)) continue;
        // keep the cell addresses and weights inside the caller's loops (hoisted, they spill)
        asm volatile("" : "+v"(g.ix[j]), "+v"(g.iy[j]), "+v"(g.a[j]), "+v"(g.b[j]));
        int ro[4], co[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ro[k] = (clampi(g.iy[j] - 1 + k, H - 1) - r.y0) * AccT::kPitch;
            co[k] = acct_col(clampi(g.ix[j] - 1 + k, W - 1) - r.x0);
        }
        const float a = g.a[j], bt = g.b[j];
        const float gv = c == 0 ? go[0][j] : (c == 1 ? go[1][j] : go[2][j]);
        const float wq[4] = {gv * (1 - a) * (1 - bt), gv * a * (1 - bt), gv * (1 - a) * bt, gv * a * bt};
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int m = 0; m < 4; m++)
                lds_add_f64(acc + ro[k] + co[m], (double)(wq[(k >> 1) * 2 + (m >> 1)] * tp[k * 4 + m][j]));
    }
}

// One 64x16 tile per workgroup; 48 KiB of LDS: the staged image, then ONE accumulator plane that the colour
// channels take in turn.  Measured on MI355X, 720p batch 32 (tools/bench_ops.py --bwd-variants):
//   this kernel, 2 workgroups / CU (209 VGPRs)                                  1.82 ms
//   MINW = 3: 3 workgroups / CU at 168 VGPRs (152 B of spills)        (arm 16)  1.93 ms
//   persistent, 2 / CU, next tile's inputs prefetched during phase 2  (arm 10)  1.83 ms (2.09 with this phase 1)
//   persistent without the prefetch                                   (arm 11)  1.83 ms (1.98)
//   second workgroup of every CU delayed by half a tile; wave priority rising through phase 2      no change
//   no phase 2 at all                                                 (arm 2)   1.05 - 1.3 ms (the HBM floor)
// i.e. the time is phase 1 (HBM bound) PLUS the LDS-atomic work of phase 2, however the two are arranged: what
// is left to gain is in the number and the conflict rate of the ds_add_f64 (768 wave-instructions per tile at
// ~15 clk), not in latency hiding.
template <int ABL, int MINW = 2>
__global__ __launch_bounds__(256, MINW) void fi_bwd_tiled_c3(
    int W, int H, int tiles_x, int tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    const float *__restrict__ gout, float *__restrict__ gin1, float *__restrict__ gin2,
    float *__restrict__ gin3)
{
    constexpr int LX = 16;
    constexpr bool TR = ABL == 9;
    using G = TileGeom<LX>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    double *const acc = reinterpret_cast<double *>(smem);        // aliases the image: phase 2 needs taps only
    int *bb = reinterpret_cast<int *>(smem + G::kCapPx * 16);

    trace_mark<TR>(0);
    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, batch);
    const int b = tc.b;
    const unsigned tid = tid_now();
    const int x = tc.tx * G::kTW + 4 * (int)(tid % LX), y = tc.ty * G::kTH + (int)(tid / LX);
    const bool inb = x < W && y < H;
    const int xs = min(x, W - 4), ys = min(y, H - 1);
    // wave-uniform plane bases + one 32-bit byte offset per tensor (see ld_stream4_u)
    const float *flow_b = flow + b * s2b, *filt_b = filt + b * s3b, *gout_b = gout + b * s1b;
    float *gin2_b = gin2 + b * s2b, *gin3_b = gin3 + b * s3b;
    const unsigned o1 = 4u * (unsigned)(ys * s1h + xs), o2 = 4u * (unsigned)(ys * s2h + xs),
                   o3 = 4u * (unsigned)(ys * s3h + xs);
    f32x4 go[3], tp[16];
    const f32x4 fx4 = ld_stream4_u(flow_b, o2), fy4 = ld_stream4_u(flow_b + s2c, o2);
#pragma unroll
    for (int c = 0; c < 3; c++) go[c] = ld_stream4_u(gout_b + c * s1c, o1);
#pragma unroll
    for (int k = 0; k < 16; k++) tp[k] = ld_stream4_u(filt_b + k * s3c, o3);
    if (TR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    trace_mark<TR>(1);                                         // inputs have arrived

    FiSite4 g;
    g.valid = 0;
    int cmin = INT_MAX, cmax = -1, rmin = INT_MAX, rmax = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const FiSite s = fi_locate(x + j, y, W, H, fx4[j], fy4[j]);
        g.ix[j] = s.ix; g.iy[j] = s.iy; g.a[j] = s.a; g.b[j] = s.b;
        if (inb && s.valid) {
            g.valid |= 1u << j;
            cmin = min(cmin, max(s.ix - 1, 0));  cmax = max(cmax, min(s.ix + 2, W - 1));
            rmin = min(rmin, max(s.iy - 1, 0));  rmax = max(rmax, min(s.iy + 2, H - 1));
        }
    }
    const BBox box = tile_bbox<LX>(cmin, cmax, rmin, rmax, bb);
    const Bands bands = make_bands<LX, false>(box);
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    unsigned done = 0;
    trace_mark<TR>(2);                                         // bounding box known
    fi_bwd_zero_invalid(inb, g.valid, gin2_b, s2c, o2, gin3_b, s3c, o3);
#pragma unroll 1
    for (int bi = 0; bi < bands.n; bi++) {
    const Region r = band_region(box, bands, bi);
    const unsigned fast = inb ? fi_covered(r, g, W, H) & ~done : 0u;
    // later bands run only if some site still needs them; the vote is also the barrier that frees the LDS
    if (bi > 0 && !__syncthreads_or(fast != 0)) continue;
    done |= fast;
    tile_stage<LX, 3>(r, in_b, s1c, s1h, tile);
    __syncthreads();
    if (bi == 0) trace_mark<TR>(3);                            // image staged
    fi_bwd_phase1<ABL>(r, fast, g, tp, go, tile, W, H, gin2_b, s2c, o2, gin3_b, s3c, o3);
    if (fast != 0xFu) {                        // mixed quads (rare): their tap gradients, site by site
        unsigned todo = fast;
        while (todo) {
            const int j = __ffs(todo) - 1;
            todo &= todo - 1;
            fi_bwd_site_taps(x + j, y, W, H, in_b, s1c, s1h, flow_b + o2 / 4 + j, gin2_b + o2 / 4 + j, s2c,
                             filt_b + o3 / 4 + j, gin3_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
        }
    }
    __syncthreads();                           // everybody is done reading the image: the LDS becomes `acc`
    if (bi == 0) trace_mark<TR>(4);                            // phase 1 done
    if (ABL == 2) continue;
    acct_zero<1>(acc);
    __syncthreads();
    if (bi == 0) trace_mark<TR>(5);                            // plane zeroed
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        fi_bwd_adds<ABL>(r, fast, g, tp, go, c, acc, W, H);
        __syncthreads();
        if (bi == 0) trace_mark<TR>(6 + 2 * c);                // channel c accumulated
        if (ABL == 4) acct_zero<1>(acc);                            // measurement: accumulate, never flush
        else acct_flush_zero<ABL == 5>(r, acc, gin1_b + c * s1c, s1h);    // leaves the plane zeroed for the next channel
        __syncthreads();
        if (bi == 0) trace_mark<TR>(7 + 2 * c);                // channel c flushed
    }
    }   // bands
    trace_mark<TR>(12);
    unsigned slow = inb ? g.valid & ~done : 0u;            // not coverable within kMaxBands bands
    while (slow) {                            // rare: redone from global memory with global atomics
        const int j = __ffs(slow) - 1;
        slow &= slow - 1;
        fi_bwd_site_scalar(x + j, y, W, H, 3, 4, in_b, gin1_b, s1c, s1h, flow_b + o2 / 4 + j, gin2_b + o2 / 4 + j,
                           s2c, filt_b + o3 / 4 + j, gin3_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
    }
}

// Persistent variant (measurement arms 10 / 11): 2 workgroups per CU walk the tiles w, w + grid, ... (grid % 8 == 0:
// a workgroup stays on its XCD's chunk of the strip order); phase 1 of all bands, then -- arm 10 -- the 21 float4 of
// per-site inputs of the NEXT tile are requested so that they arrive while phase 2 runs on the LDS (two planes, the
// flush of one channel overlapping the adds of the next).  No faster than one tile per workgroup (table above).
template <bool PREFETCH>
__global__ __launch_bounds__(256, 2) void fi_bwd_tiled_c3_persistent(
    int W, int H, int tiles_x, int tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    const float *__restrict__ gout, float *__restrict__ gin1, float *__restrict__ gin2,
    float *__restrict__ gin3)
{
    constexpr int LX = 16;
    using G = TileGeom<LX>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    double *const plane0 = reinterpret_cast<double *>(smem);     // plane i at plane0 + i * AccT::kPlane
    int *bb = reinterpret_cast<int *>(smem + 2 * AccT::kPlane * 8);
    const unsigned ntiles = (unsigned)tiles_x * tiles_y * batch;

    // requests the per-site inputs of tile `v` (always a valid tile: loads stay unconditional)
    auto request = [&](unsigned v, FiBwdIn &in) {
        const TileCoord tc = strip_walk(v, ntiles, tiles_x, tiles_y, batch);
        const unsigned tid = tid_now();
        const int xs = min(tc.tx * G::kTW + 4 * (int)(tid % LX), W - 4), ys = min(tc.ty * G::kTH + (int)(tid / LX), H - 1);
        const float *flow_b = flow + tc.b * s2b, *filt_b = filt + tc.b * s3b, *gout_b = gout + tc.b * s1b;
        const unsigned o1 = 4u * (unsigned)(ys * s1h + xs), o2 = 4u * (unsigned)(ys * s2h + xs),
                       o3 = 4u * (unsigned)(ys * s3h + xs);
        in.fx = ld_stream4_u(flow_b, o2);
        in.fy = ld_stream4_u(flow_b + s2c, o2);
#pragma unroll
        for (int c = 0; c < 3; c++) in.go[c] = ld_stream4_u(gout_b + c * s1c, o1);
#pragma unroll
        for (int k = 0; k < 16; k++) in.tp[k] = ld_stream4_u(filt_b + k * s3c, o3);
    };

    unsigned v = blockIdx.x;
    FiBwdIn nx;
    if (PREFETCH) request(v, nx);
#pragma unroll 1
    for (;;) {
    FiBwdIn in;
    if (PREFETCH) in = nx; else request(v, in);
    const TileCoord tc = strip_walk(v, ntiles, tiles_x, tiles_y, batch);
    const int b = tc.b;
    const unsigned tid = tid_now();        // (and W, H below) opaque per tile: nothing derived from them is hoisted
    const int x = tc.tx * G::kTW + 4 * (int)(tid % LX), y = tc.ty * G::kTH + (int)(tid / LX);
    const bool inb = x < W && y < H;
    const int xs = min(x, W - 4), ys = min(y, H - 1);
    const float *flow_b = flow + b * s2b, *filt_b = filt + b * s3b, *gout_b = gout + b * s1b;
    float *gin2_b = gin2 + b * s2b, *gin3_b = gin3 + b * s3b;
    const unsigned o1 = 4u * (unsigned)(ys * s1h + xs), o2 = 4u * (unsigned)(ys * s2h + xs),
                   o3 = 4u * (unsigned)(ys * s3h + xs);
    FiSite4 g;
    g.valid = 0;
    int cmin = INT_MAX, cmax = -1, rmin = INT_MAX, rmax = -1;
    int Wl = W, Hl = H;
    asm volatile("" : "+s"(Wl), "+s"(Hl));
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const FiSite s = fi_locate(x + j, y, Wl, Hl, in.fx[j], in.fy[j]);
        g.ix[j] = s.ix; g.iy[j] = s.iy; g.a[j] = s.a; g.b[j] = s.b;
        if (inb && s.valid) {
            g.valid |= 1u << j;
            cmin = min(cmin, max(s.ix - 1, 0));  cmax = max(cmax, min(s.ix + 2, W - 1));
            rmin = min(rmin, max(s.iy - 1, 0));  rmax = max(rmax, min(s.iy + 2, H - 1));
        }
    }
    const BBox box = tile_bbox<LX>(cmin, cmax, rmin, rmax, bb);
    const Bands bands = make_bands<LX, false>(box);
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    fi_bwd_zero_invalid(inb, g.valid, gin2_b, s2c, o2, gin3_b, s3c, o3);
    unsigned done = 0, fastbits = 0;                           // fastbits: 4 bits per band, the sites it owns
#pragma unroll 1
    for (int bi = 0; bi < bands.n; bi++) {
        const Region r = band_region(box, bands, bi);
        const unsigned fast = inb ? fi_covered(r, g, W, H) & ~done : 0u;
        if (bi > 0 && !__syncthreads_or(fast != 0)) continue;
        done |= fast;
        fastbits |= fast << (4 * bi);
        tile_stage<LX, 3>(r, in_b, s1c, s1h, tile);
        __syncthreads();
        fi_bwd_phase1<0>(r, fast, g, in.tp, in.go, tile, W, H, gin2_b, s2c, o2, gin3_b, s3c, o3);
        if (fast != 0xFu) {                    // mixed quads (rare): their tap gradients, site by site
            unsigned todo = fast;
            while (todo) {
                const int j = __ffs(todo) - 1;
                todo &= todo - 1;
                fi_bwd_site_taps(x + j, y, W, H, in_b, s1c, s1h, flow_b + o2 / 4 + j, gin2_b + o2 / 4 + j, s2c,
                                 filt_b + o3 / 4 + j, gin3_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
            }
        }
        __syncthreads();                       // everybody is done reading the image
    }
    // next tile's inputs: in flight during phase 2 (the last iteration re-requests its own tile: unconditional)
    const unsigned vn = v + gridDim.x;
    if (PREFETCH) request(vn < ntiles ? vn : v, nx);
    acct_zero<2>(plane0);                      // the image was here; every flush below leaves its plane zeroed again
    __syncthreads();
#pragma unroll 1
    for (int bi = 0; bi < bands.n; bi++) {
        const unsigned fast = (fastbits >> (4 * bi)) & 0xFu;
        if (bi > 0 && !__syncthreads_or(fast != 0)) continue;
        const Region r = band_region(box, bands, bi);
#pragma unroll 1
        for (int c = 0; c < 4; c++) {          // channel c accumulates while channel c - 1 is flushed
            if (c > 0) acct_flush_zero<false>(r, plane0 + ((c - 1) & 1) * AccT::kPlane, gin1_b + (c - 1) * s1c, s1h);
            if (c < 3) fi_bwd_adds<0>(r, fast, g, in.tp, in.go, c, plane0 + (c & 1) * AccT::kPlane, W, H);
            __syncthreads();
        }
    }
    unsigned slow = inb ? g.valid & ~done : 0u;
    while (slow) {
        const int j = __ffs(slow) - 1;
        slow &= slow - 1;
        fi_bwd_site_scalar(x + j, y, W, H, 3, 4, in_b, gin1_b, s1c, s1h, flow_b + o2 / 4 + j, gin2_b + o2 / 4 + j,
                           s2c, filt_b + o3 / 4 + j, gin3_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
    }
    if (vn >= ntiles) break;
    v = vn;
    }   // tiles
}

// ---------------------------------------------------------------------------------------------------------
// Round 3: the image gradient in ONE LDS round instead of three.
//
// What the round-1/2 kernel (fi_bwd_tiled_c3, now a measurement arm) spends after its HBM-bound phase 1 is LDS
// atomic work: 48 ds_add_f64 per site (16 taps x 3 colours) into ONE fp64 plane that the colours take in turn --
// adds, barrier, flush, barrier, three times.  ds_add_f32 is an emulated path on this chip (0.33 lane-ops/clk/CU),
// 64-bit integer adds are the fastest LDS atomic there is (9.0; ds_add_f64: 6.7), so the three colours of a
// contribution are accumulated as 23-bit fixed-point numbers packed into TWO 64-bit words per cell:
//
//     word A = n0 * 2^26 + (n2 >> 11)            word B = n1 * 2^26 + (n2 & 2047)
//
// with n_c = round(g_c * wq * tap * 2^(22 - e)), |n_c| <= 2^22, where 2^e bounds every contribution of the TILE:
// 2^e > (the tile's largest |gradoutput|) x (its largest |tap|), by less than a factor two (the bilinear weights
// are <= 1).  Integer adds are exact and associative: a cell's sums come out bit-identical whatever order the LDS
// retires them in (the reference's fp32 atomics do not), scaling the inputs by a power of two scales the result by
// exactly that power, and nothing can overflow -- one site puts at most 9 of its 16 taps into one cell (the image
// corner, where the clamp folds 3 x 3 window positions), so a tile adds at most 1024 x 9 < 2^14 numbers to a cell:
// |sum n0| < 2^36 (word A holds 38 signed bits above bit 26), |sum (n2 >> 11)| < 2^25 (26-bit signed field),
// sum (n2 & 2047) < 2^25 (26-bit unsigned field).  Every contribution is rounded once, to a multiple of 2^(e - 22)
// <= 2^-21 x (the tile's largest possible contribution) -- about fp32's own resolution of that largest contribution;
// a cell's error is at most (its number of contributions) x 2^(e - 23).  32 adds per site instead of 48, both planes
// flushed in one pass: one round of adds / barrier / flush.
// A tile whose gradoutput or taps are not all finite takes per-site global atomics instead (NaN / Inf then land
// exactly where the reference puts them); a tile whose bound is zero has nothing to add.
// ---------------------------------------------------------------------------------------------------------
struct PkAcc {
    static constexpr int kMagic = 0x4B400000;                                  // bits of 1.5 * 2^23
    static constexpr int kShift = 26, kSplit = 11;
};
// A plane has the band's geometry: r.h rows of r.pitch 64-bit slots (pitch 96, 80 or 64: a multiple of 16 slots, so
// every row starts on the same bank).  Column c of a row is stored at (c & 3) * (pitch / 4) + (c >> 2): cells 4 apart
// (the sites of neighbouring lanes) are adjacent 8-byte slots -- see AccT in memc_tile.hpp.
__device__ __forceinline__ int pk_col(int c, int quarter) { return (c & 3) * quarter + (c >> 2); }

__device__ __forceinline__ void lds_add_u64(unsigned long long *p, unsigned long long v)
{
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_u64
}

// exponent e with |v| < 2^e for the non-negative float whose bits are `bits` (finite, non-zero), kept inside
// [-100, 128] so that every power of two formed from it is a normal float
__device__ __forceinline__ int pk_exponent(int bits)
{
    int e;
    (void)frexpf(__int_as_float(bits), &e);
    return max(e, -100);
}

// The 32 ds_add_u64 of the sites in `fast`.  sg = 2^(11 - e_g), st = 2^(11 - e_t): |g * sg| < 2^11, |w * st| <= 2^11.
__device__ __forceinline__ void fi_bwd_adds_pk(const Region &r, unsigned fast, FiSite4 &g, const f32x4 (&tp)[16],
                                               const f32x4 (&go)[3], float sg, float st,
                                               unsigned long long *accA, unsigned long long *accB, int W, int H)
{
    const float magic = __int_as_float(PkAcc::kMagic);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!((fast >> j) & 1)) continue;
        // keep the cell addresses and weights inside the caller's loops (hoisted, they spill)
        asm volatile("" : "+v"(g.ix[j]), "+v"(g.iy[j]), "+v"(g.a[j]), "+v"(g.b[j]));
        int ro[4], co[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ro[k] = (clampi(g.iy[j] - 1 + k, H - 1) - r.y0) * r.pitch;
            co[k] = pk_col(clampi(g.ix[j] - 1 + k, W - 1) - r.x0, r.pitch >> 2);
        }
        const float a = g.a[j], bt = g.b[j];
        const float wq[4] = {st * ((1 - a) * (1 - bt)), st * (a * (1 - bt)), st * ((1 - a) * bt), st * (a * bt)};
        const float g0 = sg * go[0][j], g1 = sg * go[1][j], g2 = sg * go[2][j];
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const float w = wq[(k >> 1) * 2 + (m >> 1)] * tp[k * 4 + m][j];
                const int n0 = __float_as_int(fmaf(g0, w, magic)) - PkAcc::kMagic;
                const int n1 = __float_as_int(fmaf(g1, w, magic)) - PkAcc::kMagic;
                const int n2 = __float_as_int(fmaf(g2, w, magic)) - PkAcc::kMagic;
                const long long A = ((long long)n0 << PkAcc::kShift) + (long long)(n2 >> PkAcc::kSplit);
                const long long B = ((long long)n1 << PkAcc::kShift) | (long long)(n2 & ((1 << PkAcc::kSplit) - 1));
                lds_add_u64(accA + ro[k] + co[m], (unsigned long long)A);
                lds_add_u64(accB + ro[k] + co[m], (unsigned long long)B);
            }
    }
}

// Unpacks every cell of the band and adds its three colours to gradinput1 (row-coalesced global atomics: the boxes of
// neighbouring tiles overlap).  The 256 lanes walk the box's cells in row-major order, so a wave's 64 cells are one
// run of a row (256 contiguous bytes per colour) and, in the planes, 4 x 8 consecutive slots: conflict-free.
// inv = 2^(e_g + e_t - 22) as a double (the float may not exist).
__device__ __forceinline__ void fi_bwd_flush_pk(const Region &r, const unsigned long long *accA,
                                                const unsigned long long *accB, double inv, float *gin1_b,
                                                int64_t s1c, int s1h)
{
    const unsigned tid = tid_now();
    const int w = max(r.w, 1), total = r.w * r.h;
    int row = tid / w, col = tid % w;                      // one run-time division per band
    const int drow = 256 / w, dcol = 256 % w;
    const uintptr_t b0 = pin_sgpr(gin1_b), b1 = pin_sgpr(gin1_b + s1c), b2 = pin_sgpr(gin1_b + 2 * s1c);
    constexpr int kBatch = 4;
#pragma unroll 1
    for (int base = 0; base < total; base += 256 * kBatch) {
        long long A[kBatch], B[kBatch];
        unsigned off[kBatch];
        bool on[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            on[u] = row < r.h;
            const int slot = (on[u] ? row : 0) * r.pitch + pk_col(col, r.pitch >> 2);
            A[u] = (long long)accA[slot];
            B[u] = (long long)accB[slot];
            off[u] = 4u * (unsigned)((r.y0 + row) * s1h + r.x0 + col);
            col += dcol;
            row += drow + (col >= w ? 1 : 0);
            col -= col >= w ? w : 0;
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
            if (!on[u] || (A[u] | B[u]) == 0) continue;
            const long long c0 = (A[u] + (1LL << (PkAcc::kShift - 1))) >> PkAcc::kShift;
            const long long sa = A[u] - (c0 << PkAcc::kShift);
            const long long c1 = B[u] >> PkAcc::kShift;
            const long long lb = B[u] & ((1LL << PkAcc::kShift) - 1);
            const long long c2 = (sa << PkAcc::kSplit) + lb;
            const float v0 = (float)((double)c0 * inv), v1 = (float)((double)c1 * inv), v2 = (float)((double)c2 * inv);
            if (v0 != 0.0f) (void)__builtin_amdgcn_global_atomic_fadd_f32(reinterpret_cast<MEMC_GLOBAL float *>(b0 + off[u]), v0);
            if (v1 != 0.0f) (void)__builtin_amdgcn_global_atomic_fadd_f32(reinterpret_cast<MEMC_GLOBAL float *>(b1 + off[u]), v1);
            if (v2 != 0.0f) (void)__builtin_amdgcn_global_atomic_fadd_f32(reinterpret_cast<MEMC_GLOBAL float *>(b2 + off[u]), v2);
        }
    }
}

// the image gradient of ONE site with global atomics (tiles with a non-finite gradoutput or tap)
__device__ __noinline__ void fi_bwd_site_image_atomics(int x, int y, int W, int H, float *gin1_b, int64_t s1c, int s1h,
                                                       const float *flow_p, int64_t s2c, const float *tap_p,
                                                       int64_t s3c, const float *gout_p)
{
    const FiSite s = fi_locate(x, y, W, H, flow_p[0], flow_p[s2c]);
    if (!s.valid) return;
    for (int c = 0; c < 3; c++) {
        const float gv = gout_p[c * s1c];
        float *q = gin1_b + c * s1c;
        for (int k = 0; k < 4; k++) {
            const int jj = clampi(s.iy - 1 + k, H - 1) * s1h;
            for (int m = 0; m < 4; m++) {
                const float wa = m < 2 ? (1 - s.a) : s.a, wb = k < 2 ? (1 - s.b) : s.b;
                atomic_add_f32(q + jj + clampi(s.ix - 1 + m, W - 1), gv * wa * wb * tap_p[(k * 4 + m) * s3c]);
            }
        }
    }
}

// largest value of a non-negative int over the wave (float bit patterns of |x| order like ints; NaN sorts last)
__device__ __forceinline__ int wave_max_i32(int v) { return -wave_min_i32(-v); }

// The two planes alias the staged image (48 KiB per workgroup, 3072 cells: 96 x 32, 80 x 38 or 64 x 48 by the band's
// width -- DYN; the fixed 96 x 32 of rounds 1-2 is kept as an arm).
//
// ORDER 0 / 1 (arms 20 / 22; 0 = fixed 96 x 32 geometry): image first, as rounds 1-2 had it -- stage, barrier, phase 1 (tap
//   and flow gradients from the staged image), barrier, zero the planes, barrier, adds, barrier, flush.
// ORDER 2: image gradient FIRST.  The adds need no image, so the planes are zeroed while the tile's input loads are
//   still in flight, the adds start as soon as the box is known, and the image rows -- requested before the adds --
//   arrive in registers while the LDS is busy with adds and flush; they are written to the LDS behind the flush and
//   phase 1 ends the tile with its stores.  Against ORDER 1 the tile's serial chain loses the zeroing pass with its two
//   barriers and the staging round trip:  load -> box -> adds -> flush -> (image is already here) -> phase 1.
//   (The staged rows are touched once before the flush: vmcnt is in order, and a wait behind the flush's conditional
//   atomics could only be vmcnt(0).)
// Measured and dropped (session r03_s1): image and planes side by side (78 KiB, 2496 cells each; adds straight
// behind phase 1 without a barrier) -- 1743 us against 1605 for the aliased planes at 720p (i.i.d. flow: 3462 vs
// 2688): the smaller budget sweeps more tiles in two bands, and the adds overlapped nothing (a wave's adds + barrier
// took as long as barrier + zero + barrier + adds).
struct PkGeom {
    static constexpr int kCap = 3072;                                  // pixel quads staged = slots per plane
    static constexpr int kImageBytes = kCap * 16;
    static constexpr int kLds = kImageBytes + 128;
};

template <int MINW, int ORDER, bool TR>
__global__ __launch_bounds__(256, MINW) void fi_bwd_c3_pk(
    int W, int H, int tiles_x, int tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    const float *__restrict__ gout, float *__restrict__ gin1, float *__restrict__ gin2,
    float *__restrict__ gin3)
{
    constexpr int LX = 16;
    constexpr bool DYN = ORDER != 0, P2FIRST = ORDER == 2;
    using PG = PkGeom;
    using G = TileGeom<LX, PG::kCap>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    unsigned long long *const accA = reinterpret_cast<unsigned long long *>(smem);   // the planes alias the image
    unsigned long long *const accB = accA + PG::kCap;
    int *bb = reinterpret_cast<int *>(smem + PG::kImageBytes);           // 16 ints: boxes; 8 ints: maxima
    int *mx = bb + 16;

    trace_mark<TR>(0);
    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, batch);
    const int b = tc.b;
    const unsigned tid = tid_now();
    const int x = tc.tx * G::kTW + 4 * (int)(tid % LX), y = tc.ty * G::kTH + (int)(tid / LX);
    const bool inb = x < W && y < H;
    const int xs = min(x, W - 4), ys = min(y, H - 1);
    const float *flow_b = flow + b * s2b, *filt_b = filt + b * s3b, *gout_b = gout + b * s1b;
    float *gin2_b = gin2 + b * s2b, *gin3_b = gin3 + b * s3b;
    const unsigned o1 = 4u * (unsigned)(ys * s1h + xs), o2 = 4u * (unsigned)(ys * s2h + xs),
                   o3 = 4u * (unsigned)(ys * s3h + xs);
    f32x4 go[3], tp[16];
    const f32x4 fx4 = ld_stream4_u(flow_b, o2), fy4 = ld_stream4_u(flow_b + s2c, o2);
#pragma unroll
    for (int c = 0; c < 3; c++) go[c] = ld_stream4_u(gout_b + c * s1c, o1);
#pragma unroll
    for (int k = 0; k < 16; k++) tp[k] = ld_stream4_u(filt_b + k * s3c, o3);
    auto zero_planes = [&](int cells) {        // the first `cells` slots of both planes (whole 16-byte units)
        f32x4 *pa = reinterpret_cast<f32x4 *>(accA), *pb = reinterpret_cast<f32x4 *>(accB);
        for (int i = (int)tid_now(); i < (cells >> 1); i += 256) {
            pa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if (P2FIRST) zero_planes(PG::kCap);        // while the loads are in flight
    if (TR) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    trace_mark<TR>(1);                                         // inputs have arrived

    FiSite4 g;
    g.valid = 0;
    int cmin = INT_MAX, cmax = -1, rmin = INT_MAX, rmax = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const FiSite s = fi_locate(x + j, y, W, H, fx4[j], fy4[j]);
        g.ix[j] = s.ix; g.iy[j] = s.iy; g.a[j] = s.a; g.b[j] = s.b;
        if (inb && s.valid) {
            g.valid |= 1u << j;
            cmin = min(cmin, max(s.ix - 1, 0));  cmax = max(cmax, min(s.ix + 2, W - 1));
            rmin = min(rmin, max(s.iy - 1, 0));  rmax = max(rmax, min(s.iy + 2, H - 1));
        }
    }
    {                                          // the tile's largest |gradoutput| and |tap| (bit patterns), per wave;
        int mg = 0, mt = 0;                    // handed over by the barrier inside tile_bbox
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int c = 0; c < 3; c++) mg = max(mg, __float_as_int(go[c][j]) & 0x7FFFFFFF);
#pragma unroll
            for (int k = 0; k < 16; k++) mt = max(mt, __float_as_int(tp[k][j]) & 0x7FFFFFFF);
        }
        mg = wave_max_i32(inb ? mg : 0);
        mt = wave_max_i32(inb ? mt : 0);
        if ((tid & (kWave - 1)) == 0) {
            mx[(tid / kWave) * 2] = mg;
            mx[(tid / kWave) * 2 + 1] = mt;
        }
    }
    const BBox box = tile_bbox<LX>(cmin, cmax, rmin, rmax, bb);
    const Bands bands = make_bands<LX, DYN, PG::kCap>(box);
    int mg = max(max(mx[0], mx[2]), max(mx[4], mx[6])), mt = max(max(mx[1], mx[3]), max(mx[5], mx[7]));
    mg = __builtin_amdgcn_readfirstlane(mg);
    mt = __builtin_amdgcn_readfirstlane(mt);
    // 0: nothing to add (every contribution of this tile is zero); 2: Inf / NaN among the inputs -- per-site global
    // atomics; 1: the packed planes
    const int mode = (mg == 0 || mt == 0) ? 0 : ((mg >= 0x7F800000 || mt >= 0x7F800000) ? 2 : 1);
    // block exponent: 2^(eg + et) > (largest |gradoutput|) x (largest |tap|), off by less than a factor two -- the
    // mantissas' product tells whether the sum of the two exponents is one too many
    const int eg = pk_exponent(mode == 1 ? mg : 0x3F800000);
    int et = pk_exponent(mode == 1 ? mt : 0x3F800000);
    {
        int e0, e1;
        const float mm = frexpf(__int_as_float(mode == 1 ? mg : 0x3F800000), &e0) *
                         frexpf(__int_as_float(mode == 1 ? mt : 0x3F800000), &e1);
        if (mm < 0.4999f && et > -100) et -= 1;
    }
    const float sg = ldexpf(1.0f, 11 - eg), st = ldexpf(1.0f, 11 - et);
    const double inv = ldexp(1.0, eg + et - 22);
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    unsigned done = 0;
    trace_mark<TR>(2);                                         // bounding box known
    fi_bwd_zero_invalid(inb, g.valid, gin2_b, s2c, o2, gin3_b, s3c, o3);
    auto image_atomics = [&](unsigned todo) {  // mode 2
        while (todo) {
            const int j = __ffs(todo) - 1;
            todo &= todo - 1;
            fi_bwd_site_image_atomics(x + j, y, W, H, gin1_b, s1c, s1h, flow_b + o2 / 4 + j, s2c,
                                      filt_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
        }
    };
    auto phase1 = [&](const Region &r, unsigned fast) {
        fi_bwd_phase1<0>(r, fast, g, tp, go, tile, W, H, gin2_b, s2c, o2, gin3_b, s3c, o3);
        if (fast != 0xFu) {                    // mixed quads (rare): their tap gradients, site by site
            unsigned todo = fast;
            while (todo) {
                const int j = __ffs(todo) - 1;
                todo &= todo - 1;
                fi_bwd_site_taps(x + j, y, W, H, in_b, s1c, s1h, flow_b + o2 / 4 + j, gin2_b + o2 / 4 + j, s2c,
                                 filt_b + o3 / 4 + j, gin3_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
            }
        }
    };
#pragma unroll 1
    for (int bi = 0; bi < bands.n; bi++) {
    const Region r = band_region(box, bands, bi);
    const unsigned fast = inb ? fi_covered(r, g, W, H) & ~done : 0u;
    // later bands run only if some site still needs them; the vote is also the barrier that frees the LDS
    if (bi > 0 && !__syncthreads_or(fast != 0)) continue;
    done |= fast;
    if (P2FIRST) {
        const StageSlot sl = stage_slots(r);
        StageRegs<3> sr;
        tile_stage_load<3>(r, sl, in_b, s1c, s1h, sr);         // in flight during adds and flush
        if (mode == 1) {
            if (bi > 0) {                      // (band 0: zeroed at the top, ordered by the barrier of tile_bbox)
                zero_planes(r.h * r.pitch);
                __syncthreads();
            }
            fi_bwd_adds_pk(r, fast, g, tp, go, sg, st, accA, accB, W, H);
            __syncthreads();
            if (bi == 0) trace_mark<TR>(3);                    // accumulated
#pragma unroll
            for (int it = 0; it < kStageIts; it++)             // the staged rows have landed long ago: take the wait
#pragma unroll                                                 // here, not behind the flush's atomics
                for (int c = 0; c < 3; c++)
                    asm volatile("" : "+v"(sr.v[it][c][0]), "+v"(sr.v[it][c][1]), "+v"(sr.v[it][c][2]), "+v"(sr.v[it][c][3]));
            fi_bwd_flush_pk(r, accA, accB, inv, gin1_b, s1c, s1h);
            __syncthreads();                   // the planes have been read: the LDS becomes the image
            if (bi == 0) trace_mark<TR>(4);                    // flushed
        } else if (mode == 2) {
            image_atomics(fast);
        }
        tile_stage_store<3>(r, sl, sr, tile);
        __syncthreads();
        if (bi == 0) trace_mark<TR>(5);                        // image staged
        phase1(r, fast);
        if (bi == 0) trace_mark<TR>(6);                        // phase 1 done (this wave)
        continue;
    }
    tile_stage<LX, 3>(r, in_b, s1c, s1h, tile);
    __syncthreads();
    if (bi == 0) trace_mark<TR>(3);                            // image staged
    phase1(r, fast);
    if (bi == 0) trace_mark<TR>(4);                            // phase 1 done (this wave)
    if (mode == 2) image_atomics(fast);
    if (mode != 1) continue;                   // (workgroup-uniform)
    __syncthreads();                           // everybody is done reading the image: the LDS becomes the planes
    zero_planes(r.h * r.pitch);
    __syncthreads();
    if (bi == 0) trace_mark<TR>(5);                            // planes zeroed
    fi_bwd_adds_pk(r, fast, g, tp, go, sg, st, accA, accB, W, H);
    __syncthreads();
    if (bi == 0) trace_mark<TR>(6);                            // accumulated
    fi_bwd_flush_pk(r, accA, accB, inv, gin1_b, s1c, s1h);
    if (bi == 0) trace_mark<TR>(7);                            // flushed
    }   // bands
    trace_mark<TR>(12);
    unsigned slow = inb ? g.valid & ~done : 0u;            // not coverable within kMaxBands bands
    while (slow) {                            // rare: redone from global memory with global atomics
        const int j = __ffs(slow) - 1;
        slow &= slow - 1;
        fi_bwd_site_scalar(x + j, y, W, H, 3, 4, in_b, gin1_b, s1c, s1h, flow_b + o2 / 4 + j, gin2_b + o2 / 4 + j,
                           s2c, filt_b + o3 / 4 + j, gin3_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
    }
}

// 1: taken, 0: not this kernel's case (the caller falls back to the direct kernel), -1: launch error.  `variant`
// selects a measurement arm (measurement build only; the product passes -1).
int fi_bwd_c3_launch(hipStream_t stream, int w, int h, int batch,
                     int s1b, int s1c, int s1h, int s2b, int s2c, int s2h, int s3b, int s3c, int s3h,
                     const float *input1, const float *input2, const float *input3, const float *gradoutput,
                     float *gradinput1, float *gradinput2, float *gradinput3, int variant)
{
    if (!plane_fits_u32(w, h, {s1h, s2h, s3h}) ||
        !vec4_ok(w, {s1b, s1c, s1h, s2b, s2c, s2h, s3b, s3c, s3h},
                 {input1, input2, input3, gradoutput, gradinput1, gradinput2, gradinput3}))
        return 0;
    using G = TileGeom<16>;
    static_assert(AccT::kPlane * 8 <= G::kCapPx * 16 && G::kPitch <= AccT::kMaxW && G::kRows <= AccT::kRows,
                  "the accumulator plane aliases the staged image");
    const int ntx = (w + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;
    const unsigned ntiles = (unsigned)ntx * nty * batch;
#define MEMC_FI_BWD_ARGS                                                                                           \
    w, h, ntx, nty, batch, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b,         \
        (int64_t)s3c, s3h, input1, input2, input3, gradoutput, gradinput1, gradinput2, gradinput3
#define MEMC_FI_BWD(ABL)                                                                                           \
    hipLaunchKernelGGL(fi_bwd_tiled_c3<ABL>, dim3(ntiles), dim3(256), tile_lds_bytes<16>(), stream, MEMC_FI_BWD_ARGS)
#define MEMC_FI_BWD_P(PF)                                                                                          \
    do {                                                                                                           \
        const int lds = 2 * AccT::kPlane * 8 + 64;                                                                 \
        static const bool once = (allow_big_lds(fi_bwd_tiled_c3_persistent<PF>, lds), true);                       \
        (void)once;                                                                                                \
        const unsigned grid = ntiles < persistent_grid(2) ? ntiles : persistent_grid(2);                           \
        hipLaunchKernelGGL(fi_bwd_tiled_c3_persistent<PF>, dim3(grid), dim3(256), lds, stream, MEMC_FI_BWD_ARGS);  \
    } while (0)
#define MEMC_FI_BWD_PK(MINW, AL, TR)                                                                               \
    do {                                                                                                           \
        hipLaunchKernelGGL((fi_bwd_c3_pk<MINW, AL, TR>), dim3(ntiles), dim3(256), PkGeom::kLds, stream,            \
                           MEMC_FI_BWD_ARGS);                                                                      \
    } while (0)
#ifdef MEMC_MEASURE
    switch (variant) {
    case 1: MEMC_FI_BWD(1); break;
    case 2: MEMC_FI_BWD(2); break;
    case 3: MEMC_FI_BWD(3); break;
    case 4: MEMC_FI_BWD(4); break;
    case 5: MEMC_FI_BWD(5); break;
    case 9: MEMC_FI_BWD(9); break;
    case 16:                                               // three workgroups per CU: 168 VGPRs, spills
        hipLaunchKernelGGL((fi_bwd_tiled_c3<0, 3>), dim3(ntiles), dim3(256), tile_lds_bytes<16>(), stream,
                           MEMC_FI_BWD_ARGS);
        break;
    case 10: MEMC_FI_BWD_P(true); break;
    case 11: MEMC_FI_BWD_P(false); break;
    case 20: MEMC_FI_BWD_PK(2, 0, false); break;           // packed planes, image first, fixed 96 x 32 geometry
    case 22: MEMC_FI_BWD_PK(2, 1, false); break;           // packed planes, image first, dynamic pitch
    case 23: MEMC_FI_BWD_PK(2, 2, false); break;           // packed planes, image gradient first
    case 27: MEMC_FI_BWD_PK(2, 1, true); break;            // + timestamps
    case 28: MEMC_FI_BWD_PK(2, 2, true); break;
    default: MEMC_FI_BWD(0);
    }
#else
    (void)variant;
    MEMC_FI_BWD(0);
#endif
#undef MEMC_FI_BWD
#undef MEMC_FI_BWD_P
#undef MEMC_FI_BWD_PK
#undef MEMC_FI_BWD_ARGS
    return launch_status() == 0 ? 1 : -1;
}

}  // namespace memc

#ifdef MEMC_MEASURE
// device buffer of gridDim.x * 16 uint64 for the timestamp arm (fi_bwd variant 9); tools/trace_kernel.py
extern "C" int memc_debug_set_trace_buffer(void *p)
{
    unsigned long long *q = (unsigned long long *)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(memc::g_trace_buf), &q, sizeof(q)) == hipSuccess ? 0 : -1;
}
#endif
