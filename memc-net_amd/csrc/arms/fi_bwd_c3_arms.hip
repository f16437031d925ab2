// arms/fi_bwd_c3_arms.hip -- MEASUREMENT BUILD ONLY (libmemc_hip_measure.so): the RGB FilterInterpolation backward of
// rounds 1-2 -- one fp64 LDS plane that the three colours take in turn -- with its ablation arms, and its persistent
// variants.  Kept as the A/B baseline of fi_bwd_c3.hip (tools/bench_ops.py --bwd-variants, tools/trace_kernel.py);
// several arms return WRONG results by construction (they time one phase).  Never compiled into libmemc_hip.so.
#ifndef MEMC_MEASURE
#error "measurement arms: build with -DMEMC_MEASURE (make measure)"
#endif
#include "../memc_common.hpp"
#include "../memc_internal.h"
#include "../memc_tile.hpp"
#include "../memc_fi.hpp"

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
__device__ unsigned long long *g_trace_buf_arms = nullptr;
constexpr int kTraceSlots = 16;
template <bool ON>
__device__ __forceinline__ void trace_mark(int slot)
{
    if (ON && threadIdx.x == 0) g_trace_buf_arms[(size_t)blockIdx.x * kTraceSlots + slot] = __builtin_readcyclecounter();
}

// ABL != 0 are MEASUREMENT arms (tools/bench_ops.py --bwd-variants; their results are wrong by construction):
//   1 no fp64 LDS adds (zero + flush kept; zero cells are not flushed)   2 no phase 2 at all
//   3 phase 1 without its LDS reads                                      5 flush with plain stores
//   4 accumulate but never flush (plane re-zeroed instead)
//   9 production + phase timestamps
// gradinput3 and gradinput2 of ONE site straight from global memory (mixed quads of the tiled backward: some of a
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

// 1: launched, 0: no such arm, -1: launch error.  Geometry has been checked by fi_bwd_c3_launch.
int fi_bwd_c3_arm_launch(int variant, hipStream_t stream, int w, int h, int ntx, int nty, int batch,
                         int s1b, int s1c, int s1h, int s2b, int s2c, int s2h, int s3b, int s3c, int s3h,
                         const float *input1, const float *input2, const float *input3, const float *gradoutput,
                         float *gradinput1, float *gradinput2, float *gradinput3)
{
    using G = TileGeom<16>;
    static_assert(AccT::kPlane * 8 <= G::kCapPx * 16 && G::kPitch <= AccT::kMaxW && G::kRows <= AccT::kRows,
                  "the accumulator plane aliases the staged image");
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
    switch (variant) {
    case 0: MEMC_FI_BWD(0); break;                         // the production kernel of rounds 1-2
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
    default: return 0;
    }
#undef MEMC_FI_BWD
#undef MEMC_FI_BWD_P
#undef MEMC_FI_BWD_ARGS
    return launch_status() == 0 ? 1 : -1;
}

int fi_bwd_c3_arms_set_trace_buffer(unsigned long long *q)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(memc::g_trace_buf_arms), &q, sizeof(q)) == hipSuccess ? 0 : -1;
}

}  // namespace memc
