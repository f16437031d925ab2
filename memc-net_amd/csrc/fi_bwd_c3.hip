// fi_bwd_c3.hip -- FilterInterpolation backward, RGB (C == 3), fs == 4: the LDS-tiled kernel.
//
// Replaces my_package/src/my_lib_kernel.cu:1220-1518 (kernel) / :1571-1627 (launcher) of the reference for the only
// channel count its networks back-propagate through (networks/MEMC_Net_star.py:266-277; the context warps are
// detached, :285).  Semantics: SURVEY.md appendix A.2.  Same tile / box machinery as the forward kernel:
//   * streams (flow, 16 tap planes, 3 gradoutput planes) as dwordx4;
//   * the image gradient -- 48 scattered adds per site -- is accumulated in LDS (packed fixed point, below) and
//     flushed once per cell with row-coalesced global atomics (neighbouring tiles' boxes overlap);
//   * the image box is staged into LDS pixel quads for the tap and flow gradients: gradinput3 (each site owns its
//     taps) is stored once per site as dwordx4, gradinput2 is assigned;
//   * sites whose window no band covers are redone by fi_bwd_site_scalar with global atomics.
// The packed planes are described in memc_pk.hpp.  The kernel of rounds 1-2 (one fp64 plane per colour) lives on as a
// measurement arm: arms/fi_bwd_c3_arms.hip.
#include "memc_common.hpp"
#include "memc_internal.h"
#include "memc_tile.hpp"
#include "memc_fi.hpp"
#include "memc_pk.hpp"

namespace memc {

// Per-workgroup phase timestamps (shader clock) for tools/trace_kernel.py: the TR = true instantiation exists in the
// measurement build only.
#ifdef MEMC_MEASURE
__device__ unsigned long long *g_trace_buf = nullptr;
#endif
template <bool ON>
__device__ __forceinline__ void trace_mark(int slot)
{
#ifdef MEMC_MEASURE
    if (ON && threadIdx.x == 0) g_trace_buf[(size_t)blockIdx.x * 16 + slot] = __builtin_readcyclecounter();
#else
    static_assert(!ON, "timestamps: measurement build only");
    (void)slot;
#endif
}

// Phase 1 of one band: tap and flow gradients of the sites in `fast` from the staged image.
// With s = sum_c g_c * in_c(tap cell) (3 FMAs per tap), and q the tap's quadrant:
//     gradinput3[tap] = wq * s,   gradinput2.x = sum_taps cx[q] * s * tap,   gradinput2.y likewise,
// where wq = {(1-a)(1-b), a(1-b), (1-a)b, ab}, cx = {-(1-b), (1-b), -b, b}, cy = {-(1-a), -a, (1-a), a}.
// (The reference sums per channel first -- same value up to fp32 re-association, ~1e-7 relative.)
// Tap rows are the outer loop so that only one row of tap gradients (4 float4) is live at a time.
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
    // the kernel lives or dies by staying clear of spills (252 of 256 registers with the staged rows parked beside it).
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
                const f32x4 pix = tile[ro[j] + co];
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

// The 32 ds_add_u64 of the sites in `fast`.  sg = 2^(11 - e_g), st = 2^(11 - e_t): |g * sg| < 2^11, |w * st| <= 2^11.
__device__ __forceinline__ void fi_bwd_adds_pk(const Region &r, unsigned fast, FiSite4 &g, const f32x4 (&tp)[16],
                                               const f32x4 (&go)[3], float sg, float st,
                                               unsigned long long *accA, unsigned long long *accB, int W, int H)
{
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
                pk_add3(accA, accB, ro[k] + co[m], g0, g1, g2, w);
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

// One 64 x 16 tile of sites per workgroup; 48 KiB of LDS: the two packed planes, then -- the same bytes -- the staged
// image (3072 cells: 96 x 32, 80 x 38 or 64 x 48 by the band's width).  The image gradient comes FIRST: its adds need
// no image, so
//   * the planes are zeroed while the tile's 21 input float4 per lane are still in flight;
//   * the adds start as soon as the box is known; the image rows, requested just before, arrive in registers while
//     the LDS is busy with adds and flush (they are touched once before the flush: vmcnt is in order, and a wait
//     placed behind the flush's conditional atomics could only be vmcnt(0));
//   * the rows go to the LDS behind the flush, and phase 1 (tap and flow gradients from the staged image) ends the
//     tile with its stores.
// Serial chain of a tile: load -> box -> adds -> flush -> (image is already here) -> phase 1: four barriers.
// Measured, 720p batch 32, smooth / i.i.d. flow / 448 x 256 batch 8 (profiles/r03_fi_bwd_c3_arms.txt; one box, one process):
//   rounds 1-2: fp64 plane per colour, three rounds of adds / flush      1777 / 2977 /  98.1 us
//   packed planes, image first, fixed 96 x 32 geometry                   1545 / 2594 /  85.6
//   packed planes, image first, pitch by the band's width                1364 / 1985 /  71.5
//   packed planes, image gradient first (this kernel)                    1312 / 1929 /  71.4
//   planes beside the image (78 KiB; adds straight behind phase 1)       1743 / 3462 /  90.1   (two-band sweeps; no overlap won)
// NT lanes take a tile of 64 x NT / 16 sites.  256 (64 x 16, 48 KiB, two workgroups per CU) is the product.
// 128 (64 x 8, 24 KiB, four workgroups of two waves per CU) was built in round 4 for SMALL grids -- BASELINE config 2
// (8 x 448 x 256) is 896 tiles of 64 x 16 on 512 workgroup slots, 1.75 rounds of one tile's serial chain; as 1792 tiles of
// 64 x 8 on 1024 slots the chain per tile should have been shorter and the tail round half as long -- and LOST in one
// process (profiles/r04_fi_bwd_tile_height_ab.txt): config 2 72.9 -> 102.8 us, 720p 1436 -> 2134 us, i.i.d. flow 2.7x
// slower.  A box of 8 + 3 + motion rows holds 2.3x its tile's cells (64 x 16: 1.7x): the flush's atomics, the staged rows
// and the barriers per site all grow, and nothing in the chain got shorter.  Measurement arm 61 only.
template <int NT>
struct PkGeomT {
    static constexpr int kCap = 12 * NT;                               // pixel quads staged = slots per plane: 3 float4 per lane
    static constexpr int kImageBytes = kCap * 16;
    static constexpr int kLds = kImageBytes + 128;
};
using PkGeom = PkGeomT<256>;

// PART: 0 the whole backward; 1 the image gradient alone (planes, adds, flush); 2 the tap and flow gradients alone (staged
// image, phase 1).
//   * PART 2 is what a caller gets who passes gradinput1 == NULL: it does not want the image gradient (the reference's
//     networks never do: the frames they warp are data, MEMC_Net_star.py:266-277).  720p batch 32: 1037 us against 1432 us
//     for the whole backward on the same box (+ the 70 us zero fill of gradinput1 that the caller no longer needs);
//     BASELINE config 2 (8 x 448 x 256): 46.8 us against 72.9 (profiles/r04_fi_bwd_halves_ab.txt).
//   * PART 1 + PART 2 as two launches were round 4's second attempt at SMALL grids (config 2 is 896 tiles on 512 workgroup
//     slots: 1.75 rounds of a four-barrier chain; two shorter chains, and at 2/3 of the registers three workgroups per CU,
//     were to beat that).  They need 189 / 207 VGPRs: at three per CU (168) both spill inside their hot loops; at two per
//     CU the split reads the inputs twice and LOSES -- config 2 72.9 -> 83.1 us, 720p 1432 -> 1792 us.  Measurement arm 62.
#ifdef MEMC_PART_THREE
constexpr bool kPartThree = true;              // (experiment: the halves at three workgroups per CU -- 168 VGPRs, and they SPILL:
#else                                          //  96 / 160 B per lane, reloaded inside the add loop and phase 1)
constexpr bool kPartThree = false;
#endif
// RAG: a ragged width (W % 4 != 0, round 5) -- the whole quads here (sites x < W & ~3; the image's true width in every clamp,
// validity test and box, the box's last quad staged ragged-safely: memc_tile.hpp), the one to three columns behind them in
// fi_bwd_direct_fs4 (launcher); both ADD into gradinput1.  One workgroup per CU: the kernel has no registers left for it.
template <bool TR, int NT = 256, int PART = 0, bool RAG = false>
__global__ __launch_bounds__(NT, RAG ? 1 : (PART == 0 || !kPartThree ? 2 : 3)) void fi_bwd_c3_pk(
    int W, int H, int tiles_x, int tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    const float *__restrict__ gout, float *__restrict__ gin1, float *__restrict__ gin2,
    float *__restrict__ gin3)
{
    constexpr int LX = 16;
    using PG = PkGeomT<NT>;
    using G = TileGeom<LX, PG::kCap, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    unsigned long long *const accA = reinterpret_cast<unsigned long long *>(smem);   // the planes alias the image
    unsigned long long *const accB = accA + PG::kCap;
    int *bb = reinterpret_cast<int *>(smem + PG::kImageBytes);           // 16 ints: boxes; 16 ints: the waves' bound statistics
    int *mx = bb + 16;

    trace_mark<TR>(0);
    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, batch);
    const int b = tc.b;
    const unsigned tid = tid_now();
    const int x = tc.tx * G::kTW + 4 * (int)(tid % LX), y = tc.ty * G::kTH + (int)(tid / LX);
    const int Ws = RAG ? W & ~3 : W;
    const bool inb = x < Ws && y < H;
    const int xs = min(x, Ws - 4), ys = min(y, H - 1);
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
        for (int i = (int)tid_now(); i < (cells >> 1); i += NT) {
            pa[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if (PART != 2) zero_planes(PG::kCap);      // while the loads are in flight
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
    // per-site bounds of the packed planes (memc_pk.hpp): s = (the site's largest |gradoutput|) x (its largest |tap|),
    // published per wave and handed over by the barrier inside tile_bbox
    int sbits[4] = {0, 0, 0, 0}, gbits[4] = {0, 0, 0, 0}, tmax = 0;
    if (PART != 2) {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int mg = 0, mt = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) mg = max(mg, __float_as_int(go[c][j]) & 0x7FFFFFFF);
#pragma unroll
            for (int k = 0; k < 16; k++) mt = max(mt, __float_as_int(tp[k][j]) & 0x7FFFFFFF);
            const float sv = __int_as_float(mg) * __int_as_float(mt);
            gbits[j] = mg;
            tmax = max(tmax, ((g.valid >> j) & 1u) ? mt : 0);
            // (Inf x 0 = NaN: not finite, per-site atomics put it where the reference does; a zero bound adds nothing)
            sbits[j] = (mg >= 0x7F800000 || mt >= 0x7F800000) ? 0x7FC00000 : __float_as_int(sv);
        }
        pk_tile_publish(mx, tid, sbits, gbits, g.valid, tmax);
    }
    const BBox box = tile_bbox<LX, NT>(cmin, cmax, rmin, rmax, bb);
    const Bands bands = make_bands<LX, true, PG::kCap>(box);
    PkTile ps;
    ps.sa = ps.sb = 1.0f;  ps.inv = 1.0;  ps.limit = -1.0f;  ps.any = 0;
    if (PART != 2) ps = pk_tile_resolve<NT / kWave>(mx);
    // packed: the site's image gradient goes through the planes; outl: per-site global atomics (a bound beyond the tile's
    // block exponent, or an Inf / NaN among the site's inputs -- which then land exactly where the reference puts them)
    const unsigned packed = PART != 2 ? pk_packed_sites(ps, sbits, g.valid) : 0u;
    const unsigned outl = PART != 2 ? pk_outlier_sites(ps, sbits, g.valid) : 0u;
    const int mode = ps.any;                   // 0: no packed site has anything to add (workgroup-uniform)
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    unsigned done = 0;
    trace_mark<TR>(2);                                         // bounding box known
    if (PART != 1) fi_bwd_zero_invalid(inb, g.valid, gin2_b, s2c, o2, gin3_b, s3c, o3);
    auto image_atomics = [&](unsigned todo) {  // outlier sites
        while (todo) {
            const int j = __ffs(todo) - 1;
            todo &= todo - 1;
            fi_bwd_site_image_atomics(x + j, y, W, H, gin1_b, s1c, s1h, flow_b + o2 / 4 + j, s2c,
                                      filt_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
        }
    };
    auto phase1 = [&](const Region &r, unsigned fast) {
        fi_bwd_phase1(r, fast, g, tp, go, tile, W, H, gin2_b, s2c, o2, gin3_b, s3c, o3);
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
    const Region r = band_region(box, bands, bi, RAG ? W : 0);
    const unsigned fast = inb ? fi_covered(r, g, W, H) & ~done : 0u;
    // later bands run only if some site still needs them; the vote is also the barrier that frees the LDS
    if (bi > 0 && !__syncthreads_or(fast != 0)) continue;
    done |= fast;
    const StageSlot sl = stage_slots<NT>(r);
    StageRegs<3> sr;
    if (PART != 1) tile_stage_load<3, RAG>(r, sl, in_b, s1c, s1h, sr);   // in flight during adds and flush
    if (PART != 2 && mode == 1) {
        if (bi > 0) {                      // (band 0: zeroed at the top, ordered by the barrier of tile_bbox)
            zero_planes(r.h * r.pitch);
            __syncthreads();
        }
        fi_bwd_adds_pk(r, fast & packed, g, tp, go, ps.sa, ps.sb, accA, accB, W, H);
        __syncthreads();
        if (bi == 0) trace_mark<TR>(3);                    // accumulated
        if (PART != 1) {
#pragma unroll
            for (int it = 0; it < kStageIts; it++)         // the staged rows have landed long ago: take the wait
#pragma unroll                                             // here, not behind the flush's atomics
                for (int c = 0; c < 3; c++)
                    asm volatile("" : "+v"(sr.v[it][c][0]), "+v"(sr.v[it][c][1]), "+v"(sr.v[it][c][2]), "+v"(sr.v[it][c][3]));
        }
        pk_flush<NT>(r, accA, accB, ps.inv, gin1_b, s1c, s1h);
        if (PART != 1) __syncthreads();    // the planes have been read: the LDS becomes the image
        if (bi == 0) trace_mark<TR>(4);                    // flushed
    }
    if (PART != 2 && (fast & outl)) image_atomics(fast & outl);
    if (PART != 1) {
        tile_stage_store<3, RAG>(r, sl, sr, tile);
        __syncthreads();
        if (bi == 0) trace_mark<TR>(5);                    // image staged
        phase1(r, fast);
        if (bi == 0) trace_mark<TR>(6);                    // phase 1 done (this wave)
    }
    }   // bands
    trace_mark<TR>(12);
    unsigned slow = inb ? g.valid & ~done : 0u;            // not coverable within kMaxBands bands
    while (slow) {                            // rare: redone from global memory with global atomics
        const int j = __ffs(slow) - 1;
        slow &= slow - 1;
        if (PART == 0)
            fi_bwd_site_scalar(x + j, y, W, H, 3, 4, in_b, gin1_b, s1c, s1h, flow_b + o2 / 4 + j, gin2_b + o2 / 4 + j,
                               s2c, filt_b + o3 / 4 + j, gin3_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
        else if (PART == 1)
            fi_bwd_site_image_atomics(x + j, y, W, H, gin1_b, s1c, s1h, flow_b + o2 / 4 + j, s2c, filt_b + o3 / 4 + j, s3c,
                                      gout_b + o1 / 4 + j);
        else
            fi_bwd_site_taps(x + j, y, W, H, in_b, s1c, s1h, flow_b + o2 / 4 + j, gin2_b + o2 / 4 + j, s2c,
                             filt_b + o3 / 4 + j, gin3_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
    }
}
// 1: taken; 2: taken for the whole quads of a ragged width (w % 4 != 0): the caller runs the direct kernel on the columns
// from w & ~3 on; 0: not taken (the caller takes the direct kernel for everything); -1: launch error.  `variant` >= 0
// selects a measurement arm (measurement build only; the product passes -1).
int fi_bwd_c3_launch(hipStream_t stream, int w, int h, int batch,
                     int s1b, int s1c, int s1h, int s2b, int s2c, int s2h, int s3b, int s3c, int s3h,
                     const float *input1, const float *input2, const float *input3, const float *gradoutput,
                     float *gradinput1, float *gradinput2, float *gradinput3, int variant)
{
    const int ws = w & ~3;
    if (!plane_fits_u32(w, h, {s1h, s2h, s3h}) || ws < 4) return 0;
    if (ws < w && (gradinput1 == nullptr || variant >= 0)) return 0;   // (the tail's direct kernel needs the buffer; arms: whole widths)
    using G = TileGeom<16>;
    const int ntx = (ws + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;
    const unsigned ntiles = (unsigned)ntx * nty * batch;
#define MEMC_FI_BWD_PK(TR, NT_, NTY, NTILES) MEMC_FI_BWD_PK_PART(TR, NT_, NTY, NTILES, 0)
#define MEMC_FI_BWD_PK_PART(TR, NT_, NTY, NTILES, PART_)                                                           \
    hipLaunchKernelGGL((fi_bwd_c3_pk<TR, NT_, PART_>), dim3(NTILES), dim3(NT_), PkGeomT<NT_>::kLds, stream, w, h, ntx, NTY, batch, \
                       (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b,             \
                       (int64_t)s3c, s3h, input1, input2, input3, gradoutput, gradinput1, gradinput2, gradinput3)
#ifdef MEMC_MEASURE
    bool split = false;                                    // arm 62: the two halves as two launches (LOST, see the kernel)
    if (variant == 28) {                                   // + timestamps
        MEMC_FI_BWD_PK(true, 256, nty, ntiles);
        return launch_status() == 0 ? 1 : -1;
    }
    if (variant == 61) {                                   // A/B arm: 64 x 8 tiles on 128 lanes (LOST, see PkGeomT)
        const int nty8 = (h + 7) / 8;
        MEMC_FI_BWD_PK(false, 128, nty8, (unsigned)ntx * nty8 * batch);
        return launch_status() == 0 ? 1 : -1;
    }
    if (variant >= 0 && variant != 60) {                   // arms/fi_bwd_c3_arms.hip (60: this kernel, named)
        const int r = fi_bwd_c3_arm_launch(variant, stream, w, h, ntx, nty, batch, s1b, s1c, s1h, s2b, s2c, s2h, s3b, s3c,
                                           s3h, input1, input2, input3, gradoutput, gradinput1, gradinput2, gradinput3);
        if (r != 0) return r;
    }
    if (variant == 62) split = true;                       // (60: the product kernel, named)
#else
    (void)variant;
#endif
    if (gradinput1 == nullptr) {                           // the caller does not want the image gradient
        MEMC_FI_BWD_PK_PART(false, 256, nty, ntiles, 2);
    }
#ifdef MEMC_MEASURE
    else if (split) {
        MEMC_FI_BWD_PK_PART(false, 256, nty, ntiles, 1);
        MEMC_FI_BWD_PK_PART(false, 256, nty, ntiles, 2);
    }
#endif
    else if (ws < w) {                                     // a ragged width: the RAG instantiation (+ the caller's tail launch)
        hipLaunchKernelGGL((fi_bwd_c3_pk<false, 256, 0, true>), dim3(ntiles), dim3(256), PkGeomT<256>::kLds, stream, w, h, ntx, nty,
                           batch, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b, (int64_t)s3c,
                           s3h, input1, input2, input3, gradoutput, gradinput1, gradinput2, gradinput3);
    } else {
        MEMC_FI_BWD_PK(false, 256, nty, ntiles);
    }
#undef MEMC_FI_BWD_PK_PART
#undef MEMC_FI_BWD_PK
    return launch_status() == 0 ? (ws < w ? 2 : 1) : -1;
}

}  // namespace memc

#ifdef MEMC_MEASURE
// device buffer of gridDim.x * 16 uint64 for the timestamp arms (fi_bwd variants 9 and 28); tools/trace_kernel.py
extern "C" int memc_debug_set_trace_buffer(void *p)
{
    unsigned long long *q = (unsigned long long *)p;
    const int a = hipMemcpyToSymbol(HIP_SYMBOL(memc::g_trace_buf), &q, sizeof(q)) == hipSuccess ? 0 : -1;
    return a == 0 ? memc::fi_bwd_c3_arms_set_trace_buffer(q) : a;
}
#endif
