// fi_bwd_cn.hip -- the scattering backward passes at many channels (C >= 4), gfx950: FilterInterpolation
// (fs == 4) and, with the same owner kernel on a 2 x 2 window, the bilinear warp (Interpolation / InterpolationCh; the
// second half of this file).
//
// Same operator as fi_bwd_tiled_c3 (filter_interpolation.hip; reference kernel my_lib_kernel.cu:1220-1515), built for
// many channels.  With C channels a site issues 16 * C scattered adds into gradinput1; issued as global atomics
// (fi_bwd_direct_fs4) that is 1024 atomics per site at C = 64, and the chip retires ~290 G of them per second
// whatever the working set (tools/probes/run_probe.py atomics): 163 ms for 8 x 64 x 720 x 1280, a 128-thread CPU's
// speed.  LDS accumulator planes (the RGB kernel's way) would still cost ~15 clk per wave-add.
//
// The observation this file is built on: the scatter PATTERN and its COEFFICIENTS do not depend on the channel.
//     gradinput1[c][cell] += K(site, tap) * gradoutput[c][site],     K = (bilinear weight of the tap's quadrant) * tap
//     gradinput3[tap]      = wq * S(site, tap),                       S = sum_c gradoutput[c][site] * input1[c][cell]
//     gradinput2           = sum_taps (+-w) * S * tap
// so the pattern is resolved ONCE per tile and then replayed for every channel as a gather:
//
//   fi_bwd_taps_c4n    (site tiles, 64 x 16, the forward c4n pipeline: chunks of four channels staged into LDS pixel
//                       quads while the previous chunk is consumed) accumulates S in registers -- 4 FMAs per LDS
//                       read -- and writes gradinput3 / gradinput2 (assigned) plus the target bounding boxes of the
//                       tile's four 64 x 4 strips of sites;
//   fi_bwd_image_owner (CELL tiles, 64 x 16, owner-computes) finds the site strips whose boxes reach its cells, turns
//                       their taps into a list per cell in LDS (a head table indexed by tap + a CSR tail; LDS integer
//                       atomics only), keeps the lists in registers, then per chunk of four channels stages
//                       gradoutput of the contributing sites into LDS and every cell replays its list: one
//                       ds_read_b128 + 4 FMAs per entry.  The result is STORED to
//                       gradinput1 with plain 16-byte stores: no global atomics, no read-modify-write, each cell
//                       written by exactly one workgroup.  For this class of channel counts gradinput1 is stored on
//                       EVERY path (the launcher clears it before falling back to the direct kernel), so a caller
//                       need not zero-fill it.
//
// Widths that are not multiples of four (round 6; the reference serves every width with its one kernel, my_lib_kernel.cu:10-15):
// the kernels work on the whole quads of every row, Wq = W & ~3 columns of SITES and of CELLS -- the image is still W wide for
// every clamp and validity test, and kernel A's RAG instantiation stages boxes that reach into the last, partial quad
// (memc_tile.hpp: loads moved left and rotated back).  What is left is one to three columns:
//   * their CELLS are cleared first (bwd_cn_zero_tail) and only ever reached by atomics: a site whose window touches a column
//     >= Wq counts as "far" (site_far), so the owners never see it and the far-site kernel adds its whole window;
//   * their SITES are one WAVE each in fi_bwd_tail_sites / bl_bwd_tail_sites (lanes over the channels: tap and flow gradients from
//     global memory, the image gradient by atomics), queued behind the owners' stores like the far sites.
// Rounds 2-5 sent such shapes to the direct kernels (16 C global atomics per site: 13-41x slower, profiles/r04_slow_paths.txt).
//
// Sites whose window leaves the owner's search window (kOwnRX / kOwnRY site tiles around the site's own tile: motion
// beyond ~192 px horizontally or ~64 px vertically) are "far": every owner skips them and a third kernel,
// fi_bwd_far_sites, adds their image gradient with global atomics after the owners have stored theirs (kernel A
// flags the site tiles that have any; the kernel's other workgroups exit at once).
#include "memc_common.hpp"
#include "memc_internal.h"
#include "memc_tile.hpp"
#include "memc_fi.hpp"
#include "memc_scratch.hpp"

namespace memc {

constexpr int kOwnRX = 3, kOwnRY = 4;          // owner search window, in site tiles of 64 x 16: 7 x 9 tiles, 252 strips
constexpr int kTileHasFar = 1 << 30;           // in BBox::h of a site tile's target box: some of its sites are far

// The two window shapes the owner kernel serves.  Tap (k, m), k, m < kN, of a site with integer target (ix, iy) lands on
// cell (clamp(iy + kOff + k), clamp(ix + kOff + m)); its bilinear weight is wa * wb with wa = m < kN / 2 ? 1 - a : a.
struct FpFilter {                              // FilterInterpolation: 4 x 4 window around the 2 x 2 sample, times a tap
    static constexpr int kN = 4, kOff = -1;
    static constexpr bool kTaps = true;
    static __device__ __forceinline__ FiSite locate(int x, int y, int W, int H, float fx, float fy)
    {
        return fi_locate(x, y, W, H, fx, fy);
    }
};
struct FpBilinear {                            // Interpolation / InterpolationCh: the 2 x 2 sample itself
    static constexpr int kN = 2, kOff = 0;
    static constexpr bool kTaps = false;
    static __device__ __forceinline__ FiSite locate(int x, int y, int W, int H, float fx, float fy)
    {
        const BlSite b = bl_locate<true>(x, y, W, H, fx, fy);      // R = min(L + 1, W - 1) == clamp(L + 1)
        FiSite s;
        s.ix = b.L;  s.iy = b.T;  s.a = b.a;  s.b = b.b;  s.valid = b.valid;
        return s;
    }
};

// one site: does its (clamped) window reach a cell tile outside the search window of the site's own tile?
// ... or (a ragged width: Wq = W & ~3 < W) a cell of the one to three columns no owner stores?
template <class FP>
__device__ __forceinline__ bool site_far(int x, int y, int ix, int iy, int W, int H, int Wq)
{
    const int tx = x >> 6, ty = y >> 4;
    const int cl = clampi(ix + FP::kOff + FP::kN - 1, W - 1);      // the window's last column
    const int c0 = clampi(ix + FP::kOff, W - 1) >> 6, c1 = cl >> 6;
    const int r0 = clampi(iy + FP::kOff, H - 1) >> 4, r1 = clampi(iy + FP::kOff + FP::kN - 1, H - 1) >> 4;
    return c0 < tx - kOwnRX || c1 > tx + kOwnRX || r0 < ty - kOwnRY || r1 > ty + kOwnRY || cl >= Wq;
}
__device__ __forceinline__ bool fi_site_far(int x, int y, int ix, int iy, int W, int H, int Wq)
{
    return site_far<FpFilter>(x, y, ix, iy, W, H, Wq);
}

// What the owners cull by: the target bounding box of every 64 x 4 STRIP of sites (one wave of a producer's site tile:
// 16 lanes x 4 sites per row, 4 rows), four per site tile; tbox[(tile * 4 + strip)].  BBox::h of a tile's strip 0 also
// carries kTileHasFar.  A lane passes the box of its own near sites (or an empty one); one wave-level reduction, no
// barrier.
__device__ __forceinline__ void strip_box_store(BBox *tbox, int64_t tile, int cmin, int cmax, int rmin, int rmax,
                                                bool tile_has_far)
{
    cmin = wave_min_i32(cmin);
    cmax = -wave_min_i32(-cmax);
    rmin = wave_min_i32(rmin);
    rmax = -wave_min_i32(-rmax);
    const unsigned tid = threadIdx.x;
    if ((tid & 63) == 0) {
        BBox bx;
        if (cmin > cmax) {
            bx.x0 = bx.y0 = bx.w = bx.h = 0;
        } else {
            bx.x0 = cmin & ~3;
            bx.w = (cmax | 3) + 1 - bx.x0;
            bx.y0 = rmin;
            bx.h = rmax + 1 - rmin;
        }
        if (tid == 0 && tile_has_far) bx.h |= kTileHasFar;
        tbox[tile * 4 + (tid >> 6)] = bx;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Rare paths, one site at a time from global memory.
// ---------------------------------------------------------------------------------------------------------
// image gradient of a far site: 16 * C global atomics
__device__ __noinline__ void fi_bwd_site_image_atomics(int x, int y, int W, int H, int C, float *gin1_b, int64_t s1c,
                                                       int s1h, const float *flow_p, int64_t s2c, const float *tap_p,
                                                       int64_t s3c, const float *gout_p)
{
    const FiSite s = fi_locate(x, y, W, H, flow_p[0], flow_p[s2c]);
    if (!s.valid) return;
    for (int k = 0; k < 4; k++) {
        const int ro = clampi(s.iy - 1 + k, H - 1) * s1h;
        for (int m = 0; m < 4; m++) {
            const int co = clampi(s.ix - 1 + m, W - 1);
            const float wa = m < 2 ? (1 - s.a) : s.a, wb = k < 2 ? (1 - s.b) : s.b;
            const float kk = (wa * wb) * tap_p[(k * 4 + m) * s3c];
            for (int c = 0; c < C; c++) atomic_add_f32(gin1_b + c * s1c + ro + co, gout_p[c * s1c] * kk);
        }
    }
}

// tap and flow gradients of one site (a site no band of its tile covered)
__device__ __noinline__ void fi_bwd_site_taps_cn(int x, int y, int W, int H, int C, const float *in_b, int64_t s1c,
                                                 int s1h, const float *flow_p, float *g2, int64_t s2c,
                                                 const float *tap_p, float *g3, int64_t s3c, const float *gout_p)
{
    const FiSite s = fi_locate(x, y, W, H, flow_p[0], flow_p[s2c]);
    if (!s.valid) return;
    float gx = 0.0f, gy = 0.0f;
    for (int k = 0; k < 4; k++) {
        const float *row = in_b + (int64_t)clampi(s.iy - 1 + k, H - 1) * s1h;
        for (int m = 0; m < 4; m++) {
            const float *p = row + clampi(s.ix - 1 + m, W - 1);
            float sv = 0.0f;
            for (int c = 0; c < C; c++) sv += gout_p[c * s1c] * p[c * s1c];
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

// ---------------------------------------------------------------------------------------------------------
// Kernel A: tap and flow gradients.
// ---------------------------------------------------------------------------------------------------------
// S[tap][j] += sum over the chunk's four channels of gradoutput * staged image, for the sites in `sel`.
// Sites not in `sel` read the ZERO pixel quad kept behind the staged image (index `zero`) and add 0 * 0: reading a
// staged pixel instead would turn an Inf / NaN there into 0 * Inf = NaN in the sums of sites it has nothing to do with.
__device__ __forceinline__ void fi_bwd_taps_accum(const Region &r, const FiSite4 &g, unsigned sel,
                                                  const f32x4 (&go)[4], int W, int H, const f32x4 *tile, int zero,
                                                  f32x4 (&S)[16])
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool on = (sel >> j) & 1;
        int ro[4], co[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ro[k] = on ? (clampi(g.iy[j] - 1 + k, H - 1) - r.y0) * r.pitch : zero;
            co[k] = on ? swz_col(clampi(g.ix[j] - 1 + k, W - 1) - r.x0) : 0;
        }
        // unselected sites still issue their reads (at the zero pixel) and add zeros: no control flow in the nest
        const float g0 = on ? go[0][j] : 0.0f, g1 = on ? go[1][j] : 0.0f, g2 = on ? go[2][j] : 0.0f,
                    g3 = on ? go[3][j] : 0.0f;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            f32x4 v[4];
#pragma unroll
            for (int m = 0; m < 4; m++) v[m] = tile[ro[k] + co[m]];
#pragma unroll
            for (int m = 0; m < 4; m++) {
                float acc = S[k * 4 + m][j];
                acc += g0 * v[m][0];  acc += g1 * v[m][1];  acc += g2 * v[m][2];  acc += g3 * v[m][3];
                S[k * 4 + m][j] = acc;
            }
        }
        __builtin_amdgcn_sched_barrier(0);                 // one site's sixteen reads at a time: S needs the registers
    }
}

template <bool RAG>
__global__ __launch_bounds__(256, 2) void fi_bwd_taps_c4n(
    int W, int H, int Wq, int C, int tiles_x, int tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    const float *__restrict__ gout, float *__restrict__ gin1, float *__restrict__ gin2, float *__restrict__ gin3,
    BBox *__restrict__ tbox)
{
    constexpr int LX = 16;
    using G = TileGeom<LX>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    int *bb = reinterpret_cast<int *>(smem + G::kCapPx * 16);
    constexpr int kZeroPx = G::kCapPx + 4;                 // a pixel quad of zeros behind the staged image and the boxes
    if (threadIdx.x == 0) tile[kZeroPx] = f32x4{0.f, 0.f, 0.f, 0.f};     // (visible after tile_bbox's barrier)

    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, batch);
    const int b = tc.b;
    const unsigned tid = tid_now();
    const int x = tc.tx * G::kTW + 4 * (int)(tid % LX), y = tc.ty * G::kTH + (int)(tid / LX);
    const bool inb = x < Wq && y < H;                      // (Wq: the whole quads of a row; == W unless RAG)
    const int xs = min(x, Wq - 4), ys = min(y, H - 1);
    const float *flow_b = flow + b * s2b, *filt_b = filt + b * s3b, *gout_b = gout + b * s1b;
    float *gin2_b = gin2 + b * s2b, *gin3_b = gin3 + b * s3b;
    const unsigned o1 = 4u * (unsigned)(ys * s1h + xs), o2 = 4u * (unsigned)(ys * s2h + xs),
                   o3 = 4u * (unsigned)(ys * s3h + xs);
    const f32x4 fx4 = ld_stream4_u(flow_b, o2), fy4 = ld_stream4_u(flow_b + s2c, o2);

    FiSite4 g;
    g.valid = 0;
    unsigned far = 0;
    int cmin = INT_MAX, cmax = -1, rmin = INT_MAX, rmax = -1;          // staging box: every valid site
    int ncmin = INT_MAX, ncmax = -1, nrmin = INT_MAX, nrmax = -1;      // target box of the sites the owners take
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const FiSite s = fi_locate(x + j, y, W, H, fx4[j], fy4[j]);
        g.ix[j] = s.ix; g.iy[j] = s.iy; g.a[j] = s.a; g.b[j] = s.b;
        if (inb && s.valid) {
            g.valid |= 1u << j;
            const int c0 = max(s.ix - 1, 0), c1 = min(s.ix + 2, W - 1), r0 = max(s.iy - 1, 0), r1 = min(s.iy + 2, H - 1);
            cmin = min(cmin, c0);  cmax = max(cmax, c1);  rmin = min(rmin, r0);  rmax = max(rmax, r1);
            if (fi_site_far(x + j, y, s.ix, s.iy, W, H, Wq)) {
                far |= 1u << j;
            } else {
                ncmin = min(ncmin, c0);  ncmax = max(ncmax, c1);  nrmin = min(nrmin, r0);  nrmax = max(nrmax, r1);
            }
        }
    }
    const BBox box = tile_bbox<LX>(cmin, cmax, rmin, rmax, bb);
    // the strips' target boxes; h's bit 30 of strip 0: the tile has far sites (fi_bwd_far_sites adds their image gradient
    // after the owners have stored)
    strip_box_store(tbox, ((int64_t)b * tiles_y + tc.ty) * tiles_x + tc.tx, ncmin, ncmax, nrmin, nrmax,
                    __syncthreads_or(far != 0) != 0);
    const Bands bands = make_bands<LX>(box);
    const float *in_b = in1 + b * s1b;

    f32x4 S[16];
#pragma unroll
    for (int k = 0; k < 16; k++) S[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    unsigned done = 0;
#pragma unroll 1
    for (int bi = 0; bi < bands.n; bi++) {
        const Region r = band_region(box, bands, bi, RAG ? W : 0);
        // the window corners fi_covered derives from ix / iy (sixteen clamped values) are loop-invariant: hoisted out of the
        // band loop they lived in private scratch through the channel loop (28 dwords per lane, rounds 2-5); opaque here,
        // they are recomputed per band -- sixteen integer instructions
#pragma unroll
        for (int j = 0; j < 4; j++) asm volatile("" : "+v"(g.ix[j]), "+v"(g.iy[j]));
        const unsigned sel = inb ? fi_covered(r, g, W, H) & ~done : 0u;
        if (bi > 0 && !__syncthreads_or(sel != 0)) continue;
        done |= sel;
        const StageSlot sl = stage_slots(r);
        f32x4 go[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const f32x4 gl = ld_stream4_u(gout_b + min(c, C - 1) * s1c, o1);
            go[c] = c < C ? gl : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll 1
        for (int c0 = 0; c0 < C; c0 += 4) {
            {
                // not carried across the accumulation (S needs the registers); wave-uniform plane bases + one 32-bit
                // offset per slot: per-lane 64-bit pointers, strength-reduced over this loop, are what spills here
                StageRegs<4> sr;
#pragma unroll
                for (int it = 0; it < kStageIts; it++) {
                    // (RAG: the box's last quad may reach past the row -- loaded from rs columns further left, rotated back by
                    // tile_stage_store; what lies past the row is never gathered: coordinates are clamped)
                    const int rs = RAG ? tail_shift(r.x0 + 4 * sl.q[it], W) : 0;
                    const unsigned off = sl.row[it] < r.h
                                             ? 4u * (unsigned)((r.y0 + sl.row[it]) * s1h + r.x0 + 4 * sl.q[it] - rs) : 0u;
#pragma unroll
                    for (int c = 0; c < 4; c++) sr.v[it][c] = ld_cached4_u(in_b + min(c0 + c, C - 1) * s1c, off);
                }
                tile_stage_store<4, RAG>(r, sl, sr, tile);
            }
            __syncthreads();
            fi_bwd_taps_accum(r, g, sel, go, W, H, tile, kZeroPx, S);
            // the next chunk's gradoutput, once this chunk's has been used (one register set)
            const int cn = c0 + 4 < C ? c0 + 4 : c0;
#pragma unroll
            for (int c = 0; c < 4; c++) {                  // a ragged last chunk: channels past C contribute zeros
                const f32x4 gl = ld_stream4_u(gout_b + min(cn + c, C - 1) * s1c, o1);
                go[c] = cn + c < C ? gl : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
        }
    }

    // epilogue: gradinput3 = wq * S, gradinput2 from S * tap; sites that are invalid (or not covered: redone below)
    // store zeros -- both tensors are fully defined by this kernel.  One tap row at a time, the next row's taps in
    // flight: all sixteen tap quads at once would not fit next to S.
    if (inb) {
        const unsigned live = g.valid & done;
        const f32x4 fxe = ld_cached4_u(flow_b, o2), fye = ld_cached4_u(flow_b + s2c, o2);
        f32x4 a4, b4;
#pragma unroll
        for (int j = 0; j < 4; j++) {                      // alpha / beta again (not kept through the channel loop)
            const FiSite s = fi_locate(x + j, y, W, H, fxe[j], fye[j]);
            a4[j] = s.a;
            b4[j] = s.b;
        }
        f32x4 gx4 = {0.f, 0.f, 0.f, 0.f}, gy4 = gx4;
        f32x4 tp[4], tn[4];
#pragma unroll
        for (int m = 0; m < 4; m++) tp[m] = ld_stream4_u(filt_b + m * s3c, o3);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (k < 3) {
#pragma unroll
                for (int m = 0; m < 4; m++) tn[m] = ld_stream4_u(filt_b + ((k + 1) * 4 + m) * s3c, o3);
            }
#pragma unroll
            for (int m = 0; m < 4; m++) {
                f32x4 gt;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float a = a4[j], bt = b4[j];
                    const float wa = m < 2 ? (1 - a) : a, wb = k < 2 ? (1 - bt) : bt;
                    const float sv = ((live >> j) & 1) ? S[k * 4 + m][j] : 0.0f;
                    gt[j] = (wa * wb) * sv;
                    const float st = sv * tp[m][j];
                    gx4[j] += (m < 2 ? -wb : wb) * st;
                    gy4[j] += (k < 2 ? -wa : wa) * st;
                }
                st_stream4_u(gin3_b + (k * 4 + m) * s3c, o3, gt);
            }
#pragma unroll
            for (int m = 0; m < 4; m++) tp[m] = tn[m];
            __builtin_amdgcn_sched_barrier(0);
        }
        st_stream4_u(gin2_b, o2, gx4);
        st_stream4_u(gin2_b + s2c, o2, gy4);
    }
    unsigned slow = inb ? g.valid & ~done : 0u;            // not coverable within kMaxBands bands: from global memory
    while (slow) {
        const int j = __ffs(slow) - 1;
        slow &= slow - 1;
        fi_bwd_site_taps_cn(x + j, y, W, H, C, in_b, s1c, s1h, flow_b + o2 / 4 + j, gin2_b + o2 / 4 + j, s2c,
                            filt_b + o3 / 4 + j, gin3_b + o3 / 4 + j, s3c, gout_b + o1 / 4 + j);
    }
}

// The image gradient of the far sites, after the owners have stored theirs: one workgroup per site tile, gone at once
// unless kernel A flagged the tile.  Round 6: a wave takes its far sites ONE AT A TIME with all 64 lanes, lane l the channels
// l, l + 64, ... (16 atomics per channel): one lane per site -- 16 C atomics in a row -- made the band of sites along a ragged
// width's last columns 1.5 ms of a 6.6 ms call (profiles/r06_many_channel_ragged_kernel_trace.txt).
__global__ __launch_bounds__(256) void fi_bwd_far_sites(
    int W, int H, int Wq, int C, int tiles_x, int tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ flow, const float *__restrict__ filt, const float *__restrict__ gout,
    float *__restrict__ gin1, const BBox *__restrict__ tbox)
{
    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, batch);
    if (!(tbox[(((int64_t)tc.b * tiles_y + tc.ty) * tiles_x + tc.tx) * 4].h & kTileHasFar)) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int x0 = tc.tx * 64 + 4 * (int)(threadIdx.x % 16), y = tc.ty * 16 + (int)(threadIdx.x / 16);
    const bool inq = x0 < Wq && y < H;                     // (the columns behind the whole quads: fi_bwd_tail_sites)
    const float *flow_b = flow + tc.b * s2b, *filt_b = filt + tc.b * s3b, *gout_b = gout + tc.b * s1b;
    float *gin1_b = gin1 + tc.b * s1b;
    const float *flow_p = flow_b + (int64_t)min(y, H - 1) * s2h + min(x0, Wq - 4);
    for (int j = 0; j < 4; j++) {
        const FiSite sj = fi_locate(x0 + j, y, W, H, flow_p[j], flow_p[s2c + j]);
        unsigned long long todo = __builtin_amdgcn_ballot_w64(inq && sj.valid && fi_site_far(x0 + j, y, sj.ix, sj.iy, W, H, Wq));
        while (todo) {                                     // (wave-uniform) the far site of lane `src`, by the whole wave
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int x = __builtin_amdgcn_readlane(x0, src) + j, ys = __builtin_amdgcn_readlane(y, src);
            const float *fp = flow_b + (int64_t)ys * s2h + x, *tap_p = filt_b + (int64_t)ys * s3h + x;
            const FiSite s = fi_locate(x, ys, W, H, fp[0], fp[s2c]);                      // (the same bits as sj of that lane)
            for (int c = lane; c < C; c += kWave) {
                const float g = gout_b[c * s1c + (int64_t)ys * s1h + x];
                float *q = gin1_b + c * s1c;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int ro = clampi(s.iy - 1 + k, H - 1) * s1h;
#pragma unroll
                    for (int m = 0; m < 4; m++) {
                        const float wa = m < 2 ? (1 - s.a) : s.a, wb = k < 2 ? (1 - s.b) : s.b;
                        atomic_add_f32(q + ro + clampi(s.ix - 1 + m, W - 1), g * ((wa * wb) * tap_p[(k * 4 + m) * s3c]));
                    }
                }
            }
        }
    }
}

// A ragged width's one to three columns behind the whole quads (see the top of the file).
// gradinput1 = 0 in columns [Wq, W) of a strided [batch, channel, h, w] view: one lane per row
__global__ __launch_bounds__(256) void bwd_cn_zero_tail(float *__restrict__ p, int Wq, int W, int h, int channel, int batch,
                                                        int64_t sb, int64_t sc, int sh)
{
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x, rows = (int64_t)batch * channel * h;
    if (row >= rows) return;
    const int y = (int)(row % h), c = (int)((row / h) % channel);
    float *q = p + (row / h / channel) * sb + c * sc + (int64_t)y * sh;
    for (int x = Wq; x < W; x++) q[x] = 0.0f;
}
// sum over the 64 lanes of a wave, result in every lane (a butterfly of shuffles; rare path)
__device__ __forceinline__ float wave_allsum_f32(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}
// The sites of those columns, one WAVE each, lane l the channels l, l + 64, ...: a lane adds its channels' image gradient with
// atomics (after the owners' stores, like fi_bwd_far_sites) and keeps partial tap sums, which the wave then adds up.  (One
// LANE per site -- 16 C dependent atomics and 32 C loads in a row -- took 3 ms for the 11 520 tail sites of 8 x 64 x 720 x 1278.)
__global__ __launch_bounds__(256) void fi_bwd_tail_sites(
    int W, int H, int Wq, int C, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    const float *__restrict__ gout, float *__restrict__ gin1, float *__restrict__ gin2, float *__restrict__ gin3)
{
    const int nt = W - Wq, lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave, n = (int64_t)batch * H * nt;
    if (i >= n) return;                                    // (wave-uniform)
    const int x = Wq + (int)(i % nt), y = (int)((i / nt) % H), b = (int)(i / nt / H);
    const float *flow_p = flow + b * s2b + (int64_t)y * s2h + x, *tap_p = filt + b * s3b + (int64_t)y * s3h + x;
    const float *gout_p = gout + b * s1b + (int64_t)y * s1h + x, *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    float *g2 = gin2 + b * s2b + (int64_t)y * s2h + x, *g3 = gin3 + b * s3b + (int64_t)y * s3h + x;
    const FiSite s = fi_locate(x, y, W, H, flow_p[0], flow_p[s2c]);
    if (!s.valid) {                                        // gradinput2 / gradinput3 are ASSIGNED: invalid sites store zeros
        if (lane < 16) g3[lane * s3c] = 0.0f;
        if (lane < 2) g2[lane * s2c] = 0.0f;
        return;
    }
    float sv[16];
#pragma unroll
    for (int t = 0; t < 16; t++) sv[t] = 0.0f;
    for (int c = lane; c < C; c += kWave) {
        const float g = gout_p[c * s1c];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int ro = clampi(s.iy - 1 + k, H - 1) * s1h;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int o = ro + clampi(s.ix - 1 + m, W - 1);
                const float wa = m < 2 ? (1 - s.a) : s.a, wb = k < 2 ? (1 - s.b) : s.b;
                sv[k * 4 + m] += g * in_b[c * s1c + o];
                atomic_add_f32(gin1_b + c * s1c + o, g * ((wa * wb) * tap_p[(k * 4 + m) * s3c]));
            }
        }
    }
    float gx = 0.0f, gy = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const float tot = wave_allsum_f32(sv[k * 4 + m]);
            const float wa = m < 2 ? (1 - s.a) : s.a, wb = k < 2 ? (1 - s.b) : s.b;
            if (lane == k * 4 + m) g3[(k * 4 + m) * s3c] = (wa * wb) * tot;
            const float st = tot * tap_p[(k * 4 + m) * s3c];
            gx += (m < 2 ? -wb : wb) * st;
            gy += (k < 2 ? -wa : wa) * st;
        }
    if (lane == 0) {
        g2[0] = gx;
        g2[s2c] = gy;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Kernel B: image gradient, owner-computes over 64 x 16 cell tiles, 512 lanes (one workgroup per CU).
//
//   1. candidate site tiles (target box reaches this cell tile), from the boxes kernel A left;
//   2. one pass over their sites (flow only): per-cell tap counts (LDS integer atomics) and the box of the sites that
//      contribute;
//   3. one pass over the contributing box (flow + taps): every tap that lands here appends (K, site slot) to its
//      cell's list.  A list is kListHead entries in a fixed table (slot r of every cell side by side) plus a tail in a
//      CSR area (exclusive scan of the tail lengths) -- 16 taps per cell on average, but a compressing flow gives
//      a heavy tail (the benchmark's flow: 9 % of the cells above 24, hot cells above 100);
//   4. every lane copies the heads of its two cells into REGISTERS for the whole channel loop.  The tails are cut
//      into segments of kSegLen entries and dealt out to ALL lanes (two per lane, in registers as well): a hot
//      cell's work is spread over the workgroup instead of serialising its owner;
//   5. per chunk of four channels: gradoutput of the contributing sites is staged as one float4 per site slot; every
//      entry is one ds_read_b128 + 4 FMAs; segment sums go through LDS to the owners; then a 4 x 4 transpose inside
//      lane quads (DPP) and one 16-byte read-modify-write per lane and cell row.
// A contributing box larger than the staging area, or tails that do not fit (converging flow), are processed in
// slabs of site rows, each slab one count / fill / replay round.
//
// LDS: head table (K fp32, slot u16) -- once the heads are in registers the same bytes hold the staged gradoutput,
// the segment sums and the segment table -- | tail K, tail slot | one word per cell (cursor : count) | tail offsets |
// control words.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSegLen = 8;                     // tail segment length; two segments per lane
constexpr unsigned kHeadEmpty = 0x7fc5a5a5u;   // a NaN payload no coefficient has: an unclaimed head slot
constexpr int kOwnCand = (2 * kOwnRX + 1) * (2 * kOwnRY + 1);
static_assert(kOwnCand <= 64, "one lane per candidate site tile");

// Cell tile 64 x TH, 32 * TH lanes (two cells per lane: rows r and r + TH / 2); kHead = kN * kN head slots per cell.
//   TH = 16: 512 lanes, 157 KB of LDS (4 x 4 window), one workgroup per CU.
//   (TH = 8 -- 256 lanes, 78 KB, two per CU -- was measured in round 2, +2 ... -2 %, and is no longer built: its tail
//   area, 4096 entries, is smaller than one row of candidate sites can need, see the assertion below.)
template <class FP, int TH>
struct OwnGeom {
    static constexpr int kHead = FP::kN * FP::kN;          // entries per cell a lane keeps in registers (one per tap index)
    static constexpr int kThreads = 32 * TH, kCells = 64 * TH;
    static constexpr int kTailCap = (FP::kN == 4 ? 8 : 4) * kCells;      // tail entries per slab
    // the slab loop halves its rows until a slab's tails fit; ONE row of sites -- (2 kOwnRX + 1) site tiles of 64 sites,
    // kHead taps each -- is as far as it can go, so that must fit whatever the search window is widened to
    static_assert((2 * kOwnRX + 1) * 64 * kHead <= kTailCap, "one row of candidate sites must fit the tail area");
    static constexpr int kSegCap = 2 * kThreads;
    static constexpr int kSlotCap = TH == 16 ? 3072 : 2048;        // sites staged per slab; slot kSlotCap holds zeros
    static constexpr int kHK = 0, kHS = kHK + kHead * kCells * 4, kHeadEnd = kHS + kHead * kCells * 2;
    // staging area, segment sums, segment table: inside the head table's bytes once the heads have been copied to
    // registers, if they fit there (4 x 4 window); behind it otherwise
    static constexpr int kAliasNeed = ((kSlotCap + 1) * 16 + 127) / 128 * 128 + kSegCap * 16 + kSegCap * 4;
    static constexpr bool kAlias = kAliasNeed <= kHeadEnd;
    static constexpr int kG = kAlias ? 0 : kHeadEnd, kPart = kG + ((kSlotCap + 1) * 16 + 127) / 128 * 128,
                         kSeg = kPart + kSegCap * 16, kAliasEnd = kSeg + kSegCap * 4;
    static_assert(kSlotCap <= 8 * kThreads, "two float4 slots of staging per lane");
    static_assert(kSlotCap % 256 == 0, "the slot swizzle permutes aligned blocks of 256");
    static constexpr int kTK = kAlias ? kHeadEnd : kAliasEnd, kTS = kTK + kTailCap * 4, kOc = kTS + kTailCap * 2,
                         kPres = kOc + kCells * 4, kToff = kPres + kCells * 4, kCtl = kToff + kCells * 2,
                         kBytes = kCtl + 512;
    static_assert(kBytes * (TH == 16 ? 1 : 2) <= 160 * 1024, "LDS per CU");
};

struct OwnCtl {                                // control words in LDS
    int ncand;                                 // candidate site strips (their list lives in the tail-offset area)
    int ax0, ax1, ay0, ay1;                    // box of the contributing sites
    unsigned wave_sum[8], wave_sum2[8];
};

// Which taps of the four sites of one quad land on the cell tile at (tx0, ty0)?  rowm / colm: bit k of site j's mask =
// tap row / column k lands inside; a site contributes iff both are non-zero.
struct QuadHits {
    unsigned rowm[4], colm[4], any;
    int ix[4], iy[4];
    float a[4], b[4];
};
template <class FP, int TH>
__device__ __forceinline__ QuadHits own_quad_hits(int x, int y, int W, int H, int Wq, int tx0, int ty0, f32x4 fx4, f32x4 fy4)
{
    QuadHits h;
    h.any = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const FiSite s = FP::locate(x + j, y, W, H, fx4[j], fy4[j]);
        h.ix[j] = s.ix; h.iy[j] = s.iy; h.a[j] = s.a; h.b[j] = s.b;
        unsigned rm = 0, cm = 0;
        if (s.valid && !site_far<FP>(x + j, y, s.ix, s.iy, W, H, Wq)) {
#pragma unroll
            for (int k = 0; k < FP::kN; k++) {
                rm |= (unsigned)((unsigned)(clampi(s.iy + FP::kOff + k, H - 1) - ty0) < (unsigned)TH) << k;
                cm |= (unsigned)((unsigned)(clampi(s.ix + FP::kOff + k, W - 1) - tx0) < 64u) << k;
            }
        }
        const bool hit = rm && cm;
        h.rowm[j] = hit ? rm : 0;
        h.colm[j] = hit ? cm : 0;
        h.any |= hit ? 1u << j : 0u;
    }
    return h;
}

// pres[cell] |= 1 << tap index for every tap of the quad that lands on the tile; oc[cell] += 1 << 16 for those whose
// index was already there (the cell's tail length)
template <class FP>
__device__ __forceinline__ void own_count(const QuadHits &h, unsigned *oc, unsigned *pres, int W, int H, int tx0,
                                          int ty0)
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!((h.any >> j) & 1)) continue;
#pragma unroll
        for (int k = 0; k < FP::kN; k++) {
            if (!((h.rowm[j] >> k) & 1)) continue;
            const int rc = (clampi(h.iy[j] + FP::kOff + k, H - 1) - ty0) * 64 - tx0;
#pragma unroll
            for (int m = 0; m < FP::kN; m++)
                if ((h.colm[j] >> m) & 1) {
                    const int ci = rc + clampi(h.ix[j] + FP::kOff + m, W - 1);
                    // the first tap of this index at the cell will get the head slot; every further one is tail
                    const unsigned bit = 1u << (k * FP::kN + m);
                    if (atomicOr(pres + ci, bit) & bit) atomicAdd(oc + ci, 0x10000u);
                }
        }
    }
}

// Staging slots are XOR-swizzled in their low four bits with bits 4..7: a 16-lane group of a ds_read_b128 conflicts when
// two of its lanes read different slots 16 (or 32) apart, which is exactly what neighbouring cells' sources do under a
// compressing or expanding flow (slot distance = cell distance * (1 - gradient)).  A bijection of every aligned 256.
__device__ __forceinline__ int swz_slot(int s) { return s ^ ((s >> 4) & 15); }

template <int K>
__device__ __forceinline__ float quad_bcast(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), K * 0x55, 0xf, 0xf, true));
}
// 4 x 4 transpose inside every aligned group of four lanes: lane i (of the group) gets component i of lanes 0..3
__device__ __forceinline__ f32x4 quad_transpose(f32x4 v, unsigned my)
{
    f32x4 out;
#define MEMC_QT(K)                                                                                                 \
    {                                                                                                              \
        const float t0 = quad_bcast<K>(v[0]), t1 = quad_bcast<K>(v[1]), t2 = quad_bcast<K>(v[2]),                  \
                    t3 = quad_bcast<K>(v[3]);                                                                      \
        out[K] = my == 0 ? t0 : (my == 1 ? t1 : (my == 2 ? t2 : t3));                                              \
    }
    MEMC_QT(0) MEMC_QT(1) MEMC_QT(2) MEMC_QT(3)
#undef MEMC_QT
    return out;
}

// TR (measurement build): thread 0 accumulates the shader clocks of every phase into trace[blockIdx.x * 16 + ...]
// (tools/trace_kernel.py fi_bwd_cn): 0 whole life, 1 candidates + count pass, 2 slab recounts, 3 scan, 4 fill,
// 5 lists -> registers, 6 replay, 7 slab rounds, 8 candidate tiles, 9 tail segments (last slab), 10 site-box area.
template <class FP, int TH, bool TR>
__global__ __launch_bounds__(32 * TH, TH == 16 ? 1 : 2) void fi_bwd_image_owner(
    int W, int H, int Wq, int C, int tiles_x, int tiles_y, int site_tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ flow, const float *__restrict__ filt, const float *__restrict__ gout,
    float *__restrict__ gin1, const BBox *__restrict__ tbox, unsigned long long *__restrict__ trace)
{
    unsigned long long tr_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tr_t = 0, tr_t0 = 0;
#define MEMC_TR_BEGIN() do { if (TR) tr_t = __builtin_readcyclecounter(); } while (0)
#define MEMC_TR_END(slot) do { if (TR) { const unsigned long long n_ = __builtin_readcyclecounter(); tr_acc[slot] += n_ - tr_t; tr_t = n_; } } while (0)
    if (TR) tr_t0 = __builtin_readcyclecounter();
    MEMC_TR_BEGIN();
    using Gm = OwnGeom<FP, TH>;
    constexpr int kListHead = Gm::kHead;
    constexpr int kOwnThreads = Gm::kThreads, kOwnCells = Gm::kCells, kTailCap = Gm::kTailCap, kSegCap = Gm::kSegCap,
                  kSlotCap = Gm::kSlotCap;
    constexpr int kOwnLdsHK = Gm::kHK, kOwnLdsHS = Gm::kHS, kOwnLdsG = Gm::kG, kOwnLdsPart = Gm::kPart,
                  kOwnLdsSeg = Gm::kSeg, kOwnLdsTK = Gm::kTK, kOwnLdsTS = Gm::kTS, kOwnLdsOc = Gm::kOc,
                  kOwnLdsPres = Gm::kPres, kOwnLdsToff = Gm::kToff, kOwnLdsCtl = Gm::kCtl;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *HK = reinterpret_cast<float *>(smem + kOwnLdsHK);
    unsigned short *HS = reinterpret_cast<unsigned short *>(smem + kOwnLdsHS);
    f32x4 *g4 = reinterpret_cast<f32x4 *>(smem + kOwnLdsG);                      // } aliases of the head table
    f32x4 *part = reinterpret_cast<f32x4 *>(smem + kOwnLdsPart);                 // }
    unsigned *segtab = reinterpret_cast<unsigned *>(smem + kOwnLdsSeg);          // }
    float *TK = reinterpret_cast<float *>(smem + kOwnLdsTK);
    unsigned short *TS = reinterpret_cast<unsigned short *>(smem + kOwnLdsTS);
    unsigned *oc = reinterpret_cast<unsigned *>(smem + kOwnLdsOc);
    unsigned *pres = reinterpret_cast<unsigned *>(smem + kOwnLdsPres);
    unsigned short *toff = reinterpret_cast<unsigned short *>(smem + kOwnLdsToff);
    OwnCtl *ctl = reinterpret_cast<OwnCtl *>(smem + kOwnLdsCtl);
    static_assert(sizeof(OwnCtl) <= 512, "control words");

    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, batch);
    const int b = tc.b, tx0 = tc.tx * 64, ty0 = tc.ty * TH;      // tiles_y counts CELL tiles (rows of TH)
    const unsigned tid = threadIdx.x;

    // this lane's two cells (rows r and r + TH / 2 of the tile) and, after the quad transpose, its channel of four cells;
    // the channel (my) is folded into the lane's byte offset: 4 * (my * s1c + row * s1h + col)
    // (computed where they are used -- by store_zeros and in front of every slab's replay, from a laundered copy of the thread
    // index: defined once up here they lived through the count / fill phases, whose tap registers pushed them, half
    // computed, into private scratch: 108 bytes per lane, rounds 2-5)
    struct CellOffsets {
        unsigned my, wo0, wo1;
        bool st0, st1;
    };
    auto cell_offsets = [&](unsigned t) {
        CellOffsets o;
        o.my = t & 3;
        const int cell_x = tx0 + (int)(t & 63 & ~3u), cell_y = ty0 + (int)(t >> 6);
        o.st0 = cell_x < Wq && cell_y < H;                  // (cells behind the whole quads: cleared, reached by atomics only)
        o.st1 = cell_x < Wq && cell_y + TH / 2 < H;
        o.wo0 = o.st0 ? 4u * (unsigned)((int64_t)o.my * s1c + (int64_t)cell_y * s1h + cell_x) : 0u;
        o.wo1 = o.st1 ? 4u * (unsigned)((int64_t)o.my * s1c + (int64_t)(cell_y + TH / 2) * s1h + cell_x) : 0u;
        return o;
    };
    float *gin1_b = gin1 + b * s1b;
    // gradinput1 is STORED by this kernel (the first slab assigns, later slabs add): cells nobody reaches get zeros
    auto store_zeros = [&]() {
        const CellOffsets co = cell_offsets(tid);
        const unsigned my = co.my, wo0 = co.wo0, wo1 = co.wo1;
        const bool st0 = co.st0, st1 = co.st1;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < C; c0 += 4) {
            if (c0 + (int)my >= C) continue;               // ragged last chunk: this lane's channel does not exist
            if (st0) *reinterpret_cast<MEMC_GLOBAL f32x4u *>(addr_u(gin1_b + c0 * s1c, wo0)) = z;
            if (st1) *reinterpret_cast<MEMC_GLOBAL f32x4u *>(addr_u(gin1_b + c0 * s1c, wo1)) = z;
        }
    };

    // 1. candidate site strips (64 x 4 sites, four per site tile): those of the search window whose target box reaches
    //    this cell tile.  One lane per strip; the list (tx | ty << 12 | strip << 28) lives where the tail offsets will.
    int *cand = reinterpret_cast<int *>(toff);
    static_assert(kOwnCand * 4 <= 256 && kOwnCand * 4 * 4 <= 64 * TH * 2, "one lane per strip; the list fits the toff area");
    if (tid < 256) {
        const int tile_i = (int)tid >> 2, strip = (int)tid & 3;
        const int dx = tile_i % (2 * kOwnRX + 1) - kOwnRX, dy = tile_i / (2 * kOwnRX + 1) - kOwnRY;
        const int sx = tc.tx + dx, sy = (ty0 >> 4) + dy;   // site tiles are 64 x 16; site_tiles_y of them per image
        bool hit = false;
        if (tile_i < kOwnCand && sx >= 0 && sx < tiles_x && sy >= 0 && sy < site_tiles_y) {
            BBox bx = tbox[(((int64_t)b * site_tiles_y + sy) * tiles_x + sx) * 4 + strip];
            bx.h &= ~kTileHasFar;
            hit = bx.w > 0 && bx.x0 < tx0 + 64 && bx.x0 + bx.w > tx0 && bx.y0 < ty0 + TH && bx.y0 + bx.h > ty0;
        }
        const unsigned long long m = __ballot(hit);
        if ((tid & 63) == 0) ctl->wave_sum[tid >> 6] = (unsigned)__popcll(m);
        // ranks inside the wave now, the waves' offsets after the barrier
        const int rank = __popcll(m & ((1ull << (tid & 63)) - 1ull));
        const int word = sx | (sy << 12) | (strip << 28);
        __syncthreads();
        if (hit) {
            int base = 0;
            for (int w = 0; w < (int)(tid >> 6); w++) base += (int)ctl->wave_sum[w];
            cand[base + rank] = word;
        }
        if (tid == 0) {
            ctl->ncand = (int)(ctl->wave_sum[0] + ctl->wave_sum[1] + ctl->wave_sum[2] + ctl->wave_sum[3]);
            ctl->ax0 = INT_MAX;  ctl->ax1 = -1;  ctl->ay0 = INT_MAX;  ctl->ay1 = -1;
        }
    } else {
        __syncthreads();
    }
    oc[tid] = 0;
    oc[tid + kOwnThreads] = 0;
    pres[tid] = 0;
    pres[tid + kOwnThreads] = 0;
    __syncthreads();
    const int ncand = ctl->ncand;
    if (ncand == 0) {                                      // nobody reaches these cells
        store_zeros();
        return;
    }

    const float *flow_b = flow + b * s2b, *filt_b = filt + b * s3b, *gout_b = gout + b * s1b;

    // 2. counts and the box of the contributing sites (flow only)
    {
        int bx0 = INT_MAX, bx1 = -1, by0 = INT_MAX, by1 = -1;
        const int nq = ncand * 64;                         // 64 quads per strip
#pragma unroll 2
        for (int idx = tid; idx < nq; idx += kOwnThreads) {
            const int t = cand[idx >> 6], q = idx & 63;
            const int x = (t & 0xfff) * 64 + 4 * (q & 15), y = ((t >> 12) & 0xffff) * 16 + ((t >> 28) & 3) * 4 + (q >> 4);
            if (x >= Wq || y >= H) continue;
            const float *fp = flow_b + (int64_t)y * s2h + x;
            const QuadHits h = own_quad_hits<FP, TH>(x, y, W, H, Wq, tx0, ty0, ld_cached4(fp), ld_cached4(fp + s2c));
            if (h.any) {
                bx0 = min(bx0, x + __ffs(h.any) - 1);
                bx1 = max(bx1, x + 31 - __clz(h.any));
                by0 = min(by0, y);
                by1 = max(by1, y);
                own_count<FP>(h, oc, pres, W, H, tx0, ty0);
            }
        }
        bx0 = wave_min_i32(bx0);  bx1 = -wave_min_i32(-bx1);  by0 = wave_min_i32(by0);  by1 = -wave_min_i32(-by1);
        if ((tid & 63) == 0 && bx1 >= 0) {
            atomicMin(&ctl->ax0, bx0);  atomicMax(&ctl->ax1, bx1);
            atomicMin(&ctl->ay0, by0);  atomicMax(&ctl->ay1, by1);
        }
    }
    __syncthreads();
    MEMC_TR_END(1);
    if (ctl->ax1 < 0) {
        store_zeros();
        return;
    }
    const int ax0 = ctl->ax0 & ~3, aw = (ctl->ax1 | 3) + 1 - ax0, ay0 = ctl->ay0, ay1 = ctl->ay1;
    const int nqw = aw >> 2;
    int rows = max(1, min(ay1 - ay0 + 1, kSlotCap / aw));
    bool counted = rows == ay1 - ay0 + 1;                  // one slab: the counts above are its counts

#pragma unroll 1
    for (int y_lo = ay0; y_lo <= ay1;) {
        // Everything a slab derives from the thread index (LDS addresses of its cells' counters, heads and offsets, the
        // staging slots) is invariant over the slab loop: hoisted in front of it, those ~30 registers sat in private scratch
        // across the replay -- stored once, reloaded after every slab (108 bytes per lane, rounds 2-5).  An opaque copy of the
        // index per slab keeps them inside the loop body: recomputed per slab, a few dozen integer instructions.
        unsigned tid_slab = tid;
        asm volatile("" : "+v"(tid_slab));
        const unsigned tid = tid_slab;                     // (hides the kernel's `tid` for the body of the loop)
        const int y_hi = min(y_lo + rows - 1, ay1);
        const int nqs = (y_hi - y_lo + 1) * nqw;           // float4 slots of this slab's site box (<= 2 per lane)
        const int i0 = (int)tid, i1 = (int)tid + kOwnThreads;
        const int r0 = i0 / nqw, q0 = i0 - r0 * nqw, r1 = i1 / nqw, q1 = i1 - r1 * nqw;
        const bool on0 = i0 < nqs, on1 = i1 < nqs;
        const int xq0 = ax0 + 4 * q0, yq0 = y_lo + r0, xq1 = ax0 + 4 * q1, yq1 = y_lo + r1;
        const float *fp0 = flow_b + (on0 ? (int64_t)yq0 * s2h + xq0 : 0);
        const float *fp1 = flow_b + (on1 ? (int64_t)yq1 * s2h + xq1 : 0);

        if (!counted) {                                    // several slabs: this slab's counts
            oc[tid] = 0;
            oc[tid + kOwnThreads] = 0;
            pres[tid] = 0;
            pres[tid + kOwnThreads] = 0;
            __syncthreads();
            const QuadHits h0 = own_quad_hits<FP, TH>(xq0, yq0, W, H, Wq, tx0, ty0, ld_cached4(fp0), ld_cached4(fp0 + s2c));
            const QuadHits h1 = own_quad_hits<FP, TH>(xq1, yq1, W, H, Wq, tx0, ty0, ld_cached4(fp1), ld_cached4(fp1 + s2c));
            if (on0 && h0.any) own_count<FP>(h0, oc, pres, W, H, tx0, ty0);
            if (on1 && h1.any) own_count<FP>(h1, oc, pres, W, H, tx0, ty0);
            __syncthreads();
        }
        counted = false;
        MEMC_TR_END(2);
        if (TR) tr_acc[7]++;

        // 3a. tails: lengths beyond the head table, their exclusive scan (CSR offsets) and their segments.  This lane's
        //     cells are tid and tid + 512 from here on.  Tails that do not fit: the slab is halved and counted again.
        const unsigned w0 = oc[tid], w1 = oc[tid + kOwnThreads];
        // heads are indexed by tap: one entry per tap index the cell receives; the counts are of the further ones (tails)
        const int tl0 = (int)(w0 >> 16), tl1 = (int)(w1 >> 16);
        const int ns0 = (tl0 + kSegLen - 1) / kSegLen, ns1 = (tl1 + kSegLen - 1) / kSegLen;
        unsigned toff0, sb0, nseg_total;
        bool serial_tails;
        {
            unsigned incl = (unsigned)(tl0 + tl1), incl2 = (unsigned)(ns0 + ns1);
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned up = __shfl_up(incl, o), up2 = __shfl_up(incl2, o);
                if ((int)(tid & 63) >= o) {
                    incl += up;
                    incl2 += up2;
                }
            }
            if ((tid & 63) == 63) {
                ctl->wave_sum[tid >> 6] = incl;
                ctl->wave_sum2[tid >> 6] = incl2;
            }
            // head table: unclaimed (become K = 0 on the zero slot when the heads are read)
            {
                float he = __uint_as_float(kHeadEmpty);   // (made per slab: the constant quad, hoisted, was kept in scratch)
                asm volatile("" : "+v"(he));
                const f32x4 z4 = {he, he, he, he};
                f32x4 *k4 = reinterpret_cast<f32x4 *>(HK);
#pragma unroll
                for (int i = 0; i < kListHead * kOwnCells / 4 / kOwnThreads; i++) k4[tid + i * kOwnThreads] = z4;
                static_assert(kListHead * kOwnCells / 4 % kOwnThreads == 0, "head table init");
                unsigned *s2 = reinterpret_cast<unsigned *>(HS);
                const unsigned zs = (unsigned)kSlotCap | ((unsigned)kSlotCap << 16);
#pragma unroll
                for (int i = 0; i < kListHead * kOwnCells / 2 / kOwnThreads; i++) s2[tid + i * kOwnThreads] = zs;
            }
            __syncthreads();
            unsigned base = 0, base2 = 0, total = 0, total2 = 0;
#pragma unroll
            for (int w = 0; w < kOwnThreads / 64; w++) {
                base += w < (int)(tid >> 6) ? ctl->wave_sum[w] : 0u;
                base2 += w < (int)(tid >> 6) ? ctl->wave_sum2[w] : 0u;
                total += ctl->wave_sum[w];
                total2 += ctl->wave_sum2[w];
            }
            if (total > (unsigned)kTailCap && rows > 1) {  // workgroup-uniform
                rows = (rows + 1) / 2;
                __syncthreads();                           // wave_sum is rewritten by the next round
                continue;
            }
            toff0 = base + incl - (unsigned)(tl0 + tl1);
            sb0 = base2 + incl2 - (unsigned)(ns0 + ns1);
            nseg_total = total2;
            // (rows == 1 always fits: a row holds at most (2 kOwnRX + 1) * 64 = 448 sites = 7168 taps of the 8192 entries --
            //  asserted in OwnGeom)
            serial_tails = total2 > (unsigned)kSegCap;     // more segments than lanes can keep: owners walk their tails
            toff[tid] = (unsigned short)toff0;
            toff[tid + kOwnThreads] = (unsigned short)(toff0 + tl0);
            oc[tid] = w0 & 0xffff0000u;                    // tail cursor 0 : count
            oc[tid + kOwnThreads] = w1 & 0xffff0000u;
        }
        __syncthreads();
        MEMC_TR_END(3);

        // 3b. the taps of the slab's sites into the lists
#pragma unroll 1
        for (int u = 0; u < 2; u++) {
            if (!(u ? on1 : on0)) continue;
            const int x = u ? xq1 : xq0, y = u ? yq1 : yq0;
            const float *fp = u ? fp1 : fp0;
            const f32x4 fx4 = ld_cached4(fp), fy4 = ld_cached4(fp + s2c);
            f32x4 tp[FP::kTaps ? 16 : 1];
            if (FP::kTaps) {
                const float *tp_p = filt_b + (int64_t)y * s3h + x;
#pragma unroll
                for (int k = 0; k < (FP::kTaps ? 16 : 1); k++) tp[k] = ld_cached4(tp_p + k * s3c);
            }
            const QuadHits h = own_quad_hits<FP, TH>(x, y, W, H, Wq, tx0, ty0, fx4, fy4);
            if (!h.any) continue;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (!((h.any >> j) & 1)) continue;
                const unsigned short slot = (unsigned short)swz_slot((y - y_lo) * aw + (x + j - ax0));
#pragma unroll
                for (int k = 0; k < FP::kN; k++) {
                    if (!((h.rowm[j] >> k) & 1)) continue;
                    const int rc = (clampi(h.iy[j] + FP::kOff + k, H - 1) - ty0) * 64 - tx0;
                    const float wb = k < FP::kN / 2 ? (1 - h.b[j]) : h.b[j];
#pragma unroll
                    for (int m = 0; m < FP::kN; m++) {
                        if (!((h.colm[j] >> m) & 1)) continue;
                        const int ci = rc + clampi(h.ix[j] + FP::kOff + m, W - 1);
                        const float wa = m < FP::kN / 2 ? (1 - h.a[j]) : h.a[j];
                        const float kv = FP::kTaps ? (wa * wb) * tp[FP::kTaps ? k * 4 + m : 0][j] : wa * wb;
                        // the head slot of this tap index, if nobody has it yet (neighbouring cells then hold
                        // neighbouring sites in the same slot: conflict-free replay); else the cell's tail
                        const int hi = (k * FP::kN + m) * kOwnCells + ci;
                        if (atomicCAS(reinterpret_cast<unsigned *>(HK) + hi, kHeadEmpty, __float_as_uint(kv)) == kHeadEmpty) {
                            HS[hi] = slot;
                        } else {
                            const int e = (int)toff[ci] + (int)(atomicAdd(oc + ci, 1u) & 0xffffu);
                            TK[e] = kv;
                            TS[e] = slot;
                        }
                    }
                }
            }
        }
        __syncthreads();
        MEMC_TR_END(4);

        // 4. heads of this lane's two cells -> registers; then the head table's bytes become the staging area, the
        //    segment sums and the segment table
        float kk0[kListHead], kk1[kListHead];
        unsigned ss0[kListHead / 2], ss1[kListHead / 2];   // two slots per register (the lists must not spill)
#pragma unroll
        for (int t = 0; t < kListHead; t++) {
            const float k0 = HK[t * kOwnCells + tid], k1 = HK[t * kOwnCells + tid + kOwnThreads];
            kk0[t] = __float_as_uint(k0) == kHeadEmpty ? 0.0f : k0;
            kk1[t] = __float_as_uint(k1) == kHeadEmpty ? 0.0f : k1;
        }
#pragma unroll
        for (int t = 0; t < kListHead; t += 2) {
            ss0[t / 2] = (unsigned)HS[t * kOwnCells + tid] | ((unsigned)HS[(t + 1) * kOwnCells + tid] << 16);
            ss1[t / 2] = (unsigned)HS[t * kOwnCells + tid + kOwnThreads] |
                         ((unsigned)HS[(t + 1) * kOwnCells + tid + kOwnThreads] << 16);
        }
        __syncthreads();
        if (tid == 0) {                                     // (the zero made HERE: as a hoisted constant quad it was spilled)
            float zero = 0.0f;
            asm volatile("" : "+v"(zero));
            g4[kSlotCap] = f32x4{zero, zero, zero, zero};
        }
        const unsigned sb1 = sb0 + (unsigned)ns0;
        if (!serial_tails) {                               // the owners describe their tails' segments: start | length << 16
            for (int i = 0; i < ns0; i++)
                segtab[sb0 + i] = (toff0 + (unsigned)(i * kSegLen)) | ((unsigned)min(kSegLen, tl0 - i * kSegLen) << 16);
            for (int i = 0; i < ns1; i++)
                segtab[sb1 + i] = (toff0 + (unsigned)(tl0 + i * kSegLen)) | ((unsigned)min(kSegLen, tl1 - i * kSegLen) << 16);
        }
        __syncthreads();
        // this lane's two segments (any cells' tails) -> registers
        float ks0[kSegLen], ks1[kSegLen];
        unsigned sg0[kSegLen / 2], sg1[kSegLen / 2];
        const bool has0 = !serial_tails && tid < nseg_total, has1 = !serial_tails && tid + kOwnThreads < nseg_total;
        {
            const unsigned d0 = has0 ? segtab[tid] : 0u, d1 = has1 ? segtab[tid + kOwnThreads] : 0u;
            const int b0 = (int)(d0 & 0xffffu), l0 = (int)(d0 >> 16), b1 = (int)(d1 & 0xffffu), l1 = (int)(d1 >> 16);
            unsigned z0[kSegLen], z1[kSegLen];
#pragma unroll
            for (int i = 0; i < kSegLen; i++) {
                const bool h0 = i < l0, h1 = i < l1;
                const float k0 = TK[h0 ? b0 + i : 0], k1 = TK[h1 ? b1 + i : 0];
                const unsigned s0 = TS[h0 ? b0 + i : 0], s1 = TS[h1 ? b1 + i : 0];
                ks0[i] = h0 ? k0 : 0.0f;
                ks1[i] = h1 ? k1 : 0.0f;
                z0[i] = h0 ? s0 : (unsigned)kSlotCap;
                z1[i] = h1 ? s1 : (unsigned)kSlotCap;
            }
#pragma unroll
            for (int i = 0; i < kSegLen / 2; i++) {        // two slots per register, each register defined in ONE assignment
                sg0[i] = z0[2 * i] | (z0[2 * i + 1] << 16);
                sg1[i] = z1[2 * i] | (z1[2 * i + 1] << 16);
            }
        }
        const bool any_seg = nseg_total != 0;              // workgroup-uniform
        MEMC_TR_END(5);

        // 5. replay, four channels at a time.  Wave-uniform plane bases + 32-bit byte offsets (per-lane 64-bit
        //    pointers would not fit next to the lists: a spilled pointer's reload waits for every load in flight)
        const CellOffsets co = cell_offsets(tid);
        const unsigned my = co.my, wo0 = co.wo0, wo1 = co.wo1;
        const bool st0 = co.st0, st1 = co.st1;
        const unsigned go0 = on0 ? 4u * (unsigned)(yq0 * s1h + xq0) : 0u, go1 = on1 ? 4u * (unsigned)(yq1 * s1h + xq1) : 0u;
        const int gs0 = on0 ? r0 * aw + 4 * q0 : 0, gs1 = on1 ? r1 * aw + 4 * q1 : 0;      // first of four slots
        f32x4 v0[4], v1[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            v0[c] = ld_cached4_u(gout_b + min(c, C - 1) * s1c, go0);
            v1[c] = ld_cached4_u(gout_b + min(c, C - 1) * s1c, go1);
        }
        const bool rmw = y_lo != ay0;                      // workgroup-uniform: a later slab adds to the first one's
        f32x4 old0 = {0.f, 0.f, 0.f, 0.f}, old1 = old0, nold0 = old0, nold1 = old0;
        if (rmw && (int)my < C) {
            old0 = ld_cached4_u(gin1_b, wo0);
            old1 = ld_cached4_u(gin1_b, wo1);
        }
        // Two barriers per chunk: [stage chunk i | finish chunk i - 1: segment sums, transpose, store] barrier
        // [issue the next loads | segments of chunk i -> LDS | heads of chunk i] barrier (the staging area is free).
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;        // head sums of the chunk before
        auto finish = [&](int cp) {                       // chunk cp's sums are complete: add the segments', store
            if (!serial_tails && any_seg) {
                f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0;
#pragma unroll 4
                for (int i = 0; i < ns0; i++) q0 += part[sb0 + i];
#pragma unroll 4
                for (int i = 0; i < ns1; i++) q1 += part[sb1 + i];
                a0 += q0;
                a1 += q1;
            }
            const f32x4 t0 = quad_transpose(a0, my), t1 = quad_transpose(a1, my);
            const bool ch_ok = cp + (int)my < C;           // ragged last chunk: this lane's channel may not exist
            if (st0 && ch_ok) *reinterpret_cast<MEMC_GLOBAL f32x4u *>(addr_u(gin1_b + cp * s1c, wo0)) = old0 + t0;
            if (st1 && ch_ok) *reinterpret_cast<MEMC_GLOBAL f32x4u *>(addr_u(gin1_b + cp * s1c, wo1)) = old1 + t1;
        };
#pragma unroll 1
        for (int c0 = 0; c0 < C; c0 += 4) {
            if (on0) {
#pragma unroll
                for (int i = 0; i < 4; i++) g4[swz_slot(gs0 + i)] = f32x4{v0[0][i], v0[1][i], v0[2][i], v0[3][i]};
            }
            if (on1) {
#pragma unroll
                for (int i = 0; i < 4; i++) g4[swz_slot(gs1 + i)] = f32x4{v1[0][i], v1[1][i], v1[2][i], v1[3][i]};
            }
            if (c0 > 0) {
                finish(c0 - 4);
                old0 = nold0;
                old1 = nold1;
            }
            __syncthreads();
            MEMC_TR_END(11);
            // the next chunk's gradoutput (and, in later slabs, the cells' current values): in flight during the replay
            // (the last iteration re-reads its own chunk -- harmless, keeps the loads unconditional)
            const int cn = c0 + 4 < C ? c0 + 4 : c0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                v0[c] = ld_cached4_u(gout_b + min(cn + c, C - 1) * s1c, go0);     // (channels past C: never stored)
                v1[c] = ld_cached4_u(gout_b + min(cn + c, C - 1) * s1c, go1);
            }
            if (rmw) {
                if (cn + (int)my < C) {
                    nold0 = ld_cached4_u(gin1_b + cn * s1c, wo0);
                    nold1 = ld_cached4_u(gin1_b + cn * s1c, wo1);
                }
            }
            // keep the slot unpacking inside the loop (hoisted, the 48 LDS addresses spill)
#pragma unroll
            for (int i = 0; i < kListHead / 2; i++) asm volatile("" : "+v"(ss0[i]), "+v"(ss1[i]));
#pragma unroll
            for (int i = 0; i < kSegLen / 2; i++) asm volatile("" : "+v"(sg0[i]), "+v"(sg1[i]));
            // ... and the coefficients single: hoisted, the {k, k} pairs of the packed FMAs double them
#pragma unroll
            for (int i = 0; i < kListHead; i++) asm volatile("" : "+v"(kk0[i]), "+v"(kk1[i]));
#pragma unroll
            for (int i = 0; i < kSegLen; i++) asm volatile("" : "+v"(ks0[i]), "+v"(ks1[i]));
            // Up to eight ds_read_b128 are issued back to back, then consumed in order (the scheduling barriers pin that:
            // left alone, the compiler -- short of registers -- waits for every read before issuing the next).
#define MEMC_REPLAY(N, ACC, KS, SS, T0)                                                                            \
            {                                                                                                      \
                f32x4 gq[N];                                                                                       \
                _Pragma("unroll") for (int i_ = 0; i_ < N; i_++)                                                   \
                    gq[i_] = g4[((T0 + i_) & 1) ? SS[(T0 + i_) / 2] >> 16 : SS[(T0 + i_) / 2] & 0xffffu];          \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
                _Pragma("unroll") for (int i_ = 0; i_ < N; i_++) ACC += KS[T0 + i_] * gq[i_];                      \
                __builtin_amdgcn_sched_barrier(0);                                                                 \
            }
            static_assert(kSegLen == 8 && (kListHead == 16 || kListHead == 4), "the replay's batches");
            if (any_seg && !serial_tails) {                // (workgroup-uniform) segment sums: for their owners
                f32x4 p0 = {0.f, 0.f, 0.f, 0.f}, p1 = p0;
                MEMC_REPLAY(8, p0, ks0, sg0, 0)
                if ((tid & ~63u) + kOwnThreads < nseg_total) MEMC_REPLAY(8, p1, ks1, sg1, 0)    // wave-uniform
                if (has0) part[tid] = p0;
                if (has1) part[tid + kOwnThreads] = p1;
            }
            MEMC_TR_END(12);
            a0 = f32x4{0.f, 0.f, 0.f, 0.f};
            a1 = a0;
            if constexpr (kListHead == 16) {
                MEMC_REPLAY(8, a0, kk0, ss0, 0)
                MEMC_REPLAY(8, a1, kk1, ss1, 0)
                MEMC_REPLAY(8, a0, kk0, ss0, 8)
                MEMC_REPLAY(8, a1, kk1, ss1, 8)
            } else {
                MEMC_REPLAY(4, a0, kk0, ss0, 0)
                MEMC_REPLAY(4, a1, kk1, ss1, 0)
            }
#undef MEMC_REPLAY
            if (serial_tails) {                            // (converging flow) owners walk their own tails
                for (int t = 0; t < tl0; t++) a0 += TK[toff0 + t] * g4[TS[toff0 + t]];
                for (int t = 0; t < tl1; t++) a1 += TK[toff0 + tl0 + t] * g4[TS[toff0 + tl0 + t]];
            }
            MEMC_TR_END(13);
            __syncthreads();
            MEMC_TR_END(15);
        }
        finish((C - 1) / 4 * 4);
        __syncthreads();                                   // the segment sums have been read: the next slab may rebuild
        MEMC_TR_END(6);
        if (TR && tid == 0) trace[(size_t)blockIdx.x * 16 + 9] = (unsigned long long)nseg_total;
        y_lo = y_hi + 1;
    }
    if (TR && tid == 0) {
        unsigned long long *t = trace + (size_t)blockIdx.x * 16;
        t[0] = __builtin_readcyclecounter() - tr_t0;
        for (int i = 1; i < 8; i++) t[i] = tr_acc[i];
        for (int i = 11; i < 16; i++) t[i] = tr_acc[i];     // inside the replay: stage + barrier, segments, heads,
                                                            // barrier, sums + transpose + store + barrier
        t[8] = (unsigned long long)ncand;
        t[10] = (unsigned long long)(aw * (ay1 - ay0 + 1));
    }
#undef MEMC_TR_BEGIN
#undef MEMC_TR_END
}

#ifdef MEMC_MEASURE
static unsigned long long *g_trace_cn = nullptr;       // gridDim.x * 16 uint64; tools/trace_kernel.py fi_bwd_cn
extern "C" int memc_debug_set_trace_buffer_cn(void *p)
{
    g_trace_cn = static_cast<unsigned long long *>(p);
    return 0;
}
#endif

// ---------------------------------------------------------------------------------------------------------
// Launcher.  Returns 1 when the call was taken, 0 when it is not for these kernels (the caller falls back to the
// direct kernel), -1 on a launch error.
// ---------------------------------------------------------------------------------------------------------
// =========================================================================================================
// The bilinear warp (Interpolation / InterpolationCh, reference kernel my_lib_kernel.cu:584-670) at many channels: the
// same three steps with a 2 x 2 window and no taps.
//   bl_bwd_flow_c4n   flow gradient: the four corner sums S = sum_c gradoutput * input1(corner) accumulated over
//                     chunks of four staged channels; gradinput2 = gam * (S_TR - S_TL) + ... (assigned); target boxes;
//   fi_bwd_image_owner<FpBilinear>   image gradient, stored;
//   bl_bwd_far_sites  the far sites' image gradient.
// =========================================================================================================
__device__ __noinline__ void bl_bwd_site_image_atomics(int x, int y, int W, int H, int C, float *gin1_b, int64_t s1c,
                                                       int s1h, const float *flow_p, int64_t s2c, const float *gout_p)
{
    const BlSite s = bl_locate<true>(x, y, W, H, flow_p[0], flow_p[s2c]);
    if (!s.valid) return;
    const int oTL = s.T * s1h + s.L, oTR = s.T * s1h + s.R, oBL = s.Bm * s1h + s.L, oBR = s.Bm * s1h + s.R;
    for (int c = 0; c < C; c++) {
        const float g = gout_p[c * s1c];
        float *q = gin1_b + c * s1c;
        atomic_add_f32(q + oTL, g * ((1 - s.a) * (1 - s.b)));
        atomic_add_f32(q + oTR, g * (s.a * (1 - s.b)));
        atomic_add_f32(q + oBL, g * ((1 - s.a) * s.b));
        atomic_add_f32(q + oBR, g * (s.a * s.b));
    }
}

// corner sums of one site over nch channels, everything from global memory (a site whose corners are not staged)
__device__ __noinline__ f32x4 bl_corner_sums_global(const float *plane0, const float *gout_site, int64_t s1c, int nch,
                                                    int oTL, int oTR, int oBL, int oBR)
{
    f32x4 q = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nch; c++) {
        const float *p = plane0 + c * s1c;
        const float gv = gout_site[c * s1c];
        q[0] += gv * p[oTL];  q[1] += gv * p[oTR];  q[2] += gv * p[oBL];  q[3] += gv * p[oBR];
    }
    return q;
}

template <int CAP, bool RAG>
__global__ __launch_bounds__(256, 2) void bl_bwd_flow_c4n(
    int W, int H, int Wq, int C, int tiles_x, int tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ gout,
    float *__restrict__ gin2, BBox *__restrict__ tbox)
{
    constexpr int LX = 16;
    using G = TileGeom<LX, CAP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    int *bb = reinterpret_cast<int *>(smem + G::kCapPx * 16);
    constexpr int kZeroPx = G::kCapPx + 4;                 // a pixel quad of zeros behind the staged image and the boxes
    if (threadIdx.x == 0) tile[kZeroPx] = f32x4{0.f, 0.f, 0.f, 0.f};     // (visible after tile_region's barrier)

    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, batch);
    const int b = tc.b, tile_x0 = tc.tx * G::kTW, tile_y0 = tc.ty * G::kTH;
    const int x = tile_x0 + 4 * (int)(threadIdx.x % LX), y = tile_y0 + (int)(threadIdx.x / LX);
    const bool inb = x < Wq && y < H;                      // (Wq: the whole quads of a row; == W unless RAG)
    const int xs = min(x, Wq - 4), ys = min(y, H - 1);
    const float *flow_p = flow + b * s2b + (int64_t)ys * s2h + xs;
    const f32x4 fx4 = ld_stream4(flow_p), fy4 = ld_stream4(flow_p + s2c);
    const float *gout_p = gout + b * s1b + (int64_t)ys * s1h + xs;

    BlSite st[4];
    unsigned far = 0;
    int cmin = INT_MAX, cmax = -1, rmin = INT_MAX, rmax = -1;          // staging box: every valid site
    int ncmin = INT_MAX, ncmax = -1, nrmin = INT_MAX, nrmax = -1;      // target box of the sites the owners take
#pragma unroll
    for (int j = 0; j < 4; j++) {
        st[j] = bl_locate<true>(x + j, y, W, H, fx4[j], fy4[j]);
        st[j].valid = st[j].valid && inb;
        if (st[j].valid) {
            cmin = min(cmin, st[j].L);  cmax = max(cmax, st[j].R);
            rmin = min(rmin, st[j].T);  rmax = max(rmax, st[j].Bm);
            if (site_far<FpBilinear>(x + j, y, st[j].L, st[j].T, W, H, Wq)) {
                far |= 1u << j;
            } else {
                ncmin = min(ncmin, st[j].L);  ncmax = max(ncmax, st[j].R);
                nrmin = min(nrmin, st[j].T);  nrmax = max(nrmax, st[j].Bm);
            }
        }
    }
    Region r = tile_region<LX, true, CAP>(cmin, cmax, rmin, rmax, tile_x0, tile_y0, bb);
    r.wimg = RAG ? W : 0;                                  // (the staged box may reach into a ragged row's last, partial quad)
    strip_box_store(tbox, ((int64_t)b * tiles_y + tc.ty) * tiles_x + tc.tx, ncmin, ncmax, nrmin, nrmax,
                    __syncthreads_or(far != 0) != 0);

    int oTL[4], oTR[4], oBL[4], oBR[4];
    unsigned valid = 0, staged = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const BlSite &s = st[j];
        const bool in_box = s.valid && r.covers(s.L, s.R, s.T, s.Bm);
        valid |= (s.valid ? 1u : 0u) << j;
        staged |= (in_box ? 1u : 0u) << j;
        // (sites outside the staged box read the zero pixel quad: 0 * Inf would be NaN)
        const int rT = in_box ? (s.T - r.y0) * r.pitch : kZeroPx, rB = in_box ? (s.Bm - r.y0) * r.pitch : kZeroPx;
        const int cL = in_box ? swz_col(s.L - r.x0) : 0, cR = in_box ? swz_col(s.R - r.x0) : 0;
        oTL[j] = rT + cL;  oTR[j] = rT + cR;  oBL[j] = rB + cL;  oBR[j] = rB + cR;
    }
    const float *in_b = in1 + b * s1b;
    f32x4 sTL = {0.f, 0.f, 0.f, 0.f}, sTR = sTL, sBL = sTL, sBR = sTL;         // corner sums, component = site
    // Software-pipelined over chunks of four channels (the forward's fi_fwd_tiled_c4n scheme): the next chunk's image
    // rows and gradoutput are on their way to registers while this one is consumed from LDS.  A ragged last chunk
    // reads the last plane again and multiplies it by zero.
    const StageSlot sl = stage_slots(r);
    StageRegs<4> sr;
    f32x4 go[4];
    auto fetch = [&](int c0) {
        const float *plane[4];
        int hs[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            plane[c] = in_b + min(c0 + c, C - 1) * s1c;
            hs[c] = s1h;
        }
        tile_stage_load_planes<4, RAG>(r, sl, plane, hs, sr);
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const f32x4 gl = ld_stream4(gout_p + min(c0 + c, C - 1) * s1c);
            go[c] = c0 + c < C ? gl : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    fetch(0);
#pragma unroll 1
    for (int c0 = 0; c0 < C; c0 += 4) {
        tile_stage_store<4, RAG>(r, sl, sr, tile);
        f32x4 gc[4];
#pragma unroll
        for (int c = 0; c < 4; c++) gc[c] = go[c];
        __syncthreads();
        fetch(c0 + 4 < C ? c0 + 4 : c0);                   // (the last iteration re-reads its own chunk: harmless)
#pragma unroll
        for (int j = 0; j < 4; j++) asm volatile("" : "+v"(oTL[j]), "+v"(oTR[j]), "+v"(oBL[j]), "+v"(oBR[j]));
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const bool on = (staged >> j) & 1;
            const f32x4 gj = on ? f32x4{gc[0][j], gc[1][j], gc[2][j], gc[3][j]} : f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x4 pTL = tile[oTL[j]], pTR = tile[oTR[j]], pBL = tile[oBL[j]], pBR = tile[oBR[j]];
            sTL[j] += gj[0] * pTL[0] + gj[1] * pTL[1] + gj[2] * pTL[2] + gj[3] * pTL[3];
            sTR[j] += gj[0] * pTR[0] + gj[1] * pTR[1] + gj[2] * pTR[2] + gj[3] * pTR[3];
            sBL[j] += gj[0] * pBL[0] + gj[1] * pBL[1] + gj[2] * pBL[2] + gj[3] * pBL[3];
            sBR[j] += gj[0] * pBR[0] + gj[1] * pBR[1] + gj[2] * pBR[2] + gj[3] * pBR[3];
        }
        if (valid & ~staged) {                             // rare: corners outside the staged box -> global gathers
#pragma unroll
            for (int j = 0; j < 4; j++) {                  // (compile-time j: a run-time index would put st[] in scratch)
                if (!(((valid & ~staged) >> j) & 1)) continue;
                const f32x4 q = bl_corner_sums_global(in_b + c0 * s1c, gout_p + c0 * s1c + j, s1c, min(4, C - c0),
                                                      st[j].T * s1h + st[j].L, st[j].T * s1h + st[j].R,
                                                      st[j].Bm * s1h + st[j].L, st[j].Bm * s1h + st[j].R);
                sTL[j] += q[0];  sTR[j] += q[1];  sBL[j] += q[2];  sBR[j] += q[3];
            }
        }
        __syncthreads();
    }
    if (inb) {
        f32x4 gx4, gy4;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const BlSite &s = st[j];
            const float x2 = (float)(x + j) + fx4[j], y2 = (float)y + fy4[j];
            const float gam_x = (float)s.Bm - y2, gam_y = (float)s.R - x2;     // clamped corners, my_lib_kernel.cu:634,652
            const float vx = gam_x * (sTR[j] - sTL[j]) + (1 - gam_x) * (sBR[j] - sBL[j]);
            const float vy = gam_y * (sBL[j] - sTL[j]) + (1 - gam_y) * (sBR[j] - sTR[j]);
            gx4[j] = s.valid ? vx : 0.0f;                  // gradinput2 is ASSIGNED; invalid sites store zeros
            gy4[j] = s.valid ? vy : 0.0f;
        }
        float *g2 = gin2 + b * s2b + (int64_t)y * s2h + x;
        st_stream4(g2, gx4);
        st_stream4(g2 + s2c, gy4);
    }
}

// (a ragged width's columns behind the whole quads, as fi_bwd_tail_sites: one wave per site, lane l the channels l, l + 64, ...)
__global__ __launch_bounds__(256) void bl_bwd_tail_sites(
    int W, int H, int Wq, int C, int batch, int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ gout,
    float *__restrict__ gin1, float *__restrict__ gin2)
{
    const int nt = W - Wq, lane = threadIdx.x & (kWave - 1);
    const int64_t i = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave, n = (int64_t)batch * H * nt;
    if (i >= n) return;                                    // (wave-uniform)
    const int x = Wq + (int)(i % nt), y = (int)((i / nt) % H), b = (int)(i / nt / H);
    const float *flow_p = flow + b * s2b + (int64_t)y * s2h + x, *gout_p = gout + b * s1b + (int64_t)y * s1h + x;
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b, *g2 = gin2 + b * s2b + (int64_t)y * s2h + x;
    const float fx = flow_p[0], fy = flow_p[s2c];
    const BlSite s = bl_locate<true>(x, y, W, H, fx, fy);
    float vx = 0.0f, vy = 0.0f;                            // gradinput2 is ASSIGNED; invalid sites store zeros
    if (s.valid) {                                         // (wave-uniform)
        const int oTL = s.T * s1h + s.L, oTR = s.T * s1h + s.R, oBL = s.Bm * s1h + s.L, oBR = s.Bm * s1h + s.R;
        float qTL = 0.0f, qTR = 0.0f, qBL = 0.0f, qBR = 0.0f;
        for (int c = lane; c < C; c += kWave) {
            const float g = gout_p[c * s1c];
            const float *p = in_b + c * s1c;
            float *q = gin1_b + c * s1c;
            qTL += g * p[oTL];  qTR += g * p[oTR];  qBL += g * p[oBL];  qBR += g * p[oBR];
            atomic_add_f32(q + oTL, g * ((1 - s.a) * (1 - s.b)));              // as bl_bwd_site_image_atomics
            atomic_add_f32(q + oTR, g * (s.a * (1 - s.b)));
            atomic_add_f32(q + oBL, g * ((1 - s.a) * s.b));
            atomic_add_f32(q + oBR, g * (s.a * s.b));
        }
        qTL = wave_allsum_f32(qTL);  qTR = wave_allsum_f32(qTR);  qBL = wave_allsum_f32(qBL);  qBR = wave_allsum_f32(qBR);
        const float x2 = (float)x + fx, y2 = (float)y + fy;
        const float gam_x = (float)s.Bm - y2, gam_y = (float)s.R - x2;         // as bl_bwd_flow_c4n
        vx = gam_x * (qTR - qTL) + (1 - gam_x) * (qBR - qBL);
        vy = gam_y * (qBL - qTL) + (1 - gam_y) * (qBR - qTR);
    }
    if (lane == 0) {
        g2[0] = vx;
        g2[s2c] = vy;
    }
}

__global__ __launch_bounds__(256) void bl_bwd_far_sites(
    int W, int H, int Wq, int C, int tiles_x, int tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h,
    const float *__restrict__ flow, const float *__restrict__ gout, float *__restrict__ gin1,
    const BBox *__restrict__ tbox)
{
    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, batch);
    if (!(tbox[(((int64_t)tc.b * tiles_y + tc.ty) * tiles_x + tc.tx) * 4].h & kTileHasFar)) return;
    const int lane = threadIdx.x & (kWave - 1);
    const int x0 = tc.tx * 64 + 4 * (int)(threadIdx.x % 16), y = tc.ty * 16 + (int)(threadIdx.x / 16);
    const bool inq = x0 < Wq && y < H;
    const float *flow_b = flow + tc.b * s2b, *gout_b = gout + tc.b * s1b;
    float *gin1_b = gin1 + tc.b * s1b;
    const float *flow_p = flow_b + (int64_t)min(y, H - 1) * s2h + min(x0, Wq - 4);
    for (int j = 0; j < 4; j++) {                          // (as fi_bwd_far_sites: a far site at a time, lanes over the channels)
        const BlSite sj = bl_locate<true>(x0 + j, y, W, H, flow_p[j], flow_p[s2c + j]);
        unsigned long long todo = __builtin_amdgcn_ballot_w64(inq && sj.valid && site_far<FpBilinear>(x0 + j, y, sj.L, sj.T, W, H, Wq));
        while (todo) {
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const int x = __builtin_amdgcn_readlane(x0, src) + j, ys = __builtin_amdgcn_readlane(y, src);
            const float *fp = flow_b + (int64_t)ys * s2h + x;
            const BlSite s = bl_locate<true>(x, ys, W, H, fp[0], fp[s2c]);
            const int oTL = s.T * s1h + s.L, oTR = s.T * s1h + s.R, oBL = s.Bm * s1h + s.L, oBR = s.Bm * s1h + s.R;
            for (int c = lane; c < C; c += kWave) {
                const float g = gout_b[c * s1c + (int64_t)ys * s1h + x];
                float *q = gin1_b + c * s1c;
                atomic_add_f32(q + oTL, g * ((1 - s.a) * (1 - s.b)));
                atomic_add_f32(q + oTR, g * (s.a * (1 - s.b)));
                atomic_add_f32(q + oBL, g * ((1 - s.a) * s.b));
                atomic_add_f32(q + oBR, g * (s.a * s.b));
            }
        }
    }
}

// gradinput1 = 0 over a strided [batch, channel, h, w] view (rows are contiguous)
__global__ __launch_bounds__(256) void fi_bwd_zero_rows(float *__restrict__ p, int w, int h, int channel, int64_t sb,
                                                        int64_t sc, int sh)
{
    const int64_t row = blockIdx.x;                        // (b * channel + c) * h + y
    const int y = (int)(row % h), c = (int)((row / h) % channel);
    float *q = p + (row / h / channel) * sb + c * sc + (int64_t)y * sh;
    for (int x = threadIdx.x; x < w; x += 256) q[x] = 0.0f;
}

// Channel counts this file is for: four and more (chunks of four staged channels, a ragged last chunk padded with zeros;
// RGB has its own kernels, one or two channels are few atomics).  For them gradinput1 is STORED on every path: when the owner kernels cannot run
// (odd geometry, no scratch inside a stream capture) the buffer is cleared here before the caller falls back to the
// accumulating direct kernel.
#ifdef MEMC_MEASURE
bool g_bwd_cn_allow_c3 = false;               // arm: the bilinear warp's RGB backward through the owner kernels (bl_cap 5)
#else
constexpr bool g_bwd_cn_allow_c3 = false;
#endif
bool fi_bwd_cn_class(int channel, int filter_size) { return filter_size == 4 && (channel >= 4 || (g_bwd_cn_allow_c3 && channel == 3)); }

int fi_bwd_cn_launch(hipStream_t stream, int w, int h, int channel, int batch,
                     int s1b, int s1c, int s1h, int s2b, int s2c, int s2h, int s3b, int s3c, int s3h,
                     const float *input1, const float *input2, const float *input3, const float *gradoutput,
                     float *gradinput1, float *gradinput2, float *gradinput3, bool force_direct)
{
    if (!fi_bwd_cn_class(channel, 4)) return 0;
    const int ntx = (w + 63) / 64, nty = (h + 15) / 16;
    const unsigned ntiles = (unsigned)ntx * nty * batch;
    const int wq = w & ~3;                                 // the whole quads of a row (round 6: ragged widths, see the top)
    CallScratch scratch;                                   // the site tiles' target boxes
    if (force_direct || !plane_fits_u32(w, h, {s1h, s2h, s3h}) || wq < 8 ||
        4LL * (3LL * s1c + (long long)(h - 1) * s1h + w) >= (1LL << 32) ||           // the owner's 4-plane offsets
        ntx > 0xfff || nty > 0x7fff ||
        !scratch.alloc((size_t)ntiles * 4 * sizeof(BBox), stream)) {                 // e.g. inside a stream capture
        hipLaunchKernelGGL(fi_bwd_zero_rows, dim3((unsigned)batch * channel * h), dim3(256), 0, stream, gradinput1, w, h,
                           channel, (int64_t)s1b, (int64_t)s1c, s1h);
        return launch_status() == 0 ? 0 : -1;
    }
    BBox *tbox = static_cast<BBox *>(scratch.p);
    const bool rag = wq < w;
    const unsigned tail_rows = (unsigned)(((int64_t)batch * channel * h + 255) / 256);
    const unsigned tail_sites = (unsigned)(((int64_t)batch * h * (w - wq) + 3) / 4);     // one wave per site, four per workgroup
    if (rag)                                               // the cells no owner stores: cleared before anything adds to them
        hipLaunchKernelGGL(bwd_cn_zero_tail, dim3(tail_rows), dim3(256), 0, stream, gradinput1, wq, w, h, channel, batch,
                           (int64_t)s1b, (int64_t)s1c, s1h);
#define MEMC_TAPS(RAG_)                                                                                            \
    hipLaunchKernelGGL(fi_bwd_taps_c4n<RAG_>, dim3(ntiles), dim3(256), tile_lds_bytes<16>() + 64, stream,          \
                       w, h, wq, channel, ntx, nty, batch, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, \
                       (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, gradoutput, gradinput1, gradinput2,  \
                       gradinput3, tbox)
    if (rag) MEMC_TAPS(true);
    else MEMC_TAPS(false);
#undef MEMC_TAPS
#define MEMC_OWNER(TH, TR, TRACE)                                                                                  \
    do {                                                                                                           \
        using Gm_ = OwnGeom<FpFilter, TH>;                                                                         \
        allow_big_lds(fi_bwd_image_owner<FpFilter, TH, TR>, Gm_::kBytes);   /* per launch: a per-DEVICE attribute */ \
        const int cty = (h + TH - 1) / TH;                                                                         \
        hipLaunchKernelGGL((fi_bwd_image_owner<FpFilter, TH, TR>), dim3((unsigned)ntx * cty * batch),              \
                           dim3(Gm_::kThreads), Gm_::kBytes, stream, w, h, wq, channel, ntx, cty, nty,             \
                           batch, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b,  \
                           (int64_t)s3c, s3h, input2, input3, gradoutput, gradinput1, tbox, TRACE);                \
    } while (0)
#ifdef MEMC_MEASURE
    if (g_trace_cn) MEMC_OWNER(16, true, g_trace_cn);
    else
#endif
        MEMC_OWNER(16, false, nullptr);
#undef MEMC_OWNER
    if (rag)
        hipLaunchKernelGGL(fi_bwd_tail_sites, dim3(tail_sites), dim3(256), 0, stream, w, h, wq, channel, batch, (int64_t)s1b,
                           (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b, (int64_t)s3c, s3h, input1, input2,
                           input3, gradoutput, gradinput1, gradinput2, gradinput3);
    hipLaunchKernelGGL(fi_bwd_far_sites, dim3(ntiles), dim3(256), 0, stream,
                       w, h, wq, channel, ntx, nty, batch, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                       (int64_t)s3b, (int64_t)s3c, s3h, input2, input3, gradoutput, gradinput1, tbox);
    return launch_status() == 0 ? 1 : -1;
}

// The bilinear warp's backward for the same class of channel counts (C >= 4); same return convention.
int bl_bwd_cn_launch(hipStream_t stream, int w, int h, int channel, int batch,
                     int s1b, int s1c, int s1h, int s2b, int s2c, int s2h,
                     const float *input1, const float *input2, const float *gradoutput,
                     float *gradinput1, float *gradinput2, bool force_direct)
{
    if (!fi_bwd_cn_class(channel, 4)) return 0;
    const int ntx = (w + 63) / 64, nty = (h + 15) / 16;
    const unsigned ntiles = (unsigned)ntx * nty * batch;
    const int wq = w & ~3;
    CallScratch scratch;                                   // the site tiles' target boxes
    if (force_direct || !plane_fits_u32(w, h, {s1h, s2h}) || wq < 8 ||
        4LL * (3LL * s1c + (long long)(h - 1) * s1h + w) >= (1LL << 32) || ntx > 0xfff || nty > 0x7fff ||
        !scratch.alloc((size_t)ntiles * 4 * sizeof(BBox), stream)) {
        hipLaunchKernelGGL(fi_bwd_zero_rows, dim3((unsigned)batch * channel * h), dim3(256), 0, stream, gradinput1, w, h,
                           channel, (int64_t)s1b, (int64_t)s1c, s1h);
        return launch_status() == 0 ? 0 : -1;
    }
    BBox *tbox = static_cast<BBox *>(scratch.p);
    const bool rag = wq < w;
    constexpr int kCap = 2496;                             // the forward's staging budget: 39 KiB, 2 x 2 footprint
    if (rag)
        hipLaunchKernelGGL(bwd_cn_zero_tail, dim3((unsigned)(((int64_t)batch * channel * h + 255) / 256)), dim3(256), 0, stream,
                           gradinput1, wq, w, h, channel, batch, (int64_t)s1b, (int64_t)s1c, s1h);
#define MEMC_FLOW(RAG_)                                                                                            \
    hipLaunchKernelGGL((bl_bwd_flow_c4n<kCap, RAG_>), dim3(ntiles), dim3(256), (tile_lds_bytes<16, kCap>() + 64), stream,  \
                       w, h, wq, channel, ntx, nty, batch, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, \
                       input1, input2, gradoutput, gradinput2, tbox)
    if (rag) MEMC_FLOW(true);
    else MEMC_FLOW(false);
#undef MEMC_FLOW
    using Gm = OwnGeom<FpBilinear, 16>;
    allow_big_lds(fi_bwd_image_owner<FpBilinear, 16, false>, Gm::kBytes);   // per launch: the attribute belongs to the current device
    hipLaunchKernelGGL((fi_bwd_image_owner<FpBilinear, 16, false>), dim3(ntiles), dim3(Gm::kThreads), Gm::kBytes, stream,
                       w, h, wq, channel, ntx, nty, nty, batch, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c,
                       s2h, (int64_t)0, (int64_t)0, 0, input2, static_cast<const float *>(nullptr), gradoutput,
                       gradinput1, tbox, static_cast<unsigned long long *>(nullptr));
    if (rag)
        hipLaunchKernelGGL(bl_bwd_tail_sites, dim3((unsigned)(((int64_t)batch * h * (w - wq) + 3) / 4)), dim3(256), 0, stream,
                           w, h, wq, channel, batch, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, input1,
                           input2, gradoutput, gradinput1, gradinput2);
    hipLaunchKernelGGL(bl_bwd_far_sites, dim3(ntiles), dim3(256), 0, stream,
                       w, h, wq, channel, ntx, nty, batch, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                       input2, gradoutput, gradinput1, tbox);
    return launch_status() == 0 ? 1 : -1;
}

}  // namespace memc
