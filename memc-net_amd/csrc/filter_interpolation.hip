// filter_interpolation.hip -- the fused adaptive warp (bilinear flow sample x per-pixel fs*fs filter),
// forward and backward, for gfx950.
//
// Replaces my_package/src/my_lib_kernel.cu:1087-1627 of the reference (kernels :1087,:1220, launchers
// :1520,:1571).  Semantics: SURVEY.md appendix A.1/A.2; float evaluation order follows the reference
// (quadrant sums accumulate row-major from 0, then the four-term blend).
//
// Design (HBM-bound gather stencil, ~1.5 flop/B, no MFMA):
//   * one lane = one output site; a wavefront = 64 consecutive sites of one image row, so every tap-plane,
//     flow and output access of a wave is one fully coalesced 256-B segment;
//   * the fs*fs taps and the two flow components of a site are read ONCE into registers and reused for
//     every channel (the reference re-reads all 16 taps per channel, my_lib_kernel.cu:1149-1150);
//   * taps / flow / output are single-use streams -> non-temporal, so L1/L2 keep the source image, whose
//     4x4 windows overlap between neighbouring lanes and rows;
//   * the workgroup->tile map is XCD-chunked (memc_common.hpp) so halo rows are shared inside one L2;
//   * 64-bit batch/channel offsets (4K x batch 8 x 64 channels exceeds 2^31 elements), 32-bit in-plane.
#include "memc_common.hpp"
#include "memc_internal.h"
#include "memc_tile.hpp"
#include "memc_fi.hpp"

namespace memc {

// --------------------------------------------------------------------------------------------------
// Forward, fs == 4, LDS-tiled and fully vectorised -- the production kernel.
//
//   workgroup = 256 lanes = a (4*LX) x (256/LX) tile of output sites; one lane = 4 consecutive sites of a row.
//   1. flow (2 x dwordx4) and the 16 tap planes (16 x dwordx4) of the lane's sites are requested up front
//      (non-temporal: single-use streams);
//   2. the tile's source bounding box under its own flow is reduced across the workgroup (memc_tile.hpp) and
//      staged into LDS as pixel quads, 4 channels at a time (3 real + 1 zero for RGB);
//   3. per site: 16 ds_read_b128 (all channels of a tap at once), quadrant sums in the reference's order, blend;
//      sites outside the staged box gather from global instead; invalid sites copy the input pixel;
//   4. one dwordx4 store per channel (non-temporal).
//   C > 4 loops steps 2-4 over chunks of four channels with the taps and site geometry kept in registers.
// --------------------------------------------------------------------------------------------------
// Scalar evaluation of ONE site for channels [0, nch) of `plane0`, everything read from global memory
// (flow, taps, image): the rare path for sites whose source window is not in the staged LDS region, and the
// body of the any-filter-size kernel.  Same arithmetic order as the fast path.

__device__ __noinline__ void fi_site_scalar(int x, int y, int W, int H, int nch, int fs,
                                            const float *plane0, int64_t s1c, int s1h,
                                            const float *flow_p, int64_t s2c, const float *tap_p, int64_t s3c,
                                            float *out_p)
{
    const float fx = flow_p[0], fy = flow_p[s2c];
    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    if (s.valid) {
        const int L = s.ix + 1 - fs / 2, T = s.iy + 1 - fs / 2, R = L + fs, Bm = T + fs;
        for (int c = 0; c < nch; c++) {
            const float *p = plane0 + c * s1c;
            const float TL = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, T, s.iy, L, s.ix);
            const float TR = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, T, s.iy, s.ix + 1, R - 1);
            const float BL = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, s.iy + 1, Bm - 1, L, s.ix);
            const float BR = fi_quad_sum(p, s1h, W, H, tap_p, s3c, fs, L, T, s.iy + 1, Bm - 1, s.ix + 1, R - 1);
            out_p[c * s1c] = (1 - s.a) * (1 - s.b) * TL + s.a * (1 - s.b) * TR +
                             (1 - s.a) * s.b * BL + s.a * s.b * BR;
        }
    } else {
        const float *p = plane0 + (int64_t)y * s1h + x;
        for (int c = 0; c < nch; c++) out_p[c * s1c] = p[c * s1c];
    }
}

// one chunk of NCH (1..4) channels: stage -> gather -> store.  Everything indexed by channel or site is
// compile-time unrolled (run-time indexed vectors would live in scratch) and the hot path is branch-free:
// a site whose window is not staged still issues its 16 LDS reads (at pixel 0, harmless) and is redone
// afterwards by fi_site_scalar (at the end of the kernel, once for all channels).
// Gather + blend of the sites selected by `sel` (bit j) from the staged band; other sites keep their `res`.
// Branch-free: unselected sites still issue their 16 LDS reads (at pixel 0, harmless).
template <int LX, int NCH>
__device__ __forceinline__ void fi_gather(const Region &r, const FiSite4 &g, const f32x4 (&tp)[16], unsigned sel,
                                          int W, int H, const f32x4 *tile, f32x4 (&res)[4])
{
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const bool on = (sel >> j) & 1;
        int ro[4], co[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ro[k] = on ? (clampi(g.iy[j] - 1 + k, H - 1) - r.y0) * r.pitch : 0;
            co[k] = on ? swz_col(clampi(g.ix[j] - 1 + k, W - 1) - r.x0) : 0;
        }
        // quadrant sums, row-major inside each quadrant as in the reference (rows 0,1 top; 2,3 bottom)
        f32x4 TL = {0.f, 0.f, 0.f, 0.f}, TR = TL, BL = TL, BR = TL;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            f32x4 v[4];
#pragma unroll
            for (int m = 0; m < 4; m++) v[m] = tile[ro[k] + co[m]];
            if (k < 2) {
                TL += v[0] * tp[k * 4 + 0][j];  TL += v[1] * tp[k * 4 + 1][j];
                TR += v[2] * tp[k * 4 + 2][j];  TR += v[3] * tp[k * 4 + 3][j];
            } else {
                BL += v[0] * tp[k * 4 + 0][j];  BL += v[1] * tp[k * 4 + 1][j];
                BR += v[2] * tp[k * 4 + 2][j];  BR += v[3] * tp[k * 4 + 3][j];
            }
        }
        const float a = g.a[j], bt = g.b[j];
        const f32x4 val = ((1 - a) * (1 - bt)) * TL + (a * (1 - bt)) * TR + ((1 - a) * bt) * BL + (a * bt) * BR;
        res[j] = on ? val : res[j];
    }
}

template <int LX, int NCH>
__device__ __forceinline__ void fi_gather_store(
    const Region &r, const FiSite4 &g, const f32x4 (&tp)[16], bool inb, int x, int y, int W, int H,
    const float *__restrict__ plane0, float *__restrict__ out_p, int64_t s1c, int s1h, const f32x4 *tile)
{
    if (!inb) return;
    f32x4 res[4];                                          // res[j][c]: site j, channel c
#pragma unroll
    for (int j = 0; j < 4; j++) {
        int ro[4], co[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ro[k] = clampi(g.iy[j] - 1 + k, H - 1);
            co[k] = clampi(g.ix[j] - 1 + k, W - 1);
        }
        const bool valid = (g.valid >> j) & 1;
        const bool staged = valid && r.covers(co[0], co[3], ro[0], ro[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ro[k] = staged ? (ro[k] - r.y0) * r.pitch : 0;
            co[k] = staged ? swz_col(co[k] - r.x0) : 0;
        }
        // quadrant sums, row-major inside each quadrant as in the reference (rows 0,1 top; 2,3 bottom)
        f32x4 TL = {0.f, 0.f, 0.f, 0.f}, TR = TL, BL = TL, BR = TL;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            f32x4 v[4];
#pragma unroll
            for (int m = 0; m < 4; m++) v[m] = tile[ro[k] + co[m]];
            if (k < 2) {
                TL += v[0] * tp[k * 4 + 0][j];  TL += v[1] * tp[k * 4 + 1][j];
                TR += v[2] * tp[k * 4 + 2][j];  TR += v[3] * tp[k * 4 + 3][j];
            } else {
                BL += v[0] * tp[k * 4 + 0][j];  BL += v[1] * tp[k * 4 + 1][j];
                BR += v[2] * tp[k * 4 + 2][j];  BR += v[3] * tp[k * 4 + 3][j];
            }
        }
        const float a = g.a[j], bt = g.b[j];
        res[j] = ((1 - a) * (1 - bt)) * TL + (a * (1 - bt)) * TR + ((1 - a) * bt) * BL + (a * bt) * BR;
    }
    if (g.valid != 0xFu) {                                 // out-of-range sites copy the input pixel
#pragma unroll
        for (int c = 0; c < NCH; c++) {
            const f32x4 own = ld_cached4(plane0 + c * s1c + (int64_t)y * s1h + x);
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (!((g.valid >> j) & 1)) res[j][c] = own[j];
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; c++)
        st_stream4(out_p + c * s1c, f32x4{res[0][c], res[1][c], res[2][c], res[3][c]});
}

template <int LX, int NCH, bool RAGW = false>
__device__ __forceinline__ void fi_fwd_chunk(
    const Region &r, const FiSite4 &g, const f32x4 (&tp)[16], bool inb, int x, int y, int W, int H,
    const float *__restrict__ plane0, float *__restrict__ out_p, int64_t s1c, int s1h, f32x4 *tile)
{
    tile_stage<LX, NCH, 256, RAGW>(r, plane0, s1c, s1h, tile);
    __syncthreads();
    fi_gather_store<LX, NCH>(r, g, tp, inb, x, y, W, H, plane0, out_p, s1c, s1h, tile);
}

// --------------------------------------------------------------------------------------------------
// Forward, fs == 4, channel count a multiple of 4 (the 64-channel context warp, MEMC_Net_star.py:281-285):
// software-pipelined over chunks of four channels.  The chunk loop of fi_fwd_tiled_fs4<.., 0, ..> exposes one
// L2/HBM round trip per chunk (stage -> barrier -> gather -> barrier); here the NEXT chunk's rows are already
// on their way to registers while the current chunk is gathered from LDS, so the round trip hides behind
// ~1 us of FMA work per chunk (at C = 64 the operator is about as VALU-bound as it is HBM-bound).
// --------------------------------------------------------------------------------------------------
// NT = 512 (measurement arms 32 / 35): 64 x 32 tiles, one workgroup of 512 lanes per CU instead of two of 256 -- the same
// waves per CU, and the staged box covers 1.5x its sites instead of 1.8x (PMC: the 64 x 16 tiles read the image 2.0 times,
// nothing is shared between neighbouring tiles through the L2s: they drift apart in their chunk loops).  Measured
// 1288 us (in stripes of four: 1243) against 1133 us (8 x 64 x 720 x 1280) with a 6144-cell budget (round 2 gave this arm
// 3584 cells, most tiles swept their box in two bands: 1698 us).  Two LDS buffers and ONE barrier per chunk change nothing
// (1289 us; round 3, removed): it is not the barriers of the eight-wave workgroup, and not the box traffic, that bind.
// RAGGED: any channel count >= 4 -- the last chunk re-reads the last plane for the channels it does not have and does not
// store them (a separate instantiation: the C % 4 == 0 kernel, at 239 registers, is left exactly as it was).
// RAGW: a ragged WIDTH (W % 4 != 0, round 5; instantiated together with RAGGED only) -- the whole quads of every row here, the
// image's true width in every clamp and staged box (memc_tile.hpp), the columns behind them on fi_fwd_direct_fs4 (launcher);
// one workgroup per CU (at two, the rotation of a row's last quad spills 32 B per lane).
template <int SW, int NT = 256, bool RAGGED = false, int LX = 16, int ABL = 0, bool RAGW = false>   // SW 0: one tile column per XCD strip; 2 / 4: stripes
__global__ __launch_bounds__(NT, (NT == 256 && !RAGW) ? 2 : 1) void fi_fwd_tiled_c4n(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    float *__restrict__ out)
{
    constexpr int CAP = NT == 256 ? 3072 : 6144;
    using G = TileGeom<LX, CAP, NT>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    int *bb = reinterpret_cast<int *>(smem + G::kCapPx * 16);

    // With 64 channels the image is 88 % of the bytes read, and a tile stages its box dilated by the motion: what
    // the neighbouring tiles re-read must come out of THIS XCD's L2.  Stripes keep horizontal neighbours on one
    // XCD and in flight together (the grid then covers ceil(tiles_x / SW) * SW virtual columns).
    const TileCoord tc = SW ? stripe_walk<(SW ? SW : 1)>(blockIdx.x, gridDim.x, tiles_x, tiles_y)
                            : strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, gridDim.x / (tiles_x * tiles_y));
    if (SW && tc.tx >= tiles_x) return;
    const int b = tc.b, tile_x0 = tc.tx * G::kTW, tile_y0 = tc.ty * G::kTH;
    const int x = tile_x0 + 4 * (threadIdx.x % LX), y = tile_y0 + threadIdx.x / LX;
    const int Ws = RAGW ? W & ~3 : W;
    const bool inb = x < Ws && y < H;
    const int xs = min(x, Ws - 4), ys = min(y, H - 1);
    const float *flow_p = flow + b * s2b + (int64_t)ys * s2h + xs;
    const float *tap_p = filt + b * s3b + (int64_t)ys * s3h + xs;
    const f32x4 fx4 = ld_stream4(flow_p), fy4 = ld_stream4(flow_p + s2c);
    f32x4 tp[16];
#pragma unroll
    for (int k = 0; k < 16; k++) tp[k] = ld_stream4(tap_p + k * s3c);

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
    const BBox box = tile_bbox<LX, NT>(cmin, cmax, rmin, rmax, bb);
    const Bands bands = make_bands<LX, true, CAP>(box);
    const float *in_b = in1 + b * s1b;
    float *out_p = out + b * s1b + (int64_t)y * s1h + x;
    unsigned done = 0;
#pragma unroll 1
    for (int bi = 0; bi < bands.n; bi++) {
        const Region r = band_region(box, bands, bi, RAGW ? W : 0);
        const unsigned sel = inb ? fi_covered(r, g, W, H) & ~done : 0u;
        // later bands only run when somebody still needs them; the vote is also the barrier that frees the LDS
        if (bi > 0 && !__syncthreads_or(sel != 0)) continue;
        done |= sel;
        // band 0 also writes the out-of-range sites (they copy the input pixel)
        const unsigned wr = sel | (bi == 0 && inb ? ~g.valid & 0xFu : 0u);
        const StageSlot sl = stage_slots<NT>(r);
        StageRegs<4> sr;
        auto stage_load = [&](int cb) {
            if (!RAGGED) {
                tile_stage_load<4>(r, sl, in_b + cb * s1c, s1c, s1h, sr);
            } else {                                       // planes past the last one: the last one again
                const float *plane[4];
                int hs[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    plane[c] = in_b + min(cb + c, C - 1) * s1c;
                    hs[c] = s1h;
                }
                tile_stage_load_planes<4, RAGW>(r, sl, plane, hs, sr);
            }
        };
        stage_load(0);
#pragma unroll 1
        for (int c0 = 0; c0 < C; c0 += 4) {
            tile_stage_store<4, RAGW>(r, sl, sr, tile);
            __syncthreads();
            // next chunk's rows: in flight while this chunk is gathered (the last iteration re-reads its own
            // chunk -- harmless, keeps the loads unconditional)
            const int cn = c0 + 4 < C ? c0 + 4 : c0;
            if (ABL != 2) stage_load(cn);
            // keep the loop-invariant tap splats / LDS addresses inside the loop (see fi_fwd_tiled_fs4)
#pragma unroll
            for (int k = 0; k < 16; k++)
                asm volatile("" : "+v"(tp[k][0]), "+v"(tp[k][1]), "+v"(tp[k][2]), "+v"(tp[k][3]));
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(g.ix[j]), "+v"(g.iy[j]), "+v"(g.a[j]), "+v"(g.b[j]));
            f32x4 res[4];
#pragma unroll
            for (int j = 0; j < 4; j++) res[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ABL == 1) {                                // timing arm: one LDS read per site instead of sixteen
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    float t = 0.f;
#pragma unroll
                    for (int k = 0; k < 16; k++) t += tp[k][j];
                    res[j] = tile[(threadIdx.x * 4 + j) & 2047] * t;
                }
            } else
            fi_gather<LX, 4>(r, g, tp, sel, W, H, tile, res);
            const float *plane0 = in_b + c0 * s1c;
            float *o = out_p + c0 * s1c;
            if (wr & ~g.valid) {                           // out-of-range sites copy the input pixel
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    if (RAGGED && c0 + c >= C) continue;
                    const f32x4 own = ld_cached4(plane0 + c * s1c + (int64_t)y * s1h + x);
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (!((g.valid >> j) & 1)) res[j][c] = own[j];
                }
            }
            if (wr == 0xFu) {
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (!RAGGED || c0 + c < C) st_stream4(o + c * s1c, f32x4{res[0][c], res[1][c], res[2][c], res[3][c]});
            } else if (wr) {                               // a lane whose sites are split over bands
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if ((wr >> j) & 1) {
#pragma unroll
                        for (int c = 0; c < 4; c++)
                            if (!RAGGED || c0 + c < C) o[c * s1c + j] = res[j][c];
                    }
            }
            __syncthreads();
        }
    }
    unsigned slow = inb ? g.valid & ~done : 0u;            // not coverable within kMaxBands bands
    while (slow) {
        const int j = __ffs(slow) - 1;
        slow &= slow - 1;
        fi_site_scalar(x + j, y, W, H, C, 4, in_b, s1c, s1h, flow_p + j, s2c, tap_p + j, s3c, out_p + j);
    }
}

// CT == 3: RGB, one chunk; CT == 0: any channel count, chunks of four.  MINW = waves per SIMD the register
// allocator must leave room for (3 <-> 168 VGPRs, matching the 3 workgroups per CU the 48 KiB of LDS admit).
// WALK: how workgroups map to tiles -- 0 = column strips dealt to the XCDs (the product); the others exist in the
// measurement build only (same results): 1 = hardware order, 4 = row-major chunk per XCD, 5 / 6 = stripes 2 / 4 tile
// columns wide (PMC: profiles/r03_pmc_headline_stripes.txt -- 7.6 % less traffic, 4.6 % slower).
// (Round 1-2 also timed, and dropped: an L2 prefetch of a future tile's flow 32 / 64 / 96 positions ahead (+6 %: the XCD's
// L2 turns over in ~6 us, the lines are gone before use); the second half of the tap planes requested behind the staging
// loads (+-0.3 %); the kernel without its LDS gathers (526 us) / without its staging loads: DESIGN.md section 4.)
// RAGW: a ragged width (W % 4 != 0, round 5) -- this kernel serves the whole quads, sites x < W & ~3, with the image's true
// width in every clamp, validity test and staged box (whose last quad is loaded ragged-safely: memc_tile.hpp); the one to
// three columns behind them go to fi_fwd_direct_fs4 (launcher).
// CAP: the LDS budget in pixel quads (measurement arms only: 128 x 8 tiles need 4608 for their 137 x 21 boxes; RGB path only).
// PHASE (measurement arm, RGB path): hold the results until the chip-wide write window of the 100 MHz clock opens (g_fi_phase).
#ifdef MEMC_MEASURE
__device__ unsigned g_fi_phase[2] = {1000u, 120u};         // period, window (ticks of 10 ns)
#endif
template <int LX, int CT, int MINW, int WALK, bool RAGW = false, int CAP = 3072, int PHASE = 0>
__global__ __launch_bounds__(256, MINW) void fi_fwd_tiled_fs4(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    float *__restrict__ out)
{
    using G = TileGeom<LX, CAP>;
    constexpr int ITS = (CAP + 1023) / 1024;               // staging slots per lane (3 for the product's 3072)
    static_assert(CAP == 3072 || CT == 3, "a larger budget: RGB path only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    int *bb = reinterpret_cast<int *>(smem + G::kCapPx * 16);

    int tx, ty, b;
    if (WALK == 4) {                                       // row-major chunks per XCD (measurement arm)
        const unsigned t = xcd_chunked_id(blockIdx.x, gridDim.x);
        tx = t % tiles_x;  ty = (t / tiles_x) % tiles_y;  b = t / (tiles_x * tiles_y);
    } else if (WALK == 1) {                                // hardware order (measurement arm)
        const unsigned t = blockIdx.x;
        tx = t % tiles_x;  ty = (t / tiles_x) % tiles_y;  b = t / (tiles_x * tiles_y);
    } else if (WALK == 5 || WALK == 6) {                   // stripes 2 / 4 tile columns wide
        const TileCoord tc = WALK == 5 ? stripe_walk<2>(blockIdx.x, gridDim.x, tiles_x, tiles_y)
                                      : stripe_walk<4>(blockIdx.x, gridDim.x, tiles_x, tiles_y);
        tx = tc.tx;  ty = tc.ty;  b = tc.b;
        if (tx >= tiles_x) return;
    } else {
        const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, gridDim.x / (tiles_x * tiles_y));
        tx = tc.tx;  ty = tc.ty;  b = tc.b;
    }
    const int tile_x0 = tx * G::kTW, tile_y0 = ty * G::kTH;
    const int x = tile_x0 + 4 * (threadIdx.x % LX);
    const int y = tile_y0 + threadIdx.x / LX;
    const int Ws = RAGW ? W & ~3 : W;
    const bool inb = x < Ws && y < H;         // Ws % 4 == 0: a lane's four sites are in or out together

    // 1. streams.  Loads are UNCONDITIONAL (lanes past the image edge read a clamped, in-range address and are
    // masked at the store): a load under `if` or `?:` makes its result a phi, and the compiler then waits
    // for it (s_waitcnt vmcnt(0)) at the join instead of at its first use, serialising every phase.
    const int xs = min(x, Ws - 4), ys = min(y, H - 1);
    const float *flow_p = flow + b * s2b + (int64_t)ys * s2h + xs;
    const float *tap_p = filt + b * s3b + (int64_t)ys * s3h + xs;
    const f32x4 fx4 = ld_stream4(flow_p);
    const f32x4 fy4 = ld_stream4(flow_p + s2c);
    f32x4 tp[16];
#pragma unroll
    for (int k = 0; k < 16; k++) tp[k] = ld_stream4(tap_p + k * s3c);

    // 2. site geometry and this lane's source box
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
    // 3. source box, swept in bands when it does not fit the LDS budget (memc_tile.hpp)
    const BBox box = tile_bbox<LX>(cmin, cmax, rmin, rmax, bb);
    const Bands bands = make_bands<LX, true, CAP>(box);
    const Region r = band_region(box, bands, 0, RAGW ? W : 0);
    unsigned slow = inb ? g.valid & ~fi_covered(r, g, W, H) : 0u;   // sites outside the first band

    // 3./4. channels, four at a time
    const float *in_b = in1 + b * s1b;
    float *out_p = out + b * s1b + (int64_t)y * s1h + x;
    if (CT == 3) {
        // production RGB path: band loop, results kept in registers until every band has run
        f32x4 res[4];
#pragma unroll
        for (int j = 0; j < 4; j++) res[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned done = 0;
#pragma unroll 1
        for (int bi = 0; bi < bands.n; bi++) {
            const Region rb = band_region(box, bands, bi, RAGW ? W : 0);
            const unsigned sel = inb ? fi_covered(rb, g, W, H) & ~done : 0u;
            if (bi > 0 && !__syncthreads_or(sel != 0)) continue;
            done |= sel;
            tile_stage<LX, 3, 256, RAGW, ITS>(rb, in_b, s1c, s1h, tile);
            __syncthreads();
            // keep tap splats / blend weights inside the loop (hoisted, they spill: see the chunk loop below)
#pragma unroll
            for (int k = 0; k < 16; k++)
                asm volatile("" : "+v"(tp[k][0]), "+v"(tp[k][1]), "+v"(tp[k][2]), "+v"(tp[k][3]));
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(g.ix[j]), "+v"(g.iy[j]), "+v"(g.a[j]), "+v"(g.b[j]));
            fi_gather<LX, 3>(rb, g, tp, sel, W, H, tile, res);
        }
        slow = inb ? g.valid & ~done : 0u;
#ifdef MEMC_MEASURE
        if (PHASE) {                                       // results complete: stores wait for the chip-wide write window
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(res[j]));
            const unsigned per = g_fi_phase[0], win = g_fi_phase[1];
            while ((unsigned)__builtin_amdgcn_s_memrealtime() % per < per - win) __builtin_amdgcn_s_sleep(8);
        }
#endif
        if (inb) {
            if (g.valid != 0xFu) {                         // out-of-range sites copy the input pixel
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    const f32x4 own = ld_cached4(in_b + c * s1c + (int64_t)y * s1h + x);
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (!((g.valid >> j) & 1)) res[j][c] = own[j];
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++)
                st_stream4(out_p + c * s1c, f32x4{res[0][c], res[1][c], res[2][c], res[3][c]});
        }
    } else {
        int c0 = 0;
#pragma unroll 1
        for (; c0 + 4 <= C; c0 += 4) {
            if (c0 > 0) __syncthreads();                   // the previous chunk's gathers are done
            // Everything the chunk body derives from the taps and the site geometry (tap splats for the
            // packed FMAs, 64 LDS addresses, blend weights) is loop-invariant; hoisted out of the loop it
            // needs ~400 more registers than exist and lands in scratch (1.5 KB per lane).  Laundering the
            // inputs through empty asm statements once per iteration keeps that arithmetic in the loop.
#pragma unroll
            for (int k = 0; k < 16; k++)
                asm volatile("" : "+v"(tp[k][0]), "+v"(tp[k][1]), "+v"(tp[k][2]), "+v"(tp[k][3]));
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(g.ix[j]), "+v"(g.iy[j]), "+v"(g.a[j]), "+v"(g.b[j]));
            fi_fwd_chunk<LX, 4, RAGW>(r, g, tp, inb, x, y, W, H, in_b + c0 * s1c, out_p + c0 * s1c, s1c, s1h, tile);
        }
        if (c0 < C) {                                      // tail of 1..3 channels
            if (c0 > 0) __syncthreads();
            const int nch = C - c0;
            const float *plane0 = in_b + c0 * s1c;
            float *o = out_p + c0 * s1c;
#pragma unroll
            for (int k = 0; k < 16; k++)                   // as above: keep the tail's arithmetic in the tail
                asm volatile("" : "+v"(tp[k][0]), "+v"(tp[k][1]), "+v"(tp[k][2]), "+v"(tp[k][3]));
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(g.ix[j]), "+v"(g.iy[j]), "+v"(g.a[j]), "+v"(g.b[j]));
            if (nch == 3)      fi_fwd_chunk<LX, 3, RAGW>(r, g, tp, inb, x, y, W, H, plane0, o, s1c, s1h, tile);
            else if (nch == 2) fi_fwd_chunk<LX, 2, RAGW>(r, g, tp, inb, x, y, W, H, plane0, o, s1c, s1h, tile);
            else               fi_fwd_chunk<LX, 1, RAGW>(r, g, tp, inb, x, y, W, H, plane0, o, s1c, s1h, tile);
        }
    }
    while (slow) {                            // rare: redo those sites from global memory, all channels
        const int j = __ffs(slow) - 1;
        slow &= slow - 1;
        fi_site_scalar(x + j, y, W, H, C, 4, in_b, s1c, s1h, flow_p + j, s2c, tap_p + j, s3c, out_p + j);
    }
}

// --------------------------------------------------------------------------------------------------
// EXTENSION (no reference counterpart): the two adaptive warps of a frame pair and
// their occlusion-weighted blend in one pass,
//     out = occ0 * FI(in0, flow0, filt0) + occ1 * FI(in2, flow1, filt1)          (MEMC_Net_star.py:266-277)
// RGB, fs == 4.  Per site this moves 2 * (12 + 8 + 64) + 8 + 12 = 188 B instead of the 2 * 96 + 44 = 236 B of two
// warps and a separate blend: the two warped frames never exist in memory.  Same tile machinery as
// fi_fwd_tiled_fs4: both directions' streams are requested up front, then each direction runs its own
// box -> stage -> gather round on the same LDS bytes.
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 fi_site_vals3(int x, int y, int W, int H, const float *plane0, int64_t s1c,
                                               int s1h, float fx, float fy, const float *tap_p, int64_t s3c)
{
    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (!s.valid) return v;                                // (callers only pass valid sites)
    const int L = s.ix - 1, T = s.iy - 1, R = L + 4, Bm = T + 4;
#pragma unroll 1
    for (int c = 0; c < 3; c++) {
        const float *p = plane0 + c * s1c;
        const float TL = fi_quad_sum(p, s1h, W, H, tap_p, s3c, 4, L, T, T, s.iy, L, s.ix);
        const float TR = fi_quad_sum(p, s1h, W, H, tap_p, s3c, 4, L, T, T, s.iy, s.ix + 1, R - 1);
        const float BL = fi_quad_sum(p, s1h, W, H, tap_p, s3c, 4, L, T, s.iy + 1, Bm - 1, L, s.ix);
        const float BR = fi_quad_sum(p, s1h, W, H, tap_p, s3c, 4, L, T, s.iy + 1, Bm - 1, s.ix + 1, R - 1);
        v[c] = (1 - s.a) * (1 - s.b) * TL + s.a * (1 - s.b) * TR + (1 - s.a) * s.b * BL + s.a * s.b * BR;
    }
    return v;
}

__global__ __launch_bounds__(256, 2) void fi_fwd_blend_c3(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    int64_t sob, int soh,
    const float *__restrict__ in0, const float *__restrict__ in2, const float *__restrict__ flow0,
    const float *__restrict__ flow1, const float *__restrict__ filt0, const float *__restrict__ filt1,
    const float *__restrict__ occ0, const float *__restrict__ occ1, float *__restrict__ out)
{
    constexpr int LX = 16;
    using G = TileGeom<LX>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    int *bb = reinterpret_cast<int *>(smem + G::kCapPx * 16);

    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, gridDim.x / (tiles_x * tiles_y));
    const int b = tc.b;
    const unsigned tid = tid_now();
    const int x = tc.tx * G::kTW + 4 * (int)(tid % LX), y = tc.ty * G::kTH + (int)(tid / LX);
    const bool inb = x < W && y < H;
    const int xs = min(x, W - 4), ys = min(y, H - 1);
    const unsigned o2 = 4u * (unsigned)(ys * s2h + xs), o3 = 4u * (unsigned)(ys * s3h + xs),
                   oo = 4u * (unsigned)(ys * soh + xs);
    // (pinned in SGPRs here: after the divergent exits below the compiler refuses to move it there)
    const uintptr_t out_u = pin_sgpr(out + b * s1b);
    // all streams of both directions first: 2 * 18 + 2 float4 per lane in flight
    f32x4 fx[2], fy[2], oc[2], tp0[16], tp1[16];
    fx[0] = ld_stream4_u(flow0 + b * s2b, o2);  fy[0] = ld_stream4_u(flow0 + b * s2b + s2c, o2);
    fx[1] = ld_stream4_u(flow1 + b * s2b, o2);  fy[1] = ld_stream4_u(flow1 + b * s2b + s2c, o2);
#pragma unroll
    for (int k = 0; k < 16; k++) tp0[k] = ld_stream4_u(filt0 + b * s3b + k * s3c, o3);
#pragma unroll
    for (int k = 0; k < 16; k++) tp1[k] = ld_stream4_u(filt1 + b * s3b + k * s3c, o3);
    oc[0] = ld_stream4_u(occ0 + b * sob, oo);
    oc[1] = ld_stream4_u(occ1 + b * sob, oo);

    // one direction: box -> (bands of) stage -> gather; returns the warped RGB of the lane's four sites
    auto warp = [&](const float *in_b, const float *flow_b, const float *filt_b, const f32x4 &fx4, const f32x4 &fy4,
                    f32x4 (&tp)[16], f32x4 (&res)[4]) {
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
        const Bands bands = make_bands<LX>(box);
#pragma unroll
        for (int j = 0; j < 4; j++) res[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        unsigned done = 0;
#pragma unroll 1
        for (int bi = 0; bi < bands.n; bi++) {
            const Region rb = band_region(box, bands, bi);
            const unsigned sel = inb ? fi_covered(rb, g, W, H) & ~done : 0u;
            if (bi > 0 && !__syncthreads_or(sel != 0)) continue;
            done |= sel;
            tile_stage<LX, 3>(rb, in_b, s1c, s1h, tile);
            __syncthreads();
            // keep tap splats / blend weights inside the loop (hoisted, they spill)
#pragma unroll
            for (int k = 0; k < 16; k++)
                asm volatile("" : "+v"(tp[k][0]), "+v"(tp[k][1]), "+v"(tp[k][2]), "+v"(tp[k][3]));
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(g.ix[j]), "+v"(g.iy[j]), "+v"(g.a[j]), "+v"(g.b[j]));
            fi_gather<LX, 3>(rb, g, tp, sel, W, H, tile, res);
        }
        if (!inb) return;
        unsigned slow = g.valid & ~done;                   // rare: not coverable within kMaxBands bands
        while (slow) {
            const int j = __ffs(slow) - 1;
            slow &= slow - 1;
            const f32x4 v = fi_site_vals3(x + j, y, W, H, in_b, s1c, s1h, flow_b[(int64_t)y * s2h + x + j],
                                          flow_b[s2c + (int64_t)y * s2h + x + j],
                                          filt_b + (int64_t)y * s3h + x + j, s3c);
#pragma unroll
            for (int jj = 0; jj < 4; jj++) res[jj] = jj == j ? v : res[jj];
        }
        if (g.valid != 0xFu) {                             // out-of-range sites copy the input pixel
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const f32x4 own = ld_cached4(in_b + c * s1c + (int64_t)y * s1h + x);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (!((g.valid >> j) & 1)) res[j][c] = own[j];
            }
        }
    };

    f32x4 w0[4], w2[4];
    warp(in0 + b * s1b, flow0 + b * s2b, filt0 + b * s3b, fx[0], fy[0], tp0, w0);
    __syncthreads();                                       // direction 0's gathers are done: the LDS is free again
    warp(in2 + b * s1b, flow1 + b * s2b, filt1 + b * s3b, fx[1], fy[1], tp1, w2);
    if (!inb) return;
    const unsigned o1 = 4u * (unsigned)(y * s1h + x);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float p0 = oc[0][j] * w0[j][c], p2 = oc[1][j] * w2[j][c];     // two products, one sum: the
            v[j] = p0 + p2;                                                      // reference's torch expression
        }
        __builtin_nontemporal_store(v, reinterpret_cast<MEMC_GLOBAL f32x4u *>(out_u + (uintptr_t)(c * s1c) * 4u + o1));
    }
}

// --------------------------------------------------------------------------------------------------
// EXTENSION (no reference counterpart): the IMAGE warp and the 64-channel CONTEXT warp of one
// direction in one pass.  networks/MEMC_Net_star.py:273-285 warps the frame and its context features with the SAME
// flow and the same 16 filter planes (FilterInterpolate, then FilterInterpolate_ctx): as two launches the 72 B per
// site of flow + taps are streamed twice.  Here the image is simply one more chunk in front of fi_fwd_tiled_c4n's
// chunk loop (3 channels + a duplicate plane in the pixel quad), and with BLEND the image chunk's epilogue is the
// occlusion-weighted blend of MEMC_Net_star.py:277,
//     image_out = occ_prev * prev + occ_this * FI(image, flow, filter)
// where `prev` is the other direction's warp (written by the BLEND == false launch before): per frame pair
// 2 * (584 + 24) + 20 = 1236 B per site instead of 188 + 2 * 584 = 1356.
// Same arithmetic as fi_fwd_blend_c3 (two products, one sum) and as fi_fwd_tiled_c4n: identical results.
// --------------------------------------------------------------------------------------------------
template <bool BLEND>
__global__ __launch_bounds__(256, 2) void fi_fwd_ctx_img(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t sib, int64_t sic, int sih,               // image, prev, image_out  [B, 3, H, W]
    int64_t s1b, int64_t s1c, int s1h,               // context, context_out    [B, C, H, W], C % 4 == 0
    int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h, int64_t sob, int soh,
    const float *__restrict__ img, const float *__restrict__ in1, const float *__restrict__ flow,
    const float *__restrict__ filt, const float *__restrict__ prev, const float *__restrict__ occ_prev,
    const float *__restrict__ occ_this, float *__restrict__ img_out, float *__restrict__ out)
{
    constexpr int LX = 16;
    using G = TileGeom<LX>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    int *bb = reinterpret_cast<int *>(smem + G::kCapPx * 16);

    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, gridDim.x / (tiles_x * tiles_y));
    const int b = tc.b, tile_x0 = tc.tx * G::kTW, tile_y0 = tc.ty * G::kTH;
    const int x = tile_x0 + 4 * (threadIdx.x % LX), y = tile_y0 + threadIdx.x / LX;
    const bool inb = x < W && y < H;
    const int xs = min(x, W - 4), ys = min(y, H - 1);
    const float *flow_p = flow + b * s2b + (int64_t)ys * s2h + xs;
    const float *tap_p = filt + b * s3b + (int64_t)ys * s3h + xs;
    const f32x4 fx4 = ld_stream4(flow_p), fy4 = ld_stream4(flow_p + s2c);
    f32x4 tp[16];
#pragma unroll
    for (int k = 0; k < 16; k++) tp[k] = ld_stream4(tap_p + k * s3c);

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
    const Bands bands = make_bands<LX>(box);
    const float *img_b = img + b * sib;
    const float *in_b = in1 + b * s1b;
    float *iout_p = img_out + b * sib + (int64_t)y * sih + x;
    float *out_p = out + b * s1b + (int64_t)y * s1h + x;
    // blend of ONE element (rare paths): two products, one sum, as the reference's torch expression (and
    // fi_fwd_blend_c3).  Addresses are rebuilt at the use: six more live pointer registers do not fit this kernel.
    auto blend1 = [&](float w, int c, int j) {
        if (!BLEND) return w;
        // (the lane's coordinates rebuilt from its index, opaque: hoisted out of the chunk loop, these addresses were spilled)
        unsigned t = tid_now();
        asm volatile("" : "+v"(t));
        const int x = tile_x0 + 4 * (int)(t % LX), y = tile_y0 + (int)(t / LX);
        const int64_t oi = b * sib + c * sic + (int64_t)y * sih + x + j, oo = b * sob + (int64_t)y * soh + x + j;
        const float p0 = occ_prev[oo] * prev[oi], p2 = occ_this[oo] * w;
        return p0 + p2;
    };
    unsigned done = 0;
    const int nchunks = C / 4 + 1;                 // chunk 0: the image (RGB + a duplicate plane); then the context
#pragma unroll 1
    for (int bi = 0; bi < bands.n; bi++) {
        const Region r = band_region(box, bands, bi);
        const unsigned sel = inb ? fi_covered(r, g, W, H) & ~done : 0u;
        // later bands only run when somebody still needs them; the vote is also the barrier that frees the LDS
        if (bi > 0 && !__syncthreads_or(sel != 0)) continue;
        done |= sel;
        // band 0 also writes the out-of-range sites (they copy the input pixel)
        const unsigned wr = sel | (bi == 0 && inb ? ~g.valid & 0xFu : 0u);
        const StageSlot sl = stage_slots(r);
        StageRegs<4> sr;
        {
            const float *const planes[4] = {img_b, img_b + sic, img_b + 2 * sic, img_b + 2 * sic};
            const int hs[4] = {sih, sih, sih, sih};
            tile_stage_load_planes<4>(r, sl, planes, hs, sr);
        }
#pragma unroll 1
        for (int vc = 0; vc < nchunks; vc++) {
            tile_stage_store<4>(r, sl, sr, tile);
            __syncthreads();
            // next chunk's rows: in flight while this chunk is gathered (the last iteration re-reads its own
            // chunk -- harmless, keeps the loads unconditional)
            const int cn = vc + 1 < nchunks ? vc * 4 : (vc - 1) * 4;      // context channel of the next chunk
            tile_stage_load<4>(r, sl, in_b + cn * s1c, s1c, s1h, sr);
            // keep the loop-invariant tap splats / LDS addresses inside the loop (see fi_fwd_tiled_fs4)
#pragma unroll
            for (int k = 0; k < 16; k++)
                asm volatile("" : "+v"(tp[k][0]), "+v"(tp[k][1]), "+v"(tp[k][2]), "+v"(tp[k][3]));
#pragma unroll
            for (int j = 0; j < 4; j++) asm volatile("" : "+v"(g.ix[j]), "+v"(g.iy[j]), "+v"(g.a[j]), "+v"(g.b[j]));
            f32x4 res[4];
#pragma unroll
            for (int j = 0; j < 4; j++) res[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            fi_gather<LX, 4>(r, g, tp, sel, W, H, tile, res);
            if (vc == 0) {                         // ---- the image chunk: three channels, optional blend ----
                if (wr & ~g.valid) {               // out-of-range sites copy the input pixel
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const f32x4 own = ld_cached4(img_b + c * sic + (int64_t)y * sih + x);
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (!((g.valid >> j) & 1)) res[j][c] = own[j];
                    }
                }
                if (wr == 0xFu) {
                    // (wave-uniform bases + one 32-bit lane offset per layout, see ld_stream4_u)
                    const unsigned oi = 4u * (unsigned)(y * sih + x), oo = 4u * (unsigned)(y * soh + x);
                    f32x4 oa = {0.f, 0.f, 0.f, 0.f}, ob = oa;
                    if (BLEND) {
                        oa = ld_stream4_u(occ_prev + b * sob, oo);
                        ob = ld_stream4_u(occ_this + b * sob, oo);
                    }
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        f32x4 v = {res[0][c], res[1][c], res[2][c], res[3][c]};
                        if (BLEND) {
                            const f32x4 pv = ld_stream4_u(prev + b * sib + c * sic, oi);
#pragma unroll
                            for (int j = 0; j < 4; j++) {
                                const float p0 = oa[j] * pv[j], p2 = ob[j] * v[j];
                                v[j] = p0 + p2;
                            }
                        }
                        st_stream4(iout_p + c * sic, v);
                    }
                } else if (wr) {                   // a lane whose sites are split over bands
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if ((wr >> j) & 1) {
#pragma unroll
                            for (int c = 0; c < 3; c++) iout_p[c * sic + j] = blend1(res[j][c], c, j);
                        }
                }
            } else {                               // ---- a context chunk: exactly fi_fwd_tiled_c4n ----
                const int c0 = (vc - 1) * 4;
                const float *plane0 = in_b + c0 * s1c;
                float *o = out_p + c0 * s1c;
                if (wr & ~g.valid) {
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const f32x4 own = ld_cached4(plane0 + c * s1c + (int64_t)y * s1h + x);
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            if (!((g.valid >> j) & 1)) res[j][c] = own[j];
                    }
                }
                if (wr == 0xFu) {
#pragma unroll
                    for (int c = 0; c < 4; c++) st_stream4(o + c * s1c, f32x4{res[0][c], res[1][c], res[2][c], res[3][c]});
                } else if (wr) {
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if ((wr >> j) & 1) {
#pragma unroll
                            for (int c = 0; c < 4; c++) o[c * s1c + j] = res[j][c];
                        }
                }
            }
            __syncthreads();
        }
    }
    unsigned slow = inb ? g.valid & ~done : 0u;            // not coverable within kMaxBands bands
    if (slow == 0) return;
    // (rare) the lane's pointers are rebuilt HERE from an opaque copy of its coordinates: as the values computed at the top
    // of the kernel they lived through the chunk loop for this path alone -- in private scratch (64 bytes per lane, rounds 3-5)
    unsigned tl = tid_now();
    asm volatile("" : "+v"(tl));
    const int xl = tile_x0 + 4 * (int)(tl % LX), yl = tile_y0 + (int)(tl / LX);
    const float *flow_q = flow + b * s2b + (int64_t)min(yl, H - 1) * s2h + min(xl, W - 4);
    const float *tap_q = filt + b * s3b + (int64_t)min(yl, H - 1) * s3h + min(xl, W - 4);
    float *iout_q = img_out + b * sib + (int64_t)yl * sih + xl;
    float *out_q = out + b * s1b + (int64_t)yl * s1h + xl;
    while (slow) {
        const int j = __ffs(slow) - 1;
        slow &= slow - 1;
        fi_site_scalar(xl + j, yl, W, H, C, 4, in_b, s1c, s1h, flow_q + j, s2c, tap_q + j, s3c, out_q + j);
        fi_site_scalar(xl + j, yl, W, H, 3, 4, img_b, sic, sih, flow_q + j, s2c, tap_q + j, s3c, iout_q + j);
        if (BLEND) {
#pragma unroll
            for (int c = 0; c < 3; c++) iout_q[c * sic + j] = blend1(iout_q[c * sic + j], c, j);
        }
    }
}

// --------------------------------------------------------------------------------------------------
// Forward, fs == 4, direct gather from global memory through L1/L2.
//   ROWS waves per workgroup, one image row each: tile = 64 x ROWS sites.
//   CT > 0: channel count known at compile time (fully unrolled); CT == 0: run-time channel loop.
// --------------------------------------------------------------------------------------------------
template <int CT, int ROWS>
__global__ __launch_bounds__(64 * ROWS) void fi_fwd_direct_fs4(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    float *__restrict__ out, int x0)
{
    // (x0: first column served -- 0, or the first column behind a ragged width's whole quads, which the tiled kernel takes)
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = x0 + tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * ROWS + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = ld_stream(flow_b);
    const float fy = ld_stream(flow_b + s2c);
    const float *tap_p = filt + b * s3b + (int64_t)y * s3h + x;
    float t[16];
#pragma unroll
    for (int k = 0; k < 16; k++) t[k] = ld_stream(tap_p + k * s3c);

    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    const float *in_b = in1 + b * s1b;
    float *out_p = out + b * s1b + (int64_t)y * s1h + x;
    const int nc = CT > 0 ? CT : C;
    constexpr int kUnroll = CT > 0 ? CT : 4;

    if (s.valid) {
        int ro[4], co[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            ro[k] = clampi(s.iy - 1 + k, H - 1) * s1h;
            co[k] = clampi(s.ix - 1 + k, W - 1);
        }
        const float w00 = (1 - s.a) * (1 - s.b), w01 = s.a * (1 - s.b);
        const float w10 = (1 - s.a) * s.b, w11 = s.a * s.b;
#pragma unroll kUnroll
        for (int c = 0; c < nc; c++) {
            const float *p = in_b + c * s1c;
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int i = 0; i < 4; i++) v[j * 4 + i] = p[ro[j] + co[i]];
            // quadrant sums: rows {0,1} top / {2,3} bottom, cols {0,1} left / {2,3} right
            float TL = 0.0f, TR = 0.0f, BL = 0.0f, BR = 0.0f;
            TL += v[0] * t[0];   TL += v[1] * t[1];   TL += v[4] * t[4];   TL += v[5] * t[5];
            TR += v[2] * t[2];   TR += v[3] * t[3];   TR += v[6] * t[6];   TR += v[7] * t[7];
            BL += v[8] * t[8];   BL += v[9] * t[9];   BL += v[12] * t[12]; BL += v[13] * t[13];
            BR += v[10] * t[10]; BR += v[11] * t[11]; BR += v[14] * t[14]; BR += v[15] * t[15];
            st_stream(out_p + c * s1c, w00 * TL + w01 * TR + w10 * BL + w11 * BR);
        }
    } else {
        // out-of-range site copies the input pixel (my_lib_kernel.cu:1209-1214)
        const float *p = in_b + (int64_t)y * s1h + x;
        for (int c = 0; c < nc; c++) st_stream(out_p + c * s1c, p[c * s1c]);
    }
}

// --------------------------------------------------------------------------------------------------
// Forward, any filter size (run-time loops, taps read from global per use).  Rare path: the networks
// only ever use fs == 4 (MEMC_Net_star.py:30 `filter_size = 4`).
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fi_fwd_generic(
    int W, int H, int C, int fs, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    float *__restrict__ out)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
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

#ifdef MEMC_MEASURE
#include "arms/fi_fwd_arms.hpp"           // fi_fwd_refshape: measurement build only
#endif

// --------------------------------------------------------------------------------------------------
// Backward, fs == 4, direct.  Per valid site (my_lib_kernel.cu:1248-1515):
//   gradinput1 += scatter of g * wq * tap      (fp32 atomics; neighbouring lanes hit neighbouring cells)
//   gradinput3 += sum_c g * wq * in            (each site owns its 16 taps: accumulated in registers over
//                                               the channels, ONE read-modify-write per tap; the
//                                               reference issues 16*C atomics for this)
//   gradinput2  = flow gradients from the quadrant sums (assignment; the sums are computed once, the
//                 reference recomputes them twice more)
// Invalid sites store zeros to gradinput2 / gradinput3 (which are therefore fully defined by the kernel) and
// leave gradinput1, an accumulation target, alone.
// --------------------------------------------------------------------------------------------------
template <int CT, int ROWS>
__global__ __launch_bounds__(64 * ROWS) void fi_bwd_direct_fs4(
    int W, int H, int C, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    const float *__restrict__ gout, float *__restrict__ gin1, float *__restrict__ gin2,
    float *__restrict__ gin3, int x0)
{
    // x0: the first column this launch serves (0; w & ~3 when fi_bwd_c3_pk took the whole quads of a ragged width)
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = x0 + tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * ROWS + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = ld_stream(flow_b);
    const float fy = ld_stream(flow_b + s2c);
    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    if (!s.valid) {                            // gradinput2 / gradinput3 are fully DEFINED by the kernels: an
        float *z3 = gin3 + b * s3b + (int64_t)y * s3h + x;    // invalid site stores the zeros the reference's
#pragma unroll                                                  // caller-side memset would have left there
        for (int k = 0; k < 16; k++) z3[k * s3c] = 0.0f;
        float *z2 = gin2 + b * s2b + (int64_t)y * s2h + x;
        z2[0] = 0.0f;
        z2[s2c] = 0.0f;
        return;
    }

    const float *tap_p = filt + b * s3b + (int64_t)y * s3h + x;
    float t[16], gt[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        t[k] = ld_stream(tap_p + k * s3c);
        gt[k] = 0.0f;
    }
    int ro[4], co[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        ro[k] = clampi(s.iy - 1 + k, H - 1) * s1h;
        co[k] = clampi(s.ix - 1 + k, W - 1);
    }
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    const float *gout_p = gout + b * s1b + (int64_t)y * s1h + x;
    const int nc = CT > 0 ? CT : C;
    constexpr int kUnroll = CT > 0 ? CT : 2;
    float botx = 0.0f, boty = 0.0f;
    const float gam_x = 1.0f - s.b, gam_y = 1.0f - s.a;
#pragma unroll kUnroll
    for (int c = 0; c < nc; c++) {
        const float *p = in_b + c * s1c;
        float *q = gin1_b + c * s1c;
        const float g = ld_stream(gout_p + c * s1c);
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) v[j * 4 + i] = p[ro[j] + co[i]];
        const float wq[4] = {g * (1 - s.a) * (1 - s.b), g * s.a * (1 - s.b),
                             g * (1 - s.a) * s.b, g * s.a * s.b};
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int k = j * 4 + i;
                const float wgt = wq[(j >> 1) * 2 + (i >> 1)];
                atomic_add_f32(q + ro[j] + co[i], wgt * t[k]);
                gt[k] += wgt * v[k];
            }
        float TL = 0.0f, TR = 0.0f, BL = 0.0f, BR = 0.0f;
        TL += v[0] * t[0];   TL += v[1] * t[1];   TL += v[4] * t[4];   TL += v[5] * t[5];
        TR += v[2] * t[2];   TR += v[3] * t[3];   TR += v[6] * t[6];   TR += v[7] * t[7];
        BL += v[8] * t[8];   BL += v[9] * t[9];   BL += v[12] * t[12]; BL += v[13] * t[13];
        BR += v[10] * t[10]; BR += v[11] * t[11]; BR += v[14] * t[14]; BR += v[15] * t[15];
        float tmp = 0.0f;
        tmp += gam_x * (TR - TL);
        tmp += (1.0f - gam_x) * (BR - BL);
        botx += g * tmp;
        tmp = 0.0f;
        tmp += gam_y * (BL - TL);
        tmp += (1.0f - gam_y) * (BR - TR);
        boty += g * tmp;
    }
    float *g3 = gin3 + b * s3b + (int64_t)y * s3h + x;
#pragma unroll
    for (int k = 0; k < 16; k++) g3[k * s3c] = gt[k];         // stored: the site owns its taps
    float *g2 = gin2 + b * s2b + (int64_t)y * s2h + x;
    st_stream(g2, botx);
    st_stream(g2 + s2c, boty);
}


// Backward, any filter size (rare path; run-time loops).
__global__ __launch_bounds__(256) void fi_bwd_generic(
    int W, int H, int C, int fs, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t s2b, int64_t s2c, int s2h, int64_t s3b, int64_t s3c, int s3h,
    const float *__restrict__ in1, const float *__restrict__ flow, const float *__restrict__ filt,
    const float *__restrict__ gout, float *__restrict__ gin1, float *__restrict__ gin2,
    float *__restrict__ gin3)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;
    const float *flow_b = flow + b * s2b + (int64_t)y * s2h + x;
    const float fx = flow_b[0], fy = flow_b[s2c];
    const FiSite s = fi_locate(x, y, W, H, fx, fy);
    float *g3 = gin3 + b * s3b + (int64_t)y * s3h + x;
    if (!s.valid) {                            // gradinput2 / gradinput3 are fully defined by the kernels
        for (int k = 0; k < fs * fs; k++) g3[k * s3c] = 0.0f;
        float *z2 = gin2 + b * s2b + (int64_t)y * s2h + x;
        z2[0] = 0.0f;
        z2[s2c] = 0.0f;
        return;
    }
    const int L = s.ix + 1 - fs / 2, T = s.iy + 1 - fs / 2, R = L + fs, Bm = T + fs;
    const float *tap_p = filt + b * s3b + (int64_t)y * s3h + x;
    const float *in_b = in1 + b * s1b;
    float *gin1_b = gin1 + b * s1b;
    const float *gout_p = gout + b * s1b + (int64_t)y * s1h + x;
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
                // this site owns its taps: stored by the first channel, accumulated by the others
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
    float *g2 = gin2 + b * s2b + (int64_t)y * s2h + x;
    g2[0] = botx;
    g2[s2c] = boty;
}

}  // namespace memc

using namespace memc;

// Variant selection for A/B measurement (memc_internal.h); -1 = automatic.  Measurement build only: in the product
// build both are compile-time constants (-1), every `variant == n` branch below folds away and the ablation kernels
// are never instantiated.  Nothing is read from the environment in either build.
#ifdef MEMC_MEASURE
MEMC_KNOB_STATIC(g_fi_fwd_variant, -1);
MEMC_KNOB_STATIC(g_fi_bwd_variant, -1);
extern "C" void memc_debug_set_fi_fwd_variant(int v) { g_fi_fwd_variant = v; }
extern "C" void memc_debug_set_fi_phase(int per_win)          // period * 65536 + window, ticks of 10 ns (arm 26)
{
    const unsigned pw[2] = {(unsigned)per_win >> 16, (unsigned)per_win & 0xffffu};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(memc::g_fi_phase), pw, sizeof(pw));
}
extern "C" void memc_debug_set_fi_bwd_variant(int v) { g_fi_bwd_variant = v; }
#endif
extern "C" int FilterInterpolationLayer_gpu_forward_kernel(
    memc_stream_t stream_, const int nElement, const int w, const int h, const int channel, const int batch,
    const int filter_size,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const int s3b, const int s3c, const int s3h, const int s3w,
    const float *input1, const float *input2, const float *input3, float *output)
{
    (void)nElement; (void)s1w; (void)s2w; (void)s3w;
    hipStream_t stream = (hipStream_t)stream_;
    if (w <= 0 || h <= 0 || channel <= 0 || batch <= 0) return 0;
    // production path: LDS-tiled, 16 B per lane (needs 4-element-aligned geometry)
    const bool vec = vec4_ok(w, {s1b, s1c, s1h, s2b, s2c, s2h, s3b, s3c, s3h}, {input1, input2, input3, output});
#define MEMC_FI_TILED(LX, CT, MINW)   MEMC_FI_TILED_A(LX, CT, MINW, 0)
#define MEMC_FI_TILED_A(LX, CT, MINW, WALK)                                                                     \
    do {                                                                                                   \
        using G = TileGeom<LX>;                                                                            \
        const int ntx = (w + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;                        \
        hipLaunchKernelGGL((fi_fwd_tiled_fs4<LX, CT, MINW, WALK>), dim3((unsigned)ntx * nty * batch), dim3(256), \
                           tile_lds_bytes<LX>(), stream, w, h, channel, ntx, nty, (int64_t)s1b,            \
                           (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b, (int64_t)s3c, \
                           s3h, input1, input2, input3, output);                                           \
    } while (0)
#define MEMC_FI_C4N(SW) MEMC_FI_C4N_NT(SW, 256, false)
#define MEMC_FI_C4N_NT(SW, NT, RAG) MEMC_FI_C4N_LX(SW, NT, RAG, 16)
#define MEMC_FI_C4N_LX(SW, NT, RAG, LX) MEMC_FI_C4N_ABL(SW, NT, RAG, LX, 0)
#define MEMC_FI_C4N_ABL(SW, NT, RAG, LX, ABL)                                                                   \
    do {                                                                                                   \
        using G = TileGeom<LX, (NT == 256 ? 3072 : 6144), NT>;                                             \
        const int ntx = (w + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;                        \
        const int lds = G::kCapPx * 16 + 4 * 4 * (NT / 64);                                                \
        allow_big_lds(fi_fwd_tiled_c4n<SW, NT, RAG, LX, ABL>, lds);   /* per launch: a per-DEVICE attribute */ \
        hipLaunchKernelGGL((fi_fwd_tiled_c4n<SW, NT, RAG, LX, ABL>), dim3((unsigned)((ntx + (SW ? SW : 1) - 1) / (SW ? SW : 1) * \
                                                                    (SW ? SW : 1)) * nty * batch),          \
                           dim3(NT), lds, stream, w, h, channel, ntx, nty, (int64_t)s1b,                   \
                           (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b, (int64_t)s3c,    \
                           s3h, input1, input2, input3, output);                                           \
    } while (0)
#define MEMC_FI_FWD_LAUNCH(CT, ROWS)                                                                       \
    do {                                                                                                   \
        const int tiles_x = (w + kWave - 1) / kWave, tiles_y = (h + (ROWS) - 1) / (ROWS);                  \
        const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;                                          \
        hipLaunchKernelGGL((fi_fwd_direct_fs4<CT, ROWS>), dim3(nwg), dim3(64 * (ROWS)), 0, stream, w, h,   \
                           channel, tiles_x, tiles_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b,       \
                           (int64_t)s2c, s2h, (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3,     \
                           output, 0);                                                                     \
    } while (0)

#ifdef MEMC_MEASURE
#define MEMC_FI_FWD_NO_ARM && g_fi_fwd_variant < 0
#else
#define MEMC_FI_FWD_NO_ARM
#endif
#ifdef MEMC_MEASURE
    // ---- measurement build only: A/B and ablation arms (tools/bench_ops.py); several return WRONG results ----
    const int variant = g_fi_fwd_variant;
    if (variant == 0) {  // reference-structure measurement arm
        dim3 block(32, 16, 1), grid((w + 31) / 32, (h + 15) / 16, batch);
        hipLaunchKernelGGL(fi_fwd_refshape, grid, block, 0, stream, w, h, channel, filter_size,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, output);
        return launch_status();
    }
    if (filter_size == 4 && variant >= 1 && variant <= 3) {       // force the scalar direct-gather kernels
        if (channel == 3) {
            if (variant == 2) MEMC_FI_FWD_LAUNCH(3, 8);
            else if (variant == 3) MEMC_FI_FWD_LAUNCH(3, 2);
            else MEMC_FI_FWD_LAUNCH(3, 4);
        } else {
            if (variant == 2) MEMC_FI_FWD_LAUNCH(0, 8);
            else if (variant == 3) MEMC_FI_FWD_LAUNCH(0, 2);
            else MEMC_FI_FWD_LAUNCH(0, 4);
        }
        return launch_status();
    }
    if (filter_size == 4 && vec && variant >= 4) {
        bool handled = true;
        if (variant == 5) {
            if (channel == 3) MEMC_FI_TILED(8, 3, 3); else MEMC_FI_TILED(8, 0, 3);
        } else if (variant == 6) {
            if (channel == 3) MEMC_FI_TILED(16, 3, 2); else MEMC_FI_TILED(16, 0, 2);
        } else if (variant == 7) {
            if (channel == 3) MEMC_FI_TILED(8, 3, 2); else MEMC_FI_TILED(8, 0, 2);
        } else if (variant == 4) {
            if (channel == 3) MEMC_FI_TILED(16, 3, 3); else MEMC_FI_TILED(16, 0, 3);
        } else if (variant == 8 && channel == 3) {
            MEMC_FI_TILED_A(16, 3, 2, 1);
        } else if (variant == 11 && channel == 3) {
            MEMC_FI_TILED_A(16, 3, 3, 4);
        } else if ((variant == 15 || variant == 16 || variant == 17) && channel == 3) {
            using G = TileGeom<16>;
            const int ntx = (w + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;
            const int sw = variant == 15 ? 2 : 4;
            const unsigned grid = (unsigned)((ntx + sw - 1) / sw * sw) * nty * batch;
#define MEMC_FI_STRIPE(WALK, MINW)                                                                             \
            hipLaunchKernelGGL((fi_fwd_tiled_fs4<16, 3, MINW, WALK>), dim3(grid), dim3(256), tile_lds_bytes<16>(), \
                               stream, w, h, channel, ntx, nty, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b,     \
                               (int64_t)s2c, s2h, (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, output)
            if (variant == 15) MEMC_FI_STRIPE(5, 2);
            else if (variant == 16) MEMC_FI_STRIPE(6, 2);
            else MEMC_FI_STRIPE(4, 2);                     // 17: row-major chunk per XCD at 2 waves/SIMD
#undef MEMC_FI_STRIPE
        } else if (variant == 26 && channel == 3) {       // the product kernel with phased stores (memc_debug_set_fi_phase)
            using G = TileGeom<16>;
            const int ntx = (w + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;
            hipLaunchKernelGGL((fi_fwd_tiled_fs4<16, 3, 2, 0, false, 3072, 1>), dim3((unsigned)ntx * nty * batch), dim3(256),
                               tile_lds_bytes<16>(), stream, w, h, channel, ntx, nty, (int64_t)s1b, (int64_t)s1c, s1h,
                               (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, output);
        } else if (variant >= 20 && variant <= 25 && channel == 3) {
            // 20: 128 x 8 tiles (LX = 32) with a 4608-pixel budget, strips; 21: the same in hardware order; 22: 64 x 16 tiles
            // with a 4096-pixel budget (no band sweeps on i.i.d. flow); 23: 128 x 8 tiles on the product's 3072 pixels; 24 / 25:
            // 128 x 8 tiles on 3392 pixels (53 KiB: the most that leaves three workgroups per CU), registers for two / three
#define MEMC_FI_WIDE(LX, WALK, CAP) MEMC_FI_WIDE_M(LX, WALK, CAP, 2)
#define MEMC_FI_WIDE_M(LX, WALK, CAP, MINW)                                                                             \
            do {                                                                                                  \
                using G = TileGeom<LX, CAP>;                                                                      \
                const int ntx = (w + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;                       \
                allow_big_lds(fi_fwd_tiled_fs4<LX, 3, MINW, WALK, false, CAP>, tile_lds_bytes<LX, CAP>());        \
                hipLaunchKernelGGL((fi_fwd_tiled_fs4<LX, 3, MINW, WALK, false, CAP>), dim3((unsigned)ntx * nty * batch), \
                                   dim3(256), (tile_lds_bytes<LX, CAP>()), stream, w, h, channel, ntx, nty,       \
                                   (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,              \
                                   (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, output);              \
            } while (0)
            if (variant == 20) MEMC_FI_WIDE(32, 0, 4608);
            else if (variant == 21) MEMC_FI_WIDE(32, 1, 4608);
            else if (variant == 22) MEMC_FI_WIDE(16, 0, 4096);
            else if (variant == 23) MEMC_FI_WIDE(32, 0, 3072);
            else if (variant == 24) MEMC_FI_WIDE(32, 0, 3392);
            else MEMC_FI_WIDE_M(32, 0, 3392, 3);
#undef MEMC_FI_WIDE
#undef MEMC_FI_WIDE_M
        } else if (variant == 30 && channel % 4 == 0 && channel >= 8) {
            MEMC_FI_C4N(2);
        } else if (variant == 31 && channel % 4 == 0 && channel >= 8) {
            MEMC_FI_C4N(4);
        } else if (variant == 32 && channel % 4 == 0 && channel >= 8) {
            MEMC_FI_C4N_NT(0, 512, false);                 // 64 x 32 tiles, 512 lanes
        } else if (variant == 35 && channel % 4 == 0 && channel >= 8) {
            MEMC_FI_C4N_NT(4, 512, false);                 // ... in stripes four tile columns wide
        } else if (variant == 36 && channel % 4 == 0 && channel >= 8) {
            MEMC_FI_C4N_ABL(0, 256, false, 16, 1);         // timing: no gathers
        } else if (variant == 37 && channel % 4 == 0 && channel >= 8) {
            MEMC_FI_C4N_ABL(0, 256, false, 16, 2);         // timing: no image loads after the first chunk
        } else if (variant == 38 && channel % 4 == 0 && channel >= 8) {
            MEMC_FI_C4N_ABL(0, 512, false, 16, 1);
        } else if (variant == 39 && channel % 4 == 0 && channel >= 8) {
            MEMC_FI_C4N_ABL(0, 512, false, 16, 2);
        } else if (variant == 33 && channel % 4 == 0 && channel >= 8) {
            MEMC_FI_C4N_LX(0, 256, false, 8);              // 32 x 32 tiles: the box of a square tile is the least dilated
        } else if (variant == 34 && channel % 4 == 0 && channel >= 8) {
            MEMC_FI_C4N_LX(4, 256, false, 8);              // ... in stripes four tile columns wide
        } else {
            handled = false;
        }
        if (handled) return launch_status();
    }
#endif  // MEMC_MEASURE

    if (filter_size == 4 && vec) {
        if (channel % 4 == 0 && channel >= 8) {                        // e.g. the 64-channel context warp
            MEMC_PATH("fi_fwd:tiled_c4n");
            MEMC_FI_C4N(0);
        } else if (channel == 3) {                                     // default: 64x16 tiles, strip walk
            MEMC_PATH("fi_fwd:tiled_c3");
            MEMC_FI_TILED(16, 3, 2);
        } else if (channel >= 4) {                                     // any other count from four up: the same pipeline,
            MEMC_PATH("fi_fwd:tiled_c4n_ragged");                      // ragged last chunk
            MEMC_FI_C4N_NT(0, 256, true);
        } else {
            MEMC_PATH("fi_fwd:tiled_chunks");
            MEMC_FI_TILED(16, 0, 2);
        }
        return launch_status();
    }
    if (filter_size != 4) {
        const int tiles_x = (w + kWave - 1) / kWave, tiles_y = (h + 3) / 4;
        const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
        MEMC_PATH("fi_fwd:generic");
        hipLaunchKernelGGL(fi_fwd_generic, dim3(nwg), dim3(256), 0, stream, w, h, channel, filter_size,
                           tiles_x, tiles_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, output);
        return launch_status();
    }
    // A width that is not a multiple of four (round 5): the tiled kernel takes the whole quads (sites x < ws), the one-lane-
    // per-site kernel the one to three columns behind them.
    const int ws = w & ~3;
    if (filter_size == 4 && !vec && ws >= 8 MEMC_FI_FWD_NO_ARM) {
        using G = TileGeom<16>;
        const int ntx = (ws + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;
        const int tail_y = (h + 3) / 4;
        MEMC_PATH(channel == 3 ? "fi_fwd:tiled_c3" : channel >= 4 ? "fi_fwd:tiled_c4n_ragged" : "fi_fwd:tiled_chunks");
#define MEMC_FI_TILED_RAGW(CT)                                                                                  \
        hipLaunchKernelGGL((fi_fwd_tiled_fs4<16, CT, 2, 0, true>), dim3((unsigned)ntx * nty * batch), dim3(256),    \
                           tile_lds_bytes<16>(), stream, w, h, channel, ntx, nty, (int64_t)s1b, (int64_t)s1c, s1h,   \
                           (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, output)
#define MEMC_FI_TAIL(CT)                                                                                        \
        hipLaunchKernelGGL((fi_fwd_direct_fs4<CT, 4>), dim3((unsigned)tail_y * batch), dim3(256), 0, stream, w, h,  \
                           channel, 1, tail_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,     \
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, output, ws)
        if (channel >= 4) {                    // the chunk pipeline's any-channel-count instantiation, ragged rows
            using G4 = TileGeom<16, 3072, 256>;
            const int lds = G4::kCapPx * 16 + 4 * 4 * (256 / 64);
            allow_big_lds(fi_fwd_tiled_c4n<0, 256, true, 16, 0, true>, lds);
            hipLaunchKernelGGL((fi_fwd_tiled_c4n<0, 256, true, 16, 0, true>), dim3((unsigned)ntx * nty * batch), dim3(256), lds,
                               stream, w, h, channel, ntx, nty, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                               (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, output);
            MEMC_FI_TAIL(0);
        } else if (channel == 3) {
            MEMC_FI_TILED_RAGW(3);
            MEMC_FI_TAIL(3);
        } else {
            MEMC_FI_TILED_RAGW(0);
            MEMC_FI_TAIL(0);
        }
#undef MEMC_FI_TILED_RAGW
#undef MEMC_FI_TAIL
        return launch_status();
    }
    // everything else: the one-lane-per-site kernels
    MEMC_PATH("fi_fwd:direct");
    if (channel == 3) MEMC_FI_FWD_LAUNCH(3, 4);
    else MEMC_FI_FWD_LAUNCH(0, 4);
#undef MEMC_FI_FWD_LAUNCH
#undef MEMC_FI_FWD_NO_ARM
#undef MEMC_FI_C4N
#undef MEMC_FI_C4N_NT
#undef MEMC_FI_C4N_LX
#undef MEMC_FI_TILED
#undef MEMC_FI_TILED_A
    return launch_status();
}

extern "C" int FilterInterpolationLayer_gpu_backward_kernel(
    memc_stream_t stream_, const int nElement, const int w, const int h, const int channel, const int batch,
    const int filter_size,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const int s3b, const int s3c, const int s3h, const int s3w,
    const float *input1, const float *input2, const float *input3,
    const float *gradoutput, float *gradinput1, float *gradinput2, float *gradinput3)
{
    (void)nElement; (void)s1w; (void)s2w; (void)s3w;
    hipStream_t stream = (hipStream_t)stream_;
    if (w <= 0 || h <= 0 || channel <= 0 || batch <= 0) return 0;
    const int tiles_x = (w + kWave - 1) / kWave;
    const int tiles_y = (h + 3) / 4;
    const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
    int taken = 0;
    // EXTENSION: gradinput1 == NULL ("the image gradient is not wanted", include/memc_warp.h) is served by the RGB tiled
    // kernel only; every other shape returns -1 and the caller passes a buffer
    if (gradinput1 == nullptr && (filter_size != 4 || channel != 3)) return -1;
#ifdef MEMC_MEASURE
    const bool direct_only = g_fi_bwd_variant == 40;       // A/B: the direct kernel (global atomics) for any channel count
#else
    constexpr bool direct_only = false;
#endif
    if (filter_size != 4) {
        MEMC_PATH("fi_bwd:generic");
        hipLaunchKernelGGL(fi_bwd_generic, dim3(nwg), dim3(256), 0, stream, w, h, channel, filter_size,
                           tiles_x, tiles_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, gradoutput,
                           gradinput1, gradinput2, gradinput3);
    } else if (channel == 3 &&
               (taken = fi_bwd_c3_launch(stream, w, h, batch, s1b, s1c, s1h, s2b, s2c, s2h, s3b, s3c, s3h, input1, input2,
                                         input3, gradoutput, gradinput1, gradinput2, gradinput3,
#ifdef MEMC_MEASURE
                                         g_fi_bwd_variant
#else
                                         -1
#endif
                                         )) != 0) {
        MEMC_PATH("fi_bwd:tiled_c3");
        if (taken == 2) {                                  // a ragged width: the columns behind the whole quads (both ADD into gradinput1)
            const int ws = w & ~3, tail_y = (h + 3) / 4;
            hipLaunchKernelGGL((fi_bwd_direct_fs4<3, 4>), dim3((unsigned)tail_y * batch), dim3(256), 0, stream, w, h, channel, 1,
                               tail_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b,
                               (int64_t)s3c, s3h, input1, input2, input3, gradoutput, gradinput1, gradinput2, gradinput3, ws);
            return launch_status();
        }
        return taken > 0 ? 0 : -1;                         // RGB: fi_bwd_c3.hip
    } else if (channel != 3 &&
               (taken = fi_bwd_cn_launch(stream, w, h, channel, batch, s1b, s1c, s1h, s2b, s2c, s2h, s3b, s3c, s3h, input1,
                                         input2, input3, gradoutput, gradinput1, gradinput2, gradinput3,
                                         direct_only)) != 0) {
        MEMC_PATH("fi_bwd:owner");
        return taken > 0 ? 0 : -1;                         // many channels: fi_bwd_cn.hip
    } else if (channel == 3) {
        if (gradinput1 == nullptr) return -1;              // (unaligned geometry: the direct kernel needs the buffer)
        MEMC_PATH("fi_bwd:direct");
        hipLaunchKernelGGL((fi_bwd_direct_fs4<3, 4>), dim3(nwg), dim3(256), 0, stream, w, h, channel,
                           tiles_x, tiles_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, gradoutput,
                           gradinput1, gradinput2, gradinput3, 0);
    } else {
        MEMC_PATH("fi_bwd:direct");
        hipLaunchKernelGGL((fi_bwd_direct_fs4<0, 4>), dim3(nwg), dim3(256), 0, stream, w, h, channel,
                           tiles_x, tiles_y, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h,
                           (int64_t)s3b, (int64_t)s3c, s3h, input1, input2, input3, gradoutput,
                           gradinput1, gradinput2, gradinput3, 0);
    }
    return launch_status();
}

// EXTENSION (no reference counterpart): fused dual warp + occlusion blend, see fi_fwd_blend_c3.
// Strides: in0 / in2 / out share (s1b, s1c, s1h); flow0 / flow1 (s2b, s2c, s2h); filter0 / filter1 (s3b, s3c, s3h);
// occlusion0 / occlusion1 are [B, 1, H, W] with (sob, soh).  RGB and fs == 4 only, 16-byte aligned geometry;
// returns -1 otherwise (the caller composes the result from the two forward calls instead).
extern "C" int FilterInterpolationBlend_gpu_forward_kernel(
    memc_stream_t stream_, const int w, const int h, const int channel, const int batch, const int filter_size,
    const int s1b, const int s1c, const int s1h, const int s2b, const int s2c, const int s2h,
    const int s3b, const int s3c, const int s3h, const int sob, const int soh,
    const float *input0, const float *input2, const float *flow0, const float *flow1,
    const float *filter0, const float *filter1, const float *occlusion0, const float *occlusion1, float *output)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    if (channel != 3 || filter_size != 4) return -1;
    if (!vec4_ok(w, {s1b, s1c, s1h, s2b, s2c, s2h, s3b, s3c, s3h, sob, soh},
                 {input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1, output}) ||
        !plane_fits_u32(w, h, {s1h, s2h, s3h, soh}))
        return -1;
    using G = TileGeom<16>;
    const int ntx = (w + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;
    hipLaunchKernelGGL(fi_fwd_blend_c3, dim3((unsigned)ntx * nty * batch), dim3(256), tile_lds_bytes<16>(), stream, w,
                       h, ntx, nty, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b,
                       (int64_t)s3c, s3h, (int64_t)sob, soh, input0, input2, flow0, flow1, filter0, filter1,
                       occlusion0, occlusion1, output);
    return launch_status();
}

// EXTENSION (no reference counterpart): image + context warp of one direction in one pass, see fi_fwd_ctx_img.
// image / prev / image_out share (sib, sic, sih) and have 3 channels; context / context_out share (s1b, s1c, s1h) and
// have `channel` (a multiple of 4, >= 4) channels; prev / occlusion_prev / occlusion_this are all NULL (plain warp of the
// image) or all given (image_out = occlusion_prev * prev + occlusion_this * warp).  fs == 4 only, 16-byte aligned
// geometry; returns -1 otherwise (the caller then uses the separate entry points).
extern "C" int FilterInterpolationCtx_gpu_forward_kernel(
    memc_stream_t stream_, const int w, const int h, const int channel, const int batch, const int filter_size,
    const int sib, const int sic, const int sih, const int s1b, const int s1c, const int s1h,
    const int s2b, const int s2c, const int s2h, const int s3b, const int s3c, const int s3h,
    const int sob, const int soh,
    const float *image, const float *context, const float *flow, const float *filter,
    const float *prev, const float *occlusion_prev, const float *occlusion_this, float *image_out, float *context_out)
{
    hipStream_t stream = (hipStream_t)stream_;
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    if (filter_size != 4 || channel < 4 || channel % 4 != 0) return -1;
    const bool blend = prev != nullptr;
    if (blend != (occlusion_prev != nullptr) || blend != (occlusion_this != nullptr)) return -1;
    if (!vec4_ok(w, {sib, sic, sih, s1b, s1c, s1h, s2b, s2c, s2h, s3b, s3c, s3h, blend ? sob : 0, blend ? soh : 0},
                 {image, context, flow, filter, prev, occlusion_prev, occlusion_this, image_out, context_out}))
        return -1;
    if (!plane_fits_u32(w, h, {sih, blend ? soh : 0})) return -1;
    using G = TileGeom<16>;
    const int ntx = (w + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;
#define MEMC_FI_CTX(BLEND)                                                                                       \
    hipLaunchKernelGGL(fi_fwd_ctx_img<BLEND>, dim3((unsigned)ntx * nty * batch), dim3(256), tile_lds_bytes<16>(), stream, \
                       w, h, channel, ntx, nty, (int64_t)sib, (int64_t)sic, sih, (int64_t)s1b, (int64_t)s1c, s1h,       \
                       (int64_t)s2b, (int64_t)s2c, s2h, (int64_t)s3b, (int64_t)s3c, s3h, (int64_t)sob, soh, image,      \
                       context, flow, filter, prev, occlusion_prev, occlusion_this, image_out, context_out)
    if (blend) MEMC_FI_CTX(true);
    else MEMC_FI_CTX(false);
#undef MEMC_FI_CTX
    return launch_status();
}
