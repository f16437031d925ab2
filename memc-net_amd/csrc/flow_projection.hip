// flow_projection.hip -- FlowProjection / DepthFlowProjection: forward splat of -flow to the intermediate
// frame, count-normalisation, hole filling; and their backward passes.  gfx950 only.
//
// Replaces my_package/src/my_lib_kernel.cu:1630-2516 of the reference (scatter :1630/:2053, averaging
// :1696/:2122, hole fill :1742/:2169, backward :1837/:2265; launchers :1905,:1994,:2365,:2458).
// Semantics: SURVEY.md appendix A.3/A.4, including the reference's observable quirks: duplicate adds when the
// clamped right/bottom neighbour coincides with the left/top one, and the hole fill's dead downward search.
//
// The depth-weighted operator is the same code with a per-site weight d (DEPTH == true); the plain
// operator is d == 1.
#include "memc_common.hpp"
#include "memc_internal.h"
#include "memc_tile.hpp"
#include "memc_scratch.hpp"

#include <atomic>
#include <mutex>
#include <type_traits>

namespace memc {

constexpr int kFarWords = 8;                  // words per tile of the owner kernel's far table (proj_owner5.hpp)
constexpr int kFarMaxTiles = 262144;          // tiles the owner-computes fast path serves (proj_owner_far deals the stamped ones out from a table in LDS)
constexpr int kFlagWords = 256;               // far flags of the fast path: image b -> word b % 256, + 1 summary word
// (measurement arm, proj_owner5 MOT = 5, projection variant -54; round 6) The images' motion estimates cached in the call's
// scratch: 64-bit word b % 256 behind the far flags holds (tag << 32) | packed motion, tag = this call's nonce ^ hash(b).  The
// first workgroups of an image (per XCD) find a foreign tag, sample the flow and publish; the later ones read ONE word instead of
// 128 scattered cache lines.  LOST: +2.5 ... 4 % on the benchmark's flow -- one hot word per image is worse than 64 warm lines
// (and read with device scope, past the L2, it doubled the call); profiles/r06_proj_motion_cache_arm.txt.
constexpr int kMotionCacheAt = 320, kMotionCacheWords = 256;      // (in ints: 8-byte aligned; 2 ints per entry)
__device__ __forceinline__ unsigned motion_tag(int nonce, int b) { return (unsigned)nonce ^ ((unsigned)b * 0x9E3779B1u); }
__device__ __forceinline__ unsigned motion_pack(int mx, int my) { return ((unsigned)(mx & 0xffff) << 16) | (unsigned)(my & 0xffff); }
__device__ __forceinline__ void motion_unpack(unsigned v, int &mx, int &my)
{
    mx = (int)(short)(v >> 16);
    my = (int)(short)(v & 0xffffu);
}

// --------------------------------------------------------------------------------------------------
// Pass 1: scatter.  One lane = one source site; 12 fp32 atomics per valid site (8 flow + 4 count) into
// the caller-zeroed output / count planes.  With smooth flow the 64 lanes of a wave target 64 (nearly)
// consecutive cells, so each wave-level atomic touches two or three cache lines.
// --------------------------------------------------------------------------------------------------
template <bool DEPTH>
__global__ __launch_bounds__(256) void proj_scatter(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_p = flow + b * s1b + (int64_t)y * s1h + x;
    const float fx = ld_stream(flow_p);
    const float fy = ld_stream(flow_p + s1c);
    const BlSite s = bl_locate<false>(x, y, W, H, fx, fy);
    if (!s.valid) return;
    float vx = -fx, vy = -fy, vc = 1.0f;
    if (DEPTH) {
        const float d = ld_stream(depth + b * sdb + (int64_t)y * sdh + x);
        vx = -d * fx;                                       // my_lib_kernel.cu:2102-2109
        vy = -d * fy;
        vc = d * 1.0f;                                      // :2111-2114
    }
    float *ox = out + b * s1b, *oy = ox + s1c, *cn = count + b * scb;
    const int oT = s.T * s1h, oB = s.Bm * s1h, cT = s.T * sch, cB = s.Bm * sch;
    atomic_add_f32(ox + oT + s.L, vx);  atomic_add_f32(ox + oT + s.R, vx);
    atomic_add_f32(ox + oB + s.L, vx);  atomic_add_f32(ox + oB + s.R, vx);
    atomic_add_f32(oy + oT + s.L, vy);  atomic_add_f32(oy + oT + s.R, vy);
    atomic_add_f32(oy + oB + s.L, vy);  atomic_add_f32(oy + oB + s.R, vy);
    atomic_add_f32(cn + cT + s.L, vc);  atomic_add_f32(cn + cT + s.R, vc);
    atomic_add_f32(cn + cB + s.L, vc);  atomic_add_f32(cn + cB + s.R, vc);
}

// --------------------------------------------------------------------------------------------------
// Pass 2: out /= count where count > 0 (my_lib_kernel.cu:1730-1735).
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proj_average(
    int W, int H, int tiles_x, int tiles_y, int64_t s1b, int64_t s1c, int s1h, int64_t scb, int sch,
    const float *__restrict__ count, float *__restrict__ out)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;
    const float t = count[b * scb + (int64_t)y * sch + x];
    if (t > 0.0f) {
        float *o = out + b * s1b + (int64_t)y * s1h + x;
        o[0] = o[0] / t;
        o[s1c] = o[s1c] / t;
    }
}

// --------------------------------------------------------------------------------------------------
// Pass 3 (fillhole != 0): every cell with count <= 0 takes the mean of the nearest non-empty cells to
// its left / right / above (my_lib_kernel.cu:1776-1832).  The reference's downward search never runs
// (`while (down_temp = 0.0f && ...)`, :1799): its term is 0 * out[own cell] and contributes nothing.
// Reads touch only cells with count != 0, writes only cells with count <= 0: race-free, deterministic.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void proj_fillhole(
    int W, int H, int tiles_x, int tiles_y, int64_t s1b, int64_t s1c, int s1h, int64_t scb, int sch,
    const float *__restrict__ count, float *out)
{
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;
    const float *cn = count + b * scb;
    if (!(cn[(int64_t)y * sch + x] <= 0.0f)) return;

    int lo = x;  float lt = 0.0f;
    while (lt == 0.0f && lo - 1 >= 0) { lo--; lt = cn[(int64_t)y * sch + lo]; }
    int ro = x;  float rt = 0.0f;
    while (rt == 0.0f && ro + 1 <= W - 1) { ro++; rt = cn[(int64_t)y * sch + ro]; }
    int uo = y;  float ut = 0.0f;
    while (ut == 0.0f && uo - 1 >= 0) { uo--; ut = cn[(int64_t)uo * sch + x]; }
    const float dt = 0.0f;                                  // dead downward search
    if (lt + rt + ut + dt <= 0.0f) return;
    const float fl = lt > 0.0f ? 1.0f : 0.0f, fr = rt > 0.0f ? 1.0f : 0.0f;
    const float fu = ut > 0.0f ? 1.0f : 0.0f, fd = 0.0f;
    float *o = out + b * s1b;
#pragma unroll
    for (int k = 0; k < 2; k++, o += s1c) {
        float *self = o + (int64_t)y * s1h + x;
        *self = (fl * o[(int64_t)y * s1h + lo] + fr * o[(int64_t)y * s1h + ro] +
                 fu * o[(int64_t)uo * s1h + x] + fd * *self) / (fl + fr + fu + fd);
    }
}

// --------------------------------------------------------------------------------------------------
// Backward (my_lib_kernel.cu:1866-1897 and :2296-2360): pure gather, no atomics.  Scalar fallback of
// proj_bwd_tiled: the reference's `+=` onto the caller's zero-filled buffer is a plain store here (each site
// owns its elements; four sequential terms per component, in the reference's order).
// --------------------------------------------------------------------------------------------------
template <bool DEPTH>
__global__ __launch_bounds__(256) void proj_bwd(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth, const float *__restrict__ count,
    const float *__restrict__ fwd_out, const float *__restrict__ gout,
    float *__restrict__ gin1, float *__restrict__ gin2, int x0)
{
    // (x0: first column served -- 0, or the first column behind a ragged width's whole quads)
    const unsigned tile = xcd_chunked_id(blockIdx.x, gridDim.x);
    const int tx = tile % tiles_x;
    const int ty = (tile / tiles_x) % tiles_y;
    const int b = tile / (tiles_x * tiles_y);
    const int x = x0 + tx * kWave + (threadIdx.x & (kWave - 1));
    const int y = ty * 4 + (threadIdx.x / kWave);
    if (x >= W || y >= H) return;

    const float *flow_p = flow + b * s1b + (int64_t)y * s1h + x;
    const float fx = ld_stream(flow_p);
    const float fy = ld_stream(flow_p + s1c);
    const BlSite s = bl_locate<false>(x, y, W, H, fx, fy);
    // every site owns its gradient elements: they are STORED (invalid sites store the zero the reference leaves in
    // the caller's zero-filled buffer), so the buffers need no memset beforehand -- same rule as proj_bwd_tiled
    if (!s.valid) {
#pragma unroll
        for (int k = 0; k < 2; k++) gin1[b * s1b + k * s1c + (int64_t)y * s1h + x] = 0.0f;
        if (DEPTH) gin2[b * sdb + (int64_t)y * sdh + x] = 0.0f;
        return;
    }
    const float *cn = count + b * scb;
    const float c00 = cn[s.T * sch + s.L], c01 = cn[s.T * sch + s.R];
    const float c10 = cn[s.Bm * sch + s.L], c11 = cn[s.Bm * sch + s.R];
    const int o00 = s.T * s1h + s.L, o01 = s.T * s1h + s.R, o10 = s.Bm * s1h + s.L, o11 = s.Bm * s1h + s.R;
    float d = 1.0f;
    if (DEPTH) d = ld_stream(depth + b * sdb + (int64_t)y * sdh + x);
    float gd = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const float *go = gout + b * s1b + k * s1c;
        float *gp = gin1 + b * s1b + k * s1c + (int64_t)y * s1h + x;
        const float g00 = go[o00], g01 = go[o01], g10 = go[o10], g11 = go[o11];
        float g = 0.0f;
        if (DEPTH) {
            g += -g00 * d / c00;  g += -g01 * d / c01;  g += -g10 * d / c10;  g += -g11 * d / c11;
            const float *fo = fwd_out + b * s1b + k * s1c;
            const float f = k ? fy : fx;
            gd += -g00 / c00 * (f - fo[o00]);
            gd += -g01 / c01 * (f - fo[o01]);
            gd += -g10 / c10 * (f - fo[o10]);
            gd += -g11 / c11 * (f - fo[o11]);
        } else {
            g += -g00 / c00;  g += -g01 / c01;  g += -g10 / c10;  g += -g11 / c11;
        }
        *gp = g;
    }
    if (DEPTH) gin2[b * sdb + (int64_t)y * sdh + x] = gd;
}

// ==================================================================================================
// Vectorised, LDS-tiled production kernels (W % 4 == 0, 16-B aligned geometry); the scalar kernels above
// remain as the fallback for odd widths / unaligned views and as measurement arms.
// ==================================================================================================

// Pass 1, tiled: a workgroup owns a 64x16 tile of SOURCE sites (4 per lane, flow read as dwordx4).
//
// A site adds the SAME value to its four target cells (L|R) x (T|B), so the splat factors into
//     P[T][L] += v                                   (one add per plane per site: the "point splat")
//     out[y][x] += sum_{dy,dx in {0,1}} wy * wx * P[y-dy][x-dx]        (a 2x2 box sum, a pure gather)
// with wx = 2 for (x == W-1, dx == 0) and 1 otherwise (the clamped right neighbour R = min(L+1, W-1) coincides
// with L on the last column, which the reference adds twice), likewise wy on the last row.  The point splat goes
// to LDS planes covering the bounding box of the tile's (T, L) points -- WITHOUT LDS atomics (ds_add_f32
// retires only ~0.35 lane-ops per clock per CU, measured): sites sharing a cell are serialised by tag
// arbitration (see below) -- and the box sum is evaluated when the box is flushed, once per cell, with
// row-coalesced global atomics (neighbouring tiles' boxes overlap).  Sites whose point falls outside the LDS
// budget scatter their 12 adds straight to global memory.
// ABL (measurement arms, results WRONG): 2 = no flush, 3 = no LDS adds.
template <bool DEPTH, int ABL>
__device__ __forceinline__ void proj_scatter_tile(
    unsigned blk, unsigned nblk, char *smem, int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out, const int *__restrict__ far_flag)
{
    constexpr int LX = 16;
    using G = TileGeom<LX>;
    using A = AccGeom<LX>;
    float *acc = reinterpret_cast<float *>(smem);
    int *bb = reinterpret_cast<int *>(smem + 4 * A::kPlane * 4);

    const TileCoord tc = strip_walk(blk, nblk, tiles_x, tiles_y, nblk / (tiles_x * tiles_y));
    const int b = tc.b, tile_x0 = tc.tx * G::kTW, tile_y0 = tc.ty * G::kTH;
    if (far_flag && far_flag[b % kFlagWords] == 0) return;   // this image was complete on the fast path
    const int x = tile_x0 + 4 * (threadIdx.x % LX), y = tile_y0 + threadIdx.x / LX;
    const bool inb = x < W && y < H;
    const int xs = min(x, W - 4), ys = min(y, H - 1);
    const float *flow_p = flow + b * s1b + (int64_t)ys * s1h + xs;
    const f32x4 fx4 = ld_stream4(flow_p), fy4 = ld_stream4(flow_p + s1c);
    f32x4 d4 = {1.f, 1.f, 1.f, 1.f};
    if (DEPTH) d4 = ld_stream4(depth + b * sdb + (int64_t)ys * sdh + xs);

    // zero the three value planes and set the tag plane to "free" (vector stores)
    {
        f32x4 *a4 = reinterpret_cast<f32x4 *>(acc);
        for (int i = threadIdx.x; i < 3 * A::kPlane / 4; i += G::kThreads) a4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        int *tag0 = reinterpret_cast<int *>(acc + 3 * A::kPlane);
        for (int i = threadIdx.x; i < A::kPlane; i += G::kThreads) tag0[i] = -1;
    }
    BlSite st[4];
    int cmin = INT_MAX, cmax = -1, rmin = INT_MAX, rmax = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        st[j] = bl_locate<false>(x + j, y, W, H, fx4[j], fy4[j]);
        st[j].valid = st[j].valid && inb;
        if (st[j].valid) {                     // box of the (T, L) points only
            cmin = min(cmin, st[j].L);  cmax = max(cmax, st[j].L);
            rmin = min(rmin, st[j].T);  rmax = max(rmax, st[j].T);
        }
    }
    // the flush also writes column cmax+1 / row rmax+1 (the R / B neighbours), which need no LDS cell
    const Region r = tile_region<LX>(cmin, cmax, rmin, rmax, tile_x0, tile_y0, bb);   // barrier: planes are ready

    float *ox = out + b * s1b, *oy = ox + s1c, *cn = count + b * scb;
    int *tag = reinterpret_cast<int *>(acc + 3 * A::kPlane);
    float vx[4], vy[4], vc[4];
    int cell[4];
    unsigned pending = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const BlSite &s = st[j];
        vx[j] = -fx4[j];  vy[j] = -fy4[j];  vc[j] = 1.0f;
        if (DEPTH) {
            vx[j] = -d4[j] * fx4[j];
            vy[j] = -d4[j] * fy4[j];
            vc[j] = d4[j] * 1.0f;
        }
        cell[j] = 0;
        if (!s.valid || ABL == 3) continue;
        if (r.covers(s.L, s.L, s.T, s.T)) {
            cell[j] = (s.T - r.y0) * A::kPitch + (s.L - r.x0);
            pending |= 1u << j;
        } else {                              // point clipped out of the LDS budget: straight to global
            const int oT = s.T * s1h, oB = s.Bm * s1h, cT = s.T * sch, cB = s.Bm * sch;
            atomic_add_f32(ox + oT + s.L, vx[j]);  atomic_add_f32(ox + oT + s.R, vx[j]);
            atomic_add_f32(ox + oB + s.L, vx[j]);  atomic_add_f32(ox + oB + s.R, vx[j]);
            atomic_add_f32(oy + oT + s.L, vy[j]);  atomic_add_f32(oy + oT + s.R, vy[j]);
            atomic_add_f32(oy + oB + s.L, vy[j]);  atomic_add_f32(oy + oB + s.R, vy[j]);
            atomic_add_f32(cn + cT + s.L, vc[j]);  atomic_add_f32(cn + cT + s.R, vc[j]);
            atomic_add_f32(cn + cB + s.L, vc[j]);  atomic_add_f32(cn + cB + s.R, vc[j]);
        }
    }
    // Point splat WITHOUT LDS atomics (ds_add_f32 retires ~0.35 lane-ops per clock per CU): sites that share a
    // cell are serialised by tag arbitration.  Per round every pending site writes its id into the cell's tag;
    // after a barrier exactly one of them reads its own id back -- the winner -- and does a plain
    // read-modify-write of the three value planes; the others stay pending.  Unique targets (the bulk) finish
    // in round 0; a cell hit by m sites takes m rounds.  After kRounds rounds the stragglers use atomics.
    constexpr int kRounds = 4;
    const int uid0 = threadIdx.x * 4;
#pragma unroll 1
    for (int round = 0; round < kRounds; round++) {
        if (round > 0 && !__syncthreads_or(pending != 0)) break;   // also orders the previous winners' writes
#pragma unroll
        for (int j = 0; j < 4; j++)
            if ((pending >> j) & 1) tag[cell[j]] = uid0 + j;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (((pending >> j) & 1) && tag[cell[j]] == uid0 + j) {
                acc[cell[j]] += vx[j];
                acc[A::kPlane + cell[j]] += vy[j];
                acc[2 * A::kPlane + cell[j]] += vc[j];
                pending &= ~(1u << j);
            }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; j++)
        if ((pending >> j) & 1) {             // more than kRounds sites on one cell (strongly compressive flow)
            lds_add_f32(acc + cell[j], vx[j]);
            lds_add_f32(acc + A::kPlane + cell[j], vy[j]);
            lds_add_f32(acc + 2 * A::kPlane + cell[j], vc[j]);
        }
    __syncthreads();
    if (ABL == 2 || r.w == 0) return;

    // flush: cells [x0, x0 + w] x [y0, y0 + h] (one extra column / row for the R / B neighbours), clipped to
    // the image; 64 lanes walk a row, 4 waves take rows round-robin.
    float *const dst[3] = {ox, oy, cn};
    const int hs[3] = {s1h, s1h, sch};
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int fw = min(r.w + 1, W - r.x0), fh = min(r.h + 1, H - r.y0);
    for (int row = wave; row < fh; row += G::kThreads / kWave) {
        const int gy = r.y0 + row;
        const bool up = row >= 1, here_y = row < r.h;
        const float wy0 = (gy == H - 1) ? 2.0f : 1.0f;         // dy == 0 term weight
        for (int col = lane; col < fw; col += kWave) {
            const int gx = r.x0 + col;
            const bool left = col >= 1, here_x = col < r.w;
            const float wx0 = (gx == W - 1) ? 2.0f : 1.0f;
#pragma unroll
            for (int pl = 0; pl < 3; pl++) {
                const float *p = acc + pl * A::kPlane + row * A::kPitch + col;
                // reference order of a cell's contributions is arbitrary (atomics); keep a fixed one here
                float v = 0.0f;
                if (here_y && here_x) v += wy0 * wx0 * p[0];
                if (here_y && left) v += wy0 * p[-1];
                if (up && here_x) v += wx0 * p[-A::kPitch];
                if (up && left) v += p[-A::kPitch - 1];
                if (v != 0.0f) atomic_add_f32(dst[pl] + (int64_t)gy * hs[pl] + gx, v);
            }
        }
    }
}

// One workgroup per tile, or -- queued behind proj_owner, where it normally has nothing to do -- a short grid
// whose workgroups stride over the tiles: leaving an idle grid of 2048 workgroups costs ~3 us, one of 28800 ~10.
template <bool DEPTH, int ABL>
__global__ __launch_bounds__(256) void proj_scatter_tiled(
    int W, int H, int tiles_x, int tiles_y, unsigned ntiles,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out, const int *__restrict__ far_flag)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (far_flag && far_flag[kFlagWords] == 0) return;   // queued behind proj_owner: nothing to redo at all
#pragma unroll 1
    for (unsigned blk = blockIdx.x; blk < ntiles; blk += gridDim.x) {
        proj_scatter_tile<DEPTH, ABL>(blk, ntiles, smem, W, H, tiles_x, tiles_y, s1b, s1c, s1h, sdb, sdh, scb, sch,
                                      flow, depth, count, out, far_flag);
        __syncthreads();                                 // the planes are rebuilt by the next tile
    }
}

// ==================================================================================================
// OWNER-COMPUTES forward (the fast path).  The splat is a scatter only because the reference walks SOURCE
// sites; flow is cheap to re-read (8 B per site), so here a workgroup owns a 64 x TH tile of OUTPUT cells and
// scans every source that can reach it:
//   * sources in the tile dilated by kReach (+ alignment) are read as dwordx4 (the halo comes out of L2);
//   * a source whose point (T, L) falls into the tile's point window [ty0-1, ty0+TH-1] x [tx0-1, tx0+63] is
//     splatted into fp64 LDS planes (ds_add_f64: fast on this chip, and more accurate than fp32 atomics);
//   * then every lane owns four cells: 2x2 box sum of the points (border duplicates as weights 2, see
//     proj_scatter_tiled), normalisation by the count, one dwordx4 STORE per plane.
// No global atomics, no separate averaging pass, no dependence on the caller's zero fill; HBM traffic is the
// algorithmic 20 B per site (24 with depth) plus the scan's halo.
//
// Reach: a source with |fx| >= kReach or |fy| >= kReach is invisible to the owners of its targets.  Such "far"
// sources are skipped consistently by every owner, and the workgroup that is HOME to one raises a device flag.
// The launcher queues proj_owner_far behind it (rounds 1-2: the general path), which returns at once when no flag is
// up and otherwise redoes the flagged images exactly: correctness never depends on the motion being small, only
// speed does.
//
// Round 2 (proj_owner2).  The round-1 kernel (kept as a measurement arm, proj_owner) scanned 7.6x the sources it
// owned and ran the whole per-source body -- locate, window test, three fp64 LDS atomics -- under exec masks with
// ~1 lane in 8 active: VALU-bound (74 % busy) with 151 LDS-atomic wave-instructions per tile at 22 lanes each.
//   * The scan is now a TEST: x2 = x + fx, y2 = y + fy and two compares per axis against wave-uniform bounds (the
//     upper one on the float's bit pattern, which folds "inside the image" and "inside the window" into one
//     integer compare for the non-negative values that passed the lower one).  Quads whose rows cannot reach the
//     tile leave after the four y tests (a scalar branch on a ballot).
//   * Hits are COMPACTED: each wave appends its hits (cell, vx, vy, vc: 16 B) to a private 128-entry LDS ring --
//     rank by v_mbcnt over the ballot, no atomics, no workgroup barrier -- and whenever 64 are waiting all 64 lanes
//     splat one each: dense fp64 atomics (1 wave-instruction per 64 lane-operations instead of per ~22).
//   * The tile is 64 x TH with TH a template parameter (256 / 512 / 1024 threads): a taller tile scans fewer
//     sources per owned cell (7.6x / 4.7x / 3.3x at reach 24) for more LDS (36 / 71 / 139 KiB).
//   * Tiles are walked in stripes `sw` tile columns wide per XCD (tile_walk), so that the horizontal halo of the
//     scan is an L2 hit on the same XCD instead of a second HBM read (strips: 1.41x the algorithmic traffic).
// ==================================================================================================
#ifdef MEMC_MEASURE
constexpr int kPtW = 68;                      // point-plane pitch of the measurement-build kernels (65 columns used)
#endif

// Per-workgroup phase timestamps (shader clock) for tools/trace_kernel.py; written by the TRACE instance only.
__device__ unsigned long long *g_trace_buf_proj = nullptr;
template <bool ON>
__device__ __forceinline__ void trace_mark_proj(int slot)
{
    if (ON && threadIdx.x == 0) g_trace_buf_proj[(size_t)blockIdx.x * 16 + slot] = __builtin_readcyclecounter();
}

#include "proj_fill.hpp"                  // pass 3: masks, the owner kernels' fill epilogue, proj_fill_pending

#ifdef MEMC_MEASURE
#define MEMC_PROJ_ARMS_PART_A
#include "arms/proj_owner_arms.hpp"      // proj_owner2, proj_owner3: superseded, measurement build only
#undef MEMC_PROJ_ARMS_PART_A
#endif  // MEMC_MEASURE

// --------------------------------------------------------------------------------------------------
// The production owner kernels: proj_owner5 (proj_owner5.hpp, round 4) and proj_owner_far below.  What led to their
// shape (rounds 2-3; the kernels named here live in arms/proj_owner_arms.hpp now).  proj_owner2's timing arms
//  showed that the kernel was bound by how many workgroups a CU holds -- the scan's loads are issued at the start of a
// workgroup's life and nothing is in flight while it tests, splats and reads out, so the bytes in flight per CU are
// (workgroups per CU) x 64 KB.  With 71 KiB of LDS (three fp64 planes + the compaction rings) two 64x32 tiles fit a
// CU: 240 us; the same kernel with two planes (53 KiB, three per CU): 183 us.  A persistent, software-pipelined form
// (proj_owner3, next tile's fy prefetched) LOST (285 us): on gfx9 a wave's loads and stores share one in-order
// counter, so the first wait of a tile also waits for the previous tile's stores.  Hence:
//   * no LDS rings: round 3's proj_owner4 compacted a wave's hits in REGISTERS (ds_permute_b32 pushes the hits of one
//     ballot to consecutive lanes of a cyclic 64-entry batch); round 4's proj_owner5 drops the compaction altogether
//     and splats under the exec mask -- the compaction was two thirds of the kernel's instructions;
//     proj_owner_far still uses the register batch (OwnerTile::source).  -16 KiB per workgroup either way.
//   * FlowProjection (count = number of sources, an integer) keeps TWO planes: A = count * 2^20 + sum(vx), B =
//     sum(vy).  At most (2 kReach + 1)^2 = 2401 sources can reach one point and |vx| < kReach, so |sum(vx)| < 2^19
//     splits off exactly (count = rint(A / 2^20)); a double holding count * 2^20 <= 2^32 still resolves 2^-20 px,
//     2^-32 px for the counts that actually occur -- finer than the fp32 atomics of the reference.  The depth
//     operator (count = sum of depths, not an integer) keeps three planes.
//   * point-plane pitch 66 (65 columns used): the depth operator's three planes then leave room for three
//     workgroups per CU (52.8 KiB), FlowProjection's two for four (35.3 KiB).
//
// Far sources (|f| >= kReach), round 2: the owner kernel also records, per tile, the largest |fx| and |fy| of its own FAR
// sources (in the cold branch that raises the flag: the hot loop pays nothing).  The images whose flag was raised are then redone by proj_owner_far -- the same owner-computes tile, but
// scanning whole SOURCE TILES, and only those whose recorded motion bound lets them reach the window: exact for any
// flow, no atomics, no zeroing or averaging pass, cost proportional to the actual motion.  One normally idle launch
// instead of the general path's three or four (each idle launch costs ~5 us + a 1.5 us boundary).
// --------------------------------------------------------------------------------------------------
constexpr int kPtW4 = 66;
constexpr double kCountUnit = 1048576.0;          // 2^20

// (tail_shift / tail_fix / st_tail4 -- ragged rows -- live in memc_tile.hpp)
// The image's dominant motion: mean flow over an 8 x 8 grid of sites, rounded to a multiple of 4 (quads stay quads),
// 0 for anything non-finite or absurd and below kMotionDeadZone.  One wave; lane l holds site l.  Deterministic: a butterfly of commutative adds.
__device__ __forceinline__ void motion_sample_issue(const float *flow_b, int64_t s1c, int s1h, int W, int H, int lane,
                                                    float &fxs, float &fys)
{
    const int xs = ((2 * (lane & 7) + 1) * W) >> 4, ys = ((2 * (lane >> 3) + 1) * H) >> 4;
    const float *p = flow_b + (int64_t)ys * s1h + xs;
    fxs = *p;
    fys = p[s1c];
}
#ifdef MEMC_MEASURE
// The same 64 samples straight into LDS (global_load_lds_dword: lane l's dword lands at dst[l], no vector register is held while
// the loads are in flight -- proj_owner5 has none to spare over its first scan iteration): fx at dst[0 .. 63], fy at dst[64 .. 127].
// Issued from inline assembly ON PURPOSE: told about an LDS-DMA load (__builtin_amdgcn_global_load_lds), the compiler's wait-count
// pass makes every later LDS access that "may alias" wait for it -- s_waitcnt vmcnt(0) in front of the plane zeroing, the first
// barrier and every splat, i.e. the very round trip this is here to take off the path.  Unknown to the compiler, the two loads
// are simply the OLDEST of the wave's outstanding vector-memory operations: loads return in order, so every count the compiler
// waits for still implies what it meant (never satisfied early, at most later), and the reader waits for them explicitly
// (motion_samples_wait).
__device__ __forceinline__ void motion_sample_issue_lds(const float *flow_b, int64_t s1c, int s1h, int W, int H, int lane, float *dst)
{
    const int xs = ((2 * (lane & 7) + 1) * W) >> 4, ys = ((2 * (lane >> 3) + 1) * H) >> 4;
    const float *p = flow_b + (int64_t)ys * s1h + xs;
    typedef __attribute__((address_space(3))) void lds_void;
    const unsigned lds = (unsigned)(uintptr_t)(lds_void *)dst;       // the destination's LDS byte address (wave-uniform)
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" : : "v"(p), "s"(lds) : "memory", "m0");
    asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dword %0, off" : : "v"(p + s1c), "s"(lds + 256u) : "memory", "m0");
}
// Before the samples are read: at most `younger` vector-memory operations -- those the wave issued AFTER the two sample loads and
// has not waited for yet; a smaller number is always safe -- may still be outstanding.
template <int younger>
__device__ __forceinline__ void motion_samples_wait()
{
    static_assert(younger >= 0 && younger < 64, "vmcnt has six bits");
    // gfx9 s_waitcnt: vmcnt [3:0] and [15:14], expcnt [6:4] = 7 (no wait), lgkmcnt [11:8] = 15 (no wait)
    __builtin_amdgcn_s_waitcnt((younger & 15) | ((younger >> 4) << 14) | (7 << 4) | (15 << 8));
    asm volatile("" ::: "memory");
}
#else                                          // (the product instantiates MOT = 0 only: the arms' helpers are stubs)
__device__ __forceinline__ void motion_sample_issue_lds(const float *, int64_t, int, int, int, int, float *) {}
template <int younger>
__device__ __forceinline__ void motion_samples_wait() {}
#endif
// A mean below kMotionDeadZone counts as none: a shifted scan costs ~15 us at 720p batch 32 whatever the shift (the loads
// requested for m = 0 before m was known are thrown away, the tile's own sources are tested on their own), and a shift of
// one quad buys nothing -- with m = 0 a source stays inside the scan up to 24 px, i.e. local motion of 18 px against a mean
// of 6 (a hand-held camera's drift: 122 instead of 137 us without, 164 instead of 179 us with hole filling at a pan of 3 px).
constexpr float kMotionDeadZone = 6.0f;
__device__ __forceinline__ int motion_round4(float sum)
{
    const float mean = sum * (1.0f / 64.0f);
    if (!(fabsf(mean) < 4096.0f) || fabsf(mean) < kMotionDeadZone) return 0;
    return 4 * (int)__builtin_rintf(mean * 0.25f);
}
// sum over the 64 lanes of a wave in a FIXED order (every workgroup must arrive at the same bits): an inclusive scan inside
// each row of 16 lanes (row_shr 1, 2, 4, 8 on the DPP path: a few cycles each -- ds_bpermute shuffles here put half a
// microsecond of LDS round trips in front of every workgroup's first barrier: +8 % on the whole kernel, measured), then
// the four row sums.
__device__ __forceinline__ float wave_sum_f32(float v)
{
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, true));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, true));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
    const float b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
    const float d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    return (a + b) + (c + d);
}
__device__ __forceinline__ void motion_reduce(float fxs, float fys, int &mx, int &my)
{
    mx = motion_round4(wave_sum_f32(fxs));
    my = motion_round4(wave_sum_f32(fys));
}
// The estimate through the SCALAR unit: 16 sites (a 4 x 4 grid), their addresses wave-uniform, so the loads are s_load_dword
// through the scalar cache and touch neither the texture path nor a vector register.  Round 6 measured what round 5's 64
// per-lane loads cost the benchmark's flow (profiles/r06_proj_motion_estimate_arms.txt): 2.7-3.4 % of the call, half of it
// gone with 16 lanes instead of 64 -- it is the 128 scattered cache lines per workgroup on the vector memory path, not the wait.
// Summed in a fixed order (every workgroup and proj_owner_far arrive at the same bits).  For ONE wave (uniform control flow).
__device__ __forceinline__ void motion_estimate_scalar(const float *__restrict__ flow_b, int64_t s1c, int s1h, int W, int H, int &mx, int &my)
{
    float sx = 0.0f, sy = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int xs = ((2 * (k & 3) + 1) * W) >> 3, ys = ((2 * (k >> 2) + 1) * H) >> 3;
        // (read through the CONSTANT address space: that is what makes the compiler select scalar loads; the flow tensor is an
        // input of the call, nothing writes it while this kernel runs)
        typedef const __attribute__((address_space(4))) float cfloat;
        cfloat *p = reinterpret_cast<cfloat *>(reinterpret_cast<uintptr_t>(flow_b + ((int64_t)ys * s1h + xs)));
        sx += p[0];
        sy += p[s1c];
    }
    mx = __builtin_amdgcn_readfirstlane(motion_round4(4.0f * sx));   // (motion_round4 divides by 64)
    my = __builtin_amdgcn_readfirstlane(motion_round4(4.0f * sy));
}
#ifdef MEMC_MEASURE
// (timing arm, proj_owner5 MOT = 3) sixteen samples -- a 4 x 4 grid, lanes 0 .. 15 -- instead of sixty-four
__device__ __forceinline__ void motion_sample_issue16(const float *flow_b, int64_t s1c, int s1h, int W, int H, int lane,
                                                      float &fxs, float &fys)
{
    if (lane < 16) {
        const int xs = ((2 * (lane & 3) + 1) * W) >> 3, ys = ((2 * (lane >> 2) + 1) * H) >> 3;
        const float *p = flow_b + (int64_t)ys * s1h + xs;
        fxs = *p;
        fys = p[s1c];
    }
}
__device__ __forceinline__ void motion_reduce16(float fxs, float fys, int &mx, int &my)
{
    mx = motion_round4(4.0f * wave_sum_f32(fxs));      // (lanes 16 .. 63 hold zeros; motion_round4 divides by 64)
    my = motion_round4(4.0f * wave_sum_f32(fys));
}
#else
__device__ __forceinline__ void motion_sample_issue16(const float *, int64_t, int, int, int, int, float &, float &) {}
__device__ __forceinline__ void motion_reduce16(float, float, int &, int &) {}
#endif

// Tiles the owner-computes fast path serves: proj_owner_far deals the stamped ones out from two tables (4 + 8 bytes per 64
// tiles) that live in its point planes' bytes; a call with more tiles takes the general path.
template <int TH>
constexpr int far_max_tiles()
{
    constexpr int plane_bytes = 3 * (TH + 1) * 66 * 8;
    constexpr int groups = (plane_bytes - 16) / 12 / 64 * 64;
    return 64 * (groups < kFarMaxTiles / 64 ? groups : kFarMaxTiles / 64);
}

// One owned 64 x TH tile of the far-source kernel: three fp64 point planes (without a limit on |flow| the sum of vx at a
// point is not bounded by 2^19, which the packed count * 2^20 + sum(vx) plane of proj_owner5 relies on), the window test of
// proj_owner5 (one subtract and one unsigned compare per axis on the bit patterns) and hits splatted straight under their
// exec mask.  (Rounds 2-3: OwnerTile with a register batch of waiting hits -- arms/proj_owner_arms.hpp.)
template <bool DEPTH, int TH>
struct FarTile {
    static constexpr int NP = 3;                  // planes: count, vx, vy
    static constexpr int kPitch = 66;             // columns tx0 - 1 .. tx0 + 64 (proj_owner5 pads to 68)
    static constexpr int kPlane = (TH + 1) * kPitch;
    static_assert(kPlane % 2 == 0, "P is zeroed 16 bytes at a time");
    double *P;
    int tx0, ty0, xlo_b, ylo_b;
    unsigned xrange, yrange, ucell8;

    __device__ __forceinline__ void begin(double *P_, int tx0_, int ty0_, int W, int H)
    {
        P = P_;  tx0 = tx0_;  ty0 = ty0_;
        xlo_b = __float_as_int((float)max(tx0 - 1, 0));
        ylo_b = __float_as_int((float)max(ty0 - 1, 0));
        xrange = (unsigned)(min(__float_as_int((float)(tx0 + 64)), __float_as_int((float)(W - 1)) + 1) - xlo_b);
        yrange = (unsigned)(min(__float_as_int((float)(ty0 + TH)), __float_as_int((float)(H - 1)) + 1) - ylo_b);
        ucell8 = (unsigned)-(8 * ((ty0 - 1) * kPitch + (tx0 - 1)));
    }
    template <int NT>
    __device__ __forceinline__ void zero(int tid) const
    {
        f32x4 *pz = reinterpret_cast<f32x4 *>(P);
        for (int i = tid; i < NP * kPlane / 2; i += NT) pz[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // a quad of sources (sx .. sx + 3, sy); dead quads (outside the image) carry a NaN row
    __device__ __forceinline__ void quad(float syf, float sxf, const f32x4 &fx4, const f32x4 &fy4, const f32x4 &d4) const
    {
        float y2[4];
        bool wy[4], rowany = false;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            y2[j] = syf + fy4[j];
            wy[j] = (unsigned)(__float_as_int(y2[j]) - ylo_b) < yrange;
            rowany = rowany || wy[j];
        }
        if (__builtin_amdgcn_ballot_w64(rowany) == 0) return;        // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float x2 = (sxf + (float)j) + fx4[j];                // (float)x + fx, as the reference rounds it
            if (wy[j] && (unsigned)(__float_as_int(x2) - xlo_b) < xrange) {
                const unsigned a = __umul24((unsigned)(int)y2[j], 8u * kPitch) + ((((unsigned)(int)x2) << 3) + ucell8);
                double *q = reinterpret_cast<double *>(reinterpret_cast<char *>(P) + a);
                const float d = DEPTH ? d4[j] : 1.0f;                  // my_lib_kernel.cu:2102-2114: v = -d * f, count += d
                lds_add_f64(q, (double)(d * 1.0f));
                lds_add_f64(q + kPlane, -(double)(DEPTH ? d * fx4[j] : fx4[j]));
                lds_add_f64(q + 2 * kPlane, -(double)(DEPTH ? d * fy4[j] : fy4[j]));
            }
        }
    }
    // After a barrier: the lane's four cells (cx .. cx + 3, cy) -- 2x2 box sums of the points of columns c-1 .. c+3,
    // rows cy-1 and cy (border duplicates as weights 2, see proj_scatter_tiled), normalised by the count.
    __device__ __forceinline__ void readout(int cx, int cy, int W, int H, f32x4 &ox, f32x4 &oy, f32x4 &oc) const
    {
        const double wy0 = (cy == H - 1) ? 2.0 : 1.0;
        typedef double f64x2 __attribute__((ext_vector_type(2)));
        const double *r0 = P + (cy - ty0) * kPitch + (cx - tx0);   // column offset a multiple of 4: 16-byte pairs
        float v[NP][4];
#pragma unroll
        for (int pl = 0; pl < NP; pl++) {
            const double *a = r0 + pl * kPlane, *c = a + kPitch;
            const f64x2 a01 = *reinterpret_cast<const f64x2 *>(a), a23 = *reinterpret_cast<const f64x2 *>(a + 2);
            const f64x2 c01 = *reinterpret_cast<const f64x2 *>(c), c23 = *reinterpret_cast<const f64x2 *>(c + 2);
            const double top[5] = {a01[0], a01[1], a23[0], a23[1], a[4]};
            const double bot[5] = {c01[0], c01[1], c23[0], c23[1], c[4]};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const double wx0 = (cx + j == W - 1) ? 2.0 : 1.0;
                v[pl][j] = (float)__builtin_fma(wy0, __builtin_fma(wx0, bot[j + 1], bot[j]), __builtin_fma(wx0, top[j + 1], top[j]));
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) {
            float v0 = v[0][j], v1 = v[1][j], v2 = v[2][j];
            if (v0 > 0.0f) {                   // my_lib_kernel.cu:1730-1735; one reciprocal for both components
                const float inv = DEPTH ? 1.0f / v0 : __builtin_amdgcn_rcpf(v0);   // as proj_owner5: a recomputed tile rounds like the others
                v1 = v1 * inv;
                v2 = v2 * inv;
            }
            oc[j] = v0;  ox[j] = v1;  oy[j] = v2;
        }
    }
};

// The images flagged by the owner kernel, completed exactly: the tiles of such an image in which a far source of another
// tile lands (the owner kernel recorded in bounds[], per tile, the box its far sources land in) are recomputed from whole
// source tiles -- those within kReach of the window and those whose far sources can land in it -- with no limit on |flow|.
// Queued behind the owner kernel as a short grid that strides over the tiles; returns at once when no flag was raised.
// Round 4 measured what this path cost: 4.0 ms for 32 x 720p as soon as one source per image moved 24 px -- 33x the fast
// path (profiles/r04_proj_motion_sweep_before.txt), for what is ordinary motion in 720p video.  It walked ALL source tiles
// of the image per target tile (460 scalar culling steps for ~9 candidates), one workgroup per CU, with round 3's
// register compaction per source.  Now: a tile is redone only if a far source of another tile lands in it (landing boxes
// instead of motion bounds: a fast object dirties the tiles it lands in, not its whole image); the culling is one source
// tile per lane (a ballot names the ones to scan; per wave, its own four rows of a near tile), FarTile's cheap test and
// direct splats, two workgroups per CU.
// MINW / NCAND: the product is <4, 16 TH> (4 waves per SIMD = two workgroups per CU).  Round 4 tried <6, 384> -- three per CU,
// which its LDS then allowed; 80 VGPRs, 76 bytes of private scratch per lane: 266 -> 373 us under a 40 px pan -- and saw wrong
// results next to it; round 5 rebuilt that arm from round 4's code and took it apart (DESIGN.md section 4f,
// tools/probes/far_spill_streams.py, profiles/r05_far_spill_*): the spill only TRIGGERS (the runtime allocates a queue's
// scratch at its first such dispatch, ~1.4 ms between two kernels of a call).  RAG: ragged rows (tail_fix above).
template <bool DEPTH, int TH, int kReach, int MINW = 4, int NCAND = 16 * TH, bool RAG = false>
__global__ __launch_bounds__(16 * TH, MINW) void proj_owner_far(

    int W, int H, int tiles_x, int tiles_y, int batch,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth,
    float *__restrict__ count, float *__restrict__ out, const int *__restrict__ far_flag,
    const int *__restrict__ bounds, const int *__restrict__ stamps, FillWs ws, int nonce_arg)
{
    const int nonce = nonce_arg ? nonce_arg : __builtin_amdgcn_readfirstlane(far_flag[kFlagWords + 1]);   // (see proj_owner5)
    using FT = FarTile<DEPTH, TH>;
    constexpr int NT = 16 * TH;
    __shared__ __attribute__((aligned(16))) double P[FT::NP * FT::kPlane];
    __shared__ union {                         // (the list is dead when the fill epilogue's masks come to life)
        unsigned cand[NCAND];                  // the source tiles to scan (bit 31: its far sources can land in the window; bit 30:
                                               // it has other sources, within reach of the window)
        FillLds<TH> fl;
    } u;
    unsigned *const cand = u.cand;
    FillLds<TH> &fl = u.fl;
    __shared__ int ncand;
    __shared__ int motion[2];                  // the image's dominant motion (proj_owner5.hpp), per recomputed tile
    constexpr int kGroups = far_max_tiles<TH>() / 64;   // 64-tile groups of the work list
    __shared__ unsigned mytiles[NT];           // this workgroup's tiles (one per lane at most: see the launcher's grid)
    // (the work list's tables live in the point planes' bytes: built before the first tile zeroes them)
    unsigned *const gprefix = reinterpret_cast<unsigned *>(P);                                  // [kGroups + 1]
    unsigned long long *const gmask = reinterpret_cast<unsigned long long *>(P) + (kGroups + 2) / 2;   // [kGroups]
    static_assert(sizeof(P) >= (kGroups + 2) * 4 + kGroups * 8, "the work list's tables fit the planes");
    if (far_flag[kFlagWords] != nonce) return;
    const unsigned per_image = (unsigned)tiles_x * tiles_y, ntiles = per_image * batch;
    const int tid = tid_now(), lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int wrow0 = 4 * wave;                // the wave's four rows of a source tile
    // Which tiles?  The owner kernel's result for a tile is complete unless a far source of ANOTHER tile lands in its
    // window (it scanned every source within kReach of the tile -- shifted by the image's motion --, and splatted whatever
    // landed, far or not): the tile that owns such a source stamped this call's nonce on the tiles its far sources'
    // landing box meets (itself included when the image's motion is not zero).  Every other tile keeps what the owner
    // kernel wrote -- outputs, summaries, masks.
    // Round 5: the stamped tiles are DEALT OUT.  Round 4 gave workgroup i the tiles i, i + grid, ... and let it recompute
    // whichever of those were stamped: with 3 % of the tiles stamped (the benchmark's flow twice as large) the slowest of
    // 512 workgroups found five, the average 0.9 -- the launch took five tiles' time.  Now every workgroup counts the
    // stamps of all tiles (one ballot per 64 tiles, a prefix over the groups: ~30 loads per lane, cold path only) and takes
    // the stamped tiles of rank i, i + grid, ...: the same work in max(1, n / grid) tiles' time.
    const unsigned ngroups = (ntiles + 63u) / 64u;                           // <= kGroups (launcher)
    {   // eight groups' stamps in flight per wave (one dependent round trip per group made this prologue most of the launch)
        constexpr int kIn = 8;
        for (unsigned g0 = wave * kIn; g0 < ngroups; g0 += (NT / kWave) * kIn) {
            int st[kIn];
#pragma unroll
            for (int k = 0; k < kIn; k++) {
                const unsigned t = (g0 + k) * 64u + lane;
                st[k] = stamps[t < ntiles ? t : 0];
            }
#pragma unroll
            for (int k = 0; k < kIn; k++) {
                const unsigned t = (g0 + k) * 64u + lane;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(t < ntiles && st[k] == nonce);
                if (lane == 0 && g0 + k < ngroups) {
                    gprefix[g0 + k] = (unsigned)__builtin_popcountll(m);
                    gmask[g0 + k] = m;
                }
            }
        }
    }
    __syncthreads();
    if (wave == 0) {                           // exclusive prefix over the groups: a run per lane, then across the lanes
        constexpr int kPer = kGroups / kWave;
        unsigned sum = 0;
#pragma unroll 1
        for (int k = 0; k < kPer; k++) {
            const unsigned g = lane * kPer + k;
            sum += g < ngroups ? gprefix[g] : 0u;
        }
        unsigned incl = sum;
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const unsigned up = __shfl_up(incl, off, kWave);
            incl += lane >= off ? up : 0u;
        }
        unsigned run = incl - sum;
#pragma unroll 1
        for (int k = 0; k < kPer; k++) {
            const unsigned g = lane * kPer + k;
            if (g < ngroups) {
                const unsigned c = gprefix[g];
                gprefix[g] = run;
                run += c;
            }
        }
        if (lane == kWave - 1) gprefix[ngroups] = incl;
    }
    __syncthreads();
    {                                          // lane k of the workgroup: the stamped tile of rank blockIdx.x + k gridDim.x
        const unsigned total = gprefix[ngroups];
        const uint64_t rank = blockIdx.x + (uint64_t)tid * gridDim.x;
        unsigned tile = ~0u;
        if (rank < total) {
            unsigned glo = 0, ghi = ngroups;   // gprefix[glo] <= rank < gprefix[glo + 1]
            while (ghi - glo > 1) {
                const unsigned mid = (glo + ghi) / 2;
                if (gprefix[mid] <= (unsigned)rank) glo = mid; else ghi = mid;
            }
            unsigned long long m = gmask[glo];
            for (unsigned k = (unsigned)rank - gprefix[glo]; k; k--) m &= m - 1;         // its k-th stamped tile
            tile = glo * 64u + (unsigned)__builtin_ctzll(m);
        }
        mytiles[tid] = tile;
    }
    __syncthreads();                           // (the tables are dead: the first tile zeroes the planes)
#pragma unroll 1
    for (int mine = 0; mine < NT; mine++) {
        const unsigned tile = (unsigned)__builtin_amdgcn_readfirstlane((int)mytiles[mine]);   // (uniform: scalar registers)
        if (tile == ~0u) break;
        const int b = tile / per_image, tx = (tile % per_image) % tiles_x, ty = (tile % per_image) / tiles_x;
        const int tx0 = tx * 64, ty0 = ty * TH;
        const float *flow_b = flow + b * s1b;
        const float *depth_b = DEPTH ? depth + b * sdb : nullptr;
        float msx = 0.0f, msy = 0.0f;          // the image's motion, as the owner kernel computed it (wave 0; posted below)
        if (wave == 0) motion_sample_issue(flow_b, s1c, s1h, W, H, lane, msx, msy);
        const int4 *boxes = reinterpret_cast<const int4 *>(bounds) + 2 * (int64_t)b * per_image;   // [2 * s]: tile s of the image
        // Can a far source of a tile land in this tile's window?  box: where the tile's far sources land (min x2, max x2,
        // min y2, max y2 as the owner kernel recorded them; max < 0: it has none).  The window takes x2 in
        // [tx0 - 1, tx0 + 64), y2 in [ty0 - 1, ty0 + TH) (FarTile::begin).
        auto far_hits = [&](const int4 &box) {
            return box.y >= 0 && __int_as_float(box.y) >= (float)(tx0 - 1) && __int_as_float(box.x) < (float)(tx0 + 64) &&
                   __int_as_float(box.w) >= (float)(ty0 - 1) && __int_as_float(box.z) < (float)(ty0 + TH);
        };
        FT t;
        t.begin(P, tx0, ty0, W, H);
        t.template zero<NT>(tid);
        if (wave == 0) {
            int mx, my;
            motion_reduce(msx, msy, mx, my);
            if (tid == 0) {
                motion[0] = mx;
                motion[1] = my;
            }
        }
        __syncthreads();
        const float mxf = (float)__builtin_amdgcn_readfirstlane(motion[0]), myf = (float)__builtin_amdgcn_readfirstlane(motion[1]);
        // Can a source that is NOT far -- it moves by the image's motion give or take less than kReach (+1: the rounding of
        // x + fx) -- from rows [y_lo, y_lo + y_n) of the tile column stx land in the window?
        auto near_hits = [&](int stx, int y_lo, int y_n) {
            const float sx0 = (float)(stx * 64) + mxf, sy0 = (float)y_lo + myf, r = (float)kReach;
            return sx0 + 64.0f + r >= (float)(tx0 - 1) && sx0 - r - 1.0f < (float)(tx0 + 64) &&
                   sy0 + (float)y_n + r >= (float)(ty0 - 1) && sy0 - r - 1.0f < (float)(ty0 + TH);
        };
        // The source tiles to scan: listed once per tile by the whole workgroup, one source tile per lane (a single round
        // trip to the table), then walked by every wave -- near tiles only by the waves whose own four rows of the tile can
        // reach the window, tiles whose far sources can land in it by all.  (A tile that recorded no source that is NOT far is
        // not scanned for its near sources: after a camera pan that is every tile.)
#pragma unroll 1
        for (unsigned s0 = 0; s0 < per_image; s0 += NCAND) {
            if (s0) __syncthreads();           // the previous round's list has been walked (images of more than NCAND tiles)
            if (tid == 0) ncand = 0;
            __syncthreads();
            {
                const unsigned sl = s0 + tid;
                bool c = false, fh = false, nr = false;
                if (tid < NCAND && sl < per_image) {
                    const int sty = sl / (unsigned)tiles_x, stx = sl - sty * tiles_x;
                    fh = far_hits(boxes[2 * sl]);
                    nr = boxes[2 * sl + 1].y != 0 && near_hits(stx, sty * TH, TH);   // (.y: the tile has sources that are not far)
                    c = fh || nr;
                }
                const unsigned long long m = __builtin_amdgcn_ballot_w64(c);
                if (m) {                       // the wave's entries, in lane order, behind one atomic
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&ncand, __builtin_popcountll(m));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (c) cand[base + __builtin_popcountll(m & ((1ull << lane) - 1))] = sl | (fh ? 0x80000000u : 0u) | (nr ? 0x40000000u : 0u);
                }
            }
            __syncthreads();
            const int n = ncand;
#pragma unroll 1
            for (int e0 = 0; e0 < n; e0 += kWave) {
            const unsigned ent = e0 + lane < n ? cand[e0 + lane] : 0u;
            bool use = false;
            if (e0 + lane < n) {
                const unsigned sl = ent & 0x3fffffffu;
                const int sty = sl / (unsigned)tiles_x, stx = sl - sty * tiles_x;
                use = (ent >> 31) != 0 || ((ent & 0x40000000u) != 0 && near_hits(stx, sty * TH + wrow0, 4));
            }
            unsigned long long todo = __builtin_amdgcn_ballot_w64(use);
#pragma unroll 1
            while (todo) {                     // wave-uniform.  kBatch tiles' loads in flight before the first splat: this
                constexpr int kBatch = 4;      // loop is a chain of L2 / HBM round trips otherwise
                f32x4 fxq[kBatch], fyq[kBatch], ddq[kBatch];
                float sxf[kBatch], syf[kBatch];
#pragma unroll
                for (int k = 0; k < kBatch; k++) {     // (past the last tile: dead slots -- a NaN row, loads of element 0)
                    const bool have = todo != 0;
                    const unsigned st = (unsigned)__builtin_amdgcn_readlane((int)ent, __builtin_ctzll(todo | (1ull << 63))) & 0x3fffffffu;
                    todo &= todo - 1;          // (0 stays 0)
                    const int sty = st / (unsigned)tiles_x, stx = st - sty * tiles_x;
                    const int sx = stx * 64 + 4 * (tid % 16), sy = sty * TH + tid / 16;   // one quad of sources per lane
                    const bool lv = have && sx < W && sy < H;
                    const int rq = RAG ? tail_shift(sx, W) : 0;    // (the same for every tile of the batch: sx % 64 is the lane's)
                    const unsigned off = lv ? 4u * (unsigned)(sy * s1h + sx - rq) : 0u;
                    fxq[k] = ld_cached4_u(flow_b, off);
                    fyq[k] = ld_cached4_u(flow_b + s1c, off);
                    if (DEPTH) ddq[k] = ld_cached4_u(depth_b, lv ? 4u * (unsigned)(sy * sdh + sx - rq) : 0u);
                    sxf[k] = (float)sx;
                    syf[k] = lv ? (float)sy : __int_as_float(0x7fc00000);
                }
#pragma unroll
                for (int k = 0; k < kBatch; k++) {
                    if (RAG) {                 // the row's last quad: rotated back, the sites past the row NaN (they hit nothing)
                        const int rq = tail_shift((int)sxf[k], W);
                        fxq[k] = tail_fix(fxq[k], rq, __int_as_float(0x7fc00000));
                        fyq[k] = tail_fix(fyq[k], rq, __int_as_float(0x7fc00000));
                        if (DEPTH) ddq[k] = tail_fix(ddq[k], rq, __int_as_float(0x7fc00000));
                    }
                    t.quad(syf[k], sxf[k], fxq[k], fyq[k], DEPTH ? ddq[k] : f32x4{1.f, 1.f, 1.f, 1.f});
                }
            }
            }
        }
        __syncthreads();                       // every wave's points are in P (and the list has been walked)
        fill_lds_init(fl, tid);                // (in the list's bytes; the epilogue's vote is the barrier before its first use)
        const int cx = tx0 + 4 * (tid % 16), cy = ty0 + tid / 16;
        const bool inb = cx < W && cy < H;     // (no early exit: barriers below)
        f32x4 ox, oy, oc;
        t.readout(cx, cy, W, H, ox, oy, oc);
        if (ws.up)                             // pass 3 follows: fill what the tile can, summaries, masks (proj_fill.hpp)
            owner_fill_epilogue<TH, NT>(fl, reinterpret_cast<float *>(P), ws, tid, b, tx, ty, W, H, tiles_x, tiles_y, inb, ox,
                                        oy, oc);
        if (inb) {
            float *o = out + b * s1b + (int64_t)cy * s1h + cx;
            const int rs = RAG ? tail_shift(cx, W) : 0;
            st_tail4<false>(o, ox, rs);
            st_tail4<false>(o + s1c, oy, rs);
            st_tail4<false>(count + b * scb + (int64_t)cy * sch + cx, oc, rs);
        }
        __syncthreads();                       // P and the masks are rebuilt by the next tile
    }
}

#include "proj_owner5.hpp"               // the production owner kernel

// A call recorded into a HIP graph is replayed with the kernel arguments it was recorded with: the per-call tag of the far
// flags then comes from the DEVICE, advanced by this one-lane kernel in front of the owner kernel -- with one tag for all replays
// the flags and stamps of earlier replays would stay "raised" and ever more tiles would be recomputed (exact, and ever slower).
// The counter is a device global of this module (one per GPU, never allocated, never recycled); round 5 kept it in the call's
// workspace, which torch's graph pool hands to later tensors of the same capture between two replays: the counter then read
// whatever those kernels had written, possibly the same value at every replay (round-5 review).  The kernels of the call read
// the tag from the workspace word this kernel writes.
__device__ unsigned g_proj_replay_counter = 0;
__global__ void proj_bump_nonce(int *far_flag)
{
    const unsigned n = atomicAdd(&g_proj_replay_counter, 1u) + 1u;
    // (sign bit set, as the host counter's tags: never one of the small non-negative numbers the tables hold; never -1)
    const unsigned tag = (n & 0x7fffffffu) | 0x80000000u;
    far_flag[kFlagWords + 1] = (int)(tag == 0xffffffffu ? 0x80000000u : tag);
}

#ifdef MEMC_MEASURE
#define MEMC_PROJ_ARMS_PART_C
#include "arms/proj_owner_arms.hpp"      // proj_owner4 and the carry filler (round 3's production set): measurement build only
#undef MEMC_PROJ_ARMS_PART_C
#endif  // MEMC_MEASURE

#ifdef MEMC_MEASURE
#define MEMC_PROJ_ARMS_PART_B
#include "arms/proj_owner_arms.hpp"      // proj_owner (round 1): superseded, measurement build only
#undef MEMC_PROJ_ARMS_PART_B
#endif  // MEMC_MEASURE

// general path, queued behind proj_owner: each kernel returns at once unless a far source was seen
__global__ __launch_bounds__(256) void proj_redo_zero(int W, int H, int64_t s1b, int64_t s1c, int s1h, int64_t scb,
                                                      int sch, int batch, float *__restrict__ count,
                                                      float *__restrict__ out, const int *__restrict__ far_flag)
{
    // far_flag == nullptr: unconditional (the general path when it runs on its own: the forward pass DEFINES count
    // and output, it does not rely on the caller's zero fill)
    if (far_flag && far_flag[kFlagWords] == 0) return;
    const int w4 = W / 4;
    const int64_t n = (int64_t)batch * H * w4;
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % w4) * 4, y = (int)((i / w4) % H), b = (int)(i / ((int64_t)w4 * H));
        if (far_flag && far_flag[b % kFlagWords] == 0) continue;
        float *o = out + b * s1b + (int64_t)y * s1h + x;
        *reinterpret_cast<f32x4u *>(o) = z;
        *reinterpret_cast<f32x4u *>(o + s1c) = z;
        *reinterpret_cast<f32x4u *>(count + b * scb + (int64_t)y * sch + x) = z;
    }
}

// the same for the scalar path (odd widths, unaligned views)
__global__ __launch_bounds__(256) void proj_zero_scalar(int W, int H, int batch, int64_t s1b, int64_t s1c, int s1h,
                                                        int64_t scb, int sch, float *__restrict__ count,
                                                        float *__restrict__ out)
{
    const int64_t n = (int64_t)batch * H * W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((int64_t)W * H));
        float *o = out + b * s1b + (int64_t)y * s1h + x;
        o[0] = 0.0f;
        o[s1c] = 0.0f;
        count[b * scb + (int64_t)y * sch + x] = 0.0f;
    }
}

// Pass 2, vectorised: out /= count where count > 0, four cells per lane.
__global__ __launch_bounds__(256) void proj_average_v4(
    int W, int H, int64_t s1b, int64_t s1c, int s1h, int64_t scb, int sch, int batch,
    const float *__restrict__ count, float *__restrict__ out, const int *__restrict__ far_flag)
{
    if (far_flag && far_flag[kFlagWords] == 0) return;
    const int w4 = W / 4;
    const int64_t n = (int64_t)batch * H * w4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int x = (int)(i % w4) * 4;
        const int y = (int)((i / w4) % H);
        const int b = (int)(i / ((int64_t)w4 * H));
        if (far_flag && far_flag[b % kFlagWords] == 0) continue;
        const f32x4 c = ld_cached4(count + b * scb + (int64_t)y * sch + x);
        if (!(c[0] > 0.0f || c[1] > 0.0f || c[2] > 0.0f || c[3] > 0.0f)) continue;
        float *o = out + b * s1b + (int64_t)y * s1h + x;
        f32x4 vx = ld_cached4(o), vy = ld_cached4(o + s1c);
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (c[j] > 0.0f) {
                vx[j] = vx[j] / c[j];
                vy[j] = vy[j] / c[j];
            }
        *reinterpret_cast<f32x4u *>(o) = vx;
        *reinterpret_cast<f32x4u *>(o + s1c) = vy;
    }
}

// Pass 3, vectorised scan: only lanes that find a hole (count <= 0) do the reference's walk.
__device__ __forceinline__ void fill_one_hole(int x, int y, int W, int H, int64_t s1c, int s1h, int sch,
                                              const float *cn, float *o)
{
    int lo = x;  float lt = 0.0f;
    while (lt == 0.0f && lo - 1 >= 0) { lo--; lt = cn[(int64_t)y * sch + lo]; }
    int ro = x;  float rt = 0.0f;
    while (rt == 0.0f && ro + 1 <= W - 1) { ro++; rt = cn[(int64_t)y * sch + ro]; }
    int uo = y;  float ut = 0.0f;
    while (ut == 0.0f && uo - 1 >= 0) { uo--; ut = cn[(int64_t)uo * sch + x]; }
    const float dt = 0.0f;                                  // dead downward search (my_lib_kernel.cu:1799)
    if (lt + rt + ut + dt <= 0.0f) return;
    const float fl = lt > 0.0f ? 1.0f : 0.0f, fr = rt > 0.0f ? 1.0f : 0.0f;
    const float fu = ut > 0.0f ? 1.0f : 0.0f, fd = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; k++, o += s1c) {
        float *self = o + (int64_t)y * s1h + x;
        *self = (fl * o[(int64_t)y * s1h + lo] + fr * o[(int64_t)y * s1h + ro] +
                 fu * o[(int64_t)uo * s1h + x] + fd * *self) / (fl + fr + fu + fd);
    }
}

// count-plane reader for the walks: LDS copy of the workgroup's neighbourhood when the cell is inside it,
// global memory otherwise (a walk that leaves the neighbourhood is rare and simply continues there)
struct CountView {
    const float *lds;      // [kFhRows][kFhPitch] floats
    const float *glob;     // count plane of this image
    int x0, y0, w, h, sch; // staged neighbourhood (image coordinates) and the global row stride
    __device__ __forceinline__ float at(int y, int x) const
    {
        const int ly = y - y0, lx = x - x0;
        if ((unsigned)ly < (unsigned)h && (unsigned)lx < (unsigned)w) return lds[ly * 80 + lx];
        return glob[(int64_t)y * sch + x];
    }
};
constexpr int kFhPitch = 80, kFhRows = 24;     // 64x16 tile + 8 columns left/right, 8 rows above (no downward search)

__device__ __forceinline__ void fill_one_hole_lds(int x, int y, int W, int H, int64_t s1c, int s1h,
                                                  const CountView &cv, float *o)
{
    int lo = x;  float lt = 0.0f;
    while (lt == 0.0f && lo - 1 >= 0) { lo--; lt = cv.at(y, lo); }
    int ro = x;  float rt = 0.0f;
    while (rt == 0.0f && ro + 1 <= W - 1) { ro++; rt = cv.at(y, ro); }
    int uo = y;  float ut = 0.0f;
    while (ut == 0.0f && uo - 1 >= 0) { uo--; ut = cv.at(uo, x); }
    const float dt = 0.0f;                                  // dead downward search (my_lib_kernel.cu:1799)
    if (lt + rt + ut + dt <= 0.0f) return;
    const float fl = lt > 0.0f ? 1.0f : 0.0f, fr = rt > 0.0f ? 1.0f : 0.0f;
    const float fu = ut > 0.0f ? 1.0f : 0.0f, fd = 0.0f;
#pragma unroll
    for (int k = 0; k < 2; k++, o += s1c) {
        float *self = o + (int64_t)y * s1h + x;
        *self = (fl * o[(int64_t)y * s1h + lo] + fr * o[(int64_t)y * s1h + ro] +
                 fu * o[(int64_t)uo * s1h + x] + fd * *self) / (fl + fr + fu + fd);
    }
}

// Pass 3, tiled: every hole's walk is a chain of DEPENDENT reads of `count` (each an L2 round trip from global
// memory); the workgroup therefore copies the count cells around its 64x16 tile into LDS first and walks there.
__global__ __launch_bounds__(256) void proj_fillhole_v4(
    int W, int H, int tiles_x, int tiles_y, int64_t s1b, int64_t s1c, int s1h, int64_t scb, int sch,
    const float *__restrict__ count, float *out, int skip_fill /* measurement arm: detect holes, fill none */)
{
    __shared__ __attribute__((aligned(16))) float cnt_lds[kFhRows * kFhPitch];
    const TileCoord tc = strip_walk(blockIdx.x, gridDim.x, tiles_x, tiles_y, gridDim.x / (tiles_x * tiles_y));
    const int b = tc.b, tile_x0 = tc.tx * 64, tile_y0 = tc.ty * 16;
    const float *cn = count + b * scb;
    // most tiles have no hole at all (0.5 % of the cells, 13 % of the tiles on the benchmark's smooth flow): every
    // lane reads its own four cells (exactly the compulsory 4 B/site, unconditional at a clamped address) and the
    // workgroup leaves at once unless somebody saw a hole -- only then is the neighbourhood staged for the walks
    const int x = tile_x0 + 4 * (threadIdx.x % 16), y = tile_y0 + threadIdx.x / 16;
    const bool inb = x < W && y < H;
    const f32x4 own = ld_cached4(cn + (int64_t)min(y, H - 1) * sch + min(x, W - 4));
    const bool hole = inb && (own[0] <= 0.0f || own[1] <= 0.0f || own[2] <= 0.0f || own[3] <= 0.0f);
    if (!__syncthreads_or(hole) || skip_fill) return;
    CountView cv;
    cv.lds = cnt_lds;  cv.glob = cn;  cv.sch = sch;
    cv.x0 = max(tile_x0 - 8, 0);
    cv.y0 = max(tile_y0 - 8, 0);
    cv.w = min(tile_x0 + 64 + 8, W) - cv.x0;                // multiple of 4 (W % 4 == 0)
    cv.h = min(tile_y0 + 16, H) - cv.y0;
    const int wq = cv.w / 4;
    for (int i = threadIdx.x; i < cv.h * wq; i += 256) {
        const int row = i / wq, q = i % wq;
        *reinterpret_cast<f32x4 *>(cnt_lds + row * kFhPitch + 4 * q) =
            ld_cached4(cn + (int64_t)(cv.y0 + row) * sch + cv.x0 + 4 * q);
    }
    // A hole costs three dependent walks and six scattered reads; holes come in clusters, so left to the lanes
    // that own them a few lanes would do four of those chains in a row while the rest of the workgroup idles.
    // They are listed in LDS instead and dealt out one per lane.
    __shared__ int n_holes;
    __shared__ unsigned short hole_list[1024];
    if (threadIdx.x == 0) n_holes = 0;
    __syncthreads();                                        // count staged, list empty
    if (hole) {
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (own[j] <= 0.0f) hole_list[atomicAdd(&n_holes, 1)] = (unsigned short)(((y - tile_y0) << 6) | (x + j - tile_x0));
    }
    __syncthreads();
    const int n = n_holes;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int cell = hole_list[i];
        fill_one_hole_lds(tile_x0 + (cell & 63), tile_y0 + (cell >> 6), W, H, s1c, s1h, cv, out + b * s1b);
    }
}

// Backward, tiled: the four corner reads of gradoutput / count (/ forward output) come from an LDS image of
// the tile's target box -- one pixel quad (gx, gy, count, ox) per cell, plus a planar oy for the depth
// operator -- instead of 8..16 scattered global loads per site.
// RAG: a ragged width (W % 4 != 0, round 5) -- the whole quads (sites x < W & ~3) here, with the image's true width in every
// clamp and in the staged box (ragged-safe loads of a row's last quad, memc_tile.hpp); the columns behind them on proj_bwd.
template <bool DEPTH, int CAP, bool RAG = false>
__global__ __launch_bounds__(256) void proj_bwd_tiled(
    int W, int H, int tiles_x, int tiles_y,
    int64_t s1b, int64_t s1c, int s1h, int64_t sdb, int sdh, int64_t scb, int sch,
    const float *__restrict__ flow, const float *__restrict__ depth, const float *__restrict__ count,
    const float *__restrict__ fwd_out, const float *__restrict__ gout,
    float *__restrict__ gin1, float *__restrict__ gin2, int sw)
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
    const float *flow_p = flow + b * s1b + (int64_t)ys * s1h + xs;
    const f32x4 fx4 = ld_stream4(flow_p), fy4 = ld_stream4(flow_p + s1c);
    f32x4 d4 = {1.f, 1.f, 1.f, 1.f};
    if (DEPTH) d4 = ld_stream4(depth + b * sdb + (int64_t)ys * sdh + xs);
    // every site owns its gradient elements and the caller zero-fills the buffers (FlowProjectionLayer.py:54): they
    // are STORED once (invalid sites store the zero they already hold) instead of read, added to and written
    float *g1p = gin1 + b * s1b + (int64_t)ys * s1h + xs;
    f32x4 acc_x = {0.f, 0.f, 0.f, 0.f}, acc_y = acc_x, acc_d = acc_x;
    float *g2p = DEPTH ? gin2 + b * sdb + (int64_t)ys * sdh + xs : nullptr;

    BlSite st[4];
    int cmin = INT_MAX, cmax = -1, rmin = INT_MAX, rmax = -1;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        st[j] = bl_locate<false>(x + j, y, W, H, fx4[j], fy4[j]);
        st[j].valid = st[j].valid && inb;
        if (st[j].valid) {
            cmin = min(cmin, st[j].L);  cmax = max(cmax, st[j].R);
            rmin = min(rmin, st[j].T);  rmax = max(rmax, st[j].Bm);
        }
    }
    Region r = tile_region<LX, true, CAP>(cmin, cmax, rmin, rmax, tile_x0, tile_y0, bb);
    r.wimg = RAG ? W : 0;
    const float *go = gout + b * s1b, *cn = count + b * scb, *fo = DEPTH ? fwd_out + b * s1b : nullptr;
    // Every use of a corner is gout / count (times something of the site) or the forward output: the staged pixel
    // quad is therefore (gout_x / count, gout_y / count, out_x, out_y) -- the divisions (~12 VALU instructions
    // each; this kernel was VALU-bound on sixteen of them per site) are done once per staged cell, four
    // components hold everything (no separate plane for out_y: 48 instead of 61 KiB of LDS, 3 workgroups per CU),
    // and a site is eight or sixteen FMAs.  (g / c) * d differs from the reference's (g * d) / c by <= 1 ulp.
    {
        constexpr int NP = DEPTH ? 5 : 3;
        const StageSlot sl = stage_slots(r);
        StageRegs<NP> sr;
        if constexpr (DEPTH) {
            const float *const planes[5] = {go, go + s1c, cn, fo, fo + s1c};
            const int hs[5] = {s1h, s1h, sch, s1h, s1h};
            tile_stage_load_planes<5, RAG>(r, sl, planes, hs, sr);
        } else {
            const float *const planes[3] = {go, go + s1c, cn};
            const int hs[3] = {s1h, s1h, sch};
            tile_stage_load_planes<3, RAG>(r, sl, planes, hs, sr);
        }
#pragma unroll
        for (int it = 0; it < kStageIts; it++) {
            if (sl.row[it] < r.h) {
                f32x4 *dst = tile + sl.row[it] * r.pitch;
                if (RAG) {                     // the row's last quad was loaded to END at the row's end: rotated back
                    const int rs = tail_shift(r.x0 + 4 * sl.q[it], r.wimg);
#pragma unroll
                    for (int c = 0; c < NP; c++) sr.v[it][c] = tail_fix(sr.v[it][c], rs, 0.0f);
                }
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const float inv = 1.0f / sr.v[it][2][i];       // cells nobody projected to: inf, never read
                    f32x4 px = {sr.v[it][0][i] * inv, sr.v[it][1][i] * inv, 0.f, 0.f};
                    if (DEPTH) {
                        px[2] = sr.v[it][NP - 2][i];
                        px[3] = sr.v[it][NP - 1][i];
                    }
                    dst[swz_col(4 * sl.q[it] + i)] = px;
                }
            }
        }
    }
    __syncthreads();
    if (!inb) return;

#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!st[j].valid) continue;
        const BlSite &s = st[j];
        f32x4 q[4];          // (gx / count, gy / count, ox, oy) at TL, TR, BL, BR
        if (r.covers(s.L, s.R, s.T, s.Bm)) {
            const int rT = (s.T - r.y0) * r.pitch, rB = (s.Bm - r.y0) * r.pitch;
            const int cL = swz_col(s.L - r.x0), cR = swz_col(s.R - r.x0);
            q[0] = tile[rT + cL];  q[1] = tile[rT + cR];  q[2] = tile[rB + cL];  q[3] = tile[rB + cR];
        } else {             // rare: the corner is outside the staged box
            const int o[4] = {s.T * s1h + s.L, s.T * s1h + s.R, s.Bm * s1h + s.L, s.Bm * s1h + s.R};
            const int c[4] = {s.T * sch + s.L, s.T * sch + s.R, s.Bm * sch + s.L, s.Bm * sch + s.R};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const float inv = 1.0f / cn[c[k]];
                q[k] = f32x4{go[o[k]] * inv, go[s1c + o[k]] * inv, DEPTH ? fo[o[k]] : 0.f, DEPTH ? fo[s1c + o[k]] : 0.f};
            }
        }
        float gx = 0.0f, gy = 0.0f, gd = 0.0f;
        if (DEPTH) {
            const float d = d4[j];
#pragma unroll
            for (int k = 0; k < 4; k++) gx += -q[k][0] * d;
#pragma unroll
            for (int k = 0; k < 4; k++) gy += -q[k][1] * d;
#pragma unroll
            for (int k = 0; k < 4; k++) gd += -q[k][0] * (fx4[j] - q[k][2]);
#pragma unroll
            for (int k = 0; k < 4; k++) gd += -q[k][1] * (fy4[j] - q[k][3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) gx += -q[k][0];
#pragma unroll
            for (int k = 0; k < 4; k++) gy += -q[k][1];
        }
        acc_x[j] = gx;  acc_y[j] = gy;  acc_d[j] = gd;
    }
    *reinterpret_cast<f32x4u *>(g1p) = acc_x;
    *reinterpret_cast<f32x4u *>(g1p + s1c) = acc_y;
    if (DEPTH) *reinterpret_cast<f32x4u *>(g2p) = acc_d;
}

MEMC_KNOB_STATIC(g_proj_variant, -1);          // measurement build only (memc_common.hpp)
MEMC_KNOB_STATIC(g_proj_scratch_blocks, kBlocks);   // measurement build: how many cached scratch blocks a call may look at
MEMC_KNOB_STATIC(g_proj_stall_us, 0);          // measurement build: idle this long between the owner kernel and what follows it

#ifdef MEMC_MEASURE
// What a queue's first dispatch with private scratch does to a call (round 4's spilling far kernel: the runtime allocates the
// queue's scratch, ~1.4 ms): the call's later kernels start LATE.  Injected here to test that nothing the call relies on
// can be touched by another stream's call in the gap (tests/test_gpu_workspace_and_streams.py).
__global__ void proj_stall(int us)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                       // (the 100 MHz real-time counter)
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100ull) __builtin_amdgcn_s_sleep(32);
}
#endif

// Owner kernel geometry of the product build (measured): tile height and stripe width of the walk.
constexpr int kOwnerTH = 32, kOwnerSW = 4;

struct ProjArgs {
    hipStream_t stream;
    int w, h, batch, fillhole;
    int s1b, s1c, s1h, sdb, sdh, scb, sch;
    const float *flow, *depth;
    float *count, *out;
    void *ws;                                    // caller's workspace (the _ws entry points) or nullptr: the library's own block
    size_t ws_bytes;
};

// The call's scratch (library block or caller's workspace), in ints: [0, kHead) far flags (image b -> word b % 256, word
// 256 = "any"), the tiles' far table (kFarWords ints per tile: proj_owner5.hpp), then -- with hole filling -- the three tables
// of per-tile summaries, the per-tile hole flags and (8-byte aligned) the tiles' masks.
struct ProjWsLayout {
    size_t n_bnd, n_up, n_row, ints, mask_words;
    size_t bytes() const { return ints * sizeof(int) + mask_words * 8; }
};
#ifdef MEMC_MEASURE
constexpr size_t kProjWsHead = kMotionCacheAt + 2 * kMotionCacheWords;   // far flags (+ summary, replay tag) + the motion cache arm's words
#else
constexpr size_t kProjWsHead = 320;            // far flags (+ summary, replay tag)
#endif
template <int TH>
static ProjWsLayout proj_ws_layout(int w, int h, int batch, bool fast, bool carry, bool masks)
{
    const int ntx = (w + 63) / 64, nty = (h + TH - 1) / TH;
    const size_t ntiles = (size_t)ntx * nty * batch;
    ProjWsLayout l;
    l.n_bnd = fast ? kFarWords * ntiles + (ntiles + 3) / 4 * 4 : 0;        // the tiles' far table, then their stamps (dense)
    l.n_up = (size_t)batch * nty * w;
    l.n_row = (size_t)batch * h * ntx;
    l.ints = kProjWsHead + l.n_bnd + (carry ? l.n_up + 2 * l.n_row + ntiles : 0);
    l.ints = (l.ints + 1) / 2 * 2;
    l.mask_words = 0;
    if constexpr (TH <= 32) l.mask_words = carry && masks ? ntiles * tile_mask_words<TH>() : 0;
    return l;
}

// The far-source flags carry a per-call nonce instead of being cleared: ONE process-wide counter for every
// instantiation of run_proj_fwd (a function-local static would give FlowProjection and DepthFlowProjection their own
// counters, and two calls handed the same scratch block could carry the same nonce: a needless whole-image redo).
static std::atomic<unsigned> g_proj_call_counter{0};

// vectorised forward: owner-computes fast path (proj_owner5; images with a far source redone by proj_owner_far behind a
// device flag), hole filling from masks (proj_fill.hpp), with the tile height TH of the owner kernel and the filler;
// without scratch the general path (zero, scatter with atomics, average) and the literal hole walker.
// variant: measurement build only (-1 otherwise).
// RAG: a width that is not a multiple of four (the owner kernels' ragged-row instantiations; returns 1 -- not served, the
// caller takes the scalar kernels -- where the fast path cannot run: no scratch block, a plane beyond 4 GiB).
template <bool DEPTH, int TH, bool RAG = false>
static int run_proj_fwd(const ProjArgs &a, int sw, int variant)
{
    using A = AccGeom<16>;
    const hipStream_t stream = a.stream;
    const int w = a.w, h = a.h, batch = a.batch;
    const int ntx = (w + 63) / 64, nty = (h + TH - 1) / TH;
    const unsigned ntiles = (unsigned)ntx * nty * batch;
    const int snty = (h + 15) / 16;                          // the general path scatters from 64x16 SOURCE tiles
    const unsigned sntiles = (unsigned)ntx * snty * batch;
    const unsigned gs = 256 * 8;                             // grid-stride: 8 workgroups per CU
    const int64_t s1b = a.s1b, s1c = a.s1c, sdb = a.sdb, scb = a.scb;
    const int s1h = a.s1h, sdh = a.sdh, sch = a.sch;

    // which set of kernels (measurement build: the rounds 1-3 arms keep their own summaries and filler)
    bool r3_set = false;                                     // proj_owner4 / proj_owner_far_r3 / proj_fillhole_carry
    bool legacy_owner = false;                               // proj_owner, proj_owner2, proj_owner3 (+ general path behind the flag)
#ifdef MEMC_MEASURE
    r3_set = variant == -40 || variant == -20;
    legacy_owner = variant == -10 || variant == -7 || variant == -6 || variant == -30 || variant == -31 ||
                   (variant <= -21 && variant >= -29);
#endif
    constexpr bool kNewOk = TH <= 32;                        // a column mask of proj_fill.hpp is one 32-bit word
    if (!kNewOk && !legacy_owner) return -1;
    const bool old_fill = r3_set || legacy_owner;

    // (the owner kernel addresses the flow / depth planes with 32-bit offsets)
    const bool want_fast = variant != 1 && variant < 2 && plane_fits_u32(w, h, {s1h, sdh}) && ntiles <= (unsigned)far_max_tiles<TH>();
    const bool want_carry = a.fillhole && variant != -8 && variant != -9;
    // scratch layout: ProjWsLayout above
    constexpr size_t kHead = kProjWsHead;
    const ProjWsLayout lay = proj_ws_layout<TH>(w, h, batch, want_fast, want_carry, !old_fill);
    const size_t n_bnd = lay.n_bnd, n_up = lay.n_up, n_row = lay.n_row, ints = lay.ints;
    CallScratch scratch;
    int *flag = nullptr, *bounds = nullptr, *stamps = nullptr;
    FillWs ws = {nullptr, nullptr, nullptr, nullptr, nullptr};
    // The production pair (proj_owner5 / proj_owner_far) needs no cleared flag words: a flag is "raised" when it holds
    // this call's nonce -- a process-wide counter, never 0, so consecutive calls (which the pool hands the same block)
    // never see each other's flags; a stale or uninitialised word equal to the nonce would only cause a needless redo (and
    // cannot be one of the small numbers the tables hold: the tag's sign bit is set).  That saves a 5 us memset launch per call.  The measurement build's older kernels keep 0 / 1
    // flags and the memset.
    unsigned nonce_u = g_proj_call_counter.fetch_add(1, std::memory_order_relaxed) + 1u;
    if (nonce_u == 0) nonce_u = g_proj_call_counter.fetch_add(1, std::memory_order_relaxed) + 1u;
    // (sign bit set: the block is reused across calls of different shapes and holds row / column indices, hole flags, -1 --
    // a small sequential tag could meet one of those in a stale word and cause a needless recomputation)
    int nonce = (int)((nonce_u & 0x7fffffffu) | 0x80000000u);
    if (nonce == -1) nonce = (int)0x80000000u;
    if (a.ws) {                                              // (only a caller's workspace can be inside a capture at all)
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &capture) != hipSuccess) (void)hipGetLastError();
        if (capture != hipStreamCaptureStatusNone) nonce = 0;            // the device counter: proj_bump_nonce
    }
    // A caller's workspace (the _ws entry points: memc_flow_projection_workspace_bytes says how much) replaces the library's
    // block -- nothing is allocated, nothing is kept, a stream capture takes the same kernels as an eager call.
    if (a.ws && (a.ws_bytes < lay.bytes() || (reinterpret_cast<uintptr_t>(a.ws) & 15u) != 0)) return -1;
    void *block = nullptr;
    if (want_fast || want_carry) {
        if (a.ws) block = a.ws;
        else if (scratch.alloc(lay.bytes(), stream, g_proj_scratch_blocks)) block = scratch.p;
    }
    if (block) {
        int *base = static_cast<int *>(block);
        if (legacy_owner && hipMemsetAsync(base, 0, kHead * sizeof(int), stream) != hipSuccess) return -1;
        if (want_fast) {
            flag = base;
            bounds = base + kHead;
            stamps = bounds + kFarWords * (size_t)ntiles;
        }
        if (want_carry) {
            ws.up = base + kHead + n_bnd;
            ws.left = ws.up + n_up;
            ws.right = ws.left + n_row;
            ws.hole = ws.right + n_row;
            ws.masks = reinterpret_cast<unsigned long long *>(base + ints);
        }
    }
    // Without scratch (inside a stream capture, or the allocation failed): the general path on its own and the
    // literal hole walker -- slower, same results.
    if (RAG && !(flag && (ws.up || !a.fillhole))) return 1;  // (the general path's kernels want whole quads: scalar kernels instead)

#define MEMC_PROJ_SCATTER(ABL, FLAG)                                                                        \
    hipLaunchKernelGGL((proj_scatter_tiled<DEPTH, ABL>), dim3((FLAG) != nullptr && sntiles > gq ? gq : sntiles), \
                       dim3(256), 4 * A::kPlane * 4 + 64, stream, w, h, ntx, snty, sntiles, s1b, s1c, s1h, sdb, sdh,  \
                       scb, sch, a.flow, a.depth, a.count, a.out, FLAG)
    bool only_part = false, skip_pending = false;            // measurement arms that time one piece
    MEMC_PATH(flag ? (DEPTH ? "dproj_fwd:owner" : "proj_fwd:owner") : (DEPTH ? "dproj_fwd:general" : "proj_fwd:general"));
    // waves per SIMD the register allocator must leave room for = what the LDS admits: FlowProjection 4 workgroups per
    // CU at TH = 32 (2 planes, 35 KiB), the depth operator 3 (53 KiB)
    constexpr int kWgCu = TH == 16 ? (DEPTH ? 4 : 6) : (TH == 32 ? (DEPTH ? 3 : 4) : 1);
    constexpr int kMinW = (kWgCu * (16 * TH / 64) + 3) / 4 > 8 ? 8 : (kWgCu * (16 * TH / 64) + 3) / 4;
    // (the ragged-row instantiation needs a few registers more: two waves per SIMD fewer rather than a spill)
    [[maybe_unused]] constexpr int kMinWR = RAG && kMinW > 2 ? kMinW - 2 : kMinW;
    if (flag && !legacy_owner) {
        if constexpr (kNewOk) {
            if (nonce == 0) hipLaunchKernelGGL(proj_bump_nonce, dim3(1), dim3(1), 0, stream, flag);
            WalkPlan plan = make_walk_plan(ntx, nty, batch, sw);
#ifdef MEMC_MEASURE
            if (variant == -43) plan.fast = 0;      // test arm: the kernel's own tile_walk (what grids beyond n * d < 2^32 take)
#endif
#ifdef MEMC_MEASURE
            only_part = variant == -5 || variant == -20 || variant == -41;
            skip_pending = variant == -42;          // timing arm: everything but proj_fill_pending (pending holes stay unfilled)
            if (r3_set) {
                hipLaunchKernelGGL((proj_owner4<DEPTH, TH, 24, kMinW>), dim3(walk_grid(ntx, nty, batch, sw)), dim3(16 * TH),
                                   0, stream, w, h, ntx, nty, s1b, s1c, s1h, sdb, sdh, scb, sch, a.flow, a.depth, a.count,
                                   a.out, flag, bounds, ws, sw, nonce);
            } else if (variant == -46 && DEPTH) {   // timing arm: 64-bit fixed-point planes on ds_add_u64 (proj_owner5.hpp, FIX64)
                hipLaunchKernelGGL((proj_owner5<DEPTH, TH, 24, kMinW, false, true>), dim3(plan.nwg), dim3(16 * TH), 0, stream, w, h,
                                   ntx, nty, s1b, s1c, s1h, sdb, sdh, scb, sch, a.flow, a.depth, a.count, a.out, flag,
                                   bounds, stamps, ws, plan, nonce);
            } else if ((variant <= -47 && variant >= -50) || variant == -54) {   // how the motion estimate reaches the scan (proj_owner5.hpp, MOT):
#define MEMC_PROJ_MOT(M)                                                                                              \
                hipLaunchKernelGGL((proj_owner5<DEPTH, TH, 24, kMinWR, false, false, RAG, M>), dim3(plan.nwg), dim3(16 * TH), 0, stream, \
                                   w, h, ntx, nty, s1b, s1c, s1h, sdb, sdh, scb, sch, a.flow, a.depth, a.count, a.out, flag,  \
                                   bounds, stamps, ws, plan, nonce)
                if (variant == -47) MEMC_PROJ_MOT(1);         // -47: speculative m = 0 pass, samples by LDS DMA (round 6, lost)
                else if (variant == -48) MEMC_PROJ_MOT(2);    // -48: no estimate (timing arm)
                else if (variant == -49) MEMC_PROJ_MOT(3);    // -49: 16 samples, one lane each (timing arm)
                else if (variant == -50) MEMC_PROJ_MOT(4);    // -50: 16 samples through the scalar unit (timing arm)
                else MEMC_PROJ_MOT(5);                        // -54: the estimate cached per image in the call's scratch
#undef MEMC_PROJ_MOT
            } else if (variant <= -51 && variant >= -53) {   // tiles with many holes leave ALL of them pending (proj_fill.hpp, PENDT)
#define MEMC_PROJ_PENDT(T)                                                                                            \
                hipLaunchKernelGGL((proj_owner5<DEPTH, TH, 24, kMinWR, false, false, RAG, 0, T>), dim3(plan.nwg), dim3(16 * TH), 0, stream, \
                                   w, h, ntx, nty, s1b, s1c, s1h, sdb, sdh, scb, sch, a.flow, a.depth, a.count, a.out, flag,  \
                                   bounds, stamps, ws, plan, nonce)
                if (variant == -51) MEMC_PROJ_PENDT(0);       // -51: every tile with a hole
                else if (variant == -52) MEMC_PROJ_PENDT(8);  // -52: more than 8 lanes with a hole
                else MEMC_PROJ_PENDT(32);                     // -53: more than 32
#undef MEMC_PROJ_PENDT
            } else if (variant == -41) {       // timestamps (tools/trace_kernel.py proj5)
                hipLaunchKernelGGL((proj_owner5<DEPTH, TH, 24, kMinW, true>), dim3(plan.nwg), dim3(16 * TH), 0, stream, w, h,
                                   ntx, nty, s1b, s1c, s1h, sdb, sdh, scb, sch, a.flow, a.depth, a.count, a.out, flag,
                                   bounds, stamps, ws, plan, nonce);
            } else
#endif
            hipLaunchKernelGGL((proj_owner5<DEPTH, TH, 24, kMinWR, false, false, RAG>), dim3(plan.nwg), dim3(16 * TH), 0, stream, w,
                               h, ntx, nty, s1b, s1c, s1h, sdb, sdh, scb, sch, a.flow, a.depth, a.count, a.out, flag, bounds, stamps, ws,
                               plan, nonce);
            if (launch_status() != 0) return -1;
#ifdef MEMC_MEASURE
            if (g_proj_stall_us > 0) hipLaunchKernelGGL(proj_stall, dim3(1), dim3(64), 0, stream, g_proj_stall_us);
#endif
            if (!only_part) {
                const unsigned pg = r3_set ? persistent_grid(1) : persistent_grid(2);   // (53 KiB of LDS, 114 VGPRs: two per CU)
#ifdef MEMC_MEASURE
                if (r3_set)
                    hipLaunchKernelGGL((proj_owner_far_r3<DEPTH, TH, 24>), dim3(ntiles < pg ? ntiles : pg), dim3(16 * TH), 0,
                                       stream, w, h, ntx, nty, batch, s1b, s1c, s1h, sdb, sdh, scb, sch, a.flow, a.depth,
                                       a.count, a.out, flag, bounds, ws, nonce);
                else
#endif
                {   // (a lane of the grid per stamped tile at most: proj_owner_far's work list)
                    const unsigned need = (ntiles + 16u * TH - 1u) / (16u * TH), fg0 = ntiles < pg ? ntiles : pg;
                    hipLaunchKernelGGL((proj_owner_far<DEPTH, TH, 24, 4, 16 * TH, RAG>), dim3(fg0 > need ? fg0 : need), dim3(16 * TH), 0, stream,
                                       w, h, ntx, nty, batch, s1b, s1c, s1h, sdb, sdh, scb, sch, a.flow, a.depth, a.count,
                                       a.out, flag, bounds, stamps, ws, nonce);
                }
                if (launch_status() != 0) return -1;
            }
        }
    }
#ifdef MEMC_MEASURE
    else if (flag) {
        // rounds 1-2 owner kernels: 0 / 1 flags, flagged images redone by the general path queued behind the flag
        const unsigned gq = 256 * 2;
        bool launched = false;
        if (variant == -10 || variant == -7 || variant == -6) {       // round-1 owner kernel (64x16, strips)
            const unsigned nwg = ntiles;
#define MEMC_PROJ_OWNER(REACH, TRACE)                                                                              \
            hipLaunchKernelGGL((proj_owner<DEPTH, REACH, TRACE>), dim3(nwg), dim3(256), 0, stream, w, h, ntx, nty, s1b, \
                               s1c, s1h, sdb, sdh, scb, sch, a.flow, a.depth, a.count, a.out, flag, ws)
            if (TH == 16) {
                if (variant == -7) MEMC_PROJ_OWNER(24, true);
                else if (variant == -6) MEMC_PROJ_OWNER(16, false);
                else MEMC_PROJ_OWNER(24, false);
                launched = true;
            }
#undef MEMC_PROJ_OWNER
        }
        only_part = variant == -7;
#define MEMC_PROJ_OWNER2(ABL, TRACE)                                                                              \
            hipLaunchKernelGGL((proj_owner2<DEPTH, TH, 24, ABL, TRACE>), dim3(walk_grid(ntx, nty, batch, sw)),          \
                               dim3(16 * TH), 0, stream, w, h, ntx, nty, s1b, s1c, s1h, sdb, sdh, scb, sch, a.flow,       \
                               a.depth, a.count, a.out, flag, ws, sw)
        if (!launched && variant <= -21 && variant > -30) {   // -21 .. -26: timing arms of proj_owner2 (wrong results); -29: timestamps
            only_part = true;
            launched = true;
            if (variant == -21) MEMC_PROJ_OWNER2(1, false);
            else if (variant == -22) MEMC_PROJ_OWNER2(2, false);
            else if (variant == -23) MEMC_PROJ_OWNER2(3, false);
            else if (variant == -24) MEMC_PROJ_OWNER2(4, false);
            else if (variant == -25) MEMC_PROJ_OWNER2(5, false);
            else if (variant == -26) MEMC_PROJ_OWNER2(6, false);
            else if (variant == -29) MEMC_PROJ_OWNER2(0, true);
            else launched = false;
        }
        if (!launched && variant == -30) {     // proj_owner2: LDS rings, three planes
            MEMC_PROJ_OWNER2(0, false);
            launched = true;
        }
        if (!launched && variant == -31) {     // proj_owner3: persistent, next tile's fy prefetched
            if constexpr (TH == 32) {
                const unsigned npos = walk_grid(ntx, nty, batch, sw), pg = persistent_grid(2);
                hipLaunchKernelGGL((proj_owner3<DEPTH, 32, 24, 2>), dim3(npos < pg ? (npos + 7) / 8 * 8 : pg), dim3(512), 0,
                                   stream, w, h, ntx, nty, npos, s1b, s1c, s1h, sdb, sdh, scb, sch, a.flow, a.depth,
                                   a.count, a.out, flag, ws, sw);
                launched = true;
            }
        }
#undef MEMC_PROJ_OWNER2
        if (!launched) return -1;
        if (launch_status() != 0) return -1;
        if (!only_part) {
            hipLaunchKernelGGL(proj_redo_zero, dim3(gq), dim3(256), 0, stream, w, h, s1b, s1c, s1h, scb, sch, batch,
                               a.count, a.out, flag);
            MEMC_PROJ_SCATTER(0, flag);
            hipLaunchKernelGGL(proj_average_v4, dim3(gq), dim3(256), 0, stream, w, h, s1b, s1c, s1h, scb, sch, batch,
                               a.count, a.out, flag);
            if (ws.up)                          // summaries of the images the general path redid
                hipLaunchKernelGGL(proj_fill_summary<TH>, dim3(gq), dim3(256), 0, stream, w, h, ntx, nty, batch, scb,
                                   sch, a.count, ws, flag);
            if (launch_status() != 0) return -1;
        }
    }
#endif
    else {
        // the general path on its own: it DEFINES count and output (zero, scatter, average), it does not rely on
        // the caller's zero fill
        const unsigned gq = 0;                 // (unused: no flag)
        (void)gq;
#ifdef MEMC_MEASURE
        only_part = variant >= 2;               // (the ablation arms 2 / 3 time the scatter pass alone)
        if (variant == 2) MEMC_PROJ_SCATTER(2, (const int *)nullptr);
        else if (variant == 3) MEMC_PROJ_SCATTER(3, (const int *)nullptr);
#endif
        if (!only_part) {
            hipLaunchKernelGGL(proj_redo_zero, dim3(gs), dim3(256), 0, stream, w, h, s1b, s1c, s1h, scb, sch, batch,
                               a.count, a.out, (const int *)nullptr);
            MEMC_PROJ_SCATTER(0, (const int *)nullptr);
            hipLaunchKernelGGL(proj_average_v4, dim3(gs), dim3(256), 0, stream, w, h, s1b, s1c, s1h, scb, sch, batch,
                               a.count, a.out, (const int *)nullptr);
            if (ws.up) {
#ifdef MEMC_MEASURE
                if (old_fill)
                    hipLaunchKernelGGL(proj_fill_summary<TH>, dim3(gs), dim3(256), 0, stream, w, h, ntx, nty, batch, scb,
                                       sch, a.count, ws, (const int *)nullptr);
                else
#endif
                if constexpr (kNewOk)
                    hipLaunchKernelGGL(proj_fill_masks<TH>, dim3(ntiles < gs ? ntiles : gs), dim3(16 * TH), 0, stream, w, h,
                                       ntx, nty, batch, scb, sch, a.count, ws);
            }
        }
        if (launch_status() != 0) return -1;
    }
#undef MEMC_PROJ_SCATTER
    if (a.fillhole && !only_part && !skip_pending) {
        if (ws.up) {
            // round 3's filler: workgroup i looks after the tiles i, i + grid, ... (one flag per lane of a wave)
            [[maybe_unused]] const unsigned fg = ntiles < 4096u ? ntiles : (ntiles + 63u) / 64u > 4096u ? (ntiles + 63u) / 64u : 4096u;
            // proj_fill_pending: workgroup i lists the flagged ones among the tiles i, i + grid, ... and its sixteen waves take
            // one tile at a time; one workgroup per CU is what the chip holds at once (LDS, registers).  The grid is
            // coprime to the tiles per row: the tiles of an image's left or right edge (a camera pan's uncovered band: the
            // heavy ones) are tiles_x apart and would otherwise meet in a few workgroups.
            constexpr unsigned kW = (unsigned)kFillWaves;
            unsigned pg = (ntiles + kW - 1u) / kW < 256u ? (ntiles + kW - 1u) / kW : 256u;
            auto coprime = [](unsigned x, unsigned y) {
                while (y) {
                    const unsigned t = x % y;
                    x = y;
                    y = t;
                }
                return x == 1u;
            };
            while (pg > 1u && !coprime(pg, (unsigned)ntx)) pg--;
#ifdef MEMC_MEASURE
            if (old_fill)
                hipLaunchKernelGGL(proj_fillhole_carry<TH>, dim3(fg), dim3(256), 0, stream, w, h, ntx, nty, batch, s1b, s1c,
                                   s1h, scb, sch, a.count, a.out, ws);
            else
#endif
            if constexpr (kNewOk)
                hipLaunchKernelGGL(proj_fill_pending<TH>, dim3(pg), dim3(kFillWaves * kWave), 0, stream, w, h, ntx, nty, batch, s1b, s1c, s1h,
                                   scb, sch, a.count, a.out, ws);
        } else {
            hipLaunchKernelGGL(proj_fillhole_v4, dim3(sntiles), dim3(256), 0, stream, w, h, ntx, snty, s1b, s1c, s1h,
                               scb, sch, a.count, a.out, variant == -8 ? 1 : 0);
        }
        if (launch_status() != 0) return -1;
    }
    return 0;
}

template <bool DEPTH>
static int launch_proj_fwd(hipStream_t stream, int w, int h, int batch, int fillhole,
                           int s1b, int s1c, int s1h, int sdb, int sdh, int scb, int sch,
                           const float *flow, const float *depth, float *count, float *out,
                           void *ws = nullptr, size_t ws_bytes = 0)
{
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    const bool vec = vec4_ok(w, {s1b, s1c, s1h, sdb, sdh, scb, sch}, {flow, depth, count, out});
    if (!vec && w >= 8 && g_proj_variant < 0) {            // a ragged width: the owner kernels' RAG instantiations (round 5)
        const ProjArgs a = {stream, w, h, batch, fillhole, s1b, s1c, s1h, sdb, sdh, scb, sch, flow, depth, count, out,
                            ws, ws_bytes};
        const int r = run_proj_fwd<DEPTH, kOwnerTH, true>(a, kOwnerSW, -1);
        if (r <= 0) return r;                                // served (0) or failed (-1); 1: the scalar kernels below
    }
    if (vec && g_proj_variant != 0) {
        const ProjArgs a = {stream, w, h, batch, fillhole, s1b, s1c, s1h, sdb, sdh, scb, sch, flow, depth, count, out,
                            ws, ws_bytes};
#ifdef MEMC_MEASURE
        // 100 + 10 * log2(TH / 16) + stripe width: owner geometry under test; -10 / -7 / -6: the round-1 owner
        int v = g_proj_variant, th = kOwnerTH, sw = kOwnerSW;
        if (v >= 100 && v < 120) {             // the production kernel (proj_owner5) in another geometry (TH 16 / 32)
            th = 16 << ((v - 100) / 10);
            sw = (v - 100) % 10;
            v = -1;
        } else if (v >= 400 && v < 420) {      // round 3's production set (proj_owner4 + carry filler), same geometry code + 300
            th = 16 << ((v - 400) / 10);
            sw = (v - 400) % 10;
            v = -40;
        } else if (v >= 130 && v < 160) {      // proj_owner2 (LDS rings, three planes), same geometry code + 30
            th = 16 << ((v - 130) / 10);
            sw = (v - 130) % 10;
            v = -30;
        } else if (v >= 160 && v < 170) {      // proj_owner3 (persistent, TH = 32), stripe width v - 160
            th = 32;
            sw = v - 160;
            v = -31;
        } else if (v == -10 || v == -7 || v == -6) {
            th = 16;
            sw = 0;
        } else if (v >= 200 && v < 300) {      // 200 + 10 * arm + log2(TH / 16): timing arms / timestamps of proj_owner2
            th = 16 << (v % 10);
            sw = 0;
            v = -(20 + (v - 200) / 10);
        }
        if (th == 16) return run_proj_fwd<DEPTH, 16>(a, sw, v);
        if (th == 64) return run_proj_fwd<DEPTH, 64>(a, sw, v);
        return run_proj_fwd<DEPTH, 32>(a, sw, v);
#else
        return run_proj_fwd<DEPTH, kOwnerTH>(a, kOwnerSW, -1);
#endif
    }
    const int tiles_x = (w + kWave - 1) / kWave, tiles_y = (h + 3) / 4;
    const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
    MEMC_PATH(DEPTH ? "dproj_fwd:scalar" : "proj_fwd:scalar");
    hipLaunchKernelGGL(proj_zero_scalar, dim3(256 * 8), dim3(256), 0, stream, w, h, batch, (int64_t)s1b, (int64_t)s1c, s1h,
                       (int64_t)scb, sch, count, out);
    hipLaunchKernelGGL(proj_scatter<DEPTH>, dim3(nwg), dim3(256), 0, stream, w, h, tiles_x, tiles_y,
                       (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)sdb, sdh, (int64_t)scb, sch, flow, depth, count, out);
    if (launch_status() != 0) return -1;
    hipLaunchKernelGGL(proj_average, dim3(nwg), dim3(256), 0, stream, w, h, tiles_x, tiles_y,
                       (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)scb, sch, count, out);
    if (launch_status() != 0) return -1;
    if (fillhole) {
        hipLaunchKernelGGL(proj_fillhole, dim3(nwg), dim3(256), 0, stream, w, h, tiles_x, tiles_y,
                           (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)scb, sch, count, out);
        if (launch_status() != 0) return -1;
    }
    return 0;
}

template <bool DEPTH>
static int launch_proj_bwd(hipStream_t stream, int w, int h, int batch,
                           int s1b, int s1c, int s1h, int sdb, int sdh, int scb, int sch,
                           const float *flow, const float *depth, const float *count, const float *fwd_out,
                           const float *gout, float *gin1, float *gin2)
{
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    const bool vec = vec4_ok(w, {s1b, s1c, s1h, sdb, sdh, scb, sch}, {flow, depth, count, fwd_out, gout, gin1, gin2});
    // A width that is not a multiple of four (round 5): the tiled kernel takes the whole quads (sites x < ws), the one-lane-
    // per-site kernel the one to three columns behind them.
    const int ws = w & ~3;
    if ((vec || (ws >= 8 && g_proj_variant < 0)) && g_proj_variant != 0) {
        using G = TileGeom<16>;
        const int ntx = (ws + G::kTW - 1) / G::kTW, nty = (h + G::kTH - 1) / G::kTH;
        const int sw = g_tile_walk_sw >= 0 ? g_tile_walk_sw : kDefaultStripe;
        const unsigned nwg = walk_grid(ntx, nty, batch, sw);
        MEMC_PATH(DEPTH ? "dproj_bwd:tiled" : "proj_bwd:tiled");
#define MEMC_PROJ_BWD_R(CAP, RAG)                                                                               \
        hipLaunchKernelGGL((proj_bwd_tiled<DEPTH, CAP, RAG>), dim3(nwg), dim3(256), (tile_lds_bytes<16, CAP>()), stream, w, \
                           h, ntx, nty, (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)sdb, sdh, (int64_t)scb, sch, flow,   \
                           depth, count, fwd_out, gout, gin1, gin2, sw)
#define MEMC_PROJ_BWD(CAP)                                                                                      \
        do {                                                                                                        \
            if (ws < w) MEMC_PROJ_BWD_R(CAP, true);                                                                 \
            else MEMC_PROJ_BWD_R(CAP, false);                                                                       \
        } while (0)
        // 39 KiB of staged cells instead of 48 -> 4 workgroups per CU: 205 -> 180 us (depth 271 -> 256), same results
#ifdef MEMC_MEASURE
        if (g_cap_sel == 0) MEMC_PROJ_BWD(3072);
        else if (g_cap_sel == 2) MEMC_PROJ_BWD(1984);          // 5 per CU
        else
#endif
        MEMC_PROJ_BWD(2496);
#undef MEMC_PROJ_BWD
#undef MEMC_PROJ_BWD_R
        if (ws < w) {                          // the ragged row's last columns
            const int tail_y = (h + 3) / 4;
            hipLaunchKernelGGL(proj_bwd<DEPTH>, dim3((unsigned)tail_y * batch), dim3(256), 0, stream, w, h, 1, tail_y,
                               (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)sdb, sdh, (int64_t)scb, sch, flow, depth, count,
                               fwd_out, gout, gin1, gin2, ws);
        }
        return launch_status();
    }
    const int tiles_x = (w + kWave - 1) / kWave, tiles_y = (h + 3) / 4;
    const unsigned nwg = (unsigned)tiles_x * tiles_y * batch;
    MEMC_PATH(DEPTH ? "dproj_bwd:scalar" : "proj_bwd:scalar");
    hipLaunchKernelGGL(proj_bwd<DEPTH>, dim3(nwg), dim3(256), 0, stream, w, h, tiles_x, tiles_y,
                       (int64_t)s1b, (int64_t)s1c, s1h, (int64_t)sdb, sdh, (int64_t)scb, sch, flow, depth, count,
                       fwd_out, gout, gin1, gin2, 0);
    return launch_status();
}

}  // namespace memc

using namespace memc;

#ifdef MEMC_MEASURE
extern "C" void memc_debug_set_projection_variant(int v) { g_proj_variant = v; }
extern "C" void memc_debug_set_projection_scratch_blocks(int n) { g_proj_scratch_blocks = n < 1 ? 1 : (n > kBlocks ? kBlocks : n); }
extern "C" void memc_debug_set_projection_stall_us(int us) { g_proj_stall_us = us; }
extern "C" int memc_debug_set_trace_buffer_proj(void *p)
{
    unsigned long long *q = (unsigned long long *)p;
    return hipMemcpyToSymbol(HIP_SYMBOL(memc::g_trace_buf_proj), &q, sizeof(q)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int FlowProjection_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int fillhole,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int scb, const int scc, const int sch, const int scw,
    const float *input1, float *count, float *output)
{
    (void)nElement; (void)channel; (void)s1w; (void)scc; (void)scw;
    return launch_proj_fwd<false>((hipStream_t)stream, w, h, batch, fillhole, s1b, s1c, s1h, 0, 0, scb, sch,
                                  input1, nullptr, count, output);
}

extern "C" int FlowProjection_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int scb, const int scc, const int sch, const int scw,
    const float *input1, const float *count, const float *gradoutput, float *gradinput1)
{
    (void)nElement; (void)channel; (void)s1w; (void)scc; (void)scw;
    return launch_proj_bwd<false>((hipStream_t)stream, w, h, batch, s1b, s1c, s1h, 0, 0, scb, sch, input1, nullptr,
                                  count, nullptr, gradoutput, gradinput1, nullptr);
}

extern "C" int DepthFlowProjection_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int fillhole,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const int scb, const int scc, const int sch, const int scw,
    const float *input1, const float *input2, float *count, float *output)
{
    (void)nElement; (void)channel; (void)s1w; (void)s2c; (void)s2w; (void)scc; (void)scw;
    return launch_proj_fwd<true>((hipStream_t)stream, w, h, batch, fillhole, s1b, s1c, s1h, s2b, s2h, scb, sch,
                                 input1, input2, count, output);
}

extern "C" int DepthFlowProjection_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const int scb, const int scc, const int sch, const int scw,
    const float *input1, const float *input2, const float *count, const float *output,
    const float *gradoutput, float *gradinput1, float *gradinput2)
{
    (void)nElement; (void)channel; (void)s1w; (void)s2c; (void)s2w; (void)scc; (void)scw;
    return launch_proj_bwd<true>((hipStream_t)stream, w, h, batch, s1b, s1c, s1h, s2b, s2h, scb, sch, input1, input2,
                                 count, output, gradoutput, gradinput1, gradinput2);
}

// ------------------------------------------------------------------------------------------------------------------
// Caller-supplied workspace (include/memc_warp.h, "EXTENSION: workspace").  The reference's launchers touch borrowed buffers
// only (my_lib_kernel.cu:1905-1992; my_lib_cuda.c:752-799: "callee never allocates"); with a workspace so does this one.
// ------------------------------------------------------------------------------------------------------------------
extern "C" size_t memc_flow_projection_workspace_bytes(int w, int h, int batch, int fillhole, int depth)
{
    (void)depth;                                 // (the same tables for both operators)
    if (w <= 0 || h <= 0 || batch <= 0) return 0;
    // what run_proj_fwd<., kOwnerTH> lays out when it takes its fast path with hole filling as asked; shapes that take
    // another path (odd widths, planes beyond 4 GiB) use less or nothing
    const ProjWsLayout l = proj_ws_layout<kOwnerTH>(w, h, batch, true, fillhole != 0, true);
    return (l.bytes() + 255) / 256 * 256;
}

extern "C" int FlowProjection_gpu_forward_kernel_ws(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int fillhole,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int scb, const int scc, const int sch, const int scw,
    const float *input1, float *count, float *output, void *workspace, size_t workspace_bytes)
{
    (void)nElement; (void)channel; (void)s1w; (void)scc; (void)scw;
    if (!workspace) return -1;
    return launch_proj_fwd<false>((hipStream_t)stream, w, h, batch, fillhole, s1b, s1c, s1h, 0, 0, scb, sch,
                                  input1, nullptr, count, output, workspace, workspace_bytes);
}

extern "C" int DepthFlowProjection_gpu_forward_kernel_ws(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int fillhole,
    const int s1b, const int s1c, const int s1h, const int s1w,
    const int s2b, const int s2c, const int s2h, const int s2w,
    const int scb, const int scc, const int sch, const int scw,
    const float *input1, const float *input2, float *count, float *output, void *workspace, size_t workspace_bytes)
{
    (void)nElement; (void)channel; (void)s1w; (void)s2c; (void)s2w; (void)scc; (void)scw;
    if (!workspace) return -1;
    return launch_proj_fwd<true>((hipStream_t)stream, w, h, batch, fillhole, s1b, s1c, s1h, s2b, s2h, scb, sch,
                                 input1, input2, count, output, workspace, workspace_bytes);
}
