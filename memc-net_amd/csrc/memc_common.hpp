// memc_common.hpp -- device-side helpers shared by the gfx950 kernels of libmemc_hip.so.
//
// Target: MI355X (gfx950, CDNA4) only: 64-lane wavefronts, 256 CUs in 8 XCDs (one L2 per XCD),
// 160 KiB LDS per CU.  Everything here is HBM/L2-bound gather/scatter work; there is no MFMA.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include <initializer_list>

namespace memc {

constexpr int kWave = 64;      // wavefront width on CDNA
constexpr int kXcds = 8;       // XCDs (private L2s) on MI355X

__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }

// Workgroup b is dispatched to XCD b % 8 (observed placement, speed only).  Remap the linear
// workgroup id so that every XCD owns one contiguous run of tiles: vertically adjacent tiles
// (which share halo rows of the source image) then hit the same L2.  Bijective for any nwg.
__device__ __forceinline__ unsigned xcd_chunked_id(unsigned bid, unsigned nwg)
{
    const unsigned q = nwg / kXcds, r = nwg % kXcds;
    const unsigned xcd = bid % kXcds, idx = bid / kXcds;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Tile walk used by the LDS-tiled kernels.  Two things are wanted at once:
//   (a) tiles that share source rows (vertical neighbours: the staged box is ~1.7x taller than the tile) must
//       run on the SAME XCD, close in time, so the halo is an L2 hit, and
//   (b) the eight XCDs together should stream one neighbourhood of one image at a time (eight far-apart
//       streams cost ~7 % of the achievable HBM bandwidth: tools/probes/io_skeleton.hip).
// So whole column strips (one tile wide, full image height) are dealt round-robin to the XCDs -- strip s of the
// batch (s = image * tiles_x + tile column) belongs to XCD s % 8 -- and an XCD walks its strips top to bottom.
// Any position p of that "XCD-major" order is mapped back to (image, tx, ty); xcd_chunked_id() gives the
// position of hardware workgroup b.  Bijective for any grid.
struct TileCoord {
    int b, tx, ty;
};

// tile at position p of the XCD-major order
__device__ __forceinline__ TileCoord strip_at(unsigned p, int tiles_x, int tiles_y, int batch)
{
    const unsigned S = (unsigned)tiles_x * batch;          // strips
    const unsigned Q = S / kXcds, R = S % kXcds;
    const unsigned big = (Q + 1) * tiles_y;                // tiles of an XCD class that owns Q+1 strips
    unsigned k, rem;
    if (p < R * big) {
        k = p / big;
        rem = p % big;
    } else {
        const unsigned pp = p - R * big, small = Q * tiles_y;
        k = R + pp / small;
        rem = pp % small;
    }
    const unsigned s = k + kXcds * (rem / tiles_y);
    TileCoord c;
    c.ty = rem % tiles_y;
    c.b = s / tiles_x;
    c.tx = s % tiles_x;
    return c;
}

__device__ __forceinline__ TileCoord strip_walk(unsigned bid, unsigned nwg, int tiles_x, int tiles_y, int batch)
{
    return strip_at(xcd_chunked_id(bid, nwg), tiles_x, tiles_y, batch);
}

// Position `ahead` places later in the SAME XCD's part of the walk (what a workgroup dispatched ~`ahead`/8 slots
// later on this XCD will process), or -1 past the end of this XCD's chunk.  Used to pull that tile's flow into
// this XCD's L2 before its workgroup starts.
__device__ __forceinline__ int strip_ahead(unsigned bid, unsigned nwg, unsigned ahead)
{
    const unsigned q = nwg / kXcds, r = nwg % kXcds, xcd = bid % kXcds;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const unsigned end = base + q + (xcd < r ? 1u : 0u);
    const unsigned p = base + bid / kXcds + ahead;
    return p < end ? (int)p : -1;
}

// Same idea with stripes SW tile columns wide, walked row-major inside the stripe: horizontal neighbours then
// also share an XCD (and are in flight together).  The grid is launched over ceil(tiles_x / SW) * SW virtual
// columns per image; tiles with tx >= tiles_x do not exist (the caller returns at once).
template <int SW>
__device__ __forceinline__ TileCoord stripe_walk(unsigned bid, unsigned nwg, int tiles_x, int tiles_y)
{
    const int stripes_x = (tiles_x + SW - 1) / SW;
    const int batch = nwg / (stripes_x * SW * tiles_y);
    // position in XCD-major order, in units of tiles; a stripe holds SW * tiles_y tiles
    const unsigned p = xcd_chunked_id(bid, nwg);
    const unsigned per = SW * tiles_y;
    const unsigned S = (unsigned)stripes_x * batch;        // stripes
    const unsigned Q = S / kXcds, R = S % kXcds;
    const unsigned big = (Q + 1) * per;
    unsigned k, rem;
    if (p < R * big) {
        k = p / big;
        rem = p % big;
    } else {
        const unsigned pp = p - R * big, small = Q * per;
        k = R + pp / small;
        rem = pp % small;
    }
    const unsigned s = k + kXcds * (rem / per);
    const unsigned in_stripe = rem % per;
    TileCoord c;
    c.b = s / stripes_x;
    c.tx = (s % stripes_x) * SW + in_stripe % SW;
    c.ty = in_stripe / SW;
    return c;
}

// Walk chosen at run time: sw == 0 -> strip_walk over a grid of tiles_x * tiles_y * batch workgroups; sw > 0 ->
// stripes sw tile columns wide over a grid of walk_grid(...) workgroups, tx >= tiles_x means "no such tile".
__device__ __forceinline__ TileCoord tile_walk(unsigned bid, unsigned nwg, int tiles_x, int tiles_y, int sw)
{
    if (sw == 0) return strip_walk(bid, nwg, tiles_x, tiles_y, nwg / (tiles_x * tiles_y));
    const int stripes_x = (tiles_x + sw - 1) / sw;
    const unsigned p = xcd_chunked_id(bid, nwg);
    const unsigned per = (unsigned)sw * tiles_y;
    const unsigned S = nwg / per;                          // stripes in the launch
    const unsigned Q = S / kXcds, R = S % kXcds;
    const unsigned big = (Q + 1) * per;
    unsigned k, rem;
    if (p < R * big) {
        k = p / big;
        rem = p % big;
    } else {
        const unsigned pp = p - R * big, small = Q * per;
        k = R + pp / small;
        rem = pp % small;
    }
    const unsigned s = k + kXcds * (rem / per);
    const unsigned in_stripe = rem % per;
    TileCoord c;
    c.b = s / stripes_x;
    c.tx = (s % stripes_x) * sw + in_stripe % sw;
    c.ty = in_stripe / sw;
    return c;
}

inline unsigned walk_grid(int tiles_x, int tiles_y, int batch, int sw)
{
    const int cols = sw > 0 ? (tiles_x + sw - 1) / sw * sw : tiles_x;
    return (unsigned)cols * tiles_y * batch;
}

// tile_walk with its launch constants -- and the reciprocals of its divisors -- precomputed on the host: the walk's
// eight integer divisions are ~25 scalar instructions each when the divisor is a kernel argument, and every wave of
// every workgroup executes them; for a kernel whose scalar pipe is as busy as its vector pipe (the projection's owner
// kernel) that was a third of its scalar instructions.  n / d == mulhi(n, ceil(2^32 / d)) exactly while n * d < 2^32
// (make_walk_plan checks; `fast` == 0 -> the kernel calls tile_walk).
struct WalkPlan {
    unsigned nwg, q, r;                  // grid; nwg / 8, nwg % 8
    unsigned per, Q, R, big, sml;        // tiles per stripe; stripes / 8, stripes % 8; tiles of a big / small XCD class
    unsigned stripes_x, sw;              // stripes per image, tile columns per stripe (strips: tiles_x, 1)
    unsigned m_big, m_sml, m_per, m_sx, m_sw;
    int tiles_x, tiles_y, fast;
};

inline unsigned walk_recip(unsigned d) { return d <= 1 ? 0u : (unsigned)((0x100000000ull + d - 1) / d); }

inline WalkPlan make_walk_plan(int tiles_x, int tiles_y, int batch, int sw)
{
    WalkPlan p;
    p.tiles_x = tiles_x;
    p.tiles_y = tiles_y;
    p.sw = sw > 0 ? (unsigned)sw : 1u;
    p.stripes_x = sw > 0 ? (unsigned)((tiles_x + sw - 1) / sw) : (unsigned)tiles_x;
    p.nwg = walk_grid(tiles_x, tiles_y, batch, sw);
    p.q = p.nwg / kXcds;
    p.r = p.nwg % kXcds;
    p.per = p.sw * (unsigned)tiles_y;
    const unsigned S = p.nwg / p.per;                      // stripes in the launch
    p.Q = S / kXcds;
    p.R = S % kXcds;
    p.big = (p.Q + 1) * p.per;
    p.sml = p.Q * p.per;
    p.m_big = walk_recip(p.big);
    p.m_sml = walk_recip(p.sml);
    p.m_per = walk_recip(p.per);
    p.m_sx = walk_recip(p.stripes_x);
    p.m_sw = walk_recip(p.sw);
    const unsigned long long dmax = p.big > p.stripes_x ? p.big : p.stripes_x;
    p.fast = (unsigned long long)p.nwg * dmax < 0x100000000ull ? 1 : 0;
    return p;
}

__device__ __forceinline__ unsigned walk_div(unsigned n, unsigned d, unsigned m) { return d <= 1 ? n : __umulhi(n, m); }

// == tile_walk(bid, p.nwg, p.tiles_x, p.tiles_y, sw) for p.fast != 0
__device__ __forceinline__ TileCoord tile_walk_plan(unsigned bid, const WalkPlan &p)
{
    const unsigned xcd = bid % kXcds, idx = bid / kXcds;
    const unsigned pos = (xcd < p.r ? xcd * (p.q + 1) : p.r * (p.q + 1) + (xcd - p.r) * p.q) + idx;
    unsigned k, rem;
    if (pos < p.R * p.big) {
        k = walk_div(pos, p.big, p.m_big);
        rem = pos - k * p.big;
    } else {
        const unsigned pp = pos - p.R * p.big, kk = walk_div(pp, p.sml, p.m_sml);
        k = p.R + kk;
        rem = pp - kk * p.sml;
    }
    const unsigned st = walk_div(rem, p.per, p.m_per), in_stripe = rem - st * p.per;
    const unsigned s = k + kXcds * st;
    const unsigned b = walk_div(s, p.stripes_x, p.m_sx), sx = s - b * p.stripes_x;
    const unsigned ty = walk_div(in_stripe, p.sw, p.m_sw);
    TileCoord c;
    c.b = (int)b;
    c.tx = (int)(sx * p.sw + (in_stripe - ty * p.sw));
    c.ty = (int)ty;
    return c;
}

// Measurement knobs.  They exist only in the MEASUREMENT build (-DMEMC_MEASURE -> lib/libmemc_hip_measure.so, used by
// tools/ and by the tests that force a particular kernel).  In the product build (libmemc_hip.so) every knob is a
// compile-time constant at its default: the ablation arms -- some of which return wrong results by construction --
// are not even instantiated, no memc_debug_* symbol is exported and nothing is read from the environment.
#ifdef MEMC_MEASURE
#define MEMC_KNOB_STATIC(name, dflt) static int name = dflt
extern int g_tile_walk_sw;                                 // memc_debug_set_walk; < 0: the default below
#else
#define MEMC_KNOB_STATIC(name, dflt) static constexpr int name = dflt
constexpr int g_tile_walk_sw = -1;
#endif
// Default: one tile column per XCD strip (0).  Stripes n tile columns wide keep horizontal neighbours on one XCD:
// measured on the bilinear warp / projection backward (rocprofv3 FETCH_SIZE, tools/probes/pmc_walk.py) they read 22 %
// less (865 -> 677 MB) but run within +-2 % of the strips (those kernels are latency-, not traffic-bound), so the
// strips stay.
constexpr int kDefaultStripe = 0;
#ifdef MEMC_MEASURE
extern int g_cap_sel;                                      // LDS staging budget of the 2x2-footprint kernels
extern int g_extra_lds;                                    // occupancy experiment: unused dynamic LDS added to bl_fwd_tiled<3>
#else
constexpr int g_cap_sel = -1;
constexpr int g_extra_lds = 0;
#endif

// Streaming accesses: every filter-tap / flow / output element is touched exactly once per launch,
// so keep it from displacing the (re-used) source-image lines in L1/L2.
__device__ __forceinline__ float ld_stream(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(float *p, float v) { __builtin_nontemporal_store(v, p); }

// Hardware fp32 atomic add without return (global_atomic_add_f32); device (agent) scope.
// Buffers are ordinary coarse-grained device allocations, for which the hardware atomic is exact.
__device__ __forceinline__ void atomic_add_f32(float *p, float v) { (void)unsafeAtomicAdd(p, v); }

// One output site of the adaptive warp: validity test and window origin.
// Restates my_lib_kernel.cu:1126-1138 (== my_lib.c:973-987).
struct FiSite {
    int ix, iy;       // (int)x2, (int)y2  (meaningful only when valid)
    float a, b;       // alpha, beta
    bool valid;
};

__device__ __forceinline__ FiSite fi_locate(int x, int y, int W, int H, float fx, float fy)
{
    FiSite s;
    const float x2 = (float)x + fx, y2 = (float)y + fy;
    s.valid = x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1) &&
              fabsf(fx) < (float)W / 2.0f && fabsf(fy) < (float)H / 2.0f;
    s.ix = s.valid ? (int)x2 : 0;
    s.iy = s.valid ? (int)y2 : 0;
    s.a = x2 - (float)s.ix;
    s.b = y2 - (float)s.iy;
    return s;
}

// Projection / bilinear sites.  strict == false: x2 <= W-1 (FlowProjection, my_lib_kernel.cu:1670);
// strict == true: x2 < W (Interpolation, my_lib_kernel.cu:543).
struct BlSite {
    int L, T, R, Bm;
    float a, b;
    bool valid;
};

template <bool STRICT>
__device__ __forceinline__ BlSite bl_locate(int x, int y, int W, int H, float fx, float fy)
{
    BlSite s;
    const float x2 = (float)x + fx, y2 = (float)y + fy;
    if (STRICT)
        s.valid = x2 >= 0.0f && y2 >= 0.0f && x2 < (float)W && y2 < (float)H;
    else
        s.valid = x2 >= 0.0f && y2 >= 0.0f && x2 <= (float)(W - 1) && y2 <= (float)(H - 1);
    s.L = s.valid ? (int)x2 : 0;
    s.T = s.valid ? (int)y2 : 0;
    s.R = min(s.L + 1, W - 1);
    s.Bm = min(s.T + 1, H - 1);
    s.a = x2 - (float)s.L;
    s.b = y2 - (float)s.T;
    return s;
}

// Error-checked launch epilogue shared by the extern "C" launchers: the reference returns -1 after a
// failed cudaGetLastError() (my_lib_kernel.cu:1559-1566).
inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// Kernels that carve more than 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU) must opt in -- per DEVICE
// (hipFuncSetAttribute applies to the current device): the launchers call this before every such launch, it is cheap,
// and a process that drives several GPUs stays correct (a once-per-process flag would leave device 1 without it).
template <typename K>
inline void allow_big_lds(K kernel, int bytes)
{
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                                     hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

// Grid of a persistent kernel: `per_cu` workgroups on every CU of the current device, rounded down to a
// multiple of the XCD count so that workgroup w and w + grid sit on the same XCD.
inline unsigned persistent_grid(int per_cu)
{
    static unsigned cus[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = 0;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = (unsigned)n;
    }
    const unsigned g = (unsigned)per_cu * cus[dev] / kXcds * kXcds;
    return g ? g : kXcds;
}

}  // namespace memc
