// skeleton_walks.hip -- measurement probe, NOT part of libmemc_hip.so (round 6).
//
// The adaptive warp's I/O skeleton (io_skeleton.hip: per site 2 flow + 16 tap + 3 image values read at the site's own
// position, 3 written -- 96 B/site, no gathers, no LDS) with the ORDER in which the chip walks the tiles as a run-time
// parameter.  The product kernel runs in the time of its skeleton (profiles/r06_headline_walks_and_skeleton.txt), so the only
// thing left to learn is which orders HBM likes -- and whether one of them keeps the vertical neighbours of a tile on one XCD
// (the real kernel's source boxes overlap by ~40 % vertically; an order that breaks that pays for it in L2 misses).
//
// walk 0  raster order cut in eight contiguous chunks, one per XCD            (io_skeleton "xcd")
//      1  raster order, workgroup b = tile b (XCD = b % 8)                      (io_skeleton "blockIdx-order")
//      2  column strips dealt to the XCDs, each walked top to bottom          (the product: strip_walk)
//      3  column classes: XCD k owns the strips s = k (mod 8) as in 2, but walks G of them side by side, row by row -- the
//         eight XCDs move down the same G * 8 strips together (a raster-compact front) and a tile's vertical neighbour
//         is G positions later on the SAME XCD
//      4  stripes two tile columns wide per XCD, row-major inside
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 ldnt(const float *p) { return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p)); }
__device__ __forceinline__ void stnt(float *p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p)); }

// stores with every combination of the cache-policy bits (WR = 8 + bits: 1 = sc0, 2 = sc1, 4 = nt)
template <int BITS>
__device__ __forceinline__ void st_policy(float *p, f32x4 v)
{
    if (BITS == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (BITS == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (BITS == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (BITS == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if (BITS == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if (BITS == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    if (BITS == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if (BITS == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

struct Tile { int b, tx, ty; };

__device__ __forceinline__ Tile walk_tile(int walk, int G, unsigned bid, unsigned nwg, int tiles_x, int tiles_y)
{
    Tile c;
    const unsigned q = nwg / 8, xcd = bid % 8, idx = bid / 8;          // (the probe's grids are multiples of 8)
    if (walk == 0 || walk == 1) {
        const unsigned t = walk == 0 ? xcd * q + idx : bid;
        c.tx = t % tiles_x;  c.ty = (t / tiles_x) % tiles_y;  c.b = t / (tiles_x * tiles_y);
    } else if (walk == 2) {
        const unsigned s = xcd + 8 * (idx / tiles_y);
        c.ty = idx % tiles_y;  c.b = s / tiles_x;  c.tx = s % tiles_x;
    } else if (walk == 3) {
        const unsigned per = (unsigned)G * tiles_y, g = idx / per, r = idx % per;
        const unsigned s = xcd + 8 * (g * G + r % G);
        c.ty = r / G;  c.b = s / tiles_x;  c.tx = s % tiles_x;
    } else {
        const unsigned per = 2u * tiles_y, k = idx / per, r = idx % per;
        const unsigned s2 = xcd + 8 * k;                                // stripe number, tiles_x / 2 stripes per image
        c.ty = r / 2;  c.b = s2 / (tiles_x / 2);  c.tx = (s2 % (tiles_x / 2)) * 2 + r % 2;
    }
    return c;
}

// LX lanes per tile row (4 sites each), 256 threads.  WR: 0 = nt stores, 1 = plain stores, 2 = no stores (read-only mix)
template <int LX, int WR>
__global__ __launch_bounds__(256, 2) void skeleton_walk(int W, int H, int64_t plane, const float *__restrict__ in1,
                                                        const float *__restrict__ flow, const float *__restrict__ filt,
                                                        float *__restrict__ out, int tiles_x, int tiles_y, int walk, int G)
{
    const Tile t = walk_tile(walk, G, blockIdx.x, gridDim.x, tiles_x, tiles_y);
    const int b = t.b;
    const int x = t.tx * 4 * LX + 4 * (threadIdx.x % LX), y = t.ty * (256 / LX) + threadIdx.x / LX;
    if (x >= W || y >= H) return;
    const int64_t o = (int64_t)y * W + x;
    f32x4 acc = ldnt(flow + (b * 2 + 0) * plane + o) + ldnt(flow + (b * 2 + 1) * plane + o);
    f32x4 tp[16];
#pragma unroll
    for (int k = 0; k < 16; k++) tp[k] = ldnt(filt + (b * 16 + k) * plane + o);
    f32x4 im[3];
#pragma unroll
    for (int c = 0; c < 3; c++) im[c] = *reinterpret_cast<const f32x4 *>(in1 + (b * 3 + c) * plane + o);
#pragma unroll
    for (int k = 0; k < 16; k++) acc += tp[k];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const f32x4 v = acc * im[c];
        if (WR == 0) stnt(out + (b * 3 + c) * plane + o, v);
        else if (WR == 1) *reinterpret_cast<f32x4 *>(out + (b * 3 + c) * plane + o) = v;
        else if (WR >= 8) st_policy<(WR >= 8 ? WR - 8 : 0)>(out + (b * 3 + c) * plane + o, v);
        else if (v.x == 12345.678f) out[0] = v.y;
    }
}

extern "C" int probe_skeleton_walk(void *stream, int lx, int wr, int walk, int G, int B, int H, int W, const float *in1,
                                   const float *flow, const float *filt, float *out)
{
    const int tx = (W + 4 * lx - 1) / (4 * lx), ty = (H + 256 / lx - 1) / (256 / lx);
    const unsigned grid = (unsigned)tx * ty * B;
    if (grid % 8) return -2;
    if (walk == 3 && ((unsigned)tx * B / 8) % G) return -3;
    if (walk == 4 && (tx % 2 || (unsigned)(tx / 2) * B % 8)) return -4;
#define GO(LX, WR) hipLaunchKernelGGL((skeleton_walk<LX, WR>), dim3(grid), dim3(256), 0, (hipStream_t)stream, W, H, \
                                      (int64_t)W * H, in1, flow, filt, out, tx, ty, walk, G)
    if (lx == 16 && wr == 0) GO(16, 0);
    else if (lx == 16 && wr == 1) GO(16, 1);
    else if (lx == 16 && wr == 2) GO(16, 2);
    else if (lx == 16 && wr == 8) GO(16, 8);
    else if (lx == 16 && wr == 9) GO(16, 9);
    else if (lx == 16 && wr == 10) GO(16, 10);
    else if (lx == 16 && wr == 11) GO(16, 11);
    else if (lx == 16 && wr == 12) GO(16, 12);
    else if (lx == 16 && wr == 13) GO(16, 13);
    else if (lx == 16 && wr == 14) GO(16, 14);
    else if (lx == 16 && wr == 15) GO(16, 15);
    else if (lx == 32 && wr == 0) GO(32, 0);
    else if (lx == 64 && wr == 0) GO(64, 0);
    else if (lx == 64 && wr == 2) GO(64, 2);
    else return -1;
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
