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

// LOAD POLICIES: the skeleton's 18 stream loads with every combination of the cache-policy bits (inline assembly; BITS: 1 = sc0,
// 2 = sc1, 4 = nt), strips, nt stores.  Does any of them do for CONTIGUOUS tensors what padded strides do (profiles/r06_plane_strides.txt)?
template <int BITS>
__device__ __forceinline__ f32x4 ld_policy(const float *p)
{
    f32x4 v;
    if (BITS == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if (BITS == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if (BITS == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if (BITS == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if (BITS == 4) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if (BITS == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    if (BITS == 6) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if (BITS == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int BITS>
__global__ __launch_bounds__(256, 2) void skeleton_loads(int W, int H, int64_t plane, int64_t rowp, const float *__restrict__ in1,
                                                         const float *__restrict__ flow, const float *__restrict__ filt,
                                                         float *__restrict__ out, int tiles_x, int tiles_y)
{
    constexpr int LX = 16;
    const Tile t = walk_tile(2, 0, blockIdx.x, gridDim.x, tiles_x, tiles_y);
    const int b = t.b;
    const int x = t.tx * 4 * LX + 4 * (threadIdx.x % LX), y = t.ty * (256 / LX) + threadIdx.x / LX;
    if (x >= W || y >= H) return;
    const int64_t o = (int64_t)y * rowp + x;               // rowp: the row pitch in floats (W, or padded)
    f32x4 fl[2], tp[16], im[3];
#pragma unroll
    for (int k = 0; k < 2; k++) fl[k] = ld_policy<BITS>(flow + (b * 2 + k) * plane + o);
#pragma unroll
    for (int k = 0; k < 16; k++) tp[k] = ld_policy<BITS>(filt + (b * 16 + k) * plane + o);
#pragma unroll
    for (int c = 0; c < 3; c++) im[c] = ld_policy<BITS>(in1 + (b * 3 + c) * plane + o);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 2; k++) asm volatile("" : "+v"(fl[k]));
#pragma unroll
    for (int k = 0; k < 16; k++) asm volatile("" : "+v"(tp[k]));
#pragma unroll
    for (int c = 0; c < 3; c++) asm volatile("" : "+v"(im[c]));
    f32x4 acc = fl[0] + fl[1];
#pragma unroll
    for (int k = 0; k < 16; k++) acc += tp[k];
#pragma unroll
    for (int c = 0; c < 3; c++) stnt(out + (b * 3 + c) * plane + o, acc * im[c]);
}

// (plane, rowp: strides in floats -- W * H and W for contiguous tensors)
extern "C" int probe_skeleton_loads(void *stream, int bits, int B, int H, int W, int64_t plane, int64_t rowp, const float *in1,
                                    const float *flow, const float *filt, float *out)
{
    const int tx = W / 64, ty = H / 16;
    const unsigned grid = (unsigned)tx * ty * B;
    if (W % 64 || H % 16 || grid % 8) return -1;
#define GO(BITS) hipLaunchKernelGGL((skeleton_loads<BITS>), dim3(grid), dim3(256), 0, (hipStream_t)stream, W, H, plane, rowp, in1, flow, \
                                    filt, out, tx, ty)
    switch (bits) {
    case 0: GO(0); break; case 1: GO(1); break; case 2: GO(2); break; case 3: GO(3); break;
    case 4: GO(4); break; case 5: GO(5); break; case 6: GO(6); break; case 7: GO(7); break;
    default: return -1;
    }
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// PHASED STORES: every workgroup holds its results until the chip-wide "write window" opens -- the last `win` ticks of every
// period of `per` ticks of the constant 100 MHz clock all CUs share (s_memrealtime) -- so that HBM sees its stores in bursts
// instead of one eighth of its traffic at any time.  (A question to the memory system, asked on the skeleton first.)
// MODE 0: the stores wait for the window; 1: the workgroup's START (its loads) waits for the window, stores as they come;
// 2: the loads wait for the window and the stores for the middle of the period.
template <int LX, int MODE>
__global__ __launch_bounds__(256, 2) void skeleton_phased(int W, int H, int64_t plane, const float *__restrict__ in1,
                                                          const float *__restrict__ flow, const float *__restrict__ filt,
                                                          float *__restrict__ out, int tiles_x, int tiles_y, int walk,
                                                          unsigned per, unsigned win)
{
    if (MODE >= 1) while ((unsigned)(__builtin_amdgcn_s_memrealtime() % per) < per - win) __builtin_amdgcn_s_sleep(8);
    const Tile t = walk_tile(walk, 0, blockIdx.x, gridDim.x, tiles_x, tiles_y);
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
    f32x4 v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) v[c] = acc * im[c];
    asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));           // results are complete: now wait for the window
    if (MODE == 0) while ((unsigned)(__builtin_amdgcn_s_memrealtime() % per) < per - win) __builtin_amdgcn_s_sleep(8);
    if (MODE == 2) while ((unsigned)((__builtin_amdgcn_s_memrealtime() + per / 2) % per) < per - win) __builtin_amdgcn_s_sleep(8);
    if (MODE >= 3) {
        // nobody waits: write-back (cached) stores leave the dirty lines in the XCD's L2, and a workgroup that finishes inside
        // the window asks its L2 to write everything back (buffer_wbl2) -- the stores reach HBM in bursts.  MODE 4: plain stores
        // and no write-back request at all (the control)
#pragma unroll
        for (int c = 0; c < 3; c++) *reinterpret_cast<f32x4 *>(out + (b * 3 + c) * plane + o) = v[c];
        if (MODE == 3 && (unsigned)(__builtin_amdgcn_s_memrealtime() % per) >= per - win) {
            asm volatile("s_waitcnt vmcnt(0)\n\tbuffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
        }
        return;
    }
#pragma unroll
    for (int c = 0; c < 3; c++) stnt(out + (b * 3 + c) * plane + o, v[c]);
}

// PHASED STORES WITHOUT WAITING: persistent workgroups (grid = what fits the chip) that keep up to NP finished tiles' results in
// registers and go on reading the next tile; everything held is stored when the write window opens (or when NP are held: then
// the workgroup does wait).  NP = 0: a plain persistent loop, stores as they come.
// PAUSE: a workgroup that stored in the window also waits for the window to CLOSE before it reads again (reads and writes
// separated in time chip-wide, not just the stores clustered).
template <int NP, bool PAUSE = false>
__global__ __launch_bounds__(256, 2) void skeleton_persistent(int W, int H, int64_t plane, const float *__restrict__ in1,
                                                              const float *__restrict__ flow, const float *__restrict__ filt,
                                                              float *__restrict__ out, int tiles_x, int tiles_y, unsigned ntiles,
                                                              unsigned per, unsigned win)
{
    constexpr int LX = 16, NQ = NP > 0 ? NP : 1;
    f32x4 pend[NQ][3];
    int64_t poff[NQ];
    int np = 0;
    for (unsigned vb = blockIdx.x; vb < ntiles; vb += gridDim.x) {
        const Tile t = walk_tile(2, 0, vb, ntiles, tiles_x, tiles_y);
        const int b = t.b;
        const int x = t.tx * 4 * LX + 4 * (threadIdx.x % LX), y = t.ty * (256 / LX) + threadIdx.x / LX;
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
        f32x4 v[3];
#pragma unroll
        for (int c = 0; c < 3; c++) v[c] = acc * im[c];
        asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]));
        const int64_t ob = (int64_t)b * 3 * plane + o;
        if (NP == 0) {
#pragma unroll
            for (int c = 0; c < 3; c++) stnt(out + ob + c * plane, v[c]);
            continue;
        }
        bool open = (unsigned)__builtin_amdgcn_s_memrealtime() % per >= per - win;
        if (np == NP && !open) {
            while ((unsigned)__builtin_amdgcn_s_memrealtime() % per < per - win) __builtin_amdgcn_s_sleep(4);
            open = true;
        }
        if (open) {
#pragma unroll
            for (int q = 0; q < NQ; q++)
                if (q < np) {
#pragma unroll
                    for (int c = 0; c < 3; c++) stnt(out + poff[q] + c * plane, pend[q][c]);
                }
            np = 0;
#pragma unroll
            for (int c = 0; c < 3; c++) stnt(out + ob + c * plane, v[c]);
            if (PAUSE) while ((unsigned)__builtin_amdgcn_s_memrealtime() % per >= per - win) __builtin_amdgcn_s_sleep(2);
        } else {
#pragma unroll
            for (int q = 0; q < NQ; q++)
                if (q == np) {
                    poff[q] = ob;
#pragma unroll
                    for (int c = 0; c < 3; c++) pend[q][c] = v[c];
                }
            np++;
        }
    }
    if (NP > 0 && np > 0) {
        while ((unsigned)__builtin_amdgcn_s_memrealtime() % per < per - win) __builtin_amdgcn_s_sleep(4);
#pragma unroll
        for (int q = 0; q < NQ; q++)
            if (q < np) {
#pragma unroll
                for (int c = 0; c < 3; c++) stnt(out + poff[q] + c * plane, pend[q][c]);
            }
    }
}

extern "C" int probe_skeleton_persistent(void *stream, int np, int wg_per_cu, int per, int win, int B, int H, int W,
                                         const float *in1, const float *flow, const float *filt, float *out)
{
    const int tx = W / 64, ty = H / 16;
    const unsigned ntiles = (unsigned)tx * ty * B, grid = 256u * wg_per_cu;
    if (W % 64 || H % 16 || ntiles % 8 || per <= 0 || win <= 0 || win > per) return -1;
#define GO(NP) hipLaunchKernelGGL((skeleton_persistent<NP>), dim3(grid), dim3(256), 0, (hipStream_t)stream, W, H, (int64_t)W * H, \
                                  in1, flow, filt, out, tx, ty, ntiles, (unsigned)per, (unsigned)win)
    if (np == 0) GO(0); else if (np == 1) GO(1); else if (np == 2) GO(2); else if (np == 3) GO(3); else if (np == 4) GO(4);
    else if (np >= 101 && np <= 104) {
#define GOP(NP) hipLaunchKernelGGL((skeleton_persistent<NP, true>), dim3(grid), dim3(256), 0, (hipStream_t)stream, W, H, (int64_t)W * H, \
                                   in1, flow, filt, out, tx, ty, ntiles, (unsigned)per, (unsigned)win)
        if (np == 101) GOP(1); else if (np == 102) GOP(2); else if (np == 103) GOP(3); else GOP(4);
#undef GOP
    } else return -1;
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int probe_skeleton_phased(void *stream, int lx, int walk, int per, int win, int B, int H, int W, const float *in1,
                                     const float *flow, const float *filt, float *out)
{
    const int mode = lx >> 8;               // (lx = 16 + 256 * mode)
    lx &= 255;
    const int tx = (W + 4 * lx - 1) / (4 * lx), ty = (H + 256 / lx - 1) / (256 / lx);
    const unsigned grid = (unsigned)tx * ty * B;
    if (grid % 8 || lx != 16 || per <= 0 || win <= 0 || win > per) return -1;
#define GO(MODE) hipLaunchKernelGGL((skeleton_phased<16, MODE>), dim3(grid), dim3(256), 0, (hipStream_t)stream, W, H, (int64_t)W * H, \
                                    in1, flow, filt, out, tx, ty, walk, (unsigned)per, (unsigned)win)
    if (mode == 0) GO(0); else if (mode == 1) GO(1); else if (mode == 2) GO(2); else if (mode == 3) GO(3); else GO(4);
#undef GO
    return hipGetLastError() == hipSuccess ? 0 : -1;
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
