// io_skeleton.hip -- measurement probe, NOT part of libmemc_hip.so.
//
// The streaming skeleton of the adaptive-warp forward with all data-dependent work removed: per output site
// read 2 flow + 16 tap values + C image values at the site's own position, write C outputs -- exactly the
// kernel's ALGORITHMIC traffic (96 B/site at C=3) with no gathers, no LDS and no barriers.  Its time is the
// ceiling any real kernel with this I/O pattern can reach on the device; variants change only the tile shape,
// the cache policy of the streams and the residency.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ f32x4 ld4(const float *p)
{
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p));
    return *reinterpret_cast<const f32x4 *>(p);
}
template <bool NT>
__device__ __forceinline__ void st4(float *p, f32x4 v)
{
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(p));
    else *reinterpret_cast<f32x4 *>(p) = v;
}

__device__ __forceinline__ unsigned xcd_chunked_id(unsigned bid, unsigned nwg)
{
    const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8, idx = bid / 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// LX lanes per tile row (4 sites each); 256 threads; MINW = min waves per SIMD for the register allocator
template <int LX, bool NT, bool XCD, int MINW>
__global__ __launch_bounds__(256, MINW) void skeleton(int W, int H, int64_t plane, const float *__restrict__ in1,
                                                      const float *__restrict__ flow, const float *__restrict__ filt,
                                                      float *__restrict__ out, int tiles_x, int tiles_y)
{
    const unsigned t = XCD ? xcd_chunked_id(blockIdx.x, gridDim.x) : blockIdx.x;
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
    const int x = tx * 4 * LX + 4 * (threadIdx.x % LX), y = ty * (256 / LX) + threadIdx.x / LX;
    if (x >= W || y >= H) return;
    const int64_t o = (int64_t)y * W + x;
    f32x4 acc = ld4<NT>(flow + (b * 2 + 0) * plane + o) + ld4<NT>(flow + (b * 2 + 1) * plane + o);
    f32x4 tp[16];
#pragma unroll
    for (int k = 0; k < 16; k++) tp[k] = ld4<NT>(filt + (b * 16 + k) * plane + o);
#pragma unroll
    for (int k = 0; k < 16; k++) acc += tp[k];
#pragma unroll
    for (int c = 0; c < 3; c++) st4<NT>(out + (b * 3 + c) * plane + o, acc * ld4<false>(in1 + (b * 3 + c) * plane + o));
}

// flat grid-stride float4 copy of n4 float4s (the classic bandwidth test), for calibration
template <bool NT>
__global__ __launch_bounds__(256) void copy4(const f32x4 *__restrict__ a, f32x4 *__restrict__ b, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 v = NT ? __builtin_nontemporal_load(a + i) : a[i];
        if (NT) __builtin_nontemporal_store(v, b + i); else b[i] = v;
    }
}
// read-only: sums into a tiny output (tests pure read bandwidth)
__global__ __launch_bounds__(256) void read4(const f32x4 *__restrict__ a, float *__restrict__ sink, int64_t n4)
{
    f32x4 acc = {0, 0, 0, 0};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
        acc += __builtin_nontemporal_load(a + i);
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

// fp32 global atomic-add throughput: every lane adds to its own element (n elements, each exactly once per pass),
// REP passes shifted by `shift` elements (shift 0: same lines again back to back; 1: the neighbouring element,
// i.e. the same cache lines, as the 4-neighbour splat does).
__global__ __launch_bounds__(256) void atomics(float *__restrict__ dst, int64_t n, int rep, int shift)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        for (int r = 0; r < rep; r++) (void)unsafeAtomicAdd(dst + ((i + (int64_t)r * shift) % n), 1.0f);
}
// the same update done as plain read-modify-write (no atomicity): what a non-atomic owner would pay
__global__ __launch_bounds__(256) void rmw(f32x4 *__restrict__ dst, int64_t n4)
{
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) dst[i] += 1.0f;
}
extern "C" int probe_atomics(void *stream, int variant, int blocks, float *dst, int64_t n, int rep, int shift)
{
    if (variant == 0) hipLaunchKernelGGL(atomics, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dst, n, rep, shift);
    else hipLaunchKernelGGL(rmw, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (f32x4 *)dst, n / 4);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// LDS fp32 atomic throughput: every lane does REP ds_add_f32 to address (lane * stride + r * rstep) % 8192.
__global__ __launch_bounds__(256) void lds_atomics(float *__restrict__ sink, int rep, int stride, int rstep, int mode)
{
    __shared__ float buf[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) buf[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x;
    for (int r = 0; r < rep; r++) {
        const int a = (lane * stride + r * rstep) & 8191;
        if (mode == 0) (void)unsafeAtomicAdd(buf + a, 1.0f);                 // ds_add_f32
        else if (mode == 1) (void)atomicAdd(reinterpret_cast<unsigned *>(buf) + a, 1u);   // ds_add_u32
        else buf[a] += 1.0f;                                                  // plain read-modify-write
    }
    __syncthreads();
    if (buf[threadIdx.x] == 12345.f) sink[0] = 1.f;
}
__global__ __launch_bounds__(256) void lds_atomics64(float *__restrict__ sink, int rep, int stride, int rstep, int mode)
{
    __shared__ double buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x;
    for (int r = 0; r < rep; r++) {
        const int a = (lane * stride + r * rstep) & 4095;
        if (mode == 3) (void)unsafeAtomicAdd(buf + a, 1.0);                                            // ds_add_f64
        else (void)atomicAdd(reinterpret_cast<unsigned long long *>(buf) + a, 1ull);                   // ds_add_u64
    }
    __syncthreads();
    if (buf[threadIdx.x] == 12345.0) sink[0] = 1.f;
}
// ds_add_f64 under the access patterns of the tiled backward kernels.  pattern:
//   0 consecutive cells, all lanes          1 consecutive, every 4th lane active      2 consecutive, every 8th
//   3 kernel layout (row = lane/16, col = 4*(lane%16), pitch 97, bit-1 swizzle), no flow noise
//   4 kernel layout with +-2 cells of per-lane noise     5 as 4 without the swizzle / with pitch 96
//   6 as 3 with half of the lanes inactive (lane%2)      7 kernel layout, col = lane%16 * 4 + (lane/16) (rows share a line)
__global__ __launch_bounds__(256) void lds_atomics_pattern(float *__restrict__ sink, int rep, int pattern)
{
    __shared__ double buf[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) buf[i] = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned h = (lane * 2654435761u) >> 13;
    bool active = true;
    int base;
    if (pattern <= 2) {
        base = wave * 64 + lane;
        if (pattern == 1) { active = (lane & 3) == 0; base = wave * 64 + (lane >> 2); }
        if (pattern == 2) { active = (lane & 7) == 0; base = wave * 64 + (lane >> 3); }
    } else if (pattern == 7) {
        base = wave * 4 * 97 + (lane & 15) * 4 + (lane >> 4);
    } else {
        const int row = wave * 4 + (lane >> 4);
        int col = 4 * (lane & 15) + ((pattern == 4 || pattern == 5) ? (int)(h % 5) : 0);
        const int pitch = pattern == 5 ? 96 : 97;
        if (pattern != 5) col ^= (col >> 4) & 2;
        base = row * pitch + col;
        if (pattern == 6) active = (lane & 1) == 0;
    }
    for (int r = 0; r < rep; r++) {
        const int a = (base + (r & 15) * 194 + (r >> 4)) & 4095;
        if (active) (void)unsafeAtomicAdd(buf + a, 1.0);
    }
    __syncthreads();
    if (buf[threadIdx.x] == 12345.0) sink[0] = 1.f;
}
extern "C" int probe_lds_pattern(void *stream, int blocks, float *sink, int rep, int pattern)
{
    hipLaunchKernelGGL(lds_atomics_pattern, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, rep, pattern);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int probe_lds_atomics(void *stream, int blocks, float *sink, int rep, int stride, int rstep, int mode)
{
    if (mode >= 3) {
        hipLaunchKernelGGL(lds_atomics64, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, rep, stride, rstep, mode);
        return hipGetLastError() == hipSuccess ? 0 : -1;
    }
    hipLaunchKernelGGL(lds_atomics, dim3(blocks), dim3(256), 0, (hipStream_t)stream, sink, rep, stride, rstep, mode);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

#define LAUNCH(LX, NT, XCD, MINW)                                                                            \
    do {                                                                                                     \
        const int tx = (W + 4 * LX - 1) / (4 * LX), ty = (H + 256 / LX - 1) / (256 / LX);                    \
        hipLaunchKernelGGL((skeleton<LX, NT, XCD, MINW>), dim3((unsigned)tx * ty * B), dim3(256), 0,         \
                           (hipStream_t)stream, W, H, (int64_t)W * H, in1, flow, filt, out, tx, ty);         \
    } while (0)

extern "C" int probe_skeleton(void *stream, int variant, int B, int H, int W, const float *in1, const float *flow,
                              const float *filt, float *out)
{
    switch (variant) {
    case 0: LAUNCH(16, true, true, 1); break;     // 64x16 tile, nt streams, XCD-chunked
    case 1: LAUNCH(8, true, true, 1); break;      // 32x32
    case 2: LAUNCH(64, true, true, 1); break;     // 256x4: one wave = 1 KiB contiguous per plane
    case 3: LAUNCH(32, true, true, 1); break;     // 128x8
    case 4: LAUNCH(16, false, true, 1); break;    // default cache policy
    case 5: LAUNCH(16, true, false, 1); break;    // blockIdx order (round-robin over XCDs)
    case 6: LAUNCH(64, true, false, 1); break;
    case 7: LAUNCH(16, true, true, 2); break;     // register cap as the real kernel (2 waves / SIMD)
    case 8: LAUNCH(64, false, false, 1); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

extern "C" int probe_copy(void *stream, int variant, int blocks, const float *a, float *b, int64_t n4)
{
    if (variant == 0) hipLaunchKernelGGL(copy4<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)a, (f32x4 *)b, n4);
    else if (variant == 1) hipLaunchKernelGGL(copy4<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)a, (f32x4 *)b, n4);
    else hipLaunchKernelGGL(read4, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const f32x4 *)a, b, n4);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
