// calibration.hip -- the streaming rates THIS device gives the access pattern of the tiled kernels: 16 bytes per lane,
// non-temporal, workgroups dealt to the XCDs in contiguous runs (xcd_chunked_id, the walk of memc_common.hpp), R read
// streams per written stream.  Not an operator of the reference: bench.py reports the rate as `roofline.achievable_peak`
// beside the 8 TB/s specification (rounds 2-3 used torch's copy_, which the headline kernel outran).
//
//     dst[i] = src[i] + src[n + i] + ... + src[(R - 1) n + i]      i in [0, n) float4, R = reads_per_write (1 .. 8)
//
// R = 1 is a copy (50 % reads); R = 7 has the read : write mix of the RGB adaptive warp (21 float4 read, 3 written).
#include "memc_common.hpp"
#include "memc_tile.hpp"
#include "../../include/memc_warp.h"

namespace memc {

constexpr int kCalUnroll = 4;                  // float4 per lane and stream: 16 KiB per stream and workgroup

template <int R>
__global__ __launch_bounds__(256) void calibration_stream(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, int64_t n)
{
    const int64_t chunk = (int64_t)xcd_chunked_id(blockIdx.x, gridDim.x) * (256 * kCalUnroll);
    f32x4 v[kCalUnroll][R];
#pragma unroll
    for (int u = 0; u < kCalUnroll; u++) {
        const int64_t i = min(chunk + u * 256 + threadIdx.x, n - 1);   // (the tail re-reads the last element)
#pragma unroll
        for (int r = 0; r < R; r++) v[u][r] = __builtin_nontemporal_load(src + r * n + i);
    }
#pragma unroll
    for (int u = 0; u < kCalUnroll; u++) {
        const int64_t i = chunk + u * 256 + threadIdx.x;
        f32x4 s = v[u][0];
#pragma unroll
        for (int r = 1; r < R; r++) s += v[u][r];
        if (i < n) __builtin_nontemporal_store(s, dst + i);
    }
}

}  // namespace memc

using namespace memc;

extern "C" int memc_calibration_stream(memc_stream_t stream, const float *src, float *dst, int64_t n_float4,
                                       int reads_per_write)
{
    if (n_float4 <= 0) return 0;
    if (reads_per_write < 1 || reads_per_write > 8 || !src || !dst) return -1;
    if (reinterpret_cast<uintptr_t>(src) % 16 || reinterpret_cast<uintptr_t>(dst) % 16) return -1;
    const int64_t per_wg = 256 * kCalUnroll;
    const int64_t nwg = (n_float4 + per_wg - 1) / per_wg;
    if (nwg > 0x7fffffff) return -1;
    const f32x4 *s = reinterpret_cast<const f32x4 *>(src);
    f32x4 *d = reinterpret_cast<f32x4 *>(dst);
#define MEMC_CAL(R) \
    case R: hipLaunchKernelGGL(calibration_stream<R>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, s, d, n_float4); break
    switch (reads_per_write) {
        MEMC_CAL(1); MEMC_CAL(2); MEMC_CAL(3); MEMC_CAL(4); MEMC_CAL(5); MEMC_CAL(6); MEMC_CAL(7); MEMC_CAL(8);
    }
#undef MEMC_CAL
    return launch_status();
}
