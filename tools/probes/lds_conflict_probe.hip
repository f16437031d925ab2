// tools/probes/lds_conflict_probe.hip -- what do SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE say for known LDS access patterns?
// Every kernel: 256 lanes, 48 KiB of pixel quads, each lane issues 64 x 16 ds_read_b128 (or the write / atomic pattern) and
// keeps the sum alive.  Patterns:
//   0  linear: lane l reads slot (l + 64 it) & mask                    -- conflict-free by construction
//   1  stride 4, no swizzle: slot 4 * (l % 16) + 96 * (l / 16)         -- 4-way conflicts
//   2  the tiled kernels' gather for a UNIFORM flow: row l / 16 (+k), column 4 * (l % 16) + j + m, XOR-swizzled, pitch 96
//   3  as 2 with pitch 80
//   4  as 2, but the column of every lane perturbed by a pseudo-random 0..3 (a rough flow)
//   5  ds_write_b128 in the staging pattern (lane q of a row writes pixel 4 q + i)
//   6  as 2 with a smooth drift: the column offset grows by one every five quads along the row, and differs by one between rows
//   7  TRANSPOSED layout (column c of a row at (c & 3) * 32 + (c >> 2), row pitch 128 slots), uniform flow
//   8  transposed, the random 0..3 perturbation of 4
//   9  transposed, the smooth drift of 6
//  10  transposed, ds_write_b128 staging pattern
//  11-13  PLANAR layout (one float per pixel and channel, row pitch 76 floats; what an LDS-DMA staging would leave): a lane reads
//         the four consecutive floats of a window row, 4-byte aligned -- two ds_read2_b32; uniform flow / random 0..3 / drift
//  14-16  the same three as ONE ds_read_b128 at that 4-byte alignment (inline asm; does the hardware take it, and at what rate?)
//  17-19  the same three with a row pitch of 80 floats, two ds_read2_b32
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int swz_col(int c) { return c ^ ((c >> 2) & 15); }

template <int P>
__global__ __launch_bounds__(256) void lds_probe(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    f32x4 *tile = reinterpret_cast<f32x4 *>(smem);
    const int l = threadIdx.x;
    for (int i = l; i < 3072; i += 256) tile[i] = f32x4{(float)i, 1.f, 2.f, 3.f};
    __syncthreads();
    const int pitch = P == 3 ? 80 : 96;
    const int row = l / 16, q = l % 16;
    const int rnd = (l * 2654435761u >> 13) & 3;
    const int drift = q / 5 + (row & 1);
    auto tr = [](int c) { return (c & 3) * 32 + (c >> 2); };
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (P >= 11) {
        const float *pl = reinterpret_cast<const float *>(smem);
        const int pf = P >= 17 ? 80 : 76;
        const int mode = (P - 11) % 3;                             // 0 uniform, 1 random, 2 drift
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int ch = 0; ch < 4; ch++) {                   // four "channels": planes 3040 floats apart
                    const int c = 4 * q + 1 + (it & 3) + (mode == 1 ? rnd : 0) + (mode == 2 ? drift : 0);
                    const int fi = ch * 3040 + (row + k) * pf + c;
                    if (P >= 14 && P <= 16) {
                        f32x4 v;
                        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((unsigned)(fi * 4)));
                        acc += v;
                    } else {
                        acc[0] += pl[fi];  acc[1] += pl[fi + 1];  acc[2] += pl[fi + 2];  acc[3] += pl[fi + 3];
                    }
                }
            asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
        }
        out[blockIdx.x * 256 + l] = acc[0] + acc[1] + acc[2] + acc[3];
        return;
    }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int m = 0; m < 4; m++) {
                int idx;
                if (P == 0) idx = (l + 64 * (k * 4 + m) + it) & 2047;
                else if (P == 1) idx = (4 * q + 96 * row + k * 96 + m + it) & 2047;
                else if (P >= 7) {
                    const int c = 4 * q + 1 + m + (it & 3) + (P == 8 ? rnd : 0) + (P == 9 ? drift : 0);
                    idx = ((row + k) * 128 + tr(c)) & 2047;        // (24 KiB of the 48: rows wrap, the banks do not care)
                } else {
                    const int c = 4 * q + 1 + m + (it & 3) + (P == 4 ? rnd : 0) + (P == 6 ? drift : 0);
                    idx = (row + k) * pitch + swz_col(c);
                }
                if (P == 5) {
                    tile[(row + k) * pitch + swz_col(4 * q + m)] = acc + (float)it;
                } else if (P == 10) {
                    tile[((row + k) * 128 + tr(4 * q + m)) & 2047] = acc + (float)it;
                } else {
                    acc += tile[idx];
                }
            }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
    }
    if (P == 5 || P == 10) { __syncthreads(); acc = tile[l]; }
    out[blockIdx.x * 256 + l] = acc[0] + acc[1] + acc[2] + acc[3];
}

extern "C" int lds_probe_run(void *stream, int pattern, float *out, int blocks, int iters)
{
    hipStream_t s = (hipStream_t)stream;
#define RUN(P) hipLaunchKernelGGL(lds_probe<P>, dim3(blocks), dim3(256), 49152, s, out, iters)
    switch (pattern) {
    case 0: RUN(0); break;
    case 1: RUN(1); break;
    case 2: RUN(2); break;
    case 3: RUN(3); break;
    case 4: RUN(4); break;
    case 5: RUN(5); break;
    case 6: RUN(6); break;
    case 7: RUN(7); break;
    case 8: RUN(8); break;
    case 9: RUN(9); break;
    case 10: RUN(10); break;
    case 11: RUN(11); break;
    case 12: RUN(12); break;
    case 13: RUN(13); break;
    case 14: RUN(14); break;
    case 15: RUN(15); break;
    case 16: RUN(16); break;
    case 17: RUN(17); break;
    case 18: RUN(18); break;
    case 19: RUN(19); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
