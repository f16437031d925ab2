// Which XCD does workgroup b of a 1-D grid land on?  Reads XCC_ID (s_getreg_b32) in every workgroup.
// The XCD-aware tile walks (memc_common.hpp: xcd_chunked_id) assume b % 8.
#include <hip/hip_runtime.h>
__global__ void xcc_of_block(unsigned *out)
{
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x] = x;
    }
}
extern "C" int probe_xcc(void *stream, void *out, int nblocks, int threads, int lds_bytes)
{
    hipLaunchKernelGGL(xcc_of_block, dim3(nblocks), dim3(threads), lds_bytes, (hipStream_t)stream, (unsigned *)out);
    return (int)hipGetLastError();
}
