// tools/probes/scratch_streams.hip -- does a kernel that uses PRIVATE SCRATCH (ScratchSize > 0) disturb, or get disturbed by,
// work on other HIP streams?  The minimal reproducer the round-4 review asked for: round 4 saw a wrong
// result on a stream that ran next to an experimental build of proj_owner_far which spilled 76 bytes per lane.
//
//   hipcc --offload-arch=gfx950 -O2 -mllvm -disable-promote-alloca-to-vector -mllvm -disable-promote-alloca-to-lds \
//         -o tools/probes/scratch_streams tools/probes/scratch_streams.hip      (144 B and 4112 B of scratch per lane)
//   tools/probes/scratch_streams [rounds=10000]
//
// Every stream runs the projection's launch SHAPE:  producer (no scratch: fills a buffer with the round's pattern)  ->
// scratch kernel  ->  consumer (no scratch: checks the producer's buffer, adds up what the scratch kernel reported).
// The scratch kernel keeps a per-lane private array that the compiler cannot promote to registers (indexed by a value loaded
// at run time), fills it with a pattern that names stream, round, workgroup and lane, idles, and re-reads it.  Modes:
//   mode 0: every stream's scratch kernel works                       (scratch users next to scratch users)
//   mode 1: only the LAST stream's works, the others return at once   (the projection on a stream without far sources: a
//           dispatch that ASKS for scratch and never touches it, next to one that does)
//   mode 2: as 1, and the idle streams' scratch kernel is replaced by a register-only one   (control: no scratch there)
// each with 2, 4 and 8 streams, with the scratch array at 128 B and at 4 KiB per lane.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));       \
            exit(2);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int kN = 1 << 18;                    // elements of a stream's buffer

__global__ void producer(unsigned *buf, unsigned pattern)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kN; i += gridDim.x * blockDim.x) buf[i] = pattern ^ (unsigned)i;
}

// WORDS private words per lane, indexed through `perm` (device memory: the compiler cannot resolve the index)
template <int WORDS>
__global__ __launch_bounds__(256) void scratch_user(const int *__restrict__ perm, const int *__restrict__ go, unsigned pattern,
                                                    int spin, unsigned *__restrict__ errors)
{
    if (*go == 0) return;                      // the dispatch asked for scratch; this stream has nothing to do
    unsigned a[WORDS];
    const unsigned me = pattern ^ (blockIdx.x * 1315423911u) ^ (threadIdx.x * 2654435761u);
    for (int i = 0; i < WORDS; i++) a[perm[i]] = me + (unsigned)i;
    unsigned acc = 0;
    for (int s = 0; s < spin; s++) {           // idle with the array live, touching it (keeps it in scratch, gives others time)
        acc += a[perm[s % WORDS]];
        __builtin_amdgcn_s_sleep(8);
    }
    unsigned bad = 0;
    for (int i = 0; i < WORDS; i++) bad += a[perm[i]] != me + (unsigned)i;
    if (bad) atomicAdd(errors, bad);
    if (acc == 0x12345u) errors[1] = acc;      // (keeps acc alive)
}

__global__ __launch_bounds__(256) void register_only(const int *__restrict__ go, unsigned *__restrict__ errors)
{
    if (*go == 0) return;
    errors[1] = 1;
}

__global__ void consumer(const unsigned *buf, unsigned pattern, const unsigned *errors, unsigned *result)
{
    unsigned bad = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < kN; i += gridDim.x * blockDim.x) bad += buf[i] != (pattern ^ (unsigned)i);
    if (bad) atomicAdd(result, bad);
    if (blockIdx.x == 0 && threadIdx.x == 0 && errors[0]) atomicAdd(result + 1, errors[0]);
}

struct Lane {
    hipStream_t s;
    unsigned *buf, *errors, *result;
    int *go;
};

template <int WORDS>
static long run(int nstreams, int mode, int rounds, const int *perm)
{
    std::vector<Lane> ln(nstreams);
    for (int k = 0; k < nstreams; k++) {
        CHECK(hipStreamCreateWithFlags(&ln[k].s, hipStreamNonBlocking));
        CHECK(hipMalloc(&ln[k].buf, kN * sizeof(unsigned)));
        CHECK(hipMalloc(&ln[k].errors, 2 * sizeof(unsigned)));
        CHECK(hipMalloc(&ln[k].result, 2 * sizeof(unsigned)));
        CHECK(hipMalloc(&ln[k].go, sizeof(int)));
        CHECK(hipMemset(ln[k].errors, 0, 2 * sizeof(unsigned)));
        CHECK(hipMemset(ln[k].result, 0, 2 * sizeof(unsigned)));
        const int go = (mode == 0 || k == nstreams - 1) ? 1 : 0;
        CHECK(hipMemcpy(ln[k].go, &go, sizeof(int), hipMemcpyHostToDevice));
    }
    CHECK(hipDeviceSynchronize());
    for (int r = 0; r < rounds; r++) {
        for (int k = 0; k < nstreams; k++) {
            const unsigned pattern = 0x9E3779B1u * (unsigned)(r * 16 + k + 1);
            const bool works = mode == 0 || k == nstreams - 1;
            hipLaunchKernelGGL(producer, dim3(64), dim3(256), 0, ln[k].s, ln[k].buf, pattern);
            if (mode == 2 && !works)
                hipLaunchKernelGGL(register_only, dim3(512), dim3(256), 0, ln[k].s, ln[k].go, ln[k].errors);
            else
                hipLaunchKernelGGL((scratch_user<WORDS>), dim3(works ? 2048 : 512), dim3(256), 0, ln[k].s, perm, ln[k].go, pattern,
                                   works ? 40 : 0, ln[k].errors);
            hipLaunchKernelGGL(consumer, dim3(64), dim3(256), 0, ln[k].s, ln[k].buf, pattern, ln[k].errors, ln[k].result);
        }
        if (r % 8 == 7) CHECK(hipDeviceSynchronize());   // (the projection test synchronises every round; keep queues shallow)
    }
    CHECK(hipDeviceSynchronize());
    CHECK(hipGetLastError());
    long bad = 0;
    for (int k = 0; k < nstreams; k++) {
        unsigned res[2];
        CHECK(hipMemcpy(res, ln[k].result, sizeof(res), hipMemcpyDeviceToHost));
        if (res[0] || res[1]) printf("    stream %d of %d: %u producer words wrong at the consumer, %u private words wrong\n", k, nstreams, res[0], res[1]);
        bad += res[0] + res[1];
        CHECK(hipFree(ln[k].buf));  CHECK(hipFree(ln[k].errors));  CHECK(hipFree(ln[k].result));  CHECK(hipFree(ln[k].go));
        CHECK(hipStreamDestroy(ln[k].s));
    }
    return bad;
}

// Round 4's scratch cache took one stream-ordered allocation per STREAM from a private pool (hipMallocFromPoolAsync) and
// kept it.  What does the pool hand two different streams that both hold their allocation?
static void pool_probe()
{
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = 0;
    hipMemPool_t pool = nullptr;
    CHECK(hipMemPoolCreate(&pool, &props));
    uint64_t keep = UINT64_MAX;
    CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    hipStream_t s[3];
    for (int k = 0; k < 3; k++) CHECK(hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking));
    for (size_t bytes : {(size_t)14760, (size_t)1 << 20, (size_t)40 << 20}) {
        void *p[3] = {};
        for (int k = 0; k < 3; k++) CHECK(hipMallocFromPoolAsync(&p[k], bytes, pool, s[k]));
        printf("pool probe: %9zu B on three streams, none freed: %p %p %p%s\n", bytes, p[0], p[1], p[2],
               (p[0] == p[1] || p[1] == p[2] || p[0] == p[2]) ? "   <-- THE SAME BLOCK TWICE" : "");
        for (int k = 0; k < 3; k++) CHECK(hipStreamSynchronize(s[k]));
        void *q[3] = {};
        for (int k = 0; k < 3; k++) CHECK(hipMallocFromPoolAsync(&q[k], bytes, pool, s[k]));
        printf("            a second round, the first still held:      %p %p %p\n", q[0], q[1], q[2]);
        for (int k = 0; k < 3; k++) {
            CHECK(hipFreeAsync(p[k], s[k]));
            CHECK(hipFreeAsync(q[k], s[k]));
        }
        CHECK(hipDeviceSynchronize());
    }
    for (int k = 0; k < 3; k++) CHECK(hipStreamDestroy(s[k]));
}

// ... and what does it hand stream B when stream A has FREED its block (hipFreeAsync, in stream order) but A's kernels that
// use the block are still running?  The runtime may give B the same memory only behind a dependency on A's free.  `hold`
// fills the block, idles (~spin x 0.25 us) and verifies it; `scribble` on B overwrites whatever B was given at once.
__global__ void hold(unsigned *p, int n, unsigned pat, int spin, unsigned *bad)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = pat ^ (unsigned)i;
    for (int s = 0; s < spin; s++) __builtin_amdgcn_s_sleep(8);
    __threadfence();
    unsigned b = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) b += p[i] != (pat ^ (unsigned)i);
    if (b) atomicAdd(bad, b);
}
__global__ void scribble(unsigned *q, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) q[i] = 0xDEADBEEFu;
}
static void reuse_probe(int rounds)
{
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = 0;
    hipMemPool_t pool = nullptr;
    CHECK(hipMemPoolCreate(&pool, &props));
    uint64_t keep = UINT64_MAX;
    CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
    hipStream_t A, B;
    CHECK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
    unsigned *bad = nullptr;
    CHECK(hipMalloc(&bad, sizeof(unsigned)));
    CHECK(hipMemset(bad, 0, sizeof(unsigned)));
    const int n = 4096;                        // 16 KiB: the size of a small projection call's block
    for (int spin : {0, 400, 4000}) {          // A's kernel takes ~2 us / ~0.1 ms / ~1 ms
        int same = 0;
        CHECK(hipMemset(bad, 0, sizeof(unsigned)));
        for (int r = 0; r < rounds; r++) {
            void *p = nullptr, *q = nullptr;
            CHECK(hipMallocFromPoolAsync(&p, n * sizeof(unsigned), pool, A));
            hipLaunchKernelGGL(hold, dim3(4), dim3(256), 0, A, (unsigned *)p, n, 0x9E3779B1u * (unsigned)(r + 1), spin, bad);
            CHECK(hipFreeAsync(p, A));         // (what rounds 2-3 of the library did at the end of every call)
            CHECK(hipMallocFromPoolAsync(&q, n * sizeof(unsigned), pool, B));
            hipLaunchKernelGGL(scribble, dim3(4), dim3(256), 0, B, (unsigned *)q, n);
            CHECK(hipFreeAsync(q, B));
            same += p == q;
            if (r % 16 == 15) CHECK(hipDeviceSynchronize());
        }
        CHECK(hipDeviceSynchronize());
        unsigned h = 0;
        CHECK(hipMemcpy(&h, bad, sizeof(h), hipMemcpyDeviceToHost));
        printf("reuse probe: A holds its block ~%4d sleeps, frees it in stream order; B allocates at once: B got A's block in %d of %d "
               "rounds; words of A's block overwritten while A still used it: %u\n", spin, same, rounds, h);
    }
    CHECK(hipFree(bad));
    CHECK(hipStreamDestroy(A));
    CHECK(hipStreamDestroy(B));
}

int main(int argc, char **argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 10000;
    pool_probe();
    reuse_probe(2000);
    if (rounds <= 0) return 0;
    int h_perm[1024];
    for (int i = 0; i < 1024; i++) h_perm[i] = i;
    int *perm = nullptr;
    CHECK(hipMalloc(&perm, sizeof(h_perm)));
    CHECK(hipMemcpy(perm, h_perm, sizeof(h_perm), hipMemcpyHostToDevice));
    hipFuncAttributes fa;
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(scratch_user<32>)));
    printf("scratch_user<32>: %zu B of private memory per lane, %d registers;  ", fa.localSizeBytes, fa.numRegs);
    CHECK(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(scratch_user<1024>)));
    printf("scratch_user<1024>: %zu B per lane\n", fa.localSizeBytes);
    long total = 0;
    for (int words : {32, 1024})
        for (int mode = 0; mode < 3; mode++)
            for (int ns : {2, 4, 8}) {
                const int rr = words == 32 ? rounds : rounds / 10;
                const long bad = words == 32 ? run<32>(ns, mode, rr, perm) : run<1024>(ns, mode, rr, perm);
                printf("%5d B/lane  mode %d  %d streams  %6d rounds: %ld wrong words\n", 4 * words, mode, ns, rr, bad);
                fflush(stdout);
                total += bad;
            }
    printf("TOTAL wrong words: %ld\n", total);
    return total ? 1 : 0;
}
