// tools/probes/r4_cache/r4_cache_probe.hip -- round 4's scratch cache (memc_scratch_r4.hpp: the header as round 4 shipped it, kept
// here as the specimen) driven directly: which block do two streams get?
//   hipcc --offload-arch=gfx950 -O2 -o tools/probes/r4_cache/r4_cache_probe tools/probes/r4_cache/r4_cache_probe.hip
#include "memc_scratch_r4.hpp"

#include <cstdio>

int main()
{
    hipStream_t s[4];
    for (int k = 0; k < 4; k++)
        if (hipStreamCreateWithFlags(&s[k], hipStreamNonBlocking) != hipSuccess) return 2;
    for (int round = 0; round < 3; round++) {
        for (int k = 0; k < 4; k++) {
            memc::CallScratch c;
            const bool ok = c.alloc(11808, s[k]);
            printf("round %d stream %d (%p): ok %d block %p from the stream's cache entry: %d (entry %p, entry.stream %p, entry.bytes %zu)\n",
                   round, k, (void *)s[k], (int)ok, c.p, c.claimed != nullptr, (void *)c.claimed,
                   c.claimed ? (void *)c.claimed->stream : nullptr, c.claimed ? c.claimed->bytes : (size_t)0);
        }
    }
    return 0;
}
