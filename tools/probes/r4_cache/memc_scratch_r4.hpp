// memc_scratch.hpp -- stream-ordered per-call scratch from a private memory pool (host side).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>

namespace memc {

// The forward pass needs a few device words that outlive a kernel (per-image "far source" flags of the fast path)
// and, with hole filling, the filler's carry tables: ONE stream-ordered allocation per call, released in stream
// order before the call returns (hipFreeAsync) -- nothing is shared between calls, streams or threads.  It comes
// from a private memory pool per device (created on first use, kept for the life of the process, release threshold
// "never": with the default threshold a pool hands its memory back at every synchronisation and the next call pays
// for a fresh allocation, measured +200 us); the device's default pool and its attributes are left alone.
inline hipMemPool_t pool_for_device(int dev)
{
    static std::mutex mu;
    static hipMemPool_t pools[64] = {};
    static bool tried[64] = {};
    if (dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    if (!tried[dev]) {
        tried[dev] = true;
        hipMemPoolProps props = {};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = dev;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool) {
            uint64_t keep = UINT64_MAX;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep);
            pools[dev] = pool;
        }
        (void)hipGetLastError();
    }
    return pools[dev];
}

// Scratch of one call.  Round 4 measured what a stream-ordered allocation and its release cost the HOST: ~30 us per call
// (hipMallocFromPoolAsync + hipFreeAsync), a quarter of the projection's GPU time and the whole of its launch latency.
// So the block is CACHED per (device, stream): calls enqueued on one stream run in order, the next call may reuse the
// block the previous one used.  The cache entry is claimed for the duration of the host-side call (two host threads
// enqueueing on the same stream at once do not share: the second takes a stream-ordered allocation of its own, as
// rounds 2-3 did for every call); a block that is too small is released in stream order and replaced.  Blocks live for
// the life of the process (at most one per stream that ever ran a projection forward: 0.8 bytes per pixel of the
// largest call with hole filling).
struct CachedBlock {
    int dev = -1;
    hipStream_t stream = nullptr;
    void *p = nullptr;
    size_t bytes = 0;
    bool used = false, busy = false;
};
constexpr int kScratchSlots = 64;              // streams remembered (a 65th takes per-call allocations, as rounds 2-3 did)
inline std::mutex &scratch_mutex()
{
    static std::mutex mu;
    return mu;
}
// the entry of (dev, stream), claimed for a new pair if there is room; nullptr: no room.  Caller holds scratch_mutex().
// (a plain table, no std::map: the library exports nothing but its C surface -- tests/test_abi.py)
inline CachedBlock *scratch_entry(int dev, hipStream_t s)
{
    static CachedBlock table[kScratchSlots];
    CachedBlock *free_slot = nullptr;
    for (int i = 0; i < kScratchSlots; i++) {
        if (table[i].used && table[i].dev == dev && table[i].stream == s) return &table[i];
        if (!table[i].used && !free_slot) free_slot = &table[i];
    }
    if (free_slot) {
        free_slot->used = true;
        free_slot->dev = dev;
        free_slot->stream = s;
    }
    return free_slot;
}

struct CallScratch {
    void *p = nullptr;
    hipStream_t stream = nullptr;
    CachedBlock *claimed = nullptr;            // the cache entry this call holds (released, not freed, at the end)
    bool alloc(size_t bytes, hipStream_t s)
    {
        stream = s;
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &capture);
        if (capture != hipStreamCaptureStatusNone) return false;      // no allocation inside a stream capture
        int dev = -1;
        if (hipStreamGetDevice(s, &dev) != hipSuccess) {              // the STREAM's device, not the current one
            (void)hipGetLastError();
            if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        }
        hipMemPool_t pool = pool_for_device(dev);
        {
            std::lock_guard<std::mutex> lock(scratch_mutex());
            CachedBlock *slot = scratch_entry(dev, s);
            if (slot && !slot->busy) {
                CachedBlock &b = *slot;
                if (b.p && b.bytes < bytes) {                          // too small: released behind the work that used it
                    (void)hipFreeAsync(b.p, s);
                    b.p = nullptr;
                    b.bytes = 0;
                }
                if (!b.p) {
                    const size_t want = bytes + bytes / 4;             // (room for a somewhat larger call)
                    void *q = nullptr;
                    hipError_t e = pool ? hipMallocFromPoolAsync(&q, want, pool, s) : hipMallocAsync(&q, want, s);
                    if (e == hipSuccess && q) {
                        b.p = q;
                        b.bytes = want;
                    } else {
                        (void)hipGetLastError();
                    }
                }
                if (b.p) {
                    b.busy = true;
                    claimed = &b;
                    p = b.p;
                    return true;
                }
            }
        }
        // the stream's block is held by another host thread right now (or could not be had): a block of this call's own
        hipError_t e = pool ? hipMallocFromPoolAsync(&p, bytes, pool, s) : hipMallocAsync(&p, bytes, s);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
        }
        return p != nullptr;
    }
    ~CallScratch()
    {
        if (claimed) {
            std::lock_guard<std::mutex> lock(scratch_mutex());
            claimed->busy = false;
        } else if (p) {
            (void)hipFreeAsync(p, stream);
        }
    }
};

}  // namespace memc
