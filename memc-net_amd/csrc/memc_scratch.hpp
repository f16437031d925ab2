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

// stream-ordered scratch of one call; freed (in stream order) when it goes out of scope, on every exit path
struct CallScratch {
    void *p = nullptr;
    hipStream_t stream = nullptr;
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
        hipError_t e = pool ? hipMallocFromPoolAsync(&p, bytes, pool, s) : hipMallocAsync(&p, bytes, s);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            p = nullptr;
        }
        return p != nullptr;
    }
    ~CallScratch()
    {
        if (p) (void)hipFreeAsync(p, stream);
    }
};

}  // namespace memc
