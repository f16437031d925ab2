// memc_scratch.hpp -- the scratch block of a call that needs device words of its own and was NOT handed a workspace (host side):
// the (Depth)FlowProjection forward through the reference-signature entry points (flow_projection.hip) and the many-channel
// backward passes (fi_bwd_cn.hip: the site tiles' target boxes, FilterInterpolation and InterpolationCh alike).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>

namespace memc {

// The forward pass needs a few device words that outlive a kernel (far-source flags, the tiles' landing boxes and stamps)
// and, with hole filling, the filler's tables.  A caller that wants the library to own NOTHING passes a workspace
// (<Op>Layer_gpu_forward_ws, include/memc_warp.h: what the shipped Python layer does -- torch's caching allocator, capturable
// into a HIP graph).  The reference-signature entry points have no such parameter; for them the block comes from here.
//
// Pool.  A private memory pool per device (created on first use, kept for the life of the process, release threshold
// "never": with the default threshold a pool hands its memory back at every synchronisation and the next call pays for a
// fresh allocation, measured +200 us); the device's default pool and its attributes are left alone.
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

// Block cache.  A stream-ordered allocation and its release cost the HOST ~30 us per call (round 4), so blocks are kept.
// Round 4 keyed them by the stream HANDLE and trusted "calls on one stream run in order" -- which a handle does not
// promise: hipStreamPerThread is one handle and a different queue in every host thread, and the handle of a destroyed
// stream can be handed out again while the old queue's last work is in flight (round 4 review).  Now the ORDER is explicit:
//   * a block is owned by exactly one host-side call at a time (`busy`, under the mutex);
//   * at release the call records an event behind its last kernel; the next claimant -- whatever its stream, thread or
//     device queue -- makes its stream wait for that event before its first kernel.  On the same queue that wait is free;
//   * a small set of blocks per device (kBlocks), any stream may take any free one (preferring the one it used last: no
//     cross-queue wait); no per-stream entry, nothing to leak when streams come and go; all of them busy (more than
//     kBlocks host threads inside a projection forward at once) -> a stream-ordered allocation of the call's own, as rounds 2-3
//     did for every call; a block that is too small is released in stream order and replaced.
// Device memory held: at most kBlocks blocks of ~0.8 B per pixel of the largest call each, per device.
struct CachedBlock {
    void *p = nullptr;
    size_t bytes = 0;
    hipEvent_t done = nullptr;                 // recorded behind the last call that used the block
    hipStream_t last = nullptr;                // (preference only: never trusted for ordering)
    bool recorded = false, busy = false;
    uint64_t stamp = 0;                        // last use (LRU: the replacement victim)
};
constexpr int kBlocks = 8, kDevices = 16;
inline std::mutex &scratch_mutex()
{
    static std::mutex mu;
    return mu;
}
// (plain tables, no std::map: the library exports nothing but its C surface -- tests/test_abi.py)
inline CachedBlock *scratch_table(int dev)
{
    static CachedBlock table[kDevices][kBlocks];
    return dev >= 0 && dev < kDevices ? table[dev] : nullptr;
}

struct CallScratch {
    void *p = nullptr;
    hipStream_t stream = nullptr;
    CachedBlock *claimed = nullptr;            // the cache entry this call holds (released, not freed, at the end)
    // max_blocks: how many of the device's kBlocks this call may look at (the measurement build's test of the ordering:
    // with one block every call takes the block of the call before it, whatever their streams)
    bool alloc(size_t bytes, hipStream_t s, int max_blocks = kBlocks)
    {
        stream = s;
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        (void)hipStreamIsCapturing(s, &capture);
        if (capture != hipStreamCaptureStatusNone) return false;      // no allocation / event inside a stream capture
        int dev = -1;
        if (hipStreamGetDevice(s, &dev) != hipSuccess) {              // the STREAM's device, not the current one
            (void)hipGetLastError();
            if (hipGetDevice(&dev) != hipSuccess) dev = -1;
        }
        hipMemPool_t pool = pool_for_device(dev);
        CachedBlock *table = scratch_table(dev);
        // The cache's events are created and recorded with the CURRENT device's context: a stream of another device (the caller
        // did not switch) gets a stream-ordered block of its own instead -- an event of the wrong device would fail at every
        // record, silently, and the block would be freed and re-allocated per call anyway (round-5 review).
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) (void)hipGetLastError();
        if (cur != dev) table = nullptr;
        // hipEventQuery / hipEventCreate / the pool allocation are "potentially unsafe" calls while ANOTHER thread captures a
        // stream in global mode (torch's default): this thread declares itself relaxed for the length of the claim, so that it
        // cannot invalidate somebody else's capture (its own stream is not capturing: checked above)
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        const bool exchanged = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
        if (!exchanged) (void)hipGetLastError();
        struct Restore {
            hipStreamCaptureMode *m;
            bool on;
            ~Restore() { if (on && hipThreadExchangeStreamCaptureMode(m) != hipSuccess) (void)hipGetLastError(); }
        } restore{&mode, exchanged};
        if (table) {
            static uint64_t clock = 0;
            std::lock_guard<std::mutex> lock(scratch_mutex());
            // Which free block?  (4) large enough and last used on this stream handle: the steady state of a caller with one
            // stream -- found without asking the runtime anything; (3) large enough and its previous user's kernels are DONE
            // (hipEventQuery): nothing to wait for; (2) a slot without memory yet: a fresh allocation, so that streams that
            // run side by side each end up with a block of their own instead of queueing behind one; (1) large enough, its
            // previous user still running on another stream: this call's kernels wait for it (correct, not concurrent); (0)
            // too small: released behind that wait and replaced.  Ties: least recently used.
            CachedBlock *best = nullptr;
            int best_rank = -1;
            for (int i = 0; i < kBlocks && i < max_blocks && best_rank < 4; i++) {
                CachedBlock &b = table[i];
                if (b.busy) continue;
                int rank;
                if (!b.p) rank = 2;
                else if (b.bytes < bytes) rank = 0;
                else if (b.last == s) rank = 4;
                else if (!b.recorded || hipEventQuery(b.done) == hipSuccess) rank = 3;
                else {
                    (void)hipGetLastError();   // (hipErrorNotReady is not an error)
                    rank = 1;
                }
                if (rank > best_rank || (rank == best_rank && b.stamp < best->stamp)) {
                    best = &b;
                    best_rank = rank;
                }
            }
            if (best) {
                CachedBlock &b = *best;
                bool ordered = true;
                if (b.recorded && hipStreamWaitEvent(s, b.done, 0) != hipSuccess) {   // behind the block's previous user
                    (void)hipGetLastError();
                    ordered = false;
                }
                if (ordered && b.p && b.bytes < bytes) {               // too small: released behind that wait, replaced
                    (void)hipFreeAsync(b.p, s);
                    b.p = nullptr;
                    b.bytes = 0;
                }
                if (ordered && !b.p) {
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
                if (ordered && b.p && !b.done && hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) {
                    (void)hipGetLastError();
                    b.done = nullptr;
                }
                if (ordered && b.p && b.done) {
                    b.busy = true;
                    b.last = s;
                    b.stamp = ++clock;
                    claimed = &b;
                    p = b.p;
                    return true;
                }
            }
        }
        // every block is held by another host thread right now (or could not be had): a block of this call's own
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
            // behind the call's last kernel; a failed record leaves the block unusable for others until it succeeds once
            const bool ok = hipEventRecord(claimed->done, stream) == hipSuccess;
            if (!ok) (void)hipGetLastError();
            std::lock_guard<std::mutex> lock(scratch_mutex());
            if (ok) {
                claimed->recorded = true;
                claimed->busy = false;
            } else {                           // cannot order the next user behind this call: give the block up in stream order
                (void)hipFreeAsync(claimed->p, stream);
                claimed->p = nullptr;
                claimed->bytes = 0;
                claimed->recorded = false;
                claimed->busy = false;
            }
        } else if (p) {
            (void)hipFreeAsync(p, stream);
        }
    }
};

}  // namespace memc
