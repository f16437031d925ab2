/*
 * memc_warp.h -- C ABI of libmemc_hip.so: MEMC-Net's adaptive-warp / flow-projection operators as
 * hand-written HIP kernels for gfx950 (MI355X).
 *
 * This is the drop-in boundary.  The reference reaches its CUDA kernels through two C layers that its
 * cffi loader binds (my_package/_ext/my_lib/__init__.py:5-11):
 *
 *   layer entry points   my_package/src/my_lib_cuda.h:37-117   <Op>Layer_gpu_{forward,backward}(THCudaTensor*...)
 *   kernel launchers     my_package/src/my_lib_kernel.h:67-220 <Op>_gpu_{forward,backward}_kernel(cudaStream_t, ...)
 *
 * Both are exported here under the reference's own names.  The kernel launchers keep the reference's
 * parameter lists verbatim (stream, nElement, w, h, channel, batch, [filter_size | fillhole], four element
 * strides per tensor, device pointers); only `cudaStream_t` becomes a HIP stream handle.  The layer entry
 * points keep the reference's names, argument order, checks and return codes; `THCudaTensor*` (of which
 * the reference reads only size[], stride[] and the data pointer) becomes the POD descriptor
 * `memc_tensor4`, and the stream the reference took from its global THCState is passed explicitly.
 *
 * Contract (identical to the reference):
 *   - fp32, NCHW, strides in ELEMENTS, w-stride must be 1;
 *   - every output / gradient buffer is borrowed: allocated AND zero-filled by the caller
 *     (my_package/functions/FilterInterpolationLayer.py:26-29,46-48; FlowProjectionLayer.py:27-29,54).
 *     (Of these, the library itself only RELIES on zeros in the two scattered-into gradients -- FilterInterpolation
 *     gradinput1 and Interpolation gradinput1: the forward outputs of every operator (projection count / output
 *     included), FilterInterpolation gradinput2 / gradinput3, Interpolation gradinput2 and both projection
 *     backward gradients are fully defined by the kernels, which is what lets the shipped Python layer skip
 *     those memsets.)
 *     With zero-filled buffers the results are the reference's.  With anything else they are unspecified,
 *     and in two places differ from the reference's `+=`: the FilterInterpolation backward STORES gradinput3
 *     (each site owns its taps) and the (Depth)FlowProjection backward STORES gradinput1 / gradinput2 (each site
 *     owns its elements) instead of adding to them; FilterInterpolation gradinput2 is assigned and gradinput1
 *     added to, as in the reference -- except with four or more channels and the 4x4 filter, for which
 *     gradinput1 is STORED as well, on every path (owner-computes kernels: every cell has exactly one writer),
 *     so that a caller may skip that memset; the same holds for the gradinput1 of the Interpolation(Ch) backward
 *     with four or more channels;
 *   - `output`, `gradoutput` and `gradinput1` are indexed with input1's b/c/h strides
 *     (my_lib_kernel.cu:1184,1276-1283), `gradinput2`/`gradinput3` with input2's/input3's;
 *   - work is enqueued asynchronously on `stream`; no host synchronisation, no state carried from one call to
 *     the next (one process-wide call COUNTER excepted: it only makes every call's "far source" flag value unique),
 *     nothing shared between concurrent calls (any number of streams / host threads), nothing read from the
 *     environment.  One exception to "never allocates": a (Depth)FlowProjection FORWARD call needs a scratch block on the
 *     device (1.3 KiB of "far source" flags and 32 bytes per 64 x 32 tile for its fast path -- where the tile's sources that
 *     move 24 px or more AGAINST THE IMAGE'S DOMINANT MOTION land, and whether such a source of another tile lands in it:
 *     those tiles are recomputed by a second kernel, exact for any motion; with hole filling also the filler's per-tile
 *     summaries and masks, about 0.8 bytes per pixel).  Two ways to provide it:
 *       (a) the reference-signature entry points below take it from a PRIVATE memory pool of the stream's device (created
 *           on the first such call; the device's default pool and its attributes are not touched) and keep it for later
 *           calls -- a stream-ordered allocation per call cost the host ~30 us (round 4): at most eight blocks per device, for
 *           the life of the process, each owned by one host-side call at a time; the next call that takes a block makes
 *           its stream wait for an event recorded behind the previous user's last kernel, so any mix of streams, host
 *           threads, hipStreamPerThread and recycled stream handles is ordered explicitly.  Inside a stream capture, or
 *           if the allocation fails, these entry points use their general path and a hole filler that need no scratch
 *           (slower, same results);
 *       (b) the `_ws` entry points ("EXTENSION: workspace" below) take a caller-supplied workspace of
 *           memc_flow_projection_workspace_bytes() bytes: the library then allocates nothing and keeps nothing -- the
 *           reference's own contract (my_lib_cuda.c:752-799) -- and a stream capture (HIP graph) records the same
 *           kernels as an eager call.  The shipped Python layer uses these;
 *   - precision of the two scattered-into image gradients at three channels (FilterInterpolation / Interpolation
 *     gradinput1): every contribution g * weight is rounded ONCE to a multiple of 2^(e - 22), where 2^e bounds the
 *     contributions of the PACKED sites of its 64 x 16 (64 x 32) tile: 2^e < 2 * min(largest site bound, 16 x a robust
 *     mean of the tile's site bounds), a site's bound being (its largest |gradoutput|) x (its largest |tap|).  A cell's
 *     error is at most (its number of contributions) x 2^(e - 23) -- ~2e-6 x the tile's typical contribution at the 16
 *     contributions of an ordinary cell.  Sites beyond that bound, and sites with an Inf / NaN input, add with fp32
 *     atomics exactly as the reference does (my_lib_kernel.cu:1276-1288); four and more channels never round;
 *   - return 0 on success, -1 on a failed shape/stride check or a launch error (my_lib_cuda.c:611-646,
 *     my_lib_kernel.cu:1559-1566).
 *
 * No torch, HIP or C++ types appear in this header; a HIP stream is passed as an opaque pointer
 * (`hipStream_t` is itself a pointer type; 0 / NULL is the default stream).
 */
#ifndef MEMC_WARP_H
#define MEMC_WARP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: its C surface is exactly the functions declared in this header
 * (tests/test_abi.py).  The dynamic symbol table also holds the mangled handles of the HIP kernels, which the HIP runtime
 * resolves by name; they are not an interface. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef void *memc_stream_t;            /* hipStream_t */

/* Replaces THCudaTensor at the layer boundary (my_lib_cuda.c reads only these three things through
 * THCudaTensor_size / THCudaTensor_stride / THCudaTensor_data). */
typedef struct memc_tensor4 {
    float  *data;                       /* device pointer */
    int64_t size[4];                    /* N, C, H, W */
    int64_t stride[4];                  /* element strides */
} memc_tensor4;

/* Library / build identification: returns a static string such as "memc_hip 0.1 gfx950". */
const char *memc_hip_version(void);

/* Which kernel family did the most recent operator call made BY THE CALLING THREAD take?  A static string
 * "<operator>:<family>", e.g. "fi_fwd:tiled_c3", "proj_fwd:owner", "bl_bwd:direct"; "" before the first call.
 * Base pointers and strides may be anything (round 5: quads are accessed at dword alignment).  A WIDTH that is not a multiple of
 * four is served by the same families, from width 8 on: the (Depth)FlowProjection forward by ragged-row instantiations, the
 * FilterInterpolation forward and RGB backward, the Interpolation forward / backward and the (Depth)FlowProjection backward
 * by the whole quads of every row on the tiled kernel and the one to three columns behind them on the one-lane-per-site
 * kernel (1.1-2.0x the aligned time at 1278 x 720, profiles/r05_slow_paths.txt); the many-channel backward passes at such
 * widths, widths below 8 and FilterInterpolation with filter_size != 4 take scalar kernels -- same results, ~2x slower (the
 * many-channel backward more) -- whose family names are
 *     "direct"  (FilterInterpolation / Interpolation: one lane per site, global gathers and atomics),
 *     "generic" (FilterInterpolation, filter_size != 4),
 *     "scalar"  ((Depth)FlowProjection forward / backward),
 *     "general" ((Depth)FlowProjection forward without scratch: the reference-signature entry points inside a stream
 *               capture -- the `_ws` entry points keep the fast path there --, or a plane beyond 4 GiB).
 * The reference has no counterpart (it has one kernel per operator). */
const char *memc_last_kernel_path(void);

/* Does the backward of this operator STORE gradinput1 (1) or ACCUMULATE into it like the reference's atomicAdd (0)?
 * filter_size: FilterInterpolation's fs (e.g. 4); 0 for Interpolation / InterpolationCh.
 * The reference accumulates for every channel count (my_lib_kernel.cu:1276, :690) and its callers hand over a
 * zero-filled buffer; both contracts give the same result on a zero-filled buffer.  This library accumulates for
 * one to three channels (RGB) and stores for four and more (fs == 4 or the bilinear warp): a caller that wants to
 * skip the zero fill, or that accumulates several calls into one buffer, asks here instead of copying the rule. */
int memc_gradinput1_is_stored(int filter_size, int channel);

/* ======================================================================================================
 * Layer entry points -- replace my_lib_cuda.h:37-117 (implemented in the reference by my_lib_cuda.c).
 * ==================================================================================================== */

/* my_lib_cuda.h:37-42 / my_lib_cuda.c:364-417.  Bilinear warp, channel must be 3. */
int InterpolationLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input1,
                                   const memc_tensor4 *input2, const memc_tensor4 *output);
/* my_lib_cuda.h:44-52 / my_lib_cuda.c:419-479 */
int InterpolationLayer_gpu_backward(memc_stream_t stream, const memc_tensor4 *input1,
                                    const memc_tensor4 *input2, const memc_tensor4 *gradoutput,
                                    const memc_tensor4 *gradinput1, const memc_tensor4 *gradinput2);

/* my_lib_cuda.h:53-58 / my_lib_cuda.c:481-534.  Bilinear warp, any channel count. */
int InterpolationChLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input1,
                                     const memc_tensor4 *input2, const memc_tensor4 *output);
/* my_lib_cuda.h:60-68 / my_lib_cuda.c:536-596 */
int InterpolationChLayer_gpu_backward(memc_stream_t stream, const memc_tensor4 *input1,
                                      const memc_tensor4 *input2, const memc_tensor4 *gradoutput,
                                      const memc_tensor4 *gradinput1, const memc_tensor4 *gradinput2);

/* my_lib_cuda.h:69-75 / my_lib_cuda.c:598-668.  Flow sample fused with the fs x fs adaptive filter;
 * input3 has fs*fs channels. */
int FilterInterpolationLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input1,
                                         const memc_tensor4 *input2, const memc_tensor4 *input3,
                                         const memc_tensor4 *output);
/* my_lib_cuda.h:77-85 / my_lib_cuda.c:669-749.
 * EXTENSION: gradinput1 may be NULL -- "the image gradient is not wanted" (the reference's networks never use it: the frames
 * they warp are data, networks/MEMC_Net_star.py:266-277): three channels, filter_size 4 and a width that is a multiple of four
 * then skip its accumulation and its zero fill; any other shape returns -1 with nothing written and the caller
 * passes a buffer.  The launcher below takes a NULL gradinput1 pointer the same way. */
int FilterInterpolationLayer_gpu_backward(memc_stream_t stream, const memc_tensor4 *input1,
                                          const memc_tensor4 *input2, const memc_tensor4 *input3,
                                          const memc_tensor4 *gradoutput,
                                          const memc_tensor4 *gradinput1,
                                          const memc_tensor4 *gradinput2,
                                          const memc_tensor4 *gradinput3);

/* my_lib_cuda.h:87-92 / my_lib_cuda.c:752-799.  Forward splat of -flow, count-normalise, optional
 * hole fill (fillhole != 0). */
int FlowProjectionLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input1,
                                    const memc_tensor4 *count, const memc_tensor4 *output,
                                    int fillhole);
/* my_lib_cuda.h:94-99 / my_lib_cuda.c:801-855 */
int FlowProjectionLayer_gpu_backward(memc_stream_t stream, const memc_tensor4 *input1,
                                     const memc_tensor4 *count, const memc_tensor4 *gradoutput,
                                     const memc_tensor4 *gradinput1);

/* my_lib_cuda.h:101-107 / my_lib_cuda.c:857-914.  Depth-weighted projection; input2 = depth [N,1,H,W]. */
int DepthFlowProjectionLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input1,
                                         const memc_tensor4 *input2, const memc_tensor4 *count,
                                         const memc_tensor4 *output, int fillhole);
/* my_lib_cuda.h:109-117 / my_lib_cuda.c:916-983 */
int DepthFlowProjectionLayer_gpu_backward(memc_stream_t stream, const memc_tensor4 *input1,
                                          const memc_tensor4 *input2, const memc_tensor4 *count,
                                          const memc_tensor4 *output, const memc_tensor4 *gradoutput,
                                          const memc_tensor4 *gradinput1,
                                          const memc_tensor4 *gradinput2);

/* ======================================================================================================
 * Kernel launchers -- replace my_lib_kernel.h:67-220 (implemented in the reference by
 * my_lib_kernel.cu).  Parameter lists are the reference's, with cudaStream_t -> memc_stream_t.
 * `nElement` is accepted and ignored, as in the reference.
 * ==================================================================================================== */

/* my_lib_kernel.h:67-81 */
int InterpolationLayer_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int input2_b_stride, const int input2_c_stride, const int input2_h_stride, const int input2_w_stride,
    const float *input1, const float *input2, float *output);

/* my_lib_kernel.h:83-98 */
int InterpolationLayer_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int input2_b_stride, const int input2_c_stride, const int input2_h_stride, const int input2_w_stride,
    const float *input1, const float *input2, const float *gradoutput, float *gradinput1, float *gradinput2);

/* my_lib_kernel.h:101-115 */
int InterpolationChLayer_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int input2_b_stride, const int input2_c_stride, const int input2_h_stride, const int input2_w_stride,
    const float *input1, const float *input2, float *output);

/* my_lib_kernel.h:117-132 */
int InterpolationChLayer_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int input2_b_stride, const int input2_c_stride, const int input2_h_stride, const int input2_w_stride,
    const float *input1, const float *input2, const float *gradoutput, float *gradinput1, float *gradinput2);

/* my_lib_kernel.h:133-144 */
int FilterInterpolationLayer_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int filter_size,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int input2_b_stride, const int input2_c_stride, const int input2_h_stride, const int input2_w_stride,
    const int input3_b_stride, const int input3_c_stride, const int input3_h_stride, const int input3_w_stride,
    const float *input1, const float *input2, const float *input3, float *output);

/* my_lib_kernel.h:146-158 */
int FilterInterpolationLayer_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int filter_size,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int input2_b_stride, const int input2_c_stride, const int input2_h_stride, const int input2_w_stride,
    const int input3_b_stride, const int input3_c_stride, const int input3_h_stride, const int input3_w_stride,
    const float *input1, const float *input2, const float *input3,
    const float *gradoutput, float *gradinput1, float *gradinput2, float *gradinput3);

/* my_lib_kernel.h:161-171 */
int FlowProjection_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int fillhole,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int count_b_stride, const int count_c_stride, const int count_h_stride, const int count_w_stride,
    const float *input1, float *count, float *output);

/* my_lib_kernel.h:173-187 */
int FlowProjection_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int count_b_stride, const int count_c_stride, const int count_h_stride, const int count_w_stride,
    const float *input1, const float *count, const float *gradoutput, float *gradinput1);

/* my_lib_kernel.h:189-200 */
int DepthFlowProjection_gpu_forward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int fillhole,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int input2_b_stride, const int input2_c_stride, const int input2_h_stride, const int input2_w_stride,
    const int count_b_stride, const int count_c_stride, const int count_h_stride, const int count_w_stride,
    const float *input1, const float *input2, float *count, float *output);

/* my_lib_kernel.h:202-220 */
int DepthFlowProjection_gpu_backward_kernel(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int input2_b_stride, const int input2_c_stride, const int input2_h_stride, const int input2_w_stride,
    const int count_b_stride, const int count_c_stride, const int count_h_stride, const int count_w_stride,
    const float *input1, const float *input2, const float *count, const float *output,
    const float *gradoutput, float *gradinput1, float *gradinput2);

/* ------------------------------------------------------------------------------------------------------
 * EXTENSION -- no reference counterpart.  The two adaptive warps of a frame pair and
 * their occlusion-weighted blend, networks/MEMC_Net_star.py:266-277 (`FilterInterpolate`), in one pass:
 *
 *     output = occlusion0 * FilterInterpolation(input0, flow0, filter0)
 *            + occlusion1 * FilterInterpolation(input2, flow1, filter1)
 *
 * so that the two warped frames never exist in memory (188 instead of 236 bytes per site).  Forward only (the
 * shipped Python layer differentiates it through the reference-API entry points above).  RGB inputs,
 * filter_size 4, a width that is a multiple of four: anything else returns -1 and the caller composes the result from two
 * FilterInterpolationLayer_gpu_forward calls.  input0 / input2 / output share one layout, flow0 / flow1 another,
 * filter0 / filter1 another, the occlusions ([B, 1, H, W]) another.  `output` need not be zero-filled.
 * ------------------------------------------------------------------------------------------------------ */
int FilterInterpolationBlendLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input0,
                                              const memc_tensor4 *input2, const memc_tensor4 *flow0,
                                              const memc_tensor4 *flow1, const memc_tensor4 *filter0,
                                              const memc_tensor4 *filter1, const memc_tensor4 *occlusion0,
                                              const memc_tensor4 *occlusion1, const memc_tensor4 *output);

int FilterInterpolationBlend_gpu_forward_kernel(
    memc_stream_t stream, const int w, const int h, const int channel, const int batch, const int filter_size,
    const int input_b_stride, const int input_c_stride, const int input_h_stride,
    const int flow_b_stride, const int flow_c_stride, const int flow_h_stride,
    const int filter_b_stride, const int filter_c_stride, const int filter_h_stride,
    const int occlusion_b_stride, const int occlusion_h_stride,
    const float *input0, const float *input2, const float *flow0, const float *flow1,
    const float *filter0, const float *filter1, const float *occlusion0, const float *occlusion1, float *output);

/* ------------------------------------------------------------------------------------------------------
 * EXTENSION -- no reference counterpart.  networks/MEMC_Net_star.py:273-285 warps a frame
 * and its 64-channel context features with the SAME flow and the same 16 filter planes (FilterInterpolate, then
 * FilterInterpolate_ctx).  One pass per direction: flow + taps are streamed once for both,
 *
 *     context_out = FilterInterpolation(context, flow, filter)
 *     image_out   = FilterInterpolation(image, flow, filter)                                   (prev == NULL)
 *     image_out   = occlusion_prev * prev + occlusion_this * FilterInterpolation(image, ...)   (otherwise)
 *
 * the second form being MEMC_Net_star.py:277's blend with `prev` = the other direction's image_out.  image, prev,
 * image_out: [B, 3, H, W], one layout; context, context_out: [B, C, H, W], C a multiple of 4, one layout; the two
 * occlusions [B, 1, H, W], one layout; prev / occlusion_prev / occlusion_this all NULL or all given.  Forward only,
 * filter_size 4, a width that is a multiple of four: anything else returns -1 and the caller uses the entry points above.
 * Outputs need not be zero-filled.
 * ------------------------------------------------------------------------------------------------------ */
int FilterInterpolationCtxLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *image,
                                            const memc_tensor4 *context, const memc_tensor4 *flow,
                                            const memc_tensor4 *filter, const memc_tensor4 *prev,
                                            const memc_tensor4 *occlusion_prev, const memc_tensor4 *occlusion_this,
                                            const memc_tensor4 *image_out, const memc_tensor4 *context_out);

int FilterInterpolationCtx_gpu_forward_kernel(
    memc_stream_t stream, const int w, const int h, const int channel, const int batch, const int filter_size,
    const int image_b_stride, const int image_c_stride, const int image_h_stride,
    const int context_b_stride, const int context_c_stride, const int context_h_stride,
    const int flow_b_stride, const int flow_c_stride, const int flow_h_stride,
    const int filter_b_stride, const int filter_c_stride, const int filter_h_stride,
    const int occlusion_b_stride, const int occlusion_h_stride,
    const float *image, const float *context, const float *flow, const float *filter,
    const float *prev, const float *occlusion_prev, const float *occlusion_this, float *image_out, float *context_out);

/* ------------------------------------------------------------------------------------------------------
 * EXTENSION: workspace -- the (Depth)FlowProjection forward with a CALLER-SUPPLIED workspace.  The reference's launcher
 * touches borrowed buffers only (my_lib_kernel.cu:1905-1992; contract my_lib_cuda.c:752-799: the callee never allocates),
 * so it can be enqueued inside a stream capture at full speed; with these entry points so can this library's fast path
 * (owner-computes kernel, hole filling from masks): nothing is allocated, cached or shared between calls.
 *
 *   memc_flow_projection_workspace_bytes(w, h, batch, fillhole, depth)   bytes a call of that shape needs (a multiple of
 *                                                                        256; 0 for an empty shape)
 *   workspace        device pointer, 16-byte aligned, at least that many bytes, contents arbitrary (never read before
 *                    written, except flag words compared against this call's unique tag); private to the call until its
 *                    last kernel has run -- allocate it in stream order (e.g. torch.empty on the current stream)
 *   return           0 / -1 as the entry point it extends; -1 also for a NULL, misaligned or too small workspace
 * Same arguments, checks and results otherwise as FlowProjectionLayer_gpu_forward / FlowProjection_gpu_forward_kernel /
 * DepthFlowProjection... above.
 * ------------------------------------------------------------------------------------------------------ */
size_t memc_flow_projection_workspace_bytes(int w, int h, int batch, int fillhole, int depth);

int FlowProjectionLayer_gpu_forward_ws(memc_stream_t stream, const memc_tensor4 *input1, const memc_tensor4 *count,
                                       const memc_tensor4 *output, int fillhole, void *workspace, size_t workspace_bytes);
int DepthFlowProjectionLayer_gpu_forward_ws(memc_stream_t stream, const memc_tensor4 *input1,
                                            const memc_tensor4 *input2, const memc_tensor4 *count,
                                            const memc_tensor4 *output, int fillhole, void *workspace,
                                            size_t workspace_bytes);
int FlowProjection_gpu_forward_kernel_ws(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int fillhole,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int count_b_stride, const int count_c_stride, const int count_h_stride, const int count_w_stride,
    const float *input1, float *count, float *output, void *workspace, size_t workspace_bytes);
int DepthFlowProjection_gpu_forward_kernel_ws(
    memc_stream_t stream, const int nElement, const int w, const int h, const int channel, const int batch,
    const int fillhole,
    const int input1_b_stride, const int input1_c_stride, const int input1_h_stride, const int input1_w_stride,
    const int input2_b_stride, const int input2_c_stride, const int input2_h_stride, const int input2_w_stride,
    const int count_b_stride, const int count_c_stride, const int count_h_stride, const int count_w_stride,
    const float *input1, const float *input2, float *count, float *output, void *workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------------------------------------------
 * EXTENSION -- no reference counterpart.  The prologue of FlowProjection in the networks,
 * networks/MEMC_Net_star.py:172-176: output = bilinear_x4((mul * input) / div), i.e.
 * F.interpolate(div_flow * flow / 2.0, scale_factor=4, mode="bilinear", align_corners=...) as one kernel (torch's
 * sampling rule, ATen area_pixel_compute_source_index).  input [B, C, h, w], output [B, C, 4h, 4w] (16-byte aligned
 * rows); forward only (the Python layer differentiates it with ATen's upsample_bilinear2d_backward).
 * ------------------------------------------------------------------------------------------------------ */
int FlowUpsample4Layer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input, const memc_tensor4 *output,
                                   float mul, float div, int align_corners);

int FlowUpsample4_gpu_forward_kernel(
    memc_stream_t stream, const int w, const int h, const int channel, const int batch,
    const int input_b_stride, const int input_c_stride, const int input_h_stride,
    const int output_b_stride, const int output_c_stride, const int output_h_stride,
    const float mul, const float div, const int align_corners, const float *input, float *output);

/* ------------------------------------------------------------------------------------------------------
 * MEASUREMENT AID -- no reference counterpart.  What this device's HBM gives the access pattern of the library's tiled
 * kernels (16 bytes per lane, non-temporal, workgroups dealt to the XCDs in contiguous runs):
 *
 *     dst[i] = src[i] + src[n + i] + ... + src[(reads_per_write - 1) n + i]     i in [0, n_float4), 16-byte elements
 *
 * reads_per_write 1 .. 8 (1: a copy; 7: the read : write mix of the RGB adaptive warp).  src holds reads_per_write *
 * n_float4 elements, dst n_float4; both 16-byte aligned device pointers.  bench.py reports the rate as
 * `roofline.achievable_peak`.  Returns 0 / -1 like the operators.
 * ------------------------------------------------------------------------------------------------------ */
int memc_calibration_stream(memc_stream_t stream, const float *src, float *dst, int64_t n_float4, int reads_per_write);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MEMC_WARP_H */
