// layer_api.cpp -- the layer entry points of libmemc_hip.so (host code only).
//
// Replaces my_package/src/my_lib_cuda.c:364-983 of the reference: read sizes/strides from the tensor
// descriptors, validate, call the kernel launcher on the caller's stream, return 0 / -1.
//
// Checks: every check the reference performs (cited per function) is performed here.  In addition the
// descriptors that the kernels index with ANOTHER tensor's strides (output / gradoutput / gradinput1 with
// input1's, gradinput2 with input2's, gradinput3 with input3's, exactly as my_lib_kernel.cu does) must
// really have those b/c/h strides, all w-strides must be 1 and all strides must fit the launcher ABI's
// `int`; the reference leaves those cases unchecked and silently reads/writes the wrong cells.
#include "memc_internal.h"

#include <math.h>
#include <stdint.h>

namespace {

constexpr int kErr = -1;

inline bool fits_int(const memc_tensor4 *t)
{
    for (int i = 0; i < 4; i++)
        if (t->size[i] < 0 || t->size[i] > INT32_MAX || t->stride[i] < 0 || t->stride[i] > INT32_MAX)
            return false;
    return true;
}

inline bool same_shape(const memc_tensor4 *a, const memc_tensor4 *b)
{
    return a->size[0] == b->size[0] && a->size[1] == b->size[1] && a->size[2] == b->size[2] &&
           a->size[3] == b->size[3];
}

// same b/c/h strides (the kernels index `b` with `a`'s strides) and unit w stride
inline bool same_layout(const memc_tensor4 *a, const memc_tensor4 *b)
{
    if (!same_shape(a, b)) return false;
    for (int i = 0; i < 3; i++)                 // the stride of a size-1 dimension is never used
        if (a->size[i] > 1 && a->stride[i] != b->stride[i]) return false;
    return true;
}

inline int64_t numel(const memc_tensor4 *t) { return t->size[0] * t->size[1] * t->size[2] * t->size[3]; }

// usable descriptor: sizes/strides fit the launcher ABI, unit w stride, non-null data unless empty
inline bool ok(const memc_tensor4 *t)
{
    return t && fits_int(t) && (t->stride[3] == 1 || t->size[3] <= 1) && (t->data || numel(t) == 0);
}

inline int nelem(const memc_tensor4 *t) { return (int)numel(t); }   // ignored by the launchers

#define S4(t) (int)(t)->stride[0], (int)(t)->stride[1], (int)(t)->stride[2], (int)(t)->stride[3]

// flow tensor [N,2,H,W] matching input1 [N,C,H,W]; my_lib_cuda.c:375-381
inline bool flow_matches(const memc_tensor4 *in1, const memc_tensor4 *flow)
{
    return flow->size[0] == in1->size[0] && flow->size[1] == 2 && flow->size[2] == in1->size[2] &&
           flow->size[3] == in1->size[3];
}

int bilinear_forward(bool require_c3, memc_stream_t stream, const memc_tensor4 *input1,
                     const memc_tensor4 *input2, const memc_tensor4 *output)
{
    if (!ok(input1) || !ok(input2) || !ok(output)) return kErr;
    if (require_c3 && input1->size[1] != 3) return kErr;                       // my_lib_cuda.c:373
    if (!flow_matches(input1, input2)) return kErr;                            // :375-381
    if (!same_layout(input1, output)) return kErr;                             // :397-398 (+h, w)
    return InterpolationLayer_gpu_forward_kernel(
        stream, nelem(output), (int)input1->size[3], (int)input1->size[2], (int)input1->size[1],
        (int)input1->size[0], S4(input1), S4(input2), input1->data, input2->data, output->data);
}

int bilinear_backward(bool require_c3, memc_stream_t stream, const memc_tensor4 *input1,
                      const memc_tensor4 *input2, const memc_tensor4 *gradoutput,
                      const memc_tensor4 *gradinput1, const memc_tensor4 *gradinput2)
{
    if (!ok(input1) || !ok(input2) || !ok(gradoutput) || !ok(gradinput1) || !ok(gradinput2)) return kErr;
    if (require_c3 && input1->size[1] != 3) return kErr;                       // my_lib_cuda.c:430
    if (!flow_matches(input1, input2)) return kErr;                            // :432-438
    if (!same_layout(input1, gradinput1) || !same_layout(input2, gradinput2)) return kErr;   // :455-458
    if (!same_layout(input1, gradoutput)) return kErr;
    return InterpolationLayer_gpu_backward_kernel(
        stream, nelem(gradoutput), (int)input1->size[3], (int)input1->size[2], (int)input1->size[1],
        (int)input1->size[0], S4(input1), S4(input2), input1->data, input2->data, gradoutput->data,
        gradinput1->data, gradinput2->data);
}

}  // namespace

namespace memc {
thread_local const char *t_last_path = "";
}
extern "C" const char *memc_last_kernel_path(void) { return memc::t_last_path; }
#ifdef MEMC_MEASURE
extern "C" const char *memc_debug_last_path(void) { return memc::t_last_path; }
#endif

extern "C" {

#ifdef MEMC_MEASURE
const char *memc_hip_version(void) { return "memc_hip 0.6 gfx950 MEASUREMENT BUILD (ablation arms present)"; }
#else
const char *memc_hip_version(void) { return "memc_hip 0.6 gfx950"; }
#endif

int memc_gradinput1_is_stored(int filter_size, int channel)
{
    return memc::fi_bwd_cn_class(channel, filter_size == 0 ? 4 : filter_size) ? 1 : 0;
}

int InterpolationLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input1,
                                   const memc_tensor4 *input2, const memc_tensor4 *output)
{ return bilinear_forward(true, stream, input1, input2, output); }

int InterpolationLayer_gpu_backward(memc_stream_t stream, const memc_tensor4 *input1,
                                    const memc_tensor4 *input2, const memc_tensor4 *gradoutput,
                                    const memc_tensor4 *gradinput1, const memc_tensor4 *gradinput2)
{ return bilinear_backward(true, stream, input1, input2, gradoutput, gradinput1, gradinput2); }

// The reference's Ch entry points call the InterpolationLayer_*_kernel launchers (my_lib_cuda.c:519,579).
int InterpolationChLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input1,
                                     const memc_tensor4 *input2, const memc_tensor4 *output)
{ return bilinear_forward(false, stream, input1, input2, output); }

int InterpolationChLayer_gpu_backward(memc_stream_t stream, const memc_tensor4 *input1,
                                      const memc_tensor4 *input2, const memc_tensor4 *gradoutput,
                                      const memc_tensor4 *gradinput1, const memc_tensor4 *gradinput2)
{ return bilinear_backward(false, stream, input1, input2, gradoutput, gradinput1, gradinput2); }

int FilterInterpolationLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input1,
                                         const memc_tensor4 *input2, const memc_tensor4 *input3,
                                         const memc_tensor4 *output)
{
    if (!ok(input1) || !ok(input2) || !ok(input3) || !ok(output)) return kErr;    // my_lib_cuda.c:641-643
    if (!flow_matches(input1, input2)) return kErr;                                 // :611-617
    if (input3->size[0] != input1->size[0] || input3->size[2] != input1->size[2] ||
        input3->size[3] != input1->size[3])
        return kErr;
    const int filter_size = (int)sqrt((float)input3->size[1]);                      // :619-620
    if (filter_size < 1) return kErr;
    if (!same_layout(input1, output)) return kErr;                                  // :644-645 (+h)
    return FilterInterpolationLayer_gpu_forward_kernel(
        stream, nelem(output), (int)input1->size[3], (int)input1->size[2], (int)input1->size[1],
        (int)input1->size[0], filter_size, S4(input1), S4(input2), S4(input3), input1->data, input2->data,
        input3->data, output->data);
}

int FilterInterpolationLayer_gpu_backward(memc_stream_t stream, const memc_tensor4 *input1,
                                          const memc_tensor4 *input2, const memc_tensor4 *input3,
                                          const memc_tensor4 *gradoutput, const memc_tensor4 *gradinput1,
                                          const memc_tensor4 *gradinput2, const memc_tensor4 *gradinput3)
{
    // EXTENSION: gradinput1 == NULL -- the caller does not want the image gradient (include/memc_warp.h)
    if (!ok(input1) || !ok(input2) || !ok(input3) || !ok(gradoutput) || (gradinput1 && !ok(gradinput1)) ||
        !ok(gradinput2) || !ok(gradinput3))
        return kErr;                                                                // :716-718
    if (!flow_matches(input1, input2)) return kErr;                                 // :685-691
    if (input3->size[0] != input1->size[0] || input3->size[2] != input1->size[2] ||
        input3->size[3] != input1->size[3])
        return kErr;
    const int filter_size = (int)sqrt((float)input3->size[1]);                      // :693-694
    if (filter_size < 1) return kErr;
    if ((gradinput1 && !same_layout(input1, gradinput1)) || !same_layout(input2, gradinput2) ||
        !same_layout(input3, gradinput3))
        return kErr;                                                                // :719-723
    if (!same_layout(input1, gradoutput)) return kErr;
    return FilterInterpolationLayer_gpu_backward_kernel(
        stream, nelem(gradoutput), (int)input1->size[3], (int)input1->size[2], (int)input1->size[1],
        (int)input1->size[0], filter_size, S4(input1), S4(input2), S4(input3), input1->data, input2->data,
        input3->data, gradoutput->data, gradinput1 ? gradinput1->data : nullptr, gradinput2->data, gradinput3->data);
}

// EXTENSION (memc_warp.h): fused dual warp + occlusion blend.  Same checks as two FilterInterpolation forwards,
// plus pairwise-equal layouts (one set of strides per tensor kind reaches the kernel).
int FilterInterpolationBlendLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input0,
                                              const memc_tensor4 *input2, const memc_tensor4 *flow0,
                                              const memc_tensor4 *flow1, const memc_tensor4 *filter0,
                                              const memc_tensor4 *filter1, const memc_tensor4 *occlusion0,
                                              const memc_tensor4 *occlusion1, const memc_tensor4 *output)
{
    const memc_tensor4 *all[] = {input0, input2, flow0, flow1, filter0, filter1, occlusion0, occlusion1, output};
    for (const memc_tensor4 *t : all)
        if (!ok(t)) return kErr;
    if (!flow_matches(input0, flow0)) return kErr;
    if (!same_layout(input0, input2) || !same_layout(input0, output) || !same_layout(flow0, flow1) ||
        !same_layout(filter0, filter1) || !same_layout(occlusion0, occlusion1))
        return kErr;
    if (filter0->size[0] != input0->size[0] || filter0->size[2] != input0->size[2] ||
        filter0->size[3] != input0->size[3])
        return kErr;
    if (occlusion0->size[0] != input0->size[0] || occlusion0->size[1] != 1 ||
        occlusion0->size[2] != input0->size[2] || occlusion0->size[3] != input0->size[3])
        return kErr;
    const int filter_size = (int)sqrt((float)filter0->size[1]);
    return FilterInterpolationBlend_gpu_forward_kernel(
        stream, (int)input0->size[3], (int)input0->size[2], (int)input0->size[1], (int)input0->size[0], filter_size,
        (int)input0->stride[0], (int)input0->stride[1], (int)input0->stride[2], (int)flow0->stride[0],
        (int)flow0->stride[1], (int)flow0->stride[2], (int)filter0->stride[0], (int)filter0->stride[1],
        (int)filter0->stride[2], (int)occlusion0->stride[0], (int)occlusion0->stride[2], input0->data, input2->data,
        flow0->data, flow1->data, filter0->data, filter1->data, occlusion0->data, occlusion1->data, output->data);
}

// EXTENSION (memc_warp.h): image + context warp of one direction in one pass, optional blend epilogue.
int FilterInterpolationCtxLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *image,
                                            const memc_tensor4 *context, const memc_tensor4 *flow,
                                            const memc_tensor4 *filter, const memc_tensor4 *prev,
                                            const memc_tensor4 *occlusion_prev, const memc_tensor4 *occlusion_this,
                                            const memc_tensor4 *image_out, const memc_tensor4 *context_out)
{
    const memc_tensor4 *need[] = {image, context, flow, filter, image_out, context_out};
    for (const memc_tensor4 *t : need)
        if (!ok(t)) return kErr;
    const bool blend = prev != nullptr;
    if (blend != (occlusion_prev != nullptr) || blend != (occlusion_this != nullptr)) return kErr;
    if (blend && (!ok(prev) || !ok(occlusion_prev) || !ok(occlusion_this))) return kErr;
    if (image->size[1] != 3 || !flow_matches(image, flow) || !flow_matches(context, flow)) return kErr;
    if (!same_layout(image, image_out) || !same_layout(context, context_out)) return kErr;
    if (filter->size[0] != image->size[0] || filter->size[2] != image->size[2] || filter->size[3] != image->size[3])
        return kErr;
    if (blend) {
        if (!same_layout(image, prev) || !same_layout(occlusion_prev, occlusion_this)) return kErr;
        if (occlusion_prev->size[0] != image->size[0] || occlusion_prev->size[1] != 1 ||
            occlusion_prev->size[2] != image->size[2] || occlusion_prev->size[3] != image->size[3])
            return kErr;
    }
    const int filter_size = (int)sqrt((float)filter->size[1]);
    return FilterInterpolationCtx_gpu_forward_kernel(
        stream, (int)image->size[3], (int)image->size[2], (int)context->size[1], (int)image->size[0], filter_size,
        (int)image->stride[0], (int)image->stride[1], (int)image->stride[2], (int)context->stride[0],
        (int)context->stride[1], (int)context->stride[2], (int)flow->stride[0], (int)flow->stride[1],
        (int)flow->stride[2], (int)filter->stride[0], (int)filter->stride[1], (int)filter->stride[2],
        blend ? (int)occlusion_prev->stride[0] : 0, blend ? (int)occlusion_prev->stride[2] : 0, image->data,
        context->data, flow->data, filter->data, blend ? prev->data : nullptr, blend ? occlusion_prev->data : nullptr,
        blend ? occlusion_this->data : nullptr, image_out->data, context_out->data);
}

// EXTENSION (memc_warp.h): scale + x4 bilinear upsampling of the quarter-resolution flow in one kernel.
int FlowUpsample4Layer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input, const memc_tensor4 *output,
                                   float mul, float div, int align_corners)
{
    if (!ok(input) || !ok(output)) return kErr;
    if (output->size[0] != input->size[0] || output->size[1] != input->size[1] ||
        output->size[2] != 4 * input->size[2] || output->size[3] != 4 * input->size[3])
        return kErr;
    return FlowUpsample4_gpu_forward_kernel(
        stream, (int)input->size[3], (int)input->size[2], (int)input->size[1], (int)input->size[0],
        (int)input->stride[0], (int)input->stride[1], (int)input->stride[2], (int)output->stride[0],
        (int)output->stride[1], (int)output->stride[2], mul, div, align_corners, input->data, output->data);
}

// count tensor [N,1,H,W] matching the flow tensor; my_lib_cuda.c:813-817
static bool count_matches(const memc_tensor4 *flow, const memc_tensor4 *count)
{
    return count->size[0] == flow->size[0] && count->size[1] == 1 && count->size[2] == flow->size[2] &&
           count->size[3] == flow->size[3];
}

int FlowProjectionLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input1,
                                    const memc_tensor4 *count, const memc_tensor4 *output, int fillhole)
{
    if (!ok(input1) || !ok(count) || !ok(output)) return kErr;
    if (input1->size[1] != 2) return kErr;                                          // my_lib_cuda.c:762
    if (!count_matches(input1, count)) return kErr;
    if (!same_layout(input1, output)) return kErr;                                  // :781-782 (+h)
    return FlowProjection_gpu_forward_kernel(
        stream, nelem(output), (int)input1->size[3], (int)input1->size[2], (int)input1->size[1],
        (int)input1->size[0], fillhole, S4(input1), S4(count), input1->data, count->data, output->data);
}

int FlowProjectionLayer_gpu_forward_ws(memc_stream_t stream, const memc_tensor4 *input1, const memc_tensor4 *count,
                                       const memc_tensor4 *output, int fillhole, void *workspace, size_t workspace_bytes)
{
    if (!ok(input1) || !ok(count) || !ok(output) || !workspace) return kErr;
    if (input1->size[1] != 2) return kErr;
    if (!count_matches(input1, count)) return kErr;
    if (!same_layout(input1, output)) return kErr;
    return FlowProjection_gpu_forward_kernel_ws(
        stream, nelem(output), (int)input1->size[3], (int)input1->size[2], (int)input1->size[1],
        (int)input1->size[0], fillhole, S4(input1), S4(count), input1->data, count->data, output->data, workspace,
        workspace_bytes);
}

int FlowProjectionLayer_gpu_backward(memc_stream_t stream, const memc_tensor4 *input1,
                                     const memc_tensor4 *count, const memc_tensor4 *gradoutput,
                                     const memc_tensor4 *gradinput1)
{
    if (!ok(input1) || !ok(count) || !ok(gradoutput) || !ok(gradinput1)) return kErr;
    if (input1->size[1] != 2) return kErr;                                          // :811
    if (!count_matches(input1, count)) return kErr;                                 // :813-817
    if (!same_layout(input1, gradinput1)) return kErr;                              // :835-836
    if (!same_layout(input1, gradoutput)) return kErr;
    return FlowProjection_gpu_backward_kernel(
        stream, nelem(gradoutput), (int)input1->size[3], (int)input1->size[2], (int)input1->size[1],
        (int)input1->size[0], S4(input1), S4(count), input1->data, count->data, gradoutput->data,
        gradinput1->data);
}

int DepthFlowProjectionLayer_gpu_forward(memc_stream_t stream, const memc_tensor4 *input1,
                                         const memc_tensor4 *input2, const memc_tensor4 *count,
                                         const memc_tensor4 *output, int fillhole)
{
    if (!ok(input1) || !ok(input2) || !ok(count) || !ok(output)) return kErr;
    if (input1->size[1] != 2) return kErr;                                          // :868
    if (input2->size[1] != 1) return kErr;                                          // :870
    if (!count_matches(input1, input2) || !count_matches(input1, count)) return kErr;
    if (!same_layout(input1, output)) return kErr;                                  // :895-896 (+h)
    return DepthFlowProjection_gpu_forward_kernel(
        stream, nelem(output), (int)input1->size[3], (int)input1->size[2], (int)input1->size[1],
        (int)input1->size[0], fillhole, S4(input1), S4(input2), S4(count), input1->data, input2->data,
        count->data, output->data);
}

int DepthFlowProjectionLayer_gpu_forward_ws(memc_stream_t stream, const memc_tensor4 *input1,
                                            const memc_tensor4 *input2, const memc_tensor4 *count,
                                            const memc_tensor4 *output, int fillhole, void *workspace,
                                            size_t workspace_bytes)
{
    if (!ok(input1) || !ok(input2) || !ok(count) || !ok(output) || !workspace) return kErr;
    if (input1->size[1] != 2) return kErr;
    if (input2->size[1] != 1) return kErr;
    if (!count_matches(input1, input2) || !count_matches(input1, count)) return kErr;
    if (!same_layout(input1, output)) return kErr;
    return DepthFlowProjection_gpu_forward_kernel_ws(
        stream, nelem(output), (int)input1->size[3], (int)input1->size[2], (int)input1->size[1],
        (int)input1->size[0], fillhole, S4(input1), S4(input2), S4(count), input1->data, input2->data,
        count->data, output->data, workspace, workspace_bytes);
}

int DepthFlowProjectionLayer_gpu_backward(memc_stream_t stream, const memc_tensor4 *input1,
                                          const memc_tensor4 *input2, const memc_tensor4 *count,
                                          const memc_tensor4 *output, const memc_tensor4 *gradoutput,
                                          const memc_tensor4 *gradinput1, const memc_tensor4 *gradinput2)
{
    if (!ok(input1) || !ok(input2) || !ok(count) || !ok(output) || !ok(gradoutput) || !ok(gradinput1) ||
        !ok(gradinput2))
        return kErr;
    if (input1->size[1] != 2) return kErr;                                          // :929
    if (input2->size[1] != 1) return kErr;                                          // :933
    if (!count_matches(input1, input2) || !count_matches(input1, count)) return kErr;   // :931-936
    if (!same_layout(input1, gradinput1)) return kErr;                              // :959-960
    if (!same_layout(input1, gradoutput) || !same_layout(input1, output)) return kErr;
    if (!same_layout(input2, gradinput2)) return kErr;
    return DepthFlowProjection_gpu_backward_kernel(
        stream, nelem(gradoutput), (int)input1->size[3], (int)input1->size[2], (int)input1->size[1],
        (int)input1->size[0], S4(input1), S4(input2), S4(count), input1->data, input2->data, count->data,
        output->data, gradoutput->data, gradinput1->data, gradinput2->data);
}

}  // extern "C"
