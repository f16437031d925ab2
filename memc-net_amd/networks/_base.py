"""What MEMC_Net and MEMC_Net_star share: FlowNetS motion estimation -> flow projection to t = 0.5, the two
U-Nets (interpolation filters, occlusion masks), the adaptive warp + blend, and a residual rectifier.

Reference: networks/MEMC_Net.py:77-170, networks/MEMC_Net_star.py:78-176 (the two forward() bodies differ only
in the context-feature branch and in the rectifier).
"""
import warnings

import torch
import torch.nn as nn
import torch.nn.functional as F

from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
from my_package.modules.FlowProjectionModule import FlowProjectionModule
try:        # extension of this repository's my_package; absent from the reference's
    from my_package.modules.FilterInterpolationBlendModule import FilterInterpolationBlendModule
except ImportError:
    FilterInterpolationBlendModule = None
try:        # likewise: scaling + x4 bilinear upsampling of the estimated flow as one kernel
    from my_package.modules.FlowUpsample4Module import FlowUpsample4Module
except ImportError:
    FlowUpsample4Module = None
try:        # likewise: frames + context features warped with one stream of flow / taps per direction
    from my_package.modules.FilterInterpolationCtxBlendModule import FilterInterpolationCtxBlendModule
except ImportError:
    FilterInterpolationCtxBlendModule = None

from ._blocks import FlowEstimator, run_unet, unet_head, unet_trunk


class MEMCNetBase(nn.Module):
    div_flow = 20                                      # FlowNetS predicts flow / 20

    def __init__(self, channel, filter_size, training, align_corners, batch_norm=False):
        super().__init__()
        self.filter_size = filter_size
        self.training = training
        self.align_corners = align_corners
        self.fused_blend = True             # one kernel for both warps + the blend where my_package offers it
        # FlowProjection's prologue (x div_flow / 2, x4 upsampling) as one kernel: measured 95 us against 84 us for the
        # torch expression (ATen's upsampling kernel is already write-bound), so off by default
        self.fused_upsample = False
        # frames + context features in one launch per direction (section 8f-3): measured SLOWER than the fused blend
        # plus two context warps (2705 vs 2485 us at 8x720x1280, 64 context channels: the context kernel sits at 239
        # of 256 VGPRs and the extra image chunk makes the compiler's chunk loop ~15 % slower), so off by default
        self.fused_context = False
        fs2 = filter_size * filter_size
        self.initScaleNets_filter = unet_trunk(2 * channel, align_corners, batch_norm)
        self.initScaleNets_filter1 = unet_head(fs2)
        self.initScaleNets_filter2 = unet_head(fs2)
        self.initScaleNets_occlusion = unet_trunk(2 * channel, align_corners, batch_norm)
        self.initScaleNets_occlusion1 = unet_head(1)
        self.initScaleNets_occlusion2 = unet_head(1)

    def load_state_dict(self, state_dict, *args, **kwargs):
        """The published MEMC-Net checkpoints were trained under PyTorch 0.2, whose bilinear `nn.Upsample` sampled
        with what is now `align_corners=True`; the reference's unmodified code under a current PyTorch -- and this
        class by default -- samples with `align_corners=False`, which shifts the x4 flow upsampling and the U-Net
        upsamplings by up to 1.5 pixels and visibly degrades the interpolated frames.  Loading weights into a model
        built with the default therefore warns (once per model); pass `align_corners=True` for published weights.
        Checkpoints trained WITH this code at align_corners=False are fine: silence the note with
        `model.warn_align_corners = False` (or `load_state_dict(..., warn_align_corners=False)`)."""
        warn = kwargs.pop("warn_align_corners", getattr(self, "warn_align_corners", True))
        if warn and not self.align_corners and not getattr(self, "_warned_align_corners", False):
            self._warned_align_corners = True
            warnings.warn("%s was built with align_corners=False (what the reference's code does under current "
                          "PyTorch).  Checkpoints published with MEMC-Net were trained with PyTorch 0.2 "
                          "(align_corners=True semantics): construct the model with align_corners=True to reproduce "
                          "their results." % type(self).__name__, stacklevel=2)
        return super().load_state_dict(state_dict, *args, **kwargs)

    def _init_convs(self, init_fn):
        """The reference initialises every Conv2d that exists at this point of its constructor and lets modules
        created afterwards keep their own initialisation."""
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                init_fn(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()

    def _add_flow_estimator(self):
        self.flownets = FlowEstimator()

    # ---- pieces -----------------------------------------------------------------------------------------
    def _bidirectional_flow(self, pair):
        """quarter-resolution single-direction flow -> full resolution, halved for the middle frame"""
        flow = self.flownets(pair)
        if self.fused_upsample and FlowUpsample4Module is not None and flow.is_cuda:
            return FlowUpsample4Module(self.div_flow, 2.0, self.align_corners)(flow)      # (div_flow * flow) / 2.0, x4
        return F.interpolate(self.div_flow * flow / 2.0, scale_factor=4, mode="bilinear", align_corners=self.align_corners)

    @staticmethod
    def _project(flow):
        # holes are filled only when no gradient is needed (FlowProjectionLayer.py:15)
        return FlowProjectionModule(flow.requires_grad)(flow)

    @staticmethod
    def _two_heads(trunk, head_a, head_b, x):
        feat = run_unet(trunk, x)
        return run_unet(head_a, feat), run_unet(head_b, feat)

    def _context(self, frame0, frame2, flows, filters, warp):
        """extra rectifier inputs (none in MEMC_Net; warped context features in MEMC_Net_star)"""
        return ()

    def _context_features(self, frame0, frame2):
        """unwarped context features of the two frames, or None when the model has none (MEMC_Net)"""
        return None

    # ---- forward ----------------------------------------------------------------------------------------
    def forward(self, input):
        if self.training:
            assert input.size(0) == 3
            frame0, frame1, frame2 = input[0], input[1], input[2]
        else:
            assert input.size(0) == 2
            frame0, frame2 = input[0], input[1]
        both = torch.cat((frame0, frame2), dim=1)
        swapped = torch.cat((frame2, frame0), dim=1)

        flows = [self._project(self._bidirectional_flow(both)), self._project(self._bidirectional_flow(swapped))]
        filters = list(self._two_heads(self.initScaleNets_filter, self.initScaleNets_filter1,
                                       self.initScaleNets_filter2, both))
        occlusions = [0.5 + o for o in self._two_heads(self.initScaleNets_occlusion, self.initScaleNets_occlusion1,
                                                       self.initScaleNets_occlusion2, both)]

        warp = FilterInterpolationModule()
        feats = self._context_features(frame0, frame2) if self.fused_context else None
        if feats is not None and FilterInterpolationCtxBlendModule is not None:
            # frames and context features ride the same flow + taps: one launch per direction, blend included
            blended, ctx0, ctx2 = FilterInterpolationCtxBlendModule()(
                frame0, frame2, feats[0], feats[1], flows[0], flows[1], filters[0], filters[1], occlusions[0],
                occlusions[1])
            extra = (ctx0, ctx2)
        else:
            if self.fused_blend and FilterInterpolationBlendModule is not None:
                blended = FilterInterpolationBlendModule()(frame0, frame2, flows[0], flows[1], filters[0], filters[1],
                                                           occlusions[0], occlusions[1])
            else:
                blended = (occlusions[0] * warp(frame0, flows[0], filters[0])
                           + occlusions[1] * warp(frame2, flows[1], filters[1]))
            extra = self._context(frame0, frame2, flows, filters, warp)

        rect_in = torch.cat((blended, flows[0], flows[1], filters[0], filters[1], occlusions[0], occlusions[1])
                            + tuple(extra), dim=1)
        rectified = blended + self._rectify(rect_in)

        if self.training:
            return [blended - frame1, rectified - frame1], [flows], [filters], [occlusions]
        return [blended, rectified], flows, filters, occlusions
