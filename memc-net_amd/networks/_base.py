"""What MEMC_Net and MEMC_Net_star share: FlowNetS motion estimation -> flow projection to t = 0.5, the two
U-Nets (interpolation filters, occlusion masks), the adaptive warp + blend, and a residual rectifier.

Reference: networks/MEMC_Net.py:77-170, networks/MEMC_Net_star.py:78-176 (the two forward() bodies differ only
in the context-feature branch and in the rectifier).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
from my_package.modules.FlowProjectionModule import FlowProjectionModule
try:        # extension of this repository's my_package; absent from the reference's
    from my_package.modules.FilterInterpolationBlendModule import FilterInterpolationBlendModule
except ImportError:
    FilterInterpolationBlendModule = None

from ._blocks import FlowEstimator, run_unet, unet_head, unet_trunk


class MEMCNetBase(nn.Module):
    div_flow = 20                                      # FlowNetS predicts flow / 20

    def __init__(self, channel, filter_size, training, align_corners, batch_norm=False):
        super().__init__()
        self.filter_size = filter_size
        self.training = training
        self.align_corners = align_corners
        self.fused_blend = True             # one kernel for both warps + the blend where my_package offers it
        fs2 = filter_size * filter_size
        self.initScaleNets_filter = unet_trunk(2 * channel, align_corners, batch_norm)
        self.initScaleNets_filter1 = unet_head(fs2)
        self.initScaleNets_filter2 = unet_head(fs2)
        self.initScaleNets_occlusion = unet_trunk(2 * channel, align_corners, batch_norm)
        self.initScaleNets_occlusion1 = unet_head(1)
        self.initScaleNets_occlusion2 = unet_head(1)

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
        flow = self.div_flow * self.flownets(pair) / 2.0
        return F.interpolate(flow, scale_factor=4, mode="bilinear", align_corners=self.align_corners)

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
