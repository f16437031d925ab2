"""MEMC_Net_star, build-owned.

Interface of the reference class (networks/MEMC_Net_star.py:16-166): `MEMC_Net_star(channel=3, filter_size=4,
training=True)`; `forward(input)` with input [3, B, 3, H, W] when training, [2, B, 3, H, W] otherwise (H, W
multiples of 128 -- five poolings, demo_HD720p.py:90-108); inference returns
`([frame, rectified_frame], [flow0, flow1], [filter0, filter1], [occlusion0, occlusion1])`.
State-dict keys are the reference's, so `load_state_dict(reference_checkpoint)` works (checked in
tests/test_network_star.py against the reference class itself).

Differences, deliberate:
  * no file I/O in the constructor (the reference loads "models/flownets_pytorch.pth" and downloads ResNet-18
    when training=True, :39,47); use `load_state_dict`;
  * `align_corners` of the bilinear upsamplings is a constructor argument.  Default False = what the
    reference's unmodified code does under current PyTorch; True = PyTorch 0.2 behaviour, which the published
    weights were trained with.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
from my_package.modules.FlowProjectionModule import FlowProjectionModule

from ._blocks import ContextConv, FlowEstimator, Rectifier, run_unet, unet_head, unet_trunk


class MEMC_Net_star(nn.Module):
    def __init__(self, channel=3, filter_size=4, training=True, align_corners=False):
        super().__init__()
        self.filter_size = filter_size
        self.training = training
        self.align_corners = align_corners
        fs2 = filter_size * filter_size
        self.initScaleNets_filter = unet_trunk(2 * channel, align_corners)
        self.initScaleNets_filter1 = unet_head(fs2)
        self.initScaleNets_filter2 = unet_head(fs2)
        self.initScaleNets_occlusion = unet_trunk(2 * channel, align_corners)
        self.initScaleNets_occlusion1 = unet_head(1)
        self.initScaleNets_occlusion2 = unet_head(1)
        self.ctx_ch = 64
        self.rectifyNet = Rectifier(channel + 2 * 2 + 2 * fs2 + 2 * self.ctx_ch + 2 * 1, blocks=10, feats=128)
        for m in self.modules():                      # reference init of everything built so far, :56-76
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()
        self.flownets = FlowEstimator()               # these two keep their own initialisation
        self.ctxNet = ContextConv()
        self.div_flow = 20

    # ---- pieces -----------------------------------------------------------------------------------------
    def _bidirectional_flow(self, pair):
        """quarter-resolution single-direction flow -> full resolution, halved for the middle frame (:172-176)"""
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
        contexts = [self.ctxNet(frame0), self.ctxNet(frame2)]
        occlusions = [0.5 + o for o in self._two_heads(self.initScaleNets_occlusion, self.initScaleNets_occlusion1,
                                                       self.initScaleNets_occlusion2, both)]

        warp = FilterInterpolationModule()
        blended = occlusions[0] * warp(frame0, flows[0], filters[0]) + occlusions[1] * warp(frame2, flows[1], filters[1])
        ctx0 = warp(contexts[0], flows[0], filters[0]).detach()
        ctx2 = warp(contexts[1], flows[1], filters[1]).detach()

        rect_in = torch.cat((blended, flows[0], flows[1], filters[0], filters[1], occlusions[0], occlusions[1],
                             ctx0, ctx2), dim=1)
        rectified = blended + self.rectifyNet(rect_in)

        if self.training:
            return [blended - frame1, rectified - frame1], [flows], [filters], [occlusions]
        return [blended, rectified], flows, filters, occlusions
