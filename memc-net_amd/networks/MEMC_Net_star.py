"""MEMC_Net_star, build-owned.

Interface of the reference class (networks/MEMC_Net_star.py:16-166): `MEMC_Net_star(channel=3, filter_size=4,
training=True)`; `forward(input)` with input [3, B, 3, H, W] when training, [2, B, 3, H, W] otherwise (H, W
multiples of 128, demo_HD720p.py:90-108); inference returns
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
import torch.nn as nn

from ._base import MEMCNetBase
from ._blocks import ContextConv, Rectifier


class MEMC_Net_star(MEMCNetBase):
    def __init__(self, channel=3, filter_size=4, training=True, align_corners=False):
        super().__init__(channel, filter_size, training, align_corners)
        fs2 = filter_size * filter_size
        self.ctx_ch = 64
        self.rectifyNet = Rectifier(channel + 2 * 2 + 2 * fs2 + 2 * self.ctx_ch + 2 * 1, blocks=10, feats=128)
        self._init_convs(nn.init.xavier_uniform_)      # reference :56-76
        self._add_flow_estimator()                     # these two keep their own initialisation
        self.ctxNet = ContextConv()

    def _context(self, frame0, frame2, flows, filters, warp):
        # context features ride the same flow + filters as the frames; no gradient reaches ctxNet (:284-285)
        ctx0 = warp(self.ctxNet(frame0), flows[0], filters[0]).detach()
        ctx2 = warp(self.ctxNet(frame2), flows[1], filters[1]).detach()
        return ctx0, ctx2

    def _context_features(self, frame0, frame2):
        return self.ctxNet(frame0), self.ctxNet(frame2)

    def _rectify(self, x):
        return self.rectifyNet(x)
