"""Sub-networks of MEMC_Net_star as plain torch.nn stacks.  They are stock dense convolutions (MIOpen /
rocBLAS through PyTorch-ROCm); nothing here is on the hand-written hot path.

Parameter names are chosen so that `state_dict()` keys equal the reference's:
  flow estimator   networks/FlowNetS/FlowNetS.py:38-112   (conv1..conv6_1, deconv2..5, predict_flow2..6, upsampled_*)
  context conv     networks/ResNet/Resnet_conv1.py:216-245 (conv1, 7x7, 64 channels, no bias)
  rectifier        networks/EDSR/EDSR.py:9-46, common.py:25-44 (head / body / tail)
  U-Nets           networks/MEMC_Net_star.py:204-262 (a FLAT ModuleList of conv / relu / pool / upsample)
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------ flow estimator
def _lrelu_conv(cin, cout, k=3, stride=1):
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=True), nn.LeakyReLU(0.1, inplace=True))


def _lrelu_deconv(cin, cout):
    return nn.Sequential(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=True), nn.LeakyReLU(0.1, inplace=True))


class FlowEstimator(nn.Module):
    """FlowNetS without batch norm; returns the quarter-resolution flow (`flow2`)."""

    # (name, in, out, kernel, stride) of the contracting part
    _ENC = (("conv1", 6, 64, 7, 2), ("conv2", 64, 128, 5, 2), ("conv3", 128, 256, 5, 2), ("conv3_1", 256, 256, 3, 1),
            ("conv4", 256, 512, 3, 2), ("conv4_1", 512, 512, 3, 1), ("conv5", 512, 512, 3, 2),
            ("conv5_1", 512, 512, 3, 1), ("conv6", 512, 1024, 3, 2), ("conv6_1", 1024, 1024, 3, 1))
    # level -> channels entering the flow predictor / the next deconvolution
    _DEC_IN = {6: 1024, 5: 1026, 4: 770, 3: 386, 2: 194}
    _DEC_OUT = {5: 512, 4: 256, 3: 128, 2: 64}

    def __init__(self):
        super().__init__()
        for name, cin, cout, k, s in self._ENC:
            setattr(self, name, _lrelu_conv(cin, cout, k, s))
        for lvl in (5, 4, 3, 2):
            setattr(self, "deconv%d" % lvl, _lrelu_deconv(self._DEC_IN[lvl + 1], self._DEC_OUT[lvl]))
        for lvl in (6, 5, 4, 3, 2):
            setattr(self, "predict_flow%d" % lvl, nn.Conv2d(self._DEC_IN[lvl], 2, 3, 1, 1, bias=False))
        for lvl in (6, 5, 4, 3):
            setattr(self, "upsampled_flow%d_to_%d" % (lvl, lvl - 1), nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=False))
        for m in self.modules():                      # reference init, FlowNetS.py:72-80
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, 0.02 / n)
                if m.bias is not None:
                    m.bias.data.zero_()

    def forward(self, x):
        skip = {}
        x = self.conv2(self.conv1(x));   skip[2] = x
        x = self.conv3_1(self.conv3(x)); skip[3] = x
        x = self.conv4_1(self.conv4(x)); skip[4] = x
        x = self.conv5_1(self.conv5(x)); skip[5] = x
        feat = self.conv6_1(self.conv6(x))
        flow = self.predict_flow6(feat)
        for lvl in (5, 4, 3, 2):
            up = getattr(self, "upsampled_flow%d_to_%d" % (lvl + 1, lvl))(flow)
            feat = torch.cat((skip[lvl], getattr(self, "deconv%d" % lvl)(feat), up), 1)
            flow = getattr(self, "predict_flow%d" % lvl)(feat)
        return flow


# ------------------------------------------------------------------------------------------ context features
class ContextConv(nn.Module):
    """First ResNet-18 convolution at stride 1: 64 context channels per frame."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 1, 3, bias=False)
        self.conv1.weight.data.normal_(0, math.sqrt(2.0 / (7 * 7 * 64)))
        # the reference normalises ALL three channels with mean 0.485 and std (0.229, 0.224, 0.224)
        # (Resnet_conv1.py:236-238) -- reproduced as is
        self.register_buffer("_mean", torch.tensor([0.485, 0.485, 0.485]).view(1, 3, 1, 1), persistent=False)
        self.register_buffer("_std", torch.tensor([0.229, 0.224, 0.224]).view(1, 3, 1, 1), persistent=False)

    def forward(self, x):
        return self.conv1((x - self._mean) / self._std)


# ------------------------------------------------------------------------------------------ rectifier
class _ResBlock(nn.Module):
    def __init__(self, feats):
        super().__init__()
        self.body = nn.Sequential(nn.Conv2d(feats, feats, 3, padding=1), nn.ReLU(True), nn.Conv2d(feats, feats, 3, padding=1))

    def forward(self, x):
        return self.body(x) + x


class Rectifier(nn.Module):
    """EDSR-style residual stack: head conv, `blocks` residual blocks + conv, tail conv to RGB.
    (No global skip inside: the caller adds the interpolated frame, MEMC_Net_star.py:148.)"""

    def __init__(self, cin, blocks=10, feats=128):
        super().__init__()
        self.head = nn.Sequential(nn.Conv2d(cin, feats, 3, padding=1))
        self.body = nn.Sequential(*([_ResBlock(feats) for _ in range(blocks)] + [nn.Conv2d(feats, feats, 3, padding=1)]))
        self.tail = nn.Sequential(nn.Conv2d(feats, 3, 3, padding=1))

    def forward(self, x):
        return self.tail(self.body(self.head(x)))


# ------------------------------------------------------------------------------------------ U-Nets
class _Upsample2x(nn.Module):
    """Parameter-free bilinear x2.  `align_corners` is explicit: PyTorch 0.2 (which the reference targeted)
    interpolated with aligned corners, today's default for nn.Upsample(mode='bilinear') is not aligned."""

    def __init__(self, align_corners, factor=2):
        super().__init__()
        self.align_corners, self.factor = align_corners, factor

    def forward(self, x):
        return F.interpolate(x, scale_factor=self.factor, mode="bilinear", align_corners=self.align_corners)


# trunk of the filter / occlusion estimators: 'c<out>' = conv3x3 + ReLU, 'p' = 2x2 max-pool, 'u' = bilinear x2
_TRUNK = ("c32 c32 c32 p  c64 c64 p  c128 c128 p  c256 c256 p  c512 c512 p  c512 c512 "
          "c512 u c256  c256 u c128  c128 u c64  c64 u c32  c32 u c16").split()


def unet_trunk(cin, align_corners, batch_norm=False):
    """Flat list of modules (the reference extends a python list with Sequential objects, which flattens them:
    state-dict keys are indices into that flat list).  `batch_norm`: MEMC_Net normalises in front of every
    pooling and upsampling layer (MEMC_Net.py:293-320); MEMC_Net_star has those lines commented out."""
    mods, ch = [], cin
    for tok in _TRUNK:
        if tok in ("p", "u") and batch_norm:
            mods.append(nn.BatchNorm2d(ch))
        if tok == "p":
            mods.append(nn.MaxPool2d((2, 2)))
        elif tok == "u":
            mods.append(_Upsample2x(align_corners))
        else:
            out = int(tok[1:])
            mods += [nn.Conv2d(ch, out, (3, 3), 1, (1, 1)), nn.ReLU(inplace=False)]
            ch = out
    return nn.ModuleList(mods)


def unet_head(cout):
    return nn.ModuleList([nn.Conv2d(16, 16, (3, 3), 1, (1, 1)), nn.ReLU(inplace=False), nn.Conv2d(16, cout, (3, 3), 1, (1, 1))])


def run_unet(mods, x):
    """Encoder inputs of every pooling layer are added back after the matching upsampling layer
    (MEMC_Net_star.py:178-202, the non-'offset' branch)."""
    skips = []
    for m in mods:
        if isinstance(m, nn.MaxPool2d):
            skips.append(x)
        x = m(x)
        if isinstance(m, _Upsample2x):
            x = x + skips.pop()
    return x
