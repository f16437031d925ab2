"""MEMC_Net (the base model of the paper), build-owned.

Interface and state-dict keys of the reference class (networks/MEMC_Net.py:13-170): like MEMC_Net_star but with
no context branch, batch norm in the two U-Nets (:293-320) and a plain 8-convolution rectifier (`get_RectifyNet2`, :240-250 -- a flat ModuleList, keys
`rectifyNet.0.weight`, `rectifyNet.2.weight`, ... `rectifyNet.14.weight`); convolutions are Kaiming-uniform
initialised (:56).  Same deliberate differences as MEMC_Net_star (no checkpoint I/O in the constructor,
explicit `align_corners`).
"""
import torch.nn as nn

from ._base import MEMCNetBase


def _kaiming(w):
    nn.init.kaiming_uniform_(w, a=0, mode="fan_in")


class MEMC_Net(MEMCNetBase):
    def __init__(self, channel=3, filter_size=4, training=True, align_corners=False):
        super().__init__(channel, filter_size, training, align_corners, batch_norm=True)
        fs2 = filter_size * filter_size
        mods, ch = [], channel + 2 * 2 + 2 * fs2 + 2 * 1            # 3 + 4 + 32 + 2
        for _ in range(7):
            mods += [nn.Conv2d(ch, 64, (3, 3), 1, (1, 1)), nn.ReLU(inplace=False)]
            ch = 64
        mods.append(nn.Conv2d(64, channel, (3, 3), 1, (1, 1)))
        self.rectifyNet = nn.ModuleList(mods)
        self._init_convs(_kaiming)                     # reference :35, before FlowNetS exists (:37-40)
        self._add_flow_estimator()                     # keeps its own initialisation

    def _rectify(self, x):
        for m in self.rectifyNet:
            x = m(x)
        return x
