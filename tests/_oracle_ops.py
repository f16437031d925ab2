"""TEST-ONLY stand-ins for `my_package.modules.*` backed by the CPU oracle, so that network-level code (the
reference's and the build-owned one) can run on CPU tensors in this container.  The product package has no CPU
path and never imports this file.  Forward and backward; semantics are the GPU ones (hole fill included)."""
import sys
import types

import torch
from torch import nn

from oracle import memc_oracle as O


def _np(t):
    return t.detach().contiguous().numpy()


class _FilterInterpolation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow, filt):
        ctx.save_for_backward(x, flow, filt)
        return torch.from_numpy(O.filter_interpolation_forward(_np(x), _np(flow), _np(filt)))

    @staticmethod
    def backward(ctx, g):
        x, flow, filt = ctx.saved_tensors
        return tuple(torch.from_numpy(a) for a in O.filter_interpolation_backward(_np(x), _np(flow), _np(filt), _np(g)))


class _FlowProjection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, fillhole):
        out, count = O.flow_projection_forward(_np(flow), fillhole)
        ctx.save_for_backward(flow, torch.from_numpy(count))
        return torch.from_numpy(out)

    @staticmethod
    def backward(ctx, g):
        flow, count = ctx.saved_tensors
        return torch.from_numpy(O.flow_projection_backward(_np(flow), _np(count), _np(g))), None


class _Interpolation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow):
        ctx.save_for_backward(x, flow)
        return torch.from_numpy(O.interpolation_forward(_np(x), _np(flow)))

    @staticmethod
    def backward(ctx, g):
        x, flow = ctx.saved_tensors
        return tuple(torch.from_numpy(a) for a in O.interpolation_backward(_np(x), _np(flow), _np(g)))


class FilterInterpolationModule(nn.Module):
    def forward(self, input1, input2, input3):
        return _FilterInterpolation.apply(input1, input2, input3)


class FlowProjectionModule(nn.Module):
    def __init__(self, requires_grad=True):
        super().__init__()
        self.fillhole = 1 if requires_grad == False else 0      # noqa: E712 -- reference FlowProjectionLayer.py:15

    def forward(self, input1):
        return _FlowProjection.apply(input1, self.fillhole)


class InterpolationModule(nn.Module):
    def forward(self, input1, input2):
        return _Interpolation.apply(input1, input2)


_SAVED = {}


def install():
    """Put oracle-backed `my_package.modules.*` into sys.modules (the real entries are kept for uninstall())."""
    if _SAVED:
        return
    for k in [k for k in sys.modules if k == "my_package" or k.startswith("my_package.")]:
        _SAVED[k] = sys.modules.pop(k)
    _SAVED.setdefault("", None)
    pkg = types.ModuleType("my_package"); pkg.__path__ = []
    mods = types.ModuleType("my_package.modules"); mods.__path__ = []
    sys.modules["my_package"], sys.modules["my_package.modules"] = pkg, mods
    for cls in (FilterInterpolationModule, FlowProjectionModule, InterpolationModule):
        m = types.ModuleType("my_package.modules." + cls.__name__)
        setattr(m, cls.__name__, cls)
        sys.modules[m.__name__] = m


def uninstall():
    """Remove the stand-ins (and anything imported against them) and put the real package entries back."""
    for k in [k for k in sys.modules if k == "my_package" or k.startswith("my_package.")
              or k == "networks" or k.startswith("networks.")]:
        del sys.modules[k]
    for k, v in _SAVED.items():
        if k:
            sys.modules[k] = v
    _SAVED.clear()
