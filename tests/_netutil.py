"""Helpers shared by the network-level tests: deterministic weights by parameter NAME (so that the reference
class and the build-owned class get identical tensors iff their state-dict keys and shapes agree), and the
import shims the reference's `networks` package needs in this image."""
import importlib
import math
import sys
import types
import zlib

import torch

REF_ROOT = "/root/reference"


def named_weights(state_dict):
    """name -> tensor, a pure function of (name, shape)."""
    out = {}
    for name, t in state_dict.items():
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        if t.dim() >= 2:
            fan_in = max(1, int(torch.tensor(t.shape[1:]).prod()))
            gain = 1.0                          # keeps the random flow within a few pixels and outputs O(1)
            v = torch.randn(t.shape, generator=g) * (gain / math.sqrt(fan_in))
        elif name.endswith("num_batches_tracked"):
            v = torch.zeros(t.shape)
        elif name.endswith("running_var"):
            v = 0.5 + torch.rand(t.shape, generator=g)
        elif name.endswith(".weight"):                                   # 1-D weight: a batch-norm scale
            v = 1.0 + 0.1 * torch.randn(t.shape, generator=g)
        else:                                                            # biases, running means
            v = torch.randn(t.shape, generator=g) * 0.01
        out[name] = v.to(t.dtype)
    return out


def frames(seed, B, H, W):
    g = torch.Generator().manual_seed(seed)
    return torch.rand((2, B, 3, H, W), generator=g)


def training_frames(seed, B, H, W):
    two = frames(seed, B, H, W)
    return torch.stack((two[0], 0.5 * (two[0] + two[1]), two[1]))        # (frame0, middle ground truth, frame2)


def grad_l1_by_module(net):
    """sum |grad| per top-level submodule (a compact fingerprint of a whole backward pass)."""
    acc = {}
    for name, p in net.named_parameters():
        if p.grad is not None:
            top = name.split(".")[0]
            acc[top] = acc.get(top, 0.0) + float(p.grad.detach().double().abs().sum())
    return acc


def import_reference_networks():
    """Imports the reference's `networks` package (needs /root/reference on disk) with stand-ins for the
    modules it imports at module scope but never uses on the inference path."""
    for name in ("torchvision", "torchvision.models", "skimage", "skimage.io"):
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules.setdefault(name, types.ModuleType(name))
    import scipy.misc as sm
    for fn in ("imread", "imsave", "imshow", "imresize"):
        if not hasattr(sm, fn):
            setattr(sm, fn, lambda *a, **k: None)
    purge_networks()
    sys.path.insert(0, REF_ROOT)
    try:
        return importlib.import_module("networks")
    finally:
        sys.path.remove(REF_ROOT)


def purge_networks():
    for k in [k for k in sys.modules if k == "networks" or k.startswith("networks.") or k == "Stack"]:
        del sys.modules[k]
