"""Generates tests/golden/ve_call_trace.json: the operator calls the REFERENCE class `networks.MEMC_Net_VE` (imported from
/root/reference, unmodified; its custom operators provided by the CPU oracle, tests/_oracle_ops.py) makes in one
inference pass and one training step -- operator, argument shapes, requires_grad of each argument, constructor
arguments -- at the shape the reference's Vimeo demo feeds it (a 448 x 256 septuplet padded to 512 x 320,
demo_Vimeo_VE.py:118-137) and at a small one, plus a checksum of every call's output at the small shape.
Only the trace is stored; the reference source never enters the repository.
Run in the build container:  python tests/golden/make_ve_call_trace.py
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _netutil      # noqa: E402
import _oracle_ops   # noqa: E402

TRACE = []


def _wrap(cls, op):
    orig_init, orig_fwd = cls.__init__, cls.forward

    def init(self, *a, **k):
        self._ctor = {"args": [bool(x) if isinstance(x, (bool, np.bool_)) else x for x in a], "kwargs": dict(k)}
        orig_init(self, *a, **k)

    def fwd(self, *tensors):
        out = orig_fwd(self, *tensors)
        TRACE.append({"op": op, "ctor": getattr(self, "_ctor", {"args": [], "kwargs": {}}),
                      "shapes": [list(t.shape) for t in tensors],
                      "requires_grad": [bool(t.requires_grad) for t in tensors],
                      "contiguous": [bool(t.is_contiguous()) for t in tensors],
                      "fillhole": getattr(self, "fillhole", None),
                      "out_shape": list(out.shape), "out_abs_sum": float(out.detach().double().abs().sum())})
        return out
    cls.__init__, cls.forward = init, fwd


def run(ref, H, W, training):
    torch.manual_seed(0)
    # (constructed as the demo does, training=False: with training=True the constructor wants models/flownets_pytorch.pth;
    # the forward's training branch looks at nn.Module.training, which .train() sets)
    net = ref.MEMC_Net_VE(batch=1, channel=3, width=None, height=None, scale_num=1, scale_ratio=2, temporal=False,
                          filter_size=4, save_which=1, debug=False, offset_scale=None, cuda_available=False, cuda_id=None,
                          training=False)
    net.load_state_dict(_netutil.named_weights(net.state_dict()))
    g = torch.Generator().manual_seed(11)
    xs = [torch.rand((1, 3, H, W), generator=g) for _ in range(7)]
    del TRACE[:]
    if training:
        net.train()
        losses = net(xs, torch.rand((1, 3, H, W), generator=g))
        sum(l.abs().mean() for l in losses).backward()
    else:
        net.eval()
        with torch.no_grad():
            net(xs)
    return list(TRACE)


def main():
    _oracle_ops.install()
    for cls, op in ((_oracle_ops.FilterInterpolationModule, "FilterInterpolationModule"),
                    (_oracle_ops.FlowProjectionModule, "FlowProjectionModule"),
                    (_oracle_ops.InterpolationModule, "InterpolationModule")):
        _wrap(cls, op)
    # the constructor downloads ImageNet weights for its context network (ResNet/Resnet_conv1.py:290): no network here,
    # and every weight is replaced by the name-derived ones anyway -- the download returns nothing
    import torch.utils.model_zoo as model_zoo
    model_zoo.load_url = lambda *a, **k: {}
    ref = _netutil.import_reference_networks()
    out = {"reference": "networks/MEMC_Net_VE.py (unmodified), demo_Vimeo_VE.py:40-44 constructor arguments",
           "torch_version": torch.__version__, "runs": {}}
    for name, H, W, training in (("inference 64x64", 64, 64, False), ("training 64x64", 64, 64, True),
                                 ("inference 320x512 (448x256 septuplet padded as the demo does)", 320, 512, False)):
        calls = run(ref, H, W, training)
        if H > 64:
            for c in calls:
                c.pop("out_abs_sum")            # (name-derived weights at this size: shapes only)
        out["runs"][name] = calls
        print(name, len(calls), "calls:", sorted(set((c["op"], tuple(c["shapes"][0])) for c in calls)))
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ve_call_trace.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
