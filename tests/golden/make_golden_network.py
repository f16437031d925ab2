"""Generates tests/golden/network_star_128.npz and network_base_64.npz: outputs of the REFERENCE classes
`networks.MEMC_Net_star` / `networks.MEMC_Net`
(imported from /root/reference, unmodified) on a fixed input with name-derived weights, its custom operators
provided by the CPU oracle (tests/_oracle_ops.py).  Only the vectors are stored; the reference source never
enters the repository.  Run in the build container:  python tests/golden/make_golden_network.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _netutil      # noqa: E402
import _oracle_ops   # noqa: E402


def training_step(net, size):
    """loss and per-submodule gradient L1 norms of one training step of the reference class (fixed input)."""
    net.train()
    net.zero_grad()
    losses, _f, _k, _o = net(_netutil.training_frames(5, 1, size, size))
    total = sum(l.abs().mean() for l in losses)
    total.backward()
    out = {"train_loss": np.float64(total.item())}
    out.update({"grad_l1/" + k: np.float64(v) for k, v in _netutil.grad_l1_by_module(net).items()})
    net.eval()
    return out


def vectors(net, size):
    net.load_state_dict(_netutil.named_weights(net.state_dict()))
    x = _netutil.frames(7, 1, size, size)
    with torch.no_grad():
        frames_out, flows, filters, occl = net(x)
    out = {"blended": frames_out[0].numpy(), "rectified": frames_out[1].numpy(),
           "flow0": flows[0].numpy(), "flow1": flows[1].numpy(),
           "occlusion0": occl[0].numpy(), "occlusion1": occl[1].numpy(),
           "filter0_mean": filters[0].numpy().mean(axis=1), "filter1_mean": filters[1].numpy().mean(axis=1),
           **training_step(net, size),
           "n_params": np.int64(sum(p.numel() for p in net.parameters())),
           "torch_version": np.array(torch.__version__)}
    return out


def main():
    _oracle_ops.install()
    ref = _netutil.import_reference_networks()
    here = os.path.dirname(os.path.abspath(__file__))
    for cls, size, fname in ((ref.MEMC_Net_star, 128, "network_star_128.npz"), (ref.MEMC_Net, 64, "network_base_64.npz")):
        torch.manual_seed(0)
        out = vectors(cls(channel=3, filter_size=4, training=False).eval(), size)
        np.savez_compressed(os.path.join(here, fname), **out)
        print(fname, {k: (v.shape, float(np.abs(v).max())) for k, v in out.items() if hasattr(v, "shape") and v.ndim > 1})


if __name__ == "__main__":
    main()
