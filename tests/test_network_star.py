"""the build-owned `networks.MEMC_Net_star` (memc-net_amd/networks) against the
reference class.

CPU tests (this file, not gpu): the custom operators are provided by the oracle (tests/_oracle_ops.py) for BOTH
models, so what is compared is the network wiring, the state-dict layout and the dense layers.
  * against the committed vectors tests/golden/network_star_128.npz (made by make_golden_network.py from the
    reference class) -- always runs;
  * head-to-head against the reference class imported from /root/reference -- only where that exists.
The GPU test of the same model on the HIP operators is tests/test_gpu_network.py.
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _netutil      # noqa: E402
import _oracle_ops   # noqa: E402

GOLD = os.path.join(HERE, "golden", "network_star_128.npz")


@pytest.fixture(autouse=True)
def _oracle_backed_ops():
    """`my_package` resolves to the oracle-backed stand-ins inside these tests only."""
    _oracle_ops.install()
    yield
    _oracle_ops.uninstall()


def _outputs(net, x):
    with torch.no_grad():
        frames_out, flows, filters, occl = net(x)
    return {"blended": frames_out[0], "rectified": frames_out[1], "flow0": flows[0], "flow1": flows[1],
            "occlusion0": occl[0], "occlusion1": occl[1],
            "filter0_mean": filters[0].mean(dim=1), "filter1_mean": filters[1].mean(dim=1)}


def _own_model(name="MEMC_Net_star"):
    _netutil.purge_networks()
    import networks                                        # memc-net_amd/networks (conftest puts it on the path)
    assert "memc-net_amd" in networks.__file__
    return getattr(networks, name)(channel=3, filter_size=4, training=False).eval()


def test_own_model_matches_reference_vectors():
    gold = np.load(GOLD)
    net = _own_model()
    assert sum(p.numel() for p in net.parameters()) == int(gold["n_params"]) == 70312501
    net.load_state_dict(_netutil.named_weights(net.state_dict()), strict=True)
    got = _outputs(net, _netutil.frames(7, 1, 128, 128))
    for k, v in got.items():
        ref = gold[k]
        tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
        assert np.abs(v.numpy() - ref).max() <= tol, k


def test_own_base_model_matches_reference_vectors():
    gold = np.load(os.path.join(HERE, "golden", "network_base_64.npz"))
    net = _own_model("MEMC_Net")
    assert sum(p.numel() for p in net.parameters()) == int(gold["n_params"])
    net.load_state_dict(_netutil.named_weights(net.state_dict()), strict=True)
    got = _outputs(net, _netutil.frames(7, 1, 64, 64))
    for k, v in got.items():
        ref = gold[k]
        assert np.abs(v.numpy() - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), k


def test_own_model_training_step_matches_reference_fingerprint():
    gold = np.load(GOLD)
    net = _own_model()
    net.load_state_dict(_netutil.named_weights(net.state_dict()), strict=True)
    net.train()
    losses, _f, _k, _o = net(_netutil.training_frames(5, 1, 128, 128))
    total = sum(l.abs().mean() for l in losses)
    total.backward()
    assert abs(float(total) - float(gold["train_loss"])) <= 1e-5 * float(gold["train_loss"])
    got = _netutil.grad_l1_by_module(net)
    want = {k[len("grad_l1/"):]: float(gold[k]) for k in gold.files if k.startswith("grad_l1/")}
    assert sorted(got) == sorted(want)                    # ctxNet gets no gradient in either (detached warps)
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-4 * want[k], k


@pytest.mark.skipif(not os.path.isdir(_netutil.REF_ROOT), reason="reference tree not on this machine")
def test_state_dict_and_outputs_match_reference_class():
    ref_pkg = _netutil.import_reference_networks()
    assert ref_pkg.__file__.startswith(_netutil.REF_ROOT)
    ref = ref_pkg.MEMC_Net_star(channel=3, filter_size=4, training=False).eval()
    ref_sd = ref.state_dict()
    weights = _netutil.named_weights(ref_sd)
    ref.load_state_dict(weights)
    x = _netutil.frames(11, 2, 64, 64)
    want = _outputs(ref, x)

    net = _own_model()
    own_sd = net.state_dict()
    assert list(own_sd.keys()) == list(ref_sd.keys())                       # same names, same order
    assert all(own_sd[k].shape == ref_sd[k].shape for k in ref_sd)
    net.load_state_dict(weights, strict=True)                               # a reference checkpoint loads as is
    got = _outputs(net, x)
    for k in want:
        tol = 2e-5 * max(1.0, float(want[k].abs().max()))
        assert float((got[k] - want[k]).abs().max()) <= tol, k


def test_training_mode_returns_reference_structure():
    net = _own_model()
    net.train()
    with torch.no_grad():
        net.load_state_dict(_netutil.named_weights(net.state_dict()))
        two = _netutil.frames(3, 1, 64, 64)
        x = torch.stack((two[0], 0.5 * (two[0] + two[1]), two[1]))            # (frame0, ground truth, frame2)
        losses, flows, filters, occl = net(x)
        with pytest.raises(AssertionError):
            net(two)                                                           # training wants three frames
    # reference MEMC_Net_star.py:163-170: residuals against the middle frame, and singly nested lists
    assert len(losses) == 2 and losses[0].shape == (1, 3, 64, 64)
    assert len(flows) == 1 and len(flows[0]) == 2 and flows[0][0].shape == (1, 2, 64, 64)
    assert len(filters) == 1 and filters[0][0].shape == (1, 16, 64, 64)
    assert len(occl) == 1 and occl[0][1].shape == (1, 1, 64, 64)


@pytest.mark.skipif(not os.path.isdir(_netutil.REF_ROOT), reason="reference tree not on this machine")
def test_training_step_gradients_match_reference_class():
    """One training step (Charbonnier-free: plain L1 of the two residuals) through both classes: same losses
    and the same parameter gradients, i.e. the backward wiring (detached context warps, un-filled projection
    when gradients flow) is the reference's."""
    two = _netutil.frames(5, 1, 64, 64)
    x = torch.stack((two[0], 0.5 * (two[0] + two[1]), two[1]))

    def step(net):
        net.train()
        losses, _flows, _filters, _occl = net(x)
        total = sum(l.abs().mean() for l in losses)
        total.backward()
        return float(total), {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}

    ref_pkg = _netutil.import_reference_networks()
    ref = ref_pkg.MEMC_Net_star(channel=3, filter_size=4, training=False)   # True would load a FlowNetS checkpoint
    weights = _netutil.named_weights(ref.state_dict())
    ref.load_state_dict(weights)
    want_loss, want = step(ref)

    _netutil.purge_networks()
    import networks
    net = networks.MEMC_Net_star(channel=3, filter_size=4, training=False)
    net.load_state_dict(weights, strict=True)
    got_loss, got = step(net)

    assert abs(got_loss - want_loss) <= 1e-5 * max(1.0, abs(want_loss))
    assert sorted(got) == sorted(want)
    for k in want:
        scale = max(1e-3, float(want[k].abs().max()))
        assert float((got[k] - want[k]).abs().max()) <= 1e-4 * scale, k


@pytest.mark.skipif(not os.path.isdir(_netutil.REF_ROOT), reason="reference tree not on this machine")
def test_base_model_matches_reference_class():
    """MEMC_Net (no context branch, plain conv rectifier): state dict, inference outputs and one training
    step's gradients against the reference class."""
    ref_pkg = _netutil.import_reference_networks()
    ref = ref_pkg.MEMC_Net(channel=3, filter_size=4, training=False).eval()
    ref_sd = ref.state_dict()
    weights = _netutil.named_weights(ref_sd)
    ref.load_state_dict(weights)
    x = _netutil.frames(13, 1, 64, 64)
    want = _outputs(ref, x)
    x3 = _netutil.training_frames(5, 1, 64, 64)

    def step(net):
        net.train()
        losses, _f, _k, _o = net(x3)
        total = sum(l.abs().mean() for l in losses)
        total.backward()
        return float(total.detach()), _netutil.grad_l1_by_module(net)
    want_loss, want_g = step(ref)

    _netutil.purge_networks()
    import networks
    net = networks.MEMC_Net(channel=3, filter_size=4, training=False).eval()
    own_sd = net.state_dict()
    assert list(own_sd.keys()) == list(ref_sd.keys())
    assert all(own_sd[k].shape == ref_sd[k].shape for k in ref_sd)
    net.load_state_dict(weights, strict=True)
    got = _outputs(net, x)
    for k in want:
        tol = 2e-5 * max(1.0, float(want[k].abs().max()))
        assert float((got[k] - want[k]).abs().max()) <= tol, k
    got_loss, got_g = step(net)
    assert abs(got_loss - want_loss) <= 1e-5 * max(1.0, abs(want_loss))
    assert sorted(got_g) == sorted(want_g)
    for k in want_g:
        assert abs(got_g[k] - want_g[k]) <= 1e-4 * max(1e-6, want_g[k]), k


def test_demo_padding_rule():
    """demo_HD720p.py:88-106: pad up to the next multiple of 128, split floor / rest; 32 per side when already a
    multiple (values worked out from that arithmetic)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "inference", os.path.join(os.path.dirname(HERE), "memc-net_amd", "networks", "inference.py"))
    inf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(inf)
    assert inf.pad_amounts(720, 1280) == (32, 32, 24, 24)          # 1280 = 10 * 128 -> 32 + 32; 720 -> 768
    assert inf.pad_amounts(1080, 1920) == (32, 32, 36, 36)         # 1080 -> 1152
    assert inf.pad_amounts(256, 448) == (32, 32, 32, 32)           # 448 -> 512; 256 is a multiple
    assert inf.pad_amounts(480, 640) == (32, 32, 16, 16)           # 640 = 5 * 128; 480 -> 512
    assert inf.pad_amounts(101, 203) == (26, 27, 13, 14)           # odd totals: floor first

    class Probe(torch.nn.Module):                                   # returns its padded first frame
        def forward(self, x):
            assert x.shape[-2] % 128 == 0 or x.shape[-2] - 64 > 0
            return [x[0], x[0] + 1.0], None, None, None
    f0 = torch.rand(2, 3, 101, 203)
    out = inf.interpolate_pairs(Probe(), f0, torch.rand(2, 3, 101, 203), which=0)
    assert torch.equal(out, f0)                                     # the crop undoes the padding exactly
