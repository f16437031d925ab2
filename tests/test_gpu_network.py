"""SURVEY.md section 8(f)-1 on the GPU: the build-owned `networks.MEMC_Net_star` running on the HIP operators
(my_package -> libmemc_hip.so) against vectors of the reference class (tests/golden/network_star_128.npz,
made on CPU from /root/reference by tests/golden/make_golden_network.py with the oracle behind the operators).
Dense layers run in MIOpen/rocBLAS fp32 here and in oneDNN there, so the tolerance is relative (stated below),
not the operators' 1e-4; the operators themselves are held to 1e-4 in test_gpu_parity.py."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _netutil      # noqa: E402

pytestmark = pytest.mark.gpu
MODELS = {"MEMC_Net_star": ("network_star_128.npz", 128), "MEMC_Net": ("network_base_64.npz", 64)}
REL_TOL = 2e-3          # of each output's max magnitude; 70 M-parameter fp32 net, two different conv libraries


@pytest.fixture(scope="module", params=sorted(MODELS))
def net(request):
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    _netutil.purge_networks()
    import my_package._ext.my_lib as my_lib                # the real one: raises if libmemc_hip.so is missing
    assert hasattr(my_lib, "FilterInterpolationLayer_gpu_forward")
    import networks
    assert "memc-net_amd" in networks.__file__
    m = getattr(networks, request.param)(channel=3, filter_size=4, training=False)
    m.load_state_dict(_netutil.named_weights(m.state_dict()), strict=True)
    m = m.cuda().eval()
    m.gold = np.load(os.path.join(HERE, "golden", MODELS[request.param][0]))
    m.size = MODELS[request.param][1]
    return m


def test_inference_matches_reference_vectors(net):
    gold = net.gold
    x = _netutil.frames(7, 1, net.size, net.size).cuda()
    with torch.no_grad():
        frames_out, flows, filters, occl = net(x)
    got = {"blended": frames_out[0], "rectified": frames_out[1], "flow0": flows[0], "flow1": flows[1],
           "occlusion0": occl[0], "occlusion1": occl[1],
           "filter0_mean": filters[0].mean(dim=1), "filter1_mean": filters[1].mean(dim=1)}
    report = {}
    for k, v in got.items():
        ref = gold[k]
        report[k] = float(np.abs(v.cpu().numpy() - ref).max() / max(1.0, np.abs(ref).max()))
    print("network rel err:", report)
    assert max(report.values()) <= REL_TOL, report


def test_training_step_matches_reference_fingerprint(net):
    gold = net.gold
    weights = {k: v.clone() for k, v in net.state_dict().items()}       # batch-norm statistics move in train()
    net.train()
    net.zero_grad()
    try:
        losses, _f, _k, _o = net(_netutil.training_frames(5, 1, net.size, net.size).cuda())
        total = sum(l.abs().mean() for l in losses)
        total.backward()
        torch.cuda.synchronize()
    finally:
        net.eval()
        net.load_state_dict(weights)
    assert abs(float(total.detach()) - float(gold["train_loss"])) <= 1e-3 * float(gold["train_loss"])
    got = _netutil.grad_l1_by_module(net)
    want = {k[len("grad_l1/"):]: float(gold[k]) for k in gold.files if k.startswith("grad_l1/")}
    assert sorted(got) == sorted(want)
    rel = {k: abs(got[k] - want[k]) / want[k] for k in want}
    print("network grad-L1 rel err:", rel)
    assert max(rel.values()) <= 1e-2, rel


def test_inference_720p_runs_and_is_deterministic_in_shape(net):
    """BASELINE config 4 shape (one 1280x720 pair; the U-Nets need H, W multiples of 64 -> 1280x768 padded
    the way the reference demo pads, demo_MiddleBury.py:74-95)."""
    x = torch.rand(2, 1, 3, 768, 1280, device="cuda")
    with torch.no_grad():
        frames_out, flows, filters, occl = net(x)
    torch.cuda.synchronize()
    assert frames_out[1].shape == (1, 3, 768, 1280) and torch.isfinite(frames_out[1]).all()
    assert flows[0].shape == (1, 2, 768, 1280) and filters[1].shape == (1, 16, 768, 1280)


def test_fused_context_path_gives_the_same_network(net):
    """section 8f-3: with `fused_context` the frames and their context features ride one launch per direction
    (FilterInterpolationCtxBlendModule); the network's outputs must not move (MEMC_Net has no context branch: no-op)."""
    x = _netutil.frames(11, 1, net.size, net.size).cuda()
    with torch.no_grad():
        base = net(x)[0]
        net.fused_context = True
        try:
            fused = net(x)[0]
        finally:
            net.fused_context = False
    for a, b in zip(base, fused):
        assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(a.abs().max()))


def test_config4_shard_of_four_pairs_through_the_demo_padding(net, tmp_path):
    """BASELINE config 4 as one GPU of eight sees it: 4 frame pairs of 1280 x 720 through networks/inference.py (replicate
    padding to 1344 x 768 the way demo_HD720p.py:88-113 pads, crop back) on the HIP operators.  Shard independence
    (see the comment at the check), and the same frames through the YUV 4:2:0 demo loop (networks/yuv_io.py) give the same interpolated frames."""
    import networks
    torch.manual_seed(4)
    f0, f2 = torch.rand(4, 3, 720, 1280, device="cuda"), torch.rand(4, 3, 720, 1280, device="cuda")
    assert networks.pad_amounts(720, 1280) == (32, 32, 24, 24)
    mid = networks.interpolate_pairs(net, f0, f2)
    torch.cuda.synchronize()
    assert mid.shape == (4, 3, 720, 1280) and torch.isfinite(mid).all()
    # Pair k alone against pair k inside the batch.  The hot-path operators are per frame pair and bit-identical between
    # the two (tests/test_gpu_baseline_configs.py checks that at batch 32); the dense layers are not -- MIOpen picks its
    # solver per shape -- and with these untrained weights the flow is rough, so an fp32 rounding difference that moves a
    # projected source across a pixel boundary changes a few output pixels by O(0.01).  Hence a distribution check: the
    # typical pixel agrees to 1e-4, and only a small fraction moves at all.
    def agree(a, b, what):
        diff = (a - b).abs().flatten()
        median, worst, moved = float(diff.median()), float(diff.max()), float((diff > 1e-3).float().mean())
        print("%s: median %.3g, max %.3g, fraction beyond 1e-3: %.4f" % (what, median, worst, moved))
        assert median <= 1e-4 and moved <= 0.02, (what, median, worst, moved)
    # (not even the same batch twice is bit-identical: some of MIOpen's solvers accumulate with atomics)
    agree(networks.interpolate_pairs(net, f0, f2), mid, "the same batch twice")
    for k in (0, 3):
        agree(networks.interpolate_pairs(net, f0[k:k + 1], f2[k:k + 1])[0], mid[k], "pair %d alone vs in the batch" % k)
    # the demo loop on a small YUV file: frames 0 and 2 in, frame 1 interpolated, batched or not
    h, w = 128, 192
    rng = np.random.default_rng(3)
    src = str(tmp_path / "clip.yuv")
    wr = networks.Yuv420Writer(src)
    for i in range(5):
        wr.write(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    wr.close()
    outs = []
    for pairs in (1, 2):
        dst = str(tmp_path / ("out%d.yuv" % pairs))
        scores = networks.interpolate_yuv_sequence(net, src, dst, h, w, torch.device("cuda"), first=0, last=3,
                                                   pairs_per_step=pairs)
        assert [s[0] for s in scores] == [1, 3]
        outs.append(np.fromfile(dst, dtype=np.uint8))
    assert outs[0].size == 4 * (h * w * 3 // 2)
    d = np.abs(outs[0].astype(int) - outs[1].astype(int))
    assert np.median(d) == 0 and float((d > 1).mean()) <= 0.02        # (same reasoning as above, in 8-bit steps)


def test_hd_demo_command_line(tmp_path, capsys):
    """tools/demo_hd720p.py (the reference's demo_HD720p.py loop as a command) on a small clip: output file and scores."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import importlib
    demo = importlib.import_module("demo_hd720p")
    import networks
    h, w = 128, 192
    rng = np.random.default_rng(8)
    src, dst = str(tmp_path / "clip.yuv"), str(tmp_path / "out.yuv")
    wr = networks.Yuv420Writer(src)
    for _ in range(5):
        wr.write(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    wr.close()
    demo.main(["--input", src, "--output", dst, "--height", str(h), "--width", str(w), "--model", "MEMC_Net",
               "--last", "3", "--pairs-per-step", "2"])
    out = capsys.readouterr().out
    assert "frame    1" in out and "frame    3" in out and "average over 2 interpolated frames" in out
    assert os.path.getsize(dst) == 4 * (h * w * 3 // 2)


def test_still_image_demo_command_line(tmp_path, capsys):
    """tools/demo_middlebury.py (the reference's demo_MiddleBury.py loop as a command) on a small scene tree: PNG files in,
    the interpolated frame and the difference picture out, scores printed; a grey scene is skipped as the reference skips it."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import importlib
    demo = importlib.import_module("demo_middlebury")
    import networks
    rng = np.random.default_rng(9)
    data, gt, out = tmp_path / "data", tmp_path / "gt", tmp_path / "out"
    for scene, (h, w) in (("Alpha", (128, 192)), ("Beta", (128, 192))):     # the HD demo test's size: no new MIOpen searches
        os.makedirs(str(data / scene))
        os.makedirs(str(gt / scene))
        base = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        networks.write_png(str(data / scene / "frame10.png"), base)
        networks.write_png(str(data / scene / "frame11.png"), np.roll(base, 2, axis=1))
        if scene == "Alpha":
            networks.write_png(str(gt / scene / "frame10i11.png"), np.roll(base, 1, axis=1))
    os.makedirs(str(data / "Grey"))
    for name in ("frame10.png", "frame11.png"):
        networks.write_png(str(data / "Grey" / name), rng.integers(0, 256, (64, 64), dtype=np.uint8))
    demo.main(["--data", str(data), "--gt", str(gt), "--output", str(out), "--model", "MEMC_Net"])
    text = capsys.readouterr().out
    assert "Alpha" in text and "interpolation error / PSNR" in text and "Beta" in text and "no ground truth" in text
    assert "for all 1 images" in text and "Grey" not in text
    rec = networks.read_png(str(out / "Alpha" / "frame10i11.png"))
    assert rec.shape == (128, 192, 3) and rec.dtype == np.uint8
    assert networks.read_png(str(out / "Beta" / "frame10i11.png")).shape == (128, 192, 3)
    diffs = [f for f in os.listdir(str(out / "Alpha")) if f.startswith("frame10i11_diff")]
    assert len(diffs) == 1 and networks.read_png(str(out / "Alpha" / diffs[0])).shape == (128, 192, 3)
    assert not os.path.exists(str(out / "Grey"))

