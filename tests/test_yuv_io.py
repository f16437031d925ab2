"""The HD demo's YUV 4:2:0 frame files and its per-pair loop (networks/yuv_io.py; reference yuv_frame_io.py:31-200,
demo_HD720p.py:60-170), on the CPU: byte layout, the colour matrices, round trips, the loop's file and scores."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(os.path.dirname(HERE), "memc-net_amd")


def _load(name):
    # networks/__init__ imports the HIP-backed operators; the IO helpers are plain numpy / torch: load them by path
    if "networks_cpu" not in sys.modules:
        pkg = importlib.util.module_from_spec(importlib.util.spec_from_loader("networks_cpu", loader=None, is_package=True))
        pkg.__path__ = [os.path.join(PKG, "networks")]
        sys.modules["networks_cpu"] = pkg
    full = "networks_cpu." + name
    if full not in sys.modules:
        spec = importlib.util.spec_from_file_location(full, os.path.join(PKG, "networks", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
    return sys.modules[full]


def test_matrices():
    Y = _load("yuv_io")
    assert np.allclose(Y.RGB_FROM_YUV @ Y.YUV_FROM_RGB, np.eye(3), atol=1e-12)
    assert np.allclose(Y.rgb_to_yuv([1.0, 1.0, 1.0]), [1.0, 0.0, 0.0], atol=1e-7)       # white: no chroma
    assert np.allclose(Y.rgb_to_yuv([0.0, 0.0, 1.0])[1], 0.436, atol=1e-3)                # U of pure blue
    assert np.allclose(Y.rgb_to_yuv([1.0, 0.0, 0.0])[2], 0.615, atol=1e-3)                # V of pure red


def test_plane_layout_and_raw_round_trip(tmp_path):
    """I420: Y (h*w), U (h/2*w/2), V, row-major; chroma comes back by pixel repetition."""
    Y = _load("yuv_io")
    h, w = 6, 8
    rng = np.random.default_rng(0)
    frames = []
    path = str(tmp_path / "raw.yuv")
    wr = Y.Yuv420Writer(path, from_rgb=False)
    for _ in range(3):
        y = rng.integers(0, 256, (h, w), dtype=np.uint8)
        u = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
        v = rng.integers(0, 256, (h // 2, w // 2), dtype=np.uint8)
        full = np.stack((y, u.repeat(2, 0).repeat(2, 1), v.repeat(2, 0).repeat(2, 1)), axis=-1)
        frames.append((y, u, v, full))
        wr.write(full)
    wr.close()
    raw = np.fromfile(path, dtype=np.uint8)
    assert raw.size == 3 * Y.frame_bytes(h, w) == 3 * (48 + 12 + 12)
    y0, u0, v0, _ = frames[0]
    assert np.array_equal(raw[:48], y0.ravel()) and np.array_equal(raw[48:60], u0.ravel()) and np.array_equal(raw[60:72], v0.ravel())
    rd = Y.Yuv420Reader(path, h, w, to_rgb=False)
    got, ok = rd.read(2)                                   # random access by frame index
    assert ok and np.array_equal(got, frames[2][3])
    got, ok = rd.read(0)
    assert ok and np.array_equal(got, frames[0][3])
    got, ok = rd.read()                                    # sequential: the frame after the last one read
    assert ok and np.array_equal(got, frames[1][3])
    assert rd.read(3) == (None, False)                     # past the end
    rd.close()
    with pytest.raises(ValueError):
        Y.Yuv420Reader(path, 5, 8)


def test_rgb_round_trip_through_a_file(tmp_path):
    """RGB -> file -> RGB: grey ramps are exact up to the two truncations; flat 2x2 colour blocks survive the chroma
    subsampling within two 8-bit steps."""
    Y = _load("yuv_io")
    h, w = 16, 32
    grey = np.repeat(np.linspace(0, 255, w).astype(np.uint8)[None, :, None], h, axis=0).repeat(3, axis=2)
    rng = np.random.default_rng(1)
    blocks = rng.integers(0, 256, (h // 2, w // 2, 3), dtype=np.uint8).repeat(2, 0).repeat(2, 1)
    path = str(tmp_path / "rgb.yuv")
    wr = Y.Yuv420Writer(path)
    wr.write(grey); wr.write(blocks)
    wr.close()
    rd = Y.Yuv420Reader(path, h, w)
    g, ok = rd.read(0)
    assert ok and np.abs(g.astype(int) - grey.astype(int)).max() <= 3      # (U = V = 127.5 truncates to 127: -0.002)
    b, ok = rd.read(1)
    rd.close()
    # chroma saturates for strongly coloured pixels (U is kept in [-0.5, 0.5] of a +-0.436 range: fine; V's +-0.615 clips)
    yuv = Y.rgb_to_yuv(blocks / 255.0)
    unclipped = np.abs(yuv[:, :, 2]) < 0.49
    assert ok and np.abs(b.astype(int) - blocks.astype(int))[unclipped].max() <= 3


def test_demo_loop_files_scores_and_batching(tmp_path):
    """Frames i and i + 2 in, frame i and the interpolated one out, scored against the real frame i + 1; batching pairs
    changes nothing."""
    Y = _load("yuv_io")
    h, w, n = 16, 32, 7
    rng = np.random.default_rng(2)
    base = rng.integers(40, 200, (h, w, 3)).astype(np.float64)
    src = str(tmp_path / "in.yuv")
    wr = Y.Yuv420Writer(src)
    for k in range(n):                                     # brightness ramps linearly: the mean of k and k + 2 is k + 1
        wr.write(np.clip(base + 6.0 * k, 0, 255).astype(np.uint8))
    wr.close()

    class Mean(torch.nn.Module):                           # stands in for the network: blend = rectified = the mean
        def forward(self, x):
            m = 0.5 * (x[0] + x[1])
            return [m, m], None, None, None

    outs = []
    for pairs in (1, 2):
        dst = str(tmp_path / ("out%d.yuv" % pairs))
        scores = Y.interpolate_yuv_sequence(Mean(), src, dst, h, w, torch.device("cpu"), first=0, last=n - 2,
                                            pairs_per_step=pairs)
        assert [s[0] for s in scores] == [1, 3, 5]         # pairs (0,2), (2,4), (4,6)
        assert all(err <= 2.0 and psnr >= 40.0 for _, err, psnr in scores), scores
        raw = np.fromfile(dst, dtype=np.uint8)
        assert raw.size == 6 * Y.frame_bytes(h, w)         # frame i, interpolated, for each of the three pairs
        outs.append(raw)
    assert np.array_equal(outs[0], outs[1])
    # the first frame written is the first input frame, re-encoded: its luma plane survives within the truncations
    # (the chroma of this per-pixel noise does not: it is resampled)
    a = np.fromfile(src, dtype=np.uint8)[:h * w].astype(int)
    b = outs[0][:h * w].astype(int)
    assert np.abs(a - b).mean() <= 2.0          # (a few saturated pixels move by more)
