"""Known-answer tests of the CPU checker (oracle/memc_oracle.c).  CPU only.

The reference has no tests, golden vectors or fixtures for this path, so these pin the
oracle as far as it can be pinned here:
  * the known answers SURVEY.md appendix A.7 recorded from the reference's own C code (same numpy Generator
    seeds and draw order as the survey's probe: default_rng(0): x=random, flow=normal(0,3), taps=random);
  * closed-form cases derived from the reference source (citations in the test bodies);
  * hand-computed cases of the hole-filling pass, which exists only in the reference's CUDA file
    (my_lib_kernel.cu:1742-1836) and therefore has no executable reference at all -- "restated, not
    oracle-checked".
"""
import numpy as np
import pytest


def test_a7_identity_warp(oracle):
    # zero flow + one-hot tap 5 reproduces the input exactly (first bullet)
    rng = np.random.default_rng(1)
    x = rng.random((1, 3, 8, 8)).astype(np.float32)
    flow = np.zeros((1, 2, 8, 8), np.float32)
    k = np.zeros((1, 16, 8, 8), np.float32)
    k[:, 5] = 1
    assert np.array_equal(oracle.filter_interpolation_forward(x, flow, k), x)
    # at zero flow only taps {0,1,4,5} (the top-left quadrant, weight (1-a)(1-b)=1) reach the output
    for tap in range(16):
        k2 = np.zeros_like(k)
        k2[:, tap] = 1
        nz = np.abs(oracle.filter_interpolation_forward(x, flow, k2)).max() > 0
        assert nz == (tap in (0, 1, 4, 5)), tap


def test_a7_half_pixel_box_filter(oracle):
    # flow == (0.5, 0.5), all taps 1: interior out = 0.25 * sum of the 4x4 window rows y-1..y+2, cols x-1..x+2;
    # last row/column have x2 > W-1 -> passthrough (second bullet)
    rng = np.random.default_rng(1)
    x = rng.random((1, 3, 8, 8)).astype(np.float32)
    flow = np.full((1, 2, 8, 8), 0.5, np.float32)
    k = np.ones((1, 16, 8, 8), np.float32)
    out = oracle.filter_interpolation_forward(x, flow, k)
    xp = np.pad(x[0], ((0, 0), (2, 2), (2, 2)), mode="edge")
    for c in range(3):
        for y in range(0, 7):
            for xx in range(0, 7):
                want = 0.25 * xp[c, y - 1 + 2:y + 3 + 2, xx - 1 + 2:xx + 3 + 2].astype(np.float64).sum()
                assert abs(out[0, c, y, xx] - want) < 1e-5
    assert np.array_equal(out[0, :, :, 7], x[0, :, :, 7])
    assert np.array_equal(out[0, :, 7, :], x[0, :, 7, :])


def test_a7_half_width_guard(oracle):
    # |fx| < W/2 is strict: fx == W/2 at column 0 (x2 in range) -> passthrough (third bullet)
    rng = np.random.default_rng(1)
    x = rng.random((1, 3, 8, 8)).astype(np.float32)
    flow = np.zeros((1, 2, 8, 8), np.float32)
    flow[:, 0, :, 0] = 4.0
    k = np.zeros((1, 16, 8, 8), np.float32)
    k[:, 9] = 1                                  # anything but the identity tap
    out = oracle.filter_interpolation_forward(x, flow, k)
    assert np.array_equal(out[0, :, :, 0], x[0, :, :, 0])


def test_a7_projection_constant_flow(oracle):
    # FlowProjection 1x2x4x4, flow == (1,0): duplicate adds on the clamped last column
    flow = np.zeros((1, 2, 4, 4), np.float32)
    flow[:, 0] = 1.0
    out, count = oracle.flow_projection_forward(flow, 0)
    want = np.array([[0, 1, 2, 3], [0, 2, 4, 6], [0, 2, 4, 6], [0, 3, 6, 9]], np.float32)
    assert np.array_equal(count[0, 0], want)
    assert np.array_equal(out[0, 0], np.where(want > 0, -1.0, 0.0).astype(np.float32))
    assert np.array_equal(out[0, 1], np.zeros((4, 4), np.float32))


def test_a7_random_128_checksums(oracle):
    # SURVEY A.7 last bullet, recorded from the reference C code on BASELINE config 1's tensors:
    # sum(out) = 48114.199 and FlowProjection leaves 452 holes.
    rng = np.random.default_rng(0)
    x = rng.random((1, 3, 128, 128)).astype(np.float32)
    flow = rng.normal(0, 3, (1, 2, 128, 128)).astype(np.float32)
    k = rng.random((1, 16, 128, 128)).astype(np.float32)
    out = oracle.filter_interpolation_forward(x, flow, k)
    assert abs(float(out.sum()) - 48114.199) < 0.01
    _, count = oracle.flow_projection_forward(flow, 0)
    assert int((count == 0).sum()) == 452


def test_bilinear_closed_forms(oracle):
    # Interpolation: zero flow is the identity; integer shift reads the shifted pixel; x2 == W-0.5 is still
    # valid (strict `< W`, my_lib.c:495) and blends the last column with itself; x2 < 0 -> 0.
    rng = np.random.default_rng(2)
    x = rng.random((2, 3, 6, 7)).astype(np.float32)
    z = np.zeros((2, 2, 6, 7), np.float32)
    assert np.array_equal(oracle.interpolation_forward(x, z), x)
    f = z.copy(); f[:, 0] = 2.0
    out = oracle.interpolation_forward(x, f)
    assert np.array_equal(out[..., :5], x[..., 2:])
    assert np.array_equal(out[..., 5:], np.zeros_like(out[..., 5:]))     # x2 >= W -> zero fill
    f = z.copy(); f[:, 0] = 0.5
    out = oracle.interpolation_forward(x, f)
    assert np.allclose(out[..., 6], x[..., 6])                           # R clamps to L at the border
    f = z.copy(); f[:, 1] = -0.25
    out = oracle.interpolation_forward(x, f)
    assert np.array_equal(out[:, :, 0], np.zeros_like(out[:, :, 0]))
    # channel check: Interpolation rejects C != 3 (my_lib.c:450), InterpolationCh accepts it (:678)
    x5 = rng.random((1, 5, 6, 7)).astype(np.float32)
    with pytest.raises(RuntimeError):
        oracle.interpolation_forward(x5, z[:1])
    assert np.array_equal(oracle.interpolation_ch_forward(x5, z[:1]), x5)


def test_fillhole_hand_cases(oracle):
    """Hole filling, restated from my_lib_kernel.cu:1776-1832 (no executable reference exists)."""
    H, W = 5, 6
    # flow that sends every pixel out of range except a handful we choose -> mostly holes
    far = 1000.0
    flow = np.full((1, 2, H, W), far, np.float32)
    # pixel (y=2,x=1) stays put with flow (0,0): hits cells (2,1),(2,2),(3,1),(3,2) with value -0 -> count 1
    flow[0, :, 2, 1] = 0.0
    # pixel (y=0,x=4) moves by (+0,+0) too but carries flow value via a second source: use (0.0, 0.0) + make it
    # distinguishable with a non-zero flow that still lands in range: (x=4,y=0) + (-1, +1) -> (3,1)
    flow[0, 0, 0, 4] = -1.0
    flow[0, 1, 0, 4] = 1.0
    out0, count = oracle.flow_projection_forward(flow, 0)
    out1, count1 = oracle.flow_projection_forward(flow, 1)
    assert np.array_equal(count, count1)
    want_count = np.zeros((H, W), np.float32)
    want_count[2:4, 1:3] += 1
    want_count[1:3, 3:5] += 1
    assert np.array_equal(count[0, 0], want_count)
    # non-holes are untouched by the fill
    nz = want_count > 0
    assert np.array_equal(out1[0][:, nz], out0[0][:, nz])
    # cell (1,0): left none, right -> first non-zero count in row 1 is x=3 (value +1,-1), up none => copy
    assert out1[0, 0, 1, 0] == out0[0, 0, 1, 3] == 1.0 and out1[0, 1, 1, 0] == -1.0
    # cell (4,1): left/right none in row 4; up -> (3,1) whose flow is (-0,-0): filled with 0 (flag set)
    assert out1[0, 0, 4, 1] == 0.0
    # cell (4,3): row 4 empty; up: rows 3 (count 0 at x=3), 2 (count 1 at x=3, value (1,-1)) -> copy
    assert out1[0, 0, 4, 3] == 1.0 and out1[0, 1, 4, 3] == -1.0
    # cell (0,0): row 0 has no valid cell, nothing above, and the DOWNWARD search is dead
    # (my_lib_kernel.cu:1799: `while (down_temp = 0.0f && ...)`): stays 0 although column 0... has none anyway
    assert out1[0, 0, 0, 0] == 0.0
    # cell (0,1): nothing left/right/up; the cell (2,1) BELOW is valid but the down search never runs -> 0
    assert out1[0, 0, 0, 1] == 0.0 and out1[0, 1, 0, 1] == 0.0
    # cell (2,0): right neighbour (2,1) valid with value (-0,-0) -> 0; and cell (2,5): left -> (2,4) = (1,-1)
    assert out1[0, 0, 2, 5] == 1.0 and out1[0, 1, 2, 5] == -1.0
    # cell (3,3): left -> (3,2) value 0 ; right none (row 3: x=4,5 count 0... ) ; up -> (2,3) value (1,-1):
    # mean of the two flagged neighbours
    assert out1[0, 0, 3, 3] == 0.5 and out1[0, 1, 3, 3] == -0.5


def test_depth_projection_reduces_to_plain_for_unit_depth(oracle):
    rng = np.random.default_rng(3)
    flow = rng.normal(0, 2, (2, 2, 16, 20)).astype(np.float32)
    ones = np.ones((2, 1, 16, 20), np.float32)
    for fh in (0, 1):
        a, ca = oracle.flow_projection_forward(flow, fh)
        b, cb = oracle.depth_flow_projection_forward(flow, ones, fh)
        assert np.array_equal(a, b) and np.array_equal(ca, cb)


def test_gradients_match_finite_differences(oracle):
    """The analytic backward of the bilinear warp w.r.t. the image is linear and exact: check it as the
    transpose of the forward (sum(gout * fwd(x)) is linear in x)."""
    rng = np.random.default_rng(4)
    x = rng.random((1, 3, 9, 11)).astype(np.float32)
    flow = rng.normal(0, 1.5, (1, 2, 9, 11)).astype(np.float32)
    k = rng.random((1, 16, 9, 11)).astype(np.float32)
    gout = rng.random((1, 3, 9, 11)).astype(np.float32)
    g1, _, g3 = oracle.filter_interpolation_backward(x, flow, k, gout)
    # <gout, FI(x)> restricted to valid sites is linear in x and in k: directional derivatives are exact
    dx = rng.random(x.shape).astype(np.float32)
    valid = oracle.filter_interpolation_forward(np.zeros_like(x), flow, k)   # passthrough sites give 0 here
    f0 = oracle.filter_interpolation_forward(x, flow, k)
    f1 = oracle.filter_interpolation_forward(x + dx, flow, k)
    # passthrough (invalid) sites copy x but get no gradient (A.2): mask them out of the forward difference
    rng2 = np.random.default_rng(0)
    probe = rng2.random(x.shape).astype(np.float32) + 1.0
    passthrough = oracle.filter_interpolation_forward(probe, flow, np.zeros_like(k)) != 0
    lhs = float(((f1 - f0).astype(np.float64) * gout * (~passthrough)).sum())
    rhs = float((g1.astype(np.float64) * dx).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(rhs))
    dk = rng.random(k.shape).astype(np.float32)
    f2 = oracle.filter_interpolation_forward(x, flow, k + dk)
    lhs = float(((f2 - f0).astype(np.float64) * gout * (~passthrough)).sum())
    rhs = float((g3.astype(np.float64) * dk).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(rhs))
    assert valid.shape == x.shape
