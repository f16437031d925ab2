"""Parity of the HIP path (through the C ABI / the drop-in modules) with the CPU oracle.  GPU only.

Tolerance: BASELINE.json's north_star states "within 1e-4 abs float tolerance"; every comparison below goes through
tests/_parity.py: `|hip - oracle| <= 1e-4` wherever |oracle| <= 10, a relative term (1e-5 unless stated) only beyond
(sums of many scattered contributions: fp32 atomics add in a different order than the oracle's sequential loop), and
the observed maximum of every check is recorded (gpurun_out/parity_errors.json, copied to profiles/ per round).
Integer-valued results (FlowProjection's `count`) must match bit-for-bit.
"""
import glob
import os

import numpy as np
import pytest
import torch

import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _parity as P                        # noqa: E402
from tools import synth                    # noqa: E402

pytestmark = pytest.mark.gpu

ATOL = P.ATOL
RTOL = P.RTOL


def dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU test run without a GPU: the HIP path cannot be exercised (no fallback exists)")
    return torch.device("cuda:0")


def T(a, requires_grad=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev())
    return t.requires_grad_(True) if requires_grad else t


def N(t):
    return t.detach().cpu().numpy()


def close(got, want, what, rtol=RTOL, cancel=0.0):
    """tests/_parity.py: 1e-4 abs where |want| <= 10, max(1e-4, rtol * |want|) beyond; the observed error is recorded.
    cancel (P.CANCEL, DepthFlowProjection's gradinput2 only): the absolute bound is max(1e-4, cancel * max|want|) -- see _parity.py."""
    return P.close(np.asarray(got), np.asarray(want), what, rtol, cancel)  # (rtol = 0.0: 1e-4 absolute everywhere)


CASES = [
    # (B, C, H, W, flow kind, sigma, seed)
    (1, 3, 128, 128, "iid", 3.0, 0),       # BASELINE configs[0] shape
    (2, 3, 37, 53, "iid", 6.0, 1),         # ragged: not a multiple of the 64x4 tile
    (1, 3, 5, 3, "iid", 1.0, 2),           # smaller than one wavefront
    (3, 1, 17, 70, "smooth", 4.0, 3),
    (1, 64, 24, 40, "smooth", 4.0, 4),     # the context-feature warp (C = 64)
    (2, 3, 64, 192, "smooth", 12.0, 5),    # large smooth motion
    (1, 3, 40, 40, "iid", 30.0, 6),        # mostly out of range: passthrough / |f| < W/2 guard paths
    (2, 5, 33, 65, "zero", None, 7),
    # vector (16 B per lane, LDS-tiled) path: W % 4 == 0
    (2, 3, 64, 256, "iid", 20.0, 8),       # source box far larger than the LDS budget: per-site global fallback
    (1, 5, 16, 32, "iid", 2.0, 9),         # channel chunks 4 + 1
    (1, 6, 20, 64, "smooth", 4.0, 10),     # 4 + 2
    (1, 7, 20, 64, "smooth", 4.0, 11),     # 4 + 3
    (2, 3, 100, 132, "smooth", 8.0, 12),   # ragged tiles (132 = 2 * 64 + 4, 100 = 6 * 16 + 4)
    (1, 8, 48, 96, "iid", 3.0, 13),        # two full chunks
    # source boxes beyond the LDS budget: band sweep (and, past 6 bands, the scalar path)
    (1, 8, 64, 128, "smooth", 14.0, 14),   # pipelined multi-chunk kernel, several bands, lanes split over bands
    (1, 12, 40, 128, "iid", 8.0, 15),
    (1, 3, 96, 256, "smooth", 25.0, 16),
]
IDS = ["%dx%dx%dx%d-%s" % c[:5] for c in CASES]


def make(case):
    B, C, H, W, kind, sigma, seed = case
    rng = np.random.default_rng(seed)
    return dict(x=synth.np_image(rng, B, C, H, W), flow=synth.np_flow(rng, B, H, W, kind, sigma),
                filt=synth.np_filter(rng, B, H, W), gout=synth.np_image(rng, B, C, H, W),
                depth=synth.np_depth(rng, B, H, W), gflow=rng.random((B, 2, H, W), dtype=np.float32))


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_filter_interpolation(oracle, case):
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    d = make(case)
    x, f, k = T(d["x"], True), T(d["flow"], True), T(d["filt"], True)
    out = FilterInterpolationModule()(x, f, k)
    out.backward(T(d["gout"]))
    close(N(out), oracle.filter_interpolation_forward(d["x"], d["flow"], d["filt"]), "forward")
    g1, g2, g3 = oracle.filter_interpolation_backward(d["x"], d["flow"], d["filt"], d["gout"])
    close(N(x.grad), g1, "gradinput1", RTOL)
    close(N(f.grad), g2, "gradinput2", RTOL)
    close(N(k.grad), g3, "gradinput3", RTOL)


C3_ARMS = [-1, 60, 61, 0]
C3_ARM_IDS = ["product: packed planes, image gradient first (tile height by the grid)", "64 x 16 tiles whatever the grid",
              "64 x 8 tiles whatever the grid", "arm: fp64 plane per colour (rounds 1-2)"]
C3_FLOWS = [(2, 100, 132, "smooth", 8.0), (1, 96, 256, "smooth", 25.0), (1, 64, 192, "converge", None),
            (2, 64, 256, "iid", 20.0), (1, 48, 64, "zero", None), (1, 37, 52, "smooth", 3.0)]


def _c3_inputs(case, seed=0, signed=False):
    B, H, W, kind, sigma = case
    rng = np.random.default_rng(1000 + seed + H + W)
    if kind == "converge":
        fn = _many_channel_flows("converge", rng, B, H, W)
    else:
        fn = synth.np_flow(rng, B, H, W, kind, sigma)
    xn, kn, gn = synth.np_image(rng, B, 3, H, W), synth.np_filter(rng, B, H, W), synth.np_image(rng, B, 3, H, W)
    if signed:
        kn = (rng.standard_normal(kn.shape) * 0.3).astype(np.float32)
        gn = rng.standard_normal(gn.shape).astype(np.float32)
    return xn, fn, kn, gn


def _c3_backward(my_lib, xn, fn, kn, gn, fill=0.0):
    h1 = torch.full(xn.shape, fill, device=dev())
    h2, h3 = torch.full(fn.shape, 3.0, device=dev()), torch.full(kn.shape, 3.0, device=dev())      # defined by the kernel
    assert my_lib.FilterInterpolationLayer_gpu_backward(T(xn), T(fn), T(kn), T(gn), h1, h2, h3) == 0
    return N(h1), N(h2), N(h3)


@pytest.mark.parametrize("arm", C3_ARMS, ids=C3_ARM_IDS)
def test_filter_interpolation_backward_rgb_arms(oracle, arm):
    """The RGB backward (fi_bwd_c3.hip) -- the product's kernel and the arms kept beside it in the measurement build -- on
    smooth, banded (box beyond the LDS budget), converging (hundreds of taps per cell), i.i.d. (scalar fallback sites),
    zero and odd-sized flows, with positive and with signed gradients / taps."""
    from tools import measure as M          # forced paths exist in the measurement build only
    my_lib = M.bound()
    try:
        M.set_variant("fi_bwd", arm)
        for ci, case in enumerate(C3_FLOWS):
            for signed in (False, True):
                xn, fn, kn, gn = _c3_inputs(case, ci, signed)
                g1, g2, g3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
                h1, h2, h3 = _c3_backward(my_lib, xn, fn, kn, gn, fill=0.5)
                tag = "%s signed=%d arm %d" % (case, signed, arm)
                close(h1, g1 + 0.5, "gradinput1 += " + tag, 3 * RTOL)
                close(h2, g2, "gradinput2 " + tag, RTOL)
                close(h3, g3, "gradinput3 " + tag, RTOL)
    finally:
        M.reset()


@pytest.mark.parametrize("arm", C3_ARMS, ids=C3_ARM_IDS)
def test_filter_interpolation_backward_rgb_scaling_and_special_values(oracle, arm):
    """Properties of the RGB backward's accumulation (fi_bwd_c3.hip): (a) scaling gradoutput or the taps by a power
    of two scales gradinput1 by that power (packed planes: the tile's block exponent moves with it, the same integers
    are added; what is left is the order in which the tiles' fp32 flushes reach a cell: a few ulp);
    (b) a zero gradoutput leaves gradinput1 untouched; (c) NaN / Inf in gradoutput or in a tap land exactly where the
    reference puts them (the tile takes per-site atomics) and nowhere else; (d) huge and tiny magnitudes."""
    from tools import measure as M
    my_lib = M.bound()
    case = (2, 80, 192, "smooth", 6.0)
    xn, fn, kn, gn = _c3_inputs(case, 7, signed=True)
    try:
        M.set_variant("fi_bwd", arm)
        base = _c3_backward(my_lib, xn, fn, kn, gn)[0]
        close(base, oracle.filter_interpolation_backward(xn, fn, kn, gn)[0], "gradinput1 base", 3 * RTOL)
        for sg, st in ((2.0 ** 40, 1.0), (2.0 ** -40, 1.0), (1.0, 2.0 ** 30), (2.0 ** -20, 2.0 ** -30), (2.0 ** 60, 2.0 ** 50)):
            got = _c3_backward(my_lib, xn, fn, (kn * np.float32(st)), (gn * np.float32(sg)))[0]
            want = base.astype(np.float64) * sg * st
            scale = float(np.abs(want).max())
            bad = np.abs(got.astype(np.float64) - want) > 4e-7 * np.abs(want) + 2e-7 * scale
            assert not bad.any(), "power-of-two scaling (%g, %g): %d cells off by more than a few ulp" % (sg, st, int(bad.sum()))
        # (b)
        h1 = _c3_backward(my_lib, xn, fn, kn, np.zeros_like(gn), fill=0.25)[0]
        assert np.array_equal(h1, np.full_like(h1, 0.25)), "zero gradoutput must add nothing"
        # (c)
        g_bad, k_bad = gn.copy(), kn.copy()
        g_bad[0, 1, 10, 20] = np.nan
        g_bad[0, 2, 50, 100] = np.inf
        k_bad[1, 5, 30, 60] = -np.inf
        k_bad[1, 9, 70, 150] = np.nan
        w1, w2, w3 = oracle.filter_interpolation_backward(xn, fn, k_bad, g_bad)
        h1, h2, h3 = _c3_backward(my_lib, xn, fn, k_bad, g_bad)
        assert np.array_equal(np.isnan(h1), np.isnan(w1)), "NaN cells of gradinput1: %d vs %d" % (np.isnan(h1).sum(), np.isnan(w1).sum())
        assert np.array_equal(np.isinf(h1), np.isinf(w1)), "Inf cells of gradinput1"
        assert 0 < np.isnan(w1).sum() < 200
        close(h1, w1, "gradinput1 with NaN / Inf inputs", 3 * RTOL)
        # (d) magnitudes far from 1: relative accuracy against the oracle in float64 terms
        for sg, st in ((3.7e12, 5.1e-3), (1.3e-17, 9.0e-9)):
            kk, gg = (kn * np.float32(st)), (gn * np.float32(sg))
            want = oracle.filter_interpolation_backward(xn, fn, kk, gg)[0].astype(np.float64)
            got = _c3_backward(my_lib, xn, fn, kk, gg)[0].astype(np.float64)
            scale = float(np.abs(want).max())
            assert float(np.abs(got - want).max()) <= 1e-5 * scale, "magnitudes (%g, %g): %.3g of the largest gradient" % (
                sg, st, float(np.abs(got - want).max()) / scale)
    finally:
        M.reset()


def test_filter_interpolation_backward_without_image_gradient(oracle):
    """EXTENSION (include/memc_warp.h): gradinput1 == NULL -- the caller does not want the image gradient (the reference's
    networks warp frames that are data, MEMC_Net_star.py:266-277): the RGB kernel then computes the flow and tap gradients
    alone.  Same gradinput2 / gradinput3 as the full call, for every flow kind; shapes the extension does not serve return
    -1 and write nothing; the Python layer asks for it exactly when autograd does not need input1's gradient."""
    import my_package._ext.my_lib as my_lib
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    for ci, case in enumerate(C3_FLOWS):
        xn, fn, kn, gn = _c3_inputs(case, 40 + ci, signed=True)
        w1, w2, w3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
        h2, h3 = torch.full(fn.shape, 3.0, device=dev()), torch.full(kn.shape, 3.0, device=dev())
        rc = my_lib.FilterInterpolationLayer_gpu_backward(T(xn), T(fn), T(kn), T(gn), None, h2, h3)
        if xn.shape[3] % 4:
            assert rc == -1 and float(h2.min()) == 3.0 and float(h3.min()) == 3.0       # not served, nothing written
            continue
        assert rc == 0 and my_lib.last_kernel_path() == "fi_bwd:tiled_c3"
        close(N(h2), w2, "gradinput2 without gradinput1 %s" % (case,), 3 * RTOL)
        close(N(h3), w3, "gradinput3 without gradinput1 %s" % (case,), RTOL)
        # the module: frames without requires_grad
        x, f, k = T(xn), T(fn, True), T(kn, True)
        FilterInterpolationModule()(x, f, k).backward(T(gn))
        assert x.grad is None
        close(N(f.grad), w2, "module gradinput2, frames are data %s" % (case,), 3 * RTOL)
        close(N(k.grad), w3, "module gradinput3, frames are data %s" % (case,), RTOL)
    # more than three channels: not served
    xn, fn, kn, gn = synth.np_image(np.random.default_rng(1), 1, 8, 16, 64), np.zeros((1, 2, 16, 64), np.float32), \
        synth.np_filter(np.random.default_rng(2), 1, 16, 64), synth.np_image(np.random.default_rng(3), 1, 8, 16, 64)
    assert my_lib.FilterInterpolationLayer_gpu_backward(T(xn), T(fn), T(kn), T(gn), None, T(fn), T(kn)) == -1


def _heavy_tail_variants(rng, kn, gn):
    """(name, taps, gradoutput): a few values four to five orders of magnitude above their O(0.05) neighbours."""
    out = []
    k1 = kn.copy(); k1[0, 5, 20, 70] = 1.0e4                       # one tap of one site
    out.append(("one tap of 1e4", k1, gn))
    g1 = gn.copy(); g1[0, 1, 21, 33] = 1.0e4; g1[0, 2, 40, 100] = -3.0e3
    out.append(("gradoutput of 1e4 and -3e3 at two sites", kn, g1))
    g2 = gn.copy()                                                  # a heavy tail: 3 % of the sites a hundred times larger
    m = rng.random(g2.shape[0:1] + g2.shape[2:]) < 0.03
    g2 *= np.where(m, 100.0, 1.0).astype(np.float32)[:, None]
    out.append(("3 % of the sites with 100 x the gradient", kn, g2))
    g3 = gn.copy(); g3[:, :, :, 64:] *= np.float32(2.0 ** 12)      # half of a tile row 4096 x the other half
    out.append(("a step of 2^12 across the tiles", kn, g3))
    return out


def test_rgb_backward_packed_planes_heavy_tailed_tile(oracle):
    """The packed fixed-point planes of the RGB backward passes round every contribution to the TILE's grid (memc_pk.hpp).
    The reference's fp32 atomics (my_lib_kernel.cu:1276-1288) have no coupling between sites: a tap or a gradient of 1e4
    beside O(0.05) values must not cost the small cells their 1e-4.  Round 4: per-site bounds, block exponent from
    min(max, 16 x mean), sites beyond it by per-site atomics -- gradinput1 against the oracle under the ordinary rule
    (1e-4 absolute up to |want| = 10, 1e-5 relative beyond)."""
    import my_package._ext.my_lib as my_lib
    rng = np.random.default_rng(77)
    B, H, W = 1, 48, 128
    xn, fn = synth.np_image(rng, B, 3, H, W), synth.np_flow(rng, B, H, W, "smooth", 2.0)
    kn = (rng.random((B, 16, H, W)) * 0.05).astype(np.float32)
    gn = (rng.standard_normal((B, 3, H, W)) * 0.05).astype(np.float32)
    for name, kk, gg in _heavy_tail_variants(rng, kn, gn):
        w1, w2, w3 = oracle.filter_interpolation_backward(xn, fn, kk, gg)
        h1, h2, h3 = _c3_backward(my_lib, xn, fn, kk, gg)
        v1, v2 = oracle.interpolation_backward(xn, fn, gg)
        i1, i2 = torch.zeros(xn.shape, device=dev()), torch.full(fn.shape, 3.0, device=dev())
        assert my_lib.InterpolationLayer_gpu_backward(T(xn), T(fn), T(gg), i1, i2) == 0
        if name.startswith("a step"):
            # No outlier here: every site of the right-hand tiles carries a gradient of ~200 (up to 800).  The planes'
            # contract (include/memc_warp.h) is relative to the TILE: contributions are rounded to 2^-22 of its packed bound,
            # so small cells next to such a tile see an absolute error of up to (contributions) x 2^-23 x that bound --
            # 1e-4 absolute only holds while the tile's bound stays below ~64.  Checked against the contract.
            for got, want, bound, what in ((h1, w1, float(np.abs(gg).max() * np.abs(kk).max()), "FilterInterpolation"),
                                           (N(i1), v1, float(np.abs(gg).max()), "Interpolation")):
                err = float(np.abs(got.astype(np.float64) - want).max())
                assert err <= 64 * 2.0 ** -22 * bound + 1e-5 * float(np.abs(want).max()), (what, name, err, bound)
        else:
            close(h1, w1, "FilterInterpolation gradinput1, " + name, RTOL)
            close(N(i1), v1, "Interpolation gradinput1, " + name, RTOL)
            for got, want, what in ((h1, w1, "FilterInterpolation"), (N(i1), v1, "Interpolation")):
                small = np.abs(want) <= 1.0                             # the cells the outliers do not reach
                assert small.mean() > 0.3
                err_small = float(np.abs(got[small].astype(np.float64) - want[small]).max())
                assert err_small <= 2e-5, "%s gradinput1, %s: small cells off by %.3g" % (what, name, err_small)
        close(h2, w2, "FilterInterpolation gradinput2, " + name, 3 * RTOL)
        close(h3, w3, "FilterInterpolation gradinput3, " + name, RTOL)
        close(N(i2), v2, "Interpolation gradinput2, " + name, 3 * RTOL)


def test_interpolation_backward_rgb_packed_planes_special_values(oracle):
    """The RGB bilinear-warp backward (interpolation.hip: bl_bwd_c3_pk, packed fixed-point planes): signed and huge /
    tiny gradients against the oracle, a zero gradient, NaN / Inf in gradoutput landing where the reference puts them,
    large motion (sites outside the staged box scatter with global atomics)."""
    import my_package._ext.my_lib as my_lib
    for ci, case in enumerate(((2, 80, 192, "smooth", 6.0), (1, 64, 256, "iid", 20.0), (1, 37, 52, "smooth", 3.0))):
        B, H, W, kind, sigma = case
        rng = np.random.default_rng(500 + ci)
        xn, fn = synth.np_image(rng, B, 3, H, W), synth.np_flow(rng, B, H, W, kind, sigma)
        gn = rng.standard_normal((B, 3, H, W)).astype(np.float32)

        def run(g, fill=0.0):
            h1, h2 = torch.full(xn.shape, fill, device=dev()), torch.full(fn.shape, 3.0, device=dev())
            assert my_lib.InterpolationLayer_gpu_backward(T(xn), T(fn), T(g), h1, h2) == 0
            return N(h1), N(h2)
        w1, w2 = oracle.interpolation_backward(xn, fn, gn)
        h1, h2 = run(gn, 0.5)
        close(h1, w1 + 0.5, "gradinput1 += %s" % (case,), RTOL)
        close(h2, w2, "gradinput2 %s" % (case,), RTOL)
        for scale in (3.7e12, 1.3e-17):
            g = gn * np.float32(scale)
            want = oracle.interpolation_backward(xn, fn, g)[0].astype(np.float64)
            got = run(g)[0].astype(np.float64)
            assert float(np.abs(got - want).max()) <= 1e-5 * float(np.abs(want).max()), (case, scale)
        z1 = run(np.zeros_like(gn), 0.25)[0]
        assert np.array_equal(z1, np.full_like(z1, 0.25)), "zero gradoutput must add nothing"
        if W % 4 == 0:
            g_bad = gn.copy()
            g_bad[0, 1, 10, 20] = np.nan
            g_bad[0, 2, 30, 40] = -np.inf
            w1 = oracle.interpolation_backward(xn, fn, g_bad)[0]
            h1 = run(g_bad)[0]
            assert np.array_equal(np.isnan(h1), np.isnan(w1)) and np.array_equal(np.isinf(h1), np.isinf(w1)), case
            close(h1, w1, "gradinput1 with NaN / Inf in gradoutput %s" % (case,), RTOL)


def _many_channel_flows(kind, rng, B, H, W):
    """Flow fields that take fi_bwd_cn.hip's paths one by one (FilterInterpolation backward, C % 4 == 0)."""
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    if kind == "smooth":                       # the common case: a handful of candidate site tiles per cell tile
        return synth.np_flow(rng, B, H, W, "smooth", 6.0)
    if kind == "pan":                          # coherent motion of 100 px: the target boxes sit two tiles away
        f = synth.np_flow(rng, B, H, W, "smooth", 1.5)
        f[:, 0] += -min(100.25, 0.4 * W)
        f[:, 1] += min(9.5, 0.3 * H)
        return f
    if kind == "far":                          # beyond the owners' search window (> 3 tile columns): global atomics
        f = synth.np_flow(rng, B, H, W, "smooth", 2.0)
        f[:, 0, :, W // 2:] -= 0.47 * W                    # (still valid: |flow| < W / 2)
        return f
    if kind == "converge":                     # every site lands near one point: lists far beyond the LDS budget
        f = np.empty((B, 2, H, W), np.float32)
        f[:, 0] = (W / 2 - xx) * 0.97 + 0.3
        f[:, 1] = (H / 2 - yy) * 0.97 + 0.3
        return f
    if kind == "iid":
        return synth.np_flow(rng, B, H, W, "iid", 20.0)
    raise ValueError(kind)


MANY = [(2, 8, 70, 200, "smooth"), (1, 16, 96, 384, "pan"), (1, 8, 48, 512, "far"), (1, 8, 64, 192, "converge"),
        (1, 12, 80, 256, "iid"), (1, 64, 36, 132, "smooth"),
        (1, 8, 20, 50, "smooth"),          # width not a multiple of 4 (round 6: whole quads on the owner kernels, the tail by lanes)
        (1, 8, 5, 12, "smooth"), (1, 8, 16, 64, "iid"), (3, 8, 33, 68, "pan"),      # tiny, exactly one tile, ragged
        (1, 4, 40, 128, "smooth"), (2, 5, 40, 132, "smooth"), (1, 6, 24, 64, "iid"), (1, 7, 70, 200, "converge"),
        # (the last four: one chunk; ragged last chunks of 1, 2 and 3 channels)
        # round 6, ragged WIDTHS (W % 4 = 1, 2, 3) through every path of fi_bwd_cn.hip: near sites next to the tail columns,
        # coherent motion into / out of them, far sites, converging lists, i.i.d. flow, 64 channels, a ragged last chunk too
        (1, 8, 40, 134, "smooth"), (2, 16, 70, 201, "pan"), (1, 8, 48, 510, "far"), (1, 8, 64, 190, "converge"),
        (1, 12, 80, 254, "iid"), (1, 64, 36, 131, "smooth"), (1, 5, 40, 133, "smooth"), (1, 8, 17, 9, "smooth")]


@pytest.mark.parametrize("case", MANY, ids=["%dx%dx%dx%d-%s" % c for c in MANY])
def test_filter_interpolation_backward_many_channels(oracle, case):
    """C % 4 == 0, C >= 8: tap-gradient kernel + owner-computes image gradient (fi_bwd_cn.hip)."""
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    B, C, H, W, kind = case
    rng = np.random.default_rng(sum(case[:4]))
    xn, kn, gn = synth.np_image(rng, B, C, H, W), synth.np_filter(rng, B, H, W), synth.np_image(rng, B, C, H, W)
    fn = _many_channel_flows(kind, rng, B, H, W)
    x, f, k = T(xn, True), T(fn, True), T(kn, True)
    out = FilterInterpolationModule()(x, f, k)
    out.backward(T(gn))
    g1, g2, g3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
    close(N(x.grad), g1, "gradinput1", 3 * RTOL)      # a cell can collect hundreds of taps ("converge")
    close(N(f.grad), g2, "gradinput2", RTOL)
    close(N(k.grad), g3, "gradinput3", RTOL)
    # The owner kernels STORE gradinput1 (every cell has exactly one writer; unreached cells get zeros; far sites are
    # added afterwards): outside a stream capture the result does not depend on what the buffers held before.
    import my_package._ext.my_lib as my_lib
    h1, h2, h3 = torch.full_like(x, 7.0), torch.full_like(f, 7.0), torch.full_like(k, 7.0)
    assert my_lib.FilterInterpolationLayer_gpu_backward(x.detach(), f.detach(), k.detach(), T(gn), h1, h2, h3) == 0
    close(N(h1), g1, "gradinput1 (stored)", 3 * RTOL)
    close(N(h2), g2, "gradinput2 (stored)", RTOL)
    close(N(h3), g3, "gradinput3 (stored)", RTOL)


@pytest.mark.parametrize("case", MANY, ids=["%dx%dx%dx%d-%s" % c for c in MANY])
def test_interpolation_ch_backward_many_channels(oracle, case):
    """The bilinear warp's backward for C % 4 == 0, C >= 8: flow-gradient kernel + the owner kernel with a 2 x 2 window
    (fi_bwd_cn.hip).  Through the module, then through the C ABI from garbage-filled buffers (gradinput1 is stored)."""
    from my_package.modules.InterpolationChModule import InterpolationChModule
    import my_package._ext.my_lib as my_lib
    B, C, H, W, kind = case
    rng = np.random.default_rng(sum(case[:4]) + 1)
    xn, gn = synth.np_image(rng, B, C, H, W), synth.np_image(rng, B, C, H, W)
    fn = _many_channel_flows(kind, rng, B, H, W)
    x, f = T(xn, True), T(fn, True)
    out = InterpolationChModule()(x, f)
    out.backward(T(gn))
    close(N(out), oracle.interpolation_ch_forward(xn, fn), "forward")
    g1, g2 = oracle.interpolation_ch_backward(xn, fn, gn)
    close(N(x.grad), g1, "gradinput1", 3 * RTOL)
    close(N(f.grad), g2, "gradinput2", 3 * RTOL)
    h1, h2 = torch.full_like(x, 7.0), torch.full_like(f, 7.0)
    assert my_lib.InterpolationChLayer_gpu_backward(x.detach(), f.detach(), T(gn), h1, h2) == 0
    close(N(h1), g1, "gradinput1 (stored)", 3 * RTOL)
    close(N(h2), g2, "gradinput2 (stored)", 3 * RTOL)
    from tools import measure as M          # the direct kernel it replaced: same answers from garbage-filled buffers
    ml = M.bound()
    try:
        M.set_variant("bl_bwd_direct", 1)
        h1.fill_(5.0); h2.fill_(5.0)
        assert ml.InterpolationChLayer_gpu_backward(x.detach(), f.detach(), T(gn), h1, h2) == 0
    finally:
        M.reset()
    close(N(h1), g1, "gradinput1 (direct arm)", 3 * RTOL)
    close(N(h2), g2, "gradinput2 (direct arm)", 3 * RTOL)


@pytest.mark.parametrize("arm", [("fi_bwd", 40)], ids=["direct kernel"])
def test_filter_interpolation_backward_many_channels_measurement_arms(oracle, arm):
    """The A/B arm of the many-channel backward (measurement build): the direct global-atomics kernel the owner kernels
    replaced -- it must give the oracle's gradients (from garbage-filled buffers:
    gradinput1 is stored on every path)."""
    from tools import measure as M          # forced paths exist in the measurement build only
    my_lib = M.bound()
    for case in (MANY[0], MANY[3]):
        B, C, H, W, kind = case
        rng = np.random.default_rng(sum(case[:4]))
        xn, kn, gn = synth.np_image(rng, B, C, H, W), synth.np_filter(rng, B, H, W), synth.np_image(rng, B, C, H, W)
        fn = _many_channel_flows(kind, rng, B, H, W)
        g1, g2, g3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
        h1, h2, h3 = (torch.full(s, 3.0, device=dev()) for s in (xn.shape, fn.shape, kn.shape))
        try:
            M.set_variant(*arm)
            assert my_lib.FilterInterpolationLayer_gpu_backward(T(xn), T(fn), T(kn), T(gn), h1, h2, h3) == 0
        finally:
            M.reset()
        close(N(h1), g1, "gradinput1 %s" % (arm,), 3 * RTOL)
        close(N(h2), g2, "gradinput2 %s" % (arm,), RTOL)
        close(N(h3), g3, "gradinput3 %s" % (arm,), RTOL)


@pytest.mark.parametrize("variant", [33, 34, 31, 32, 35],
                         ids=["32x32 strips", "32x32 stripes of 4", "64x16 stripes of 4", "64x32 on 512 lanes",
                              "64x32 on 512 lanes, stripes of 4"])
def test_context_warp_forward_tile_shape_arms(oracle, variant):
    """The many-channel forward on 32 x 32 tiles (eight lanes per tile row) and the stripe walks -- measurement arms of
    fi_fwd_tiled_c4n -- must give the oracle's results like the 64 x 16 product kernel."""
    from tools import measure as M
    my_lib = M.bound()
    try:
        M.set_variant("fi_fwd", variant)
        for ci, (B, C, H, W, kind, sigma) in enumerate(((1, 8, 70, 200, "smooth", 6.0), (2, 16, 96, 132, "smooth", 14.0),
                                                        (1, 64, 40, 256, "iid", 5.0), (1, 12, 33, 64, "smooth", 3.0),
                                                        (2, 6, 45, 72, "smooth", 4.0), (1, 7, 18, 300, "iid", 2.0))):
            rng = np.random.default_rng(700 + ci)
            xn, fn, kn = synth.np_image(rng, B, C, H, W), synth.np_flow(rng, B, H, W, kind, sigma), synth.np_filter(rng, B, H, W)
            out = torch.full(xn.shape, float("nan"), device=dev())
            assert my_lib.FilterInterpolationLayer_gpu_forward(T(xn), T(fn), T(kn), out) == 0
            close(N(out), oracle.filter_interpolation_forward(xn, fn, kn), "forward %s variant %d" % ((B, C, H, W, kind), variant))
    finally:
        M.reset()


def test_shapes_take_the_documented_kernel_paths():
    """DESIGN.md section 5 says which kernel family a shape takes; the library records the launcher's choice per host
    thread and hands it out through its C ABI (memc_last_kernel_path, include/memc_warp.h: round 4 -- rounds 2-3 could ask
    the measurement build only), so the claim is checked ON THE SHIPPED BINARY instead of inferred from timings: aligned
    shapes must not silently drop to the scalar kernels, odd ones must not reach the vector kernels."""
    import my_package._ext.my_lib as my_lib

    def last():
        return my_lib.last_kernel_path().encode()
    rng = np.random.default_rng(5)

    def run(B, C, H, W, fs=4, sliced=False):
        Wa = W + 1 if sliced else W                     # sliced: views whose rows start at unaligned addresses

        def view(t):
            return t[..., 1:] if sliced else t

        def buf(ch):
            return view(torch.zeros(B, ch, H, Wa, device=dev()))
        x, g = view(T(synth.np_image(rng, B, C, H, Wa))), view(T(synth.np_image(rng, B, C, H, Wa)))
        f, k = view(T(synth.np_flow(rng, B, H, Wa, "smooth", 2.0))), view(T(synth.np_filter(rng, B, H, Wa, fs)))
        got = {}
        out, g1, g2, g3 = buf(C), buf(C), buf(2), buf(fs * fs)
        assert my_lib.FilterInterpolationLayer_gpu_forward(x, f, k, out) == 0
        got["fi_fwd"] = last().decode()
        assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
        got["fi_bwd"] = last().decode()
        if fs == 4:
            assert my_lib.InterpolationChLayer_gpu_forward(x, f, out) == 0
            got["bl_fwd"] = last().decode()
            assert my_lib.InterpolationChLayer_gpu_backward(x, f, g, g1, g2) == 0
            got["bl_bwd"] = last().decode()
            cnt, po, gf = buf(1), buf(2), view(T(synth.np_flow(rng, B, H, Wa, "smooth", 1.0)))
            assert my_lib.FlowProjectionLayer_gpu_forward(f, cnt, po, 1) == 0
            got["proj_fwd"] = last().decode()
            assert my_lib.FlowProjectionLayer_gpu_backward(f, cnt, gf, g2) == 0
            got["proj_bwd"] = last().decode()
        return got

    assert run(2, 3, 40, 128) == {"fi_fwd": "fi_fwd:tiled_c3", "fi_bwd": "fi_bwd:tiled_c3", "bl_fwd": "bl_fwd:tiled_c3",
                                  "bl_bwd": "bl_bwd:tiled_c3", "proj_fwd": "proj_fwd:owner", "proj_bwd": "proj_bwd:tiled"}
    assert run(1, 64, 40, 128) == {"fi_fwd": "fi_fwd:tiled_c4n", "fi_bwd": "fi_bwd:owner", "bl_fwd": "bl_fwd:tiled_chunks",
                                   "bl_bwd": "bl_bwd:owner", "proj_fwd": "proj_fwd:owner", "proj_bwd": "proj_bwd:tiled"}
    assert run(1, 5, 40, 128) == {"fi_fwd": "fi_fwd:tiled_c4n_ragged", "fi_bwd": "fi_bwd:owner", "bl_fwd": "bl_fwd:tiled_chunks",
                                  "bl_bwd": "bl_bwd:owner", "proj_fwd": "proj_fwd:owner", "proj_bwd": "proj_bwd:tiled"}
    # a width that is not a multiple of four stays on the tiled kernels since round 5 (the scattering passes were 13-41x
    # slower on the scalar ones in round 4, the gathers 2x): the projection forward's owner kernels have a ragged-row
    # instantiation, every other operator takes the whole quads in its tiled kernel and the one to three columns behind
    # them in the one-lane-per-site kernel (the many-channel backward passes since round 6)
    assert run(1, 3, 20, 50) == {"fi_fwd": "fi_fwd:tiled_c3", "fi_bwd": "fi_bwd:tiled_c3", "bl_fwd": "bl_fwd:tiled_c3",
                                 "bl_bwd": "bl_bwd:tiled_c3", "proj_fwd": "proj_fwd:owner", "proj_bwd": "proj_bwd:tiled"}
    got = run(1, 8, 20, 50)                                            # (round 6: the many-channel backward passes too)
    assert got["fi_fwd"] == "fi_fwd:tiled_c4n_ragged" and got["fi_bwd"] == "fi_bwd:owner" and got["bl_bwd"] == "bl_bwd:owner"
    assert run(1, 8, 20, 7)["fi_bwd"] == "fi_bwd:direct"               # (fewer than two whole quads per row: the direct kernel)
    assert run(1, 3, 20, 6)["proj_fwd"] == "proj_fwd:scalar"           # (narrower than two quads: scalar)
    assert run(1, 3, 20, 3) == {"fi_fwd": "fi_fwd:direct", "fi_bwd": "fi_bwd:direct", "bl_fwd": "bl_fwd:direct",
                                "bl_bwd": "bl_bwd:direct", "proj_fwd": "proj_fwd:scalar", "proj_bwd": "proj_bwd:scalar"}
    # ... but a multiple of four seen through a view that starts one element in (rows of 65 elements: every row at another
    # alignment) takes the tiled kernels since round 5: a quad in global memory needs dword alignment only (memc_tile.hpp)
    assert run(1, 3, 20, 64, sliced=True) == {"fi_fwd": "fi_fwd:tiled_c3", "fi_bwd": "fi_bwd:tiled_c3", "bl_fwd": "bl_fwd:tiled_c3",
                                             "bl_bwd": "bl_bwd:tiled_c3", "proj_fwd": "proj_fwd:owner", "proj_bwd": "proj_bwd:tiled"}
    assert run(1, 8, 20, 64, sliced=True)["fi_bwd"] == "fi_bwd:owner"
    assert run(1, 3, 24, 64, fs=2) == {"fi_fwd": "fi_fwd:generic", "fi_bwd": "fi_bwd:generic"}


@pytest.mark.parametrize("off", [1, 2, 3])
@pytest.mark.parametrize("C", [3, 8])
def test_unaligned_views_on_the_tiled_kernels(oracle, off, C):
    """Round 5: the tiled kernels serve any view whose WIDTH is a multiple of four -- base pointers and row strides at any
    dword boundary (a quad in global memory is loaded / stored with dword alignment, memc_tile.hpp: f32x4u).  Every operator,
    forward and backward, on views that start `off` elements into rows of W + off elements (so every row sits at another
    alignment), against the oracle; the reference serves every view with its one kernel per operator
    (my_lib_kernel.cu:10-15).  Until round 5 these views took the scalar kernels: 13-41x slower for the scattering passes."""
    import my_package._ext.my_lib as my_lib
    rng = np.random.default_rng(100 * C + off)
    B, H, W = 2, 70, 132

    def view(a, fill=None):                                            # numpy [B, ch, H, W] -> a view into a wider tensor
        t = torch.full((a.shape[0], a.shape[1], H, W + off), 123.0 if fill is None else fill, device=dev())
        v = t[..., off:]
        if a is not None:
            v.copy_(T(a))
        assert v.data_ptr() % 16 != 0 or (W + off) % 4 != 0
        return v

    def buf(ch, fill=0.0):
        return view(np.zeros((B, ch, H, W), np.float32), fill)
    xn, gn = synth.np_image(rng, B, C, H, W), synth.np_image(rng, B, C, H, W)
    fn, kn = synth.np_flow(rng, B, H, W, "smooth", 5.0), synth.np_filter(rng, B, H, W)
    dn = synth.np_depth(rng, B, H, W)
    g2n = synth.np_flow(rng, B, H, W, "smooth", 1.0)
    x, g, f, k, d, gf = view(xn), view(gn), view(fn), view(kn), view(dn), view(g2n)
    # FilterInterpolation
    out = buf(C, 7.0)
    assert my_lib.FilterInterpolationLayer_gpu_forward(x, f, k, out) == 0
    assert my_lib.last_kernel_path() in ("fi_fwd:tiled_c3", "fi_fwd:tiled_c4n")
    close(N(out), oracle.filter_interpolation_forward(xn, fn, kn), "FI fwd, view + %d" % off)
    g1, g2, g3 = buf(C), buf(2, 7.0), buf(16, 7.0)
    assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
    assert my_lib.last_kernel_path() in ("fi_bwd:tiled_c3", "fi_bwd:owner")
    w1, w2, w3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
    close(N(g1), w1, "FI gradinput1, view + %d" % off, RTOL)
    close(N(g2), w2, "FI gradinput2, view + %d" % off, RTOL)
    close(N(g3), w3, "FI gradinput3, view + %d" % off, RTOL)
    # Interpolation(Ch)
    out = buf(C, 7.0)
    assert my_lib.InterpolationChLayer_gpu_forward(x, f, out) == 0
    assert "tiled" in my_lib.last_kernel_path()
    close(N(out), oracle.interpolation_ch_forward(xn, fn), "Interpolation fwd, view + %d" % off)
    g1, g2 = buf(C), buf(2, 7.0)
    assert my_lib.InterpolationChLayer_gpu_backward(x, f, g, g1, g2) == 0
    assert my_lib.last_kernel_path() in ("bl_bwd:tiled_c3", "bl_bwd:owner")
    w1, w2 = oracle.interpolation_ch_backward(xn, fn, gn)
    close(N(g1), w1, "Interpolation gradinput1, view + %d" % off, RTOL)
    close(N(g2), w2, "Interpolation gradinput2, view + %d" % off, RTOL)
    if C != 3:
        return
    # (Depth)FlowProjection, forward with and without hole filling, backward
    for fill in (0, 1):
        cnt, po = buf(1, 7.0), buf(2, 7.0)
        assert my_lib.FlowProjectionLayer_gpu_forward(f, cnt, po, fill) == 0
        assert my_lib.last_kernel_path() == "proj_fwd:owner"
        want_o, want_c = oracle.flow_projection_forward(fn, fill)
        assert np.array_equal(N(cnt), want_c)
        close(N(po), want_o, "FlowProjection fwd fill %d, view + %d" % (fill, off))
        cnt, po = buf(1, 7.0), buf(2, 7.0)
        assert my_lib.DepthFlowProjectionLayer_gpu_forward(f, d, cnt, po, fill) == 0
        assert my_lib.last_kernel_path() == "dproj_fwd:owner"
        want_o, want_c = oracle.depth_flow_projection_forward(fn, dn, fill)
        close(N(cnt), want_c, "DepthFlowProjection count fill %d, view + %d" % (fill, off), RTOL)
        close(N(po), want_o, "DepthFlowProjection fwd fill %d, view + %d" % (fill, off), RTOL)
    gin = buf(2, 7.0)
    assert my_lib.FlowProjectionLayer_gpu_backward(f, view(oracle.flow_projection_forward(fn, 0)[1]), gf, gin) == 0
    assert my_lib.last_kernel_path() == "proj_bwd:tiled"
    close(N(gin), oracle.flow_projection_backward(fn, oracle.flow_projection_forward(fn, 0)[1], g2n), "FlowProjection bwd, view + %d" % off, RTOL)
    wo, wc = oracle.depth_flow_projection_forward(fn, dn, 0)
    gin, gd = buf(2, 7.0), buf(1, 7.0)
    assert my_lib.DepthFlowProjectionLayer_gpu_backward(f, d, view(wc), view(wo), gf, gin, gd) == 0
    w1, w2 = oracle.depth_flow_projection_backward(fn, dn, wc, wo, g2n)
    close(N(gin), w1, "DepthFlowProjection gradinput1, view + %d" % off, RTOL)
    close(N(gd), w2, "DepthFlowProjection gradinput2, view + %d" % off, RTOL)


@pytest.mark.parametrize("fs", [2, 3, 6])
def test_filter_interpolation_other_filter_sizes(oracle, fs):
    """fs = (int)sqrt(channels of input3) (my_lib.c:925); 3 is odd: window [ix, ix+2]."""
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    rng = np.random.default_rng(20 + fs)
    B, C, H, W = 2, 3, 21, 34
    xn, fn = synth.np_image(rng, B, C, H, W), synth.np_flow(rng, B, H, W, "iid", 3.0)
    kn, gn = synth.np_filter(rng, B, H, W, fs), synth.np_image(rng, B, C, H, W)
    x, f, k = T(xn, True), T(fn, True), T(kn, True)
    out = FilterInterpolationModule()(x, f, k)
    out.backward(T(gn))
    close(N(out), oracle.filter_interpolation_forward(xn, fn, kn), "forward fs=%d" % fs)
    g1, g2, g3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
    close(N(x.grad), g1, "gradinput1", RTOL)
    close(N(f.grad), g2, "gradinput2", RTOL)
    close(N(k.grad), g3, "gradinput3", RTOL)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_interpolation_ch(oracle, case):
    from my_package.modules.InterpolationChModule import InterpolationChModule
    d = make(case)
    x, f = T(d["x"], True), T(d["flow"], True)
    out = InterpolationChModule()(x, f)
    out.backward(T(d["gout"]))
    close(N(out), oracle.interpolation_ch_forward(d["x"], d["flow"]), "forward")
    g1, g2 = oracle.interpolation_ch_backward(d["x"], d["flow"], d["gout"])
    close(N(x.grad), g1, "gradinput1", RTOL)
    close(N(f.grad), g2, "gradinput2", RTOL)


def test_interpolation_three_channels_only(oracle):
    from my_package.modules.InterpolationModule import InterpolationModule
    d = make(CASES[1])
    x, f = T(d["x"], True), T(d["flow"], True)
    out = InterpolationModule()(x, f)
    out.backward(T(d["gout"]))
    close(N(out), oracle.interpolation_forward(d["x"], d["flow"]), "forward")
    g1, g2 = oracle.interpolation_backward(d["x"], d["flow"], d["gout"])
    close(N(x.grad), g1, "gradinput1", RTOL)
    close(N(f.grad), g2, "gradinput2", RTOL)
    with pytest.raises(RuntimeError):          # channel != 3 -> -1 (my_lib_cuda.c:373); we raise
        InterpolationModule()(T(make(CASES[4])["x"]), T(make(CASES[4])["flow"]))


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_flow_projection(oracle, case):
    from my_package.modules.FlowProjectionModule import FlowProjectionModule
    import my_package._ext.my_lib as my_lib
    d = make(case)
    # inference: requires_grad False -> fillhole 1 (reference FlowProjectionLayer.py:15)
    with torch.no_grad():
        out1 = FlowProjectionModule(requires_grad=False)(T(d["flow"]))
    want1, want_count = oracle.flow_projection_forward(d["flow"], 1)
    close(N(out1), want1, "forward fillhole=1")
    # training: fillhole 0, then backward
    f = T(d["flow"], True)
    out0 = FlowProjectionModule(requires_grad=True)(f)
    out0.backward(T(d["gflow"]))
    want0, _ = oracle.flow_projection_forward(d["flow"], 0)
    close(N(out0), want0, "forward fillhole=0")
    close(N(f.grad), oracle.flow_projection_backward(d["flow"], want_count, d["gflow"]), "gradinput1", RTOL)
    # count is integer-valued: bit-exact
    flow = T(d["flow"])
    count = flow.new_zeros((flow.size(0), 1, flow.size(2), flow.size(3)))
    out = torch.zeros_like(flow)
    assert my_lib.FlowProjectionLayer_gpu_forward(flow, count, out, 0) == 0
    assert np.array_equal(N(count), want_count)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_depth_flow_projection(oracle, case):
    from my_package.modules.DepthFlowProjectionModule import DepthFlowProjectionModule
    d = make(case)
    with torch.no_grad():
        out1 = DepthFlowProjectionModule(requires_grad=False)(T(d["flow"]), T(d["depth"]))
    want1, _ = oracle.depth_flow_projection_forward(d["flow"], d["depth"], 1)
    close(N(out1), want1, "forward fillhole=1", RTOL)
    f, dp = T(d["flow"], True), T(d["depth"], True)
    out0 = DepthFlowProjectionModule(requires_grad=True)(f, dp)
    out0.backward(T(d["gflow"]))
    want0, wcount = oracle.depth_flow_projection_forward(d["flow"], d["depth"], 0)
    close(N(out0), want0, "forward fillhole=0", RTOL)
    g1, g2 = oracle.depth_flow_projection_backward(d["flow"], d["depth"], wcount, want0, d["gflow"])
    # 1/count amplifies the rounding of tiny depth sums: relative bound
    close(N(f.grad), g1, "gradinput1", 1e-4)
    close(N(dp.grad), g2, "gradinput2", 1e-4)


def test_projection_forward_unusual_depths_and_flows(oracle):
    """What the mask-based hole filling (proj_fill.hpp) must get right beyond the ordinary case, against the oracle (which
    restates my_lib_kernel.cu:2053-2264 line by line): cells whose depth sum is NEGATIVE are holes (count <= 0, :2184) that
    nevertheless stop a walk (count != 0, :2205-2224) and contribute nothing (flag = count > 0, :2238); zero depths add nothing;
    a NaN / Inf flow makes its site invalid (every compare fails); NaN depths poison their four cells.  Several tile shapes:
    one tile, tiles cut by the image edge, many tiles."""
    import my_package._ext.my_lib as my_lib
    for ci, (B, H, W) in enumerate(((1, 20, 64), (2, 45, 132), (1, 100, 260))):
        rng = np.random.default_rng(900 + ci)
        flow = synth.np_flow(rng, B, H, W, "smooth", 5.0)
        depth = (rng.random((B, 1, H, W)) + 0.1).astype(np.float32)
        depth[rng.random(depth.shape) < 0.15] *= -1.0                 # negative depths: negative and near-zero count cells
        depth[rng.random(depth.shape) < 0.10] = 0.0
        flow[0, :, 3:9, 10:30] = 200.0                                # an uncovered patch (its sources leave the image)
        flow[0, 0, 12, 40] = np.nan
        flow[-1, 1, 15, 20] = np.inf
        for fill in (0, 1):
            want_out, want_cnt = oracle.depth_flow_projection_forward(flow, depth, fill)
            cnt, out = torch.full((B, 1, H, W), 7.0, device=dev()), torch.full((B, 2, H, W), 7.0, device=dev())
            assert my_lib.DepthFlowProjectionLayer_gpu_forward(T(flow), T(depth), cnt, out, fill) == 0
            close(N(cnt), want_cnt, "count, negative / zero depths, fill %d, case %d" % (fill, ci), RTOL)
            # where the depth sum cancels to within rounding the two disagree on which side of zero it fell: compare the
            # cells whose count is clearly non-zero or exactly zero in both
            clear = (np.abs(want_cnt) > 1e-4) | ((want_cnt == 0) & (N(cnt) == 0))
            assert clear.mean() > 0.9
            sel = np.broadcast_to(clear, want_out.shape)
            if fill == 0:
                close(N(out)[sel], want_out[sel], "output, negative / zero depths, fill 0, case %d" % ci, 1e-4)
            else:                                                      # a filled hole next to an ambiguous cell inherits the ambiguity
                bad = np.abs(N(out) - want_out)[sel] > 1e-4 + 1e-4 * np.abs(want_out[sel])
                assert bad.mean() < 0.02, "fill 1, case %d: %.2f %% of the clear cells differ" % (ci, 100 * bad.mean())
        d2 = np.abs(depth) + 0.1
        d2[0, 0, 5, 7] = np.nan
        want_out, want_cnt = oracle.depth_flow_projection_forward(flow, d2, 1)
        cnt, out = torch.zeros((B, 1, H, W), device=dev()), torch.zeros((B, 2, H, W), device=dev())
        assert my_lib.DepthFlowProjectionLayer_gpu_forward(T(flow), T(d2), cnt, out, 1) == 0
        assert np.array_equal(np.isnan(N(cnt)), np.isnan(want_cnt)) and 0 < np.isnan(want_cnt).sum() <= 4
        ok = ~np.isnan(want_out)
        close(N(out)[ok], want_out[ok], "output beside a NaN depth, case %d" % ci, RTOL)
        # the plain operator with the same non-finite flows
        want_out, want_cnt = oracle.flow_projection_forward(flow, 1)
        cnt, out = torch.zeros((B, 1, H, W), device=dev()), torch.zeros((B, 2, H, W), device=dev())
        assert my_lib.FlowProjectionLayer_gpu_forward(T(flow), cnt, out, 1) == 0
        assert np.array_equal(N(cnt), want_cnt)
        close(N(out), want_out, "FlowProjection with NaN / Inf flow and an uncovered patch, case %d" % ci)


def test_projection_with_a_few_far_sources(oracle):
    """Sources that move 24 px or more are not seen by the owner kernel's scan unless they lie within 24 px of the tile they
    land in: their image is flagged and proj_owner_far redoes the tiles such a source can reach -- and ONLY those (the other
    tiles keep what the owner kernel wrote, summaries and hole masks included).  A few far sources in an otherwise gentle
    image: single sites thrown across many tiles, a patch moving 40 px, far sources landing in what would otherwise be a
    hole, one image with none (never flagged).  Counts bit-exact, outputs to 1e-4, both operators, with and without fill."""
    import my_package._ext.my_lib as my_lib
    B, H, W = 3, 170, 530                                              # 6 x 9 tiles of 64 x 32, cut by both edges
    rng = np.random.default_rng(4711)
    flow = synth.np_flow(rng, B, H, W, "smooth", 4.0)
    depth = (rng.random((B, 1, H, W)) + 0.1).astype(np.float32)
    flow[0, :, 60:70, 300:330] = 300.0                                 # a hole (its sources leave the image) ...
    for (y, x, fx, fy) in ((5, 7, 150.0, 0.0), (100, 500, -200.5, 40.25), (160, 20, 30.0, -100.0), (64, 310, 0.0, 25.0),
                           (40, 280, 35.5, 24.75), (90, 100, -24.0, 3.0), (12, 400, 23.99, -23.99)):
        flow[0, 0, y, x], flow[0, 1, y, x] = fx, fy                    # ... that (40, 280) -> (315.5, 64.75) lands in
    flow[2, 0, 120:128, 200:208] += 40.0                               # a patch moving 40 px right, 30 px up
    flow[2, 1, 120:128, 200:208] -= 30.0
    for fill in (0, 1):
        want_out, want_cnt = oracle.flow_projection_forward(flow, fill)
        cnt, out = torch.full((B, 1, H, W), 7.0, device=dev()), torch.full((B, 2, H, W), 7.0, device=dev())
        assert my_lib.FlowProjectionLayer_gpu_forward(T(flow), cnt, out, fill) == 0
        assert np.array_equal(N(cnt), want_cnt), "count, fill %d" % fill
        close(N(out), want_out, "FlowProjection with a few far sources, fill %d" % fill)
        want_out, want_cnt = oracle.depth_flow_projection_forward(flow, depth, fill)
        cnt, out = torch.full((B, 1, H, W), 7.0, device=dev()), torch.full((B, 2, H, W), 7.0, device=dev())
        assert my_lib.DepthFlowProjectionLayer_gpu_forward(T(flow), T(depth), cnt, out, fill) == 0
        close(N(cnt), want_cnt, "depth count, fill %d" % fill, RTOL)
        close(N(out), want_out, "DepthFlowProjection with a few far sources, fill %d" % fill, RTOL)
    # more source tiles per image than proj_owner_far lists in one round (512): far sources at both ends of the table
    Hb, Wb = 1060, 1030                                                # 34 x 17 = 578 tiles
    big = synth.np_flow(rng, 1, Hb, Wb, "smooth", 3.0)
    for (y, x, fx, fy) in ((3, 5, 900.0, 1000.0), (1050, 1020, -1000.25, -1040.5), (500, 500, 0.0, 540.0), (1040, 10, 64.0, -700.0),
                           (30, 1000, -60.0, 1020.0)):
        big[0, 0, y, x], big[0, 1, y, x] = fx, fy
    want_out, want_cnt = oracle.flow_projection_forward(big, 1)
    cnt, out = torch.zeros((1, 1, Hb, Wb), device=dev()), torch.zeros((1, 2, Hb, Wb), device=dev())
    assert my_lib.FlowProjectionLayer_gpu_forward(T(big), cnt, out, 1) == 0
    assert np.array_equal(N(cnt), want_cnt), "count, 578 tiles"
    close(N(out), want_out, "FlowProjection, far sources across 578 tiles")
    # more images than the owner kernel has per-image flag words (256): far sources in images 3 and 259 only
    many = synth.np_flow(rng, 260, 40, 136, "smooth", 2.0)
    many[3, 0, 10, 5], many[259, 1, 30, 100], many[259, 0, 30, 100] = 100.0, -27.0, -80.0
    want_out, want_cnt = oracle.flow_projection_forward(many, 1)
    cnt, out = torch.zeros((260, 1, 40, 136), device=dev()), torch.zeros((260, 2, 40, 136), device=dev())
    assert my_lib.FlowProjectionLayer_gpu_forward(T(many), cnt, out, 1) == 0
    assert np.array_equal(N(cnt), want_cnt), "count, 260 images"
    close(N(out), want_out, "FlowProjection, 260 images, far sources in two of them")
    # every image far, by 30-90 px: (nearly) every tile redone
    for ci, sigma in enumerate((30.0, 90.0)):
        big = synth.np_flow(rng, 2, 100, 300, "smooth", sigma)
        for fill in (0, 1):
            want_out, want_cnt = oracle.flow_projection_forward(big, fill)
            cnt, out = torch.zeros((2, 1, 100, 300), device=dev()), torch.zeros((2, 2, 100, 300), device=dev())
            assert my_lib.FlowProjectionLayer_gpu_forward(T(big), cnt, out, fill) == 0
            assert np.array_equal(N(cnt), want_cnt), "count, sigma %g, fill %d" % (sigma, fill)
            close(N(out), want_out, "FlowProjection, smooth flow of sigma %g, fill %d" % (sigma, fill))


PAN_CASES = [
    # (pan x, pan y, local sigma): the image's dominant motion m is the mean of 64 sample sites rounded to 4 px
    (40.0, -20.0, 3.0), (-37.5, 55.25, 3.0), (6.0, 0.0, 3.0), (2.1, -1.9, 3.0), (130.0, 0.0, 2.0), (0.0, -97.0, 2.0),
    (600.0, 10.0, 1.0),            # beyond the image: nothing lands
    (26.0, 30.5, 9.0),             # local motion up to ~24 px around the pan: some sources far FROM THE PAN
]


@pytest.mark.parametrize("case", PAN_CASES, ids=["pan%g_%g_s%g" % c for c in PAN_CASES])
def test_projection_scan_shifted_by_the_dominant_motion(oracle, case):
    """Round 5: the owner kernel scans [tile - m] for the image's dominant motion m, "far" means far from m.  Pans of any size
    and direction on top of a gentle local flow, images of one batch moving differently (each has its own m), fast
    objects against the pan (far sources: their landing tiles recomputed, the home tile itself included), a few NaN / Inf
    samples (m falls back to 0 or is pulled off: results must not care) -- counts bit for bit, outputs to 1e-4, both
    operators, with and without hole filling, against the oracle (the reference's cost and results are motion-independent,
    my_lib_kernel.cu:1630-1690)."""
    import my_package._ext.my_lib as my_lib
    px, py, sigma = case
    B, H, W = 3, 200, 456                                              # 7 x 8 tiles, cut by both edges; 8 x 8 sample sites
    rng = np.random.default_rng(int(abs(px) * 7 + abs(py) * 13 + sigma))
    flow = synth.np_flow(rng, B, H, W, "smooth", sigma)
    flow[0, 0] += px;  flow[0, 1] += py                                # image 0: the pan
    flow[1, 0] -= py;  flow[1, 1] += px * 0.5                          # image 1: another one
    flow[2, 0] += px;  flow[2, 1] += py                                # image 2: the pan + objects moving against it
    flow[2, :, 50:70, 100:140] = 0.0                                   #   a static patch (far from the pan when |pan| >= 24)
    flow[2, 0, 120:130, 300:320] -= 60.0                               #   a fast object
    flow[2, 1, 150, 40], flow[2, 0, 150, 40] = 190.0, -33.0           #   single far sites
    flow[2, 0, 10, 400] = -395.5
    depth = (rng.random((B, 1, H, W)) + 0.1).astype(np.float32)
    for poison in (False, True):
        if poison:                                                     # two of image 0's sample sites non-finite: m = 0 there
            flow = flow.copy()
            ys, xs = ((2 * 3 + 1) * H) >> 4, ((2 * 5 + 1) * W) >> 4
            flow[0, 0, ys, xs] = np.nan
            flow[0, 1, ((2 * 6 + 1) * H) >> 4, ((2 * 1 + 1) * W) >> 4] = np.inf
        for fill in (0, 1):
            want_out, want_cnt = oracle.flow_projection_forward(flow, fill)
            cnt, out = torch.full((B, 1, H, W), 7.0, device=dev()), torch.full((B, 2, H, W), 7.0, device=dev())
            assert my_lib.FlowProjectionLayer_gpu_forward(T(flow), cnt, out, fill) == 0
            assert my_lib.last_kernel_path() == "proj_fwd:owner"
            assert np.array_equal(N(cnt), want_cnt), "count, fill %d, poison %s" % (fill, poison)
            close(N(out), want_out, "FlowProjection under pan %s, fill %d, poison %s" % (case, fill, poison))
            want_out, want_cnt = oracle.depth_flow_projection_forward(flow, depth, fill)
            cnt, out = torch.full((B, 1, H, W), 7.0, device=dev()), torch.full((B, 2, H, W), 7.0, device=dev())
            assert my_lib.DepthFlowProjectionLayer_gpu_forward(T(flow), T(depth), cnt, out, fill) == 0
            close(N(cnt), want_cnt, "depth count under pan %s, fill %d" % (case, fill), RTOL)
            close(N(out), want_out, "DepthFlowProjection under pan %s, fill %d, poison %s" % (case, fill, poison), RTOL)


@pytest.mark.parametrize("pan", [216.0, -300.0, 48.0, 0.0])
def test_projection_pan_with_heavy_convergence_on_a_corner_cell(oracle, pan):
    """Round-5 review (proj_owner5.hpp, packed plane): FlowProjection keeps count * 2^20 + sum(vx) in ONE double per point and
    splits it at the readout, which needs |sum of the weighted vx| < 2^19.  Under a pan m the sources that are "not far" carry
    |fx| up to |m| + 24: a block of 47 x 24 sources converging on the image's bottom-right cell (border weights 4) under a
    216 px pan sums to 975 k -- the count came out wrong by one or more, silently.  The plane now holds the residual mx - fx.
    Counts bit for bit, outputs against the oracle; the reference has no such limit (fp32 atomics, my_lib_kernel.cu:1676-1689)."""
    import my_package._ext.my_lib as my_lib
    B, H, W = 2, 128, 512
    flow = np.zeros((B, 2, H, W), np.float32)
    flow[:, 0] = pan
    ys, xs = np.mgrid[0:H, 0:W].astype(np.float32)
    # image 0: every source within 24 px of (the corner - pan) lands exactly on the corner cell (W - 1, H - 1) -- or, for a
    # negative pan, on (0, H - 1): column weight 1, row weight 2
    tx = float(W - 1) if pan >= 0 else 0.0
    block = (np.abs((tx - xs) - pan) < 23.5) & ((H - 1) - ys < 23.5)
    flow[0, 0][block] = (tx - xs)[block]
    flow[0, 1][block] = ((H - 1) - ys)[block]
    # image 1: convergence on an interior point under the same pan, sub-pixel landing (all four cells get it)
    cx, cy = (W // 2 + 0.5 if abs(pan) < 100 else (W - 40.25 if pan > 0 else 40.25)), 70.25
    block = (np.abs((cx - xs) - pan) < 23.0) & (np.abs(cy - ys) < 23.0)
    flow[1, 0][block] = (cx - xs)[block]
    flow[1, 1][block] = (cy - ys)[block]
    for fill in (0, 1):
        want_out, want_cnt = oracle.flow_projection_forward(flow, fill)
        assert want_cnt[0].max() >= 2000 and want_cnt[1].max() >= 2000                 # the test is what it says it is:
        if abs(pan) > 200:                                                              # count * |vx| at the corner cell > 2^19
            assert want_cnt[0].max() * abs(pan) > 524288.0
        cnt, out = torch.full((B, 1, H, W), 7.0, device=dev()), torch.full((B, 2, H, W), 7.0, device=dev())
        assert my_lib.FlowProjectionLayer_gpu_forward(T(flow), cnt, out, fill) == 0
        assert my_lib.last_kernel_path() == "proj_fwd:owner"
        assert np.array_equal(N(cnt), want_cnt), "count under pan %g with convergence, fill %d" % (pan, fill)
        close(N(out), want_out, "FlowProjection under pan %g with convergence, fill %d" % (pan, fill))


RAGGED_CASES = [(2, 70, 130, 0.0), (1, 45, 67, 0.0), (2, 64, 133, 0.0), (1, 100, 1278, 0.0), (2, 70, 130, 37.0), (1, 33, 9, 0.0),
                (2, 96, 255, -50.0)]


@pytest.mark.parametrize("case", RAGGED_CASES, ids=["%dx%dx%d_pan%g" % c for c in RAGGED_CASES])
def test_projection_on_widths_that_are_not_multiples_of_four(oracle, case):
    """Round 5: the owner kernels' ragged-row instantiation (the row's last quad holds W % 4 sites: loads moved left and
    rotated back, stores site by site) instead of the scalar kernels (40x slower at 1278 x 720, round 4) -- smooth and i.i.d.
    flow, far sources, pans, hole filling, both operators, against the oracle; the reference serves every width with the
    same kernel (my_lib_kernel.cu:10-15)."""
    import my_package._ext.my_lib as my_lib
    B, H, W, pan = case
    rng = np.random.default_rng(W * 7 + H)
    depth = synth.np_depth(rng, B, H, W)
    for kind, sigma in (("smooth", 4.0), ("iid", 3.0), ("smooth", 30.0)):
        flow = synth.np_flow(rng, B, H, W, kind, sigma)
        flow[:, 0] += np.float32(pan)
        flow[:, 1] -= np.float32(pan / 2)
        flow[0, 0, H // 2, W - 1] = -float(W // 2)                     # a far source in the row's last (partial) quad
        flow[0, 1, H // 3, W - 2] = 31.0
        for fill in (0, 1):
            want_out, want_cnt = oracle.flow_projection_forward(flow, fill)
            cnt, out = torch.full((B, 1, H, W), 7.0, device=dev()), torch.full((B, 2, H, W), 7.0, device=dev())
            assert my_lib.FlowProjectionLayer_gpu_forward(T(flow), cnt, out, fill) == 0
            assert my_lib.last_kernel_path() == "proj_fwd:owner"
            assert np.array_equal(N(cnt), want_cnt), "count, %s, fill %d" % (kind, fill)
            close(N(out), want_out, "FlowProjection at width %d, %s, fill %d" % (W, kind, fill))
            want_out, want_cnt = oracle.depth_flow_projection_forward(flow, depth, fill)
            cnt, out = torch.full((B, 1, H, W), 7.0, device=dev()), torch.full((B, 2, H, W), 7.0, device=dev())
            assert my_lib.DepthFlowProjectionLayer_gpu_forward(T(flow), T(depth), cnt, out, fill) == 0
            close(N(cnt), want_cnt, "depth count at width %d, %s, fill %d" % (W, kind, fill), RTOL)
            close(N(out), want_out, "DepthFlowProjection at width %d, %s, fill %d" % (W, kind, fill), RTOL)
    # ... and seen through a view (row stride != width), with the buffer's bytes behind each row left alone
    flow = synth.np_flow(rng, B, H, W, "smooth", 4.0)
    wide = torch.full((B, 2, H, W + 3), 55.0, device=dev())
    cw, ow = torch.full((B, 1, H, W + 3), 7.0, device=dev()), torch.full((B, 2, H, W + 3), 7.0, device=dev())
    wide[..., :W].copy_(T(flow))
    assert my_lib.FlowProjectionLayer_gpu_forward(wide[..., :W], cw[..., :W], ow[..., :W], 1) == 0
    want_out, want_cnt = oracle.flow_projection_forward(flow, 1)
    assert np.array_equal(N(cw[..., :W]), want_cnt)
    close(N(ow[..., :W]), want_out, "FlowProjection at width %d through a view" % W)
    assert float((cw[..., W:] - 7.0).abs().max()) == 0 and float((ow[..., W:] - 7.0).abs().max()) == 0


@pytest.mark.parametrize("W", [50, 133, 258, 1278])
def test_ragged_widths_on_the_tiled_backward_passes(oracle, W):
    """Round 5: a width that is not a multiple of four no longer sends the RGB backward passes (and the bilinear warp) to the
    one-lane-per-site kernels (FilterInterpolation backward 20x, Interpolation backward 13x slower at 1278 x 720): the
    tiled kernel serves the whole quads of every row -- the image's true width in every clamp, validity test and staged box,
    the box's last quad loaded to end at the row's end --, the direct kernel the one to three columns behind them, both
    adding into gradinput1.  Against the oracle, smooth and i.i.d. flow (boxes across the ragged edge), through a view
    whose rows are longer than the width (the elements behind each row must stay untouched)."""
    import my_package._ext.my_lib as my_lib
    rng = np.random.default_rng(W)
    B, C, H = 2, 3, 40 if W > 1000 else 70
    for kind, sigma in (("smooth", 5.0), ("iid", 4.0)):
        xn, gn = synth.np_image(rng, B, C, H, W), synth.np_image(rng, B, C, H, W)
        fn, kn = synth.np_flow(rng, B, H, W, kind, sigma), synth.np_filter(rng, B, H, W)
        pad = 3

        def wide(a, fill):
            t = torch.full((a.shape[0], a.shape[1], H, W + pad), fill, device=dev())
            t[..., :W].copy_(T(a))
            return t
        x, g, f, k = wide(xn, 9.0), wide(gn, 9.0), wide(fn, 9.0), wide(kn, 9.0)
        v = lambda t: t[..., :W]                                             # noqa: E731
        g1, g2, g3 = wide(np.zeros_like(xn), 0.0), wide(np.zeros_like(fn), 7.0), wide(np.zeros_like(kn), 7.0)
        g1[..., W:] = 7.0
        assert my_lib.FilterInterpolationLayer_gpu_backward(v(x), v(f), v(k), v(g), v(g1), v(g2), v(g3)) == 0
        assert my_lib.last_kernel_path() == "fi_bwd:tiled_c3"
        w1, w2, w3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
        close(N(v(g1)), w1, "FI gradinput1 at width %d, %s" % (W, kind), RTOL)
        close(N(v(g2)), w2, "FI gradinput2 at width %d, %s" % (W, kind), RTOL)
        close(N(v(g3)), w3, "FI gradinput3 at width %d, %s" % (W, kind), RTOL)
        for t in (g1, g2, g3):
            assert float((t[..., W:] - 7.0).abs().max()) == 0
        out = wide(np.zeros_like(xn), 7.0)
        assert my_lib.InterpolationLayer_gpu_forward(v(x), v(f), v(out)) == 0
        assert my_lib.last_kernel_path() == "bl_fwd:tiled_c3"
        close(N(v(out)), oracle.interpolation_forward(xn, fn), "Interpolation fwd at width %d, %s" % (W, kind))
        g1, g2 = wide(np.zeros_like(xn), 0.0), wide(np.zeros_like(fn), 7.0)
        g1[..., W:] = 7.0
        assert my_lib.InterpolationLayer_gpu_backward(v(x), v(f), v(g), v(g1), v(g2)) == 0
        assert my_lib.last_kernel_path() == "bl_bwd:tiled_c3"
        w1, w2 = oracle.interpolation_backward(xn, fn, gn)
        close(N(v(g1)), w1, "Interpolation gradinput1 at width %d, %s" % (W, kind), RTOL)
        close(N(v(g2)), w2, "Interpolation gradinput2 at width %d, %s" % (W, kind), RTOL)
        for t in (out, g1, g2):
            assert float((t[..., W:] - 7.0).abs().max()) == 0


@pytest.mark.parametrize("W", [50, 133, 258, 1278])
def test_ragged_widths_on_the_tiled_gathers(oracle, W):
    """Round 5, second half: the two gathers that had kept their one-lane-per-site kernels at widths that are not multiples of
    four (FilterInterpolation forward 2.4x, the projections' backward 1.8x slower at 1278 x 720) take the same split (the
    forward at any channel count: RGB kernel, chunk loop, chunk pipeline) -- the
    tiled kernel on the whole quads with the image's true width in every clamp and staged box, the scalar kernel on the one
    to three columns behind them.  Against the oracle, smooth and i.i.d. flow, through views whose rows are longer than the
    width: what lies behind a row stays untouched."""
    import my_package._ext.my_lib as my_lib
    rng = np.random.default_rng(1000 + W)
    B, H = 2, 40 if W > 1000 else 70
    pad = 3

    def wide(a, fill):
        t = torch.full((a.shape[0], a.shape[1], H, W + pad), fill, device=dev())
        t[..., :W].copy_(T(a))
        return t
    v = lambda t: t[..., :W]                                                 # noqa: E731
    for kind, sigma in (("smooth", 5.0), ("iid", 4.0)):
        for C, path in ((3, "fi_fwd:tiled_c3"), (2, "fi_fwd:tiled_chunks"), (5, "fi_fwd:tiled_c4n_ragged"), (8, "fi_fwd:tiled_c4n_ragged")):
            xn, fn, kn = synth.np_image(rng, B, C, H, W), synth.np_flow(rng, B, H, W, kind, sigma), synth.np_filter(rng, B, H, W)
            x, f, k, out = wide(xn, 9.0), wide(fn, 9.0), wide(kn, 9.0), wide(np.zeros_like(xn), 7.0)
            assert my_lib.FilterInterpolationLayer_gpu_forward(v(x), v(f), v(k), v(out)) == 0
            assert my_lib.last_kernel_path() == path
            close(N(v(out)), oracle.filter_interpolation_forward(xn, fn, kn), "FI fwd, %d channels, width %d, %s" % (C, W, kind))
            assert float((out[..., W:] - 7.0).abs().max()) == 0
        # the projections' backward passes: the forward results from the oracle (so that only the backward is under test)
        fn, dn, gn = synth.np_flow(rng, B, H, W, kind, sigma), synth.np_depth(rng, B, H, W), synth.np_flow(rng, B, H, W, "iid", 1.0)
        won, wcn = oracle.flow_projection_forward(fn, 0)
        f, cnt, g, gin = wide(fn, 9.0), wide(wcn, 9.0), wide(gn, 9.0), wide(np.zeros_like(fn), 7.0)
        assert my_lib.FlowProjectionLayer_gpu_backward(v(f), v(cnt), v(g), v(gin)) == 0
        assert my_lib.last_kernel_path() == "proj_bwd:tiled"
        close(N(v(gin)), oracle.flow_projection_backward(fn, wcn, gn), "FlowProjection bwd at width %d, %s" % (W, kind), RTOL)
        assert float((gin[..., W:] - 7.0).abs().max()) == 0
        won, wcn = oracle.depth_flow_projection_forward(fn, dn, 0)
        d, cnt, fo = wide(dn, 9.0), wide(wcn, 9.0), wide(won, 9.0)
        gin, gd = wide(np.zeros_like(fn), 7.0), wide(np.zeros_like(dn), 7.0)
        assert my_lib.DepthFlowProjectionLayer_gpu_backward(v(f), v(d), v(cnt), v(fo), v(g), v(gin), v(gd)) == 0
        assert my_lib.last_kernel_path() == "dproj_bwd:tiled"
        w1, w2 = oracle.depth_flow_projection_backward(fn, dn, wcn, won, gn)
        close(N(v(gin)), w1, "DepthFlowProjection gradinput1 at width %d, %s" % (W, kind), RTOL)
        close(N(v(gd)), w2, "DepthFlowProjection gradinput2 at width %d, %s" % (W, kind), RTOL)
        for t in (gin, gd):
            assert float((t[..., W:] - 7.0).abs().max()) == 0


def test_projection_far_tiles_are_dealt_out_evenly(oracle):
    """proj_owner_far takes the STAMPED tiles by rank (round 5), whatever their number and position: a few (one workgroup
    each), a run of consecutive ones, more than the grid holds workgroups (720p: 460 tiles per image, batch 3), all of
    them -- against the oracle."""
    import my_package._ext.my_lib as my_lib
    rng = np.random.default_rng(99)
    B, H, W = 3, 720, 1280
    base = synth.np_flow(rng, B, H, W, "smooth", 3.0)
    variants = []
    f = base.copy();  f[1, 0, 100:110, 200:210] += 90.0;  variants.append(("one object", f))
    f = base.copy();  f[:, 0, ::37, ::53] += 333.0;  f[:, 1, ::41, ::59] -= 111.0;  variants.append(("far sites everywhere", f))
    f = base.copy();  f[0] *= 9.0;  variants.append(("image 0 far throughout", f))
    for name, f in variants:
        want_out, want_cnt = oracle.flow_projection_forward(f, 1)
        cnt, out = torch.full((B, 1, H, W), 7.0, device=dev()), torch.full((B, 2, H, W), 7.0, device=dev())
        assert my_lib.FlowProjectionLayer_gpu_forward(T(f), cnt, out, 1) == 0
        assert np.array_equal(N(cnt), want_cnt), name
        close(N(out), want_out, "FlowProjection, %s" % name)


def test_empty_batches_and_single_pixel_images(oracle):
    """Nothing to do is not an error: an empty batch returns 0 from every entry point without touching anything (the
    reference would launch a grid of zero blocks and report the launch error).  The smallest image there is -- one pixel,
    and one row of four -- against the oracle, every operator."""
    import my_package._ext.my_lib as my_lib
    z = lambda c, h=8, w=16: torch.zeros((0, c, h, w), device=dev())          # noqa: E731
    assert my_lib.FilterInterpolationLayer_gpu_forward(z(3), z(2), z(16), z(3)) == 0
    assert my_lib.FilterInterpolationLayer_gpu_backward(z(3), z(2), z(16), z(3), z(3), z(2), z(16)) == 0
    assert my_lib.InterpolationLayer_gpu_forward(z(3), z(2), z(3)) == 0
    assert my_lib.InterpolationChLayer_gpu_backward(z(5), z(2), z(5), z(5), z(2)) == 0
    assert my_lib.FlowProjectionLayer_gpu_forward(z(2), z(1), z(2), 1) == 0
    assert my_lib.FlowProjectionLayer_gpu_backward(z(2), z(1), z(2), z(2)) == 0
    assert my_lib.DepthFlowProjectionLayer_gpu_forward(z(2), z(1), z(1), z(2), 1) == 0
    for ci, (H, W) in enumerate(((1, 1), (1, 4), (2, 2))):
        rng = np.random.default_rng(70 + ci)
        x, k, g = synth.np_image(rng, 2, 3, H, W), synth.np_filter(rng, 2, H, W), synth.np_image(rng, 2, 3, H, W)
        f = (rng.random((2, 2, H, W), dtype=np.float32) - 0.5) * 1.5
        dpt = synth.np_depth(rng, 2, H, W)
        out = torch.full(x.shape, 7.0, device=dev())
        assert my_lib.FilterInterpolationLayer_gpu_forward(T(x), T(f), T(k), out) == 0
        close(N(out), oracle.filter_interpolation_forward(x, f, k), "FI forward %dx%d" % (H, W))
        g1, g2, g3 = torch.zeros(x.shape, device=dev()), torch.full(f.shape, 7.0, device=dev()), torch.full(k.shape, 7.0, device=dev())
        assert my_lib.FilterInterpolationLayer_gpu_backward(T(x), T(f), T(k), T(g), g1, g2, g3) == 0
        for got, want, what in zip((g1, g2, g3), oracle.filter_interpolation_backward(x, f, k, g), ("gradinput1", "gradinput2", "gradinput3")):
            close(N(got), want, "FI backward %s %dx%d" % (what, H, W), RTOL)
        out = torch.full(x.shape, 7.0, device=dev())
        assert my_lib.InterpolationLayer_gpu_forward(T(x), T(f), out) == 0
        close(N(out), oracle.interpolation_forward(x, f), "Interpolation forward %dx%d" % (H, W))
        for fill in (0, 1):
            cnt, po = torch.full((2, 1, H, W), 7.0, device=dev()), torch.full(f.shape, 7.0, device=dev())
            assert my_lib.FlowProjectionLayer_gpu_forward(T(f), cnt, po, fill) == 0
            want_out, want_cnt = oracle.flow_projection_forward(f, fill)
            assert np.array_equal(N(cnt), want_cnt)
            close(N(po), want_out, "FlowProjection %dx%d fill %d" % (H, W, fill))
            cnt, po = torch.full((2, 1, H, W), 7.0, device=dev()), torch.full(f.shape, 7.0, device=dev())
            assert my_lib.DepthFlowProjectionLayer_gpu_forward(T(f), T(dpt), cnt, po, fill) == 0
            want_out, want_cnt = oracle.depth_flow_projection_forward(f, dpt, fill)
            close(N(cnt), want_cnt, "depth count %dx%d fill %d" % (H, W, fill), RTOL)
            close(N(po), want_out, "DepthFlowProjection %dx%d fill %d" % (H, W, fill), RTOL)


GOLDEN = sorted(p for p in glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz"))
                if os.path.basename(p).startswith(("small_", "config1_")))   # the oracle-made operator fixtures


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_against_committed_golden_vectors(path):
    """Same comparison against the stored bytes (no oracle run involved)."""
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    from my_package.modules.FlowProjectionModule import FlowProjectionModule
    g = np.load(path)
    x, f, k = T(g["x"], True), T(g["flow"], True), T(g["filt"], True)
    out = FilterInterpolationModule()(x, f, k)
    close(N(out), g["fi_out"], "forward")
    if "fi_g1" in g:
        out.backward(T(g["gout"]))
        close(N(x.grad), g["fi_g1"], "gradinput1", RTOL)
        close(N(f.grad), g["fi_g2"], "gradinput2", RTOL)
        close(N(k.grad), g["fi_g3"], "gradinput3", RTOL)
    with torch.no_grad():
        p = FlowProjectionModule(requires_grad=False)(T(g["flow"]))
    close(N(p), g["fp_out1"], "projection fillhole=1")


def test_non_contiguous_inputs_and_strided_descriptors(oracle):
    """(a) the modules accept non-contiguous views (made contiguous on the host side);
    (b) the C ABI honours b/c/h strides, as the reference kernels do (my_lib_kernel.cu:1123-1150)."""
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    import my_package._ext.my_lib as my_lib
    rng = np.random.default_rng(9)
    B, C, H, W = 2, 3, 20, 36
    big = synth.np_image(rng, B, C + 2, H + 3, W)
    xn = big[:, 1:1 + C, 2:2 + H, :]                              # b, c, h strided view, unit w stride
    fn, kn = synth.np_flow(rng, B, H, W, "iid", 2.0), synth.np_filter(rng, B, H, W)
    want = oracle.filter_interpolation_forward(np.ascontiguousarray(xn), fn, kn)
    tbig = T(big)
    xv = tbig[:, 1:1 + C, 2:2 + H, :]
    assert not xv.is_contiguous()
    close(N(FilterInterpolationModule()(xv, T(fn), T(kn))), want, "module on a view")
    # strided descriptor straight through the ABI: output must share input1's b/c/h strides
    obig = torch.zeros_like(tbig)
    ov = obig[:, 1:1 + C, 2:2 + H, :]
    assert my_lib.FilterInterpolationLayer_gpu_forward(xv, T(fn), T(kn), ov) == 0
    close(N(ov), want, "strided descriptors")
    assert float(obig[:, 0].abs().max()) == 0.0                   # nothing written outside the view
    # mismatching output strides are rejected, not silently mis-written
    assert my_lib.FilterInterpolationLayer_gpu_forward(xv, T(fn), T(kn), torch.zeros(B, C, H, W, device=dev())) == -1


def test_backward_accumulates_into_caller_buffers(oracle):
    """gradinput1 is a `+=` target (scatter; the reference's atomicAdd): a pre-filled buffer must come back as
    prefill + gradient.  gradinput2 is assigned.  gradinput3 must be zero-filled by the caller
    (FilterInterpolationLayer.py:48) -- the tiled kernel then stores each site's tap gradients once instead of
    read-modify-writing 64 B per site; invalid sites are never touched."""
    import my_package._ext.my_lib as my_lib
    d = make(CASES[1])
    x, f, k, g = T(d["x"]), T(d["flow"]), T(d["filt"]), T(d["gout"])
    g1 = torch.full_like(x, 0.5); g2 = torch.zeros_like(f); g3 = torch.zeros_like(k)
    assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
    w1, w2, w3 = oracle.filter_interpolation_backward(d["x"], d["flow"], d["filt"], d["gout"])
    close(N(g1), w1 + 0.5, "gradinput1 += ", RTOL)
    close(N(g3), w3, "gradinput3 ", RTOL)
    close(N(g2), w2, "gradinput2 = ", RTOL)


def test_streams_and_repeatability(oracle):
    """Work is enqueued on the caller's current stream; two streams give the same answer."""
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    d = make(CASES[0])
    x, f, k = T(d["x"]), T(d["flow"]), T(d["filt"])
    a = FilterInterpolationModule()(x, f, k)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        b = FilterInterpolationModule()(x, f, k)
    s.synchronize()
    torch.cuda.synchronize()
    assert torch.equal(a, b)                                       # the forward gather is deterministic


def test_headline_size_image_gradients_are_the_adjoints_of_the_warps():
    """BASELINE's headline size (32 x 3 x 720 x 1280), a size-independent property of the two scattering backward passes:
    for a fixed flow (and taps) the warp is linear in the image, and gradinput1 is its transpose applied to gradoutput --
    <warp(x), g> == <x, gradinput1(g)> (float64 sums; FilterInterpolation copies the input pixel at out-of-range sites and
    back-propagates nothing there, so its g is masked to the valid sites)."""
    import my_package._ext.my_lib as my_lib
    B, C, H, W = 32, 3, 720, 1280
    t = synth.torch_inputs(dev(), B, C, H, W, flow_kind="smooth", seed=77, with_grad=True)
    x, f, k, g = t["x"], t["flow"], t["filt"], t["gout"]
    xs = torch.arange(W, device=dev(), dtype=torch.float32).view(1, 1, W)
    ys = torch.arange(H, device=dev(), dtype=torch.float32).view(1, H, 1)
    x2, y2 = xs + f[:, 0], ys + f[:, 1]
    valid = (x2 >= 0) & (y2 >= 0) & (x2 <= W - 1) & (y2 <= H - 1) & (f[:, 0].abs() < W / 2.0) & (f[:, 1].abs() < H / 2.0)
    gm = g * valid.unsqueeze(1)

    def dot(a, b):
        return float((a.double() * b.double()).sum())
    out = torch.empty_like(x)
    g1, g2, g3 = torch.zeros_like(x), torch.zeros_like(f), torch.zeros_like(k)
    assert my_lib.FilterInterpolationLayer_gpu_forward(x, f, k, out) == 0
    assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, gm, g1, g2, g3) == 0
    lhs, rhs = dot(out, gm), dot(x, g1)
    assert abs(lhs - rhs) <= 2e-6 * abs(lhs), ("FilterInterpolation", lhs, rhs)
    del g3
    out.zero_(); g1.zero_()
    assert my_lib.InterpolationLayer_gpu_forward(x, f, out) == 0
    assert my_lib.InterpolationLayer_gpu_backward(x, f, g, g1, g2) == 0
    lhs, rhs = dot(out, g), dot(x, g1)
    assert abs(lhs - rhs) <= 2e-6 * abs(lhs), ("Interpolation", lhs, rhs)


def test_full_size_properties():
    """BASELINE sizes (720p, batch 8 here to bound host time) through size-independent properties:
      * zero flow + one-hot tap 5 is the identity;
      * linearity in the image: FI(a*x1 + x2) == a*FI(x1) + FI(x2) for the same flow/taps;
      * FlowProjection: count sums to 4 * (number of valid source sites) and sum(out*count) == -4 * sum of
        valid flows (checksum of checksums), holes are all filled when every row has a valid cell."""
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    import my_package._ext.my_lib as my_lib
    B, C, H, W = 8, 3, 720, 1280
    t = synth.torch_inputs(dev(), B, C, H, W, flow_kind="smooth", seed=99)
    fi = FilterInterpolationModule()
    onehot = torch.zeros_like(t["filt"]); onehot[:, 5] = 1
    assert torch.equal(fi(t["x"], torch.zeros_like(t["flow"]), onehot), t["x"])
    x2 = torch.rand_like(t["x"])
    lhs = fi(2.5 * t["x"] + x2, t["flow"], t["filt"])
    rhs = 2.5 * fi(t["x"], t["flow"], t["filt"]) + fi(x2, t["flow"], t["filt"])
    assert float((lhs - rhs).abs().max()) <= ATOL
    flow = t["flow"]
    count = flow.new_zeros((B, 1, H, W)); out = torch.zeros_like(flow)
    assert my_lib.FlowProjectionLayer_gpu_forward(flow, count, out, 0) == 0
    xs = torch.arange(W, device=dev(), dtype=torch.float32).view(1, 1, W)
    ys = torch.arange(H, device=dev(), dtype=torch.float32).view(1, H, 1)
    x2f, y2f = xs + flow[:, 0], ys + flow[:, 1]
    valid = (x2f >= 0) & (y2f >= 0) & (x2f <= W - 1) & (y2f <= H - 1)
    assert float(count.sum(dtype=torch.float64)) == 4.0 * float(valid.sum())
    for k in range(2):
        got = float((out[:, k] * count[:, 0]).sum(dtype=torch.float64))
        want = -4.0 * float((flow[:, k] * valid).sum(dtype=torch.float64))
        assert abs(got - want) <= 1e-3 * max(1.0, abs(want))


@pytest.mark.parametrize("case", [CASES[i] for i in (0, 5, 6, 8, 12, 16)], ids=[IDS[i] for i in (0, 5, 6, 8, 12, 16)])
def test_tile_walks_and_staging_budgets(oracle, case):
    """The 2x2-footprint kernels take their tile walk (strips / stripes of n tile columns per XCD) and their LDS
    staging budget (48 / 39 / 31 KiB) from measurement knobs; every combination must give the oracle's results
    (a smaller budget only moves sites from the staged to the global-gather path)."""
    from tools import measure as M          # forced paths exist in the measurement build only
    my_lib = M.bound()                      # (the same sources, -DMEMC_MEASURE; my_package stays on the product library)
    d = make(case)
    if d["x"].shape[1] != 3:
        pytest.skip("RGB kernels")
    x, f, g, dep, gf = T(d["x"]), T(d["flow"]), T(d["gout"]), T(d["depth"]), T(d["gflow"])
    want_fwd = oracle.interpolation_forward(d["x"], d["flow"])
    want_g1, want_g2 = oracle.interpolation_backward(d["x"], d["flow"], d["gout"])
    _, cnt = oracle.flow_projection_forward(d["flow"], 0)
    want_p = oracle.flow_projection_backward(d["flow"], cnt, d["gflow"])
    dout, dcnt = oracle.depth_flow_projection_forward(d["flow"], d["depth"], 0)
    want_q1, want_q2 = oracle.depth_flow_projection_backward(d["flow"], d["depth"], dcnt, dout, d["gflow"])
    try:
        for walk, cap in ((0, 0), (2, 1), (4, 2), (3, -1), (-1, 1), (-1, 3), (1, 4)):
            M.set_variant("walk", walk)
            M.set_variant("bl_cap", cap)
            tag = "walk %d budget %d" % (walk, cap)
            o = torch.full_like(x, float("nan"))
            assert my_lib.InterpolationLayer_gpu_forward(x, f, o) == 0
            close(N(o), want_fwd, "interpolation forward, " + tag)
            g1, g2 = torch.zeros_like(x), torch.full_like(f, float("nan"))
            assert my_lib.InterpolationLayer_gpu_backward(x, f, g, g1, g2) == 0
            close(N(g1), want_g1, "interpolation gradinput1, " + tag, RTOL)
            close(N(g2), want_g2, "interpolation gradinput2, " + tag, RTOL)
            p1 = torch.full_like(f, float("nan"))
            assert my_lib.FlowProjectionLayer_gpu_backward(f, T(cnt), gf, p1) == 0
            close(N(p1), want_p, "projection backward, " + tag, RTOL)
            q1, q2 = torch.full_like(f, float("nan")), torch.full_like(dep, float("nan"))
            assert my_lib.DepthFlowProjectionLayer_gpu_backward(f, dep, T(dcnt), T(dout), gf, q1, q2) == 0
            close(N(q1), want_q1, "depth projection gradinput1, " + tag, RTOL)
            close(N(q2), want_q2, "depth projection gradinput2, " + tag, RTOL)
    finally:
        M.set_variant("walk", -1)
        M.set_variant("bl_cap", -1)


def test_tensors_beyond_2G_elements():
    """Maximum sizes: 288 GB of HBM holds tensors with more than 2^31 elements (here 40 x 64 x 720 x 1280 = 2.36e9,
    9.4 GB each), where any 32-bit element index would wrap.  Batch items are independent, so the last item of
    the big call must equal the same item run alone (forward and flow/tap gradients: bit-exact gathers; image
    gradient: atomics, tolerance)."""
    import my_package._ext.my_lib as my_lib
    free, _ = torch.cuda.mem_get_info()
    if free < 70 * 2**30:
        pytest.skip("needs 70 GB of free device memory")
    B, C, H, W = 40, 64, 720, 1280
    assert B * C * H * W > 2**31
    t = synth.torch_inputs(dev(), B, C, H, W, flow_kind="smooth", seed=5, with_grad=True)
    x, f, k, g = t["x"], t["flow"], t["filt"], t["gout"]
    last = lambda a: a[B - 1:].contiguous()
    out = torch.empty_like(x)
    assert my_lib.FilterInterpolationLayer_gpu_forward(x, f, k, out) == 0
    out1 = torch.empty_like(last(x))
    assert my_lib.FilterInterpolationLayer_gpu_forward(last(x), last(f), last(k), out1) == 0
    assert torch.equal(out[B - 1:], out1)
    assert my_lib.InterpolationChLayer_gpu_forward(x, f, out) == 0
    assert my_lib.InterpolationChLayer_gpu_forward(last(x), last(f), out1) == 0
    assert torch.equal(out[B - 1:], out1)
    del out
    g1, g2, g3 = torch.zeros_like(x), torch.empty_like(f), torch.empty_like(k)
    assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
    h1, h2, h3 = torch.zeros_like(out1), torch.empty_like(last(f)), torch.empty_like(last(k))
    assert my_lib.FilterInterpolationLayer_gpu_backward(last(x), last(f), last(k), last(g), h1, h2, h3) == 0
    assert torch.equal(g2[B - 1:], h2) and torch.equal(g3[B - 1:], h3)
    assert float((g1[B - 1:] - h1).abs().max()) <= ATOL
    assert float(g1[B - 1:].abs().max()) > 0


def test_stream_capture_and_replay(oracle):
    """The launchers allocate nothing and synchronise nothing, so a forward can be captured into a HIP graph and
    replayed (FlowProjection's one-time scratch allocation happens at its first call, made before capture)."""
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    import my_package._ext.my_lib as my_lib
    d = make(CASES[0])
    x, f, k = T(d["x"]), T(d["flow"]), T(d["filt"])
    flow = T(d["flow"])
    cnt = flow.new_zeros((flow.size(0), 1, flow.size(2), flow.size(3)))
    pout = torch.zeros_like(flow)
    assert my_lib.FlowProjectionLayer_gpu_forward(flow, cnt, pout, 1) == 0      # first call: allocates scratch
    out = torch.zeros_like(x)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out.zero_(); cnt.zero_(); pout.zero_()
            assert my_lib.FilterInterpolationLayer_gpu_forward(x, f, k, out) == 0
            assert my_lib.FlowProjectionLayer_gpu_forward(flow, cnt, pout, 1) == 0
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    close(N(out), oracle.filter_interpolation_forward(d["x"], d["flow"], d["filt"]), "captured FI forward")
    close(N(pout), oracle.flow_projection_forward(d["flow"], 1)[0], "captured projection")


def test_many_channel_backward_random_shapes(oracle):
    """Twelve seeded random shapes (batch, channels 4..13, ragged heights / widths -- multiples of four and not --,
    five flow kinds): both many-channel backward passes against the oracle, from garbage-filled buffers."""
    import my_package._ext.my_lib as my_lib
    rng = np.random.default_rng(20260927)
    kinds = ["smooth", "pan", "far", "converge", "iid"]
    for _ in range(12):
        B, C = int(rng.integers(1, 3)), int(rng.integers(4, 14))
        H, W = int(rng.integers(5, 90)), int(rng.integers(3, 60)) * 4 + (int(rng.integers(0, 4)) if rng.random() < 0.25 else 0)
        kind = kinds[int(rng.integers(0, len(kinds)))]
        xn, kn, gn = synth.np_image(rng, B, C, H, W), synth.np_filter(rng, B, H, W), synth.np_image(rng, B, C, H, W)
        fn = _many_channel_flows(kind, rng, B, H, W)
        what = "%dx%dx%dx%d %s" % (B, C, H, W, kind)
        x, f, k, g = T(xn), T(fn), T(kn), T(gn)
        g1, g2, g3 = torch.full_like(x, -3.0), torch.full_like(f, -3.0), torch.full_like(k, -3.0)
        assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
        w1, w2, w3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
        close(N(g1), w1, "gradinput1 " + what, 3 * RTOL)
        close(N(g2), w2, "gradinput2 " + what, 3 * RTOL)
        close(N(g3), w3, "gradinput3 " + what, 3 * RTOL)
        h1, h2 = torch.full_like(x, -3.0), torch.full_like(f, -3.0)
        assert my_lib.InterpolationChLayer_gpu_backward(x, f, g, h1, h2) == 0
        v1, v2 = oracle.interpolation_ch_backward(xn, fn, gn)
        close(N(h1), v1, "bilinear gradinput1 " + what, 3 * RTOL)
        close(N(h2), v2, "bilinear gradinput2 " + what, 3 * RTOL)


def test_rgb_backward_random_shapes_and_strided_views(oracle):
    """Sixteen seeded random shapes (ragged heights / widths, multiples of four and not, five flow kinds, signed taps and
    gradients of random magnitude) through the RGB backward passes -- packed fixed-point planes, fi_bwd_c3.hip and
    interpolation.hip -- against the oracle; then the same kernels on channel slices of wider tensors with padded rows
    (every stride different)."""
    import my_package._ext.my_lib as my_lib
    rng = np.random.default_rng(20260928)
    kinds = ["smooth", "pan", "far", "converge", "iid"]
    for _ in range(16):
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(5, 120)), int(rng.integers(3, 80)) * 4 + (int(rng.integers(0, 4)) if rng.random() < 0.2 else 0)
        kind = kinds[int(rng.integers(0, len(kinds)))]
        mag_g, mag_k = float(10.0 ** rng.uniform(-3, 2)), float(10.0 ** rng.uniform(-2, 0.5))
        xn = synth.np_image(rng, B, 3, H, W)
        kn = (rng.standard_normal((B, 16, H, W)) * mag_k / 4).astype(np.float32)
        gn = (rng.standard_normal((B, 3, H, W)) * mag_g).astype(np.float32)
        fn = _many_channel_flows(kind, rng, B, H, W)
        what = "%dx3x%dx%d %s |g|~%.3g |k|~%.3g" % (B, H, W, kind, mag_g, mag_k)
        x, f, k, g = T(xn), T(fn), T(kn), T(gn)
        g1, g2, g3 = torch.zeros_like(x), torch.full_like(f, -3.0), torch.full_like(k, -3.0)
        assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
        w1, w2, w3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
        # the operator's own scale: errors are judged against the largest gradient of the case (inputs span 1e-3 .. 1e2)
        for got, want, name in ((g1, w1, "gradinput1"), (g2, w2, "gradinput2"), (g3, w3, "gradinput3")):
            sc = max(1.0, float(np.abs(want).max()))
            close(N(got) / sc, want / sc, "%s (scaled by %.3g) %s" % (name, sc, what), 3 * RTOL)
        h1, h2 = torch.zeros_like(x), torch.full_like(f, -3.0)
        assert my_lib.InterpolationLayer_gpu_backward(x, f, g, h1, h2) == 0
        v1, v2 = oracle.interpolation_backward(xn, fn, gn)
        for got, want, name in ((h1, v1, "bilinear gradinput1"), (h2, v2, "bilinear gradinput2")):
            sc = max(1.0, float(np.abs(want).max()))
            close(N(got) / sc, want / sc, "%s (scaled by %.3g) %s" % (name, sc, what), 3 * RTOL)
    # strided views: RGB = channels 2..4 of a 6-channel tensor whose rows are 8 columns wider
    B, H, W = 2, 48, 128
    xn, kn, gn = synth.np_image(rng, B, 3, H, W), synth.np_filter(rng, B, H, W), synth.np_image(rng, B, 3, H, W)
    fn = synth.np_flow(rng, B, H, W, "smooth", 5.0)

    def wide(a, cpad, wpad, fill=0.0):
        big = torch.full((a.shape[0], a.shape[1] + cpad, a.shape[2], a.shape[3] + wpad), fill, device=dev())
        v = big[:, cpad:, :, :a.shape[3]]
        v.copy_(T(a))
        return v
    x, g = wide(xn, 2, 8), wide(gn, 2, 8)
    f, k = wide(fn, 1, 4), wide(kn, 3, 12)
    g1 = wide(np.zeros_like(xn), 2, 8)                     # same batch / channel strides as x (the layer checks that)
    g2, g3 = wide(np.zeros_like(fn), 1, 4, fill=5.0), wide(np.zeros_like(kn), 3, 12, fill=5.0)
    assert not x.is_contiguous() and not k.is_contiguous()
    assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
    w1, w2, w3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
    close(N(g1), w1, "strided RGB gradinput1", RTOL)
    close(N(g2), w2, "strided RGB gradinput2", RTOL)
    close(N(g3), w3, "strided RGB gradinput3", RTOL)
    h1, h2 = wide(np.zeros_like(xn), 2, 8), wide(np.zeros_like(fn), 1, 4, fill=5.0)
    assert my_lib.InterpolationLayer_gpu_backward(x, f, g, h1, h2) == 0
    v1, v2 = oracle.interpolation_backward(xn, fn, gn)
    close(N(h1), v1, "strided RGB bilinear gradinput1", RTOL)
    close(N(h2), v2, "strided RGB bilinear gradinput2", RTOL)


def test_many_channel_backward_on_strided_views(oracle):
    """The owner kernels index every tensor through its own batch / channel / row strides: channel slices of wider
    tensors, rows of wider images (16-byte aligned, so still the vector path), all strides different."""
    import my_package._ext.my_lib as my_lib
    from tools import measure as M
    B, C, H, W = 2, 8, 40, 128
    rng = np.random.default_rng(91)
    xn, kn, gn = synth.np_image(rng, B, C, H, W), synth.np_filter(rng, B, H, W), synth.np_image(rng, B, C, H, W)
    fn = synth.np_flow(rng, B, H, W, "smooth", 5.0)

    def wide(a, cpad, wpad, fill=7.0):
        """a copy of `a` living inside a tensor with cpad extra channels in front and wpad extra columns behind"""
        big = torch.full((a.shape[0], a.shape[1] + cpad, a.shape[2], a.shape[3] + wpad), fill, device=dev())
        v = big[:, cpad:, :, :a.shape[3]]
        v.copy_(T(a))
        return big, v
    _, x = wide(xn, 2, 4)
    _, f = wide(fn, 1, 8)
    _, k = wide(kn, 3, 12)
    g = wide(gn, 2, 4)[1]                                   # gradoutput shares input1's strides (the contract)
    b1, g1 = wide(np.zeros_like(xn), 2, 4)
    b2, g2 = wide(np.zeros_like(fn), 1, 8)
    b3, g3 = wide(np.zeros_like(kn), 3, 12)
    assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
    w1, w2, w3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
    close(N(g1), w1, "strided gradinput1", 3 * RTOL)
    close(N(g2), w2, "strided gradinput2", RTOL)
    close(N(g3), w3, "strided gradinput3", RTOL)
    for big in (b1, b2, b3):                                # nothing written outside the views
        assert float(big[:, 0].min()) == 7.0 and float(big[..., -1].min()) == 7.0
    h1, h2 = wide(np.zeros_like(xn), 2, 4)[1], wide(np.zeros_like(fn), 1, 8)[1]
    assert my_lib.InterpolationChLayer_gpu_backward(x, f, g, h1, h2) == 0
    v1, v2 = oracle.interpolation_ch_backward(xn, fn, gn)
    close(N(h1), v1, "strided bilinear gradinput1", 3 * RTOL)
    close(N(h2), v2, "strided bilinear gradinput2", 3 * RTOL)
    # and they really took the owner path
    ml = M.bound()
    import ctypes
    last = M.lib().memc_debug_last_path
    last.restype = ctypes.c_char_p
    assert ml.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0 and last().decode() == "fi_bwd:owner"
    assert ml.InterpolationChLayer_gpu_backward(x, f, g, h1, h2) == 0 and last().decode() == "bl_bwd:owner"


def test_stream_capture_many_channel_backward(oracle):
    """Inside a stream capture nothing may be allocated: the many-channel backward passes then clear gradinput1 themselves
    and take the direct kernels -- same gradients, replayable, still independent of what the buffers held."""
    import my_package._ext.my_lib as my_lib
    B, C, H, W = 1, 8, 40, 128
    rng = np.random.default_rng(77)
    xn, kn, gn = synth.np_image(rng, B, C, H, W), synth.np_filter(rng, B, H, W), synth.np_image(rng, B, C, H, W)
    fn = synth.np_flow(rng, B, H, W, "smooth", 5.0)
    x, f, k, g = T(xn), T(fn), T(kn), T(gn)
    g1, g2, g3 = torch.full_like(x, 9.0), torch.full_like(f, 9.0), torch.full_like(k, 9.0)
    h1, h2 = torch.full_like(x, 9.0), torch.full_like(f, 9.0)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
            assert my_lib.InterpolationChLayer_gpu_backward(x, f, g, h1, h2) == 0
    for _ in range(2):
        graph.replay()
    torch.cuda.synchronize()
    w1, w2, w3 = oracle.filter_interpolation_backward(xn, fn, kn, gn)
    close(N(g1), w1, "captured FI gradinput1", 3 * RTOL)
    close(N(g2), w2, "captured FI gradinput2", RTOL)
    close(N(g3), w3, "captured FI gradinput3", RTOL)
    v1, v2 = oracle.interpolation_ch_backward(xn, fn, gn)
    close(N(h1), v1, "captured bilinear gradinput1", 3 * RTOL)
    close(N(h2), v2, "captured bilinear gradinput2", 3 * RTOL)


def test_concurrent_streams_projection(oracle):
    """Two streams running the projection fast path at the same time must not share far-source flags."""
    import my_package._ext.my_lib as my_lib
    rng = np.random.default_rng(31)
    near = synth.np_flow(rng, 2, 64, 128, "smooth", 3.0)            # stays on the fast path
    far = synth.np_flow(rng, 2, 64, 128, "iid", 40.0)               # every image needs the general path
    want_near = oracle.flow_projection_forward(near, 1)[0]
    want_far = oracle.flow_projection_forward(far, 1)[0]
    tn, tf = T(near), T(far)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for it in range(20):
        cn, on = tn.new_zeros(2, 1, 64, 128), torch.zeros_like(tn)
        cf, of = tf.new_zeros(2, 1, 64, 128), torch.zeros_like(tf)
        torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            assert my_lib.FlowProjectionLayer_gpu_forward(tn, cn, on, 1) == 0
        with torch.cuda.stream(s2):
            assert my_lib.FlowProjectionLayer_gpu_forward(tf, cf, of, 1) == 0
        outs.append((on, of))
    torch.cuda.synchronize()
    for on, of in outs:
        close(N(on), want_near, "near-flow stream")
        close(N(of), want_far, "far-flow stream")


def test_4k_frame_properties():
    """BASELINE configs[4] size (3840x2160): offsets beyond 2^31 bytes per tensor, identity and linearity."""
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    B, C, H, W = 2, 3, 2160, 3840
    t = synth.torch_inputs(dev(), B, C, H, W, flow_kind="smooth", seed=5)
    fi = FilterInterpolationModule()
    onehot = torch.zeros_like(t["filt"]); onehot[:, 5] = 1
    assert torch.equal(fi(t["x"], torch.zeros_like(t["flow"]), onehot), t["x"])
    x2 = torch.rand_like(t["x"])
    lhs = fi(0.5 * t["x"] + x2, t["flow"], t["filt"])
    rhs = 0.5 * fi(t["x"], t["flow"], t["filt"]) + fi(x2, t["flow"], t["filt"])
    assert float((lhs - rhs).abs().max()) <= ATOL


def test_large_channel_count_offsets(oracle):
    """64 channels at 720p: the per-batch stride exceeds 2^25 elements; compare a strip against the oracle."""
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    rng = np.random.default_rng(41)
    B, C, H, W = 1, 64, 96, 1280
    xn, fn, kn = synth.np_image(rng, B, C, H, W), synth.np_flow(rng, B, H, W, "smooth"), synth.np_filter(rng, B, H, W)
    out = FilterInterpolationModule()(T(xn), T(fn), T(kn))
    close(N(out), oracle.filter_interpolation_forward(xn, fn, kn), "C=64 forward")


# ------------------------------------------------------------------------------------------------------------
# EXTENSION: fused dual warp + occlusion blend.  Expected value = the composition of two
# oracle forwards, blended in fp32 exactly as the reference network writes it (MEMC_Net_star.py:277); gradients =
# the oracle backward of each warp fed with occlusion * gradoutput, occlusion gradients = sum_c gout * warp.
# ------------------------------------------------------------------------------------------------------------
BLEND_CASES = [
    (2, 3, 40, 64, "smooth", 4.0, 31), (1, 3, 100, 132, "iid", 3.0, 32), (2, 3, 64, 256, "iid", 20.0, 33),
    (1, 3, 96, 256, "smooth", 25.0, 34),        # bands
    (1, 5, 24, 48, "smooth", 4.0, 35),          # not RGB: composed path of the Python layer
    (1, 3, 19, 23, "iid", 2.0, 36),             # width not a multiple of 4: composed path
]


@pytest.mark.parametrize("case", BLEND_CASES, ids=["%dx%dx%dx%d-%s" % c[:5] for c in BLEND_CASES])
def test_filter_interpolation_blend(oracle, case):
    from my_package.modules.FilterInterpolationBlendModule import FilterInterpolationBlendModule
    B, C, H, W, kind, sigma, seed = case
    a, b2 = make(case), make(case[:6] + (seed + 100,))
    rng = np.random.default_rng(seed + 7)
    o0 = rng.random((B, 1, H, W), dtype=np.float32)
    o1 = rng.random((B, 1, H, W), dtype=np.float32)
    names = ("x0", "x2", "f0", "f1", "k0", "k1", "o0", "o1")
    host = dict(x0=a["x"], x2=b2["x"], f0=a["flow"], f1=b2["flow"], k0=a["filt"], k1=b2["filt"], o0=o0, o1=o1)
    t = {n: T(host[n], True) for n in names}
    out = FilterInterpolationBlendModule()(*[t[n] for n in names])
    out.backward(T(a["gout"]))
    w0 = oracle.filter_interpolation_forward(host["x0"], host["f0"], host["k0"])
    w2 = oracle.filter_interpolation_forward(host["x2"], host["f1"], host["k1"])
    close(N(out), o0 * w0 + o1 * w2, "blend forward")
    gout = a["gout"]
    for x, f, k, o, w in (("x0", "f0", "k0", "o0", w0), ("x2", "f1", "k1", "o1", w2)):
        g1, g2, g3 = oracle.filter_interpolation_backward(host[x], host[f], host[k], (gout * host[o]).astype(np.float32))
        close(N(t[x].grad), g1, "grad " + x, RTOL)
        close(N(t[f].grad), g2, "grad " + f, RTOL)
        close(N(t[k].grad), g3, "grad " + k, RTOL)
        close(N(t[o].grad), (gout * w).sum(axis=1, keepdims=True), "grad " + o, RTOL)


def test_filter_interpolation_blend_c_abi_checks():
    """the C entry point: needs no zero-filled output; rejects what the fused kernel does not cover"""
    import my_package._ext.my_lib as my_lib
    B, H, W = 1, 32, 64
    z = lambda c: torch.rand(B, c, H, W, device=dev())          # noqa: E731
    out = torch.full((B, 3, H, W), 7.0, device=dev())
    args = [z(3), z(3), z(2) * 4 - 2, z(2) * 4 - 2, z(16) / 16, z(16) / 16, z(1), z(1)]
    assert my_lib.FilterInterpolationBlendLayer_gpu_forward(*args, out) == 0
    ref = args[6] * _fi(my_lib, args[0], args[2], args[4]) + args[7] * _fi(my_lib, args[1], args[3], args[5])
    assert float((out - ref).abs().max()) <= 1e-5
    bad = list(args); bad[0] = z(4); bad[1] = z(4)
    assert my_lib.FilterInterpolationBlendLayer_gpu_forward(*bad, torch.zeros(B, 4, H, W, device=dev())) == -1
    bad = list(args); bad[7] = z(2)
    assert my_lib.FilterInterpolationBlendLayer_gpu_forward(*bad, out) == -1


def _fi(my_lib, x, f, k):
    o = torch.zeros_like(x)
    assert my_lib.FilterInterpolationLayer_gpu_forward(x, f, k, o) == 0
    return o


def test_views_with_huge_row_strides(oracle):
    """The owner-computes projection and the tiled FI backward address a plane with 32-bit byte offsets.  A view
    whose rows are 2^21 elements (8 MiB) apart spans 4.7 GiB per plane at H = 600: the launcher must route it to
    the 64-bit kernels -- same numbers as on a dense copy."""
    from my_package.modules.FlowProjectionModule import FlowProjectionModule  # noqa: F401  (package import check)
    import my_package._ext.my_lib as my_lib
    if torch.cuda.get_device_properties(0).total_memory < 64 << 30:
        pytest.skip("needs ~25 GB of address space on the device")
    H, W, S = 600, 64, 1 << 21
    rng = np.random.default_rng(77)
    fn = synth.np_flow(rng, 1, H, W, "smooth", 4.0)

    def view(c):
        base = torch.zeros(c * H * S, device=dev())
        return base.as_strided((1, c, H, W), (0, H * S, S, 1))
    flow, cnt, out = view(2), view(1), view(2)
    flow.copy_(T(fn))
    assert my_lib.FlowProjectionLayer_gpu_forward(flow, cnt, out, 1) == 0
    want_out, want_cnt = oracle.flow_projection_forward(fn, 1)
    close(N(out.contiguous()), want_out, "projection through a 4.7 GiB-per-plane view")
    close(N(cnt.contiguous()), want_cnt, "count")


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_forward_outputs_need_no_zero_fill(oracle, case):
    """The Python layer hands FilterInterpolation / Interpolation(Ch) forward an UNINITIALISED output (no memset):
    every element must be written by the kernels -- valid sites, invalid sites (copy / zero), every channel, every
    kernel variant the shapes of CASES select.  Checked with a NaN-filled buffer through the C ABI."""
    import my_package._ext.my_lib as my_lib
    d = make(case)
    x, f, k = T(d["x"]), T(d["flow"]), T(d["filt"])
    out = torch.full_like(x, float("nan"))
    assert my_lib.FilterInterpolationLayer_gpu_forward(x, f, k, out) == 0
    assert not torch.isnan(out).any()
    close(N(out), oracle.filter_interpolation_forward(d["x"], d["flow"], d["filt"]), "FI forward into a NaN buffer")
    out = torch.full_like(x, float("nan"))
    assert my_lib.InterpolationChLayer_gpu_forward(x, f, out) == 0
    assert not torch.isnan(out).any()
    close(N(out), oracle.interpolation_ch_forward(d["x"], d["flow"]), "InterpolationCh forward into a NaN buffer")
    for fs in (2, 3):                                            # the any-filter-size kernel
        kk = T(d["filt"][:, :fs * fs])
        out = torch.full_like(x, float("nan"))
        assert my_lib.FilterInterpolationLayer_gpu_forward(x, f, kk, out) == 0
        assert not torch.isnan(out).any()


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_backward_defines_flow_and_tap_gradients(oracle, case):
    """The Python layer hands the FilterInterpolation backward UNINITIALISED gradinput2 / gradinput3: every element
    must be stored by the kernels (invalid sites: zero), whatever kernel the shape selects (tiled RGB, direct,
    any filter size; quads split between bands, sites only the scalar path can reach).  NaN-filled buffers."""
    import my_package._ext.my_lib as my_lib
    d = make(case)
    x, f, k, g = T(d["x"]), T(d["flow"]), T(d["filt"]), T(d["gout"])
    for kk, fs in ((k, 4), (T(d["filt"][:, :9]), 3)):
        g1 = torch.zeros_like(x)
        g2, g3 = torch.full_like(f, float("nan")), torch.full_like(kk, float("nan"))
        assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, kk, g, g1, g2, g3) == 0
        assert not torch.isnan(g2).any() and not torch.isnan(g3).any(), "fs=%d" % fs
        w1, w2, w3 = oracle.filter_interpolation_backward(d["x"], d["flow"], N(kk), d["gout"])
        close(N(g1), w1, "gradinput1 fs=%d" % fs, RTOL)
        close(N(g2), w2, "gradinput2 fs=%d" % fs, RTOL)
        close(N(g3), w3, "gradinput3 fs=%d" % fs, RTOL)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_backward_defines_bilinear_flow_and_projection_gradients(oracle, case):
    """Interpolation(Ch) gradinput2 and both (Depth)FlowProjection backward gradients are handed over UNINITIALISED
    by the Python layers: the kernels (tiled or scalar, whichever the shape selects) must store every element,
    zeros at the sites the reference leaves at the caller's zero.  NaN-filled buffers through the C ABI."""
    import my_package._ext.my_lib as my_lib
    d = make(case)
    x, f, g = T(d["x"]), T(d["flow"]), T(d["gout"])
    pairs = [("InterpolationChLayer_gpu_backward", oracle.interpolation_ch_backward)]
    if x.shape[1] == 3:
        pairs.append(("InterpolationLayer_gpu_backward", oracle.interpolation_backward))
    for name, want in pairs:
        g1, g2 = torch.zeros_like(x), torch.full_like(f, float("nan"))
        assert getattr(my_lib, name)(x, f, g, g1, g2) == 0
        assert not torch.isnan(g2).any(), name
        w1, w2 = want(d["x"], d["flow"], d["gout"])
        close(N(g1), w1, name + " gradinput1", RTOL)
        close(N(g2), w2, name + " gradinput2", RTOL)

    dep, gf = T(d["depth"]), T(d["gflow"])
    _, cnt = oracle.flow_projection_forward(d["flow"], 0)
    g1 = torch.full_like(f, float("nan"))
    assert my_lib.FlowProjectionLayer_gpu_backward(f, T(cnt), gf, g1) == 0
    assert not torch.isnan(g1).any()
    close(N(g1), oracle.flow_projection_backward(d["flow"], cnt, d["gflow"]), "projection gradinput1", RTOL)
    out, cnt = oracle.depth_flow_projection_forward(d["flow"], d["depth"], 0)
    g1, g2 = torch.full_like(f, float("nan")), torch.full_like(dep, float("nan"))
    assert my_lib.DepthFlowProjectionLayer_gpu_backward(f, dep, T(cnt), T(out), gf, g1, g2) == 0
    assert not torch.isnan(g1).any() and not torch.isnan(g2).any()
    w1, w2 = oracle.depth_flow_projection_backward(d["flow"], d["depth"], cnt, out, d["gflow"])
    close(N(g1), w1, "depth projection gradinput1", RTOL)
    close(N(g2), w2, "depth projection gradinput2", RTOL)


@pytest.mark.parametrize("case", CASES, ids=IDS)
def test_projection_forward_needs_no_zero_fill(oracle, case):
    """(Depth)FlowProjection forward DEFINES count and output on every path (owner-computes fast path, the general
    path behind its far flag or on its own, the scalar kernels of odd widths): the Python layer passes
    uninitialised buffers.  NaN-filled buffers through the C ABI, every path forced in turn."""
    from tools import measure as M          # forced paths exist in the measurement build only
    my_lib = M.bound()                      # (the same sources, -DMEMC_MEASURE; my_package stays on the product library)
    d = make(case)
    f, dep = T(d["flow"]), T(d["depth"])
    want = {fh: oracle.flow_projection_forward(d["flow"], fh) for fh in (0, 1)}
    dwant = oracle.depth_flow_projection_forward(d["flow"], d["depth"], 1)
    try:
        # automatic (fast path where it applies), general, scalar; then the owner kernel's geometries (tile height
        # 16 / 32 x walk in strips / stripes 2 / 4 tile columns wide: 10x-11x the production kernel, 40x-41x round 3's
        # production set -- proj_owner4 and the carry filler --, 13x-15x the LDS-ring kernel (tile height up to 64), 16x the
        # persistent one) and the round-1 owner kernel
        # (-43: the product kernel computing its tile coordinates itself, as grids too large for the host-made reciprocals do)
        for variant in (-1, -43, 1, 0, 100, 104, 110, 112, -40, 400, 404, 412, 130, 142, 154, 160, 164, -10):
            M.set_variant("projection", variant)
            for fh in (0, 1):
                cnt = torch.full((f.shape[0], 1, f.shape[2], f.shape[3]), float("nan"), device=dev())
                out = torch.full_like(f, float("nan"))
                assert my_lib.FlowProjectionLayer_gpu_forward(f, cnt, out, fh) == 0
                assert not torch.isnan(cnt).any() and not torch.isnan(out).any(), (variant, fh)
                assert np.array_equal(N(cnt), want[fh][1])
                close(N(out), want[fh][0], "projection variant %d fillhole %d" % (variant, fh))
            cnt, out = torch.full_like(dep, float("nan")), torch.full_like(f, float("nan"))
            assert my_lib.DepthFlowProjectionLayer_gpu_forward(f, dep, cnt, out, 1) == 0
            close(N(cnt), dwant[1], "depth count variant %d" % variant, RTOL)
            close(N(out), dwant[0], "depth out variant %d" % variant)
    finally:
        M.set_variant("projection", -1)


@pytest.mark.parametrize("shift", [(10.3, 0.0), (7.5, -5.2), (-12.0, 9.0), (0.0, 20.0), (40.0, -20.0), (-70.5, 3.0), (26.0, 30.5)])
def test_hole_filling_on_camera_pans(oracle, shift):
    """A pan leaves an uncovered strip along one or two image borders: holes whose walks run the whole length of
    the strip (the carry tables of proj_fillhole_carry), and holes with no neighbour at all in a direction.
    Both hole fillers -- the carry-based one and the literal walker kept for stream captures -- against the oracle.
    Pans of 24 px and more: every source is far, no tile has a near source (proj_owner_far recomputes every tile the pan
    lands in, from the tiles it comes from only)."""
    from tools import measure as M          # forced paths exist in the measurement build only
    my_lib = M.bound()                      # (the same sources, -DMEMC_MEASURE; my_package stays on the product library)
    rng = np.random.default_rng(41)
    B, H, W = 2, 100, 196
    flow = np.empty((B, 2, H, W), np.float32)
    flow[:, 0], flow[:, 1] = shift
    flow += rng.normal(0, 0.05, flow.shape).astype(np.float32)
    want_out, want_cnt = oracle.flow_projection_forward(flow, 1)
    assert (want_cnt == 0).mean() > 0.02                         # the strip is there
    f = T(flow)
    try:
        # mask filler (in the owner's epilogue + proj_fill_pending), literal walker, general path + masks; 16-row bands;
        # round 3's carry filler over 16 / 32 / 64-row bands
        for variant in (-1, -9, 1, 100, 114, -40, 400, 414, 134, 140, 154, 164, -10):
            M.set_variant("projection", variant)
            cnt, out = torch.full((B, 1, H, W), float("nan"), device=dev()), torch.full_like(f, float("nan"))
            assert my_lib.FlowProjectionLayer_gpu_forward(f, cnt, out, 1) == 0
            assert np.array_equal(N(cnt), want_cnt), variant
            close(N(out), want_out, "pan %s variant %d" % (shift, variant))
    finally:
        M.set_variant("projection", -1)


# ------------------------------------------------------------------------------------------------------------
# EXTENSION: frames AND context features warped with one stream of flow + taps per direction, blend included
# (MEMC_Net_star.py:273-285).  Expected values = the oracle composition; the fused results
# must also agree with the separate HIP operators they replace to a few ulps (same arithmetic, other kernels).
# ------------------------------------------------------------------------------------------------------------
CTX_CASES = [
    # (B, Cctx, H, W, flow kind, sigma, seed)
    (2, 8, 40, 64, "smooth", 4.0, 41), (1, 64, 24, 128, "smooth", 4.0, 42), (1, 12, 100, 132, "iid", 3.0, 43),
    (1, 8, 64, 256, "iid", 20.0, 44),           # box far beyond the LDS budget: bands + per-site fallback
    (1, 8, 96, 256, "smooth", 25.0, 45),        # bands, lanes split over bands
    (1, 6, 24, 48, "smooth", 4.0, 46),          # context channels not a multiple of 4: composed path
    (1, 8, 19, 23, "iid", 2.0, 47),             # width not a multiple of 4: composed path
]


@pytest.mark.parametrize("case", CTX_CASES, ids=["%dx%dx%dx%d-%s" % c[:5] for c in CTX_CASES])
def test_filter_interpolation_ctx_blend(oracle, case):
    from my_package.modules.FilterInterpolationCtxBlendModule import FilterInterpolationCtxBlendModule
    from my_package.modules.FilterInterpolationBlendModule import FilterInterpolationBlendModule
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    B, Cc, H, W, kind, sigma, seed = case
    img_case = (B, 3, H, W, kind, sigma, seed)
    a, b2 = make(img_case), make(img_case[:6] + (seed + 100,))
    rng = np.random.default_rng(seed + 7)
    host = dict(x0=a["x"], x2=b2["x"], c0=rng.random((B, Cc, H, W), dtype=np.float32),
                c2=rng.random((B, Cc, H, W), dtype=np.float32), f0=a["flow"], f1=b2["flow"], k0=a["filt"],
                k1=b2["filt"], o0=rng.random((B, 1, H, W), dtype=np.float32), o1=rng.random((B, 1, H, W), dtype=np.float32))
    names = ("x0", "x2", "c0", "c2", "f0", "f1", "k0", "k1", "o0", "o1")
    t = {n: T(host[n], n not in ("c0", "c2")) for n in names}
    blended, c0w, c2w = FilterInterpolationCtxBlendModule()(*[t[n] for n in names])
    assert not c0w.requires_grad and not c2w.requires_grad           # detached, as MEMC_Net_star.py:284-285
    blended.backward(T(a["gout"]))
    w0 = oracle.filter_interpolation_forward(host["x0"], host["f0"], host["k0"])
    w2 = oracle.filter_interpolation_forward(host["x2"], host["f1"], host["k1"])
    close(N(blended), host["o0"] * w0 + host["o1"] * w2, "blended frame")
    close(N(c0w), oracle.filter_interpolation_forward(host["c0"], host["f0"], host["k0"]), "context 0")
    close(N(c2w), oracle.filter_interpolation_forward(host["c2"], host["f1"], host["k1"]), "context 2")
    gout = a["gout"]
    for x, f, k, o, w in (("x0", "f0", "k0", "o0", w0), ("x2", "f1", "k1", "o1", w2)):
        g1, g2, g3 = oracle.filter_interpolation_backward(host[x], host[f], host[k], (gout * host[o]).astype(np.float32))
        close(N(t[x].grad), g1, "grad " + x, RTOL)
        close(N(t[f].grad), g2, "grad " + f, RTOL)
        close(N(t[k].grad), g3, "grad " + k, RTOL)
        close(N(t[o].grad), (gout * w).sum(axis=1, keepdims=True), "grad " + o, RTOL)
    # bit-identical to the operators it replaces
    with torch.no_grad():
        d = {n: t[n].detach() for n in names}
        ref_b = FilterInterpolationBlendModule()(d["x0"], d["x2"], d["f0"], d["f1"], d["k0"], d["k1"], d["o0"], d["o1"])
        warp = FilterInterpolationModule()
        # (the same expressions compiled into different kernels: fused multiply-adds may contract differently)
        assert float((blended.detach() - ref_b).abs().max()) <= 2e-6
        assert float((c0w - warp(d["c0"], d["f0"], d["k0"])).abs().max()) <= 2e-6
        assert float((c2w - warp(d["c2"], d["f1"], d["k1"])).abs().max()) <= 2e-6


def test_filter_interpolation_ctx_c_abi_checks():
    """the C entry point: outputs need no zero fill; prev / occlusions all given or all NULL; rejects what the kernel
    does not cover (the Python layer then composes the result)"""
    import my_package._ext.my_lib as my_lib
    B, H, W = 1, 32, 64
    z = lambda c: torch.rand(B, c, H, W, device=dev())          # noqa: E731
    img, ctxf, flow, filt = z(3), z(8), z(2) * 4 - 2, z(16) / 16
    io, co = torch.full((B, 3, H, W), float("nan"), device=dev()), torch.full((B, 8, H, W), float("nan"), device=dev())
    assert my_lib.FilterInterpolationCtxLayer_gpu_forward(img, ctxf, flow, filt, None, None, None, io, co) == 0
    assert float((io - _fi(my_lib, img, flow, filt)).abs().max()) <= 2e-6
    assert float((co - _fi(my_lib, ctxf, flow, filt)).abs().max()) <= 2e-6
    prev, oa, ob = z(3), z(1), z(1)
    io2 = torch.full_like(io, float("nan"))
    assert my_lib.FilterInterpolationCtxLayer_gpu_forward(img, ctxf, flow, filt, prev, oa, ob, io2, co) == 0
    assert float((io2 - (oa * prev + ob * io)).abs().max()) <= 1e-6
    assert my_lib.FilterInterpolationCtxLayer_gpu_forward(img, ctxf, flow, filt, prev, None, ob, io2, co) == -1
    assert my_lib.FilterInterpolationCtxLayer_gpu_forward(img, z(6), flow, filt, None, None, None, io,
                                                          torch.zeros(B, 6, H, W, device=dev())) == -1
    assert my_lib.FilterInterpolationCtxLayer_gpu_forward(z(4), ctxf, flow, filt, None, None, None,
                                                          torch.zeros(B, 4, H, W, device=dev()), co) == -1


# ------------------------------------------------------------------------------------------------------------
# EXTENSION: FlowProjection's prologue in the networks -- (mul * flow) / div, x4 bilinear upsampling -- as one
# kernel (MEMC_Net_star.py:172-176).  Expected value = the torch expression it replaces.
# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("align", [False, True])
@pytest.mark.parametrize("shape", [(2, 2, 9, 13), (1, 2, 48, 80), (3, 2, 1, 1), (1, 5, 32, 17)])
def test_flow_upsample4(shape, align):
    import torch.nn.functional as F
    from my_package.modules.FlowUpsample4Module import FlowUpsample4Module
    g = torch.Generator(device=dev()); g.manual_seed(sum(shape) + int(align))
    flow = torch.randn(shape, device=dev(), generator=g) * 0.7
    fr = flow.clone().requires_grad_(True)
    ft = flow.clone().requires_grad_(True)
    got = FlowUpsample4Module(20, 2.0, align)(fr)
    want = F.interpolate(20 * ft / 2.0, scale_factor=4, mode="bilinear", align_corners=align)
    assert got.shape == want.shape
    # (align_corners: src = dst * (in - 1) / (out - 1) lands within an ulp of an integer now and then; the bilinear
    # form is continuous there, the two evaluations differ by rounding only)
    err = float((got - want).abs().max())
    assert err <= 2e-5 * max(1.0, float(want.abs().max())), err
    gout = torch.rand(want.shape, device=dev(), generator=g)
    got.backward(gout); want.backward(gout)
    assert float((fr.grad - ft.grad).abs().max()) <= 1e-5 * max(1.0, float(ft.grad.abs().max()))


def _random_shape_cases(n, seed):
    """Seeded random shapes: batch 1-3, channels 1-9 (+ 16 / 64 now and then), heights 1-90, widths 1-260 of every residue
    mod 4, a flow kind and scale per case -- every operator sees widths below one quad, ragged widths, partial tiles in both
    directions, one-pixel images and motion larger than the image."""
    rng = np.random.default_rng(seed)
    cases = []
    big = os.environ.get("MEMC_RANDOM_BIG", "") not in ("", "0")
    for i in range(n if big else 0):
        # MEMC_RANDOM_BIG=1 (sweeps run once, tools/sessions/r06_s36.sh): ANY height up to 420 and width up to 900, batch up to 5 -- tens
        # of tile rows and columns, every remainder of the strip count modulo the eight XCDs (the walk's two XCD classes)
        kind = str(rng.choice(["smooth", "iid", "iid", "zero", "pan", "pan"]))
        cases.append((int(rng.integers(1, 6)), int(rng.choice([1, 2, 3, 3, 3, 4, 5, 8])), int(rng.integers(1, 421)),
                      int(rng.integers(1, 901)), kind, float(rng.choice([0.5, 2.0, 5.0, 15.0, 40.0])), 5000 + i))
    for i in range(0 if big else n):
        B = int(rng.integers(1, 4))
        C = int(rng.choice([1, 2, 3, 3, 3, 4, 5, 7, 8, 9, 16, 64], p=None))
        H = int(rng.choice([1, 2, 5, 16, 17, 31, 33, 48, 64, 90]))
        W = int(rng.choice([1, 3, 4, 7, 8, 9, 13, 30, 63, 64, 65, 66, 67, 100, 129, 130, 131, 200, 258, 260]))
        if C >= 16:
            H, W = min(H, 40), min(W, 140)                 # (keep the oracle's time per case in milliseconds)
        kind = str(rng.choice(["smooth", "iid", "iid", "zero", "pan", "pan"]))
        sigma = float(rng.choice([0.5, 2.0, 5.0, 15.0, 40.0]))
        cases.append((B, C, H, W, kind, sigma, 1000 + i))
    return cases


def _make_random(case):
    """make(), plus the flow kind "pan": a smooth field on top of a whole-image motion of up to +-0.75 of the image's size per
    axis, another one per image of the batch (the projection's scan follows it, far sources, sites leaving the image)."""
    B, C, H, W, kind, sigma, seed = case
    if kind != "pan":
        return make(case)
    d = make((B, C, H, W, "smooth", sigma, seed))
    rng = np.random.default_rng(seed + 77)
    for b in range(B):
        d["flow"][b, 0] += np.float32(rng.uniform(-0.75, 0.75) * W)
        d["flow"][b, 1] += np.float32(rng.uniform(-0.75, 0.75) * H)
    return d


# (64 cases in the suite; MEMC_RANDOM_CASES=N for a longer sweep -- round 6 ran 600 once: tools/sessions/r06_s13.sh)
# (MEMC_RANDOM_SEED: another seed for such a sweep -- tools/sessions/r06_s31.sh ran 1500 + 600 fresh cases once)
RANDOM_CASES = _random_shape_cases(int(os.environ.get("MEMC_RANDOM_CASES", "64")), int(os.environ.get("MEMC_RANDOM_SEED", "20260601")))
_N_STRIDED = int(os.environ.get("MEMC_STRIDED_CASES", "48"))


@pytest.mark.parametrize("case", RANDOM_CASES, ids=["%dx%dx%dx%d-%s-%g" % c[:6] for c in RANDOM_CASES])
def test_random_shapes_every_operator_forward_and_backward(oracle, case):
    """Round 6: a seeded sweep of random shapes through EVERY operator of the path, forward and backward, through the C ABI,
    against the oracle -- the fixed case lists above were written around the kernels' known edges; this one is not.  The
    reference serves every shape with one kernel per operator (my_lib_kernel.cu:10-15)."""
    import my_package._ext.my_lib as my_lib
    B, C, H, W, kind, sigma, seed = case
    d = _make_random(case)
    x, f, k, g, dep, gf = (T(d[n]) for n in ("x", "flow", "filt", "gout", "depth", "gflow"))
    junk = lambda *shape: torch.full(shape, 7.0, device=dev())     # noqa: E731 -- outputs are DEFINED by the calls
    # FilterInterpolation
    out = junk(B, C, H, W)
    assert my_lib.FilterInterpolationLayer_gpu_forward(x, f, k, out) == 0
    close(N(out), oracle.filter_interpolation_forward(d["x"], d["flow"], d["filt"]), "FI fwd")
    g1 = junk(B, C, H, W) if my_lib.gradinput1_is_stored(4, C) else torch.zeros(B, C, H, W, device=dev())
    g2, g3 = junk(B, 2, H, W), junk(B, 16, H, W)
    assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
    w1, w2, w3 = oracle.filter_interpolation_backward(d["x"], d["flow"], d["filt"], d["gout"])
    close(N(g1), w1, "FI bwd gradinput1", 3 * RTOL)
    close(N(g2), w2, "FI bwd gradinput2")
    close(N(g3), w3, "FI bwd gradinput3")
    # Interpolation (C == 3 only, as the reference) / InterpolationCh
    fwd, bwd = ((my_lib.InterpolationLayer_gpu_forward, my_lib.InterpolationLayer_gpu_backward) if C == 3 else
                (my_lib.InterpolationChLayer_gpu_forward, my_lib.InterpolationChLayer_gpu_backward))
    out = junk(B, C, H, W)
    assert fwd(x, f, out) == 0
    close(N(out), oracle.interpolation_ch_forward(d["x"], d["flow"]), "Interpolation fwd")
    g1 = junk(B, C, H, W) if my_lib.gradinput1_is_stored(0, C) else torch.zeros(B, C, H, W, device=dev())
    g2 = junk(B, 2, H, W)
    assert bwd(x, f, g, g1, g2) == 0
    w1, w2 = oracle.interpolation_ch_backward(d["x"], d["flow"], d["gout"])
    close(N(g1), w1, "Interpolation bwd gradinput1", 3 * RTOL)
    close(N(g2), w2, "Interpolation bwd gradinput2")
    # FlowProjection / DepthFlowProjection, with and without hole filling, and their backward passes
    for fill in (0, 1):
        cnt, po = junk(B, 1, H, W), junk(B, 2, H, W)
        assert my_lib.FlowProjectionLayer_gpu_forward(f, cnt, po, fill) == 0
        want_out, want_cnt = oracle.flow_projection_forward(d["flow"], fill)
        assert np.array_equal(N(cnt), want_cnt), "FlowProjection count, fill %d" % fill
        close(N(po), want_out, "FlowProjection fwd, fill %d" % fill)
        dcnt, dpo = junk(B, 1, H, W), junk(B, 2, H, W)
        assert my_lib.DepthFlowProjectionLayer_gpu_forward(f, dep, dcnt, dpo, fill) == 0
        want_dout, want_dcnt = oracle.depth_flow_projection_forward(d["flow"], d["depth"], fill)
        close(N(dcnt), want_dcnt, "DepthFlowProjection count, fill %d" % fill)
        close(N(dpo), want_dout, "DepthFlowProjection fwd, fill %d" % fill)
        if fill == 0:                                          # (backward sees the planes of a forward without hole filling)
            gin = junk(B, 2, H, W)
            assert my_lib.FlowProjectionLayer_gpu_backward(f, T(want_cnt), gf, gin) == 0
            close(N(gin), oracle.flow_projection_backward(d["flow"], want_cnt, d["gflow"]), "FlowProjection bwd")
            gin, gd = junk(B, 2, H, W), junk(B, 1, H, W)
            assert my_lib.DepthFlowProjectionLayer_gpu_backward(f, dep, T(want_dcnt), T(want_dout), gf, gin, gd) == 0
            wg1, wg2 = oracle.depth_flow_projection_backward(d["flow"], d["depth"], want_dcnt, want_dout, d["gflow"])
            close(N(gin), wg1, "DepthFlowProjection bwd gradinput1")
            close(N(gd), wg2, "DepthFlowProjection bwd gradinput2", cancel=P.CANCEL)


@pytest.mark.parametrize("case", RANDOM_CASES[:_N_STRIDED], ids=["%dx%dx%dx%d-%s-%g" % c[:6] for c in RANDOM_CASES[:_N_STRIDED]])
def test_random_strided_views_every_operator(oracle, case):
    """The same sweep on VIEWS: every tensor is a window of a larger buffer -- random extra channels, rows and columns around it and a
    random offset inside, one layout per tensor shape (image-like, flow-like, filter-like, one-plane: tensors of one shape
    share their strides, which is what the reference's checks ask of some pairs, my_lib_cuda.c:611-646) -- so batch, channel
    and row strides are all unrelated to the sizes and every row starts at some dword.  What surrounds a window must not change."""
    import my_package._ext.my_lib as my_lib
    B, C, H, W, kind, sigma, seed = case
    d = _make_random(case)
    rng = np.random.default_rng(seed + 4242)
    layouts = {}

    def window(channels, fill=None, src=None):
        """A [B, channels, H, W] view inside a buffer with this shape class's margins; returns (view, buffer)."""
        if channels not in layouts:
            layouts[channels] = [int(v) for v in rng.integers(0, 4, size=6)]           # channel / row / column margins, before + after
        c0, c1, h0, h1, w0, w1 = layouts[channels]
        buf = torch.full((B, channels + c0 + c1, H + h0 + h1, W + w0 + w1), -3.0, device=dev())
        view = buf[:, c0:c0 + channels, h0:h0 + H, w0:w0 + W]
        if src is not None:
            view.copy_(T(src))
        elif fill is not None:
            view.fill_(fill)
        return view, buf

    def untouched(buf, channels):
        c0, c1, h0, h1, w0, w1 = layouts[channels]
        inner = torch.zeros_like(buf, dtype=torch.bool)
        inner[:, c0:c0 + channels, h0:h0 + H, w0:w0 + W] = True
        return bool((buf[~inner] == -3.0).all())

    x, _ = window(C, src=d["x"])
    g, _ = window(C, src=d["gout"])
    f, _ = window(2, src=d["flow"])
    gf, _ = window(2, src=d["gflow"])
    k, _ = window(16, src=d["filt"])
    dep, _ = window(1, src=d["depth"])
    out, out_b = window(C, fill=7.0)
    assert my_lib.FilterInterpolationLayer_gpu_forward(x, f, k, out) == 0
    close(N(out), oracle.filter_interpolation_forward(d["x"], d["flow"], d["filt"]), "FI fwd on views")
    g1, g1_b = window(C, fill=7.0 if my_lib.gradinput1_is_stored(4, C) else 0.0)
    g2, g2_b = window(2, fill=7.0)
    g3, g3_b = window(16, fill=7.0)
    assert my_lib.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
    w1, w2, w3 = oracle.filter_interpolation_backward(d["x"], d["flow"], d["filt"], d["gout"])
    close(N(g1), w1, "FI bwd gradinput1 on views", 3 * RTOL)
    close(N(g2), w2, "FI bwd gradinput2 on views")
    close(N(g3), w3, "FI bwd gradinput3 on views")
    assert untouched(out_b, C) and untouched(g1_b, C) and untouched(g2_b, 2) and untouched(g3_b, 16)
    fwd, bwd = ((my_lib.InterpolationLayer_gpu_forward, my_lib.InterpolationLayer_gpu_backward) if C == 3 else
                (my_lib.InterpolationChLayer_gpu_forward, my_lib.InterpolationChLayer_gpu_backward))
    out, out_b = window(C, fill=7.0)
    assert fwd(x, f, out) == 0
    close(N(out), oracle.interpolation_ch_forward(d["x"], d["flow"]), "Interpolation fwd on views")
    g1, g1_b = window(C, fill=7.0 if my_lib.gradinput1_is_stored(0, C) else 0.0)
    g2, g2_b = window(2, fill=7.0)
    assert bwd(x, f, g, g1, g2) == 0
    w1, w2 = oracle.interpolation_ch_backward(d["x"], d["flow"], d["gout"])
    close(N(g1), w1, "Interpolation bwd gradinput1 on views", 3 * RTOL)
    close(N(g2), w2, "Interpolation bwd gradinput2 on views")
    assert untouched(out_b, C) and untouched(g1_b, C) and untouched(g2_b, 2)
    for fill in (0, 1):
        cnt, cnt_b = window(1, fill=7.0)
        po, po_b = window(2, fill=7.0)
        assert my_lib.FlowProjectionLayer_gpu_forward(f, cnt, po, fill) == 0
        want_out, want_cnt = oracle.flow_projection_forward(d["flow"], fill)
        assert np.array_equal(N(cnt), want_cnt), "FlowProjection count on views, fill %d" % fill
        close(N(po), want_out, "FlowProjection fwd on views, fill %d" % fill)
        dcnt, dcnt_b = window(1, fill=7.0)
        dpo, dpo_b = window(2, fill=7.0)
        assert my_lib.DepthFlowProjectionLayer_gpu_forward(f, dep, dcnt, dpo, fill) == 0
        want_dout, want_dcnt = oracle.depth_flow_projection_forward(d["flow"], d["depth"], fill)
        close(N(dcnt), want_dcnt, "DepthFlowProjection count on views, fill %d" % fill)
        close(N(dpo), want_dout, "DepthFlowProjection fwd on views, fill %d" % fill)
        assert untouched(cnt_b, 1) and untouched(po_b, 2) and untouched(dcnt_b, 1) and untouched(dpo_b, 2)
        if fill == 0:
            gin, gin_b = window(2, fill=7.0)
            assert my_lib.FlowProjectionLayer_gpu_backward(f, cnt, gf, gin) == 0
            close(N(gin), oracle.flow_projection_backward(d["flow"], want_cnt, d["gflow"]), "FlowProjection bwd on views")
            gin2, gin2_b = window(2, fill=7.0)
            gd, gd_b = window(1, fill=7.0)
            # (the SAME forward planes on both sides, as in the contiguous sweep: the library's own differ from the oracle's in the last
            # bit or two, and gradinput2 multiplies that by terms of 1e3 under +-160 px of flow)
            dcnt_in, _ = window(1, src=want_dcnt)
            dpo_in, _ = window(2, src=want_dout)
            assert my_lib.DepthFlowProjectionLayer_gpu_backward(f, dep, dcnt_in, dpo_in, gf, gin2, gd) == 0
            wg1, wg2 = oracle.depth_flow_projection_backward(d["flow"], d["depth"], want_dcnt, want_dout, d["gflow"])
            close(N(gin2), wg1, "DepthFlowProjection bwd gradinput1 on views")
            close(N(gd), wg2, "DepthFlowProjection bwd gradinput2 on views", cancel=P.CANCEL)
            assert untouched(gin_b, 2) and untouched(gin2_b, 2) and untouched(gd_b, 1)
