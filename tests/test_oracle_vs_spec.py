"""The C oracle against the independent pure-Python restatement of SURVEY.md appendix A (tests/_spec_py.py), on small
inputs that exercise every branch: image borders (clamped windows, duplicate adds), invalid sites (passthrough /
zero / skipped), the |f| < W/2 guard, holes and their filling, depth weighting."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _spec_py as S     # noqa: E402

CASES = [(1, 3, 9, 11, 2.0, 1), (2, 2, 7, 8, 4.0, 2), (1, 1, 5, 6, 0.4, 3), (1, 3, 6, 5, 6.0, 4)]


def _inputs(case):
    B, C, H, W, sigma, seed = case
    rng = np.random.default_rng(seed)
    x = rng.random((B, C, H, W), dtype=np.float32)
    flow = rng.normal(0, sigma, (B, 2, H, W)).astype(np.float32)
    flow[:, :, 0, 0] = 0.0                      # exact integer hits
    flow[:, 0, -1, -1] = 0.5                    # lands beyond the last column: invalid in A.1/A.3, valid-less in A.5
    filt = rng.random((B, 16, H, W), dtype=np.float32)
    depth = (rng.random((B, 1, H, W), dtype=np.float32) + np.float32(0.1)).astype(np.float32)
    return x, flow, filt, depth


def _close(got, want, what):
    err = float(np.abs(got - want).max()) if got.size else 0.0
    assert err <= 2e-6 * max(1.0, float(np.abs(want).max())), (what, err)


@pytest.mark.parametrize("case", CASES)
def test_filter_interpolation_forward(oracle, case):
    x, flow, filt, _ = _inputs(case)
    _close(oracle.filter_interpolation_forward(x, flow, filt), S.fi_forward(x, flow, filt), "A.1")
    k9 = filt[:, :9]                             # fs = 3: odd window [ix, ix + 2]
    _close(oracle.filter_interpolation_forward(x, flow, k9), S.fi_forward(x, flow, k9), "A.1 fs=3")


@pytest.mark.parametrize("case", CASES)
def test_projection_forward_and_fill(oracle, case):
    _, flow, _, depth = _inputs(case)
    out, count = oracle.flow_projection_forward(flow, 0)
    sout, scount = S.flow_projection_forward(flow)
    _close(count, scount, "A.3 count")
    _close(out, sout, "A.3 out")
    filled, _ = oracle.flow_projection_forward(flow, 1)
    _close(filled, S.fill_holes(sout, scount), "A.3 fill-hole")
    dout, dcount = oracle.depth_flow_projection_forward(flow, depth, 0)
    sdout, sdcount = S.flow_projection_forward(flow, depth)
    _close(dcount, sdcount, "A.4 count")
    _close(dout, sdout, "A.4 out")


@pytest.mark.parametrize("case", CASES)
def test_interpolation_forward(oracle, case):
    x, flow, _, _ = _inputs(case)
    if x.shape[1] == 3:
        _close(oracle.interpolation_forward(x, flow), S.interpolation_forward(x, flow), "A.5")
    _close(oracle.interpolation_ch_forward(x, flow), S.interpolation_forward(x, flow), "A.5 any C")
