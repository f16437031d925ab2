"""Generates the committed fixtures tests/golden/*.npz.

PROVENANCE: these vectors are produced by the repo's own CPU checker (oracle/memc_oracle.c), NOT by the
reference -- no admissible reference build or reference test vector exists for this path (DESIGN.md
"Oracle").  They freeze the checker's behaviour (so an accidental edit of the oracle is caught) and let the
GPU parity tests compare against stored bytes as well as against a live oracle run.

    python tests/golden/make_golden.py        # rewrites the .npz files in place
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import memc_oracle as O   # noqa: E402
from tools import synth               # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def case(name, seed, B, C, H, W, flow_kind, sigma, fs=4):
    rng = np.random.default_rng(seed)
    x = synth.np_image(rng, B, C, H, W)
    flow = synth.np_flow(rng, B, H, W, flow_kind, sigma)
    filt = synth.np_filter(rng, B, H, W, fs)
    gout = synth.np_image(rng, B, C, H, W)
    depth = synth.np_depth(rng, B, H, W)
    gflow = rng.random((B, 2, H, W), dtype=np.float32)
    d = dict(x=x, flow=flow, filt=filt, gout=gout, depth=depth, gflow=gflow)
    d["fi_out"] = O.filter_interpolation_forward(x, flow, filt)
    d["fi_g1"], d["fi_g2"], d["fi_g3"] = O.filter_interpolation_backward(x, flow, filt, gout)
    d["ich_out"] = O.interpolation_ch_forward(x, flow)
    d["ich_g1"], d["ich_g2"] = O.interpolation_ch_backward(x, flow, gout)
    for fh in (0, 1):
        d["fp_out%d" % fh], d["fp_count"] = O.flow_projection_forward(flow, fh)
        d["dfp_out%d" % fh], d["dfp_count"] = O.depth_flow_projection_forward(flow, depth, fh)
    # backward only ever divides by the count of cells the same source site hit in forward (>= 1 hit)
    d["fp_g1"] = O.flow_projection_backward(flow, d["fp_count"], gflow)
    d["dfp_g1"], d["dfp_g2"] = O.depth_flow_projection_backward(flow, depth, d["dfp_count"], d["dfp_out0"], gflow)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
    print(name, {k: v.shape for k, v in d.items() if k in ("x", "filt")})


def config1():
    """BASELINE.json configs[0]: FilterInterpolation forward, 1x3x128x128, random input / N(0,3^2) flow /
    random 4x4 filters -- the tensors of SURVEY.md A.7's last known answer (default_rng(0), same draw order)."""
    rng = np.random.default_rng(0)
    x = rng.random((1, 3, 128, 128)).astype(np.float32)
    flow = rng.normal(0, 3, (1, 2, 128, 128)).astype(np.float32)
    filt = rng.random((1, 16, 128, 128)).astype(np.float32)
    out = O.filter_interpolation_forward(x, flow, filt)
    pout, pcount = O.flow_projection_forward(flow, 1)
    np.savez_compressed(os.path.join(HERE, "config1_fi_fwd_128.npz"), x=x, flow=flow, filt=filt, fi_out=out,
                        fp_out1=pout, fp_count=pcount)
    print("config1 sum(out) = %.3f (SURVEY A.7: 48114.199), holes = %d (A.7: 452)"
          % (float(out.sum()), int((pcount == 0).sum())))


if __name__ == "__main__":
    O.build()
    case("small_iid_2x3x19x23", 11, 2, 3, 19, 23, "iid", 2.5)
    case("small_smooth_1x3x32x48", 12, 1, 3, 32, 48, "smooth", 4.0)
    case("small_c5_1x5x16x16", 13, 1, 5, 16, 16, "iid", 1.5)
    config1()
