"""Inputs shared by the reference-kernel tests and the fixture generator (tests/golden/make_golden_ref_gpu.py)."""
import numpy as np

from tools import synth

# (B, C, H, W, flow kind, sigma, seed)
REF_CASES = [
    (1, 3, 32, 48, "smooth", 4.0, 101),
    (2, 3, 19, 23, "iid", 3.0, 102),       # odd sizes, many holes, invalid sites at the borders
    (1, 3, 40, 64, "iid", 9.0, 103),       # large motion
    (1, 5, 16, 24, "smooth", 4.0, 104),    # not RGB (InterpolationCh, FI with C = 5)
    (1, 3, 48, 136, "smooth", 30.0, 105),  # motion of up to ~80 px over 3 x 2 projection tiles: sources far beyond the owner kernel's reach
]


def make(case):
    B, C, H, W, kind, sigma, seed = case
    rng = np.random.default_rng(seed)
    return dict(x=synth.np_image(rng, B, C, H, W), flow=synth.np_flow(rng, B, H, W, kind, sigma),
                filt=synth.np_filter(rng, B, H, W), gout=synth.np_image(rng, B, C, H, W),
                depth=synth.np_depth(rng, B, H, W), gflow=rng.random((B, 2, H, W), dtype=np.float32))


def name(case):
    return "%dx%dx%dx%d_%s" % case[:5]


def reference_outputs(R, d, T, N):
    """Every reference entry point of the path on the inputs `d`; R = oracle.ref_gpu, T/N = to/from device."""
    x, f, k, g, dep, gf = (T(d[n]) for n in ("x", "flow", "filt", "gout", "depth", "gflow"))
    out = {}
    out["fi_fwd"] = N(R.filter_interpolation_forward(x, f, k))
    out["fi_g1"], out["fi_g2"], out["fi_g3"] = (N(t) for t in R.filter_interpolation_backward(x, f, k, g))
    out["blch_fwd"] = N(R.interpolation_forward(x, f, ch=True))
    out["blch_g1"], out["blch_g2"] = (N(t) for t in R.interpolation_backward(x, f, g, ch=True))
    for fh in (0, 1):
        o, c = R.flow_projection_forward(f, fh)
        out["fp_out%d" % fh], out["fp_cnt%d" % fh] = N(o), N(c)
        o, c = R.depth_flow_projection_forward(f, dep, fh)
        out["dfp_out%d" % fh], out["dfp_cnt%d" % fh] = N(o), N(c)
    o0, c0 = R.flow_projection_forward(f, 0)
    out["fp_g1"] = N(R.flow_projection_backward(f, c0, gf))
    o0, c0 = R.depth_flow_projection_forward(f, dep, 0)
    g1, g2 = R.depth_flow_projection_backward(f, dep, c0, o0, gf)
    out["dfp_g1"], out["dfp_g2"] = N(g1), N(g2)
    return out


def oracle_outputs(O, d):
    """The same quantities from the CPU oracle."""
    x, f, k, g, dep, gf = (d[n] for n in ("x", "flow", "filt", "gout", "depth", "gflow"))
    out = {}
    out["fi_fwd"] = O.filter_interpolation_forward(x, f, k)
    out["fi_g1"], out["fi_g2"], out["fi_g3"] = O.filter_interpolation_backward(x, f, k, g)
    out["blch_fwd"] = O.interpolation_ch_forward(x, f)
    out["blch_g1"], out["blch_g2"] = O.interpolation_ch_backward(x, f, g)
    for fh in (0, 1):
        out["fp_out%d" % fh], out["fp_cnt%d" % fh] = O.flow_projection_forward(f, fh)
        out["dfp_out%d" % fh], out["dfp_cnt%d" % fh] = O.depth_flow_projection_forward(f, dep, fh)
    o0, c0 = O.flow_projection_forward(f, 0)
    out["fp_g1"] = O.flow_projection_backward(f, c0, gf)
    o0, c0 = O.depth_flow_projection_forward(f, dep, 0)
    out["dfp_g1"], out["dfp_g2"] = O.depth_flow_projection_backward(f, dep, c0, o0, gf)
    return out
