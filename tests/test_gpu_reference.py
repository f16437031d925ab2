"""The reference's OWN GPU kernels (my_package/src/my_lib_kernel.cu, built for gfx950 by `make -C oracle ref`, see
oracle/ref_gpu.py) run on this GPU, against
  (a) the CPU oracle  -- this is what pins the oracle: same inputs, every entry point of the path, forward and
      backward, with and without the hole-filling pass that only the GPU file has;
  (b) the HIP path    -- the north star's own criterion, "outputs that match the reference CUDA kernels within 1e-4".
Skipped when oracle/_ref/libmemc_ref_gpu.so was not built (it is built wherever /root/reference exists and travels to
the GPU box as a binary)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _refcases as RC                     # noqa: E402
from oracle import ref_gpu as R            # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not R.available(), reason="oracle/_ref/libmemc_ref_gpu.so not built")]
import _parity as P                        # noqa: E402
ATOL, RTOL = P.ATOL, P.RTOL                # the rule and the record of observed errors: tests/_parity.py


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def N(t):
    return t.detach().cpu().numpy()


def close(got, want, what, rtol=RTOL):
    return P.close(got, want, what, rtol)


@pytest.mark.parametrize("case", RC.REF_CASES, ids=[RC.name(c) for c in RC.REF_CASES])
def test_oracle_matches_reference_kernels(oracle, case):
    d = RC.make(case)
    want = RC.reference_outputs(R, d, T, N)
    got = RC.oracle_outputs(oracle, d)
    assert sorted(got) == sorted(want)
    for k in sorted(want):
        if "cnt" in k and not k.startswith("dfp"):
            assert np.array_equal(got[k], want[k]), k          # integer-valued counts: bit for bit
        else:
            close(got[k], want[k], "oracle vs reference kernel: " + k)


@pytest.mark.parametrize("case", RC.REF_CASES, ids=[RC.name(c) for c in RC.REF_CASES])
def test_hip_path_matches_reference_kernels(case):
    import my_package._ext.my_lib as L
    d = RC.make(case)
    want = RC.reference_outputs(R, d, T, N)
    x, f, k, g, dep, gf = (T(d[n]) for n in ("x", "flow", "filt", "gout", "depth", "gflow"))
    z = torch.zeros_like
    out = z(x); assert L.FilterInterpolationLayer_gpu_forward(x, f, k, out) == 0
    close(N(out), want["fi_fwd"], "fi_fwd")
    g1, g2, g3 = z(x), z(f), z(k)
    assert L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
    close(N(g1), want["fi_g1"], "fi_g1"); close(N(g2), want["fi_g2"], "fi_g2"); close(N(g3), want["fi_g3"], "fi_g3")
    out = z(x); assert L.InterpolationChLayer_gpu_forward(x, f, out) == 0
    close(N(out), want["blch_fwd"], "blch_fwd")
    g1, g2 = z(x), z(f); assert L.InterpolationChLayer_gpu_backward(x, f, g, g1, g2) == 0
    close(N(g1), want["blch_g1"], "blch_g1"); close(N(g2), want["blch_g2"], "blch_g2")
    for fh in (0, 1):
        cnt, out = z(dep), z(f); assert L.FlowProjectionLayer_gpu_forward(f, cnt, out, fh) == 0
        assert np.array_equal(N(cnt), want["fp_cnt%d" % fh]); close(N(out), want["fp_out%d" % fh], "fp_out%d" % fh)
        cnt, out = z(dep), z(f); assert L.DepthFlowProjectionLayer_gpu_forward(f, dep, cnt, out, fh) == 0
        close(N(cnt), want["dfp_cnt%d" % fh], "dfp_cnt"); close(N(out), want["dfp_out%d" % fh], "dfp_out%d" % fh)
    cnt, out = z(dep), z(f); assert L.FlowProjectionLayer_gpu_forward(f, cnt, out, 0) == 0
    g1 = z(f); assert L.FlowProjectionLayer_gpu_backward(f, cnt, gf, g1) == 0
    close(N(g1), want["fp_g1"], "fp_g1")
    cnt, out = z(dep), z(f); assert L.DepthFlowProjectionLayer_gpu_forward(f, dep, cnt, out, 0) == 0
    g1, g2 = z(f), z(dep); assert L.DepthFlowProjectionLayer_gpu_backward(f, dep, cnt, out, gf, g1, g2) == 0
    close(N(g1), want["dfp_g1"], "dfp_g1"); close(N(g2), want["dfp_g2"], "dfp_g2")


def test_interpolation_rgb_entry_point():
    """InterpolationLayer (C == 3 only) of the reference against the HIP path."""
    import my_package._ext.my_lib as L
    d = RC.make(RC.REF_CASES[0])
    x, f, g = T(d["x"]), T(d["flow"]), T(d["gout"])
    out = torch.zeros_like(x); assert L.InterpolationLayer_gpu_forward(x, f, out) == 0
    close(N(out), N(R.interpolation_forward(x, f)), "bl_fwd")
    g1, g2 = torch.zeros_like(x), torch.zeros_like(f)
    assert L.InterpolationLayer_gpu_backward(x, f, g, g1, g2) == 0
    w1, w2 = R.interpolation_backward(x, f, g)
    close(N(g1), N(w1), "bl_g1"); close(N(g2), N(w2), "bl_g2")


@pytest.mark.parametrize("kind", ["smooth", "iid"])
def test_benchmark_size_against_reference_kernels(kind):
    """BASELINE's 1280x720 frames (batch 2, device-generated like bench.py): the HIP path against the reference
    kernels themselves -- real parity at full size, where the CPU oracle would take minutes."""
    import my_package._ext.my_lib as L
    from tools import synth
    B, C, H, W = 2, 3, 720, 1280
    t = synth.torch_inputs(torch.device("cuda:0"), B, C, H, W, flow_kind=kind, seed=5, with_grad=True, with_depth=True)
    x, f, k, g, dep = t["x"], t["flow"], t["filt"], t["gout"], t["depth"]
    z = torch.zeros_like
    out = z(x); assert L.FilterInterpolationLayer_gpu_forward(x, f, k, out) == 0
    close(N(out), N(R.filter_interpolation_forward(x, f, k)), "fi_fwd 720p")
    g1, g2, g3 = z(x), z(f), z(k)
    assert L.FilterInterpolationLayer_gpu_backward(x, f, k, g, g1, g2, g3) == 0
    w1, w2, w3 = R.filter_interpolation_backward(x, f, k, g)
    close(N(g1), N(w1), "fi_g1 720p"); close(N(g2), N(w2), "fi_g2 720p"); close(N(g3), N(w3), "fi_g3 720p")
    cnt, o = z(dep), z(f); assert L.FlowProjectionLayer_gpu_forward(f, cnt, o, 1) == 0
    wo, wc = R.flow_projection_forward(f, 1)
    assert torch.equal(cnt, wc); close(N(o), N(wo), "fp + fill 720p")
    cnt, o = z(dep), z(f); assert L.DepthFlowProjectionLayer_gpu_forward(f, dep, cnt, o, 1) == 0
    wo, wc = R.depth_flow_projection_forward(f, dep, 1)
    close(N(cnt), N(wc), "dfp count 720p"); close(N(o), N(wo), "dfp + fill 720p")
    gf = torch.rand_like(f)
    cnt0, o0 = z(dep), z(f); assert L.DepthFlowProjectionLayer_gpu_forward(f, dep, cnt0, o0, 0) == 0
    g1, g2 = z(f), z(dep); assert L.DepthFlowProjectionLayer_gpu_backward(f, dep, cnt0, o0, gf, g1, g2) == 0
    w1, w2 = R.depth_flow_projection_backward(f, dep, cnt0, o0, gf)
    close(N(g1), N(w1), "dfp_g1 720p"); close(N(g2), N(w2), "dfp_g2 720p")
    o = z(x); assert L.InterpolationLayer_gpu_forward(x, f, o) == 0
    close(N(o), N(R.interpolation_forward(x, f)), "bl_fwd 720p")
