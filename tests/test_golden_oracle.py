"""The committed fixtures (tests/golden/*.npz, produced by tests/golden/make_golden.py from the oracle) must be
reproduced bit-for-bit by the oracle as built here: freezes the checker.  CPU only."""
import glob
import os

import numpy as np
import pytest

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "small_*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p) for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    d = np.load(path)
    x, flow, filt, gout, depth, gflow = (d[k] for k in ("x", "flow", "filt", "gout", "depth", "gflow"))
    assert np.array_equal(oracle.filter_interpolation_forward(x, flow, filt), d["fi_out"])
    g1, g2, g3 = oracle.filter_interpolation_backward(x, flow, filt, gout)
    assert np.array_equal(g1, d["fi_g1"]) and np.array_equal(g2, d["fi_g2"]) and np.array_equal(g3, d["fi_g3"])
    assert np.array_equal(oracle.interpolation_ch_forward(x, flow), d["ich_out"])
    h1, h2 = oracle.interpolation_ch_backward(x, flow, gout)
    assert np.array_equal(h1, d["ich_g1"]) and np.array_equal(h2, d["ich_g2"])
    for fh in (0, 1):
        o, c = oracle.flow_projection_forward(flow, fh)
        assert np.array_equal(o, d["fp_out%d" % fh]) and np.array_equal(c, d["fp_count"])
        o, c = oracle.depth_flow_projection_forward(flow, depth, fh)
        assert np.array_equal(o, d["dfp_out%d" % fh]) and np.array_equal(c, d["dfp_count"])
    assert np.array_equal(oracle.flow_projection_backward(flow, d["fp_count"], gflow), d["fp_g1"])
    p1, p2 = oracle.depth_flow_projection_backward(flow, depth, d["dfp_count"], d["dfp_out0"], gflow)
    assert np.array_equal(p1, d["dfp_g1"]) and np.array_equal(p2, d["dfp_g2"])


def test_config1_fixture(oracle):
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "config1_fi_fwd_128.npz"))
    out = oracle.filter_interpolation_forward(d["x"], d["flow"], d["filt"])
    assert np.array_equal(out, d["fi_out"])
    # the reference's own result on these tensors
    assert abs(float(d["fi_out"].sum()) - 48114.199) < 0.01
    assert int((d["fp_count"] == 0).sum()) == 452


def test_oracle_is_thread_count_independent(oracle):
    """OpenMP runs batch items in parallel; inside one item the order is the reference's sequential order,
    so the result must not depend on the number of threads."""
    import ctypes
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r);"
        "from oracle import memc_oracle as O; from tools import synth;"
        "rng=np.random.default_rng(5); x=synth.np_image(rng,4,3,24,28); f=synth.np_flow(rng,4,24,28,'iid',3.0);"
        "k=synth.np_filter(rng,4,24,28); g=synth.np_image(rng,4,3,24,28);"
        "r=O.filter_interpolation_backward(x,f,k,g); o,c=O.flow_projection_forward(f,1);"
        "import hashlib; print(hashlib.sha1(b''.join(a.tobytes() for a in (*r,o,c))).hexdigest())"
        % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for n in ("1", "4"):
        env = dict(os.environ, OMP_NUM_THREADS=n)
        outs.append(subprocess.run([sys.executable, "-c", code], env=env, check=True,
                                   stdout=subprocess.PIPE, text=True).stdout.strip())
    assert outs[0] == outs[1]
    assert ctypes  # keep import (documented dependency of the oracle wrapper)
