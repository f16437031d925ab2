"""The CPU oracle against vectors recorded from the REFERENCE'S OWN GPU KERNELS (my_package/src/my_lib_kernel.cu
built for gfx950 and run on an MI355X by tests/golden/make_golden_ref_gpu.py; fixtures tests/golden/ref_gpu_*.npz
hold outputs only -- the inputs are regenerated from the case's seed).  This is what pins the oracle: every entry
point of the path, forward and backward, with and without the hole-filling pass.  Runs anywhere, no GPU.

Tolerance: 1e-4 absolute (+ 1e-5 relative): the reference accumulates its scatters with fp32 atomics in hardware
order, the oracle sequentially; integer-valued projection counts must match bit for bit."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refcases as RC     # noqa: E402

ATOL, RTOL = 1e-4, 1e-5


@pytest.mark.parametrize("case", RC.REF_CASES, ids=[RC.name(c) for c in RC.REF_CASES])
def test_oracle_matches_reference_gpu_vectors(oracle, case):
    path = os.path.join(HERE, "golden", "ref_gpu_%s.npz" % RC.name(case))
    want = np.load(path)
    got = RC.oracle_outputs(oracle, RC.make(case))
    keys = sorted(k for k in want.files if k != "device")
    assert keys == sorted(got)
    for k in keys:
        if k.startswith("fp_cnt"):
            assert np.array_equal(got[k], want[k]), k
            continue
        err = np.abs(got[k].astype(np.float64) - want[k].astype(np.float64))
        bound = ATOL + RTOL * np.abs(want[k])
        assert float((err - bound).max()) <= 0, "%s: max abs err %.3g" % (k, float(err.max()))
    # the fixtures do contain what they are meant to pin
    assert (want["fp_cnt0"] == 0).any() or case[4] == "smooth"          # holes exist in the iid cases ...
    if (want["fp_cnt0"] == 0).any():
        assert np.abs(want["fp_out1"] - want["fp_out0"]).max() > 0      # ... and the fill pass changed them
