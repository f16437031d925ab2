"""The drop-in claim, at the import surface: the reference's own, UNMODIFIED `networks/MEMC_Net_star.py` is imported
from /root/reference with the REAL `memc-net_amd/my_package` (the ctypes loader over libmemc_hip.so -- not the
oracle-backed stand-in the other network tests install) on `sys.path`.  It must construct, its operator modules must
be this repository's classes, and the first operator call on CPU tensors must fail with the loader's "no CPU path"
error -- i.e. the reference code reached the real HIP binding and nothing fell back.

CPU only; skipped where /root/reference does not exist (the GPU box).  On a GPU the same reference file running on
the HIP operators is covered by tests/test_gpu_network.py."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _netutil      # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(_netutil.REF_ROOT, "networks")),
                                reason="needs the reference tree at /root/reference")


def _fresh_real_my_package():
    for k in [k for k in sys.modules if k == "my_package" or k.startswith("my_package.")]:
        del sys.modules[k]
    import my_package._ext.my_lib as my_lib
    assert "memc-net_amd" in my_lib.__file__ and os.path.exists(my_lib.LIB_PATH)
    return my_lib


def test_unmodified_reference_network_reaches_the_real_loader():
    my_lib = _fresh_real_my_package()
    ref = _netutil.import_reference_networks()
    try:
        assert ref.__file__.startswith(_netutil.REF_ROOT)
        star = sys.modules["networks.MEMC_Net_star"]
        # the names the reference imports (MEMC_Net_star.py:6-7) resolved to this repository's classes
        from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
        from my_package.modules.FlowProjectionModule import FlowProjectionModule
        assert star.FilterInterpolationModule is FilterInterpolationModule
        assert star.FlowProjectionModule is FlowProjectionModule
        assert sys.modules["my_package._ext.my_lib"] is my_lib
        net = ref.MEMC_Net_star(channel=3, filter_size=4, training=False).eval()
        assert sum(p.numel() for p in net.parameters()) == 70312501
        x = _netutil.frames(3, 1, 128, 128)
        with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU path"):
            net(x)                                   # first custom operator on the path: FlowProjectionModule
    finally:
        _netutil.purge_networks()


def test_unmodified_reference_operator_call_sites():
    """The call shapes of MEMC_Net_star.py:264-285 against the real modules: constructor arguments, argument
    counts, and the error type on CPU tensors (the reference's own CPU branches die with NameError instead)."""
    _fresh_real_my_package()
    from my_package.modules.FilterInterpolationModule import FilterInterpolationModule
    from my_package.modules.FlowProjectionModule import FlowProjectionModule
    from my_package.modules.InterpolationModule import InterpolationModule
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU path"):
        FlowProjectionModule(z(1, 2, 8, 8).requires_grad)(z(1, 2, 8, 8))        # :266
    with pytest.raises(RuntimeError, match="no CPU path"):
        FilterInterpolationModule()(z(1, 3, 8, 8), z(1, 2, 8, 8), z(1, 16, 8, 8))   # :274-275
    with pytest.raises(RuntimeError, match="no CPU path"):
        InterpolationModule()(z(1, 3, 8, 8), z(1, 2, 8, 8))                      # MEMC_Net_VE.py:454
