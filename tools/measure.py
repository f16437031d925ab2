"""tools/measure.py -- binds lib/libmemc_hip_measure.so, the MEASUREMENT build of the HIP library (-DMEMC_MEASURE).

The product library libmemc_hip.so, which my_package loads, carries no ablation / A-B kernels, exports no memc_debug_*
hook and reads nothing from the environment.  The measurement build is the same sources plus those hooks; `use()`
re-binds my_package's operator entry points to it for the current process, so that tools/ (and the tests that force
a particular kernel path) drive exactly the code they would otherwise, with the knobs available.

    from tools import measure as M
    M.use()                                 # tools: everything through the measurement build
    M.set_variant("projection", 110)        # < 0 restores automatic selection
    ML = M.bound()                          # tests: a separate binding, my_package stays on the product library

Never imported by my_package or the networks (tests/test_host_api.py checks)."""
import ctypes
import os

import my_package._ext.my_lib as L

MEASURE_LIB = os.path.join(os.path.dirname(L.LIB_PATH), "libmemc_hip_measure.so")
_lib = None

_SETTERS = {
    "fi_fwd": "memc_debug_set_fi_fwd_variant",
    "fi_bwd": "memc_debug_set_fi_bwd_variant",
    "fi_phase": "memc_debug_set_fi_phase",
    "projection": "memc_debug_set_projection_variant",
    "proj_scratch_blocks": "memc_debug_set_projection_scratch_blocks",
    "proj_stall_us": "memc_debug_set_projection_stall_us",
    "walk": "memc_debug_set_walk",
    "extra_lds": "memc_debug_set_extra_lds",
    "bl_cap": "memc_debug_set_bl_cap",
    "bl_bwd_direct": "memc_debug_set_bl_bwd_direct",
}


def available():
    return os.path.exists(MEASURE_LIB)


def lib():
    """The measurement build, loaded on first use.  Loading it changes nothing in my_package."""
    global _lib
    if _lib is None:
        if not available():
            raise ImportError("%s not built: `make -C memc-net_amd/csrc measure`" % MEASURE_LIB)
        _lib = ctypes.CDLL(MEASURE_LIB)                 # RTLD_LOCAL: its symbol names equal the product library's
        for sym in _SETTERS.values():
            f = getattr(_lib, sym)
            f.argtypes = [ctypes.c_int]
            f.restype = None
        _lib.memc_hip_version.restype = ctypes.c_char_p
    return _lib


class _Bound(object):
    """The operator entry points of my_package._ext.my_lib, bound to the measurement build."""


_bound = None


def bound():
    """A namespace with my_lib's entry points bound to the measurement build; my_package itself stays on the
    product library (what the forced-path tests use)."""
    global _bound
    if _bound is None:
        b = _Bound()
        for name, (n, flag) in list(L._SYMBOLS.items()) + list(L._EXTENSIONS.items()):
            setattr(b, name, L._bind(name, n, flag, lib=lib(), optional=L._OPTIONAL.get(name, ())))
        for name, n in L._WS_SYMBOLS.items():          # the projection forward with a caller-supplied workspace
            setattr(b, name, L._bind_ws(name, n, lib=lib()))
        _bound = b
    return _bound


def use():
    """Route my_package's operator calls (modules, functions, networks) through the measurement build for the rest
    of this process (idempotent).  For tools/ only -- tests that must exercise the product library use bound()."""
    b = bound()
    for name in list(L._SYMBOLS) + list(L._EXTENSIONS) + list(L._WS_SYMBOLS):
        setattr(L, name, getattr(b, name))
    return lib()


def version():
    return lib().memc_hip_version().decode()


def set_variant(op, variant):
    getattr(lib(), _SETTERS[op])(int(variant))


def reset():
    for op in _SETTERS:
        set_variant(op, 0 if op in ("extra_lds", "bl_bwd_direct", "proj_stall_us") else 8 if op == "proj_scratch_blocks"
                    else 1000 * 65536 + 120 if op == "fi_phase" else -1)
