"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/sobfu_hip.h
declares; host-only entry points (filter table, reduction sizing, error strings) agree with the oracle."""
import ctypes as C

import numpy as np
import pytest


@pytest.fixture(scope="module")
def lib():
    from sobfu_amd import build

    build.build_hip()
    from sobfu_amd import _lib

    return _lib.lib()


def test_exports_every_declared_symbol(lib):
    from sobfu_amd import _lib

    names = _lib.declared_symbols()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), n
    assert lib.sobfu_hip_abi_version() == 3


def test_error_strings(lib):
    assert lib.sobfu_hip_error_string(0) == b"success"
    assert b"bad argument" in lib.sobfu_hip_error_string(-1)
    assert b"filter" in lib.sobfu_hip_error_string(-2)
    # argument validation happens before any device call, so it is testable without a GPU
    assert lib.sobfu_hip_clear_volume(None, 4, 4, 4, None) == -1
    assert lib.sobfu_hip_apply(None, None, None, 4, 4, 4, None) == -1
    assert lib.sobfu_hip_init_identity(C.c_void_p(16), 0, 4, 4, None) == -1


def test_sobolev_filter_matches_oracle(lib, oracle):
    from sobfu_amd import ops

    for s, lam in ((3, 0.1), (7, 0.05), (7, 0.1), (7, 0.2), (7, 0.4), (9, 0.05), (9, 0.1), (11, 0.1)):
        a, b = ops.sobolev_filter(s, lam), oracle.sobolev_filter(s, lam)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (s, lam)
    out = (C.c_float * 16)()
    assert lib.sobfu_hip_sobolev_filter(7, C.c_float(0.3), out) == -2
    assert lib.sobfu_hip_sobolev_filter(5, C.c_float(0.1), out) == -2


def test_reduce_config_matches_oracle(lib, oracle):
    from sobfu_amd import ops

    for n in (1, 2, 3, 100, 765, 1023, 1024, 1025, 19200, 64 ** 3, 256 ** 3, 512 ** 3):
        assert ops.reduce_config(n) == oracle.reduce_config(n), n


def test_cpp_host_shells_compile():
    """include/sobfu_amd/sobfu.hpp (the reference's class surface over the C ABI) + its test driver build with g++."""
    import os

    from sobfu_amd import build_host

    exe = build_host.build_host()
    assert os.path.exists(exe) and os.access(exe, os.X_OK)
    # the reference include paths resolve to the shells
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for h in ("sobfu/solver.hpp", "sobfu/vector_fields.hpp", "kfusion/cuda/tsdf_volume.hpp", "kfusion/internal.hpp"):
        assert os.path.exists(os.path.join(root, "include", h))
