"""The stated tolerance against a real CUDA build ("distance to the reference as built"): BASELINE configs 1, 2
and 3 on the oracle's IEEE build and on its SO_NVCC_MODE build (flush-to-zero, <= 2-ulp divide, approximate sqrtf / powf / __expf,
fmad contraction -- what nvcc's flags in /root/reference/CMakeLists.txt:40-46 permit), warp fields compared.  The asserted bounds are
the ones the measurements support (tests/nvcc_distance.py prints the full table); where the north star's 1e-5 does not hold against
such a build, the assert says what does and why.  CPU only."""
import pytest

import nvcc_distance as D


@pytest.fixture(scope="module")
def nvcc_lib(oracle, tmp_path_factory):
    try:
        return oracle.build_nvcc(str(tmp_path_factory.mktemp("nvcc")))
    except RuntimeError as e:
        pytest.skip(str(e))


def test_mode_is_off_by_default_and_restored(oracle, nvcc_lib):
    import numpy as np

    assert not hasattr(oracle.lib(), "so_nvcc_ftz")  # the oracle everything else loads is the IEEE build
    tiny = np.float32(1e-39)
    with oracle.nvcc_mode(nvcc_lib, 1):
        v = oracle.new_volume((4, 4, 4))
        v[..., 0] = tiny
        g = oracle.new_field((4, 4, 4))
        oracle.tsdf_gradient(v, g)  # subnormal operands: flushed
        flushed = float(np.abs(g).max())
        a = np.full((4, 4, 4, 2), 3.0, np.float32)
        b = np.full((4, 4, 4, 2), 7.0, np.float32)
        oracle.integrate_fuse(a, b, 64.0)  # (3*3 + 7) / 4 = 4: an inexact-free quotient by a power of two stays exact
        assert float(a[0, 0, 0, 0]) == 4.0
    assert flushed == 0.0
    assert float(tiny * np.float32(0.5)) > 0.0  # the calling thread is back to IEEE gradual underflow


def test_config3_roofline_config_stays_under_1e_5(oracle, nvcc_lib):
    """256^3, params_boxing.ini solver values, 50 iterations from two initSphere volumes (approximate sqrtf, powf, divide and contracted
    voxel centres perturb 6.1 M voxels of each volume by <= 1.1e-6): the warp field moves by 7.8e-6 in TOTAL L2 over the 16.7 M voxels."""
    worst, vols = D.distance(D.config3(50), nvcc_lib, seeds=(1,))
    w = worst[0]
    assert vols[0]["phi_global"]["voxels"] > 1_000_000 and vols[0]["phi_global"]["max_abs"] < 2e-6
    assert w["l2"] < 1e-5 and w["max_abs"] < 4e-6 and w["iters_equal"], w


def test_config1_depth_driven(oracle, nvcc_lib):
    """64^3, two depth frames, 10 iterations.  Without __expf's error the TSDF inputs move by <= 5e-6 and psi by <= 8e-6 per component
    (total L2 1.1e-4: above 1e-5 because thousands of voxels each move by ~1e-6).  With it, the bilateral filter's integer output flips by
    1 mm on a pixel or two, each flip moves the TSDF voxels behind that pixel by 1 mm / trunc, and psi locally by up to 1e-2 voxels."""
    run = D.config1()
    worst, vols = D.distance(run, nvcc_lib, seeds=(1,), mask=D.MASKS["all but __expf"])
    assert worst[0]["max_abs"] < 2e-5 and worst[0]["l2"] < 3e-4 and worst[0]["rms"] < 1e-6, worst
    assert vols[0]["filtered_px"] == 0 and vols[0]["phi_n"]["max_abs"] < 1e-5
    worst, vols = D.distance(run, nvcc_lib, seeds=(1,))
    assert worst[0]["max_abs"] < 3e-2 and worst[0]["rms"] < 1e-4, worst
    assert 0 < vols[0]["filtered_px"] < 20


def test_config2_pixel_flips_dominate(oracle, nvcc_lib):
    """128^3, 7 frames, 3 solved frames of 16 iterations.  The contracted voxel centre x * vs + vs / 2 (tsdf_volume.cu:70-71) differs from the
    two-rounding one in the last ulp; where the projected coordinate sits on a pixel boundary (this volume is centred on the optical axis)
    floor(coo) picks the neighbouring depth pixel and the voxel's TSDF jumps by up to 3e-2.  The field stays within 1e-2 voxels max,
    2e-5 RMS per voxel -- the 1e-5 TOTAL-L2 bar is not attainable against a build whose voxel->pixel indexing differs."""
    worst, vols = D.distance(D.config2(), nvcc_lib, seeds=(1,))
    assert len(worst) == 3
    for w, v in zip(worst, vols):
        assert w["max_abs"] < 2e-2 and w["rms"] < 3e-5 and w["iters_equal"], w
        assert v["phi_n"]["max_abs"] > 1e-3  # a pixel flip, not rounding
