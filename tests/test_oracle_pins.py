"""Pins the CPU oracle (oracle/sobfu_oracle.c) before anything trusts it.

(1) The six value-pinning gtest cases of the reference, restated on the oracle:
    test/deformation_field_test.cpp:92-336 and test/reductions_test.cpp:86-101.
(2) Known-answer values recorded in SURVEY.md Appendix B -- numbers the reference's own code printed
    (solver energies / max update norms, warp-field statistics, TSDF sums) and convolution impulse responses.
"""
import numpy as np
import pytest

from sobfu_amd.synthetic import render_sphere_depth


def _params64(size=0.25, trunc_vox=10.0, eta_vox=2.0):
    size = np.float32(size)
    vs = np.array([size / np.float32(64)] * 3, np.float32)
    return (64, 64, 64), vs, np.float32(trunc_vox) * vs[0], np.float32(eta_vox) * vs[0]


# ---------------------------------------------------------------------------------------------------
# (1) reference gtest cases
# ---------------------------------------------------------------------------------------------------
def test_ref_ClearTest_identity(oracle):
    """deformation_field_test.cpp:92-108: a fresh DeformationField is psi(i,j,k) = (i,j,k)."""
    psi = oracle.new_field((64, 64, 64))
    oracle.init_identity(psi)
    k, j, i = np.meshgrid(np.arange(64), np.arange(64), np.arange(64), indexing="ij")
    assert np.array_equal(psi[..., 0], i) and np.array_equal(psi[..., 1], j) and np.array_equal(psi[..., 2], k)
    assert not psi[..., 3].any()


def test_ref_TsdfGradientTest(oracle):
    """deformation_field_test.cpp:111-149: |grad phi| ~ voxel/trunc = 0.1 (tol 0.15) on interior non-truncated voxels."""
    dims, vs, trunc, eta = _params64()
    vol = oracle.new_volume(dims)
    oracle.init_sphere(vol, vs, trunc, eta, (0.16, 0.16, 0.16), 0.01)
    grad = oracle.new_field(dims)
    oracle.tsdf_gradient(vol, grad)
    n = np.sqrt((grad[1:-1, 1:-1, 1:-1, :3] ** 2).sum(-1))
    m = np.abs(vol[1:-1, 1:-1, 1:-1, 0]) < 1.0
    assert m.sum() > 1000
    assert np.all(np.abs(n[m] - vs[0] / trunc) <= 0.15)


def test_ref_UniformFieldJacobianTest(oracle):
    """deformation_field_test.cpp:152-196: psi == (1,1,1) => J == 0 everywhere (mode 0)."""
    psi = oracle.new_field((64, 64, 64))
    psi[..., :3] = 1.0
    J = np.zeros((64, 64, 64, 4, 4), np.float32)
    oracle.jacobian(psi, J, 0)
    assert np.all(np.abs(J[..., :3, :3]) <= 1e-5)


def test_ref_JacobianTestSimple(oracle):
    """deformation_field_test.cpp:199-249: psi = (i,j,k) => J = I on the interior."""
    psi = oracle.new_field((64, 64, 64))
    oracle.init_identity(psi)
    J = np.zeros((64, 64, 64, 4, 4), np.float32)
    oracle.jacobian(psi, J, 0)
    assert np.all(np.abs(J[1:-1, 1:-1, 1:-1, :3, :3] - np.eye(3, dtype=np.float32)) <= 1e-5)


def test_ref_JacobianLaplacianTestComplicated(oracle):
    """deformation_field_test.cpp:252-336: psi = (i(1-j), exp(-k)+j, k); J and the NEGATIVE Laplacian, tol 0.1."""
    k, j, i = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(64, dtype=np.float32),
                          np.arange(64, dtype=np.float32), indexing="ij")
    psi = oracle.new_field((64, 64, 64))
    psi[..., 0] = i * (1.0 - j)
    psi[..., 1] = np.exp(-k) + j
    psi[..., 2] = k
    J = np.zeros((64, 64, 64, 4, 4), np.float32)
    oracle.jacobian(psi, J, 0)
    s = (slice(1, -1),) * 3
    exp = np.zeros((62, 62, 62, 3, 3), np.float32)
    exp[..., 0, 0] = 1.0 - j[s]
    exp[..., 0, 1] = -i[s]
    exp[..., 1, 1] = 1.0
    exp[..., 1, 2] = -np.exp(-k[s])
    exp[..., 2, 2] = 1.0
    assert np.all(np.abs(J[s][..., :3, :3] - exp) <= 0.1)
    L = oracle.new_field((64, 64, 64))
    oracle.laplacian(psi, L)
    assert np.all(np.abs(L[s][..., 0]) <= 0.1)
    assert np.all(np.abs(L[s][..., 1] + np.exp(-k[s])) <= 0.1)
    assert np.all(np.abs(L[s][..., 2]) <= 0.1)


def test_ref_DataTermTest(oracle):
    """reductions_test.cpp:86-101: phi_n = 0, phi_global = 1 everywhere => data energy = 0.5*N (tol 0.1)."""
    dims, vs, trunc, eta = _params64(trunc_vox=5.0)
    pg, pn = oracle.new_volume(dims), oracle.new_volume(dims)
    oracle.init_sphere(pg, vs, trunc, eta, (5.0, 5.0, 5.0), 0.01)
    assert np.all(pg[..., 0] == 1.0)
    assert abs(oracle.data_energy(pg, pn) - 0.5 * 64 ** 3) <= 0.1
    assert oracle.reduce_config(64 ** 3) == (256, 512)


# ---------------------------------------------------------------------------------------------------
# (2) SURVEY.md Appendix B known answers
# ---------------------------------------------------------------------------------------------------
RUN1 = {  # iter: (e_data, e_reg, max ||update||)  -- test/solver_test.cpp:109-132 set-up, verbosity 2
    1: (24.5457, 0.0, 0.000383393), 2: (24.5319, 3.69403e-05, 0.000383125), 3: (24.5182, 0.000145988, 0.000382842),
    4: (24.5045, 0.000324705, 0.00038255), 5: (24.4909, 0.00057081, 0.000382319),
    10: (24.4234, 0.00274294, 0.000380488)}


def _six(x, ref):  # the reference prints 6 significant digits (std::cout default)
    return float(f"{x:.6g}") == pytest.approx(ref, rel=2e-6, abs=1e-12)


def test_appendixB_run1_solver_trace(oracle):
    dims, vs, trunc, eta = _params64()
    pg, pn = oracle.new_volume(dims), oracle.new_volume(dims)
    oracle.init_sphere(pg, vs, trunc, eta, (0.13, 0.13, 0.13), 0.012)
    oracle.init_sphere(pn, vs, trunc, eta, (0.125, 0.13, 0.13), 0.012)
    ident = oracle.new_field(dims)
    oracle.init_identity(ident)

    psi = ident.copy()
    oracle.estimate_psi(pg, pn, psi, max_iter=3, alpha=0.01, w_reg=0.4, inverse_iters=1)
    d = (psi - ident)[..., :3].astype(np.float64)
    assert _six(d.sum(), -3.32055855) or abs(d.sum() + 3.32055855) < 5e-8
    assert abs(np.sqrt((d ** 2).sum()) - 0.059968784) < 1e-9
    assert abs(np.sqrt((d ** 2).sum(-1)).max() - 0.00114924996) < 1e-10
    assert np.allclose(psi[32, 32, 30, :3], (29.9992485, 31.9996338, 31.9996338), rtol=0, atol=2e-6)

    psi = ident.copy()
    r = oracle.estimate_psi(pg, pn, psi, max_iter=10, alpha=0.01, w_reg=0.4, verbosity=2, inverse_iters=1)
    assert r["iters"] == 10  # max_update_norm = -1 never converges
    for it, (ed, er, mx) in RUN1.items():
        row = r["trace"][it - 1]
        assert _six(row[0], ed) and _six(row[1], er) and _six(row[2], mx), (it, row)
    d = (psi - ident)[..., :3].astype(np.float64)
    assert abs(d.sum() + 11.0575466) < 5e-7
    assert abs(np.sqrt((d ** 2).sum()) - 0.197894352) < 1e-9
    assert abs(np.sqrt((d ** 2).sum(-1)).max() - 0.00381812943) < 1e-10
    assert np.allclose(psi[32, 32, 30, :3], (29.9975681, 31.9988365, 31.9988365), rtol=0, atol=2e-6)


RUN2 = [(1059.91, 0.0, 0.234661), (634.563, 44.8031, 0.117406), (480.888, 95.3376, 0.0869742),
        (401.086, 133.849, 0.0658413), (352.758, 162.537, 0.0520266), (320.525, 184.27, 0.0423408),
        (297.475, 201.126, 0.0350061), (280.093, 214.493, 0.0293397), (266.423, 225.299, 0.0258926),
        (255.322, 234.173, 0.0237178)]


def _stats(v):
    t, w = v[..., 0], v[..., 1]
    return t.astype(np.float64).sum(), float(w.sum()), int(((np.abs(t) < 1) & (w > 0)).sum())


def test_appendixB_run2_frame_pipeline(oracle):
    """Config 1 of BASELINE.json: 64^3, two synthetic translating-sphere depth frames, 10 iterations
    (params_advent.ini values): bilateral -> truncate -> dists -> integrate -> solve -> fuse -> inverse."""
    dims = (64, 64, 64)
    size = np.float32(0.5)
    vs = np.array([size / np.float32(64)] * 3, np.float32)
    trunc, eta = np.float32(5) * vs[0], np.float32(2) * vs[0]
    intr = (570.342, 570.342, 320.0, 240.0)
    R = np.eye(3, dtype=np.float32)
    t = np.array([-size / np.float32(2), -size / np.float32(2), 0.5], np.float32)
    vols = []
    for cx in (0.0, 0.005):
        d = render_sphere_depth((cx, 0.0, 0.75), 0.1, intr)
        d = oracle.bilateral(d, 7, 4.5, 0.005)
        oracle.truncate_depth(d, 1.5)
        dist = oracle.compute_dists(d, intr)
        v = oracle.new_volume(dims)
        oracle.integrate_depth(dist, v, vs, trunc, eta, R, t, intr)
        vols.append(v)
    s0, s1 = _stats(vols[0]), _stats(vols[1])
    assert abs(s0[0] + 19581.2063) < 5e-5 and s0[1:] == (8468.0, 2909)
    assert abs(s1[0] + 19552.2192) < 5e-5 and s1[1:] == (8470.0, 2921)
    assert int(((vols[0][..., 0] != 0) | (vols[0][..., 1] != 0)).sum()) == 34832  # Appendix B run 7

    ident = oracle.new_field(dims)
    oracle.init_identity(ident)
    psi = ident.copy()
    r = oracle.estimate_psi(vols[0], vols[1], psi, max_iter=10, alpha=0.1, w_reg=0.2, verbosity=2)
    for row, (ed, er, mx) in zip(r["trace"], RUN2):
        assert _six(row[0], ed) and _six(row[1], er) and _six(row[2], mx), row
    s = _stats(r["phi_n_psi"])
    assert abs(s[0] + 19567.7977) < 5e-5 and s[1:] == (8561.0, 4363)
    fused = vols[0].copy()
    oracle.integrate_fuse(fused, r["phi_n_psi"], 128.0)
    s = _stats(fused)
    assert abs(s[0] + 19566.7973) < 5e-5 and s[1:] == (17029.0, 4531)
    s = _stats(r["phi_global_psi_inv"])
    assert abs(s[0] + 19571.7471) < 5e-5 and s[1:] == (8601.0, 4392)
    d = (psi - ident)[..., :3].astype(np.float64)
    assert abs(np.sqrt((d ** 2).sum()) - 25.954649) < 5e-7
    assert abs(np.sqrt((d ** 2).sum(-1)).max() - 0.57107548) < 5e-9


def test_appendixB_run3_convolution_impulses(oracle):
    """Taps 1..7 expose orientation: out(x) = sum_j S[3-j] in(x+j), SUM of three 1-D passes, clamp-to-edge."""
    dims = (64, 64, 64)
    S = np.arange(1, 8, dtype=np.float32)
    src, dst = oracle.new_field(dims), oracle.new_field(dims)
    src[32, 32, 32, :3] = 1.0
    oracle.convolution_rows(dst, src, S)
    oracle.convolution_columns(dst, src, S)
    oracle.convolution_depth(dst, src, S)
    exp = [1, 2, 3, 12, 5, 6, 7]
    assert list(dst[32, 32, 29:36, 0]) == exp and list(dst[32, 29:36, 32, 1]) == exp
    assert list(dst[29:36, 32, 32, 2]) == exp
    assert dst[32, 33, 33, 0] == 0
    src[:] = 0
    src[0, 0, 0, :3] = 1.0
    oracle.convolution_rows(dst, src, S)
    oracle.convolution_columns(dst, src, S)
    oracle.convolution_depth(dst, src, S)
    assert list(dst[0, 0, 0:5, 0]) == [66, 18, 13, 7, 0]
    assert list(dst[0, 0:5, 0, 0]) == [66, 18, 13, 7, 0] and list(dst[0:5, 0, 0, 0]) == [66, 18, 13, 7, 0]


def test_sobolev_filter_table(oracle):
    """solver.cpp:160-262; normalised S=7, lambda=0.1 taps bit patterns quoted in SURVEY.md section 0.3."""
    S = oracle.sobolev_filter(7, 0.1)
    assert [hex(v) for v in S.view(np.uint32)[:4]] == ["0x398a658b", "0x3b7e4dc8", "0x3d6cd2f5", "0x3f60466d"]
    assert np.array_equal(S, S[::-1])
    with pytest.raises(ValueError):
        oracle.sobolev_filter(7, 0.3)
    for s, lam in ((3, 0.1), (7, 0.05), (7, 0.2), (7, 0.4), (9, 0.05), (9, 0.1), (11, 0.1)):
        h = oracle.sobolev_filter(s, lam)
        assert abs(float(h.sum()) - 1.0) < 1e-6 and h.size == s


def test_reduce_config_and_energy_small_sizes(oracle):
    """precomp.cpp:20-43 launch sizing incl. n < 1024 and non-power-of-two n (Appendix B runs 5/6 shapes)."""
    assert oracle.reduce_config(32 ** 3) == (32, 512)
    assert oracle.reduce_config(40 * 24 * 20) == (19, 512)
    assert oracle.reduce_config(17 * 9 * 5) == (1, 512)
    assert oracle.reduce_config(512 ** 3) == (65536, 512)
    assert oracle.reduce_config(100) == (1, 64)
    a, b = oracle.new_volume((17, 9, 5)), oracle.new_volume((17, 9, 5))
    a[..., 0] = np.linspace(-1, 1, 765, dtype=np.float32).reshape(5, 9, 17)
    e = oracle.data_energy(a, b)
    assert abs(e - 0.5 * float((a[..., 0].astype(np.float64) ** 2).sum())) < 1e-3
