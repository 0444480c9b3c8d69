"""The reference's six value-pinning gtest cases (test/deformation_field_test.cpp:92-336, test/reductions_test.cpp:86-101) run
on the HIP path through the C ABI -- the same assertions, tolerances and set-ups the reference's own tests hold (the oracle is
held to them in tests/test_oracle_pins.py; tests/cpp/host_shell_tests.cpp runs them through the C++ shells)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from sobfu_amd import ops as O

    return O


def _params64(size=0.25, trunc_vox=10.0, eta_vox=2.0):
    size = np.float32(size)
    vs = np.array([size / np.float32(64)] * 3, np.float32)
    return (64, 64, 64), vs, np.float32(trunc_vox) * vs[0], np.float32(eta_vox) * vs[0]


def host(t):
    return t.cpu().numpy()


def dev(a):
    import torch

    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_ref_ClearTest_identity(ops):
    """deformation_field_test.cpp:92-108: a fresh DeformationField is psi(i,j,k) = (i,j,k)."""
    psi = ops.new_field((64, 64, 64))
    ops.init_identity(psi)
    psi = host(psi)
    k, j, i = np.meshgrid(np.arange(64), np.arange(64), np.arange(64), indexing="ij")
    assert np.array_equal(psi[..., 0], i) and np.array_equal(psi[..., 1], j) and np.array_equal(psi[..., 2], k)
    assert not psi[..., 3].any()


def test_ref_TsdfGradientTest(ops):
    """deformation_field_test.cpp:111-149: |grad phi| ~ voxel/trunc = 0.1 (tol 0.15) on interior non-truncated voxels."""
    dims, vs, trunc, eta = _params64()
    vol = ops.new_volume(dims)
    ops.init_sphere(vol, vs, trunc, eta, (0.16, 0.16, 0.16), 0.01)
    grad = ops.new_field(dims)
    ops.tsdf_gradient(vol, grad)
    vol, grad = host(vol), host(grad)
    n = np.sqrt((grad[1:-1, 1:-1, 1:-1, :3] ** 2).sum(-1))
    m = np.abs(vol[1:-1, 1:-1, 1:-1, 0]) < 1.0
    assert m.sum() > 1000
    assert np.all(np.abs(n[m] - vs[0] / trunc) <= 0.15)


def test_ref_UniformFieldJacobianTest(ops):
    """deformation_field_test.cpp:152-196: psi == (1,1,1) => J == 0 everywhere (mode 0)."""
    psi = np.zeros((64, 64, 64, 4), np.float32)
    psi[..., :3] = 1.0
    J = ops.new_jacobian((64, 64, 64))
    ops.jacobian(dev(psi), J, 0)
    assert np.all(np.abs(host(J)[..., :3, :3]) <= 1e-5)


def test_ref_JacobianTestSimple(ops):
    """deformation_field_test.cpp:199-249: psi = (i,j,k) => J = I on the interior."""
    psi = ops.new_field((64, 64, 64))
    ops.init_identity(psi)
    J = ops.new_jacobian((64, 64, 64))
    ops.jacobian(psi, J, 0)
    assert np.all(np.abs(host(J)[1:-1, 1:-1, 1:-1, :3, :3] - np.eye(3, dtype=np.float32)) <= 1e-5)


def test_ref_JacobianLaplacianTestComplicated(ops):
    """deformation_field_test.cpp:252-336: psi = (i(1-j), exp(-k)+j, k); J and the NEGATIVE Laplacian, tol 0.1."""
    k, j, i = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(64, dtype=np.float32),
                          np.arange(64, dtype=np.float32), indexing="ij")
    psi = np.zeros((64, 64, 64, 4), np.float32)
    psi[..., 0] = i * (1.0 - j)
    psi[..., 1] = np.exp(-k) + j
    psi[..., 2] = k
    psi_d = dev(psi)
    J = ops.new_jacobian((64, 64, 64))
    ops.jacobian(psi_d, J, 0)
    J = host(J)
    s = (slice(1, -1),) * 3
    exp = np.zeros((62, 62, 62, 3, 3), np.float32)
    exp[..., 0, 0] = 1.0 - j[s]
    exp[..., 0, 1] = -i[s]
    exp[..., 1, 1] = 1.0
    exp[..., 1, 2] = -np.exp(-k[s])
    exp[..., 2, 2] = 1.0
    assert np.all(np.abs(J[s][..., :3, :3] - exp) <= 0.1)
    L = ops.new_field((64, 64, 64))
    ops.laplacian(psi_d, L)
    L = host(L)
    assert np.all(np.abs(L[s][..., 0]) <= 0.1)
    assert np.all(np.abs(L[s][..., 1] + np.exp(-k[s])) <= 0.1)
    assert np.all(np.abs(L[s][..., 2]) <= 0.1)


def test_ref_DataTermTest(ops):
    """reductions_test.cpp:86-101: phi_n = 0, phi_global = 1 everywhere => data energy = 0.5*N (tol 0.1); the wave-shuffle
    tail of the tree reduction included."""
    dims, vs, trunc, eta = _params64(trunc_vox=5.0)
    pg, pn = ops.new_volume(dims), ops.new_volume(dims)
    ops.init_sphere(pg, vs, trunc, eta, (5.0, 5.0, 5.0), 0.01)
    assert np.all(host(pg)[..., 0] == 1.0)
    assert abs(ops.data_energy(pg, pn) - 0.5 * 64 ** 3) <= 0.1
    assert ops.reduce_config(64 ** 3) == (256, 512)


def test_reference_own_test_binary():
    """The reference's OWN test translation units (test/*.cpp, compiled unchanged against include/ by oracle/ref_callers.py in the build
    container -- the binary travels, the reference does not) run on this GPU: its nine gtest cases pass on the HIP path."""
    import os
    import subprocess

    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "reference_gtests")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/reference_gtests was not built (needs /root/reference: __graft_entry__.build() in the build container)")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    tail = r.stdout[-4000:] + r.stderr[-2000:]
    assert r.returncode == 0, tail
    assert "9 tests ran, 0 failed" in r.stdout, tail
    for name in ("DeformationFieldTest.ClearTest", "DeformationFieldTest.TsdfGradientTest", "ReductionsTest", "SolverTest"):
        assert name in r.stdout, (name, tail)
