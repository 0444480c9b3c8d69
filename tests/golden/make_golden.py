"""Regenerates tests/golden/*.npz from the CPU oracle (oracle/sobfu_oracle.c), AFTER the oracle has been pinned against the
reference's own known answers (tests/test_oracle_pins.py, tests/golden/appendix_b.json).

The fixtures freeze a few small end-to-end results so that both the oracle (CPU suite) and the HIP path (GPU suite) are
also compared against committed data, independent of each other's builds:

    solver_20x12x9.npz   inputs (phi_global, phi_n, psi0) and outputs of estimate_psi (4 iterations, alpha 0.05, w_reg 0.4):
                         psi, phi_n_psi, psi_inv, phi_global_psi_inv, per-iteration max update norms, energies
    kernels_17x9x5.npz   per-kernel outputs on an odd-sized volume: gradient, negative Laplacian, Jacobian (mode 1), potential
                         gradient, the three convolutions, psi update, warp, 5-sweep inverse, fusion

    mesh_14x11x9.npz     marching cubes of a small ellipsoid-like volume with unobserved holes: occupied-voxel table, vertices, normals
                         (volume -> world pose with a rotation)

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle as O  # noqa: E402
from sobfu_amd.synthetic import hash_field  # noqa: E402


def rand_volume(dims, seed):
    X, Y, Z = dims
    v = hash_field((Z, Y, X, 2), seed, 1.0)
    v[..., 1] = (hash_field((Z, Y, X), seed + 7) > 0).astype(np.float32)
    return v


def warped_identity(dims, seed, amp):
    X, Y, Z = dims
    psi = O.new_field(dims)
    O.init_identity(psi)
    psi[..., :3] += hash_field((Z, Y, X, 3), seed, amp)
    return psi


def solver_fixture():
    dims = (20, 12, 9)
    pg, pn, psi0 = rand_volume(dims, 901), rand_volume(dims, 902), warped_identity(dims, 903, 0.7)
    psi = psi0.copy()
    r = O.estimate_psi(pg, pn, psi, max_iter=4, alpha=0.05, w_reg=0.4, verbosity=2)
    np.savez_compressed(os.path.join(HERE, "solver_20x12x9.npz"), phi_global=pg, phi_n=pn, psi0=psi0, psi=psi, phi_n_psi=r["phi_n_psi"],
                        psi_inv=r["psi_inv"], phi_global_psi_inv=r["phi_global_psi_inv"], trace=r["trace"])


def kernel_fixture():
    dims = (17, 9, 5)
    X, Y, Z = dims
    vol, pg = rand_volume(dims, 911), rand_volume(dims, 912)
    psi = warped_identity(dims, 913, 1.4)
    S = O.sobolev_filter(7, 0.1)
    grad, L, nU, nUS, upd, inv = (O.new_field(dims) for _ in range(6))
    J = np.zeros((Z, Y, X, 4, 4), np.float32)
    O.tsdf_gradient(vol, grad)
    O.laplacian(psi, L)
    O.jacobian(psi, J, 1)
    O.potential_gradient(vol, pg, grad, L, nU, 0.6)
    conv = []
    for fn in (O.convolution_rows, O.convolution_columns, O.convolution_depth):
        fn(nUS, nU, S)
        conv.append(nUS.copy())
    psi_new = psi.copy()
    O.update_psi(psi_new, nUS, upd, 0.1)
    warped = O.new_volume(dims)
    O.apply(vol, warped, psi_new)
    O.init_identity(inv)
    O.estimate_inverse(psi_new, inv, 5)
    fused = pg.copy()
    fused[..., 1] = np.floor(np.abs(hash_field((Z, Y, X), 914, 6.0)))
    fuse_in = fused.copy()
    O.integrate_fuse(fused, warped, 4.0)
    np.savez_compressed(os.path.join(HERE, "kernels_17x9x5.npz"), vol=vol, phi_global=pg, psi=psi, taps=S, grad=grad, laplacian=L, jacobian=J,
                        nabla_U=nU, conv_rows=conv[0], conv_cols=conv[1], conv_depth=conv[2], psi_new=psi_new, updates=upd, warped=warped,
                        psi_inv5=inv, fuse_in=fuse_in, fused=fused, data_energy=np.float32(O.data_energy(pg, vol)),
                        reg_energy=np.float32(O.reg_energy_sobolev(J)), max_update=np.array(O.max_update_norm(upd), np.float32))


def mesh_fixture():
    dims = (14, 11, 9)
    X, Y, Z = dims
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    sdf = np.sqrt(((x - 6.3) / 5.1) ** 2 + ((y - 5.2) / 3.9) ** 2 + ((z - 4.1) / 3.2) ** 2) - 1.0
    vol = np.stack([np.clip(sdf, -1, 1), (hash_field((Z, Y, X), 77) > -0.9).astype(np.float32)], -1).astype(np.float32)
    R = np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], np.float32)
    t = np.array([0.05, -0.1, 0.2], np.float32)
    size = (0.7, 0.55, 0.45)
    occ, count = O.mc_occupied_voxels(vol, 4096)
    total = O.mc_offsets(occ, count)
    v, n = O.marching_cubes(vol, size, R, t)
    assert len(v) == total > 300
    np.savez_compressed(os.path.join(HERE, "mesh_14x11x9.npz"), vol=vol, R=R, t=t, size=np.array(size, np.float32), occupied=occ[:, :count],
                        vertices=v, normals=n)


if __name__ == "__main__":
    O.build()
    solver_fixture()
    kernel_fixture()
    mesh_fixture()
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))
