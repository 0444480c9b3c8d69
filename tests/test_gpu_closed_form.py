"""Closed-form known answers (tests/closed_form.py) on the HIP kernels through the C ABI -- the launcher-for-launcher kernels AND
the fused passes the solver runs: a check of the product path that involves neither the oracle nor the reference's code, only the
mathematics of the operators (linear / quadratic fields under unit-sum symmetric taps, the documented DC gain 3 and impulse
responses, affine warps, exact central differences of quadratics)."""
import numpy as np
import pytest
import torch

import closed_form as cf

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


class HipApi:
    def __init__(self, ops, fused):
        self.ops, self.fused = ops, fused

    def _field(self, a):
        f = np.zeros(cf.DIMS[::-1] + (4,), np.float32)
        f[..., :a.shape[-1]] = a
        return dev(f)

    def _vol(self, tsdf, weight=1.0):
        v = np.zeros(cf.DIMS[::-1] + (2,), np.float32)
        v[..., 0], v[..., 1] = tsdf, weight
        return dev(v)

    def run_conv(self, f, S):
        ops = self.ops
        src, dst = self._field(f[..., :3]), ops.new_field(cf.DIMS)
        ops.convolution_rows(dst, src, S)
        ops.convolution_columns(dst, src, S)
        ops.convolution_depth(dst, src, S)
        return dst.cpu().numpy()

    def run_apply(self, phi, psi, weight=1.0):
        out = self.ops.new_volume(cf.DIMS)
        self.ops.apply(self._vol(phi, weight), out, self._field(psi))
        return out.cpu().numpy()

    def run_potential_gradient(self, F, G, psi, w_reg):
        ops = self.ops
        nU = ops.new_field(cf.DIMS)
        if self.fused:
            ops.fused_potential_gradient(self._vol(F), self._vol(G), self._field(psi), nU, w_reg)
        else:
            g, L = ops.new_field(cf.DIMS), ops.new_field(cf.DIMS)
            ops.tsdf_gradient(self._vol(F), g)
            ops.laplacian(self._field(psi), L)
            ops.potential_gradient(self._vol(F), self._vol(G), g, L, nU, w_reg)
        return nU.cpu().numpy()

    def run_smooth_update_apply(self, nU, psi, phi, S, alpha):
        ops = self.ops
        p, out = self._field(psi), ops.new_volume(cf.DIMS)
        if self.fused:
            ops.fused_smooth_update_apply(self._field(nU), p, self._vol(phi), out, S, alpha)
        else:
            nUS, upd, src = ops.new_field(cf.DIMS), ops.new_field(cf.DIMS), self._field(nU)
            ops.convolution_rows(nUS, src, S)
            ops.convolution_columns(nUS, src, S)
            ops.convolution_depth(nUS, src, S)
            ops.update_psi(p, nUS, upd, alpha)
            ops.apply(self._vol(phi), out, p)
        return p.cpu().numpy(), out.cpu().numpy()


    def run_inverse(self, psi, sweeps):
        inv = self.ops.new_field(cf.DIMS)
        self.ops.estimate_inverse(self._field(psi), inv, sweeps)
        return inv.cpu().numpy()

    def run_max_norm(self, updates):
        return self.ops.max_update_norm(dev(updates.astype(np.float32)))

    def run_fuse(self, g, n, max_weight):
        vg, vn = dev(g.astype(np.float32)), dev(n.astype(np.float32))
        self.ops.integrate_fuse(vg, vn, max_weight)
        return vg.cpu().numpy()


@pytest.mark.parametrize("fused", [False, True])
def test_closed_form_known_answers_hip(fused):
    from sobfu_amd import ops

    S = ops.sobolev_filter(7, 0.1)
    cf.check_all(HipApi(ops, fused), S)


def test_closed_form_solver_iteration_hip():
    """the solver handle's own format (compact state, both passes fused, the marching variants small grids run): one iteration on the
    closed-form potential / update inputs lands on the closed-form psi"""
    from sobfu_amd import ops

    S = ops.sobolev_filter(7, 0.1)
    m0, _, _ = cf.taps_moments(S)
    x, y, z = cf.grid()
    # phi_global == phi_n o psi at the start (identity psi, phi_n = phi_global): nabla_U = w_reg * (-Lap psi) = 0 for the identity ->
    # psi stays the identity EXACTLY, whatever alpha; the warp of an affine phi_n by the identity returns it exactly
    phi = 0.125 + 0.015625 * x - 0.0078125 * y + 0.03125 * z
    v = np.zeros(cf.DIMS[::-1] + (2,), np.float32)
    v[..., 0], v[..., 1] = phi, 1.0
    pg, pn, pnp = dev(v), dev(v), ops.new_volume(cf.DIMS)
    psi = ops.new_field(cf.DIMS)
    ops.init_identity(psi)
    sv = ops.Solver(cf.DIMS, max_iter=3, alpha=0.25, w_reg=0.5)
    rep, hist = sv.iterate(pg, pn, pnp, psi, 3)
    sv.close()
    ident = np.stack([x, y, z], -1)
    assert np.array_equal(psi.cpu().numpy()[..., :3], ident.astype(np.float32)) and float(np.abs(hist).max()) == 0.0
    assert np.array_equal(pnp.cpu().numpy(), v)
