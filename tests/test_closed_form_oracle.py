"""Closed-form known answers (tests/closed_form.py) on the ORACLE: a check of the CPU restatement that does not come from any
implementation of the operators -- the hot path (convolution, update, warp, potential gradient) has no reference-held vectors
(reference test/solver_test.cpp:109-208 asserts nothing), so its pin is these identities plus SURVEY Appendix B."""
import numpy as np

import closed_form as cf


class OracleApi:
    def __init__(self, O):
        self.O = O

    def _field(self, a):
        f = self.O.new_field(cf.DIMS)
        f[..., :a.shape[-1]] = a
        return f

    def _vol(self, tsdf, weight=1.0):
        v = self.O.new_volume(cf.DIMS)
        v[..., 0], v[..., 1] = tsdf, weight
        return v

    def run_conv(self, f, S):
        O = self.O
        src, dst = self._field(f[..., :3]), O.new_field(cf.DIMS)
        O.convolution_rows(dst, src, S)
        O.convolution_columns(dst, src, S)
        O.convolution_depth(dst, src, S)
        return dst

    def run_apply(self, phi, psi, weight=1.0):
        out = self.O.new_volume(cf.DIMS)
        self.O.apply(self._vol(phi, weight), out, self._field(psi))
        return out

    def run_potential_gradient(self, F, G, psi, w_reg):
        O = self.O
        g, L, nU = O.new_field(cf.DIMS), O.new_field(cf.DIMS), O.new_field(cf.DIMS)
        vf, vg = self._vol(F), self._vol(G)
        O.tsdf_gradient(vf, g)
        O.laplacian(self._field(psi), L)
        O.potential_gradient(vf, vg, g, L, nU, w_reg)
        return nU

    def run_smooth_update_apply(self, nU, psi, phi, S, alpha):
        O = self.O
        nUS, upd = O.new_field(cf.DIMS), O.new_field(cf.DIMS)
        src = self._field(nU)
        O.convolution_rows(nUS, src, S)
        O.convolution_columns(nUS, src, S)
        O.convolution_depth(nUS, src, S)
        p = self._field(psi)
        O.update_psi(p, nUS, upd, alpha)
        out = O.new_volume(cf.DIMS)
        O.apply(self._vol(phi), out, p)
        return p, out


    def run_inverse(self, psi, sweeps):
        inv = self.O.new_field(cf.DIMS)
        self.O.estimate_inverse(self._field(psi), inv, sweeps)
        return inv

    def run_max_norm(self, updates):
        return self.O.max_update_norm(np.ascontiguousarray(updates, np.float32))

    def run_fuse(self, g, n, max_weight):
        vg, vn = self.O.new_volume(cf.DIMS), self.O.new_volume(cf.DIMS)
        vg[...], vn[...] = g, n
        self.O.integrate_fuse(vg, vn, max_weight)
        return vg


def test_closed_form_known_answers_oracle(oracle):
    S = oracle.sobolev_filter(7, 0.1)
    m0, m1, m2 = cf.taps_moments(S)
    assert abs(m0 - 1.0) < 1e-7 and abs(m1) < 1e-9 and 0.1 < m2 < 0.2  # unit sum, symmetric; the second moment is what "quadratic" tests
    worst = cf.check_all(OracleApi(oracle), S)
    assert worst["conv_constant_ulp"] <= 4 and worst["warp_abs_err"] < 1e-6
