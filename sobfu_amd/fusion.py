"""The per-frame pipeline of the reference's SobFusion::operator() (src/sobfu/sob_fusion.cpp:71-145) on one GPU, through the C ABI:
depth pre-steps -> frame 0: integrate into phi_global | frame n: clear + integrate phi_n, estimate_psi (iterations + 48-sweep
inverse + canonical warp), fuse phi_n o psi into phi_global.  The Python twin of sobfu::SobFusion in include/sobfu_amd/sobfu.hpp
(what apps/sobfu_headless drives); bench.py times it for the frames/s figures (BASELINE config 5)."""
from __future__ import annotations


class SobFusion:
    """params: a dict as sobfu_amd.params.read_ini returns (dims, vs, trunc, eta, max_weight, intr, R, t, start_frame, bilateral,
    trunc_depth, max_iter, max_update_norm, s, lam, alpha, w_reg)."""

    def __init__(self, params, max_iter=None):
        from . import ops

        self.ops, self.P = ops, params
        self.max_iter = int(params["max_iter"] if max_iter is None else max_iter)
        self.frame = 0
        self.phi_global = self.phi_global_psi_inv = self.phi_n = self.phi_n_psi = self.psi = self.psi_inv = self.solver = None
        self.last_report = None

    def __call__(self, depth_u16):
        ops, P = self.ops, self.P
        dims, vs = P["dims"], tuple(float(v) for v in P["vs"])
        ks, ss, sd = P["bilateral"]
        d = ops.bilateral_filter(depth_u16, ks, ss, sd)                                       # sob_fusion.cpp:78
        ops.truncate_depth(d, P["trunc_depth"])                                               # :85
        dists = ops.compute_dists(d, P["intr"])                                               # :91
        if self.frame == 0:                                                                   # :93-123
            self.phi_global = ops.new_volume(dims)
            ops.integrate_depth(dists, self.phi_global, vs, P["trunc"], P["eta"], P["R"], P["t"], P["intr"])
            self.phi_global_psi_inv, self.phi_n, self.phi_n_psi = ops.new_volume(dims), ops.new_volume(dims), ops.new_volume(dims)
            self.psi, self.psi_inv = ops.new_field(dims), ops.new_field(dims)
            ops.init_identity(self.psi)
            ops.init_identity(self.psi_inv)
            self.solver = ops.Solver(dims, max_iter=self.max_iter, alpha=P["alpha"], w_reg=P["w_reg"], s=P["s"], lam=P["lam"],
                                     max_update_norm=P["max_update_norm"])
            self.frame += 1
            return None
        ops.clear_volume(self.phi_n)                                                          # :129
        ops.integrate_depth(dists, self.phi_n, vs, P["trunc"], P["eta"], P["R"], P["t"], P["intr"])  # :130
        if self.frame < P["start_frame"]:                                                     # :136-139
            ops.integrate_fuse(self.phi_global, self.phi_n, P["max_weight"])
        else:
            self.last_report = self.solver.estimate_psi(self.phi_global, self.phi_global_psi_inv, self.phi_n, self.phi_n_psi, self.psi,
                                                        self.psi_inv)                        # :141
            ops.integrate_fuse(self.phi_global, self.phi_n_psi, P["max_weight"])              # :142
        self.frame += 1
        return self.last_report

    def close(self):
        if self.solver is not None:
            self.solver.close()
            self.solver = None
