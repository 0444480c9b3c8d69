"""Thin torch-tensor front end over the C ABI (include/sobfu_hip.h).

torch is used for device memory and streams only; every function below is a single C-ABI call on raw device
pointers.  Tensor conventions (float32, contiguous, on a HIP device):
  TSDF volume (Z, Y, X, 2) {tsdf, weight}; vector field (Z, Y, X, 4) (w == 0); Jacobian (Z, Y, X, 4, 4).
Names follow the reference's launcher names (include/sobfu/*.hpp `namespace device`, kfusion/internal.hpp).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import SolverParams, SolverReport, check

_F3 = C.c_float * 3
_F2 = C.c_float * 2
_F7 = C.c_float * 7
_F9 = C.c_float * 9


def _require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("sobfu_amd needs a HIP device (MI355X); there is no CPU fallback")


def _ptr(t: torch.Tensor, dtype=torch.float32):
    if not (t.is_cuda and t.is_contiguous() and t.dtype == dtype):
        raise ValueError(f"expected a contiguous {dtype} tensor on the GPU, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _xyz(t: torch.Tensor):
    Z, Y, X = t.shape[:3]
    return C.c_int(X), C.c_int(Y), C.c_int(Z)


def _f(x):
    return C.c_float(float(x))


def new_volume(dims, device="cuda"):
    X, Y, Z = dims
    return torch.zeros((Z, Y, X, 2), dtype=torch.float32, device=device)


def new_field(dims, device="cuda"):
    X, Y, Z = dims
    return torch.zeros((Z, Y, X, 4), dtype=torch.float32, device=device)


def new_jacobian(dims, device="cuda"):
    X, Y, Z = dims
    return torch.zeros((Z, Y, X, 4, 4), dtype=torch.float32, device=device)


# ---- TSDF --------------------------------------------------------------------------------------
def clear_volume(vol):
    check(_lib.lib().sobfu_hip_clear_volume(_ptr(vol), *_xyz(vol), _stream()), "clear_volume")


def integrate_depth(dists, vol, voxel_size, trunc, eta, R, t, intr):
    Rm = _F9(*[float(v) for v in np.asarray(R, np.float32).reshape(9)])
    tv = _F3(*[float(v) for v in np.asarray(t, np.float32).reshape(3)])
    assert dists.is_cuda and dists.dtype == torch.float32 and dists.stride(1) == 1
    check(_lib.lib().sobfu_hip_integrate_depth(C.c_void_p(dists.data_ptr()), C.c_int(dists.stride(0) * 4),
                                               C.c_int(dists.shape[0]), C.c_int(dists.shape[1]), _ptr(vol), *_xyz(vol),
                                               _F3(*[float(v) for v in voxel_size]), _f(trunc), _f(eta), Rm, tv,
                                               _f(intr[0]), _f(intr[1]), _f(intr[2]), _f(intr[3]), _stream()),
          "integrate_depth")


def tile3_integrate_depth(dists, vol_local, base, voxel_size, trunc, eta, R, t, intr):
    """integrate(depth) into a tile whose local cell (0, 0, 0) is global cell `base` = (xb, yb, zb) (multi-GPU tiles)"""
    Rm = _F9(*[float(v) for v in np.asarray(R, np.float32).reshape(9)])
    tv = _F3(*[float(v) for v in np.asarray(t, np.float32).reshape(3)])
    assert dists.is_cuda and dists.dtype == torch.float32 and dists.stride(1) == 1
    check(_lib.lib().sobfu_hip_tile3_integrate_depth(C.c_void_p(dists.data_ptr()), C.c_int(dists.stride(0) * 4), C.c_int(dists.shape[0]),
                                                     C.c_int(dists.shape[1]), _ptr(vol_local), *_xyz(vol_local), *[C.c_int(int(b)) for b in base],
                                                     _F3(*[float(v) for v in voxel_size]), _f(trunc), _f(eta), Rm, tv, _f(intr[0]),
                                                     _f(intr[1]), _f(intr[2]), _f(intr[3]), _stream()), "tile3_integrate_depth")


def integrate_fuse(phi_global, phi_n_psi, max_weight):
    check(_lib.lib().sobfu_hip_integrate_fuse(_ptr(phi_global), _ptr(phi_n_psi), *_xyz(phi_global), _f(max_weight),
                                              _stream()), "integrate_fuse")


def init_sphere(vol, voxel_size, trunc, eta, centre, radius):
    check(_lib.lib().sobfu_hip_init_sphere(_ptr(vol), *_xyz(vol), _F3(*[float(v) for v in voxel_size]), _f(trunc),
                                           _f(eta), _F3(*[float(np.float32(v)) for v in centre]), _f(np.float32(radius)),
                                           _stream()), "init_sphere")


def init_box(vol, voxel_size, trunc, b):
    check(_lib.lib().sobfu_hip_init_box(_ptr(vol), *_xyz(vol), _F3(*[float(v) for v in voxel_size]), _f(trunc),
                                        _F3(*[float(v) for v in b]), _stream()), "init_box")


def init_ellipsoid(vol, voxel_size, trunc, r):
    check(_lib.lib().sobfu_hip_init_ellipsoid(_ptr(vol), *_xyz(vol), _F3(*[float(v) for v in voxel_size]), _f(trunc),
                                              _F3(*[float(v) for v in r]), _stream()), "init_ellipsoid")


def init_plane(vol, voxel_size, trunc, z):
    check(_lib.lib().sobfu_hip_init_plane(_ptr(vol), *_xyz(vol), _F3(*[float(v) for v in voxel_size]), _f(trunc), _f(z),
                                          _stream()), "init_plane")


def init_torus(vol, voxel_size, trunc, t):
    check(_lib.lib().sobfu_hip_init_torus(_ptr(vol), *_xyz(vol), _F3(*[float(v) for v in voxel_size]), _f(trunc),
                                          _F2(*[float(v) for v in t]), _stream()), "init_torus")


# ---- depth pre-steps -----------------------------------------------------------------------------
def _u16(t):
    # torch has no uint16 arithmetic on all builds; depth images travel as int16 views of the same bits
    if not (t.is_cuda and t.is_contiguous() and t.dtype in (torch.int16, torch.uint16)):
        raise ValueError("depth images must be contiguous 16-bit tensors on the GPU")
    return C.c_void_p(t.data_ptr())


def bilateral_filter(src, ksz, sigma_spatial, sigma_depth):
    dst = torch.empty_like(src)
    rows, cols = src.shape
    check(_lib.lib().sobfu_hip_bilateral_filter(_u16(src), C.c_int(cols * 2), _u16(dst), C.c_int(cols * 2), C.c_int(rows),
                                                C.c_int(cols), C.c_int(ksz), _f(sigma_spatial), _f(sigma_depth), _stream()),
          "bilateral_filter")
    return dst


def truncate_depth(depth, max_dist_m):
    rows, cols = depth.shape
    check(_lib.lib().sobfu_hip_truncate_depth(_u16(depth), C.c_int(cols * 2), C.c_int(rows), C.c_int(cols), _f(max_dist_m),
                                              _stream()), "truncate_depth")


def compute_dists(depth, intr):
    rows, cols = depth.shape
    dists = torch.empty((rows, cols), dtype=torch.float32, device=depth.device)
    check(_lib.lib().sobfu_hip_compute_dists(_u16(depth), C.c_int(cols * 2), _ptr(dists), C.c_int(cols * 4), C.c_int(rows),
                                             C.c_int(cols), _f(intr[0]), _f(intr[1]), _f(intr[2]), _f(intr[3]), _stream()),
          "compute_dists")
    return dists


# ---- vector fields -----------------------------------------------------------------------------
def clear_field(f):
    check(_lib.lib().sobfu_hip_clear_field(_ptr(f), *_xyz(f), _stream()), "clear_field")


def init_identity(psi):
    check(_lib.lib().sobfu_hip_init_identity(_ptr(psi), *_xyz(psi), _stream()), "init_identity")


def apply(phi, phi_warped, psi):
    check(_lib.lib().sobfu_hip_apply(_ptr(phi), _ptr(phi_warped), _ptr(psi), *_xyz(phi), _stream()), "apply")


def estimate_inverse(psi, psi_inv, n_sweeps=48):
    check(_lib.lib().sobfu_hip_estimate_inverse(_ptr(psi), _ptr(psi_inv), *_xyz(psi), C.c_int(n_sweeps), _stream()),
          "estimate_inverse")


def tsdf_gradient(vol, grad):
    check(_lib.lib().sobfu_hip_tsdf_gradient(_ptr(vol), _ptr(grad), *_xyz(vol), _stream()), "tsdf_gradient")


def laplacian(psi, L):
    check(_lib.lib().sobfu_hip_laplacian(_ptr(psi), _ptr(L), *_xyz(psi), _stream()), "laplacian")


def jacobian(psi, J, mode):
    check(_lib.lib().sobfu_hip_jacobian(_ptr(psi), _ptr(J), *_xyz(psi), C.c_int(mode), _stream()), "jacobian")


# ---- solver launchers ----------------------------------------------------------------------------
def sobolev_filter(s, lam):
    out = (C.c_float * 16)()
    check(_lib.lib().sobfu_hip_sobolev_filter(C.c_int(s), _f(np.float32(lam)), out), "sobolev_filter")
    return np.array(out[:s], np.float32)


def _taps(S):
    S = np.asarray(S, np.float32)
    assert S.size >= 7
    return _F7(*[float(v) for v in S[:7]])


def potential_gradient(phi_n_psi, phi_global, grad, L, nabla_U, w_reg):
    check(_lib.lib().sobfu_hip_potential_gradient(_ptr(phi_n_psi), _ptr(phi_global), _ptr(grad), _ptr(L), _ptr(nabla_U),
                                                  _f(w_reg), *_xyz(phi_n_psi), _stream()), "potential_gradient")


def convolution_rows(dst, src, S):
    check(_lib.lib().sobfu_hip_convolution_rows(_ptr(dst), _ptr(src), _taps(S), *_xyz(src), _stream()), "convolution_rows")


def convolution_columns(dst, src, S):
    check(_lib.lib().sobfu_hip_convolution_columns(_ptr(dst), _ptr(src), _taps(S), *_xyz(src), _stream()), "convolution_columns")


def convolution_depth(dst, src, S):
    check(_lib.lib().sobfu_hip_convolution_depth(_ptr(dst), _ptr(src), _taps(S), *_xyz(src), _stream()), "convolution_depth")


def update_psi(psi, nabla_U_S, updates, alpha):
    check(_lib.lib().sobfu_hip_update_psi(_ptr(psi), _ptr(nabla_U_S), _ptr(updates), _f(alpha), *_xyz(psi), _stream()),
          "update_psi")


# ---- reductions ----------------------------------------------------------------------------------
def reduce_config(n):
    b, t = C.c_int(), C.c_int()
    check(_lib.lib().sobfu_hip_reduce_config(C.c_int(n), C.byref(b), C.byref(t)), "reduce_config")
    return b.value, t.value


def _scratch(dev):
    return torch.empty(65536 * 2, dtype=torch.float32, device=dev)


def data_energy(phi_global, phi_n):
    out, sc = C.c_float(), _scratch(phi_global.device)
    check(_lib.lib().sobfu_hip_data_energy(_ptr(phi_global), _ptr(phi_n), C.c_int(phi_global.numel() // 2), _ptr(sc),
                                           C.byref(out), _stream()), "data_energy")
    return out.value


def reg_energy_sobolev(J):
    out, sc = C.c_float(), _scratch(J.device)
    check(_lib.lib().sobfu_hip_reg_energy_sobolev(_ptr(J), C.c_int(J.numel() // 16), _ptr(sc), C.byref(out), _stream()),
          "reg_energy_sobolev")
    return out.value


def reg_energy_sobolev_from_psi(psi):
    out, sc = C.c_float(), _scratch(psi.device)
    check(_lib.lib().sobfu_hip_reg_energy_sobolev_from_psi(_ptr(psi), *_xyz(psi), _ptr(sc), C.byref(out), _stream()),
          "reg_energy_sobolev_from_psi")
    return out.value


def max_update_norm(updates):
    out, sc = _F2(), _scratch(updates.device)
    check(_lib.lib().sobfu_hip_max_update_norm(_ptr(updates), C.c_int(updates.numel() // 4), _ptr(sc), out, _stream()),
          "max_update_norm")
    return out[0], out[1]


# ---- fused passes --------------------------------------------------------------------------------
def fused_potential_gradient(phi_n_psi, phi_global, psi, nabla_U, w_reg):
    check(_lib.lib().sobfu_hip_fused_potential_gradient(_ptr(phi_n_psi), _ptr(phi_global), _ptr(psi), _ptr(nabla_U),
                                                        _f(w_reg), *_xyz(psi), _stream()), "fused_potential_gradient")


def fused_smooth_update_apply(nabla_U, psi, phi_n, phi_n_psi, S, alpha, updates=None, slots=None):
    """Returns max ||u|| (sqrt rounded down, as Reductor::max_update_norm().x)."""
    if slots is None:
        slots = torch.zeros(256, dtype=torch.int32, device=psi.device)
    up = _ptr(updates) if updates is not None else None
    check(_lib.lib().sobfu_hip_fused_smooth_update_apply(_ptr(nabla_U), _ptr(psi), _ptr(phi_n), _ptr(phi_n_psi), up,
                                                         _ptr(slots, torch.int32), _taps(S), _f(alpha), *_xyz(psi),
                                                         _stream()), "fused_smooth_update_apply")
    m = np.float32(slots.max().cpu().numpy().view(np.float32))
    r = np.sqrt(m, dtype=np.float32)
    if r > 0 and np.float64(r) * np.float64(r) > np.float64(m):
        r = np.nextafter(r, np.float32(-np.inf), dtype=np.float32)
    return float(r)


# ---- solver handle -------------------------------------------------------------------------------
class Solver:
    """sobfu::cuda::Solver (reference include/sobfu/solver.hpp:52-101) over the opaque C handle."""

    def __init__(self, dims, *, max_iter, alpha, w_reg, s=7, lam=0.1, max_update_norm=-1.0, verbosity=0, quiet=True):
        _require_gpu()
        self.dims = tuple(int(d) for d in dims)
        self.params = SolverParams(verbosity, max_iter, s, max_update_norm, np.float32(lam), alpha, w_reg)
        self._h = C.c_void_p()
        check(_lib.lib().sobfu_hip_solver_create(C.byref(self._h), *[C.c_int(d) for d in self.dims],
                                                 C.byref(self.params)), "solver_create")
        self.log_lines = []
        self._cb = _lib.LOG_FN(self._on_log)
        self._quiet = quiet
        check(_lib.lib().sobfu_hip_solver_set_logger(self._h, self._cb, None), "solver_set_logger")

    def _on_log(self, line, _user):
        self.log_lines.append(line.decode())
        if not self._quiet:
            print(line.decode())

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().sobfu_hip_solver_destroy(self._h)
            self._h = None

    __del__ = close

    def workspace_bytes(self):
        return int(_lib.lib().sobfu_hip_solver_workspace_bytes(self._h))

    def set_compact(self, enable=True):
        check(_lib.lib().sobfu_hip_solver_set_compact(self._h, C.c_int(1 if enable else 0)), "set_compact")

    def set_profiling(self, stride=1):
        """HIP events around the two launches of every stride-th iteration (0 / False = off)"""
        check(_lib.lib().sobfu_hip_solver_set_profiling(self._h, C.c_int(int(stride))), "set_profiling")

    def get_profile(self, reset=True):
        """(ms in pass A, ms in pass B, iterations timed) from HIP events on the solver's stream."""
        a, b, n = C.c_float(), C.c_float(), C.c_int()
        check(_lib.lib().sobfu_hip_solver_get_profile(self._h, C.byref(a), C.byref(b), C.byref(n), C.c_int(1 if reset else 0)),
              "get_profile")
        return a.value, b.value, n.value

    def keep_updates(self, keep=True):
        check(_lib.lib().sobfu_hip_solver_keep_updates(self._h, C.c_int(1 if keep else 0)), "keep_updates")

    def updates(self):
        """Tensor view (no copy) of Reductor::updates (reference src/sobfu/reductor.cpp:26)."""
        p = _lib.lib().sobfu_hip_solver_updates(self._h)
        if not p:
            raise _lib.HipError("solver_updates: allocation failed")
        X, Y, Z = self.dims

        class _Raw:
            __cuda_array_interface__ = {"shape": (Z, Y, X, 4), "typestr": "<f4", "data": (int(p), False), "version": 2}

        return torch.as_tensor(_Raw(), device="cuda")

    def estimate_psi(self, phi_global, phi_global_psi_inv, phi_n, phi_n_psi, psi, psi_inv):
        """Solver::estimate_psi (reference src/sobfu/solver.cpp:69-101).  Returns (report, per-iteration max norms)."""
        rep = SolverReport()
        hist = (C.c_float * max(1, self.params.max_iter))()
        check(_lib.lib().sobfu_hip_solver_estimate_psi(self._h, _ptr(phi_global), _ptr(phi_global_psi_inv), _ptr(phi_n),
                                                       _ptr(phi_n_psi), _ptr(psi), _ptr(psi_inv), C.byref(rep), hist,
                                                       _stream()), "solver_estimate_psi")
        return rep, np.array(hist[:rep.iterations], np.float32)

    def begin(self, phi_global, phi_n, phi_n_psi, psi, max_iters):
        """the loop in pieces (sobfu_hip_solver_begin / step / end): begin enters the iteration format, step(n) ENQUEUES n
        iterations without synchronising, end() synchronises and returns what iterate() would"""
        self._session = (phi_global, phi_n, phi_n_psi, psi, int(max_iters))  # keeps the buffers alive
        check(_lib.lib().sobfu_hip_solver_begin(self._h, _ptr(phi_global), _ptr(phi_n), _ptr(phi_n_psi), _ptr(psi), C.c_int(int(max_iters)),
                                                _stream()), "solver_begin")

    def step(self, n_iters):
        check(_lib.lib().sobfu_hip_solver_step(self._h, C.c_int(int(n_iters)), _stream()), "solver_step")

    def end(self):
        rep = SolverReport()
        hist = (C.c_float * max(1, self._session[4] if getattr(self, "_session", None) else 1))()
        check(_lib.lib().sobfu_hip_solver_end(self._h, C.byref(rep), hist, _stream()), "solver_end")
        self._session = None
        return rep, np.array(hist[:rep.iterations], np.float32)

    def iterate(self, phi_global, phi_n, phi_n_psi, psi, n_iters):
        rep = SolverReport()
        hist = (C.c_float * max(1, n_iters))()
        check(_lib.lib().sobfu_hip_solver_iterate(self._h, _ptr(phi_global), _ptr(phi_n), _ptr(phi_n_psi), _ptr(psi),
                                                  C.c_int(n_iters), C.byref(rep), hist, _stream()), "solver_iterate")
        return rep, np.array(hist[:rep.iterations], np.float32)


# ---- marching cubes (include/kfusion/internal.hpp:213-225) ---------------------------------------------------------------
def mc_workspace(vol):
    """scratch for mc_occupied_voxels / mc_offsets on volumes of this shape (keep it between frames: no allocation per call)"""
    _lib.lib().sobfu_hip_mc_workspace_bytes.restype = C.c_size_t
    n = int(_lib.lib().sobfu_hip_mc_workspace_bytes(*_xyz(vol)))
    return torch.empty(n, dtype=torch.uint8, device=vol.device)


def _ws(workspace):
    if workspace is None:
        return None, C.c_size_t(0)
    return _ptr(workspace, torch.uint8), C.c_size_t(workspace.numel())


def mc_occupied_voxels(vol, max_size, workspace=None):
    """getOccupiedVoxels -> (occupied int32 (3, max_size) on the GPU: voxel index / vertex count / vertex offset rows, count)"""
    occ = torch.zeros((3, int(max_size)), dtype=torch.int32, device=vol.device)
    n = C.c_int(0)
    check(_lib.lib().sobfu_hip_mc_occupied_voxels(_stream(), _ptr(vol), *_xyz(vol), _ptr(occ, torch.int32), C.c_int(occ.shape[1]),
                                                  C.c_int(int(max_size)), C.byref(n), *_ws(workspace)), "mc_occupied_voxels")
    return occ, n.value


def mc_offsets(occ, count, workspace=None):
    """computeOffsetsAndTotalVertices: row 2 = exclusive scan of row 1 -> total vertices"""
    total = C.c_int(0)
    check(_lib.lib().sobfu_hip_mc_offsets(_stream(), _ptr(occ, torch.int32), C.c_int(occ.shape[1]), C.c_int(int(count)), C.byref(total),
                                          *_ws(workspace)), "mc_offsets")
    return total.value


def mc_generate_triangles(vol, occ, count, volume_size, R, t, vertices, normals):
    """generateTriangles into float4 buffers (n, 4)"""
    Rm = _F9(*[float(v) for v in np.asarray(R, np.float32).reshape(9)])
    tv = _F3(*[float(v) for v in np.asarray(t, np.float32).reshape(3)])
    assert vertices.shape == normals.shape and vertices.shape[1] == 4
    check(_lib.lib().sobfu_hip_mc_generate_triangles(_stream(), _ptr(vol), *_xyz(vol), _ptr(occ, torch.int32), C.c_int(occ.shape[1]),
                                                     C.c_int(int(count)), _f(volume_size[0]), _f(volume_size[1]), _f(volume_size[2]), Rm, tv,
                                                     _ptr(vertices), _ptr(normals), C.c_int(vertices.shape[0])), "mc_generate_triangles")


def marching_cubes(vol, volume_size, R=np.eye(3), t=(0, 0, 0), max_voxels=2_000_000, max_vertices=None, workspace=None):
    """kfusion::cuda::MarchingCubes::run (src/kfusion/marching_cubes.cpp:23-79) -> (vertices (n, 4), normals (n, 4)) GPU tensors"""
    max_vertices = max_vertices or 3 * max_voxels
    occ, count = mc_occupied_voxels(vol, max_voxels, workspace)
    if count == 0:
        e = torch.zeros((0, 4), dtype=torch.float32, device=vol.device)
        return e, e.clone()
    total = min(mc_offsets(occ, count, workspace), max_vertices // 3 * 3)  # whole triangles only
    v = torch.zeros((max_vertices, 4), dtype=torch.float32, device=vol.device)
    n = torch.zeros_like(v)
    mc_generate_triangles(vol, occ, count, volume_size, R, t, v, n)
    return v[:total], n[:total]
