"""ctypes binding of the CPU parity oracle (oracle/sobfu_oracle.c).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's
``cpu_baseline`` leg -- as the checker, never as the product path.  Nothing under sobfu_amd/ imports it.

All arrays are numpy, C-contiguous, float32:
  TSDF volume   (Z, Y, X, 2)   {tsdf, weight}     (x fastest, reference layout idx = x + X*(y + Y*z))
  vector field  (Z, Y, X, 4)   float4, w == 0
  Jacobian      (Z, Y, X, 4, 4)
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(force: bool = False) -> None:
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    tgt = [os.path.join(_HERE, n) for n in ("liboracle.so", "liboracle_v3.so")]
    src = os.path.join(_HERE, "sobfu_oracle.c")
    if not force and all(os.path.exists(t) and os.path.getmtime(t) >= os.path.getmtime(src) for t in tgt):
        return
    subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


def _cpu_has_v3() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    fl = set(line.split(":", 1)[1].split())
                    return {"avx2", "fma", "bmi2"} <= fl
    except OSError:
        pass
    return False


class SolverParams(C.Structure):
    _fields_ = [("verbosity", C.c_int), ("max_iter", C.c_int), ("s", C.c_int), ("max_update_norm", C.c_float),
                ("lambda_", C.c_float), ("alpha", C.c_float), ("w_reg", C.c_float)]


_lib = None


def usable_cpus() -> int:
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota.  A container that shows 256 CPUs but
    is throttled to 16 runs the OpenMP loops on all of them several times slower than on 16."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max":
            n = min(n, max(1, int(float(q) / float(p))))
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and p > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return n


def _load(path):
    L = C.CDLL(path)
    L.so_data_energy.restype = C.c_float
    L.so_reg_energy_sobolev.restype = C.c_float
    L.so_estimate_psi.restype = C.c_int
    L.so_sobolev_filter.restype = C.c_int
    L.so_num_threads.restype = C.c_int
    L.so_set_num_threads(C.c_int(min(int(L.so_num_threads()), usable_cpus())))
    return L


def lib():
    global _lib
    if _lib is None:
        name = "liboracle_v3.so" if _cpu_has_v3() else "liboracle.so"
        path = os.path.join(_HERE, name)
        if not os.path.exists(path):
            build()
        _lib = _load(path)
    return _lib


def build_mutant(k: int, out_dir: str) -> str:
    """Negative control (tests/test_oracle_mutants.py): this restatement with ONE deliberate deviation on the hot path
    (-DSO_MUTANT=k, see the top of sobfu_oracle.c), built into out_dir.  Never loaded by anything but that test."""
    out = os.path.join(out_dir, f"liboracle_mutant{k}.so")
    subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", f"-DSO_MUTANT={int(k)}",
                           "-o", out, os.path.join(_HERE, "sobfu_oracle.c"), "-lm"])
    return out


def build_nvcc(out_dir: str) -> str:
    """This restatement evaluated the way the reference AS BUILT by nvcc may evaluate it (-DSO_NVCC_MODE=1: flush-to-zero, approximate
    divide / sqrt / powf / __expf, fmad contraction -- see the top of sobfu_oracle.c), built into out_dir.  x86-64-v3 only: the FTZ
    emulation needs hardware fma.  Loaded by tests/test_nvcc_distance.py and tools/nvcc_distance.py, nothing else."""
    if not _cpu_has_v3():
        raise RuntimeError("SO_NVCC_MODE needs avx2+fma (flush-to-zero of fmaf goes through the hardware instruction)")
    out = os.path.join(out_dir, "liboracle_nvcc.so")
    subprocess.check_call(["gcc", "-O3", "-fPIC", "-shared", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-march=x86-64-v3",
                           "-DSO_NVCC_MODE=1", "-o", out, os.path.join(_HERE, "sobfu_oracle.c"), "-lm"])
    return out


class nvcc_mode:
    """`with nvcc_mode(path, seed):` -- oracle calls inside run on the SO_NVCC_MODE build with FTZ|DAZ switched on in every OpenMP
    thread; on exit the IEEE library and the threads' IEEE behaviour are restored."""

    def __init__(self, path, seed=1):
        self.path, self.seed = path, seed

    def __enter__(self):
        global _lib
        self.old = _lib if _lib is not None else lib()
        _lib = _load(self.path)
        _lib.so_nvcc_set_seed(C.c_uint(self.seed))
        _lib.so_nvcc_ftz(C.c_int(1))
        return self

    def __exit__(self, *exc):
        global _lib
        _lib.so_nvcc_ftz(C.c_int(0))
        _lib = self.old
        return False


class use_library:
    """context manager: every function of this module calls the given build of the oracle instead of the regular one"""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _lib
        lib()
        self.saved, _lib = _lib, _load(self.path)
        return self

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


def _p(a):
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be C-contiguous"
    return a.ctypes.data_as(C.c_void_p)


def _f(x):
    return C.c_float(float(x))


def _dims(a):
    Z, Y, X = a.shape[:3]
    return C.c_int(X), C.c_int(Y), C.c_int(Z)


def new_volume(dims):
    X, Y, Z = dims
    return np.zeros((Z, Y, X, 2), np.float32)


def new_field(dims):
    X, Y, Z = dims
    return np.zeros((Z, Y, X, 4), np.float32)


# ---- TSDF ------------------------------------------------------------------------------------
def clear_volume(vol):
    lib().so_clear_volume(_p(vol), *_dims(vol))


def init_sphere(vol, voxel_size, trunc, eta, centre, radius):
    lib().so_init_sphere(_p(vol), *_dims(vol), _f(voxel_size[0]), _f(voxel_size[1]), _f(voxel_size[2]), _f(trunc),
                         _f(eta), _f(centre[0]), _f(centre[1]), _f(centre[2]), _f(radius))


def init_box(vol, voxel_size, trunc, b):
    lib().so_init_box(_p(vol), *_dims(vol), _f(voxel_size[0]), _f(voxel_size[1]), _f(voxel_size[2]), _f(trunc),
                      _f(b[0]), _f(b[1]), _f(b[2]))


def init_ellipsoid(vol, voxel_size, trunc, r):
    lib().so_init_ellipsoid(_p(vol), *_dims(vol), _f(voxel_size[0]), _f(voxel_size[1]), _f(voxel_size[2]), _f(trunc),
                            _f(r[0]), _f(r[1]), _f(r[2]))


def init_plane(vol, voxel_size, trunc, z):
    lib().so_init_plane(_p(vol), *_dims(vol), _f(voxel_size[0]), _f(voxel_size[1]), _f(voxel_size[2]), _f(trunc), _f(z))


def init_torus(vol, voxel_size, trunc, t):
    lib().so_init_torus(_p(vol), *_dims(vol), _f(voxel_size[0]), _f(voxel_size[1]), _f(voxel_size[2]), _f(trunc),
                        _f(t[0]), _f(t[1]))


def integrate_depth(dists, vol, voxel_size, trunc, eta, R, t, intr):
    """dists: (rows, cols) float32 (any row stride); R: 3x3 vol->cam rotation; t: translation; intr (fx,fy,cx,cy)."""
    assert dists.dtype == np.float32 and dists.strides[1] == 4
    Rm = np.ascontiguousarray(np.asarray(R, np.float32).reshape(9))
    tv = np.ascontiguousarray(np.asarray(t, np.float32).reshape(3))
    lib().so_integrate_depth(C.c_void_p(dists.ctypes.data), C.c_int(dists.strides[0]), C.c_int(dists.shape[0]),
                             C.c_int(dists.shape[1]), _p(vol), *_dims(vol), _f(voxel_size[0]), _f(voxel_size[1]),
                             _f(voxel_size[2]), _f(trunc), _f(eta), _p(Rm), _p(tv), _f(intr[0]), _f(intr[1]),
                             _f(intr[2]), _f(intr[3]))


def integrate_fuse(phi_global, phi_n_psi, max_weight):
    lib().so_integrate_fuse(_p(phi_global), _p(phi_n_psi), *_dims(phi_global), _f(max_weight))


# ---- depth pre-steps -----------------------------------------------------------------------------
def bilateral(src, ksz, sigma_spatial, sigma_depth):
    assert src.dtype == np.uint16 and src.flags["C_CONTIGUOUS"]
    dst = np.zeros_like(src)
    lib().so_bilateral(_p(src), C.c_int(src.strides[0]), _p(dst), C.c_int(dst.strides[0]), C.c_int(src.shape[0]),
                       C.c_int(src.shape[1]), C.c_int(ksz), _f(sigma_spatial), _f(sigma_depth))
    return dst


def truncate_depth(depth, max_dist_m):
    lib().so_truncate_depth(_p(depth), C.c_int(depth.strides[0]), C.c_int(depth.shape[0]), C.c_int(depth.shape[1]),
                            _f(max_dist_m))


def compute_dists(depth, intr):
    dists = np.zeros(depth.shape, np.float32)
    lib().so_compute_dists(_p(depth), C.c_int(depth.strides[0]), _p(dists), C.c_int(dists.strides[0]),
                           C.c_int(depth.shape[0]), C.c_int(depth.shape[1]), _f(intr[0]), _f(intr[1]), _f(intr[2]),
                           _f(intr[3]))
    return dists


# ---- vector fields ---------------------------------------------------------------------------
def clear_field(f):
    lib().so_clear_field(_p(f), *_dims(f))


def init_identity(psi):
    lib().so_init_identity(_p(psi), *_dims(psi))


def apply(phi, phi_warped, psi):
    lib().so_apply(_p(phi), _p(phi_warped), _p(psi), *_dims(phi))


def apply_tile(phi_full, phi_warped_tile, psi_tile):
    """apply on a tile (any local extents): phi_full is the whole volume, psi holds absolute voxel coordinates"""
    lib().so_apply_tile3(_p(phi_full), *_dims(phi_full), _p(phi_warped_tile), _p(psi_tile), *_dims(psi_tile))


def estimate_inverse(psi, psi_inv, n_iters=48):
    lib().so_estimate_inverse(_p(psi), _p(psi_inv), *_dims(psi), C.c_int(n_iters))


def tsdf_gradient(vol, grad):
    lib().so_tsdf_gradient(_p(vol), _p(grad), *_dims(vol))


def laplacian(psi, L):
    lib().so_laplacian(_p(psi), _p(L), *_dims(psi))


def jacobian(psi, J, mode):
    lib().so_jacobian(_p(psi), _p(J), *_dims(psi), C.c_int(mode))


# ---- solver pieces ---------------------------------------------------------------------------
def potential_gradient(phi_n_psi, phi_global, grad, L, nabla_U, w_reg):
    lib().so_potential_gradient(_p(phi_n_psi), _p(phi_global), _p(grad), _p(L), _p(nabla_U), _f(w_reg),
                                *_dims(phi_n_psi))


def _taps(S):
    S = np.ascontiguousarray(np.asarray(S, np.float32))
    assert S.size >= 7
    return S


def convolution_rows(dst, src, S):
    S = _taps(S)
    lib().so_convolution_rows(_p(dst), _p(src), _p(S), *_dims(src))


def convolution_columns(dst, src, S):
    S = _taps(S)
    lib().so_convolution_columns(_p(dst), _p(src), _p(S), *_dims(src))


def convolution_depth(dst, src, S):
    S = _taps(S)
    lib().so_convolution_depth(_p(dst), _p(src), _p(S), *_dims(src))


def update_psi(psi, nabla_U_S, updates, alpha):
    lib().so_update_psi(_p(psi), _p(nabla_U_S), _p(updates), _f(alpha), *_dims(psi))


def sobolev_filter(s, lam):
    h = np.zeros(16, np.float32)
    rc = lib().so_sobolev_filter(C.c_int(s), _f(np.float32(lam)), _p(h))
    if rc != 0:
        raise ValueError(f"(s={s}, lambda={lam}) is not in the reference's filter table (src/sobfu/solver.cpp:160-251)")
    return h[:s].copy()


# ---- reductions ------------------------------------------------------------------------------
def reduce_config(n):
    b, t = C.c_int(), C.c_int()
    lib().so_reduce_config(C.c_int(n), C.byref(b), C.byref(t))
    return b.value, t.value


def data_energy(phi_global, phi_n):
    return float(lib().so_data_energy(_p(phi_global), _p(phi_n), C.c_int(phi_global.size // 2)))


def reg_energy_sobolev(J):
    return float(lib().so_reg_energy_sobolev(_p(J), C.c_int(J.size // 16)))


def max_update_norm(updates):
    out = np.zeros(2, np.float32)
    lib().so_max_update_norm(_p(updates), C.c_int(updates.size // 4), _p(out))
    return float(out[0]), float(out[1])


# ---- solver ----------------------------------------------------------------------------------
def estimate_psi(phi_global, phi_n, psi, *, max_iter, alpha, w_reg, s=7, lam=0.1, max_update_norm=-1.0, verbosity=0,
                 compute_jacobian=True, inverse_iters=48, phi_n_psi=None):
    """Runs sobfu::device::estimate_psi (solver.cu:85-205) on numpy arrays.  psi is updated in place.

    Returns dict(iters, phi_n_psi, psi_inv, phi_global_psi_inv, trace[iters, 4] = e_data, e_reg, max_norm, max_idx).
    """
    Z, Y, X = psi.shape[:3]
    dims = (X, Y, Z)
    if phi_n_psi is None:
        phi_n_psi = new_volume(dims)
    out_inv = new_volume(dims)
    psi_inv = new_field(dims)
    grad, L, nU, nUS, upd = (new_field(dims) for _ in range(5))
    J = np.zeros((Z, Y, X, 4, 4), np.float32) if (compute_jacobian or verbosity) else None
    trace = np.full((max(max_iter, 1), 4), np.nan, np.float32)
    p = SolverParams(verbosity, max_iter, s, max_update_norm, np.float32(lam), alpha, w_reg)
    it = lib().so_estimate_psi(_p(phi_global), _p(out_inv), _p(phi_n), _p(phi_n_psi), _p(psi), _p(psi_inv), _p(grad),
                               _p(L), _p(nU), _p(nUS), _p(upd), _p(J) if J is not None else None, C.c_int(X),
                               C.c_int(Y), C.c_int(Z), C.byref(p), C.c_int(1 if compute_jacobian else 0),
                               C.c_int(inverse_iters), _p(trace))
    if it < 0:
        raise ValueError("unsupported (s, lambda)")
    return dict(iters=it, phi_n_psi=phi_n_psi, psi_inv=psi_inv, phi_global_psi_inv=out_inv, trace=trace[:it],
                updates=upd, nabla_U=nU, nabla_U_S=nUS)


def num_threads():
    return lib().so_num_threads()


def set_num_threads(n):
    lib().so_set_num_threads(C.c_int(n))


# ---- marching cubes (parity unpinned; see the block comment in sobfu_oracle.c) -------------------------------------
def mc_num_verts(cube):
    lib().so_mc_num_verts.restype = C.c_int
    return int(lib().so_mc_num_verts(C.c_int(int(cube))))


def mc_occupied_voxels(vol, max_size):
    """-> (occupied int32 (3, max_size): voxel index / vertex count / vertex offset rows, count)"""
    occ = np.zeros((3, int(max_size)), np.int32)
    lib().so_mc_occupied_voxels.restype = C.c_int
    n = lib().so_mc_occupied_voxels(_p(vol), *_dims(vol), _p(occ), C.c_int(occ.shape[1]), C.c_int(int(max_size)))
    return occ, int(n)


def mc_offsets(occ, count):
    lib().so_mc_offsets.restype = C.c_int
    return int(lib().so_mc_offsets(_p(occ), C.c_int(occ.shape[1]), C.c_int(int(count))))


def mc_generate_triangles(vol, occ, count, volume_size, R, t, max_vertices):
    """-> (vertices, normals) float32 (max_vertices, 4); rows past the total stay zero"""
    v, n = np.zeros((int(max_vertices), 4), np.float32), np.zeros((int(max_vertices), 4), np.float32)
    R = np.ascontiguousarray(R, np.float32).reshape(9)
    t = np.ascontiguousarray(t, np.float32).reshape(3)
    lib().so_mc_generate_triangles(_p(vol), *_dims(vol), _p(occ), C.c_int(occ.shape[1]), C.c_int(int(count)), _f(volume_size[0]),
                                   _f(volume_size[1]), _f(volume_size[2]), _p(R), _p(t), _p(v), _p(n), C.c_int(int(max_vertices)))
    return v, n


def marching_cubes(vol, volume_size, R=np.eye(3), t=(0, 0, 0), max_voxels=None, max_vertices=None):
    """kfusion::cuda::MarchingCubes::run (src/kfusion/marching_cubes.cpp:23-79) -> (vertices (n, 4), normals (n, 4))"""
    max_voxels = max_voxels or 2_000_000
    max_vertices = max_vertices or 3 * max_voxels
    occ, count = mc_occupied_voxels(vol, max_voxels)
    if count == 0:
        return np.zeros((0, 4), np.float32), np.zeros((0, 4), np.float32)
    total = min(mc_offsets(occ, count), max_vertices // 3 * 3)  # whole triangles only
    v, n = mc_generate_triangles(vol, occ, count, volume_size, R, t, max_vertices)
    return v[:total], n[:total]
