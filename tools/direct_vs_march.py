"""Timing probe: a WHOLE 128^3 grid evaluated lane-per-cell ("direct" boxes: every tap read through the caches, no LDS, no barrier, no
z pipeline, 4096 independent workgroups) against the z-marching passes, compact format, same launches the tile loop uses
(sobfu_hip_tile3_potential_gradient / sobfu_hip_tile3_smooth_update_apply with thin = 1 / 0).  Bits compared.
    python tools/direct_vs_march.py            # DVM_DIM=128 DVM_ITERS=200
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from sobfu_amd import _lib, ops

L = _lib.lib()
dim = int(os.environ.get("DVM_DIM", "128"))
iters = int(os.environ.get("DVM_ITERS", "200"))
P = bench.boxing_params(dim)
dims = P["dims"]
N = dim ** 3
c0, c1, r = bench.sphere_pair(P)
pg, pn = ops.new_volume(dims), ops.new_volume(dims)
ops.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r)
ops.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
S = ops.sobolev_filter(P["s"], P["lam"])
taps = (C.c_float * 7)(*[float(v) for v in np.asarray(S, np.float32).reshape(-1)[:7]])
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
I6 = C.c_int * 6
box = I6(0, dim, 0, dim, 0, dim)
f32 = lambda n: torch.zeros(n, dtype=torch.float32, device="cuda")


def state():
    psi4 = ops.new_field(dims)
    ops.init_identity(psi4)
    psi4[..., :3] += 0.3 * torch.sin(torch.arange(N * 3, device="cuda", dtype=torch.float32).reshape(dim, dim, dim, 3) * 0.37)
    psi3, g, n1, f, nu = f32(3 * N), f32(N), f32(N), f32(N), f32(3 * N)
    _lib.check(L.sobfu_hip_pack_vec3(p(psi4), p(psi3), C.c_size_t(N), st), "pack")
    _lib.check(L.sobfu_hip_extract_tsdf(p(pg), p(g), C.c_size_t(N), st), "x")
    _lib.check(L.sobfu_hip_extract_tsdf(p(pn), p(n1), C.c_size_t(N), st), "x")
    _lib.check(L.sobfu_hip_tile3_apply_tsdf_only(p(n1), C.c_int(dim), C.c_int(dim), C.c_int(dim), p(f), p(psi3), C.c_int(dim), C.c_int(dim), C.c_int(dim), st), "apply")
    return psi3, g, n1, f, nu


slots = torch.zeros(256, dtype=torch.int32, device="cuda")


def pass_a(s, thin):
    psi3, g, n1, f, nu = s
    _lib.check(L.sobfu_hip_tile3_potential_gradient(p(f), p(g), p(psi3), p(nu), C.c_float(P["w_reg"]), C.c_int(dim), C.c_int(dim), C.c_int(dim), box,
                                                    C.c_int(thin), None, C.c_float(-1.0), C.c_int(1), st), "A")


def pass_b(s, thin):
    psi3, g, n1, f, nu = s
    _lib.check(L.sobfu_hip_tile3_smooth_update_apply(p(nu), p(psi3), p(n1), p(f), None, p(slots), taps, C.c_float(P["alpha"]), C.c_int(dim), C.c_int(dim),
                                                     C.c_int(dim), C.c_int(dim), C.c_int(dim), C.c_int(dim), box, box, C.c_int(thin), None, C.c_float(-1.0),
                                                     C.c_int(1), st), "B")


def timed(fn, n):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / n)
    return 1e6 * sorted(ts)[2]


res = {}
for thin in (0, 1):
    s = state()
    for _ in range(3):
        pass_a(s, thin)
        pass_b(s, thin)
    torch.cuda.synchronize()
    res[thin] = (s[0].clone(), s[3].clone())
    s = state()
    ta = timed(lambda: pass_a(s, thin), iters)
    tb = timed(lambda: pass_b(s, thin), iters)
    tab = timed(lambda: (pass_a(s, thin), pass_b(s, thin)), iters)
    print(f"dim {dim} {'DIRECT (lane per cell)' if thin else 'marching'}: pass A {ta:.1f} us, pass B {tb:.1f} us, A + B {tab:.1f} us per iteration", flush=True)
same = torch.equal(res[0][0].view(torch.int32), res[1][0].view(torch.int32)) and torch.equal(res[0][1].view(torch.int32), res[1][1].view(torch.int32))
print("bit-identical after 3 iterations:", same)
