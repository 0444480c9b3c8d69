"""Go / no-go for VERDICT round 4, item 1: the single-launch iteration in which nabla_U never leaves the CU
(tools/calib/fused_iteration.hip) beside today's two-pass loop, SAME grid, SAME process, bits compared.

    python tools/fused_go_nogo.py            # FUSED_DIM=128 (a tile of the 2x2x2 split), FUSED_ITERS=200
    FUSED_NCH=4,8,12,16 FUSED_DIM=128 python tools/fused_go_nogo.py

Prints one JSON line per configuration: us per iteration of the two-pass solver loop (sobfu_hip_solver_step) and of the fused
kernel for every z-chunk count, and whether psi / phi_n o psi after FUSED_CHECK iterations are bit-identical.
"""
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from sobfu_amd import _lib, ops

SO = os.path.join(ROOT, "build", "libfused_iteration.so")


def build():
    src = os.path.join(ROOT, "tools", "calib", "fused_iteration.hip")
    if os.path.exists(SO) and os.path.getmtime(SO) > os.path.getmtime(src):
        return
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
                           "-shared", f"-I{ROOT}/sobfu_amd/csrc", f"-I{ROOT}/include", src, "-o", SO])


def main():
    if "--build-only" in sys.argv:
        build()
        return
    build()
    F = C.CDLL(SO)
    L = _lib.lib()
    dim = int(os.environ.get("FUSED_DIM", "128"))
    iters = int(os.environ.get("FUSED_ITERS", "200"))
    check = int(os.environ.get("FUSED_CHECK", "6"))
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

    def two_pass(n, timed):
        psi, pnp = ops.new_field(dims), ops.new_volume(dims)
        ops.init_identity(psi)
        sv = ops.Solver(dims, max_iter=100000, alpha=P["alpha"], w_reg=P["w_reg"], s=P["s"], lam=P["lam"], max_update_norm=P["max_update_norm"])
        sv.begin(pg, pn, pnp, psi, 100000)
        dt = None
        if timed:
            sv.step(50)
            torch.cuda.synchronize()
            best = []
            for _ in range(5):
                t0 = time.perf_counter()
                sv.step(n)
                torch.cuda.synchronize()
                best.append((time.perf_counter() - t0) / n)
            dt = sorted(best)[len(best) // 2]
        else:
            sv.step(n)
        sv.end()
        sv.close()
        return psi, pnp, dt

    def compact_state():
        psi4 = ops.new_field(dims)
        ops.init_identity(psi4)
        psi3 = [torch.zeros(N * 3, dtype=torch.float32, device="cuda") for _ in range(2)]
        f = [torch.zeros(N, dtype=torch.float32, device="cuda") for _ in range(2)]
        g, n1 = torch.zeros(N, dtype=torch.float32, device="cuda"), torch.zeros(N, dtype=torch.float32, device="cuda")
        _lib.check(L.sobfu_hip_pack_vec3(p(psi4), p(psi3[0]), C.c_size_t(N), st), "pack")
        _lib.check(L.sobfu_hip_extract_tsdf(p(pg), p(g), C.c_size_t(N), st), "extract")
        _lib.check(L.sobfu_hip_extract_tsdf(p(pn), p(n1), C.c_size_t(N), st), "extract")
        _lib.check(L.sobfu_hip_tile3_apply_tsdf_only(p(n1), C.c_int(dim), C.c_int(dim), C.c_int(dim), p(f[0]), p(psi3[0]), C.c_int(dim), C.c_int(dim), C.c_int(dim), st), "apply")
        return psi3, f, g, n1

    slots = torch.zeros(256 * 4, dtype=torch.int32, device="cuda")

    def fused(psi3, f, g, n1, n, nch, nt=0):
        P2 = C.c_void_p * 2
        rc = F.calib_fused_iterate(P2(psi3[0].data_ptr(), psi3[1].data_ptr()), P2(f[0].data_ptr(), f[1].data_ptr()), p(g), p(n1), p(slots), taps,
                                   C.c_float(P["alpha"]), C.c_float(P["w_reg"]), C.c_int(dim), C.c_int(dim), C.c_int(dim), C.c_int(nch), C.c_int(nt),
                                   C.c_int(n), st)
        assert rc == 0, rc
        return n & 1  # index of the half that holds the result

    # bits: `check` iterations of both
    psi_ref, pnp_ref, _ = two_pass(check, False)
    psi_ref3 = psi_ref.reshape(-1, 4)[:, :3].contiguous().reshape(-1)
    f_ref = pnp_ref.reshape(-1, 2)[:, 0].contiguous()
    _, _, dt2 = two_pass(iters, True)
    out = {"dim": dim, "two_pass_us": 1e6 * dt2, "fused": []}
    nchs = [int(v) for v in os.environ.get("FUSED_NCH", "4,6,8,10,12,16").split(",")]
    for nch in nchs:
        psi3, f, g, n1 = compact_state()
        h = fused(psi3, f, g, n1, check, nch)
        torch.cuda.synchronize()
        same = bool(torch.equal(psi3[h].view(torch.int32), psi_ref3.view(torch.int32)) and torch.equal(f[h].view(torch.int32), f_ref.view(torch.int32)))
        fused(psi3, f, g, n1, 50, nch)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            fused(psi3, f, g, n1, iters, nch)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / iters)
        tiles = ((dim + 63) // 64) * ((dim + 7) // 8)
        out["fused"].append({"z_chunks": nch, "workgroups": tiles * nch, "planes_per_march": dim / nch, "us": 1e6 * sorted(ts)[2], "bit_identical": same})
        print(json.dumps(out["fused"][-1]), flush=True)
    best = min(out["fused"], key=lambda e: e["us"])
    out["best_fused_us"] = best["us"]
    out["verdict"] = "go (<= 32 us)" if (dim == 128 and best["us"] <= 32.0) else ("no-go" if dim == 128 else "n/a")
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
