"""How fast does the reference's OWN decomposition run on this GPU?  oracle/_ref/reference_hip_ieee (the reference's kernels compiled for
gfx950 by oracle/ref_hipbuild/build.py; arrays bit-identical to the host emulation and to this repo's kernels) runs Solver::estimate_psi on
bench.py's workload (256^3, params_boxing.ini solver values, two initSphere volumes) as the reference drives it: ten kernels, a host
synchronisation and a 128 KB read-back per iteration (solver.cu:114-193), then 48 inverse launches and a warp.  The iteration rate is the
difference of a 100-iteration and a 50-iteration solve; this repo's solver runs the same workload beside it.

    python tests/reference_time.py [dim]      # on the GPU box; prints one JSON line

Shim evidence / a measured baseline, not the product: the binary exists only where the build container made it."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def reference_rate(dim=256, repeat=3):
    """-> dict(iterations_per_s, s_per_solve_50, s_not_iterations): bench.reference_build_on_this_gpu on bench.py's workload at dim^3"""
    import bench

    r = bench.reference_build_on_this_gpu(bench.boxing_params(dim), repeat=repeat)
    assert r and r.get("value"), r
    return {"iterations_per_s": r["value"], "s_per_solve_50": r["s_per_solve_50"], "s_not_iterations": r["s_not_iterations"]}


def main():
    import numpy as np
    import torch

    import bench
    import ref_hip_runner as R
    from sobfu_amd import ops

    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    if not R.available():
        raise SystemExit("oracle/_ref/reference_hip_ieee is missing: build it in the build container (python oracle/ref_hipbuild/build.py)")
    ref = reference_rate(dim)
    P = bench.boxing_params(dim)
    c0, c1, r = bench.sphere_pair(P)
    pg, pn = ops.new_volume(P["dims"]), ops.new_volume(P["dims"])
    ops.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r)
    ops.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
    sv = ops.Solver(P["dims"], max_iter=100, alpha=P["alpha"], w_reg=P["w_reg"], max_update_norm=P["max_update_norm"])
    t = {}
    for n in (50, 100):
        best = 1e9
        for _ in range(3):
            psi, pnp = ops.new_field(P["dims"]), ops.new_volume(P["dims"])
            ops.init_identity(psi)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sv.iterate(pg, pn, pnp, psi, n)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        t[n] = best
    sv.close()
    ours = 50.0 / (t[100] - t[50])
    print(json.dumps({"workload": "%d^3, params_boxing.ini solver values, two initSphere volumes 1.3 voxels apart (bench.py's)" % dim,
                      "reference_build": dict(ref, what="the reference's .cu / .cpp files compiled for gfx950 by hipcc through a CUDA -> HIP name map "
                                                      "(oracle/ref_hipbuild), IEEE flavour, driven by its own Solver::estimate_psi"),
                      "this_repo": {"iterations_per_s": ours, "s_per_solve_50": t[50]}, "speedup_iterations": ours / ref["iterations_per_s"],
                      "speedup_whole_solve_50": ref["s_per_solve_50"] / t[50]}))


if __name__ == "__main__":
    main()
