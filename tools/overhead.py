import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sobfu_amd import ops
P = bench.boxing_params(256); dims = P["dims"]
c0, c1, r = bench.sphere_pair(P)
pg, pn, pnp = ops.new_volume(dims), ops.new_volume(dims), ops.new_volume(dims)
ops.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r); ops.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
psi = ops.new_field(dims); ops.init_identity(psi)
for thr in (1e-10, -1.0):
    sv = ops.Solver(dims, max_iter=400, alpha=P["alpha"], w_reg=P["w_reg"], max_update_norm=thr)
    sv.iterate(pg, pn, pnp, psi, 50)
    for n in (1, 10, 30, 100, 300, 30, 10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sv.iterate(pg, pn, pnp, psi, n)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"thr={thr} n={n}: total {1e3*dt:.2f} ms  = {1e6*dt/n:.1f} us/iter")
    sv.close()
