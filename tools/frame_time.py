"""Times one full Solver::estimate_psi at 256^3 (BASELINE config 3: 50 iterations + 48-sweep inverse + canonical warp)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sobfu_amd import ops
P = bench.boxing_params(256); dims = P["dims"]
c0, c1, r = bench.sphere_pair(P)
pg, pn, pnp, pgi = (ops.new_volume(dims) for _ in range(4))
ops.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r); ops.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
psi, psi_inv = ops.new_field(dims), ops.new_field(dims); ops.init_identity(psi)
sv = ops.Solver(dims, max_iter=50, alpha=P["alpha"], w_reg=P["w_reg"], max_update_norm=P["max_update_norm"])
for k in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rep, hist = sv.estimate_psi(pg, pgi, pn, pnp, psi, psi_inv)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"estimate_psi #{k}: {1e3*dt:.2f} ms for {rep.iterations} iterations (+inverse +warp)")
for name, fn in (("init_identity", lambda: ops.init_identity(psi_inv)), ("estimate_inverse x48", lambda: ops.estimate_inverse(psi, psi_inv, 48)),
                 ("apply", lambda: ops.apply(pg, pgi, psi_inv)), ("integrate_fuse", lambda: ops.integrate_fuse(pg, pnp, 128.0))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); print(f"  {name}: {1e3*(time.perf_counter()-t0)/5:.3f} ms")
