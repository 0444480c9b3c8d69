"""Compute-side cost of one rank's share of the 256^3 solve in the NATIVE tiled loop (sobfu_hip_tiled_iterate) for N
z-slabs, timed on one GPU with communicator-less handles: every kernel launch and stream/event dependency of a middle
rank's schedule, no peers (the numbers exclude the exchange and the all-reduce themselves)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sobfu_amd import ops, tiled
P = bench.boxing_params(256); dims = P["dims"]; X, Y, Z = dims
c0, c1, r = bench.sphere_pair(P)
pg_full, pn_full = ops.new_volume(dims), ops.new_volume(dims)
ops.init_sphere(pg_full, P["vs"], P["trunc"], P["eta"], c0, r); ops.init_sphere(pn_full, P["vs"], P["trunc"], P["eta"], c1, r)
for world in [int(w) for w in os.environ.get("SLAB_WORLDS", "1,2,4,8").split(",")]:
    for thr in (-1.0, 1e-10):
        sv = tiled.NativeTiledSolver(dims, alpha=P["alpha"], w_reg=P["w_reg"], max_update_norm=thr, dry=(world, world // 2))
        L = sv.layout
        pg = L.take(pg_full).clone(); pnp = sv.new_local(2); psi = sv.identity_psi()
        sv.iterate(pg, pn_full, pnp, psi, 50)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sv.iterate(pg, pn_full, pnp, psi, 300)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
        print(f"N={world} thr={thr:g}: slab {L.z1-L.z0}+{L.lo}+{L.hi} planes: {1e6*dt:.1f} us/iteration -> {1/dt:.0f} it/s "
              f"(speed-up bound {bench_ref/dt if (bench_ref:=globals().get('bench_ref')) else 1:.2f}x)", flush=True)
        if world == 1 and thr < 0: bench_ref = dt
        sv.close()
