"""Compute-side cost of one rank's share of the 256^3 solve when cut into N z-slabs (no communication): what the
overlapped schedule of sobfu_amd.tiled issues per iteration (6 launches), timed on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from sobfu_amd import ops, tiled
P = bench.boxing_params(256); dims = P["dims"]; X, Y, Z = dims
c0, c1, r = bench.sphere_pair(P)
pg_full, pn_full = ops.new_volume(dims), ops.new_volume(dims)
ops.init_sphere(pg_full, P["vs"], P["trunc"], P["eta"], c0, r); ops.init_sphere(pn_full, P["vs"], P["trunc"], P["eta"], c1, r)
be = tiled.HipBackend()
S = ops.sobolev_filter(7, 0.1)
for world in (1, 2, 4, 8):
    rank = world // 2
    L = tiled.SlabLayout(dims, world, rank)
    pg = L.take(pg_full).clone(); pnp = torch.zeros(L.local_shape(2), device="cuda"); psi = torch.zeros(L.local_shape(4), device="cuda")
    be.init_identity(psi, L)
    st = be.begin(L, pg, pn_full, pnp, psi)
    slots = torch.zeros((402, 256), dtype=torch.int32, device="cuda")
    lo, hi, H = L.own_lo, L.own_hi, tiled.HALO
    a_lo = min(lo + H, hi) if L.lo else lo; a_hi = max(hi - H, a_lo) if L.hi else hi
    b_lo = min(lo + 3, hi) if L.lo else lo; b_hi = max(hi - 3, b_lo) if L.hi else hi
    b_first = lo - 1 if L.lo else lo; b_last = hi + 1 if L.hi else hi
    def it(k):
        row = slots[k]
        be.pass_a(st, lo, a_lo, P["w_reg"], None, -1.0); be.pass_a(st, a_hi, hi, P["w_reg"], None, -1.0)
        be.pass_a(st, a_lo, a_hi, P["w_reg"], None, -1.0)
        be.pass_b(st, b_lo, b_hi, row, S, P["alpha"], None, -1.0)
        be.pass_b(st, b_first, b_lo, row, S, P["alpha"], None, -1.0); be.pass_b(st, b_hi, b_last, row, S, P["alpha"], None, -1.0)
    for k in range(1, 101): it(k)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(101, 401): it(k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    print(f"N={world}: slab {L.z1-L.z0}+{L.lo}+{L.hi} planes: {1e6*dt:.1f} us/iteration (compute only) -> ideal {1/dt:.0f} it/s, "
          f"exchange {2*(1 if world>1 else 0)*tiled.HALO*X*Y*12/1e6:.1f} MB/iter out")
