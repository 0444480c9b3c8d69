"""Compute-side cost of one rank's share of the 256^3 solve in the NATIVE tiled loop (sobfu_hip_tiled_iterate), timed on one
GPU with communicator-less handles: every kernel launch of a rank with the most neighbours, no peers.

  direct   the direct transport's iteration: pass A (push boxes storing the messages -- here into the send buffer instead of a
           peer -- + tickets + the owned block) and pass B (owned block + thin shells): TWO launches, nothing else
  packed   the RCCL / callback transports' launches: the same pass A, the scatter kernel, pass B (the transfer itself excluded)
  slab schedules (1 x 1 x N only, packed): serial / overlapped z-slab schedules of round 1

    python tools/tile_time_native.py                 # 1x1x1, z-slabs 1x1x{2,4,8}, 1x2x2, 2x2x2, 1x2x4
    TILE_GRIDS=2x2x2,1x1x8 python tools/tile_time_native.py
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from sobfu_amd import ops, tiled
dim = int(os.environ.get("TILE_DIM", "256"))
P = bench.boxing_params(dim); dims = P["dims"]
c0, c1, r = bench.sphere_pair(P)
pg_full, pn_full = ops.new_volume(dims), ops.new_volume(dims)
ops.init_sphere(pg_full, P["vs"], P["trunc"], P["eta"], c0, r); ops.init_sphere(pn_full, P["vs"], P["trunc"], P["eta"], c1, r)
grids = os.environ.get("TILE_GRIDS", "1x1x1,1x1x2,1x1x4,1x1x8,1x2x2,2x2x2,1x2x4")
iters = int(os.environ.get("TILE_ITERS", "300"))
ref = None
for g in grids.split(","):
    grid = tuple(int(v) for v in g.split("x"))
    world = grid[0] * grid[1] * grid[2]
    lays = [tiled.TileLayout(dims, grid, q) for q in range(world)]
    rank = max(range(world), key=lambda q: (lays[q].L[0] * lays[q].L[1] * lays[q].L[2], q))  # a tile with the most halos
    for thr in [float(v) for v in os.environ.get("TILE_THR", "-1,1e-10").split(",")]:
        for mode in os.environ.get("TILE_MODES", "direct,packed").split(","):
            os.environ["SOBFU_TILED_DRY_PACKED"] = "1" if mode == "packed" else "0"
            for sched in ((3,) if (mode == "direct" or not lays[rank].slab) else (0, 3)):
                sv = tiled.NativeTiledSolver(dims, alpha=P["alpha"], w_reg=P["w_reg"], max_update_norm=thr, dry=(world, rank), grid=grid)
                sv.set_schedule(sched)
                L = sv.layout
                pg = L.take(pg_full).clone().contiguous(); pnp = sv.new_local(2); psi = sv.identity_psi()
                sv.iterate(pg, pn_full, pnp, psi, 50)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                sv.iterate(pg, pn_full, pnp, psi, iters)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / iters
                if ref is None: ref = dt
                own = tuple(L.g1[a] - L.g0[a] for a in range(3))
                what = mode if not (mode == "packed" and L.slab) else f"packed, slab schedule {'serial' if sched == 3 else 'heuristic (overlapped)'}"
                print(f"grid {g} rank {rank} thr={thr:g} {what}: owns {own}, local {L.L}: "
                      f"{1e6 * dt:.1f} us/iteration compute side -> bound {ref / dt:.2f}x of the first line ({1e6 * ref:.1f} us)", flush=True)
                sv.close()
