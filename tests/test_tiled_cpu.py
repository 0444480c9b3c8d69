"""Multi-GPU path on CPU: world-size-2/3/4 gloo runs of sobfu_amd.tiled (tile decomposition -- z-slabs, x / y splits, 2-D grids
with edge strips -- + halo exchange + max-norm all-reduce + device-gate semantics) must reproduce the single-process oracle
solve bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(world, thr, tmp_path, grid=""):
    out = str(tmp_path / f"tiled_{world}_{thr}_{grid}.npz")
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_tiled_worker.py"), str(r), str(world), port, out, str(thr)] + ([grid] if grid else []),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return np.load(out)


def _reference(oracle, thr):
    sys.path.insert(0, HERE)
    import _tiled_worker as W

    pg, pn = W.inputs()
    psi = oracle.new_field(W.DIMS)
    oracle.init_identity(psi)
    r1 = oracle.estimate_psi(pg, pn, psi, max_iter=W.ITERS, alpha=0.05, w_reg=0.4, max_update_norm=thr, inverse_iters=0,
                             compute_jacobian=False)
    r2 = oracle.estimate_psi(pg, pn, psi, max_iter=3, alpha=0.05, w_reg=0.4, max_update_norm=thr, inverse_iters=0,
                             compute_jacobian=False)
    return psi, r1, r2


@pytest.mark.parametrize("world", [2, 3])
def test_slabs_match_single_process(oracle, tmp_path, world):
    got = _run(world, -1.0, tmp_path)
    psi, r1, r2 = _reference(oracle, -1.0)
    assert int(got["done"]) == r1["iters"] == 6 and int(got["done2"]) == 3
    assert np.array_equal(got["psi"].view(np.uint32), psi.view(np.uint32))
    assert np.array_equal(got["pnp"].view(np.uint32), r2["phi_n_psi"].view(np.uint32))
    assert np.array_equal(got["norms"].view(np.uint32), np.ascontiguousarray(r1["trace"][:, 2]).view(np.uint32))
    assert np.array_equal(got["norms2"].view(np.uint32), np.ascontiguousarray(r2["trace"][:, 2]).view(np.uint32))


@pytest.mark.parametrize("grid", ["2x1x1", "1x2x1", "2x2x1", "1x2x2"])
def test_tiles_match_single_process(oracle, tmp_path, grid):
    """x / y splits (non-contiguous faces, the thin x shell) and 2-D grids (edge strips between diagonal neighbours)"""
    world = int(np.prod([int(v) for v in grid.split("x")]))
    got = _run(world, -1.0, tmp_path, grid)
    psi, r1, r2 = _reference(oracle, -1.0)
    assert int(got["done"]) == r1["iters"] == 6 and int(got["done2"]) == 3
    assert np.array_equal(got["psi"].view(np.uint32), psi.view(np.uint32))
    assert np.array_equal(got["pnp"].view(np.uint32), r2["phi_n_psi"].view(np.uint32))
    assert np.array_equal(got["norms"].view(np.uint32), np.ascontiguousarray(r1["trace"][:, 2]).view(np.uint32))
    assert np.array_equal(got["norms2"].view(np.uint32), np.ascontiguousarray(r2["trace"][:, 2]).view(np.uint32))


def test_tiles_convergence_gate(oracle, tmp_path):
    psi0, r_free, _ = _reference(oracle, -1.0)
    tr = r_free["trace"][:, 2]
    thr = float(np.float32((tr[2] + tr[3]) / 2)) if tr[3] < tr[2] else float(tr.min())
    got = _run(2, thr, tmp_path, "2x1x1")
    psi, r1, r2 = _reference(oracle, thr)
    assert r1["iters"] < 6 and int(got["done"]) == r1["iters"] and int(got["done2"]) == r2["iters"]
    assert np.array_equal(got["psi"].view(np.uint32), psi.view(np.uint32))


def test_slabs_convergence_gate(oracle, tmp_path):
    """positive threshold: the all-reduced max-norm gate must stop every rank at the reference's iteration"""
    psi0, r_free, _ = _reference(oracle, -1.0)
    tr = r_free["trace"][:, 2]
    thr = float(np.float32((tr[2] + tr[3]) / 2)) if tr[3] < tr[2] else float(tr.min())
    got = _run(2, thr, tmp_path)
    psi, r1, r2 = _reference(oracle, thr)
    assert r1["iters"] < 6
    assert int(got["done"]) == r1["iters"] and int(got["done2"]) == r2["iters"]
    assert np.array_equal(got["psi"].view(np.uint32), psi.view(np.uint32))
    assert np.array_equal(got["pnp"].view(np.uint32), r2["phi_n_psi"].view(np.uint32))


def test_tile_layout_properties():
    """every cell is owned exactly once; messages pair up (my send box has the shape of the peer's recv box for me)"""
    from sobfu_amd.tiled import TileLayout, default_grid, parse_grid

    assert default_grid(8) == (2, 2, 2) and default_grid(4) == (1, 2, 2) and default_grid(2) == (1, 1, 2) and default_grid(1) == (1, 1, 1)
    assert default_grid(6) == (1, 2, 3) and parse_grid("2x1x4", 8) == (2, 1, 4)
    with pytest.raises(ValueError):
        parse_grid("2x2x2", 4)
    dims = (21, 17, 26)
    for grid in ((2, 2, 2), (1, 2, 2), (3, 1, 2), (2, 1, 1), (1, 1, 5)):
        world = grid[0] * grid[1] * grid[2]
        lays = [TileLayout(dims, grid, r) for r in range(world)]
        count = np.zeros(dims[::-1], np.int32)
        for L in lays:
            L.owned_global(count)[...] += 1
            assert all(L.L[a] == L.g1[a] - L.g0[a] + L.lo3[a] + L.hi3[a] and L.base[a] == L.g0[a] - L.lo3[a] for a in range(3))
            assert L.take(count).shape == L.local_shape()
        assert (count == 1).all()
        for L in lays:
            peers = [m[0] for m in L.messages()]
            assert len(set(peers)) == len(peers)  # one message per neighbour
            for peer, sb, rb in L.messages():
                back = [m for m in lays[peer].messages() if m[0] == L.rank]
                assert len(back) == 1
                shape = lambda b: (b[1] - b[0], b[3] - b[2], b[5] - b[4])  # noqa: E731
                assert shape(sb) == shape(back[0][2]) and shape(rb) == shape(back[0][1])
    assert len(TileLayout((64, 64, 64), (2, 2, 2), 0).messages()) == 6  # 3 faces + 3 edges, no corner
    assert len(TileLayout((64, 64, 64), (3, 3, 3), 13).messages()) == 18
    with pytest.raises(ValueError):
        TileLayout((6, 64, 64), (2, 1, 1), 0)


def test_tile_messages_carry_the_right_cells():
    """random extents and grids: what a rank sends are cells it OWNS, they land on the SAME global cells in the peer's halo, and
    together the messages a rank receives cover exactly the face and edge halo cells pass B's stencils reach (4 cells along one
    axis, or along two -- never all three)"""
    from sobfu_amd.tiled import HALO, TileLayout

    rng = np.random.default_rng(7)
    cases = 0
    while cases < 40:
        grid = tuple(int(g) for g in rng.integers(1, 4, 3))
        dims = tuple(int(d) for d in rng.integers(9, 40, 3))
        try:
            lays = [TileLayout(dims, grid, r) for r in range(grid[0] * grid[1] * grid[2])]
        except ValueError:
            continue  # a tile thinner than the halo: refused (covered above)
        cases += 1
        for L in lays:
            got = np.zeros(L.local_shape()[:3], np.int32)  # (Lz, Ly, Lx)
            for peer, sb, rb in L.messages():
                P = lays[peer]
                back = [m for m in P.messages() if m[0] == L.rank][0]
                # my send box, in global cells, is the peer's receive box for me, in global cells
                mine = tuple(sb[2 * a + k] + L.base[a] for a in range(3) for k in range(2))
                theirs = tuple(back[2][2 * a + k] + P.base[a] for a in range(3) for k in range(2))
                assert mine == theirs, (dims, grid, L.rank, peer)
                # ... cells I own ...
                assert all(L.g0[a] <= mine[2 * a] and mine[2 * a + 1] <= L.g1[a] for a in range(3))
                # ... and what I receive lies in my halo, outside the cells I own
                g = tuple(rb[2 * a + k] + L.base[a] for a in range(3) for k in range(2))
                assert any(g[2 * a + 1] <= L.g0[a] or g[2 * a] >= L.g1[a] for a in range(3))
                got[rb[4]:rb[5], rb[2]:rb[3], rb[0]:rb[1]] += 1
            # cells of the local array outside the owned box, by the number of axes along which they are outside
            out = [np.zeros(L.L[a], bool) for a in range(3)]
            for a in range(3):
                o0 = L.g0[a] - L.base[a]
                out[a][:o0] = True
                out[a][o0 + (L.g1[a] - L.g0[a]):] = True
            n_out = out[2][:, None, None].astype(np.int32) + out[1][None, :, None] + out[0][None, None, :]
            assert ((got == 1) == ((n_out == 1) | (n_out == 2))).all(), (dims, grid, L.rank)
            assert (got <= 1).all()
            assert all(L.lo3[a] in (0, HALO) and L.hi3[a] in (0, HALO) for a in range(3))


def test_layout_properties():
    from sobfu_amd.tiled import SlabLayout

    for world in (1, 2, 3, 4, 8):
        own = []
        for r in range(world):
            L = SlabLayout((16, 16, 64), world, r)
            own += list(range(L.z0, L.z1))
            assert L.Lz == (L.z1 - L.z0) + L.lo + L.hi and L.zbase == L.z0 - L.lo
            assert (L.lo == 0) == (r == 0) and (L.hi == 0) == (r == world - 1)
        assert own == list(range(64))
    with pytest.raises(ValueError):
        SlabLayout((16, 16, 8), 4, 0)


@pytest.mark.parametrize("grid,dims,w", [("1x1x2", "20,12,24", 3), ("2x2x1", "20,12,24", 2), ("1x2x2", "20,12,24", 5), ("2x2x2", "16,18,20", 3),
                                         ("1x1x4", "8,8,24", 6),  # w == the tiles' extent: a window reaches the next ring's owner ... not yet: exactly the neighbour
                                         ("1x1x4", "8,8,24", 9)])  # ... and beyond it: cells two tiles away travel too (nothing assumes 26 neighbours)
def test_bounded_reach_windows_over_gloo(grid, dims, w):
    """the communication of the per-frame tail on tiles (sobfu_amd.tiled.DistHalo / window_plan): every rank ends up with its owned cells
    widened by w, fetched point-to-point from their owners; the bytes received are exactly the cells it does not own"""
    world = int(np.prod([int(v) for v in grid.split("x")]))
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_halo_worker.py"), str(r), str(world), port, grid, dims, str(w)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)


def test_window_plan_is_symmetric_and_covers_the_window():
    from sobfu_amd import tiled

    for dims, grid, w in (((256, 256, 256), (2, 2, 2), 3), ((70, 33, 40), (2, 1, 3), 4), ((64, 64, 64), (1, 2, 4), 17)):
        world = grid[0] * grid[1] * grid[2]
        plans = [tiled.window_plan(tiled.TileLayout(dims, grid, r), w) for r in range(world)]
        for r, (wb, recvs, sends) in enumerate(plans):
            cells = sum((b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4]) for _, b in recvs)
            L = tiled.TileLayout(dims, grid, r)
            own = (L.g1[0] - L.g0[0]) * (L.g1[1] - L.g0[1]) * (L.g1[2] - L.g0[2])
            assert cells + own == (wb[1] - wb[0]) * (wb[3] - wb[2]) * (wb[5] - wb[4])  # disjoint owners tile the window exactly
            for q, b in recvs:  # what r expects from q is what q sends to r
                assert (r, b) in plans[q][2]
        if dims == (256, 256, 256):  # BASELINE config 4: 7 neighbours, 1.8 MB of psi + 1.2 MB of phi_global instead of 268 + 134 MB
            assert len(plans[0][1]) == 7 and sum((b[1] - b[0]) * (b[3] - b[2]) * (b[5] - b[4]) for _, b in plans[0][1]) == 131 ** 3 - 128 ** 3
