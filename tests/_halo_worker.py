"""Worker for tests/test_tiled_cpu.py::test_bounded_reach_windows_over_gloo: every rank fetches its WINDOW (owned cells widened by w) of a
field from the ranks that own the cells (sobfu_amd.tiled.DistHalo, point-to-point over gloo) and checks every window cell against the
closed-form pattern the owners wrote; halo cells of the local arrays hold garbage (they are not maintained between frames) and must
never travel.  Usage: python _halo_worker.py <rank> <world> <port> <PxxPyxPz> <X,Y,Z> <w>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sobfu_amd import tiled  # noqa: E402


def pattern(box, ch):
    """value of channel c at global cell (x, y, z): distinct for every (cell, channel)"""
    x0, x1, y0, y1, z0, z1 = box
    z, y, x = np.meshgrid(np.arange(z0, z1), np.arange(y0, y1), np.arange(x0, x1), indexing="ij")
    return np.stack([(x + 1000 * y + 1000000 * z + 0.25 * c).astype(np.float32) for c in range(ch)], -1)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    grid = tuple(int(v) for v in sys.argv[4].split("x"))
    dims = tuple(int(v) for v in sys.argv[5].split(","))
    w = int(sys.argv[6])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = tiled.TileLayout(dims, grid, rank)
    halo = tiled.DistHalo(L)
    assert halo.allreduce_max(float(rank) + 0.5) == world - 0.5
    for ch, nch in ((4, 3), (2, 2)):
        local = torch.full(L.local_shape(ch), -777.0)  # halo cells: garbage
        L.owned(local).copy_(torch.from_numpy(pattern(L.owned_box_global(), ch)))
        before = halo.bytes_received
        win, wb = halo.window(local, w, nch)
        assert wb == L.window_box(w)
        want = pattern(wb, ch)
        got = win.numpy()
        assert np.array_equal(got[..., :nch], want[..., :nch]), (rank, ch)
        if nch < ch:  # channels that do not travel: the owner's own cells keep theirs, received cells hold 0
            own = tiled._cut(torch.from_numpy(np.ascontiguousarray(got)), L.owned_box_global(), (wb[0], wb[2], wb[4])).numpy()
            assert np.array_equal(own[..., nch:], pattern(L.owned_box_global(), ch)[..., nch:])
        cells = (wb[1] - wb[0]) * (wb[3] - wb[2]) * (wb[5] - wb[4]) - (L.g1[0] - L.g0[0]) * (L.g1[1] - L.g0[1]) * (L.g1[2] - L.g0[2])
        assert halo.bytes_received - before == 4 * nch * cells, (halo.bytes_received - before, cells)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
