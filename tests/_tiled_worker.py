"""Worker for tests/test_tiled_cpu.py: runs tests/tiled_reference.TiledSolver over gloo with an ORACLE-backed per-tile kernel
backend (test infrastructure: checks the decomposition / halo-exchange / reduction logic on CPU, bit for bit against the
single-process oracle solve).  Usage: python _tiled_worker.py <rank> <world> <port> <out.npz> <thr> [PxxPyxPz]"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import oracle as O  # noqa: E402
from sobfu_amd import tiled  # noqa: E402
import tiled_reference  # noqa: E402
from sobfu_amd.synthetic import hash_field  # noqa: E402

DIMS = (20, 12, 24)
ITERS = 6


class _State:
    pass


class OracleBackend:
    """begin / pass_a / pass_b / end protocol of sobfu_amd.tiled backends, on numpy views of CPU tensors (API format)."""

    device = "cpu"

    def init_identity(self, psi, layout):
        a = psi.numpy()
        O.init_identity(a)
        for k in range(3):
            a[..., k] += np.float32(layout.base[k])

    def begin(self, layout, pg, pn_full, pnp, psi):
        st = _State()
        st.layout, st.pg, st.pn, st.pnp, st.psi = layout, pg.numpy(), pn_full.numpy(), pnp.numpy(), psi.numpy()
        st.nabla_U = torch.zeros(layout.local_shape(4), dtype=torch.float32)
        O.apply_tile(st.pn, st.pnp, st.psi)
        return st

    @staticmethod
    def _gate(prev, thr):
        if prev is None:
            return False
        return tiled._sqrt_rd(int(prev.numpy().view(np.uint32).max())) <= thr

    @staticmethod
    def _sl(box):
        return (slice(box[4], box[5]), slice(box[2], box[3]), slice(box[0], box[1]))

    def pass_a(self, st, box, w_reg, prev, thr, thin=False):
        """whole-array oracle kernels, only the cells of `box` are committed (a launch of the HIP kernel produces exactly those)"""
        if min(box[1] - box[0], box[3] - box[2], box[5] - box[4]) <= 0 or self._gate(prev, thr):
            return
        dims = st.layout.L
        g, Lap, out = O.new_field(dims), O.new_field(dims), O.new_field(dims)
        O.tsdf_gradient(st.pnp, g)
        O.laplacian(st.psi, Lap)
        O.potential_gradient(st.pnp, st.pg, g, Lap, out, w_reg)
        st.nabla_U.numpy()[self._sl(box)] = out[self._sl(box)]

    def pass_b(self, st, box, slots, taps, alpha, prev, thr, thin=False):
        if min(box[1] - box[0], box[3] - box[2], box[5] - box[4]) <= 0 or self._gate(prev, thr):
            return
        L = st.layout
        dims = L.L
        nU = st.nabla_U.numpy()
        nUS, upd = O.new_field(dims), O.new_field(dims)
        O.convolution_rows(nUS, nU, taps)
        O.convolution_columns(nUS, nU, taps)
        O.convolution_depth(nUS, nU, taps)
        psi_new = st.psi.copy()
        O.update_psi(psi_new, nUS, upd, alpha)
        sl = self._sl(box)
        st.psi[sl] = psi_new[sl]
        warped = O.new_volume(dims)
        O.apply_tile(st.pn, warped, psi_new)
        st.pnp[sl] = warped[sl]
        ob = L.own_box()
        lo = [max(box[2 * k], ob[2 * k]) for k in range(3)]
        hi = [min(box[2 * k + 1], ob[2 * k + 1]) for k in range(3)]
        if all(h > l for l, h in zip(lo, hi)):
            u = upd[lo[2]:hi[2], lo[1]:hi[1], lo[0]:hi[0]]
            sq = (u[..., 0] * u[..., 0] + u[..., 1] * u[..., 1]) + u[..., 2] * u[..., 2]
            s = slots.numpy().view(np.uint32)
            s[0] = max(s[0], np.float32(sq.max()).view(np.uint32))

    def end(self, st):
        pass

    def sobolev_filter(self, s, lam):
        return O.sobolev_filter(s, lam)

    def synchronize(self):
        pass


def inputs():
    X, Y, Z = DIMS
    pg = hash_field((Z, Y, X, 2), 101)
    pn = hash_field((Z, Y, X, 2), 102)
    pg[..., 1] = 1
    pn[..., 1] = (hash_field((Z, Y, X), 103) > -0.5).astype(np.float32)
    return pg, pn


def main():
    rank, world, port, out, thr = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], float(sys.argv[5])
    grid = tiled.parse_grid(sys.argv[6], world) if len(sys.argv) > 6 else (1, 1, world)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    O.set_num_threads(1)
    torch.set_num_threads(1)
    pg, pn = inputs()
    sv = tiled_reference.TiledSolver(DIMS, alpha=0.05, w_reg=0.4, max_update_norm=thr, backend=OracleBackend(), grid=grid)
    L = sv.layout
    pg_l = torch.from_numpy(np.ascontiguousarray(L.take(pg)))
    pn_full = torch.from_numpy(pn)
    pnp = sv.new_local(2)
    psi = sv.identity_psi()
    done, norms = sv.iterate(pg_l, pn_full, pnp, psi, ITERS)
    # second solve, warm-started (psi persists across frames)
    done2, norms2 = sv.iterate(pg_l, pn_full, pnp, psi, 3)
    psi_full = sv.gather_owned(psi)
    pnp_full = sv.gather_owned(pnp)
    if rank == 0:
        np.savez(out, psi=psi_full.numpy(), pnp=pnp_full.numpy(), norms=norms, done=done, norms2=norms2, done2=done2)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
