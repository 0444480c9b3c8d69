"""The tile iteration restated as a plain torch.distributed loop -- TEST INFRASTRUCTURE (moved out of sobfu_amd/tiled.py in round 4).

`TiledSolver` issues one iteration as the product's native loop does (sobfu_amd/csrc/tiled_capi.hip: pass A on the owned cells, one
halo exchange of nabla_U, pass B on owned +- 1), from Python, with a pluggable per-tile kernel backend:
  * tests/_tiled_worker.py plugs in an ORACLE-backed backend and runs it over gloo with 2 - 4 ranks (tests/test_tiled_cpu.py): the
    decomposition logic -- which cells a launch produces, which faces and edge strips travel -- checked without a GPU;
  * `HipBackend` calls the per-tile HIP kernels through the C ABI (sobfu_hip_tile3_*): tests/test_gpu_parity.py checks the tile
    kernels against the full-volume kernels with it.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from sobfu_amd.tiled import HALO, SLOTS, TileLayout, _sqrt_rd, estimate_psi_tiled, gather_owned


def _cut(t, box):
    return t[box[4]:box[5], box[2]:box[3], box[0]:box[1]]


def halo_ops(layout: TileLayout, fields, group=None):
    """P2P op list of one exchange (built once per solve: the buffers stay valid while the solve lives).
    fields: list of (tensor (Lz, Ly, Lx, ...), width).  z-slabs: zero-copy views of the planes.  3-D tiles: staging buffers;
    returns (ops, pack, unpack) where pack() copies the send boxes out before the ops start and unpack() scatters the received
    boxes after they finished."""
    L = layout
    ops, packs, unpacks = [], [], []
    for t, w in fields:
        assert tuple(t.shape[:3]) == L.local_shape() and t.is_contiguous() and 0 < w <= L.halo
        for peer, sb, rb in L.messages(w):
            src, dst = _cut(t, sb), _cut(t, rb)
            if src.is_contiguous() and dst.is_contiguous():
                ops.append(dist.P2POp(dist.isend, src, peer, group))
                ops.append(dist.P2POp(dist.irecv, dst, peer, group))
            else:
                sbuf, rbuf = torch.empty_like(src, memory_format=torch.contiguous_format), torch.empty_like(dst, memory_format=torch.contiguous_format)
                packs.append((sbuf, src))
                unpacks.append((dst, rbuf))
                ops.append(dist.P2POp(dist.isend, sbuf, peer, group))
                ops.append(dist.P2POp(dist.irecv, rbuf, peer, group))

    def pack():
        for buf, view in packs:
            buf.copy_(view)

    def unpack():
        for view, buf in unpacks:
            view.copy_(buf)

    return ops, pack, unpack


def start_halo_ops(ops):
    """One grouped RCCL launch on RCCL's own stream, ordered after everything queued so far on the current stream."""
    return dist.batch_isend_irecv(ops) if ops else []


def finish_halo_ops(works):
    for w in works:  # for RCCL this only makes the current stream wait; the host does not block
        w.wait()


def run_halo_ops(ops):
    finish_halo_ops(start_halo_ops(ops))


def exchange_halos(layout: TileLayout, fields, group=None):
    """fields: list of (tensor (Lz, Ly, Lx, ...), width): neighbour exchange of the `width`-cell faces / edge strips."""
    ops, pack, unpack = halo_ops(layout, fields, group)
    pack()
    run_halo_ops(ops)
    unpack()


class _SlabState:
    """What a backend keeps for one solve: nabla_U (exchanged by the driver) + whatever format it iterates in."""

    def __init__(self, layout, nabla_U):
        self.layout, self.nabla_U = layout, nabla_U


class HipBackend:
    """Per-tile kernels through the C ABI (include/sobfu_hip.h `sobfu_hip_tile3_*`).

    Iterates in the compact format (12-byte psi / nabla_U, tsdf-only phi_global / phi_n / phi_n o psi -- fewer bytes both
    through HBM and over xGMI); `begin` converts the caller's API-format arrays, `end` rebuilds them."""

    device = "cuda"

    def __init__(self, compact=True):
        from sobfu_amd import _lib, ops

        self._lib, self._ops, self.compact = _lib, ops, bool(compact)
        self._cache = {}

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _call(self, name, *args):
        self._lib.check(getattr(self._lib.lib(), name)(*args, self._stream()), name)

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr())

    def init_identity(self, psi, layout):
        self._call("sobfu_hip_tile3_init_identity", self._p(psi), *layout.L, *layout.base)

    def _buf(self, key, shape):
        t = self._cache.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.zeros(shape, dtype=torch.float32, device="cuda")
            self._cache[key] = t
        return t

    def begin(self, layout, pg_local, pn_full, pnp_local, psi_local):
        X, Y, Z = layout.dims
        Lx, Ly, Lz = layout.L
        st = _SlabState(layout, None)
        st.pn_full, st.pnp, st.psi = pn_full, pnp_local, psi_local
        if not self.compact:
            st.nabla_U = self._buf("nU4", (Lz, Ly, Lx, 4))
            st.c_psi, st.c_f, st.c_g, st.c_n = psi_local, pnp_local, pg_local, pn_full
            self._call("sobfu_hip_tile3_apply", self._p(pn_full), X, Y, Z, self._p(pnp_local), self._p(psi_local), Lx, Ly, Lz)
            return st
        st.nabla_U = self._buf("nU3", (Lz, Ly, Lx, 3))
        st.c_psi, st.c_f, st.c_g = self._buf("psi3", (Lz, Ly, Lx, 3)), self._buf("f", (Lz, Ly, Lx)), self._buf("g", (Lz, Ly, Lx))
        st.c_n = self._buf("n", (Z, Y, X))
        nl, nf = C.c_size_t(Lz * Ly * Lx), C.c_size_t(Z * Y * X)
        self._call("sobfu_hip_pack_vec3", self._p(psi_local), self._p(st.c_psi), nl)
        self._call("sobfu_hip_extract_tsdf", self._p(pg_local), self._p(st.c_g), nl)
        self._call("sobfu_hip_extract_tsdf", self._p(pn_full), self._p(st.c_n), nf)
        self._call("sobfu_hip_tile3_apply_tsdf_only", self._p(st.c_n), X, Y, Z, self._p(st.c_f), self._p(st.c_psi), Lx, Ly, Lz)  # solver.cu:106
        return st

    def pass_a(self, st, box, w_reg, prev_slots, thr, thin=False):
        """nabla_U on the local cells of `box` = (x0, x1, y0, y1, z0, z1)"""
        if min(box[1] - box[0], box[3] - box[2], box[5] - box[4]) <= 0:
            return
        prev = self._p(prev_slots) if prev_slots is not None else None
        self._call("sobfu_hip_tile3_potential_gradient", self._p(st.c_f), self._p(st.c_g), self._p(st.c_psi), self._p(st.nabla_U),
                   C.c_float(w_reg), *st.layout.L, (C.c_int * 6)(*box), 1 if thin else 0, prev, C.c_float(thr), 1 if self.compact else 0)

    def pass_b(self, st, box, slots, taps, alpha, prev_slots, thr, thin=False):
        """psi update + warp on the local cells of `box`"""
        if min(box[1] - box[0], box[3] - box[2], box[5] - box[4]) <= 0:
            return
        L = st.layout
        prev = self._p(prev_slots) if prev_slots is not None else None
        self._call("sobfu_hip_tile3_smooth_update_apply", self._p(st.nabla_U), self._p(st.c_psi), self._p(st.c_n), self._p(st.c_f), None,
                   self._p(slots), (C.c_float * 7)(*[float(v) for v in taps[:7]]), C.c_float(alpha), *L.L, *L.dims, (C.c_int * 6)(*L.own_box()),
                   (C.c_int * 6)(*box), 1 if thin else 0, prev, C.c_float(thr), 1 if self.compact else 0)

    def end(self, st):
        if not self.compact:
            return
        L = st.layout
        self._call("sobfu_hip_unpack_vec3", self._p(st.c_psi), self._p(st.psi), C.c_size_t(L.L[0] * L.L[1] * L.L[2]))
        self._call("sobfu_hip_tile3_apply", self._p(st.pn_full), *L.dims, self._p(st.pnp), self._p(st.psi), *L.L)  # state of solver.cu:168

    def sobolev_filter(self, s, lam):
        return self._ops.sobolev_filter(s, lam)

    def synchronize(self):
        torch.cuda.synchronize()


class TiledSolver:
    """The gradient-descent loop of sobfu::device::estimate_psi (reference src/sobfu/cuda/solver.cu:106-193) on tiles."""

    def __init__(self, dims, *, alpha, w_reg, s=7, lam=0.1, max_update_norm=-1.0, backend=None, group=None, grid=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.group = group
        self.backend = backend or HipBackend()
        self.layout = TileLayout(dims, grid or (1, 1, self.world), self.rank)
        if self.layout.world != self.world:
            raise ValueError(f"tile grid {grid} needs {self.layout.world} ranks, the group has {self.world}")
        self.alpha, self.w_reg, self.thr = float(alpha), float(w_reg), float(max_update_norm)
        if s < 7:
            raise ValueError("S < 7 is unsupported (the kernels use 7 taps, reference solver.cu:211-234)")
        self.taps = np.asarray(self.backend.sobolev_filter(s, lam), np.float32)[:7]
        self.slots = None

    # -- state helpers ------------------------------------------------------------------------------------------
    def new_local(self, channels):
        return torch.zeros(self.layout.local_shape(channels), dtype=torch.float32, device=self.backend.device)

    def identity_psi(self):
        psi = self.new_local(4)
        self.backend.init_identity(psi, self.layout)
        return psi

    def iterate(self, phi_global_local, phi_n_full, phi_n_psi_local, psi_local, n_iters):
        """Runs n_iters iterations (fewer if the convergence test fires).  Returns (iterations, per-iteration max norms)."""
        L, be = self.layout, self.backend
        can_converge = self.thr >= 0.0
        st = be.begin(L, phi_global_local, phi_n_full, phi_n_psi_local, psi_local)  # includes the warp of solver.cu:106
        slots = torch.zeros((n_iters + 1, SLOTS), dtype=torch.int32, device=be.device)
        self.slots = slots
        xch, pack, unpack = halo_ops(L, [(st.nabla_U, HALO)], self.group) if self.world > 1 else ([], lambda: None, lambda: None)
        ox, oy = (L.o0[0], L.o1[0]), (L.o0[1], L.o1[1])
        lo, hi, H = L.own_lo, L.own_hi, HALO
        b_boxes = L.pass_b_boxes()
        if L.slab:
            # z-slabs: planes next to an interior face (sent to the neighbour) vs the rest, so that the exchange overlaps the
            # interior compute; ranges are local plane indices
            a_lo = min(lo + H, hi) if L.lo else lo          # [lo, a_lo)  : lower boundary planes of pass A
            a_hi = max(hi - H, a_lo) if L.hi else hi        # [a_hi, hi)  : upper boundary planes of pass A
            b_lo = min(lo + 3, hi) if L.lo else lo          # pass B planes >= b_lo have all -3 taps inside the owned range
            b_hi = max(hi - 3, b_lo) if L.hi else hi
            b_first = lo - 1 if L.lo else lo                # pass B also refreshes the first halo plane (owned +-1)
            b_last = hi + 1 if L.hi else hi
        for it in range(1, n_iters + 1):
            prev = slots[it - 1] if (it > 1 and can_converge) else None
            row = slots[it]
            if L.slab:
                be.pass_a(st, ox + oy + (lo, a_lo), self.w_reg, prev, self.thr)
                be.pass_a(st, ox + oy + (a_hi, hi), self.w_reg, prev, self.thr)
                works = start_halo_ops(xch)
                be.pass_a(st, ox + oy + (a_lo, a_hi), self.w_reg, prev, self.thr)
                be.pass_b(st, ox + oy + (b_lo, b_hi), row, self.taps, self.alpha, prev, self.thr)
                finish_halo_ops(works)
                be.pass_b(st, ox + oy + (b_first, b_lo), row, self.taps, self.alpha, prev, self.thr)
                be.pass_b(st, ox + oy + (b_hi, b_last), row, self.taps, self.alpha, prev, self.thr)
            else:
                be.pass_a(st, L.own_box(), self.w_reg, prev, self.thr)
                pack()
                finish_halo_ops(start_halo_ops(xch))
                unpack()
                for box, tr in b_boxes:
                    be.pass_b(st, box, row, self.taps, self.alpha, prev, self.thr, thin=tr)
            if self.world > 1 and can_converge:
                dist.all_reduce(slots[it], op=dist.ReduceOp.MAX, group=self.group)  # the gate needs the GLOBAL max
        if self.world > 1 and not can_converge:
            dist.all_reduce(slots, op=dist.ReduceOp.MAX, group=self.group)
        be.end(st)
        be.synchronize()
        mx = slots[1:].max(dim=1).values.cpu().numpy().view(np.uint32)
        norms = np.array([_sqrt_rd(int(b)) for b in mx], np.float32)
        done = n_iters
        if can_converge:
            for k, v in enumerate(norms):
                if v <= self.thr:  # solver.cu:183 -- later iterations were device-side no-ops
                    done = k + 1
                    break
        return done, norms[:done]

    def estimate_psi(self, *args, **kw):
        return estimate_psi_tiled(self, *args, **kw)

    def gather_owned(self, local):
        """all_gather of the owned cells -> full volume on every rank"""
        return gather_owned(self.layout, local, self.group)
