"""Multi-GPU solver loop: the volume is cut into a Px x Py x Pz grid of TILES, one rank (one process, one GPU) per tile.

SURVEY.md section 8(e).  Frames of a sequence are sequentially dependent (psi persists), so the path shards by VOLUME TILE:
2 x 2 x 2 tiles of 128^3 on 8 GPUs for the 256^3 grid (BASELINE config 4); 1 x 1 x N are z-slabs, whose halos are whole planes
(z is the slowest-varying axis of the layout) and travel zero-copy straight out of / into the field arrays.

Per rank:  psi, phi_n o psi, phi_global, nabla_U are LOCAL arrays = owned cells + HALO (= 4) cells on every side that faces a
neighbour; phi_n is replicated (the warp gathers at absolute coordinates anywhere in the volume).

One iteration = ONE exchange (Option B of the survey):
    A      nabla_U on the owned cells (reads psi, phi_n o psi at owned +- 1 along each axis)
    E      the 4-cell faces of nabla_U go to the face neighbours and 4 x 4 edge strips to the edge neighbours (one message per
           neighbour, one grouped RCCL send/recv); corners are never needed: every stencil is axis-aligned
    B      psi, phi_n o psi on owned +- 1 along each axis: the radius-3 convolution is exact there, so psi and phi_n o psi stay
           exact on the one-cell shells -- all the next pass A reads -- without ever being exchanged (invariant; identity psi
           satisfies it at the start).  A shell cell one step outside along axis a reads nabla_U up to 3 further along a (face
           halo, hence width 4) and up to 3 along any other axis b (edge strip).  max ||u||^2 takes OWNED cells only.
    R      all_reduce(MAX) of the 256 max-norm slots -- only when max_update_norm >= 0 (otherwise the test never fires)
z-slabs overlap E with the interior compute (the kernels take the range of planes a launch produces):
    A_bnd (4 owned planes next to each interior face) -> E || A_int, B_int (planes whose +-3 taps are all owned) -> wait E -> B_bnd
Halo cells further out are never read.  Clamp / mirror rules act at a tile's array edge, which is the volume boundary exactly
where the tile has no halo, so the result equals the single-GPU run bit for bit (the max is order-independent).

The kernel backend is pluggable: `HipBackend` (product; C ABI on torch CUDA tensors over RCCL) -- the CPU tests inject an
oracle-backed backend over gloo to check the decomposition logic without a GPU.
"""
from __future__ import annotations

import ctypes as C
import itertools
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HALO = 4
SLOTS = 256


def _x_pad():
    import os

    return os.environ.get("SOBFU_TILE_XPAD", "0") == "1"


def default_grid(world):
    """Tile grid for `world` ranks: as cubic as the factorisation allows, larger factors on the slower axes (x, the axis the
    64 lanes of a wave run along, is split last): 8 -> 2 x 2 x 2 (BASELINE config 4), 4 -> 1 x 2 x 2, 2 -> 1 x 1 x 2."""
    best = None
    for px in range(1, world + 1):
        if world % px:
            continue
        for py in range(1, world // px + 1):
            if (world // px) % py:
                continue
            pz = world // (px * py)
            if px <= py <= pz:
                key = (pz - px, pz)
                if best is None or key < best[0]:
                    best = (key, (px, py, pz))
    return best[1]


def parse_grid(text, world):
    """'2x2x2' -> (2, 2, 2); '' -> default_grid(world)"""
    if not text:
        return default_grid(world)
    g = tuple(int(v) for v in text.lower().split("x"))
    if len(g) != 3 or g[0] * g[1] * g[2] != world or min(g) < 1:
        raise ValueError(f"tile grid {text!r} does not have {world} tiles")
    return g


class TileLayout:
    """Which cells a rank owns and how its local arrays are laid out.  rank = cx + Px * (cy + Py * cz); index triples are
    (x, y, z); the cells of every axis are split as evenly as possible (sobfu_hip_tiled_create3 computes the same layout)."""

    def __init__(self, dims, grid, rank, halo=HALO):
        dims, grid = tuple(int(d) for d in dims), tuple(int(g) for g in grid)
        world = grid[0] * grid[1] * grid[2]
        if not 0 <= rank < world:
            raise ValueError("rank outside the tile grid")
        self.dims, self.grid, self.world, self.rank, self.halo = dims, grid, world, rank, halo
        self.coords = (rank % grid[0], (rank // grid[0]) % grid[1], rank // (grid[0] * grid[1]))
        g0, g1, lo, hi = [], [], [], []
        for a in range(3):
            base, rem = divmod(dims[a], grid[a])
            if grid[a] > 1 and base < halo:
                raise ValueError(f"axis {'xyz'[a]} ({dims[a]} cells) is too thin for {grid[a]} tiles with halo {halo}")
            c = self.coords[a]
            g0.append(c * base + min(c, rem))
            g1.append(g0[-1] + base + (1 if c < rem else 0))
            lo.append(halo if c > 0 else 0)
            hi.append(halo if c < grid[a] - 1 else 0)
            if a == 0 and _x_pad() and grid[0] > 1 and base >= 64:  # aligned rows: see make_layout in csrc/tiled_capi.hip
                if lo[-1]:
                    lo[-1] = 32
                if hi[-1]:
                    hi[-1] = halo + (32 - (lo[-1] + (g1[-1] - g0[-1]) + halo) % 32) % 32
        self.g0, self.g1, self.lo3, self.hi3 = tuple(g0), tuple(g1), tuple(lo), tuple(hi)
        self.L = tuple(g1[a] - g0[a] + lo[a] + hi[a] for a in range(3))              # local extents
        self.o0 = tuple(lo)                                                           # owned local range [o0, o1)
        self.o1 = tuple(lo[a] + g1[a] - g0[a] for a in range(3))
        self.base = tuple(g0[a] - lo[a] for a in range(3))                            # global coordinate of local cell 0
        self.slab = grid[0] == 1 and grid[1] == 1
        # the z entries under the names the slab schedules use
        self.z0, self.z1, self.lo, self.hi = g0[2], g1[2], lo[2], hi[2]
        self.Lz, self.own_lo, self.own_hi, self.zbase = self.L[2], self.o0[2], self.o1[2], self.base[2]

    def local_shape(self, channels=None):
        shp = (self.L[2], self.L[1], self.L[0])
        return shp if channels is None else shp + (channels,)

    def take(self, full):
        """local array (with halos) cut out of a full-volume array / tensor"""
        b, L = self.base, self.L
        return full[b[2]:b[2] + L[2], b[1]:b[1] + L[1], b[0]:b[0] + L[0]]

    def owned(self, local):
        return local[self.o0[2]:self.o1[2], self.o0[1]:self.o1[1], self.o0[0]:self.o1[0]]

    def owned_global(self, full):
        return full[self.g0[2]:self.g1[2], self.g0[1]:self.g1[1], self.g0[0]:self.g1[0]]

    def own_box(self):
        return (self.o0[0], self.o1[0], self.o0[1], self.o1[1], self.o0[2], self.o1[2])

    def pass_b_boxes(self):
        """[(box, thin)] pass B produces: the owned cells with their one-cell z shells (extra planes of the march) and the one-cell
        y / x shells as THIN boxes (evaluated lane per cell).  Cells on tile edges (two shells at once) are never read."""
        o0, o1, lo, hi = self.o0, self.o1, self.lo3, self.hi3
        ext = lambda a: (o0[a] - (1 if lo[a] else 0), o1[a] + (1 if hi[a] else 0))  # noqa: E731
        own = lambda a: (o0[a], o1[a])  # noqa: E731
        boxes = [(own(0) + own(1) + ext(2), False)]
        if lo[1]:
            boxes.append((own(0) + (o0[1] - 1, o0[1]) + own(2), True))
        if hi[1]:
            boxes.append((own(0) + (o1[1], o1[1] + 1) + own(2), True))
        if lo[0]:
            boxes.append(((o0[0] - 1, o0[0]) + own(1) + own(2), True))
        if hi[0]:
            boxes.append(((o1[0], o1[0] + 1) + own(1) + own(2), True))
        return boxes

    def messages(self, width=None):
        """[(peer, send_box, recv_box)] of one exchange: every face neighbour (one non-zero offset) and edge neighbour (two).
        Along an axis with offset +1 the `width` owned cells next to that face go out and the `width` halo cells beyond it come
        in; along an axis with offset 0 the owned range (the neighbour shares that coordinate, hence the range)."""
        w = self.halo if width is None else width
        out = []
        for dz, dy, dx in itertools.product((-1, 0, 1), repeat=3):
            d = (dx, dy, dz)
            nnz = sum(1 for v in d if v)
            if nnz < 1 or nnz > 2:
                continue
            n = tuple(self.coords[a] + d[a] for a in range(3))
            if any(n[a] < 0 or n[a] >= self.grid[a] for a in range(3)):
                continue
            sb, rb = [], []
            for a in range(3):
                if d[a] > 0:
                    sb += [self.o1[a] - w, self.o1[a]]
                    rb += [self.o1[a], self.o1[a] + w]
                elif d[a] < 0:
                    sb += [self.o0[a], self.o0[a] + w]
                    rb += [self.o0[a] - w, self.o0[a]]
                else:
                    sb += [self.o0[a], self.o1[a]]
                    rb += [self.o0[a], self.o1[a]]
            out.append((n[0] + self.grid[0] * (n[1] + self.grid[1] * n[2]), tuple(sb), tuple(rb)))
        return out


class SlabLayout(TileLayout):
    """z-slabs: the 1 x 1 x world tile grid"""

    def __init__(self, dims, world, rank, halo=HALO):
        if int(dims[2]) < world * halo:
            raise ValueError(f"Z={dims[2]} is too thin for {world} slabs with halo {halo}")
        super().__init__(dims, (1, 1, world), rank, halo)
        if world > 1 and (self.z1 - self.z0) < halo:
            raise ValueError("a slab must own at least `halo` planes")


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
        from . import _lib, ops

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


def _sqrt_rd(m_bits: int) -> float:
    m = np.array([m_bits], np.uint32).view(np.float32)[0]
    r = np.sqrt(m, dtype=np.float32)
    if r > 0 and np.float64(r) * np.float64(r) > np.float64(m):
        r = np.nextafter(r, np.float32(-np.inf), dtype=np.float32)
    return float(r)


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


def gather_owned(layout, local, group=None):
    L = layout
    own = L.owned(local).contiguous()
    if L.world == 1:
        return own
    lays = [TileLayout(L.dims, L.grid, r) for r in range(L.world)]
    parts = [torch.empty((l.g1[2] - l.g0[2], l.g1[1] - l.g0[1], l.g1[0] - l.g0[0]) + tuple(own.shape[3:]), dtype=own.dtype, device=own.device)
             for l in lays]
    dist.all_gather(parts, own, group=group)
    if L.slab:
        return torch.cat(parts, dim=0)
    X, Y, Z = L.dims
    full = torch.empty((Z, Y, X) + tuple(own.shape[3:]), dtype=own.dtype, device=own.device)
    for l, part in zip(lays, parts):
        l.owned_global(full).copy_(part)
    return full


def estimate_psi_tiled(solver, phi_global_local, phi_global_psi_inv_local, phi_n_full, phi_n_psi_local, psi_local, psi_inv_local,
                       n_iters, inverse_iters=48, gather=None):
    """One frame's Solver::estimate_psi (src/sobfu/cuda/solver.cu:85-205) on tiles: the tiled iteration loop, then the
    two per-frame collectives of SURVEY 8(e) -- all-gather psi for the 48-sweep inverse (it gathers psi at arbitrary
    psi^-1(x)), all-gather phi_global for the canonical -> live warp -- each followed by the tile kernel.  HIP backend only
    (C ABI: sobfu_hip_tile3_init_identity / tile3_estimate_inverse / tile3_apply).  `gather` overrides solver.gather_owned
    (tests).  Returns (iterations, per-iteration max norms)."""
    from . import _lib

    L = solver.layout
    lib, st = _lib.lib(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    gather = gather or solver.gather_owned
    done, norms = solver.iterate(phi_global_local, phi_n_full, phi_n_psi_local, psi_local, n_iters)
    psi_full = gather(psi_local)                                                              # solver.cu:196-197
    _lib.check(lib.sobfu_hip_tile3_init_identity(C.c_void_p(psi_inv_local.data_ptr()), *L.L, *L.base, st), "tile3_init_identity")
    _lib.check(lib.sobfu_hip_tile3_estimate_inverse(C.c_void_p(psi_full.data_ptr()), *L.dims, C.c_void_p(psi_inv_local.data_ptr()), *L.L,
                                                    *L.base, C.c_int(inverse_iters), st), "tile3_estimate_inverse")
    pg_full = gather(phi_global_local)                                                        # solver.cu:199
    _lib.check(lib.sobfu_hip_tile3_apply(C.c_void_p(pg_full.data_ptr()), *L.dims, C.c_void_p(phi_global_psi_inv_local.data_ptr()),
                                         C.c_void_p(psi_inv_local.data_ptr()), *L.L, st), "tile3_apply")
    torch.cuda.current_stream().synchronize()  # psi_full / pg_full die with this frame
    return done, norms


class TiledFusion:
    """SobFusion::operator() (src/sobfu/sob_fusion.cpp:71-145) with the volume cut into tiles: every rank runs the depth
    pre-steps (640 x 480: negligible), integrates ITS tile of phi_global and the whole phi_n (the warp gathers anywhere), the
    solve runs tiled, the fusion is per cell (run on the whole local array: only owned cells are ever read back).
    phi_global / phi_n o psi / psi / psi^-1 live as local arrays.

    params: dims, size (metres), trunc, eta (metres), max_weight, intr (fx, fy, cx, cy), R, t (volume -> camera), start_frame,
    bilateral (ksz, sigma_spatial, sigma_depth), trunc_depth, max_iter; `solver` is a TiledSolver / NativeTiledSolver."""

    def __init__(self, solver, params, gather=None):
        from . import ops

        self.ops, self.solver, self.P, self.gather = ops, solver, params, gather
        self.L = solver.layout
        self.frame = 0
        X, Y, Z = self.L.dims
        self.vs = tuple(float(params["size"][i]) / self.L.dims[i] for i in range(3))
        self.phi_global = self.phi_n = self.phi_n_psi = self.phi_global_psi_inv = self.psi = self.psi_inv = None

    def __call__(self, depth_u16):
        ops, P, L, s = self.ops, self.P, self.L, self.solver
        ks, ss, sd = P["bilateral"]
        d = ops.bilateral_filter(depth_u16, ks, ss, sd)                              # sob_fusion.cpp:78
        ops.truncate_depth(d, P["trunc_depth"])                                      # :85
        dists = ops.compute_dists(d, P["intr"])                                      # :91
        if self.frame == 0:                                                          # :93-123
            self.phi_global = s.new_local(2)
            ops.tile3_integrate_depth(dists, self.phi_global, L.base, self.vs, P["trunc"], P["eta"], P["R"], P["t"], P["intr"])
            self.phi_n = ops.new_volume(L.dims)
            self.phi_n_psi, self.phi_global_psi_inv = s.new_local(2), s.new_local(2)
            self.psi, self.psi_inv = s.identity_psi(), s.identity_psi()
            self.frame += 1
            return None
        ops.clear_volume(self.phi_n)                                                 # :129
        ops.integrate_depth(dists, self.phi_n, self.vs, P["trunc"], P["eta"], P["R"], P["t"], P["intr"])  # :130
        result = None
        if self.frame < P["start_frame"]:                                            # :136-139
            ops.integrate_fuse(self.phi_global, L.take(self.phi_n).contiguous(), P["max_weight"])
        else:
            result = s.estimate_psi(self.phi_global, self.phi_global_psi_inv, self.phi_n, self.phi_n_psi, self.psi, self.psi_inv,
                                    P["max_iter"], gather=self.gather)               # :141
            ops.integrate_fuse(self.phi_global, self.phi_n_psi, P["max_weight"])     # :142 (halo cells come out stale; never read)
        self.frame += 1
        return result


class TiledMsg(C.Structure):
    """sobfu_hip_tiled_msg"""
    _fields_ = [("peer", C.c_int), ("send_off", C.c_size_t), ("recv_off", C.c_size_t), ("count", C.c_size_t)]


class TiledExports(C.Structure):
    """sobfu_hip_tiled_exports"""
    _fields_ = [("arena", C.c_void_p), ("nabla_u_off", C.c_size_t * 2), ("rows_off", C.c_size_t), ("flags", C.c_void_p)]


# hipIpc bookkeeping of this process (direct transport): a device block is exported ONCE (the library parks and reuses the blocks
# it has exported) and a peer's handle is opened ONCE -- the mappings live as long as the process
_IPC_EXPORTED, _IPC_OPENED = {}, {}


class NativeTiledSolver:
    """The same tile loop run entirely in C++ (sobfu_amd/csrc/tiled_capi.hip), no Python per iteration.  Transports:
    "direct" -- halo cells stored straight into the neighbours' arrays over xGMI from pass A's launch (peer-mapped with hipIpc;
    torch.distributed is used once, to hand the 64-byte handles around), two launches per iteration and no collective in the
    loop; "rccl" -- RCCL send/recv issued from the library (z-slabs: on a dedicated communication stream, overlapped with the
    interior compute; torch.distributed hands the unique id to every rank)."""

    def __init__(self, dims, *, alpha, w_reg, s=7, lam=0.1, max_update_norm=-1.0, group=None, dry=None, grid=None, transport=None):
        """grid: (Px, Py, Pz) tiles (default: z-slabs, 1 x 1 x world).  dry=(world, rank): a communicator-less handle with that
        rank's layout -- every launch of the rank's schedule without peers (halos are stale unless a transport is plugged in with
        set_transport / connect_local; alone, only its TIMING means anything: tools/tile_time_native.py).  transport: "rccl"
        (default with a process group) or "direct"."""
        import os

        from . import _lib
        from ._lib import SolverParams

        self._lib = _lib
        L = _lib.lib()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.transport = transport or ("none" if dry is not None else "rccl")
        self._opened = []
        if self.transport == "direct" and dry is None:
            dry = (self.world, self.rank)  # a communicator-less handle; connect_ipc() below maps the peers
            self._connect_group = True
        else:
            self._connect_group = False
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        _lib.check(L.sobfu_hip_tiled_load_rccl(path.encode()), "tiled_load_rccl")
        uid = (C.c_char * 128)()
        if dry is not None:
            self.world, self.rank = dry
        elif self.rank == 0:
            _lib.check(L.sobfu_hip_tiled_unique_id(uid), "tiled_unique_id")
        if self.world > 1 and dry is None:  # noqa: SIM102
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        grid = tuple(int(g) for g in grid) if grid else (1, 1, self.world)
        if grid[0] * grid[1] * grid[2] != self.world:
            raise ValueError(f"tile grid {grid} does not have {self.world} tiles")
        self.grid = grid
        # the layout is validated HERE, identically on every rank (same dims, same grid), before any collective: a grid that is too
        # thin for its halos raises everywhere instead of failing inside create3 on some ranks while the others sit in ncclCommInitRank
        for q in range(self.world):
            TileLayout(dims, grid, q)
        self.params = SolverParams(0, 0, s, max_update_norm, np.float32(lam), alpha, w_reg)
        self._h = C.c_void_p()
        X, Y, Z = (int(d) for d in dims)
        _lib.check(L.sobfu_hip_tiled_create3(C.byref(self._h), X, Y, Z, *grid, self.rank, uid, C.byref(self.params)), "tiled_create3")
        # opt-in ("1"; "force": also on a world of one, bring-up / tests): two communicators with operations in flight at once
        # are a known RCCL hang hazard and this mode has never run on >= 2 real GPUs -- the default keeps the max-norm
        # all-reduce on the main communicator, in line behind the exchange
        want2 = os.environ.get("SOBFU_TILED_REDUCE_COMM", "0")
        if dry is None and ((self.world > 1 and want2 == "1") or want2 == "force"):
            # a communicator of its own for the max-norm all-reduce (collective; every rank or none)
            uid2 = (C.c_char * 128)()
            if self.rank == 0:
                _lib.check(L.sobfu_hip_tiled_unique_id(uid2), "tiled_unique_id")
            if self.world > 1:
                box = [bytes(uid2)]
                dist.broadcast_object_list(box, src=0, group=group)
                uid2 = (C.c_char * 128).from_buffer_copy(box[0])
            _lib.check(L.sobfu_hip_tiled_add_reduce_comm(self._h, uid2), "tiled_add_reduce_comm")
        self.has_comm = dry is None
        self.schedule = 0
        self.layout = TileLayout(dims, grid, self.rank)
        v = (C.c_int * 24)()
        _lib.check(L.sobfu_hip_tiled_layout3(self._h, v), "tiled_layout3")
        lay = self.layout
        want = lay.grid + lay.coords + lay.g0 + lay.g1 + lay.lo3 + lay.hi3 + lay.L + lay.base
        assert tuple(v) == want, (tuple(v), want)
        # the library's message table must be the one TileLayout.messages() describes (both sides of the C ABI build it)
        mm, sb, rb = (TiledMsg * 18)(), (C.c_int * (6 * 18))(), (C.c_int * (6 * 18))()
        n = L.sobfu_hip_tiled_messages(self._h, mm, sb, rb, 18)
        mine = lay.messages()
        assert n == len(mine), (n, len(mine))
        for i, (peer, sbox, rbox) in enumerate(mine):
            assert mm[i].peer == peer and tuple(sb[6 * i:6 * i + 6]) == sbox and tuple(rb[6 * i:6 * i + 6]) == rbox, (i, peer, sbox, rbox)
        if self._connect_group:
            self.connect_ipc()

    # -- direct transport ----------------------------------------------------------------------------------------------
    def exports(self):
        e = TiledExports()
        self._lib.check(self._lib.lib().sobfu_hip_tiled_exports_get(self._h, C.byref(e)), "tiled_exports_get")
        return e

    def connect(self, ranks, exports):
        """hand the library the device pointers (valid in THIS process) of the other ranks' exported arrays"""
        n = len(ranks)
        self._lib.check(self._lib.lib().sobfu_hip_tiled_connect(self._h, n, (C.c_int * max(n, 1))(*ranks), (TiledExports * max(n, 1))(*exports)),
                        "tiled_connect")
        self.transport = "direct"

    @staticmethod
    def connect_local(solvers):
        """ranks that live in ONE process (tests): plain pointers, no IPC"""
        ex = [s.exports() for s in solvers]
        for s in solvers:
            others = [q for q in range(len(solvers)) if q != s.rank]
            s.connect(others, [ex[q] for q in others])

    def connect_ipc(self):
        """one process per rank: every rank exports its two allocations as 64-byte hipIpc handles, all ranks gather them over
        torch.distributed (any backend) and map the others' arrays; a barrier, so that nobody begins a solve before every rank
        is connected (and has cleared its halo cells).  Every step is agreed on collectively: a rank that fails reports it in the
        NEXT gather, so the ranks never sit in different collectives -- all of them raise."""
        lib, check = self._lib.lib(), self._lib.check
        mine, err = [], None
        try:
            e = self.exports()
            import os

            if os.environ.get("SOBFU_TEST_FAIL_EXPORT_RANK") == str(self.rank):  # tests: this rank cannot export
                raise RuntimeError("ipc_export(flags): invalid argument (code 1) [forced by SOBFU_TEST_FAIL_EXPORT_RANK]")
            for what, ptr in (("arena", e.arena), ("flags", e.flags)):
                if ptr not in _IPC_EXPORTED:
                    h = (C.c_char * 64)()
                    check(lib.sobfu_hip_ipc_export(C.c_void_p(ptr), h), f"ipc_export({what})")
                    _IPC_EXPORTED[ptr] = bytes(h)
                mine.append(_IPC_EXPORTED[ptr])
            mine.append((e.nabla_u_off[0], e.nabla_u_off[1], e.rows_off))
        except Exception as ex:  # noqa: BLE001
            err = f"rank {self.rank}: export: {ex!r}"
        allh = [None] * self.world
        dist.all_gather_object(allh, (mine, err), group=self.group)
        errs = [a[1] for a in allh if a[1]]
        if not errs:
            try:
                ranks, exps = [], []
                for q in range(self.world):
                    if q == self.rank:
                        continue
                    ptrs = []
                    for hb in allh[q][0][:2]:
                        if (q, hb) not in _IPC_OPENED:
                            out = C.c_void_p()
                            check(lib.sobfu_hip_ipc_open((C.c_char * 64).from_buffer_copy(hb), C.byref(out)), f"ipc_open (rank {q})")
                            _IPC_OPENED[(q, hb)] = out.value
                        ptrs.append(_IPC_OPENED[(q, hb)])
                    x = TiledExports()
                    x.arena, x.flags = ptrs
                    x.nabla_u_off[0], x.nabla_u_off[1], x.rows_off = allh[q][0][2]
                    ranks.append(q)
                    exps.append(x)
                self.connect(ranks, exps)
            except Exception as ex:  # noqa: BLE001
                err = f"rank {self.rank}: map: {ex!r}"
            st = [None] * self.world
            dist.all_gather_object(st, err, group=self.group)  # doubles as the barrier
            errs = [a for a in st if a]
        if errs:
            print(f"[rank {self.rank}] direct transport: {errs}", file=sys.stderr, flush=True)
            raise RuntimeError("direct transport: peer mapping failed: " + "; ".join(errs))

    def status(self):
        """(ok, missing_peer): whether every peer has answered within the deadline so far"""
        m = C.c_int(-1)
        rc = self._lib.lib().sobfu_hip_tiled_status(self._h, C.byref(m))
        return rc == 0, m.value

    def set_wait(self, wait):
        self._lib.check(self._lib.lib().sobfu_hip_tiled_set_wait(self._h, C.c_int(1 if wait else 0)), "tiled_set_wait")

    def step_phase(self, phase):
        self._lib.check(self._lib.lib().sobfu_hip_tiled_step_phase(self._h, C.c_int(int(phase)), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "tiled_step_phase")

    EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(TiledMsg), C.c_int, C.c_void_p)
    ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)

    def set_transport(self, exchange, allreduce_max):
        """communicator-less (dry) handles only: callables exchange(rank, send_ptr, recv_ptr, [(peer, send_off, recv_off, count)],
        stream) (offsets / counts in floats) and allreduce_max(rank, buf_ptr, n, stream) -> 0"""
        self._cb = (self.EXCHANGE_FN(lambda ctx, r, sp, rp, m, n, st: exchange(r, sp or 0, rp or 0,
                                                                             [(m[i].peer, m[i].send_off, m[i].recv_off, m[i].count) for i in range(n)], st)),
                    self.ALLREDUCE_FN(lambda ctx, r, b, n, st: allreduce_max(r, b, n, st)))
        self._lib.check(self._lib.lib().sobfu_hip_tiled_set_transport(self._h, self._cb[0], self._cb[1], None), "tiled_set_transport")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lib().sobfu_hip_tiled_destroy(self._h)
            self._h = None
            for ptr in getattr(self, "_opened", []):
                self._lib.lib().sobfu_hip_ipc_close(C.c_void_p(ptr))
            self._opened = []

    __del__ = close

    def new_local(self, channels):
        return torch.zeros(self.layout.local_shape(channels), dtype=torch.float32, device="cuda")

    def identity_psi(self):
        psi = self.new_local(4)
        self._lib.check(self._lib.lib().sobfu_hip_tile3_init_identity(C.c_void_p(psi.data_ptr()), *self.layout.L, *self.layout.base,
                                                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tile3_init_identity")
        return psi

    def iterate(self, phi_global_local, phi_n_full, phi_n_psi_local, psi_local, n_iters):
        from ._lib import SolverReport

        rep = SolverReport()
        hist = (C.c_float * max(1, n_iters))()
        self._lib.check(self._lib.lib().sobfu_hip_tiled_iterate(
            self._h, C.c_void_p(phi_global_local.data_ptr()), C.c_void_p(phi_n_full.data_ptr()), C.c_void_p(phi_n_psi_local.data_ptr()),
            C.c_void_p(psi_local.data_ptr()), C.c_int(n_iters), C.byref(rep), hist, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            "tiled_iterate")
        return rep.iterations, np.array(hist[:rep.iterations], np.float32)

    def begin(self, phi_global_local, phi_n_full, phi_n_psi_local, psi_local, max_iters):
        """the loop in pieces (sobfu_hip_tiled_begin / step / end): step(n) ENQUEUES n iterations without synchronising"""
        self._session = (phi_global_local, phi_n_full, phi_n_psi_local, psi_local, int(max_iters))  # keeps the buffers alive
        self._lib.check(self._lib.lib().sobfu_hip_tiled_begin(
            self._h, C.c_void_p(phi_global_local.data_ptr()), C.c_void_p(phi_n_full.data_ptr()), C.c_void_p(phi_n_psi_local.data_ptr()),
            C.c_void_p(psi_local.data_ptr()), C.c_int(int(max_iters)), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tiled_begin")

    def step(self, n_iters):
        self._lib.check(self._lib.lib().sobfu_hip_tiled_step(self._h, C.c_int(int(n_iters)), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "tiled_step")

    def end(self):
        from ._lib import SolverReport

        rep = SolverReport()
        hist = (C.c_float * max(1, self._session[4]))()
        self._lib.check(self._lib.lib().sobfu_hip_tiled_end(self._h, C.byref(rep), hist, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "tiled_end")
        self._session = None
        return rep.iterations, np.array(hist[:rep.iterations], np.float32)

    def set_profiling(self, stride, max_samples=256):
        self._lib.check(self._lib.lib().sobfu_hip_tiled_set_profiling(self._h, C.c_int(int(stride)), C.c_int(int(max_samples))), "tiled_set_profiling")

    def get_profile(self, reset=True):
        """(ms in pass A, ms in the exchange incl. pack / unpack, ms in pass B, iterations timed) -- serial schedule only"""
        ms, n = (C.c_float * 3)(), C.c_int()
        self._lib.check(self._lib.lib().sobfu_hip_tiled_get_profile(self._h, ms, C.byref(n), C.c_int(1 if reset else 0)), "tiled_get_profile")
        return ms[0], ms[1], ms[2], n.value

    def gather_owned(self, local):
        return gather_owned(self.layout, local, self.group)

    def estimate_psi(self, *args, **kw):
        return estimate_psi_tiled(self, *args, **kw)

    # z-slabs: 1 / 2 = exchange overlapped, pass A split into boundary + interior launches or whole; 3-D tiles: 1 = 2 = push boxes as
    # their own launch, exchange + scatter on the communication stream beside pass A's owned block and pass B's interior
    SCHEDULES = {1: "overlapped exchange, pass A split", 2: "overlapped exchange, pass A whole", 3: "serial (no overlap, no events)"}

    def set_schedule(self, schedule):
        self._lib.check(self._lib.lib().sobfu_hip_tiled_set_schedule(self._h, C.c_int(int(schedule))), "tiled_set_schedule")
        self.schedule = int(schedule)

    def autotune(self, phi_global_local, phi_n_full, iters=40):
        """Times the three ways of issuing an iteration on THIS machine (they give identical results) on scratch state and keeps
        the fastest; every rank takes part and all agree (MAX over ranks).  Returns {schedule: us per iteration}."""
        pnp, psi = self.new_local(2), self.identity_psi()
        times = {}
        for sched in (self.SCHEDULES if self.layout.slab else (1, 3)):
            self.set_schedule(sched)
            self.iterate(phi_global_local, phi_n_full, pnp, psi, 4)
            torch.cuda.synchronize()
            if dist.is_initialized():
                dist.barrier(group=self.group)
            t0 = time.perf_counter()
            self.iterate(phi_global_local, phi_n_full, pnp, psi, iters)
            torch.cuda.synchronize()
            t = torch.tensor([(time.perf_counter() - t0) / iters * 1e6], dtype=torch.float64, device="cuda")
            if dist.is_initialized():
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            times[sched] = float(t.item())
        self.set_schedule(min(times, key=times.get))
        return times


class GlooTransport:
    """Transport of a communicator-less native handle over a gloo process group, staged through host memory: lets N ranks that
    SHARE one GPU (bench.py with SOBFU_BENCH_SHARE_GPU=1, bring-up on a machine with fewer GPUs than ranks -- RCCL refuses two
    ranks on one device) run the real multi-process tile loop.  Not a performance path."""

    def __init__(self, group=None):
        self.group = group
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def _ok(self, rc):
        if rc != 0:
            raise RuntimeError(f"hip call failed: {rc}")

    def exchange(self, rank, send, recv, msgs, stream):
        try:
            self._ok(self.hip.hipStreamSynchronize(stream))
            ops, bufs = [], []
            for peer, soff, roff, cnt in msgs:
                out, inn = torch.empty(cnt, dtype=torch.float32), torch.empty(cnt, dtype=torch.float32)
                self._ok(self.hip.hipMemcpy(out.data_ptr(), send + 4 * soff, 4 * cnt, 2))
                ops.append(dist.P2POp(dist.isend, out, peer, self.group))
                ops.append(dist.P2POp(dist.irecv, inn, peer, self.group))
                bufs.append((inn, roff, cnt, out))
            for w in dist.batch_isend_irecv(ops):
                w.wait()
            for inn, roff, cnt, _ in bufs:
                self._ok(self.hip.hipMemcpy(recv + 4 * roff, inn.data_ptr(), 4 * cnt, 1))
            return 0
        except Exception as e:  # noqa: BLE001
            print("gloo transport: exchange failed:", repr(e), file=sys.stderr, flush=True)
            return -1

    def allreduce(self, rank, buf, n, stream):
        try:
            self._ok(self.hip.hipStreamSynchronize(stream))
            h = torch.empty(n, dtype=torch.int32)  # max ||u||^2 bit patterns of non-negative floats order like int32
            self._ok(self.hip.hipMemcpy(h.data_ptr(), buf, 4 * n, 2))
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=self.group)
            self._ok(self.hip.hipMemcpy(buf, h.data_ptr(), 4 * n, 1))
            return 0
        except Exception as e:  # noqa: BLE001
            print("gloo transport: allreduce failed:", repr(e), file=sys.stderr, flush=True)
            return -1


def _tiled_diagnostics(solver, kw, dims, grid, pg, pn_full, world, rank, ranks, reps=30, iters=60):
    """Per-piece timings of the native loop on the machine at hand (microseconds; rank 0's view after a MAX over ranks)."""
    import os

    L, lib, check = solver.layout, solver._lib.lib(), solver._lib.check
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        ranks.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return ranks.max([(time.perf_counter() - t0) / n * 1e6])[0]

    direct = solver.transport == "direct"
    if not direct:  # (the direct transport has no separate exchange step: the stores are part of pass A's launch)
        field = torch.zeros(L.local_shape(3), dtype=torch.float32, device="cuda")
        for planes in ((1, 2, HALO) if L.slab else (HALO,)):  # latency vs bandwidth of a face message (z-slabs: 1/4, 1/2 and all of the halo)
            out[f"exchange_{planes}_cells_us"] = timed(
                lambda: check(lib.sobfu_hip_tiled_exchange(solver._h, C.c_void_p(field.data_ptr()), C.c_int(planes), st), "exchange"), reps)
    msgs = L.messages()
    out["exchange_messages"] = len(msgs)
    out["exchange_bytes_out"] = sum((m[1][1] - m[1][0]) * (m[1][3] - m[1][2]) * (m[1][5] - m[1][4]) for m in msgs) * 12
    out["exchange_largest_message_bytes"] = max([(m[1][1] - m[1][0]) * (m[1][3] - m[1][2]) * (m[1][5] - m[1][4]) for m in msgs] or [0]) * 12
    if solver.has_comm:
        slots = torch.zeros(SLOTS, dtype=torch.int32, device="cuda")
        out["allreduce_256_slots_us"] = timed(lambda: check(lib.sobfu_hip_tiled_allreduce_max_u32(solver._h, C.c_void_p(slots.data_ptr()), C.c_size_t(SLOTS), st), "allreduce"), reps)
    pnp, psi = solver.new_local(2), solver.identity_psi()

    def loop(s):
        return lambda: s.iterate(pg, pn_full, pnp, psi, iters)

    if L.slab and not direct:
        prev = os.environ.get("SOBFU_TILED_SPLIT_A")
        for name, val in (("iteration_us_pass_a_unsplit", "0"), ("iteration_us_pass_a_split", "1")):
            os.environ["SOBFU_TILED_SPLIT_A"] = val
            out[name] = timed(loop(solver), 2) / iters
        if prev is None:
            os.environ.pop("SOBFU_TILED_SPLIT_A", None)
        else:
            os.environ["SOBFU_TILED_SPLIT_A"] = prev
        os.environ["SOBFU_TILED_SERIAL"] = "1"  # pass A, exchange, pass B in line on one stream (no overlap, no events)
        out["iteration_us_serial_schedule"] = timed(loop(solver), 2) / iters
        os.environ.pop("SOBFU_TILED_SERIAL", None)
    out["iteration_us_default_schedule"] = timed(loop(solver), 2) / iters
    lib.sobfu_hip_tiled_last_enqueue_us.restype = C.c_double
    out["host_enqueue_us_per_iteration"] = float(lib.sobfu_hip_tiled_last_enqueue_us(solver._h))
    dry = NativeTiledSolver(dims, dry=(world, rank), grid=grid, **kw)  # same tile, no peers: the compute side alone
    out["iteration_us_compute_only"] = timed(loop(dry), 2) / iters
    dry.close()
    return {k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}


def candidate_grids(world, dims):
    """every (Px, Py, Pz) with Px * Py * Pz == world whose tiles keep >= HALO cells per split axis, x split last (Px <= Py <= Pz
    is not required: 1 x 2 x 4 and 2 x 2 x 2 and 1 x 1 x 8 are all candidates at 8; permutations that split x more than z are not)"""
    out = []
    for px in range(1, world + 1):
        for py in range(1, world + 1):
            if world % (px * py):
                continue
            pz = world // (px * py)
            if px <= py <= pz and all(g == 1 or dims[a] // g >= HALO for a, g in enumerate((px, py, pz))):
                out.append((px, py, pz))
    return out


def make_native_solver(dims, grid, ranks, kw, transport):
    """NativeTiledSolver on the given transport ("direct" / "rccl"); ranks that SHARE a GPU (bring-up) cannot use RCCL -- their
    "rccl" is the same buffers over gloo (GlooTransport).  Returns (solver, keep-alive)."""
    if transport == "direct":
        return NativeTiledSolver(dims, grid=grid, transport="direct", **kw), None
    if ranks.share and ranks.world > 1:
        sv = NativeTiledSolver(dims, dry=(ranks.world, ranks.rank), grid=grid, **kw)
        tr = GlooTransport()
        sv.set_transport(tr.exchange, tr.allreduce)
        return sv, tr
    return NativeTiledSolver(dims, grid=grid, **kw), None


def direct_transport_precheck(P, ranks, kw, grid, iters=6):
    """The direct transport on THIS machine, before anything is timed: a few iterations of the bench workload on tiles against the
    single-GPU solver, bit for bit on every rank (peer mapping, flags and deadline all exercised).  Returns None when every rank
    agrees it works, else a reason string (collective: every rank gets the same verdict)."""
    from . import ops

    dims = P["dims"]
    c0, c1, r = (0.375,) * 3, (0.375 + 1.3 * float(P["vs"][0]), 0.375, 0.375), 0.2
    why, sv = None, None
    try:
        sv = NativeTiledSolver(dims, grid=grid, transport="direct", **kw)
    except Exception as e:  # noqa: BLE001
        why = f"setup failed: {e!r}"
    if ranks.min([0 if why else 1])[0] == 0:  # some rank could not map its peers: nobody uses the transport
        if sv is not None:
            sv.close()
        return why or "setup failed on another rank"
    try:
        pg_full, pn_full = ops.new_volume(dims), ops.new_volume(dims)
        ops.init_sphere(pg_full, P["vs"], P["trunc"], P["eta"], c0, r)
        ops.init_sphere(pn_full, P["vs"], P["trunc"], P["eta"], c1, r)
        L = sv.layout
        pg = L.take(pg_full).clone().contiguous()
        pnp, psi = sv.new_local(2), sv.identity_psi()
        done, norms = sv.iterate(pg, pn_full, pnp, psi, iters)
        one = ops.Solver(dims, max_iter=iters, **kw)
        psi_f, pnp_f = ops.new_field(dims), ops.new_volume(dims)
        ops.init_identity(psi_f)
        _, norms_one = one.iterate(pg_full, pn_full, pnp_f, psi_f, iters)
        one.close()
        same = (np.array_equal(np.asarray(norms_one, np.float32).view(np.uint32), np.asarray(norms, np.float32).view(np.uint32))
                and torch.equal(L.owned_global(psi_f)[..., :3].contiguous().view(torch.int32), L.owned(psi)[..., :3].contiguous().view(torch.int32))
                and torch.equal(L.owned_global(pnp_f).contiguous().view(torch.int32), L.owned(pnp).contiguous().view(torch.int32)))
        if not same:
            dn = int((np.asarray(norms_one, np.float32).view(np.uint32) != np.asarray(norms, np.float32).view(np.uint32)).sum())
            dp = int((L.owned_global(psi_f)[..., :3].contiguous().view(torch.int32) != L.owned(psi)[..., :3].contiguous().view(torch.int32)).sum())
            df = int((L.owned_global(pnp_f).contiguous().view(torch.int32) != L.owned(pnp).contiguous().view(torch.int32)).sum())
            why = (f"tiles differ from the single-GPU solve (rank {ranks.rank}: {dn} of {len(norms)} max-norms, {dp} psi words, {df} phi_n o psi words; "
                   f"iterations done {done}; norms {[float(v) for v in norms]} vs {[float(v) for v in norms_one]})")
    except Exception as e:  # noqa: BLE001 -- e.g. SOBFU_E_TIMEOUT: a peer's flag did not arrive
        why = f"{e!r}"
    ok = ranks.min([0 if why else 1])[0]
    sv.close()
    return None if ok else (why or "failed on another rank")


def direct_transport_sandbox(P, ranks, kw, grid, timeout=240):
    """The same check as direct_transport_precheck, one step earlier and somewhere safer: in a CHILD process of every rank
    (sobfu_amd/ipc_probe.py).  A transport that stores into other GPUs' memory from inside a kernel fails, when the mapping is not
    what it looks like, with a GPU memory fault -- which kills the process that launched the kernel.  The children take that risk;
    the ranks themselves only learn the verdict.  Returns None when every rank's child exited 0, else a reason (collective)."""
    import json
    import os
    import subprocess

    from ._lib import ROOT

    if os.environ.get("SOBFU_TILED_SANDBOX", "1") != "1":
        return None
    import socket

    port = 0
    if ranks.rank == 0:  # a port that is free right now, agreed on through the ranks' own process group
        with socket.socket() as sk:
            sk.bind(("", 0))
            port = sk.getsockname()[1]
    port = int(ranks.max([port])[0])
    args = dict(addr=os.environ.get("MASTER_ADDR", "127.0.0.1"), port=port, grid=list(grid),
                dims=list(P["dims"]), vs=[float(v) for v in P["vs"]], trunc=float(P["trunc"]), eta=float(P["eta"]), kw=kw, iters=4, timeout=int(os.environ.get("SOBFU_PROBE_TIMEOUT_S", "90")))
    # the children rendezvous among themselves: without the launcher's agent store (TORCHELASTIC_USE_AGENT_STORE would make rank 0's
    # child a client of a store nobody serves on that port)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["SOBFU_PROBE_ARGS"] = json.dumps(args)
    ok, why = False, None
    try:
        r = subprocess.run([sys.executable, "-m", "sobfu_amd.ipc_probe"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
        ok = r.returncode == 0
        if not ok:
            tail = " | ".join((r.stderr or r.stdout or "").strip().splitlines()[-3:])
            why = f"sandboxed probe exited {r.returncode}: {tail[-400:]}"
    except subprocess.TimeoutExpired:
        why = f"sandboxed probe did not finish in {timeout} s"
    except OSError as e:
        why = f"sandboxed probe could not start: {e!r}"
    all_ok = ranks.min([1 if ok else 0])[0] == 1
    if all_ok:
        time.sleep(float(os.environ.get("SOBFU_TILED_SETTLE_S", "0.5")))  # the children's device memory and IPC state are torn down asynchronously
    return None if all_ok else (why or "sandboxed probe failed on another rank")


def autotune_grid(P, ranks, kw, iters=40, transport="rccl"):
    """us per iteration of the native loop for every candidate tile grid on the machine at hand (MAX over ranks: all agree)"""
    from . import ops

    dims = P["dims"]
    c0, c1, r = (0.375,) * 3, (0.375 + 1.3 * float(P["vs"][0]), 0.375, 0.375), 0.2
    pg_full, pn_full = ops.new_volume(dims), ops.new_volume(dims)
    ops.init_sphere(pg_full, P["vs"], P["trunc"], P["eta"], c0, r)
    ops.init_sphere(pn_full, P["vs"], P["trunc"], P["eta"], c1, r)
    times = {}
    for grid in candidate_grids(ranks.world, dims):
        sv, _ = make_native_solver(dims, grid, ranks, kw, transport)
        pg = sv.layout.take(pg_full).clone().contiguous()
        pnp, psi = sv.new_local(2), sv.identity_psi()
        sv.iterate(pg, pn_full, pnp, psi, 4)
        torch.cuda.synchronize()
        ranks.barrier()
        t0 = time.perf_counter()
        sv.iterate(pg, pn_full, pnp, psi, iters)
        torch.cuda.synchronize()
        times[grid] = ranks.max([(time.perf_counter() - t0) / iters * 1e6])[0]
        sv.close()
    return times


def bench_tiled(args, P, ranks, timed_regions):
    """bench.py leg for --gpus N > 1: the SAME 256^3 solve cut into N tiles (strong scaling; 2 x 2 x 2 at N = 8).  The direct
    transport is the default; should it -- after passing its precheck -- still break during the run (a peer missing its deadline,
    or tiles that differ from the single-GPU solve in the final bitwise self-check), every rank repeats the whole leg on RCCL
    and the line says so: a wrong or wedged transport never becomes the reported number."""
    import os

    res = _bench_tiled_once(args, P, ranks, timed_regions, os.environ.get("SOBFU_TILED_TRANSPORT", "direct"))
    if res.get("retry"):
        why = res["retry"]
        print(f"[rank {ranks.rank}] direct transport abandoned ({why}): repeating the run on RCCL", file=sys.stderr, flush=True)
        res = _bench_tiled_once(args, P, ranks, timed_regions, "rccl", plain=True)
        res["transport_fallback"] = why
    return res


def _bench_tiled_once(args, P, ranks, timed_regions, want, plain=False):
    import os

    from . import ops

    rank, world = ranks.rank, ranks.world
    dims = P["dims"]
    X, Y, Z = dims
    spec = args.tiles or os.environ.get("SOBFU_TILES", "")
    c0, c1, r = (0.375,) * 3, (0.375 + 1.3 * float(P["vs"][0]), 0.375, 0.375), 0.2
    kw = dict(alpha=P["alpha"], w_reg=P["w_reg"], s=P["s"], lam=P["lam"], max_update_norm=P["max_update_norm"])
    K, W, R, PR = args.steps, args.warmup, args.repeats, args.profile_repeats
    total = W + (R + PR) * K
    native = os.environ.get("SOBFU_TILED_NATIVE", "1") == "1"
    if world == 1 and not dist.is_initialized():  # SOBFU_FORCE_TILED=1 on one GPU: a world of one
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", ranks.device))
    # transport of the native loop: "direct" (default: peer-mapped stores over xGMI, no RCCL in the loop) is taken only after
    # it has reproduced the single-GPU solve bit for bit on THIS machine on every rank (direct_transport_precheck, a few
    # iterations before anything is timed); otherwise every rank falls back to "rccl" and the line says why
    transport_name, fallback = ("rccl" if want != "direct" else "direct"), None
    if native and transport_name == "direct":
        probe_grid = parse_grid("" if spec == "auto" else spec, world)
        fallback = direct_transport_sandbox(P, ranks, kw, probe_grid)  # first in child processes (a GPU fault there costs nothing) ...
        if fallback is None:
            fallback = direct_transport_precheck(P, ranks, kw, probe_grid)  # ... then in this one
            if fallback is not None:
                # seen twice in ~35 multi-process start-ups right behind the children's exit (one refused export, one mismatch), never
                # in 360 start-ups without children: a second attempt, on fresh state, before the transport is given up
                print(f"[rank {rank}] direct transport precheck failed ({fallback}); trying once more", file=sys.stderr, flush=True)
                time.sleep(1.0)
                first, fallback = fallback, direct_transport_precheck(P, ranks, kw, probe_grid)
                if fallback is not None:
                    fallback = f"{fallback} (first attempt: {first})"
        if fallback is not None:
            print(f"[rank {rank}] direct transport not used: {fallback}", file=sys.stderr, flush=True)
            transport_name = "rccl"
    grid_times = None
    if spec == "auto":  # time every tile grid of `world` tiles on THIS machine (real exchange included) and keep the fastest
        grid_times = autotune_grid(P, ranks, kw, transport=transport_name)
        grid = min(grid_times, key=grid_times.get)
    else:
        grid = parse_grid(spec, world)
    # if ANY rank fails to set the native loop up, every rank falls back to the torch.distributed loop (same decomposition, same
    # results).  Ranks that share a GPU (bring-up) run "rccl" over the gloo transport.
    solver, transport = None, None
    if native:
        try:
            solver, transport = make_native_solver(dims, grid, ranks, kw, transport_name)
        except Exception as e:  # noqa: BLE001 -- report and agree on the fallback collectively
            print(f"[rank {rank}] native tiled loop unavailable: {e!r}", file=sys.stderr, flush=True)
        ok = ranks.min([1 if solver is not None else 0])[0]
        if ok == 0:
            if solver is not None:
                solver.close()
            solver, native = None, False
    if solver is None:
        solver = TiledSolver(dims, grid=grid, **kw)
    L = solver.layout
    # every rank builds the full analytic TSDFs (replicated phi_n; phi_global is then cut to the local tile)
    pg_full, pn_full = ops.new_volume(dims), ops.new_volume(dims)
    ops.init_sphere(pg_full, P["vs"], P["trunc"], P["eta"], c0, r)
    ops.init_sphere(pn_full, P["vs"], P["trunc"], P["eta"], c1, r)
    pg = L.take(pg_full).clone().contiguous()
    del pg_full
    pnp = solver.new_local(2)
    psi = solver.identity_psi()
    tuned = None
    # schedule autotuning (serial vs overlapped exchange; outside the timed region) runs only where RCCL was ASKED for: z-slabs by
    # default, 3-D tiles with SOBFU_TILED_AUTOTUNE=tiles.  When RCCL is the fallback of a direct transport that just failed, the run
    # takes the plainest schedule there is (serial, one stream) -- nothing that has never executed on >= 2 GPUs is tried first.
    at = os.environ.get("SOBFU_TILED_AUTOTUNE", "1")
    if (native and transport is None and transport_name == "rccl" and want == "rccl" and fallback is None and not plain and world > 1
            and ((L.slab and at == "1") or at == "tiles")):
        tuned = solver.autotune(pg, pn_full)
    def timed_native():
        solver.begin(pg, pn_full, pnp, psi, total)  # the solve is open and its state resident before anything is timed
        solver.step(W)
        secs = timed_regions(ranks, torch, lambda: solver.step(K), R)
        prof = None
        if PR > 0:  # the split of an iteration (pass A incl. the message stores / transfer + scatter / pass B), outside the timed regions
            sched = getattr(solver, "schedule", 0)
            slab_sched = L.slab and transport_name == "rccl"
            if slab_sched:
                solver.set_schedule(3)  # the split is measured on the serial schedule (same results whatever the schedule)
            solver.set_profiling(1, PR * K)
            solver.get_profile(reset=True)
            for _ in range(PR):
                solver.step(K)
            torch.cuda.synchronize()
            pa, px, pb, n = solver.get_profile()
            solver.set_profiling(0)
            if slab_sched:
                solver.set_schedule(sched)
            if n > 0:
                prof = ranks.max([pa / n, px / n, pb / n]) + [n]
        done, norms = solver.end()
        assert done == total and np.isfinite(norms).all() and float(norms.max()) > 0, (done, total)
        return secs, prof, norms

    if native:
        broke = None
        try:
            secs, prof, norms = timed_native()
        except Exception as e:  # noqa: BLE001
            if transport_name != "direct":
                raise
            broke = repr(e)  # e.g. SOBFU_E_TIMEOUT: a peer's flag did not arrive within the deadline
        if transport_name == "direct" and ranks.max([1 if broke else 0])[0] > 0:  # every rank leaves the transport together
            solver.close()
            return {"retry": broke or "another rank's direct transport broke"}
    else:  # the torch loop has no open-solve form: a region is a whole iterate() of K iterations
        prof, total = None, W + R * K
        if W > 0:
            solver.iterate(pg, pn_full, pnp, psi, W)
        hist = []
        secs = timed_regions(ranks, torch, lambda: hist.append(solver.iterate(pg, pn_full, pnp, psi, K)), R)
        norms = np.concatenate([h[1] for h in hist])
        assert all(h[0] == K for h in hist) and np.isfinite(norms).all()
    # self-check, outside the timed region: every rank repeats the WHOLE solve on its own GPU with the single-GPU solver
    # handle and compares its owned cells and the max-norm history bit for bit (tiling must not change a single bit)
    parity = None
    if os.environ.get("SOBFU_TILED_SELFCHECK", "1") == "1":
        pg_full = ops.new_volume(dims)
        ops.init_sphere(pg_full, P["vs"], P["trunc"], P["eta"], c0, r)
        psi_full, pnp_full = ops.new_field(dims), ops.new_volume(dims)
        ops.init_identity(psi_full)
        one = ops.Solver(dims, max_iter=max(total, 1), **kw)
        if native:
            _, norms_one = one.iterate(pg_full, pn_full, pnp_full, psi_full, total)
        else:
            parts = [one.iterate(pg_full, pn_full, pnp_full, psi_full, n)[1] for n in ([W] if W > 0 else []) + [K] * R]
            norms_one = np.concatenate(parts[(1 if W > 0 else 0):])
        one.close()
        same = (np.array_equal(np.asarray(norms_one, np.float32).view(np.uint32), np.asarray(norms, np.float32).view(np.uint32))
                and torch.equal(L.owned_global(psi_full)[..., :3].contiguous().view(torch.int32), L.owned(psi)[..., :3].contiguous().view(torch.int32))
                and torch.equal(L.owned_global(pnp_full).contiguous().view(torch.int32), L.owned(pnp).contiguous().view(torch.int32)))
        parity = bool(ranks.min([1 if same else 0])[0])
        if not parity:
            print(f"[rank {rank}] tiled self-check: tile differs from the single-GPU solve (local: {same})", file=sys.stderr, flush=True)
        del pg_full, psi_full, pnp_full
        if not parity and native and transport_name == "direct":
            solver.close()
            return {"retry": "tiles differed from the single-GPU solve in the final bitwise self-check"}
    # diagnostics for the next tuning round, outside the timed region (every rank takes part in the collective ones):
    # what one halo exchange, one slot all-reduce and the compute side alone cost on THIS machine
    diag, hung = None, False
    if native and os.environ.get("SOBFU_TILED_DIAG", "1") == "1":
        # in a worker thread with a deadline: whatever happens in there (an exception on one rank would leave the others
        # waiting in a collective), the benchmark line is still printed
        import threading

        box, dev_index = {}, torch.cuda.current_device()

        def work():
            try:
                torch.cuda.set_device(dev_index)
                box["diag"] = _tiled_diagnostics(solver, kw, dims, grid, pg, pn_full, world, rank, ranks)
            except Exception as e:  # noqa: BLE001
                box["error"] = repr(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("SOBFU_TILED_DIAG_TIMEOUT", "120")))
        hung = th.is_alive()
        diag = box.get("diag") or {"error": "timed out" if hung else box.get("error", "unknown")}
        if "error" in diag:
            print(f"[rank {rank}] tiled diagnostics: {diag['error']}", file=sys.stderr, flush=True)
    # whether ANY rank hung is agreed over the rendezvous store, not over the (possibly wedged) communicator: every rank then
    # takes the same exit path (no rank waits in a barrier the hung rank never reaches)
    hung_any = hung
    if world > 1 and native and os.environ.get("SOBFU_TILED_DIAG", "1") == "1":
        try:
            store = dist.distributed_c10d._get_default_store()
            store.set(f"sobfu_diag_hung_{rank}", "1" if hung else "0")
            hung_any = any(store.get(f"sobfu_diag_hung_{q}") == b"1" for q in range(world))
        except Exception as e:  # noqa: BLE001
            print(f"[rank {rank}] could not agree on the diagnostics verdict: {e!r}", file=sys.stderr, flush=True)
            hung_any = True
    own = tuple(L.g1[a] - L.g0[a] for a in range(3))
    what = (f"{world} z-slabs of {own[2]} planes" if L.slab else f"{grid[0]}x{grid[1]}x{grid[2]} tiles of {own[0]}x{own[1]}x{own[2]} cells")
    via = {"direct": "peer-mapped stores over xGMI issued by pass A's own launch (no pack / unpack, no RCCL in the loop; arrival flags "
                     "and max-norm rows travel the same way)",
           "rccl": "gloo (ranks share a GPU: bring-up transport)" if transport else "RCCL send/recv (packed by pass A's launch, one scatter kernel)"}
    return dict(diag_hung=hung, diag_hung_any=hung_any, transport=(transport_name if native else "torch.distributed"),
                transport_fallback=fallback, region_seconds=secs, N=X * Y * Z, ms_a=(prof[0] if prof else None), ms_b=(prof[2] if prof else None),
                ms_exchange=(prof[1] if prof else None), n_prof=(prof[3] if prof else None),
                launch_cells=max((l.g1[0] - l.g0[0]) * (l.g1[1] - l.g0[1]) * (l.g1[2] - l.g0[2]) for l in (TileLayout(dims, grid, q) for q in range(world))), last_norm=float(norms[-1]), workspace=None, tiled_parity=parity,
                tiled_diag=diag, tiles={"grid": list(grid), "owned_cells_rank0": list(own), "halo": HALO,
                                        "messages_per_exchange_rank0": len(L.messages())},
                parallelism=f"{what} (+{HALO}-cell halos), one nabla_U halo exchange per iteration over "
                            + (via[transport_name] if native else "RCCL send/recv") + ", "
                            + ("native C++ loop" if native else "torch.distributed loop")
                            + (f", schedule: {solver.SCHEDULES[solver.schedule]} (autotuned)" if tuned else ""),
                tiled_autotune_us=({solver.SCHEDULES[k]: round(v, 2) for k, v in tuned.items()} if tuned else None)
                if not grid_times else {"x".join(map(str, g)): round(v, 2) for g, v in grid_times.items()})
