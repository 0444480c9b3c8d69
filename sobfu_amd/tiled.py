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

This module is the PRODUCT side: the tile layout, the native loop's Python handle (NativeTiledSolver: the whole iteration runs in
C++, sobfu_amd/csrc/tiled_capi.hip), one frame on tiles (estimate_psi_tiled, TiledFusion).  The slow torch.distributed restatement of
the loop that the CPU tests drive with an oracle backend lives in tests/tiled_reference.py; everything bench.py needs around a
multi-GPU run (transport probing, grid timing, diagnostics) in bench_tiled.py.
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

    def owned_box_global(self):
        return (self.g0[0], self.g1[0], self.g0[1], self.g1[1], self.g0[2], self.g1[2])

    def window_box(self, w):
        """global box of the owned cells widened by w cells on every side, clipped to the volume"""
        return tuple(v for a in range(3) for v in (max(self.g0[a] - w, 0), min(self.g1[a] + w, self.dims[a])))

    def min_owned_extent(self):
        """smallest owned extent of ANY tile along an axis that is split (a halo wider than that reaches past the neighbours)"""
        return min([self.dims[a] // self.grid[a] for a in range(3) if self.grid[a] > 1] or [max(self.dims)])

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
            out.append((n[0] + self.grid[0] * (n[1] + self.grid[1] * n[2]), tuple(sb), tuple(rb), d[0] == 0 and d[1] == 0))
        # the z faces go last (the library's order: on the packed transports of a 3-D tile they travel in place, as whole planes)
        return [m[:3] for m in sorted(out, key=lambda m: m[3])]


class SlabLayout(TileLayout):
    """z-slabs: the 1 x 1 x world tile grid"""

    def __init__(self, dims, world, rank, halo=HALO):
        if int(dims[2]) < world * halo:
            raise ValueError(f"Z={dims[2]} is too thin for {world} slabs with halo {halo}")
        super().__init__(dims, (1, 1, world), rank, halo)
        if world > 1 and (self.z1 - self.z0) < halo:
            raise ValueError("a slab must own at least `halo` planes")


def _sqrt_rd(m_bits: int) -> float:
    m = np.array([m_bits], np.uint32).view(np.float32)[0]
    r = np.sqrt(m, dtype=np.float32)
    if r > 0 and np.float64(r) * np.float64(r) > np.float64(m):
        r = np.nextafter(r, np.float32(-np.inf), dtype=np.float32)
    return float(r)


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


def _box_and(a, b):
    r = tuple(v for k in range(3) for v in (max(a[2 * k], b[2 * k]), min(a[2 * k + 1], b[2 * k + 1])))
    return r if all(r[2 * k] < r[2 * k + 1] for k in range(3)) else None


def window_plan(layout, w):
    """Who sends what so that every rank holds its WINDOW (owned cells widened by w, clipped to the volume) of a field:
    (window box, recvs, sends) in GLOBAL cell coordinates -- recvs = [(rank q, owned(q) & window(me))], sends = [(q, owned(me) &
    window(q))], q != me.  Owned boxes are disjoint and a window is a box, so a pair of ranks exchanges at most one box each way; with
    w below the tiles' extents these are the 26 face / edge / corner neighbours, but nothing here assumes that."""
    me = layout
    lays = [TileLayout(me.dims, me.grid, q, me.halo) for q in range(me.world)]
    wb = me.window_box(w)
    recvs = [(q, b) for q, l in enumerate(lays) if q != me.rank for b in [_box_and(l.owned_box_global(), wb)] if b]
    sends = [(q, b) for q, l in enumerate(lays) if q != me.rank for b in [_box_and(me.owned_box_global(), l.window_box(w))] if b]
    return wb, recvs, sends


def _cut(t, box, origin):
    """view of tensor t (z, y, x, ...) on global box `box`, t's cell (0, 0, 0) being global cell `origin`"""
    return t[box[4] - origin[2]:box[5] - origin[2], box[2] - origin[1]:box[3] - origin[1], box[0] - origin[0]:box[1] - origin[0]]


class DistHalo:
    """What the per-frame tail needs from the other ranks, over torch.distributed: one MAX reduction of a float and the window of a
    field (point-to-point: every rank receives exactly the cells of its window it does not own -- a few MB at 256^3 where the
    all-gather moved the whole volume to everybody).  via_host: stage through host memory (ranks that share a GPU run on gloo)."""

    def __init__(self, layout, group=None, via_host=None):
        self.L, self.group = layout, group
        self.via_host = (dist.get_backend(group) == "gloo") if via_host is None else via_host
        self.bytes_received = 0

    def _peer(self, q):
        return q if self.group is None else dist.get_global_rank(self.group, q)

    def allreduce_max(self, value):
        if self.L.world == 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device="cpu" if self.via_host else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return float(t.item())

    def window(self, local, w, nch):
        """-> (window tensor with local's channel count, window box).  nch: leading channels that travel (psi: 3 of 4, w stays 0)"""
        L = self.L
        wb, recvs, sends = window_plan(L, w)
        win = torch.zeros((wb[5] - wb[4], wb[3] - wb[2], wb[1] - wb[0]) + tuple(local.shape[3:]), dtype=local.dtype, device=local.device)
        origin = (wb[0], wb[2], wb[4])
        _cut(win, L.owned_box_global(), origin).copy_(L.owned(local))
        dev = "cpu" if self.via_host else local.device
        rbuf = [torch.empty((b[5] - b[4], b[3] - b[2], b[1] - b[0], nch), dtype=local.dtype, device=dev) for _, b in recvs]
        sbuf = [_cut(local, b, L.base)[..., :nch].contiguous().to(dev) for _, b in sends]
        if not self.via_host:
            torch.cuda.current_stream().synchronize()  # the send buffers are final before the communication stream reads them
        ops = [dist.P2POp(dist.irecv, t, self._peer(q), self.group) for (q, _), t in zip(recvs, rbuf)]
        ops += [dist.P2POp(dist.isend, t, self._peer(q), self.group) for (q, _), t in zip(sends, sbuf)]
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        for (q, b), t in zip(recvs, rbuf):
            _cut(win, b, origin)[..., :nch].copy_(t.to(local.device))
            self.bytes_received += t.numel() * t.element_size()
        return win, wb


def estimate_psi_tiled(solver, phi_global_local, phi_global_psi_inv_local, phi_n_full, phi_n_psi_local, psi_local, psi_inv_local,
                       n_iters, inverse_iters=48, gather=None, halo=None):
    """One frame's Solver::estimate_psi (src/sobfu/cuda/solver.cu:85-205) on tiles: the tiled iteration loop, then the per-frame
    tail (solver.cu:196-199): psi^-1 by 48 fixed-point sweeps from the identity, and phi_global o psi^-1.

    The tail gathers psi at psi^-1(x) and phi_global at psi^-1(x) -- points within r = max |psi - id| (over the whole volume) of x
    (src/sobfu/cuda/vector_fields.cu:111-138, include/sobfu/cuda/utils.hpp:124-164).  So, after ONE global MAX reduction, every rank
    fetches psi and phi_global on its owned cells widened by ceil(r) + 2 cells (`halo`: DistHalo over torch.distributed, or a test
    double) and runs the windowed tile kernels (sobfu_hip_tile3_{estimate_inverse,apply}_window): a few MB per frame instead of the
    two all-gathers of SURVEY 8(e) (psi 268 MB + phi_global 134 MB at 256^3, which do not shrink with N).  Same clamps on the global
    extents, same arithmetic: bit-identical.  When the window would reach past the neighbours (r larger than the smallest tile), or a
    sample is reported outside its window, the tail runs on all-gathered sources as before (`gather`).  Passing `gather` WITHOUT
    `halo` forces that path (tests).  HIP backend only.  Returns (iterations, per-iteration max norms); solver.tail_stats says which
    path ran and what it moved."""
    done, norms = solver.iterate(phi_global_local, phi_n_full, phi_n_psi_local, psi_local, n_iters)
    frame_tail(solver, phi_global_local, phi_global_psi_inv_local, psi_local, psi_inv_local, inverse_iters, gather, halo)
    return done, norms


def frame_tail(solver, phi_global_local, phi_global_psi_inv_local, psi_local, psi_inv_local, inverse_iters=48, gather=None, halo=None):
    """the tail of estimate_psi_tiled on its own: psi^-1 and phi_global o psi^-1 on the owned cells (see there); fills solver.tail_stats"""
    from . import _lib

    L = solver.layout
    lib, st = _lib.lib(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if halo is None and gather is None and L.world > 1:
        halo = solver.halo_comm() if hasattr(solver, "halo_comm") else DistHalo(L, getattr(solver, "group", None))
    gather = gather or solver.gather_owned
    I6 = C.c_int * 6
    own = I6(*L.own_box())
    stats = {"mode": "all-gather", "reach": None, "halo_width": None, "bytes_received": 0}
    if halo is not None:
        bits = torch.zeros(1, dtype=torch.int32, device=psi_local.device)
        _lib.check(lib.sobfu_hip_tile3_max_displacement(C.c_void_p(psi_local.data_ptr()), *L.L, *L.base, own, C.c_void_p(bits.data_ptr()), st),
                   "tile3_max_displacement")
        r = halo.allreduce_max(float(np.array([bits.item()], np.int32).view(np.float32)[0]))
        w = int(np.ceil(r)) + 2 if r < 1e30 else 1 << 30
        stats["reach"] = r
        if w <= L.min_owned_extent():
            got0 = getattr(halo, "bytes_received", 0)
            viol = torch.zeros(1, dtype=torch.int32, device=psi_local.device)
            psi_win, wb = halo.window(psi_local, w, 3)
            win6 = I6(wb[1] - wb[0], wb[3] - wb[2], wb[5] - wb[4], wb[0], wb[2], wb[4])
            _lib.check(lib.sobfu_hip_tile3_init_identity(C.c_void_p(psi_inv_local.data_ptr()), *L.L, *L.base, st), "tile3_init_identity")
            _lib.check(lib.sobfu_hip_tile3_estimate_inverse_window(C.c_void_p(psi_win.data_ptr()), win6, *L.dims, C.c_void_p(psi_inv_local.data_ptr()),
                                                                   *L.L, *L.base, own, C.c_int(inverse_iters), C.c_void_p(viol.data_ptr()), st),
                       "tile3_estimate_inverse_window")
            pg_win, wb2 = halo.window(phi_global_local, w, 2)
            assert wb2 == wb
            _lib.check(lib.sobfu_hip_tile3_apply_window(C.c_void_p(pg_win.data_ptr()), win6, *L.dims, C.c_void_p(phi_global_psi_inv_local.data_ptr()),
                                                        C.c_void_p(psi_inv_local.data_ptr()), *L.L, own, C.c_void_p(viol.data_ptr()), st),
                       "tile3_apply_window")
            bad = halo.allreduce_max(float(viol.item()))  # (synchronises: the windows die with this frame) -- every rank takes the same path
            stats.update(mode="halo", halo_width=w, bytes_received=getattr(halo, "bytes_received", 0) - got0)
            if bad == 0.0:
                solver.tail_stats = stats
                return
            stats["mode"] = "all-gather (a sample left its window: the reach bound did not hold)"
    psi_full = gather(psi_local)                                                              # solver.cu:196-197
    _lib.check(lib.sobfu_hip_tile3_init_identity(C.c_void_p(psi_inv_local.data_ptr()), *L.L, *L.base, st), "tile3_init_identity")
    _lib.check(lib.sobfu_hip_tile3_estimate_inverse(C.c_void_p(psi_full.data_ptr()), *L.dims, C.c_void_p(psi_inv_local.data_ptr()), *L.L,
                                                    *L.base, C.c_int(inverse_iters), st), "tile3_estimate_inverse")
    pg_full = gather(phi_global_local)                                                        # solver.cu:199
    _lib.check(lib.sobfu_hip_tile3_apply(C.c_void_p(pg_full.data_ptr()), *L.dims, C.c_void_p(phi_global_psi_inv_local.data_ptr()),
                                         C.c_void_p(psi_inv_local.data_ptr()), *L.L, st), "tile3_apply")
    torch.cuda.current_stream().synchronize()  # psi_full / pg_full die with this frame
    X, Y, Z = L.dims
    own_cells = (L.g1[0] - L.g0[0]) * (L.g1[1] - L.g0[1]) * (L.g1[2] - L.g0[2])
    stats["bytes_received"] += (X * Y * Z - own_cells) * (16 + 8) if L.world > 1 else 0
    solver.tail_stats = stats


class TiledFusion:
    """SobFusion::operator() (src/sobfu/sob_fusion.cpp:71-145) with the volume cut into tiles: every rank runs the depth
    pre-steps (640 x 480: negligible), integrates ITS tile of phi_global and the whole phi_n (the warp gathers anywhere), the
    solve runs tiled, the fusion is per cell (run on the whole local array: only owned cells are ever read back).
    phi_global / phi_n o psi / psi / psi^-1 live as local arrays.

    params: dims, size (metres), trunc, eta (metres), max_weight, intr (fx, fy, cx, cy), R, t (volume -> camera), start_frame,
    bilateral (ksz, sigma_spatial, sigma_depth), trunc_depth, max_iter; `solver` is a TiledSolver / NativeTiledSolver."""

    def __init__(self, solver, params, gather=None, halo=None):
        from . import ops

        self.ops, self.solver, self.P, self.gather, self.halo = ops, solver, params, gather, halo
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
                                    P["max_iter"], gather=self.gather, halo=self.halo)  # :141
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

    def messages_inplace(self):
        """-> (the in-place message list of one exchange [(peer, send_off, recv_off, count) in floats of the nabla_U array], n_packed):
        what a pluggable transport sees in the SECOND call of a 3-D tile's exchange (include/sobfu_hip.h, sobfu_hip_tiled_messages_inplace)"""
        mm, n_packed = (TiledMsg * 18)(), C.c_int(0)
        n = self._lib.lib().sobfu_hip_tiled_messages_inplace(self._h, mm, 18, C.byref(n_packed))
        return [(mm[i].peer, mm[i].send_off, mm[i].recv_off, mm[i].count) for i in range(n)], n_packed.value

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
            mine.append(int(torch.cuda.current_device()))
        except Exception as ex:  # noqa: BLE001
            err = f"rank {self.rank}: export: {ex!r}"
        allh = [None] * self.world
        dist.all_gather_object(allh, (mine, err), group=self.group)
        errs = [a[1] for a in allh if a[1]]
        if not errs:
            try:
                # can this GPU reach every rank it stores to at all?  (asked BEFORE anything is mapped: a missing peer path fails here,
                # with its reason, and the run takes RCCL -- not later with a memory fault inside a kernel)
                me = int(torch.cuda.current_device())
                self.peer_links = {}
                for q in range(self.world):
                    if q == self.rank:
                        continue
                    info = (C.c_int * 4)()
                    check(lib.sobfu_hip_p2p_info(C.c_int(me), C.c_int(int(allh[q][0][3])), info), "p2p_info")
                    self.peer_links[q] = dict(device=int(allh[q][0][3]), can_access=info[0], link_type=info[1], hops=info[2], perf_rank=info[3])
                    if info[0] == 0:
                        raise RuntimeError(f"device {me} cannot access device {allh[q][0][3]} of rank {q} as a peer "
                                           f"(hipDeviceCanAccessPeer = 0, link type {info[1]}, hops {info[2]})")
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

    def max_iterations(self):
        """iterations one solve may run on this handle (the direct transport's peer-mapped max-norm rows are fixed at creation)"""
        return int(self._lib.lib().sobfu_hip_tiled_max_iterations(self._h))

    # diagnostics of the direct transport (collective: every rank makes the same calls in the same order, outside a solve)
    def wait_stats(self, reset=True):
        """(microseconds pass A's signalling workgroup has spent waiting for the peers' arrival flags, number of waits)"""
        us, n = C.c_double(), C.c_int()
        self._lib.check(self._lib.lib().sobfu_hip_tiled_wait_stats(self._h, C.byref(us), C.byref(n), C.c_int(1 if reset else 0)), "tiled_wait_stats")
        return us.value, n.value

    def pingpong(self, rank_a, rank_b, reps):
        self._lib.check(self._lib.lib().sobfu_hip_tiled_pingpong(self._h, C.c_int(rank_a), C.c_int(rank_b), C.c_int(reps),
                                                                 C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tiled_pingpong")

    def probe_push(self, reps):
        self._lib.check(self._lib.lib().sobfu_hip_tiled_probe_push(self._h, C.c_int(reps), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
                        "tiled_probe_push")

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

    tail_stats = None  # what the last frame's tail did (estimate_psi_tiled)

    def halo_comm(self):
        """the per-frame tail's communication over this handle's process group (kept: it counts the bytes it has received)"""
        if getattr(self, "_halo", None) is None:
            self._halo = DistHalo(self.layout, self.group)
        return self._halo

    def estimate_psi(self, *args, **kw):
        return estimate_psi_tiled(self, *args, **kw)

    # z-slabs on the RCCL transport: 1 / 2 = exchange overlapped, pass A split into boundary + interior launches or whole; 3-D tiles and
    # the direct transport have ONE way of issuing an iteration
    SCHEDULES = {1: "overlapped exchange, pass A split", 2: "overlapped exchange, pass A whole", 3: "serial (no overlap, no events)"}

    def set_schedule(self, schedule):
        self._lib.check(self._lib.lib().sobfu_hip_tiled_set_schedule(self._h, C.c_int(int(schedule))), "tiled_set_schedule")
        self.schedule = int(schedule)

    def autotune(self, phi_global_local, phi_n_full, iters=40):
        """Times the three ways of issuing an iteration on THIS machine (they give identical results) on scratch state and keeps
        the fastest; every rank takes part and all agree (MAX over ranks).  Returns {schedule: us per iteration}."""
        pnp, psi = self.new_local(2), self.identity_psi()
        times = {}
        for sched in (self.SCHEDULES if self.layout.slab else (3,)):
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
