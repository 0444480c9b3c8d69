"""Multi-GPU solver loop: the volume is cut into z-slabs, one rank (one process, one GPU) per slab.

SURVEY.md section 8(e).  Frames of a sequence are sequentially dependent (psi persists), so the path shards by
VOLUME TILE.  Slabs along z (z is the slowest-varying axis of the layout) make every halo a contiguous block of
planes: halo exchange is zero-copy `isend/irecv` straight out of / into the field arrays -- no pack kernels.

Per rank:  psi, phi_n o psi, phi_global, nabla_U are LOCAL slabs (X, Y, Lz) = owned planes + HALO (=4) planes towards
each neighbour; phi_n is replicated (the warp gathers at absolute coordinates anywhere in the volume).

One iteration = ONE exchange (Option B of the survey, which in a 1-D decomposition needs no edge data), overlapped
with the interior compute (the kernels take the range of planes a launch produces):
    A_bnd  nabla_U on the 4 owned planes next to each interior face (reads psi, phi_n o psi at owned +-1)
    E      start the exchange of those planes (one grouped RCCL send/recv on RCCL's stream) -> nabla_U exact on owned +-4
    A_int  nabla_U on the remaining owned planes                       } run while E is in flight: they touch
    B_int  psi, phi_n o psi on the planes whose +-3 taps are all owned } neither the planes being sent nor received
    wait E
    B_bnd  psi, phi_n o psi on the remaining planes out to owned +-1: the radius-3 convolution is exact there, so psi
           and phi_n o psi stay exact on owned +-1 -- all the next pass A reads -- without ever being exchanged
           (invariant; identity psi satisfies it at the start).  max ||u||^2 takes OWNED planes only.
    R      all_reduce(MAX) of the 256 max-norm slots -- only when max_update_norm >= 0 (otherwise the test never fires)
Halo planes further out are never read.  Clamp / mirror rules act at a slab's array edge, which is the volume
boundary exactly where the slab has no halo, so the result equals the single-GPU run bit for bit (the max is
order-independent).

The kernel backend is pluggable: `HipBackend` (product; C ABI on torch CUDA tensors over RCCL) -- the CPU tests inject an
oracle-backed backend over gloo to check the decomposition logic without a GPU.
"""
from __future__ import annotations

import ctypes as C
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

HALO = 4
SLOTS = 256


class SlabLayout:
    """Which z planes a rank owns and how its local slab is laid out."""

    def __init__(self, dims, world, rank, halo=HALO):
        X, Y, Z = (int(d) for d in dims)
        if Z < world * halo:
            raise ValueError(f"Z={Z} is too thin for {world} slabs with halo {halo}")
        base, rem = divmod(Z, world)
        starts = [r * base + min(r, rem) for r in range(world + 1)]
        self.dims, self.world, self.rank, self.halo = (X, Y, Z), world, rank, halo
        self.z0, self.z1 = starts[rank], starts[rank + 1]
        self.lo = halo if rank > 0 else 0            # halo planes below
        self.hi = halo if rank < world - 1 else 0    # halo planes above
        self.Lz = (self.z1 - self.z0) + self.lo + self.hi
        self.own_lo, self.own_hi = self.lo, self.lo + (self.z1 - self.z0)
        self.zbase = self.z0 - self.lo               # global z of local plane 0
        if world > 1 and (self.z1 - self.z0) < halo:
            raise ValueError("a slab must own at least `halo` planes")

    def local_shape(self, channels):
        X, Y, _ = self.dims
        return (self.Lz, Y, X, channels)

    def take(self, full):
        """local slab (with halos) cut out of a full-volume array/tensor"""
        return full[self.zbase:self.zbase + self.Lz]

    def owned(self, local):
        return local[self.own_lo:self.own_hi]


def halo_ops(layout: SlabLayout, fields, group=None):
    """P2P op list of one exchange (built once per solve: the views stay valid while the buffers live)."""
    L = layout
    ops = []
    for t, w in fields:
        assert t.shape[0] == L.Lz and t.is_contiguous() and 0 < w <= L.halo
        if L.rank > 0:
            ops.append(dist.P2POp(dist.isend, t[L.own_lo:L.own_lo + w], L.rank - 1, group))
            ops.append(dist.P2POp(dist.irecv, t[L.own_lo - w:L.own_lo], L.rank - 1, group))
        if L.rank < L.world - 1:
            ops.append(dist.P2POp(dist.isend, t[L.own_hi - w:L.own_hi], L.rank + 1, group))
            ops.append(dist.P2POp(dist.irecv, t[L.own_hi:L.own_hi + w], L.rank + 1, group))
    return ops


def start_halo_ops(ops):
    """One grouped RCCL launch on RCCL's own stream, ordered after everything queued so far on the current stream."""
    return dist.batch_isend_irecv(ops) if ops else []


def finish_halo_ops(works):
    for w in works:  # for RCCL this only makes the current stream wait; the host does not block
        w.wait()


def run_halo_ops(ops):
    finish_halo_ops(start_halo_ops(ops))


def exchange_halos(layout: SlabLayout, fields, group=None):
    """fields: list of (tensor (Lz, ...), radius).  Zero-copy neighbour exchange of `radius` owned planes."""
    run_halo_ops(halo_ops(layout, fields, group))


class _SlabState:
    """What a backend keeps for one solve: nabla_U (exchanged by the driver) + whatever format it iterates in."""

    def __init__(self, layout, nabla_U):
        self.layout, self.nabla_U = layout, nabla_U


class HipBackend:
    """Per-slab kernels through the C ABI (include/sobfu_hip.h `sobfu_hip_tile_*`).

    Iterates in the compact format (12-byte psi / nabla_U, tsdf-only phi_global / phi_n / phi_n o psi -- fewer bytes both
    through HBM and over xGMI); `begin` converts the caller's API-format slabs, `end` rebuilds them."""

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
        X, Y, _ = layout.dims
        self._call("sobfu_hip_tile_init_identity", self._p(psi), X, Y, layout.Lz, layout.zbase)

    def _buf(self, key, shape):
        t = self._cache.get(key)
        if t is None or tuple(t.shape) != tuple(shape):
            t = torch.empty(shape, dtype=torch.float32, device="cuda")
            self._cache[key] = t
        return t

    def begin(self, layout, pg_local, pn_full, pnp_local, psi_local):
        X, Y, Z = layout.dims
        Lz = layout.Lz
        st = _SlabState(layout, None)
        st.pn_full, st.pnp, st.psi = pn_full, pnp_local, psi_local
        if not self.compact:
            st.nabla_U = self._buf("nU4", (Lz, Y, X, 4))
            st.c_psi, st.c_f, st.c_g, st.c_n = psi_local, pnp_local, pg_local, pn_full
            self._call("sobfu_hip_tile_apply", self._p(pn_full), Z, self._p(pnp_local), self._p(psi_local), X, Y, Lz)
            return st
        st.nabla_U = self._buf("nU3", (Lz, Y, X, 3))
        st.c_psi, st.c_f, st.c_g = self._buf("psi3", (Lz, Y, X, 3)), self._buf("f", (Lz, Y, X)), self._buf("g", (Lz, Y, X))
        st.c_n = self._buf("n", (Z, Y, X))
        nl, nf = C.c_size_t(Lz * Y * X), C.c_size_t(Z * Y * X)
        self._call("sobfu_hip_pack_vec3", self._p(psi_local), self._p(st.c_psi), nl)
        self._call("sobfu_hip_extract_tsdf", self._p(pg_local), self._p(st.c_g), nl)
        self._call("sobfu_hip_extract_tsdf", self._p(pn_full), self._p(st.c_n), nf)
        self._call("sobfu_hip_tile_apply_tsdf_only", self._p(st.c_n), Z, self._p(st.c_f), self._p(st.c_psi), X, Y, Lz)  # solver.cu:106
        return st

    def pass_a(self, st, z0, z1, w_reg, prev_slots, thr):
        """nabla_U on local planes [z0, z1)"""
        if z1 <= z0:
            return
        L = st.layout
        X, Y, _ = L.dims
        prev = self._p(prev_slots) if prev_slots is not None else None
        self._call("sobfu_hip_tile_potential_gradient", self._p(st.c_f), self._p(st.c_g), self._p(st.c_psi), self._p(st.nabla_U),
                   C.c_float(w_reg), X, Y, L.Lz, z0, z1, prev, C.c_float(thr), 1 if self.compact else 0)

    def pass_b(self, st, z0, z1, slots, taps, alpha, prev_slots, thr):
        """psi update + warp on local planes [z0, z1)"""
        if z1 <= z0:
            return
        L = st.layout
        X, Y, Z = L.dims
        prev = self._p(prev_slots) if prev_slots is not None else None
        self._call("sobfu_hip_tile_smooth_update_apply", self._p(st.nabla_U), self._p(st.c_psi), self._p(st.c_n), self._p(st.c_f), None,
                   self._p(slots), (C.c_float * 7)(*[float(v) for v in taps[:7]]), C.c_float(alpha), X, Y, L.Lz, Z, L.own_lo, L.own_hi,
                   z0, z1, prev, C.c_float(thr), 1 if self.compact else 0)

    def end(self, st):
        if not self.compact:
            return
        L = st.layout
        X, Y, Z = L.dims
        self._call("sobfu_hip_unpack_vec3", self._p(st.c_psi), self._p(st.psi), C.c_size_t(L.Lz * Y * X))
        self._call("sobfu_hip_tile_apply", self._p(st.pn_full), Z, self._p(st.pnp), self._p(st.psi), X, Y, L.Lz)  # state of solver.cu:168

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
    """The gradient-descent loop of sobfu::device::estimate_psi (reference src/sobfu/cuda/solver.cu:106-193) on slabs."""

    def __init__(self, dims, *, alpha, w_reg, s=7, lam=0.1, max_update_norm=-1.0, backend=None, group=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.group = group
        self.backend = backend or HipBackend()
        self.layout = SlabLayout(dims, self.world, self.rank)
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
        xch = halo_ops(L, [(st.nabla_U, HALO)], self.group) if self.world > 1 else []
        lo, hi, H = L.own_lo, L.own_hi, HALO
        # planes next to an interior face (sent to the neighbour) vs the rest; ranges are local plane indices
        a_lo = min(lo + H, hi) if L.lo else lo          # [lo, a_lo)  : lower boundary planes of pass A
        a_hi = max(hi - H, a_lo) if L.hi else hi        # [a_hi, hi)  : upper boundary planes of pass A
        b_lo = min(lo + 3, hi) if L.lo else lo          # pass B planes >= b_lo have all -3 taps inside the owned range
        b_hi = max(hi - 3, b_lo) if L.hi else hi
        b_first = lo - 1 if L.lo else lo                # pass B also refreshes the first halo plane (owned +-1)
        b_last = hi + 1 if L.hi else hi
        for it in range(1, n_iters + 1):
            prev = slots[it - 1] if (it > 1 and can_converge) else None
            row = slots[it]
            be.pass_a(st, lo, a_lo, self.w_reg, prev, self.thr)
            be.pass_a(st, a_hi, hi, self.w_reg, prev, self.thr)
            works = start_halo_ops(xch)
            be.pass_a(st, a_lo, a_hi, self.w_reg, prev, self.thr)
            be.pass_b(st, b_lo, b_hi, row, self.taps, self.alpha, prev, self.thr)
            finish_halo_ops(works)
            be.pass_b(st, b_first, b_lo, row, self.taps, self.alpha, prev, self.thr)
            be.pass_b(st, b_hi, b_last, row, self.taps, self.alpha, prev, self.thr)
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
        """all_gather of the owned planes -> full volume on every rank (z-concatenation)."""
        own = self.layout.owned(local).contiguous()
        if self.world == 1:
            return own
        base, rem = divmod(self.layout.dims[2], self.world)
        parts = [torch.empty((base + (1 if r < rem else 0),) + tuple(own.shape[1:]), dtype=own.dtype, device=own.device)
                 for r in range(self.world)]
        dist.all_gather(parts, own, group=self.group)
        return torch.cat(parts, dim=0)


def estimate_psi_tiled(solver, phi_global_local, phi_global_psi_inv_local, phi_n_full, phi_n_psi_local, psi_local, psi_inv_local,
                       n_iters, inverse_iters=48, gather=None):
    """One frame's Solver::estimate_psi (src/sobfu/cuda/solver.cu:85-205) on slabs: the tiled iteration loop, then the
    two per-frame collectives of SURVEY 8(e) -- all-gather psi for the 48-sweep inverse (it gathers psi at arbitrary
    psi^-1(x)), all-gather phi_global for the canonical -> live warp -- each followed by the slab kernel.  HIP backend only
    (C ABI: sobfu_hip_tile_init_identity / tile_estimate_inverse / tile_apply).  `gather` overrides solver.gather_owned
    (tests).  Returns (iterations, per-iteration max norms)."""
    from . import _lib

    L = solver.layout
    X, Y, Z = L.dims
    lib, st = _lib.lib(), C.c_void_p(torch.cuda.current_stream().cuda_stream)
    gather = gather or solver.gather_owned
    done, norms = solver.iterate(phi_global_local, phi_n_full, phi_n_psi_local, psi_local, n_iters)
    psi_full = gather(psi_local)                                                              # solver.cu:196-197
    _lib.check(lib.sobfu_hip_tile_init_identity(C.c_void_p(psi_inv_local.data_ptr()), X, Y, L.Lz, L.zbase, st), "tile_init_identity")
    _lib.check(lib.sobfu_hip_tile_estimate_inverse(C.c_void_p(psi_full.data_ptr()), Z, C.c_void_p(psi_inv_local.data_ptr()), X, Y, L.Lz,
                                                   L.zbase, C.c_int(inverse_iters), st), "tile_estimate_inverse")
    pg_full = gather(phi_global_local)                                                        # solver.cu:199
    _lib.check(lib.sobfu_hip_tile_apply(C.c_void_p(pg_full.data_ptr()), Z, C.c_void_p(phi_global_psi_inv_local.data_ptr()),
                                        C.c_void_p(psi_inv_local.data_ptr()), X, Y, L.Lz, st), "tile_apply")
    torch.cuda.current_stream().synchronize()  # psi_full / pg_full die with this frame
    return done, norms


class TiledFusion:
    """SobFusion::operator() (src/sobfu/sob_fusion.cpp:71-145) with the volume cut into z-slabs: every rank runs the depth
    pre-steps (640 x 480: negligible), integrates ITS slab of phi_global and the whole phi_n (the warp gathers anywhere), the
    solve runs tiled, the fusion touches owned planes only.  phi_global / phi_n o psi / psi / psi^-1 live as local slabs.

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
            ops.tile_integrate_depth(dists, self.phi_global, L.zbase, self.vs, P["trunc"], P["eta"], P["R"], P["t"], P["intr"])
            self.phi_n = ops.new_volume(L.dims)
            self.phi_n_psi, self.phi_global_psi_inv = s.new_local(2), s.new_local(2)
            self.psi, self.psi_inv = s.identity_psi(), s.identity_psi()
            self.frame += 1
            return None
        ops.clear_volume(self.phi_n)                                                 # :129
        ops.integrate_depth(dists, self.phi_n, self.vs, P["trunc"], P["eta"], P["R"], P["t"], P["intr"])  # :130
        own = slice(L.own_lo, L.own_hi)
        result = None
        if self.frame < P["start_frame"]:                                            # :136-139
            ops.integrate_fuse(self.phi_global[own], L.owned(L.take(self.phi_n)).contiguous(), P["max_weight"])
        else:
            result = s.estimate_psi(self.phi_global, self.phi_global_psi_inv, self.phi_n, self.phi_n_psi, self.psi, self.psi_inv,
                                    P["max_iter"], gather=self.gather)               # :141
            ops.integrate_fuse(self.phi_global[own], self.phi_n_psi[own], P["max_weight"])  # :142 (owned planes)
        self.frame += 1
        return result


class NativeTiledSolver:
    """The same slab loop run entirely in C++ (sobfu_amd/csrc/tiled_capi.hip): RCCL send/recv issued from the library on
    a dedicated communication stream, overlapped with the interior compute, no Python per iteration.  torch.distributed is
    used once, to hand the RCCL unique id to every rank."""

    def __init__(self, dims, *, alpha, w_reg, s=7, lam=0.1, max_update_norm=-1.0, group=None, dry=None):
        """dry=(world, rank): a communicator-less handle with that slab layout -- every launch and stream dependency of the
        rank's schedule without peers (halos are stale, so only its TIMING means anything; tools/slab_time_native.py)."""
        import os

        from . import _lib
        from ._lib import SolverParams

        self._lib = _lib
        L = _lib.lib()
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        _lib.check(L.sobfu_hip_tiled_load_rccl(path.encode()), "tiled_load_rccl")
        uid = (C.c_char * 128)()
        if dry is not None:
            self.world, self.rank = dry
        elif self.rank == 0:
            _lib.check(L.sobfu_hip_tiled_unique_id(uid), "tiled_unique_id")
        if self.world > 1 and dry is None:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        self.params = SolverParams(0, 0, s, max_update_norm, np.float32(lam), alpha, w_reg)
        self._h = C.c_void_p()
        X, Y, Z = (int(d) for d in dims)
        _lib.check(L.sobfu_hip_tiled_create(C.byref(self._h), X, Y, Z, self.world, self.rank, uid, C.byref(self.params)), "tiled_create")
        want2 = os.environ.get("SOBFU_TILED_REDUCE_COMM", "1")  # "force": also on a world of one (bring-up / tests)
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
        self.layout = SlabLayout(dims, self.world, self.rank)
        v = [C.c_int() for _ in range(6)]
        _lib.check(L.sobfu_hip_tiled_layout(self._h, *[C.byref(x) for x in v]), "tiled_layout")
        got = tuple(x.value for x in v)
        want = (self.layout.z0, self.layout.z1, self.layout.lo, self.layout.hi, self.layout.Lz, self.layout.zbase)
        assert got == want, (got, want)

    EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p)
    ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p)

    def set_transport(self, exchange, allreduce_max):
        """communicator-less (dry) handles only: callables (rank, field_ptr, planes, stream) / (rank, buf_ptr, n, stream) -> 0"""
        self._cb = (self.EXCHANGE_FN(lambda ctx, r, f, p, st: exchange(r, f, p, st)),
                    self.ALLREDUCE_FN(lambda ctx, r, b, n, st: allreduce_max(r, b, n, st)))
        self._lib.check(self._lib.lib().sobfu_hip_tiled_set_transport(self._h, self._cb[0], self._cb[1], None), "tiled_set_transport")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.lib().sobfu_hip_tiled_destroy(self._h)
            self._h = None

    __del__ = close

    def new_local(self, channels):
        return torch.zeros(self.layout.local_shape(channels), dtype=torch.float32, device="cuda")

    def identity_psi(self):
        psi = self.new_local(4)
        X, Y, _ = self.layout.dims
        self._lib.check(self._lib.lib().sobfu_hip_tile_init_identity(C.c_void_p(psi.data_ptr()), X, Y, self.layout.Lz, self.layout.zbase,
                                                                     C.c_void_p(torch.cuda.current_stream().cuda_stream)), "tile_init_identity")
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

    def gather_owned(self, local):
        return TiledSolver.gather_owned(self, local)

    def estimate_psi(self, *args, **kw):
        return estimate_psi_tiled(self, *args, **kw)

    SCHEDULES = {1: "overlapped exchange, pass A split", 2: "overlapped exchange, pass A whole", 3: "serial (no overlap, no events)"}

    def set_schedule(self, schedule):
        self._lib.check(self._lib.lib().sobfu_hip_tiled_set_schedule(self._h, C.c_int(int(schedule))), "tiled_set_schedule")
        self.schedule = int(schedule)

    def autotune(self, phi_global_local, phi_n_full, iters=40):
        """Times the three ways of issuing an iteration on THIS machine (they give identical results) on scratch state and keeps
        the fastest; every rank takes part and all agree (MAX over ranks).  Returns {schedule: us per iteration}."""
        pnp, psi = self.new_local(2), self.identity_psi()
        times = {}
        for sched in self.SCHEDULES:
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


def _tiled_diagnostics(solver, kw, dims, pg, pn_full, world, rank, reps=30, iters=60):
    """Per-piece timings of the native loop on the machine at hand (microseconds; rank 0's view after a MAX over ranks)."""
    import os

    L, lib, check = solver.layout, solver._lib.lib(), solver._lib.check
    X, Y, _ = dims
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - t0) / n * 1e6], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    field = torch.zeros((L.Lz, Y, X, 3), dtype=torch.float32, device="cuda")
    for planes in (1, 2, HALO):  # latency vs bandwidth of a face message: 1/4, 1/2 and all of the loop's halo
        out[f"exchange_{planes}_planes_us"] = timed(
            lambda: check(lib.sobfu_hip_tiled_exchange(solver._h, C.c_void_p(field.data_ptr()), C.c_int(planes), st), "exchange"), reps)
    out["exchange_bytes_per_face"] = HALO * X * Y * 12
    slots = torch.zeros(SLOTS, dtype=torch.int32, device="cuda")
    out["allreduce_256_slots_us"] = timed(lambda: check(lib.sobfu_hip_tiled_allreduce_max_u32(solver._h, C.c_void_p(slots.data_ptr()), C.c_size_t(SLOTS), st), "allreduce"), reps)
    pnp, psi = solver.new_local(2), solver.identity_psi()

    def loop(s):
        return lambda: s.iterate(pg, pn_full, pnp, psi, iters)

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
    dry = NativeTiledSolver(dims, dry=(world, rank), **kw)  # same slab, no peers: the compute side alone
    out["iteration_us_compute_only"] = timed(loop(dry), 2) / iters
    dry.close()
    return {k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}


def bench_tiled(P, steps, warmup, rank, world):
    """bench.py leg for --gpus N > 1: the SAME 256^3 solve cut into N z-slabs (strong scaling)."""
    from . import ops

    dims = P["dims"]
    X, Y, Z = dims
    c0, c1, r = (0.375,) * 3, (0.375 + 1.3 * float(P["vs"][0]), 0.375, 0.375), 0.2
    import os

    kw = dict(alpha=P["alpha"], w_reg=P["w_reg"], s=P["s"], lam=P["lam"], max_update_norm=P["max_update_norm"])
    # default: the native C++ loop (RCCL issued from the library, exchange overlapped with the interior compute); if ANY
    # rank fails to set it up, every rank falls back to the torch.distributed loop (same schedule, same results)
    native = os.environ.get("SOBFU_TILED_NATIVE", "1") == "1"
    solver = None
    if native:
        try:
            solver = NativeTiledSolver(dims, **kw)
        except Exception as e:  # noqa: BLE001 -- report and agree on the fallback collectively
            print(f"[rank {rank}] native tiled loop unavailable: {e}", file=sys.stderr, flush=True)
        ok = torch.tensor([1 if solver is not None else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            if solver is not None:
                solver.close()
            solver, native = None, False
    if solver is None:
        solver = TiledSolver(dims, **kw)
    L = solver.layout
    # every rank builds the full analytic TSDFs (replicated phi_n; phi_global is then cut to the local slab)
    pg_full, pn_full = ops.new_volume(dims), ops.new_volume(dims)
    ops.init_sphere(pg_full, P["vs"], P["trunc"], P["eta"], c0, r)
    ops.init_sphere(pn_full, P["vs"], P["trunc"], P["eta"], c1, r)
    pg = L.take(pg_full).clone()
    del pg_full
    pnp = solver.new_local(2)
    psi = solver.identity_psi()
    tuned = None
    if native and os.environ.get("SOBFU_TILED_AUTOTUNE", "1") == "1":  # outside the timed region: pick this machine's best schedule
        tuned = solver.autotune(pg, pn_full)
    if warmup > 0:
        solver.iterate(pg, pn_full, pnp, psi, warmup)
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    done, norms = solver.iterate(pg, pn_full, pnp, psi, steps)
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    assert done == steps and np.isfinite(norms).all() and float(norms.max()) > 0
    # self-check, outside the timed region: every rank repeats the WHOLE solve on its own GPU with the single-GPU solver
    # handle and compares its owned planes and the max-norm history bit for bit (tiling must not change a single bit)
    parity = None
    if os.environ.get("SOBFU_TILED_SELFCHECK", "1") == "1":
        pg_full = ops.new_volume(dims)
        ops.init_sphere(pg_full, P["vs"], P["trunc"], P["eta"], c0, r)
        psi_full, pnp_full = ops.new_field(dims), ops.new_volume(dims)
        ops.init_identity(psi_full)
        one = ops.Solver(dims, max_iter=max(steps, warmup, 1), **kw)
        if warmup > 0:
            one.iterate(pg_full, pn_full, pnp_full, psi_full, warmup)
        _, norms_one = one.iterate(pg_full, pn_full, pnp_full, psi_full, steps)
        one.close()
        same = (np.array_equal(np.asarray(norms_one, np.float32).view(np.uint32), np.asarray(norms, np.float32).view(np.uint32))
                and torch.equal(L.owned(L.take(psi_full))[..., :3].contiguous().view(torch.int32), L.owned(psi)[..., :3].contiguous().view(torch.int32))
                and torch.equal(L.owned(L.take(pnp_full)).contiguous().view(torch.int32), L.owned(pnp).contiguous().view(torch.int32)))
        ok = torch.tensor([1 if same else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        parity = bool(int(ok.item()))
        if not parity:
            print(f"[rank {rank}] tiled self-check: slab differs from the single-GPU solve (local: {same})", file=sys.stderr, flush=True)
    # diagnostics for the next tuning round, outside the timed region (every rank takes part in the collective ones):
    # what one halo exchange, one slot all-reduce and the compute side alone cost on THIS machine, and both pass-A schedules
    diag, hung = None, False
    if native and os.environ.get("SOBFU_TILED_DIAG", "1") == "1":
        # in a worker thread with a deadline: whatever happens in there (an exception on one rank would leave the others
        # waiting in a collective), the benchmark line is still printed
        import threading

        box, dev_index = {}, torch.cuda.current_device()

        def work():
            try:
                torch.cuda.set_device(dev_index)
                box["diag"] = _tiled_diagnostics(solver, kw, dims, pg, pn_full, world, rank)
            except Exception as e:  # noqa: BLE001
                box["error"] = repr(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("SOBFU_TILED_DIAG_TIMEOUT", "120")))
        hung = th.is_alive()
        diag = box.get("diag") or {"error": "timed out" if hung else box.get("error", "unknown")}
        if "error" in diag:
            print(f"[rank {rank}] tiled diagnostics: {diag['error']}", file=sys.stderr, flush=True)
    return dict(diag_hung=hung, seconds=dt, N=X * Y * Z, ms_a=None, ms_b=None, last_norm=float(norms[-1]), workspace=None, tiled_parity=parity, tiled_diag=diag,
                parallelism=f"{world} z-slabs of {(Z + world - 1) // world} planes (+{HALO}-plane halos), RCCL halo exchange, "
                            + ("native C++ loop" if native else "torch.distributed loop")
                            + (f", schedule: {solver.SCHEDULES[solver.schedule]} (autotuned)" if tuned else ""),
                tiled_autotune_us={solver.SCHEDULES[k]: round(v, 2) for k, v in tuned.items()} if tuned else None)
