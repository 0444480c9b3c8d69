"""N ranks of the NATIVE tiled loop (sobfu_hip_tiled_iterate) on ONE GPU: communicator-less handles, one host thread per
rank, and an in-process loopback transport (device-to-device copies between the ranks' buffers + a host max) plugged in
through sobfu_hip_tiled_set_transport.  Everything but RCCL itself -- tile layout (z-slabs, x / y splits, 2 x 2 x 2), message
boxes, pass A's push boxes and the scatter kernel, boundary / interior plane ranges, the multi-box launches with the thin x / y
shells, halo widths, ungated pass A, stream and event order -- runs exactly as on N GPUs, and the gathered result must equal the
single-GPU solve bit for bit.  The DIRECT transport (halo cells stored straight into the neighbours' arrays, arrival flags and
max-norm rows the same way) runs here too: the ranks' arrays live in one process, so "peer-mapped" is a plain pointer."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Loopback:
    """transport of sobfu_hip_tiled_set_transport between rank threads of one process: every rank posts where its messages
    start, a barrier, every rank copies what its peers posted for it into its receive segments, a barrier"""

    def __init__(self, solvers):
        self.sv, self.N = solvers, len(solvers)
        self.bar = threading.Barrier(self.N)
        self.posted, self.host = [None] * self.N, [None] * self.N
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipStreamSynchronize.argtypes = [C.c_void_p]

    def _ok(self, rc):
        if rc != 0:
            self.bar.abort()
            raise RuntimeError(f"hip call failed: {rc}")

    def exchange(self, rank, send, recv, msgs, stream):
        try:
            self._ok(self.hip.hipStreamSynchronize(stream))  # what this rank sends is final
            self.posted[rank] = {peer: (send + 4 * soff, cnt) for peer, soff, _, cnt in msgs}
            self.bar.wait(timeout=60)
            for peer, _, roff, cnt in msgs:
                src, n = self.posted[peer][rank]
                assert n == cnt, (rank, peer, n, cnt)
                self._ok(self.hip.hipMemcpy(recv + 4 * roff, src, 4 * cnt, 3))
            self._ok(self.hip.hipDeviceSynchronize())
            self.bar.wait(timeout=60)  # nobody overwrites data a peer is still copying
            return 0
        except Exception as e:  # noqa: BLE001
            print("loopback exchange failed:", repr(e), flush=True)
            return -1

    def allreduce(self, rank, buf, n, stream):
        try:
            self._ok(self.hip.hipStreamSynchronize(stream))
            h = np.empty(n, np.uint32)
            self._ok(self.hip.hipMemcpy(h.ctypes.data, buf, 4 * n, 2))
            self.host[rank] = h
            self.bar.wait(timeout=60)
            m = np.maximum.reduce([x for x in self.host])
            self.bar.wait(timeout=60)
            self._ok(self.hip.hipMemcpy(buf, m.ctypes.data, 4 * n, 1))
            return 0
        except Exception as e:  # noqa: BLE001
            print("loopback allreduce failed:", repr(e), flush=True)
            return -1


def as_grid(world_or_grid):
    return (1, 1, world_or_grid) if isinstance(world_or_grid, int) else tuple(world_or_grid)


def assemble(solvers, parts):
    """owned parts of every rank -> the full volume"""
    X, Y, Z = solvers[0].layout.dims
    full = np.zeros((Z, Y, X) + parts[0].shape[3:], parts[0].dtype)
    for s, part in zip(solvers, parts):
        s.layout.owned_global(full)[...] = part
    return full


def run_world(dims, grid, psi0, pg, pn, n_iters, thr, schedule=None):
    import torch

    from sobfu_amd import tiled

    grid = as_grid(grid)
    world = grid[0] * grid[1] * grid[2]
    solvers = [tiled.NativeTiledSolver(dims, alpha=0.05, w_reg=0.4, max_update_norm=thr, dry=(world, r), grid=grid) for r in range(world)]
    lb = Loopback(solvers)
    for s in solvers:
        s.set_transport(lb.exchange, lb.allreduce)
        if schedule is not None:
            s.set_schedule(schedule)
    pn_d = torch.from_numpy(pn).cuda()
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            s = solvers[r]
            L = s.layout
            with torch.cuda.stream(torch.cuda.Stream()):
                pg_l = torch.from_numpy(np.ascontiguousarray(L.take(pg))).cuda()
                psi_l = torch.from_numpy(np.ascontiguousarray(L.take(psi0))).cuda()
                pnp_l = s.new_local(2)
                done, hist = s.iterate(pg_l, pn_d, pnp_l, psi_l, n_iters)
                torch.cuda.current_stream().synchronize()
            out[r] = (done, hist, L.owned(psi_l).cpu().numpy(), L.owned(pnp_l).cpu().numpy())
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
            lb.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    assert not errs, errs
    assert all(o is not None for o in out)
    full = (assemble(solvers, [o[2] for o in out]), assemble(solvers, [o[3] for o in out]))
    for s in solvers:
        s.close()
    return out, full


def test_transport_sees_what_the_queries_announce():
    """the pluggable-transport contract of include/sobfu_hip.h on a 2 x 2 x 2 split: every exchange is TWO calls -- the packed list
    (send buffer -> receive buffer: the first n_packed messages of sobfu_hip_tiled_messages) and the in-place list (the nabla_U array
    onto itself: exactly sobfu_hip_tiled_messages_inplace) -- so a transport that pre-posts its requests can build them up front"""
    import torch

    from sobfu_amd import tiled
    from sobfu_amd.synthetic import hash_field

    dims, grid = (24, 20, 16), (2, 2, 2)
    X, Y, Z = dims
    solvers = [tiled.NativeTiledSolver(dims, alpha=0.05, w_reg=0.4, max_update_norm=-1.0, dry=(8, r), grid=grid) for r in range(8)]
    lb = Loopback(solvers)
    calls = [[] for _ in range(8)]

    def spy(rank, send, recv, msgs, stream):
        calls[rank].append((send == recv, [tuple(m) for m in msgs]))
        return lb.exchange(rank, send, recv, msgs, stream)

    for s in solvers:
        s.set_transport(spy, lb.allreduce)
    pg = hash_field((Z, Y, X, 2), 41, 1.0)
    pn = hash_field((Z, Y, X, 2), 42, 1.0)
    psi0 = np.zeros((Z, Y, X, 4), np.float32)
    psi0[..., 0], psi0[..., 1], psi0[..., 2] = np.arange(X)[None, None, :], np.arange(Y)[None, :, None], np.arange(Z)[:, None, None]
    pn_d, errs = torch.from_numpy(pn).cuda(), []

    def rank_main(r):
        try:
            s = solvers[r]
            with torch.cuda.stream(torch.cuda.Stream()):
                s.iterate(torch.from_numpy(np.ascontiguousarray(s.layout.take(pg))).cuda(), pn_d, s.new_local(2),
                          torch.from_numpy(np.ascontiguousarray(s.layout.take(psi0))).cuda(), 2)
                torch.cuda.current_stream().synchronize()
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
            lb.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    assert not errs, errs
    for r, s in enumerate(solvers):
        inplace, n_packed = s.messages_inplace()
        assert len(inplace) == 1 and n_packed == 5  # x face, y face and the three edge strips through the buffers; the z face in place
        assert len(s.layout.messages()) == n_packed + len(inplace)
        assert len(calls[r]) == 4  # two exchanges (one per iteration), two calls each
        for k in (0, 2):
            same_base, msgs = calls[r][k]
            assert not same_base and len(msgs) == n_packed and len({m[0] for m in msgs}) == n_packed  # one message per peer in a call
            same_base, msgs = calls[r][k + 1]
            assert same_base and msgs == inplace
        Lx, Ly = s.layout.L[0], s.layout.L[1]
        assert all(cnt == 3 * Lx * Ly * 4 for _, _, _, cnt in inplace)  # 4 whole padded planes of 12-byte cells
        s.close()


def run_world_direct(dims, grid, psi0, pg, pn, n_iters, thr, stepped, solves=1, kw=None, keep_halo_lines_hot=False):
    """N ranks on the DIRECT transport in one process.  stepped: one host thread drives all ranks phase by phase (pass A incl.
    the pushes | pass B | ... | end-of-solve handshake) with the in-kernel waits off -- any number of ranks; else one thread
    per rank runs the real loop, in-kernel waits live (few ranks: every rank's stream needs a hardware queue of its own).
    keep_halo_lines_hot (stepped): right before every pass A -- whose push boxes store into the OTHER ranks' nabla_U halo cells -- a
    copy kernel reads every rank's nabla_U arena, so that the lines those stores are about to change sit in the reader's caches when
    the stores happen: pass B must still see the new cells (on connected handles it reads nabla_U and the max-norm rows at system scope)."""
    import ctypes as C

    import torch

    from sobfu_amd import tiled

    grid = as_grid(grid)
    world = grid[0] * grid[1] * grid[2]
    kw = kw or dict(alpha=0.05, w_reg=0.4)
    solvers = [tiled.NativeTiledSolver(dims, max_update_norm=thr, dry=(world, r), grid=grid, **kw) for r in range(world)]
    tiled.NativeTiledSolver.connect_local(solvers)
    torch.cuda.synchronize()
    pn_d = torch.from_numpy(pn).cuda()
    streams = [torch.cuda.Stream() for _ in range(world)]
    state = []
    for r, s in enumerate(solvers):
        L = s.layout
        with torch.cuda.stream(streams[r]):
            state.append([torch.from_numpy(np.ascontiguousarray(L.take(pg))).cuda(), torch.from_numpy(np.ascontiguousarray(L.take(psi0))).cuda(), s.new_local(2)])
    torch.cuda.synchronize()
    out = [None] * world
    for _ in range(solves):  # a second solve warm-starts from the first one's psi (sequence numbers and flags carry over)
        if stepped:
            for s in solvers:
                s.set_wait(False)
            for r, s in enumerate(solvers):
                with torch.cuda.stream(streams[r]):
                    s.begin(state[r][0], pn_d, state[r][2], state[r][1], n_iters)
            torch.cuda.synchronize()
            hip = C.CDLL("libamdhip64.so")
            hip.hipMemcpyAsync.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
            sink = torch.empty(2 * max(sv.layout.L[0] * sv.layout.L[1] * sv.layout.L[2] for sv in solvers) * 3, dtype=torch.float32, device="cuda")
            for phase in [p for _ in range(n_iters) for p in (0, 1)] + [2]:
                if keep_halo_lines_hot and phase == 0:
                    for r, s in enumerate(solvers):  # read both halves of this rank's nabla_U (halo cells included) -> its lines are cache-resident
                        e, nl = s.exports(), s.layout.L[0] * s.layout.L[1] * s.layout.L[2]
                        for h in (0, 1):
                            assert hip.hipMemcpyAsync(C.c_void_p(sink.data_ptr() + h * nl * 12), C.c_void_p(e.arena + e.nabla_u_off[h]), nl * 12, 3,
                                                      C.c_void_p(streams[r].cuda_stream)) == 0
                    torch.cuda.synchronize()
                for r, s in enumerate(solvers):
                    with torch.cuda.stream(streams[r]):
                        s.step_phase(phase)
                torch.cuda.synchronize()
            for r, s in enumerate(solvers):
                with torch.cuda.stream(streams[r]):
                    out[r] = s.end()
        else:
            errs = []

            def rank_main(r):
                try:
                    with torch.cuda.stream(streams[r]):
                        out[r] = solvers[r].iterate(state[r][0], pn_d, state[r][2], state[r][1], n_iters)
                except Exception as e:  # noqa: BLE001
                    errs.append((r, repr(e)))

            th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
            for t in th:
                t.start()
            for t in th:
                t.join(timeout=180)
            assert not errs, errs
        torch.cuda.synchronize()
        assert all(s.status()[0] for s in solvers)
    res = [(out[r][0], out[r][1], solvers[r].layout.owned(state[r][1]).cpu().numpy(), solvers[r].layout.owned(state[r][2]).cpu().numpy()) for r in range(world)]
    full = (assemble(solvers, [o[2] for o in res]), assemble(solvers, [o[3] for o in res]))
    for s in solvers:
        s.close()
    return res, full


@pytest.mark.parametrize("dims,grid,stepped", [((40, 24, 36), (2, 2, 2), True), ((64, 64, 64), (2, 2, 2), True), ((33, 17, 16), (1, 2, 2), True),
                                                ((40, 24, 36), (1, 1, 3), True), ((70, 33, 23), (2, 1, 1), True), ((141, 19, 17), (2, 1, 2), True),
                                                ((36, 36, 36), (3, 3, 3), True), ((8, 9, 10), (2, 2, 2), True),
                                                # wide rows (owned x >= 64) meeting y neighbours: the marched y-face push box stores its 8 rows at
                                                # home too and the owned block leaves them out (y_home), with and without the same along z
                                                ((130, 40, 20), (1, 2, 1), True), ((130, 40, 20), (2, 2, 1), True), ((130, 40, 20), (1, 2, 2), True),
                                                ((130, 60, 12), (1, 3, 1), True),
                                                # in-kernel waits live: one thread per rank, the real loop
                                                ((40, 24, 36), (1, 1, 2), False), ((70, 33, 23), (2, 1, 1), False), ((40, 24, 36), (1, 2, 1), False)])
def test_native_loop_direct_transport(dims, grid, stepped):
    """The direct transport against the single-GPU solve, bit for bit: psi, phi_n o psi, the GLOBAL max-norm history (made global
    by stores into the peers' rows, no collective), dead / live / firing thresholds, and a warm-started second solve."""
    import torch

    import oracle
    from sobfu_amd import ops

    rng = np.random.default_rng(5)
    X, Y, Z = dims
    pg = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    pn = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    psi0 = oracle.new_field(dims)
    oracle.init_identity(psi0)
    psi0[..., :3] += rng.uniform(-0.7, 0.7, psi0[..., :3].shape).astype(np.float32)
    n_iters = 6

    def single(thr, solves):
        sv = ops.Solver(dims, max_iter=n_iters, alpha=0.05, w_reg=0.4, max_update_norm=thr)
        psi, pnp = torch.from_numpy(psi0.copy()).cuda(), ops.new_volume(dims)
        for _ in range(solves):
            rep, hist = sv.iterate(torch.from_numpy(pg).cuda(), torch.from_numpy(pn).cuda(), pnp, psi, n_iters)
        sv.close()
        return rep.iterations, np.asarray(hist[:rep.iterations], np.float32), psi.cpu().numpy(), pnp.cpu().numpy()

    _, hist_dead, _, _ = single(-1.0, 1)
    for thr, solves in ((-1.0, 1), (1e-10, 2), (float(hist_dead[2]), 1)):
        done_e, hist_e, psi_e, pnp_e = single(thr, solves)
        out, (psi_t, pnp_t) = run_world_direct(dims, grid, psi0, pg, pn, n_iters, thr, stepped, solves)
        for done, hist, _, _ in out:
            assert done == done_e
            assert np.array_equal(np.asarray(hist, np.float32).view(np.uint32), hist_e.view(np.uint32))
        assert np.array_equal(psi_t[..., :3].view(np.uint32), psi_e[..., :3].view(np.uint32))
        assert np.array_equal(pnp_t.view(np.uint32), pnp_e.view(np.uint32))


@pytest.mark.parametrize("dims,grid", [((40, 24, 36), (2, 2, 2)), ((64, 64, 64), (2, 2, 2)), ((40, 24, 36), (1, 1, 3))])
def test_direct_transport_with_stale_halo_lines_in_cache(dims, grid):
    """pass B reads halo cells OTHER ranks stored (on real hardware: other GPUs).  Here every rank's
    nabla_U halo lines are deliberately made cache-resident right before the peers overwrite them (keep_halo_lines_hot): the
    solve must still equal the single-GPU one bit for bit.  On ONE GPU the ordinary kernel boundary already guarantees that (this
    test cannot fail for the reason it guards against -- only a second GPU can show that); what it does exercise is the explicit
    system-scope loads of pass B on connected handles, under exactly the access pattern they exist for."""
    import torch

    import oracle
    from sobfu_amd import ops

    rng = np.random.default_rng(11)
    X, Y, Z = dims
    pg = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    pn = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    psi0 = oracle.new_field(dims)
    oracle.init_identity(psi0)
    psi0[..., :3] += rng.uniform(-0.7, 0.7, psi0[..., :3].shape).astype(np.float32)
    sv = ops.Solver(dims, max_iter=6, alpha=0.05, w_reg=0.4, max_update_norm=1e-10)
    psi, pnp = torch.from_numpy(psi0.copy()).cuda(), ops.new_volume(dims)
    rep, hist = sv.iterate(torch.from_numpy(pg).cuda(), torch.from_numpy(pn).cuda(), pnp, psi, 6)
    sv.close()
    out, (psi_t, pnp_t) = run_world_direct(dims, grid, psi0, pg, pn, 6, 1e-10, True, keep_halo_lines_hot=True)
    for done, h, _, _ in out:
        assert done == rep.iterations and np.array_equal(np.asarray(h, np.float32).view(np.uint32), np.asarray(hist, np.float32).view(np.uint32))
    assert np.array_equal(psi_t[..., :3].view(np.uint32), psi.cpu().numpy()[..., :3].view(np.uint32))
    assert np.array_equal(pnp_t.view(np.uint32), pnp.cpu().numpy().view(np.uint32))


@pytest.mark.parametrize("dims,grid", [((70, 33, 23), (2, 1, 1)), ((40, 24, 36), (1, 1, 2)), ((64, 64, 64), (2, 2, 2))])
def test_direct_transport_on_tiles_beyond_the_cache(dims, grid, monkeypatch):
    """N = 2 and N = 4 cut 256^3 into tiles that do NOT fit the Infinity Cache: streaming hints on, and -- on connected handles -- still
    the pipelined march of pass B (its loads carry the system scope for the cells other GPUs stored).  Forced here on small grids with
    SOBFU_CACHE_CELLS=0: the instantiation <thin boxes, streaming hints, pipelined> of pass B and the streaming variant of the tile's
    pass A, bit for bit against the single-GPU solve."""
    import torch

    import oracle
    from sobfu_amd import ops

    rng = np.random.default_rng(23)
    X, Y, Z = dims
    pg = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    pn = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    psi0 = oracle.new_field(dims)
    oracle.init_identity(psi0)
    psi0[..., :3] += rng.uniform(-0.7, 0.7, psi0[..., :3].shape).astype(np.float32)
    sv = ops.Solver(dims, max_iter=6, alpha=0.05, w_reg=0.4, max_update_norm=1e-10)
    psi, pnp = torch.from_numpy(psi0.copy()).cuda(), ops.new_volume(dims)
    rep, hist = sv.iterate(torch.from_numpy(pg).cuda(), torch.from_numpy(pn).cuda(), pnp, psi, 6)
    sv.close()
    monkeypatch.setenv("SOBFU_CACHE_CELLS", "0")
    out, (psi_t, pnp_t) = run_world_direct(dims, grid, psi0, pg, pn, 6, 1e-10, True)
    for done, h, _, _ in out:
        assert done == rep.iterations and np.array_equal(np.asarray(h, np.float32).view(np.uint32), np.asarray(hist, np.float32).view(np.uint32))
    assert np.array_equal(psi_t[..., :3].view(np.uint32), psi.cpu().numpy()[..., :3].view(np.uint32))
    assert np.array_equal(pnp_t.view(np.uint32), pnp.cpu().numpy().view(np.uint32))


def test_direct_transport_deadline():
    """A peer that never shows up: the wait in pass A's tail gives up at the deadline, records WHOM it missed, the handle reports
    SOBFU_E_TIMEOUT -- and the GPU is never hung."""
    import os

    import torch

    from sobfu_amd import tiled

    os.environ["SOBFU_TILED_DEADLINE_S"] = "0.5"
    try:
        dims, grid = (40, 24, 36), (1, 1, 2)
        solvers = [tiled.NativeTiledSolver(dims, alpha=0.05, w_reg=0.4, dry=(2, r), grid=grid) for r in range(2)]
    finally:
        os.environ.pop("SOBFU_TILED_DEADLINE_S")
    tiled.NativeTiledSolver.connect_local(solvers)
    s = solvers[0]  # rank 1 never runs
    pg, pn = s.new_local(2), torch.zeros((36, 24, 40, 2), device="cuda")
    pnp, psi = s.new_local(2), s.identity_psi()
    with pytest.raises(RuntimeError, match="deadline"):
        s.iterate(pg, pn, pnp, psi, 3)
    ok, missing = s.status()
    assert not ok and missing == 1
    torch.cuda.synchronize()
    for q in solvers:
        q.close()


@pytest.mark.parametrize("dims,world,split", [((40, 24, 36), 3, None), ((40, 24, 36), 3, "1"), ((33, 17, 16), 4, None), ((20, 12, 120), 2, None),
                                               ((70, 33, 23), 2, "1"), ((64, 64, 64), 4, "0"), ((40, 24, 36), 3, "serial"), ((33, 17, 16), 4, "serial"),
                                               # 3-D tile grids: BASELINE config 4's 2 x 2 x 2, the N = 4 default 1 x 2 x 2, single-axis x / y
                                               # splits, ragged extents, an interior tile with both neighbours on every axis
                                               ((40, 24, 36), (2, 2, 2), None), ((64, 64, 64), (2, 2, 2), None), ((33, 17, 16), (1, 2, 2), None),
                                               ((70, 33, 23), (2, 1, 1), None), ((40, 24, 36), (1, 2, 1), None), ((40, 24, 36), (2, 2, 1), None),
                                               ((141, 19, 17), (2, 1, 2), None), ((36, 36, 36), (3, 3, 3), None),
                                               # wide rows meeting y neighbours on one side / on both sides, with and without a z split (y_home)
                                               ((130, 40, 20), (1, 2, 1), None), ((130, 40, 20), (2, 2, 1), None), ((130, 40, 20), (1, 2, 2), None),
                                               ((130, 60, 12), (1, 3, 1), None),
                                               # the smallest tiles the layout allows (4 owned cells per split axis: a message is the whole tile)
                                               ((8, 9, 10), (2, 2, 2), None), ((12, 8, 8), (3, 2, 1), None)])
def test_native_loop_n_ranks_loopback(dims, world, split, monkeypatch):
    import torch

    import oracle
    from sobfu_amd import ops

    # how the z-slab iteration is issued (sobfu_hip_tiled_set_schedule): 3 serial, 1 overlapped with pass A split, 2 overlapped, pass A whole
    schedule = {None: None, "serial": 3, "1": 1, "0": 2}[split]
    rng = np.random.default_rng(5)
    X, Y, Z = dims
    pg = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    pn = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    psi0 = oracle.new_field(dims)
    oracle.init_identity(psi0)
    psi0[..., :3] += rng.uniform(-0.7, 0.7, psi0[..., :3].shape).astype(np.float32)
    n_iters = 6
    ref = ops.Solver(dims, max_iter=n_iters, alpha=0.05, w_reg=0.4)
    psi_r, pnp_r = torch.from_numpy(psi0.copy()).cuda(), ops.new_volume(dims)
    _, hist_r = ref.iterate(torch.from_numpy(pg).cuda(), torch.from_numpy(pn).cuda(), pnp_r, psi_r, n_iters)
    ref.close()
    for thr, expect in ((-1.0, n_iters), (1e-10, n_iters), (float(hist_r[2]), 3)):
        if expect < n_iters:
            a = ops.Solver(dims, max_iter=n_iters, alpha=0.05, w_reg=0.4, max_update_norm=thr)
            psi_e, pnp_e = torch.from_numpy(psi0.copy()).cuda(), ops.new_volume(dims)
            rep, _ = a.iterate(torch.from_numpy(pg).cuda(), torch.from_numpy(pn).cuda(), pnp_e, psi_e, n_iters)
            a.close()
            assert rep.iterations == expect
        else:
            psi_e, pnp_e = psi_r, pnp_r
        out, (psi_t, pnp_t) = run_world(dims, world, psi0, pg, pn, n_iters, thr, schedule)
        for done, hist, _, _ in out:
            assert done == expect
            assert np.array_equal(np.asarray(hist, np.float32).view(np.uint32), np.asarray(hist_r[:expect], np.float32).view(np.uint32))
        assert np.array_equal(psi_t[..., :3].view(np.uint32), psi_e.cpu().numpy()[..., :3].view(np.uint32))
        assert np.array_equal(pnp_t.view(np.uint32), pnp_e.cpu().numpy().view(np.uint32))


class LoopGather:
    """all_gather of the owned cells between the rank threads of one process (stands in for dist.all_gather)"""

    def __init__(self, solvers):
        self.sv, n = solvers, len(solvers)
        self.parts, self.bar = [None] * n, threading.Barrier(n)

    def make(self, rank, layout):
        import torch

        def gather(local):
            torch.cuda.current_stream().synchronize()
            self.parts[rank] = layout.owned(local).contiguous()
            torch.cuda.current_stream().synchronize()
            self.bar.wait(timeout=60)
            X, Y, Z = layout.dims
            full = torch.empty((Z, Y, X) + tuple(local.shape[3:]), dtype=local.dtype, device=local.device)
            for s, part in zip(self.sv, self.parts):
                s.layout.owned_global(full).copy_(part)
            torch.cuda.current_stream().synchronize()
            self.bar.wait(timeout=60)  # nobody replaces its part while a peer is still assembling
            return full

        return gather


class LoopHalo:
    """the per-frame tail's communication (sobfu_amd.tiled.DistHalo's interface) between the rank threads of one process: a MAX
    reduction and the bounded-reach window of a field, assembled from the owners' parts by the same window_plan the real one uses"""

    def __init__(self, solvers, understate_reach=1.0):
        self.sv, n = solvers, len(solvers)
        self.parts, self.vals, self.bar = [None] * n, [None] * n, threading.Barrier(n)
        self.understate = understate_reach  # < 1: the first reduction of a tail (the reach) comes back too small -> windows too narrow

    def make(self, rank, layout):
        import torch

        from sobfu_amd import tiled

        outer = self

        class Halo:
            bytes_received = 0
            calls = 0

            def allreduce_max(self, v):
                outer.vals[rank] = float(v)
                outer.bar.wait(timeout=60)
                m = max(outer.vals)
                outer.bar.wait(timeout=60)
                self.calls += 1
                return m * outer.understate if self.calls % 2 == 1 else m  # (a tail reduces twice: the reach, then the violation flag)

            def window(self, local, w, nch):
                torch.cuda.current_stream().synchronize()
                outer.parts[rank] = layout.owned(local).contiguous()
                torch.cuda.current_stream().synchronize()
                outer.bar.wait(timeout=60)
                wb, recvs, _ = tiled.window_plan(layout, w)
                win = torch.zeros((wb[5] - wb[4], wb[3] - wb[2], wb[1] - wb[0]) + tuple(local.shape[3:]), dtype=local.dtype, device=local.device)
                origin = (wb[0], wb[2], wb[4])
                tiled._cut(win, layout.owned_box_global(), origin).copy_(layout.owned(local))
                for q, b in recvs:
                    Lq = outer.sv[q].layout
                    src = tiled._cut(outer.parts[q], b, Lq.g0)
                    tiled._cut(win, b, origin)[..., :nch].copy_(src[..., :nch])
                    self.bytes_received += src[..., :nch].numel() * 4
                torch.cuda.current_stream().synchronize()
                outer.bar.wait(timeout=60)
                return win, wb

        return Halo()


@pytest.mark.parametrize("tail,amp", [("gather", 0.5), ("halo", 0.5), ("halo", 2.6), ("halo-overflow", 13.0), ("halo-violation", 2.6)])
@pytest.mark.parametrize("dims,world", [((40, 24, 36), 3), ((33, 17, 16), 4), ((40, 24, 36), (2, 2, 2)), ((33, 17, 16), (1, 2, 2))])
def test_tiled_frame_estimate_psi_loopback(dims, world, tail, amp):
    """A whole frame on tiles = the single-GPU Solver::estimate_psi, bit for bit, on every rank's owned cells -- with the tail
    (48-sweep inverse, canonical warp) on all-gathered sources ("gather": the two collectives of SURVEY 8(e)), on bounded-reach WINDOWS
    of psi / phi_global fetched from the neighbours ("halo": one MAX reduction of |psi - id| sizes them; amp 2.6 makes them 5 cells
    wide and the samples really leave the tile), and with a displacement that outgrows the tiles ("halo-overflow": the windows would
    reach past the neighbours, so the tail falls back to the all-gather), and with a reach bound that does NOT hold ("halo-violation": the
    test double reports a reduced reach of zero, the windows come out two cells wide under a 2.6-voxel displacement, samples leave them -- the kernels raise their flag
    instead of reading outside the arrays, every rank learns of it and the tail is redone on all-gathered sources)."""
    import torch

    import oracle
    from sobfu_amd import ops, tiled

    rng = np.random.default_rng(9)
    X, Y, Z = dims
    pg = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    pn = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    psi0 = oracle.new_field(dims)
    oracle.init_identity(psi0)
    psi0[..., :3] += rng.uniform(-amp, amp, psi0[..., :3].shape).astype(np.float32)
    n_iters = 5
    ref = ops.Solver(dims, max_iter=n_iters, alpha=0.05, w_reg=0.4)
    psi_r, inv_r, pnp_r, pgi_r = torch.from_numpy(psi0.copy()).cuda(), ops.new_field(dims), ops.new_volume(dims), ops.new_volume(dims)
    ref.estimate_psi(torch.from_numpy(pg).cuda(), pgi_r, torch.from_numpy(pn).cuda(), pnp_r, psi_r, inv_r)
    ref.close()

    grid = as_grid(world)
    world = grid[0] * grid[1] * grid[2]
    solvers = [tiled.NativeTiledSolver(dims, alpha=0.05, w_reg=0.4, dry=(world, r), grid=grid) for r in range(world)]
    lb, lg, lh = Loopback(solvers), LoopGather(solvers), LoopHalo(solvers, 0.0 if tail == "halo-violation" else 1.0)
    for s in solvers:
        s.set_transport(lb.exchange, lb.allreduce)
    pn_d = torch.from_numpy(pn).cuda()
    out, errs, modes = [None] * world, [], [None] * world

    def rank_main(r):
        try:
            s = solvers[r]
            L = s.layout
            with torch.cuda.stream(torch.cuda.Stream()):
                pg_l = torch.from_numpy(np.ascontiguousarray(L.take(pg))).cuda()
                psi_l = torch.from_numpy(np.ascontiguousarray(L.take(psi0))).cuda()
                pnp_l, pgi_l, inv_l = s.new_local(2), s.new_local(2), s.new_local(4)
                inv_l[...] = -5.0  # whatever the tail leaves outside the owned cells, it must produce the owned ones itself
                done, _ = s.estimate_psi(pg_l, pgi_l, pn_d, pnp_l, psi_l, inv_l, n_iters, gather=lg.make(r, L),
                                         halo=(None if tail == "gather" else lh.make(r, L)))
                torch.cuda.current_stream().synchronize()
            modes[r] = dict(s.tail_stats)
            out[r] = (done, L.owned(psi_l).cpu().numpy(), L.owned(pnp_l).cpu().numpy(), L.owned(inv_l).cpu().numpy(), L.owned(pgi_l).cpu().numpy())
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
            lb.bar.abort()
            lg.bar.abort()
            lh.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=180)
    assert not errs, errs
    cat = lambda i: assemble(solvers, [o[i] for o in out])  # noqa: E731
    for s in solvers:
        s.close()
    assert all(o[0] == n_iters for o in out)
    assert np.array_equal(cat(1)[..., :3].view(np.uint32), psi_r.cpu().numpy()[..., :3].view(np.uint32))
    assert np.array_equal(cat(2).view(np.uint32), pnp_r.cpu().numpy().view(np.uint32))
    assert np.array_equal(cat(3)[..., :3].view(np.uint32), inv_r.cpu().numpy()[..., :3].view(np.uint32))
    assert np.array_equal(cat(4).view(np.uint32), pgi_r.cpu().numpy().view(np.uint32))
    full = dims[0] * dims[1] * dims[2] * 24
    fits = int(np.ceil(modes[0]["reach"] or 0)) + 2 <= tiled.TileLayout(dims, grid, 0).min_owned_extent()  # (33, 17, 16) on four 4-plane slabs: 5 > 4
    if tail == "halo-violation":
        # every rank takes the same path; whether a sample really leaves a 2-cell window depends on the data (trilinear averaging of
        # uncorrelated noise rarely keeps a 2.6-voxel displacement): it does on the three 12-plane slabs of (40, 24, 36) -- there the fallback must have run
        assert len({m["mode"] for m in modes}) == 1 and all(m["halo_width"] == 2 for m in modes), modes
        assert modes[0]["mode"] in ("halo",) or modes[0]["mode"].startswith("all-gather (a sample left its window"), modes
        if dims == (40, 24, 36) and grid == (1, 1, 3):
            assert modes[0]["mode"].startswith("all-gather (a sample left its window"), modes
    elif tail == "halo" and fits:
        assert all(m["mode"] == "halo" and m["halo_width"] == int(np.ceil(m["reach"])) + 2 for m in modes), modes
        assert all(0 < m["bytes_received"] < full for m in modes) and (amp < 1 or modes[0]["halo_width"] >= 5)
    else:
        assert all(m["mode"] == "all-gather" for m in modes), modes
        assert tail == "gather" or not fits


@pytest.mark.parametrize("tail", ["gather", "halo"])
@pytest.mark.parametrize("world", [2, 4, (2, 2, 2), (1, 2, 2)])
def test_tiled_fusion_frames_loopback(world, tail):
    """BASELINE config 1 (64^3, translating sphere, three frames) through TiledFusion on N tiles = the same frames through the
    single-GPU launchers: phi_global after every frame and psi at the end, bit for bit."""
    import torch

    from sobfu_amd import ops, synthetic, tiled

    dims, size = (64, 64, 64), (0.5, 0.5, 0.5)
    vs = 0.5 / 64
    intr = (570.342, 570.342, 320.0, 240.0)
    P = dict(size=size, trunc=5 * vs, eta=2 * vs, max_weight=128.0, intr=intr, R=np.eye(3, dtype=np.float32),
             t=np.array([-0.25, -0.25, 0.5], np.float32), start_frame=1, bilateral=(7, 4.5, 0.005), trunc_depth=1.5, max_iter=6)
    frames = [torch.from_numpy(synthetic.render_sphere_depth((0.005 * n, 0.0, 0.75), 0.1, intr, 480, 640)).cuda() for n in range(3)]
    kw = dict(alpha=0.1, w_reg=0.2)

    # single-GPU pipeline through the same launchers
    sv = ops.Solver(dims, max_iter=P["max_iter"], **kw)
    pg, pn, pnp, pgi = (ops.new_volume(dims) for _ in range(4))
    psi, psi_inv = ops.new_field(dims), ops.new_field(dims)
    ops.init_identity(psi)
    ops.init_identity(psi_inv)
    ref_pg = []
    for n, f in enumerate(frames):
        d = ops.bilateral_filter(f, *P["bilateral"])
        ops.truncate_depth(d, P["trunc_depth"])
        dists = ops.compute_dists(d, intr)
        if n == 0:
            ops.integrate_depth(dists, pg, (vs,) * 3, P["trunc"], P["eta"], P["R"], P["t"], intr)
        else:
            ops.clear_volume(pn)
            ops.integrate_depth(dists, pn, (vs,) * 3, P["trunc"], P["eta"], P["R"], P["t"], intr)
            sv.estimate_psi(pg, pgi, pn, pnp, psi, psi_inv)
            ops.integrate_fuse(pg, pnp, P["max_weight"])
        ref_pg.append(pg.cpu().numpy().copy())
    sv.close()

    grid = as_grid(world)
    world = grid[0] * grid[1] * grid[2]
    solvers = [tiled.NativeTiledSolver(dims, dry=(world, r), grid=grid, **kw) for r in range(world)]
    lb, lg, lh = Loopback(solvers), LoopGather(solvers), LoopHalo(solvers)
    for s in solvers:
        s.set_transport(lb.exchange, lb.allreduce)
    out, errs = [None] * world, []

    def rank_main(r):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                fu = tiled.TiledFusion(solvers[r], P, gather=lg.make(r, solvers[r].layout), halo=(lh.make(r, solvers[r].layout) if tail == "halo" else None))
                L, pgs = solvers[r].layout, []
                for f in frames:
                    fu(f)
                    torch.cuda.current_stream().synchronize()
                    pgs.append(L.owned(fu.phi_global).cpu().numpy().copy())
                out[r] = (pgs, L.owned(fu.psi).cpu().numpy(), L.owned(fu.psi_inv).cpu().numpy())
                assert solvers[r].tail_stats["mode"] == ("halo" if tail == "halo" else "all-gather"), solvers[r].tail_stats
        except Exception as e:  # noqa: BLE001
            errs.append((r, repr(e)))
            lb.bar.abort()
            lg.bar.abort()
            lh.bar.abort()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=240)
    assert not errs, errs
    for n in range(3):
        got = assemble(solvers, [o[0][n] for o in out])
        assert np.array_equal(got.view(np.uint32), ref_pg[n].view(np.uint32)), n
    assert np.array_equal(assemble(solvers, [o[1] for o in out])[..., :3].view(np.uint32), psi.cpu().numpy()[..., :3].view(np.uint32))
    assert np.array_equal(assemble(solvers, [o[2] for o in out])[..., :3].view(np.uint32), psi_inv.cpu().numpy()[..., :3].view(np.uint32))
    for s in solvers:
        s.close()
    assert float(np.abs(psi.cpu().numpy()[..., 0] - np.arange(64)[None, None, :]).max()) > 1e-3  # the solves really moved psi


def test_config4_256_cubed_on_2x2x2_tiles_loopback():
    """BASELINE config 4 at full size: the 256^3 roofline workload (params_boxing.ini solver values, two analytic spheres) cut
    into 2 x 2 x 2 tiles of 128^3, eight ranks of the native loop on this one GPU over the loopback transport -- everything of the
    8-GPU run except RCCL itself.  psi, phi_n o psi and the max-norm history equal the single-GPU solve bit for bit, with the
    ini's 1e-10 threshold live (the max-norm all-reduce and the late gate run every iteration)."""
    import torch

    import bench
    from sobfu_amd import ops, tiled

    free, _ = torch.cuda.mem_get_info()
    if free < 24 * 2 ** 30:
        pytest.skip("needs ~16 GiB of HBM")
    P = bench.boxing_params(256)
    dims, grid, n_iters = P["dims"], (2, 2, 2), 8
    c0, c1, r = bench.sphere_pair(P)
    pg, pn = ops.new_volume(dims), ops.new_volume(dims)
    ops.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r)
    ops.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
    kw = dict(alpha=P["alpha"], w_reg=P["w_reg"], s=P["s"], lam=P["lam"], max_update_norm=P["max_update_norm"])
    ref = ops.Solver(dims, max_iter=n_iters, **kw)
    psi_r, pnp_r = ops.new_field(dims), ops.new_volume(dims)
    ops.init_identity(psi_r)
    rep, hist_r = ref.iterate(pg, pn, pnp_r, psi_r, n_iters)
    ref.close()
    assert rep.iterations == n_iters and float(hist_r.min()) > 0

    solvers = [tiled.NativeTiledSolver(dims, dry=(8, q), grid=grid, **kw) for q in range(8)]
    assert all(tuple(s.layout.L) == (132, 132, 132) and len(s.layout.messages()) == 6 for s in solvers)
    lb = Loopback(solvers)
    for s in solvers:
        s.set_transport(lb.exchange, lb.allreduce)
    ok, errs = [None] * 8, []

    def rank_main(q):
        try:
            s = solvers[q]
            L = s.layout
            with torch.cuda.stream(torch.cuda.Stream()):
                pg_l = L.take(pg).clone().contiguous()
                pnp_l, psi_l = s.new_local(2), s.identity_psi()
                done, hist = s.iterate(pg_l, pn, pnp_l, psi_l, n_iters)
                torch.cuda.current_stream().synchronize()
                same = (done == n_iters and np.array_equal(np.asarray(hist, np.float32).view(np.uint32), np.asarray(hist_r, np.float32).view(np.uint32))
                        and torch.equal(L.owned(psi_l)[..., :3].contiguous().view(torch.int32), L.owned_global(psi_r)[..., :3].contiguous().view(torch.int32))
                        and torch.equal(L.owned(pnp_l).contiguous().view(torch.int32), L.owned_global(pnp_r).contiguous().view(torch.int32)))
            ok[q] = bool(same)
        except Exception as e:  # noqa: BLE001
            errs.append((q, repr(e)))
            lb.bar.abort()

    th = [threading.Thread(target=rank_main, args=(q,)) for q in range(8)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errs, errs
    for s in solvers:
        s.close()
    assert ok == [True] * 8, ok


def test_config4_256_cubed_on_2x2x2_tiles_direct_transport():
    """BASELINE config 4 at full size on the DIRECT transport: the 256^3 roofline workload cut into 2 x 2 x 2 tiles of 128^3, eight
    ranks in this process (phase-stepped: pass A incl. the stores into the neighbours' arrays | pass B | ... | handshake), the ini's
    1e-10 threshold live (max-norm rows made global by stores, the late gate reads them every iteration): psi, phi_n o psi and the
    max-norm history equal the single-GPU solve bit for bit."""
    import torch

    import bench
    from sobfu_amd import ops

    free, _ = torch.cuda.mem_get_info()
    if free < 24 * 2 ** 30:
        pytest.skip("needs ~16 GiB of HBM")
    P = bench.boxing_params(256)
    dims, n_iters = P["dims"], 8
    c0, c1, r = bench.sphere_pair(P)
    pg, pn = ops.new_volume(dims), ops.new_volume(dims)
    ops.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r)
    ops.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
    kw = dict(alpha=P["alpha"], w_reg=P["w_reg"], s=P["s"], lam=P["lam"])
    ref = ops.Solver(dims, max_iter=n_iters, max_update_norm=P["max_update_norm"], **kw)
    psi_r, pnp_r = ops.new_field(dims), ops.new_volume(dims)
    ops.init_identity(psi_r)
    rep, hist_r = ref.iterate(pg, pn, pnp_r, psi_r, n_iters)
    ref.close()
    assert rep.iterations == n_iters and float(hist_r.min()) > 0
    psi0 = ops.new_field(dims)
    ops.init_identity(psi0)
    out, (psi_t, pnp_t) = run_world_direct(dims, (2, 2, 2), psi0.cpu().numpy(), pg.cpu().numpy(), pn.cpu().numpy(), n_iters,
                                           P["max_update_norm"], True, 1, kw)
    for done, hist, _, _ in out:
        assert done == n_iters
        assert np.array_equal(np.asarray(hist, np.float32).view(np.uint32), np.asarray(hist_r, np.float32).view(np.uint32))
    assert np.array_equal(psi_t[..., :3].view(np.uint32), psi_r.cpu().numpy()[..., :3].view(np.uint32))
    assert np.array_equal(pnp_t.view(np.uint32), pnp_r.cpu().numpy().view(np.uint32))


def test_config4_tiles_against_the_emulated_reference_256():
    """BASELINE config 4 against the reference's own Solver::estimate_psi under emulation (tests/golden/ref_config3_256.npz: config 4 is
    config 3's workload on tiles): 2 x 2 x 2 tiles of 128^3 on the direct transport, 50 iterations with the ini's threshold live; the
    assembled psi and phi_n o psi hash to the digests of the reference's arrays."""
    import torch

    import oracle as O
    from test_reference_fixtures import _sphere_pair, check, load

    if torch.cuda.mem_get_info()[0] < 24 * 2 ** 30:
        pytest.skip("needs ~16 GiB of HBM")
    f = load("ref_config3_256")
    P, dims = f["P"], (256, 256, 256)
    pg, pn = _sphere_pair(O, P, dims)  # the reference's input volumes (libm powf), checked against the fixture
    check(f, "phi_global", pg), check(f, "phi_n", pn)
    psi0 = np.zeros((256, 256, 256, 4), np.float32)
    psi0[..., 0], psi0[..., 1], psi0[..., 2] = np.arange(256)[None, None, :], np.arange(256)[None, :, None], np.arange(256)[:, None, None]
    kw = dict(alpha=P["alpha"], w_reg=P["w_reg"], s=7, lam=0.1)
    out, (psi_t, pnp_t) = run_world_direct(dims, (2, 2, 2), psi0, pg, pn, 50, P["max_update_norm"], True, 1, kw)
    assert all(done == 50 for done, _, _, _ in out)
    psi4 = np.zeros_like(psi0)
    psi4[..., :3] = psi_t[..., :3]  # (the w lane of the reference's psi is 0; the tiles carry xyz)
    check(f, "psi", psi4), check(f, "phi_n_psi", pnp_t)


def test_box_list_cache_turnover_leaves_live_handles_alone():
    """A process that keeps creating tile handles keeps creating box lists; the library's cache of their device copies is bounded
    (256 per device) and retires its entries when full.  A handle that was created BEFORE the turnover -- its planned pass-A launches
    own their lists -- must compute the same bits after it, and so must handles created during it (found in round 5: the bound first
    freed lists that live plans still pointed to; the full GPU suite, > 256 handles in one process, caught it once in four runs)."""
    import torch

    from sobfu_amd import ops, tiled

    dims = (40, 24, 36)
    rng = np.random.default_rng(3)
    X, Y, Z = dims
    pg = torch.from_numpy(np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)).cuda()
    pn = torch.from_numpy(np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)).cuda()

    def run(sv):
        L = sv.layout
        pg_l, pnp_l, psi_l = L.take(pg).clone().contiguous(), sv.new_local(2), sv.identity_psi()
        sv.iterate(pg_l, pn, pnp_l, psi_l, 3)
        torch.cuda.synchronize()
        return psi_l.clone(), pnp_l.clone()

    old = tiled.NativeTiledSolver(dims, alpha=0.05, w_reg=0.4, dry=(8, 7), grid=(2, 2, 2))  # planned launches (push boxes into its send buffer)
    one = tiled.NativeTiledSolver(dims, alpha=0.05, w_reg=0.4, dry=(1, 0), grid=(1, 1, 1))  # a world of one: the cached (un-planned) path
    want_old, want_one = run(old), run(one)
    for i in range(300):  # 300 distinct lists through the cache (a world of one looks its list up at every launch)
        sv = tiled.NativeTiledSolver((8 + i % 150, 8 + i // 150, 8), alpha=0.05, w_reg=0.4, dry=(1, 0), grid=(1, 1, 1))
        L = sv.layout
        sv.iterate(sv.new_local(2), torch.zeros((8, 8 + i // 150, 8 + i % 150, 2), device="cuda"), sv.new_local(2), sv.identity_psi(), 1)
        sv.close()
    for sv, want in ((old, want_old), (one, want_one)):
        got = run(sv)
        assert torch.equal(got[0].view(torch.int32), want[0].view(torch.int32)) and torch.equal(got[1].view(torch.int32), want[1].view(torch.int32))
        sv.close()


@pytest.mark.parametrize("dims,grid", [((40, 24, 36), (2, 2, 2)), ((70, 33, 23), (2, 1, 1)), ((40, 24, 36), (1, 1, 2))])
def test_packed_transport_on_tiles_beyond_the_cache(dims, grid, monkeypatch):
    """The RCCL / callback transports on tiles that do NOT fit the Infinity Cache (N = 2 and N = 4 of the 256^3 split): the plain march with
    streaming hints -- since round 5 the instantiation whose psi load / store carry the hint for real (buffer instructions, NTBUF), with
    thin boxes in the launch -- and the z faces in place.  Forced on small grids with SOBFU_CACHE_CELLS=0; bit for bit against the
    single-GPU solve."""
    import torch

    import oracle
    from sobfu_amd import ops

    rng = np.random.default_rng(29)
    X, Y, Z = dims
    pg = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    pn = np.stack([rng.uniform(-1, 1, (Z, Y, X)), rng.integers(0, 3, (Z, Y, X))], -1).astype(np.float32)
    psi0 = oracle.new_field(dims)
    oracle.init_identity(psi0)
    psi0[..., :3] += rng.uniform(-0.7, 0.7, psi0[..., :3].shape).astype(np.float32)
    sv = ops.Solver(dims, max_iter=6, alpha=0.05, w_reg=0.4, max_update_norm=1e-10)
    psi, pnp = torch.from_numpy(psi0.copy()).cuda(), ops.new_volume(dims)
    rep, hist = sv.iterate(torch.from_numpy(pg).cuda(), torch.from_numpy(pn).cuda(), pnp, psi, 6)
    sv.close()
    monkeypatch.setenv("SOBFU_CACHE_CELLS", "0")
    out, (psi_t, pnp_t) = run_world(dims, grid, psi0, pg, pn, 6, 1e-10)
    for done, h, _, _ in out:
        assert done == rep.iterations and np.array_equal(np.asarray(h, np.float32).view(np.uint32), np.asarray(hist, np.float32).view(np.uint32))
    assert np.array_equal(psi_t[..., :3].view(np.uint32), psi.cpu().numpy()[..., :3].view(np.uint32))
    assert np.array_equal(pnp_t.view(np.uint32), pnp.cpu().numpy().view(np.uint32))
