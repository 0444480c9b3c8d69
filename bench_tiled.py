"""bench.py's multi-GPU leg (`--gpus N` without --replicas): the SAME 256^3 solve cut into N tiles (strong scaling; 2 x 2 x 2 at N = 8).

Harness, not product (moved out of sobfu_amd/tiled.py in round 4): probing the direct transport in child processes, the bitwise
precheck, timing every tile grid, per-piece diagnostics, the gloo bring-up transport for ranks that share a GPU.  The product side --
TileLayout, NativeTiledSolver, TiledFusion -- is sobfu_amd/tiled.py.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from sobfu_amd.tiled import HALO, SLOTS, DistHalo, NativeTiledSolver, TileLayout, parse_grid

ROOT = os.path.dirname(os.path.abspath(__file__))


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


def _sphere_volumes(P):
    """the bench workload's two analytic TSDFs (every rank builds the full volumes: phi_n is replicated, phi_global is cut to the tile)"""
    from sobfu_amd import ops

    dims = P["dims"]
    c0, c1, r = (0.375,) * 3, (0.375 + 1.3 * float(P["vs"][0]), 0.375, 0.375), 0.2
    pg_full, pn_full = ops.new_volume(dims), ops.new_volume(dims)
    ops.init_sphere(pg_full, P["vs"], P["trunc"], P["eta"], c0, r)
    ops.init_sphere(pn_full, P["vs"], P["trunc"], P["eta"], c1, r)
    return pg_full, pn_full


def _timed(ranks, fn, n):
    """microseconds per fn() over n calls, barrier + synchronize around the lot, MAX over ranks"""
    fn()
    torch.cuda.synchronize()
    ranks.barrier()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return ranks.max([(time.perf_counter() - t0) / n * 1e6])[0]


def candidate_grids(world, dims):
    """every (Px, Py, Pz) with Px * Py * Pz == world and Px <= Py <= Pz (x, the axis the 64 lanes of a wave run along, is split last)
    whose tiles keep >= HALO cells per split axis: 1 x 1 x 8, 1 x 2 x 4 and 2 x 2 x 2 at 8"""
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
    "rccl" is the same packed buffers over gloo (GlooTransport).  Returns (solver, keep-alive)."""
    if transport == "direct":
        return NativeTiledSolver(dims, grid=grid, transport="direct", **kw), None
    if ranks.share and ranks.world > 1:
        sv = NativeTiledSolver(dims, dry=(ranks.world, ranks.rank), grid=grid, **kw)
        tr = GlooTransport()
        sv.set_transport(tr.exchange, tr.allreduce)
        return sv, tr
    return NativeTiledSolver(dims, grid=grid, **kw), None


def _single_gpu_reference(P, kw, pg_full, pn_full, iters):
    from sobfu_amd import ops

    dims = P["dims"]
    one = ops.Solver(dims, max_iter=max(iters, 1), **kw)
    psi_f, pnp_f = ops.new_field(dims), ops.new_volume(dims)
    ops.init_identity(psi_f)
    _, norms = one.iterate(pg_full, pn_full, pnp_f, psi_f, iters)
    one.close()
    return psi_f, pnp_f, norms


def _tile_equals(L, psi, pnp, norms, psi_f, pnp_f, norms_f):
    """(same, description of the difference): this rank's owned cells and the max-norm history against the single-GPU solve, bit for bit"""
    n32 = lambda v: np.asarray(v, np.float32).view(np.uint32)  # noqa: E731
    dn = int((n32(norms_f) != n32(norms)).sum()) if len(norms) == len(norms_f) else -1
    dp = int((L.owned_global(psi_f)[..., :3].contiguous().view(torch.int32) != L.owned(psi)[..., :3].contiguous().view(torch.int32)).sum())
    df = int((L.owned_global(pnp_f).contiguous().view(torch.int32) != L.owned(pnp).contiguous().view(torch.int32)).sum())
    return dn == 0 and dp == 0 and df == 0, f"{dn} of {len(norms)} max-norms, {dp} psi words, {df} phi_n o psi words differ"


def direct_transport_precheck(P, ranks, kw, grid, iters=6):
    """The direct transport on THIS machine, before anything is timed: a few iterations of the bench workload on tiles against the
    single-GPU solver, bit for bit on every rank (peer mapping, flags and deadline all exercised).  Returns None when every rank
    agrees it works, else a reason string (collective: every rank gets the same verdict)."""
    dims = P["dims"]
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
        pg_full, pn_full = _sphere_volumes(P)
        L = sv.layout
        pg = L.take(pg_full).clone().contiguous()
        pnp, psi = sv.new_local(2), sv.identity_psi()
        done, norms = sv.iterate(pg, pn_full, pnp, psi, iters)
        psi_f, pnp_f, norms_one = _single_gpu_reference(P, kw, pg_full, pn_full, iters)
        same, what = _tile_equals(L, psi, pnp, norms, psi_f, pnp_f, norms_one)
        if not same:
            why = (f"tiles differ from the single-GPU solve (rank {ranks.rank}: {what}; iterations done {done}; "
                   f"norms {[float(v) for v in norms]} vs {[float(v) for v in norms_one]})")
    except Exception as e:  # noqa: BLE001 -- e.g. SOBFU_E_TIMEOUT: a peer's flag did not arrive
        why = f"{e!r}"
    ok = ranks.min([0 if why else 1])[0]
    sv.close()
    return None if ok else (why or "failed on another rank")


def direct_transport_sandbox(P, ranks, kw, grid, timeout=240):
    """The same check as direct_transport_precheck, one step earlier and somewhere safer: in a CHILD process of every rank
    (bench_probe.py).  A transport that stores into other GPUs' memory from inside a kernel fails, when the mapping is not
    what it looks like, with a GPU memory fault -- which kills the process that launched the kernel.  The children take that risk;
    the ranks themselves only learn the verdict.  Returns None when every rank's child exited 0, else a reason (collective)."""
    import json
    import socket
    import subprocess

    if os.environ.get("SOBFU_TILED_SANDBOX", "1") != "1":
        return None
    # the children's rendezvous port: derived from the launcher's own (MASTER_PORT + 1 ...), tried for real by rank 0 -- a port picked
    # by bind(0) + close can be taken by another process before the children bind it
    port = 0
    if ranks.rank == 0:
        base = int(os.environ.get("MASTER_PORT", "29500"))
        for cand in [base + 1 + k for k in range(40)] + [0]:
            with socket.socket() as sk:
                try:
                    sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                    sk.bind(("", cand))
                    port = sk.getsockname()[1]
                    break
                except OSError:
                    continue
    port = int(ranks.max([port])[0])
    args = dict(addr=os.environ.get("MASTER_ADDR", "127.0.0.1"), port=port, grid=list(grid),
                dims=list(P["dims"]), vs=[float(v) for v in P["vs"]], trunc=float(P["trunc"]), eta=float(P["eta"]), kw=kw, iters=4, timeout=int(os.environ.get("SOBFU_PROBE_TIMEOUT_S", "90")))
    # the children rendezvous among themselves: without the launcher's agent store (TORCHELASTIC_USE_AGENT_STORE would make rank 0's
    # child a client of a store nobody serves on that port)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}
    env["SOBFU_PROBE_ARGS"] = json.dumps(args)
    ok, why = False, None
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_probe.py")], env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
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


def direct_transport_usable(P, ranks, kw, grid, left=float("inf")):
    """sandboxed probe in child processes, then the bitwise precheck in the ranks themselves (repeated once); None = usable.
    left: seconds of the run's budget (the probe may take a third of it, at most 240 s)"""
    probe_s = 240 if left == float("inf") else int(ranks.min([max(30.0, min(240.0, left / 3.0))])[0])
    why = direct_transport_sandbox(P, ranks, kw, grid, timeout=probe_s)  # first in child processes (a GPU fault there costs nothing) ...
    if why is None:
        why = direct_transport_precheck(P, ranks, kw, grid)  # ... then in this one
        if why is not None:
            # seen twice in ~35 multi-process start-ups right behind the children's exit (one refused export, one mismatch), never
            # in 360 start-ups without children: a second attempt, on fresh state, before the transport is given up
            print(f"[rank {ranks.rank}] direct transport precheck failed ({why}); trying once more", file=sys.stderr, flush=True)
            time.sleep(1.0)
            first, why = why, direct_transport_precheck(P, ranks, kw, grid)
            if why is not None:
                why = f"{why} (first attempt: {first})"
    return why


def topology_snapshot(ranks):
    """what the runtime reports about the paths between the node's GPUs (rank 0's view): hipDeviceCanAccessPeer / link type / hops /
    performance rank per ordered pair (sobfu_hip_p2p_info) and, when rocm-smi is there, the text of `rocm-smi --showtopo`"""
    import shutil
    import subprocess

    from sobfu_amd import _lib

    out = {"devices_visible": int(torch.cuda.device_count()), "link_type_names": {"0": "HyperTransport", "1": "QPI", "2": "PCIe", "3": "InfiniBand", "4": "xGMI"}}
    try:
        lib, n, pairs = _lib.lib(), min(int(torch.cuda.device_count()), 16), {}
        for a in range(n):
            for b in range(n):
                if a != b:
                    info = (C.c_int * 4)()
                    _lib.check(lib.sobfu_hip_p2p_info(C.c_int(a), C.c_int(b), info), "p2p_info")
                    pairs[f"{a}->{b}"] = dict(can_access=info[0], link_type=info[1], hops=info[2], perf_rank=info[3])
        out["pairs"] = pairs
    except Exception as e:  # noqa: BLE001
        out["pairs_error"] = repr(e)
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    if os.path.exists(exe):
        try:
            r = subprocess.run([exe, "--showtopo"], capture_output=True, text=True, timeout=30)
            out["rocm_smi_showtopo"] = [ln.rstrip() for ln in r.stdout.splitlines() if ln.strip()][:120]
        except Exception as e:  # noqa: BLE001
            out["rocm_smi_error"] = repr(e)
    return out


def time_grids(P, ranks, kw, transport, iters=40):
    """us per iteration of the native loop for every candidate tile grid on the machine at hand with the real exchange (MAX over
    ranks: all agree); a grid that fails on some rank is reported as null"""
    dims = P["dims"]
    pg_full, pn_full = _sphere_volumes(P)
    times = {}
    for grid in candidate_grids(ranks.world, dims):
        sv, t = None, None
        try:
            sv, _keep = make_native_solver(dims, grid, ranks, kw, transport)
        except Exception as e:  # noqa: BLE001
            print(f"[rank {ranks.rank}] grid {grid} on {transport}: {e!r}", file=sys.stderr, flush=True)
        if ranks.min([1 if sv is not None else 0])[0] == 1:
            pg = sv.layout.take(pg_full).clone().contiguous()
            pnp, psi = sv.new_local(2), sv.identity_psi()
            try:
                sv.iterate(pg, pn_full, pnp, psi, 4)
                torch.cuda.synchronize()
                ranks.barrier()
                t0 = time.perf_counter()
                sv.iterate(pg, pn_full, pnp, psi, iters)
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / iters * 1e6
            except Exception as e:  # noqa: BLE001
                print(f"[rank {ranks.rank}] grid {grid} on {transport}: {e!r}", file=sys.stderr, flush=True)
            ok = ranks.min([1 if t is not None else 0])[0] == 1
            worst = ranks.max([t if t is not None else 0.0])[0]
            times["x".join(map(str, grid))] = round(worst, 2) if ok else None
        else:
            times["x".join(map(str, grid))] = None
        if sv is not None:
            sv.close()
    return times


def direct_micro_diagnostics(solver, ranks, reps=400):
    """The direct transport piece by piece on THIS machine (collective, outside a solve): flag round trips between every pair of ranks,
    pass A's push boxes alone with the production store path (-> the rate the halo faces leave at), against the same launches storing
    into local memory."""
    L, out = solver.layout, {}
    world, rank = ranks.world, ranks.rank
    rt = {}
    for a in range(world):
        for b in range(a + 1, world):
            solver.pingpong(a, b, 8)  # warm
            torch.cuda.synchronize()
            ranks.barrier()
            t0 = time.perf_counter()
            solver.pingpong(a, b, reps)
            torch.cuda.synchronize()
            mine = (time.perf_counter() - t0) / reps * 1e6 if rank in (a, b) else 0.0
            rt[f"{a}<->{b}"] = round(ranks.max([mine])[0], 2)
    out["flag_round_trip_us"] = rt  # one kernel of `reps` round trips per pair, host-timed incl. one launch: an upper bound by launch / reps
    msgs = L.messages()
    cells = [(m[1][1] - m[1][0]) * (m[1][3] - m[1][2]) * (m[1][5] - m[1][4]) for m in msgs]
    out["messages"] = len(msgs)
    out["bytes_out_per_iteration"] = 12 * sum(cells)
    out["largest_message_bytes"] = 12 * max(cells or [0])
    if msgs:
        push_us = _timed(ranks, lambda: solver.probe_push(10), 5) / 10
        dry = NativeTiledSolver(L.dims, dry=(world, rank), grid=L.grid, alpha=0.1, w_reg=0.1)  # same boxes, stores into local memory, no peers
        local_us = _timed(ranks, lambda: dry.probe_push(10), 5) / 10
        dry.close()
        out["push_boxes_only_us"] = round(push_us, 2)             # rim cells evaluated + stored into the PEERS' halo cells + handshake
        out["push_boxes_only_local_stores_us"] = round(local_us, 2)  # the same launches storing locally (no peers, no handshake)
        out["push_rate_GBps_of_bytes_out"] = round(12 * sum(cells) / (push_us * 1e-6) / 1e9, 3)
    ok, missing = solver.status()
    out["status_ok"] = bool(ok)
    return out


def rccl_micro_diagnostics(solver, ranks, reps=30):
    """the RCCL leg's pieces: one halo exchange (packed send / recv + scatter; z-slabs: planes in place) and one 256-slot all-reduce"""
    L, lib, check = solver.layout, solver._lib.lib(), solver._lib.check
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    field = torch.zeros(L.local_shape(3), dtype=torch.float32, device="cuda")
    for planes in ((1, 2, HALO) if L.slab else (HALO,)):  # latency vs bandwidth of a face message (z-slabs: 1/4, 1/2 and all of the halo)
        out[f"exchange_{planes}_cells_us"] = round(_timed(ranks, lambda: check(lib.sobfu_hip_tiled_exchange(solver._h, C.c_void_p(field.data_ptr()), C.c_int(planes), st), "exchange"), reps), 2)
    if solver.has_comm:
        slots = torch.zeros(SLOTS, dtype=torch.int32, device="cuda")
        out["allreduce_256_slots_us"] = round(_timed(ranks, lambda: check(lib.sobfu_hip_tiled_allreduce_max_u32(solver._h, C.c_void_p(slots.data_ptr()), C.c_size_t(SLOTS), st), "allreduce"), reps), 2)
    return out


def frames_on_tiles(args, ranks, grid, transport, max_frames):
    """frames/s of the WHOLE per-frame pipeline on tiles (sobfu_amd.tiled.TiledFusion = SobFusion::operator(), reference
    src/sobfu/sob_fusion.cpp:71-145, with the volume cut into tiles): the tiled solve plus the two per-frame collectives that do
    not shrink with N -- all-gather psi before the 48-sweep inverse, all-gather phi_global for the canonical -> live warp
    (reference src/sobfu/cuda/solver.cu:196-199).  One synthetic sequence; frame 0 (initialisation) untimed."""
    import bench
    from sobfu_amd import params, synthetic
    from sobfu_amd.tiled import TiledFusion, gather_owned

    ini, dim0 = bench.FRAME_CONFIGS[args.frame_config]
    dim = args.dim if args.frame_dim <= 0 else args.frame_dim
    P = params.read_ini(os.path.join(ROOT, ini), dims=(dim if dim != dim0 else None))
    size, tz, vx = float(P["size"][0]), float(P["t"][2]), float(P["vs"][0])
    depth = [torch.from_numpy(synthetic.render_sphere_depth((1.3 * vx * n, 0.0, tz + 0.5 * size), 0.2 * size, P["intr"])).cuda() for n in range(max_frames)]
    kw = dict(alpha=P["alpha"], w_reg=P["w_reg"], s=P["s"], lam=P["lam"], max_update_norm=P["max_update_norm"])
    solver, keep = make_native_solver(P["dims"], grid, ranks, kw, transport)
    L = solver.layout

    def gather_cpu(local):  # ranks that share a GPU run their process group on gloo: staged through the host
        own = L.owned(local).contiguous().cpu()
        parts = [None] * ranks.world
        dist.all_gather_object(parts, own)
        X, Y, Z = L.dims
        full = torch.empty((Z, Y, X) + tuple(own.shape[3:]), dtype=own.dtype, device="cuda")
        for q, part in enumerate(parts):
            TileLayout(L.dims, L.grid, q).owned_global(full).copy_(part.cuda())
        return full

    fP = dict(P, max_iter=args.frame_iters)
    # the per-frame tail fetches bounded-reach windows of psi / phi_global (DistHalo; all-gather only when the displacement outgrows a
    # tile); SOBFU_TILED_TAIL=gather forces the two all-gathers of round 4 (A/B)
    force_gather = os.environ.get("SOBFU_TILED_TAIL", "halo") == "gather"
    gather_fn = gather_cpu if ranks.share else solver.gather_owned
    halo = None if (force_gather or ranks.world == 1) else DistHalo(L, via_host=ranks.share)
    fu = TiledFusion(solver, fP, gather=gather_fn, halo=halo)
    tails = []
    fu(depth[0])
    torch.cuda.synchronize()
    ranks.barrier()
    ms, iters = [], []
    for n in range(1, max_frames):
        torch.cuda.synchronize()
        ranks.barrier()
        t0 = time.perf_counter()
        rep = fu(depth[n])
        torch.cuda.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
        iters.append(int(rep[0]) if rep is not None else 0)
        if rep is not None and solver.tail_stats:
            tails.append(dict(solver.tail_stats))
    # self-check of the bounded-reach tail on the state the last frame left: the same tail on all-gathered sources, bit for bit on
    # every rank's owned cells (MIN over ranks)
    tail_ok = None
    if halo is not None and tails:
        from sobfu_amd.tiled import frame_tail

        inv_a, pgi_a, inv_b, pgi_b = solver.new_local(4), solver.new_local(2), solver.new_local(4), solver.new_local(2)
        frame_tail(solver, fu.phi_global, pgi_a, fu.psi, inv_a, gather=gather_fn, halo=halo)
        mode_a = solver.tail_stats["mode"]
        frame_tail(solver, fu.phi_global, pgi_b, fu.psi, inv_b, gather=gather_fn, halo=None)
        same = torch.equal(L.owned(inv_a)[..., :3].contiguous().view(torch.int32), L.owned(inv_b)[..., :3].contiguous().view(torch.int32)) and \
            torch.equal(L.owned(pgi_a).contiguous().view(torch.int32), L.owned(pgi_b).contiguous().view(torch.int32))
        tail_ok = ranks.min([1.0 if same else 0.0])[0] == 1.0  # the BITS decide; which path produced them (halo windows, or the all-gather
        # fallback when the displacement outgrew a tile) is reported under tail.mode
    solver.close()
    worst = ranks.max(ms)
    timed = [w for w, i in zip(worst, iters) if i > 0] or worst  # frames before START_FRAME only fuse
    med = sorted(timed)[len(timed) // 2]
    all_gather_bytes = P["dims"][0] * P["dims"][1] * P["dims"][2] * (16 + 8)  # psi (float4) + phi_global (float2) all-gathered: what round 4 moved per frame
    recv = int(ranks.max([float(np.median([t["bytes_received"] for t in tails])) if tails else 0.0])[0])
    return {"config": f"{ini} values, {P['dims'][0]}^3, {args.frame_iters} solver iterations per frame, synthetic 640x480 depth sequence",
            "pipeline": "bilateral + truncation + ray lengths -> integrate(depth) (own tile of phi_global, whole phi_n) -> tiled estimate_psi: iterations, "
                        "one MAX reduction of |psi - id|, bounded-reach windows of psi and phi_global from the neighbours (all-gather only when the "
                        "displacement outgrows a tile), 48-sweep inverse, canonical->live warp -> fuse   [reference src/sobfu/sob_fusion.cpp:71-145]",
            "transport": transport, "tiles": "x".join(map(str, grid)), "frames_timed": len(timed), "ms_per_frame": med,
            "ms_per_frame_all": [round(v, 3) for v in worst], "frames_per_s": 1e3 / med, "iterations_per_frame": iters,
            "tail": {"mode": sorted({t["mode"] for t in tails}) if tails else None, "halo_width_cells": sorted({t["halo_width"] for t in tails if t["halo_width"]}) or None,
                     "max_displacement_voxels": max([t["reach"] for t in tails if t["reach"] is not None], default=None),
                     "bytes_received_per_frame_per_rank": recv, "all_gather_would_move_bytes": all_gather_bytes,
                     "parity_vs_all_gather_tail": None if tail_ok is None else ("bit-exact" if tail_ok else "MISMATCH")},
            "all_gathered_bytes_per_frame": recv}


def run_leg(args, P, ranks, timed_regions, grid, transport_name, kw):
    """One timed run of the tile loop on one transport + the bitwise self-check of the whole run.  Returns a dict; {"failed": reason}
    when the transport broke (collective: every rank returns the same kind)."""
    rank, world = ranks.rank, ranks.world
    dims = P["dims"]
    K, W, R, PR = args.steps, args.warmup, args.repeats, args.profile_repeats
    total = W + (R + PR) * K
    solver, keep, why = None, None, None
    try:
        solver, keep = make_native_solver(dims, grid, ranks, kw, transport_name)
        if total > solver.max_iterations():
            why = (f"a solve of {total} iterations (warm-up + regions) exceeds the {solver.max_iterations()} iterations the direct transport's "
                   "peer-mapped max-norm rows hold")
    except Exception as e:  # noqa: BLE001
        why = f"setup failed: {e!r}"
    if ranks.min([0 if why else 1])[0] == 0:
        if solver is not None:
            solver.close()
        return {"failed": why or "setup failed on another rank"}
    L = solver.layout
    pg_full, pn_full = _sphere_volumes(P)
    pg = L.take(pg_full).clone().contiguous()
    pnp, psi = solver.new_local(2), solver.identity_psi()
    tuned = None
    if transport_name == "rccl" and keep is None and L.slab and world > 1 and os.environ.get("SOBFU_TILED_AUTOTUNE", "1") == "1":
        tuned = solver.autotune(pg, pn_full)  # z-slabs: serial vs overlapped exchange, outside the timed region
    broke, secs, prof, norms, wait = None, None, None, None, None
    try:
        solver.begin(pg, pn_full, pnp, psi, total)  # the solve is open and its state resident before anything is timed
        solver.step(W)
        if transport_name == "direct":
            torch.cuda.synchronize()
            solver.wait_stats(reset=True)
        secs = timed_regions(ranks, torch, lambda: solver.step(K), R)
        if transport_name == "direct":
            wait = solver.wait_stats(reset=True)
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
    except Exception as e:  # noqa: BLE001 -- e.g. SOBFU_E_TIMEOUT: a peer's flag did not arrive within the deadline
        broke = repr(e)
    if ranks.max([1 if broke else 0])[0] > 0:  # every rank leaves the transport together
        solver.close()
        return {"failed": broke or "the transport broke on another rank"}
    # self-check, outside the timed region: every rank repeats the WHOLE solve on its own GPU with the single-GPU solver
    # handle and compares its owned cells and the max-norm history bit for bit (tiling must not change a single bit)
    parity = None
    if os.environ.get("SOBFU_TILED_SELFCHECK", "1") == "1":
        psi_f, pnp_f, norms_one = _single_gpu_reference(P, kw, pg_full, pn_full, total)
        same, what = _tile_equals(L, psi, pnp, norms, psi_f, pnp_f, norms_one)
        parity = bool(ranks.min([1 if same else 0])[0])
        if not same:
            print(f"[rank {rank}] tiled self-check on {transport_name}: {what}", file=sys.stderr, flush=True)
        del psi_f, pnp_f
    own = tuple(L.g1[a] - L.g0[a] for a in range(3))
    s = sorted(secs)
    med = s[len(s) // 2] if len(s) % 2 else 0.5 * (s[len(s) // 2 - 1] + s[len(s) // 2])
    leg = dict(transport=transport_name, value=K / med, ms_per_step=1e3 * med / K, region_seconds=secs, parity=parity,
               ms_a=(prof[0] if prof else None), ms_exchange=(prof[1] if prof else None), ms_b=(prof[2] if prof else None),
               n_prof=(prof[3] if prof else None), last_norm=float(norms[-1]), owned=own, slab=L.slab, over_gloo=keep is not None,
               schedule=(f"{solver.SCHEDULES[solver.schedule]} (autotuned)" if tuned else None),
               schedule_times_us=({solver.SCHEDULES[k]: round(v, 2) for k, v in tuned.items()} if tuned else None),
               host_enqueue_us=float(_last_enqueue_us(solver)))
    if wait is not None and wait[1] > 0:  # what of the exchange an iteration did NOT hide: the signalling workgroup's wait for its peers' flags
        leg["peer_wait_us_per_iteration"] = round(wait[0] / wait[1], 3)
    leg["_solver"], leg["_keep"], leg["_state"] = solver, keep, (pg, pn_full)
    return leg


def seconds_left(args):
    """seconds until the run's --budget-s deadline (inf: no budget)"""
    dl = getattr(args, "deadline", None)
    return float("inf") if dl is None else dl - time.time()


def agree(ranks, key, value):
    """rank 0's decision, for every rank (over the rendezvous store: a skip decided on local clocks could leave some ranks in a
    collective the others never enter)"""
    if ranks.world == 1 or not dist.is_initialized():
        return value
    store = dist.distributed_c10d._get_default_store()
    if ranks.rank == 0:
        store.set(key, "1" if value else "0")
        return value
    return store.get(key) == b"1"


def _last_enqueue_us(solver):
    lib = solver._lib.lib()
    lib.sobfu_hip_tiled_last_enqueue_us.restype = C.c_double
    return lib.sobfu_hip_tiled_last_enqueue_us(solver._h)


def bench_tiled(args, P, ranks, timed_regions):
    """bench.py leg for --gpus N > 1: the SAME 256^3 solve cut into N tiles (strong scaling; BASELINE config 4's 2 x 2 x 2 at N = 8).

    ONE run harvests everything the machine can tell:
      legs        BOTH transports are timed on config 4's grid -- "direct" (peer-mapped stores issued by pass A's own launch, after a
                  sandboxed probe and a bitwise precheck) and "rccl" (packed by pass A's launch, one grouped send / recv, one scatter
                  kernel): `value` is the better BIT-EXACT one, `legs` holds both, so the RCCL figure north_star names always exists;
      tiled_autotune_us   every grid of N tiles timed on every usable transport (is 1 x 2 x 4 faster than 2 x 2 x 2 here?);
      direct_diag / rccl_diag   flag round trips per pair of ranks, the push boxes alone and the rate the faces leave at, the part of the
                  exchange an iteration does not hide (the signalling workgroup's wait); one exchange and one all-reduce on RCCL;
      topology    link type / hops / peer access per pair of GPUs and `rocm-smi --showtopo`;
      per_frame   frames/s of the whole per-frame pipeline on tiles, incl. the two all-gathers that do not shrink with N.
    A transport that fails its probe, breaks during the run (deadline) or fails the bitwise self-check never becomes the reported
    number; a MISMATCH with no correct leg left exits non-zero."""
    rank, world = ranks.rank, ranks.world
    dims = P["dims"]
    X, Y, Z = dims
    spec = args.tiles or os.environ.get("SOBFU_TILES", "")
    kw = dict(alpha=P["alpha"], w_reg=P["w_reg"], s=P["s"], lam=P["lam"], max_update_norm=P["max_update_norm"])
    if world == 1 and not dist.is_initialized():  # SOBFU_FORCE_TILED=1 on one GPU: a world of one
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", ranks.device))
    want = os.environ.get("SOBFU_TILED_TRANSPORT", "both")  # both | direct | rccl
    grid = parse_grid("" if spec == "auto" else spec, world)
    legs, usable = {}, {}
    direct_why = None
    if want in ("both", "direct") and world > 1:
        direct_why = direct_transport_usable(P, ranks, kw, grid, seconds_left(args))
        if direct_why is not None:
            print(f"[rank {rank}] direct transport not used: {direct_why}", file=sys.stderr, flush=True)
            legs["direct"] = {"failed": direct_why}
    elif want in ("both", "direct"):
        direct_why = None  # a world of one: the tile path's launches, no peers
    grid_times = {}
    if spec == "auto":  # `--tiles auto`: the timed grid is the fastest one on the first usable transport
        first = "direct" if (want != "rccl" and direct_why is None) else "rccl"
        grid_times[first] = time_grids(P, ranks, kw, first)
        best = min((v, k) for k, v in grid_times[first].items() if v is not None)[1]
        grid = tuple(int(v) for v in best.split("x"))
    for name in (["direct"] if want in ("both", "direct") and direct_why is None else []) + (["rccl"] if want in ("both", "rccl") or direct_why is not None else []):
        if name == "rccl" and world == 1 and want == "both":
            continue  # a world of one has nothing to exchange: one leg
        leg = run_leg(args, P, ranks, timed_regions, grid, name, kw)
        if "failed" in leg:
            print(f"[rank {rank}] {name} transport abandoned: {leg['failed']}", file=sys.stderr, flush=True)
        elif leg["parity"] is False:
            leg["_solver"].close()
            leg = {"failed": "tiles differed from the single-GPU solve in the final bitwise self-check", "parity": False}
        legs[name] = leg
    good = {k: v for k, v in legs.items() if "failed" not in v}
    if not good:
        why = "; ".join(f"{k}: {v.get('failed')}" for k, v in legs.items())
        if rank == 0:
            print(f"bench_tiled: no transport produced a correct run ({why})", file=sys.stderr, flush=True)
        mismatch = any(v.get("parity") is False for v in legs.values())
        raise SystemExit(3 if mismatch else 2)
    best = max(good, key=lambda k: good[k]["value"])
    main = good[best]

    def result(extras, hung, hung_any, grid_times, skipped):
        own = main["owned"]
        what = (f"{world} z-slabs of {own[2]} planes" if main["slab"] else f"{grid[0]}x{grid[1]}x{grid[2]} tiles of {own[0]}x{own[1]}x{own[2]} cells")
        via = {"direct": "peer-mapped stores over xGMI issued by pass A's own launch (no pack / unpack, no RCCL in the loop; arrival flags "
                         "and max-norm rows travel the same way)",
               "rccl": "gloo (ranks share a GPU: bring-up transport)" if main["over_gloo"] else "RCCL send/recv (packed by pass A's launch, one scatter kernel)"}
        pub = {}
        for name, leg in legs.items():
            if "failed" in leg:
                pub[name] = {"failed": leg["failed"]}
            else:
                pub[name] = {k: v for k, v in leg.items() if not k.startswith("_") and k not in ("region_seconds", "owned", "slab")}
                pub[name]["region_its"] = [round(args.steps / s, 1) for s in leg["region_seconds"]]
                pub[name]["tiled_parity_vs_single_gpu"] = "bit-exact" if leg["parity"] else ("MISMATCH" if leg["parity"] is False else None)
        layouts = [TileLayout(dims, grid, q) for q in range(world)]
        return dict(diag_hung=hung, diag_hung_any=hung_any, transport=best, transport_fallback=direct_why, legs=pub,
                    region_seconds=main["region_seconds"], N=X * Y * Z, ms_a=main["ms_a"], ms_b=main["ms_b"], ms_exchange=main["ms_exchange"],
                    n_prof=main["n_prof"], last_norm=main["last_norm"], workspace=None, tiled_parity=main["parity"],
                    launch_cells=max((l.g1[0] - l.g0[0]) * (l.g1[1] - l.g0[1]) * (l.g1[2] - l.g0[2]) for l in layouts),
                    tiled_diag={k: v for k, v in extras.items() if k in ("direct_diag", "rccl_diag", "iteration_us_compute_only", "error")} or None,
                    topology=extras.get("topology"), per_frame=extras.get("per_frame"),
                    tiles={"grid": list(grid), "owned_cells_rank0": list(TileLayout(dims, grid, 0).g1[a] - TileLayout(dims, grid, 0).g0[a] for a in range(3)),
                           "halo": HALO, "messages_per_exchange_rank0": len(TileLayout(dims, grid, 0).messages())},
                    parallelism=f"{what} (+{HALO}-cell halos), one nabla_U halo exchange per iteration over {via[best]}, native C++ loop"
                                + (f", schedule: {main['schedule']}" if main.get("schedule") else ""),
                    tiled_autotune_us=grid_times or None, skipped=skipped or None)

    # THE CORE OF THE LINE EXISTS NOW (value, both legs, parity): hand it to bench.py before any harvest step runs -- it keeps an early
    # copy and, if the run is still busy when its budget ends, prints that copy instead of nothing
    if getattr(args, "on_core", None) is not None:
        args.on_core(result({}, False, False, dict(grid_times), ["everything after the timed legs (the run reached its --budget-s deadline)"]))
    skipped = []

    def want_step(name, estimate_s):
        """does the budget leave room for a harvest step of about estimate_s seconds?  (rank 0 decides for everybody)"""
        ok = agree(ranks, f"sobfu_step_{name}", seconds_left(args) > estimate_s + 15.0)
        if not ok:
            skipped.append(name)
        return ok
    # everything below is outside the timed regions and runs in a worker thread with a deadline: whatever happens in there (an
    # exception on one rank would leave the others waiting in a collective), the benchmark line is still printed
    extras, hung = {}, False
    if os.environ.get("SOBFU_TILED_DIAG", "1") == "1":
        import threading

        box, dev_index = {}, torch.cuda.current_device()

        def work():
            try:
                torch.cuda.set_device(dev_index)
                if os.environ.get("SOBFU_BENCH_TEST_HANG") == "1":  # test hook: a harvest step that never returns (a sick node)
                    time.sleep(3600)
                if want_step("topology", 35) and rank == 0:
                    box["topology"] = topology_snapshot(ranks)
                if "direct" in good and world > 1 and want_step("direct_diag", 30):
                    box["direct_diag"] = direct_micro_diagnostics(good["direct"]["_solver"], ranks)
                if "rccl" in good and world > 1 and want_step("rccl_diag", 20):
                    box["rccl_diag"] = rccl_micro_diagnostics(good["rccl"]["_solver"], ranks)
                # the compute side alone (same tile, no peers) next to the measured iteration
                if want_step("iteration_us_compute_only", 10):
                    pg, pn_full = main["_state"]
                    dry = NativeTiledSolver(dims, dry=(world, rank), grid=grid, **kw)
                    pnp, psi = dry.new_local(2), dry.identity_psi()
                    box["iteration_us_compute_only"] = round(_timed(ranks, lambda: dry.iterate(pg, pn_full, pnp, psi, 60), 2) / 60, 2)
                    dry.close()
                for leg in good.values():  # the timed handles are done: their memory goes before the grid sweep
                    leg["_solver"].close()
                frames = args.frames if args.frames >= 2 else 0
                if frames and want_step("per_frame", 45):  # (before the grid sweep: frames/s on tiles is the scarcer number)
                    box["per_frame"] = frames_on_tiles(args, ranks, grid, best, frames)
                if world > 1:
                    for name in good:
                        if name not in grid_times and want_step(f"tiled_autotune_us.{name}", 60):
                            box.setdefault("grids", {})[name] = time_grids(P, ranks, kw, name)
            except Exception as e:  # noqa: BLE001
                box["error"] = repr(e)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(timeout=max(5.0, min(float(os.environ.get("SOBFU_TILED_DIAG_TIMEOUT", "420")), seconds_left(args) - 10.0)))
        hung = th.is_alive()
        extras = dict(box)
        if hung or "error" in box:
            extras["error"] = "timed out" if hung else box["error"]
            print(f"[rank {rank}] tiled diagnostics: {extras['error']}", file=sys.stderr, flush=True)
    # whether ANY rank hung is agreed over the rendezvous store, not over the (possibly wedged) communicator: every rank then
    # takes the same exit path (no rank waits in a barrier the hung rank never reaches)
    hung_any = hung
    if world > 1 and os.environ.get("SOBFU_TILED_DIAG", "1") == "1":
        try:
            store = dist.distributed_c10d._get_default_store()
            store.set(f"sobfu_diag_hung_{rank}", "1" if hung else "0")
            hung_any = any(store.get(f"sobfu_diag_hung_{q}") == b"1" for q in range(world))
        except Exception as e:  # noqa: BLE001
            print(f"[rank {rank}] could not agree on the diagnostics verdict: {e!r}", file=sys.stderr, flush=True)
            hung_any = True
    for name, t in (extras.get("grids") or {}).items():
        grid_times[name] = t
    if hung:
        skipped.append("harvest steps still running when the diagnostics deadline / the run's budget ended")
    return result(extras, hung, hung_any, grid_times, skipped)
