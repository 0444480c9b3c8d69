#!/usr/bin/env python
"""bench.py -- solver iterations/s of the SobolevFusion inner loop on a 256^3 grid (BASELINE.json metric).

One "step" = one solver iteration (one pass of the `while` body at reference src/sobfu/cuda/solver.cu:114-193 with
verbosity 0) over the 256^3 roofline config (BASELINE.json configs[2]: params_boxing.ini solver values, dims
overridden to 256, two analytic spheres 1.3 voxels apart).  Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 50 --warmup 5

Prints ONE JSON line (rank 0).  Adds `roofline` for the dominant kernel (pass B: Sobolev smoothing + psi update +
warp + max-norm, 64 algorithmic B/voxel) measured with HIP events on the solver's stream, and `cpu_baseline`
(the oracle's OpenMP port of the same iteration timed on the host cores, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only does dmabuf IPC (RCCL across processes needs it)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
B_PASS_B = 64           # nabla_U r16 + psi r16 w16 + phi_n gather 8 + phi_n o psi w8
B_PASS_A = 48           # phi_n o psi r8 + phi_global r8 + psi r16 + nabla_U w16
B_ITER = B_PASS_A + B_PASS_B


def boxing_params(dim):
    """params/params_boxing.ini with VOL_DIMS overridden (SURVEY 8(d) input 3)."""
    size = np.float32(0.75)
    vs = np.array([size / np.float32(dim)] * 3, np.float32)
    return dict(dims=(dim, dim, dim), vs=vs, trunc=np.float32(48) * vs[0], eta=np.float32(3) * vs[0], alpha=0.001,
                w_reg=0.6, s=7, lam=0.1, max_update_norm=1e-10)


def sphere_pair(P, shift_vox=1.3):
    c = 0.375
    r = 0.2
    return (c, c, c), (c + shift_vox * float(P["vs"][0]), c, c), r


def cpu_baseline(P, budget_s=15.0):
    """Oracle (OpenMP port, kind='port') timed on the host cores on a bounded sample of the same workload."""
    import oracle as O

    O.build()
    dims = P["dims"]
    c0, c1, r = sphere_pair(P)
    pg, pn = O.new_volume(dims), O.new_volume(dims)
    O.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r)
    O.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
    psi = O.new_field(dims)
    O.init_identity(psi)
    kw = dict(alpha=P["alpha"], w_reg=P["w_reg"], max_update_norm=-1.0, compute_jacobian=False, inverse_iters=0)
    t0 = time.perf_counter()
    O.estimate_psi(pg, pn, psi, max_iter=1, **kw)  # warm-up iteration (also sizes the sample)
    t1 = time.perf_counter() - t0
    n = int(max(2, min(20, budget_s / max(t1, 1e-3))))
    t0 = time.perf_counter()
    O.estimate_psi(pg, pn, psi, max_iter=n, **kw)
    dt = time.perf_counter() - t0
    # estimate_psi also runs one extra apply + identity init per call; negligible next to n iterations
    return {"value": n / dt, "unit": "iterations/s", "cores": O.num_threads(), "kind": "port",
            "sample": f"{n} solver iterations of the same {dims[0]}^3 workload after 1 warm-up iteration "
                      f"(oracle/sobfu_oracle.c, OpenMP over all host cores, -O3 -ffp-contract=off, Jacobian pass skipped)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--dim", type=int, default=256, help="grid edge (256 = the BASELINE metric's grid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: every rank solves its OWN grid (independent sequences, BASELINE config 5 style: no exchange, weak "
                         "scaling) instead of the default -- ONE grid cut into N z-slabs with RCCL halo exchange (strong scaling)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank != 0:  # only rank 0 owns stdout: libraries (RCCL's version banner) write there from every process, some at exit
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(minutes=5))

    from sobfu_amd import ops

    P = boxing_params(args.dim)
    force_tiled = os.environ.get("SOBFU_FORCE_TILED") == "1"  # exercise the slab path on one GPU (debugging)
    if (world > 1 and not args.replicas) or force_tiled:
        if world == 1 and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
        from sobfu_amd import tiled

        res = tiled.bench_tiled(P, args.steps, args.warmup, rank, world)
    else:
        dims = P["dims"]
        N = dims[0] * dims[1] * dims[2]
        c0, c1, r = sphere_pair(P)
        pg, pn, pnp = ops.new_volume(dims), ops.new_volume(dims), ops.new_volume(dims)
        ops.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r)
        ops.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
        psi = ops.new_field(dims)
        ops.init_identity(psi)
        sv = ops.Solver(dims, max_iter=max(args.steps, args.warmup, 1), alpha=P["alpha"], w_reg=P["w_reg"], s=P["s"],
                        lam=P["lam"], max_update_norm=P["max_update_norm"])
        if args.warmup > 0:
            sv.iterate(pg, pn, pnp, psi, args.warmup)
        sv.set_profiling(True)
        sv.get_profile(reset=True)
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        t0 = time.perf_counter()
        rep, hist = sv.iterate(pg, pn, pnp, psi, args.steps)
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        dt = time.perf_counter() - t0
        assert rep.iterations == args.steps, (rep.iterations, args.steps)
        assert np.isfinite(hist).all() and float(hist.max()) > 0
        ms_a, ms_b, n = sv.get_profile()
        res = dict(seconds=dt, N=N, ms_a=ms_a / max(n, 1), ms_b=ms_b / max(n, 1), last_norm=float(hist[-1]),
                   workspace=sv.workspace_bytes(),
                   parallelism="single" if world == 1 else f"{world} independent {args.dim}^3 sequences, one per GPU (replicas, no exchange)")
        sv.close()

    if dist.is_initialized():
        t = torch.tensor([res["seconds"]], dtype=torch.float64, device="cuda")
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        res["seconds"] = float(t.item())

    if rank == 0:
        replicas = world > 1 and args.replicas
        its = (world if replicas else 1) * args.steps / res["seconds"]
        N = res["N"]
        ach_b = N * B_PASS_B / (res["ms_b"] * 1e-3) / 1e9 if res.get("ms_b") else None
        pmc = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc_path):
            with open(pmc_path) as f:
                pmc = json.load(f).get("pass_b_hbm_bytes_per_launch")
        out = {
            "metric": f"solver iterations/sec on {args.dim}^3 voxel grid",
            "value": its, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * res["seconds"] / args.steps, "higher_is_better": True,
            "scaling": "strong" if (world > 1 and not replicas) else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.dim}^3 TSDF, params_boxing.ini solver values (alpha 0.001, w_reg 0.6, S=7, "
                                   f"lambda 0.1, max_update_norm 1e-10), two analytic spheres 1.3 voxels apart, "
                                   f"{args.steps} solver iterations per solve",
                       "grid": [args.dim] * 3, "parallelism": res["parallelism"]},
            "iteration_hbm_frac": (N * B_ITER * its / 1e9) / (HBM_PEAK_GBPS * world),
            "iteration_GBps": N * B_ITER * its / 1e9,
            "roofline": {"kernel": "fused_smooth_update_apply_kernel (pass B: sum of three 1-D Sobolev convolutions + psi "
                                   "update + phi_n o psi warp + max-norm)",
                         "bound": "hbm", "achieved": ach_b, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": (ach_b / HBM_PEAK_GBPS) if ach_b else None, "traffic": pmc,
                         "algorithmic_bytes_per_launch": N * B_PASS_B, "avg_launch_ms": res.get("ms_b"),
                         # transparency: the solver iterates on a compact copy of the state (12-byte psi / nabla_U, tsdf-only
                         # TSDF streams), so the bytes it must move are below the survey's algorithmic figure
                         "compact_format_bytes_per_launch": N * 44,
                         "compact_format_GBps": (N * 44 / (res["ms_b"] * 1e-3) / 1e9) if res.get("ms_b") else None,
                         "traffic_GBps": (pmc / (res["ms_b"] * 1e-3) / 1e9) if (pmc and res.get("ms_b")) else None,
                         "pass_a_avg_launch_ms": res.get("ms_a"),
                         "pass_a_GBps": (N * B_PASS_A / (res["ms_a"] * 1e-3) / 1e9) if res.get("ms_a") else None},
            "last_max_update_norm": res.get("last_norm"),
            "solver_workspace_bytes": res.get("workspace"),
        }
        if res.get("tiled_autotune_us"):
            out["tiled_autotune_us"] = res["tiled_autotune_us"]
        if res.get("tiled_diag"):  # N > 1: what the pieces of the native loop cost on this machine (outside the timed region)
            out["tiled_diag"] = res["tiled_diag"]
        if res.get("tiled_parity") is not None:  # N > 1: every rank re-ran the whole solve alone and compared its slab bitwise
            out["tiled_parity_vs_single_gpu"] = "bit-exact" if res["tiled_parity"] else "MISMATCH"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(P)
    if res.get("diag_hung"):  # a diagnostics collective never returned on this rank: report what was measured and leave
        if rank == 0:
            import ctypes

            ctypes.CDLL(None).fflush(None)
            print(json.dumps(out), flush=True)
        os._exit(0)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:  # after the process group is gone, and after flushing C stdio (RCCL's version banner sits in libc's stdout
        # buffer until exit when stdout is a pipe), so that the JSON is the LAST line on stdout
        import ctypes

        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
        # nothing may follow the line on stdout (RCCL writes banners from destructors / at exit): point fd 1 at /dev/null for the
        # rest of the process instead of killing it -- exit hooks (rocprofv3 writing its results) must still run
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)


if __name__ == "__main__":
    main()
