#!/usr/bin/env python
"""bench.py -- solver iterations/s of the SobolevFusion inner loop on a 256^3 grid (BASELINE.json metric).

One "step" = one solver iteration (one pass of the `while` body at reference src/sobfu/cuda/solver.cu:114-193 with
verbosity 0: pass A + pass B, including the warp and the max-norm) over the 256^3 roofline config (BASELINE.json
configs[2]: params_boxing.ini solver values, dims overridden to 256, two analytic spheres 1.3 voxels apart).  Inputs are
resident in HBM and the solve is open (sobfu_hip_solver_begin) before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 --steps 20 --warmup 5          # re-executes itself under torch.distributed.run

Timing: W untimed warm-up iterations, then `--repeats` (default 7) timed regions of EXACTLY K iterations each, every region
bracketed by barrier + synchronize on both sides, MAX over ranks per region; `value` is the MEDIAN region (K iterations /
seconds), every region's rate is listed in `region_its`.  The per-solve fixed cost (entering / leaving the iteration format,
reading the max-norm rows) is not an iteration and is reported separately (`per_solve`).  Afterwards, outside the timed
regions, `--profile-repeats` more regions run with HIP events around every launch for the per-kernel split (`roofline`).

Prints ONE JSON line (rank 0), last on stdout.  `roofline` = the dominant kernel (pass B: Sobolev smoothing + psi update +
warp + max-norm) incl. `traffic` (fabric bytes per launch from two live rocprofv3 PMC passes, N=1), `cpu_baseline` = the oracle's
OpenMP port of the same iteration timed on the host cores (N=1 only), `per_frame` = the whole per-frame pipeline of the reference's
SobFusion::operator() (frames/s), `gpu_state` = shader clock / package power sampled inside the timed regions.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only does dmabuf IPC (RCCL across processes needs it)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
B_PASS_B = 64           # SURVEY 8(d) algorithmic bytes: nabla_U r16 + psi r16 w16 + phi_n gather 8 + phi_n o psi w8
B_PASS_A = 48           # phi_n o psi r8 + phi_global r8 + psi r16 + nabla_U w16
B_ITER = B_PASS_A + B_PASS_B
C_PASS_B = 44           # bytes the compact iteration format must move: nabla_U r12 + psi r12 w12 + phi_n gather 4 + F w4
C_PASS_A = 32           # F r4 + G r4 + psi r12 + nabla_U w12
C_ITER = C_PASS_A + C_PASS_B


def kernel_source_sha256():
    """SHA-256 of the iteration kernels' source: solver_kernels.hip and the parts it includes (what profiles/pmc_latest.json is stamped with)"""
    import hashlib

    d = os.path.join(ROOT, "sobfu_amd", "csrc")
    h = hashlib.sha256()
    for name in ["solver_kernels.hip"] + sorted(f for f in os.listdir(d) if f.startswith("solver_") and f.endswith(".inl")):
        with open(os.path.join(d, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


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


def _cpu_rate(O, P, threads, n_iters, warm):
    """iterations/s of the oracle's estimate_psi loop on `threads` host threads"""
    O.set_num_threads(threads)
    dims = P["dims"]
    c0, c1, r = sphere_pair(P)
    pg, pn = O.new_volume(dims), O.new_volume(dims)
    O.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r)
    O.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
    psi = O.new_field(dims)
    O.init_identity(psi)
    kw = dict(alpha=P["alpha"], w_reg=P["w_reg"], max_update_norm=-1.0, compute_jacobian=False, inverse_iters=0)
    if warm:
        O.estimate_psi(pg, pn, psi, max_iter=warm, **kw)
    t0 = time.perf_counter()
    O.estimate_psi(pg, pn, psi, max_iter=n_iters, **kw)  # also one extra apply per call: negligible next to the iterations
    return n_iters / (time.perf_counter() - t0)


def cpu_baseline(P, budget_s=12.0):
    """Oracle (OpenMP port, kind='port') timed on the host cores on bounded samples of the same workload: all cores and one
    core, on the bench grid and on BASELINE config 1's 64^3 grid."""
    import oracle as O

    O.build()
    all_cores = O.num_threads()  # the affinity mask capped by the container's cgroup CPU quota (oracle.usable_cpus)
    dim = P["dims"][0]
    one = _cpu_rate(O, P, all_cores, 1, 0)  # sizes the sample (and warms the pages)
    n = int(max(2, min(20, budget_s * one)))
    v_all = _cpu_rate(O, P, all_cores, n, 0)
    n1 = 2 if dim >= 256 else 4
    v_one = _cpu_rate(O, P, 1, n1, 0)
    # BASELINE config 1: 64^3, params_advent.ini solver values, 10 iterations
    vs = np.array([np.float32(0.5) / np.float32(64)] * 3, np.float32)
    P1 = dict(dims=(64, 64, 64), vs=vs, trunc=np.float32(5) * vs[0], eta=np.float32(2) * vs[0], alpha=0.1, w_reg=0.2)
    # sized to >= ~0.5 s each: 10 iterations (27 ms on 16 threads) swung 2x between boxes
    n_all = int(max(10, min(2000, 0.6 * _cpu_rate(O, P1, all_cores, 10, 2))))
    c1_all = _cpu_rate(O, P1, all_cores, n_all, 0)
    n_one = int(max(10, min(400, 0.6 * _cpu_rate(O, P1, 1, 10, 2))))
    c1_one = _cpu_rate(O, P1, 1, n_one, 0)
    O.set_num_threads(all_cores)
    what = (f"oracle/sobfu_oracle.c, OpenMP over z-planes on the {all_cores} CPUs the container may use (affinity mask "
            f"{len(os.sched_getaffinity(0))}, capped by its cgroup CPU quota), -O3 -ffp-contract=off, Jacobian pass skipped")
    return {"value": v_all, "unit": "iterations/s", "cores": all_cores, "kind": "port",
            "sample": f"{n} solver iterations of the same {dim}^3 workload after 1 warm-up iteration ({what})",
            "one_core": {"value": v_one, "unit": "iterations/s", "cores": 1, "sample": f"{n1} solver iterations of the same {dim}^3 workload"},
            "config1_64": {"all_cores": {"value": c1_all, "cores": all_cores}, "one_core": {"value": c1_one, "cores": 1},
                           "unit": "iterations/s",
                           "sample": f"{n_all} (all cores) / {n_one} (one core) solver iterations on BASELINE config 1's grid (64^3, alpha 0.1, "
                                     "w_reg 0.2, S=7, lambda 0.1, two analytic spheres) after 12 warm-up iterations: >= 0.5 s each"}}


def reference_build_on_this_gpu(P, repeat=2):
    """Part of the baseline leg: oracle/_ref/reference_hip_ieee -- the reference's own .cu / .cpp files compiled for gfx950 by hipcc through a
    CUDA -> HIP name-map header in the build container (oracle/ref_hipbuild; shim evidence, arrays bit-identical to this repo's) -- runs its
    Solver::estimate_psi on this workload.  The iteration rate is the difference of a 100- and a 50-iteration solve.  None when the binary
    is not there (it cannot be built on the GPU box: the reference does not travel)."""
    import shutil
    import subprocess
    import tempfile

    exe = os.path.join(ROOT, "oracle", "_ref", "reference_hip_ieee")
    if not os.path.exists(exe):
        return None
    dim = P["dims"][0]
    c0, c1, r = sphere_pair(P)
    kw = dict(X=dim, Y=dim, Z=dim, size_x=0.75, size_y=0.75, size_z=0.75, trunc_vox=48.0, eta_vox=3.0, max_weight=128.0, s=P["s"], alpha=P["alpha"], w_reg=P["w_reg"],
              max_update_norm=P["max_update_norm"], verbosity=0, sphere_cx=c0[0], sphere_cy=c0[1], sphere_cz=c0[2], sphere2_cx=c1[0], sphere2_cy=c1[1],
              sphere2_cz=c1[2], sphere_r=r, repeat=repeat)
    kw["lambda"] = P["lam"]
    t = {}
    for n in (50, 100):
        d = tempfile.mkdtemp(prefix="ref_hip_")
        try:
            a = subprocess.run([exe, "time", d, "max_iter=%d" % n] + ["%s=%r" % (k, float(v)) for k, v in kw.items()], capture_output=True, text=True, timeout=120)
            if a.returncode != 0:
                return {"error": (a.stdout + a.stderr)[-300:]}
            t[n] = min(float(x) for x in open(os.path.join(d, "out_time.txt")).read().split())
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"value": 50.0 / (t[100] - t[50]), "unit": "iterations/s", "s_per_solve_50": t[50], "s_not_iterations": t[50] - (t[100] - t[50]),
            "kind": "the reference's own kernels and host loop (ten kernels, a host synchronisation and a 128 KB read-back per iteration, solver.cu:114-193), "
                    "compiled by hipcc for gfx950 through a CUDA -> HIP name-map header (oracle/ref_hipbuild; shim evidence, not a supported build of the reference)",
            "sample": "Solver::estimate_psi of 50 and of 100 iterations on the same %d^3 workload, best of %d each; rate = 50 / (t100 - t50)" % (dim, repeat)}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run the same command under torch.distributed.run (one rank per GPU) and
    hand its JSON line on as the last line of stdout."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, stdout=subprocess.PIPE)
    lines = p.stdout.decode(errors="replace").splitlines()
    line = next((ln for ln in reversed(lines) if ln.startswith('{"metric"')), None)
    for ln in lines:
        if ln is not line:
            print(ln, file=sys.stderr)
    sys.stderr.flush()
    if line is not None:
        print(line, flush=True)
    raise SystemExit(p.returncode if (p.returncode != 0 or line is not None) else 1)


class Ranks:
    """barrier / MAX over ranks.  Normal runs: torch.distributed on the nccl (= RCCL) backend, one GPU per rank.
    SOBFU_BENCH_SHARE_GPU=1 (bring-up on a machine with fewer GPUs than ranks): every rank uses cuda:0 and the process group
    runs on gloo -- RCCL refuses two ranks on one device."""

    def __init__(self, torch, dist):
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.share = os.environ.get("SOBFU_BENCH_SHARE_GPU") == "1"
        self.device = 0 if self.share else self.local_rank
        if self.world > 1 and not self.share and torch.cuda.device_count() < self.world:
            raise SystemExit(f"--gpus {self.world} but only {torch.cuda.device_count()} GPU(s) visible "
                             "(SOBFU_BENCH_SHARE_GPU=1 runs all ranks on cuda:0 over gloo for bring-up)")
        torch.cuda.set_device(self.device)
        if self.world > 1:
            import datetime

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.share:
                dist.init_process_group("gloo", timeout=datetime.timedelta(minutes=5))
            else:
                dist.init_process_group("nccl", device_id=torch.device("cuda", self.device), timeout=datetime.timedelta(minutes=5))

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def _reduce(self, values, op):
        if self.world == 1:
            return [float(v) for v in values]
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device="cpu" if self.share else "cuda")
        self.dist.all_reduce(t, op=op)
        return [float(v) for v in t.tolist()]

    def max(self, values):
        return self._reduce(values, self.dist.ReduceOp.MAX)

    def min(self, values):
        return self._reduce(values, self.dist.ReduceOp.MIN)

    def close(self):
        if self.world > 1:
            self.dist.barrier()
            self.dist.destroy_process_group()


def measure_traffic(args):
    """HBM-side bytes per launch of pass A / pass B, LIVE: two short rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- counters only,
    with --kernel-trace for the kernel names, never combined with other trace domains) over this same script, corrected with the
    factors measured on this part (profiles/r02_counter_calibration.json: FETCH_SIZE reports 1/2 of the bytes read, WRITE_SIZE the
    bytes written; KiB units) -> bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  These are L2 <-> fabric bytes (Infinity-Cache hits
    included): an upper bound of the HBM bytes.  Any failure (no rocprofv3, a time-out) -> None: the line is printed regardless."""
    import shutil
    import sqlite3
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCPROFILER")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under a profiler: no nested rocprofv3"
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
                cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", tmp, "-o", "r", "--", sys.executable, os.path.abspath(__file__),
                       "--steps", "20", "--warmup", "5", "--repeats", "1", "--profile-repeats", "0", "--frames", "0", "--dim", str(args.dim),
                       "--no-cpu-baseline", "--no-traffic"]
                subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=150, check=True)
                db = os.path.join(tmp, "r_results.db")
                c = sqlite3.connect(db)
                for name, cn, avg, n in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events "
                                                  "where name like '%fused_%' group by name, counter_name"):
                    vals[("a" if "potential" in name else "b", cn)] = (float(avg), int(n))
                c.close()
        out = {}
        for k in ("a", "b"):
            f, w = vals[(k, "FETCH_SIZE")], vals[(k, "WRITE_SIZE")]
            out[k] = {"bytes": (2.0 * f[0] + w[0]) * 1024.0, "launches": min(f[1], w[1])}
        return out, None
    except Exception as e:  # noqa: BLE001
        return None, repr(e)


class GpuState:
    """Shader clock and package power of THIS rank's GPU, read from the amdgpu hwmon files (freq1_input, power1_input) by a
    sampling thread while the timed regions run -- evidence for (or against) attributing box-to-box / run-to-run differences to
    the device's clock / power state.  Silent when sysfs does not expose the device."""

    def __init__(self, torch, device):
        import glob
        import threading

        self.dir, self.samples, self.marks, self._stop, self.why, self.after = None, [], [], threading.Event(), None, []
        try:
            pr = torch.cuda.get_device_properties(device)
            bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
            for d in glob.glob("/sys/class/drm/card*/device"):
                if os.path.realpath(d).endswith(bdf):
                    hw = glob.glob(os.path.join(d, "hwmon", "hwmon*"))
                    if hw and os.path.exists(os.path.join(hw[0], "freq1_input")):
                        self.dir, self.bdf = hw[0], bdf
            if self.dir is None:
                self.why = f"no hwmon directory for PCI device {bdf} under /sys/class/drm"
        except Exception as e:  # noqa: BLE001
            self.why = repr(e)
        if self.dir:
            self.thread = threading.Thread(target=self._run, daemon=True)
            self.thread.start()

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name)) as f:
                return float(f.read().strip())
        except (OSError, ValueError):
            return float("nan")

    def _run(self):
        while not self._stop.is_set():
            self.samples.append((time.perf_counter(), self._read("freq1_input") / 1e6, self._read("power1_input") / 1e6))
            time.sleep(0.002)

    def _partition(self):
        """compute / memory partition mode of the device (SPX / NPS1 ...) and its fabric clock -- static, read once"""
        out, dev = {}, os.path.dirname(os.path.dirname(self.dir))
        for key, name in (("compute", "current_compute_partition"), ("memory", "current_memory_partition"), ("fclk", "pp_dpm_fclk"), ("vbios", "vbios_version"), ("unique_id", "unique_id")):
            try:
                with open(os.path.join(dev, name)) as f:
                    out[key] = " ".join(f.read().split())
            except OSError:
                out[key] = None
        return out

    def mark(self, t0, t1):
        self.marks.append((t0, t1))
        if self.dir:  # read right behind a region, outside every timed one: memory clock, junction / HBM temperature
            self.after.append((self._read("freq2_input") / 1e6, self._read("temp2_input") / 1e3, self._read("temp3_input") / 1e3))

    def report(self):
        if not self.dir:
            return {"available": False, "why": self.why}
        self._stop.set()
        self.thread.join(timeout=1.0)
        inside = [(c, p) for t, c, p in self.samples if any(a <= t <= b for a, b in self.marks) and c == c]
        per = []
        for a, b in self.marks:
            v = [c for t, c, _ in self.samples if a <= t <= b and c == c]
            per.append(round(float(np.median(v)), 0) if v else None)
        if not inside:
            return {"available": True, "samples_in_timed_regions": 0, "source": f"hwmon of {self.bdf}"}
        c, p = np.array([x[0] for x in inside]), np.array([x[1] for x in inside])
        return {"available": True, "source": f"amdgpu hwmon (freq1_input, power1_input) of PCI device {self.bdf}, sampled every ~2 ms",
                "samples_in_timed_regions": len(inside),
                "sclk_mhz": {"median": float(np.median(c)), "min": float(c.min()), "max": float(c.max())},
                "power_w": {"median": float(np.nanmedian(p)), "min": float(np.nanmin(p)), "max": float(np.nanmax(p))},
                "sclk_mhz_per_region": per,
                "partition": self._partition(),
                "after_each_region": {"mclk_mhz": [round(a[0]) for a in self.after if a[0] == a[0]],
                                      "junction_c": [round(a[1]) for a in self.after if a[1] == a[1]],
                                      "hbm_c": [round(a[2]) for a in self.after if a[2] == a[2]]},
                "note": "freq1_input is a slowly updated average: within one run it climbs by several hundred MHz from region to region while "
                        "the regions' rates agree to 0.1 %, and SQ_BUSY_CYCLES / kernel time of the PMC passes says ~2.1 GHz under load"}


GPU_STATE = None


def timed_regions(ranks, torch, fn, repeats):
    """`repeats` regions of one fn() each (fn enqueues exactly K iterations), barrier + synchronize on both sides of every
    region; returns the per-region seconds after a MAX over ranks.  A rank's clock runs from its exit of the opening barrier to
    the moment its own GPU has drained; the closing barrier follows the stamp, so the region lasts until the SLOWEST rank is
    done (the MAX) without the host round trip of a barrier -- 0.1 ms and more on the nccl backend -- inside it."""
    secs = []
    for _ in range(repeats):
        torch.cuda.synchronize()
        ranks.barrier()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ranks.barrier()
        secs.append(t1 - t0)
        if GPU_STATE is not None:
            GPU_STATE.mark(t0, t1)
    return ranks.max(secs)


def bench_single(args, P, ranks, torch):
    """one whole grid per rank: the N=1 metric, and `--replicas` (N independent sequences, BASELINE config 5 style)"""
    from sobfu_amd import ops

    dims = P["dims"]
    N = dims[0] * dims[1] * dims[2]
    c0, c1, r = sphere_pair(P)
    pg, pn, pnp = ops.new_volume(dims), ops.new_volume(dims), ops.new_volume(dims)
    ops.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r)
    ops.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
    psi = ops.new_field(dims)
    ops.init_identity(psi)
    K, W, R, PR = args.steps, args.warmup, args.repeats, args.profile_repeats
    total = W + (R + PR) * K
    sv = ops.Solver(dims, max_iter=max(total, 50), alpha=P["alpha"], w_reg=P["w_reg"], s=P["s"], lam=P["lam"],
                    max_update_norm=P["max_update_norm"])
    sv.begin(pg, pn, pnp, psi, total)  # the solve is open and its state resident before anything is timed
    sv.step(W)
    secs = timed_regions(ranks, torch, lambda: sv.step(K), R)
    # per-kernel split: same loop, HIP events on the solver's stream around EVERY launch (events drain the pipeline between
    # two kernels, so these regions are not the ones whose wall time is quoted)
    ms_a = ms_b = n_prof = 0
    if PR > 0:
        sv.set_profiling(1)
        sv.get_profile(reset=True)
        for _ in range(PR):
            sv.step(K)
        torch.cuda.synchronize()
        ms_a, ms_b, n_prof = sv.get_profile()
        sv.set_profiling(0)
    rep, hist = sv.end()
    assert rep.iterations == total, (rep.iterations, total)
    assert np.isfinite(hist).all() and float(hist.max()) > 0
    # per-solve fixed cost: whole iterate() calls (begin + K iterations + end, host-synchronised) against K timed iterations
    solve = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sv.iterate(pg, pn, pnp, psi, 50)
        solve.append(time.perf_counter() - t0)
    res = dict(region_seconds=secs, N=N, ms_a=ms_a / max(n_prof, 1), ms_b=ms_b / max(n_prof, 1), n_prof=n_prof, last_norm=float(hist[-1]),
               workspace=sv.workspace_bytes(), solve50_s=float(np.median(solve)),
               parallelism="single" if ranks.world == 1 else f"{ranks.world} independent {dims[0]}^3 sequences, one per GPU (replicas, no exchange)")
    sv.close()
    return res


FRAME_CONFIGS = {"config3": ("params/config3_boxing_256.ini", 256), "config5": ("params/config5_umbrella_512.ini", 512),
                 "config2": ("params/config2_snoopy_128.ini", 128), "config1": ("params/config1_sphere_64.ini", 64)}


def bench_frames(args, ranks, torch):
    """Frames/s through the WHOLE per-frame pipeline the reference runs per sequence (src/sobfu/sob_fusion.cpp:71-145): depth
    pre-steps (bilateral, truncation, ray lengths) -> integrate(depth) into phi_n -> estimate_psi (MAX_ITER iterations + 48-sweep
    inverse + canonical->live warp) -> fuse.  One synthetic depth sequence per rank (a sphere translating 1.3 voxels per frame),
    frame 0 (initialisation of phi_global) not timed; every later frame is timed on its own, host-synchronised."""
    from sobfu_amd import fusion, params, synthetic

    ini, dim0 = FRAME_CONFIGS[args.frame_config]
    dim = args.dim if args.frame_dim <= 0 else args.frame_dim
    P = params.read_ini(os.path.join(ROOT, ini), dims=(dim if dim != dim0 else None))
    size, tz = float(P["size"][0]), float(P["t"][2])
    vx = float(P["vs"][0])
    radius = 0.2 * size
    depth = [torch.from_numpy(synthetic.render_sphere_depth((1.3 * vx * n, 0.0, tz + 0.5 * size), radius, P["intr"])).cuda()
             for n in range(args.frames)]
    fu = fusion.SobFusion(P, max_iter=args.frame_iters)
    fu(depth[0])
    torch.cuda.synchronize()
    ranks.barrier()
    ms, iters = [], []
    for n in range(1, args.frames):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rep = fu(depth[n])
        torch.cuda.synchronize()
        ms.append(1e3 * (time.perf_counter() - t0))
        iters.append(int(rep[0].iterations) if rep is not None else 0)
    moved = float((fu.psi[..., 0] - torch.arange(P["dims"][0], device="cuda", dtype=torch.float32)).abs().max().item())
    fu.close()
    worst = ranks.max(ms)  # per frame index, the slowest rank
    med = sorted(worst)[len(worst) // 2]
    return {"config": f"{ini} values, {P['dims'][0]}^3, {args.frame_iters} solver iterations per frame (MAX_ITER; max_update_norm "
                      f"{P['max_update_norm']:g} never fires), synthetic 640x480 depth sequence (sphere translating 1.3 voxels / frame)",
            "pipeline": "bilateral + truncation + ray lengths -> integrate(depth) -> estimate_psi (iterations + 48-sweep inverse + "
                        "canonical->live warp) -> fuse   [reference src/sobfu/sob_fusion.cpp:71-145]",
            "frames_timed": len(worst), "ms_per_frame": med, "ms_per_frame_all": [round(v, 3) for v in worst],
            "frames_per_s_per_gpu": 1e3 / med, "frames_per_s_aggregate": ranks.world * 1e3 / med, "sequences": ranks.world,
            "iterations_per_frame": iters, "psi_moved_max_abs": moved}


def _left(args):
    return float("inf") if getattr(args, "deadline", None) is None else args.deadline - time.time()


CORE_LINE = None    # rank 0: the early copy of a tiled run's line (value, legs, parity)
CORE_READY_AT = None  # every rank: when the timed legs were done (on_core)
LINE_LOCK = None      # whoever takes it prints THE line (the normal path or the watchdog): exactly one JSON line on stdout either way


def _core_line_with(reason):
    """the early core line, marked: the run did NOT complete its harvest -- `late_failure` says why (the exit code stays 0 so that the
    measured legs are not lost; a consumer that wants to tell a complete run from a salvaged one reads this key)"""
    d = json.loads(CORE_LINE)
    d["late_failure"] = reason
    return json.dumps(d)


def _arm_watchdog(args, rank):
    """--budget-s is a promise about when the line is on stdout.  Once the core of a tiled run's line exists (timed legs + bitwise
    self-checks done: on_core), a run still busy 25 s past its deadline (a harvest collective that never returns, a sick node) prints
    that core line (rank 0) and every rank leaves.  Before the core exists there is nothing valid to print: the run goes on."""
    import threading

    global LINE_LOCK
    LINE_LOCK = threading.Lock()
    if args.deadline is None:
        return

    grace = float(os.environ.get("SOBFU_BENCH_WATCHDOG_GRACE_S", "25"))  # (tests shorten it)

    def run():
        while True:
            if CORE_READY_AT is not None and time.time() > max(args.deadline + grace, CORE_READY_AT + grace + 5.0):
                if not LINE_LOCK.acquire(blocking=False):
                    return  # the normal path is printing the full line
                if rank == 0 and CORE_LINE is not None:
                    import ctypes

                    ctypes.CDLL(None).fflush(None)
                    print(_core_line_with("watchdog: still running %.0f s past --budget-s; the harvest behind the timed legs was abandoned"
                                          % (time.time() - args.deadline)), flush=True)
                os._exit(0)
            time.sleep(0.5)

    threading.Thread(target=run, daemon=True).start()


def make_line(args, P, res, world, force_tiled, full=True):
    """the JSON line from a result dict.  full=False: the EARLY copy of a tiled run (no live PMC passes, no CPU baseline)"""
    replicas = world > 1 and args.replicas
    K = args.steps
    secs = sorted(res["region_seconds"])
    med = secs[len(secs) // 2] if len(secs) % 2 else 0.5 * (secs[len(secs) // 2 - 1] + secs[len(secs) // 2])
    mult = world if replicas else 1
    its = mult * K / med
    N = res["N"]
    ms_b, ms_a = res.get("ms_b"), res.get("ms_a")

    def gbps(nbytes, ms):
        return (nbytes / (ms * 1e-3) / 1e9) if ms else None

    pmc = pmc_file = pmc_stale = None
    for name in ("pmc_latest.json",):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            import hashlib

            with open(path) as f:
                pj = json.load(f)
            pmc, pmc_file = pj.get("pass_b_hbm_bytes_per_launch"), "profiles/" + name
            pmc_stale = pj.get("kernel_source_sha256") != kernel_source_sha256()  # measured on other kernels?
    NL = res.get("launch_cells") or N  # cells one launch produces on one GPU (the largest tile's owned cells when tiled)
    ach_b = gbps(NL * B_PASS_B, ms_b)
    phys_b = gbps(NL * C_PASS_B, ms_b)
    out = {
        "metric": f"solver iterations/sec on {args.dim}^3 voxel grid",
        "value": its, "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": args.warmup,
        "ms_per_step": 1e3 * med / K, "higher_is_better": True,
        "scaling": ("strong" if not replicas else "weak") if world > 1 else "single", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.dim}^3 TSDF, params_boxing.ini solver values (alpha 0.001, w_reg 0.6, S=7, "
                               f"lambda 0.1, max_update_norm 1e-10), two analytic spheres 1.3 voxels apart; a step = one solver "
                               f"iteration of an open solve",
                   "grid": [args.dim] * 3, "parallelism": res["parallelism"]},
        "repeats": len(secs), "timing": f"median of {len(secs)} regions of {K} iterations (barrier + synchronize around each, MAX over ranks)",
        "region_its": [round(mult * K / s, 1) for s in res["region_seconds"]],
        # whole-iteration view, per GPU: the 76 B/voxel the compact format must move per iteration against the HBM peak.
        # (SURVEY 8(d) prices an iteration at 112 algorithmic B/voxel; at that price the same run is
        # `iteration_algorithmic_GBps`, which can exceed the peak precisely because the format moves fewer bytes.)
        "iteration_physical_GBps": N * C_ITER * its / world / 1e9,
        "iteration_hbm_frac_physical": (N * C_ITER * its / world / 1e9) / HBM_PEAK_GBPS,
        "iteration_algorithmic_GBps": N * B_ITER * its / world / 1e9,
        "last_max_update_norm": res.get("last_norm"),
        "solver_workspace_bytes": res.get("workspace"),
    }
    traffic, traffic_err = (None, "not measured (--no-traffic, N > 1 or another grid path)")
    skipped = list(res.get("skipped") or [])
    if world == 1 and not args.no_traffic and not force_tiled and ms_b and full:
        if _left(args) > 40.0:
            traffic, traffic_err = measure_traffic(args)
        else:
            traffic_err = "skipped: the run's --budget-s left no room for the two PMC passes"
            skipped.append("roofline.traffic")
    if ms_b:
        out["roofline"] = {
            "kernel": "fused_smooth_update_apply_kernel (pass B: sum of three 1-D Sobolev convolutions + psi update + phi_n o psi "
                      "warp + max-norm)",
            "bound": "hbm", "achieved": ach_b, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach_b / HBM_PEAK_GBPS,
            # L2 <-> fabric bytes per launch of this kernel from rocprofv3's FETCH_SIZE / WRITE_SIZE counters, measured NOW by two
            # short PMC passes over this script (measure_traffic); null when that was not possible
            "traffic": traffic["b"]["bytes"] if traffic else None,
            "traffic_GBps": gbps(traffic["b"]["bytes"], ms_b) if traffic else None,
            "traffic_frac": (gbps(traffic["b"]["bytes"], ms_b) / HBM_PEAK_GBPS) if traffic else None,
            "traffic_how": ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, 20 iterations each) of this script just now; "
                            "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 with the factors calibrated on this part; " +
                            f"{traffic['b']['launches']} launches averaged") if traffic else traffic_err,
            "algorithmic_bytes_per_launch": NL * B_PASS_B, "avg_launch_ms": ms_b, "launches_timed": res.get("n_prof"),
            "how": "HIP events on the solver's stream around every pass-A / pass-B launch of the profiled regions"
                   + ("; per GPU: one launch produces the largest tile's owned cells (MAX over ranks of the averages)" if res.get("launch_cells") else ""),
            # the solver iterates on a compact copy of the state (12-byte psi / nabla_U, tsdf-only TSDF streams): the bytes
            # the kernel must physically move are below the survey's algorithmic figure
            "physical_bytes_per_launch": NL * C_PASS_B, "physical_GBps": phys_b, "frac_physical": phys_b / HBM_PEAK_GBPS,
            "traffic_from_profiles": {"file": pmc_file, "bytes_per_launch": pmc, "GBps": gbps(pmc, ms_b) if pmc else None,
                                      "stale": pmc_stale,  # true: the kernel source has changed since the counters were collected
                                      "note": "rocprofv3 PMC pass committed under profiles/, NOT measured in this run"},
            "pass_a": {"avg_launch_ms": ms_a, "algorithmic_bytes_per_launch": NL * B_PASS_A, "physical_bytes_per_launch": NL * C_PASS_A,
                       "traffic": traffic["a"]["bytes"] if traffic else None,
                       "physical_GBps": gbps(NL * C_PASS_A, ms_a), "frac_physical": gbps(NL * C_PASS_A, ms_a) / HBM_PEAK_GBPS},
            "event_sum_vs_step": (ms_a + ms_b + (res.get("ms_exchange") or 0.0)) / (1e3 * med / K),
        }
        if res.get("ms_exchange") is not None:  # N > 1, serial schedule: what an iteration is made of
            out["tiled_iteration_ms"] = {"pass_a_incl_message_stores" + ("_and_peer_wait" if res.get("transport") == "direct" else ""): ms_a,
                                         "exchange_transfer_and_scatter": res["ms_exchange"], "pass_b": ms_b}
    if res.get("solve50_s"):
        s50 = res["solve50_s"]
        out["per_solve"] = {"iterations": 50, "ms": 1e3 * s50, "fixed_ms": 1e3 * s50 - 50 * 1e3 * med / K,
                            "iterations_per_s_incl_fixed": 50 / s50,
                            "note": "one whole sobfu_hip_solver_iterate call of 50 iterations (BASELINE config 3's frame): enter the "
                                    "compact format + 50 iterations + max-norm rows to the host + leave, host-synchronised"}
    for k in ("tiles", "legs", "tiled_autotune_us", "tiled_diag", "transport", "transport_fallback", "per_frame", "topology"):
        if res.get(k):
            out[k] = res[k]
    if res.get("legs"):
        # `value` stays on BASELINE config 4's grid (2 x 2 x 2 at N = 8).  The fastest (grid, transport) this machine showed is a
        # first-class number beside it: the timed legs on the run's grid, and -- when the harvest ran -- every grid of N tiles on
        # both transports (a 40-iteration sweep per grid: tiled_autotune_us)
        here = "x".join(map(str, res["tiles"]["grid"]))
        cands = [(leg["value"], here, t, "timed leg (the regions `value` is the median of)") for t, leg in res["legs"].items() if leg.get("value")]
        for t, grids in (res.get("tiled_autotune_us") or {}).items():
            cands += [(1e6 / us, g, t, "40-iteration sweep of every grid of N tiles (tiled_autotune_us), outside the timed regions") for g, us in grids.items()
                      if us and g != here]
        if cands:
            v, g, t, how = max(cands)
            out["best_grid"] = {"grid": [int(x) for x in g.split("x")], "transport": t, "value": v, "unit": "iterations/s", "how": how,
                                "candidates": len(cands)}
    if GPU_STATE is not None:
        out["gpu_state"] = GPU_STATE.report()
    if res.get("tiled_parity") is not None:  # N > 1: every rank re-ran the whole solve alone and compared its tile bitwise
        out["tiled_parity_vs_single_gpu"] = "bit-exact" if res["tiled_parity"] else "MISMATCH"
    if world == 1 and not args.no_cpu_baseline and full:
        if _left(args) > 45.0:
            out["cpu_baseline"] = cpu_baseline(P)
            try:
                rb = reference_build_on_this_gpu(P)
            except Exception as e:  # a baseline beside the line, never a reason to lose it
                rb = {"error": repr(e)[:300]}
            if rb is not None:
                if rb.get("value"):
                    rb["this_repo_over_reference_build"] = out["value"] / rb["value"]
                out["reference_build_on_this_gpu"] = rb
        else:
            skipped.append("cpu_baseline")
    if args.budget_s > 0:
        out["budget_s"] = args.budget_s
    if skipped:
        out["skipped"] = skipped
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=7, help="timed regions of --steps iterations each; the median is reported")
    ap.add_argument("--profile-repeats", type=int, default=2, help="extra regions with HIP events around every launch (kernel split)")
    ap.add_argument("--dim", type=int, default=256, help="grid edge (256 = the BASELINE metric's grid)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 PMC passes that measure roofline.traffic (N = 1)")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: every rank solves its OWN grid (independent sequences, BASELINE config 5 style: no exchange, weak "
                         "scaling) instead of the default -- ONE grid cut into N tiles with a halo exchange per iteration (strong scaling; "
                         "both transports, peer-mapped stores and RCCL, are timed: bench_tiled.py)")
    ap.add_argument("--tiles", type=str, default="", help="N > 1, strong scaling: tile grid PxxPyxPz (default sobfu_amd.tiled.default_grid: "
                                                         "2x2x2 at N=8, 1x2x2 at N=4, 1x1x2 at N=2), or 'auto': time every grid of N "
                                                         "tiles on this machine before the timed region and keep the fastest")
    ap.add_argument("--frames", type=int, default=-1, help="frames of the per-frame pipeline to run for the `per_frame` block (frame 0 = "
                                                           "initialisation, untimed); default: 5 at N = 1 and with --replicas, 0 otherwise")
    ap.add_argument("--frame-config", choices=sorted(FRAME_CONFIGS), default="config3",
                    help="parameter set of the per-frame pipeline: config3 = params_boxing.ini values (the bench grid), config5 = "
                         "params_umbrella.ini values at 512^3 (BASELINE config 5; use with --replicas for batched sequences)")
    ap.add_argument("--frame-dim", type=int, default=0, help="grid edge of the per-frame pipeline (default: --dim)")
    ap.add_argument("--frame-iters", type=int, default=50, help="solver iterations per frame (MAX_ITER; BASELINE config 3 states 50)")
    ap.add_argument("--budget-s", type=float, default=-1.0,
                    help="wall-clock budget of the WHOLE run in seconds, counted from the first process's start.  The timed legs always run; "
                         "every harvest step behind them (diagnostics, other tile grids, frames on tiles, topology; at N = 1 the PMC traffic "
                         "passes, the CPU baseline, the per-frame pipeline) runs only while the budget leaves room for it and is listed under "
                         "`skipped` otherwise; a run still busy 25 s past its budget prints the core line it already holds and exits.  "
                         "Default: 300 s at N > 1 (a healthy 8-GPU run takes about two minutes), none at N = 1; 0 = none")
    args = ap.parse_args()
    os.environ.setdefault("SOBFU_BENCH_T0", repr(time.time()))  # (inherited by the ranks of a self-launched run: one clock for all)
    if args.budget_s < 0:
        args.budget_s = 300.0 if (args.gpus > 1 and not args.replicas) else 0.0
    args.deadline = (float(os.environ["SOBFU_BENCH_T0"]) + args.budget_s) if args.budget_s > 0 else None
    if args.frames < 0:
        args.frames = 5 if (args.gpus == 1 or args.replicas) else 4  # tiles: frame 0 + three timed frames of the tiled pipeline
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if rank != 0:  # only rank 0 owns stdout: libraries (RCCL's version banner) write there from every process, some at exit
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    ranks = Ranks(torch, dist)
    global GPU_STATE
    if rank == 0:
        GPU_STATE = GpuState(torch, ranks.device)

    P = boxing_params(args.dim)
    force_tiled = os.environ.get("SOBFU_FORCE_TILED") == "1"  # exercise the tile path on one GPU (debugging)
    _arm_watchdog(args, rank)
    if (world > 1 and not args.replicas) or force_tiled:
        import bench_tiled

        def on_core(core_res):  # called by every rank as soon as the timed legs and their bitwise self-checks are done
            global CORE_LINE, CORE_READY_AT
            CORE_READY_AT = time.time()
            if rank == 0:
                CORE_LINE = json.dumps(make_line(args, P, core_res, world, force_tiled, full=False))
                print("[bench] core line (early copy; the full line is the last line on stdout): " + CORE_LINE, file=sys.stderr, flush=True)

        args.on_core = on_core

        res = bench_tiled.bench_tiled(args, P, ranks, timed_regions)
    else:
        res = bench_single(args, P, ranks, torch)
        if args.frames >= 2 and not force_tiled:
            go = _left(args) > 30.0
            if world > 1:  # replicas: rank 0's clock decides for everybody (bench_frames ends in a collective)
                go = ranks.min([1.0 if (rank != 0 or go) else 0.0])[0] == 1.0
            if go:
                res["per_frame"] = bench_frames(args, ranks, torch)
            else:
                res["skipped"] = list(res.get("skipped") or []) + ["per_frame"]

    out = make_line(args, P, res, world, force_tiled) if rank == 0 else None
    mismatch = res.get("tiled_parity") is False  # every rank holds the same verdict (MIN over ranks)
    # ... and so does a frame tail on tiles that differs from the same tail on all-gathered sources
    mismatch = mismatch or ((res.get("per_frame") or {}).get("tail") or {}).get("parity_vs_all_gather_tail") == "MISMATCH"
    if res.get("diag_hung_any"):
        # a diagnostics collective never returned on SOME rank (the verdict was agreed over a side channel that is not the
        # wedged communicator): nobody enters another collective -- every rank reports what was measured and leaves
        LINE_LOCK.acquire()  # (never released: the watchdog must not print a second line)
        if rank == 0:
            import ctypes

            ctypes.CDLL(None).fflush(None)
            print(json.dumps(out), flush=True)
        os._exit(3 if mismatch else 0)
    if os.environ.get("SOBFU_BENCH_TEST_HANG") == "2" and rank == world - 1 and world > 1:  # test hook: a rank that never reaches the final barrier
        time.sleep(3600)
    ranks.close()
    if rank == 0:  # after the process group is gone, and after flushing C stdio (RCCL's version banner sits in libc's stdout
        # buffer until exit when stdout is a pipe), so that the JSON is the LAST line on stdout
        import ctypes

        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
        LINE_LOCK.acquire()  # (never released: the watchdog must not print a second line)
        print(json.dumps(out), flush=True)
        # nothing may follow the line on stdout (RCCL writes banners from destructors / at exit): point fd 1 at /dev/null for the
        # rest of the process instead of killing it -- exit hooks (rocprofv3 writing its results) must still run
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if mismatch:  # the line above says MISMATCH; the exit code says so too
        raise SystemExit(3)


def _main_guarded():
    """main(), except that once the core of a tiled run's line exists nothing that happens later may cost the line: if a peer has
    already left (its watchdog fired first -- the ranks' clocks differ -- or it died) a late collective on this rank raises; rank 0
    then prints the core line it holds and every rank exits 0."""
    try:
        main()
    except BaseException as e:  # noqa: BLE001 -- incl. SystemExit from a late step
        if CORE_READY_AT is None or (isinstance(e, SystemExit) and e.code in (0, None)):
            raise
        rank = int(os.environ.get("RANK", "0"))
        print(f"[bench rank {rank}] after the timed legs: {e!r}; the core line stands", file=sys.stderr, flush=True)
        if rank == 0 and CORE_LINE is not None and LINE_LOCK is not None and LINE_LOCK.acquire(blocking=False):
            import ctypes

            ctypes.CDLL(None).fflush(None)
            print(_core_line_with("after the timed legs: %r" % (e,)), flush=True)
        mismatch = isinstance(e, SystemExit) and e.code == 3  # a MISMATCH verdict keeps its exit code
        os._exit(3 if mismatch else 0)


if __name__ == "__main__":
    _main_guarded()
