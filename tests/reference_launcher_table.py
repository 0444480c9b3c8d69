"""Launcher by launcher on the GPU at hand: the reference's own kernels (oracle/_ref/reference_hip_ieee, oracle/ref_hipbuild) and this repo's C-ABI
launchers on the same arrays -- every row of SURVEY 8(a) that launches a kernel.

    python tests/reference_launcher_table.py [dim] > profiles/r06/launcher_table_256.md        # on the GPU box

reference side: the driver's `launchers` scenario (each launcher `repeat` times back to back) under `rocprofv3 --kernel-trace --stats`; the
                kernels of one launcher call are summed (estimate_inverse = 48 sweeps, the reductions' device part only);
this repo:      the same launcher through sobfu_amd.ops, HIP events around `repeat` back-to-back launches on torch's current stream (the stream
                the launchers run on);
bytes:          SURVEY 8(a)'s "traffic today" column -- the compulsory reads + writes of the launcher AS THE REFERENCE DECOMPOSES the work (so a
                fraction near 1 says the launcher streams; the fused iteration kernels are measured by bench.py, not here); a17 is 64, not the
                survey's 80 (8 + 8 + 16 + 16 read, 16 written: the counters agree).
Shim evidence on the reference side (a name-map header stands in for the CUDA toolkit): a measured baseline, not a supported build."""
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
F32 = np.float32

# launcher -> (substrings of the reference kernels one call launches with calls of each per launcher call, this repo's kernels likewise, bytes per voxel (None: per pixel))
ROWS = [
    ("apply (a12)", [("apply_kernel", 1)], [("apply_kernel", 1)], 32),
    ("estimate_gradient (a14)", [("estimate_gradient_kernel", 1)], [("tsdf_gradient_kernel", 1)], 24),
    ("laplacian (a15)", [("estimate_laplacian_kernel", 1)], [("laplacian_kernel", 1)], 32),
    ("deformation Jacobian (a16)", [("estimate_deformation_jacobian_kernel", 1)], [("jacobian_kernel", 1)], 80),
    ("calculate_potential_gradient (a17)", [("calculate_potential_gradient_kernel", 1)], [("potential_gradient_kernel", 1)], 64),
    ("convolution_rows (a18)", [("convolution_rows_kernel", 1)], [("conv1d_kernel<0", 1)], 32),
    ("convolution_columns (a18)", [("convolution_columns_kernel", 1)], [("conv1d_kernel<1", 1)], 48),
    ("convolution_depth (a18)", [("convolution_depth_kernel", 1)], [("conv_depth_march_kernel", 1)], 48),
    ("update_psi (a19)", [("update_psi_kernel", 1)], [("update_psi_kernel", 1)], 64),
    ("max_update_norm (a20), device part", [("reduce_max_kernel", 1)], [("tree_max_kernel", 1)], 16),
    ("data_energy (a21), device part", [("reduce_data_kernel", 1)], [("DataEl", 1)], 16),
    ("reg_energy_sobolev (a21), device part", [("reduce_reg_sobolev_kernel", 1)], [("RegEl", 1)], 64),
    ("init_identity + estimate_inverse, 48 sweeps (a13)", [("init_identity_kernel", 1), ("estimate_inverse_kernel", 48)],
     [("init_identity_kernel", 1), ("inverse_fixed_point_kernel", 1)], 16 + 48 * 48),
    ("integrate(phi_global, phi_n_psi) (a5)", [("TsdfVolume, kfusion::device::TsdfVolume)", 1)], [("integrate_fuse_kernel", 1)], 24),
    ("clear_volume (a3)", [("clear_volume_kernel", 1)], [("fillBuffer", 1)], 8),
    ("init_sphere (a6)", [("init_sphere_kernel", 1)], [("init_prim_kernel", 1)], 8),
    ("integrate(dists) (a4)", [("integrate_kernel", 1)], [("integrate_depth_kernel", 1)], 8),
    ("bilateralFilter 640x480 (a7)", [("bilateral_kernel", 1)], [("bilateral_kernel", 1)], None),
    ("truncateDepth 640x480 (a7)", [("truncate_depth_kernel", 1)], [("truncate_depth_kernel", 1)], None),
    ("compute_dists 640x480 (a7)", [("compute_dists_kernel", 1)], [("compute_dists_kernel", 1)], None),
]


def top_kernels(db):
    return {name: (calls, avg) for name, calls, avg in sqlite3.connect(db).execute("select name,total_calls,average from top_kernels")}


def pick(table, kernels, used):
    """sum of the average durations of the kernels of one launcher call; -> (us, description)"""
    tot, found = 0.0, []
    for sub, per_call in kernels:
        for kname, (calls, avg) in table.items():
            if sub in kname and kname not in used:
                used.add(kname)
                tot += avg * per_call
                found.append("%s %.1f%s" % (kname.split("(")[0].split("::")[-1], avg, " × %d" % per_call if per_call > 1 else ""))
                break
    return tot, "; ".join(found) if found else "?"


def reference_times(exe, d, kw, repeat):
    out = os.path.join(d, "prof")
    env = dict(os.environ, TMPDIR="/tmp")
    r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", out, "-o", "r", "--", exe, "launchers", d, "repeat=%d" % repeat, "flush=1"] +
                       ["%s=%r" % (k, float(v)) for k, v in kw.items()], capture_output=True, text=True, timeout=900, cwd="/tmp", env=env)
    db = os.path.join(out, "r_results.db")
    assert os.path.exists(db), (r.stdout + r.stderr)[-3000:]
    return top_kernels(db)


def workload(dim):
    import fixture_inputs as FI

    P = dict(rows=480, cols=640, fx=570.342, fy=570.342, cx=320.0, cy=240.0, trunc_depth=1.5, bilateral_ksz=7, bilateral_ss=4.5, bilateral_sd=0.04, X=dim, Y=dim, Z=dim,
             size_x=0.75, size_y=0.75, size_z=0.75, trunc_vox=48.0, eta_vox=3.0, t_z=0.3, max_weight=128.0, s=7, alpha=0.001, w_reg=0.6, sphere_cx=0.375, sphere_cy=0.375,
             sphere_cz=0.375, sphere_r=0.2)
    P["lambda"] = 0.1
    ins = FI.kernel_inputs((dim, dim, dim), 11, 0.45)
    intr = (P["fx"], P["fy"], P["cx"], P["cy"])
    return P, ins, intr, FI.bench_sequence_frame(intr, 0.75, P["t_z"], 0.75 / dim, 1)


def ours(dim, repeat, flush=False):
    """every launcher through sobfu_amd.ops `repeat` times -> microseconds per call from HIP events, back to back (in the order of ROWS); flush: 512 MB are
    overwritten before every call instead (cold Infinity Cache; the kernel times are then read from rocprofv3 by the parent)"""
    import torch

    from sobfu_amd import ops
    from test_reference_fixtures import _pose, _tsdf_params

    P, ins, intr, depth = workload(dim)
    dims = (dim, dim, dim)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    vol, pg, psi, fuse = dev(ins["phi_n_psi"]), dev(ins["phi_global"]), dev(ins["psi"]), dev(ins["fuse_in"])
    S = ins["taps"]
    grad, L, nU, nUS, upd, inv = (ops.new_field(dims) for _ in range(6))
    J, warped, v = ops.new_jacobian(dims), ops.new_volume(dims), ops.new_volume(dims)
    size, vs, trunc, eta = _tsdf_params(P, dims)
    R, t = _pose(P, size)
    draw = dev(depth.view(np.int16))
    filt = ops.bilateral_filter(draw, 7, 4.5, 0.04)
    dists = ops.compute_dists(filt, intr)

    scratch = torch.empty(512 << 20, dtype=torch.uint8, device="cuda") if flush else None

    def timed(fn):
        if flush:
            for i in range(repeat):
                scratch.fill_(i)
                fn()
            torch.cuda.synchronize()
            return 0.0
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(repeat):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return 1e3 * e0.elapsed_time(e1) / repeat

    def inverse():
        ops.init_identity(inv)
        ops.estimate_inverse(psi, inv, 48)

    fns = [
        lambda: ops.apply(vol, warped, psi), lambda: ops.tsdf_gradient(vol, grad), lambda: ops.laplacian(psi, L), lambda: ops.jacobian(psi, J, 1),
        lambda: ops.potential_gradient(vol, pg, grad, L, nU, 0.6), lambda: ops.convolution_rows(nUS, nU, S), lambda: ops.convolution_columns(nUS, nU, S),
        lambda: ops.convolution_depth(nUS, nU, S), lambda: ops.update_psi(psi, nUS, upd, 0.001), lambda: ops.max_update_norm(upd), lambda: ops.data_energy(pg, vol),
        lambda: ops.reg_energy_sobolev(J), inverse, lambda: ops.integrate_fuse(fuse, warped, 128.0), lambda: ops.clear_volume(v),
        lambda: ops.init_sphere(v, vs, trunc, eta, (0.375, 0.375, 0.375), 0.2), lambda: ops.integrate_depth(dists, v, vs, trunc, eta, R, t, intr),
        lambda: ops.bilateral_filter(draw, 7, 4.5, 0.04), lambda: ops.truncate_depth(filt, 1.5), lambda: ops.compute_dists(filt, intr),
    ]
    out = []
    for i, fn in enumerate(fns):
        if i == 5:
            ops.convolution_rows(nUS, nU, S)
        out.append(timed(fn))
        if i == 7:
            ops.convolution_rows(nUS, nU, S)  # bounded again after the accumulating repeats
    return out


CHAIN = [("estimate_gradient", "tsdf_gradient_kernel", "estimate_gradient_kernel"), ("deformation Jacobian", "jacobian_kernel", "estimate_deformation_jacobian_kernel"),
         ("laplacian", "laplacian_kernel", "estimate_laplacian_kernel"), ("calculate_potential_gradient", "potential_gradient_kernel", "calculate_potential_gradient_kernel"),
         ("convolution_rows", "conv1d_kernel<0", "convolution_rows_kernel"), ("convolution_columns", "conv1d_kernel<1", "convolution_columns_kernel"),
         ("convolution_depth", "conv_depth_march_kernel", "convolution_depth_kernel"), ("update_psi", "update_psi_kernel", "update_psi_kernel"),
         ("apply", "apply_kernel", "apply_kernel"), ("max_update_norm (device part)", "tree_max_kernel", "reduce_max_kernel")]


def chain(dim, iters):
    """this repo's launchers in the order of the reference's iteration (solver.cu:114-193: gradient, Jacobian, Laplacian, potential gradient, three convolutions,
    update, warp, max-norm with its read-back) on bench.py's workload, one stream -> microseconds per iteration (HIP events)"""
    import torch

    import bench
    from sobfu_amd import ops

    P = bench.boxing_params(dim)
    dims = P["dims"]
    c0, c1, r = bench.sphere_pair(P)
    pg, pn, pnp = ops.new_volume(dims), ops.new_volume(dims), ops.new_volume(dims)
    ops.init_sphere(pg, P["vs"], P["trunc"], P["eta"], c0, r)
    ops.init_sphere(pn, P["vs"], P["trunc"], P["eta"], c1, r)
    psi, grad, L, nU, nUS, upd = (ops.new_field(dims) for _ in range(6))
    J = ops.new_jacobian(dims)
    ops.init_identity(psi)
    S = ops.sobolev_filter(P["s"], P["lam"])
    ops.apply(pn, pnp, psi)

    def one():
        ops.tsdf_gradient(pnp, grad)
        ops.jacobian(psi, J, 1)
        ops.laplacian(psi, L)
        ops.potential_gradient(pnp, pg, grad, L, nU, P["w_reg"])
        ops.convolution_rows(nUS, nU, S)
        ops.convolution_columns(nUS, nU, S)
        ops.convolution_depth(nUS, nU, S)
        ops.update_psi(psi, nUS, upd, P["alpha"])
        ops.apply(pn, pnp, psi)
        return ops.max_update_norm(upd)

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        one()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters


def chain_table(dim, exe):
    """the reference's iteration as the reference decomposes it: its own kernels in its own loop (three streams, a host synchronisation and a read-back per
    iteration) and this repo's launcher-shaped kernels chained in the same order, kernel by kernel IN the chain (no call finds the previous call's arrays cached:
    2.3 GB cycle through per iteration at 256^3) -- with the streaming hints and without -- beside the two fused passes the solver really runs"""
    import json

    import bench

    P = bench.boxing_params(dim)
    c0, c1, r = bench.sphere_pair(P)
    kw = dict(X=dim, Y=dim, Z=dim, size_x=0.75, size_y=0.75, size_z=0.75, trunc_vox=48.0, eta_vox=3.0, max_weight=128.0, s=P["s"], alpha=P["alpha"], w_reg=P["w_reg"],
              max_update_norm=P["max_update_norm"], verbosity=0, sphere_cx=c0[0], sphere_cy=c0[1], sphere_cz=c0[2], sphere2_cx=c1[0], sphere2_cy=c1[1], sphere2_cz=c1[2],
              sphere_r=r, repeat=2, max_iter=50)
    kw["lambda"] = P["lam"]
    d = tempfile.mkdtemp(prefix="chain_")
    env = dict(os.environ, TMPDIR="/tmp")
    res = {}
    try:
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", os.path.join(d, "ref"), "-o", "r", "--", exe, "time", d] + ["%s=%r" % (k, float(v)) for k, v in kw.items()],
                       capture_output=True, text=True, timeout=900, cwd="/tmp", env=env)
        res["ref"] = top_kernels(os.path.join(d, "ref", "r_results.db"))
        subprocess.run([exe, "time", d] + ["%s=%r" % (k, float(v)) for k, v in kw.items()], capture_output=True, text=True, timeout=900)  # (wall time: not under the profiler)
        ref_wall = min(float(x) for x in open(os.path.join(d, "out_time.txt")).read().split()) / 50 * 1e6
        for tag, nt in (("nt", "1"), ("plain", "0")):
            r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", os.path.join(d, tag), "-o", "r", "--", sys.executable, os.path.abspath(__file__), "--chain", str(dim), "30"],
                               capture_output=True, text=True, timeout=900, cwd="/tmp", env=dict(env, SOBFU_LAUNCHER_NT=nt))
            assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
            res[tag] = top_kernels(os.path.join(d, tag, "r_results.db"))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--chain", str(dim), "30"], capture_output=True, text=True, timeout=900, env=dict(os.environ, SOBFU_LAUNCHER_NT=nt))
            res[tag + "_wall"] = json.loads([l for l in r.stdout.splitlines() if l.startswith("[")][-1])[0]
    finally:
        shutil.rmtree(d, ignore_errors=True)
    print("\n### The reference's iteration, kernel by kernel IN its loop, %d^3 (bench.py's workload)\n" % dim)
    print("Kernel time from `rocprofv3 --kernel-trace --stats` inside the running loop: the reference's own kernels in its own `estimate_psi` (gradient, Jacobian and Laplacian on "
          "three concurrent streams, a host synchronisation and a 128 KB read-back per iteration), this repo's launcher-shaped kernels chained in the same order on one stream "
          "with the streaming hints (the default beyond 3.3 M cells) and without (`SOBFU_LAUNCHER_NT=0`). Nothing is warm here: the chain cycles 2.3 GB per iteration at 256^3.\n")
    print("| launcher | reference kernel in its loop, µs | this repo, streaming hints, µs | this repo, plain, µs |\n|---|---|---|---|")
    tot = {"ref": 0.0, "nt": 0.0, "plain": 0.0}
    for name, mine, theirs in CHAIN:
        row = []
        for tag, sub in (("ref", theirs), ("nt", mine), ("plain", mine)):
            v = [avg for k, (calls, avg) in res[tag].items() if sub in k and (tag != "ref" or "rocclr" not in k)]
            row.append(v[0] if v else float("nan"))
            tot[tag] += row[-1]
        print("| %s | %.1f | %.1f | %.1f |" % (name, *row))
    print("| **sum of kernel times** | %.0f (overlapping streams) | %.0f | %.0f |" % (tot["ref"], tot["nt"], tot["plain"]))
    print("| **wall time per iteration** (not under the profiler) | %.0f (a 50-iteration `estimate_psi` / 50: its inverse and warps included) | %.0f | %.0f |" % (ref_wall, res["nt_wall"], res["plain_wall"]))
    print("\n(The solver's two fused passes do the same iteration in ≈ 235 – 245 µs: `bench.py`.)")


def main():
    import json

    if sys.argv[1:2] == ["--chain"]:
        print(json.dumps([chain(int(sys.argv[2]), int(sys.argv[3]))]))
        return

    if sys.argv[1:2] == ["--ours"]:  # the child: this repo's launchers; event times on the last line (back to back), or flushed launches under rocprofv3
        print(json.dumps(ours(int(sys.argv[2]), int(sys.argv[3]), flush=len(sys.argv) > 4 and sys.argv[4] == "flush")))
        return
    dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    repeat = 10
    N = dim ** 3
    exe = os.path.join(ROOT, "oracle", "_ref", "reference_hip_ieee")
    if not os.path.exists(exe):
        raise SystemExit("oracle/_ref/reference_hip_ieee is missing: build it in the build container (python oracle/ref_hipbuild/build.py)")
    P, ins, intr, depth = workload(dim)
    d = tempfile.mkdtemp(prefix="launchers_")
    try:
        for k, a in ins.items():
            np.ascontiguousarray(a).tofile(os.path.join(d, k + ".bin"))
        depth.tofile(os.path.join(d, "depth.bin"))
        ref = reference_times(exe, d, P, repeat)
        out = os.path.join(d, "ours")
        r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", out, "-o", "r", "--", sys.executable, os.path.abspath(__file__), "--ours", str(dim), str(repeat), "flush"],
                           capture_output=True, text=True, timeout=1800, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        mine = top_kernels(os.path.join(out, "r_results.db"))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--ours", str(dim), str(repeat)], capture_output=True, text=True, timeout=1800)
        assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
        t_events = json.loads([l for l in r.stdout.splitlines() if l.startswith("[")][-1])
    finally:
        shutil.rmtree(d, ignore_errors=True)

    print("Launcher by launcher at %d^3 on one MI355X, kernel time (`rocprofv3 --kernel-trace --stats`, average of %d calls, both sides, **512 MB overwritten before every call** so "
          "that no call finds its inputs in the 256 MB Infinity Cache because the call before it left them there): the reference's own kernels (hipcc build through a name-map "
          "header, `oracle/ref_hipbuild`) and this repo's C-ABI launchers on the same arrays%s. `warm call` = HIP events around %d back-to-back calls of this repo's launcher on the "
          "same arrays (launch gaps and, for the reductions, the read-back and the host finish included; inputs up to 256 MB come from the Infinity Cache there). Bytes per voxel: "
          "SURVEY 8(a)'s column for the launcher as the reference decomposes the work; fraction = kernel bytes/s over 8 TB/s. The solver does not run these kernels in its loop "
          "(it runs the two fused passes `bench.py` measures); they are the drop-in surface.\n"
          % (dim, repeat, " (SOBFU_LAUNCHER_NT=%s)" % os.environ["SOBFU_LAUNCHER_NT"] if "SOBFU_LAUNCHER_NT" in os.environ else "", repeat))
    print("| launcher (SURVEY row) | reference kernels, µs | this repo's kernels, µs | × | warm call, µs | B/voxel | GB/s | fraction of 8 TB/s |\n|---|---|---|---|---|---|---|---|")
    used_r, used_m = set(), set()
    for (name, rk, mk, bpv), ev in zip(ROWS, t_events):
        tr, dr = pick(ref, rk, used_r)
        tm, dm = pick(mine, mk, used_m)
        gbps = "" if bpv is None or not tm else "%.0f" % (N * bpv / (tm * 1e-6) / 1e9)
        frac = "" if bpv is None or not tm else "%.2f" % (N * bpv / (tm * 1e-6) / 8e12)
        print("| %s | %s | %s | %s | %.1f | %s | %s | %s |" % (name, dr, dm, "%.1f" % (tr / tm) if tr and tm else "", ev, "" if bpv is None else bpv, gbps, frac))
    rest = [k for k in ref if k not in used_r and "rocclr" not in k]
    if rest:
        print("\nreference kernels not in a row: " + "; ".join("%s (%d calls, %.1f µs)" % (k.split("(")[0], ref[k][0], ref[k][1]) for k in rest))
    chain_table(dim, exe)


if __name__ == "__main__":
    main()
