"""Regenerates tests/golden/ref_*.npz: full arrays computed by the REFERENCE'S OWN SOURCE LINES, run on the CPU in the build
container through the host emulation of tools/ref_emulation/ (builder-authored stand-in headers + one launch-rewriting regex
+ a coroutine block scheduler; the reference's files are compiled where they lie, generated copies live in a temp dir that is
removed at the end -- only arrays come back).

    python tests/golden/make_reference_fixtures.py            # rewrites the fixtures (byte-identical on every run; ~6 min on 8 cores: BASELINE config 2's 808 iterations
                                                              # at 128^3 and config 3's 50 at 256^3 under emulation)
    python tests/golden/make_reference_fixtures.py --check    # regenerates into memory and compares with the committed files
    python tests/golden/make_reference_fixtures.py --check --only=ref_kernels_17x9x5,ref_mc_14x11x9   # a quick subset (the CPU suite runs this)

WHAT THIS EVIDENCE IS: shim evidence.  The emulation needs stand-ins for CUDA / OpenCV / PCL headers the image lacks, so by the
task's rules it is not a reference build and pins nothing: the oracle's parity grade stays "partial" (DESIGN.md section 2).  It
is still the strongest link this image allows between oracle/sobfu_oracle.c + the HIP kernels and
/root/reference/src/sobfu/cuda/solver.cu:15-205,237-459, vector_fields.cu:28-138,144-472, reductor.cu, tsdf_volume.cu:23-373,
imgproc.cu:8-77,233-254, src/sobfu/solver.cpp:7-101,160-262, src/sobfu/sob_fusion.cpp:71-145, marching_cubes.{cpp,cu}.
Arithmetic of the emulation: IEEE binary32, nothing contracted, libm powf/expf (tools/ref_emulation/shim/cuda_runtime.h).

Consumers: tests/test_reference_fixtures.py (oracle on CPU, HIP on the GPU box, bitwise).  Nothing else imports this recipe.
The inputs are made here with numpy only (no oracle, no sobfu_amd kernels), so the fixtures do not depend on the code they check.
"""
import hashlib
import io
import os
import shutil
import subprocess
import sys
import tempfile
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_emulation"))
sys.path.insert(0, ROOT)

import build as emu_build  # noqa: E402  (tools/ref_emulation/build.py)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fixture_inputs import (bench_sequence_frame, digest, identity, kernel_inputs, mc_volume, rand_volume, snoopy_frame, sphere_volume,  # noqa: E402  (numpy-only generators)
                            translating_sphere_frame, warped_identity)
from sobfu_amd.synthetic import render_sphere_depth  # noqa: E402,F401  (pure numpy)

F32 = np.float32


class Emu:
    def __init__(self, arch="610"):
        self.work = tempfile.mkdtemp(prefix="ref_emu_")
        self.exe = emu_build.build(self.work, arch=arch)
        self.n = 0

    def run(self, scenario, inputs, outputs, **kw):
        """inputs: name -> array; outputs: name -> (dtype, shape).  Returns dict of arrays + 'log' (the reference's stdout)."""
        d = os.path.join(self.work, "run%03d" % self.n)
        self.n += 1
        os.makedirs(d)
        for k, a in inputs.items():
            np.ascontiguousarray(a).tofile(os.path.join(d, k + ".bin"))
        subprocess.check_call([self.exe, scenario, d] + ["%s=%r" % (k, float(v)) for k, v in kw.items()])
        out = {}
        for k, (dt, shape) in outputs.items():
            a = np.fromfile(os.path.join(d, "out_" + k + ".bin"), dtype=dt)
            out[k] = a.reshape(shape) if shape is not None else a
        out["log"] = open(os.path.join(d, "out_log.txt")).read()
        shutil.rmtree(d)
        return out

    def close(self):
        shutil.rmtree(self.work, ignore_errors=True)


def npz_bytes(arrays):
    """A .npz whose bytes depend only on the arrays (fixed member timestamps, sorted members, fixed compression)."""
    buf = io.BytesIO()
    with zipfile.ZipFile(buf, "w", zipfile.ZIP_DEFLATED, compresslevel=9) as z:
        for k in sorted(arrays):
            a = arrays[k]
            if isinstance(a, str):
                a = np.frombuffer(a.encode(), np.uint8)
            b = io.BytesIO()
            np.lib.format.write_array(b, np.ascontiguousarray(a), version=(1, 0), allow_pickle=False)
            info = zipfile.ZipInfo(k + ".npy", date_time=(1980, 1, 1, 0, 0, 0))
            info.compress_type, info.external_attr = zipfile.ZIP_DEFLATED, 0o644 << 16
            z.writestr(info, b.getvalue(), compresslevel=9)
    return buf.getvalue()


def vol_shape(dims):
    return (dims[2], dims[1], dims[0], 2)


def fld_shape(dims):
    return (dims[2], dims[1], dims[0], 4)


# ------------------------------------------------------------------------------------------------- scenarios
def stats(v):
    """(sum of tsdf in float64, sum of weights, voxels observed and not truncated) -- the figures SURVEY Appendix B quotes"""
    t, w = v[..., 0], v[..., 1]
    return np.array([t.astype(np.float64).sum(), float(w.astype(np.float64).sum()), float(((np.abs(t) < 1) & (w > 0)).sum())], np.float64)


def displacement_stats(psi):
    d = (psi - identity(psi.shape[2::-1]))[..., :3].astype(np.float64)
    return np.array([d.sum(), np.sqrt((d ** 2).sum()), np.sqrt((d ** 2).sum(-1)).max()], np.float64)


def compact(arrays, keep=()):
    """Digest form for grids too large to commit: every float array becomes sha256 + its statistics."""
    out = {}
    for k, v in arrays.items():
        if isinstance(v, str) or k in keep or k in ("params", "param_names"):
            out[k] = v
            continue
        out["sha256_" + k] = digest(v)
        if v.ndim == 4 and v.shape[-1] == 2:
            out["stats_" + k] = stats(v)
        elif v.ndim == 4 and v.shape[-1] == 4 and k.split("_f")[0] in ("psi", "psi_inv", "psi_after", "in_psi0"):
            out["stats_" + k] = displacement_stats(v)
    return out


def kernels_fixture(emu, dims, seed, amp, w_reg=0.6, alpha=0.1, max_weight=4.0):
    X, Y, Z = dims
    ins = kernel_inputs(dims, seed, amp)
    V, Fd, J = (np.float32, vol_shape(dims)), (np.float32, fld_shape(dims)), (np.float32, (Z, Y, X, 4, 4))
    outs = dict(grad=Fd, laplacian=Fd, jacobian0=J, jacobian1=J, nabla_U=Fd, conv_rows=Fd, conv_cols=Fd, conv_depth=Fd, psi_new=Fd, updates=Fd,
                warped=V, psi_inv=Fd, fused=V, scalars=(np.float32, (6,)))
    r = emu.run("kernels", ins, outs, X=X, Y=Y, Z=Z, w_reg=w_reg, alpha=alpha, max_weight=max_weight)
    r.pop("log")
    r.update({"in_" + k: v for k, v in ins.items()})
    r["params"] = np.array([seed, amp, w_reg, alpha, max_weight], np.float64)
    return r


def solver_fixture(emu, dims, phi_global, phi_n, psi0, iters, verbosity=2, per_iteration=False, **kw):
    X, Y, Z = dims
    P = dict(X=X, Y=Y, Z=Z, size_x=X * 0.004, size_y=Y * 0.004, size_z=Z * 0.004, trunc_vox=5.0, eta_vox=2.0, max_weight=64.0, s=7,
             max_update_norm=-1.0, verbosity=verbosity, max_iter=iters)
    P["lambda"] = 0.1
    P.update(kw)
    V, Fd = (np.float32, vol_shape(dims)), (np.float32, fld_shape(dims))
    ins = dict(phi_global=phi_global, phi_n=phi_n, psi0=psi0)
    outs = dict(psi=Fd, phi_n_psi=V, psi_inv=Fd, phi_global_psi_inv=V)
    r = emu.run("solver", ins, outs, **P)
    r.update({"in_" + k: v for k, v in ins.items()})
    r["params"] = np.array([P["max_iter"], P["alpha"], P["w_reg"], P["s"], P["lambda"], P["max_update_norm"], P["verbosity"]], np.float64)
    if per_iteration:  # the state after k iterations = the same deterministic run stopped at max_iter = k
        psis = []
        for k in range(1, iters):
            Pk = dict(P, max_iter=k, verbosity=0)
            psis.append(emu.run("solver", ins, dict(psi=Fd), **Pk)["psi"])
        r["psi_after"] = np.stack(psis + [r["psi"]])
    return r


def sphere_solver_fixture(emu, P, after=None):
    """Solver::estimate_psi on two initSphere volumes from the identity (the shape of the reference's own test/solver_test.cpp:109-132), in
    digest form.  after = k: also the state after k iterations (the same deterministic run stopped at max_iter = k)."""
    dims = (int(P["X"]), int(P["Y"]), int(P["Z"]))
    P = dict(P)
    P.setdefault("lambda", 0.1)
    V, Fd = (np.float32, vol_shape(dims)), (np.float32, fld_shape(dims))
    outs = dict(phi_global=V, phi_n=V, psi=Fd, phi_n_psi=V, psi_inv=Fd, phi_global_psi_inv=V)
    r = emu.run("solver", {}, outs, **P)
    X, Y = dims[0], dims[1]
    probe = (dims[2] // 2, Y // 2, X // 2 - 2)
    r["probe_psi"] = r["psi"][probe][None].copy()
    keep = ["probe_psi"]
    if after:
        ra = emu.run("solver", {}, dict(psi=Fd), **dict(P, max_iter=after, verbosity=0))
        r["psi_after%d" % after] = ra["psi"]
        r["probe_psi"] = np.stack([ra["psi"][probe], r["psi"][probe]])
    c = compact(r, keep=tuple(keep))
    if after:
        c["stats_psi_after%d" % after] = displacement_stats(r["psi_after%d" % after])
    c["params"] = np.array([P[k] for k in sorted(P)], np.float64)
    c["param_names"] = ",".join(sorted(P))
    return c


def solver_test_fixture(emu):
    """The set-up of the reference's own test/solver_test.cpp:109-132 (AlignmentTestSphereTranslation), cut to 10 iterations at
    verbosity 2: 64^3, size 0.25, trunc 10 vox, eta 2 vox, max weight 128, S=7, lambda 0.1, alpha 0.01, w_reg 0.4 -- SURVEY
    Appendix B run 1.  The state after 3 iterations comes from the same run stopped at max_iter = 3."""
    c = sphere_solver_fixture(emu, dict(X=64, Y=64, Z=64, size_x=0.25, size_y=0.25, size_z=0.25, trunc_vox=10.0, eta_vox=2.0, max_weight=128.0, s=7, alpha=0.01,
                                        w_reg=0.4, max_update_norm=-1.0, verbosity=2, max_iter=10, sphere_cx=0.13, sphere_cy=0.13, sphere_cz=0.13,
                                        sphere2_cx=0.125, sphere2_cy=0.13, sphere2_cz=0.13, sphere_r=0.012), after=3)
    c["probe_psi_30_32_32"] = c.pop("probe_psi")  # voxel (30, 32, 32): the one SURVEY Appendix B quotes
    return c


def tsdf_fixture(emu, dims):
    X, Y, Z = dims
    size = (0.30, 0.24, 0.18)
    P = dict(X=X, Y=Y, Z=Z, size_x=size[0], size_y=size[1], size_z=size[2], trunc_vox=4.0, eta_vox=2.0,
             sphere_cx=0.14, sphere_cy=0.125, sphere_cz=0.09, sphere_r=0.06, box_x=0.07, box_y=0.05, box_z=0.04,
             ell_x=0.09, ell_y=0.06, ell_z=0.05, plane_z=0.05, torus_R=0.07, torus_r=0.025)
    V = (np.float32, vol_shape(dims))
    r = emu.run("tsdf", {}, dict(sphere=V, box=V, ellipsoid=V, plane=V, torus=V), **P)
    r.pop("log")
    r["params"] = np.array([P[k] for k in sorted(P)], np.float64)
    r["param_names"] = ",".join(sorted(P))
    return r


DEPTH_P = dict(rows=96, cols=128, fx=114.0, fy=114.0, cx=64.0, cy=48.0, trunc_depth=0.9, bilateral_ksz=7, bilateral_ss=4.5, bilateral_sd=0.04)


def depth_fixture(emu, dims):
    X, Y, Z = dims
    P = dict(DEPTH_P, X=X, Y=Y, Z=Z, size_x=0.5, size_y=0.5, size_z=0.5, trunc_vox=5.0, eta_vox=2.0, t_z=0.5)
    intr = (P["fx"], P["fy"], P["cx"], P["cy"])
    depth = render_sphere_depth((0.01, -0.02, 0.75), 0.1, intr, rows=P["rows"], cols=P["cols"])
    depth[10:14, 20:30] = 1500  # beyond trunc_depth: removed by depthTruncation
    depth[60:62, 90:100] = 0    # holes inside the object
    u16, f32 = np.uint16, np.float32
    r = emu.run("depth", dict(depth=depth), dict(bilateral=(u16, (P["rows"], P["cols"])), truncated=(u16, (P["rows"], P["cols"])),
                                                  dists=(f32, (P["rows"], P["cols"])), volume=(f32, vol_shape(dims))), **P)
    r.pop("log")
    r["in_depth"] = depth
    r["params"] = np.array([P[k] for k in sorted(P)], np.float64)
    r["param_names"] = ",".join(sorted(P))
    return r


def frames_fixture(emu, P, depths, full_last=True):
    """SobFusion::operator() (sob_fusion.cpp:71-145) over a list of depth frames.  Every array of every frame in digest form; the last
    frame's arrays in full when full_last."""
    dims = (int(P["X"]), int(P["Y"]), int(P["Z"]))
    n = len(depths)
    P = dict(P, frames=n)
    ins = {"depth_%d" % f: d for f, d in enumerate(depths)}
    V, Fd = (np.float32, vol_shape(dims)), (np.float32, fld_shape(dims))
    outs = {}
    for f in range(n):
        outs["phi_global_f%d" % f] = V
        if f > 0:
            outs["phi_n_f%d" % f] = V
        if f >= max(1, int(P["start_frame"])):
            outs.update({"psi_f%d" % f: Fd, "psi_inv_f%d" % f: Fd, "phi_n_psi_f%d" % f: V, "phi_global_psi_inv_f%d" % f: V})
    r = emu.run("frames", ins, outs, **P)
    last = "_f%d" % (n - 1)
    c = compact(r, keep=tuple(k for k in r if full_last and k.endswith(last)))
    for k in list(c):
        if full_last and k.endswith(last) and not k.startswith(("sha256_", "stats_")):
            c["sha256_" + k] = digest(c[k])
    c.update({"in_" + k: v for k, v in ins.items()})
    c["params"] = np.array([P[k] for k in sorted(P)], np.float64)
    c["param_names"] = ",".join(sorted(P))
    return c


def mc_fixture(emu, dims, buffer=3 * 4096, as_digest=False):
    X, Y, Z = dims
    size = (0.28, 0.22, 0.18)
    vol = mc_volume(dims)
    P = dict(X=X, Y=Y, Z=Z, size_x=size[0], size_y=size[1], size_z=size[2], trunc_vox=4.0, eta_vox=2.0, t_z=0.4, buffer=buffer)
    r = emu.run("mc", dict(volume=vol), dict(vertices=(np.float32, None), normals=(np.float32, None)), **P)
    r["vertices"], r["normals"] = r["vertices"].reshape(-1, 4), r["normals"].reshape(-1, 4)
    if as_digest:  # the SET of triangles (the reference's order is run-dependent): rows of 3 x (vertex, normal), sorted by their bit patterns
        a = np.concatenate([r.pop("vertices").reshape(-1, 3, 4), r.pop("normals").reshape(-1, 3, 4)], axis=2).reshape(-1, 24)
        b = a.view(np.uint32)
        r["n_triangles"] = np.array([len(a)], np.int64)
        r["sha256_triangle_set"] = digest(b[np.lexsort(b.T[::-1])])
        r["sha256_in_volume"] = digest(vol)
    else:
        r["in_volume"] = vol
    r["params"] = np.array([P[k] for k in sorted(P)], np.float64)
    r["param_names"] = ",".join(sorted(P))
    return r


def check_appendix_b(fx):
    """The numbers SURVEY.md Appendix B recorded (tests/golden/appendix_b.json) must fall out of this recipe again."""
    import json
    import re

    B = json.load(open(os.path.join(HERE, "appendix_b.json")))
    six = lambda x, ref: float("%.6g" % x) == float("%.6g" % ref)  # noqa: E731  (std::cout prints 6 significant digits)
    close = lambda x, ref: abs(x - ref) <= 5e-9 * max(1.0, abs(ref)) * 10  # noqa: E731  (the survey quoted 9 significant digits)

    def trace(log):
        e = [(float(a), float(b)) for a, b in re.findall(r"data energy \+ w_reg \* reg energy = (\S+) \+ \S+ \* (\S+) =", log)]
        m = [float(v) for v in re.findall(r"max\. update norm (\S+) at voxel", log)]
        return [(a, b, c) for (a, b), c in zip(e, m)]

    t = trace(fx["ref_solver_test_64"]["log"])
    for k, row in B["run1"]["trace"].items():
        assert all(six(a, b) for a, b in zip(t[int(k) - 1], row)), ("run 1 iteration", k, t[int(k) - 1], row)
    s3, s10 = fx["ref_solver_test_64"]["stats_psi_after3"], fx["ref_solver_test_64"]["stats_psi"]
    for s, ref in ((s3, B["run1"]["after3"]), (s10, B["run1"]["after10"])):
        assert close(s[0], ref["sum"]) and close(s[1], ref["l2"]) and close(s[2], ref["max"]), (s, ref)
    for row, ref in zip(fx["ref_solver_test_64"]["probe_psi_30_32_32"], (B["run1"]["after3"]["psi_30_32_32"], B["run1"]["after10"]["psi_30_32_32"])):
        assert all(close(float(a), b) for a, b in zip(row[:3], ref)), (row, ref)
    c1 = fx["ref_config1_64"]
    t = trace(c1["log"])
    assert len(t) == 10 and all(six(a, b) for row, ref in zip(t, B["run2"]["trace"]) for a, b in zip(row, ref)), t
    stat = lambda k, ref: abs(c1[k][0] - ref[0]) < 5e-5 and tuple(c1[k][1:]) == tuple(float(v) for v in ref[1:])  # noqa: E731
    assert stat("stats_phi_global_f0", B["run2"]["frame0_phi_global"]) and stat("stats_phi_n_f1", B["run2"]["frame1_phi_n"])
    a10 = B["run2"]["after10"]
    assert stat("stats_phi_n_psi_f1", a10["phi_n_psi"]) and stat("stats_phi_global_f1", a10["fused_phi_global"])
    assert stat("stats_phi_global_psi_inv_f1", a10["phi_global_psi_inv"])
    assert close(c1["stats_psi_f1"][1], a10["l2"]) and close(c1["stats_psi_f1"][2], a10["max"])


def make_some(emu, emu_smem, names):
    """a subset (the quick regeneration check of tests/test_reference_recipe.py): small fixtures only"""
    fx = {}
    for n in names:
        if n == "ref_kernels_17x9x5":
            fx[n] = kernels_fixture(emu, (17, 9, 5), 911, 1.4)
            smem = kernels_fixture(emu_smem, (17, 9, 5), 911, 1.4)
            assert all(np.array_equal(np.asarray(fx[n][k]).view(np.uint8), np.asarray(smem[k]).view(np.uint8)) for k in fx[n])
        elif n == "ref_solver_20x12x9":
            d = (20, 12, 9)
            fx[n] = solver_fixture(emu, d, rand_volume(d, 901), rand_volume(d, 902), warped_identity(d, 903, 0.7), 4, alpha=0.05, w_reg=0.4)
        elif n == "ref_mc_14x11x9":
            fx[n] = mc_fixture(emu, (14, 11, 9))
        elif n == "ref_mc_40x33x29":
            fx[n] = mc_fixture(emu, (40, 33, 29), buffer=3 * 16384, as_digest=True)
        elif n == "ref_depth_32x32x32":
            fx[n] = depth_fixture(emu, (32, 32, 32))
        elif n == "ref_config5_values_96":
            fx[n] = make_all(emu, emu_smem, only_config5=True)[n]
        else:
            raise SystemExit("--only knows ref_kernels_17x9x5, ref_solver_20x12x9, ref_mc_14x11x9, ref_depth_32x32x32")
    return fx


def _config5(emu, fx):
    # BASELINE config 5's parameter set (params_umbrella.ini values, params/config5_umbrella_512.ini) on a grid the emulation can run: 96^3
    # (the voxel-unit parameters follow the voxel size, as apps/sobfu_headless --dims does), bench.py's own depth sequence, 6 iterations
    # per frame.  The 512^3 size itself is covered HIP-vs-oracle (tests/test_gpu_configs.py::test_config5_512_vs_oracle).
    cfg5 = dict(rows=480, cols=640, fx=570.342, fy=570.342, cx=320.0, cy=240.0, trunc_depth=1.5, bilateral_ksz=7, bilateral_ss=4.5, bilateral_sd=0.04,
                X=96, Y=96, Z=96, size_x=1.0, size_y=1.0, size_z=1.0, trunc_vox=8.0, eta_vox=3.0, t_z=0.3, max_weight=128.0, start_frame=1, s=7,
                alpha=0.001, w_reg=0.2, max_iter=6, max_update_norm=1e-10, verbosity=2)
    cfg5["lambda"] = 0.1
    vx5 = float(F32(1.0) / F32(96))
    fx["ref_config5_values_96"] = frames_fixture(emu, cfg5, [bench_sequence_frame((570.342, 570.342, 320.0, 240.0), 1.0, 0.3, vx5, f) for f in range(3)],
                                                 full_last=False)
    k5 = "ref_config5_values_96"
    for k in [k for k in fx[k5] if k.startswith("in_depth")]:
        fx[k5]["sha256_" + k] = digest(fx[k5].pop(k))
    return fx


def make_all(emu, emu_smem, only_config5=False):
    fx = {}
    if only_config5:
        return _config5(emu, fx)
    # per-launcher outputs on the sizes of SURVEY Appendix B run 4 (odd sizes exercise every clamp / partial tile); the two larger
    # grids in digest form (their inputs are regenerated by tests/fixture_inputs.py from the seed in params)
    for dims, seed, amp, full in (((17, 9, 5), 911, 1.4, True), ((20, 12, 9), 921, 0.7, True), ((40, 24, 20), 931, 2.5, False), ((32, 32, 32), 941, 0.9, False)):
        name = "ref_kernels_%dx%dx%d" % dims
        r = kernels_fixture(emu, dims, seed, amp)
        smem = kernels_fixture(emu_smem, dims, seed, amp)  # __CUDA_ARCH__ undefined: the reductions' shared-memory tails
        for k in r:
            assert np.array_equal(np.asarray(r[k]).view(np.uint8), np.asarray(smem[k]).view(np.uint8)), (name, k, "arch 610 vs pre-Kepler branch")
        fx[name] = r if full else compact(r, keep=("scalars",))
    # whole Solver::estimate_psi: random volumes from a warped start (every clamp and lattice hit), then well-posed sphere pairs
    d = (17, 9, 5)
    fx["ref_solver_17x9x5"] = solver_fixture(emu, d, rand_volume(d, 951), rand_volume(d, 952), warped_identity(d, 953, 0.9), 10, alpha=0.05, w_reg=0.4,
                                             per_iteration=True)
    d = (20, 12, 9)
    fx["ref_solver_20x12x9"] = solver_fixture(emu, d, rand_volume(d, 901), rand_volume(d, 902), warped_identity(d, 903, 0.7), 4, alpha=0.05, w_reg=0.4)
    d = (40, 24, 20)
    fx["ref_solver_40x24x20"] = solver_fixture(emu, d, sphere_volume(d, (19.5, 12.0, 10.0), 6.0, 5.0), sphere_volume(d, (20.8, 11.6, 10.3), 6.0, 5.0), identity(d),
                                               6, alpha=0.05, w_reg=0.4)
    d = (32, 32, 32)
    fx["ref_solver_32x32x32"] = solver_fixture(emu, d, sphere_volume(d, (15.5, 16.0, 16.0), 8.0, 5.0), sphere_volume(d, (16.8, 16.3, 15.6), 8.0, 5.0), identity(d),
                                               8, alpha=0.1, w_reg=0.2, verbosity=1)
    # convergence break (max_update_norm reached at iteration k < max_iter) at verbosity 0
    d = (20, 12, 9)
    fx["ref_solver_break_20x12x9"] = solver_fixture(emu, d, sphere_volume(d, (9.5, 6.0, 4.5), 3.0, 4.0), sphere_volume(d, (10.2, 6.2, 4.4), 3.0, 4.0), identity(d),
                                                    40, alpha=0.1, w_reg=0.2, verbosity=0, max_update_norm=0.0105)
    fx["ref_solver_test_64"] = solver_test_fixture(emu)
    fx["ref_tsdf_30x24x18"] = tsdf_fixture(emu, (30, 24, 18))
    fx["ref_depth_32x32x32"] = depth_fixture(emu, (32, 32, 32))
    small = dict(DEPTH_P, X=32, Y=32, Z=32, size_x=0.5, size_y=0.5, size_z=0.5, trunc_vox=5.0, eta_vox=2.0, t_z=0.5, max_weight=64.0, start_frame=1, s=7,
                 alpha=0.1, w_reg=0.2, max_iter=12, max_update_norm=1e-4, verbosity=2)
    small["lambda"] = 0.1
    small_intr = (small["fx"], small["fy"], small["cx"], small["cy"])
    small_depths = [translating_sphere_frame(small_intr, f, rows=small["rows"], cols=small["cols"]) for f in range(3)]
    fx["ref_frames_32x32x32"] = frames_fixture(emu, small, small_depths)
    # START_FRAME = 2: frame 1 is fused without a solve (sob_fusion.cpp:136-139)
    fx["ref_frames_gated_32x32x32"] = frames_fixture(emu, dict(small, start_frame=2, max_iter=5), small_depths, full_last=False)
    # BASELINE config 1 = SURVEY 8(d) input 1 = Appendix B run 2: 64^3, 640 x 480, two frames, 10 iterations (params/config1_sphere_64.ini)
    cfg1 = dict(rows=480, cols=640, fx=570.342, fy=570.342, cx=320.0, cy=240.0, trunc_depth=1.5, bilateral_ksz=7, bilateral_ss=4.5, bilateral_sd=0.005,
                X=64, Y=64, Z=64, size_x=0.5, size_y=0.5, size_z=0.5, trunc_vox=5.0, eta_vox=2.0, t_z=0.5, max_weight=128.0, start_frame=1, s=7,
                alpha=0.1, w_reg=0.2, max_iter=10, max_update_norm=-1.0, verbosity=2)
    cfg1["lambda"] = 0.1
    cfg1_intr = (cfg1["fx"], cfg1["fy"], cfg1["cx"], cfg1["cy"])
    fx["ref_config1_64"] = frames_fixture(emu, cfg1, [translating_sphere_frame(cfg1_intr, f) for f in range(2)], full_last=False)
    # BASELINE config 2 as the ini states it: 128^3, params_snoopy.ini values (params/config2_snoopy_128.ini: MAX_ITER 2048, MAX_UPDATE_NORM
    # 1e-3), the 7-frame VolumeDeform-style sequence: frames 1 - 3 are fused without a solve (START_FRAME 4), frames 4 - 6 solve from a
    # warm-started psi until the threshold fires (after 612, 143 and 53 iterations)
    cfg2 = dict(rows=480, cols=640, fx=517.0, fy=517.0, cx=320.0, cy=240.0, trunc_depth=3.0, bilateral_ksz=7, bilateral_ss=4.5, bilateral_sd=0.01,
                X=128, Y=128, Z=128, size_x=0.9, size_y=0.9, size_z=0.9, trunc_vox=10.0, eta_vox=5.0, t_z=0.05, max_weight=128.0, start_frame=4, s=7,
                alpha=0.1, w_reg=0.2, max_iter=2048, max_update_norm=1e-3, verbosity=1)
    cfg2["lambda"] = 0.1
    fx["ref_config2_128"] = frames_fixture(emu, cfg2, [snoopy_frame((517.0, 517.0, 320.0, 240.0), f) for f in range(7)], full_last=False)
    _config5(emu, fx)
    for name in ("ref_config1_64", "ref_config2_128"):  # 640 x 480 inputs: regenerated by the tests from tests/fixture_inputs.py
        for k in [k for k in fx[name] if k.startswith("in_depth")]:
            fx[name]["sha256_" + k] = digest(fx[name].pop(k))
    # BASELINE config 3, the roofline config and bench.py's own workload: 256^3, params_boxing.ini solver values, 50 iterations from two
    # initSphere volumes 1.3 voxels apart (bench.boxing_params / sphere_pair), then the 48-sweep inverse and the canonical warp
    vs3 = float(F32(0.75) / F32(256))
    fx["ref_config3_256"] = sphere_solver_fixture(emu, dict(X=256, Y=256, Z=256, size_x=0.75, size_y=0.75, size_z=0.75, trunc_vox=48.0, eta_vox=3.0, max_weight=128.0,
                                                            s=7, alpha=0.001, w_reg=0.6, max_update_norm=1e-10, verbosity=1, max_iter=50, sphere_cx=0.375,
                                                            sphere_cy=0.375, sphere_cz=0.375, sphere2_cx=0.375 + 1.3 * vs3, sphere2_cy=0.375, sphere2_cz=0.375,
                                                            sphere_r=0.2))
    fx["ref_mc_14x11x9"] = mc_fixture(emu, (14, 11, 9))
    fx["ref_mc_40x33x29"] = mc_fixture(emu, (40, 33, 29), buffer=3 * 16384, as_digest=True)
    check_appendix_b(fx)
    return fx
def main():
    check = "--check" in sys.argv
    only = [a.split("=", 1)[1].split(",") for a in sys.argv if a.startswith("--only=")]
    emu, emu_smem = Emu("610"), Emu(None)
    try:
        fx = make_some(emu, emu_smem, only[0]) if only else make_all(emu, emu_smem)
    finally:
        emu.close(), emu_smem.close()
    bad = 0
    for name, arrays in sorted(fx.items()):
        data, path = npz_bytes(arrays), os.path.join(HERE, name + ".npz")
        if check:
            same = os.path.exists(path) and open(path, "rb").read() == data
            print("%-28s %8d B  %s" % (name, len(data), "identical" if same else "DIFFERS"))
            bad += not same
        else:
            open(path, "wb").write(data)
            print("%-28s %8d B  sha256 %s" % (name, len(data), hashlib.sha256(data).hexdigest()[:16]))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
