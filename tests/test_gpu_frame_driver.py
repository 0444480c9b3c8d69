"""Headless frame driver (apps/sobfu_headless.cpp = SobFusion::operator() counterpart over the C++ shells) on the GPU,
BASELINE config 1, against the known-answer values of SURVEY.md Appendix B run 2.

The depth pre-step runs the bilateral filter with the device's expf (the reference uses __expf): a few pixels may differ
by 1 mm from the oracle's libm version, so sums are compared with a small tolerance instead of bit for bit (the bit-exact
checks of every stage downstream of the filter are in test_gpu_parity.py)."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args):
    from sobfu_amd import build, build_host

    build.build_hip()
    exe = build_host.build_app()
    r = subprocess.run([exe, *args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


def _stats(out, name):
    return [(float(a), float(b), int(c)) for a, b, c in
            re.findall(rf"^{name}: sum_tsdf=(\S+) sum_weight=(\S+) non_truncated_observed=(\d+)", out, re.M)]


def test_config1_two_frames():
    out = _run(os.path.join(ROOT, "params", "config1_sphere_64.ini"), "--synthetic", "2", "--vverbose")
    assert "--- FRAME NO. 0 ---" in out and "--- FRAME NO. 1 ---" in out
    pg, pn = _stats(out, "phi_global"), _stats(out, "phi_n")
    # frame 0: phi_global = integrate(depth 0); Appendix B: -19581.2063 / 8468 / 2909
    assert abs(pg[0][0] + 19581.2063) < 0.5 and abs(pg[0][1] - 8468) <= 3 and abs(pg[0][2] - 2909) <= 6
    # frame 1: phi_n: -19552.2192 / 8470 / 2921
    assert abs(pn[0][0] + 19552.2192) < 0.5 and abs(pn[0][1] - 8470) <= 3 and abs(pn[0][2] - 2921) <= 6
    # energies printed by the solver, iteration 1 and 10 (Appendix B run 2): 1059.91 + 0.2 * 0, 255.322 + 0.2 * 234.173
    e = [(float(a), float(b)) for a, b in re.findall(r"data energy \+ w_reg \* reg energy = (\S+) \+ 0\.2 \* (\S+) =", out)]
    assert len(e) == 10
    assert abs(e[0][0] - 1059.91) < 0.5 and e[0][1] == 0.0
    assert abs(e[9][0] - 255.322) < 0.3 and abs(e[9][1] - 234.173) < 0.3
    norms = [float(v) for v in re.findall(r"max\. update norm (\S+) at voxel", out)]
    assert len(norms) == 10 and abs(norms[0] - 0.234661) < 2e-3 and abs(norms[9] - 0.0237178) < 5e-4
    assert "SOLVER REACHED MAX. NO. OF ITERATIONS WITHOUT CONVERGING" in out
    # after the solve: phi_n o psi -19567.7977 / 8561 / 4363; fused phi_global -19566.7973 / 17029 / 4531;
    # phi_global o psi^-1 -19571.7471 / 8601 / 4392
    s = _stats(out, "phi_n_psi")[0]
    assert abs(s[0] + 19567.7977) < 0.6 and abs(s[1] - 8561) <= 4 and abs(s[2] - 4363) <= 10
    assert abs(pg[1][0] + 19566.7973) < 0.8 and abs(pg[1][1] - 17029) <= 6 and abs(pg[1][2] - 4531) <= 12
    s = _stats(out, "phi_global_psi_inv")[0]
    assert abs(s[0] + 19571.7471) < 0.6 and abs(s[1] - 8601) <= 4 and abs(s[2] - 4392) <= 10


def test_start_frame_gating_and_unknown_keys(tmp_path):
    """START_FRAME > frame index fuses phi_n directly (sob_fusion.cpp:136-139); unknown keys (RHO_0) are ignored."""
    ini = tmp_path / "p.ini"
    src = open(os.path.join(ROOT, "params", "config1_sphere_64.ini")).read().replace("START_FRAME=1", "START_FRAME=2")
    ini.write_text(src + "\nRHO_0=1.0\nSOME_FUTURE_KEY=abc\n")
    out = _run(str(ini), "--synthetic", "3", "--max-iter", "3")
    assert out.count("solver: iterations=3") == 1          # only frame 2 runs the solver
    pg = _stats(out, "phi_global")
    assert pg[1][1] > pg[0][1] and pg[2][1] > pg[1][1]     # weights accumulate on every frame


def test_png_frames_and_npy_dumps(tmp_path):
    """The same two frames as 16-bit PNG files (what the reference's datasets ship): identical output to the built-in renderer;
    --dump writes loadable .npy fields whose statistics match the printed ones."""
    import numpy as np

    from sobfu_amd import synthetic
    from test_depth_io import png_bytes

    ini = os.path.join(ROOT, "params", "config1_sphere_64.ini")
    ref = _run(ini, "--synthetic", "2", "--max-iter", "4")
    intr = (570.342, 570.342, 320.0, 240.0)
    files = []
    for n in range(2):
        d = synthetic.render_sphere_depth((0.005 * n, 0.0, 0.75), 0.1, intr, 480, 640)
        p = tmp_path / f"frame{n}.png"
        p.write_bytes(png_bytes(d, [4, 1, 2] if n else [0]))
        files.append(str(p))
    dump = tmp_path / "dump"
    dump.mkdir()
    out = _run(ini, "--max-iter", "4", "--dump", str(dump), *files)
    strip = lambda s: [l for l in s.splitlines() if l.startswith(("phi_", "solver:"))]  # noqa: E731
    assert strip(out) == strip(ref)
    psi, psi_inv = np.load(dump / "psi.npy"), np.load(dump / "psi_inv.npy")
    assert psi.shape == (64, 64, 64, 4) and psi_inv.shape == psi.shape and psi.dtype == np.float32
    z, y, x = np.meshgrid(np.arange(64), np.arange(64), np.arange(64), indexing="ij")
    assert float(np.abs(psi[..., 0] - x).max()) < 2.0 and float(np.abs(psi[..., 0] - x).max()) > 1e-3
    assert float(np.abs(psi[..., 3]).max()) == 0.0
    pg = np.load(dump / "phi_global.npy")
    s = _stats(out, "phi_global")[-1]
    assert pg.shape == (64, 64, 64, 2) and abs(float(pg[..., 0].astype(np.float64).sum()) - s[0]) < 1e-2 and float(pg[..., 1].sum()) == s[1]
    for name in ("phi_n", "phi_n_psi", "phi_global_psi_inv"):
        assert np.load(dump / f"{name}.npy").shape == (64, 64, 64, 2)


def _read_vtk(path):
    import numpy as np

    lines = open(path).read().split("\n")
    assert lines[0].startswith("# vtk DataFile") and lines[2] == "ASCII" and lines[3] == "DATASET POLYDATA"
    n = int(lines[4].split()[1])
    pts = np.array([[float(v) for v in l.split()] for l in lines[5:5 + n]], np.float32)
    i = next(k for k, l in enumerate(lines) if l.startswith("POLYGONS"))
    nt = int(lines[i].split()[1])
    polys = np.array([[int(v) for v in l.split()] for l in lines[i + 1:i + 1 + nt]])
    return pts, polys


def test_mesh_files_match_oracle_marching_cubes(tmp_path):
    """--mesh: the .vtk polydata of phi_global after frame 1 = the oracle's marching cubes of the dumped volume, vertex for vertex."""
    import numpy as np

    import oracle

    ini = os.path.join(ROOT, "params", "config1_sphere_64.ini")
    out_dir = tmp_path / "out"
    out_dir.mkdir()
    out = _run(ini, "--synthetic", "2", "--max-iter", "4", "--mesh", str(out_dir), "--dump", str(out_dir))
    assert "no. of active voxels:" in out and "mesh phi_global:" in out and "mesh phi_global_psi_inv:" in out
    for name in ("phi_global_0", "phi_global_1", "phi_n_1", "phi_n_psi_1", "phi_global_psi_inv_1"):
        assert (out_dir / f"{name}.vtk").exists(), name
    pts, polys = _read_vtk(out_dir / "phi_global_1.vtk")
    assert len(pts) == 3 * len(polys) and np.array_equal(polys[:, 0], np.full(len(polys), 3))
    assert np.array_equal(polys[:, 1:].ravel(), np.arange(len(pts)))
    vol = np.load(out_dir / "phi_global.npy")
    # params: VOL_SIZE 0.5, volume pose = translate(-0.25, -0.25, VOL_POSE_T_Z = 0.5) (demo.cpp:73-74)
    v, _ = oracle.marching_cubes(vol, (0.5, 0.5, 0.5), np.eye(3, dtype=np.float32), (-0.25, -0.25, 0.5))
    assert len(v) == len(pts) > 3000
    assert np.array_equal(v[:, :3].view(np.uint32), pts.view(np.uint32))  # %.9g text round-trips float32 exactly
