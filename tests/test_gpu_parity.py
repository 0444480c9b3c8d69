"""GPU parity tests: HIP kernels (through the C ABI) vs the CPU oracle on identical inputs.

Bar: bit-exact for every per-iteration kernel and for the whole psi recurrence (all arithmetic is rn ops /
explicit fma); stated tolerances only where the reference itself uses approximate
device math (powf in init_sphere, __expf in the bilateral filter).
"""
import numpy as np
import pytest

from sobfu_amd.synthetic import hash_field, render_sphere_depth

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

SIZES = [(64, 64, 64), (40, 24, 20), (17, 9, 5), (70, 33, 19)]


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device; the product path has no CPU fallback")
    from sobfu_amd import ops as _ops

    return _ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def same(a, b):
    """bitwise equality (NaN-safe, distinguishes -0/+0)"""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def nmis(a, b):
    return int((np.ascontiguousarray(a).view(np.uint32) != np.ascontiguousarray(b).view(np.uint32)).sum())


def rand_field(dims, seed, scale=1.0):
    X, Y, Z = dims
    f = hash_field((Z, Y, X, 4), seed, scale)
    f[..., 3] = 0
    return f


def rand_volume(dims, seed):
    X, Y, Z = dims
    v = hash_field((Z, Y, X, 2), seed, 1.0)
    v[..., 1] = (hash_field((Z, Y, X), seed + 7) > 0).astype(np.float32)
    return v


def warped_identity(oracle, dims, seed, amp):
    X, Y, Z = dims
    psi = oracle.new_field(dims)
    oracle.init_identity(psi)
    psi[..., :3] += hash_field((Z, Y, X, 3), seed, amp)
    return psi


# ---------------------------------------------------------------------------------------------------
# per-kernel parity
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dims", SIZES)
def test_identity_gradient_laplacian_jacobian(ops, oracle, dims):
    X, Y, Z = dims
    psi_d = ops.new_field(dims)
    ops.init_identity(psi_d)
    ident = oracle.new_field(dims)
    oracle.init_identity(ident)
    assert same(host(psi_d), ident)

    vol = rand_volume(dims, 3)
    g_o = oracle.new_field(dims)
    oracle.tsdf_gradient(vol, g_o)
    g_d = ops.new_field(dims)
    ops.tsdf_gradient(dev(vol), g_d)
    assert same(host(g_d), g_o)

    psi = warped_identity(oracle, dims, 5, 0.7)
    L_o = oracle.new_field(dims)
    oracle.laplacian(psi, L_o)
    L_d = ops.new_field(dims)
    ops.laplacian(dev(psi), L_d)
    assert same(host(L_d), L_o)

    for mode in (0, 1):
        J_o = np.zeros((Z, Y, X, 4, 4), np.float32)
        oracle.jacobian(psi, J_o, mode)
        J_d = ops.new_jacobian(dims)
        ops.jacobian(dev(psi), J_d, mode)
        assert same(host(J_d), J_o)
        assert ops.reg_energy_sobolev(J_d) == oracle.reg_energy_sobolev(J_o)
        if mode == 1:
            assert ops.reg_energy_sobolev_from_psi(dev(psi)) == oracle.reg_energy_sobolev(J_o)


@pytest.mark.parametrize("dims", SIZES)
def test_potential_gradient_conv_update(ops, oracle, dims):
    a, b = rand_volume(dims, 11), rand_volume(dims, 12)
    grad, L = rand_field(dims, 13), rand_field(dims, 14, 3.0)
    nU_o = oracle.new_field(dims)
    oracle.potential_gradient(a, b, grad, L, nU_o, 0.6)
    nU_d = ops.new_field(dims)
    ops.potential_gradient(dev(a), dev(b), dev(grad), dev(L), nU_d, 0.6)
    assert same(host(nU_d), nU_o)

    S = oracle.sobolev_filter(7, 0.1)
    assert same(ops.sobolev_filter(7, 0.1), S)
    dst_o = oracle.new_field(dims)
    dst_d = ops.new_field(dims)
    for name in ("convolution_rows", "convolution_columns", "convolution_depth"):
        getattr(oracle, name)(dst_o, nU_o, S)
        getattr(ops, name)(dst_d, nU_d, S)
        assert same(host(dst_d), dst_o), name

    psi = warped_identity(oracle, dims, 15, 0.3)
    upd_o = oracle.new_field(dims)
    psi_d, upd_d = dev(psi), ops.new_field(dims)
    oracle.update_psi(psi, dst_o, upd_o, 0.001)
    ops.update_psi(psi_d, dst_d, upd_d, 0.001)
    assert same(host(psi_d), psi) and same(host(upd_d), upd_o)
    mo, md = oracle.max_update_norm(upd_o), ops.max_update_norm(upd_d)
    assert mo == md


@pytest.mark.parametrize("dims", SIZES)
def test_apply_inverse_fuse_energy(ops, oracle, dims):
    X, Y, Z = dims
    phi = rand_volume(dims, 21)
    # displacements large enough to cross voxels and to leave the volume (exercises the clamps), plus exact
    # lattice hits at 0 and dim-1 (the `hi = g` rule, utils.hpp:61-72)
    psi = warped_identity(oracle, dims, 22, 2.5)
    psi[0, 0, :, :3] = 0.0
    psi[-1, -1, :, 0] = X - 1
    psi[-1, -1, :, 1] = Y - 1
    psi[-1, -1, :, 2] = Z - 1
    out_o = oracle.new_volume(dims)
    oracle.apply(phi, out_o, psi)
    out_d = ops.new_volume(dims)
    ops.apply(dev(phi), out_d, dev(psi))
    assert same(host(out_d), out_o)

    inv_o = oracle.new_field(dims)
    oracle.init_identity(inv_o)
    oracle.estimate_inverse(psi, inv_o, 5)
    inv_d = ops.new_field(dims)
    ops.init_identity(inv_d)
    ops.estimate_inverse(dev(psi), inv_d, 5)
    assert same(host(inv_d), inv_o)

    g = rand_volume(dims, 23)
    g[..., 1] = np.floor(np.abs(hash_field((Z, Y, X), 24, 6.0)))
    n = rand_volume(dims, 25)
    n[::2, ::3, ::2, 0] = -1.0  # exercise the skip rule (tsdf_volume.cu:118)
    n[1::2, ::3, ::2, 0] = 0.0
    g_d = dev(g)
    oracle.integrate_fuse(g, n, 4.0)
    ops.integrate_fuse(g_d, dev(n), 4.0)
    assert same(host(g_d), g)

    assert ops.data_energy(dev(phi), dev(n)) == oracle.data_energy(phi, n)
    assert ops.reduce_config(X * Y * Z) == oracle.reduce_config(X * Y * Z)


def test_tsdf_builders(ops, oracle):
    dims = (64, 64, 64)
    size = np.float32(0.25)
    vs = np.array([size / np.float32(64)] * 3, np.float32)
    trunc, eta = np.float32(10) * vs[0], np.float32(2) * vs[0]
    # init_sphere: the reference calls powf(d, 2) (tsdf_volume.cu:262); HIP squares with a correctly rounded
    # multiply, the oracle calls libm powf -> values agree to a few ulp, weights (sdf > -eta) may flip only on ties
    o = oracle.new_volume(dims)
    oracle.init_sphere(o, vs, trunc, eta, (0.13, 0.13, 0.13), 0.012)
    d = ops.new_volume(dims)
    ops.init_sphere(d, vs, trunc, eta, (0.13, 0.13, 0.13), 0.012)
    hd = host(d)
    assert np.max(np.abs(hd[..., 0] - o[..., 0])) <= 4e-6
    assert int((hd[..., 1] != o[..., 1]).sum()) <= 2
    # the other primitives use only IEEE ops + fma: bit-exact
    for name, arg in (("init_box", (0.03, 0.02, 0.04)), ("init_ellipsoid", (0.05, 0.03, 0.04)), ("init_plane", 0.11),
                      ("init_torus", (0.05, 0.02))):
        o = oracle.new_volume(dims)
        getattr(oracle, name)(o, vs, trunc, arg)
        d = ops.new_volume(dims)
        getattr(ops, name)(d, vs, trunc, arg)
        assert same(host(d), o), name
    ops.clear_volume(d)
    assert not host(d).any()


def test_depth_pipeline(ops, oracle):
    """config-1 inputs: depth -> bilateral -> truncate -> dists -> integrate, GPU vs oracle."""
    intr = (570.342, 570.342, 320.0, 240.0)
    depth = render_sphere_depth((0.005, 0.0, 0.75), 0.1, intr)
    dd = dev(depth.view(np.int16))
    # bilateral: the reference uses __expf; oracle = libm expf, HIP = ocml expf -> allow rare +-1 mm flips
    b_o = oracle.bilateral(depth, 7, 4.5, 0.005)
    b_d = host(ops.bilateral_filter(dd, 7, 4.5, 0.005)).view(np.uint16)
    diff = np.abs(b_d.astype(np.int32) - b_o.astype(np.int32))
    assert diff.max() <= 1 and (diff != 0).mean() < 1e-3
    # from here on feed both sides the oracle's filtered image: everything is bit-exact
    t_o = b_o.copy()
    oracle.truncate_depth(t_o, 0.7)
    t_d = dev(b_o.view(np.int16))
    ops.truncate_depth(t_d, 0.7)
    assert np.array_equal(host(t_d).view(np.uint16), t_o)
    assert (t_o == 0).sum() > (b_o == 0).sum()
    oracle.truncate_depth(b_o, 1.5)
    dist_o = oracle.compute_dists(b_o, intr)
    dist_d = ops.compute_dists(dev(b_o.view(np.int16)), intr)
    assert same(host(dist_d), dist_o)
    dims = (64, 64, 64)
    size = np.float32(0.5)
    vs = np.array([size / np.float32(64)] * 3, np.float32)
    trunc, eta = np.float32(5) * vs[0], np.float32(2) * vs[0]
    R = np.eye(3, dtype=np.float32)
    t = np.array([-size / np.float32(2), -size / np.float32(2), 0.5], np.float32)
    v_o = oracle.new_volume(dims)
    oracle.integrate_depth(dist_o, v_o, vs, trunc, eta, R, t, intr)
    v_d = ops.new_volume(dims)
    ops.integrate_depth(dist_d, v_d, vs, trunc, eta, R, t, intr)
    assert same(host(v_d), v_o)
    assert int(((v_o[..., 0] != 0) | (v_o[..., 1] != 0)).sum()) == 34812


# ---------------------------------------------------------------------------------------------------
# fused passes == launcher sequence == oracle
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dims", SIZES + [(130, 37, 41)])
def test_fused_passes(ops, oracle, dims):
    pnp, pg, pn = rand_volume(dims, 31), rand_volume(dims, 32), rand_volume(dims, 33)
    psi = warped_identity(oracle, dims, 34, 1.5)
    S = oracle.sobolev_filter(7, 0.1)
    w_reg, alpha = 0.6, 0.1
    # oracle: the reference's launcher sequence
    grad, L, nU, nUS, upd = (oracle.new_field(dims) for _ in range(5))
    oracle.tsdf_gradient(pnp, grad)
    oracle.laplacian(psi, L)
    oracle.potential_gradient(pnp, pg, grad, L, nU, w_reg)
    nU_d = ops.new_field(dims)
    ops.fused_potential_gradient(dev(pnp), dev(pg), dev(psi), nU_d, w_reg)
    assert nmis(host(nU_d), nU) == 0

    oracle.convolution_rows(nUS, nU, S)
    oracle.convolution_columns(nUS, nU, S)
    oracle.convolution_depth(nUS, nU, S)
    psi_o = psi.copy()
    oracle.update_psi(psi_o, nUS, upd, alpha)
    out_o = oracle.new_volume(dims)
    oracle.apply(pn, out_o, psi_o)
    mo = oracle.max_update_norm(upd)[0]

    psi_d, out_d, upd_d = dev(psi), ops.new_volume(dims), ops.new_field(dims)
    md = ops.fused_smooth_update_apply(nU_d, psi_d, dev(pn), out_d, S, alpha, updates=upd_d)
    assert nmis(host(psi_d), psi_o) == 0
    assert nmis(host(upd_d), upd) == 0
    assert nmis(host(out_d), out_o) == 0
    assert md == mo
    # without the updates store
    psi_d2, out_d2 = dev(psi), ops.new_volume(dims)
    md2 = ops.fused_smooth_update_apply(nU_d, psi_d2, dev(pn), out_d2, S, alpha)
    assert same(host(psi_d2), psi_o) and same(host(out_d2), out_o) and md2 == mo


# ---------------------------------------------------------------------------------------------------
# the solver
# ---------------------------------------------------------------------------------------------------
def _run1_inputs(oracle):
    dims = (64, 64, 64)
    size = np.float32(0.25)
    vs = np.array([size / np.float32(64)] * 3, np.float32)
    trunc, eta = np.float32(10) * vs[0], np.float32(2) * vs[0]
    pg, pn = oracle.new_volume(dims), oracle.new_volume(dims)
    oracle.init_sphere(pg, vs, trunc, eta, (0.13, 0.13, 0.13), 0.012)
    oracle.init_sphere(pn, vs, trunc, eta, (0.125, 0.13, 0.13), 0.012)
    return dims, pg, pn


@pytest.mark.parametrize("verbosity", [0, 2])
def test_solver_estimate_psi_matches_oracle(ops, oracle, verbosity):
    """test/solver_test.cpp:109-132 set-up (SURVEY Appendix B run 1), 10 iterations, full estimate_psi."""
    dims, pg, pn = _run1_inputs(oracle)
    psi = oracle.new_field(dims)
    oracle.init_identity(psi)
    r = oracle.estimate_psi(pg, pn, psi, max_iter=10, alpha=0.01, w_reg=0.4, verbosity=2)

    sv = ops.Solver(dims, max_iter=10, alpha=0.01, w_reg=0.4, verbosity=verbosity)
    psi_d, psi_inv_d = ops.new_field(dims), ops.new_field(dims)
    ops.init_identity(psi_d)
    pnp_d, pgi_d = ops.new_volume(dims), ops.new_volume(dims)
    rep, hist = sv.estimate_psi(dev(pg), pgi_d, dev(pn), pnp_d, psi_d, psi_inv_d)
    assert rep.iterations == 10 and rep.converged == 0
    assert same(host(psi_d), psi)
    assert same(host(pnp_d), r["phi_n_psi"])
    assert same(host(psi_inv_d), r["psi_inv"])
    assert same(host(pgi_d), r["phi_global_psi_inv"])
    assert same(hist, r["trace"][:, 2])
    # warp-field L2 error vs the reference restatement (north-star bar: < 1e-5) -- here exactly 0
    assert float(np.sqrt(((host(psi_d) - psi).astype(np.float64) ** 2).sum())) == 0.0
    if verbosity == 2:
        assert rep.last_e_data == r["trace"][-1, 0] and rep.last_e_reg == r["trace"][-1, 1]
        assert rep.last_max_update_index == r["trace"][-1, 3]
        assert same(host(sv.updates()), r["updates"])
        assert any(l.startswith("data energy + w_reg * reg energy = 24.5457 + 0.4 * 0 = 24.5457") for l in sv.log_lines)
        assert any(l.startswith("max. update norm 0.000383393 at voxel") for l in sv.log_lines)
    assert sv.log_lines[0] == "iter. no. 1"
    assert sv.log_lines[-1] == "SOLVER REACHED MAX. NO. OF ITERATIONS WITHOUT CONVERGING"
    sv.close()


@pytest.mark.parametrize("case", ["120 iterations", "break in a quiet run", "break on a reporting iteration", "API format"])
def test_solver_verbosity1_fast_loop(ops, oracle, case):
    """verbosity 1 = the quiet loop between the reporting iterations 1, 50 k, max_iter (solver.cu:132-133,173): same lines, same arrays,
    same stopping iteration as the reference's loop, which evaluates everything on every iteration."""
    from test_reference_fixtures import expected_log

    dims, pg, pn = _run1_inputs(oracle)
    ident = oracle.new_field(dims)
    oracle.init_identity(ident)
    full = oracle.estimate_psi(pg, pn, ident.copy(), max_iter=120, alpha=0.01, w_reg=0.4, verbosity=2, inverse_iters=0)["trace"]
    assert np.all(np.diff(full[:, 2]) < 0)  # the max norms fall monotonically: a threshold between two of them fires at a known iteration
    thr = {"break in a quiet run": float(full[60:62, 2].mean()), "break on a reporting iteration": float(full[48:50, 2].mean())}.get(case, -1.0)
    psi = ident.copy()
    r = oracle.estimate_psi(pg, pn, psi, max_iter=120, alpha=0.01, w_reg=0.4, verbosity=1, max_update_norm=thr)
    assert r["iters"] == {"break in a quiet run": 62, "break on a reporting iteration": 50}.get(case, 120)
    sv = ops.Solver(dims, max_iter=120, alpha=0.01, w_reg=0.4, verbosity=1, max_update_norm=thr)
    sv.set_compact(case != "API format")
    psi_d, psi_inv_d, pnp_d, pgi_d = dev(ident), ops.new_field(dims), ops.new_volume(dims), ops.new_volume(dims)
    rep, hist = sv.estimate_psi(dev(pg), pgi_d, dev(pn), pnp_d, psi_d, psi_inv_d)
    assert rep.iterations == r["iters"] and bool(rep.converged) == (thr > 0)
    assert same(host(psi_d), psi) and same(host(pnp_d), r["phi_n_psi"]) and same(host(psi_inv_d), r["psi_inv"]) and same(host(pgi_d), r["phi_global_psi_inv"])
    assert same(hist, r["trace"][:, 2])
    assert "\n".join(sv.log_lines) + "\n" == expected_log(r["trace"], dims, 120, 0.4, thr, 1)
    n_reports = sum(l.startswith("data energy") for l in sv.log_lines)
    assert n_reports == {"break in a quiet run": 2, "break on a reporting iteration": 2}.get(case, 4)  # iterations 1, 50, 100, 120
    last = r["iters"]
    if last == 1 or last % 50 == 0 or last == 120:
        assert rep.last_max_update_index == r["trace"][last - 1, 3]
    else:
        assert np.isnan(rep.last_max_update_index)  # the reference prints no arg-max for that iteration either
    k = max(i for i in (1, 50, 100, 120) if i <= last) - 1
    assert rep.last_e_data == r["trace"][k, 0] and rep.last_e_reg == r["trace"][k, 1]
    sv.close()


def test_solver_convergence_break(ops, oracle):
    """max_update_norm > 0: the device-side gate must stop at exactly the reference's iteration."""
    dims, pg, pn = _run1_inputs(oracle)
    psi = oracle.new_field(dims)
    oracle.init_identity(psi)
    thr = 0.00038255  # between the max norms of iterations 3 and 4... (trace decreases monotonically)
    r = oracle.estimate_psi(pg, pn, psi, max_iter=70, alpha=0.01, w_reg=0.4, max_update_norm=thr, inverse_iters=48)
    assert 1 < r["iters"] < 70
    sv = ops.Solver(dims, max_iter=70, alpha=0.01, w_reg=0.4, max_update_norm=thr)
    psi_d, psi_inv_d = ops.new_field(dims), ops.new_field(dims)
    ops.init_identity(psi_d)
    pnp_d, pgi_d = ops.new_volume(dims), ops.new_volume(dims)
    rep, hist = sv.estimate_psi(dev(pg), pgi_d, dev(pn), pnp_d, psi_d, psi_inv_d)
    assert rep.iterations == r["iters"] and rep.converged == 1
    assert same(host(psi_d), psi) and same(host(pnp_d), r["phi_n_psi"]) and same(host(psi_inv_d), r["psi_inv"])
    assert sv.log_lines[-1] == f"SOLVER CONVERGED AFTER {r['iters']} ITERATIONS"
    sv.close()


def test_solver_serial_frames_warm_start(ops, oracle):
    """test/solver_test.cpp:162-208 SerialAlignmentTest shape: psi persists across two estimate_psi calls."""
    dims = (64, 64, 64)
    size = np.float32(0.25)
    vs = np.array([size / np.float32(64)] * 3, np.float32)
    trunc, eta = np.float32(10) * vs[0], np.float32(2) * vs[0]
    pg = oracle.new_volume(dims)
    oracle.init_sphere(pg, vs, trunc, eta, (0.13, 0.13, 0.13), 0.02)
    psi = oracle.new_field(dims)
    oracle.init_identity(psi)
    sv = ops.Solver(dims, max_iter=6, alpha=0.005, w_reg=0.4)
    psi_d, psi_inv_d = ops.new_field(dims), ops.new_field(dims)
    ops.init_identity(psi_d)
    pnp_d, pgi_d = ops.new_volume(dims), ops.new_volume(dims)
    for c in ((0.125, 0.13, 0.132), (0.123, 0.13, 0.132)):
        pn = oracle.new_volume(dims)
        oracle.init_sphere(pn, vs, trunc, eta, c, 0.02)
        r = oracle.estimate_psi(pg, pn, psi, max_iter=6, alpha=0.005, w_reg=0.4)
        sv.estimate_psi(dev(pg), pgi_d, dev(pn), pnp_d, psi_d, psi_inv_d)
        assert same(host(psi_d), psi) and same(host(pgi_d), r["phi_global_psi_inv"])
    sv.close()


@pytest.mark.parametrize("seed", range(12))
def test_solver_randomised_configurations(ops, oracle, seed):
    """Differential run of the whole estimate_psi on random shapes, parameters, filter choices, thresholds and code paths
    (quiet compact two-pass, verbose API-format, single-kernel iteration): everything bit-equal to the oracle."""
    rng = np.random.default_rng(1000 + seed)
    dims = tuple(int(v) for v in rng.integers(2, 48, 3))
    if seed % 4 == 0:
        dims = (int(rng.integers(60, 140)), int(rng.integers(2, 20)), int(rng.integers(2, 20)))
    s, lam = [(7, 0.1), (7, 0.05), (7, 0.2), (7, 0.4)][seed % 4]
    alpha, w_reg = float(rng.uniform(0.005, 0.08)), float(rng.uniform(0.0, 0.9))
    iters = int(rng.integers(1, 7))
    pg, pn = rand_volume(dims, 2000 + seed), rand_volume(dims, 3000 + seed)
    psi0 = warped_identity(oracle, dims, 4000 + seed, float(rng.uniform(0.0, 1.2)))
    psi = psi0.copy()
    r = oracle.estimate_psi(pg, pn, psi, max_iter=iters, alpha=alpha, w_reg=w_reg, s=s, lam=lam, verbosity=2, inverse_iters=48)
    norms = r["trace"][:, 2]
    thr = float(norms[len(norms) // 2]) if seed % 3 == 0 and iters > 1 else -1.0
    if thr >= 0:
        psi = psi0.copy()
        r = oracle.estimate_psi(pg, pn, psi, max_iter=iters, alpha=alpha, w_reg=w_reg, s=s, lam=lam, max_update_norm=thr, verbosity=2,
                                inverse_iters=48)
    for mode in ("quiet", "verbose", "api-format"):
        sv = ops.Solver(dims, max_iter=iters, alpha=alpha, w_reg=w_reg, s=s, lam=lam, max_update_norm=thr, verbosity=2 if mode == "verbose" else 0)
        if mode == "api-format":
            sv.set_compact(False)
        psi_d, psi_inv_d, pnp_d, pgi_d = dev(psi0), ops.new_field(dims), ops.new_volume(dims), ops.new_volume(dims)
        rep, hist = sv.estimate_psi(dev(pg), pgi_d, dev(pn), pnp_d, psi_d, psi_inv_d)
        assert rep.iterations == r["iters"], (mode, dims)
        assert same(host(psi_d), psi) and same(host(pnp_d), r["phi_n_psi"]), (mode, dims)
        assert same(host(psi_inv_d), r["psi_inv"]) and same(host(pgi_d), r["phi_global_psi_inv"]), (mode, dims)
        assert same(hist[:r["iters"]], r["trace"][:r["iters"], 2]), (mode, dims)
        sv.close()


def test_solver_rejects_bad_filter(ops):
    from sobfu_amd._lib import HipError

    with pytest.raises(HipError):
        ops.Solver((16, 16, 16), max_iter=1, alpha=0.1, w_reg=0.2, lam=0.3)
    with pytest.raises(HipError):
        ops.Solver((16, 16, 16), max_iter=1, alpha=0.1, w_reg=0.2, s=3)


# ---------------------------------------------------------------------------------------------------
# full-size (256^3) launcher kernels vs fused passes in the API format, and size-independent properties (the oracle itself is put
# against the bench's own code path at 256^3 in tests/test_gpu_configs.py::test_config3_256_bench_path_vs_oracle)
# ---------------------------------------------------------------------------------------------------
def test_full_size_256_fused_equals_launchers_and_properties(ops, oracle):
    dims = (256, 256, 256)
    X, Y, Z = dims
    g = torch.Generator(device="cuda").manual_seed(0)
    pnp = torch.rand((Z, Y, X, 2), device="cuda", generator=g) * 2 - 1
    pg = torch.rand((Z, Y, X, 2), device="cuda", generator=g) * 2 - 1
    pn = torch.rand((Z, Y, X, 2), device="cuda", generator=g) * 2 - 1
    psi = ops.new_field(dims)
    ops.init_identity(psi)
    psi[..., :3] += (torch.rand((Z, Y, X, 3), device="cuda", generator=g) - 0.5)
    S = oracle.sobolev_filter(7, 0.1)
    # launcher sequence (independent kernels) vs fused passes, bit for bit, on 16.7M voxels
    grad, L, nU, nUS, upd = (ops.new_field(dims) for _ in range(5))
    ops.tsdf_gradient(pnp, grad)
    ops.laplacian(psi, L)
    ops.potential_gradient(pnp, pg, grad, L, nU, 0.6)
    nU_f = ops.new_field(dims)
    ops.fused_potential_gradient(pnp, pg, psi, nU_f, 0.6)
    assert torch.equal(nU.view(torch.int32), nU_f.view(torch.int32))
    del grad, L
    ops.convolution_rows(nUS, nU, S)
    ops.convolution_columns(nUS, nU, S)
    ops.convolution_depth(nUS, nU, S)
    psi_l = psi.clone()
    ops.update_psi(psi_l, nUS, upd, 0.001)
    out_l = ops.new_volume(dims)
    ops.apply(pn, out_l, psi_l)
    m_l = ops.max_update_norm(upd)[0]
    psi_f, out_f = psi.clone(), ops.new_volume(dims)
    m_f = ops.fused_smooth_update_apply(nU_f, psi_f, pn, out_f, S, 0.001)
    assert torch.equal(psi_l.view(torch.int32), psi_f.view(torch.int32))
    assert torch.equal(out_l.view(torch.int32), out_f.view(torch.int32))
    assert m_l == m_f
    del nUS, upd, psi_l, out_l, psi_f, out_f, nU_f
    # DC gain 3 of the sum of three unit-sum filters, clamp-to-edge => exact everywhere up to
    # rounding of the tap sums; psi = identity warps phi onto itself exactly (lerp with t = 0)
    nU[...] = 0
    nU[..., 0] = 1.0
    ident = ops.new_field(dims)
    ops.init_identity(ident)
    psi_c, out_c = ident.clone(), ops.new_volume(dims)
    m = ops.fused_smooth_update_apply(nU, psi_c, pn, out_c, S, 0.5)
    u = (ident - psi_c)[..., 0]
    assert float((u - 1.5).abs().max()) < 1e-6 and float((ident - psi_c)[..., 1:].abs().max()) == 0.0
    assert abs(m - 1.5) < 1e-6
    out_i = ops.new_volume(dims)
    ops.apply(pn, out_i, ident)
    assert torch.equal(out_i.view(torch.int32), pn.view(torch.int32))


# ---------------------------------------------------------------------------------------------------
# multi-GPU slab kernels (sobfu_hip_tile_*) on one GPU: slabs with exchanged halos == full volume
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("compact", [False, True])
@pytest.mark.parametrize("grid", [(1, 1, 2), (1, 1, 3), (2, 1, 1), (1, 2, 1), (2, 2, 2), (3, 2, 1)])
def test_tile_kernels_match_full_volume(ops, oracle, grid, compact):
    """One iteration on every tile of a cut (z-slabs, x / y splits, 2 x 2 x 2), with the single nabla_U exchange emulated by
    slicing the full-volume result: the owned cells and their one-cell shells of psi / phi_n o psi, and the owned max, must
    equal the full-volume kernels.  The x and y shells run as thin (lane-per-cell) boxes."""
    from sobfu_amd import tiled

    dims = (70, 24, 36)
    X, Y, Z = dims
    pg, pn = rand_volume(dims, 41), rand_volume(dims, 42)
    psi0 = warped_identity(oracle, dims, 43, 1.2)
    S = oracle.sobolev_filter(7, 0.1)
    w_reg, alpha = 0.6, 0.1
    import tiled_reference

    be = tiled_reference.HipBackend(compact=compact)
    psi_f, pnp_f, nU_f = dev(psi0), ops.new_volume(dims), ops.new_field(dims)
    ops.apply(dev(pn), pnp_f, psi_f)
    ops.fused_potential_gradient(pnp_f, dev(pg), psi_f, nU_f, w_reg)
    m_full = ops.fused_smooth_update_apply(nU_f, psi_f, dev(pn), pnp_f, S, alpha)
    pn_d = dev(pn)
    m_tiles = 0.0
    ident = oracle.new_field(dims)
    oracle.init_identity(ident)
    for r in range(grid[0] * grid[1] * grid[2]):
        L = tiled.TileLayout(dims, grid, r)
        idl = torch.zeros(L.local_shape(4), device="cuda")
        be.init_identity(idl, L)
        assert same(host(idl), L.take(ident))
        psi_l, pg_l = (L.take(t).clone().contiguous() for t in (dev(psi0), dev(pg)))
        pnp_l = torch.zeros(L.local_shape(2), device="cuda")
        st = be.begin(L, pg_l, pn_d, pnp_l, psi_l)
        # pass A on the owned cells, in three z ranges (as an overlapped schedule issues them); one of them as a thin box
        ob = L.own_box()
        z0, z1 = ob[4], ob[5]
        be.pass_a(st, ob[:4] + (z0, z0 + 2), w_reg, None, 0.0)
        be.pass_a(st, ob[:4] + (z1 - 3, z1), w_reg, None, 0.0, thin=True)
        be.pass_a(st, ob[:4] + (z0 + 2, z1 - 3), w_reg, None, 0.0)
        nU_own = L.owned(st.nabla_U)[..., :3]
        assert torch.equal(nU_own.contiguous().view(torch.int32), L.owned_global(nU_f)[..., :3].contiguous().view(torch.int32))
        st.nabla_U[..., :3] = L.take(nU_f)[..., :3]  # the exchange: halo cells of nabla_U from their owners
        slots = torch.zeros(256, dtype=torch.int32, device="cuda")
        for box, tr in L.pass_b_boxes():
            mid = (box[4] + box[5]) // 2
            be.pass_b(st, box[:4] + (mid, box[5]), slots, S, alpha, None, 0.0, thin=tr)
            be.pass_b(st, box[:4] + (box[4], mid), slots, S, alpha, None, 0.0, thin=tr)
        be.end(st)
        torch.cuda.synchronize()
        # owned cells and the one-cell shell along each axis (cells on tile edges / corners are not part of the contract)
        ref_psi, ref_pnp = L.take(psi_f), L.take(pnp_f)
        checks = [L.own_box()]
        for a in range(3):
            for side, has in ((0, L.lo3[a]), (1, L.hi3[a])):
                if has:
                    bx = list(L.own_box())
                    bx[2 * a], bx[2 * a + 1] = (L.o0[a] - 1, L.o0[a]) if side == 0 else (L.o1[a], L.o1[a] + 1)
                    checks.append(tuple(bx))
        assert len(checks) == 1 + sum(1 for a in range(3) for v in (L.lo3[a], L.hi3[a]) if v)
        for bx in checks:
            cut = lambda t: t[bx[4]:bx[5], bx[2]:bx[3], bx[0]:bx[1]].contiguous().view(torch.int32)  # noqa: E731
            assert torch.equal(cut(psi_l), cut(ref_psi)), (r, bx)
            assert torch.equal(cut(pnp_l), cut(ref_pnp)), (r, bx)
        m_tiles = max(m_tiles, tiled._sqrt_rd(int(slots.max().cpu().numpy().view(np.uint32))))
    assert m_tiles == m_full


def test_tile_message_pack_unpack(ops):
    """sobfu_hip_tile3_pack / unpack: the send boxes of a tile, packed and scattered into the matching recv boxes of its
    neighbours, reproduce the full field in every halo cell a message covers (2 x 2 x 2 and an interior tile of 3 x 3 x 3)"""
    import ctypes as C

    from sobfu_amd import _lib, tiled

    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for dims, grid in (((40, 24, 36), (2, 2, 2)), ((36, 30, 33), (3, 3, 3))):
        X, Y, Z = dims
        full = torch.rand((Z, Y, X, 3), device="cuda")
        lays = [tiled.TileLayout(dims, grid, r) for r in range(grid[0] * grid[1] * grid[2])]
        for L in lays:
            field = torch.zeros(L.local_shape(3), device="cuda")
            L.owned(field).copy_(L.owned_global(full))
            for peer, _, rbox in L.messages():
                Lp = lays[peer]
                back = [m for m in Lp.messages() if m[0] == L.rank][0]
                src = Lp.take(full).clone().contiguous()
                n = (back[1][1] - back[1][0]) * (back[1][3] - back[1][2]) * (back[1][5] - back[1][4])
                buf = torch.zeros(3 * n, device="cuda")
                _lib.check(lib.sobfu_hip_tile3_pack(C.c_void_p(src.data_ptr()), *Lp.L, C.c_void_p(buf.data_ptr()), (C.c_int * 6)(*back[1]), 1, st), "pack")
                _lib.check(lib.sobfu_hip_tile3_unpack(C.c_void_p(field.data_ptr()), *L.L, C.c_void_p(buf.data_ptr()), (C.c_int * 6)(*rbox), 1, st), "unpack")
            want = L.take(full)
            covered = torch.zeros(L.local_shape(), dtype=torch.bool, device="cuda")
            L.owned(covered)[...] = True
            for _, _, rb in L.messages():
                covered[rb[4]:rb[5], rb[2]:rb[3], rb[0]:rb[1]] = True
            assert torch.equal(field[covered], want[covered])
            assert float(field[~covered].abs().max() if (~covered).any() else 0.0) == 0.0  # corners stay untouched
        # all messages of a tile in ONE launch, as the loop issues them
        L = lays[len(lays) // 2]
        msgs = L.messages()
        field = L.take(full).clone().contiguous()
        cells = [(m[1][1] - m[1][0]) * (m[1][3] - m[1][2]) * (m[1][5] - m[1][4]) for m in msgs]
        buf = torch.zeros(3 * sum(cells), device="cuda")
        boxes = (C.c_int * (6 * len(msgs)))(*[v for m in msgs for v in m[1]])
        _lib.check(lib.sobfu_hip_tile3_pack(C.c_void_p(field.data_ptr()), *L.L, C.c_void_p(buf.data_ptr()), boxes, len(msgs), st), "pack")
        off = 0
        for m, n in zip(msgs, cells):
            sb = m[1]
            assert torch.equal(buf[3 * off:3 * (off + n)], field[sb[4]:sb[5], sb[2]:sb[3], sb[0]:sb[1]].reshape(-1))
            off += n


# ---------------------------------------------------------------------------------------------------
# edge sizes: the smallest legal volume, a volume thinner than one tile / one z-chunk, and the 512^3 grid of
# BASELINE config 5 (maximum size; int32 indices, 2 GiB fields)
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dims", [(2, 2, 2), (3, 2, 5), (65, 9, 2), (5, 70, 3)])
def test_tiny_and_thin_volumes(ops, oracle, dims):
    pg, pn = rand_volume(dims, 51), rand_volume(dims, 52)
    psi = warped_identity(oracle, dims, 53, 0.8)
    r = oracle.estimate_psi(pg, pn, psi, max_iter=3, alpha=0.05, w_reg=0.4, inverse_iters=48)
    for compact in (True, False):
        sv = ops.Solver(dims, max_iter=3, alpha=0.05, w_reg=0.4)
        sv.set_compact(compact)
        psi_d, psi_inv_d = dev(warped_identity(oracle, dims, 53, 0.8)), ops.new_field(dims)
        pnp_d, pgi_d = ops.new_volume(dims), ops.new_volume(dims)
        rep, hist = sv.estimate_psi(dev(pg), pgi_d, dev(pn), pnp_d, psi_d, psi_inv_d)
        assert rep.iterations == 3
        assert same(host(psi_d), psi) and same(host(pnp_d), r["phi_n_psi"])
        assert same(host(psi_inv_d), r["psi_inv"]) and same(host(pgi_d), r["phi_global_psi_inv"])
        assert same(hist, r["trace"][:, 2])
        sv.close()


def test_solver_psi_w_lane_untouched(ops, oracle):
    """update_psi leaves psi.w alone (utils.hpp:260-265); the compact path must write back xyz only."""
    dims = (20, 12, 9)
    pg, pn = rand_volume(dims, 61), rand_volume(dims, 62)
    psi0 = warped_identity(oracle, dims, 63, 0.5)
    psi0[..., 3] = 7.25
    sv = ops.Solver(dims, max_iter=2, alpha=0.05, w_reg=0.4)
    psi_d, pnp_d = dev(psi0), ops.new_volume(dims)
    sv.iterate(dev(pg), dev(pn), pnp_d, psi_d, 2)
    out = host(psi_d)
    assert np.all(out[..., 3] == 7.25) and not np.array_equal(out[..., :3], psi0[..., :3])
    sv.close()


def test_max_size_512(ops, oracle):
    """512^3 (BASELINE config 5 grid): fused passes through the solver handle, checked by size-independent
    properties -- zero data term + identity psi is a fixed point; a constant nabla_U shifts psi by alpha*3*c."""
    dims = (512, 512, 512)
    free, _ = torch.cuda.mem_get_info()
    if free < 20 * 2 ** 30:
        pytest.skip("needs ~14 GiB of HBM")
    vs = np.array([np.float32(1.0 / 512)] * 3, np.float32)
    pg = ops.new_volume(dims)
    ops.init_sphere(pg, vs, np.float32(24) * vs[0], np.float32(3) * vs[0], (0.5, 0.5, 0.5), 0.3)
    psi, pnp = ops.new_field(dims), ops.new_volume(dims)
    ops.init_identity(psi)
    sv = ops.Solver(dims, max_iter=2, alpha=0.1, w_reg=0.2)
    rep, hist = sv.iterate(pg, pg, pnp, psi, 2)  # phi_n == phi_global, psi = id: nabla_U == 0 exactly
    assert rep.iterations == 2 and float(hist.max()) == 0.0
    ident = ops.new_field(dims)
    ops.init_identity(ident)
    assert torch.equal(psi.view(torch.int32), ident.view(torch.int32))
    assert torch.equal(pnp.view(torch.int32), pg.view(torch.int32))
    del ident
    sv.close()
    S = oracle.sobolev_filter(7, 0.1)
    nU = ops.new_field(dims)
    nU[..., 1] = 2.0
    m = ops.fused_smooth_update_apply(nU, psi, pg, pnp, S, 0.25)
    assert abs(m - 1.5) < 1e-6
    y = torch.arange(512, device="cuda", dtype=torch.float32).view(1, 512, 1)
    assert float((psi[..., 1] - (y - 1.5)).abs().max()) < 1e-4
    assert float(psi[-1, -1, -1, 0]) == 511.0 and float(psi[-1, -1, -1, 2]) == 511.0


@pytest.mark.parametrize("amp", [0.05, 0.45, 1.7])
def test_inverse_early_exit_is_exact(ops, oracle, amp):
    """The one-kernel inverse stops a lane once its fixed-point iteration has provably entered a period-1 or period-2 cycle:
    every sweep count (odd and even, before and after the cycle starts) must give the oracle's plain n-sweep result."""
    dims = (33, 18, 11)
    X, Y, Z = dims
    psi = oracle.new_field(dims)
    oracle.init_identity(psi)
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    smooth = np.stack([np.sin(0.4 * y + 0.3 * z), np.cos(0.5 * x + 0.2 * z), np.sin(0.3 * x + 0.6 * y)], -1)
    psi[..., :3] += (amp * (0.7 * smooth + 0.3 * hash_field((Z, Y, X), 31, 1.0)[..., None])).astype(np.float32)
    psi_d = dev(psi)
    cycles = 0
    for n in (0, 1, 2, 3, 6, 7, 20, 47, 48, 49):
        inv_o = oracle.new_field(dims)
        oracle.init_identity(inv_o)
        inv_o[..., 3] = 5.0  # a w lane that only n = 0 may keep
        inv_d = dev(inv_o)
        oracle.estimate_inverse(psi, inv_o, n)
        ops.estimate_inverse(psi_d, inv_d, n)
        assert same(host(inv_d), inv_o), n
        if n == 48:
            a = oracle.new_field(dims)
            oracle.init_identity(a)
            oracle.estimate_inverse(psi, a, 47)
            cycles = int((a[..., :3].view(np.uint32) != inv_o[..., :3].view(np.uint32)).any(-1).sum())
    if amp >= 1.0:
        assert cycles > 0  # the rough field really has lanes that never settle on a fixed point


def test_native_tiled_loop_single_rank(ops, oracle):
    """sobfu_hip_tiled_* (C++ loop + RCCL) with one rank: identical to the solver handle; the RCCL entry points it uses
    (communicator bootstrap, grouped send/recv, MAX all-reduce) are exercised on the real library through self-transfers."""
    import ctypes as C

    from sobfu_amd import _lib, tiled

    dims = (40, 24, 20)
    pg, pn = rand_volume(dims, 71), rand_volume(dims, 72)
    psi0 = warped_identity(oracle, dims, 73, 0.6)
    ref = ops.Solver(dims, max_iter=5, alpha=0.05, w_reg=0.4)
    psi_r, pnp_r = dev(psi0), ops.new_volume(dims)
    rep_r, hist_r = ref.iterate(dev(pg), dev(pn), pnp_r, psi_r, 5)
    nt = tiled.NativeTiledSolver(dims, alpha=0.05, w_reg=0.4)
    psi_n, pnp_n = dev(psi0), ops.new_volume(dims)
    done, hist_n = nt.iterate(dev(pg), dev(pn), pnp_n, psi_n, 5)
    assert done == 5 and same(hist_n, hist_r)
    assert torch.equal(psi_n.view(torch.int32), psi_r.view(torch.int32))
    assert torch.equal(pnp_n.view(torch.int32), pnp_r.view(torch.int32))
    # identity slab, convergence gate
    assert torch.equal(nt.identity_psi().view(torch.int32), dev(oracle.new_field(dims) * 0 + 0).view(torch.int32)) is False
    L = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    src = torch.arange(3000, dtype=torch.float32, device="cuda")
    dst = torch.zeros_like(src)
    _lib.check(L.sobfu_hip_tiled_self_sendrecv(nt._h, C.c_void_p(src.data_ptr()), C.c_void_p(dst.data_ptr()), C.c_size_t(3000), st), "self")
    buf = torch.tensor([3, 9, 1, 7], dtype=torch.int32, device="cuda")
    _lib.check(L.sobfu_hip_tiled_allreduce_max_u32(nt._h, C.c_void_p(buf.data_ptr()), C.c_size_t(4), st), "allreduce")
    torch.cuda.synchronize()
    assert torch.equal(src, dst) and buf.tolist() == [3, 9, 1, 7]
    nt.close()
    ref.close()
    # positive threshold: same stopping iteration as the handle
    thr = float(hist_r[2])
    a = ops.Solver(dims, max_iter=5, alpha=0.05, w_reg=0.4, max_update_norm=thr)
    psi_a, pnp_a = dev(psi0), ops.new_volume(dims)
    rep_a, _ = a.iterate(dev(pg), dev(pn), pnp_a, psi_a, 5)
    b = tiled.NativeTiledSolver(dims, alpha=0.05, w_reg=0.4, max_update_norm=thr)
    psi_b, pnp_b = dev(psi0), ops.new_volume(dims)
    done_b, _ = b.iterate(dev(pg), dev(pn), pnp_b, psi_b, 5)
    assert done_b == rep_a.iterations < 5
    assert torch.equal(psi_b.view(torch.int32), psi_a.view(torch.int32))
    a.close()
    b.close()


def test_native_tiled_loop_comm_choreography_on_one_rank(ops, oracle, monkeypatch):
    """SOBFU_TILED_FORCE_COMM=1: the multi-rank stream/event schedule (comm stream, exchange group, all-reduce overlapped with
    the next iteration's ungated pass A) on a world of one -- results must not change, with and without a live threshold."""
    from sobfu_amd import tiled

    dims = (40, 24, 20)
    pg, pn = rand_volume(dims, 71), rand_volume(dims, 72)
    psi0 = warped_identity(oracle, dims, 73, 0.6)
    ref = ops.Solver(dims, max_iter=6, alpha=0.05, w_reg=0.4)
    psi_r, pnp_r = dev(psi0), ops.new_volume(dims)
    _, hist_r = ref.iterate(dev(pg), dev(pn), pnp_r, psi_r, 6)
    ref.close()
    monkeypatch.setenv("SOBFU_TILED_FORCE_COMM", "1")
    cases = [(thr, expect, own, sched) for thr, expect in ((-1.0, 6), (1e-10, 6), (float(hist_r[2]), 3))
             for own, sched in (("0", 0), ("force", 1), ("force", 3))]  # shared / own reduce communicator, overlapped / serial
    for thr, expect, own, sched in cases:
        monkeypatch.setenv("SOBFU_TILED_REDUCE_COMM", own)
        nt = tiled.NativeTiledSolver(dims, alpha=0.05, w_reg=0.4, max_update_norm=thr)
        nt.set_schedule(sched)
        psi_n, pnp_n = dev(psi0), ops.new_volume(dims)
        done, hist_n = nt.iterate(dev(pg), dev(pn), pnp_n, psi_n, 6)
        assert done == expect and same(hist_n[:done], hist_r[:done])
        if expect == 6:
            assert torch.equal(psi_n.view(torch.int32), psi_r.view(torch.int32))
            assert torch.equal(pnp_n.view(torch.int32), pnp_r.view(torch.int32))
        else:
            a = ops.Solver(dims, max_iter=6, alpha=0.05, w_reg=0.4, max_update_norm=thr)
            psi_a, pnp_a = dev(psi0), ops.new_volume(dims)
            a.iterate(dev(pg), dev(pn), pnp_a, psi_a, 6)
            assert torch.equal(psi_n.view(torch.int32), psi_a.view(torch.int32))
            a.close()
        nt.close()


# ---------------------------------------------------------------------------------------------------
# long marches on a small grid: the pass-B variant big grids run (halo requests three planes ahead through the LDS FIFO) is
# picked when a z-chunk has >= 24 planes -- forced here with the tuning override, on grids the oracle finishes in seconds,
# including chunk lengths that leave a short last chunk and a chunk that is the whole volume
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("pipe", ["0", "1"])
@pytest.mark.parametrize("dims,zc", [((70, 33, 80), 24), ((70, 33, 80), 31), ((40, 24, 96), 96), ((65, 9, 50), 27)])
def test_long_march_variant_on_small_grids(ops, oracle, dims, zc, pipe, monkeypatch):
    """the instantiations big grids run (halo lead, streaming hints; plain and software-pipelined march), forced onto small ones"""
    monkeypatch.setenv("SOBFU_ZC_B", str(zc))
    monkeypatch.setenv("SOBFU_CACHE_CELLS", "0")  # "does not fit the Infinity Cache"
    monkeypatch.setenv("SOBFU_PIPE_B", pipe)
    pg, pn = rand_volume(dims, 91), rand_volume(dims, 92)
    psi = warped_identity(oracle, dims, 93, 0.9)
    r = oracle.estimate_psi(pg, pn, psi, max_iter=4, alpha=0.05, w_reg=0.4, inverse_iters=0, compute_jacobian=False)
    sv = ops.Solver(dims, max_iter=4, alpha=0.05, w_reg=0.4)
    psi_d, pnp_d = dev(warped_identity(oracle, dims, 93, 0.9)), ops.new_volume(dims)
    _, hist = sv.iterate(dev(pg), dev(pn), pnp_d, psi_d, 4)
    assert nmis(host(psi_d), psi) == 0
    assert nmis(host(pnp_d), r["phi_n_psi"]) == 0
    assert same(hist, r["trace"][:, 2])
    sv.close()


@pytest.mark.parametrize("pipe", ["0", "1"])
@pytest.mark.parametrize("cache_cells", ["0", "1000000000"])
@pytest.mark.parametrize("dims", [(70, 33, 19), (40, 24, 20), (65, 9, 2), (130, 37, 41)])
def test_march_variants_agree_with_oracle(ops, oracle, dims, cache_cells, pipe, monkeypatch):
    """every (streaming hints, pipelined march) instantiation of the compact solver format against the oracle: psi, phi_n o psi and
    the max-norm history bit for bit (the pipelined march issues the phi_n gather one plane early and consumes it one plane late)"""
    monkeypatch.setenv("SOBFU_CACHE_CELLS", cache_cells)
    monkeypatch.setenv("SOBFU_PIPE_B", pipe)
    pg, pn = rand_volume(dims, 191), rand_volume(dims, 192)
    psi = warped_identity(oracle, dims, 193, 1.1)
    r = oracle.estimate_psi(pg, pn, psi, max_iter=4, alpha=0.05, w_reg=0.4, inverse_iters=0, compute_jacobian=False)
    sv = ops.Solver(dims, max_iter=4, alpha=0.05, w_reg=0.4)
    psi_d, pnp_d = dev(warped_identity(oracle, dims, 193, 1.1)), ops.new_volume(dims)
    _, hist = sv.iterate(dev(pg), dev(pn), pnp_d, psi_d, 4)
    assert nmis(host(psi_d), psi) == 0
    assert nmis(host(pnp_d), r["phi_n_psi"]) == 0
    assert same(hist, r["trace"][:, 2])
    sv.close()


# ---------------------------------------------------------------------------------------------------
# the loop in pieces (sobfu_hip_solver_begin / step / end): same bits as iterate(), whatever the step sizes
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dims", [(64, 64, 64), (40, 24, 20), (17, 9, 5), (70, 33, 19), (2, 2, 2), (65, 9, 2)])
@pytest.mark.parametrize("compact", [True, False])
def test_session_begin_step_end(ops, oracle, dims, compact):
    pg, pn = rand_volume(dims, 81), rand_volume(dims, 82)
    psi = warped_identity(oracle, dims, 83, 0.9)
    r = oracle.estimate_psi(pg, pn, psi, max_iter=5, alpha=0.05, w_reg=0.4, inverse_iters=0, compute_jacobian=False)
    sv = ops.Solver(dims, max_iter=5, alpha=0.05, w_reg=0.4)
    sv.set_compact(compact)
    psi_d, pnp_d = dev(warped_identity(oracle, dims, 83, 0.9)), ops.new_volume(dims)
    sv.begin(dev(pg), dev(pn), pnp_d, psi_d, 5)
    sv.step(2)
    sv.step(0)
    sv.step(3)
    rep, hist = sv.end()
    assert rep.iterations == 5 and rep.converged == 0
    assert nmis(host(psi_d), psi) == 0
    assert nmis(host(pnp_d), r["phi_n_psi"]) == 0
    assert same(hist, r["trace"][:, 2])
    # fewer iterations than the session's capacity: the state after exactly 3
    psi3 = warped_identity(oracle, dims, 83, 0.9)
    r3 = oracle.estimate_psi(pg, pn, psi3, max_iter=3, alpha=0.05, w_reg=0.4, inverse_iters=0, compute_jacobian=False)
    psi_d3, pnp_d3 = dev(warped_identity(oracle, dims, 83, 0.9)), ops.new_volume(dims)
    sv.begin(dev(pg), dev(pn), pnp_d3, psi_d3, 5)
    sv.step(1)
    sv.step(2)
    rep3, hist3 = sv.end()
    assert rep3.iterations == 3 and nmis(host(psi_d3), psi3) == 0 and nmis(host(pnp_d3), r3["phi_n_psi"]) == 0 and same(hist3, r3["trace"][:, 2])
    # convergence break inside a step: later launches are device-side no-ops, end() reports the reference's iteration
    thr = float(r["trace"][2, 2])
    psi2 = warped_identity(oracle, dims, 83, 0.9)
    r2 = oracle.estimate_psi(pg, pn, psi2, max_iter=5, alpha=0.05, w_reg=0.4, max_update_norm=thr, inverse_iters=0, compute_jacobian=False)
    sv2 = ops.Solver(dims, max_iter=5, alpha=0.05, w_reg=0.4, max_update_norm=thr)
    sv2.set_compact(compact)
    psi_d2, pnp_d2 = dev(warped_identity(oracle, dims, 83, 0.9)), ops.new_volume(dims)
    sv2.begin(dev(pg), dev(pn), pnp_d2, psi_d2, 5)
    sv2.step(4)
    sv2.step(1)
    rep2, _ = sv2.end()
    assert rep2.iterations == r2["iters"] and rep2.converged == 1
    assert nmis(host(psi_d2), psi2) == 0 and nmis(host(pnp_d2), r2["phi_n_psi"]) == 0
    # misuse: a second begin on an open session, step beyond the capacity, end without begin
    from sobfu_amd._lib import HipError

    sv.begin(dev(pg), dev(pn), pnp_d3, psi_d3, 2)
    with pytest.raises(HipError):
        sv.begin(dev(pg), dev(pn), pnp_d3, psi_d3, 2)
    with pytest.raises(HipError):
        sv.step(3)
    sv.end()
    with pytest.raises(HipError):
        sv.end()
    sv.close()
    sv2.close()


# ---------------------------------------------------------------------------------------------------
# the windowed per-frame tail of a tile (sobfu_hip_tile3_*_window): same bits as the whole-volume kernels when the window is wide
# enough, and a raised flag -- not a wild read -- when it is not
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dims,grid,rank", [((40, 24, 36), (2, 2, 2), 5), ((33, 17, 16), (1, 2, 2), 0), ((70, 33, 23), (2, 1, 1), 1)])
def test_window_tail_kernels_and_their_guard(ops, oracle, dims, grid, rank):
    import ctypes as C

    from sobfu_amd import _lib, tiled

    lib = _lib.lib()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    I6 = C.c_int * 6
    amp = 2.6
    psi = warped_identity(oracle, dims, 401, amp)
    pg = rand_volume(dims, 402)
    psi_d, pg_d = dev(psi), dev(pg)
    L = tiled.TileLayout(dims, grid, rank)
    own = I6(*L.own_box())
    # reference: the whole-volume tile kernels
    inv_ref, pgi_ref = torch.zeros(L.local_shape(4), device="cuda"), torch.zeros(L.local_shape(2), device="cuda")
    _lib.check(lib.sobfu_hip_tile3_init_identity(C.c_void_p(inv_ref.data_ptr()), *L.L, *L.base, st), "id")
    _lib.check(lib.sobfu_hip_tile3_estimate_inverse(C.c_void_p(psi_d.data_ptr()), *dims, C.c_void_p(inv_ref.data_ptr()), *L.L, *L.base, C.c_int(48), st), "inv")
    _lib.check(lib.sobfu_hip_tile3_apply(C.c_void_p(pg_d.data_ptr()), *dims, C.c_void_p(pgi_ref.data_ptr()), C.c_void_p(inv_ref.data_ptr()), *L.L, st), "apply")
    # the reach the tail measures for itself
    bits = torch.zeros(1, dtype=torch.int32, device="cuda")
    psi_l = L.take(psi_d).clone().contiguous()
    _lib.check(lib.sobfu_hip_tile3_max_displacement(C.c_void_p(psi_l.data_ptr()), *L.L, *L.base, own, C.c_void_p(bits.data_ptr()), st), "maxdisp")
    r = float(np.array([bits.item()], np.int32).view(np.float32)[0])
    want_r = float(np.abs(L.owned_global(torch.from_numpy(psi))[..., :3].numpy()
                          - np.stack(np.meshgrid(np.arange(L.g0[2], L.g1[2]), np.arange(L.g0[1], L.g1[1]), np.arange(L.g0[0], L.g1[0]), indexing="ij")[::-1], -1)).max())
    assert r == want_r and amp * 0.9 < r <= amp

    def run(w):
        wb = L.window_box(w)
        cut = lambda t: t[wb[4]:wb[5], wb[2]:wb[3], wb[0]:wb[1]].contiguous()  # noqa: E731
        psi_w, pg_w = cut(psi_d), cut(pg_d)
        win = I6(wb[1] - wb[0], wb[3] - wb[2], wb[5] - wb[4], wb[0], wb[2], wb[4])
        inv, pgi = torch.full(L.local_shape(4), -7.0, device="cuda"), torch.full(L.local_shape(2), -7.0, device="cuda")
        viol = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(lib.sobfu_hip_tile3_estimate_inverse_window(C.c_void_p(psi_w.data_ptr()), win, *dims, C.c_void_p(inv.data_ptr()), *L.L, *L.base, own,
                                                               C.c_int(48), C.c_void_p(viol.data_ptr()), st), "inv_w")
        _lib.check(lib.sobfu_hip_tile3_apply_window(C.c_void_p(pg_w.data_ptr()), win, *dims, C.c_void_p(pgi.data_ptr()), C.c_void_p(inv.data_ptr()), *L.L, own,
                                                    C.c_void_p(viol.data_ptr()), st), "apply_w")
        return inv, pgi, int(viol.item())

    inv, pgi, viol = run(int(np.ceil(r)) + 2)
    assert viol == 0
    assert torch.equal(L.owned(inv)[..., :3].contiguous().view(torch.int32), L.owned(inv_ref)[..., :3].contiguous().view(torch.int32))
    assert torch.equal(L.owned(pgi).contiguous().view(torch.int32), L.owned(pgi_ref).contiguous().view(torch.int32))
    outside = torch.ones(L.local_shape(), dtype=torch.bool, device="cuda")
    L.owned(outside)[...] = False
    assert bool((inv[outside] == -7.0).all()) and bool((pgi[outside] == -7.0).all())  # cells outside the box are left alone
    if min(L.lo3 + L.hi3) >= 0 and any(L.lo3) or any(L.hi3):
        _, _, viol = run(1)  # a window narrower than the reach: samples leave it -> the flag, never a read outside the array
        assert viol == 1
