"""Closed-form known answers for the hot path -- convolution, psi update, trilinear warp, gradient / Laplacian / potential gradient --
derived in float64 from the MATHEMATICS of the operators (reference src/sobfu/cuda/solver.cu:15-79,211-459,
include/sobfu/cuda/utils.hpp:33-86, src/sobfu/cuda/vector_fields.cu:81-109,144-208,291-337), not from any implementation of them.
Every case runs against BOTH the oracle (tests/test_closed_form_oracle.py, CPU suite) and the HIP kernels through the C ABI
(tests/test_gpu_closed_form.py): a second, structurally independent check next to "HIP == oracle bit for bit".

`api` is a thin adapter: .new_field(dims) / .new_volume(dims) -> writable numpy-like (Z, Y, X, C) arrays, .run_*(...) -> numpy.
Tolerances are stated per case, in units of the float32 spacing of the expected value (ulp)."""
import numpy as np

DIMS = (24, 20, 18)  # X, Y, Z


def grid(dims=DIMS):
    X, Y, Z = dims
    z, y, x = np.meshgrid(np.arange(Z, dtype=np.float64), np.arange(Y, dtype=np.float64), np.arange(X, dtype=np.float64), indexing="ij")
    return x, y, z


def ulps(got, exp, floor=1e-30):
    """|got - exp| in units of the float32 spacing at max(|exp|, floor) (float64 inputs).  `floor` = the magnitude of the terms the
    result is summed from: where the expected value itself cancels towards 0, rounding errors are those of the terms"""
    mag = np.maximum(np.abs(np.asarray(exp, np.float64)), floor).astype(np.float32)
    return np.abs(np.asarray(got, np.float64) - np.asarray(exp, np.float64)) / np.spacing(mag).astype(np.float64)


def interior(a, r):
    return a[r:-r, r:-r, r:-r]


def taps_moments(S):
    """(sum, first moment, second moment) of the 7 taps about the centre, in float64; out(x) = sum_j S[3 - j] in(x + j)"""
    S = np.asarray(S, np.float64)[:7]
    j = np.arange(-3, 4, dtype=np.float64)
    w = S[::-1]  # weight of in(x + j)
    return w.sum(), (w * j).sum(), (w * j * j).sum()


def conv_cases(S):
    """[(name, field (Z, Y, X, 3) float64, expected sum-of-three-1-D-convolutions away from the boundary, ulp tolerance)]:
    a constant (DC gain: the reference's three passes ADD up, solver.cu:290,366,443), a linear field (symmetric unit-sum taps
    reproduce it: 3 f) and a quadratic one (+ the taps' second moment along the axis the square runs along: 3 f + m2)."""
    x, y, z = grid()
    m0, m1, m2 = taps_moments(S)
    const = np.stack([np.full_like(x, 0.75), np.full_like(x, -2.0), np.full_like(x, 5.5)], -1)
    lin = np.stack([1.0 + 0.5 * x - 0.25 * y + 2.0 * z, -3.0 + 0.125 * x + y - 0.5 * z, 0.25 * x + 0.75 * y + 1.5 * z], -1)
    quad = np.stack([0.0625 * x * x, 0.125 * y * y, 0.03125 * z * z], -1)
    exp_const = 3.0 * m0 * const
    exp_lin = 3.0 * m0 * lin + m1 * np.stack([0.5 - 0.25 + 2.0 + 0 * x, 0.125 + 1.0 - 0.5 + 0 * x, 0.25 + 0.75 + 1.5 + 0 * x], -1)
    exp_quad = 3.0 * m0 * quad + np.stack([0.0625 * (2 * m1 * x + m2), 0.125 * (2 * m1 * y + m2), 0.03125 * (2 * m1 * z + m2)], -1)
    # 21 rounded products and 20 rounded sums of terms no larger than the result: a few ulp of the largest partial sum
    return [("constant", const, exp_const, 4.0), ("linear", lin, exp_lin, 12.0), ("quadratic", quad, exp_quad, 12.0)]


def impulse_case():
    """taps 1..7 and a unit impulse: exact small integers -- orientation (out(x) = sum_j S[3-j] in(x+j)), SUM of the three passes,
    clamp-to-edge at a corner (SURVEY Appendix B run 3)"""
    return np.arange(1, 8, dtype=np.float32), [1, 2, 3, 12, 5, 6, 7], [66, 18, 13, 7, 0]


def warp_case():
    """affine phi (tsdf = p0 + p . x, weight 1) warped by an affine psi that stays inside the volume: trilinear interpolation is
    exact on affine functions -> (phi o psi)(x) = p0 + p . psi(x); weight = phi(floor(psi)).y = 1.  Tolerance: 7 nested lerps, each
    two roundings of values <= max|phi| -> absolute 16 * eps32 * max|phi|."""
    x, y, z = grid()
    X, Y, Z = DIMS
    psi = np.stack([1.5 + 0.875 * x + 0.0625 * y, 2.25 + 0.03125 * x + 0.75 * y + 0.0625 * z, 1.125 + 0.0625 * y + 0.8125 * z], -1)
    assert psi[..., 0].min() > 0 and psi[..., 0].max() < X - 1 and psi[..., 1].max() < Y - 1 and psi[..., 2].max() < Z - 1
    p0, p = 0.125, np.array([0.015625, -0.0078125, 0.03125])
    phi = p0 + p[0] * x + p[1] * y + p[2] * z
    exp = p0 + psi @ p
    return psi, phi, exp, 16.0 * float(np.finfo(np.float32).eps) * float(np.abs(phi).max())


def potential_case(w_reg=0.5):
    """quadratic F = phi_n o psi and G = phi_global, quadratic psi: central differences of a quadratic are its derivative exactly,
    the 7-point Laplacian of a quadratic is its (constant) Laplacian:
        nabla_U = (F - G) * grad F + w_reg * (-Lap psi)          (solver.cu:28-31, vector_fields.cu:165-191,299-331)
    All coefficients dyadic and small: every operation is exact in float32 -> tolerance 0 ulp away from the boundary."""
    x, y, z = grid()
    F = 0.0625 * x * x + 0.125 * y + 0.25 * z * z - 0.5 * z
    G = 0.03125 * y * y + 0.5 * x - 1.0
    psi = np.stack([x + 0.0625 * x * x, y + 0.125 * y * z + 0.03125 * z * z, z - 0.0625 * x * y + 0.25 * y * y], -1)
    gradF = np.stack([0.125 * x, 0.125 + 0 * x, 0.5 * z - 0.5], -1)
    lap = np.stack([0.125 + 0 * x, 0.0625 + 0 * x, 0.5 + 0 * x], -1)
    exp = (F - G)[..., None] * gradF + w_reg * (-lap)
    return F, G, psi, exp, w_reg


def update_case(S, alpha=0.25):
    """one whole pass B on a linear nabla_U: psi' = psi - alpha * 3 m0 nabla_U (interior), phi_n o psi' by the affine-warp argument"""
    x, y, z = grid()
    m0, m1, _ = taps_moments(S)
    nU = np.stack([0.5 + 0.03125 * x, -0.25 + 0.015625 * y, 0.125 + 0.0078125 * z], -1)
    psi0 = np.stack([x + 0.5, y + 0.25, z + 0.75], -1)
    conv = 3.0 * m0 * nU + m1 * np.stack([0.03125 + 0 * x, 0.015625 + 0 * x, 0.0078125 + 0 * x], -1)
    psi1 = psi0 - alpha * conv
    p0, p = 0.25, np.array([0.03125, 0.015625, -0.0078125])
    phi = p0 + p[0] * x + p[1] * y + p[2] * z
    warped = p0 + psi1 @ p
    return nU, psi0, phi, psi1, warped, alpha


def inverse_case():
    """psi = identity + a constant displacement c (dyadic): the fixed point psi_inv(x) = x - u(psi_inv(x)) (vector_fields.cu:111-138,
    u = psi - id interpolated trilinearly) has u == c wherever the sample point and its 8 corners are inside the volume, so psi_inv =
    x - c EXACTLY after the first sweep and for all 48 -- away from the faces the displacement points out of (there the sampler's
    clamp changes the point)."""
    x, y, z = grid()
    c = np.array([1.5, -0.75, 2.25])
    psi = np.stack([x + c[0], y + c[1], z + c[2]], -1)
    inv = np.stack([x - c[0], y - c[1], z - c[2]], -1)
    X, Y, Z = DIMS
    ok = (inv[..., 0] >= 0) & (inv[..., 0] <= X - 1) & (inv[..., 1] >= 0) & (inv[..., 1] <= Y - 1) & (inv[..., 2] >= 0) & (inv[..., 2] <= Z - 1)
    return psi, inv, ok


def fuse_case(max_weight=4.0):
    """integrate(phi_global, phi_n o psi) (tsdf_volume.cu:103-130): tsdf = fma(w_g, tsdf_g, tsdf_n) / (w_g + 1), w = min(w_g + 1, max); a
    voxel is SKIPPED when w_n == 0, or when w_n == 1 and tsdf_n is 0 or -1.  Dyadic values: exact."""
    X, Y, Z = DIMS
    g = np.zeros((Z, Y, X, 2))
    n = np.zeros((Z, Y, X, 2))
    g[..., 0], g[..., 1] = 0.5, 1.0
    n[..., 0], n[..., 1] = 0.25, 1.0
    exp = np.zeros_like(g)
    exp[..., 0], exp[..., 1] = (1.0 * 0.5 + 0.25) / 2.0, 2.0
    n[0, :, :, 1] = 0.0                      # w_n == 0: untouched
    n[1, :, :, 0] = 0.0                      # w_n == 1, tsdf_n == 0: untouched
    n[2, :, :, 0] = -1.0                     # w_n == 1, tsdf_n == -1: untouched
    for k in (0, 1, 2):
        exp[k] = g[k]
    g[3, :, :, 1] = max_weight               # weight saturates: tsdf = (4 * 0.5 + 0.25) / 5, w = min(5, 4)
    exp[3, :, :, 0], exp[3, :, :, 1] = (max_weight * 0.5 + 0.25) / (max_weight + 1.0), max_weight
    n[4, :, :, 1] = 2.0                      # w_n == 2 with tsdf_n == -1 is NOT skipped
    n[4, :, :, 0] = -1.0
    exp[4, :, :, 0], exp[4, :, :, 1] = (0.5 - 1.0) / 2.0, 2.0
    return g, n, exp, max_weight


def lattice_case():
    """sample points that sit EXACTLY on the lattice at coordinate 0 (include/sobfu/cuda/utils.hpp:61-72: the upper index is g + 1
    except when the clamped coordinate is exactly 0 or dim - 1, where it is g): the sampler must not touch x = 1 at all.  The plane
    x = 1 holds +inf, so an implementation that reads it gets 0 * inf = NaN; the expected value is phi(0, y, z) itself, exactly
    (every lerp has t = 0: fma(0, hi, fma(-0, lo, lo)) = lo)."""
    x, y, z = grid()
    phi = 0.25 + 0.015625 * x - 0.03125 * y + 0.0078125 * z
    phi[:, :, 1] = np.inf
    psi = np.stack([0.0 * x, y, z], -1)  # every cell samples (0, y, z)
    exp = np.broadcast_to(phi[:, :, :1], phi.shape)
    return psi, phi, exp


def boundary_case():
    """the boundary rules of the two differentiators on the FACES of the volume (away from edges), for the quadratic fields of
    potential_case (all arithmetic exact):
      TsdfDifferentiator (vector_fields.cu:165-191): the missing neighbour is mirrored -> the component normal to a face is exactly 0,
          the tangential ones are the central differences;
      SecondOrderDifferentiator (vector_fields.cu:299-331): on a face both neighbours along the normal are the centre -> that axis
          contributes nothing: the negative Laplacian there is -(sum of the OTHER axes' second derivatives)."""
    x, y, z = grid()
    F = 0.0625 * x * x + 0.125 * y + 0.25 * z * z - 0.5 * z
    psi = np.stack([x + 0.0625 * x * x, y + 0.125 * y * z + 0.03125 * z * z, z - 0.0625 * x * y + 0.25 * y * y], -1)
    gradF = np.stack([0.125 * x, 0.125 + 0 * x, 0.5 * z - 0.5], -1)
    d2 = np.array([[0.125, 0.0, 0.0], [0.0, 0.0, 0.0625], [0.0, 0.5, 0.0]])  # d2[component][axis]: pure second derivatives
    return F, psi, gradF, d2


def maxnorm_case():
    """Reductor::max_update_norm (reductor.cu:342-456, utils.hpp:279-281): ||u|| = __fsqrt_rd(ux^2 + uy^2 + uz^2), the square root
    rounded DOWN.  u = (1, 2, 0): sqrt(5) = 2.2360679..., whose nearest float32 lies ABOVE it -- the expected value is one float below
    what sqrtf returns.  Second-largest norm 2 elsewhere; the arg-max is the voxel's linear index."""
    X, Y, Z = DIMS
    u = np.zeros((Z, Y, X, 4), np.float32)
    at = (Z // 3, Y // 2, X // 4)
    u[at][:3] = (1.0, 2.0, 0.0)
    u[1, 2, 3, 0] = 2.0
    r = np.float32(np.sqrt(5.0))
    assert float(r) ** 2 > 5.0
    return u, float(np.nextafter(r, np.float32(0))), float(at[2] + X * (at[1] + Y * at[0]))


def check_all(api, S):
    """runs every case through `api`; returns the worst deviations for the record"""
    out = {}
    X, Y, Z = DIMS
    for name, f, exp, tol in conv_cases(S):
        got = api.run_conv(f.astype(np.float32), S)
        floor = float(np.abs(f).max())  # the largest term of any of the sums
        u = interior(ulps(got[..., :3], exp, floor), 3)
        assert u.max() <= tol, (name, float(u.max()))
        out["conv_" + name + "_ulp"] = float(u.max())
        if name == "constant":  # clamp-to-edge keeps the gain at the boundary too
            assert ulps(got[..., :3], exp, floor).max() <= tol
    St, row, corner = impulse_case()
    src = np.zeros((Z, Y, X, 4), np.float32)
    src[Z // 2, Y // 2, X // 2, :3] = 1.0
    got = api.run_conv(src, St)
    c = (Z // 2, Y // 2, X // 2)
    assert list(got[c[0], c[1], c[2] - 3:c[2] + 4, 0]) == row and list(got[c[0], c[1] - 3:c[1] + 4, c[2], 1]) == row
    assert list(got[c[0] - 3:c[0] + 4, c[1], c[2], 2]) == row and got[c[0], c[1] + 1, c[2] + 1, 0] == 0
    src[:] = 0
    src[0, 0, 0, :3] = 1.0
    got = api.run_conv(src, St)
    assert list(got[0, 0, 0:5, 0]) == corner and list(got[0, 0:5, 0, 1]) == corner and list(got[0:5, 0, 0, 2]) == corner
    psi, phi, exp, atol = warp_case()
    got = api.run_apply(phi.astype(np.float32), psi.astype(np.float32))
    err = np.abs(got[..., 0].astype(np.float64) - exp)
    assert err.max() <= atol and (got[..., 1] == 1).all(), float(err.max())
    out["warp_abs_err"] = float(err.max())
    # the weight of a warped voxel is that of the LOWER corner, phi(floor(psi)).y (utils.hpp:78-85): every voxel carries its own linear
    # index as weight (exact below 2^24), so the expected weight is the index of floor(psi)
    X, Y, Z = DIMS
    gx, gy, gz = grid()
    widx = gx + X * (gy + Y * gz)
    fl = np.floor(psi)
    got = api.run_apply(phi.astype(np.float32), psi.astype(np.float32), widx.astype(np.float32))
    assert np.array_equal(got[..., 1].astype(np.float64), fl[..., 0] + X * (fl[..., 1] + Y * fl[..., 2])), "warp weight is not phi(floor(psi)).y"
    F, G, psi, exp, w_reg = potential_case()
    got = api.run_potential_gradient(F.astype(np.float32), G.astype(np.float32), psi.astype(np.float32), w_reg)
    u = interior(ulps(got[..., :3], exp), 1)
    assert u.max() == 0, float(u.max())
    nU, psi0, phi, psi1, warped, alpha = update_case(S)
    gpsi, gw = api.run_smooth_update_apply(nU.astype(np.float32), psi0.astype(np.float32), phi.astype(np.float32), S, alpha)
    u = interior(ulps(gpsi[..., :3], psi1, 1.0), 3)
    assert u.max() <= 4.0, float(u.max())
    e = interior(np.abs(gw[..., 0].astype(np.float64) - warped), 3)
    assert e.max() <= 32.0 * float(np.finfo(np.float32).eps) * float(np.abs(phi).max()), float(e.max())
    out["update_ulp"], out["update_warp_abs_err"] = float(u.max()), float(e.max())
    psi, inv, ok = inverse_case()
    got = api.run_inverse(psi.astype(np.float32), 48)
    assert np.array_equal(got[..., :3][ok], inv.astype(np.float32)[ok]) and ok.sum() > 0.5 * ok.size
    g, n, exp, mw = fuse_case()
    got = api.run_fuse(g.astype(np.float32), n.astype(np.float32), mw)
    u = ulps(got, exp)
    assert u.max() <= 1.0, float(u.max())  # (the one inexact quotient, 2.25 / 5, is correctly rounded)
    # lattice hits at coordinate 0: no read of index 1
    psi, phi, exp = lattice_case()
    got = api.run_apply(phi.astype(np.float32), psi.astype(np.float32))
    assert np.array_equal(got[..., 0], exp.astype(np.float32)), "the sampler touched index 1 at coordinate exactly 0 (or lerp(t = 0) is not the lower sample)"
    # boundary rules of the two differentiators, face by face
    F, psi, gradF, d2 = boundary_case()
    zero = np.zeros_like(F)
    g = api.run_potential_gradient(F.astype(np.float32), (F - 1.0).astype(np.float32), psi.astype(np.float32), 0.0)  # (F - G) = 1, w_reg = 0: grad F
    L = api.run_potential_gradient(F.astype(np.float32), F.astype(np.float32), psi.astype(np.float32), 1.0)          # (F - G) = 0, w_reg = 1: -Lap psi
    lap_full = -d2.sum(1)
    for ax in range(3):  # array axis of coordinate ax: x -> 2, y -> 1, z -> 0
        for side in (0, -1):
            sl = [slice(1, -1)] * 3
            sl[2 - ax] = side
            sl = tuple(sl)
            gf, ef = g[sl][..., :3].astype(np.float64), gradF[sl].copy()
            ef[..., ax] = 0.0
            assert np.array_equal(gf, ef), ("gradient on face", ax, side)
            expL = lap_full + d2[:, ax]  # the normal axis drops out
            assert np.array_equal(L[sl][..., :3].astype(np.float64), np.broadcast_to(expL, L[sl][..., :3].shape)), ("Laplacian on face", ax, side)
    del zero
    # max-norm: square root rounded down, arg-max index
    u, norm, index = maxnorm_case()
    got = api.run_max_norm(u)
    assert got[0] == norm and got[1] == index, (got, norm, index)
    return out
