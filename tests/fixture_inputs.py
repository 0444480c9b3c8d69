"""Deterministic numpy-only input generators shared by tests/golden/make_reference_fixtures.py (which feeds them to the
emulated reference) and tests/test_reference_fixtures.py (which regenerates the larger inputs instead of loading them from
the .npz; the fixtures carry a sha256 of every input, so a drift here is caught).  No oracle, no sobfu_amd kernels."""
import hashlib

import numpy as np

from sobfu_amd.synthetic import hash_field, render_ellipsoid_depth, render_sphere_depth

F32 = np.float32
# raw S=7, lambda=0.1 taps of the reference's table (src/sobfu/solver.cpp:190-198), normalised the way decompose_sobolev_filter does
_RAW = np.array([0.00030, 0.00441, 0.06571, 0.99565, 0.06571, 0.00441, 0.00030], F32)


def taps_s7_l01():
    s = F32(0)
    for v in _RAW:
        s = F32(s + v)
    return (_RAW / s).astype(F32)


def rand_volume(dims, seed):
    """TSDF-like volume: values in [-1, 1), weights 0/1 from an independent hash."""
    X, Y, Z = dims
    v = hash_field((Z, Y, X, 2), seed, 1.0)
    v[..., 1] = (hash_field((Z, Y, X), seed + 7) > 0).astype(F32)
    return v


def identity(dims):
    X, Y, Z = dims
    psi = np.zeros((Z, Y, X, 4), F32)
    psi[..., 0] = np.arange(X, dtype=F32)[None, None, :]
    psi[..., 1] = np.arange(Y, dtype=F32)[None, :, None]
    psi[..., 2] = np.arange(Z, dtype=F32)[:, None, None]
    return psi


def warped_identity(dims, seed, amp):
    X, Y, Z = dims
    psi = identity(dims)
    psi[..., :3] += hash_field((Z, Y, X, 3), seed, amp)
    return psi


def sphere_volume(dims, centre, radius, trunc):
    """Smooth analytic TSDF (float64 distance, rounded once), weight 1 everywhere: a well-posed registration pair."""
    X, Y, Z = dims
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    d = np.sqrt((x - centre[0]) ** 2.0 + (y - centre[1]) ** 2.0 + (z - centre[2]) ** 2.0) - radius
    v = np.zeros((Z, Y, X, 2), F32)
    v[..., 0] = np.clip(d / trunc, -1.0, 1.0).astype(F32)
    v[..., 1] = 1.0
    return v


def kernel_inputs(dims, seed, amp):
    X, Y, Z = dims
    ins = dict(phi_n_psi=rand_volume(dims, seed), phi_global=rand_volume(dims, seed + 1), psi=warped_identity(dims, seed + 2, amp),
               fuse_in=rand_volume(dims, seed + 3), taps=taps_s7_l01())
    ins["fuse_in"][..., 1] = np.floor(np.abs(hash_field((Z, Y, X), seed + 4, 6.0)))  # accumulated weights 0..5
    return ins


def mc_volume(dims):
    X, Y, Z = dims
    z, y, x = np.meshgrid(np.arange(Z), np.arange(Y), np.arange(X), indexing="ij")
    d = np.sqrt(((x - X / 2 + 0.3) / (X * 0.36)) ** 2.0 + ((y - Y / 2 - 0.2) / (Y * 0.33)) ** 2.0 + ((z - Z / 2 + 0.1) / (Z * 0.3)) ** 2.0) - 1.0
    vol = np.zeros((Z, Y, X, 2), F32)
    vol[..., 0] = np.clip(d * 3.0, -1.0, 1.0).astype(F32)
    vol[..., 1] = (hash_field((Z, Y, X), 77) > -0.9).astype(F32)  # ~5 % unobserved voxels: cubes touching them are skipped
    return vol


def digest(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8).copy()


def snoopy_frame(intr, n):
    """frame n (0..6) of the synthetic VolumeDeform-style sequence of BASELINE config 2: a breathing, drifting ellipsoid in front of the camera"""
    a = [0.0, 0.35, 0.7, 1.0, 1.25, 1.32, 1.33][n]
    c = (0.004 * np.sin(a), 0.003 * (1 - np.cos(a)), 0.50 + 0.003 * a)
    r = (0.15 * (1 + 0.04 * np.sin(a)), 0.13 * (1 - 0.03 * np.sin(a)), 0.14 * (1 + 0.02 * a))
    return render_ellipsoid_depth(c, r, intr)


def translating_sphere_frame(intr, n, rows=480, cols=640):
    """frame n of SURVEY 8(d) input 1: a sphere of radius 0.1 m at 0.75 m moving 5 mm along x per frame"""
    return render_sphere_depth((0.005 * n, 0.0, 0.75), 0.1, intr, rows=rows, cols=cols)


def bench_sequence_frame(intr, size, t_z, vx, n):
    """frame n of the sequence bench.py's per-frame pipeline uses (bench_frames): a sphere of radius 0.2 * size in the middle of the
    volume's depth range, translating 1.3 voxels per frame"""
    return render_sphere_depth((1.3 * vx * n, 0.0, t_z + 0.5 * size), 0.2 * size, intr)
