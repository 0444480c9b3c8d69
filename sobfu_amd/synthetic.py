"""Deterministic synthetic inputs: depth frames of analytic scenes.

Pure numpy, no RNG.  Used by the tests, bench.py and the headless frame driver's examples.
"""
from __future__ import annotations

import numpy as np


def render_sphere_depth(centre, radius, intr, rows=480, cols=640):
    """uint16 millimetre depth image of a sphere seen by a pinhole camera at the origin looking down +z.

    Pixel (u, v) casts the ray ((u-cx)/fx, (v-cy)/fy, 1); depth = z of the nearer ray/sphere
    intersection, rounded (half-to-even) to millimetres; 0 where the ray misses.  Evaluated in float64.
    """
    fx, fy, cx, cy = (np.float64(v) for v in intr)
    u, v = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    dx, dy = (u - cx) / fx, (v - cy) / fy
    c = np.asarray(centre, np.float64)
    a = dx * dx + dy * dy + 1.0
    b = -2.0 * (dx * c[0] + dy * c[1] + c[2])
    cc = float((c * c).sum()) - float(radius) ** 2
    disc = b * b - 4.0 * a * cc
    z = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0.0))) / (2.0 * a), 0.0)
    return np.rint(z * 1000.0).astype(np.uint16)


def render_ellipsoid_depth(centre, radii, intr, rows=480, cols=640):
    """Same as render_sphere_depth for an axis-aligned ellipsoid (VolumeDeform-style breathing shapes)."""
    fx, fy, cx, cy = (np.float64(v) for v in intr)
    u, v = np.meshgrid(np.arange(cols, dtype=np.float64), np.arange(rows, dtype=np.float64))
    r = np.asarray(radii, np.float64)
    c = np.asarray(centre, np.float64) / r
    dx, dy, dz = (u - cx) / fx / r[0], (v - cy) / fy / r[1], np.full_like(u, 1.0 / r[2])
    a = dx * dx + dy * dy + dz * dz
    b = -2.0 * (dx * c[0] + dy * c[1] + dz * c[2])
    cc = float((c * c).sum()) - 1.0
    disc = b * b - 4.0 * a * cc
    z = np.where(disc >= 0, (-b - np.sqrt(np.maximum(disc, 0.0))) / (2.0 * a), 0.0)
    return np.rint(z * 1000.0).astype(np.uint16)


def hash_field(shape, seed=1, scale=1.0):
    """Reproducible pseudo-random float32 array in [-scale, scale): splitmix64-style hash of the flat index.

    Stateless (no numpy RNG), so every rank / test regenerates identical data from (shape, seed).
    """
    n = int(np.prod(shape))
    with np.errstate(over="ignore"):
        h = (np.arange(n, dtype=np.uint64) + np.uint64(seed)) * np.uint64(0x9E3779B97F4A7C15)
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        h ^= h >> np.uint64(32)
    x = (h & np.uint64(0xFFFFFFFF)).astype(np.float64) / 4294967296.0
    return ((x * 2.0 - 1.0) * scale).astype(np.float32).reshape(shape)
