"""Marching cubes: the packed case table, the CPU restatement's geometric properties, and -- on the GPU -- the
HIP path against the restatement bit for bit (same deterministic voxel order)."""
import os
from collections import Counter

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EDGE = [(0, 1), (1, 2), (2, 3), (3, 0), (4, 5), (5, 6), (6, 7), (7, 4), (0, 4), (1, 5), (2, 6), (3, 7)]


def load_table():
    import re

    txt = open(os.path.join(ROOT, "sobfu_amd", "csrc", "mc_table.inc")).read()
    assert txt == open(os.path.join(ROOT, "oracle", "mc_table.inc")).read()
    vals = [int(v, 16) for v in re.findall(r"0x([0-9a-f]{16})ull", txt)]
    assert len(vals) == 256
    return [[(v >> (4 * k)) & 15 for k in range(16)] for v in vals]


def test_case_table_is_a_valid_marching_cubes_table(oracle):
    table = load_table()
    for case, row in enumerate(table):
        n = row.index(15) if 15 in row else 16
        assert n % 3 == 0 and all(v == 15 for v in row[n:]) and n == oracle.mc_num_verts(case)
        inside = [(case >> c) & 1 for c in range(8)]
        cut = {e for e, (a, b) in enumerate(EDGE) if inside[a] != inside[b]}
        assert set(row[:n]) == cut  # the triangles of a case use exactly the edges the surface cuts
        assert all(len(set(row[t:t + 3])) == 3 for t in range(0, n, 3))
    assert table[0] == [15] * 16 and table[255] == [15] * 16 and table[1][:3] == [0, 8, 3]
    if os.path.exists("/root/reference/src/kfusion/marching_cubes.cpp"):  # dev container only: same constants as the reference
        import sys

        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import pack_mc_table

        ref = pack_mc_table.parse("/root/reference/src/kfusion/marching_cubes.cpp")
        assert [[15 if v < 0 else v for v in r] for r in ref] == table


def sphere_volume(oracle, n=32, centre=(0.25, 0.26, 0.24), r=0.1):
    vs = 0.5 / n
    vol = oracle.new_volume((n, n, n))
    oracle.init_sphere(vol, (vs,) * 3, 5 * vs, 2 * vs, centre, r)
    return vol, vs


def edge_counts(v):
    tri = np.round(v[:, :3].astype(np.float64) * 1e6).astype(np.int64).reshape(-1, 3, 3)
    ec = Counter()
    for t in tri:
        k = [tuple(p) for p in t]
        for a, b in ((0, 1), (1, 2), (2, 0)):
            ec[tuple(sorted((k[a], k[b])))] += 1
    return ec


def test_oracle_sphere_mesh_is_closed_and_on_the_sphere(oracle):
    centre, r = (0.25, 0.26, 0.24), 0.1
    vol, vs = sphere_volume(oracle, 32, centre, r)
    v, n = oracle.marching_cubes(vol, (0.5,) * 3)
    assert len(v) % 3 == 0 and len(v) > 1000 and np.all(v[:, 3] == 1) and np.all(n[:, 3] == 1)
    p = v[:, :3] * np.array([1, -1, -1], np.float32)  # store_point flips y and z
    rad = np.sqrt(((p - np.array(centre, np.float32)) ** 2).sum(1))
    assert rad.min() > r - 0.05 * vs and rad.max() < r + 0.05 * vs  # linear interpolation of an exact SDF
    assert set(edge_counts(v).values()) == {2}  # watertight: every edge is shared by exactly two triangles
    nn = n[:, :3] * np.array([1, -1, -1], np.float32)
    out = (p - np.array(centre, np.float32)) / rad[:, None]
    assert float((nn * out).sum(1).min()) > 0.9  # outward unit normals
    assert np.allclose(np.sqrt((nn ** 2).sum(1)), 1.0, atol=1e-5)
    # pose: a translation + 90 degree rotation about z moves every vertex rigidly
    R = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    t = np.array([0.1, -0.2, 0.3], np.float32)
    v2, n2 = oracle.marching_cubes(vol, (0.5,) * 3, R, t)
    p2 = v2[:, :3] * np.array([1, -1, -1], np.float32)
    assert np.allclose(p2, p @ R.T + t, atol=1e-6) and np.array_equal(n2, n)  # normals are not rotated (marching_cubes.cu:258-261)


def test_oracle_edge_cases(oracle):
    vol = oracle.new_volume((8, 8, 8))
    v, n = oracle.marching_cubes(vol, (1, 1, 1))  # nothing observed
    assert v.shape == (0, 4)
    vol[..., 0], vol[..., 1] = 1.0, 1.0  # observed, all outside
    assert oracle.marching_cubes(vol, (1, 1, 1))[0].shape == (0, 4)
    vol[4, 4, 4, 0] = -1.0  # one inside voxel: 8 cells, one triangle each
    occ, count = oracle.mc_occupied_voxels(vol, 100)
    assert count == 8 and list(occ[0, :8]) == sorted(occ[0, :8]) and set(occ[1, :8]) == {3}
    assert oracle.mc_offsets(occ, count) == 24 and list(occ[2, :8]) == list(range(0, 24, 3))
    vol[4, 4, 5, 1] = 0.0  # an unobserved corner silences every cell that touches it
    occ2, count2 = oracle.mc_occupied_voxels(vol, 100)
    assert count2 == 4
    occ3, count3 = oracle.mc_occupied_voxels(vol, 3)  # cap: the first max_size cells in index order
    assert count3 == 3 and list(occ3[0, :3]) == list(occ2[0, :3])


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["sphere32", "sphere_odd", "random", "capped", "empty"])
def test_hip_matches_oracle(oracle, case):
    import torch

    from sobfu_amd import ops

    rng = np.random.default_rng(11)
    R = np.array([[0.36, 0.48, -0.8], [-0.8, 0.6, 0.0], [0.48, 0.64, 0.6]], np.float32)
    t = np.array([0.05, -0.1, 0.2], np.float32)
    size, kw = (0.5, 0.45, 0.55), {}
    if case == "sphere32":
        vol, _ = sphere_volume(oracle, 32)
    elif case == "sphere_odd":
        vs = 0.5 / 40
        vol = oracle.new_volume((70, 33, 19))
        oracle.init_sphere(vol, (vs,) * 3, 5 * vs, 2 * vs, (0.4, 0.2, 0.12), 0.09)
    elif case in ("random", "capped"):
        vol = np.stack([rng.uniform(-1, 1, (20, 24, 40)), (rng.uniform(0, 1, (20, 24, 40)) > 0.02)], -1).astype(np.float32)
        if case == "capped":
            kw = dict(max_voxels=1000, max_vertices=2500)
    else:
        vol = oracle.new_volume((16, 16, 16))
    v_o, n_o = oracle.marching_cubes(vol, size, R, t, **kw)
    d = torch.from_numpy(vol).cuda()
    if case != "empty":
        mv = kw.get("max_voxels", 2_000_000)
        occ_o, count_o = oracle.mc_occupied_voxels(vol, mv)
        occ_d, count_d = ops.mc_occupied_voxels(d, mv)
        assert count_d == count_o > 0
        assert np.array_equal(occ_d[:2, :count_d].cpu().numpy(), occ_o[:2, :count_o])
        assert ops.mc_offsets(occ_d, count_d) == oracle.mc_offsets(occ_o, count_o)
        assert np.array_equal(occ_d[2, :count_d].cpu().numpy(), occ_o[2, :count_o])
    v_d, n_d = ops.marching_cubes(d, size, R, t, **kw)
    assert v_d.shape[0] == v_o.shape[0]
    # with a caller-kept workspace (no allocation inside the scan steps; what kfusion::cuda::MarchingCubes passes): same mesh
    ws = ops.mc_workspace(d)
    for _ in range(2):  # reused across calls
        v_w, n_w = ops.marching_cubes(d, size, R, t, workspace=ws, **kw)
        assert torch.equal(v_w.view(torch.int32), v_d.view(torch.int32)) and torch.equal(n_w.view(torch.int32), n_d.view(torch.int32))
    v_w, _ = ops.marching_cubes(d, size, R, t, workspace=ws[:16], **kw)  # too small: falls back to per-call scratch
    assert torch.equal(v_w.view(torch.int32), v_d.view(torch.int32))
    assert np.array_equal(v_d.cpu().numpy().view(np.uint32), v_o.view(np.uint32))
    assert np.array_equal(n_d.cpu().numpy(), n_o, equal_nan=True)  # degenerate triangles carry NaN normals on both sides
    if case == "capped":
        assert v_o.shape[0] == 2499 and bool(np.all(v_o[:, 3] == 1))


@pytest.mark.gpu
def test_hip_256_cubed_mesh_is_closed(oracle):
    """full-size property test: 256^3 sphere -> closed surface, vertices on the sphere (no oracle run at this size)"""
    import torch

    from sobfu_amd import ops

    n, r, c = 256, 0.2, (0.375, 0.37, 0.38)
    vs = 0.75 / n
    vol = ops.new_volume((n, n, n))
    ops.init_sphere(vol, (vs,) * 3, 48 * vs, 3 * vs, c, r)
    v, _ = ops.marching_cubes(vol, (0.75,) * 3)
    v = v.cpu().numpy()
    assert len(v) % 3 == 0 and len(v) > 300_000
    p = v[:, :3] * np.array([1, -1, -1], np.float32)
    rad = np.sqrt(((p - np.array(c, np.float32)) ** 2).sum(1))
    assert rad.min() > r - 0.05 * vs and rad.max() < r + 0.05 * vs
    # closedness without matching vertices across cells (neighbouring cells walk a shared edge in opposite directions, so their
    # copies of a vertex may differ in the last bit): a closed oriented surface has zero total area vector and encloses the volume
    # the divergence theorem gives
    t = p.astype(np.float64).reshape(-1, 3, 3)
    area_vec = 0.5 * np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0])
    area = np.sqrt((area_vec ** 2).sum(1)).sum()
    assert abs(area / (4 * np.pi * r * r) - 1) < 2e-3
    assert float(np.abs(area_vec.sum(0)).max()) < 1e-9 * area / vs  # cancels to rounding noise
    vol_enclosed = abs(float((t[:, 0] * np.cross(t[:, 1], t[:, 2])).sum()) / 6)
    assert abs(vol_enclosed / (4 / 3 * np.pi * r ** 3) - 1) < 2e-3
