"""Pins against REFERENCE-HELD constants (tests/golden/reference_tables.json, extracted from the reference tree by
tests/golden/make_reference_tables.py): the marching-cubes vertex-count table and case table, and every raw tap of the
Sobolev filter table -- checked on the oracle, on the C-ABI library's host code and, on the GPU, on the HIP kernels."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with open(os.path.join(ROOT, "tests", "golden", "reference_tables.json")) as f:
    REF = json.load(f)


def packed_table():
    import re

    txt = open(os.path.join(ROOT, "sobfu_amd", "csrc", "mc_table.inc")).read()
    assert txt == open(os.path.join(ROOT, "oracle", "mc_table.inc")).read()  # the HIP library and the oracle compile the same constants
    vals = [int(v, 16) for v in re.findall(r"0x([0-9a-f]{16})ull", txt)]
    assert len(vals) == 256
    return [[(v >> (4 * k)) & 15 for k in range(16)] for v in vals]


def test_num_verts_table_is_the_references(oracle):
    """numVertsTable (src/kfusion/marching_cubes.cpp:358) vs the vertex count the packed table implies (entries before the
    first 0xF nibble -- what both num_verts() implementations compute)"""
    nv = REF["numVertsTable"]
    assert len(nv) == 256
    table = packed_table()
    for case in range(256):
        row = table[case]
        assert (row.index(15) if 15 in row else 16) == nv[case], case
        assert oracle.mc_num_verts(case) == nv[case], case


def test_case_table_hash_is_the_references():
    """every one of the 4096 triTable entries (src/kfusion/marching_cubes.cpp:81-355), through a hash of the table"""
    flat = [(-1 if v == 15 else v) for row in packed_table() for v in row]
    assert hashlib.sha256(",".join(str(v) for v in flat).encode()).hexdigest() == REF["triTable_sha256"]


def _normalised(raw):
    """decompose_sobolev_filter's tail (solver.cpp:253-261): float sum in index order, then one division per tap"""
    h = np.array([np.float32(t) for t in raw], np.float32)
    s = np.float32(0)
    for v in h:
        s = np.float32(s + v)
    return (h / s).astype(np.float32)


def test_every_raw_filter_tap_is_the_references(oracle):
    from sobfu_amd import build, ops

    build.build_hip()
    assert len(REF["sobolev_filters"]) == 8
    for f in REF["sobolev_filters"]:
        s, lam = f["s"], np.float32(f["lambda"])
        want = _normalised(f["raw_taps"])
        assert abs(float(want.sum()) - 1) < 1e-6 and np.array_equal(want, want[::-1])
        got_o = oracle.sobolev_filter(s, lam)
        got_h = ops.sobolev_filter(s, lam)  # host code of libsobfu_hip.so: runs without a GPU
        assert np.array_equal(got_o.view(np.uint32), want.view(np.uint32)), (s, f["lambda"])
        assert np.array_equal(got_h.view(np.uint32), want.view(np.uint32)), (s, f["lambda"])
    # a perturbed tap is detected (the normalised comparison has the resolution of one raw digit)
    raw = list(REF["sobolev_filters"][2]["raw_taps"])
    raw[1] = "0.00442"
    assert not np.array_equal(_normalised(raw).view(np.uint32), oracle.sobolev_filter(7, np.float32(0.1)).view(np.uint32))


@pytest.mark.gpu
def test_hip_vertex_counts_are_the_references():
    """all 256 corner-sign configurations through the HIP classify kernel: vertex count per cell == numVertsTable"""
    import torch

    from sobfu_amd import ops

    nv = REF["numVertsTable"]
    X, Y, Z = 2 * 256, 2, 2
    vol = np.zeros((Z, Y, X, 2), np.float32)
    vol[..., 1] = 1.0
    dx, dy, dz = (0, 1, 1, 0, 0, 1, 1, 0), (0, 0, 1, 1, 0, 0, 1, 1), (0, 0, 0, 0, 1, 1, 1, 1)  # corner numbering, marching_cubes.cu:38-79
    for case in range(256):
        for c in range(8):
            vol[dz[c], dy[c], 2 * case + dx[c], 0] = -0.5 if (case >> c) & 1 else 0.5
    occ, count = ops.mc_occupied_voxels(torch.from_numpy(vol).cuda(), X * Y * Z)
    occ = occ.cpu().numpy()
    got = {int(occ[0, i]): int(occ[1, i]) for i in range(count)}
    for case in range(256):
        assert got.get(2 * case, 0) == nv[case], case  # cell (2*case, 0, 0) has linear index 2*case
