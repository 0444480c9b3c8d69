"""Multi-GPU path on CPU: world-size-2/3 gloo runs of sobfu_amd.tiled (slab decomposition + halo exchange + max-norm
all-reduce + device-gate semantics) must reproduce the single-process oracle solve bit for bit."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


def _run(world, thr, tmp_path):
    out = str(tmp_path / f"tiled_{world}_{thr}.npz")
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_tiled_worker.py"), str(r), str(world), port, out, str(thr)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=240)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return np.load(out)


def _reference(oracle, thr):
    sys.path.insert(0, HERE)
    import _tiled_worker as W

    pg, pn = W.inputs()
    psi = oracle.new_field(W.DIMS)
    oracle.init_identity(psi)
    r1 = oracle.estimate_psi(pg, pn, psi, max_iter=W.ITERS, alpha=0.05, w_reg=0.4, max_update_norm=thr, inverse_iters=0,
                             compute_jacobian=False)
    r2 = oracle.estimate_psi(pg, pn, psi, max_iter=3, alpha=0.05, w_reg=0.4, max_update_norm=thr, inverse_iters=0,
                             compute_jacobian=False)
    return psi, r1, r2


@pytest.mark.parametrize("world", [2, 3])
def test_slabs_match_single_process(oracle, tmp_path, world):
    got = _run(world, -1.0, tmp_path)
    psi, r1, r2 = _reference(oracle, -1.0)
    assert int(got["done"]) == r1["iters"] == 6 and int(got["done2"]) == 3
    assert np.array_equal(got["psi"].view(np.uint32), psi.view(np.uint32))
    assert np.array_equal(got["pnp"].view(np.uint32), r2["phi_n_psi"].view(np.uint32))
    assert np.array_equal(got["norms"].view(np.uint32), np.ascontiguousarray(r1["trace"][:, 2]).view(np.uint32))
    assert np.array_equal(got["norms2"].view(np.uint32), np.ascontiguousarray(r2["trace"][:, 2]).view(np.uint32))


def test_slabs_convergence_gate(oracle, tmp_path):
    """positive threshold: the all-reduced max-norm gate must stop every rank at the reference's iteration"""
    psi0, r_free, _ = _reference(oracle, -1.0)
    tr = r_free["trace"][:, 2]
    thr = float(np.float32((tr[2] + tr[3]) / 2)) if tr[3] < tr[2] else float(tr.min())
    got = _run(2, thr, tmp_path)
    psi, r1, r2 = _reference(oracle, thr)
    assert r1["iters"] < 6
    assert int(got["done"]) == r1["iters"] and int(got["done2"]) == r2["iters"]
    assert np.array_equal(got["psi"].view(np.uint32), psi.view(np.uint32))
    assert np.array_equal(got["pnp"].view(np.uint32), r2["phi_n_psi"].view(np.uint32))


def test_layout_properties():
    from sobfu_amd.tiled import SlabLayout

    for world in (1, 2, 3, 4, 8):
        own = []
        for r in range(world):
            L = SlabLayout((16, 16, 64), world, r)
            own += list(range(L.z0, L.z1))
            assert L.Lz == (L.z1 - L.z0) + L.lo + L.hi and L.zbase == L.z0 - L.lo
            assert (L.lo == 0) == (r == 0) and (L.hi == 0) == (r == world - 1)
        assert own == list(range(64))
    with pytest.raises(ValueError):
        SlabLayout((16, 16, 8), 4, 0)
