"""Reader for the reference's parameter files (params/*.ini; schema src/apps/demo.cpp:87-160, derived values :71-74) -- the
Python twin of sobfu_amd::read_params_ini (include/sobfu_amd/sobfu.hpp): same keys, same derived values, same refusals."""
from __future__ import annotations

import numpy as np

_INT = ("VOL_DIMS_X", "VOL_DIMS_Y", "VOL_DIMS_Z", "BILATERAL_KERNEL_SIZE", "START_FRAME", "MAX_ITER", "S")
_REQUIRED = ("TSDF_TRUNC_DIST", "ETA", "VOL_POSE_T_Z")  # vm[...].as<float>() in the reference: missing = exception


def read_ini(path, dims=None):
    """-> dict with the raw keys plus the derived entries the frame driver uses (float32 arithmetic as in the reference):
    dims, size, vs (voxel sizes), trunc / eta in metres, intr, R / t (volume pose), bilateral, trunc_depth, max_weight,
    start_frame, max_iter, max_update_norm, s, lam, alpha, w_reg.  `dims` overrides VOL_DIMS_* (a cubic grid edge or a triple);
    the voxel-unit parameters follow the new voxel size, as in apps/sobfu_headless --dims.  Unknown keys are ignored."""
    kv = {}
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0]
            if "=" not in line:
                continue
            k, v = line.split("=", 1)
            kv[k.strip()] = v.strip()
    for k in _REQUIRED:
        if k not in kv:
            raise ValueError(f"required key {k} is missing from {path}")
    raw = {k: (int(v) if k in _INT else float(v)) for k, v in kv.items() if k in _INT or _is_float(v)}
    if dims is not None:
        d = (int(dims),) * 3 if np.isscalar(dims) else tuple(int(x) for x in dims)
        raw["VOL_DIMS_X"], raw["VOL_DIMS_Y"], raw["VOL_DIMS_Z"] = d
    d = tuple(raw.get(f"VOL_DIMS_{a}", 0) for a in "XYZ")
    size = np.array([raw.get(f"VOL_SIZE_{a}", 0.0) for a in "XYZ"], np.float32)
    if min(d) <= 0 or not (size > 0).all():
        raise ValueError(f"VOL_DIMS_* / VOL_SIZE_* must be positive in {path}")
    if not raw["TSDF_TRUNC_DIST"] > 0:
        raise ValueError(f"TSDF_TRUNC_DIST must be positive in {path}")
    vs = (size / np.array(d, np.float32)).astype(np.float32)  # Params::voxel_sizes
    P = dict(raw)
    P.update(
        dims=d, size=size, vs=vs,
        trunc=np.float32(raw["TSDF_TRUNC_DIST"]) * vs[0], eta=np.float32(raw["ETA"]) * vs[0],        # demo.cpp:71-72
        max_weight=float(raw.get("TSDF_MAX_WEIGHT", 64.0)),
        intr=tuple(float(raw.get(k, 0.0)) for k in ("INTR_FX", "INTR_FY", "INTR_CX", "INTR_CY")),
        R=np.eye(3, dtype=np.float32),
        t=np.array([-size[0] / np.float32(2), -size[1] / np.float32(2), np.float32(raw["VOL_POSE_T_Z"])], np.float32),  # :73-74
        bilateral=(int(raw.get("BILATERAL_KERNEL_SIZE", 7)), float(raw.get("BILATERAL_SIGMA_SPATIAL", 4.5)),
                   float(raw.get("BILATERAL_SIGMA_DEPTH", 0.04))),
        trunc_depth=float(raw.get("TRUNC_DEPTH", 0.0)), start_frame=int(raw.get("START_FRAME", 1)),
        max_iter=int(raw.get("MAX_ITER", 0)), max_update_norm=float(raw.get("MAX_UPDATE_NORM", 0.0)),
        s=int(raw.get("S", 7)), lam=float(raw.get("LAMBDA", 0.1)), alpha=float(raw.get("ALPHA", 0.0)), w_reg=float(raw.get("W_REG", 0.0)))
    return P


def _is_float(v):
    try:
        float(v)
        return True
    except ValueError:
        return False
