"""ctypes loader for libsobfu_hip.so (the C-ABI HIP library).  No fallbacks: if the library is missing or a
symbol of include/sobfu_hip.h is absent, importing the product path fails loudly."""
from __future__ import annotations

import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB_PATH = os.environ.get("SOBFU_HIP_LIB") or os.path.join(HERE, "libsobfu_hip.so")  # env override: kernel-tuning experiments
HEADER = os.path.join(ROOT, "include", "sobfu_hip.h")


class SolverParams(C.Structure):
    """sobfu_hip_solver_params (= SolverParams, reference include/sobfu/solver.hpp:16-19)."""
    _fields_ = [("verbosity", C.c_int), ("max_iter", C.c_int), ("s", C.c_int), ("max_update_norm", C.c_float),
                ("lambda_", C.c_float), ("alpha", C.c_float), ("w_reg", C.c_float)]


class SolverReport(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("last_max_update_norm", C.c_float),
                ("last_max_update_index", C.c_float), ("last_e_data", C.c_float), ("last_e_reg", C.c_float)]


LOG_FN = C.CFUNCTYPE(None, C.c_char_p, C.c_void_p)


def declared_symbols():
    """Every function name declared in include/sobfu_hip.h."""
    with open(HEADER) as f:
        txt = f.read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sobfu_hip_[a-z0-9_]+)\s*\(", txt)) - {"sobfu_hip_log_fn"})


class HipError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -m sobfu_amd.build` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # The library needs libamdhip64, and so does PyTorch -- which ships its OWN copy.  Whichever is loaded first serves both (same
    # soname); PyTorch goes first, so that the tensors this front end hands over and the library's kernels live in ONE HIP runtime
    # (loaded the other way round, the library saw "no ROCm-capable device" in a process where torch.cuda was fine).
    import torch  # noqa: F401

    L = C.CDLL(LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    if missing:
        raise ImportError(f"{LIB_PATH} does not export: {missing}")
    L.sobfu_hip_error_string.restype = C.c_char_p
    L.sobfu_hip_solver_workspace_bytes.restype = C.c_size_t
    L.sobfu_hip_solver_updates.restype = C.c_void_p
    _lib = L
    return L


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().sobfu_hip_error_string(C.c_int(rc)).decode()
        raise HipError(f"{what}: {msg} (code {rc})")
