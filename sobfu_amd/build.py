"""Builds libsobfu_hip.so (the C-ABI HIP library) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerical contract (see
csrc/sobfu_device.hpp): the reference writes its products with non-contractible intrinsics.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsobfu_hip.so")
SOURCES = ["tsdf_kernels.hip", "field_kernels.hip", "reduce_kernels.hip", "solver_kernels.hip", "launcher_kernels.hip", "solver_capi.hip", "tiled_capi.hip", "mc_kernels.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]


# solver_kernels.hip: hipcc's SLP vectoriser turns the 7-tap y chain of the compact pass B into <7 x float> shuffles that
# it lowers through 64 B/lane of scratch (pass B 390 us instead of 180 us at 256^3); without SLP the kernels also need
# ~30 fewer VGPRs.  Measured, interleaved A/B: profiles/LABBOOK.md section 4.1.
PER_FILE_FLAGS = {"solver_kernels.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_hip(force: bool = False, extra_flags=(), verbose: bool = False) -> str:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".inc", ".inl"))] + [os.path.join(ROOT, "include", "sobfu_hip.h")]
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(op)
        if force or _stale(op, [sp] + hdrs):
            jobs.append([_hipcc(), *FLAGS, *PER_FILE_FLAGS.get(src, []), *extra_flags, "-c", sp, "-o", op])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build_hip(force="--force" in sys.argv, verbose=True))
