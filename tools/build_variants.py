"""Builds tile-shape variants of libsobfu_hip.so for tuning experiments (build/variants/*.so; build/ is git-ignored but travels with gpurun)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sobfu_amd import build
out = os.path.join(build.ROOT, "build", "variants")
os.makedirs(out, exist_ok=True)
for spec in sys.argv[1:]:
    parts = spec.split("x")
    rpt, wy = int(parts[0]), int(parts[1])
    extra = [(d if d.startswith("-") else f"-D{d}") for d in parts[2:]]  # e.g. 1x8xSOBFU_XCD_SWIZZLE=1 or 1x8x-fno-slp-vectorize
    objs = []
    for src in build.SOURCES:
        o = os.path.join(out, f"{src[:-4]}_{spec}.o")
        flags = build.FLAGS + build.PER_FILE_FLAGS.get(src, []) + ([f"-DSOBFU_RPT={rpt}", f"-DSOBFU_WY={wy}", *extra] if src == "solver_kernels.hip" else [])
        if src != "solver_kernels.hip":
            o = os.path.join(build.HERE, "build", src.replace(".hip", ".o"))
        else:
            subprocess.check_call([build._hipcc(), *flags, "-c", os.path.join(build.CSRC, src), "-o", o])
        objs.append(o)
    lib = os.path.join(out, f"libsobfu_hip_{spec}.so")
    subprocess.check_call([build._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib])
    print(lib)
