"""Builds the reference's hot-path sources for gfx950 with hipcc -- the .cu and .cpp files compiled WHERE THEY LIE under
/root/reference (no copies, no rewriting: hipcc understands `kernel<<<...>>>(...)` and `extern __shared__`), through the CUDA -> HIP
name map of shim/cuda_runtime.h and the OpenCV / PCL / Boost stand-ins of tools/ref_emulation/shim, with the driver of
tools/ref_emulation/driver.cpp (scenarios kernels / solver / tsdf / depth / frames / launchers / time) -> oracle/_ref/reference_hip_ieee and
oracle/_ref/reference_hip_fast (git-ignored; they travel to the GPU box, the reference does not).

  ieee   -ffp-contract=off, correctly rounded divide / sqrt, denormals kept: the arithmetic of the host emulation, the oracle and this
         repo's kernels -- its arrays must equal tests/golden/ref_*.npz bit for bit (tests/test_gpu_reference_hipbuild.py)
  fast   -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero: hipcc's analogues of the
         reference's nvcc flags (CMakeLists.txt:40-46: --fmad=true by default, --prec-div=false --prec-sqrt=false --ftz=true)

`fast` with -DOCML_BASIC_ROUNDED_OPERATIONS (the *_rn intrinsics become correctly rounded library calls, never contracted -- nvcc's
guarantee); both with -fgpu-rdc (the reference launches kernels of one .cu file from another).  Not compiled: marching_cubes.cu (32-wide warp
intrinsics, PTX, texture fetches) -- mc_unavailable.cpp holds its five launchers as refusing stubs so that SobFusion links.  SHIM EVIDENCE -- see shim/cuda_runtime.h.  Test / measurement infrastructure; nothing under sobfu_amd/ uses it.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SOBFU_REFERENCE", "/root/reference")
OUT_DIR = os.path.join(ROOT, "oracle", "_ref")
CU = ["src/sobfu/cuda/solver.cu", "src/sobfu/cuda/vector_fields.cu", "src/sobfu/cuda/reductor.cu", "src/kfusion/cuda/tsdf_volume.cu", "src/kfusion/cuda/imgproc.cu"]
CPP = ["src/sobfu/solver.cpp", "src/sobfu/vector_fields.cpp", "src/sobfu/reductor.cpp", "src/sobfu/precomp.cpp", "src/kfusion/device_memory.cpp",
       "src/kfusion/precomp.cpp", "src/kfusion/tsdf_volume.cpp", "src/kfusion/imgproc.cpp", "src/sobfu/sob_fusion.cpp", "src/kfusion/marching_cubes.cpp"]
# ieee: the *_rn intrinsics are HIP's plain operators, which -ffp-contract=off keeps un-contracted and IEEE-rounded (the same values as
#       rounded library calls, at full speed: this is also the build whose SPEED is quoted).
# fast: contraction is on, so the intrinsics must be opaque: -DOCML_BASIC_ROUNDED_OPERATIONS turns them into correctly rounded OCML calls
#       that the compiler cannot fuse -- what nvcc guarantees for them.
FLAVOURS = {"ieee": ["-ffp-contract=off"],
            "fast": ["-ffp-contract=fast", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero", "-DOCML_BASIC_ROUNDED_OPERATIONS"]}


def available():
    return os.path.isdir(os.path.join(REF, "src", "sobfu", "cuda"))


def binary(flavour):
    return os.path.join(OUT_DIR, "reference_hip_" + flavour)


def build(force=False, verbose=False):
    if not available():
        raise FileNotFoundError("%s is not mounted: the reference is only present in the build container" % REF)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OUT_DIR, exist_ok=True)
    srcs = [os.path.join(REF, r) for r in CU + CPP] + [os.path.join(HERE, "texture_state.cpp"), os.path.join(HERE, "mc_unavailable.cpp"),
            os.path.join(ROOT, "tools", "ref_emulation", "driver.cpp")]
    deps = srcs + [os.path.join(HERE, "shim", "cuda_runtime.h"), __file__]
    for flavour, fl in FLAVOURS.items():
        exe = binary(flavour)
        if not force and os.path.exists(exe) and all(os.path.getmtime(exe) >= os.path.getmtime(d) for d in deps):
            continue
        work = os.path.join(OUT_DIR, "hip_obj_" + flavour)
        os.makedirs(work, exist_ok=True)
        flags = ["--offload-arch=gfx950", "-x", "hip", "-fgpu-rdc", "-O3", "-std=c++14", "-w", "-fpermissive", "-DREF_HIP_BUILD=1",
                 *fl, "-I" + os.path.join(HERE, "shim"), "-I" + os.path.join(ROOT, "tools", "ref_emulation", "shim"), "-I" + os.path.join(REF, "include")]
        # host-only translation units go through g++ (the reference's host code needs -fpermissive, which clang does not have)
        host_flags = ["-std=c++14", "-O2", "-w", "-fpermissive", "-D__HIP_PLATFORM_AMD__", "-DREF_HIP_BUILD=1", "-I" + os.path.join(HERE, "shim"),
                      "-I" + os.path.join(ROOT, "tools", "ref_emulation", "shim"), "-I" + os.path.join(REF, "include"), "-I/opt/rocm/include"]
        procs, objs = [], []
        for i, src in enumerate(srcs):
            obj = os.path.join(work, "%02d_%s.o" % (i, os.path.basename(src)))
            objs.append(obj)
            device = src.endswith(".cu") or src.endswith("texture_state.cpp")
            cmd = [hipcc, *flags, "-c", src, "-o", obj] if device else ["g++", *host_flags, "-c", src, "-o", obj]
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        bad = False
        for src, p in procs:
            out = p.communicate()[0].decode()
            if p.returncode:
                bad = True
                sys.stderr.write("== %s\n%s\n" % (src, out[-5000:]))
        if bad:
            raise RuntimeError("reference HIP build (%s) failed" % flavour)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fgpu-rdc", "--hip-link", *objs, "-o", exe])
        for o in objs:
            os.remove(o)
        os.rmdir(work)
        if verbose:
            print(exe)
    return [binary(f) for f in FLAVOURS]


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
