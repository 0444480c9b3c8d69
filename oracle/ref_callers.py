"""Builds the REFERENCE'S OWN test translation units (test/{solver_test,reductions_test,deformation_field_test,main}.cpp), unchanged and
from where they lie under /root/reference, against this repo's include/ (the header-only C++ shells over the C ABI) and links them with
libsobfu_hip.so -> oracle/_ref/reference_gtests.

TEST INFRASTRUCTURE ONLY, and NOT a reference build in the sense of an oracle: no line of the reference's IMPLEMENTATION is compiled here
(its .cu / .cpp sources of the path are exactly what this repo replaces) -- only its test drivers, against this repo's library.  The
binary therefore pins nothing about the oracle; the oracle's pin status (oracle/sobfu_oracle.c header) is unchanged.
No reference source is copied: the compiler reads the four files in place, only the binary lands in
oracle/_ref/ (git-ignored, but it travels to the GPU box with the snapshot, where /root/reference does not exist).  The image has no
GoogleTest; tests/cpp/gtest_stub/gtest/gtest.h supplies the five names those files use (TEST_F, ASSERT_NEAR, ::testing::Test,
InitGoogleTest, RUN_ALL_TESTS).  What this is evidence for: the drop-in claim of INTEGRATION.md -- the reference's callers of the path
compile against the replacement without an edit -- and, run on the GPU (tests/test_gpu_reference_gtests.py), that the six value-pinning
gtest cases the reference holds pass on the HIP path THROUGH THE REFERENCE'S OWN TEST CODE.

Not built: src/sobfu/sob_fusion.cpp and src/apps/demo.cpp -- their signatures carry pcl::PolygonMesh / cv::viz / boost::program_options
(PCL, OpenCV viz, Boost: absent here); the shell's SobFusion class REPLACES sob_fusion.cpp.
"""
from __future__ import annotations

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
REF = os.environ.get("SOBFU_REFERENCE_DIR", "/root/reference")
OUT_DIR = os.path.join(_HERE, "_ref")
OUT = os.path.join(OUT_DIR, "reference_gtests")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
UNITS = ("solver_test", "reductions_test", "deformation_field_test", "main")


def sources():
    return [os.path.join(REF, "test", u + ".cpp") for u in UNITS]


def available() -> bool:
    return all(os.path.exists(s) for s in sources())


def compile_flags():
    return ["-std=c++14", "-O1", "-w", "-D__HIP_PLATFORM_AMD__", f"-I{os.path.join(ROOT, 'tests', 'cpp', 'gtest_stub')}", f"-I{ROCM}/include",
            f"-I{os.path.join(ROOT, 'include')}"]


def build(force: bool = False) -> str:
    """-> path of the binary; raises when /root/reference is absent (callers check available())"""
    if not available():
        raise FileNotFoundError(f"{REF}/test/*.cpp not found: the reference is only mounted in the build container")
    lib = os.path.join(ROOT, "sobfu_amd", "libsobfu_hip.so")
    deps = sources() + [os.path.join(ROOT, "include", "sobfu_amd", "sobfu.hpp"), os.path.join(ROOT, "include", "sobfu_hip.h"),
                        os.path.join(ROOT, "tests", "cpp", "gtest_stub", "gtest", "gtest.h"), lib]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    objs = []
    for u, src in zip(UNITS, sources()):
        o = os.path.join(OUT_DIR, u + ".o")
        subprocess.check_call(["g++", *compile_flags(), "-c", src, "-o", o])
        objs.append(o)
    subprocess.check_call(["g++", *objs, "-o", OUT, f"-L{os.path.join(ROOT, 'sobfu_amd')}", "-lsobfu_hip", f"-L{ROCM}/lib", "-lamdhip64", "-lz",
                           "-Wl,-rpath,$ORIGIN/../../sobfu_amd", f"-Wl,-rpath,{ROCM}/lib"])
    for o in objs:
        os.remove(o)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
