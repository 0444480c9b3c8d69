"""Builds the C++ host-shell test driver (tests/cpp/host_shell_tests.cpp over include/sobfu_amd/sobfu.hpp) with g++.

The shells are header-only host code; the driver links libsobfu_hip.so (C ABI) and libamdhip64.so."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "build", "host_shell_tests")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


def build_host(force: bool = False) -> str:
    src = os.path.join(ROOT, "tests", "cpp", "host_shell_tests.cpp")
    deps = [src, os.path.join(ROOT, "include", "sobfu_amd", "sobfu.hpp"), os.path.join(ROOT, "include", "sobfu_hip.h"),
            os.path.join(HERE, "libsobfu_hip.so")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__",
                           f"-I{ROCM}/include", f"-I{os.path.join(ROOT, 'include')}", src, "-o", OUT, f"-L{HERE}", "-lsobfu_hip",
                           f"-L{ROCM}/lib", "-lamdhip64", f"-Wl,-rpath,{HERE}", f"-Wl,-rpath,{ROCM}/lib", "-Wl,-rpath,$ORIGIN/../sobfu_amd"])
    return OUT


if __name__ == "__main__":
    print(build_host(force=True))
