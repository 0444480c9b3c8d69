"""Builds the C++ host-shell test driver (tests/cpp/host_shell_tests.cpp over include/sobfu_amd/sobfu.hpp) with g++.

The shells are header-only host code; the driver links libsobfu_hip.so (C ABI) and libamdhip64.so."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "build", "host_shell_tests")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")


APP = os.path.join(ROOT, "build", "sobfu_headless")


def _compile(src: str, out: str) -> None:
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__",
                           f"-I{ROCM}/include", f"-I{os.path.join(ROOT, 'include')}", src, "-o", out, f"-L{HERE}", "-lsobfu_hip",
                           f"-L{ROCM}/lib", "-lamdhip64", "-lz", f"-Wl,-rpath,{HERE}", f"-Wl,-rpath,{ROCM}/lib", "-Wl,-rpath,$ORIGIN/../sobfu_amd"])


def build_app(force: bool = False) -> str:
    """apps/sobfu_headless.cpp: the headless frame-loop app over the shells."""
    src = os.path.join(ROOT, "apps", "sobfu_headless.cpp")
    deps = [src, os.path.join(ROOT, "include", "sobfu_amd", "sobfu.hpp"), os.path.join(ROOT, "include", "sobfu_amd", "depth_io.hpp"),
            os.path.join(HERE, "libsobfu_hip.so")]
    if force or not os.path.exists(APP) or any(os.path.getmtime(APP) < os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(APP), exist_ok=True)
        _compile(src, APP)
    return APP


IO_TOOL = os.path.join(ROOT, "build", "depth_io_tool")


def build_io_tool(force: bool = False) -> str:
    """tests/cpp/depth_io_tool.cpp: CPU-only driver of the depth readers / .npy writer (needs zlib only)."""
    src = os.path.join(ROOT, "tests", "cpp", "depth_io_tool.cpp")
    deps = [src, os.path.join(ROOT, "include", "sobfu_amd", "depth_io.hpp")]
    if force or not os.path.exists(IO_TOOL) or any(os.path.getmtime(IO_TOOL) < os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(IO_TOOL), exist_ok=True)
        subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", f"-I{os.path.join(ROOT, 'include')}", src, "-o", IO_TOOL, "-lz"])
    return IO_TOOL


def build_host(force: bool = False) -> str:
    build_app(force)
    build_io_tool(force)
    src = os.path.join(ROOT, "tests", "cpp", "host_shell_tests.cpp")
    deps = [src, os.path.join(ROOT, "include", "sobfu_amd", "sobfu.hpp"), os.path.join(ROOT, "include", "sobfu_hip.h"),
            os.path.join(HERE, "libsobfu_hip.so")]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    _compile(src, OUT)
    return OUT


if __name__ == "__main__":
    print(build_host(force=True))
