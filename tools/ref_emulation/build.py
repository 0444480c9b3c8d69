"""Builds the host emulation of the reference's hot path (SURVEY.md Appendix B's recipe, made regenerable).

What happens, all inside a scratch directory (default: a fresh tempfile.mkdtemp(); NOTHING generated lands in the repo):

  1. the reference's .cu files are copied and their `kernel<<<cfg>>>(args);` launch statements are rewritten, by ONE regular
     expression, into `CUEMU_LAUNCH((kernel), (cfg), (args));` -- a macro of shim/cuda_runtime.h that runs the same kernel
     function over the same grid on the CPU (cuemu.cpp: blocks spread over host threads, the threads of a block as coroutines);
  2. a copy of include/sobfu/cuda/utils.hpp gets its one `extern __shared__ int __smem[];` turned into a read of the launch's
     dynamic shared buffer (g++ has no notion of an unsized extern shared array);
  3. the copies, the reference's host .cpp files IN PLACE under /root/reference, cuemu.cpp and driver.cpp are compiled with
     g++ -O2 -ffp-contract=off against shim/ (stand-ins for CUDA / OpenCV / PCL / Boost headers, written for this repo) and
     the reference's own include/ directory, and linked into one binary, `ref_emu`.

This is SHIM EVIDENCE (stand-in headers for a toolkit the image lacks), not a reference build: by the task's rules it pins
nothing, and DESIGN.md section 2 says so.  It is used only by tests/golden/make_reference_fixtures.py, in the build container,
to produce arrays; nothing under tests/, sobfu_amd/, bench.py or __graft_entry__.py imports or executes it.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SOBFU_REFERENCE", "/root/reference")

CU = ["src/sobfu/cuda/solver.cu", "src/sobfu/cuda/vector_fields.cu", "src/sobfu/cuda/reductor.cu", "src/kfusion/cuda/tsdf_volume.cu",
      "src/kfusion/cuda/imgproc.cu", "src/kfusion/cuda/marching_cubes.cu"]
CPP = ["src/sobfu/solver.cpp", "src/sobfu/vector_fields.cpp", "src/sobfu/reductor.cpp", "src/sobfu/precomp.cpp", "src/sobfu/sob_fusion.cpp",
       "src/kfusion/device_memory.cpp", "src/kfusion/precomp.cpp", "src/kfusion/tsdf_volume.cpp", "src/kfusion/imgproc.cpp",
       "src/kfusion/marching_cubes.cpp"]

# name (optionally qualified, optionally with one template-argument list) <<< cfg >>> ( args );
LAUNCH = re.compile(r"([A-Za-z_][\w:]*(?:<[^<>;()]*?>)?)\s*<<<(.*?)>>>\s*\((.*?)\);", re.S)


def rewrite_launches(text):
    out, n = LAUNCH.subn(lambda m: "CUEMU_LAUNCH((%s), (%s), (%s));" % (m.group(1), m.group(2), m.group(3)), text)
    if "<<<" in out:
        raise RuntimeError("launch statement the rewrite did not match")
    return out, n


def build(work=None, arch="610", verbose=False):
    """Returns the path of the emulator binary; arch=None leaves __CUDA_ARCH__ undefined (pre-Kepler shared-memory tails)."""
    if not os.path.isdir(REF):
        raise RuntimeError("%s is not mounted: the emulation only runs in the build container" % REF)
    work = work or tempfile.mkdtemp(prefix="ref_emu_")
    gen = os.path.join(work, "gen")
    os.makedirs(os.path.join(gen, "sobfu", "cuda"), exist_ok=True)
    launches = 0
    objs = []
    flags = ["-std=c++14", "-O2", "-ffp-contract=off", "-fopenmp", "-fpermissive", "-w", "-I" + gen, "-I" + os.path.join(HERE, "shim"),
             "-I" + os.path.join(REF, "include")]
    if arch:
        flags.append("-DCUEMU_ARCH=" + arch)
    utils = open(os.path.join(REF, "include/sobfu/cuda/utils.hpp")).read()
    assert utils.count("extern __shared__ int __smem[];") == 2
    open(os.path.join(gen, "sobfu/cuda/utils.hpp"), "w").write(
        utils.replace("extern __shared__ int __smem[];", "int *__smem = (int *) cuemu::dynamic_smem();"))
    jobs = []
    for rel in CU:
        text, n = rewrite_launches(open(os.path.join(REF, rel)).read())
        launches += n
        dst = os.path.join(gen, rel.replace("/", "_") + ".cpp")
        open(dst, "w").write(text)
        jobs.append(dst)
    jobs += [os.path.join(REF, rel) for rel in CPP] + [os.path.join(HERE, "cuemu.cpp"), os.path.join(HERE, "driver.cpp")]
    procs = []
    for src in jobs:
        obj = os.path.join(work, "%02d_%s.o" % (len(objs), os.path.basename(src)))
        objs.append(obj)
        procs.append((src, subprocess.Popen(["g++", *flags, "-c", src, "-o", obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode:
            failed = True
            sys.stderr.write("== %s\n%s\n" % (src, out[-6000:]))
    if failed:
        raise RuntimeError("emulation build failed")
    exe = os.path.join(work, "ref_emu")
    subprocess.check_call(["g++", "-fopenmp", *objs, "-o", exe])
    if verbose:
        print("ref_emu: %d launch statements rewritten across %d files -> %s" % (launches, len(CU), exe))
    return exe


if __name__ == "__main__":
    print(build(verbose=True))
    if "--keep" not in sys.argv:
        pass  # the scratch directory is the caller's to remove; make_reference_fixtures.py does
