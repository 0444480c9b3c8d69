"""The emulated-reference recipe is regenerable: in the build container (where /root/reference is mounted) a subset of
tests/golden/ref_*.npz is rebuilt from a clean temp dir through tools/ref_emulation/ and must come out BYTE-IDENTICAL to the
committed files (the full set: `python tests/golden/make_reference_fixtures.py --check`, ~6 min).  Also: nothing outside
tests/golden/ imports the recipe, and the recipe keeps no reference text in the repo."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("SOBFU_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src", "sobfu", "cuda")), reason="the reference is only mounted in the build container")
def test_subset_regenerates_byte_identically():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_reference_fixtures.py"), "--check",
                        "--only=ref_kernels_17x9x5,ref_solver_20x12x9,ref_mc_14x11x9,ref_depth_32x32x32"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("identical") == 4 and "DIFFERS" not in r.stdout, r.stdout


def test_recipe_is_imported_by_the_fixture_script_only():
    users = []
    for base, _, files in os.walk(ROOT):
        if any(part in base for part in (".git", "gpurun_out", "__pycache__", os.path.join("tools", "ref_emulation"))):  # the recipe itself
            continue
        for f in files:
            if f.endswith((".py", ".sh")):
                text = open(os.path.join(base, f), errors="replace").read()
                # imports or executions of the recipe (mentions in docstrings do not count)
                if re.search(r"^\s*(import|from)\s+make_reference_fixtures|import build as emu_build|\"ref_emulation\"|make_reference_fixtures\.py\"\)", text, re.M):
                    users.append(os.path.relpath(os.path.join(base, f), ROOT))
    # oracle/ref_hipbuild/build.py shares the scenario driver and the OpenCV / PCL / Boost stand-ins (the hipcc build of the reference, GPU tests only)
    assert sorted(users) == ["oracle/ref_hipbuild/build.py", "tests/golden/make_reference_fixtures.py", "tests/test_reference_recipe.py"], users


def test_launch_rewrite_is_one_regular_expression():
    sys.path.insert(0, os.path.join(ROOT, "tools", "ref_emulation"))
    try:
        import build as emu_build
    finally:
        sys.path.pop(0)
    out, n = emu_build.rewrite_launches("k<<<g, b>>>(a, f(x), *(p->q));\nns::t<512, true>\n    <<<dim3(1), dim3(2), s, st>>>(u, v);\n")
    assert n == 2 and out == "CUEMU_LAUNCH((k), (g, b), (a, f(x), *(p->q)));\nCUEMU_LAUNCH((ns::t<512, true>), (dim3(1), dim3(2), s, st), (u, v));\n"
    with pytest.raises(RuntimeError):
        emu_build.rewrite_launches("k<<<g, b>>> stray;")
    # stand-in headers are this repo's: none of them is a copy of a reference file
    shim = os.path.join(ROOT, "tools", "ref_emulation", "shim")
    assert not any(f.endswith((".cu", ".cuh")) for _, _, fs in os.walk(shim) for f in fs)
