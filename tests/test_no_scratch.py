"""No kernel of the shipped library may use scratch (private segment): a spill or a stack object turns register traffic into memory traffic on the
critical path.  Found twice by reading the ISA -- the 7-tap y chain of pass B under the SLP vectoriser (round 1: 390 instead of 180 us) and a select
between two float4 objects in the inverse fixed point, compiled to a load through a selected stack address (round 6: the frame's inverse + warp 537
instead of 260 us) -- so it is checked here on the code objects inside sobfu_amd/libsobfu_hip.so (CPU only: llvm-objdump / llvm-readelf)."""
import glob
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="ROCm's llvm tools are not installed")
def test_no_kernel_uses_scratch():
    lib = os.path.join(ROOT, "sobfu_amd", "libsobfu_hip.so")
    assert os.path.exists(lib), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    d = tempfile.mkdtemp(prefix="scratch_check_")
    try:
        shutil.copy(lib, d)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "libsobfu_hip.so"], cwd=d, capture_output=True, check=True)
        objs = glob.glob(os.path.join(d, "*gfx950*"))
        assert len(objs) >= 6, objs  # one code object per translation unit with kernels
        kernels, bad = 0, []
        for o in objs:
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", o], capture_output=True, text=True, check=True).stdout
            for name, size, spills in re.findall(r"\.name:\s+(\S+)\n\s*\.private_segment_fixed_size:\s+(\d+)\n(?:[^\n]*\n)*?\s*\.vgpr_spill_count:\s+(\d+)", notes):
                kernels += 1
                if int(size) != 0 or int(spills) != 0:
                    bad.append((name, int(size), int(spills)))
        assert kernels >= 80, kernels
        assert not bad, "kernels with scratch: %s" % bad
    finally:
        shutil.rmtree(d, ignore_errors=True)
