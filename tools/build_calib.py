"""Builds tools/calib/calib_copy.hip -> build/calib_copy (hipcc, gfx950): the counter-calibration copies (tools/calibrate_counters.sh)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, "build", "calib_copy")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", os.path.join(ROOT, "tools", "calib", "calib_copy.hip"), "-o", out])
print(out)
