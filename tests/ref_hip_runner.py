"""Runs oracle/_ref/reference_hip_{ieee,fast} (the reference's own kernels compiled for gfx950 by tools/ref_hipbuild/build.py in the build
container; the binaries travel to the GPU box, the reference does not): writes the inputs as raw files, runs one scenario of the driver,
reads the outputs back.  Test infrastructure (shim evidence, see tools/ref_hipbuild/shim/cuda_runtime.h)."""
import os
import shutil
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def binary(flavour):
    return os.path.join(ROOT, "oracle", "_ref", "reference_hip_" + flavour)


def available():
    return all(os.path.exists(binary(f)) for f in ("ieee", "fast"))


def run(flavour, scenario, inputs, outputs, **kw):
    """inputs: name -> array; outputs: name -> (dtype, shape or None).  -> dict of arrays + 'log' (the reference's stdout) [+ 'time': seconds]"""
    d = tempfile.mkdtemp(prefix="ref_hip_")
    try:
        for k, a in inputs.items():
            np.ascontiguousarray(a).tofile(os.path.join(d, k + ".bin"))
        r = subprocess.run([binary(flavour), scenario, d] + ["%s=%r" % (k, float(v)) for k, v in kw.items()], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out = {}
        for k, (dt, shape) in outputs.items():
            a = np.fromfile(os.path.join(d, "out_" + k + ".bin"), dtype=dt)
            out[k] = a.reshape(shape) if shape is not None else a
        out["log"] = open(os.path.join(d, "out_log.txt")).read()
        t = os.path.join(d, "out_time.txt")
        if os.path.exists(t):
            out["time"] = [float(x) for x in open(t).read().split()]
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)
