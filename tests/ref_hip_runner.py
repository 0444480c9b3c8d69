"""Runs oracle/_ref/reference_hip_{ieee,fast} (the reference's own kernels compiled for gfx950 by oracle/ref_hipbuild/build.py in the build
container; the binaries travel to the GPU box, the reference does not): writes the inputs as raw files, runs one scenario of the driver,
reads the outputs back.  Test infrastructure (shim evidence, see oracle/ref_hipbuild/shim/cuda_runtime.h)."""
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


def word_digest(a, chunk=1 << 26):
    """the driver's digest=1 checksum of an array: sum over its 32-bit words w_i of w_i * (2 i + 1) mod 2^64"""
    w = np.ascontiguousarray(a).reshape(-1).view(np.uint32)
    total = np.uint64(0)
    with np.errstate(over="ignore"):
        for i in range(0, w.size, chunk):
            part = w[i:i + chunk].astype(np.uint64)
            idx = np.arange(i, i + part.size, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
            total = total + (part * idx).sum(dtype=np.uint64)
    return int(total)


def word_digest_device(t, chunk=1 << 27):
    """word_digest of a CUDA tensor, computed on the device (int64 arithmetic wraps like uint64: the low 64 bits are the same)"""
    import torch

    w = t.contiguous().view(-1).view(torch.int32)
    total = 0
    for i in range(0, w.numel(), chunk):
        part = w[i:i + chunk].to(torch.int64) & 0xFFFFFFFF
        idx = torch.arange(i, i + part.numel(), dtype=torch.int64, device=w.device) * 2 + 1
        total = (total + int((part * idx).sum().item())) & 0xFFFFFFFFFFFFFFFF
    return total


def run(flavour, scenario, inputs, outputs, **kw):
    """inputs: name -> array; outputs: name -> (dtype, shape or None).  -> dict of arrays + 'log' (the reference's stdout) [+ 'time': seconds].
    With digest=1 every output is the 64-bit word_digest of the array instead (an int)."""
    d = tempfile.mkdtemp(prefix="ref_hip_")
    try:
        for k, a in inputs.items():
            np.ascontiguousarray(a).tofile(os.path.join(d, k + ".bin"))
        r = subprocess.run([binary(flavour), scenario, d] + ["%s=%r" % (k, float(v)) for k, v in kw.items()], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        out = {}
        for k, (dt, shape) in outputs.items():
            if kw.get("digest"):
                out[k] = int(np.fromfile(os.path.join(d, "out_" + k + ".bin"), dtype=np.uint64)[0])
                continue
            a = np.fromfile(os.path.join(d, "out_" + k + ".bin"), dtype=dt)
            out[k] = a.reshape(shape) if shape is not None else a
        out["log"] = open(os.path.join(d, "out_log.txt")).read()
        t = os.path.join(d, "out_time.txt")
        if os.path.exists(t):
            out["time"] = [float(x) for x in open(t).read().split()]
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)
