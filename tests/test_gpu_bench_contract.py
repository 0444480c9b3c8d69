"""bench.py prints ONE JSON line with the driver's contract fields (plus roofline / cpu_baseline) -- checked on a short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None):
    e = dict(os.environ, **(env or {}))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0]  # exactly one JSON line, and it is the last line
    return json.loads(lines[0])


def test_single_gpu_line():
    d = run_bench("--gpus", "1", "--steps", "24", "--warmup", "8", "--dim", "128")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 24 and d["warmup"] == 8 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "iterations/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["avg_launch_ms"] > 0 and r["algorithmic_bytes_per_launch"] == 128 ** 3 * 64
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "iterations/s" and c["sample"]


def test_slab_path_line_on_one_gpu():
    d = run_bench("--gpus", "1", "--steps", "20", "--warmup", "6", "--dim", "128", "--no-cpu-baseline", env={"SOBFU_FORCE_TILED": "1"})
    assert d["tiled_parity_vs_single_gpu"] == "bit-exact" and "native C++ loop" in d["config"]["parallelism"]
    assert d["tiled_diag"]["exchange_bytes_per_face"] == 4 * 128 * 128 * 12 and d["tiled_diag"]["iteration_us_compute_only"] > 0
