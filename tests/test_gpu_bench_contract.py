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
    assert r.returncode == 0, r.stderr[-12000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.rstrip().splitlines()[-1] == lines[0]  # exactly one JSON line, and it is the last line
    return json.loads(lines[0])


def test_single_gpu_line():
    d = run_bench("--gpus", "1", "--steps", "24", "--warmup", "8", "--dim", "128", "--repeats", "5")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "repeats", "region_its", "per_solve", "per_frame"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 24 and d["warmup"] == 8 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "iterations/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    assert d["repeats"] == 5 and len(d["region_its"]) == 5 and min(d["region_its"]) <= d["value"] <= max(d["region_its"]) + 0.1
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["avg_launch_ms"] > 0 and r["algorithmic_bytes_per_launch"] == 128 ** 3 * 64 and r["launches_timed"] == 2 * 24
    # roofline.traffic is measured live (two rocprofv3 PMC passes): at least the bytes the compact format must move, not absurdly more
    assert r["traffic"] is not None, r["traffic_how"]
    assert 0.3 * 128 ** 3 * 44 < r["traffic"] < 3.0 * 128 ** 3 * 44 and 0 < r["traffic_frac"] < 1  # (a 128^3 grid partly lives in the L2s)
    assert 0 < r["frac_physical"] < 1 and 0 < r["pass_a"]["frac_physical"] < 1
    assert r["physical_bytes_per_launch"] == 128 ** 3 * 44
    assert d["per_solve"]["iterations"] == 50 and d["per_solve"]["ms"] > 0
    f = d["per_frame"]  # the whole per-frame pipeline on the bench grid (5 frames by default, the first one untimed)
    assert f["frames_timed"] == 4 and f["iterations_per_frame"] == [50] * 4 and f["frames_per_s_per_gpu"] > 0 and "128^3" in f["config"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "iterations/s" and c["sample"]
    assert c["one_core"]["cores"] == 1 and c["one_core"]["value"] > 0
    assert c["config1_64"]["all_cores"]["value"] > 0 and c["config1_64"]["one_core"]["value"] > 0
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "reference_hip_ieee")):  # the baseline leg's second number: the reference's kernels on this GPU
        rb = d["reference_build_on_this_gpu"]
        assert rb["value"] > 0 and rb["unit"] == "iterations/s" and rb["this_repo_over_reference_build"] > 2.0 and "shim evidence" in rb["kind"], rb


def test_gpus_n_self_launches_replicas():
    """plain `python bench.py --gpus 2` (no torchrun): bench.py re-executes itself under torch.distributed.run.  On this 1-GPU box
    both ranks share cuda:0 (gloo process group); --replicas = BASELINE config 5's batched independent sequences."""
    env = {"SOBFU_BENCH_SHARE_GPU": "1"}
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        assert k not in os.environ
    d = run_bench("--gpus", "2", "--replicas", "--steps", "10", "--warmup", "4", "--dim", "64", "--repeats", "3", "--no-cpu-baseline", env=env)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["repeats"] == 3 and "replicas" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6


def test_config5_replicas_512():
    """BASELINE config 5's shape: 512^3 grids, one independent sequence per rank (two ranks sharing this box's GPU)"""
    d = run_bench("--gpus", "2", "--replicas", "--dim", "512", "--steps", "4", "--warmup", "2", "--repeats", "2", "--profile-repeats", "1",
                  "--no-cpu-baseline", "--frames", "3", "--frame-config", "config5", "--frame-iters", "6", env={"SOBFU_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["grid"] == [512, 512, 512] and d["value"] > 0
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 512 ** 3 * 64
    # ... and the frames/s of the whole per-frame pipeline (pre-steps -> integrate -> estimate_psi -> fuse) on those sequences
    f = d["per_frame"]
    assert "config5_umbrella_512.ini" in f["config"] and "512^3" in f["config"] and f["sequences"] == 2 and f["frames_timed"] == 2
    assert f["iterations_per_frame"] == [6, 6] and f["ms_per_frame"] > 0 and f["psi_moved_max_abs"] > 0
    assert abs(f["frames_per_s_aggregate"] - 2 * f["frames_per_s_per_gpu"]) < 1e-9 and abs(f["frames_per_s_per_gpu"] - 1e3 / f["ms_per_frame"]) < 1e-9


def test_tile_path_line_on_one_gpu():
    d = run_bench("--gpus", "1", "--steps", "20", "--warmup", "6", "--dim", "128", "--repeats", "3", "--no-cpu-baseline", env={"SOBFU_FORCE_TILED": "1"})
    assert d["tiled_parity_vs_single_gpu"] == "bit-exact" and "native C++ loop" in d["config"]["parallelism"]
    assert d["tiled_diag"]["iteration_us_compute_only"] > 0 and d["tiles"]["grid"] == [1, 1, 1]


@pytest.mark.parametrize("n,grid,transport", [(8, [2, 2, 2], "direct"), (8, [2, 2, 2], "rccl"), (4, [1, 2, 2], "direct"), (2, [1, 1, 2], "rccl")])
def test_gpus_n_strong_scaling_self_launch(n, grid, transport):
    """plain `python bench.py --gpus 8`: self-launch, the default tile grid (2 x 2 x 2 at N = 8), the native loop on every rank,
    one REAL process per rank -- on this 1-GPU box the ranks share cuda:0.  direct: the ranks map each other's arrays with hipIpc
    and the halo cells, arrival flags and max-norm rows travel as plain stores between the processes (in-kernel waits live);
    rccl: the packed messages travel over gloo instead of RCCL.  Every rank checks its tile against the single-GPU solve bit for
    bit; the line names the transport."""
    d = run_bench("--gpus", str(n), "--steps", "6", "--warmup", "2", "--dim", "64", "--repeats", "2",
                  env={"SOBFU_BENCH_SHARE_GPU": "1", "SOBFU_TILED_DIAG": "0", "SOBFU_TILED_TRANSPORT": transport})
    assert d["n_gpus"] == n and d["scaling"] == "strong" and d["tiles"]["grid"] == grid
    assert d["transport"] == transport, d.get("transport_fallback")
    assert d["best_grid"]["grid"] == grid and d["best_grid"]["transport"] == transport and d["best_grid"]["value"] == d["value"]  # no sweep: the timed leg
    assert d["tiled_parity_vs_single_gpu"] == "bit-exact" and "native C++ loop" in d["config"]["parallelism"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) / d["value"] < 1e-6 and "cpu_baseline" not in d
    t = d["tiled_iteration_ms"]
    a_key = "pass_a_incl_message_stores" + ("_and_peer_wait" if transport == "direct" else "")
    assert t[a_key] > 0 and t["pass_b"] > 0 and (t["exchange_transfer_and_scatter"] > 0 or transport == "direct")
    assert d["roofline"]["algorithmic_bytes_per_launch"] == 64 ** 3 // n * 64 and d["roofline"]["launches_timed"] == 2 * 6


def test_gpus_n_tiles_auto():
    d = run_bench("--gpus", "4", "--tiles", "auto", "--steps", "5", "--warmup", "1", "--dim", "64", "--repeats", "2",
                  env={"SOBFU_BENCH_SHARE_GPU": "1", "SOBFU_TILED_DIAG": "0"})
    t = d["tiled_autotune_us"]["direct"]  # the grid is chosen on the first usable transport
    assert set(t) == {"1x1x4", "1x2x2"} and "x".join(map(str, d["tiles"]["grid"])) in t and all(v > 0 for v in t.values())
    assert d["tiled_parity_vs_single_gpu"] == "bit-exact" and set(d["legs"]) == {"direct", "rccl"}


def test_gpus_n_explicit_tiles_and_threshold():
    d = run_bench("--gpus", "4", "--tiles", "2x2x1", "--steps", "5", "--warmup", "1", "--dim", "64", "--repeats", "2",
                  env={"SOBFU_BENCH_SHARE_GPU": "1", "SOBFU_TILED_DIAG": "0"})
    assert d["tiles"]["grid"] == [2, 2, 1] and d["tiled_parity_vs_single_gpu"] == "bit-exact"


def test_direct_transport_probe_child_dies_falls_back():
    """the direct transport is first exercised in a CHILD of every rank (bench_probe.py): one child dying the way a GPU
    memory fault would kill it must cost the run nothing but the transport -- every rank agrees on RCCL and the line says why"""
    d = run_bench("--gpus", "2", "--steps", "4", "--warmup", "1", "--dim", "64", "--repeats", "2",
                  env={"SOBFU_BENCH_SHARE_GPU": "1", "SOBFU_TILED_DIAG": "0", "SOBFU_PROBE_TEST_ABORT": "1", "SOBFU_PROBE_TIMEOUT_S": "20"})
    assert d["transport"] == "rccl" and "sandboxed probe" in d["transport_fallback"], d
    assert d["tiled_parity_vs_single_gpu"] == "bit-exact" and "sandboxed probe" in d["legs"]["direct"]["failed"] and d["legs"]["rccl"]["value"] > 0


def test_gpus_8_harvests_everything():
    """ONE `python bench.py --gpus 8` records everything the machine can tell: both transports timed on
    BASELINE config 4's grid with `value` = the better bit-exact one, every grid of 8 tiles on both transports, the direct
    transport's flag round trips / push-box rate / unhidden wait, one exchange and one all-reduce of the RCCL leg, a topology
    snapshot, frames/s of the whole pipeline ON TILES.  Eight real processes sharing this box's GPU (RCCL leg over gloo)."""
    d = run_bench("--gpus", "8", "--steps", "6", "--warmup", "2", "--dim", "64", "--repeats", "2", "--frame-iters", "8",
                  env={"SOBFU_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 8 and d["tiles"]["grid"] == [2, 2, 2] and d["tiled_parity_vs_single_gpu"] == "bit-exact"
    legs = d["legs"]
    assert set(legs) == {"direct", "rccl"} and d["transport"] in legs
    for name, leg in legs.items():
        assert leg["tiled_parity_vs_single_gpu"] == "bit-exact" and leg["value"] > 0 and len(leg["region_its"]) == 2, (name, leg)
        assert leg["ms_a"] > 0 and leg["ms_b"] > 0
    assert d["value"] == max(leg["value"] for leg in legs.values()) and abs(d["value"] - legs[d["transport"]]["value"]) < 1e-9
    assert legs["direct"]["peer_wait_us_per_iteration"] >= 0 and legs["rccl"]["ms_exchange"] > 0
    bg = d["best_grid"]  # the fastest (grid, transport) of the run is a first-class number beside `value`, which stays on config 4's grid
    assert bg["value"] >= d["value"] - 1e-9 and bg["transport"] in legs and "x".join(map(str, bg["grid"])) in {"1x1x8", "1x2x4", "2x2x2"} and bg["candidates"] == 6
    grids = d["tiled_autotune_us"]
    assert set(grids) == {"direct", "rccl"} and all(set(g) == {"1x1x8", "1x2x4", "2x2x2"} and all(v and v > 0 for v in g.values()) for g in grids.values())
    dd = d["tiled_diag"]["direct_diag"]
    assert len(dd["flag_round_trip_us"]) == 28 and all(v > 0 for v in dd["flag_round_trip_us"].values())
    # (eight processes time-sharing ONE GPU make every handshake milliseconds long: only presence and sign are checked here)
    assert dd["push_boxes_only_us"] > 0 and dd["push_boxes_only_local_stores_us"] > 0 and dd["push_rate_GBps_of_bytes_out"] >= 0 and dd["status_ok"]
    assert dd["messages"] == 6 and dd["bytes_out_per_iteration"] == 12 * (3 * 4 * 32 * 32 + 3 * 4 * 4 * 32)
    rd = d["tiled_diag"]["rccl_diag"]
    assert rd["exchange_4_cells_us"] > 0 and d["tiled_diag"]["iteration_us_compute_only"] > 0
    topo = d["topology"]
    assert topo["devices_visible"] >= 1 and "pairs" in topo
    f = d["per_frame"]
    assert f["frames_timed"] == 3 and f["iterations_per_frame"][-1] == 8 and f["frames_per_s"] > 0 and f["tiles"] == "2x2x2"
    # the per-frame tail moves bounded-reach windows, not two all-gathers, and says so; checked against the
    # all-gather tail bit for bit inside the run
    t = f["tail"]
    assert t["mode"] == ["halo"] and t["parity_vs_all_gather_tail"] == "bit-exact" and t["all_gather_would_move_bytes"] == 64 ** 3 * 24
    assert 0 < f["all_gathered_bytes_per_frame"] == t["bytes_received_per_frame_per_rank"] < 0.25 * 64 ** 3 * 24
    assert t["halo_width_cells"] and max(t["halo_width_cells"]) <= 6 and 0 < t["max_displacement_voxels"] < 4


def test_budget_skips_the_harvest_and_keeps_one_valid_line():
    """with a budget that is gone before the timed legs are done, every harvest step behind them is skipped
    -- by name, agreed between the ranks -- and the line is still ONE valid JSON line with the core of the run (value, both legs, parity)"""
    d = run_bench("--gpus", "2", "--steps", "10", "--warmup", "4", "--dim", "64", "--repeats", "3", "--no-cpu-baseline", "--budget-s", "1",
                  env={"SOBFU_BENCH_SHARE_GPU": "1"})
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["budget_s"] == 1.0
    assert d["tiled_parity_vs_single_gpu"] == "bit-exact"
    assert set(d["legs"]) == {"direct", "rccl"} and all("value" in v or "failed" in v for v in d["legs"].values())
    sk = d["skipped"]
    # either the normal path skipped every step by name, or the watchdog printed the early copy (which says so in one entry)
    assert any("per_frame" in x for x in sk) or any("budget" in x for x in sk), sk
    assert "per_frame" not in d and not d.get("tiled_autotune_us")


def test_single_gpu_budget_skips_named_steps():
    d = run_bench("--gpus", "1", "--steps", "10", "--warmup", "4", "--dim", "64", "--repeats", "3", "--budget-s", "1")
    assert d["value"] > 0 and d["roofline"]["traffic"] is None and "budget" in d["roofline"]["traffic_how"]
    assert set(d["skipped"]) == {"roofline.traffic", "cpu_baseline", "per_frame"} and "cpu_baseline" not in d and "per_frame" not in d


@pytest.mark.parametrize("where", ["harvest", "final barrier"])
def test_a_hang_behind_the_timed_legs_never_costs_the_line(where):
    """the safety nets of the one real multi-GPU run (test hook SOBFU_BENCH_TEST_HANG).  "harvest": a harvest step that never returns --
    the harvest thread's join is bounded by the budget, the run goes on without it and prints the FULL line, saying what is missing.
    "final barrier": a rank that never reaches the barrier in front of the line (a sick node) -- nothing in the normal path can end
    that; 25 s past the budget (30 s behind the core; 6 / 11 s in this test) every rank's watchdog ends the run, rank 0 having printed the core line it kept when
    the timed legs finished.  Either way: exactly one valid JSON line, exit code 0."""
    import time

    t0 = time.time()
    d = run_bench("--gpus", "2", "--steps", "10", "--warmup", "4", "--dim", "64", "--repeats", "3", "--no-cpu-baseline", "--budget-s", "12",
                  env={"SOBFU_BENCH_SHARE_GPU": "1", "SOBFU_BENCH_TEST_HANG": "1" if where == "harvest" else "2", "SOBFU_BENCH_WATCHDOG_GRACE_S": "6"})
    took = time.time() - t0
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["tiled_parity_vs_single_gpu"] == "bit-exact" and set(d["legs"]) == {"direct", "rccl"}
    assert "per_frame" not in d
    if where == "harvest":
        assert any("still running" in x for x in d["skipped"]), d["skipped"]
        assert "late_failure" not in d  # the normal path printed the full line
    else:
        assert any("everything after the timed legs" in x for x in d["skipped"]), d["skipped"]  # the early copy, printed by the watchdog
        # ... which says that it is a salvaged line (the exit code stays 0): printed by the watchdog, or by the guard around main() when
        # a late collective raised first because the peer's watchdog had already taken it out
        assert "watchdog" in d["late_failure"] or "after the timed legs" in d["late_failure"], d["late_failure"]
        assert 15 < took < 120, took  # budget 12 s + 6 s grace (production: 25 s), or 11 s behind the core, + start-up
