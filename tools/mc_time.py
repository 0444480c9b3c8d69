import sys, time, torch
sys.path.insert(0, "/root/repo")
from sobfu_amd import ops
n, vs = 256, 0.75 / 256
vol = ops.new_volume((n, n, n)); ops.init_sphere(vol, (vs,) * 3, 48 * vs, 3 * vs, (0.375, 0.37, 0.38), 0.2)
ws = ops.mc_workspace(vol)  # kept between calls, as kfusion::cuda::MarchingCubes does: no allocation inside the scan steps
for k in range(3):
    for w, tag in ((None, "scratch allocated per call"), (ws, "caller-kept workspace")):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v, nn = ops.marching_cubes(vol, (0.75,) * 3, workspace=w)
        torch.cuda.synchronize(); print(f"marching cubes 256^3 ({tag}): {1e3 * (time.perf_counter() - t0):.2f} ms, {len(v) // 3} triangles")
