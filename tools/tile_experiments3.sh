#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" TILE_MODES=direct TILE_THR=-1 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //'; }
for g in 2x2x2 1x2x4; do
run TILE_GRIDS=$g SOBFU_PIPE_B=0 SOBFU_CACHE_CELLS=0
run TILE_GRIDS=$g SOBFU_PIPE_B=0
run TILE_GRIDS=$g SOBFU_PIPE_B=1
run TILE_GRIDS=$g SOBFU_PIPE_B=1 SOBFU_TILED_DEBUG_SKIP=4
run TILE_GRIDS=$g SOBFU_PIPE_B=1 SOBFU_TILED_DEBUG_SKIP=6
run TILE_GRIDS=$g SOBFU_PIPE_B=1 SOBFU_TILED_DEBUG_SKIP=8
run TILE_GRIDS=$g SOBFU_PIPE_B=1 SOBFU_TILED_DEBUG_SKIP=9
for z in 6 8 16; do run TILE_GRIDS=$g SOBFU_PIPE_B=1 SOBFU_ZC_B=$z; done
for z in 2 8; do run TILE_GRIDS=$g SOBFU_PIPE_B=1 SOBFU_ZC_A=$z; done
done
b() { echo "== bench $*"; env "$@" python bench.py --steps 50 --warmup 20 --repeats 5 --no-cpu-baseline --no-traffic --frames 0 2>/dev/null | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.0f it/s  passA %.1f us  passB %.1f us' % (d['value'], r['pass_a']['avg_launch_ms']*1e3, r['avg_launch_ms']*1e3))"; }
for i in 1 2; do
b SOBFU_PIPE_B=0
b SOBFU_PIPE_B=1
done
b SOBFU_PIPE_B=1 SOBFU_ZC_B=64
b SOBFU_PIPE_B=1 SOBFU_ZC_B=32
V=$PWD/build/variants/libsobfu_hip_1x8xSOBFU_MINW_PIPE=6.so
run TILE_GRIDS=2x2x2 SOBFU_PIPE_B=1 SOBFU_HIP_LIB=$V
b SOBFU_PIPE_B=1 SOBFU_HIP_LIB=$V
b SOBFU_PIPE_B=1 SOBFU_HIP_LIB=$V
b SOBFU_PIPE_B=0 SOBFU_HIP_LIB=$V
