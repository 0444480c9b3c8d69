#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" TILE_MODES=direct TILE_THR=-1 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //' | sed -E 's/ compute side.*//'; }
for g in 2x2x2 1x2x4 1x1x8; do
run TILE_GRIDS=$g SOBFU_TILE_PUSH_MARCH=0
run TILE_GRIDS=$g SOBFU_TILE_PUSH_MARCH=1
run TILE_GRIDS=$g SOBFU_TILE_PUSH_MARCH=0 SOBFU_TILED_DEBUG_SKIP=8
run TILE_GRIDS=$g SOBFU_TILE_PUSH_MARCH=1 SOBFU_TILED_DEBUG_SKIP=8
run TILE_GRIDS=$g SOBFU_TILE_PUSH_MARCH=1 SOBFU_TILED_DEBUG_SKIP=8 SOBFU_ZC_A=4
run TILE_GRIDS=$g SOBFU_TILE_PUSH_MARCH=1 SOBFU_TILED_DEBUG_SKIP=8 SOBFU_ZC_A=8
done
