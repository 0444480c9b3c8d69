#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" TILE_MODES=direct TILE_THR=-1 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //' | sed -E 's/ compute side.*//'; }
for g in 2x2x2 1x2x4; do
run TILE_GRIDS=$g
run TILE_GRIDS=$g SOBFU_TILED_DEBUG_SKIP=4
run TILE_GRIDS=$g SOBFU_TILED_DEBUG_SKIP=6
run TILE_GRIDS=$g SOBFU_TILED_DEBUG_SKIP=20
run TILE_GRIDS=$g SOBFU_TILED_DEBUG_SKIP=8
run TILE_GRIDS=$g SOBFU_TILED_DEBUG_SKIP=9
for zb in 17 22; do run TILE_GRIDS=$g SOBFU_ZC_B=$zb; done
done
