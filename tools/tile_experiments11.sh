#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" TILE_MODES=direct TILE_THR=1e-10 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //' | sed -E 's/ compute side.*//'; }
for i in 1 2; do
for g in 2x2x2; do
run TILE_GRIDS=$g SOBFU_TILE_XPAD=0
run TILE_GRIDS=$g SOBFU_TILE_XPAD=1
run TILE_GRIDS=$g SOBFU_TILE_XPAD=0 SOBFU_TILED_DEBUG_SKIP=8
run TILE_GRIDS=$g SOBFU_TILE_XPAD=1 SOBFU_TILED_DEBUG_SKIP=8
run TILE_GRIDS=$g SOBFU_TILE_XPAD=0 SOBFU_TILED_DEBUG_SKIP=4 TILE_THR=-1
run TILE_GRIDS=$g SOBFU_TILE_XPAD=1 SOBFU_TILED_DEBUG_SKIP=4 TILE_THR=-1
done
done
