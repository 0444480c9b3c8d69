#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" TILE_MODES=packed TILE_THR=1e-10 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //' | sed -E 's/ compute side.*//'; }
for g in 2x2x2 1x2x4; do
run TILE_GRIDS=$g SOBFU_TILED_SERIAL=1
run TILE_GRIDS=$g SOBFU_TILED_SERIAL=0
run TILE_GRIDS=$g SOBFU_TILED_SERIAL=0 TILE_THR=-1
done
