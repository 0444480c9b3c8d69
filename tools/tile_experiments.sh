#!/bin/bash
# timing experiments on the 2x2x2 tile (compute side): pieces left out / z-chunk sweeps.  usage: tools/tile_experiments.sh [grid]
cd "$(dirname "$0")/.."; grid=${1:-2x2x2}
run() { echo "== $*"; env "$@" TILE_GRIDS=$grid TILE_MODES=direct TILE_THR=1e-10 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //'; }
run X=0
run SOBFU_TILED_DEBUG_SKIP=1
run SOBFU_TILED_DEBUG_SKIP=2
run SOBFU_TILED_DEBUG_SKIP=3
run SOBFU_TILED_DEBUG_SKIP=4
run SOBFU_TILED_DEBUG_SKIP=8
run SOBFU_TILED_DEBUG_SKIP=12
for z in 4 6 8 16 24 32; do run SOBFU_ZC_B=$z; done
for z in 2 3 6 8 12 16; do run SOBFU_ZC_A=$z; done
