#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" TILE_GRIDS=2x2x2 TILE_MODES=direct TILE_THR=-1 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //' | sed -E 's/ compute side.*//'; }
run SOBFU_TILED_DEBUG_SKIP=4
run SOBFU_TILED_DEBUG_SKIP=6
run SOBFU_TILED_DEBUG_SKIP=20
run SOBFU_TILED_DEBUG_SKIP=52
run SOBFU_TILED_DEBUG_SKIP=84
run SOBFU_TILED_DEBUG_SKIP=36
run SOBFU_TILED_DEBUG_SKIP=68
run SOBFU_TILED_DEBUG_SKIP=20 SOBFU_PIPE_B=0
run SOBFU_TILED_DEBUG_SKIP=4 SOBFU_PIPE_B=0
