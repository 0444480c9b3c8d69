#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" TILE_MODES=direct TILE_THR=-1 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //' | sed -E 's/ compute side.*//'; }
g=2x2x2
for zb in 17 33 11 22; do run TILE_GRIDS=$g SOBFU_ZC_B=$zb; done
for zb in 17 33; do run TILE_GRIDS=$g SOBFU_ZC_B=$zb SOBFU_PIPE_B=0; done
for za in 16 32 11 22; do run TILE_GRIDS=$g SOBFU_ZC_B=17 SOBFU_ZC_A=$za; done
run TILE_GRIDS=$g SOBFU_ZC_B=17 SOBFU_TILED_DEBUG_SKIP=4
run TILE_GRIDS=$g SOBFU_ZC_B=17 SOBFU_TILED_DEBUG_SKIP=6
run TILE_GRIDS=$g SOBFU_ZC_A=16 SOBFU_TILED_DEBUG_SKIP=8
run TILE_GRIDS=$g SOBFU_ZC_A=16 SOBFU_TILED_DEBUG_SKIP=9
g=1x2x4
for zb in 9 17 33; do run TILE_GRIDS=$g SOBFU_ZC_B=$zb; done
for za in 8 16 32; do run TILE_GRIDS=$g SOBFU_ZC_B=17 SOBFU_ZC_A=$za; done
