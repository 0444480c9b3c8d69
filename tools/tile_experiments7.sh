#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" TILE_MODES=direct TILE_THR=-1 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //' | sed -E 's/ compute side.*//'; }
for g in 2x2x2; do
run TILE_GRIDS=$g
run TILE_GRIDS=$g SOBFU_TILED_DEBUG_SKIP=8
for v in 1x4 1x16; do
V=$PWD/build/variants/libsobfu_hip_$v.so
run TILE_GRIDS=$g SOBFU_HIP_LIB=$V
run TILE_GRIDS=$g SOBFU_HIP_LIB=$V SOBFU_TILED_DEBUG_SKIP=6
run TILE_GRIDS=$g SOBFU_HIP_LIB=$V SOBFU_TILED_DEBUG_SKIP=9
for zb in 8 17 24; do run TILE_GRIDS=$g SOBFU_HIP_LIB=$V SOBFU_ZC_B=$zb SOBFU_TILED_DEBUG_SKIP=6; done
for za in 8 16; do run TILE_GRIDS=$g SOBFU_HIP_LIB=$V SOBFU_ZC_A=$za SOBFU_TILED_DEBUG_SKIP=9; done
done
done
