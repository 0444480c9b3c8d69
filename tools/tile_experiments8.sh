#!/bin/bash
cd "$(dirname "$0")/.."
run() { echo "== $*"; env "$@" TILE_GRIDS=2x2x2 TILE_MODES=direct TILE_THR=-1 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //' | sed -E 's/ compute side.*//'; }
# pass B alone, no shells (SKIP=6): LDS per workgroup 32 KB static + pad
run SOBFU_TILED_DEBUG_SKIP=6
run SOBFU_TILED_DEBUG_SKIP=6 SOBFU_LDS_PAD_B=20000
run SOBFU_TILED_DEBUG_SKIP=6 SOBFU_LDS_PAD_B=50000
for z in 17 9 12; do
run SOBFU_TILED_DEBUG_SKIP=6 SOBFU_ZC_B=$z
run SOBFU_TILED_DEBUG_SKIP=6 SOBFU_ZC_B=$z SOBFU_LDS_PAD_B=20000
run SOBFU_TILED_DEBUG_SKIP=6 SOBFU_ZC_B=$z SOBFU_LDS_PAD_B=50000
done
run SOBFU_TILED_DEBUG_SKIP=6 SOBFU_ZC_B=17 SOBFU_LDS_PAD_B=50000 SOBFU_PIPE_B=0
run SOBFU_TILED_DEBUG_SKIP=6 SOBFU_ZC_B=33 SOBFU_LDS_PAD_B=50000
