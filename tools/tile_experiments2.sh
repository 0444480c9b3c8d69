#!/bin/bash
cd "$(dirname "$0")/.."; grid=${1:-2x2x2}
run() { echo "== $*"; env "$@" TILE_GRIDS=$grid TILE_MODES=direct TILE_THR=-1 TILE_ITERS=300 python tools/tile_time_native.py 2>&1 | grep "us/iter" | sed -E 's/.*local \([0-9, ]+\): //'; }
run X=0
run SOBFU_TILED_DEBUG_SKIP=4
run SOBFU_TILED_DEBUG_SKIP=6
run SOBFU_TILED_DEBUG_SKIP=8
run SOBFU_TILED_DEBUG_SKIP=9
run SOBFU_TILE_A_DIRECT=1
run SOBFU_TILE_A_DIRECT=1 SOBFU_TILED_DEBUG_SKIP=8
run SOBFU_TILE_B_DIRECT=1
run SOBFU_TILE_B_DIRECT=1 SOBFU_TILED_DEBUG_SKIP=4
run SOBFU_TILE_A_DIRECT=1 SOBFU_TILE_B_DIRECT=1
run SOBFU_HIP_LIB=$PWD/build/variants/libsobfu_hip_1x8xSOBFU_NT=0.so
run SOBFU_HIP_LIB=$PWD/build/variants/libsobfu_hip_1x8xSOBFU_NT=1.so
run SOBFU_HIP_LIB=$PWD/build/variants/libsobfu_hip_1x8xSOBFU_NT=0.so SOBFU_TILE_A_DIRECT=1
