#!/bin/bash
# usage (on the GPU box): tools/profile_tiles.sh <tag> [grid]  -- rocprofv3 kernel stats of one rank's iteration of the tile loop (compute side)
R="$(cd "$(dirname "$0")/.." && pwd)"; tag=${1:-r03}; grid=${2:-2x2x2}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for mode in direct packed; do
  TILE_GRIDS=$grid TILE_MODES=$mode TILE_THR=1e-10 TILE_ITERS=200 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$mode -o r -- python $R/tools/tile_time_native.py > $O/tile_${grid}_$mode.log 2>&1
  python $R/tools/rocprof_summary.py $O/kt_$mode/r_results.db > $O/tile_${grid}_${mode}_kernel_stats.md
  rm -rf $O/kt_$mode
done
cat $O/tile_${grid}_*_kernel_stats.md
