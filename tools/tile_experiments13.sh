#!/bin/bash
# usage (on the GPU box): tools/tile_experiments13.sh [grid] -- what the thin shells of pass B and the push boxes of pass A cost a tile, in time
# (kernel averages, rocprofv3 --kernel-trace) and in fabric reads (FETCH_SIZE), one run-time knob at a time
R="$(cd "$(dirname "$0")/.." && pwd)"; grid=${1:-2x2x2}; O=$R/gpurun_out/exp13; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cfg in "default:A=1" "noshells:SOBFU_TILED_DEBUG_SKIP=2" "no_y_shells:SOBFU_TILED_DEBUG_SKIP=32" "no_x_shells:SOBFU_TILED_DEBUG_SKIP=64" "nopush:SOBFU_TILED_DEBUG_SKIP=1" $EXTRA_CFGS; do
  name=${cfg%%:*}; kv=${cfg#*:}
  env ${kv//,/ } TILE_GRIDS=$grid TILE_MODES=direct TILE_THR=1e-10 TILE_ITERS=100 timeout 300 rocprofv3 --kernel-trace -d $O/kt_$name -o r -- python $R/tools/tile_time_native.py 2>/dev/null | grep "us/iteration" | sed "s/^/$name: /"
  env ${kv//,/ } TILE_GRIDS=$grid TILE_MODES=direct TILE_THR=1e-10 TILE_ITERS=60 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pm_$name -o r -- python $R/tools/tile_time_native.py >/dev/null 2>&1
done
python - <<PY
import sqlite3, glob, os
for d in sorted(glob.glob("$O/kt_*")):
    name = os.path.basename(d)[3:]
    line = name + ":"
    for db in glob.glob(d + "/r_results.db"):
        c = sqlite3.connect(db)
        for kn, avg, n in c.execute("select name, average, total_calls from top_kernels where name like '%fused_smooth%' or name like '%tile_potential%'"):
            line += f"  {'A' if 'potential' in kn else 'B'} {avg/1000:.2f} us x{n}"
    for db in glob.glob("$O/pm_" + name + "/r_results.db"):
        c = sqlite3.connect(db)
        for kn, avg in c.execute("select name, avg(counter_value) from pmc_events where counter_name = 'FETCH_SIZE' and (name like '%fused_smooth%' or name like '%tile_potential%') group by name"):
            line += f"  {'A' if 'potential' in kn else 'B'} reads {2*avg*1024/1e6:.1f} MB"
    print(line)
PY
rm -rf $O
