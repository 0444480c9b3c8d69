#!/bin/bash
# usage (on the GPU box): tools/pmc_tiles.sh <tag> [grid] -- issue / busy counters of the tile loop's two kernels (compute side, direct transport's launches)
R="$(cd "$(dirname "$0")/.." && pwd)"; tag=${1:-r03}; grid=${2:-2x2x2}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  TILE_GRIDS=$grid TILE_MODES=direct TILE_THR=1e-10 TILE_ITERS=100 timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/tp$i -o r -- python $R/tools/tile_time_native.py >/dev/null 2>&1
done
python - <<PY
import sqlite3, glob, json
rows = {}
for db in sorted(glob.glob("$O/tp*/r_results.db")):
    c = sqlite3.connect(db)
    for name, cn, avg, n in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events where name like '%fused_smooth%' or name like '%tile_potential%' group by name, counter_name"):
        k = "pass_a_tile" if "potential" in name else "pass_b_tile"
        rows.setdefault(k, {})[cn] = avg
json.dump({"note": "rocprofv3 --pmc over tools/tile_time_native.py ($grid, direct transport's launches, one rank, no peers); averages per launch; SQ_* as rocprofv3 reports them (per shader engine: SQ_WAVES x 32 = the launch's waves); FETCH_SIZE / WRITE_SIZE in KiB", "counters": rows}, open("$O/tile_pmc.json", "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
rm -rf $O/tp[0-9]
