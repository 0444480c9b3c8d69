#!/bin/bash
# usage (on the GPU box): tools/pmc_tiles_attr.sh <tag> [grid] -- where the tile kernels' fabric READS come from, without touching the code:
# FETCH_SIZE per launch of both kernels under the run-time knobs that remove one source of re-reads each (longer z-chunks: the refill
# planes; no push boxes; no thin shells).  What no knob removes is the xy-halo lines fetched twice and the phi_n gather's overlap.
R="$(cd "$(dirname "$0")/.." && pwd)"; tag=${1:-r03}; grid=${2:-2x2x2}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() {  # name, env...
  local name=$1; shift
  env "$@" TILE_GRIDS=$grid TILE_MODES=direct TILE_THR=1e-10 TILE_ITERS=60 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/ta_$name -o r -- python $R/tools/tile_time_native.py >/dev/null 2>&1
}
run default A=1
run zc16 SOBFU_ZC_A=16 SOBFU_ZC_B=17
run zc32 SOBFU_ZC_A=32 SOBFU_ZC_B=33
run zc129 SOBFU_ZC_A=129 SOBFU_ZC_B=129
run nopush SOBFU_TILED_DEBUG_SKIP=1
run noshells SOBFU_TILED_DEBUG_SKIP=2
python - <<PY
import sqlite3, glob, json, os
out = {}
for d in sorted(glob.glob("$O/ta_*")):
    name = os.path.basename(d)[3:]
    for db in glob.glob(d + "/r_results.db"):
        c = sqlite3.connect(db)
        for kn, avg, n in c.execute("select name, avg(counter_value), count(*) from pmc_events where counter_name = 'FETCH_SIZE' and (name like '%fused_smooth%' or name like '%tile_potential%') group by name"):
            k = "pass_a_tile" if "potential" in kn else "pass_b_tile"
            out.setdefault(name, {})[k + "_read_MB"] = round(2 * avg * 1024 / 1e6, 2)  # FETCH_SIZE counts half the bytes read (profiles/r02_counter_calibration.json)
            out[name][k + "_launches"] = n
json.dump({"note": "fabric reads per launch (MB) of the two tile kernels, $grid tile of 256^3, under run-time knobs (tools/pmc_tiles_attr.sh)", "reads": out}, open("$O/tile_read_attribution.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
rm -rf $O/ta_*
