#!/bin/bash
# usage (on the GPU box): tools/pmc_tile_stalls.sh <tag> [grid] -- where the wave-cycles of the tile loop's two kernels go: issue / wait / LDS counters
# (two rocprofv3 --pmc passes of 8 SQ counters each; counters only, no other trace domain)
R="$(cd "$(dirname "$0")/.." && pwd)"; tag=${1:-r04}; grid=${2:-2x2x2}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VALU"; do
  i=$((i+1))
  TILE_GRIDS=$grid TILE_MODES=direct TILE_THR=1e-10 TILE_ITERS=100 timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/ts$i -o r -- python $R/tools/tile_time_native.py > $O/ts$i.log 2>&1
done
python - <<PY
import sqlite3, glob, json
rows = {}
for db in sorted(glob.glob("$O/ts*/r_results.db")):
    c = sqlite3.connect(db)
    for name, cn, avg, n in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events where name like '%fused_smooth%' or name like '%tile_potential%' group by name, counter_name"):
        k = "pass_a_tile" if "potential" in name else "pass_b_tile"
        rows.setdefault(k, {})[cn] = avg
for k, r in rows.items():
    w = r.get("SQ_WAVE_CYCLES")
    if w:
        r["share_of_wave_cycles"] = {n: round(r[c] / w, 3) for n, c in (("parked (s_waitcnt / barrier)", "SQ_WAIT_ANY"), ("issue stall", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY"),
                                                                         ("issuing VALU", "SQ_ACTIVE_INST_VALU"), ("issuing LDS", "SQ_ACTIVE_INST_LDS"), ("stalled on LDS issue", "SQ_WAIT_INST_LDS")) if c in r}
json.dump({"note": "rocprofv3 --pmc over tools/tile_time_native.py ($grid, direct transport's launches, one rank, no peers); averages per launch as rocprofv3 reports them (per shader engine; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles: MI355X_MICROARCH.md); WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES", "counters": rows}, open("$O/tile_stalls.json", "w"), indent=1)
print(json.dumps(rows, indent=1))
PY
rm -rf $O/ts[0-9]
