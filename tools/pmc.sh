#!/bin/bash
# usage: tools/pmc.sh <tag> [bench args...]  -- separate rocprofv3 --pmc passes (never combined with sys/hip traces)
R="$(cd "$(dirname "$0")/.." && pwd)"; tag=$1; shift
cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $R/gpurun_out/pmc_$tag/p$i -o r -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --frames 0 "$@" >/dev/null 2>&1
done
python - <<PY
import sqlite3, glob
rows = {}
for db in sorted(glob.glob("$R/gpurun_out/pmc_$tag/p*/r_results.db")):
    c = sqlite3.connect(db)
    for name, cn, avg, n in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events where name like '%fused_%' group by name, counter_name"):
        k = "passA" if "potential" in name else "passB"
        rows.setdefault(k, {})[cn] = avg
for k, d in sorted(rows.items()):
    print(k, {a: round(b, 1) for a, b in sorted(d.items())})
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        print("   fabric bytes/launch = 2*FETCH + WRITE = %.4g B" % ((2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024))
PY
