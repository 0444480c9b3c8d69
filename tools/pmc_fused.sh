#!/bin/bash
R="$(cd "$(dirname "$0")/.." && pwd)"; O=$R/gpurun_out/pmc_fused; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  SOBFU_FUSED=1 rocprofv3 --pmc $set --kernel-trace -d $O/p$i -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline >/dev/null 2>&1
done
python - <<PY
import sqlite3, glob
rows = {}
for db in sorted(glob.glob("$O/p*/r_results.db")):
    c = sqlite3.connect(db)
    for name, cn, avg, n in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events where name like '%fused_iteration%' group by name, counter_name"):
        rows[cn] = avg
print({a: round(b, 1) for a, b in sorted(rows.items())})
if "FETCH_SIZE" in rows: print("fabric bytes/launch = %.4g" % ((2 * rows["FETCH_SIZE"] + rows["WRITE_SIZE"]) * 1024))
PY
rm -rf $O
