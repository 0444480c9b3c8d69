#!/bin/bash
# usage (on the GPU box): tools/profile_round.sh r01   -- bench line + rocprofv3 kernel stats + PMC passes -> gpurun_out/<tag>/
R="$(cd "$(dirname "$0")/.." && pwd)"; tag=${1:-r01}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o r -- python $R/bench.py --no-cpu-baseline --no-traffic > $O/bench_under_rocprof.json 2>/dev/null
python $R/tools/rocprof_summary.py $O/kt/r_results.db > $O/rocprof_kernel_stats.md
i=0
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d $O/pmc$i -o r -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic >/dev/null 2>&1
done
python - <<PY
import sqlite3, glob, json
rows = {}
for db in sorted(glob.glob("$O/pmc*/r_results.db")):
    c = sqlite3.connect(db)
    for name, cn, avg, n in c.execute("select name, counter_name, avg(counter_value), count(*) from pmc_events where name like '%fused_%' group by name, counter_name"):
        k = "pass_a" if "potential" in name else "pass_b"
        rows.setdefault(k, {})[cn] = avg
import hashlib, os
h = hashlib.sha256()  # solver_kernels.hip and the parts it includes (bench.kernel_source_sha256)
for name in ["solver_kernels.hip"] + sorted(f for f in os.listdir("$R/sobfu_amd/csrc") if f.startswith("solver_") and f.endswith(".inl")):
    h.update(open("$R/sobfu_amd/csrc/" + name, "rb").read())
out = {"kernel_source_sha256": h.hexdigest(),
       "note": "rocprofv3 --pmc, one counter set per run, averages per launch at 256^3; FETCH_SIZE/WRITE_SIZE in KiB. "
               "Correction factors MEASURED on this part with streaming copies of known size in the kernels' own access widths "
               "(12-byte dwordx3, 4-byte dword, 16-byte; plain and nontemporal -- tools/calibrate_counters.sh, "
               "profiles/r02_counter_calibration.json): FETCH_SIZE reports exactly 1/2 of the bytes read, WRITE_SIZE the bytes "
               "written -> bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.  These are L2 <-> fabric bytes (Infinity-Cache hits included).",
       "counters": rows}
for k in rows:
    if "FETCH_SIZE" in rows[k] and "WRITE_SIZE" in rows[k]:
        out[k + "_hbm_bytes_per_launch"] = (2 * rows[k]["FETCH_SIZE"] + rows[k]["WRITE_SIZE"]) * 1024
json.dump(out, open("$O/pmc.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k.endswith("per_launch")}))
PY
rm -rf $O/kt $O/pmc[0-9]   # keep the summaries, drop the raw databases
# the bench line LAST, on the same box, with this run's counters in place (so that its traffic_from_profiles is this run's)
cp $O/pmc.json $R/profiles/pmc_latest.json
cd $R
python bench.py 2>$O/bench.err > $O/bench.json
cd /tmp
# one rank's iteration of the 2x2x2 tile loop, per kernel (compute side)
timeout 600 $R/tools/profile_tiles.sh $tag 2x2x2 > $O/tiles.log 2>&1
head -c 600 $O/bench.json; echo; cat $O/rocprof_kernel_stats.md | head -6
