#!/bin/bash
# usage (on the GPU box): tools/calibrate_counters.sh [tag]
# FETCH_SIZE / WRITE_SIZE of streaming copies with a known byte count, in the fused passes' own access widths
# (tools/calib/calib_copy.hip, built here by `python tools/build_calib.py`) -> gpurun_out/<tag>/calibration.json
R="$(cd "$(dirname "$0")/.." && pwd)"; tag=${1:-calib}; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
N=$((1<<25))
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/f -o r -- $R/build/calib_copy $N 5 > $O/calib.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/w -o r -- $R/build/calib_copy $N 5 >> $O/calib.log 2>&1
python - <<PY
import sqlite3, glob, json
n = $N
known = {"f4": 16 * n, "x3": 12 * n, "f1": 4 * n}
out = {"elements": n, "note": "factor = known bytes / (counter KiB * 1024); FETCH for the read direction, WRITE for the write direction", "patterns": {}}
for kind, sub in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
    for db in glob.glob("$O/%s/r_results.db" % sub):
        c = sqlite3.connect(db)
        for name, avg, cnt in c.execute("select name, avg(counter_value), count(*) from pmc_events where counter_name = ? group by name", (kind,)):
            short = name.split("(")[0]
            width = "f4" if "f4" in short else ("x3" if "x3" in short else "f1")
            e = out["patterns"].setdefault(short, {"known_bytes_per_direction": known[width]})
            e[kind + "_KiB"] = avg
            moved = known[width] if not ((short.startswith("read") and kind == "WRITE_SIZE") or (short.startswith("write") and kind == "FETCH_SIZE")) else 0
            e[kind + "_factor"] = (moved / (avg * 1024)) if (avg and moved) else None
json.dump(out, open("$O/calibration.json", "w"), indent=1)
for k, v in sorted(out["patterns"].items()):
    print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
PY
rm -rf $O/f $O/w
