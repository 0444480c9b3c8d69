#!/bin/bash
# One parameterised sweep over run-time knobs of the tile loop (replaces the round-3 tile_experiments*.sh one-offs).
# usage (on the GPU box):  tools/tile_sweep.sh [-g 2x2x2,1x2x4] [-k] [-i ITERS] name:VAR=v,VAR=v ...
#   every config is one process of tools/tile_time_native.py (direct transport's launches, one rank, no peers) per grid;
#   -k adds a rocprofv3 --kernel-trace run per config and prints the average launch time of pass A / pass B;
#   -p adds two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; counters only) and prints the fabric bytes per launch.
R="$(cd "$(dirname "$0")/.." && pwd)"; grids=2x2x2; ks=0; pm=0; iters=200
while getopts "g:kpi:" o; do case $o in g) grids=$OPTARG;; k) ks=1;; p) pm=1;; i) iters=$OPTARG;; esac; done; shift $((OPTIND-1))
cd /tmp; export TMPDIR=/tmp
for cfg in "$@"; do
  name=${cfg%%:*}; kv=${cfg#*:}; [ "$kv" = "$cfg" ] && kv="SOBFU_NOP=1"
  for g in ${grids//,/ }; do
    line=$(env ${kv//,/ } TILE_GRIDS=$g TILE_MODES=${TILE_MODES:-direct} TILE_THR=1e-10 TILE_ITERS=$iters timeout 300 python $R/tools/tile_time_native.py 2>&1 | grep "us/iteration" | sed 's/.*: \([0-9.]*\) us\/iteration.*/\1/' | tr '\n' ' ')
    out="$name $g: ${line}us/iteration"
    if [ $ks = 1 ]; then
      O=/tmp/ts_$$; rm -rf $O
      env ${kv//,/ } TILE_GRIDS=$g TILE_MODES=${TILE_MODES:-direct} TILE_THR=1e-10 TILE_ITERS=100 timeout 300 rocprofv3 --kernel-trace -d $O -o r -- python $R/tools/tile_time_native.py >/dev/null 2>&1
      out="$out $(python - <<PY
import sqlite3, glob
for db in glob.glob("$O/r_results.db"):
    c = sqlite3.connect(db)
    for kn, avg, n in c.execute("select name, average, total_calls from top_kernels where name like '%fused_smooth%' or name like '%potential_gradient%' order by name"):
        print(f" {'A' if 'potential' in kn else 'B'} {avg:.2f} us x{n}", end="")
PY
)"
      rm -rf $O
    fi
    if [ $pm = 1 ]; then
      for cn in FETCH_SIZE WRITE_SIZE; do
        O=/tmp/tp_$$; rm -rf $O
        env ${kv//,/ } TILE_GRIDS=$g TILE_MODES=${TILE_MODES:-direct} TILE_THR=1e-10 TILE_ITERS=60 timeout 300 rocprofv3 --pmc $cn --kernel-trace -d $O -o r -- python $R/tools/tile_time_native.py >/dev/null 2>&1
        out="$out $(python - <<PY
import sqlite3, glob
for db in glob.glob("$O/r_results.db"):
    c = sqlite3.connect(db)
    for kn, avg in c.execute("select name, avg(counter_value) from pmc_events where counter_name = '$cn' and (name like '%fused_smooth%' or name like '%potential_gradient%') group by name order by name"):
        f = 2 if "$cn" == "FETCH_SIZE" else 1
        print(f" {'A' if 'potential' in kn else 'B'} {'rd' if f == 2 else 'wr'} {f*avg*1024/1e6:.1f} MB", end="")
PY
)"
        rm -rf $O
      done
    fi
    echo "$out"
  done
done
