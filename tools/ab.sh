#!/bin/bash
# usage: tools/ab.sh "<variants>" [rounds]  -- interleaved A/B rounds of bench.py (100 steps, 30 warm-up) per variant
cd "$(dirname "$0")/.."
for round in $(seq 1 ${2:-3}); do for v in $1; do
  lib=${v%%@*}; envs=""; [[ "$v" == *@* ]] && envs="${v#*@}"
  r=$(env $envs SOBFU_HIP_LIB=$PWD/build/variants/libsobfu_hip_$lib.so python bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.0f it/s  passA %.1f us  passB %.1f us (%.1f%%)' % (d['value'], r['pass_a_avg_launch_ms']*1e3, r['avg_launch_ms']*1e3, 100*r['frac']))")
  echo "round $round $v : $r"
done; done
