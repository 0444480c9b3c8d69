#!/bin/bash
# usage: tools/ab2.sh "<variant[@ENV=..,ENV=..]> ..." [rounds]   (like ab.sh; several env assignments separated by commas)
cd "$(dirname "$0")/.."
for round in $(seq 1 ${2:-3}); do for v in $1; do
  lib=${v%%@*}; envs=""; [[ "$v" == *@* ]] && envs="${v#*@}"; envs=${envs//,/ }
  r=$(env $envs SOBFU_HIP_LIB=$PWD/build/variants/libsobfu_hip_$lib.so python bench.py --steps 50 --warmup 20 --repeats 5 --no-cpu-baseline --no-traffic --frames 0 2>/dev/null | grep metric | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.0f it/s  passA %.1f us  passB %.1f us' % (d['value'], r['pass_a']['avg_launch_ms']*1e3, r['avg_launch_ms']*1e3))")
  echo "round $round $v : $r"
done; done
