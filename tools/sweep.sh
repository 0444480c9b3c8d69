#!/bin/bash
# usage: tools/sweep.sh "<variants>" "<zc list>"  -- prints it/s, pass A/B ms for each tile shape x z-chunk
cd "$(dirname "$0")/.."
for v in $1; do for zc in $2; do
  r=$(SOBFU_HIP_LIB=$PWD/build/variants/libsobfu_hip_$v.so SOBFU_ZC_A=${ZCA:-$zc} SOBFU_ZC_B=$zc python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-traffic --frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%.0f it/s  passA %.1f us  passB %.1f us (%.0f GB/s, %.1f%%)' % (d['value'], r['pass_a']['avg_launch_ms']*1e3, r['avg_launch_ms']*1e3, r['achieved'], 100*r['frac']))")
  echo "$v zc=$zc : $r"
done; done
