#!/bin/bash
cd "$(dirname "$0")/.."
for d in 64 128; do for za in 2 4 8 16; do for zb in 2 4 8 16; do
r=$(SOBFU_ZC_A=$za SOBFU_ZC_B=$zb python bench.py --dim $d --steps 200 --warmup 100 --repeats 5 --no-cpu-baseline --no-traffic --frames 0 2>/dev/null | grep metric | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.0f it/s  %.1f us/iter  A %.1f B %.1f' % (d['value'], 1e3*d['ms_per_step'], 1e3*r['pass_a']['avg_launch_ms'], 1e3*r['avg_launch_ms']))")
echo "dim=$d zcA=$za zcB=$zb : $r"; done; done; done
