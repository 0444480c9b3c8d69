#!/bin/bash
# usage (GPU box): tools/stress_probe.sh [runs] -- repeated multi-process `bench.py --gpus 8` (ranks sharing the GPU) on the direct transport; prints every fallback reason
export SOBFU_BENCH_SHARE_GPU=1 SOBFU_TILED_DIAG=0
for i in $(seq 1 ${1:-10}); do
  timeout 300 python bench.py --gpus 8 --steps 6 --warmup 2 --dim 64 --repeats 2 >/tmp/stress_$i.out 2>/tmp/stress_$i.err; rc=$?
  tail -1 /tmp/stress_$i.out | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('run $i rc=$rc', d.get('transport'), d.get('tiled_parity_vs_single_gpu'), (d.get('transport_fallback') or '')[:600])
except Exception as e:
    print('run $i rc=$rc NO JSON LINE', e)"
  if ! grep -q '"transport": "direct"' /tmp/stress_$i.out; then mkdir -p gpurun_out/stress; cp /tmp/stress_$i.err gpurun_out/stress/run_$i.err; cp /tmp/stress_$i.out gpurun_out/stress/run_$i.out; fi
  grep -h "refused a fresh\|trying once more" /tmp/stress_$i.err | cut -c1-700 | head -4
done
