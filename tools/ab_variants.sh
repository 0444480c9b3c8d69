#!/bin/bash
# usage (on the GPU box): tools/ab_variants.sh lib1.so lib2.so ...   -- interleaved A/B of library variants against the tree's library: 3 rounds of a short bench each
cd $GRAFT_REPO_ROOT
one() { env ${1:+SOBFU_HIP_LIB=$1} python bench.py --steps 50 --warmup 10 --repeats 5 --profile-repeats 1 --no-cpu-baseline --no-traffic --frames 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%-46s it/s %7.1f  passB %6.1f us  passA %6.1f us' % (sys.argv[1][-46:], d['value'], 1e3*r['avg_launch_ms'], 1e3*r['pass_a']['avg_launch_ms']))" "${1:-tree}"; }
for round in 1 2 3; do one ""; for lib in "$@"; do one "$lib"; done; done
