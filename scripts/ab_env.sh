#!/bin/bash
# A/B of one environment switch on ONE box, alternating runs: scripts/ab_env.sh IRMV_WRES_STAGGER 0 1 [rounds=2] [bench args...]
# prints value (FPS, HBM-resident clock) per run; boxes of the pool differ by +-2 %, so comparisons are same-box only.
var=$1; a=$2; b=$3; rounds=${4:-2}; shift 4 2>/dev/null
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
for r in $(seq 1 $rounds); do
  for v in $a $b; do
    out=$(env $var=$v IRMV_BENCH_SKIP=latency,h2d,config4 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>/dev/null | grep '^{' | tail -1)
    echo "$var=$v round $r: $(echo "$out" | python3 -c 'import sys, json; d = json.loads(sys.stdin.read()); print(d["value"], "FPS;", d["ms_per_step"], "ms/step; eager kernel sum", d["roofline"]["step_kernel_ms_eager"], "ms")')"
  done
done
